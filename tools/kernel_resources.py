"""Registers, scratch, occupancy and LDS of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
usage: python tools/kernel_resources.py encodermap_amd/csrc/sketchmap.hip [substring of the kernel name]      (build box)"""
import os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-mllvm", "-amdgpu-kernarg-preload-count=16",
       "-I" + os.path.join(root, "include"), "-I" + os.path.dirname(os.path.abspath(src)), "-I" + os.path.join(root, "encodermap_amd", "csrc"), "--cuda-device-only", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = m.group(2)
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(1).split(" ")[0]] = m.group(2)
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    if flt in dem:
        g = lambda k: str(r.get(k, "-"))
        print(f"{dem:70s} vgpr {g('VGPRs'):>4} agpr {g('AGPRs'):>3} sgpr {g('TotalSGPRs'):>4} scratch {g('ScratchSize'):>5} occ {g('Occupancy')} lds {g('LDS')}")
