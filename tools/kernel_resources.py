#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one HIP source (compiler remarks; no GPU needed).
usage: tools/kernel_resources.py encodermap_amd/csrc/mlp.hip [substring of the kernel name] [extra hipcc flags]
(headers are taken from the directory of the source first: a copy of csrc/ with an older or patched source compares builds)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = sys.argv[3:] if flt else sys.argv[2:]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(os.path.abspath(src)), "-I" + os.path.join(ROOT, "encodermap_amd", "csrc"), *extra,
           "--cuda-device-only", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    rows, cur = [], None
    for line in out.stderr.splitlines():
        m = re.search(r":\d+:\d+: remark: +(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: +(.*?) \[-Rpass", line)
        if not m:
            if "error" in line or "warning" in line:
                print(line)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        g = r.get
        if flt not in name:
            continue
        print(f"{name[:64]:64s} vgpr {g('VGPRs', '?'):>4} agpr {g('AGPRs', '?'):>3} sgpr {g('TotalSGPRs', '?'):>4} "
              f"spill s/v {g('SGPRs Spill', '?')}/{g('VGPRs Spill', '?')} scratch {g('ScratchSize [bytes/lane]', '?')} "
              f"occ {g('Occupancy [waves/SIMD]', '?')} lds {g('LDS Size [bytes/block]', '?')}")
    return out.returncode


if __name__ == "__main__":
    sys.exit(main())
