"""Timeline of two consecutive data-parallel steps out of a rocprofv3 kernel trace of tools/bench_dp_structure.py (one-rank RCCL
group): start (us, relative), duration and name of every launch -- the gaps are what the data-parallel path costs besides its two
extra kernels.   python tools/dp_timeline.py gpurun_out/prof_<tag>"""
import csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "oneRankReduce" in r["Kernel_Name"] or "ncclDevKernel" in r["Kernel_Name"]]
i = idx[len(idx) // 2]
lo = i - 6
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = None
for r in rows[lo: lo + 16]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:6.2f}"
    print(f"{(s - t0) / 1e3:9.2f} us  {(e - s) / 1e3:7.2f} us  {name:48s} {gap}")
    prev_end = e
