"""Per-kernel medians of the counters collected over the training step (one rocprofv3 --pmc pass per counter, bench.py as the
workload):   python tools/pmc_step_summary.py <dir with one sub-directory per pass> > profiles/r01_step_counters.json
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half of a wide coalesced read stream), WRITE_SIZE is
taken as it is; both are in KB."""
import collections, csv, glob, json, statistics, sys

root = sys.argv[1]
names = {"sk_tile_f16": "sk_tile_f16_kernel (all-pairs)", "chain_kernel<2,": "chain_kernel<FUSED> (forward + cost + decoder backward)",
         "chain_kernel<1,": "chain_kernel<BWD> (encoder backward)", "mlp_wgrad_kernel": "mlp_wgrad_kernel (+ scalar epilogue)",
         "adam_kernel": "adam_kernel (+ next batch)"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for pat, nm in names.items():
            if pat in r["Kernel_Name"]:
                vals[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
                vals[nm]["_dur_us_under_pmc"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for nm, d in vals.items():
    o = {k: round(statistics.median(v), 2) for k, v in d.items()}
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_bytes_per_launch"] = int(2 * o["FETCH_SIZE"] * 1024 + o["WRITE_SIZE"] * 1024)
    if "SQ_INSTS_VALU" in o and "SQ_WAVES" in o:
        o["valu_per_wave"] = round(o["SQ_INSTS_VALU"] / o["SQ_WAVES"], 1)
        o["salu_per_wave"] = round(o.get("SQ_INSTS_SALU", 0) / o["SQ_WAVES"], 1)
    out[nm] = o
doc = {"workload": "bench.py (C2: B=4096, D=128), one training step = these five launches",
       "command": "bash tools/run_pmc_step.sh: rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 40 "
                  "--warmup 8 --no-cpu-baseline (one pass per counter set; medians over the launches)",
       "corrections": "FETCH_SIZE (KB) doubled, WRITE_SIZE (KB) as reported (MI355X_MICROARCH.md, HBM section); durations are those under "
                      "the counter passes, not the kernel-trace averages",
       "kernels": out}
json.dump(doc, sys.stdout, indent=1)
print()
