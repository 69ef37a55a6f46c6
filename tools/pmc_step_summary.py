"""Per-kernel medians of the counters collected over the training step (one rocprofv3 --pmc pass per counter, bench.py as the
workload):   python tools/pmc_step_summary.py <dir with one sub-directory per pass> > profiles/r01_step_counters.json
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half of a wide coalesced read stream), WRITE_SIZE is
taken as it is; both are in KB."""
import collections, csv, glob, json, statistics, sys

root = sys.argv[1]
names = {"sk_tile_f16": "sk_tile_f16_kernel (all-pairs)", "chain_kernel<2>": "chain_kernel<FUSED> (forward + cost + decoder backward)",
         "chain_kernel<1>": "chain_kernel<BWD> (encoder backward)", "mlp_wgrad_kernel": "mlp_wgrad_kernel (+ scalar epilogue)",
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
    out[nm] = o
json.dump(out, sys.stdout, indent=1)
print()
