"""Per-kernel medians of the counters collected over a training step (one rocprofv3 --pmc pass per counter set, bench.py as the
workload):   python tools/pmc_step_summary.py <dir with one sub-directory per pass> [config] > profiles/rNN_step_counters.json
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half of a wide coalesced read stream), WRITE_SIZE is
taken as it is; both are in KB."""
import collections, csv, glob, json, os, statistics, subprocess, sys

root = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


LABEL = {"chain_kernel_small<2, 2>": "chain_kernel<FUSED> (forward + cost + decoder backward; SMALL build)",
         "chain_kernel_small<1, 2>": "chain_kernel<BWD> (encoder backward; SMALL build)", "chain_kernel_small<0, 2>": "chain_kernel<FWD> (SMALL build)",
         "chain_kernel<2, 2>": "chain_kernel<FUSED> (forward + cost + decoder backward)", "chain_kernel_c128<2, 2>": "chain_kernel<FUSED> (128-register build)",
         "chain_kernel<1, 2>": "chain_kernel<BWD> (encoder backward)", "chain_kernel_c128<1, 2>": "chain_kernel<BWD> (128-register build)",
         "chain_kernel<0, 2>": "chain_kernel<FWD>", "chain_kernel_c128<0, 2>": "chain_kernel<FWD> (128-register build)",
         "mlp_wgrad_kernel": "mlp_wgrad_kernel (+ scalar epilogue)", "adam_kernel<false>": "adam_kernel (+ next batch)", "adam_kernel<true>": "adam_kernel (+ next ADC head)",
         "mlp_wgrad_local_kernel": "mlp_wgrad_local_kernel (weight gradients + optimizer + scalar epilogue + next batch)",
         "mlp_wgrad_adam_kernel": "mlp_wgrad_adam_kernel (weight gradients, optimizer as their tail)"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith(("at::", "__amd_rocclr")) or "mirror" in k or "gather_rows_kernel" in k or "permute" in k or "l2_partials" in k:
            continue
        nm = LABEL.get(k, k)
        if k.startswith("sk_tile"):
            nm = k + " (all-pairs)"
        if k.startswith("chain_tiles_kernel"):
            nm = k + " = chain_kernel<FUSED> with the all-pairs tiles beside its decoder half"
        vals[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
        vals[nm]["_dur_us_under_pmc"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for nm, d in vals.items():
    o = {k: round(statistics.median(v), 2) for k, v in d.items()}
    o["_launches_seen"] = len(d["_dur_us_under_pmc"])
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_bytes_per_launch"] = int(2 * o["FETCH_SIZE"] * 1024 + o["WRITE_SIZE"] * 1024)
    if "SQ_INSTS_VALU" in o and "SQ_WAVES" in o and o["SQ_WAVES"]:
        o["valu_per_wave"] = round(o["SQ_INSTS_VALU"] / o["SQ_WAVES"], 1)
        o["salu_per_wave"] = round(o.get("SQ_INSTS_SALU", 0) / o["SQ_WAVES"], 1)
    if "TCC_HIT_sum" in o and "TCC_MISS_sum" in o and (o["TCC_HIT_sum"] + o["TCC_MISS_sum"]) > 0:
        o["l2_hit_rate"] = round(o["TCC_HIT_sum"] / (o["TCC_HIT_sum"] + o["TCC_MISS_sum"]), 4)
    out[nm] = o
try:
    commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip() or None
except Exception:
    commit = None
if not commit:      # (the GPU box has no .git: the stamp __graft_entry__.build() left beside the library)
    try:
        commit = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_commit")).read().strip() or None
    except Exception:
        commit = None
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import source_hash
    shash = source_hash()
except Exception:
    shash = None
doc = {"workload": f"bench.py --config {cfg}: one training step = the launches below", "commit": os.environ.get("EMX_COMMIT") or commit,
       "source_hash": shash,
       "command": f"bash tools/run_pmc_step.sh <tag> {cfg}: rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --config {cfg} "
                  "--steps 40 --warmup 8 --no-cpu-baseline --no-facade (one pass per counter set; medians over the launches)",
       "corrections": "FETCH_SIZE (KB) doubled, WRITE_SIZE (KB) as reported (MI355X_MICROARCH.md, HBM section); durations are those under "
                      "the counter passes, not the kernel-trace averages; l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)",
       "kernels": out}
json.dump(doc, sys.stdout, indent=1)
print()
