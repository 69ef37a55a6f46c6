# hardware counters of one dense-layer kernel at the C2 shape (tools/run_fwd_only.py <mode>): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mode=${1:-fused}; pat=${2:-chain_kernel}
out=gpurun_out/pmc_$mode
mkdir -p $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python3 tools/run_fwd_only.py $mode > $out/$tag.log 2>&1
done
python3 - $out $pat <<'PY'
import csv, glob, collections, sys
out, pat = sys.argv[1], sys.argv[2]
vals = {}
for f in sorted(glob.glob(out + "/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        v = v[2:]   # (the first launches include warm-up effects)
        vals[k] = sum(v) / len(v)
w = vals.get("SQ_WAVES", 1)
for k in sorted(vals):
    print(f"{k:30s} per launch {vals[k]:14.1f}   per wave {vals[k] / w:10.1f}")
PY
