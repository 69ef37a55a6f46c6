# hardware counters of the row-block chain kernel (forward, C2 shape): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_chain
rocprofv3 -L > gpurun_out/pmc_chain/counters.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS" \
           "SQ_INSTS_WAVE32_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_chain/$tag -- python3 tools/run_fwd_only.py > gpurun_out/pmc_chain/$tag.log 2>&1
  tail -c 200 gpurun_out/pmc_chain/$tag.log
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_chain/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "chain_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} mean per launch {sum(v)/len(v):14.1f}  ({len(v)} launches)")
PY
