# where a configuration's kernels spend their wave-cycles: issue / wait / LDS / co-execution counters
#   bash tools/run_pmc_stalls.sh <tag> [config=C5] [steps=12]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-stalls}; cfg=${2:-C5}; steps=${3:-12}
out=gpurun_out/pmc_stalls_$tag
mkdir -p $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "LdsUtil" "LDSBankConflict" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" "SQ_LEVEL_WAVES SQ_CYCLES"; do
  c=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$c -- python3 bench.py --config $cfg --steps $steps --warmup 4 --no-cpu-baseline --no-facade > $out/$c.log 2>&1
done
python3 tools/pmc_step_summary.py $out $cfg > gpurun_out/pmc_stalls_$tag.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/pmc_stalls_$tag.json"))
for k, v in d["kernels"].items():
    print(k[:50]); print("   ", {a: b for a, b in v.items() if not a.startswith("_")})
PY
