# hardware counters of the streaming inference launch (one rocprofv3 --pmc pass per counter set, bench.py --mode <encode|decode> as the workload)
#   bash tools/run_pmc_infer.sh <tag> [mode=encode] [config=C2]   ->  gpurun_out/pmc_infer_<tag>/ and gpurun_out/pmc_infer_<tag>.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r06}; mode=${2:-encode}; cfg=${3:-C2}

out=gpurun_out/pmc_infer_$tag
mkdir -p $out
for set in "MfmaUtil" "VALUBusy" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  c=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$c -- python3 bench.py --mode $mode --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $out/$c.log 2>&1
  tail -c 200 $out/$c.log | head -c 120 >&2; echo >&2
done
python3 tools/pmc_step_summary.py $out C2 > gpurun_out/pmc_infer_$tag.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/pmc_infer_$tag.json"))
for k, v in d["kernels"].items():
    if "infer" in k:
        print(k[:60]); print({a: b for a, b in v.items()})
PY
