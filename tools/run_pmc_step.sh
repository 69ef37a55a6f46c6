# hardware counters of every launch of a configuration's training step (one rocprofv3 --pmc pass per counter set, bench.py as the workload)
#   bash tools/run_pmc_step.sh <tag> [config=C2] [steps=40]   ->  gpurun_out/pmc_step_<tag>/ and gpurun_out/pmc_step_<tag>.json
#   (kept as profiles/rNN_step_counters.json for C2, profiles/rNN_<config>_counters.json otherwise)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r03}; cfg=${2:-C2}; steps=${3:-40}
out=gpurun_out/pmc_step_$tag
mkdir -p $out
for set in "MfmaUtil" "VALUBusy" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  c=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$c -- python3 bench.py --config $cfg --steps $steps --warmup 8 --no-cpu-baseline --no-facade > $out/$c.log 2>&1
  tail -c 200 $out/$c.log | head -c 120 >&2; echo >&2
done
python3 tools/pmc_step_summary.py $out $cfg > gpurun_out/pmc_step_$tag.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/pmc_step_$tag.json"))
for k, v in d["kernels"].items():
    print(k[:44].ljust(44), {a: v.get(a) for a in ("_dur_us_under_pmc", "MfmaUtil", "VALUBusy", "hbm_bytes_per_launch", "l2_hit_rate", "valu_per_wave")})
PY
