# hardware counters of every launch of the C2 training step (one rocprofv3 --pmc pass per counter set, bench.py as the workload)
#   bash tools/run_pmc_step.sh [tag]   ->  gpurun_out/pmc_step_<tag>/ and a JSON summary on stdout (kept as profiles/rNN_step_counters.json)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r02}
out=gpurun_out/pmc_step_$tag
mkdir -p $out
for set in "MfmaUtil" "VALUBusy" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  c=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$c -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > $out/$c.log 2>&1
  tail -c 200 $out/$c.log | head -c 200 >&2; echo >&2
done
python3 tools/pmc_step_summary.py $out
