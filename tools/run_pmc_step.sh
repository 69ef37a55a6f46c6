cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in MfmaUtil VALUBusy FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_step/$c -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/pmc_step_$c.log 2>&1
  tail -c 300 gpurun_out/pmc_step_$c.log
done
python tools/pmc_step_summary.py gpurun_out/pmc_step
