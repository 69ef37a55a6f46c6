# kernel trace of the C4-shape step (D=512 periodic, B=8192), fixed-point tiles and (EMX_SK_PHASE=0) the float form
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ph in 1 0; do
  export EMX_SK_PHASE=$ph
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4_phase$ph -- python3 tools/bench_configs.py C4 > gpurun_out/c4_phase$ph.txt 2>&1
  tail -1 gpurun_out/c4_phase$ph.txt
  python3 tools/prof_summary.py gpurun_out/c4_phase$ph | head -4
done
