cd "$GRAFT_REPO_ROOT"; timeout 300 python tools/infer_debug.py > gpurun_out/r06_infer_debug.log 2>&1; tail -12 gpurun_out/r06_infer_debug.log
