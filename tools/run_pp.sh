cd "$GRAFT_REPO_ROOT"; timeout 600 python tools/stagger_check.py > gpurun_out/r06_stagger.log 2>&1; cat gpurun_out/r06_stagger.log
