cd $GRAFT_REPO_ROOT
# the step's last launch with its row parts dealt by XCD (EMX_WGRAD_LOCAL_XROWS=1: eight parts, part p = row blocks b % 8 == p on XCD p)
EMX_WGRAD_LOCAL_XROWS=1 python3 -m pytest tests/test_mlp_gpu.py tests/test_headline_gpu.py -x -q -k "local or c2 or C2 or tail" 2>&1 | tail -3
for rep in 1 2 3; do
  for v in 0 1; do
    EMX_WGRAD_LOCAL_XROWS=$v python3 bench.py --config C2 --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('xrows=$v', round(d['ms_per_step']*1e3,2),'us/step', d['loss_first_last'])"
  done
done
