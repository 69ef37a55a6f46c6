cd $GRAFT_REPO_ROOT
# the one-launch tail (emx_mlp_wgrad_adam_local) for batches beyond 4096 rows: EMX_WGRAD_LOCAL_MAX_ROWS
for rep in 1 2; do
  for c in C5 C4; do
    for v in 4096 16384; do
      for sp in 2 4; do
        EMX_WGRAD_LOCAL_MAX_ROWS=$v EMX_WGRAD_LOCAL_SPLITS=$sp python3 bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$c max_rows=$v splits<=$sp', round(d['ms_per_step']*1e3,2),'us/step launches', d['launches_per_step'], d['loss_first_last'])"
      done
    done
  done
done
