cd $GRAFT_REPO_ROOT
# float32 rows nobody reads left out of HBM (EMX_LEAN_ROWS, engine.plan_lean) against all rows written: same box, alternating
for rep in 1 2 3; do
  for lean in 1 0; do
    for c in C2 C1 C5; do
      EMX_LEAN_ROWS=$lean python3 bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('lean=$lean $c', round(d['ms_per_step']*1e3,2),'us/step  largest', d['roofline']['kernel'], round(d['roofline']['kernel_us'],2), 'loss', d['loss_first_last'])"
    done
  done
done
