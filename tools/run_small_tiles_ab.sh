cd $GRAFT_REPO_ROOT
for w in 0 -1; do
  echo "== EMX_CHAIN_W16=$w (0: SMALL builds, -1: general builds)"
  for c in C1 C2; do
  EMX_CHAIN_W16=$w python3 bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('   $c step', round(d['ms_per_step']*1e3,2),'us  largest', round(d['roofline']['kernel_us'],2), 'loss', d['loss_first_last'])"
  done
  for b in 512 1024 1536; do EMX_CHAIN_W16=$w python3 tools/time_overlap.py $b 128 2>&1 | tail -1; done
done
