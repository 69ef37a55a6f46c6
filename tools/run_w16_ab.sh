# A/B of the dense-layer row-block kernels: eight waves SMALL build (EMX_CHAIN_W16=0, default), general build (-1)
cd $GRAFT_REPO_ROOT
for w in 0 -1; do
  echo "== EMX_CHAIN_W16=$w"
  EMX_CHAIN_W16=$w python3 tools/time_mlp.py 4096 128 2>&1 | grep -E "forward  us|forward \+ cost|encoder backward|backward us"
  EMX_CHAIN_W16=$w python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('   step', round(d['ms_per_step']*1e3,2),'us  chain', round(d['roofline']['kernel_us'],2), 'loss', d['loss_first_last'])"
done
