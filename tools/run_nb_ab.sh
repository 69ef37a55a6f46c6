cd $GRAFT_REPO_ROOT
# register ring of the weight stream two / three units deep, INSIDE the step (the weights' first touch after the optimizer launch)
for rep in 1 2; do
  for v in "EMX_CHAIN_W16=0" "EMX_CHAIN_W16=-1" "EMX_CHAIN_NB=3"; do
    env $v python3 bench.py --config C2 --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', round(d['ms_per_step']*1e3,2),'us/step  fused alone', round(d['roofline']['kernel_us'],2))"
  done
done
