cd $GRAFT_REPO_ROOT
for cfg in "2 1000" "3 1000" "3 38" "3 32" "2 32" "4 1000"; do
  set -- $cfg
  echo "== EMX_WGRAD_LOCAL_SPLITS=$1 EMX_LOCAL_GATHER_WGS=$2"
  EMX_WGRAD_LOCAL_SPLITS=$1 EMX_LOCAL_GATHER_WGS=$2 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('   step', round(d['ms_per_step']*1e3,2),'us  loss', d['loss_first_last'])"
done
