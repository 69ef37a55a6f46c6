cd $GRAFT_REPO_ROOT
for c in C2 C1 C3 C4 C5; do
  python3 bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$c', round(d['ms_per_step']*1e3,2),'us/step  largest', d['roofline']['kernel'], round(d['roofline']['kernel_us'],2), ' all-pairs', round((d.get('roofline_all_pairs') or {}).get('kernel_us', 0),2))"
done
