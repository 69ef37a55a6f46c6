cd $GRAFT_REPO_ROOT
# the all-pairs launch beside the decoder's backward launch on a second stream inside the step's graph (EMX_FORK_STREAMS=1), re-checked on this ROCm
b() { python3 bench.py --config C2 --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['ms_per_step']*1e3,2),'us/step', d['loss_first_last'][1], d.get('launches_per_step'))"; }
for rep in 1 2; do
b "default"
EMX_FORK_STREAMS=1 b "EMX_FORK_STREAMS=1"
EMX_FUSE_DECODER=0 b "EMX_FUSE_DECODER=0 (no fork: the same launches in one stream)"
done
