cd $GRAFT_REPO_ROOT
# defaults of the step's switches re-checked at C2 on the current build (same box, alternating; 300-step runs)
b() { python3 bench.py --config C2 --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['ms_per_step']*1e3,2),'us/step', d['loss_first_last'][1])"; }
for rep in 1 2; do
b "defaults"
EMX_SK_XCD=0 b "EMX_SK_XCD=0"
EMX_CHAIN_NB=3 b "EMX_CHAIN_NB=3"
EMX_WGRAD_LOCAL_SPLITS=1 b "EMX_WGRAD_LOCAL_SPLITS=1"
EMX_WGRAD_LOCAL_XCD=0 b "EMX_WGRAD_LOCAL_XCD=0"
EMX_GRAPH_STEPS=64 b "EMX_GRAPH_STEPS=64"
EMX_GRAPH_STEPS=16 b "EMX_GRAPH_STEPS=16"
EMX_LEAN_ROWS=0 b "EMX_LEAN_ROWS=0"
EMX_LOCAL_WGRAD=0 b "EMX_LOCAL_WGRAD=0"
EMX_SK_BIG_MIN_B=4096 b "EMX_SK_BIG_MIN_B=4096 (macro-tiles)"
done
