cd $GRAFT_REPO_ROOT
# The step's last launch with its row parts dealt by XCD (eight parts, part p = row blocks b % 8 == p on XCD p): an EXPERIMENT that
# is not in the tree's library.  The switch EMX_WGRAD_LOCAL_XROWS exists only in tools/experiments/wgrad_xrows.patch; build the
# patched library on the build box first,
#     tools/build_variant.sh tools/experiments/wgrad_xrows.patch build_ab/libemx_xrows.so
# (the patch was cut against commit a58b57e: re-cut it if the sources have moved on), then run this script on the GPU box.  Without
# that library both arms would run the same code, so the script refuses to start.
alt=build_ab/libemx_xrows.so
if [ ! -f "$alt" ]; then echo "run_xrows_ab.sh: $alt is missing -- see the header of this script" >&2; exit 2; fi
EMX_LIB_PATH=$alt EMX_WGRAD_LOCAL_XROWS=1 python3 -m pytest tests/test_mlp_gpu.py tests/test_headline_gpu.py -x -q -k "local or c2 or C2 or tail" 2>&1 | tail -3
for rep in 1 2 3; do
  for v in 0 1; do
    EMX_LIB_PATH=$alt EMX_WGRAD_LOCAL_XROWS=$v python3 bench.py --config C2 --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('xrows=$v', round(d['ms_per_step']*1e3,2),'us/step', d['loss_first_last'])"
  done
done
