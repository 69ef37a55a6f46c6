cd $GRAFT_REPO_ROOT
# two builds of the library on one box (EMX_LIB_PATH): the tree's against build_ab/<name>.so, C2 / C1 / C3, alternating
alt=${1:-build_ab/libemx_base.so}
for rep in 1 2 3; do
  for c in C2 C1 C3; do
    for lib in "" $alt; do
      EMX_LIB_PATH=$lib python3 bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$c', '${lib:-tree}', round(d['ms_per_step']*1e3,2),'us/step  largest alone', round(d['roofline']['kernel_us'],2), d['loss_first_last'][1])"
    done
  done
done
