cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in default build_ab/libemx_prev.so; do
  if [ "$lib" = default ]; then unset EMX_LIB_PATH; else export EMX_LIB_PATH=$GRAFT_REPO_ROOT/$lib; fi
  echo "== $lib"
  python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('   step', round(d['ms_per_step']*1e3,2),'us  chain', round(d['roofline']['kernel_us'],2), 'loss', d['loss_first_last'])"
done
done
unset EMX_LIB_PATH
python tools/abl_timing.py build_ab/libemx_timing.so fused 2>/dev/null | head -12
python tools/abl_timing.py build_ab/libemx_timing.so bwd_latent 2>/dev/null | head -4
