# A/B of library builds on one box: kernel medians of the C2 step per build   bash tools/run_ab_libs.sh build_ab/libemx_x.so ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset EMX_LIB_PATH; else export EMX_LIB_PATH=$GRAFT_REPO_ROOT/$lib; fi
  tag=$(basename $lib .so)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  echo "== $tag: $(python3 -c "import json;d=json.load(open('gpurun_out/ab_$tag.json'));print(round(d['ms_per_step']*1e3,2),'us/step under the profiler')")"
  python3 tools/prof_summary.py gpurun_out/ab_$tag | head -5
  python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('   un-profiled', round(d['ms_per_step']*1e3,2),'us/step')"
done
