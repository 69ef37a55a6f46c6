# one-rank data-parallel step under the kernel tracer: which launches does the data-parallel path add?   bash tools/prof_dp.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-dp}
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 tools/bench_dp_structure.py > gpurun_out/prof_$tag.log 2>&1
tail -3 gpurun_out/prof_$tag.log
python3 tools/prof_summary.py gpurun_out/prof_$tag | head -14
python3 tools/dp_timeline.py gpurun_out/prof_$tag
