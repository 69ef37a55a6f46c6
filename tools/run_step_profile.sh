# kernel trace of the C2 benchmark step: median duration per kernel (tools/prof_summary.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-step}
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/$tag/prof_bench.json 2> gpurun_out/$tag/prof.err
python3 tools/prof_summary.py gpurun_out/$tag/prof | head -8
cut -c1-200 gpurun_out/$tag/prof_bench.json
