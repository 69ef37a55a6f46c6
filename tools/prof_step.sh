# kernel-trace statistics of one configuration's step:  bash tools/prof_step.sh C5 [tag]   (run on the GPU box; writes gpurun_out/prof_<tag>/)
cfg=${1:-C2}; tag=${2:-$cfg}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-facade > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
python3 tools/prof_summary.py gpurun_out/prof_$tag | head -${3:-8}
