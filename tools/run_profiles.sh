cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
python tools/bench_dp_structure.py > gpurun_out/r01b/dp_eager.txt 2>&1
EMX_DP_GRAPH_ALLREDUCE=1 python tools/bench_dp_structure.py > gpurun_out/r01b/dp_captured.txt 2>&1
python bench.py > gpurun_out/r01b/bench.json 2> gpurun_out/r01b/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/prof -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r01b/prof_bench.json 2> gpurun_out/r01b/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/prof_dp -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --one-rank-dp > gpurun_out/r01b/prof_dp_bench.json 2> gpurun_out/r01b/prof_dp.err
grep -h "us\|step" gpurun_out/r01b/dp_eager.txt gpurun_out/r01b/dp_captured.txt | tail -8
cut -c1-300 gpurun_out/r01b/bench.json
