# the round's closing measurements: bench line, kernel stats of the same command, the other configurations, ADC
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01c
python bench.py > gpurun_out/r01c/bench.json 2> gpurun_out/r01c/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01c/prof -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r01c/prof_bench.json 2> gpurun_out/r01c/prof.err
python tools/bench_configs.py C1 C4 C5 > gpurun_out/r01c/configs.txt 2>&1
python tools/bench_adc.py >> gpurun_out/r01c/configs.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01c/prof_c1 -- python3 tools/bench_configs.py C1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01c/prof_adc -- python3 tools/bench_adc.py > /dev/null 2>&1
grep -v amdgpu.ids gpurun_out/r01c/configs.txt
cut -c1-200 gpurun_out/r01c/bench.json
