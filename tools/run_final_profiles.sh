# the round's closing measurements (everything that profiles/README.md cites):  bash tools/run_final_profiles.sh [tag]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r02}
o=gpurun_out/$tag
mkdir -p $o
python3 bench.py > $o/bench.json 2> $o/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $o/prof_bench.json 2> $o/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_dp -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --one-rank-dp > $o/prof_dp_bench.json 2> $o/prof_dp.err
python3 tools/bench_configs.py C1 C4 C5 > $o/configs.txt 2>&1
python3 tools/bench_adc.py >> $o/configs.txt 2>&1
for c in C1 C4 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$c -- python3 tools/bench_configs.py $c > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_adc -- python3 tools/bench_adc.py > /dev/null 2>&1
./tools/ubench > $o/valu_issue_rates.txt 2>&1
python3 tools/abl_timing.py build_ab/libemx_timing.so fused > $o/chain_timeline.txt 2>&1
python3 tools/abl_timing.py build_ab/libemx_timing.so bwd >> $o/chain_timeline.txt 2>&1
python3 tools/abl_wgrad.py build_ab/libemx_timing.so > $o/wgrad_adam_timeline.txt 2>&1
python3 tools/abl_finish.py build_ab/libemx_timing.so >> $o/wgrad_adam_timeline.txt 2>&1
EMX_LIB_PATH=build_ab/lib_d2ct.so python3 tools/abl_d2c.py > $o/d2c_timeline.txt 2>&1
python3 tools/adc_fuse_check.py > $o/adc_fuse_check.txt 2>&1
bash tools/run_pmc_step.sh $tag > $o/step_counters.json 2> $o/step_counters.err
grep -v amdgpu.ids $o/configs.txt
cut -c1-220 $o/bench.json
python3 tools/prof_summary.py $o/prof | head -6
