cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/adc_prof8 -- python3 tools/bench_adc.py $ADC_ARGS > gpurun_out/adc_prof8.log 2>&1
tail -1 gpurun_out/adc_prof8.log
