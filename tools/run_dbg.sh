cd "$GRAFT_REPO_ROOT"; rm -f gpurun_out/r06_adc_measured_errors.jsonl
EMX_RECORD_ADC_ERRORS=gpurun_out/r06_adc_measured_errors.jsonl timeout 1200 python -m pytest tests/test_adc_gpu.py -q -m gpu -k test_adc_loss_trajectory 2>&1 | tail -3; cat gpurun_out/r06_adc_measured_errors.jsonl
