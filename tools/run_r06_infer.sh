#!/bin/bash
# round 6: the streaming inference launch -- parity tests, then encode / decode timings of its two forms and of the training kernels forward-only
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r06_infer_tests.log gpurun_out/r06_infer_bench.log
timeout 900 python -m pytest tests/test_infer_gpu.py tests/test_reference_kat_gpu.py -q -m gpu > gpurun_out/r06_infer_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06_infer_tests.log
for mode in encode decode; do
  for env in EMX_INFER_STREAM=1 EMX_INFER_STREAM=0; do     # (the streaming launch / the training kernels forward-only)
    echo "== $mode $env" >> gpurun_out/r06_infer_bench.log
    env $env timeout 600 python bench.py --mode $mode --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | cut -c1-900 >> gpurun_out/r06_infer_bench.log
  done
done
echo "== C1 encode" >> gpurun_out/r06_infer_bench.log
timeout 600 python bench.py --mode encode --config C1 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | cut -c1-900 >> gpurun_out/r06_infer_bench.log
tail -5 gpurun_out/r06_infer_tests.log
grep "==\|ms_per_step" gpurun_out/r06_infer_bench.log | sed -e 's/.*"ms_per_step": \([0-9.]*\).*/\1/' | paste - -
