#!/bin/bash
# the whole GPU suite, then the driver-shaped bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_gpu_tests.log
tail -4 gpurun_out/r06_gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_c2.json 2> gpurun_out/r06_bench_c2.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_c2.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "launches_per_step")}, d.get("inference", {}).get("encode", {}).get("ms_per_pass"), d.get("inference", {}).get("decode", {}).get("ms_per_pass"))
print(d["roofline"]["kernel"], d["roofline"]["kernel_us"], d["roofline"]["frac"])
PY
