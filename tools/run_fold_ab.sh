#!/bin/bash
# the folded first decoder layer of the transposed chain (tchain.h) against a build without it (-DTC_NO_FOLD): parity tests, then
# alternating bench runs:   bash tools/run_fold_ab.sh build_ab/libemx_nofold.so
cd "$GRAFT_REPO_ROOT" || exit 1
other=$PWD/$1
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_headline_gpu.py -q -m gpu -x -k "transposed or tchain or c2_three or fused_launch or trajectory" 2>&1 | tail -4
for rep in 1 2 3; do
  for lib in "" $other; do
    t=$(EMX_LIB_PATH=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), d['roofline']['kernel'], round(d['roofline']['kernel_us'],2), d['loss_first_last'])")
    echo "rep $rep lib=${lib:-shipped(fold)} $t"
  done
done
