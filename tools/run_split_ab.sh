#!/bin/bash
# same-box A/B of the three-instruction f16 split (emx_common.h: split_f16_pair) against the compiler's form (-DEMX_SPLIT_PLAIN,
# build_ab/libemx_plain.so by tools/build_variant.sh): bit-identity fingerprints, then timings, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/split_ab.log; rm -f $out
echo "== fingerprints: shipped" >> $out
timeout 600 python tools/split_ab.py 2>&1 | grep -v amdgpu.ids | tail -4 >> $out
echo "== fingerprints: plain" >> $out
EMX_LIB_PATH=$PWD/build_ab/libemx_plain.so timeout 600 python tools/split_ab.py 2>&1 | grep -v amdgpu.ids | tail -4 >> $out
for rep in 1 2 3; do
  for lib in "" $PWD/build_ab/libemx_plain.so; do
    for args in "--mode encode --steps 20 --warmup 3" "--mode decode --steps 20 --warmup 3" "--steps 2000 --warmup 200"; do
      echo "== lib=${lib:-shipped} $args" >> $out
      EMX_LIB_PATH=$lib timeout 600 python bench.py $args --no-cpu-baseline 2>&1 | grep ms_per_step | sed -e 's/.*"ms_per_step": \([0-9.]*\).*/\1/' >> $out
    done
  done
done
timeout 900 python -m pytest tests/test_infer_gpu.py tests/test_mlp_gpu.py -q -m gpu -x 2>&1 | tail -3 >> $out
cat $out
