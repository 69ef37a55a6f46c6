#!/bin/bash
# kernel times (rocprofv3) of the inference launch under the two split forms, alternating
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/split_prof; rm -rf $o; mkdir -p $o
for rep in 1 2; do
  for v in shipped plain; do
    lib=""; [ $v = plain ] && lib=$PWD/build_ab/libemx_plain.so
    for m in encode decode; do
      EMX_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $o/${v}_${m}_$rep -- python3 bench.py --mode $m --steps 40 --warmup 5 --no-cpu-baseline > $o/${v}_${m}_$rep.json 2> $o/${v}_${m}_$rep.err
      echo "== $v $m $rep: $(python3 tools/prof_summary.py $o/${v}_${m}_$rep | grep infer16 | head -2 | tr '\n' ' ')"
    done
  done
done
