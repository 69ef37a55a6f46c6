#!/bin/bash
# same-box A/B of two builds on the wide-end inference passes: bash tools/run_wide_ab.sh build_ab/libemx_prev.so
cd "$GRAFT_REPO_ROOT" || exit 1
other=$PWD/$1
for rep in 1 2 3; do
  for lib in "" $other; do
    for cm in "C4 encode" "C4 decode" "C5 encode" "C5 decode"; do
      set -- $cm
      t=$(EMX_LIB_PATH=$lib timeout 300 python bench.py --mode $2 --config $1 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep ms_per_step | sed -e 's/.*"ms_per_step": \([0-9.]*\).*/\1/')
      echo "rep $rep lib=${lib:-shipped} $cm $t"
    done
  done
done
