#!/bin/bash
# the one-shot all-reduce (emx_p2p_*): its tests, the bench line of two ranks on one GPU, and what ONE rank costs through it against RCCL
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_p2p_gpu.py tests/test_dp_gpu.py -q -m gpu -x -k "p2p or peer_mapped" 2>&1 | tail -4
for v in "" 1; do
  echo "== one rank, EMX_DP_P2P=${v:-unset (RCCL)}"
  EMX_DP_P2P=$v timeout 600 python bench.py --one-rank-dp --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), 'us/step', d['config']['allreduce'])"
done
