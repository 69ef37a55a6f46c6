cd $GRAFT_REPO_ROOT
# every non-default step plan against the oracle: the training-trajectory tests under one switch at a time
for sw in EMX_LEAN_ROWS=0 EMX_DRAW_AHEAD=0 EMX_APPEND_BLOCKS=0 EMX_LOCAL_WGRAD=0 EMX_FUSE_DECODER=0 EMX_OVERLAP_TILES=0 EMX_JOIN_BWD=0 EMX_FOLD_FINISH=0 EMX_TAIL_ADAM=1 EMX_GRAPH_STEPS=1 EMX_CHAIN_W16=-1 EMX_SK_XCD=0 EMX_SK_PHASE=0; do
  r=$(env $sw python3 -m pytest tests/test_training_gpu.py tests/test_headline_gpu.py -x -q -k "loss_trajectory or without_distance or c2_ or graph_replay or short_calls" 2>&1 | tail -1)
  echo "$sw: $r"
done
