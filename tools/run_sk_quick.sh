cd $GRAFT_REPO_ROOT
# the 64-tile all-pairs kernel alone, in a graph, several times (same box): tools/time_sk_graph.py
for rep in 1 2 3; do EMX_SK_W16_MIN_B=1000000 python3 tools/time_sk_graph.py 128 2048 4096 2>&1 | grep -v amdgpu.ids; done
EMX_SK_W16_MIN_B=1000000 python3 tools/time_sk_graph.py 256 4096 2>&1 | grep -v amdgpu.ids
