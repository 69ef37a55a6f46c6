cd $GRAFT_REPO_ROOT
# every other resident layer of 64-tiles started late (EMX_SK_STAGGER = cycles): do the Gram phase of one half and the epilogue of the other overlap?
for st in 0 1024 2048 4096 6144 8192 12288; do
echo "== stagger $st"; EMX_SK_W16_MIN_B=1000000 EMX_SK_STAGGER=$st python3 tools/time_sk_graph.py 128 3072 4096
done
