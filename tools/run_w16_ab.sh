cd $GRAFT_REPO_ROOT
# the 128 x 128 macro-tile on sixteen waves (sk_tile_f16_w16_kernel) against the 64-tiles and the four-wave macro-tiles, one box
for rep in 1 2; do
echo "== 64-tiles (w16 off)";  EMX_SK_W16_MIN_B=1000000 python3 tools/time_sk_graph.py 128 2048 3072 4096 6144
echo "== w16 from 2048 rows";  EMX_SK_W16_MIN_B=2048 python3 tools/time_sk_graph.py 128 2048 3072 4096 6144
done
echo "== C5 shape: four-wave macro-tiles";  python3 tools/time_sk_graph.py 256 8192 16384
echo "== C5 shape: w16";  EMX_SK_W16_MIN_B=2048 EMX_SK_BIG_MIN_B=1000000 python3 tools/time_sk_graph.py 256 8192 16384
for rep in 1 2; do
for v in 1000000 3072; do
EMX_SK_W16_MIN_B=$v python3 bench.py --config C2 --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('w16_min_b=$v', round(d['ms_per_step']*1e3,2),'us/step  all-pairs alone', round(d['roofline_all_pairs']['kernel_us'],2), d['loss_first_last'])"
done
done
