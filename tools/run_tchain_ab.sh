cd $GRAFT_REPO_ROOT
# the fused dense-layer launch as the transposed, register-resident chain (csrc/tchain.h; EMX_TCHAIN=0: the row-block chain kernel), same box, alternating
python3 -m pytest tests/test_headline_gpu.py -m gpu -x -q -k "c2_three or headline_c2 or C2" 2>&1 | tail -3
for rep in 1 2 3; do
for v in 0 1; do
EMX_TCHAIN=$v python3 bench.py --config C2 --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('EMX_TCHAIN=$v', round(d['ms_per_step']*1e3,2),'us/step  largest', d['roofline']['kernel'], round(d['roofline']['kernel_us'],2), d['loss_first_last'])"
done
done
