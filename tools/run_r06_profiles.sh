# round 6: the measurements profiles/ cites -- bash tools/run_r06_profiles.sh [tag]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r06}
o=gpurun_out/$tag
mkdir -p $o
python3 bench.py > $o/bench.json 2> $o/bench.err
python3 bench.py --steps 20 --warmup 5 > $o/bench_short.json 2> $o/bench_short.err      # (the driver's command shape)
for c in C1 C3 C4 C5; do python3 bench.py --config $c --steps 200 --warmup 20 > $o/bench_$c.json 2> $o/bench_$c.err; done
for m in encode decode; do python3 bench.py --mode $m --steps 20 --warmup 3 > $o/infer_$m.json 2> $o/infer_$m.err; done
python3 bench.py --mode encode --config C1 --steps 50 --warmup 5 --no-cpu-baseline > $o/infer_encode_C1.json 2> /dev/null
python3 bench.py --mode encode --config C4 --steps 3 --warmup 1 --no-cpu-baseline > $o/infer_encode_C4.json 2> /dev/null
python3 bench.py --mode encode --config C5 --steps 10 --warmup 2 --no-cpu-baseline > $o/infer_encode_C5.json 2> /dev/null
python3 bench.py --mode decode --config C4 --steps 3 --warmup 1 --no-cpu-baseline > $o/infer_decode_C4.json 2> /dev/null
python3 bench.py --mode decode --config C5 --steps 10 --warmup 2 --no-cpu-baseline > $o/infer_decode_C5.json 2> /dev/null
for c in C2 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$c -- python3 bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-facade > $o/prof_$c.json 2> $o/prof_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_encode -- python3 bench.py --mode encode --steps 50 --warmup 5 --no-cpu-baseline > $o/prof_encode.json 2> $o/prof_encode.err
# the data-parallel path with the one rank this box allows, over RCCL: probe-decided captured all-reduce (default) and the eager form
python3 bench.py --one-rank-dp --steps 300 --warmup 30 --no-cpu-baseline > $o/dp_one_rank_captured.json 2> $o/dp_one_rank_captured.err
EMX_DP_GRAPH_ALLREDUCE=0 python3 bench.py --one-rank-dp --steps 300 --warmup 30 --no-cpu-baseline > $o/dp_one_rank_eager.json 2> $o/dp_one_rank_eager.err
bash tools/run_pmc_step.sh ${tag}_c2 C2 > $o/pmc_c2.txt 2>&1
bash tools/run_pmc_infer.sh ${tag}_encode encode > $o/pmc_encode.txt 2>&1
bash tools/run_pmc_infer.sh ${tag}_decode decode > $o/pmc_decode.txt 2>&1
bash tools/run_pmc_infer.sh ${tag}_c4_encode encode C4 > $o/pmc_c4_encode.txt 2>&1
bash tools/run_pmc_infer.sh ${tag}_c4_decode decode C4 > $o/pmc_c4_decode.txt 2>&1
for c in C2 C5 encode; do echo "== $c"; python3 tools/prof_summary.py $o/prof_$c | head -7; done
for f in bench bench_short bench_C1 bench_C3 bench_C4 bench_C5 infer_encode infer_decode infer_encode_C1 infer_encode_C4 infer_encode_C5 infer_decode_C4 infer_decode_C5 dp_one_rank_captured dp_one_rank_eager; do
  python3 -c "
import json,sys
try:
    d=json.loads(open('$o/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step']*1e3,2),'us/step', round(d['value']/1e6,2),'M/s', d['config'].get('allreduce'), (d.get('roofline') or {}).get('frac'))
except Exception as e: print('$f', 'FAILED', e)
"
done
