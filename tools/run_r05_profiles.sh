# round 5: the measurements profiles/ cites -- bash tools/run_r05_profiles.sh [tag] [quick]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r05}; quick=${2:-}
o=gpurun_out/$tag
mkdir -p $o
python3 bench.py > $o/bench.json 2> $o/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-facade > $o/bench_short.json 2> $o/bench_short.err      # (the driver's command shape)
for c in C1 C3 C4 C5; do python3 bench.py --config $c --steps 200 --warmup 20 ${quick:+--no-cpu-baseline} > $o/bench_$c.json 2> $o/bench_$c.err; done
for c in C2 C1 C3 C4 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$c -- python3 bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-facade > $o/prof_$c.json 2> $o/prof_$c.err
done
# the data-parallel path with the one rank this box allows, over RCCL: probe-decided captured all-reduce (default) and the eager form
python3 bench.py --one-rank-dp --steps 300 --warmup 30 --no-cpu-baseline > $o/dp_one_rank_captured.json 2> $o/dp_one_rank_captured.err
EMX_DP_GRAPH_ALLREDUCE=0 python3 bench.py --one-rank-dp --steps 300 --warmup 30 --no-cpu-baseline > $o/dp_one_rank_eager.json 2> $o/dp_one_rank_eager.err
python3 bench.py --one-rank-dp --dp-mode rowblock --collectives 1 --steps 300 --warmup 30 --no-cpu-baseline > $o/dp_one_rank_rowblock1.json 2> /dev/null
python3 bench.py --one-rank-dp --dp-mode rowblock --collectives 2 --steps 300 --warmup 30 --no-cpu-baseline > $o/dp_one_rank_rowblock2.json 2> /dev/null
python3 tools/time_sk_latent.py > $o/sk_latent.txt 2>&1
bash tools/run_pmc_step.sh ${tag}_c2 C2 > $o/pmc_c2.txt 2>&1
for c in C2 C1 C3 C4 C5; do echo "== $c"; python3 tools/prof_summary.py $o/prof_$c | head -7; done
for f in bench bench_short bench_C1 bench_C3 bench_C4 bench_C5 dp_one_rank_captured dp_one_rank_eager dp_one_rank_rowblock1 dp_one_rank_rowblock2; do
  python3 -c "
import json,sys
try:
    d=json.loads(open('$o/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step']*1e3,2),'us/step', round(d['value']/1e6,2),'M/s', d['config'].get('allreduce'), d['config'].get('collectives_per_step'))
except Exception as e: print('$f', 'FAILED', e)
"
done
