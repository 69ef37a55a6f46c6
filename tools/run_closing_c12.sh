# closing measurements of the configurations the one-launch step tail changes (C1, C2):  bash tools/run_closing_c12.sh [tag]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r03e}
o=gpurun_out/$tag
mkdir -p $o
python3 bench.py > $o/bench.json 2> $o/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-facade > $o/bench_short.json 2> $o/bench_short.err      # (the driver's command shape)
python3 bench.py --config C1 --steps 200 --warmup 20 > $o/bench_C1.json 2> $o/bench_C1.err
EMX_LOCAL_WGRAD=0 python3 bench.py --no-cpu-baseline --no-facade > $o/bench_two_launches.json 2> $o/bench_two_launches.err
EMX_LOCAL_WGRAD=0 python3 bench.py --config C1 --steps 200 --warmup 20 --no-cpu-baseline --no-facade > $o/bench_C1_two_launches.json 2> /dev/null
for c in C2 C1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$c -- python3 bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-facade > $o/prof_$c.json 2> $o/prof_$c.err
done
bash tools/run_pmc_step.sh ${tag}_c2 C2 > $o/pmc_c2.txt 2>&1
for c in C2 C1; do echo "== $c"; python3 tools/prof_summary.py $o/prof_$c | head -6; done
cat $o/pmc_c2.txt | tail -6
for f in bench bench_short bench_C1 bench_two_launches bench_C1_two_launches; do echo "$f: $(tail -n 1 $o/$f.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('launches_per_step'), d.get('facade_ms_per_step'), d.get('facade_progress_ms_per_step'))")"; done
