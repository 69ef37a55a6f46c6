# the round's closing measurements (everything that profiles/README.md cites for round 3):  bash tools/run_final_profiles.sh [tag]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r03}
o=gpurun_out/$tag
mkdir -p $o
python3 bench.py > $o/bench.json 2> $o/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-facade > $o/bench_short.json 2> $o/bench_short.err      # (the driver's command shape)
for c in C1 C3 C4 C5; do python3 bench.py --config $c --steps 200 --warmup 20 > $o/bench_$c.json 2> $o/bench_$c.err; done
for c in C2 C1 C3 C4 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$c -- python3 bench.py --config $c --steps 300 --warmup 30 --no-cpu-baseline --no-facade > $o/prof_$c.json 2> $o/prof_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_dp -- python3 tools/bench_dp_structure.py > $o/dp_structure.txt 2>&1
python3 tools/dp_timeline.py $o/prof_dp > $o/dp_one_rank_timeline.txt 2>&1
EMX_DP_GRAPH_ALLREDUCE=1 python3 tools/bench_dp_structure.py 2>&1 | grep "us/step" | sed 's/^/captured all-reduce: /' >> $o/dp_structure.txt
python3 tools/split_accuracy.py > $o/split_accuracy.log 2>&1; cp gpurun_out/r03_split_accuracy.txt $o/split_accuracy.txt
for b in 512 1024 2048 4096; do python3 tools/time_overlap.py $b 128 2>&1 | tail -1; done > $o/overlap_ab.txt
bash tools/run_pmc_step.sh ${tag}_c2 C2 > $o/pmc_c2.txt 2>&1
bash tools/run_pmc_step.sh ${tag}_c5 C5 20 > $o/pmc_c5.txt 2>&1
bash tools/run_pmc_step.sh ${tag}_c4 C4 12 > $o/pmc_c4.txt 2>&1
for c in C2 C1 C3 C4 C5; do echo "== $c"; python3 tools/prof_summary.py $o/prof_$c | head -7; done
grep -h "us/step" $o/dp_structure.txt
cat $o/overlap_ab.txt
tail -1 $o/bench.json | cut -c1-200; tail -1 $o/bench_short.json | cut -c1-200
