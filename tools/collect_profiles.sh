#!/bin/bash
# copy the summaries of tools/run_r06_profiles.sh (gpurun_out/<tag>/) into profiles/ under the round's names:
#   bash tools/collect_profiles.sh r06
tag=${1:-r06}
o=gpurun_out/$tag; p=profiles
cp $o/bench.json $p/${tag}_final_bench.json
cp $o/bench_short.json $p/${tag}_bench_short.json
for c in C1 C3 C4 C5; do cp $o/bench_$c.json $p/${tag}_bench_$c.json; done
for m in encode decode; do cp $o/infer_$m.json $p/${tag}_infer_$m.json; done
for c in C1 C4 C5; do cp $o/infer_encode_$c.json $p/${tag}_infer_encode_$c.json; done
for c in C4 C5; do [ -f $o/infer_decode_$c.json ] && cp $o/infer_decode_$c.json $p/${tag}_infer_decode_$c.json; done
cp $o/prof_C2.json $p/${tag}_final_bench_profile_run.json
cp $(ls -t $o/prof_C2/*/*_kernel_stats.csv | head -1) $p/${tag}_final_kernel_stats.csv
cp $(ls -t $o/prof_C5/*/*_kernel_stats.csv | head -1) $p/${tag}_c5_kernel_stats.csv
cp $(ls -t $o/prof_encode/*/*_kernel_stats.csv | head -1) $p/${tag}_encode_kernel_stats.csv
cp $o/dp_one_rank_captured.json $p/${tag}_dp_one_rank_captured.json
cp $o/dp_one_rank_eager.json $p/${tag}_dp_one_rank_eager.json
cp gpurun_out/pmc_step_${tag}_c2.json $p/${tag}_step_counters.json
for m in encode decode; do [ -f gpurun_out/pmc_infer_${tag}_$m.json ] && cp gpurun_out/pmc_infer_${tag}_$m.json $p/${tag}_infer_counters_$m.json; done
for m in encode decode; do [ -f gpurun_out/pmc_infer_${tag}_c4_$m.json ] && cp gpurun_out/pmc_infer_${tag}_c4_$m.json $p/${tag}_infer_counters_c4_$m.json; done
{
  echo "# one line per run of tools/run_r06_profiles.sh $tag (us per step or pass, M samples or rows per second, all-reduce form, roofline.frac)"
  grep -E "^(bench|infer_|dp_)" gpurun_out/${tag}_profiles.log
  echo "# per-kernel medians as the GPU box printed them (rocprofv3 --kernel-trace of bench.py --config C2 / C5, --mode encode)"
  grep -E "^== |n= " gpurun_out/${tag}_profiles.log
} > $p/${tag}_bench_lines.txt
ls -la $p/${tag}_* | wc -l
