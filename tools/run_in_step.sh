cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 tools/time_in_step.py --alone 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/in_step_prof -- python3 tools/time_in_step.py --alone 0 2>/dev/null | grep -v "^W2026\|simple_timer"
python3 tools/prof_summary.py gpurun_out/in_step_prof | head -5
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/in_step_prof/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(r["Name"][:70], r["Calls"], "avg", round(float(r["AverageNs"]) / 1e3, 2))
PY
