cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for x in 0 1; do
  export EMX_WGRAD_LOCAL_XCD=$x
  python3 bench.py --steps 300 --warmup 30 --no-facade --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('xcd-aware $x: ms/step', d['ms_per_step'])"
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    c=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pl_$x/$c -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-facade > /dev/null 2>&1
  done
  python3 - <<PY
import csv,glob,statistics,collections
v=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pl_$x/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wgrad_local" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"])); v["dur"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print({k: round(statistics.median(x),1) for k,x in v.items()})
PY
done
