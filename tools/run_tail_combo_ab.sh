cd $GRAFT_REPO_ROOT
# the step's last launch (emx_mlp_wgrad_adam_local) at C2: gather workgroups beside the 144 block workgroups (same box, alternating)
b() { python3 bench.py --config C2 --steps 300 --warmup 30 --no-cpu-baseline --no-facade 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['ms_per_step']*1e3,2),'us/step', d['loss_first_last'][1])"; }
for rep in 1 2 3; do
b "64 gather wgs (default)"
for g in 48 40 32 24 16 8; do EMX_LOCAL_GATHER_WGS=$g b "$g gather wgs"; done
done
