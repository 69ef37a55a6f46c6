import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(1)
d = json.loads(line[-1])
print(f"fork={os.environ.get('EMX_FORK_STREAMS','1')} ms/step={d['ms_per_step']:.4f} samples/s={d['value']:.4g} tile_us={d['roofline']['kernel_us']:.2f} frac={d['roofline']['frac']:.3f} loss={d['loss_first_last']}")
