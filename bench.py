#!/usr/bin/env python3
"""Benchmark of the EncoderMap training step on the configurations of BASELINE.json.

    python bench.py                                    # the headline: C2, one GPU, 300 steps after 30 warm-up steps
    python bench.py --config C5 --steps 100 --warmup 10
    python bench.py --mode encode [--config C2|C4]    # encode() of the resident training set (a "step" = one pass over all its rows)
    python bench.py --mode decode                      # decode() / generate() of as many latent rows
    python bench.py --gpus N --steps K --warmup W      # no launcher: the script starts torch.distributed.run itself (a child
                                                       # process, before anything touches the GPU) and relays the JSON line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config C4]

``--config`` (default C2, the configuration BASELINE.json's metric is quoted on):
  C1  em.EncoderMap, default Parameters, random float32 array 10000 x 10 (periodic, B = 256)
  C2  Gaussian blobs 1e6 x 128, B = 4096, Euclidean sketch-map cost, sig (16, 12, 6, 1, 2, 6)
  C3  AngleDihedralCartesianEncoderMap (dihedral -> Cartesian back-mapping costs), synthetic backbone of 76 residues
      (228 atoms), 50 000 frames, B = 256
  C4  synthetic dihedral trajectory 2e6 x 512 (periodic), B = 8192
  C5  Gaussian blobs 5e5 x 256, B = 16384, Euclidean

A "step" is one full training step (mini-batch gather, forward, all cost terms, backward, clip + Adam) on synthetic data
that is resident in HBM before the timed region starts.  One JSON line is printed by rank 0:

  value        whole-job samples/s = global batch * K / (max over ranks of the time for K steps)
  roofline     the kernel that takes the largest share of the STEP, timed here with HIP events around back-to-back launches on
               the live batch (torch's current stream is the stream the kernels are launched on); ``roofline_all_pairs`` the
               O(B^2 D) all-pairs kernel when it is not that kernel.  Euclidean configurations: every float32 product runs as
               THREE f16 matrix instructions on operands split into two 16-bit planes (2^-22 relative operand precision, f32
               accumulate; profiles/r03_split_accuracy.txt prices the result against a true float32 run), so ``achieved`` counts
               the EXECUTED FLOPs (6 per multiply-add) against the dense f16 matrix peak -- ``frac`` is on that basis in every
               line of this round and of round 2.  Periodic configurations: the all-pairs pass runs on the VALU (5 FLOP per
               pair-element) against the fp32 peak.  ``limiter`` names what the PMC counters say bounds the
               kernel; ``traffic`` = HBM bytes per launch from the PMC passes under profiles/ (``traffic_source`` names file and
               commit), null where no pass was taken for the configuration.
  cpu_baseline the reference formulation (oracle/reference_ops.py in float32: Gram-matrix / [B, B, D] temporaries, autograd,
               per-element clipped Adam) timed on this host: >= 50 steps after 5 warm-up steps at the fastest thread count of a
               one-step probe, and a few steps on ONE thread.  N = 1 only.  kind "port": TensorFlow cannot be installed here.
  inference    N = 1: encode() and decode() of the resident set as rows/s, each with bytes in + out / time against the 8 TB/s HBM roof
               (what ``--mode encode|decode`` print as a line of their own, with a CPU baseline of the same pass).
  facade_ms_per_step   C2, N = 1: the same steps through ``em.EncoderMap.train()`` (the user's entry point, callbacks and History
               included), with the defaults and with the progress read-out every ``summary_step`` steps.

N > 1: C2 -- weak scaling, every rank trains on its own mini-batch of 4096 rows and the flat gradient buffer is averaged by one
RCCL all-reduce per step (engine.DataParallel mode "replica"; ``--dp-mode rowblock`` is the objective-preserving partition
of ONE global batch, ``--global-batch``).  C4 / C5 -- ALWAYS the row-block partition of the configuration's one global
batch (the single-process objective exactly; strong scaling): dense layers on runs of 16-row blocks,
the all-pairs pass either as the rectangle (own rows) x (all rows) with ONE all-reduce per step (small passes: C2-sized) or with
its symmetric tiles dealt to the ranks and TWO all-reduces (C4, C5) -- ``--collectives``, default: chosen per configuration.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INF = float("inf")
FP32_PEAK = 157.3e12   # MI355X fp32 vector == fp32 MFMA peak (MI355X_MICROARCH.md chip table)
F16_PEAK = 2.5e15      # dense f16 / bf16 matrix peak (same table; the headline figure with 2:1 sparsity is not used)
HBM_PEAK = 8.0e12
SIMDS, CLOCK = 1024, 2.4e9

CONFIGS = {
    "C1": dict(rows=10_000, D=10, kw=dict(), kind="uniform01", model="encodermap",
               workload="C1: em.EncoderMap with default Parameters on a random float32 array 10000x10 (periodic 2 pi, batch 256)"),
    "C2": dict(rows=1_000_000, D=128, kw=dict(periodicity=INF, batch_size=4096, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6)), kind="blobs",
               model="encodermap", seed=1,
               workload="C2: EncoderMap training, 1e6x128 Gaussian blobs resident in HBM, batch 4096, sketch-map sigmoid loss (Euclidean), "
                        "sig=(16,12,6,1,2,6)"),
    "C3": dict(rows=50_000, D=228, kw=dict(batch_size=256, cartesian_pwd_start=1, cartesian_pwd_step=3,
                                           cartesian_dist_sig_parameters=(40.0, 10, 5, 1, 2, 5), dihedral_cost_reference=1.5,
                                           angle_cost_reference=0.3, cartesian_cost_reference=8.0),
               kind="backbone", model="adc", n_res=76,
               workload="C3: AngleDihedralCartesianEncoderMap (dihedral_to_cartesian back-mapping, cartesian + cartesian-distance costs), "
                        "synthetic backbone of 76 residues = 228 atoms, 50000 frames, batch 256"),
    "C4": dict(rows=2_000_000, D=512, kw=dict(batch_size=8192, dist_sig_parameters=(30.0, 12, 6, 1, 2, 6)), kind="dihedrals", model="encodermap", seed=4,
               workload="C4: EncoderMap training, synthetic dihedral trajectory 2e6x512 (periodic 2 pi) resident in HBM, batch 8192, "
                        "sig=(30,12,6,1,2,6)"),
    "C5": dict(rows=500_000, D=256, kw=dict(periodicity=INF, batch_size=16384, dist_sig_parameters=(22.6, 12, 6, 1, 2, 6)), kind="blobs",
               model="encodermap", seed=5,
               workload="C5: EncoderMap training, 5e5x256 Gaussian blobs resident in HBM, batch 16384, sketch-map sigmoid loss (Euclidean), "
                        "sig=(22.6,12,6,1,2,6)"),
}
DTYPES = {"blobs": "f32 (split-f16 x3 mfma: 2^-22 operand precision, f32 accumulate)",
          "dihedrals": "f32 (dense layers split-f16 x3 mfma; all-pairs pass on 32-bit fixed-point phases, f32 accumulate)",
          "uniform01": "f32 (dense layers split-f16 x3 mfma; all-pairs pass on 32-bit fixed-point phases, f32 accumulate)",
          "backbone": "f32 (dense layers and cartesian-distance Gram tiles split-f16 x3 mfma; back-mapping f32)"}


def git_head():
    """the commit this tree was built from: EMX_COMMIT, git itself, or -- on the GPU box, which has no .git -- the stamp
    that __graft_entry__.build() leaves beside the library (.build_commit; "-dirty" when the work tree differed from HEAD)"""
    if os.environ.get("EMX_COMMIT"):
        return os.environ["EMX_COMMIT"]
    try:
        head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=5).stdout.strip()
        if head:
            return head
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, ".build_commit")).read().strip() or None
    except Exception:
        return None


def source_hash():
    """first 16 hex digits of the SHA-256 over the kernel sources (encodermap_amd/csrc, include) in name order: what a counters file
    under profiles/ must carry to be quoted beside a line of THIS build (tools/pmc_step_summary.py writes it)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "encodermap_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        if f.endswith((".hip", ".h")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------------------------
# synthetic data, generated ON the device (nothing crosses PCIe inside or in front of the timed region)
# ------------------------------------------------------------------------------------------------------------------
def make_data(cfg, device, n_rows=None):
    import torch

    kind, d = cfg["kind"], cfg["D"]
    n = int(n_rows or cfg["rows"])
    g = torch.Generator(device=device)
    g.manual_seed(cfg.get("seed", 0))
    if kind == "blobs":
        # 16 blob centres N(0, 4), points = centre + N(0, 1)   
        centres = torch.randn(16, d, generator=g, device=device) * 4.0
        k = torch.randint(0, 16, (n,), generator=g, device=device)
        return centres[k] + torch.randn(n, d, generator=g, device=device)
    if kind == "uniform01":
        return torch.rand(n, d, generator=g, device=device)
    if kind == "dihedrals":
        # mixture of basins per dihedral (von-Mises-like: a wrapped normal of width 0.35 around one of three basin centres per
        # column), wrapped to (-pi, pi]; generated in chunks: 2e6 x 512 floats are 4 GB
        centres = (torch.rand(3, d, generator=g, device=device) * 2 - 1) * math.pi
        out = torch.empty(n, d, dtype=torch.float32, device=device)
        for s in range(0, n, 1 << 18):
            m = min(1 << 18, n - s)
            k = torch.randint(0, 3, (m, d), generator=g, device=device)
            x = torch.gather(centres, 0, k) + 0.35 * torch.randn(m, d, generator=g, device=device)
            out[s: s + m] = torch.remainder(x + math.pi, 2 * math.pi) - math.pi
        return out
    raise ValueError(kind)


def make_backbone(cfg, n_frames):
    """random-walk chain with protein-like bond lengths, featurised by the product's own kernel (emx_backbone_features)"""
    import numpy as np

    from encodermap_amd.features import featurize_backbone

    A = 3 * cfg["n_res"]
    rng = np.random.default_rng(3)
    step = rng.normal(size=(n_frames, A, 3)).astype(np.float32)
    step /= np.linalg.norm(step, axis=2, keepdims=True)
    return featurize_backbone(np.cumsum(1.4 * step, axis=1).astype(np.float32))


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference formulation in float32 on the host cores
# ------------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def cpu_baseline(name, n_steps=50, n_warm=5, budget_s=45.0):
    import numpy as np
    import torch

    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights
    from encodermap_amd.parameters import ADCParameters, Parameters
    from oracle import reference_ops as R

    cfg = CONFIGS[name]
    rng = np.random.default_rng(1)
    kw = dict(cfg["kw"])
    B_full = kw.get("batch_size", 256)
    D = cfg["D"]
    note = ""
    if cfg["model"] == "adc":
        data = R.synthetic_backbone(rng, 2000, cfg["n_res"])
        p = ADCParameters(**kw)
        A = 3 * cfg["n_res"]
        lay = build_layout(p.n_neurons, p.activation_functions, 2 * (A - 3))
        w0 = unpack_weights(lay, init_parameters(lay, 0))
        model = R.ADCOracle(A, 0, p.to_dict(), w0, dtype=torch.float32)
        B = B_full

        def run(idx):
            R.adc_train_steps(model, data, idx)

        n_rows = 2000
    else:
        # the reference formulation materialises [B, B] (Euclidean) or [B, B, D] (periodic) temporaries: at C4 that is 137 GB, at
        # C5 ~1 GB per temporary and a dozen of them alive under autograd -- those two run at a reduced batch (stated)
        B = {"C4": 512, "C5": 8192}.get(name, B_full)
        if B != B_full:
            kw["batch_size"] = B
            note = (f"; batch reduced from {B_full} to {B}: the formulation's {'[B, B, D]' if cfg['kind'] == 'dihedrals' else '[B, B]'} "
                    f"temporaries do not fit at the full batch -- pair work per sample scales with B, so the full-batch rate is lower still")
        n_rows = 20000
        if cfg["kind"] == "blobs":
            c = rng.normal(0, 4, (16, D))
            data = (c[rng.integers(0, 16, n_rows)] + rng.normal(0, 1, (n_rows, D))).astype(np.float32)
        elif cfg["kind"] == "uniform01":
            data = rng.random((10000, D)).astype(np.float32)
            n_rows = 10000
        else:
            data = rng.uniform(-np.pi, np.pi, (n_rows, D)).astype(np.float32)
        p = Parameters(**kw)
        net = 2 * D if p.periodicity < INF else D
        lay = build_layout(p.n_neurons, p.activation_functions, net)
        w0 = unpack_weights(lay, init_parameters(lay, 0))
        model = R.SequentialOracle(D, p.n_neurons, p.activation_functions, p.periodicity, w0, dtype=torch.float32)

        def run(idx):
            R.train_steps(model, data, idx, p.to_dict())

    draw = lambda k: rng.integers(0, n_rows, (k, B))
    run(draw(1))                                   # first touch
    # torch-CPU over-subscribes badly on many-core hosts (all threads can be several times slower than 16-32), so the thread
    # count is chosen by a one-step probe among {all, 64, 32, 16, 8} and the fastest is used and reported; one thread beside it
    cores = os.cpu_count() or torch.get_num_threads()
    best, best_t = torch.get_num_threads(), float("inf")
    for nt in sorted({cores, min(cores, 64), min(cores, 32), min(cores, 16), min(cores, 8)}, reverse=True):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        run(draw(1))
        dt1 = time.perf_counter() - t0
        if dt1 < best_t:
            best, best_t = nt, dt1
    torch.set_num_threads(best)
    steps = max(3, min(n_steps, int(budget_s / max(best_t, 1e-9))))
    warm = min(n_warm, max(1, steps // 4))
    run(draw(warm))
    t0 = time.perf_counter()
    run(draw(steps))
    dt = time.perf_counter() - t0
    # one thread: a few steps (bounded)
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    run(draw(1))
    one = time.perf_counter() - t0
    k1 = max(1, min(5, int(10.0 / max(one, 1e-9))))
    if k1 > 1:
        t0 = time.perf_counter()
        run(draw(k1))
        one = (time.perf_counter() - t0) / k1
    torch.set_num_threads(best)
    short = "" if steps >= n_steps else f" ({steps} steps instead of {n_steps}: bounded to ~{budget_s:.0f} s of CPU work)"
    return {"value": B * steps / dt, "unit": "samples/s", "cores": best, "kind": "port",
            "sample": f"{steps} training steps after {warm} warm-up steps at B={B}, D={D} (float32 torch-CPU restatement of the reference "
                      f"formulation, oracle/reference_ops.py): {dt / steps * 1e3:.0f} ms/step with {best} of {cores} host threads, the fastest "
                      f"of a one-step probe{short}{note}",
            "single_thread": {"value": B / one, "unit": "samples/s", "cores": 1, "ms_per_step": one * 1e3},
            "host_cpu": _cpu_model(), "host_threads": cores}


# ------------------------------------------------------------------------------------------------------------------
# per-kernel rooflines, timed live
# ------------------------------------------------------------------------------------------------------------------
def _time_launches(fn, launches):
    """average duration of ``fn`` (asynchronous launches on torch's current stream) over back-to-back calls, seconds"""
    import torch

    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / launches


def _counters_file(name):
    """(blob, path, reason): the newest per-launch counters of the configuration (rocprofv3 PMC passes over this script,
    tools/run_pmc_step.sh) IF they were taken on these kernel sources -- a file whose ``source_hash`` is not this build's is refused
    (None, path, why): numbers of another build are not quoted beside this one's timings."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join("profiles", f"{rnd}_step_counters.json" if name == "C2" else f"{rnd}_{name.lower()}_counters.json")
        try:
            blob = json.load(open(os.path.join(ROOT, path)))
        except Exception:
            continue
        have, want = blob.get("source_hash"), source_hash()
        if have != want:
            return None, path, (f"refused: {path} was taken on kernel sources {have or 'of an unrecorded build'} "
                                f"(commit {blob.get('commit') or 'not recorded'}), this build's are {want}")
        return blob, path, None
    return None, None, "no counters file for this configuration"


def _counters(name, kernel_key):
    """static per-launch counters of one kernel of the configuration; returns (counters, source)"""
    blob, path, reason = _counters_file(name)
    if blob is None:
        return {}, reason
    for kname, c in blob["kernels"].items():
        if kernel_key in kname:
            return c, f"{path} (static: PMC passes over this workload, commit {blob.get('commit')}, kernel sources {blob.get('source_hash')})"
    return {}, f"{path}: no kernel matching {kernel_key!r}"


def roofline_step(name, eng, live_us):
    """Every launch of the step on one line: in-step duration (the profiler's, under the counter passes), HBM bytes per launch against
    8 TB/s, matrix / vector busy, and for the all-pairs kernel SURVEY 8d's fraction (3 W or 5 W / t / 157.3 TF); ``alone_us``: the
    HIP-event timing of the kernel alone taken by this run, where there is one.  None (+ reason) without counters of this build."""
    blob, path, reason = _counters_file(name)
    if blob is None:
        return None, reason
    cfg = CONFIGS[name]
    Bt, D = eng.B, cfg["D"]
    W = D * Bt * (Bt - 1) / 2
    out = []
    for kname, c in blob["kernels"].items():
        t = c.get("_dur_us_under_pmc")
        if not t or c.get("_launches_seen", 0) < 8 or "infer16" in kname:   # (the run's inference passes: not launches of the step)
            continue
        rec = {"kernel": kname, "us_in_step": t, "hbm_bytes": c.get("hbm_bytes_per_launch"),
               "hbm_frac": (c["hbm_bytes_per_launch"] / (t * 1e-6) / HBM_PEAK) if c.get("hbm_bytes_per_launch") else None,
               "mfma_util": c.get("MfmaUtil"), "valu_busy": c.get("VALUBusy"), "l2_hit_rate": c.get("l2_hit_rate"), "frac_survey_8d": None,
               "alone_us": next((v for k, v in live_us.items() if k in kname), None)}
        if "sk_tile" in kname:
            rec["frac_survey_8d"] = (5 if eng.periodic else 3) * W / (t * 1e-6) / FP32_PEAK
        out.append(rec)
    out.sort(key=lambda r: -r["us_in_step"])
    return out, f"{path} (commit {blob.get('commit')}, kernel sources {blob.get('source_hash')})"


def _record(name, kernel, kernel_key, t, flops_alg, flops_exec, peak, roof, bound, limiter, note, algorithmic_bytes, flops_8d=None):
    """One roofline block.  ``bound`` names what the counters say limits the kernel ("issue": instruction issue / dependent
    latency; "valu": the vector unit's issue ceiling; "mfma" / "hbm" when one of those roofs binds), ``roof`` the roof that
    ``peak`` / ``achieved`` / ``frac`` are quoted against (ONE basis per kernel family: executed FLOPs against the dense f16
    matrix peak for the split-f16 kernels, SURVEY 8d's count against the fp32 peak for the VALU tiles);
    ``frac_survey_8d`` = SURVEY.md section 8d's figure (f W / t / 157.3 TF, f = 3 Euclidean, 5 periodic) for the all-pairs kernels."""
    c, src = _counters(name, kernel_key)
    valu = c.get("SQ_INSTS_VALU")
    return {"bound": bound, "roof": roof, "achieved": flops_exec / t / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": flops_exec / t / peak,
            "frac_basis": ("executed FLOPs (6 per float32 multiply-add on split-f16 operands) / dense f16 matrix peak" if roof == "mfma-f16"
                           else "SURVEY 8d FLOPs / fp32 vector peak"),
            "frac_survey_8d": (flops_8d / t / FP32_PEAK) if flops_8d else None,
            "traffic": c.get("hbm_bytes_per_launch"), "traffic_source": src,
            "kernel": kernel, "kernel_us": t * 1e6, "kernel_us_in_step": c.get("_dur_us_under_pmc"),
            "timing": "HIP events around back-to-back launches of the kernel alone on the live batch (torch's current stream = the stream "
                      "the kernels run on).  Inside the step a launch also pays the first touch of what the previous launch rewrote on other "
                      "XCDs: the profiler's per-launch trace shows the fused dense-layer launch 2-4 us longer there than alone "
                      "(profiles/r04_in_step.txt), and its --stats average mixes both kinds of launches of this command",
            "algorithmic_bytes": algorithmic_bytes,
            "hbm_frac_algorithmic": algorithmic_bytes / t / HBM_PEAK,
            "fractions": {"fp32_equivalent": flops_alg / t / FP32_PEAK, "executed": flops_exec / t / peak,
                          "valu_issue": (valu / (SIMDS * CLOCK * t)) if valu else None,
                          "valu_busy": c.get("VALUBusy"), "mfma_util": c.get("MfmaUtil")},
            "fp32_equiv_tflops": flops_alg / t / 1e12, "limiter": limiter, "note": note}


ISSUE = ("instruction issue / dependent latency: neither the matrix pipes nor HBM are near their roofs; valu_issue is VALU "
         "wave-instructions per SIMD and clock (measured ceiling 0.32-0.39 with >= 4 waves per SIMD, profiles/r02_valu_issue_rates.txt)")


def kernel_rooflines(name, eng, launches):
    """(the kernel that takes the largest share of the step, the all-pairs kernel or None -- possibly the same record), each timed
    alone on the engine's live batch."""
    from encodermap_amd import _lib

    cfg = CONFIGS[name]
    lay = eng.layout
    Bt, D = eng.B, cfg["D"]
    recs, sk = [], None
    dims = [(l.n_in, l.n_out) for l in lay.layers]
    n_par = sum(k * m + m for k, m in dims)
    if cfg["model"] == "encodermap" and eng._use_sketchmap():      # (data parallel: the full symmetric pass on the rank's live batch, timed alone)
        Z = eng.h[lay.n_enc - 1]
        sig = eng.p.dist_sig_parameters
        t_sk = _time_launches(lambda: _lib.sketchmap_tiles(eng.xb, Z, eng.periodicity, sig, 0, Bt, eng.sk_ws, eng.split_ws, data_span=eng._data_span),
                              launches)
        W = D * Bt * (Bt - 1) / 2          # pair-elements per launch (i<->j symmetry exploited)
        abytes = 4 * Bt * (D + 2 * eng.L + eng.L) + 4
        if eng.periodic:
            sk = _record(name, "sk_tile_kernel (VALU, fixed-point phases)", "sk_tile_kernel", t_sk, 5 * W, 5 * W, FP32_PEAK, "valu-fp32", "valu",
                         "VALU issue: the pair loop runs at the measured issue ceiling of the vector unit (profiles/r02_tile_counters.json); "
                         "the fp32 vector peak equals the fp32 matrix peak on this chip",
                         "direct wrapped differences on the VALU: sub, convert, fma per pair-element over the upper triangle; 5 FLOP per "
                         "pair-element by SURVEY.md section 8d's count for the periodic form", abytes, flops_8d=5 * W)
        else:
            sk = _record(name, "sk_tile_f16_kernel", "sk_tile_f16", t_sk, 2 * W, 6 * W, F16_PEAK, "mfma-f16", "issue", ISSUE,
                         "Gram tiles on rows split into two f16 planes, VALU sigmoid epilogue (~40 instructions per pair); 2 FLOP per "
                         "pair-element over the upper triangle (SURVEY.md section 8d), executed three times", abytes, flops_8d=3 * W)
        recs.append(sk)
    if cfg["model"] == "encodermap" and (eng.rb, eng.re) == (0, Bt):
        c0 = _lib.load().emx_transposed_chain_count()
        fused = eng._forward()
        transposed = _lib.load().emx_transposed_chain_count() > c0      # (csrc/tchain.h took the launch: small non-periodic networks)
        t_ch = _time_launches(eng._forward, launches)
        mults = Bt * (sum(k * m for k, m in dims) + (sum(k * m for k, m in dims[lay.n_enc:]) if fused else 0))
        kname = "tchain_fused_kernel" if transposed else ("chain_kernel<FUSED>" if fused else "chain_kernel<FWD>")
        recs.append(_record(name, kname, kname,
                            t_ch, 2 * mults, 6 * mults, F16_PEAK, "mfma-f16", "issue", ISSUE,
                            ("forward of all layers, auto cost and the decoder's backward pass" if fused else "forward of all layers (+ auto cost)")
                            + (" in one launch per 16-row block as the transposed chain (H^T = W^T X^T: a layer's accumulators are the next "
                               "layer's operand registers); weights from the pre-split operand store; " if transposed else
                               " in one launch per 16-row block; weights from the pre-split operand store, activations split in the epilogue; ")
                            + "algorithmic bytes: the batch in, activations + deltas out as f16 fragments for the weight-gradient kernel, one "
                            "pass over the weights", 4 * Bt * dims[0][0] + 4 * Bt * sum(m for _, m in dims) * 2 + 4 * n_par))
    if cfg["model"] == "adc":
        n = len(lay.layers)
        t_ch = _time_launches(lambda: _lib.mlp_forward(eng.plan, 0, n, eng.xin, 0, Bt), launches)
        mults = Bt * sum(k * m for k, m in dims)
        recs.append(_record(name, "chain_kernel<FWD>", "chain_kernel<FWD>", t_ch, 2 * mults, 6 * mults, F16_PEAK, "mfma-f16", "issue",
                            "launch and dependent-trip latency: 256 rows are 16 row blocks on 16 of 256 CUs; the step is priced by its "
                            "launch count (launches_per_step) and the per-launch floors, DESIGN.md section 4",
                            "dense layers forward, the largest single launch of the ADC step", 4 * Bt * dims[0][0] + 4 * n_par))
    recs.sort(key=lambda r: -r["kernel_us"])
    return recs[0], sk


def launches_per_step(eng):
    """kernel launches of one step: counted by the library itself across one eager step (state saved and restored)"""
    import torch

    from encodermap_amd import _lib

    lib = _lib.load()
    saved = [t.clone() for t in eng._state_tensors()]
    torch.cuda.synchronize()
    n0 = lib.emx_launch_count()
    eng._step_body()
    n1 = lib.emx_launch_count()
    torch.cuda.synchronize()
    for t, sv in zip(eng._state_tensors(), saved):
        t.copy_(sv)
    eng._sync_derived_state()
    if eng._prefetch:
        eng._gather_current()
    return int(n1 - n0)


def agree_on_port(world: int, rank: int) -> int:
    """a free port for the probe children's rendezvous, the SAME on every rank: rank 0 asks the kernel for one and the ranks learn it
    over a short-lived gloo group (CPU only: nothing here touches a GPU; the launcher's MASTER_PORT carries that group, the RCCL
    group is initialised later on the same store).  One rank: no exchange."""
    if world <= 1:
        return _free_port()
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        box = [_free_port() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
    finally:
        dist.destroy_process_group()
    return int(box[0])


def start_watchdog(seconds: float, what: str):
    """Ranks started by a launcher directly (no self_launch above them to repeat the run): if ``what`` is not over after ``seconds``,
    say so on stderr and leave with code 124 -- the launcher then ends the other ranks, and the caller sees a failure with a reason
    instead of a hang (never a re-exec: this process has initialised its GPU).  Returns the function that calls the watchdog off."""
    import threading

    done = threading.Event()

    def watch():
        if not done.wait(seconds):
            sys.stderr.write(f"bench.py: {what} did not finish within {seconds:.0f} s (EMX_BENCH_WATCHDOG); if the all-reduce captured in "
                             "the step's graph is at fault, EMX_DP_GRAPH_ALLREDUCE=0 runs it eagerly\n")
            sys.stderr.flush()
            os._exit(124)

    threading.Thread(target=watch, daemon=True).start()
    return done.set


def probe_graph_allreduce(world: int, mode: str = "replica", timeout_s: float = 300.0, port: int = 0, p2p: bool = False,
                          rccl_captured: bool = True) -> bool:
    """N >= 1 ranks over RCCL: one child process per rank checks, BEFORE this process touches the GPU, whether the step with its
    RCCL all-reduce(s) captured in a hipGraph works with this many ranks on this machine
    (encodermap_amd/_probe_graph_allreduce.py: a synthetic [scale, all-reduce, add] graph replayed 200 times against the host-side
    recurrence, then the ENGINE's data-parallel step itself -- the benchmark's layer widths at B = 4096 on a small resident set,
    40 captured steps, finite losses, identical weights on every rank).  Passed on every rank: the data-parallel step runs as one
    graph, 32 steps per replay; failed, or hung until the timeout (the child is killed, its queues go with
    it): the all-reduce stays eager between two graphs."""
    # the children rendezvous among themselves: nothing of the launcher's agent store (TORCHELASTIC_USE_AGENT_STORE makes
    # rank 0 a CLIENT of the launcher's store, which does not listen on the probe's port)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["EMX_PROBE_PORT"] = str(port or _free_port())     # (agreed between the ranks: agree_on_port)
    env["EMX_PROBE_MODE"] = mode
    env["EMX_PROBE_P2P"] = "1" if p2p else "0"       # (the other question: the library's one-shot all-reduce over peer-mapped memory)
    env["EMX_PROBE_RCCL_CAPTURED"] = "1" if rccl_captured else "0"   # (what it is timed against: RCCL as the run would use it)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("RANK", "0")
    env.setdefault("WORLD_SIZE", str(world))
    env.setdefault("LOCAL_RANK", "0")
    try:
        proc = subprocess.Popen([sys.executable, "-m", "encodermap_amd._probe_graph_allreduce"], cwd=ROOT, env=env,
                                stdout=subprocess.DEVNULL, stderr=sys.stderr)
    except Exception:
        return False
    try:
        return proc.wait(timeout=timeout_s) == 0
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
        return False


def _free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n_gpus: int, argv) -> int:
    """``python bench.py --gpus N`` WITHOUT a launcher: this process has not touched the GPU (nothing but argparse has run), so it
    starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a child, relays the child's JSON line and
    returns its exit code (never a re-exec: a process that has initialised the GPU must not be replaced, and this one stays
    around to relay).  Should the run with the captured all-reduce fail or hang until the timeout, it is repeated ONCE with the
    eager all-reduce (EMX_DP_GRAPH_ALLREDUCE=0); the line says which form ran (config.allreduce)."""
    import signal

    timeout_s = float(os.environ.get("EMX_BENCH_LAUNCH_TIMEOUT", "1500"))
    attempts = [dict()]
    if os.environ.get("EMX_DP_GRAPH_ALLREDUCE") != "0":
        attempts.append({"EMX_DP_GRAPH_ALLREDUCE": "0", "EMX_DP_P2P": "0"})   # (the second attempt: RCCL, eagerly)
    code = 1
    for extra in attempts:
        env = dict(os.environ, **extra)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
        proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=timeout_s)
            code = proc.returncode
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)     # (the session this call created: the launcher and its ranks, nothing else)
            except Exception:
                proc.kill()
            out, _ = proc.communicate()
            code = 124
        all_lines = (out or "").splitlines()
        lines = [i for i, ln in enumerate(all_lines) if ln.startswith("{") and '"metric"' in ln]
        if code == 0 and lines:
            for i, ln in enumerate(all_lines):
                if i != lines[-1]:
                    sys.stderr.write(ln + "\n")      # (anything else the ranks printed: not on stdout, which carries ONE line)
            sys.stdout.write(all_lines[lines[-1]] + "\n")
            sys.stdout.flush()
            return 0
        sys.stderr.write(f"bench.py: the {n_gpus}-rank run ended with code {code}" + (" and no JSON line" if not lines else "") +
                         (": repeating it with the eager all-reduce\n" if extra is not attempts[-1] else "\n"))
    return code or 1


def launch_check(args) -> int:
    """the ranks of ``--gpus N`` as the launcher started them, without any GPU work: rendezvous over gloo, one all-reduce, one
    line from rank 0.  EMX_BENCH_FAIL_CAPTURED=1 (test hook) makes every rank fail unless the eager all-reduce was forced --
    which is what self_launch's second attempt does."""
    import torch
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        return 2
    forced_eager = os.environ.get("EMX_DP_GRAPH_ALLREDUCE") == "0"
    if os.environ.get("EMX_BENCH_FAIL_CAPTURED") == "1" and not forced_eager:
        return 3
    # as main() does it: the probe's port over a short-lived gloo group, then the run's own group on the same store
    port = agree_on_port(world, rank)
    call_off = start_watchdog(float(os.environ.get("EMX_BENCH_WATCHDOG", "900")), f"the {world}-rank launch check")
    if os.environ.get("EMX_BENCH_HANG") == "1":            # (test hook: a rank that never comes back -- the watchdog's case)
        time.sleep(3600)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    lo, hi = torch.tensor([float(port)]), torch.tensor([float(port)])
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    call_off()
    if rank == 0:
        print("a stray line from a rank", flush=True)      # (must not reach the relayed stdout)
        print(json.dumps({"metric": "launch_check", "value": float(t), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "config": {"ranks_seen": dist.get_world_size(), "allreduce": "eager (forced)" if forced_eager else "probe decides",
                                     "probe_port_agreed": bool(float(lo) == float(hi) == float(port)) and port > 0}}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def facade_timing(em, cfg, data, device, steps):
    """``em.EncoderMap.train()`` -- the reference's entry point (autoencoder.py:967-1029) -- on the same workload: ms per step with
    the default callbacks and with the progress read-out (one loss per ``summary_step`` steps fetched from the device)."""
    import numpy as np
    import torch

    out = {}
    B, D = cfg["kw"].get("batch_size", 256), cfg["D"]
    for label, progress in (("defaults", False), ("progress", True)):
        warm = 64
        p = em.Parameters(n_steps=warm, seed=7, **cfg["kw"])
        stderr, stdout = sys.stderr, sys.stdout
        try:
            # (the progress read-out writes a line per summary step to stderr, save() its read-only notice to stdout -- which is
            #  reserved for the JSON line)
            sys.stderr = sys.stdout = open(os.devnull, "w")
            e = em.EncoderMap(p, np.zeros((B, D), "float32"), read_only=True, device=device, progress=progress)
            e._engine.set_data(data)
            e.train()                                # captures the step's graphs
            e.p.n_steps += steps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.train()
            torch.cuda.synchronize()
            out[label] = (time.perf_counter() - t0) / steps * 1e3
        finally:
            sys.stderr, sys.stdout = stderr, stdout
        del e
    return out


# ------------------------------------------------------------------------------------------------------------------
# inference: encode() / decode() / generate() of whole trajectories (autoencoder.py:1110-1195) against the HBM roof
# ------------------------------------------------------------------------------------------------------------------
def inference_pass(eng, which, data, passes, warmup=2):
    """seconds per pass of ``which`` ("encode": the resident rows -> latent rows; "decode": as many latent rows -> output rows), HIP
    events around ``passes`` back-to-back passes on the stream the kernels run on; everything stays on the device.  Returns
    (seconds, rows, bytes in + out, launches per pass, streaming?)."""
    import torch

    from encodermap_amd import _lib

    n = data.shape[0]
    lib = _lib.load()
    if which == "encode":
        src = data
        dst = torch.empty(n, eng.L, dtype=torch.float32, device=data.device)
        fn = lambda: eng.encode_device(src, dst)
    else:
        g = torch.Generator(device=data.device)
        g.manual_seed(11)
        src = torch.randn(n, eng.L, generator=g, device=data.device)
        dst = torch.empty(n, eng.D, dtype=torch.float32, device=data.device)
        fn = lambda: eng.decode_device(src, dst)
    for _ in range(warmup):
        fn()
    c0 = lib.emx_launch_count()
    fn()
    launches = int(lib.emx_launch_count() - c0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(passes):
        fn()
    e1.record()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dst[:: max(1, n // 4096)]).all()), "non-finite rows out of the inference pass"
    nbytes = 4 * n * (src.shape[1] + dst.shape[1])
    return e0.elapsed_time(e1) * 1e-3 / passes, n, nbytes, launches, launches == 1


def _inference_traffic(name, which):
    """HBM bytes per launch of the inference kernel from the PMC passes (the C2 step file holds them: the default run ends with an
    encode and a decode pass; other configurations: profiles/r06_infer_counters_<config>_<encode|decode>.json) -- under the rule
    of _counters_file: only counters taken on these kernel sources.  Returns (bytes or None, source)."""
    if name == "C2":
        blob, path, reason = _counters_file(name)
    else:
        path = os.path.join("profiles", f"r06_infer_counters_{name.lower()}_{which}.json")
        try:
            blob, reason = json.load(open(os.path.join(ROOT, path))), None
            if blob.get("source_hash") != source_hash():
                blob, reason = None, f"refused: {path} was taken on kernel sources {blob.get('source_hash')}, this build's are {source_hash()}"
        except Exception:
            blob, reason = None, "no counters file for this configuration"
    if blob is None:
        return None, reason
    for kname, c in blob["kernels"].items():
        if "infer16" in kname and (c.get("FETCH_SIZE", 0) > c.get("WRITE_SIZE", 0)) == (which == "encode"):
            return int(c["hbm_bytes_per_launch"]), f"{path}: {kname} (static: PMC passes, kernel sources {blob.get('source_hash')})"
    return None, f"{path}: no inference kernel among its launches"


def inference_record(eng, which, data, passes, warmup=2, name="C2"):
    t, n, nbytes, launches, streaming = inference_pass(eng, which, data, passes, warmup)
    traffic, traffic_source = _inference_traffic(name, which) if streaming else (None, "the training kernels forward-only: not profiled")
    lay = eng.layout
    layers = lay.layers[: lay.n_enc] if which == "encode" else lay.layers[lay.n_enc:]
    mults = n * sum(l.n_in * l.n_out for l in layers)
    return {"rows_per_sec": n / t, "ms_per_pass": t * 1e3, "rows": int(n), "launches_per_pass": launches,
            "kernel": "infer_kernel (one streaming launch: rows in, result out, operand pieces resident in LDS)" if streaming else
                      "chain_kernel<FWD> (the training kernels forward-only: a layer wider than 128 or pieces beyond LDS)",
            "roofline": {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": nbytes / t / HBM_PEAK,
                         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": int(nbytes),
                         "mfma_f16_frac_executed": 6 * mults / t / F16_PEAK,
                         "note": "algorithmic bytes = 4 (columns in + columns out) per row; the split-f16 matrix products of the pass (three per float32 "
                                 "product) against the dense f16 peak beside it: the two roofs are within a factor of two of each other here"}}


def cpu_inference_baseline(name, which, budget_s=15.0):
    """the same pass in float32 torch-CPU (oracle/reference_ops.py's model: Dense layers, sin / cos in, atan2 out) on a bounded sample"""
    import numpy as np
    import torch

    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights
    from encodermap_amd.parameters import Parameters
    from oracle import reference_ops as R

    cfg = CONFIGS[name]
    D = cfg["D"]
    p = Parameters(**cfg["kw"])
    net = 2 * D if p.periodicity < INF else D
    lay = build_layout(p.n_neurons, p.activation_functions, net)
    w0 = unpack_weights(lay, init_parameters(lay, 0))
    model = R.SequentialOracle(D, p.n_neurons, p.activation_functions, p.periodicity, w0, dtype=torch.float32)
    rng = np.random.default_rng(2)
    rows = 100_000
    x = torch.tensor(rng.normal(0, 1, (rows, D if which == "encode" else p.n_neurons[-1])).astype(np.float32))
    fn = (lambda: model.encoder(x)) if which == "encode" else (lambda: model.decoder(x))
    cores = os.cpu_count() or torch.get_num_threads()
    torch.set_num_threads(min(cores, 32))
    with torch.no_grad():
        fn()
        t0 = time.perf_counter()
        k = 0
        while k < 3 or (time.perf_counter() - t0 < budget_s and k < 200):
            fn()
            k += 1
        dt = (time.perf_counter() - t0) / k
    return {"value": rows / dt, "unit": "rows/s", "cores": min(cores, 32), "kind": "port",
            "sample": f"{k} passes of {which} over {rows} rows x {x.shape[1]} columns (float32 torch-CPU restatement of the reference's "
                      f"model, oracle/reference_ops.py) with {min(cores, 32)} of {cores} host threads: {dt * 1e3:.1f} ms per pass",
            "host_cpu": _cpu_model(), "host_threads": cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="train", choices=["train", "encode", "decode"],
                    help="train (default): the training step; encode / decode: one pass of encode() / decode() over the resident rows per step")
    ap.add_argument("--rows", type=int, default=0, help="rows of the resident training set (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-facade", action="store_true")
    ap.add_argument("--dp-mode", default=None, choices=["replica", "rowblock"])
    ap.add_argument("--global-batch", type=int, default=-1,
                    help="--dp-mode rowblock: rows of the ONE global batch all ranks share (default: the configuration's batch; 0: that batch per rank)")
    ap.add_argument("--collectives", type=int, default=0, choices=[0, 1, 2],
                    help="--dp-mode rowblock: 1 = the rectangle form of the all-pairs pass (one collective per step), 2 = dealt triangle "
                         "(two); 0 (default): chosen per configuration by engine.DataParallel.one_collective")
    ap.add_argument("--one-rank-dp", action="store_true",
                    help="N = 1 only: run the DATA-PARALLEL code path (probe, RCCL group, all-reduce) with its one rank -- a self-test, "
                         "not a benchmark configuration")
    ap.add_argument("--launch-check", action="store_true",
                    help="self-test of the launcher plumbing WITHOUT a GPU: the ranks meet over gloo, all-reduce one number and rank 0 "
                         "prints a one-line record (tests/test_host_logic.py)")
    args = ap.parse_args()
    name = args.config
    cfg = CONFIGS[name]
    B = cfg["kw"].get("batch_size", 256)
    D = cfg["D"]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # (no launcher: become one -- before torch or anything else that could touch the GPU is even imported)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    if args.launch_check:
        raise SystemExit(launch_check(args))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")
    # C4 / C5 are DEFINED on 8 GPUs as ONE global batch: the objective-preserving row-block partition
    dp_mode = args.dp_mode or ("rowblock" if name in ("C4", "C5") else "replica")
    if cfg["model"] == "adc":
        dp_mode = "replica"      # (the ADC step runs as replicas: one collective per step)
    # self-test hooks (not benchmark configurations): EMX_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and EMX_BENCH_BACKEND=gloo
    # replaces RCCL, so that the N > 1 plumbing of this script can be run with two ranks on a one-GPU box
    one_device = os.environ.get("EMX_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("EMX_BENCH_BACKEND", "nccl")
    device = "cuda:0" if one_device else f"cuda:{local_rank}"
    multi = world > 1 or (args.one_rank_dp and world == 1)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]   # (RCCL's version banner goes to stdout, which is reserved for the JSON line)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
    # The RCCL all-reduce(s) of the step are CAPTURED inside the step's graph (one graph per step, 32 steps per replay) whenever the
    # child-process probe -- the synthetic graph AND the engine's own data-parallel step, run by one child per rank before this
    # process touches its GPU -- passes on every rank; otherwise they run eagerly between the step's graphs.
    # EMX_DP_GRAPH_ALLREDUCE=0 forces the eager form (no probe).  The line records which form ran and the probe's result.
    probe = None
    graph_allreduce = False
    if multi and backend == "nccl" and os.environ.get("EMX_DP_GRAPH_ALLREDUCE") != "0":
        probe = probe_graph_allreduce(world, dp_mode, port=agree_on_port(world, rank))
        graph_allreduce = probe
    # The collectives as the library's ONE-SHOT all-reduce over peer-mapped memory (include/emx.h: emx_p2p_*; one launch, every rank reads
    # its peers' buffers over xGMI and adds them in rank order) instead of RCCL's: where a second child-process probe -- the all-reduce
    # itself against RCCL on changing data AND faster than RCCL's (200 timed calls each), then the engine's step with it, 40 captured
    # steps against the RCCL run -- passes on every rank.  EMX_DP_P2P=0: never; EMX_DP_P2P=1: without asking.  Ranks of one node only (which is what this script launches).
    p2p_probe = None
    use_p2p = os.environ.get("EMX_DP_P2P") == "1" and multi
    if multi and world > 1 and backend == "nccl" and os.environ.get("EMX_DP_P2P") is None:
        p2p_probe = probe_graph_allreduce(world, dp_mode, port=agree_on_port(world, rank), p2p=True, rccl_captured=bool(graph_allreduce))
        use_p2p = p2p_probe
    torch.cuda.set_device(device)
    import torch.distributed as dist

    if multi:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import numpy as np

    import encodermap_amd as em
    from encodermap_amd.engine import DataParallel, SequentialEngine

    if multi:
        # every rank must take the same path: the captured all-reduce only if the probe passed on ALL ranks
        flag = torch.tensor([1.0 if graph_allreduce else 0.0], device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        graph_allreduce = bool(float(flag) == 1.0)
        SequentialEngine._dp_graph_allreduce = graph_allreduce
        flag = torch.tensor([1.0 if use_p2p else 0.0], device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        use_p2p = bool(float(flag) == 1.0)

    rowblock = dp_mode == "rowblock" and world > 1
    gb = B if args.global_batch < 0 else args.global_batch
    batch = (gb or B * world) if rowblock else B     # rows per engine step: the rank's own batch, or the shared one
    dp = DataParallel(rank, world, dp_mode, collectives=args.collectives or None, p2p=use_p2p) if world > 1 else None
    if multi and world == 1:
        class _OneRank(DataParallel):      # the data-parallel path with the one rank a single GPU allows
            active = True

        dp = _OneRank(0, 1, dp_mode, collectives=args.collectives or None, p2p=use_p2p)
    n_rows = args.rows or cfg["rows"]
    kw = dict(cfg["kw"], batch_size=batch)
    if cfg["model"] == "adc":
        p = em.ADCParameters(n_steps=args.steps + args.warmup, seed=7, **kw)
        e = em.AngleDihedralCartesianEncoderMap(make_backbone(cfg, n_rows), p, read_only=True, device=device, dp=dp)
        eng = e._engine
        data = None
    else:
        p = em.Parameters(n_steps=args.steps + args.warmup, seed=7, **kw)
        data = make_data(cfg, device, n_rows)
        e = em.EncoderMap(p, np.zeros((batch, D), "float32"), read_only=True, device=device, dp=dp)
        eng = e._engine
        eng.set_data(data)                      # resident in HBM before the timed region (C2: 512 MB, C4: 4 GB)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode != "train":
        if cfg["model"] != "encodermap":
            raise SystemExit("--mode encode / decode: the EncoderMap configurations (C1, C2, C4, C5)")
        # a "step" = one pass over the rank's resident rows (N > 1: every rank its own replica of the set -- rows, not a model, are what
        # inference shards over: no collective); warm-up passes, then K timed passes bracketed as the contract says
        barrier()
        t0 = time.perf_counter()
        rec = inference_record(eng, args.mode, data, max(args.steps, 1), warmup=max(args.warmup, 1), name=name)
        barrier()
        dt = time.perf_counter() - t0
        t_pass = rec["ms_per_pass"] * 1e-3
        if multi:
            tt = torch.tensor([t_pass], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_pass = float(tt.item())
        if rank == 0:
            n = data.shape[0]
            out = {"metric": f"{args.mode}_rows_per_sec", "value": n * world / t_pass, "unit": "rows/s", "n_gpus": world, "steps": max(args.steps, 1),
                   "warmup": max(args.warmup, 1), "ms_per_step": t_pass * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": DTYPES[cfg["kind"]].split(";")[0].replace("all-pairs pass", "").strip(), "data": "synthetic",
                   "config": {"workload": f"{name}: em.EncoderMap.{args.mode}() of {n} rows x {D if args.mode == 'encode' else eng.L} columns resident in HBM "
                                          f"-> {eng.L if args.mode == 'encode' else D} columns (network {cfg['workload'].split(':')[0]}'s), result left on the device",
                              "name": name, "rows_resident": int(n), "features": D, "parallelism": f"replicas{world}" if world > 1 else "single",
                              "commit": git_head(), "wall_s_including_warmup": dt},
                   "roofline": dict(rec["roofline"], kernel=rec["kernel"], kernel_us=rec["ms_per_pass"] * 1e3, launches_per_pass=rec["launches_per_pass"])}
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_inference_baseline(name, args.mode)
            print(json.dumps(out), flush=True)
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

    # (multi-rank runs: a hang -- of the captured all-reduce, of a rank that lost its peers -- ends as a failure with a reason)
    call_off = start_watchdog(float(os.environ.get("EMX_BENCH_WATCHDOG", "900")), f"the {world}-rank training run") if multi else (lambda: None)
    eng.run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    eng.run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    call_off()
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    losses = eng.losses()
    assert all(l == l and abs(l) != float("inf") for l in losses), "non-finite loss in the benchmark run"

    if rank == 0:
        global_batch = batch if rowblock else B * world
        if rowblock:
            one = not eng._dealt()
            objective = (f"one sketch-map cost over one global batch of {batch} rows (the single-process objective at that batch size); "
                         f"dense layers on {world} runs of 16-row blocks, " +
                         ("the all-pairs pass as the rectangle (own rows) x (all rows): ONE collective per step" if one else
                          "the upper-triangle tiles of the all-pairs pass dealt round-robin: TWO collectives per step"))
            scaling = "strong" if batch == B else "weak"
        else:
            objective = (f"mean of {world} independent costs over {B} rows each (gradients averaged)" if world > 1
                         else f"one cost over one batch of {B} rows")
            scaling = "weak"
        rccl = None
        if multi and backend == "nccl":
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
        out = {
            "metric": "train_samples_per_sec", "value": global_batch * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "steps_per_sec": args.steps / dt, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": DTYPES[cfg["kind"]], "data": "synthetic",
            "config": {"workload": cfg["workload"] + (f" [one batch of {batch} rows over all GPUs]" if rowblock else
                                                      (f" [batch {B} per GPU]" if world > 1 else "")),
                       "name": name, "rows_resident": int(n_rows), "global_batch": global_batch, "features": D,
                       "parallelism": f"dp{world}-{dp_mode}" if world > 1 else "single", "objective": objective,
                       "collectives_per_step": (None if not multi else (1 if (not rowblock or not eng._dealt()) else 2)),
                       "ranks_seen": dist.get_world_size() if multi else 1, "backend": backend if multi else None, "rccl_version": rccl,
                       "allreduce": (("one-shot over peer-mapped memory (emx_p2p_allreduce), captured in the step graph" if use_p2p else
                                      "captured in the step graph" if graph_allreduce else "eager between the step's graphs") if multi else None),
                       "graph_allreduce_probe": probe, "p2p_allreduce_probe": p2p_probe,
                       **({"self_test": f"backend {backend}, one device: {one_device}"} if (one_device or backend != "nccl") else {}),
                       "hip_graph": True, "commit": git_head(), "kernel_sources": source_hash()},
            "loss_first_last": [float(losses[0]), float(losses[-1])],
        }
        out["roofline"], ap_rec = kernel_rooflines(name, eng, max(min(args.steps, 200), 50))
        if ap_rec is not None:
            out["roofline_all_pairs"] = ap_rec
        live = {r["kernel"].split(" ")[0].split("<")[0]: r["kernel_us"] for r in (out["roofline"], ap_rec) if r}
        out["roofline_step"], out["roofline_step_source"] = roofline_step(name, eng, live)
        if not multi:
            out["launches_per_step"] = launches_per_step(eng)
        if world == 1 and not multi and cfg["model"] == "encodermap" and data is not None:
            out["inference"] = {w: inference_record(eng, w, data, 10, name=name) for w in ("encode", "decode")}
        if world == 1 and not multi and not args.no_facade and name == "C2":
            f = facade_timing(em, cfg, data, device, max(args.steps, 2000))
            out["facade_ms_per_step"] = f["defaults"]
            out["facade_progress_ms_per_step"] = f["progress"]
        if world == 1 and not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(name)
        # RCCL writes its version banner to the C-level stdout, which is flushed at exit -- behind this line unless it is
        # flushed first; the JSON line must be the last thing on stdout
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
