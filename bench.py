#!/usr/bin/env python3
"""bench.py — image-pairs/s of the RePoseD RANSAC hot path on MI355X (BASELINE.json metric).

A step = one pass of estimate_monodepth_relative_pose over one batch of synthetic image pairs that already sit in
HBM: calibrated 3-point solver (P3P path, shift off), 2000 correspondences per pair, max_iterations = min_iterations
= 10000 (BASELINE.json configs[1]; /root/reference/make_video.py:192-194), 50 % outliers, 0.5 px / 2 % noise,
1024 pairs per GPU.  Pairs shard across ranks with no data-path collective (weak scaling); for N > 1 the step ends
with one device-side RCCL all_gather of the fixed-size result records.  --total-pairs P switches to BASELINE.json
configs[4]: P pairs in total, ceil(P/N) contiguous pairs per rank, each rank generating only its own block (strong scaling).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python -m torch.distributed.run ... bench.py --gpus 8 --total-pairs 100000
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_EVAL = 32.0      # x1,x2 as four fp64 per (model x correspondence) Sampson evaluation

WORKLOADS = {
    # name: (kind, n, iters, outliers, estimate_shift, random_focal)
    "calib_p3p_n2000_i10k": (0, 2000, 10000, 0.5, False, None),
    "calib_shift_n2000_i10k": (0, 2000, 10000, 0.5, True, None),
    "shared_n2000_i10k": (1, 2000, 10000, 0.5, False, "shared"),
    "varying_n5000_i10k": (2, 5000, 10000, 0.5, False, "varying"),  # + the shift FLAG (SHIFT_FLAG_ONLY): BASELINE configs[3]
    # SURVEY.md §8(d) C2 also asks for the outlier-free shape
    "calib_p3p_n2000_i10k_clean": (0, 2000, 10000, 0.0, False, None),
    "calib_shift_n2000_i10k_clean": (0, 2000, 10000, 0.0, True, None),
    # non-monodepth baselines on the same kernels: 5-point relative pose, 7-point fundamental matrix
    "relpose_5pt_n2000_i10k": (3, 2000, 10000, 0.5, False, None),
    "fundamental_7pt_n2000_i10k": (5, 2000, 10000, 0.5, False, None),
    "shared_6pt_n2000_i10k": (4, 2000, 10000, 0.5, False, "shared"),
}

# BASELINE.json configs[3] sets monodepth_estimate_shift=True on the varying-focal estimator.  The reference ignores the flag there
# (include/mdrp.h, tests/golden/estimate_wide.npz `varying_shiftflag`); it is passed all the same, the synthetic depths carry no shift.
SHIFT_FLAG_ONLY = {"varying_n5000_i10k"}


def make_inputs(workload, first_index, batch):
    from mdrp_amd import synth
    kind, n, iters, of, es, rf = WORKLOADS[workload]
    b = synth.make_batch(first_index, batch, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf,
                         shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
    return b


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_worker(args):
    """one host core: `count` pairs of the workload starting at `first` through the CPU oracle; returns busy seconds"""
    workload, first, count = args
    from oracle import pyorc as po
    kind, n, iters, of, es, rf = WORKLOADS[workload]
    b = make_inputs(workload, first, count)
    ro = po.ransac_opt(max_iterations=iters, min_iterations=iters, max_epipolar_error=2.0, max_reproj_error=16.0,
                       estimate_shift=es or workload in SHIFT_FLAG_ONLY)
    bo = po.bundle_opt(loss_type=4)
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    po.lib()
    t0 = time.perf_counter()
    for i in range(count):
        if kind >= 3:
            po.estimate_classic(kind, b["x1"][i], b["x2"][i], ro, bo, cam, cam, pp=(0.0, 0.0))
        else:
            po.estimate(kind, b["x1"][i], b["x2"][i], b["d1"][i], b["d2"][i], ro, bo, cam if kind == 0 else None, cam if kind == 0 else None)
    return time.perf_counter() - t0


def cpu_baseline(workload, pairs, multi=True):
    """the CPU oracle (oracle/*.c — our port of the reference algorithm, pinned against the reference binary) on the
    same workload: single thread on the first `pairs` pairs (the reported baseline), then up to 32
    worker processes with their own pairs (like the reference's `eval.py -nw`).  Must run BEFORE the process touches
    the GPU: the workers are forked."""
    import multiprocessing as mp
    dt = _cpu_worker((workload, 0, pairs))
    cores = os.cpu_count() or 1
    out = {"value": pairs / dt, "unit": "image-pairs/s", "cores": 1, "kind": "port",
           "sample": f"{pairs} pairs of {workload} (same generator, indices 0..{pairs - 1}), {dt:.1f} s wall, 1 thread",
           "cpu_model": _cpu_model(), "host_cores": cores}
    if cores > 1 and multi:
        # bounded: at most 32 workers x 6 pairs (the GPU boxes advertise 256 CPUs but schedule ~8 cores' worth of time to
        # the job: 256 workers x 16 pairs took 54 s for 75 pairs/s)
        workers = min(cores, 32)
        per = max(2, min(6, pairs // 6))
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(workers) as pool:
            busy = pool.map(_cpu_worker, [(workload, w * per, per) for w in range(workers)], chunksize=1)
        dta = time.perf_counter() - t0
        out["multi_process"] = {"value": workers * per / max(busy), "unit": "image-pairs/s", "cores": workers,
                                "sample": f"{workers} worker processes x {per} pairs, slowest worker {max(busy):.1f} s busy "
                                          f"({dta:.1f} s wall with process start-up and input generation)"}
    return out


def pmc_profile(workload, batch, kernel):
    """Per-launch PMC numbers of `kernel` from the committed rocprofv3 passes of this round (profiles/r*_pmc_*.json, written
    by tools/pmc_json.py on the GPU box): HBM bytes (FETCH_SIZE x2 + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md) and the SQ issue counters, when they were taken on this workload and batch; else {}."""
    import glob
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload or d.get("pairs_per_gpu") != batch:
            continue
        for k, v in d.get("kernels", {}).items():
            if k.split("<")[0] == "mdrp::" + kernel:
                out.update(v)
                out["source"] = os.path.basename(f)
    return out


MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (~2.5 PFLOP/s; 512 MAC per cycle and SIMD)
FLOP_PER_MFMA_EVAL = 64.0       # one (model x correspondence) evaluation in k_count = 32 bf16 MACs of the K = 32 contraction
# fp64 vector peak.  MI355X_MICROARCH.md states the FP32 vector peak, 157.3 TFLOP/s (256 CUs x 4 SIMDs x 16 lanes x 2 packed x 2 flop
# x 2.4 GHz); the guide has no fp64 line.  CDNA4's published fp64 vector rate is half of that (one 64-bit FMA per lane and cycle, no
# packing): 78.6 TFLOP/s.  tools/ubench/fp64_peak.hip measured 69 TFLOP/s of it at 8 waves/SIMD.
FP32_VALU_PEAK_TFLOPS_GUIDE = 157.3
FP64_VALU_PEAK_TFLOPS = FP32_VALU_PEAK_TFLOPS_GUIDE / 2.0
# fp64 flop per correspondence of the LM sweeps, (cost sweep: residuals | normal-equation sweep: residuals + Jacobians + J'J),
# read off the ISA of this build by tools/lm_flops.py (FMA-class instructions 2 flop, every other fp64 VALU instruction 1);
# keyed by (estimator kind, monodepth_estimate_shift on the calibrated estimator)
LM_FLOP = {(0, False): (179.0, 716.0), (0, True): (179.0, 826.0), (1, False): (187.0, 883.0), (2, False): (187.0, 973.0)}
# BASELINE.json configs[2] and configs[3]: measured after the headline on the default N = 1 run and reported under `configs`
EXTRA_CONFIGS = (("shared_n2000_i10k", 24), ("varying_n5000_i10k", 16))  # (workload, pairs of its CPU baseline)
ESTIMATOR_NAMES = ["calibrated", "shared_focal", "varying_focal", "relative_pose_5pt", "shared_focal_6pt", "fundamental_7pt"]


def port_vs_reference():
    """pairs/s of the CPU port (oracle/*.c, what `cpu_baseline` times) relative to the reference's own PoseLib binary on the same
    core and inputs, per workload: measured in the build container by tests/tools/port_vs_reference.py (the reference binary cannot
    travel to the GPU box) and committed as profiles/r04_port_vs_reference.json; {} if that file is missing"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r04_port_vs_reference.json")))
    except Exception:
        return {}


class Workload:
    """one synthetic workload resident in HBM on this rank + the handle that runs it"""

    def __init__(self, name, lo, hi, per, local_rank, dev):
        import torch
        from mdrp_amd import _capi
        self.name = name
        self.kind, self.n, self.iters, self.of, self.es, self.rf = WORKLOADS[name]
        self.B, self.per = hi - lo, per
        B1 = max(self.B, 1)
        self.b = make_inputs(name, lo, B1)
        self.x1 = torch.from_numpy(self.b["x1"]).to(dev); self.x2 = torch.from_numpy(self.b["x2"]).to(dev)
        self.d1 = torch.from_numpy(self.b["d1"]).to(dev); self.d2 = torch.from_numpy(self.b["d2"]).to(dev)
        self.mask = torch.zeros((B1, self.n), dtype=torch.uint8, device=dev)
        self.cams = np.zeros(B1, dtype=_capi.CAMERA_DTYPE)
        self.cams["params"][:, 0] = 800.0
        if self.kind == 4:
            self.cams["params"][:] = 0.0  # MDRP_SHARED_6PT: cam1 carries the principal point; the synthetic pixels are centred
        self.ro = _capi.ransac_opt_from_dict({"max_iterations": self.iters, "min_iterations": self.iters, "max_epipolar_error": 2.0,
                                              "max_reproj_error": 16.0, "monodepth_estimate_shift": self.es or name in SHIFT_FLAG_ONLY})
        self.bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
        self.with_cams = self.kind in (0, 3, 4)  # kind 4: the principal point travels in cam1 (0, 0 here)
        self.classic = self.kind >= 3
        self.h = _capi.Handle(local_rank)  # its own stream; every kernel class is timed with HIP events recorded on the stream it runs on
        torch.cuda.synchronize(dev)  # the inputs were written on torch's stream, the handle runs on its own: order them once

    def launch(self, h=None, mask=None):
        """queue one pass of the hot path over the resident batch (mdrp_estimate_batch_async on device pointers)"""
        h = h or self.h
        c = self.cams if self.with_cams else None
        h.estimate_batch_device(self.kind, self.x1.data_ptr(), self.x2.data_ptr(), 0 if self.classic else self.d1.data_ptr(),
                                0 if self.classic else self.d2.data_ptr(), self.B, self.n, self.ro, self.bo, None, c, c,
                                (mask if mask is not None else self.mask).data_ptr())

    def close(self):
        self.h.close()


def timed_steps(step, barrier, steps, warmup, stats_of=None):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; returns (seconds, last result, summed stats)"""
    res = None
    for _ in range(warmup):
        step()
    acc = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
        if stats_of is not None:
            for k, v in stats_of().items():
                acc[k] = acc.get(k, 0) + v
    barrier()
    return time.perf_counter() - t0, res, acc


def rooflines_of(w, acc, steps, dt):
    """per-kernel time of the step (HIP events around every launch, on the stream it runs on) and the rooflines that have a work model.
    k_count: executed MFMA work (64 flop per evaluation, 16 x 16 tiles, padding included, counted on the device) against the dense bf16
    MFMA peak.  The LM kernels (k_lo, k_final): fp64 flop = correspondences their sweeps evaluated (counted on the device) x the flop
    per correspondence of LM_FLOP, against the fp64 vector peak.
    With the fused tail (the default where the run's end is known) k_final's events bracket its wait for the LO queue as well, and its
    interval overlaps the last k_lo launch: `k_lo+k_final` gives the two as ONE interval (union on the device timeline is not available
    from events on two streams; the sum is an upper bound) and the dominant kernel is never chosen from k_final's overlapped time."""
    kind, B, steps = w.kind, w.B, max(steps, 1)
    kern = {"k_count": acc.get("count_ms", 0.0), "k_score": acc.get("sweep_ms", 0.0), "k_bound": acc.get("bound_ms", 0.0),
            "k_solve": acc.get("solve_ms", 0.0), "k_lo": acc.get("lo_ms", 0.0), "k_final": acc.get("final_ms", 0.0)}
    launches = {"k_count": acc.get("count_launches", 0), "k_score": acc.get("sweep_launches", 0), "k_bound": acc.get("bound_launches", 0),
                "k_solve": acc.get("solve_launches", 0), "k_lo": acc.get("lo_launches", 0), "k_final": acc.get("final_launches", 0)}
    lm_key = (kind, bool(w.es) and kind == 0)
    rooflines = {}
    cl = max(launches["k_count"], 1)
    count_s = kern["k_count"] / 1e3
    flop = FLOP_PER_MFMA_EVAL * acc.get("evals_mfma", 0)
    achieved = flop / count_s / 1e12 if count_s > 0 else 0.0
    prof = pmc_profile(w.name, B, "k_count")
    hbm = prof.get("hbm_bytes_corrected")
    avg_launch_s = count_s / cl
    rooflines["k_count"] = {
        "bound": "mfma", "kernel": "k_count (v_mfma_f32_16x16x32_bf16)", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": hbm, "traffic_unit": "HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE)",
        "traffic_source": prof.get("source"), "avg_launch_ms": 1e3 * avg_launch_s, "launches_per_step": cl / steps,
        "executed_evals_per_step": acc.get("evals_mfma", 0) / steps, "share_of_step": count_s / dt,
        "hbm_GBs": (hbm / avg_launch_s / 1e9) if (hbm and avg_launch_s > 0) else None,
        "hbm_frac": (hbm / avg_launch_s / 1e9 / HBM_PEAK_GBS) if (hbm and avg_launch_s > 0) else None,
        "mfma_busy_frac_pmc": prof.get("mfma_busy_frac"), "valu_active_per_simd_cycle_pmc": prof.get("valu_active_per_simd_cycle")}
    if lm_key in LM_FLOP:
        fc, fa = LM_FLOP[lm_key]
        for name, ce, ae in (("k_lo", "lm_cost_evals", "lm_accum_evals"), ("k_final", "final_cost_evals", "final_accum_evals")):
            ksec, nl = kern[name] / 1e3, max(launches[name], 1)
            fl = fc * acc.get(ce, 0) + fa * acc.get(ae, 0)
            ach = fl / ksec / 1e12 if ksec > 0 else 0.0
            pr = pmc_profile(w.name, B, name)
            hb = pr.get("hbm_bytes_corrected")
            rooflines[name] = {
                "bound": "fp64", "kernel": f"{name} (LM refinements: v_fma_f64)", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_VALU_PEAK_TFLOPS, "traffic": hb, "traffic_unit": "HBM bytes per launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE)",
                "traffic_source": pr.get("source"), "avg_launch_ms": 1e3 * ksec / nl, "launches_per_step": nl / steps, "share_of_step": ksec / dt,
                "fp64_flop_per_step": fl / steps, "flop_per_correspondence": {"cost_sweep": fc, "normal_equations": fa},
                "correspondences_per_step": {"cost_sweep": acc.get(ce, 0) / steps, "normal_equations": acc.get(ae, 0) / steps},
                "hbm_GBs": (hb / (ksec / nl) / 1e9) if (hb and ksec > 0) else None,
                "hbm_frac": (hb / (ksec / nl) / 1e9 / HBM_PEAK_GBS) if (hb and ksec > 0) else None,
                "valu_active_per_simd_cycle_pmc": pr.get("valu_active_per_simd_cycle"), "mean_waves_per_simd_pmc": pr.get("mean_waves_per_simd"),
                "peak_note": "fp64 vector peak = half the guide's FP32 vector peak (157.3 TFLOP/s); the guide has no fp64 line"}
        if acc.get("fuse_timeouts", 0):
            rooflines["k_final"]["note"] = "fused tail hit a bounded-wait timeout in this run: k_final's interval includes give-up waits"
    # k_final's event interval overlaps the last k_lo launch when the tail is fused and includes its wait for ready pairs:
    # the dominant kernel is chosen among the others
    cand = {k: v for k, v in kern.items() if k != "k_final"}
    top = max(cand, key=lambda k: cand[k])
    if top not in rooflines and any(kern.values()):
        # k_solve / k_score / k_bound on top (the 5-/6-point baselines, the outlier-free shape): no device-side flop counter exists for
        # these (data-dependent root finders, early exits) - the entry carries the event time and the PMC issue fraction, no `achieved`
        tsec, tl = kern[top] / 1e3, max(launches[top], 1)
        pt = pmc_profile(w.name, B, {"k_solve": "kc_solve" if w.classic else "k_solve"}.get(top, top))
        rooflines[top] = {
            "bound": "fp64" if top != "k_bound" else "fp32", "kernel": top, "achieved": None, "peak": FP64_VALU_PEAK_TFLOPS if top != "k_bound" else None,
            "unit": "TFLOP/s", "frac": None, "traffic": pt.get("hbm_bytes_corrected"), "traffic_source": pt.get("source"),
            "avg_launch_ms": 1e3 * tsec / tl, "launches_per_step": tl / steps, "share_of_step": tsec / dt,
            "valu_active_per_simd_cycle_pmc": pt.get("valu_active_per_simd_cycle"), "mean_waves_per_simd_pmc": pt.get("mean_waves_per_simd"),
            "note": "no flop model for this kernel: time and PMC issue fraction only; the LM and MFMA rooflines are in roofline_lm / roofline_count"}
    dominant = max((k for k in cand if k in rooflines), key=lambda k: cand[k]) if any(kern.values()) else "k_count"
    return kern, rooflines, dominant, top


def hbm_algorithmic(acc, steps, dt):
    """SURVEY.md 8(d)'s own figure: every (model x correspondence) Sampson evaluation of the CPU loop reads x1, x2 = 32 bytes.
    frac > 1 says what it has to say: the timed kernels do NOT stream 32 B per evaluation from HBM - a pair's correspondences stay
    in LDS / L2 while thousands of hypotheses sweep them, and 98 % of the evaluations are retired by conservative bf16-MFMA / fp32
    bounds instead of being evaluated in fp64."""
    ev = acc.get("evals_algorithmic", 0) / max(steps, 1)
    by = BYTES_PER_EVAL * ev
    gbs = by / (dt / max(steps, 1)) / 1e9 if dt > 0 else 0.0
    return {"bound": "hbm", "evals": ev, "bytes_per_eval": BYTES_PER_EVAL, "bytes": by, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS,
            "note": "algorithmic bytes of the CPU loop over the whole step time; > 1 because correspondences are reused from LDS / L2 and "
                    "hypotheses are retired by conservative bounds (evals executed: work.evals_*_per_step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="image pairs per GPU per step (weak scaling)")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="BASELINE configs[4]: this many pairs in total, ceil(P/G) contiguous pairs per rank (strong scaling); "
                         "e.g. --total-pairs 100000 --gpus 8, or --total-pairs 12500 --gpus 1 for one rank's share")
    ap.add_argument("--workload", default="calib_p3p_n2000_i10k", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-pairs", type=int, default=96, help="pairs timed on the CPU baseline (0 = skip)")
    ap.add_argument("--inflight", type=int, default=2, help="extra measurement at N = 1: the same steps with this many in flight "
                                                              "(one handle + host thread each, mdrp_amd.pipeline); 1 = skip.  Never `value`.")
    ap.add_argument("--host-steps", type=int, default=2, help="extra steps through the host-buffer (PCIe-inclusive) entry point, N = 1 only (0 = skip)")
    ap.add_argument("--extra-configs", type=int, default=-1,
                    help="after the headline, also measure BASELINE configs[2] and [3] (shared focal; varying focal N = 5000 with the shift flag) "
                         "with this many steps each and report them under `configs` (N = 1, default workload only; default 5; 0 = skip)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    default_run = world_env == 1 and args.workload == "calib_p3p_n2000_i10k" and args.total_pairs == 0 and args.batch == 1024
    extra_steps = args.extra_configs if args.extra_configs >= 0 else (5 if default_run else 0)
    if not default_run:
        extra_steps = 0
    cpu_line, cpu_extra = None, {}
    if world_env == 1 and args.cpu_pairs > 0:  # forks workers: before anything initialises the GPU
        cpu_line = cpu_baseline(args.workload, args.cpu_pairs)
        if extra_steps > 0:
            for wname, cp in EXTRA_CONFIGS:
                cpu_extra[wname] = cpu_baseline(wname, cp, multi=False)
    pvr = port_vs_reference()
    for name, cl in [(args.workload, cpu_line)] + list(cpu_extra.items()):
        if cl is not None and name in pvr.get("workloads", {}):
            cl["port_vs_reference_binary"] = pvr["workloads"][name]

    import torch
    import torch.distributed as dist
    from mdrp_amd import _capi
    from mdrp_amd import dist as mdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    strong = args.total_pairs > 0
    if strong:  # contiguous block of ceil(P/G) pairs per rank, generated by the rank that owns it (never replicated)
        total = args.total_pairs
        lo, hi, per = mdist.shard_bounds(total, rank, world)
    else:
        total = args.batch * world
        lo, hi, per = rank * args.batch, (rank + 1) * args.batch, args.batch
    w = Workload(args.workload, lo, hi, per, local_rank, dev)
    kind, n, iters, of, es, B, b, h = w.kind, w.n, w.iters, w.of, w.es, w.B, w.b, w.h
    rec_local = torch.zeros((per, mdist.RECORD_BYTES), dtype=torch.uint8, device=dev)  # this rank's block of the gather
    rec_all = torch.empty((world * per, mdist.RECORD_BYTES), dtype=torch.uint8, device=dev) if world > 1 else None

    def step():
        if B > 0:
            w.launch()
        if world > 1:  # final gather of the pose records over RCCL/xGMI, device to device: 136 B per pair
            if B > 0:
                h.copy_results_device(rec_local.data_ptr(), B)
            dist.all_gather_into_tensor(rec_all, rec_local)
            return None
        return h.fetch_results(B)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt, res, acc = timed_steps(step, barrier, args.steps, args.warmup, h.last_stats if B > 0 else None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        res = h.fetch_results(B) if B > 0 else np.zeros(0, dtype=_capi.RESULT_DTYPE)
        gathered = mdist._unpad(rec_all.cpu().numpy(), total, world, per)
        assert gathered.shape[0] == total

    host_rate = None
    if world == 1 and args.host_steps > 0 and B > 0:  # the same step through mdrp_estimate_batch with HOST buffers: H2D of the
        xs = [b["x1"], b["x2"], b["d1"], b["d2"]]      # correspondences (48 B each) and D2H of records + masks inside the timed region
        c = w.cams if w.with_cams else None
        h.estimate_batch(kind, *xs, w.ro, w.bo, None, c, c)
        th = time.perf_counter()
        for _ in range(args.host_steps):
            h.estimate_batch(kind, *xs, w.ro, w.bo, None, c, c)
        host_rate = B * args.host_steps / (time.perf_counter() - th)

    pipelined = None
    if world == 1 and args.inflight > 1 and B > 0:  # consecutive steps with several in flight: an EXTRA key, `value` stays sequential
        import threading
        L = args.inflight
        hs = [_capi.Handle(local_rank) for _ in range(L)]
        masks = [torch.zeros((B, n), dtype=torch.uint8, device=dev) for _ in range(L)]
        torch.cuda.synchronize(dev)

        def lane(i, steps_):
            for _ in range(steps_):
                w.launch(hs[i], masks[i])
                hs[i].fetch_results(B)
        for i in range(L):
            lane(i, 1)
        per_lane = max(1, args.steps // L)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        th = [threading.Thread(target=lane, args=(i, per_lane)) for i in range(L)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dtp = time.perf_counter() - tp
        pipelined = {"in_flight": L, "steps": per_lane * L, "value": B * per_lane * L / dtp, "unit": "image-pairs/s", "ms_per_step": 1e3 * dtp / (per_lane * L),
                     "note": "the same steps, this many in flight at a time (one handle, stream set and host thread each: mdrp_amd.pipeline.BatchPipeline); "
                             "the next step's solver and sweeps fill the SIMDs the LM phases leave idle.  Not `value`."}
        for hh in hs:
            hh.close()

    # ---- BASELINE configs[2] and configs[3] on the same line (N = 1 default run): 1024 pairs each, same timing protocol
    extra = []
    if extra_steps > 0 and rank == 0:
        w.close()
        del w.x1, w.x2, w.d1, w.d2, w.mask
        for wname, _cp in EXTRA_CONFIGS:
            we = Workload(wname, 0, 1024, 1024, local_rank, dev)

            def step_e():
                we.launch()
                return we.h.fetch_results(we.B)
            dte, rese, acce = timed_steps(step_e, barrier, extra_steps, 1, we.h.last_stats)
            kern_e, roof_e, dom_e, top_e = rooflines_of(we, acce, extra_steps, dte)
            r = roof_e[dom_e]
            entry = {"workload": wname, "value": we.B * extra_steps / dte, "unit": "image-pairs/s", "ms_per_step": 1e3 * dte / extra_steps,
                     "steps": extra_steps, "warmup": 1, "pairs": we.B, "correspondences": we.n, "ransac_iterations": we.iters,
                     "estimator": ESTIMATOR_NAMES[we.kind], "monodepth_estimate_shift_flag": bool(we.es or wname in SHIFT_FLAG_ONLY), "outlier_fraction": we.of,
                     "roofline": {"kernel": r.get("kernel"), "bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
                                  "frac": r.get("frac"), "traffic": r.get("traffic"), "avg_launch_ms": r.get("avg_launch_ms"),
                                  "launches_per_step": r.get("launches_per_step")},
                     "roofline_hbm_algorithmic": hbm_algorithmic(acce, extra_steps, dte),
                     "kernel_ms_per_step": {k: v / extra_steps for k, v in kern_e.items()},
                     "lo_plus_final_ms_per_step": (kern_e["k_lo"] + kern_e["k_final"]) / extra_steps,
                     "mean_inlier_ratio": float(np.mean(rese["num_inliers"] / we.n))}
            if wname in cpu_extra:
                entry["cpu_baseline"] = cpu_extra[wname]
                entry["speedup_vs_cpu_1thread"] = entry["value"] / cpu_extra[wname]["value"]
            extra.append(entry)
            we.close()
            del we

    if rank == 0:
        pairs = total * args.steps
        value = pairs / dt
        from mdrp_amd import synth
        from mdrp_amd.poselib import _quat_to_R
        R_err = None if kind == 5 else float(np.median([synth.rotation_error_deg(g["R"], _quat_to_R(r["model"]["q"])) for r, g in zip(res[:64], b["gt"][:64])]))
        steps = max(args.steps, 1)
        kern, rooflines, dominant, top = rooflines_of(w, acc, args.steps, dt)
        line = {
            "metric": "image-pairs/sec (2000 corrs, 10k RANSAC iters)", "value": value, "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "pairs_per_gpu": per, "total_pairs": total, "correspondences": n, "ransac_iterations": iters,
                       "outlier_fraction": of, "estimator": ESTIMATOR_NAMES[kind],
                       "monodepth_estimate_shift": bool(es or args.workload in SHIFT_FLAG_ONLY),
                       "parallelism": f"{'ceil(P/G) contiguous pairs per rank' if strong else 'fixed pairs per rank'} x{world}, device-side all_gather of the 136-B records"},
            # the kernel with the largest event-timed share of the step (k_final excluded: fused, its interval overlaps k_lo); the MFMA figure of k_count stays beside it
            "roofline": rooflines[dominant],
            "roofline_count": rooflines["k_count"],
            "roofline_lm": rooflines.get("k_lo"),
            "roofline_final": rooflines.get("k_final"),
            # SURVEY.md 8(d)'s figure, explicit: 32 algorithmic bytes per (model x correspondence) evaluation of the CPU loop
            "roofline_hbm_algorithmic": hbm_algorithmic(acc, args.steps, dt),
            "kernel_ms_per_step": {k: v / steps for k, v in kern.items()},
            "lo_plus_final_ms_per_step": (kern["k_lo"] + kern["k_final"]) / steps,
            "fuse_timeouts": acc.get("fuse_timeouts", 0),
            "top_kernel_by_event_time": top,
            # what the CPU loop would do vs what runs: SURVEY.md 8(d)'s 32 B per evaluation is an ALGORITHMIC figure (the
            # correspondences stay in LDS / L2, HBM is not the bound), next to the evaluations actually executed
            "work": {"evals_algorithmic_per_step": acc.get("evals_algorithmic", 0) / args.steps, "algorithmic_bytes_per_eval": BYTES_PER_EVAL,
                     "algorithmic_GBs_whole_step": BYTES_PER_EVAL * acc.get("evals_algorithmic", 0) / dt / 1e9,
                     "evals_mfma_count_per_step": acc.get("evals_mfma", 0) / args.steps,
                     "evals_fp32_bound_per_step": acc.get("evals_bound", 0) / args.steps,
                     "evals_fp64_sweep_per_step": acc.get("evals_fp64", 0) / args.steps,
                     "k_count_ms_per_step": acc.get("count_ms", 0.0) / args.steps, "k_score_ms_per_step": acc.get("sweep_ms", 0.0) / args.steps,
                     "lm_fp64_flop_per_step": sum(r.get("fp64_flop_per_step", 0.0) for r in rooflines.values())},
            "quality": {"median_rotation_error_deg_first64": R_err,
                        "mean_inlier_ratio": float(np.mean(res["num_inliers"] / n)) if len(res) else None},
        }
        if extra:
            line["configs"] = extra
        if pipelined is not None:
            line["pipelined"] = pipelined
        if host_rate is not None:
            line["host_buffers"] = {"value": host_rate, "unit": "image-pairs/s", "ratio_to_resident": host_rate / value,
                                    "note": "mdrp_estimate_batch with MDRP_MEM_HOST: pageable numpy buffers, H2D of 48 B per correspondence and D2H of "
                                            "records + inlier masks inside the timed region (PCIe-inclusive; never `value`)"}
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
            line["speedup_vs_cpu_1thread"] = value / line["cpu_baseline"]["value"]
            pv = cpu_line.get("port_vs_reference_binary")
            if pv and pv.get("port_over_reference"):
                # the port is slower than the reference binary on the same core: the honest ratio is against the reference's rate
                line["speedup_vs_reference_binary_1thread_est"] = line["speedup_vs_cpu_1thread"] * pv["port_over_reference"]
            if "multi_process" in line["cpu_baseline"]:
                line["speedup_vs_cpu_multi_process"] = value / line["cpu_baseline"]["multi_process"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
