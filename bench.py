#!/usr/bin/env python3
"""bench.py — image-pairs/s of the RePoseD RANSAC hot path on MI355X (BASELINE.json metric).

A step = one pass of estimate_monodepth_relative_pose over one batch of synthetic image pairs that already sit in
HBM: calibrated 3-point solver (P3P path, shift off), 2000 correspondences per pair, max_iterations = min_iterations
= 10000 (BASELINE.json configs[1]; /root/reference/make_video.py:192-194), 50 % outliers, 0.5 px / 2 % noise,
1024 pairs per GPU.  Pairs shard across ranks with no data-path collective (weak scaling); for N > 1 the step ends
with one RCCL all_gather of the fixed-size result records.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
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
    "varying_n5000_i10k": (2, 5000, 10000, 0.5, False, "varying"),
    # SURVEY.md §8(d) C2 also asks for the outlier-free shape
    "calib_p3p_n2000_i10k_clean": (0, 2000, 10000, 0.0, False, None),
    "calib_shift_n2000_i10k_clean": (0, 2000, 10000, 0.0, True, None),
}
FP64_PEAK_TFLOPS = 78.6    # MI355X_MICROARCH.md: fp64 vector peak
FLOPS_PER_EVAL = 35.0      # fp64 flops of one Sampson evaluation (cheirality of inliers not counted)


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
    ro = po.ransac_opt(max_iterations=iters, min_iterations=iters, max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=es)
    bo = po.bundle_opt(loss_type=4)
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    po.lib()
    t0 = time.perf_counter()
    for i in range(count):
        po.estimate(kind, b["x1"][i], b["x2"][i], b["d1"][i], b["d2"][i], ro, bo, cam if kind == 0 else None, cam if kind == 0 else None)
    return time.perf_counter() - t0


def cpu_baseline(workload, pairs):
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
    if cores > 1:
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


def pmc_traffic(workload, batch):
    """HBM bytes per k_score launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, per the gfx950
    correction in MI355X_MICROARCH.md) when they were taken on this workload and batch; else None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("pairs_per_gpu") == batch:
            for k, v in d.get("kernels", {}).items():
                if "k_score" in k:
                    best = (v["hbm_bytes_corrected"], os.path.basename(f))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="image pairs per GPU per step")
    ap.add_argument("--workload", default="calib_p3p_n2000_i10k", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-pairs", type=int, default=96, help="pairs timed on the CPU baseline (0 = skip)")
    args = ap.parse_args()

    cpu_line = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.cpu_pairs > 0:
        cpu_line = cpu_baseline(args.workload, args.cpu_pairs)  # forks workers: before anything initialises the GPU

    import torch
    import torch.distributed as dist
    from mdrp_amd import _capi

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

    kind, n, iters, of, es, rf = WORKLOADS[args.workload]
    B = args.batch
    b = make_inputs(args.workload, rank * B, B)
    x1 = torch.from_numpy(b["x1"]).to(dev); x2 = torch.from_numpy(b["x2"]).to(dev)
    d1 = torch.from_numpy(b["d1"]).to(dev); d2 = torch.from_numpy(b["d2"]).to(dev)
    mask = torch.zeros((B, n), dtype=torch.uint8, device=dev)
    cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE)
    cams["params"][:, 0] = 800.0
    ro = _capi.ransac_opt_from_dict({"max_iterations": iters, "min_iterations": iters, "max_epipolar_error": 2.0,
                                     "max_reproj_error": 16.0, "monodepth_estimate_shift": es})
    bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    h = _capi.Handle(local_rank)  # its own stream; the sweep is timed with HIP events recorded on that stream
    from mdrp_amd import dist as mdist

    def step():
        h.estimate_batch_device(kind, x1.data_ptr(), x2.data_ptr(), d1.data_ptr(), d2.data_ptr(), B, n, ro, bo, None,
                                cams if kind == 0 else None, cams if kind == 0 else None, mask.data_ptr())
        res = h.fetch_results(B)
        if world > 1:  # final gather of the pose records over RCCL/xGMI: 136 B per pair
            mdist.gather_results(res, B * world, None, dev)
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sweep_ms = 0.0
    sweep_launches = 0
    sweep_evals = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        ms, ln, ev = h.last_sweep_stats()
        sweep_ms += ms; sweep_launches += ln; sweep_evals += ev
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        pairs = B * world * args.steps
        value = pairs / dt
        # roofline of the dominant kernel (k_score): algorithmic bytes per launch / average launch duration
        avg_launch_s = (sweep_ms / 1e3) / max(sweep_launches, 1)
        bytes_per_launch = BYTES_PER_EVAL * sweep_evals / max(sweep_launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        valu_tf = FLOPS_PER_EVAL * sweep_evals / max(sweep_launches, 1) / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        from mdrp_amd import synth
        from mdrp_amd.poselib import _quat_to_R
        R_err = float(np.median([synth.rotation_error_deg(g["R"], _quat_to_R(r["model"]["q"])) for r, g in zip(res[:64], b["gt"][:64])]))
        traffic = pmc_traffic(args.workload, B)
        phys = (traffic[0] / avg_launch_s / 1e9) if (traffic and avg_launch_s > 0) else None
        line = {
            "metric": "image-pairs/sec (2000 corrs, 10k RANSAC iters)", "value": value, "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "pairs_per_gpu": B, "correspondences": n, "ransac_iterations": iters,
                       "outlier_fraction": of, "estimator": ["calibrated", "shared_focal", "varying_focal"][kind],
                       "monodepth_estimate_shift": es, "parallelism": f"pairs sharded x{world}, all_gather of results"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic or (None, None))[0], "traffic_unit": "bytes per launch (PMC)",
                         "traffic_source": (traffic or (None, None))[1], "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel": "k_score", "evals_per_launch": sweep_evals / max(sweep_launches, 1),
                         "avg_launch_ms": 1e3 * avg_launch_s, "sweep_share_of_step": (sweep_ms / 1e3) / dt,
                         # physical HBM-side rate of the sweep: PMC bytes per launch / this run's launch duration
                         "physical_GBs": phys, "physical_frac": (phys / HBM_PEAK_GBS) if phys else None},
            # SURVEY.md §8(d) asks for both fractions: the physically binding limit of the sweep is VALU issue, not HBM
            "roofline_valu": {"bound": "valu-fp64", "achieved": valu_tf, "frac": valu_tf / FP64_PEAK_TFLOPS,
                              "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s (algorithmic: 35 flop per evaluation the CPU loop would do)",
                              "note": "the exact bail-out and the fp32 pre-filter skip work, so algorithmic flops are not executed flops"},
            "quality": {"median_rotation_error_deg_first64": R_err,
                        "mean_inlier_ratio": float(np.mean(res["num_inliers"] / n))},
        }
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
            line["speedup_vs_cpu_1thread"] = value / line["cpu_baseline"]["value"]
            if "multi_process" in line["cpu_baseline"]:
                line["speedup_vs_cpu_multi_process"] = value / line["cpu_baseline"]["multi_process"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
