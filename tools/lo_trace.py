"""Per-problem timing of the LO kernel (experiment build -DMDRP_LO_TRACE), with the time inside every LM split into cost sweeps,
normal-equation sweeps and the rest (solve, step, state expansion, reductions).  Build HERE (no GPU needed), run ON THE GPU BOX:
    python tools/lo_trace.py build          -> tools/gpu/libmdrp_lo_trace.so (travels with gpurun; *.so is git-ignored)
    gpurun -- python tools/lo_trace.py [bench workload name]     -> prints the distributions (environment knobs such as MDRP_LO_THREADS apply)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.environ.get("LO_TRACE_LIB") or os.path.join(ROOT, "tools", "gpu", "libmdrp_lo_trace.so")
trace = "/tmp/lo_trace.bin"
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from mdrp_amd import build
    build.build(force=True, defines=("MDRP_LO_TRACE",), out=lib)
elif len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import torch
    import bench
    name = os.environ.get("LO_TRACE_WORKLOAD", "calib_p3p_n2000_i10k")
    w = bench.Workload(name, 0, 1024, 1024, 0, torch.device("cuda:0"))
    for _ in range(2):
        w.launch()
        w.h.fetch_results(1024)
    print(f"workload {name}")
    ev = np.fromfile(trace, dtype=np.uint64).reshape(-1, 8)
    dur = (ev[:, 5] - ev[:, 4]).astype(np.int64) / 100.0   # wall_clock64 ticks at 100 MHz -> microseconds
    t_cost = (ev[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64) / 100.0
    t_acc = (ev[:, 6] >> np.uint64(32)).astype(np.int64) / 100.0
    chunk1 = (ev[:, 7] & np.uint64(1)).astype(bool)
    its = ((ev[:, 7] >> np.uint64(8)) & np.uint64(0xFFFF)).astype(np.int64)
    accs = ((ev[:, 7] >> np.uint64(24)) & np.uint64(0xFFFF)).astype(np.int64)
    t_lm = (ev[:, 7] >> np.uint64(40)).astype(np.int64) / 100.0
    for sel, name in ((~chunk1, "chunk 0 (lo0, beside k_bound / k_score)"), (chunk1, "chunk 1 (lo1, the tail)")):
        d = dur[sel]
        if not len(d):
            continue
        start = (ev[sel, 4] - ev[sel, 4].min()).astype(np.int64) / 100.0
        end = (ev[sel, 5] - ev[sel, 4].min()).astype(np.int64) / 100.0
        rest = t_lm[sel] - t_cost[sel] - t_acc[sel]
        print(f"{name}: {len(d)} problems; duration us: mean {d.mean():.0f} median {np.median(d):.0f} p90 {np.percentile(d, 90):.0f} p99 {np.percentile(d, 99):.0f} max {d.max():.0f}; "
              f"sum {d.sum() / 1e3:.0f} ms; makespan {end.max():.0f} us; sum / 2048 waves = {d.sum() / 2048:.0f} us")
        print(f"   LM iterations mean {its[sel].mean():.1f} (accepted {accs[sel].mean():.1f}); per problem: cost sweeps {t_cost[sel].mean():.0f} us, normal equations {t_acc[sel].mean():.0f} us, "
              f"rest of the LM {rest.mean():.0f} us, outside the LM (score, publish) {(d - t_lm[sel]).mean():.0f} us")
        print(f"   per sweep: cost {t_cost[sel].sum() / (its[sel] + 1).sum():.1f} us, normal equations {t_acc[sel].sum() / np.maximum(accs[sel], 1).sum():.1f} us; rest per iteration {rest.sum() / np.maximum(its[sel], 1).sum():.1f} us")
        for lo_, hi_ in ((0.0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
            w = (start >= lo_ * end.max()) & (start < hi_ * end.max())
            if w.any():
                print(f"   started in [{lo_:.2f}, {hi_:.2f}) of the makespan: {w.sum()} problems; per sweep cost {t_cost[sel][w].sum() / (its[sel][w] + 1).sum():.1f} us, "
                      f"normal equations {t_acc[sel][w].sum() / np.maximum(accs[sel][w], 1).sum():.1f} us, rest per iteration {rest[w].sum() / np.maximum(its[sel][w], 1).sum():.1f} us")
    pos = ev[:, 1].astype(np.int64)
    print("duration by trigger ordinal within the pair (ordinal: problems, mean us, p90 us, mean LM iterations): " +
          ", ".join(f"{k}: {(pos == k).sum()}, {dur[pos == k].mean():.0f}, {np.percentile(dur[pos == k], 90):.0f}, {its[pos == k].mean():.1f}" for k in range(0, int(pos.max()) + 1) if (pos == k).sum() >= 20))
    print("long problems by trigger ordinal (ordinal: share that ran all 25 iterations, p99 us, max us): " +
          ", ".join(f"{k}: {(its[pos == k] >= 25).mean():.3f}, {np.percentile(dur[pos == k], 99):.0f}, {dur[pos == k].max():.0f}" for k in range(0, int(pos.max()) + 1) if (pos == k).sum() >= 20))
    inl = ev[:, 2].astype(np.int64)
    print("by inlier count of the triggering minimal model (range: problems, mean us, share at 25 iterations, max us): " +
          ", ".join(f"{lo_}-{hi_}: {((inl >= lo_) & (inl < hi_)).sum()}, {dur[(inl >= lo_) & (inl < hi_)].mean():.0f}, {(its[(inl >= lo_) & (inl < hi_)] >= 25).mean():.3f}, {dur[(inl >= lo_) & (inl < hi_)].max():.0f}"
                    for lo_, hi_ in ((0, 100), (100, 300), (300, 600), (600, 800), (800, 900), (900, 1000), (1000, 2001)) if ((inl >= lo_) & (inl < hi_)).sum() >= 20))
    try:
        fin = np.fromfile(trace + ".final", dtype=np.uint64).reshape(-1, 8)
        fin = fin[fin[:, 2] > 0]
        t0 = fin[:, 1].min()
        st = (fin[:, 1] - t0).astype(np.int64) / 100.0; en = (fin[:, 2] - t0).astype(np.int64) / 100.0
        d = en - st
        it1 = (fin[:, 3] & np.uint64(0xFFFF)).astype(np.int64); it0 = (fin[:, 3] >> np.uint64(32)).astype(np.int64)
        print(f"final refinements (k_final): {len(d)} pairs; duration us: mean {d.mean():.0f} median {np.median(d):.0f} p90 {np.percentile(d, 90):.0f} p99 {np.percentile(d, 99):.0f} max {d.max():.0f}; "
              f"makespan {en.max():.0f} us; sum / 512 workgroups = {d.sum() / 512:.0f} us")
        print(f"   iterations: LO from the best model mean {it0.mean():.1f} max {it0.max()}; inlier refinement mean {it1.mean():.1f} p90 {np.percentile(it1, 90):.0f} max {it1.max()}; "
              f"us per iteration of the inlier refinement ~ {(d.sum() / np.maximum(it0 + it1, 1).sum()):.1f} (both refinements pooled)")
        tc, ta, tl, acc1 = (fin[:, k].astype(np.int64) / (100.0 if k < 7 else 1.0) for k in (4, 5, 6, 7))
        have = it1 > 0
        if have.any():
            rest = tl - tc - ta
            print(f"   inlier refinement, per pair: LM {tl[have].mean():.0f} us = cost sweeps {tc[have].mean():.0f} + normal equations {ta[have].mean():.0f} + rest {rest[have].mean():.0f}; "
                  f"per sweep: cost {tc[have].sum() / (it1[have] + 1).sum():.1f} us, normal equations {ta[have].sum() / np.maximum(acc1[have], 1).sum():.1f} us; "
                  f"rest per iteration {rest[have].sum() / it1[have].sum():.1f} us; accepted steps mean {acc1[have].mean():.1f} of {it1[have].mean():.1f}")
            long_ = it1 >= np.percentile(it1, 95)
            print(f"   the 5 % longest refinements ({long_.sum()} pairs, {it1[long_].mean():.0f} iterations): LM {tl[long_].mean():.0f} us = cost {tc[long_].mean():.0f} + normal equations {ta[long_].mean():.0f} + rest {rest[long_].mean():.0f}; "
                  f"per sweep: cost {tc[long_].sum() / (it1[long_] + 1).sum():.1f} us, normal equations {ta[long_].sum() / np.maximum(acc1[long_], 1).sum():.1f} us, rest per iteration {rest[long_].sum() / it1[long_].sum():.1f} us")
        order = np.argsort(-d)[:8]
        print("   longest pairs (start us, duration us, iterations): " + ", ".join(f"({st[i]:.0f}, {d[i]:.0f}, {it0[i]}+{it1[i]})" for i in order))
        late = st > 0.5 * en.max()
        print(f"   pairs started in the second half of the makespan: {late.sum()}, their mean duration {d[late].mean() if late.any() else 0:.0f} us")
    except FileNotFoundError:
        pass
else:
    env = dict(os.environ, MDRP_LIB=lib, MDRP_LO_TRACE_FILE=trace)
    if len(sys.argv) > 1:
        env["LO_TRACE_WORKLOAD"] = sys.argv[1]
    subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
