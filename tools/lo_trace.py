"""Per-problem timing of the LO kernels (experiment build -DMDRP_LO_TRACE).  Run ON THE GPU BOX: python tools/lo_trace.py
Prints the distribution of LM problem durations, the makespan structure and how well simple keys predict the duration."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "libmdrp_lo_trace.so")
trace = os.path.join(ROOT, "gpurun_out", "lo_trace.bin")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from mdrp_amd import _capi, synth
    b = synth.make_batch(0, 1024, 2000, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
    cams = np.zeros(1024, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = _capi.ransac_opt_from_dict({"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    h = _capi.Handle(0)
    for _ in range(2):
        h.estimate_batch(0, b["x1"], b["x2"], b["d1"], b["d2"], ro, bo, None, cams, cams)
    ev = np.fromfile(trace, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    dur = (ev[:, 5] - ev[:, 4]) / 100.0   # wall_clock64 ticks at 100 MHz -> microseconds
    t0 = ev[:, 4].min()
    for c in (0, ):
        pass
    for off, name in ((0, "chunk 0 (lo0)"), (None, "chunk 1 (lo1)")):
        sel = ev[:, 7] == 0 if off == 0 else ev[:, 7] != 0
        d, e = dur[sel], ev[sel]
        start = (e[:, 4] - e[:, 4].min()) / 100.0
        end = (e[:, 5] - e[:, 4].min()) / 100.0
        print(f"{name}: {len(d)} problems, duration us: mean {d.mean():.0f} median {np.median(d):.0f} p90 {np.percentile(d, 90):.0f} p99 {np.percentile(d, 99):.0f} max {d.max():.0f}; "
              f"sum {d.sum() / 1e3:.0f} ms; makespan {end.max():.0f} us; sum / 2048 waves = {d.sum() / 2048:.0f} us")
        late = start > 0.5 * end.max()
        print(f"   problems started in the second half of the makespan: {late.sum()}, their mean duration {d[late].mean() if late.any() else 0:.0f} us, longest {d[late].max() if late.any() else 0:.0f}")
        for key, kn in ((e[:, 2], "cnt_ref (inliers of the minimal model)"), (e[:, 3], "ref_cnt (inliers after LO)"), (e[:, 6], "iteration index"), (e[:, 1], "trigger position")):
            r = np.corrcoef(key.astype(float), d)[0, 1]
            print(f"   corr(duration, {kn}) = {r:.3f}")
        # duration by trigger position
        for pos in range(0, 12, 2):
            m = e[:, 1] == pos
            if m.any():
                print(f"   position {pos}: n {m.sum()} mean {d[m].mean():.0f} us, cnt_ref mean {e[m, 2].mean():.0f}")
else:
    from mdrp_amd import build
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    build.build(force=True, defines=("MDRP_LO_TRACE",), out=lib)
    env = dict(os.environ, MDRP_LIB=lib, MDRP_LO_TRACE_FILE=trace)
    subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
