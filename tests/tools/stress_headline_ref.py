#!/usr/bin/env python3
"""A one-off extension of tests/golden/headline_ref_<workload>.npz (the 1024 pairs bench.py times, against the reference binary) to FRESH pairs of the same
shapes — other indices of the same generator:
    build container:  python3 tests/tools/stress_headline_ref.py gen FIRST COUNT build/headline_ref_more.npz [workload ...]     (reference shim, 8 workers)
    GPU box        :  python3 tests/tools/stress_headline_ref.py run build/headline_ref_more.npz
`run` sends each workload's COUNT pairs through the device-resident batch entry point (the one bench.py times) in one call and counts the pairs whose
iterations, inlier count, mask and model (1e-6) equal the reference's, and the LO-count differences."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
import gen_golden_headline_ref as gh  # noqa: E402

# the comparison rows' and the outlier-free shape's timed batches (tests/tools/gen_golden_headline_ref_classic.py): kind, n, outlier_frac, random_focal
CLASSIC = {"relpose_5pt_n2000_i10k": (3, 2000, 0.5, None), "fundamental_7pt_n2000_i10k": (5, 2000, 0.5, None), "shared_6pt_n2000_i10k": (4, 2000, 0.5, "shared"),
           "calib_p3p_n2000_i10k_clean": (0, 2000, 0.0, None)}


def _work_classic(args):  # the reference binary on pairs [lo, hi) of a CLASSIC workload
    workload, lo, hi = args
    import refshim as rs
    from mdrp_amd import synth
    kind, n, of, rf = CLASSIC[workload]
    cam = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
    rows = []
    for i in range(lo, hi):
        p = synth.make_pair(i, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf)
        gh._srand(1)
        if kind == 0:
            m, st, mask = rs.estimate(0, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(**gh.OPTS), rs.bopt(loss_type=4), cam, cam)
            full = np.r_[m, 1.0, 1.0]
        else:
            kw = {k: v for k, v in gh.OPTS.items() if k != "max_reproj_error"}
            m, st, mask = rs.estimate_classic(kind, p["x1"], p["x2"], rs.ropt(**kw), rs.bopt(loss_type=4), cam if kind == 3 else None, cam if kind == 3 else None, pp=(0.0, 0.0))
            m = np.asarray(m, float).reshape(-1)
            full = np.zeros(12); full[: len(m)] = m
        rows.append((i, full, (int(st[0]), int(st[1]), int(st[2])), (float(st[3]), float(st[4])), np.packbits(mask), 0))
    return rows


def gen(first, count, out, names):
    d = {"first": first, "count": count, "names": np.array(names)}
    for w in names:
        t0 = time.perf_counter()
        jobs = [(w, lo, min(lo + 4, first + count)) for lo in range(first, first + count, 4)]
        with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
            rows = [r for chunk in pool.imap(_work_classic if w in CLASSIC else gh._work, jobs, chunksize=1) for r in chunk]
        rows.sort(key=lambda r: r[0])
        assert [r[0] for r in rows] == list(range(first, first + count))
        d[w + "_model"] = np.array([r[1] for r in rows]); d[w + "_istats"] = np.array([r[2] for r in rows], dtype=np.int64)
        d[w + "_fstats"] = np.array([r[3] for r in rows]); d[w + "_mask"] = np.array([r[4] for r in rows])
        print(w, count, "pairs,", f"{time.perf_counter() - t0:.0f} s", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


def run(ref):
    import torch
    from helpers import model_diff
    from mdrp_amd import _capi as capi, synth
    g = np.load(ref)
    first, count = int(g["first"]), int(g["count"])
    dev = torch.device("cuda", 0)
    for w in [str(x) for x in g["names"]]:
        if w in CLASSIC:
            (kind, n, of, rf), es, s1, s2 = CLASSIC[w], False, 0.0, 0.0
        else:
            kind, es, n, of, rf, (s1, s2) = gh.HEADLINE[w]
        b = synth.make_batch(first, count, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf, shift1=s1, shift2=s2)
        t = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
        mask_t = torch.zeros((count, n), dtype=torch.uint8, device=dev)
        cams = np.zeros(count, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
        ro = capi.ransac_opt_from_dict({"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "seed": 0,
                                        "monodepth_estimate_shift": es})
        bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
        c = cams if kind in (0, 3, 4) else None  # (kind 4: cam1 carries the principal point, 0 here)
        if kind == 4:
            cams["params"][:, 0] = 0.0
        h = capi.Handle(0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        mono = kind <= 2
        h.estimate_batch_device(kind, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr() if mono else 0, t[3].data_ptr() if mono else 0, count, n, ro, bo, None, c, c, mask_t.data_ptr())
        res = h.fetch_results(count)
        dt = time.perf_counter() - t0
        mask = mask_t.cpu().numpy()
        h.close()
        ist, rm = g[w + "_istats"], g[w + "_model"]
        same_stats = (res["iterations"].astype(np.int64) == ist[:, 1]) & (res["num_inliers"].astype(np.int64) == ist[:, 2])
        same_mask = (mask == np.unpackbits(g[w + "_mask"], axis=1)[:, :n]).all(axis=1)
        def mdiff(m, r):
            if kind <= 2:
                return model_diff(m, r)
            if kind == 5:
                a, b_ = m[:9] / np.linalg.norm(m[:9]), r[:9] / np.linalg.norm(r[:9])
                return min(np.abs(a - b_).max(), np.abs(a + b_).max())
            d = min(np.abs(m[:4] - r[:4]).max(), np.abs(m[:4] + r[:4]).max()) + np.abs(m[4:7] / np.linalg.norm(m[4:7]) - r[4:7] / np.linalg.norm(r[4:7])).max()
            return max(d, abs(m[10] - r[7]) / abs(r[7])) if kind == 4 else d   # the shared focal length: mdrp_model.f1 | the reference's 8th value
        md = np.array([mdiff(capi.model_to_array(res[i]["model"]), rm[i]) for i in range(count)])
        nan_both = np.array([np.isnan(rm[i]).any() and np.isnan(capi.model_to_array(res[i]["model"])).any() for i in range(count)])
        ok = same_stats & same_mask & ((md < 1e-6) | nan_both)
        lo = np.nonzero(res["refinements"].astype(np.int64) != ist[:, 0])[0]
        bad = np.nonzero(~ok)[0]
        print(f"{w}: pairs {first} .. {first + count - 1} in one call ({dt * 1e3:.0f} ms): {int(ok.sum())} / {count} identical to the REFERENCE BINARY (iterations, inliers, mask, model 1e-6); "
              f"LO count differs on {len(lo)}; not identical: {[(int(i) + first, 'stats' if not same_stats[i] else ('mask' if not same_mask[i] else f'model {md[i]:.1e}')) for i in bad[:24]]}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "gen":
        gen(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5:] or list(gh.HEADLINE))
    else:
        run(sys.argv[2])
