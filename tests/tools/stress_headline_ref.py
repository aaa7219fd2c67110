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


def gen(first, count, out, names):
    d = {"first": first, "count": count, "names": np.array(names)}
    for w in names:
        t0 = time.perf_counter()
        jobs = [(w, lo, min(lo + 4, first + count)) for lo in range(first, first + count, 4)]
        with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
            rows = [r for chunk in pool.imap(gh._work, jobs, chunksize=1) for r in chunk]
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
        kind, es, n, of, rf, (s1, s2) = gh.HEADLINE[w]
        b = synth.make_batch(first, count, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf, shift1=s1, shift2=s2)
        t = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
        mask_t = torch.zeros((count, n), dtype=torch.uint8, device=dev)
        cams = np.zeros(count, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
        ro = capi.ransac_opt_from_dict({"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "seed": 0,
                                        "monodepth_estimate_shift": es})
        bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
        c = cams if kind == 0 else None
        h = capi.Handle(0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        h.estimate_batch_device(kind, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), count, n, ro, bo, None, c, c, mask_t.data_ptr())
        res = h.fetch_results(count)
        dt = time.perf_counter() - t0
        mask = mask_t.cpu().numpy()
        h.close()
        ist, rm = g[w + "_istats"], g[w + "_model"]
        same_stats = (res["iterations"].astype(np.int64) == ist[:, 1]) & (res["num_inliers"].astype(np.int64) == ist[:, 2])
        same_mask = (mask == np.unpackbits(g[w + "_mask"], axis=1)[:, :n]).all(axis=1)
        md = np.array([model_diff(capi.model_to_array(res[i]["model"]), rm[i]) for i in range(count)])
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
