"""The N>1 path (mdrp_amd/dist.py) under gloo with world_size 2 on CPU: contiguous sharding of pairs, no data-path
collective, one all_gather of the result records.  The per-rank estimator is injected (the CPU oracle — tests may use
it as the checker), because the product's own estimator needs a GPU."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_local_fn(kind, x1, x2, d1, d2, ro, bo, npp, c1, c2):
    from mdrp_amd import _capi
    from oracle import pyorc as po
    B, N = d1.shape
    res = np.zeros(B, dtype=_capi.RESULT_DTYPE)
    mask = np.zeros((B, N), dtype=np.uint8)
    oro = po.ransac_opt(ro.max_iterations, ro.min_iterations, ro.dyn_num_trials_mult, ro.success_prob, ro.max_reproj_error,
                        ro.max_epipolar_error, ro.seed, bool(ro.monodepth_estimate_shift), ro.monodepth_weight_sampson)
    obo = po.bundle_opt(bo.max_iterations, bo.loss_type, bo.loss_scale, bo.gradient_tol, bo.step_tol, bo.initial_lambda, bo.min_lambda, bo.max_lambda)
    for i in range(B):
        n = N if npp is None else int(npp[i])
        cams = [None, None]
        if kind == 0:
            cams = [po.cam_flat(int(c["model_id"]), list(c["params"][:4 if c["model_id"] == 1 else 3])) for c in (c1[i], c2[i])]
        m, st, mk = po.estimate(kind, x1[i, :n], x2[i, :n], d1[i, :n], d2[i, :n], oro, obo, cams[0], cams[1])
        r = res[i]
        r["model"]["q"] = m[:4]; r["model"]["t"] = m[4:7]; r["model"]["scale"] = m[7]; r["model"]["shift1"] = m[8]
        r["model"]["shift2"] = m[9]; r["model"]["f1"] = m[10]; r["model"]["f2"] = m[11]
        r["refinements"], r["iterations"], r["num_inliers"] = st.refinements, st.iterations, st.num_inliers
        r["inlier_ratio"], r["model_score"] = st.inlier_ratio, st.model_score
        mask[i, :n] = mk
    return res, mask


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from mdrp_amd import dist as mdist, synth, _capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = synth.make_batch(300, total, 120, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3, random_focal="shared")
    ro = {"max_iterations": 200, "min_iterations": 200, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    res, mask = mdist.estimate_sharded(_capi.SHARED_FOCAL, b["x1"], b["x2"], b["d1"], b["d2"], ro, {"loss_type": "TRUNCATED_CAUCHY"},
                                       local_fn=_oracle_local_fn, want_mask=True)
    np.save(os.path.join(out_dir, f"res_{rank}.npy"), res)
    np.save(os.path.join(out_dir, f"mask_{rank}.npy"), mask)
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 4])
def test_two_rank_sharding_and_gather(tmp_path, total):
    from mdrp_amd import dist as mdist, synth, _capi
    port = 29600 + os.getpid() % 300 + total
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "res_0.npy"), np.load(tmp_path / "res_1.npy")
    assert len(r0) == total and r0.tobytes() == r1.tobytes()          # every rank holds all records, identical
    assert (np.load(tmp_path / "mask_0.npy") == np.load(tmp_path / "mask_1.npy")).all()
    # single-process run of the same pairs gives the same records in the same order
    b = synth.make_batch(300, total, 120, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3, random_focal="shared")
    ro = _capi.ransac_opt_from_dict({"max_iterations": 200, "min_iterations": 200, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    ref, refmask = _oracle_local_fn(_capi.SHARED_FOCAL, b["x1"], b["x2"], b["d1"], b["d2"], ro, _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None, None, None)
    assert ref.tobytes() == r0.tobytes()
    assert (refmask == np.load(tmp_path / "mask_0.npy")).all()


def test_shard_bounds_cover_everything():
    from mdrp_amd.dist import shard_bounds
    for total in (0, 1, 7, 8, 100000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world)[:2] for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in spans) <= (total + world - 1) // world


def _calib_problem(total, N=150):
    """`total` calibrated pairs with ragged correspondence counts and per-pair cameras (PINHOLE and SIMPLE_PINHOLE mixed)"""
    from mdrp_amd import synth, _capi
    ns = np.array([N - 7 * (i % 5) for i in range(total)], dtype=np.int32)
    x1 = np.zeros((total, N, 2)); x2 = np.zeros((total, N, 2)); d1 = np.ones((total, N)); d2 = np.ones((total, N))
    cam1 = np.zeros(total, dtype=_capi.CAMERA_DTYPE); cam2 = np.zeros(total, dtype=_capi.CAMERA_DTYPE)
    for i in range(total):
        f1, f2 = 700.0 + 40 * i, 900.0 - 30 * i
        p = synth.make_pair(800 + i, int(ns[i]), noise_px=0.5, depth_noise=0.02, outlier_frac=0.25, f1=f1, f2=f2, pp=(640.0, 480.0))
        x1[i, :ns[i]] = p["x1"]; x2[i, :ns[i]] = p["x2"]; d1[i, :ns[i]] = p["d1"]; d2[i, :ns[i]] = p["d2"]
        cam1[i]["model_id"] = 0; cam1[i]["params"][:3] = [f1, 640.0, 480.0]
        cam2[i]["model_id"] = 1; cam2[i]["params"][:4] = [f2, f2, 640.0, 480.0]
    return x1, x2, d1, d2, ns, cam1, cam2


def _worker_local(rank, world, port, total, out_dir):
    """every rank builds ONLY its own block (BASELINE configs[4]: the data set is never replicated)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from mdrp_amd import dist as mdist, _capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi, per = mdist.shard_bounds(total, rank, world)
    x1, x2, d1, d2, ns, cam1, cam2 = _calib_problem(total)
    sl = slice(lo, hi)
    ro = {"max_iterations": 200, "min_iterations": 200, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "seed": 3}
    res, mask = mdist.estimate_local_shard(_capi.CALIB, total, x1[sl], x2[sl], d1[sl], d2[sl], ro, {"loss_type": "TRUNCATED_CAUCHY"},
                                           n_per_pair=ns[sl], cam1=cam1[sl], cam2=cam2[sl], local_fn=_oracle_local_fn, want_mask=True)
    np.save(os.path.join(out_dir, f"res_{rank}.npy"), res)
    np.save(os.path.join(out_dir, f"mask_{rank}.npy"), mask)
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [7, 3, 1])
def test_local_shards_uneven_with_counts_and_cameras(tmp_path, total):
    """world_size 2, uneven blocks (4 + 3, 2 + 1, 1 + 0 pairs), ragged n_per_pair, per-pair cameras: every rank ends up with
    all records in pair order, equal to a single-process run over the whole set"""
    from mdrp_amd import _capi
    port = 29900 + os.getpid() % 300 + total
    mp.spawn(_worker_local, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "res_0.npy"), np.load(tmp_path / "res_1.npy")
    assert len(r0) == total and r0.tobytes() == r1.tobytes()
    m0 = np.load(tmp_path / "mask_0.npy")
    assert (m0 == np.load(tmp_path / "mask_1.npy")).all() and m0.shape[0] == total
    x1, x2, d1, d2, ns, cam1, cam2 = _calib_problem(total)
    ro = _capi.ransac_opt_from_dict({"max_iterations": 200, "min_iterations": 200, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "seed": 3})
    ref, refmask = _oracle_local_fn(_capi.CALIB, x1, x2, d1, d2, ro, _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), ns, cam1, cam2)
    assert ref.tobytes() == r0.tobytes() and (refmask == m0).all()
    assert (r0["num_inliers"] > 40).all()
