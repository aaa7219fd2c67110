"""Boundary behaviour of the C-ABI / drop-in module on a real MI355X: streams, threads, devices."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAM = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 0.0, 0.0]}
RO = {"max_iterations": 1500, "min_iterations": 1500, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
BO = {"loss_type": "TRUNCATED_CAUCHY"}


def test_default_stream_orders_with_async_producers():
    """estimate_batch_torch on torch's DEFAULT (null) stream, inputs produced by asynchronous torch kernels queued just
    before the call (a long dependent chain, so they are certainly still running when the call is made): results must
    equal the host-buffer path.  cuda_stream == 0 must mean the legacy default stream, not a private stream."""
    import torch
    import mdrp_amd.poselib as poselib
    from mdrp_amd import synth
    b = synth.make_batch(6200, 16, 600, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3)
    geoms, infos = poselib.estimate_monodepth_relative_pose_batch(b["x1"], b["x2"], b["d1"], b["d2"], CAM, CAM, RO, BO)
    dev = torch.device("cuda", 0)
    assert torch.cuda.current_stream(dev).cuda_stream == 0
    src = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
    big = torch.randn(4096, 4096, device=dev)
    for rep in range(3):
        t = [torch.full_like(v, float("nan")) for v in src]              # the inputs do not exist yet: poison
        torch.cuda.synchronize()
        junk = big
        for _ in range(40):                                               # ~100 ms of queued work in front of the producers
            junk = junk @ big
            junk = junk / junk.abs().max()
        bump = (junk[0, 0] * 0.0).double()                                # data dependence on the long chain
        for v, o in zip(src, t):
            torch.add(v, bump, out=o)                                     # asynchronous producers of the real inputs
        res, mask = poselib.estimate_batch_torch("calibrated", *t, CAM, CAM, RO, BO)
        after = mask.sum(dim=1)                                           # consumer on the same stream, no explicit sync
        for i in range(16):
            assert int(res[i]["num_inliers"]) == infos[i]["num_inliers"], (rep, i)
            assert int(res[i]["refinements"]) == infos[i]["refinements"]
            assert np.array_equal(mask[i].cpu().numpy().astype(bool), np.array(infos[i]["inliers"]))
        assert np.array_equal(after.cpu().numpy(), np.array([sum(x["inliers"]) for x in infos]))


def test_torch_batch_accepts_tensor_n_per_pair_and_keeps_current_device():
    import torch
    import mdrp_amd.poselib as poselib
    from mdrp_amd import synth
    b = synth.make_batch(6300, 4, 300, noise_px=0.5, outlier_frac=0.2)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
    npp = torch.tensor([300, 250, 3, 0], device=dev, dtype=torch.int32)
    before = torch.cuda.current_device()
    res, mask = poselib.estimate_batch_torch("calibrated", *t, CAM, CAM, RO, BO, n_per_pair=npp)
    assert torch.cuda.current_device() == before
    assert int(res[3]["iterations"]) == 0 and int(mask[1, 250:].sum()) == 0 and int(res[0]["num_inliers"]) > 150


def test_two_threads_two_handles_run_concurrently_and_agree():
    """The reference releases the GIL around its estimators and is re-entrant.  Here: Python threads
    calling the drop-in API get one handle each (thread-local default handle) and must produce exactly the results of
    the same calls made one after the other; threads sharing ONE explicit handle are serialised by the library's lock
    and must agree as well."""
    import mdrp_amd.poselib as poselib
    from mdrp_amd import _capi, synth
    jobs = []
    for k in range(4):
        b = synth.make_batch(6400 + 50 * k, 24, 500 + 100 * k, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4)
        jobs.append(b)
    ro = dict(RO, max_iterations=3000, min_iterations=3000)

    def run(b):
        g, info = poselib.estimate_monodepth_relative_pose_batch(b["x1"], b["x2"], b["d1"], b["d2"], CAM, CAM, ro, BO)
        return [(tuple(x.pose.q), tuple(x.pose.t), x.scale) for x in g], [(i["num_inliers"], i["refinements"], i["model_score"]) for i in info]

    serial = [run(b) for b in jobs]
    out = [None] * len(jobs)
    handles = [None] * len(jobs)
    errs = []

    def worker(i):
        try:
            for _ in range(3):
                out[i] = run(jobs[i])
            handles[i] = _capi.default_handle(0)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert out == serial
    assert len({id(h) for h in handles}) == len(jobs), "threads must not share a default handle"

    # one shared handle, four threads: the library serialises the calls
    h = _capi.Handle(0)
    cams = np.zeros(24, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ropt, bopt = _capi.ransac_opt_from_dict(ro), _capi.bundle_opt_from_dict(BO)
    ref = [h.estimate_batch(0, b["x1"], b["x2"], b["d1"], b["d2"], ropt, bopt, None, cams, cams) for b in jobs]
    got = [None] * len(jobs)

    def shared(i):
        try:
            b = jobs[i]
            for _ in range(3):
                got[i] = h.estimate_batch(0, b["x1"], b["x2"], b["d1"], b["d2"], ropt, bopt, None, cams, cams)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=shared, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for (r0, m0), (r1, m1) in zip(ref, got):
        assert r0.tobytes() == r1.tobytes() and np.array_equal(m0, m1)
    h.close()


def test_initial_pose_and_score_initial_model_vs_reference_golden(golden):
    """the drop-in signatures with initial_pose / initial_image_pair against tests/golden/initial.npz (reference binary): stats,
    model, mask; the pose handed in must not matter, its scale must come back when nothing is adopted"""
    import mdrp_amd.poselib as poselib
    from helpers import model_diff
    from test_oracle_golden import initial_cases
    g = golden("initial")
    cam = {"model": "SIMPLE_PINHOLE", "width": 1280, "height": 960, "params": [800.0, 640.0, 480.0]}
    for i, kind, flag, its, seed, deg in initial_cases(g):
        ro = {"max_iterations": its, "min_iterations": its, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "seed": seed}
        ini = g[f"initial_{i}"]
        geom = poselib.MonoDepthTwoViewGeometry(poselib.CameraPose(ini[:4], ini[4:7]), ini[7], ini[8], ini[9])
        x = (g[f"x1_{i}"], g[f"x2_{i}"], g[f"d1_{i}"], g[f"d2_{i}"])
        if flag:   # the binding sets score_initial_model when an initial model is passed
            if kind == 0:
                out, info = poselib.estimate_monodepth_relative_pose(*x, cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"}, initial_pose=geom)
            else:
                fn = poselib.estimate_monodepth_shared_focal_relative_pose if kind == 1 else poselib.estimate_monodepth_varying_focal_relative_pose
                pair = poselib.MonoDepthImagePair(geom, poselib.Camera("SIMPLE_PINHOLE", [ini[10], 0, 0]), poselib.Camera("SIMPLE_PINHOLE", [ini[11], 0, 0]))
                out, info = fn(*x, ro, {"loss_type": "TRUNCATED_CAUCHY"}, initial_image_pair=pair)
        else:
            if kind == 0:
                out, info = poselib.estimate_monodepth_relative_pose(*x, cam, cam, ro, {"loss_type": "TRUNCATED_CAUCHY"})
            else:
                fn = poselib.estimate_monodepth_shared_focal_relative_pose if kind == 1 else poselib.estimate_monodepth_varying_focal_relative_pose
                out, info = fn(*x, ro, {"loss_type": "TRUNCATED_CAUCHY"})
        gm = out if kind == 0 else out.geometry
        m = np.r_[gm.pose.q, gm.pose.t, gm.scale, gm.shift1, gm.shift2, 1.0 if kind == 0 else out.camera1.focal(), 1.0 if kind == 0 else out.camera2.focal()]
        ref_m, ref_st = g[f"model_{i}"], g[f"stats_{i}"]
        assert (info["refinements"], info["iterations"], info["num_inliers"]) == tuple(int(v) for v in ref_st[:3]), (i, info, ref_st)
        assert info["model_score"] == pytest.approx(ref_st[4], rel=1e-9)
        if flag or not deg:
            assert model_diff(m, ref_m) < 1e-6, (i, m, ref_m)
        assert np.array_equal(np.array(info["inliers"], dtype=np.uint8), g[f"mask_{i}"])


def test_handle_before_torch_cuda_init_in_a_fresh_process():
    """torch imported, CUDA not yet initialised, an mdrp handle created first: torch.cuda must still come up (the wheel bundles
    its own HIP runtime; _capi brings it up before ours)"""
    import subprocess
    import sys
    code = ("import torch, numpy as np\n"
            "from mdrp_amd import _capi\n"
            "h = _capi.Handle(0)\n"
            "x = torch.ones(8, device='cuda')\n"
            "print('ok', float(x.sum()))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         cwd=__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    assert out.returncode == 0 and "ok 8.0" in out.stdout, out.stderr[-2000:]


def test_batch_pipeline_equals_sequential_calls():
    """two batches in flight (two handles, two host threads): same records and masks, in submission order, as one call after
    the other — for a monodepth estimator and for a baseline"""
    from mdrp_amd import _capi, synth
    from mdrp_amd.pipeline import BatchPipeline
    batches = [synth.make_batch(6800 + 40 * k, 20 + 3 * k, 400, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4) for k in range(6)]
    ro = {"max_iterations": 2000, "min_iterations": 2000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    bo = {"loss_type": "TRUNCATED_CAUCHY"}
    h = _capi.Handle(0)
    rop, bop = _capi.ransac_opt_from_dict(ro), _capi.bundle_opt_from_dict(bo)
    for kind in (_capi.CALIB, _capi.RELPOSE_5PT):
        def cams(b):
            c = np.zeros(len(b["x1"]), dtype=_capi.CAMERA_DTYPE); c["params"][:, 0] = 800.0
            return c
        seq = [h.estimate_batch(kind, b["x1"], b["x2"], b["d1"] if kind == 0 else None, b["d2"] if kind == 0 else None, rop, bop, None, cams(b), cams(b))
               for b in batches]
        with BatchPipeline(depth=2) as pipe:
            futs = [pipe.submit(kind, b["x1"], b["x2"], b["d1"] if kind == 0 else None, b["d2"] if kind == 0 else None, ro, bo, None, cams(b), cams(b))
                    for b in batches]
            got = [f.result() for f in futs]
        for (r0, m0), (r1, m1) in zip(seq, got):
            assert r0.tobytes() == r1.tobytes() and np.array_equal(m0, m1)
    h.close()


@pytest.mark.gpu
def test_batches_larger_than_one_pass(monkeypatch):
    """A batch that does not fit the scratch budget is processed in passes of consecutive pairs (estimate_device); forced here
    with MDRP_PAIRS_PER_PASS on a ragged batch: records and masks equal the single-pass ones bit for bit (pairs are independent
    and a pass only changes which launches they share), for a monodepth estimator and a baseline."""
    from mdrp_amd import _capi, synth
    B, N = 37, 400
    pairs = [synth.make_pair(6600 + i, [N, 250, 2, 90][i % 4] if i % 3 == 0 else N, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4) for i in range(B)]
    n_per = np.array([len(p["x1"]) for p in pairs], dtype=np.int32)
    x1, x2 = np.zeros((B, N, 2)), np.zeros((B, N, 2))
    d1, d2 = np.ones((B, N)), np.ones((B, N))
    for i, p in enumerate(pairs):
        x1[i, : n_per[i]], x2[i, : n_per[i]], d1[i, : n_per[i]], d2[i, : n_per[i]] = p["x1"], p["x2"], p["d1"], p["d2"]
    cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE)
    cams["params"][:, 0] = 800.0
    ro = _capi.ransac_opt_from_dict({"max_iterations": 2000, "min_iterations": 2000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    h = _capi.Handle(0)
    try:
        for kind in (_capi.CALIB, _capi.FUNDAMENTAL_7PT):
            out = []
            for per in (None, "16", "5"):
                monkeypatch.delenv("MDRP_PAIRS_PER_PASS", raising=False) if per is None else monkeypatch.setenv("MDRP_PAIRS_PER_PASS", per)
                res, mask = h.estimate_batch(kind, x1, x2, d1, d2, ro, bo, n_per, cams if kind == _capi.CALIB else None, cams if kind == _capi.CALIB else None)
                out.append((res.copy(), mask.copy()))
            for r, m in out[1:]:
                assert r.tobytes() == out[0][0].tobytes() and np.array_equal(m, out[0][1]), kind
            assert int(out[0][0]["refinements"].max()) > 2
    finally:
        h.close()


def test_host_buffer_call_in_h2d_slices_equals_the_resident_call():
    """a host-buffer call of >= 512 pairs copies the correspondences in 256-pair slices on the handle's copy stream and runs k_prep,
    the first chunk and the second chunk's solver of each slice beside the next slice's copy (mdrp_capi.hip run_pass).  Pairs are independent units:
    records and masks must equal those of the SAME batch resident on the device bit for bit — ragged counts (incl. pairs below the sample size), per-pair
    cameras, a last slice that is short (600 = 256 + 256 + 88), for a monodepth estimator and a classic one."""
    import torch
    from mdrp_amd import _capi, synth
    B, N = 600, 640
    rng = np.random.default_rng(77)
    ns = rng.integers(200, N + 1, size=B).astype(np.int32)
    ns[[5, 300, 599]] = [2, 0, 3]
    x1, x2 = np.zeros((B, N, 2)), np.zeros((B, N, 2))
    d1, d2 = np.ones((B, N)), np.ones((B, N))
    for i in range(B):
        n = int(ns[i])
        if n:
            p = synth.make_pair(8800 + i, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4)
            x1[i, :n], x2[i, :n], d1[i, :n], d2[i, :n] = p["x1"], p["x2"], p["d1"], p["d2"]
    cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 700.0 + rng.uniform(0, 200, B)
    ro = _capi.ransac_opt_from_dict({"max_iterations": 1500, "min_iterations": 1500, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    dev = torch.device("cuda", 0)
    h = _capi.Handle(0)
    try:
        for kind in (0, 5):
            c = cams if kind == 0 else None
            res_h, mask_h = h.estimate_batch(kind, x1, x2, d1 if kind == 0 else None, d2 if kind == 0 else None, ro, bo, ns, c, c)
            t = [torch.from_numpy(a).to(dev) for a in (x1, x2, d1, d2)]
            mask_d = torch.zeros((B, N), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)
            h.estimate_batch_device(kind, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr() if kind == 0 else 0, t[3].data_ptr() if kind == 0 else 0, B, N, ro, bo, ns, c, c,
                                    mask_d.data_ptr())
            res_d = h.fetch_results(B)
            assert res_h.tobytes() == res_d.tobytes(), kind
            assert np.array_equal(mask_h, mask_d.cpu().numpy()), kind
            assert int(res_h["num_inliers"].max()) > 200
    finally:
        h.close()


def test_first_chunk_follows_the_inlier_ratio_of_the_previous_call_and_changes_no_result():
    """The first chunk of a run (scored exactly in full: no bar yet) is sized from the results of the handle's previous call with the same estimator
    (mdrp_capi.hip run_pass; mdrp_stats::first_chunk): 256 iterations on a fresh handle; afterwards the mean over the previous call's pairs of
    6 / r^3 (r: the pair's inlier ratio; between 256 and 1024, 128 for a nearly outlier-free pair) — 80 % outliers: ~750; 50 %: 256; none: 128 — for
    runs of at least 8192 certain iterations, per estimator, never under MDRP_CHUNKS.  Records and masks of the same batch are identical whatever
    the length was."""
    from mdrp_amd import _capi, synth
    B, N = 48, 600
    ro = _capi.ransac_opt_from_dict({"max_iterations": 8192, "min_iterations": 8192, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    ro_short = _capi.ransac_opt_from_dict({"max_iterations": 2000, "min_iterations": 2000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0})
    bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0

    def batch(of, rf=None):
        b = synth.make_batch(4200, B, N, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf)
        return b["x1"], b["x2"], b["d1"], b["d2"]

    dirty, half, clean, shared = batch(0.8), batch(0.5), batch(0.0), batch(0.8, "shared")
    h = _capi.Handle(0)
    try:
        def run(kind, data, opt=ro):
            c = cams if kind == 0 else None
            res, mask = h.estimate_batch(kind, *data, opt, bo, None, c, c)
            return res.tobytes(), mask.tobytes(), int(h.last_stats()["first_chunk"])
        r1 = run(0, dirty)
        assert r1[2] == 256                                    # fresh handle
        r2 = run(0, dirty)
        assert 512 <= r2[2] <= 896 and r2[:2] == r1[:2], r2[2]  # r ~ 0.2: 6 / r^3 = 750 -> 768 (0.19 ... 0.22: 896 ... 576); the same records and masks
        assert run(0, dirty, ro_short)[2] == 128                # a short run keeps a sixteenth of its certain iterations, at least 128
        r3 = run(1, shared)
        assert r3[2] == 256                                    # another estimator: its own history
        r4 = run(0, half)
        assert r4[2] == r2[2]                                  # (sized by the calibrated call before it)
        r5 = run(0, half)
        assert r5[2] == 256 and r5[:2] == r4[:2], r5[2]         # r ~ 0.5: 48 -> 256
        run(0, clean)
        assert run(0, clean)[2] == 128                          # r ~ 1: 6 -> 128
        assert 512 <= run(1, shared)[2] <= 896                  # the shared-focal estimator's second call
        # the shift solver runs its final refinements unfused, behind the last progress record: its sums arrive with an event the next call looks at
        ro_shift = _capi.ransac_opt_from_dict({"max_iterations": 8192, "min_iterations": 8192, "max_epipolar_error": 2.0, "max_reproj_error": 16.0,
                                               "monodepth_estimate_shift": True})
        run(0, clean)                                          # (the calibrated estimators share one history: back to 128 ...)
        assert run(0, dirty, ro_shift)[2] == 128
        assert 512 <= run(0, dirty, ro_shift)[2] <= 896         # ... and up again from the shift solver's own results
        # the comparison rows keep fixed schedules: 5-point a sixteenth of the certain iterations up to 512, 7-point 128 (| 1024 | rest)
        cl = lambda kind, opt: (h.estimate_batch(kind, half[0], half[1], None, None, opt, bo, None, cams if kind == 3 else None, cams if kind == 3 else None),
                                int(h.last_stats()["first_chunk"]))[1]
        assert [cl(3, ro), cl(3, ro_short), cl(5, ro), cl(5, ro_short)] == [512, 128, 128, 128]
    finally:
        h.close()


def test_local_shard_through_the_device_gather_path_one_rank():
    """BASELINE configs[4]'s data path on ONE GPU: dist.estimate_local_shard_device — device-resident inputs,
    mdrp_estimate_batch_async, mdrp_copy_results_device into the rank's slot, all_gather_into_tensor over RCCL (a one-rank `nccl`
    group, the collective forced) — returns the records the host-buffer API returns, bit for bit; the multi-rank bookkeeping
    (uneven blocks, padding) is covered under gloo in tests/test_dist_gloo.py."""
    import os
    import socket
    import torch
    import torch.distributed as tdist
    from mdrp_amd import _capi, dist as mdist, synth
    B, N = 37, 400
    b = synth.make_batch(9900, B, N, noise_px=0.5, depth_noise=0.02, outlier_frac=0.4)
    cams = np.zeros(B, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = {"max_iterations": 800, "min_iterations": 800, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    bo = {"loss_type": "TRUNCATED_CAUCHY"}
    ref, ref_mask = _capi.default_handle(0).estimate_batch(0, b["x1"], b["x2"], b["d1"], b["d2"], _capi.ransac_opt_from_dict(ro), _capi.bundle_opt_from_dict(bo), None, cams, cams)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
    mask = torch.zeros((B, N), dtype=torch.uint8, device=dev)
    created = False
    if not tdist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        tdist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        out = mdist.estimate_local_shard_device(0, B, *t, ro, bo, None, cams, cams, mask=mask, force_collective=True)
        # what bench.py puts on its N > 1 line: every rank's physical device, gathered over RCCL (here: one rank, the collective forced)
        seen = mdist.gather_device_identities(0, force_collective=True)
        assert seen == [(0, mdist.device_identity(0))] and len(seen[0][1]) > 8 and mdist.local_device_index(5) == 5 % torch.cuda.device_count()
    finally:
        if created:
            tdist.destroy_process_group()
    assert out.tobytes() == ref.tobytes() and np.array_equal(mask.cpu().numpy(), ref_mask)


def test_prosac_is_refused_on_every_entry_point_including_the_c_abi():
    """`progressive_sampling=True` switches the reference to PROSAC (RandomSampler::initialize_prosac @0x4f8a20) — different
    samples, different results.  It is not built, so it must be refused, not dropped: MDRP_ERR_UNSUPPORTED (4) from the C ABI (ABI 0.4:
    mdrp_ransac_opt carries the reference's progressive_sampling / max_prosac_iterations / real_focal_check), NotImplementedError from
    every Python entry point — the three monodepth estimators included, which ignored the key until round 4."""
    import ctypes as C
    from mdrp_amd import _capi, poselib, synth
    p = synth.make_pair(4242, 300, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3)
    pros = {**RO, "progressive_sampling": True, "max_prosac_iterations": 1000}
    for call in (lambda: poselib.estimate_monodepth_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], CAM, CAM, pros, BO),
                 lambda: poselib.estimate_monodepth_shared_focal_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], pros, BO),
                 lambda: poselib.estimate_monodepth_varying_focal_relative_pose(p["x1"], p["x2"], p["d1"], p["d2"], pros, BO),
                 lambda: poselib.estimate_monodepth_relative_pose_batch([p["x1"]], [p["x2"]], [p["d1"]], [p["d2"]], [CAM], [CAM], pros, BO),
                 lambda: poselib.estimate_relative_pose(p["x1"], p["x2"], CAM, CAM, pros, BO),
                 lambda: poselib.estimate_fundamental(p["x1"], p["x2"], pros, BO),
                 lambda: poselib.estimate_shared_focal_relative_pose(p["x1"], p["x2"], [0.0, 0.0], pros, BO)):
        with pytest.raises(NotImplementedError):
            call()
    # the C ABI itself: return code 4 and a message, for every kind; max_prosac_iterations alone changes nothing
    lib = _capi.load_library()
    h = _capi.Handle(0)
    try:
        x1 = np.ascontiguousarray(p["x1"][None]); x2 = np.ascontiguousarray(p["x2"][None]); d1 = np.ascontiguousarray(p["d1"][None]); d2 = np.ascontiguousarray(p["d2"][None])
        cams = np.zeros(1, dtype=_capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
        out = np.zeros(1, dtype=_capi.RESULT_DTYPE)
        bo = _capi.bundle_opt_from_dict(BO)
        for kind in range(6):
            ro = _capi.ransac_opt_from_dict(pros)
            rc = lib.mdrp_estimate_batch(h._h, kind, 0, _capi._ptr(x1), _capi._ptr(x2), _capi._ptr(d1), _capi._ptr(d2), 1, 300, None, _capi._ptr(cams), _capi._ptr(cams),
                                         C.byref(ro), C.byref(bo), _capi._ptr(out), None)
            assert rc == 4 and b"PROSAC" in lib.mdrp_last_error(), (kind, rc)
        for kind in (4, 5):
            ro = _capi.ransac_opt_from_dict({**RO, "real_focal_check": True})
            rc = lib.mdrp_estimate_batch(h._h, kind, 0, _capi._ptr(x1), _capi._ptr(x2), None, None, 1, 300, None, _capi._ptr(cams), _capi._ptr(cams),
                                         C.byref(ro), C.byref(bo), _capi._ptr(out), None)
            assert rc == 4, (kind, rc)
        a = h.estimate_batch(0, x1, x2, d1, d2, _capi.ransac_opt_from_dict(RO), bo, None, cams, cams)[0]
        b_ = h.estimate_batch(0, x1, x2, d1, d2, _capi.ransac_opt_from_dict({**RO, "max_prosac_iterations": 7, "real_focal_check": True}), bo, None, cams, cams)[0]
        assert a.tobytes() == b_.tobytes()  # without progressive_sampling the PROSAC budget is not read; real_focal_check is the baselines' switch
    finally:
        h.close()


C_HOST = r"""
/* a plain C host of libmdrp_hip.so (INTEGRATION.md 3): one image pair from a file through mdrp_estimate_batch, record to stdout */
#include "mdrp.h"
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char **argv) {
    if (argc < 3 || mdrp_abi_version() != MDRP_ABI_VERSION) return 2;
    const int n = atoi(argv[2]);
    double *buf = malloc(sizeof(double) * 6 * (size_t)n); /* x1 [n][2] | x2 [n][2] | d1 [n] | d2 [n] */
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(buf, sizeof(double), 6 * (size_t)n, f) != 6 * (size_t)n) return 3;
    fclose(f);
    mdrp_ransac_opt ro = {10000, 10000, 3.0, 0.9999, 16.0, 2.0, 0, 0, 1.0f, 0, 0, 100000, 0, 0};
    mdrp_bundle_opt bo = {100, 4 /* TRUNCATED_CAUCHY */, 1.0, 1e-10, 1e-8, 1e-3, 1e-10, 1e10};
    mdrp_camera cam = {0, 0, {800.0, 0.0, 0.0, 0.0}};
    mdrp_handle *h = NULL;
    mdrp_result r;
    unsigned char *mask = malloc((size_t)n);
    int rc = mdrp_create(0, NULL, &h);
    if (!rc) rc = mdrp_estimate_batch(h, MDRP_CALIB, MDRP_MEM_HOST, buf, buf + 2 * n, buf + 4 * n, buf + 5 * n, 1, n, NULL, &cam, &cam, &ro, &bo, &r, mask);
    if (rc) { fprintf(stderr, "mdrp error %d: %s\n", rc, mdrp_last_error()); return 1; }
    ro.progressive_sampling = 1;
    if (mdrp_estimate_batch(h, MDRP_CALIB, MDRP_MEM_HOST, buf, buf + 2 * n, buf + 4 * n, buf + 5 * n, 1, n, NULL, &cam, &cam, &ro, &bo, &r, mask) != MDRP_ERR_UNSUPPORTED) return 4;
    int inl = 0;
    for (int i = 0; i < n; ++i) inl += mask[i];
    printf("%llu %llu %llu %d %.17g %.17g", (unsigned long long)r.refinements, (unsigned long long)r.iterations, (unsigned long long)r.num_inliers, inl, r.inlier_ratio, r.model_score);
    for (int i = 0; i < 4; ++i) printf(" %.17g", r.model.q[i]);
    for (int i = 0; i < 3; ++i) printf(" %.17g", r.model.t[i]);
    printf(" %.17g %.17g %.17g\n", r.model.scale, r.model.shift1, r.model.shift2);
    mdrp_destroy(h);
    return 0;
}
"""


def test_c_host_runs_an_estimate_on_the_gpu(tmp_path, golden):
    """a C host — no Python, no ctypes, its own libamdhip64 — compiles against include/mdrp.h, links libmdrp_hip.so,
    estimates pair 3 of the headline batch from a file and prints the record: iterations, inliers, mask count and model equal the
    REFERENCE binary's record for that pair (tests/golden/headline_ref_calib_p3p_n2000_i10k.npz)."""
    import os
    import shutil
    import subprocess
    from mdrp_amd import _capi, synth
    from helpers import model_diff
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    assert shutil.which("gcc") and os.path.exists(os.path.join(rocm, "lib", "libamdhip64.so")), "the image has gcc and a system HIP runtime"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    idx, n = 3, 2000
    p = synth.make_pair(idx, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
    np.concatenate([p["x1"].ravel(), p["x2"].ravel(), p["d1"].ravel(), p["d2"].ravel()]).astype(np.float64).tofile(tmp_path / "pair.bin")
    (tmp_path / "host.c").write_text(C_HOST)
    libdir = os.path.dirname(_capi.LIB_PATH)
    subprocess.run(["gcc", "-O1", str(tmp_path / "host.c"), "-I", os.path.join(root, "include"), "-L", libdir, "-lmdrp_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                    f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", f"-Wl,-rpath,{libdir}", "-o", str(tmp_path / "host")], check=True)
    out = subprocess.run([str(tmp_path / "host"), str(tmp_path / "pair.bin"), str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    v = out.stdout.split()
    ref = golden("headline_ref_calib_p3p_n2000_i10k")
    assert (int(v[1]), int(v[2]), int(v[3])) == (int(ref["istats"][idx, 1]), int(ref["istats"][idx, 2]), int(ref["istats"][idx, 2]))
    assert int(v[0]) == int(ref["istats"][idx, 0])  # pair 3 is not among the enumerated LO-count deviations
    assert abs(float(v[5]) / ref["fstats"][idx, 1] - 1.0) < 1e-9 and float(v[4]) == ref["fstats"][idx, 0]
    model = np.r_[[float(x) for x in v[6:16]], 1.0, 1.0]
    assert model_diff(model, ref["model"][idx]) < 1e-6
