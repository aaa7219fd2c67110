"""The batches bench.py times, pair by pair, through the entry point bench.py times.

Round 5: tests/golden/headline_ref_<workload>.npz (tests/tools/gen_golden_headline_ref.py) holds the output of the REFERENCE's own
PoseLib binary for every one of the 1024 pairs of the four timed workloads (calibrated P3P, calibrated with shifts, shared focal,
varying focal N = 5000) and `test_every_pair_of_the_bench_batch_vs_reference_fixture` compares the HIP path with those directly; the
oracle fixture below stays as the second witness.  Where the reference and the oracle differ (LO count only, causes enumerated per pair
in tests/golden/headline_ref_deviations.json by tests/tools/classify_ref_deviations.py) the test asserts the exact list.

tests/golden/headline_<workload>.npz (tests/tools/gen_golden_headline.py) holds the CPU oracle's output for EVERY pair of
the 1024-pair batches of BASELINE.json configs[1..3]: refinements, iterations, num_inliers, inlier_ratio, model_score, the
12-wide model and the packed inlier mask.  The oracle itself is pinned against the reference binary on 32 pairs spread over
each of these batches (tests/golden/estimate_wide.npz).  Here the same batches go through `mdrp_estimate_batch_async` on
DEVICE-RESIDENT buffers + `mdrp_fetch_results` — what bench.py's timed region calls — and every pair is compared.
A soak test repeats the headline step 200 times and demands bit-identical records and masks (fused tail on: the
stale-read race of round 3 showed once in ~10^4 pairs).  Needs an MI355X:  pytest -m gpu."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import model_diff

pytestmark = pytest.mark.gpu

WORKLOADS = {
    # name: (kind, shift flag, n, outlier_frac, random_focal, depth shifts)  == bench.py WORKLOADS / gen_golden_headline.HEADLINE
    "calib_p3p_n2000_i10k": (0, False, 2000, 0.5, None, (0.0, 0.0)),
    "calib_shift_n2000_i10k": (0, True, 2000, 0.5, None, (0.2, -0.1)),
    "shared_n2000_i10k": (1, False, 2000, 0.5, "shared", (0.0, 0.0)),
    "varying_n5000_i10k": (2, True, 5000, 0.5, "varying", (0.0, 0.0)),
}
RO = {"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}


@pytest.fixture(scope="module")
def capi():
    from mdrp_amd import _capi
    return _capi


def _digest(b, i):
    h = hashlib.sha256()
    for k in ("x1", "x2", "d1", "d2"):
        h.update(np.ascontiguousarray(b[k][i], dtype=np.float64).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


class DeviceBatch:
    """a synthetic batch uploaded once; run() = one bench.py step: mdrp_estimate_batch_async on the device pointers, then fetch"""

    def __init__(self, capi, workload, B=1024):
        import torch
        from mdrp_amd import synth
        self.capi, self.torch = capi, torch
        self.kind, es, self.n, of, rf, (s1, s2) = WORKLOADS[workload]
        self.B = B
        self.host = synth.make_batch(0, B, self.n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf, shift1=s1, shift2=s2)
        dev = torch.device("cuda", 0)
        self.t = [torch.from_numpy(self.host[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
        self.mask = torch.zeros((B, self.n), dtype=torch.uint8, device=dev)
        self.cams = np.zeros(B, dtype=capi.CAMERA_DTYPE)
        self.cams["params"][:, 0] = 800.0
        self.ro = capi.ransac_opt_from_dict({**RO, "monodepth_estimate_shift": es})
        self.bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
        self.h = capi.Handle(0)
        torch.cuda.synchronize(dev)  # uploads ran on torch's stream, the handle has its own

    def run(self):
        c = self.cams if self.kind == 0 else None
        self.h.estimate_batch_device(self.kind, *(x.data_ptr() for x in self.t), self.B, self.n, self.ro, self.bo, None, c, c, self.mask.data_ptr())
        return self.h.fetch_results(self.B)

    def close(self):
        self.h.close()


@pytest.mark.parametrize("workload", list(WORKLOADS))
def test_every_pair_of_the_bench_batch_vs_oracle_fixture(capi, golden, workload):
    """All 1024 pairs of the batch bench.py times for this workload, through the device-resident entry point: iterations,
    inlier count and inlier mask identical on every pair, model to 1e-6 (north_star's tolerance; measured 1e-9), score to 1e-9.
    The LO count (`refinements`) equals the oracle's except on the rounding-tie class of DESIGN.md §5 (v): two scores that
    agree to ~1e-14 compared with `<`, decided by FMA contraction — the pairs are listed in GPU_MINUS_ORACLE_LO (exact list),
    with everything else on those pairs identical."""
    g = golden(f"headline_{workload}")
    db = DeviceBatch(capi, workload)
    try:
        for i in range(0, db.B, 37):
            assert _digest(db.host, i) == g["digest"][i], "synthetic generator drifted"
        res = db.run()
        mask = db.mask.cpu().numpy()
    finally:
        db.close()
    n = db.n
    ist, fst = g["istats"], g["fstats"]
    assert np.array_equal(res["iterations"].astype(np.int64), ist[:, 1]) and (ist[:, 1] == 10000).all()
    bad_cnt = np.nonzero(res["num_inliers"].astype(np.int64) != ist[:, 2])[0]
    assert len(bad_cnt) == 0, (workload, bad_cnt[:8], res["num_inliers"][bad_cnt[:8]], ist[bad_cnt[:8], 2])
    ref_mask = np.unpackbits(g["mask"], axis=1)[:, :n]
    bad_mask = np.nonzero((mask != ref_mask).any(axis=1))[0]
    assert len(bad_mask) == 0, (workload, bad_mask[:8])
    assert (mask.sum(axis=1) == res["num_inliers"]).all()
    worst = 0.0
    for i in range(db.B):
        d = model_diff(capi.model_to_array(res[i]["model"]), g["model"][i])
        worst = max(worst, d)
        assert d < 1e-6, (workload, i, d)
    assert np.allclose(res["model_score"], fst[:, 1], rtol=1e-9, atol=0), workload
    assert np.allclose(res["inlier_ratio"], fst[:, 0], rtol=1e-12, atol=0), workload
    dlo = res["refinements"].astype(np.int64) - ist[:, 0]
    off = np.nonzero(dlo)[0]
    assert dict(zip(off.tolist(), dlo[off].tolist())) == GPU_MINUS_ORACLE_LO[workload], (workload, off, dlo[off])
    print(f"{workload}: 1024 / 1024 pairs identical (iterations, inliers, mask); worst model diff {worst:.2e}; LO count differs on {len(off)} pairs {off.tolist()}")


# ---- against the reference binary's own output for every pair (round 5) -------------------------------------------------------------
# LO count (`refinements`) of the HIP path minus the CPU oracle's, by pair: the rounding-tie class of DESIGN.md 5 (v) — two scores that
# agree to ~1e-14 compared with `<`, decided by FMA contraction / the LM's summation tree.  Everything else on these pairs is identical.
# An explicit list, not a budget: a build that moves a tie fails here and the list is updated deliberately.
GPU_MINUS_ORACLE_LO = {
    "calib_p3p_n2000_i10k": {},
    "calib_shift_n2000_i10k": {},
    "shared_n2000_i10k": {957: -1},  # (here the HIP path agrees with the REFERENCE and the oracle does not: right by an accident of contraction, not by construction)
    "varying_n5000_i10k": {},
}
# pairs whose final MODEL differs from the reference's by more than north_star's 1e-6: the reference's relpose_monodepth_3pt returns a NaN
# model where ours returns a true root of the minimal problem (residual 4e-16); that root's LO ends 5e-8 lower in score, the RANSAC winner
# differs in the 8th digit of its score, and the inlier-only refinement of the weakly observable shifts stops 1.2e-3 apart
# (tests/tools/diag_headline_pair.py calib_shift_n2000_i10k 897; inliers and mask identical).  On these pairs the HIP path must equal the oracle.
MODEL_FOLLOWS_ORACLE = {"calib_shift_n2000_i10k": (897,)}


@pytest.mark.parametrize("workload", list(WORKLOADS))
def test_every_pair_of_the_bench_batch_vs_reference_fixture(capi, golden, workload):
    """All 1024 pairs of the batch bench.py times for this workload against the output of the reference's own PoseLib binary
    (estimate_monodepth_relative_pose @0x224170 / estimate_shared_focal_... @0x223300 / estimate_varying_focal_... @0x223a40;
    options of /root/reference/make_video.py:192-196): iterations, inlier count and inlier mask identical on every pair, model
    within 1e-6, model_score to 1e-9; `refinements` equal to the reference's except on the pairs enumerated, with their cause, in
    tests/golden/headline_ref_deviations.json (reference solver returns NaN / misses a root / returns an extra root; rounding ties) and
    in GPU_MINUS_ORACLE_LO above — asserted as an exact list."""
    ref = golden(f"headline_ref_{workload}")
    orc = golden(f"headline_{workload}")
    dev = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "headline_ref_deviations.json")))["deviations"][workload]
    db = DeviceBatch(capi, workload)
    try:
        for i in range(0, db.B, 37):
            assert _digest(db.host, i) == ref["digest"][i] == orc["digest"][i], "synthetic generator drifted"
        res = db.run()
        mask = db.mask.cpu().numpy()
    finally:
        db.close()
    n = db.n
    ist, fst = ref["istats"], ref["fstats"]
    assert np.array_equal(res["iterations"].astype(np.int64), ist[:, 1]) and (ist[:, 1] == 10000).all()
    bad_cnt = np.nonzero(res["num_inliers"].astype(np.int64) != ist[:, 2])[0]
    assert len(bad_cnt) == 0, (workload, bad_cnt[:8], res["num_inliers"][bad_cnt[:8]], ist[bad_cnt[:8], 2])
    ref_mask = np.unpackbits(ref["mask"], axis=1)[:, :n]
    bad_mask = np.nonzero((mask != ref_mask).any(axis=1))[0]
    assert len(bad_mask) == 0, (workload, bad_mask[:8])
    assert np.allclose(res["inlier_ratio"], fst[:, 0], rtol=1e-12, atol=0), workload
    follows = set(MODEL_FOLLOWS_ORACLE.get(workload, ()))
    worst = 0.0
    for i in range(db.B):
        want = orc["model"][i] if i in follows else ref["model"][i]
        d = model_diff(capi.model_to_array(res[i]["model"]), want)
        worst = max(worst, d)
        assert d < 1e-6, (workload, i, d)
        if i in follows:
            assert str(i) in dev and "ref_" in dev[str(i)]["cause"], (workload, i)
    # the RANSAC winner's score: where the reference's solver lost a root the winner can differ in the last digits (oracle and HIP path agree there)
    sc_ref_ok = np.isclose(res["model_score"], fst[:, 1], rtol=1e-9, atol=0)
    sc_orc_ok = np.isclose(res["model_score"], orc["fstats"][:, 1], rtol=1e-9, atol=0)
    assert sc_orc_ok.all(), (workload, np.nonzero(~sc_orc_ok)[0])
    for i in np.nonzero(~sc_ref_ok)[0]:
        assert str(i) in dev and "ref_" in dev[str(i)]["cause"], (workload, int(i), res["model_score"][i], fst[i, 1])
    expected = np.zeros(db.B, dtype=np.int64)
    for k, v in dev.items():
        expected[int(k)] += v["oracle_minus_reference"]
    for k, v in GPU_MINUS_ORACLE_LO[workload].items():
        expected[k] += v
    dlo = res["refinements"].astype(np.int64) - ist[:, 0]
    wrong = np.nonzero(dlo != expected)[0]
    assert len(wrong) == 0, (workload, {int(i): (int(dlo[i]), int(expected[i])) for i in wrong})
    off = np.nonzero(dlo)[0]
    print(f"{workload} vs REFERENCE binary: 1024 / 1024 pairs identical (iterations, inliers, mask); worst model diff {worst:.2e}; score differs beyond 1e-9 on "
          f"{int((~sc_ref_ok).sum())} pairs; LO count differs on {len(off)} pairs (all enumerated): {dict(zip(off.tolist(), dlo[off].tolist()))}")


def test_headline_step_soak_is_bit_identical(capi):
    """200 consecutive steps of the headline batch (1024 pairs, N = 2000, 10^4 iterations; fused tail on, as bench.py runs it):
    records and masks of every step bit-identical to step 0's."""
    db = DeviceBatch(capi, "calib_p3p_n2000_i10k")
    try:
        r0 = db.run().tobytes()
        m0 = db.mask.clone()
        for step in range(1, 200):
            r = db.run().tobytes()
            assert r == r0, f"step {step}: records differ"
            assert db.torch.equal(db.mask, m0), f"step {step}: masks differ"
    finally:
        db.close()


def test_large_batches_are_pipelined_by_the_entry_points(capi, golden, monkeypatch):
    """8192 headline pairs (the 1024-pair bench batch eight times over) in ONE call of the drop-in entry points.
    `estimate_monodepth_relative_pose_batch(..., as_arrays=True)` from host buffers cuts the batch into 1024-pair chunks that run two in flight
    (mdrp_amd.pipeline: a chunk's H2D copy beside the previous chunk's kernels); the same chunking on resident tensors (pipeline.estimate_device)
    gives the same records and masks bit for bit, and every copy of a pair gets the same record whatever chunk it falls into.
    `estimate_batch_torch` runs the resident batch as one call (measured fastest); its records agree with the chunked ones to 1e-9 (the
    final refinements use 64 instead of 256 lanes from 4096 pairs on: another summation tree).  Every 64th pair of both is compared with the
    REFERENCE binary's fixture by index modulo 1024.  Prints the pairs/s of both calls (profiles/r05_python_entry_points.txt)."""
    import time
    import torch
    from mdrp_amd import pipeline, poselib, synth
    ref = golden("headline_ref_calib_p3p_n2000_i10k")
    n, rep = 2000, 8
    b = synth.make_batch(0, 1024, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
    x1, x2, d1, d2 = (np.ascontiguousarray(np.concatenate([b[k]] * rep)) for k in ("x1", "x2", "d1", "d2"))
    B = len(x1)
    monkeypatch.setattr(pipeline, "PIPELINE_MIN", 6144)  # (MDRP_PIPELINE_MIN=6144; off by default since round 6: one host call copies in slices by itself)
    assert pipeline.chunk_bounds(B) == [(1024 * i, 1024 * (i + 1)) for i in range(8)]
    cam = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 0.0, 0.0]}
    bo = {"loss_type": "TRUNCATED_CAUCHY"}
    poselib.estimate_monodepth_relative_pose_batch(x1[:6200], x2[:6200], d1[:6200], d2[:6200], cam, cam, RO, bo, as_arrays=True)  # warm-up: pipeline handles, scratch
    t0 = time.perf_counter()
    res, mask, ns = poselib.estimate_monodepth_relative_pose_batch(x1, x2, d1, d2, cam, cam, RO, bo, as_arrays=True)
    t_host = time.perf_counter() - t0
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(a).to(dev) for a in (x1, x2, d1, d2)]
    torch.cuda.synchronize()
    # the same chunks on resident tensors: bit-identical to the host-buffer call
    mask_c = torch.zeros((B, n), dtype=torch.uint8, device=dev)
    cams = poselib._camera_records(cam, B)
    res_c = pipeline.estimate_device(0, *(x.data_ptr() for x in t), B, n, capi.ransac_opt_from_dict(RO), capi.bundle_opt_from_dict(bo), None, cams, cams, mask_c.data_ptr(), 0)
    assert res.tobytes() == res_c.tobytes() and np.array_equal(mask, mask_c.cpu().numpy())
    first = res[:1024].tobytes()
    for r in range(1, rep):
        assert res[1024 * r:1024 * (r + 1)].tobytes() == first, f"copy {r} differs from copy 0"
        assert np.array_equal(mask[1024 * r:1024 * (r + 1)], mask[:1024])
    # one call on the resident batch
    poselib.estimate_batch_torch("calibrated", *t, cam, cam, RO, bo)
    t0 = time.perf_counter()
    res_d, mask_d = poselib.estimate_batch_torch("calibrated", *t, cam, cam, RO, bo)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    mask_dn = mask_d.cpu().numpy()
    assert np.array_equal(res_d["iterations"], res["iterations"]) and np.array_equal(res_d["num_inliers"], res["num_inliers"]) and np.array_equal(mask_dn, mask)
    ref_mask = np.unpackbits(ref["mask"], axis=1)[:, :n]
    for i in range(0, B, 64):
        j = i % 1024
        for rr, mm in ((res, mask), (res_d, mask_dn)):
            assert int(rr[i]["iterations"]) == ref["istats"][j, 1] and int(rr[i]["num_inliers"]) == ref["istats"][j, 2], i
            assert np.array_equal(mm[i], ref_mask[j]), i
            assert model_diff(capi.model_to_array(rr[i]["model"]), ref["model"][j]) < 1e-6, i
        assert model_diff(capi.model_to_array(res_d[i]["model"]), capi.model_to_array(res[i]["model"])) < 1e-7, i
    print(f"8192 headline pairs in one call of the Python entry points: {B / t_dev:.0f} pairs/s on resident tensors (estimate_batch_torch, one call), "
          f"{B / t_host:.0f} pairs/s from pageable host buffers (estimate_monodepth_relative_pose_batch as_arrays: 1024-pair chunks two in flight)")


def test_emulated_eight_way_split_equals_one_call_and_the_reference(capi, golden):
    """VERDICT r04 item 7 — BASELINE configs[4]'s bookkeeping on ONE GPU: 1100 headline pairs are cut by mdrp_amd.dist.shard_bounds into the
    blocks eight ranks would own (138 pairs each, 134 in the last); the blocks run one after the other through the device-resident path of a rank
    (mdrp_estimate_batch_async -> mdrp_copy_results_device into a zero-padded block of ceil(P / 8) records), the eight padded blocks are laid out
    as all_gather_into_tensor lays them out and trimmed with dist._unpad — the code bench.py --gpus 8 --total-pairs runs — and the result is
    compared, bit for bit, with ONE call over all 1100 pairs, and pair by pair (index modulo 1024) with the reference binary's fixture."""
    import torch
    from mdrp_amd import dist as mdist, synth
    ref = golden("headline_ref_calib_p3p_n2000_i10k")
    total, world, n = 1100, 8, 2000
    b = synth.make_batch(0, 1024, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
    idx = np.arange(total) % 1024
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(dev) for k in ("x1", "x2", "d1", "d2")]
    cams = np.zeros(total, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = capi.ransac_opt_from_dict(RO); bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    h = capi.Handle(0)
    torch.cuda.synchronize(dev)
    try:
        mask_one = torch.zeros((total, n), dtype=torch.uint8, device=dev)
        h.estimate_batch_device(0, *(x.data_ptr() for x in t), total, n, ro, bo, None, cams, cams, mask_one.data_ptr())
        one = h.fetch_results(total)
        per = mdist.shard_bounds(total, 0, world)[2]
        gathered = torch.zeros((world * per, mdist.RECORD_BYTES), dtype=torch.uint8, device=dev)  # the layout all_gather_into_tensor produces
        mask_sh = torch.zeros((total, n), dtype=torch.uint8, device=dev)
        for r in range(world):
            lo, hi, per_r = mdist.shard_bounds(total, r, world)
            assert per_r == per and hi - lo in (138, 134)
            h.estimate_batch_device(0, *(x[lo:hi].data_ptr() for x in t), hi - lo, n, ro, bo, None, cams[lo:hi], cams[lo:hi], mask_sh[lo:hi].data_ptr())
            h.copy_results_device(gathered[r * per:].data_ptr(), hi - lo)  # the rank's slot; rows past hi - lo stay zero (padding)
        rows = mdist._unpad(gathered.cpu().numpy(), total, world, per)
        sharded = np.ascontiguousarray(rows).reshape(-1).view(capi.RESULT_DTYPE)
    finally:
        h.close()
    assert len(sharded) == total and sharded.tobytes() == one.tobytes()
    assert torch.equal(mask_sh, mask_one)
    ref_mask = np.unpackbits(ref["mask"], axis=1)[:, :n]
    m = mask_sh.cpu().numpy()
    for i in range(total):
        j = i % 1024
        assert int(sharded[i]["iterations"]) == ref["istats"][j, 1] and int(sharded[i]["num_inliers"]) == ref["istats"][j, 2], i
        assert np.array_equal(m[i], ref_mask[j]), i
        assert model_diff(capi.model_to_array(sharded[i]["model"]), ref["model"][j]) < 1e-6, i


def test_one_rank_share_of_configs4_12500_pairs_vs_the_reference(capi, golden):
    """VERDICT r05 item 1 — BASELINE configs[4] at the size ONE rank sees: 100 000 pairs over 8 GPUs = 12 500 pairs per rank.  The 1024-pair headline
    batch tiled to 12 500 pairs goes through `dist.estimate_local_shard_device(force_collective=True)` — device-resident inputs, ONE
    mdrp_estimate_batch_async call (one pass, ~66 GB of scratch), mdrp_copy_results_device into the rank's slot, all_gather_into_tensor over RCCL
    (a one-rank `nccl` group, the collective forced): exactly what `bench.py --gpus 8 --total-pairs 100000` runs on each rank — and EVERY pair is
    compared with the reference binary's fixture by index modulo 1024: iterations, inlier count, mask identical, model within 1e-6."""
    import socket
    import time
    import torch
    import torch.distributed as tdist
    from mdrp_amd import dist as mdist, synth
    ref = golden("headline_ref_calib_p3p_n2000_i10k")
    total, n = 12500, 2000
    b = synth.make_batch(0, 1024, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.5)
    for i in range(0, 1024, 97):
        assert _digest(b, i) == ref["digest"][i], "synthetic generator drifted"
    idx = np.arange(total) % 1024
    dev = torch.device("cuda", 0)
    base = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
    tidx = torch.from_numpy(idx).to(dev)
    t = [x.index_select(0, tidx).contiguous() for x in base]  # tiled on the device: 1.2 GB of inputs, no 12x host copy
    del base
    mask = torch.zeros((total, n), dtype=torch.uint8, device=dev)
    cams = np.zeros(total, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = capi.ransac_opt_from_dict(RO); bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    created = False
    if not tdist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        tdist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        created = True
    h = capi.Handle(0)
    try:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = mdist.estimate_local_shard_device(0, total, *t, ro, bo, None, cams, cams, handle=h, mask=mask, force_collective=True)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        out2 = mdist.estimate_local_shard_device(0, total, *t, ro, bo, None, cams, cams, handle=h, mask=mask, force_collective=True)
        dt2 = time.perf_counter() - t0
        st = h.last_stats()
    finally:
        h.close()
        if created:
            tdist.destroy_process_group()
    assert len(out) == total and len(out2) == total
    if out.tobytes() != out2.tobytes():  # name the pairs and fields: a bare bytes comparison says nothing
        diff = np.nonzero((out.view(np.uint8).reshape(total, -1) != out2.view(np.uint8).reshape(total, -1)).any(axis=1))[0]
        detail = {int(p): {f: (out[p][f].tolist(), out2[p][f].tolist()) for f in ("refinements", "iterations", "num_inliers", "model_score")} for p in diff[:6]}
        raise AssertionError(f"two calls on the same inputs differ on {len(diff)} of {total} pairs: {diff[:12].tolist()} {detail}")
    ist = ref["istats"][idx]
    assert np.array_equal(out["iterations"].astype(np.int64), ist[:, 1]) and np.array_equal(out["num_inliers"].astype(np.int64), ist[:, 2])
    ref_mask = np.unpackbits(ref["mask"], axis=1)[:, :n]
    m = mask.cpu().numpy()
    for r0 in range(0, total, 1024):  # block by block: the comparison array stays at 2 MB
        blk = m[r0:r0 + 1024]
        assert np.array_equal(blk, ref_mask[:len(blk)]), r0
    worst = 0.0
    for i in range(total):
        d = model_diff(capi.model_to_array(out[i]["model"]), ref["model"][idx[i]])
        worst = max(worst, d)
        assert d < 1e-6, (i, d)
    # every copy of a pair gets the same record whatever its place in the batch
    first = out[:1024]
    for r0 in range(1024, total, 1024):
        blk = out[r0:r0 + 1024]
        assert blk.tobytes() == first[:len(blk)].tobytes(), r0
    print(f"one rank's share of BASELINE configs[4]: {total} pairs in one call through estimate_local_shard_device (collective forced): every pair = reference fixture "
          f"(iterations, inliers, mask; worst model diff {worst:.2e}); {total / dt2:.0f} pairs/s incl. the gather and the host copy of the records "
          f"(first call with scratch allocation {total / dt:.0f}); fused-tail time-outs {st.get('fuse_timeouts', 0)}")


@pytest.mark.parametrize("name", ["calib_p3p", "calib_shift", "shared", "varying"])
def test_dynamic_stopping_full_size_vs_reference_fixture(capi, golden, name):
    """Round 5: ransac<>'s DYNAMIC stopping rule at full size against the reference binary (tests/golden/dynamic_ref.npz,
    tests/tools/gen_golden_dynamic_ref.py): the reference's default iteration budget (max 100000, min 1000) on 96 pairs per estimator at 50-85 %
    outliers — the reference stops them between 1001 and 28714 iterations.  One batched call of the HIP path (several super-chunks, a host read-back of
    the walk's verdict after each): `iterations` and inlier count identical on every pair; inlier mask identical, model within 1e-6 and model_score to
    1e-9 on 383 of 384 (the other pair enumerated below with its cause, equal to the oracle);
    `refinements` may differ by the solver classes of DESIGN.md 5 (counted here: at most 1 pair in 12, as on the fixed-length fixtures)."""
    from test_oracle_golden import DYNAMIC_CASES, dynamic_pair
    g = golden("dynamic_ref")
    kind, es, n, rf, _ = DYNAMIC_CASES[name]
    ist, fst = g[f"{name}_istats"], g[f"{name}_fstats"]
    B = len(ist)
    pairs = [dynamic_pair(g, name, j) for j in range(B)]
    for j in range(0, B, 11):
        assert _digest({k: np.stack([p[k] for p in pairs]) for k in ("x1", "x2", "d1", "d2")}, j) == g[f"{name}_digest"][j], "synthetic generator drifted"
    x1, x2, d1, d2 = (np.ascontiguousarray(np.stack([p[k] for p in pairs])) for k in ("x1", "x2", "d1", "d2"))
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = capi.ransac_opt_from_dict({"max_iterations": 100000, "min_iterations": 1000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "monodepth_estimate_shift": es})
    h = capi.Handle(0)
    try:
        res, mask = h.estimate_batch(kind, x1, x2, d1, d2, ro, capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None, cams if kind == 0 else None, cams if kind == 0 else None)
    finally:
        h.close()
    assert np.array_equal(res["iterations"].astype(np.int64), ist[:, 1]), (name, np.nonzero(res["iterations"].astype(np.int64) != ist[:, 1])[0][:8])
    assert np.array_equal(res["num_inliers"].astype(np.int64), ist[:, 2]), name
    # Pairs on which the REFERENCE's own solver output changes the RANSAC winner (the shared-focal solver of the binary returns a root
    # ours does not at iteration 552 of pair 27 — tests/tools/classify_ref_deviations.py —, its records differ from there on, it refines 10 models
    # instead of 12 and ends on a winner whose score differs in the 5th digit: same iterations, same inlier count, 2 mask bits and 1.3e-4 in the model).
    # There the HIP path must equal the ORACLE, which is run here as the checker.
    follows_oracle = {"shared": (27,)}.get(name, ())
    ref_mask = np.unpackbits(g[f"{name}_mask"], axis=1)[:, :n]
    keep = np.array([j not in follows_oracle for j in range(B)])
    assert np.array_equal(mask[keep], ref_mask[keep]), (name, np.nonzero((mask != ref_mask).any(axis=1))[0])
    worst = max(model_diff(capi.model_to_array(res[j]["model"]), g[f"{name}_model"][j]) for j in range(B) if keep[j])
    assert worst < 1e-6, (name, worst)
    assert np.allclose(res["model_score"][keep], fst[keep, 1], rtol=1e-9, atol=0), name
    for j in follows_oracle:
        from oracle import pyorc as po
        ro_o = po.ransac_opt(max_iterations=100000, min_iterations=1000, max_epipolar_error=2.0, max_reproj_error=16.0, estimate_shift=es)
        cam_o = po.cam_flat(0, [800.0, 0.0, 0.0])
        m, st, mk = po.estimate(kind, x1[j], x2[j], d1[j], d2[j], ro_o, po.bundle_opt(loss_type=4), cam_o if kind == 0 else None, cam_o if kind == 0 else None)
        assert (int(res[j]["refinements"]), int(res[j]["iterations"]), int(res[j]["num_inliers"])) == (st.refinements, st.iterations, st.num_inliers), (name, j)
        assert np.array_equal(mask[j], mk) and model_diff(capi.model_to_array(res[j]["model"]), m) < 1e-6 and abs(res[j]["model_score"] / st.model_score - 1.0) < 1e-9, (name, j)
        assert int((mask[j] != ref_mask[j]).sum()) <= 4 and model_diff(m, g[f"{name}_model"][j]) < 1e-3, (name, j)  # ... and the reference is that close
    lo_off = np.nonzero(res["refinements"].astype(np.int64) != ist[:, 0])[0]
    assert len(lo_off) <= B // 12, (name, lo_off)
    print(f"dynamic stopping, {name}: {int(keep.sum())} / 96 pairs identical to the REFERENCE binary ({len(follows_oracle)} enumerated: equal to the oracle) (iterations {ist[:, 1].min()} ... {ist[:, 1].max()}, inliers, mask); worst model diff {worst:.2e}; "
          f"LO count differs on {len(lo_off)} pairs {lo_off.tolist()}")


def test_fresh_pairs_of_the_timed_shapes_vs_reference_fixture(capi, golden):
    """Round 6: pairs the timed batches do not contain — indices 20000 ... 20063 of the same generator, the first 64 of the 4096 per workload that
    tests/tools/stress_headline_ref.py ran through the reference binary (profiles/r06_stress_headline_16384_vs_reference.txt: 16 384 pairs, 16 375 identical) —
    in one call of the device-resident entry point per workload: iterations, inlier count, mask and model (1e-6) equal the reference's on every pair; the
    LO count too on the P3P and the varying-focal path (the shift and shared-focal solvers' reference root sets differ, DESIGN.md 5: reported, not asserted)."""
    import torch
    from mdrp_amd import synth
    g = golden("headline_ref_fresh")
    first, count = int(g["first"]), int(g["count"])
    dev = torch.device("cuda", 0)
    for w in [str(x) for x in g["names"]]:
        kind, es, n, of, rf, (s1, s2) = WORKLOADS[w]
        b = synth.make_batch(first, count, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf, shift1=s1, shift2=s2)
        t = [torch.from_numpy(b[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
        mask_t = torch.zeros((count, n), dtype=torch.uint8, device=dev)
        cams = np.zeros(count, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
        ro = capi.ransac_opt_from_dict(dict(RO, seed=0, monodepth_estimate_shift=es))
        bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
        c = cams if kind == 0 else None
        h = capi.Handle(0)
        torch.cuda.synchronize(dev)
        try:
            h.estimate_batch_device(kind, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), count, n, ro, bo, None, c, c, mask_t.data_ptr())
            res = h.fetch_results(count)
            mask = mask_t.cpu().numpy()
        finally:
            h.close()
        ist, rm = g[w + "_istats"], g[w + "_model"]
        assert np.array_equal(res["iterations"].astype(np.int64), ist[:, 1]) and np.array_equal(res["num_inliers"].astype(np.int64), ist[:, 2]), w
        assert np.array_equal(mask, np.unpackbits(g[w + "_mask"], axis=1)[:, :n]), w
        worst = max(model_diff(capi.model_to_array(res[i]["model"]), rm[i]) for i in range(count))
        assert worst < 1e-6, (w, worst)
        lo = np.nonzero(res["refinements"].astype(np.int64) != ist[:, 0])[0]
        if w in ("calib_p3p_n2000_i10k", "varying_n5000_i10k"):
            assert len(lo) == 0, (w, lo)
        print(f"{w}: fresh pairs {first} .. {first + count - 1} vs REFERENCE binary: {count} / {count} identical (iterations, inliers, mask); worst model diff {worst:.2e}; "
              f"LO count differs on {len(lo)} pairs")


# ---- the timed batches of the comparison rows and of the outlier-free shape against the reference binary (round 5) -------------------------------
BASELINE_SETS = {
    # workload (bench.py WORKLOADS): kind, pairs, outlier_frac, random_focal
    "relpose_5pt_n2000_i10k": (3, 1024, 0.5, None),
    "fundamental_7pt_n2000_i10k": (5, 1024, 0.5, None),
    "shared_6pt_n2000_i10k": (4, 256, 0.5, "shared"),
    "calib_p3p_n2000_i10k_clean": (0, 1024, 0.0, None),
}
# LO count of the HIP path minus the CPU oracle's (the oracle's own difference to the reference is stored per pair in the fixture: `oracle_refinements`)
GPU_MINUS_ORACLE_LO_BASELINES = {
    # 5-point: a degree-10 root at the edge of existence, found by the device's elimination order (LU in LDS) and not by the oracle's (Gauss-Jordan), sets a
    # record: one more LO, nothing else changes (the class of DESIGN.md 8a; 6 of 3840 pairs in the stress campaign).  On pair 939 the reference has it too.
    # (Until the null space moved into a kernel of its own — other fused multiply-adds in the elimination's last bits, DESIGN.md 8a — pairs 553 and 864
    # were in this list as well, against the reference: the batch's LO count now differs from the reference's on 11 pairs instead of 13 — the oracle's
    # own differences, stored in the fixture as `oracle_refinements`.)
    "relpose_5pt_n2000_i10k": {939: 1},
    "fundamental_7pt_n2000_i10k": {},
    "shared_6pt_n2000_i10k": {},
    "calib_p3p_n2000_i10k_clean": {},
}


# Pairs whose RANSAC winner's `model_score` differs from the reference's beyond 1e-9 (exact list).  5-point pair 519: 0.00662927 (HIP) against 0.00663005
# (reference and oracle) with the same 13 refinements, the same iterations, inlier count and mask and a final model within 1e-6: one of the pair's LO
# refinements ends in a slightly lower minimum on the HIP path.  The 5-point LO refines over the subset get_inliers(5 thr^2) of the incoming model
# (RelativePoseEstimator::refine_model); the cause was not isolated further — a membership test of that subset on a rounding boundary or the LM's
# summation order are the candidates (the tie class of DESIGN.md 8a (ii): LO results that reach the same basin from different starts).
SCORE_OFF_BASELINES = {"relpose_5pt_n2000_i10k": [519]}


def _pose_diff(a, b):
    dq = min(np.abs(a[:4] - b[:4]).max(), np.abs(a[:4] + b[:4]).max())
    return dq + np.abs(a[4:7] / np.linalg.norm(a[4:7]) - b[4:7] / np.linalg.norm(b[4:7])).max()


def _fund_diff(a, b):
    a, b = a[:9] / np.linalg.norm(a[:9]), b[:9] / np.linalg.norm(b[:9])
    return min(np.abs(a - b).max(), np.abs(a + b).max())


@pytest.mark.parametrize("workload", list(BASELINE_SETS))
def test_baseline_and_clean_bench_batches_vs_reference_fixture(capi, golden, workload):
    """Every pair of the batches bench.py times for the 5- / 7-point baselines (1024 pairs), the 6-point baseline (its 256-pair bench batch) and the
    outlier-free calibrated shape (1024 pairs) against the reference binary's own output (tests/golden/headline_ref_<workload>.npz,
    tests/tools/gen_golden_headline_ref_classic.py), through the device-resident entry point bench.py times: iterations, inlier count and inlier mask
    identical on every pair, model within 1e-6 (|t| of a 5- / 6-point pose and the scale of F are gauges: directions / normalised matrices are compared),
    model_score to 1e-9, `refinements` = the oracle's stored count (whose own difference to the reference is in the fixture: 12 / 0 / 7 / 1 pairs) plus
    the exact list above."""
    import torch
    from mdrp_amd import synth
    ref = golden(f"headline_ref_{workload}")
    kind, B, of, rf = BASELINE_SETS[workload]
    n = 2000
    host = synth.make_batch(0, B, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf)
    for i in range(0, B, 37):
        h_ = hashlib.sha256()
        for k in ("x1", "x2", "d1", "d2"):
            h_.update(np.ascontiguousarray(host[k][i], dtype=np.float64).tobytes())
        assert np.frombuffer(h_.digest()[:8], dtype=np.uint64)[0] == ref["digest"][i], "synthetic generator drifted"
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(host[k]).to(dev) for k in ("x1", "x2", "d1", "d2")]
    mask_t = torch.zeros((B, n), dtype=torch.uint8, device=dev)
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE)
    if kind != 4:
        cams["params"][:, 0] = 800.0  # (kind 4: cam1 carries the principal point, 0 here)
    ro = capi.ransac_opt_from_dict(RO)
    bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    c = cams if kind in (0, 3, 4) else None
    h = capi.Handle(0)
    torch.cuda.synchronize(dev)
    try:
        h.estimate_batch_device(kind, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr() if kind == 0 else 0, t[3].data_ptr() if kind == 0 else 0, B, n, ro, bo, None, c, c, mask_t.data_ptr())
        res = h.fetch_results(B)
        mask = mask_t.cpu().numpy()
    finally:
        h.close()
    ist, fst = ref["istats"], ref["fstats"]
    assert np.array_equal(res["iterations"].astype(np.int64), ist[:, 1]) and np.array_equal(res["num_inliers"].astype(np.int64), ist[:, 2]), workload
    assert np.array_equal(mask, np.unpackbits(ref["mask"], axis=1)[:, :n]), (workload, np.nonzero((mask != np.unpackbits(ref["mask"], axis=1)[:, :n]).any(axis=1))[0][:8])
    sc_off = np.nonzero(~np.isclose(res["model_score"], fst[:, 1], rtol=1e-9, atol=0))[0]
    assert sc_off.tolist() == SCORE_OFF_BASELINES.get(workload, []), (workload, {int(i): (float(res["model_score"][i]), float(fst[i, 1]), int(res["refinements"][i]), int(ist[i, 0]), int(ref["oracle_refinements"][i])) for i in sc_off})
    worst = 0.0
    for i in range(B):
        m = capi.model_to_array(res[i]["model"])
        r = ref["model"][i]
        if kind == 0:
            d = model_diff(m, r)
        elif kind == 5:
            d = _fund_diff(np.r_[m[:9]], r[:9])
        else:
            d = _pose_diff(m[:7], r[:7])
            if kind == 4:
                d = max(d, abs(m[10] - r[7]) / abs(r[7]))  # the shared focal length (mdrp_model.f1 | the fixture's 8th value)
        worst = max(worst, d)
        assert d < 1e-6, (workload, i, d)
    expected = ref["oracle_refinements"].copy()
    for k, v in GPU_MINUS_ORACLE_LO_BASELINES[workload].items():
        expected[k] += v
    got = res["refinements"].astype(np.int64)
    wrong = np.nonzero(got != expected)[0]
    assert len(wrong) == 0, (workload, {int(i): (int(got[i]), int(expected[i]), int(ist[i, 0])) for i in wrong})
    off = np.nonzero(got != ist[:, 0])[0]
    print(f"{workload} vs REFERENCE binary: {B} / {B} pairs identical (iterations, inliers, mask); worst model diff {worst:.2e}; LO count differs from the reference on {len(off)} pairs "
          f"{dict(zip(off.tolist(), (got[off] - ist[off, 0]).tolist()))}")
