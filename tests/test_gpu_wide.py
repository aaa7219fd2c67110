"""Parity of the batch sizes bench.py actually runs: large ragged batches of every monodepth
estimator against the sequential CPU oracle, and BASELINE.json's full-size shapes — including the 1024-pair headline
batch itself — against the reference binary's own outputs (tests/golden/estimate_wide.npz: seeds + outputs, written by
tests/tools/gen_golden_wide.py; inputs regenerate from mdrp_amd.synth and are checked against a stored digest).
Needs an MI355X:  pytest -m gpu."""
import hashlib

import numpy as np
import pytest

from helpers import model_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    from mdrp_amd import _capi
    return _capi.default_handle(0)


@pytest.fixture(scope="module")
def capi():
    from mdrp_amd import _capi
    return _capi


@pytest.fixture(scope="module")
def po():
    from oracle import pyorc
    return pyorc


def input_digest(p):
    h = hashlib.sha256()
    for k in ("x1", "x2", "d1", "d2"):
        h.update(np.ascontiguousarray(p[k], dtype=np.float64).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def wide_case(g, name):
    j = list(g["names"]).index(name)
    kind, es, n = (int(v) for v in g["cases"][j])
    return kind, es, n, float(g["outlier_frac"][j])


def wide_pair(kind, es, n, of, index):
    from mdrp_amd import synth
    rf = [None, "shared", "varying"][kind]
    return synth.make_pair(index, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf,
                           shift1=0.2 if es and kind == 0 else 0.0, shift2=-0.1 if es and kind == 0 else 0.0)


def check_against_reference(capi, g, name, n, res, mask, rows):
    """res / mask rows `rows` of a batch against the reference binary's outputs for the 32 pinned pairs: result identity
    (iterations, inlier count, mask, model <= 1e-6, score) on EVERY pair; the LO count equals the reference's or, on the few
    pairs where gen_golden_wide.py measured a solver-level deviation of the CPU port, the port's; returns
    the number of pairs whose LO count differs from the reference's"""
    ref_m, ref_st, ref_mask = g[f"{name}_model"], g[f"{name}_stats"], g[f"{name}_mask"]
    port_lo = g[f"{name}_oracle_refinements"]
    lo_dev = 0
    for j, r in enumerate(rows):
        where = (name, int(g["indices"][j]))
        assert int(res[r]["iterations"]) == int(ref_st[j][1]) == 10000, where
        assert int(res[r]["num_inliers"]) == int(ref_st[j][2]), (where, int(res[r]["num_inliers"]), ref_st[j][2])
        assert (mask[r][:n] == np.unpackbits(ref_mask[j])[:n]).all(), where
        assert model_diff(capi.model_to_array(res[r]["model"]), ref_m[j]) < 1e-6, (where, model_diff(capi.model_to_array(res[r]["model"]), ref_m[j]))
        assert res[r]["model_score"] == pytest.approx(ref_st[j][4], rel=1e-9), where
        assert res[r]["inlier_ratio"] == pytest.approx(ref_st[j][3], rel=1e-12), where
        assert int(res[r]["refinements"]) in (int(ref_st[j][0]), int(port_lo[j])), (where, int(res[r]["refinements"]), int(port_lo[j]), int(ref_st[j][0]))
        lo_dev += int(res[r]["refinements"]) != int(ref_st[j][0])
    return lo_dev


def test_headline_batch_vs_reference_binary(handle, capi, golden):
    """bench.py's own batch — 1024 pairs of calib_p3p_n2000_i10k, the kernels and grid sizes the headline number is measured
    on — with 32 pairs spread over it (indices 0, 33, ..., 1023) compared against the reference binary's output."""
    from mdrp_amd import synth
    g = golden("estimate_wide")
    kind, es, n, of = wide_case(g, "calib_p3p")
    B = 1024
    b = synth.make_batch(0, B, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of)
    rows = [int(i) for i in g["indices"]]
    for j, r in enumerate(rows):
        assert input_digest({k: b[k][r] for k in ("x1", "x2", "d1", "d2")}) == g["calib_p3p_digest"][j], "synthetic generator drifted"
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = {"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    res, mask = handle.estimate_batch(capi.CALIB, b["x1"], b["x2"], b["d1"], b["d2"], capi.ransac_opt_from_dict(ro),
                                      capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), None, cams, cams)
    lo_dev = check_against_reference(capi, g, "calib_p3p", n, res, mask, rows)
    assert lo_dev <= 1  # the reference's NaN-pose / missed-root cases: pair 957 in the port's run
    # every pair of the batch ran all iterations and found the planted geometry (50 % outliers of 2000)
    assert (res["iterations"] == 10000).all() and int(res["num_inliers"].min()) > 700


@pytest.mark.parametrize("name", ["calib_shift", "shared", "varying_shiftflag"])
def test_full_size_wide_vs_reference_binary(handle, capi, golden, name):
    """32 full-size pairs per estimator (N = 2000, varying focal N = 5000; 10^4 iterations, 50 % outliers) as one batch
    against the reference binary's outputs."""
    g = golden("estimate_wide")
    kind, es, n, of = wide_case(g, name)
    pairs = [wide_pair(kind, es, n, of, int(i)) for i in g["indices"]]
    for j, p in enumerate(pairs):
        assert input_digest(p) == g[f"{name}_digest"][j], "synthetic generator drifted"
    B = len(pairs)
    x1 = np.stack([p["x1"] for p in pairs]); x2 = np.stack([p["x2"] for p in pairs])
    d1 = np.stack([p["d1"] for p in pairs]); d2 = np.stack([p["d2"] for p in pairs])
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    ro = {"max_iterations": 10000, "min_iterations": 10000, "max_epipolar_error": 2.0, "max_reproj_error": 16.0,
          "monodepth_estimate_shift": bool(es)}
    res, mask = handle.estimate_batch(kind, x1, x2, d1, d2, capi.ransac_opt_from_dict(ro), capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}),
                                      None, cams if kind == 0 else None, cams if kind == 0 else None)
    lo_dev = check_against_reference(capi, g, name, n, res, mask, list(range(B)))
    assert lo_dev <= 1


@pytest.mark.parametrize("kind,es,rf", [(0, False, None), (0, True, None), (1, False, "shared"), (2, False, "varying")])
def test_large_ragged_batch_vs_oracle(handle, capi, po, golden, kind, es, rf):
    """B = 136 ragged pairs (N = 40 ... 300, 0-50 % outliers) of every monodepth estimator in ONE call — above every batch-size
    switch of the host schedule — against the sequential CPU oracle: every pair on the oracle's exact trajectory.  The LO
    count may equal the reference binary's instead (tests/golden/ragged_lo.npz) on the pairs where oracle and reference
    differ by one through a solver-level deviation: the HIP solvers side with one or the other; at most two
    pairs per estimator may differ by one LO from both through a rounding-level score tie (below)."""
    from mdrp_amd import synth
    ref = golden("ragged_lo")[f"k{kind}_s{int(es)}"]
    B = 136
    rng = np.random.default_rng(77 + kind + int(es))
    ns = rng.integers(40, 301, B)
    ns[:4] = [300, 40, 64, 65]
    N = int(ns.max())
    x1 = np.zeros((B, N, 2)); x2 = np.zeros((B, N, 2)); d1 = np.ones((B, N)); d2 = np.ones((B, N))
    for i, n in enumerate(ns):
        p = synth.make_pair(12000 + 50 * kind + i, int(n), noise_px=0.6, depth_noise=0.02, outlier_frac=[0.0, 0.25, 0.5][i % 3],
                            random_focal=rf, shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
        x1[i, :n] = p["x1"]; x2[i, :n] = p["x2"]; d1[i, :n] = p["d1"]; d2[i, :n] = p["d2"]
    opts = {"max_iterations": 700, "min_iterations": 700, "max_epipolar_error": 2.0, "max_reproj_error": 16.0}
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE); cams["params"][:, 0] = 800.0
    res, mask = handle.estimate_batch(kind, x1, x2, d1, d2, capi.ransac_opt_from_dict({**opts, "monodepth_estimate_shift": es}),
                                      capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"}), ns.astype(np.int32),
                                      cams if kind == 0 else None, cams if kind == 0 else None)
    oro = po.ransac_opt(estimate_shift=es, **opts)
    cam = po.cam_flat(0, [800.0, 0, 0])
    lo_dev = []
    for i, n in enumerate(ns):
        n = int(n)
        m, st, mk = po.estimate(kind, x1[i, :n], x2[i, :n], d1[i, :n], d2[i, :n], oro, po.bundle_opt(loss_type=4),
                                cam if kind == 0 else None, cam if kind == 0 else None)
        where = (kind, es, i, n)
        assert int(res[i]["iterations"]) == st.iterations == 700, where
        if int(res[i]["refinements"]) not in (st.refinements, int(ref[i][0])):
            lo_dev.append((i, n, int(res[i]["refinements"]), st.refinements, int(ref[i][0])))
        assert int(res[i]["num_inliers"]) == st.num_inliers == int(ref[i][1]) and (mask[i, :n] == mk).all() and mask[i, n:].sum() == 0, where
        assert model_diff(capi.model_to_array(res[i]["model"]), m) < 2e-6, (where, model_diff(capi.model_to_array(res[i]["model"]), m))
    # At N < 100 the 700 samples repeat triples in permuted order: the same model up to rounding, its score equal to the running
    # record to 1e-14 (pair 34 of the calibrated batch, iteration 264: tests/tools/diag_gpu_solver.py finds no solver difference);
    # whether `score < record` then holds is decided by the summation order.  One LO more or less, the same result.
    assert len(lo_dev) <= 2 and all(abs(d[2] - d[3]) == 1 for d in lo_dev), lo_dev


@pytest.mark.parametrize("kind,es", [(0, False), (0, True), (1, False)])
def test_fused_tail_is_bit_identical(capi, monkeypatch, kind, es):
    """The fused tail (the last LO launch replays each pair itself and k_final starts from the ready list while the LO drains,
    mdrp_capi.hip `fuse_tail`) against LO | k_walk | k_final one after the other: the same arithmetic in another order of
    launches - records and masks bit for bit, on a ragged batch large enough for one wavefront per LO problem (with pairs
    that have no trigger in the last chunk, pairs below the sample size, and a dynamic-stopping run, where it must stay off)."""
    from mdrp_amd import synth
    B, N = 160, 600
    rf = [None, "shared", "varying"][kind]
    pairs = [synth.make_pair(8800 + i, [N, 400, 3, 2, 150][i % 5] if i % 7 == 0 else N, noise_px=0.5, depth_noise=0.02,
                             outlier_frac=[0.5, 0.2, 0.0][i % 3], random_focal=rf) for i in range(B)]
    n_per = np.array([len(p["x1"]) for p in pairs], dtype=np.int32)
    x1, x2 = np.zeros((B, N, 2)), np.zeros((B, N, 2))
    d1, d2 = np.ones((B, N)), np.ones((B, N))
    for i, p in enumerate(pairs):
        x1[i, : n_per[i]], x2[i, : n_per[i]], d1[i, : n_per[i]], d2[i, : n_per[i]] = p["x1"], p["x2"], p["d1"], p["d2"]
    cams = np.zeros(B, dtype=capi.CAMERA_DTYPE)
    cams["params"][:, 0] = 800.0
    bo = capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    h = capi.Handle(0)
    try:
        for ropt in ({"max_iterations": 3000, "min_iterations": 3000}, {"max_iterations": 3000, "min_iterations": 100}):
            ro = capi.ransac_opt_from_dict({**ropt, "max_epipolar_error": 2.0, "max_reproj_error": 16.0, "monodepth_estimate_shift": es})
            out = []
            # "giveup": the bounded waits of the fused tail cut to 1 us - the gate opens at once, final workgroups whose pair is not ready
            # leave it to the pass behind the LO launch (what happens where kernels of two streams cannot overlap, e.g. under rocprofv3 --pmc)
            for fuse in ("0", "1", "giveup"):
                monkeypatch.setenv("MDRP_FUSE_TAIL", "0" if fuse == "0" else "1")
                for k in ("MDRP_FUSE_GATE_US", "MDRP_FUSE_WAIT_US"):
                    monkeypatch.setenv(k, "1") if fuse == "giveup" else monkeypatch.delenv(k, raising=False)
                res, mask = h.estimate_batch(kind, x1, x2, d1, d2, ro, bo, n_per, cams if kind == 0 else None, cams if kind == 0 else None)
                out.append((res.copy(), mask.copy()))
                st = h.last_stats()
                if ropt["min_iterations"] == ropt["max_iterations"]:  # (with dynamic stopping the end of the run is not known in advance: never fused)
                    # expired waits are counted and reported (mdrp_stats, ABI 0.3): none in a healthy run, some when the waits are cut to 1 us
                    assert (st["fuse_timeouts"] > 0) == (fuse == "giveup"), (fuse, st["fuse_gate_timeouts"], st["fuse_wait_timeouts"])
                else:
                    assert st["fuse_timeouts"] == 0
            (r0, m0), (r1, m1), (r2, m2) = out
            assert r0.tobytes() == r1.tobytes() and np.array_equal(m0, m1), ropt
            assert r0.tobytes() == r2.tobytes() and np.array_equal(m0, m2), ropt
            assert int(r0["refinements"].max()) > 3
    finally:
        h.close()
