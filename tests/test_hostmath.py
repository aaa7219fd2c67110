"""Per-lane arithmetic of the product (mdrp_amd/csrc/mdrp_math.h, host build) against the CPU oracle.
Checks the solvers, residual/Jacobian rows, the LM step and the sampler without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyorc as po
from helpers import match_solution_sets

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "hostmath", "libhostmath.so")
dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def hm():
    src = os.path.join(HERE, "hostmath", "hostmath.cpp")
    hdr = os.path.join(HERE, "..", "mdrp_amd", "csrc", "mdrp_math.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", src, "-o", SO])
    lib = C.CDLL(SO)
    lib.hm_loss.restype = C.c_double
    return lib


def _hm_solver(hm, solver, x1, x2, d1, d2):
    out = np.zeros((4, 12))
    n = hm.hm_solver(C.c_int(solver), P(po.f64(x1)), P(po.f64(x2)), P(po.f64(d1)), P(po.f64(d2)), P(out))
    return out[:n]


@pytest.mark.parametrize("kind", ["p3p", "calib_shift", "shared", "varying"])
def test_solvers_equal_oracle(hm, golden, kind):
    g = golden("solvers")
    src = "calib_shift" if kind == "p3p" else kind  # the P3P path consumes the same (x1,x2,d1,d2) samples
    solver = {"p3p": 0, "calib_shift": 1, "shared": 2, "varying": 3}[kind]
    ofn = {"p3p": po.solver_calib_p3p, "calib_shift": po.solver_calib_shift, "shared": po.solver_shared,
           "varying": po.solver_varying}[kind]
    n = len(g[f"{src}_n"])
    nsol = 0
    for i in range(n):
        a = (g[f"{src}_x1"][i], g[f"{src}_x2"][i], g[f"{src}_d1"][i], g[f"{src}_d2"][i])
        mine, ref = _hm_solver(hm, solver, *a), ofn(*a)
        assert match_solution_sets(list(ref), list(mine), 1e-9), (kind, i, ref, mine)
        nsol += len(mine)
    assert nsol > 0.3 * n


def test_sampler_equal_oracle(hm):
    for n, seed in ((7, 0), (200, 3), (2000, 0), (5000, 9)):
        out = np.zeros((500, 3), dtype=np.uint32)
        hm.hm_draw(C.c_uint64(n), C.c_uint64(seed), 500, out.ctypes.data_as(C.c_void_p))
        assert (out.astype(np.int64) == po.draw_samples(seed, n, 500)).all()


def test_wave_speculative_sampler_equals_sequential(hm):
    """k_samples' speculation scheme (host emulation) reproduces the sequential sampler incl. the end state, also for
    tiny N where nearly every wave step sees a rejection, and continues correctly across chunks"""
    for n, seed, count in ((3, 0, 300), (7, 1, 1000), (200, 0, 3000), (2000, 0, 10000), (5000, 4, 10000)):
        out = np.zeros((count, 3), dtype=np.uint32)
        st = C.c_uint64(0)
        hm.hm_draw_wave(C.c_uint64(n), C.c_uint64(seed), count, out.ctypes.data_as(C.c_void_p), C.byref(st))
        ref = po.draw_samples(seed, n, count + 50)
        assert (out.astype(np.int64) == ref[:count]).all(), n
        out2 = np.zeros((50, 3), dtype=np.uint32)
        hm.hm_draw_wave(C.c_uint64(n), st, 50, out2.ctypes.data_as(C.c_void_p), C.byref(st))
        assert (out2.astype(np.int64) == ref[count:]).all(), n


def test_residuals_and_jacobian_equal_oracle(hm):
    rng = np.random.default_rng(0)
    L = po.lib()
    for trial in range(40):
        focal = trial % 2
        m = po.new_model()
        q = rng.normal(size=4)
        m[:4] = q / np.linalg.norm(q)
        m[4:7] = rng.normal(0, 0.5, 3)
        m[7] = rng.uniform(0.3, 3)
        m[8], m[9] = rng.uniform(-0.3, 0.3, 2)
        if focal:
            m[10], m[11] = rng.uniform(0.5, 2, 2)
        x1, x2 = rng.uniform(-0.8, 0.8, 2), rng.uniform(-0.8, 0.8, 2)
        d1, d2 = rng.uniform(1, 6, 2)
        r, J = np.zeros(7), np.zeros(55)
        hm.hm_point(C.c_int(focal), P(m), C.c_double(0.37), P(x1), P(x2), C.c_double(d1), C.c_double(d2), P(r), P(J))
        ro, Jo = np.zeros(7), np.zeros(55)
        L.orc_debug_point(C.c_int(2 if focal else 0), P(m), C.c_double(0.37), P(x1), P(x2), C.c_double(d1), C.c_double(d2), P(ro), P(Jo))
        assert np.allclose(r, ro, rtol=1e-12, atol=1e-14)
        assert np.allclose(J, Jo, rtol=1e-10, atol=1e-12), np.abs(J - Jo).max()


def test_lm_step_and_losses_equal_oracle(hm):
    rng = np.random.default_rng(1)
    L = po.lib()
    for trial in range(20):
        m = po.new_model()
        q = rng.normal(size=4)
        m[:4] = q / np.linalg.norm(q)
        m[4:10] = rng.normal(0, 0.5, 6)
        m[10:] = rng.uniform(0.5, 2, 2)
        d = rng.normal(0, 0.05 if trial % 2 else 1e-8, 11)
        out, ref = np.zeros(12), np.zeros(12)
        hm.hm_step(1, 1, P(m), P(d), P(out))
        L.orc_debug_step(C.c_int(2), P(m), P(d), P(ref))
        assert np.allclose(out, ref, rtol=1e-13, atol=1e-15)
    # Cholesky
    A = rng.normal(size=(9, 9)); A = A @ A.T + 9 * np.eye(9); b = rng.normal(size=9); x = np.zeros(9)
    hm.hm_chol9(P(np.ascontiguousarray(A)), P(b), P(x))
    assert np.allclose(A @ x, b, rtol=1e-10)
    # losses against their closed forms
    thr = 0.7
    for r2 in (0.01, 0.3, 0.49, 0.5, 2.0):
        t2 = thr * thr
        assert hm.hm_loss(1, C.c_double(thr), C.c_double(r2), 0) == pytest.approx(min(r2, t2))
        assert hm.hm_loss(3, C.c_double(thr), C.c_double(r2), 0) == pytest.approx(t2 * np.log1p(r2 / t2))
        assert hm.hm_loss(4, C.c_double(thr), C.c_double(r2), 0) == pytest.approx(t2 * np.log1p(min(r2, t2) / t2))
        assert hm.hm_loss(4, C.c_double(thr), C.c_double(r2), 1) == pytest.approx(1 / (1 + r2 / t2) if r2 < t2 else 0.0)


def _rand_E(rng, kind):
    from mdrp_amd import synth
    R = synth.rodrigues(rng.normal(0, 0.6, 3))
    t = rng.normal(0, 1.0, 3) * 10.0 ** rng.uniform(-4, 1)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    if kind:  # fundamental matrix of a focal estimator: diag(1,1,f2) E diag(1,1,f1), normalised focals around 1
        f1, f2 = 10.0 ** rng.uniform(-1, 1, 2)
        E = np.diag([1, 1, f2]) @ E @ np.diag([1, 1, f1])
    return np.ascontiguousarray(E.reshape(9))


def test_fp32_prefilter_is_conservative(hm):
    """mdrp_math.h filter_setup / filter_keeps (phase 1 of the scoring sweep, run in fp32 on the GPU) must keep every record
    the exact fp64 test `C^2 < thr * den` accepts: random and near-degenerate hypotheses, coordinate scales 1e-3 .. 1e3,
    inliers planted exactly on the epipolar lines, thresholds from 1e-8 to 1e-2."""
    hm.hm_filter_check.restype = C.c_long
    rng = np.random.default_rng(11)
    n = 4000
    total_kept = total_exact = 0
    for trial in range(300):
        kind = trial % 2
        E = _rand_E(rng, kind)
        scale = 10.0 ** rng.uniform(-3, 3) if trial % 3 == 0 else 1.0
        x1 = rng.uniform(-1, 1, (n, 2)) * np.array([0.8, 0.6]) * scale
        x2 = rng.uniform(-1, 1, (n, 2)) * np.array([0.8, 0.6]) * scale
        # plant near-inliers: move x2 onto (or within a few thresholds of) its epipolar line l = E (x1, 1)
        thr = 10.0 ** rng.uniform(-8, -2) * scale * scale
        Em = E.reshape(3, 3)
        l = (Em @ np.c_[x1, np.ones(n)].T).T
        nl = np.hypot(l[:, 0], l[:, 1]) + 1e-300
        dist = (l[:, 0] * x2[:, 0] + l[:, 1] * x2[:, 1] + l[:, 2]) / nl
        k = n // 2
        off = rng.uniform(-3, 3, k) * np.sqrt(thr)
        x2[:k, 0] -= (dist[:k] - off) * l[:k, 0] / nl[:k]
        x2[:k, 1] -= (dist[:k] - off) * l[:k, 1] / nl[:k]
        if trial % 7 == 0:  # box corners
            x1[-8:] = np.abs(x1).max(0) * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1]] * 2)
        kept, exact = C.c_long(), C.c_long()
        missed = hm.hm_filter_check(P(E), P(np.ascontiguousarray(x1)), P(np.ascontiguousarray(x2)), C.c_long(n), C.c_double(thr),
                                    C.byref(kept), C.byref(exact))
        assert missed == 0, (trial, kind, scale, thr)
        assert kept.value >= exact.value
        total_kept += kept.value; total_exact += exact.value
    assert total_exact > 10000  # the planted inliers are really inliers
    # degenerate inputs keep everything: E = 0, NaN, fp32-overflowing magnitudes
    x1 = rng.uniform(-1, 1, (64, 2)); x2 = rng.uniform(-1, 1, (64, 2))
    for E in (np.zeros(9), np.full(9, np.nan), _rand_E(rng, 0) * 1e35):
        kept, exact = C.c_long(), C.c_long()
        assert hm.hm_filter_check(P(np.ascontiguousarray(E)), P(x1), P(x2), C.c_long(64), C.c_double(1e-5), C.byref(kept), C.byref(exact)) == 0
        assert kept.value == 64


def _planted(rng, E, n, thr, scale):
    """correspondences with half of them moved onto (or within a few thresholds of) their epipolar lines"""
    x1 = rng.uniform(-1, 1, (n, 2)) * np.array([0.8, 0.6]) * scale
    x2 = rng.uniform(-1, 1, (n, 2)) * np.array([0.8, 0.6]) * scale
    Em = E.reshape(3, 3)
    l = (Em @ np.c_[x1, np.ones(n)].T).T
    nl = np.hypot(l[:, 0], l[:, 1]) + 1e-300
    dist = (l[:, 0] * x2[:, 0] + l[:, 1] * x2[:, 1] + l[:, 2]) / nl
    k = n // 2
    off = rng.uniform(-3, 3, k) * np.sqrt(thr)
    x2[:k, 0] -= (dist[:k] - off) * l[:k, 0] / nl[:k]
    x2[:k, 1] -= (dist[:k] - off) * l[:k, 1] / nl[:k]
    return np.ascontiguousarray(x1), np.ascontiguousarray(x2)


def test_mfma_count_filter_is_conservative(hm):
    """mdrp_math.h count_setup_scaled + the arithmetic of k_count (bf16-split contraction, fp32 accumulation in three
    different orders, clamp test), emulated on the host: every correspondence the exact fp64 test accepts stays a
    candidate; the scale leaves no value between 0 and 1; and the filter stays selective (candidates < 3 x the exact
    inliers + 15 % of the outliers on these planted sets)."""
    hm.hm_count_check.restype = C.c_long
    rng = np.random.default_rng(21)
    n = 3000
    tot_kept = tot_exact = 0
    for trial in range(240):
        kind = trial % 2
        E = _rand_E(rng, kind)
        scale = 10.0 ** rng.uniform(-3, 3) if trial % 3 == 0 else 1.0
        thr = 10.0 ** rng.uniform(-8, -2) * scale * scale
        x1, x2 = _planted(rng, E, n, thr, scale)
        if trial % 7 == 0:
            x1[-8:] = np.abs(x1).max(0) * np.array([[1, 1], [1, -1], [-1, 1], [-1, -1]] * 2)
        for order in (0, 1, 2):
            kept, exact = C.c_long(), C.c_long()
            missed = hm.hm_count_check(P(E), P(x1), P(x2), C.c_long(n), C.c_double(thr), C.c_int(order), C.byref(kept), C.byref(exact))
            assert missed == 0, (trial, kind, scale, thr, order)
            assert kept.value >= exact.value
        tot_kept += kept.value; tot_exact += exact.value
    assert tot_exact > 10000
    x1 = rng.uniform(-1, 1, (64, 2)); x2 = rng.uniform(-1, 1, (64, 2))
    for E in (np.zeros(9), np.full(9, np.nan), _rand_E(rng, 0) * 1e35, _rand_E(rng, 0) * 1e-35):
        kept, exact = C.c_long(), C.c_long()
        assert hm.hm_count_check(P(np.ascontiguousarray(E)), P(x1), P(x2), C.c_long(64), C.c_double(1e-5), C.c_int(0), C.byref(kept), C.byref(exact)) == 0
        assert kept.value == 64


def test_fp32_score_lower_bound_is_a_lower_bound(hm):
    """mdrp_math.h bound_setup32 / bound_r2_32 (k_bound): the fp32 sum of min(q, thr) times (1 - BOUND_SLACK) never
    exceeds the exact MSAC score (cheirality aside), its inlier count never undercounts, and it is tight (within 5 % of the
    exact score on these sets) — otherwise it would retire nothing."""
    rng = np.random.default_rng(22)
    n = 3000
    ratios = []
    for trial in range(240):
        kind = trial % 2
        E = _rand_E(rng, kind)
        scale = 10.0 ** rng.uniform(-2, 2) if trial % 3 == 0 else 1.0
        thr = 10.0 ** rng.uniform(-7, -3) * scale * scale
        x1, x2 = _planted(rng, E, n, thr, scale)
        lb, ex = C.c_double(), C.c_double()
        cu, ce = C.c_long(), C.c_long()
        ok = hm.hm_bound_check(P(E), P(x1), P(x2), C.c_long(n), C.c_double(thr), C.byref(lb), C.byref(ex), C.byref(cu), C.byref(ce))
        assert ok == 1, (trial, kind, scale, thr, lb.value, ex.value, cu.value, ce.value)
        ratios.append(lb.value / ex.value)
    assert min(ratios) > 0.5 and np.median(ratios) > 0.95, (min(ratios), np.median(ratios))
    for E in (np.full(9, np.nan), _rand_E(rng, 0) * 1e35):
        lb, ex = C.c_double(), C.c_double()
        cu, ce = C.c_long(), C.c_long()
        assert hm.hm_bound_check(P(np.ascontiguousarray(E)), P(x1), P(x2), C.c_long(n), C.c_double(1e-5), C.byref(lb), C.byref(ex), C.byref(cu), C.byref(ce)) == 1


def test_filters_are_conservative_for_model_matrices_of_any_magnitude(hm):
    """|E| from 1e-36 to 1e36 (the GPU suite found k_count undercounting at |F| = 1e-24: thr * Dmax left the
    fp32 normal range and the threshold collapsed to the kappa term).  Both filters, emulated on the host: no correspondence the
    exact test accepts is ever dropped, no score bound exceeds the exact score, at every magnitude — a model the fp32 arithmetic
    cannot judge must keep everything."""
    hm.hm_count_check.restype = C.c_long
    rng = np.random.default_rng(23)
    n = 1500
    judged = 0
    for e in range(-36, 37, 2):
        for kind in (0, 1):
            E0 = _rand_E(rng, kind)
            thr = 10.0 ** rng.uniform(-7, -4)
            x1, x2 = _planted(rng, E0, n, thr, 1.0)
            E = np.ascontiguousarray(E0 * 10.0 ** e)
            for order in (0, 2):
                kept, exact = C.c_long(), C.c_long()
                assert hm.hm_count_check(P(E), P(x1), P(x2), C.c_long(n), C.c_double(thr), C.c_int(order), C.byref(kept), C.byref(exact)) == 0, (e, kind, order)
                assert kept.value >= exact.value and exact.value > n // 10, (e, kind, kept.value, exact.value)
            judged += kept.value < n
            lb, ex = C.c_double(), C.c_double()
            cu, ce = C.c_long(), C.c_long()
            assert hm.hm_bound_check(P(E), P(x1), P(x2), C.c_long(n), C.c_double(thr), C.byref(lb), C.byref(ex), C.byref(cu), C.byref(ce)) == 1, (e, kind, lb.value, ex.value)
    assert judged >= 20  # ... and across the ordinary magnitudes the count filter still retires something


def test_first_chunk_wish_of_a_pair(hm):
    """mdrp_math.h first_chunk_wish: what a pair of inlier ratio r (sample size k) would like as the first chunk of its run — 6 / r^k iterations between
    256 and 1024, 128 when nearly every sample is outlier-free; the host averages it over the pairs of a call to size the next one's (mdrp_capi.hip)."""
    import ctypes as C
    hm.hm_first_chunk_wish.restype = C.c_int
    hm.hm_first_chunk_wish.argtypes = [C.c_double, C.c_int]
    w = lambda r, k=3: hm.hm_first_chunk_wish(r, k)
    assert [w(1.0), w(0.8), w(0.72), w(0.7)] == [128, 128, 256, 256]     # 6 / r^3 <= 16 up to r = 0.721
    assert [w(0.5), w(0.3), w(0.25), w(0.2), w(0.18), w(0.15), w(0.0)] == [256, 256, 384, 749, 1024, 1024, 1024]  # (0.2^3 rounds up: 6 / 0.008000000000000002)
    assert w(float("nan")) == 1024
    assert [w(0.5, 5), w(0.5, 7), w(0.9, 7)] == [256, 768, 128]
    assert all(w(a) >= w(b) for a, b in zip(np.linspace(0.01, 0.71, 80), np.linspace(0.02, 0.72, 80)))  # monotone below the cut-off

