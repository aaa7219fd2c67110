"""Non-monodepth baselines: the CPU oracle and the host build of the device arithmetic against
tests/golden/classic.npz — outputs of the reference binary's relpose_5pt / relpose_7pt / refine_relpose / refine_fundamental /
estimate_relative_pose / estimate_fundamental (tests/tools/gen_golden.py::gen_classic)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyorc as po

HERE = os.path.dirname(os.path.abspath(__file__))
dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


def pose_diff(a, b):
    """rotation (quaternion up to sign) and translation DIRECTION: |t| is a gauge of the 5-parameter LM (t moves in its tangent
    plane and is never renormalised), and ties between equally good local minima pick LO results that differ only in |t|"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    dq = min(np.abs(a[:4] - b[:4]).max(), np.abs(a[:4] + b[:4]).max())
    return dq + np.abs(a[4:7] / np.linalg.norm(a[4:7]) - b[4:7] / np.linalg.norm(b[4:7])).max()


def fund_diff(a, b):
    a, b = np.asarray(a, float).reshape(-1)[:9], np.asarray(b, float).reshape(-1)[:9]
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max())


def classic_cam(f):
    return po.cam_flat(0, [f, 640.0, 480.0]), po.cam_flat(1, [f * 1.01, f * 0.99, 640.0, 480.0])


def est_cases(g):
    for row in g["est_cases"]:
        i, kind, n, its, min_its, loss = (int(v) for v in row[:6])
        yield i, kind, n, its, min_its, loss, float(row[6]), int(row[7]), float(row[8])


def test_samples_of_5_and_7(golden):
    g = golden("classic")
    l = po.lib()
    for K in (5, 7):
        for n in (9, 200, 2000):
            ref = g[f"samples_k{K}_n{n}"]
            # the oracle's K-point sampler is static inside orc_classic.c: replay it through random_int
            st = C.c_uint64(3)
            out = np.zeros_like(ref)
            for s in range(len(ref)):
                for i in range(K):
                    while True:
                        r = l.orc_random_int(C.byref(st))
                        v = (r + (1 << 64)) % (1 << 64) % n  # sign-extended int32, unsigned modulo
                        if v not in out[s, :i]:
                            out[s, i] = v
                            break
            assert np.array_equal(out, ref), (K, n)


def sample_residual(kind, sol, a, b):
    """largest |x2' E x1| of a solution over the sample's bearings"""
    M = po.essential(np.r_[sol[:7], 1, 0, 0, 1, 1]) if kind == 3 else np.asarray(sol[:9]).reshape(3, 3)
    return np.abs(np.einsum("ki,ij,kj->k", b, M / np.linalg.norm(M), a)).max()


def check_solver_lists(kind, g, solve):
    """same number of solutions in the same ORDER as the binary.  Values: equal to 1e-8, except where the binary's own root
    polishing stopped early (its residual on the sample reaches 4e-4 on ill-conditioned 5-point samples, ours stays below
    1e-8): there ours must be the more accurate one and within 1e-2 of it."""
    loose = 0
    for i in range(len(g[f"solver{kind}_n"])):
        a, b = np.ascontiguousarray(g[f"solver{kind}_x1"][i]), np.ascontiguousarray(g[f"solver{kind}_x2"][i])
        n_ref = int(g[f"solver{kind}_n"][i])
        ref = g[f"solver{kind}_sols"][i][:n_ref]
        ours = solve(a, b)
        assert len(ours) == n_ref, (kind, i, len(ours), n_ref)
        for k in range(n_ref):
            d = pose_diff(ours[k], ref[k]) if kind == 3 else fund_diff(ours[k], ref[k])
            if d > 1e-8:
                loose += 1
                assert kind == 3 and d < 1e-2, (kind, i, k, d)
                assert sample_residual(kind, ours[k], a, b) <= 10 * sample_residual(kind, ref[k], a, b) + 1e-9, (i, k)
    assert loose <= 12, loose


@pytest.mark.parametrize("kind", [3, 5])
def test_solver_lists_equal_reference_in_order(golden, kind):
    check_solver_lists(kind, golden("classic"), (lambda a, b: po.relpose_5pt(a, b)) if kind == 3 else (lambda a, b: po.relpose_7pt(a, b).reshape(-1, 9)))


def test_refine_equals_reference(golden):
    g = golden("classic")
    for i, kind, loss, its in g["refine_cases"]:
        bo = po.bundle_opt(max_iterations=int(its), loss_type=int(loss), loss_scale=0.004)
        m, st = po.refine_classic(int(kind), g[f"refine_x1_{i}"], g[f"refine_x2_{i}"], g[f"refine_m0_{i}"], bo)
        ref, rst = g[f"refine_m_{i}"], g[f"refine_stats_{i}"]
        d = pose_diff(m, ref) if kind == 3 else fund_diff(m, ref)
        assert d < 1e-9, (i, kind, loss, its, d)
        assert st.initial_cost == pytest.approx(rst[1], rel=1e-12) and st.cost == pytest.approx(rst[2], rel=1e-9)
        assert st.iterations == int(rst[0])


def test_estimators_equal_reference(golden):
    g = golden("classic")
    for i, kind, n, its, min_its, loss, thr, seed, f in est_cases(g):
        if n >= 2000:
            continue  # the BASELINE-sized cases run in test_full_size_estimators_equal_reference
        ro = po.ransac_opt(max_iterations=its, min_iterations=min_its, max_epipolar_error=thr, seed=seed)
        bo = po.bundle_opt(loss_type=loss, loss_scale=thr)
        c1, c2 = classic_cam(f)
        m, st, mask = po.estimate_classic(kind, g[f"est_x1_{i}"], g[f"est_x2_{i}"], ro, bo, c1, c2)
        ref, rst = g[f"est_model_{i}"], g[f"est_stats_{i}"]
        assert (st.refinements, st.iterations, st.num_inliers) == tuple(int(v) for v in rst[:3]), (i, kind, rst)
        assert st.model_score == pytest.approx(rst[4], rel=1e-9)
        assert np.array_equal(mask, g[f"est_mask_{i}"])
        d = pose_diff(m, ref) if kind == 3 else fund_diff(m, ref)
        assert d < 1e-7, (i, kind, d)


def test_full_size_estimators_equal_reference(golden):
    g = golden("classic")
    for i, kind, n, its, min_its, loss, thr, seed, f in est_cases(g):
        if n < 2000:
            continue
        ro = po.ransac_opt(max_iterations=its, min_iterations=min_its, max_epipolar_error=thr, seed=seed)
        bo = po.bundle_opt(loss_type=loss, loss_scale=thr)
        c1, c2 = classic_cam(f)
        m, st, mask = po.estimate_classic(kind, g[f"est_x1_{i}"], g[f"est_x2_{i}"], ro, bo, c1, c2)
        rst = g[f"est_stats_{i}"]
        assert (st.iterations, st.num_inliers) == (int(rst[1]), int(rst[2])), (i, kind, rst)
        assert int(st.refinements) == int(rst[0]), (i, st.refinements, rst[0])
        assert st.model_score == pytest.approx(rst[4], rel=1e-9)
        assert np.array_equal(mask, g[f"est_mask_{i}"])
        d = pose_diff(m, g[f"est_model_{i}"]) if kind == 3 else fund_diff(m, g[f"est_model_{i}"])
        assert d < 1e-7, (i, kind, d)


def test_initial_pose_sets_score_initial_model(golden):
    g = golden("classic")
    for i, n, its, min_its, seed in g["init_cases"]:
        ro = po.ransac_opt(max_iterations=int(its), min_iterations=int(min_its), max_epipolar_error=2.0, seed=int(seed), score_initial_model=True)
        c = po.cam_flat(0, [800.0, 640.0, 480.0])
        m, st, mask = po.estimate_classic(3, g[f"init_x1_{i}"], g[f"init_x2_{i}"], ro, po.bundle_opt(loss_type=4, loss_scale=2.0), c, c)
        rst = g[f"init_stats_{i}"]
        assert (st.refinements, st.iterations, st.num_inliers) == tuple(int(v) for v in rst[:3]), (i, rst)
        assert np.array_equal(mask, g[f"init_mask_{i}"]) and pose_diff(m, g[f"init_model_{i}"]) < 1e-7


# ---- the device arithmetic (mdrp_classic_math.h) compiled for the host
@pytest.fixture(scope="module")
def hm():
    src = os.path.join(HERE, "hostmath", "hostmath.cpp")
    so = os.path.join(HERE, "hostmath", "libhostmath.so")
    hdrs = [os.path.join(HERE, "..", "mdrp_amd", "csrc", h) for h in ("mdrp_math.h", "mdrp_classic_math.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", src, "-o", so])
    return C.CDLL(so)


@pytest.mark.parametrize("kind", [3, 5])
def test_device_solvers_equal_reference_in_order(hm, golden, kind):
    def solve(a, b):
        out = np.zeros((10, 12))
        n = (hm.hm_relpose_5pt if kind == 3 else hm.hm_relpose_7pt)(P(a), P(b), P(out))
        return out[:n]

    check_solver_lists(kind, golden("classic"), solve)


def test_device_solvers_equal_oracle_on_random_samples(hm):
    rng = np.random.default_rng(11)

    def unit(x):
        h = np.c_[x, np.ones(len(x))]
        return np.ascontiguousarray(h / np.linalg.norm(h, axis=1, keepdims=True))

    for trial in range(600):
        K = 5 if trial % 2 == 0 else 7
        x1 = rng.uniform(-1, 1, (K, 2))
        a, b = unit(x1), unit(x1 + rng.normal(size=(K, 2)) * (0.1 if trial % 4 < 2 else 1.0))
        out = np.zeros((10, 12))
        if K == 5:
            ref = po.relpose_5pt(a, b)
            n = hm.hm_relpose_5pt(P(a), P(b), P(out))
            assert n == len(ref), trial
            assert all(pose_diff(out[k], ref[k]) < 1e-4 for k in range(n)), trial
        else:
            ref = po.relpose_7pt(a, b).reshape(-1, 9)
            n = hm.hm_relpose_7pt(P(a), P(b), P(out))
            assert n == len(ref), trial
            assert all(fund_diff(out[k], ref[k]) < 1e-10 for k in range(n)), trial


def test_root_finder_with_shared_interval_storage_equals_separate_arrays(hm):
    """real_roots_fast<10> with the isolated intervals stored in the free end of the interval stack's arrays (the device's layout: what bounds the
    root kernel's LDS) against the same routine with separate arrays: the same roots bit for bit — also with all ten roots real (stack and isolated
    intervals then meet: 10 of the 12 shared entries), with clustered and with repeated roots."""
    rng = np.random.default_rng(17)
    seen10 = 0
    for trial in range(1500):
        mode = trial % 5
        if mode == 0:
            c = rng.normal(size=11)
        elif mode == 1:                                   # ten real roots
            c = np.poly(rng.uniform(-3, 3, 10))[::-1] * rng.uniform(0.1, 10)
        elif mode == 2:                                   # clusters
            c = np.poly(np.r_[rng.uniform(-2, 2, 4), 0.7 + 1e-6 * rng.normal(size=3), -1.1 + 1e-9 * rng.normal(size=3)])[::-1]
        elif mode == 3:                                   # repeated roots and a complex pair
            r = rng.uniform(-2, 2, 3)
            c = np.real(np.poly(np.r_[r, r, r[:2], 0.3 + 0.9j, 0.3 - 0.9j]))[::-1]
        else:                                             # a few real roots among complex pairs
            z = rng.normal(size=4) + 1j * rng.uniform(0.2, 2, 4)
            c = np.real(np.poly(np.r_[z, z.conj(), rng.uniform(-4, 4, 2)]))[::-1]
        c = np.ascontiguousarray(c, dtype=np.float64)
        ra, rb = np.zeros(10), np.zeros(10)
        na, nb = hm.hm_real_roots10_fast(P(c), P(ra)), hm.hm_real_roots10_fast_shared(P(c), P(rb))
        assert na == nb and ra[:na].tobytes() == rb[:nb].tobytes(), (trial, na, nb, ra, rb)
        assert np.all(np.diff(ra[:na]) >= 0), (trial, ra[:na])
        seen10 += na == 10
        if mode == 1:
            want = np.sort(np.roots(c[::-1]).real)
            assert na == 10 and np.allclose(ra, want, rtol=1e-6, atol=1e-6), (trial, ra, want)
    assert seen10 >= 250


def test_null_space_with_register_columns_equals_the_stored_factorisation(hm):
    """epipolar_nullspace<M, RC> (the device's: trailing constraint columns in registers, swaps as select chains) against epipolar_columns<M> +
    fullpiv_nullspace<M> (everything in the strided matrix): the same operations on the same values, so on the host — no fused multiply-adds — the
    two null spaces are equal bit for bit; generic samples, samples with repeated points (rank-deficient: the `degenerate` exit) and samples whose
    largest entries sit in the trailing columns (the pivot search then picks a register column in the first step)."""
    rng = np.random.default_rng(5)

    def unit(x):
        h = np.c_[x, np.ones(len(x))]
        return np.ascontiguousarray(h / np.linalg.norm(h, axis=1, keepdims=True))

    for trial in range(900):
        M = 5 if trial % 2 == 0 else 7
        x1 = rng.uniform(-1, 1, (M, 2))
        x2 = x1 + rng.normal(size=(M, 2)) * (0.1 if trial % 4 < 2 else 1.0)
        if trial % 9 == 4:
            x1[1], x2[1] = x1[0], x2[0]                      # two identical constraints
        if trial % 9 == 7:
            x1[-1] *= 30.0; x2[-2] *= 30.0                    # the largest entries in the last columns
        a, b = unit(x1), unit(x2)
        if trial % 9 == 7:
            a[-1] *= 50.0                                     # (not unit bearings: the factorisation does not care)
        n0, n1 = np.zeros((9 - M, 9)), np.zeros((9 - M, 9))
        assert hm.hm_nullspace(M, 0, P(a), P(b), P(n0)) == 0 and hm.hm_nullspace(M, 1, P(a), P(b), P(n1)) == 0
        assert n0.tobytes() == n1.tobytes(), (trial, M, np.abs(n0 - n1).max())


# ---------------------------------------------------------------------------------------------- 6-point, shared focal
def _sixpt_residual(sol, a, b):
    """how well a solution (q, t, f) reproduces the six epipolar constraints: max |x2' F x1| / |F| with F = K^-1' [t]x R K^-1
    (any F in the null space of the sample satisfies them exactly; a pose that was decomposed from an F far from an essential
    matrix does not)"""
    R = po.quat_to_rotmat(sol[:4]); t = sol[4:7]; f = sol[7]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.diag([1 / f, 1 / f, 1.0])
    F = Ki @ tx @ R @ Ki
    return max(abs(b[i] @ F @ a[i]) for i in range(6)) / np.linalg.norm(F)


def _same_sixpt(u, v, tol):
    return (abs(u[7] - v[7]) < tol * abs(v[7]) and min(np.abs(u[:4] - v[:4]).max(), np.abs(u[:4] + v[:4]).max()) < 10 * tol
            and np.abs(u[4:7] - v[4:7]).max() < 10 * tol)


def test_sixpt_solution_sets_equal_reference(golden):
    """relpose_6pt_shared_focal: every ACCURATE solution of the reference binary (epipolar residual of its own pose below 1e-9)
    is in the oracle's set to 1e-6, and the oracle returns nothing else — except where the binary's action-matrix solver lost
    accuracy on an ill-conditioned sample (its residuals reach 1e-3 there; ours stay below 1e-9 by construction: each of our
    extra / shifted solutions must then stand against an inaccurate one of the binary).  Order is not compared."""
    g = golden("sixpt")
    x1, x2, sols, cnt = g["solver_x1"], g["solver_x2"], g["solver_sols"], g["solver_n"]
    exact = inexact = total_ref = 0
    for i in range(len(cnt)):
        ref = [s for s in sols[i][:cnt[i]]]
        mine = [np.r_[m[:7], m[10]] for m in po.relpose_6pt(x1[i], x2[i])]
        for m in mine:
            assert _sixpt_residual(m, x1[i], x2[i]) < 1e-8, i
        acc = [r for r in ref if _sixpt_residual(r, x1[i], x2[i]) < 1e-9]
        bad = len(ref) - len(acc)
        total_ref += len(ref)
        used = set()
        for r in acc:
            j = next((j for j, m in enumerate(mine) if j not in used and _same_sixpt(m, r, 1e-6)), None)
            assert j is not None, (i, r[7], [m[7] for m in mine])
            used.add(j)
        assert len(mine) - len(used) <= bad, (i, len(mine), len(acc), bad)
        exact += bad == 0
        inexact += bad > 0
    assert exact >= 80 and total_ref > 100, (exact, inexact, total_ref)


def test_sixpt_estimator_vs_reference(golden):
    """estimate_shared_focal_relative_pose of the reference binary, 24 small runs (every loss type, fixed and dynamic stopping,
    with and without a principal point) + 8 full-size ones (N = 2000, 10^4 iterations, 50 % outliers): the oracle — whose
    solver returns the solutions by ascending focal length, not in the binary's eigenvalue order — lands on the same
    iterations, inlier count, mask and model; the LO count may differ by one where the order of two record breakers of one
    sample decides which of them is refined."""
    from mdrp_amd import synth
    g = golden("sixpt")
    lo_dev = 0
    for case in g["est_cases"]:
        k, n, its, min_its, loss, thr, seed = int(case[0]), int(case[1]), int(case[2]), int(case[3]), int(case[4]), float(case[5]), int(case[6])
        pp = (float(case[7]), float(case[8]))
        ro = po.ransac_opt(max_iterations=its, min_iterations=min_its, max_epipolar_error=thr, seed=seed)
        m, st, mask = po.estimate_classic(4, g[f"est_x1_{k}"], g[f"est_x2_{k}"], ro, po.bundle_opt(loss_type=loss, loss_scale=thr), pp=pp)
        ref_m, ref_st, ref_mask = g[f"est_model_{k}"], g[f"est_stats_{k}"], g[f"est_mask_{k}"]
        assert st.iterations == int(ref_st[1]) and st.num_inliers == int(ref_st[2]) and (mask == ref_mask).all(), (k, st.iterations, ref_st)
        assert np.abs(po.quat_to_rotmat(m[:4]) - po.quat_to_rotmat(ref_m[:4])).max() < 1e-6, k
        tn, tr = m[4:7] / np.linalg.norm(m[4:7]), ref_m[4:7] / np.linalg.norm(ref_m[4:7])
        assert np.abs(tn - tr).max() < 1e-6 and m[10] == pytest.approx(ref_m[7], rel=1e-6), k
        lo_dev += st.refinements != int(ref_st[0])
    for j, index in enumerate(g["full_indices"][:3]):  # three of the eight here (0.5 s each); all eight in the GPU suite
        pr = synth.make_pair(int(index), 2000, noise_px=0.5, outlier_frac=0.5, random_focal="shared", pp=(0.0, 0.0))
        ro = po.ransac_opt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0)
        m, st, mask = po.estimate_classic(4, pr["x1"], pr["x2"], ro, po.bundle_opt(loss_type=4), pp=(0.0, 0.0))
        ref_m, ref_st = g["full_model"][j], g["full_stats"][j]
        assert st.iterations == 10000 and st.num_inliers == int(ref_st[2]) and (mask == np.unpackbits(g["full_mask"][j])[:2000]).all(), index
        assert m[10] == pytest.approx(ref_m[7], rel=1e-6), index
        lo_dev += st.refinements != int(ref_st[0])
    assert lo_dev <= 3, lo_dev


# ---------------------------------------------------------------------------------------------- wide full-size pin (classic_wide.npz)
WIDE_CLASSIC = (("relpose_5pt", 3), ("shared_6pt", 4), ("fundamental_7pt", 5))


def wide_classic_pair(kind, index):
    """the inputs of tests/tools/gen_golden_wide_classic.py (seeds only are stored)"""
    from mdrp_amd import synth
    if kind == 4:
        return synth.make_pair(7000 + index, 2000, noise_px=0.5, outlier_frac=0.5, random_focal="shared", pp=(0.0, 0.0))
    return synth.make_pair(7000 + index, 2000, f1=800.0, f2=800.0, pp=(0.0, 0.0), noise_px=0.5, outlier_frac=0.5)


def wide_classic_digest(p):
    import hashlib
    h = hashlib.sha256()
    for k in ("x1", "x2"):
        h.update(np.ascontiguousarray(p[k], dtype=np.float64).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def wide_classic_model_diff(kind, m, ref):
    """m: flat model of ours (pose: q, t [, f at `fi`]; fundamental: nine entries), ref: the fixture's row (q, t, f, f | F)"""
    if kind == 5:
        return fund_diff(m, ref)
    return pose_diff(m, ref)


def test_wide_full_size_classic_pin_subsample(golden):
    """tests/golden/classic_wide.npz: 80 full-size runs of the reference binary (N = 2000, 10^4 iterations, 50 % outliers; 32 + 16 +
    32 pairs of the 5- / 6- / 7-point estimators) on which the oracle's RESULT equals the reference's on every pair (written by
    the generator into `oracle_same`) and the LO count on all but one 6-point pair; three pairs per estimator are re-run here."""
    g = golden("classic_wide")
    cam = po.cam_flat(0, [800.0, 0.0, 0.0])
    for name, kind in WIDE_CLASSIC:
        assert g[f"{name}_oracle_same"].all(), name
        dev = g[f"{name}_oracle_refinements"] - g[f"{name}_stats"][:, 0].astype(int)
        assert int(np.abs(dev).sum()) == (1 if kind == 4 else 0), (name, dev)
        for j in (0, 5, 12):
            p = wide_classic_pair(kind, j)
            assert wide_classic_digest(p) == g[f"{name}_digest"][j], "mdrp_amd.synth changed: regenerate tests/golden/classic_wide.npz"
            m, st, mask = po.estimate_classic(kind, p["x1"], p["x2"], po.ransac_opt(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, seed=0),
                                              po.bundle_opt(loss_type=4), cam if kind == 3 else None, cam if kind == 3 else None, pp=(0.0, 0.0))
            ref_st = g[f"{name}_stats"][j]
            assert (st.iterations, st.num_inliers) == (int(ref_st[1]), int(ref_st[2])) and st.refinements == int(g[f"{name}_oracle_refinements"][j])
            assert np.array_equal(mask, np.unpackbits(g[f"{name}_mask"][j])[:2000])
            assert wide_classic_model_diff(kind, m, g[f"{name}_model"][j]) < 1e-8
            if kind == 4:
                assert abs(m[10] - g[f"{name}_model"][j][7]) < 1e-8 * m[10]


@pytest.mark.parametrize("name", ["relpose_5pt", "shared_6pt", "fundamental_7pt"])
def test_randomised_options_vs_reference_fixture(golden, name):
    """tests/golden/options_ref_classic.npz (tests/tools/gen_golden_options_ref_classic.py): the reference binary on 64 cases per comparison row with size,
    outlier share, noise, threshold, seed, fixed / dynamic budget (up to 100 000 iterations), loss type, loss scale, bundle cap, cameras (5-point: two
    focal lengths, SIMPLE_PINHOLE / PINHOLE, principal point) and principal point (6-point) drawn at random.  Oracle == reference in iterations, inliers,
    mask, model (1e-6) and LO count on every 5- and 7-point case; 6-point: one other winner and two LO counts, enumerated."""
    from helpers import (CLASSIC_OPTIONS_KINDS, CLASSIC_OPTIONS_LO_DEVIATIONS, CLASSIC_OPTIONS_OTHER_WINNER, classic_options_cameras, classic_options_pair,
                         input_digest)
    g = golden("options_ref_classic")
    kind = CLASSIC_OPTIONS_KINDS[name]
    for j, row in enumerate(g["cases"]):
        n = int(row[0])
        p = classic_options_pair(name, j, row)
        assert input_digest(p) == g[f"{name}_digest"][j]
        if j in CLASSIC_OPTIONS_OTHER_WINNER.get(name, ()):
            continue
        ro = po.ransac_opt(max_iterations=int(row[5]), min_iterations=int(row[6]), max_epipolar_error=float(row[3]), seed=int(row[4]))
        bo = po.bundle_opt(max_iterations=int(row[9]), loss_type=int(row[7]), loss_scale=float(row[8]), gradient_tol=1e-10)
        c1, c2 = classic_options_cameras(row)
        cam1, cam2 = (po.cam_flat(*c1), po.cam_flat(*c2)) if kind == 3 else (None, None)
        m, st, mask = po.estimate_classic(kind, p["x1"], p["x2"], ro, bo, cam1, cam2, pp=(float(row[12]), float(row[13])))
        m, r, ist = np.asarray(m, float).reshape(-1), g[f"{name}_model"][j], g[f"{name}_istats"][j]
        assert (st.iterations, st.num_inliers) == (int(ist[1]), int(ist[2])), (name, j)
        assert (mask == np.unpackbits(g[f"{name}_mask"][j])[:n]).all(), (name, j)
        d = pose_diff(m, r[:7]) if kind == 3 else (fund_diff(m, r[:9]) if kind == 5 else pose_diff(m[:7], r[:7]) + abs(r[7] - m[10]) / abs(r[7]))
        assert d < 1e-6, (name, j, d)
        assert st.refinements - int(ist[0]) == CLASSIC_OPTIONS_LO_DEVIATIONS.get(name, {}).get(j, 0), (name, j, st.refinements, int(ist[0]))


def _classic_edge_model_equal(kind, m, r):
    """m: our flat model (q, t [, ..., f at 10] or F), r: the reference's (q, t [, f at 7] or F).  Degenerate answers (the identity pose the search starts
    from, NaN) must be equal as they stand; everything else to 1e-6 in the estimator's own gauge."""
    m, r = np.asarray(m, float).reshape(-1), np.asarray(r, float).reshape(-1)
    if kind == 5:
        return np.array_equal(m[:9], r[:9], equal_nan=True) or fund_diff(m[:9], r[:9]) < 1e-6
    f_ok = True if kind == 3 else abs(m[10] - r[7]) <= 1e-6 * abs(r[7])
    if not np.linalg.norm(r[4:7]) > 0 or not np.linalg.norm(m[4:7]) > 0:
        return bool(np.allclose(m[:7], r[:7], atol=1e-12, equal_nan=True) and f_ok)
    return bool(pose_diff(m[:7], r[:7]) < 1e-6 and f_ok)


@pytest.mark.parametrize("name", ["relpose_5pt", "shared_6pt", "fundamental_7pt"])
def test_edge_options_vs_reference_fixture(golden, name):
    """tests/golden/edge_options_ref_classic.npz: max_iterations 0 / 1 / below min_iterations, success_prob 0 / 1, dyn_num_trials_mult 0, thresholds 0 /
    1e-3 / 100 px, a 41-bit seed, loss_scale 0, pinned damping, tolerances of 1 — stats, mask and model identical to the reference binary on every case
    but the enumerated ties (threshold 0 with the 6-point solver: the first model scored stays; one LO count)."""
    from helpers import CLASSIC_EDGE_LO_DEVIATIONS, CLASSIC_EDGE_TIES, CLASSIC_OPTIONS_KINDS, classic_edge_cases, classic_edge_pair, input_digest
    g = golden("edge_options_ref_classic")
    kind = CLASSIC_OPTIONS_KINDS[name]
    p = classic_edge_pair(name)
    assert input_digest(p) == g[f"{name}_digest"]
    cam = po.cam_flat(0, [800.0, 0.0, 0.0]) if kind == 3 else None
    for j, (rod, bod) in enumerate(classic_edge_cases()):
        m, st, mask = po.estimate_classic(kind, p["x1"], p["x2"], po.ransac_opt(**rod), po.bundle_opt(**bod), cam, cam, pp=(0.0, 0.0))
        ref = g[f"{name}_stats"][j]
        assert (st.iterations, st.num_inliers) == (int(ref[1]), int(ref[2])), (name, j, rod, bod, st.iterations, st.num_inliers, ref)
        lo = next((v for k, v in CLASSIC_EDGE_LO_DEVIATIONS.get(name, {}).items() if rod.get(k.split("=")[0]) == float(k.split("=")[1])), 0)
        assert st.refinements - int(ref[0]) == lo, (name, j, rod, st.refinements, ref[0])
        if any(all(rod.get(k) == v for k, v in tie.items()) for tie in CLASSIC_EDGE_TIES.get(name, ())):
            continue
        assert (mask == np.unpackbits(g[f"{name}_mask"][j])[:300]).all(), (name, j)
        assert _classic_edge_model_equal(kind, m, g[f"{name}_model"][j]), (name, j, rod, bod, m, g[f"{name}_model"][j])
