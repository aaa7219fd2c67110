#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own compiled PoseLib (via oracle/_ref/librefshim.so).

Runs only in the build container (needs /root/reference/demo/poselib-2.0.5-*.whl; `make -C oracle ref`
first).  The fixtures are data only: inputs and the reference binary's outputs.  Re-running reproduces the
files bit for bit (all randomness is seeded).

    python3 tests/tools/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import refshim as rs  # noqa: E402
from mdrp_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "..", "golden")


def rodrigues(w):
    return synth.rodrigues(np.asarray(w, dtype=np.float64))


def pad(sols, width):
    out = np.full((4, width), np.nan)
    out[: len(sols), : sols.shape[1]] = sols
    return out


def gen_sampler():
    d = {}
    for n in (7, 200, 2000, 5000):
        for seed in (0, 5):
            d[f"n{n}_s{seed}"] = rs.draw_samples(seed, n, 64).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **d)


def minimal_problem(rng, kind, noisy):
    R = rodrigues(rng.normal(0, 0.3, 3))
    t = rng.normal(0, 0.5, 3)
    X = np.stack([rng.uniform(-2, 2, 3), rng.uniform(-1.5, 1.5, 3), rng.uniform(2, 8, 3)], 1)
    Y = X @ R.T + t
    s = rng.uniform(0.3, 3)
    u = rng.uniform(-0.5, 0.5) if kind == "calib_shift" else 0.0
    v = rng.uniform(-0.5, 0.5) if kind == "calib_shift" else 0.0
    f1 = f2 = 1.0
    if kind == "shared":
        f1 = f2 = rng.uniform(0.3, 2)
    if kind == "varying":
        f1, f2 = rng.uniform(0.3, 2, 2)
    x1 = X / X[:, 2:]
    x2 = Y / Y[:, 2:]
    x1[:, :2] *= f1
    x2[:, :2] *= f2
    d1 = X[:, 2] - u
    d2 = Y[:, 2] / s - v
    if noisy:
        x2[:, :2] += rng.normal(0, 0.1, (3, 2))
        d2 = d2 * (1 + rng.normal(0, 0.1, 3))
    return x1, x2, d1, d2


def gen_solvers(count=96):
    rng = np.random.default_rng(20240)
    d = {}
    # p3p: unit bearings + 3D points
    xs, Xs, sols, ns = [], [], [], []
    for i in range(count):
        R = rodrigues(rng.normal(0, 0.8, 3))
        t = rng.normal(size=3)
        X = rng.uniform(-2, 2, (3, 3))
        X[:, 2] += 5
        Y = X @ R.T + t
        if i % 2:
            Y = rng.uniform(-2, 2, (3, 3)) + [0, 0, 4]
        x = Y / np.linalg.norm(Y, axis=1, keepdims=True)
        s = rs.p3p(x, X)
        xs.append(x); Xs.append(X); sols.append(pad(s, 7)); ns.append(len(s))
    d.update(p3p_x=np.array(xs), p3p_X=np.array(Xs), p3p_sols=np.array(sols), p3p_n=np.array(ns, dtype=np.int32))
    for kind, fn, w in (("calib_shift", rs.solver_calib, 10), ("shared", rs.solver_shared, 12), ("varying", rs.solver_varying, 12)):
        a, b, c, e, sols, ns = [], [], [], [], [], []
        for i in range(count):
            x1, x2, d1, d2 = minimal_problem(rng, kind, i % 2)
            s = fn(x1, x2, d1, d2)
            a.append(x1); b.append(x2); c.append(d1); e.append(d2); sols.append(pad(s, w)); ns.append(len(s))
        d.update({f"{kind}_x1": np.array(a), f"{kind}_x2": np.array(b), f"{kind}_d1": np.array(c), f"{kind}_d2": np.array(e),
                  f"{kind}_sols": np.array(sols), f"{kind}_n": np.array(ns, dtype=np.int32)})
    np.savez_compressed(os.path.join(OUT, "solvers.npz"), **d)


def quat_of(R):
    from scipy.spatial.transform import Rotation as Rot
    q = Rot.from_matrix(R).as_quat()
    return np.array([q[3], q[0], q[1], q[2]])


def gen_scoring():
    d = {}
    rng = np.random.default_rng(7)
    for i in range(6):
        p = synth.make_pair(500 + i, 256, noise_px=1.0, outlier_frac=0.3)
        x1, x2 = p["x1"] / 800.0, p["x2"] / 800.0
        R = p["R"] if i % 2 == 0 else rodrigues(rng.normal(0, 1.0, 3))
        t = p["t"] if i % 3 else rng.normal(size=3)
        m = np.zeros(12)
        m[:4] = quat_of(R); m[4:7] = t; m[7] = 1.0; m[10] = 1.3; m[11] = 0.7
        thr = (2.0 / 800.0) ** 2
        s, c = rs.msac_pose(m[:7], x1, x2, thr)
        mask = rs.inliers_pose(m[:7], x1, x2, thr)
        # F of the model with the oracle convention diag(1,1,f2) E diag(1,1,f1), computed here independently
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        F = np.diag([1, 1, m[11]]) @ (tx @ R) @ np.diag([1, 1, m[10]])
        sF, cF = rs.msac_F(F, x1, x2, thr)
        maskF = rs.inliers_F(F, x1, x2, thr)
        d.update({f"x1_{i}": x1, f"x2_{i}": x2, f"model_{i}": m, f"R_{i}": R, f"F_{i}": F, f"thr_{i}": thr,
                  f"pose_score_{i}": s, f"pose_cnt_{i}": c, f"pose_mask_{i}": mask,
                  f"F_score_{i}": sF, f"F_cnt_{i}": cF, f"F_mask_{i}": maskF})
    np.savez_compressed(os.path.join(OUT, "scoring.npz"), **d)


def gen_refine():
    d = {}
    cases = []
    for i in range(9):
        kind = i % 3
        es = (i // 3) % 2 if kind == 0 else 0
        p = synth.make_pair(700 + i, 160, noise_px=1.0, depth_noise=0.05, outlier_frac=0.25,
                            random_focal=[None, "shared", "varying"][kind], shift1=0.2 if es else 0.0, shift2=-0.1 if es else 0.0)
        sc = 800.0 if kind == 0 else 700.0
        x1, x2 = p["x1"] / sc, p["x2"] / sc
        rng = np.random.default_rng(i)
        Rn = p["R"] @ rodrigues(rng.normal(0, 0.02, 3))
        m = np.zeros(12)
        m[:4] = quat_of(Rn); m[4:7] = p["t"] + rng.normal(0, 0.02, 3); m[7] = p["scale"] * 1.03
        m[10] = p["f1"] / sc * 1.02 if kind else 1.0
        m[11] = (p["f2"] / sc * 0.97) if kind == 2 else m[10]
        d.update({f"x1_{i}": x1, f"x2_{i}": x2, f"d1_{i}": p["d1"], f"d2_{i}": p["d2"], f"model_{i}": m})
        for lt in (0, 1, 2, 3, 4, 5):
            for its in (0, 1, 25):
                thr = 2.0 / sc
                bo = rs.bopt(max_iterations=its, loss_type=lt, loss_scale=thr, gradient_tol=1e-10)
                if kind == 0:
                    g, st = rs.refine_calib(x1, x2, p["d1"], p["d2"], m[:10], 1 / 64.0, 1.0, bo, es)
                    g = np.r_[g, 1.0, 1.0]
                else:
                    g, st = rs.refine_focal(kind == 2, x1, x2, p["d1"], p["d2"], m, 1 / 64.0, 1.0, bo)
                cases.append([i, kind, es, lt, its, thr])
                d[f"out_{len(cases) - 1}"] = np.r_[g, st]
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "refine.npz"), **d)


def gen_estimate():
    d = {}
    cases = []
    for i in range(18):
        kind = i % 3
        es = (i // 3) % 2 if kind == 0 else 0
        noise = 0.0 if i % 6 >= 3 else 0.5
        n = [200, 300, 400][i % 3]
        of = [0.0, 0.33, 0.5][(i // 2) % 3] if noise else 0.0
        pp = (640.0, 480.0) if kind == 0 else (0.0, 0.0)
        p = synth.make_pair(900 + i, n, noise_px=noise, depth_noise=0.02 if noise else 0.0, outlier_frac=of,
                            random_focal=[None, "shared", "varying"][kind], shift1=0.2 if es else 0.0,
                            shift2=-0.1 if es else 0.0, pp=pp, f1=800.0, f2=650.0)
        iters = 1000
        kw = dict(max_iterations=iters if i % 2 else 100000, min_iterations=iters, max_epipolar_error=2.0,
                  max_reproj_error=16.0, seed=i % 4, estimate_shift=es)
        lt = 4 if i % 4 else 3
        c1 = c2 = None
        cam1 = cam2 = np.zeros(6)
        if kind == 0:
            if i % 2:
                cam1 = np.array([0, 3, 800, 640, 480, 0.0]); cam2 = np.array([1, 4, 640, 660, 640, 480.0])
            else:
                cam1 = np.array([0, 3, 800, 640, 480, 0.0]); cam2 = np.array([0, 3, 650, 640, 480, 0.0])
            c1 = rs.cam_flat(int(cam1[0]), 1280, 960, list(cam1[2:2 + int(cam1[1])]))
            c2 = rs.cam_flat(int(cam2[0]), 1280, 960, list(cam2[2:2 + int(cam2[1])]))
        m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(**kw), rs.bopt(loss_type=lt), c1, c2)
        m12 = np.r_[m, 1.0, 1.0] if kind == 0 else m
        d.update({f"x1_{i}": p["x1"], f"x2_{i}": p["x2"], f"d1_{i}": p["d1"], f"d2_{i}": p["d2"], f"cam1_{i}": cam1,
                  f"cam2_{i}": cam2, f"model_{i}": m12, f"stats_{i}": st, f"mask_{i}": mask,
                  f"gt_R_{i}": p["R"], f"gt_t_{i}": p["t"], f"gt_{i}": np.array([p["scale"], p["shift1"], p["shift2"], p["f1"], p["f2"]])})
        cases.append([i, kind, es, noise, of, kw["max_iterations"], kw["min_iterations"], kw["seed"], lt])
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "estimate.npz"), **d)


FULL_CASES = (
    # name, kind, estimate_shift, n, outlier_frac, random_focal, first synth index, pairs   (BASELINE.json configs[1..3])
    ("calib_p3p", 0, 0, 2000, 0.5, None, 0, 2),        # bench.py's headline workload, pairs 0 and 1
    ("calib_p3p_clean", 0, 0, 2000, 0.0, None, 0, 1),  # SURVEY.md 8d C2 also names the outlier-free shape
    ("calib_shift", 0, 1, 2000, 0.5, None, 0, 2),
    ("shared", 1, 0, 2000, 0.5, "shared", 0, 2),
    ("varying_shiftflag", 2, 1, 5000, 0.5, "varying", 0, 2),  # configs[3]: the flag is set and ignored by the reference
)


def gen_estimate_full():
    """BASELINE.json's full-size shapes through the reference binary: N = 2000 / 5000, max = min = 10^4 iterations
    (make_video.py:192-194), eps = 2 px, rep = 16 px, TRUNCATED_CAUCHY; inputs exactly as bench.py generates them."""
    d = {}
    cases = []
    for name, kind, es, n, of, rf, first, pairs in FULL_CASES:
        for j in range(pairs):
            p = synth.make_pair(first + j, n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf,
                                shift1=0.2 if es and kind == 0 else 0.0, shift2=-0.1 if es and kind == 0 else 0.0)
            kw = dict(max_iterations=10000, min_iterations=10000, max_epipolar_error=2.0, max_reproj_error=16.0, seed=0, estimate_shift=bool(es))
            c1 = c2 = None
            if kind == 0:
                c1 = c2 = rs.cam_flat(0, 1600, 1200, [800.0, 0.0, 0.0])
            m, st, mask = rs.estimate(kind, p["x1"], p["x2"], p["d1"], p["d2"], rs.ropt(**kw), rs.bopt(loss_type=4), c1, c2)
            i = len(cases)
            d.update({f"x1_{i}": p["x1"], f"x2_{i}": p["x2"], f"d1_{i}": p["d1"], f"d2_{i}": p["d2"],
                      f"model_{i}": np.r_[m, 1.0, 1.0] if kind == 0 else m, f"stats_{i}": st, f"mask_{i}": np.packbits(mask)})
            cases.append([i, kind, es, n, of, first + j])
            print(name, j, "stats", st, flush=True)
    d["cases"] = np.array(cases)
    d["names"] = np.array([c[0] for c in FULL_CASES for _ in range(c[7])])
    np.savez_compressed(os.path.join(OUT, "estimate_full.npz"), **d)


def gen_initial():
    """initial_pose / score_initial_model (_core.pyi:455, RansacOptions +0x49) through the reference binary.  Black-box finding
    pinned here: the pose handed in is never read (ransac_*_relpose reset it), its scale / shifts survive only when RANSAC
    adopts nothing, and score_initial_model scores that reset model first (one more refinement, records start at N eps^2)."""
    d = {}
    cases = []
    rng = np.random.default_rng(5)
    for i in range(12):
        kind = i % 3
        degenerate = i >= 9           # all correspondences identical: the solvers return nothing (or NaN models)
        flag = (i // 3) % 2 == 0 or degenerate
        n = 150
        p = synth.make_pair(1500 + i, n, noise_px=0.5, depth_noise=0.02, outlier_frac=0.3, random_focal=[None, "shared", "varying"][kind],
                            pp=(640.0, 480.0) if kind == 0 else (0.0, 0.0))
        x1, x2, d1, d2 = p["x1"], p["x2"], p["d1"], p["d2"]
        if degenerate:
            x1 = np.tile(x1[:1], (n, 1)); x2 = np.tile(x2[:1], (n, 1)); d1 = np.full(n, 2.0); d2 = np.full(n, 3.0)
        ini = np.r_[quat_of(rodrigues(rng.normal(0, 0.5, 3))), rng.normal(0, 1, 3), 0.8 + 0.1 * i, 0.0, 0.0]
        if kind:
            ini = np.r_[ini, 1.3, 0.7]
        its = 1 if i % 4 == 1 else 200
        kw = dict(max_iterations=its, min_iterations=its, max_epipolar_error=2.0, max_reproj_error=16.0, seed=i % 3)
        c = rs.cam_flat(0, 1280, 960, [800.0, 640.0, 480.0]) if kind == 0 else None
        m, st, mask = rs.estimate(kind, x1, x2, d1, d2, rs.ropt(**kw), rs.bopt(loss_type=4), c, c, initial=ini, score_initial=flag)
        d.update({f"x1_{i}": x1, f"x2_{i}": x2, f"d1_{i}": d1, f"d2_{i}": d2, f"initial_{i}": ini,
                  f"model_{i}": np.r_[m, 1.0, 1.0] if kind == 0 else m, f"stats_{i}": st, f"mask_{i}": mask})
        cases.append([i, kind, int(flag), its, kw["seed"], int(degenerate)])
        print("initial", i, kind, flag, its, st, np.round(m[:8], 3), flush=True)
    d["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "initial.npz"), **d)


def unit_rows(x):
    h = np.c_[x, np.ones(len(x))]
    return np.ascontiguousarray(h / np.linalg.norm(h, axis=1, keepdims=True))


def gen_classic():
    """Non-monodepth baselines of the same binary: relpose_5pt / relpose_7pt solution lists IN THE BINARY'S ORDER,
    refine_relpose / refine_fundamental, estimate_relative_pose / estimate_fundamental (kind 3 / 5)."""
    d = {}
    rng = np.random.default_rng(77)
    # ---- solvers: geometric problems (with a little noise) and random ones
    for kind, K, width in ((3, 5, 7), (5, 7, 9)):
        xs1, xs2, sols, cnts = [], [], [], []
        for i in range(96):
            if i % 3 != 2:
                pr = synth.make_pair(9000 + 100 * kind + i, K, f1=1.0, f2=1.0, noise_px=0.0 if i % 3 == 0 else 1e-3)
                a, b = unit_rows(pr["x1"]), unit_rows(pr["x2"])
            else:
                a, b = unit_rows(rng.uniform(-1, 1, (K, 2))), unit_rows(rng.uniform(-1, 1, (K, 2)))
            out = rs.relpose_5pt(a, b) if kind == 3 else rs.relpose_7pt(a, b).reshape(-1, 9)
            full = np.full((10, width), np.nan)
            full[: len(out)] = out
            xs1.append(a); xs2.append(b); sols.append(full); cnts.append(len(out))
        d[f"solver{kind}_x1"] = np.array(xs1); d[f"solver{kind}_x2"] = np.array(xs2)
        d[f"solver{kind}_sols"] = np.array(sols); d[f"solver{kind}_n"] = np.array(cnts)
        print("classic solver", kind, "solutions per problem:", np.bincount(cnts), flush=True)
    # ---- refinement
    ref_cases = []
    for i in range(24):
        kind = 3 if i % 2 == 0 else 5
        loss = [1, 4, 3, 0, 5, 2][(i // 2) % 6]
        its = [1, 5, 100][(i // 2) % 3]
        pr = synth.make_pair(9500 + i, 80, f1=1.0 if kind == 3 else 1.3, f2=1.0 if kind == 3 else 1.3, noise_px=2e-3, outlier_frac=0.2)
        Rp = rodrigues(rng.normal(0, 0.03, 3)) @ pr["R"]
        tp = pr["t"] / np.linalg.norm(pr["t"]) + rng.normal(0, 0.05, 3)
        if kind == 3:
            m0 = np.r_[quat_of(Rp), tp]
        else:
            # a GENERIC rank-2 F (distinct singular values).  For an exact essential matrix the factorisation U diag(1, s, 0) V'
            # refine_fundamental works in is not unique (any common rotation of the first two singular vectors), and the
            # per-iteration path then depends on Eigen's JacobiSVD picking one member — the estimators never hit that case
            tx = np.array([[0, -tp[2], tp[1]], [tp[2], 0, -tp[0]], [-tp[1], tp[0], 0]])
            K = np.diag([1.0, 1.0, 1.3])
            F = K @ tx @ Rp @ K
            m0 = (F / np.linalg.norm(F)).reshape(-1)
        bo = rs.bopt(max_iterations=its, loss_type=loss, loss_scale=0.004)
        m, st = rs.refine_classic(kind, pr["x1"], pr["x2"], m0, bo)
        d.update({f"refine_x1_{i}": pr["x1"], f"refine_x2_{i}": pr["x2"], f"refine_m0_{i}": m0, f"refine_m_{i}": m, f"refine_stats_{i}": st})
        ref_cases.append([i, kind, loss, its])
    d["refine_cases"] = np.array(ref_cases)
    # ---- estimators: small shapes with every loss type + one BASELINE-sized case per estimator (N = 2000, 10^4 iterations)
    est_cases = []
    for i in range(18):
        kind = 3 if i % 2 == 0 else 5
        big = i >= 16
        n = 2000 if big else [60, 150, 400, 1000][(i // 2) % 4]
        outl = 0.5 if big else [0.1, 0.3, 0.5, 0.0][(i // 2) % 4]
        its = 10000 if big else [50, 300, 1000, 200][(i // 2) % 4]
        min_its = its if (big or i % 4 < 2) else 20          # half of the small cases stop by the dynamic rule
        loss = 4 if big else [4, 1, 3, 0, 5, 2][(i // 2) % 6]
        f = 800.0 if i % 3 else 1100.0
        pr = synth.make_pair(9700 + i, n, f1=f, f2=f, pp=(640.0, 480.0), noise_px=0.5, outlier_frac=outl)
        thr = [2.0, 1.0][i % 2 if not big else 0]
        ro = rs.ropt(max_iterations=its, min_iterations=min_its, max_epipolar_error=thr, seed=i % 5)
        bo = rs.bopt(loss_type=loss, loss_scale=thr)
        cam1 = rs.cam_flat(0, 1280, 960, [f, 640.0, 480.0])
        cam2 = rs.cam_flat(1, 1280, 960, [f * 1.01, f * 0.99, 640.0, 480.0])
        m, st, mask = rs.estimate_classic(kind, pr["x1"], pr["x2"], ro, bo, cam1, cam2)
        d.update({f"est_x1_{i}": pr["x1"], f"est_x2_{i}": pr["x2"], f"est_model_{i}": m, f"est_stats_{i}": st, f"est_mask_{i}": mask})
        est_cases.append([i, kind, n, its, min_its, loss, thr, i % 5, f])
        print("classic estimate", i, kind, n, its, st, flush=True)
    d["est_cases"] = np.array(est_cases)
    # ---- estimate_relative_pose with an initial pose (the binding sets score_initial_model; ransac_relpose resets the pose itself)
    init_cases = []
    for i in range(4):
        n, its = [150, 400, 60, 300][i], [300, 1000, 1, 500][i]
        pr = synth.make_pair(9900 + i, n, f1=800.0, f2=800.0, pp=(640.0, 480.0), noise_px=0.5, outlier_frac=[0.3, 0.5, 0.1, 0.4][i])
        ro = rs.ropt(max_iterations=its, min_iterations=its if i != 1 else 100, max_epipolar_error=2.0, seed=i)
        bo = rs.bopt(loss_type=4, loss_scale=2.0)
        cam = rs.cam_flat(0, 1280, 960, [800.0, 640.0, 480.0])
        ini = np.r_[quat_of(rodrigues(rng.normal(0, 0.5, 3))), rng.normal(0, 1, 3)]
        m, st, mask = rs.estimate_classic(3, pr["x1"], pr["x2"], ro, bo, cam, cam, score_initial=True, initial=ini)
        d.update({f"init_x1_{i}": pr["x1"], f"init_x2_{i}": pr["x2"], f"init_pose_{i}": ini, f"init_model_{i}": m, f"init_stats_{i}": st, f"init_mask_{i}": mask})
        init_cases.append([i, n, its, its if i != 1 else 100, i])
        print("classic initial", i, st, flush=True)
    d["init_cases"] = np.array(init_cases)
    # samples of 5 / 7 indices (draw_sample @0x4f87f0)
    for K in (5, 7):
        for n in (9, 200, 2000):
            d[f"samples_k{K}_n{n}"] = rs.draw_samples_k(3, n, K, 64).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "classic.npz"), **d)


if __name__ == "__main__":
    if not rs.available():
        sys.exit("reference shim not built: run `make -C oracle ref` in the build container")
    os.makedirs(OUT, exist_ok=True)
    gen_sampler()
    gen_solvers()
    gen_scoring()
    gen_refine()
    gen_estimate()
    gen_estimate_full()
    gen_initial()
    gen_classic()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
