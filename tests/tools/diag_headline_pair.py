#!/usr/bin/env python3
"""Build container only (needs the reference shim).  Why does a pair of a headline batch differ between the reference binary
and the CPU oracle?  Replays score_models<> over the minimal models of both sides' solvers (reference binary via refshim,
ours via the oracle), scoring every model with the reference's own scorer, and prints the iterations at which only one side
sets a record (= triggers an LO) together with the solver outputs of that sample.

    python tests/tools/diag_headline_pair.py calib_p3p_n2000_i10k|calib_shift_n2000_i10k INDEX [max_iterations]"""
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "..", ".."))
import numpy as np
import refshim as rs
from oracle import pyorc as po
import gen_golden_headline_ref as gh

w, idx = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
kind, es, n, of, rf, _ = gh.HEADLINE[w]
assert kind == 0, "calibrated estimators only (the focal solvers show no deviation on the headline batches)"
p = gh.make_pair(w, idx)
f = 800.0
a, b, d1, d2 = p["x1"] / f, p["x2"] / f, p["d1"], p["d2"]
thr = (2.0 / f) ** 2
S = po.draw_samples(0, n, iters)


def models_of(side, s):
    x1h = np.c_[a[s], np.ones(3)]; x2h = np.c_[b[s], np.ones(3)]
    if es:
        return rs.solver_calib(x1h, x2h, d1[s], d2[s]) if side == "ref" else po.solver_calib_shift(x1h, x2h, d1[s], d2[s])
    if side == "ref":
        X = x1h * d1[s][:, None]; xb = x2h / np.linalg.norm(x2h, axis=1, keepdims=True)
        return rs.p3p(xb, X)
    return po.solver_calib_p3p(x1h, x2h, d1[s], d2[s])


def run(side):
    bc, bs = 0, np.finfo(np.float64).max
    trig = {}
    for it, s in enumerate(S):
        sols = models_of(side, s)
        hit = False
        for m in sols:
            sc, c = rs.msac_pose(np.asarray(m[:7]), a, b, thr)
            if c > bc or sc < bs:  # NaN models: msac of a NaN pose is N * thr with 0 inliers (every comparison false)
                bc = max(bc, c); bs = min(bs, sc); hit = True
        if hit:
            trig[it] = (bc, bs, len(sols))
    return trig


r, o = run("ref"), run("orc")
print(f"{w} pair {idx}: LO triggers reference {len(r)}, oracle {len(o)}")
for it in sorted(set(r) ^ set(o)):
    side = "ref only" if it in r else "oracle only"
    print(f"  iteration {it}: {side}; record after it {r.get(it) or o.get(it)}")
    for sd in ("ref", "orc"):
        sols = models_of(sd, S[it])
        for m in sols:
            m = np.asarray(m)
            sc, c = rs.msac_pose(m[:7], a, b, thr)
            print(f"      {sd}: finite={np.isfinite(m).all()} inliers {c} score {sc:.9g} scale/shifts {m[7:10]}")
        if len(sols) == 0:
            print(f"      {sd}: no solutions")

# ---- the LO of every common trigger on both sides (refine_model: 25 iterations, TRUNCATED at the epipolar threshold)
thr_n, rep_n = 2.0 / f, 16.0 / f
sr = thr_n * thr_n / (rep_n * rep_n)
bo_r = rs.bopt(max_iterations=25, loss_type=1, loss_scale=thr_n, gradient_tol=1e-10, step_tol=1e-8, initial_lambda=1e-3)
bo_o = po.bundle_opt(max_iterations=25, loss_type=1, loss_scale=thr_n, gradient_tol=1e-10, step_tol=1e-8, initial_lambda=1e-3)
bc, bs = 0, np.finfo(np.float64).max
for it in sorted(set(r) & set(o)):
    trig = {}
    for sd in ("ref", "orc"):
        bc_, bs_ = bc, bs
        for m in models_of(sd, S[it]):
            sc, c = rs.msac_pose(np.asarray(m[:7]), a, b, thr)
            if c > bc_ or sc < bs_:
                bc_ = max(bc_, c); bs_ = min(bs_, sc); trig[sd] = np.asarray(m)
    bc, bs = r[it][0], r[it][1]
    mr, mo = trig["ref"], trig["orc"]
    start_diff = np.abs(mr[:10] - mo[:10]).max()
    gr, str_ = rs.refine_calib(a, b, d1, d2, mr[:10], sr, 1.0, bo_r, es)
    go, sto = po.refine(0, a, b, d1, d2, np.r_[mo[:10], 1.0, 1.0], sr, 1.0, bo_o, es)
    gx, stx = po.refine(0, a, b, d1, d2, np.r_[mr[:10], 1.0, 1.0], sr, 1.0, bo_o, es)  # the oracle's LM from the REFERENCE's start
    sr_, cr_ = rs.msac_pose(gr[:7], a, b, thr); so_, co_ = rs.msac_pose(go[:7], a, b, thr); sx_, cx_ = rs.msac_pose(gx[:7], a, b, thr)
    print(f"  LO at iteration {it}: start models differ by {start_diff:.2e}; refined score ref {sr_:.12g} ({cr_}) oracle {so_:.12g} ({co_}) "
          f"oracle-LM-from-ref-start {sx_:.12g} ({cx_}); LM iterations ref {int(str_[0])} oracle {sto.iterations}; refined models differ by {np.abs(gr[:10] - go[:10]).max():.2e}")
