"""Shared comparison helpers for the parity tests."""
import numpy as np


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def model_diff(a, b):
    """max of |dR| and relative difference of (t, scale, shifts, focals); a, b are 12-wide models."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    dR = np.abs(quat_to_R(a[:4]) - quat_to_R(b[:4])).max()
    rest = np.abs((a[4:] - b[4:]) / (1.0 + np.abs(b[4:]))).max()
    return max(dR, rest)


def match_solution_sets(A, B, tol=1e-6):
    """Greedy one-to-one matching of two solution lists; True iff same size and every pair within tol."""
    if len(A) != len(B):
        return False
    used = set()
    for a in A:
        best, bj = np.inf, -1
        for j, b in enumerate(B):
            if j in used:
                continue
            e = model_diff(a, b)
            if e < best:
                best, bj = e, j
        if best > tol:
            return False
        used.add(bj)
    return True


def widen(sol, width=12):
    out = np.ones(width)
    out[: len(sol)] = sol
    return out


# ---- enumerated deviations from the reference binary, by fixture index -------------------------------
# tests/golden/solvers.npz: problems on which the reference binary itself returns NaN models (nothing to compare)
REFERENCE_NAN_SOLUTIONS = {"p3p": (), "calib_shift": (21,), "shared": (), "varying": ()}

# tests/golden/estimate_full.npz: our LO count minus the reference's.  Found with the per-iteration replay of both
# sides' minimal models (tests/tools/diag_lo_count.py):
#   case 1 (calibrated P3P, pair 1): the reference's P3P returns four NaN poses for the sample of iteration 0; a NaN model
#           scores N * eps^2 < DBL_MAX, so it is the run's first "record" and costs the reference one LO that cannot
#           change the result; ours returns the empty set.
#   case 4 (calibrated + shift, pair 1): at iteration 2827 the reference's relpose_monodepth_3pt misses a true root
#           that ours finds; that model has 993 inliers, sets a record and costs us one LO.
# Iterations, inliers, score, mask and model are identical in both cases.
KNOWN_LO_COUNT_DEVIATIONS = {1: -1, 4: +1}

# tests/golden/initial.npz: case 11 (varying focal, ALL correspondences identical, score_initial_model): the LO that starts from
# the reset identity pose on fully degenerate data moves to a model with every correspondence as inlier in our LM; the
# reference's does not (its result keeps 0 inliers).  The HIP path does not run that LO at all (it starts from the state the
# reference ends up in) and matches the reference here; the oracle, which restates the reference's control flow, does not.
INITIAL_ORACLE_DEVIATIONS = (11,)
