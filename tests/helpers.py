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
