"""The batches bench.py times, pair by pair, through the entry point bench.py times.

tests/golden/headline_<workload>.npz (tests/tools/gen_golden_headline.py) holds the CPU oracle's output for EVERY pair of
the 1024-pair batches of BASELINE.json configs[1..3]: refinements, iterations, num_inliers, inlier_ratio, model_score, the
12-wide model and the packed inlier mask.  The oracle itself is pinned against the reference binary on 32 pairs spread over
each of these batches (tests/golden/estimate_wide.npz).  Here the same batches go through `mdrp_estimate_batch_async` on
DEVICE-RESIDENT buffers + `mdrp_fetch_results` — what bench.py's timed region calls — and every pair is compared.
A soak test repeats the headline step 200 times and demands bit-identical records and masks (fused tail on: the
stale-read race of round 3 showed once in ~10^4 pairs).  Needs an MI355X:  pytest -m gpu."""
import hashlib

import numpy as np
import pytest

from helpers import model_diff

pytestmark = pytest.mark.gpu

WORKLOADS = {
    # name: (kind, shift flag, n, outlier_frac, random_focal)  == bench.py WORKLOADS / gen_golden_headline.HEADLINE
    "calib_p3p_n2000_i10k": (0, False, 2000, 0.5, None),
    "shared_n2000_i10k": (1, False, 2000, 0.5, "shared"),
    "varying_n5000_i10k": (2, True, 5000, 0.5, "varying"),
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
        self.kind, es, self.n, of, rf = WORKLOADS[workload]
        self.B = B
        self.host = synth.make_batch(0, B, self.n, noise_px=0.5, depth_noise=0.02, outlier_frac=of, random_focal=rf)
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
    agree to ~1e-14 compared with `<`, decided by FMA contraction — at most 4 pairs in 1024 may differ, by exactly one LO,
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
    assert len(off) <= 4 and (np.abs(dlo[off]) == 1).all(), (workload, off, dlo[off])
    print(f"{workload}: 1024 / 1024 pairs identical (iterations, inliers, mask); worst model diff {worst:.2e}; LO count differs on {len(off)} pairs {off.tolist()}")


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
