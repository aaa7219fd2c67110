"""Host-side logic of the large-batch entry points (no GPU): how a batch is cut into chunks, and the camera records of a batch."""
import numpy as np
import pytest

from mdrp_amd import _capi, pipeline, poselib


def test_chunk_bounds_cover_the_batch_in_order(monkeypatch):
    assert pipeline.PIPELINE_MIN == 0 and pipeline.chunk_bounds(100000) == [(0, 100000)]  # the default since round 6: a host batch is ONE call (it copies in slices itself)
    monkeypatch.setattr(pipeline, "PIPELINE_MIN", 6144)  # MDRP_PIPELINE_MIN=6144: the chunking of rounds 4-5
    for B in (0, 1, 1024, 6144, 6145, 8192, 9000, 12500, 100000):
        for chunk in (None, 512, 1024, 4096):
            b = pipeline.chunk_bounds(B, chunk)
            assert b[0][0] == 0 and b[-1][1] == B and all(x[1] == y[0] for x, y in zip(b, b[1:])), (B, chunk, b[:3])
            size = chunk or pipeline.PIPELINE_CHUNK
            if B <= max(pipeline.PIPELINE_MIN, size):
                assert b == [(0, B)]
            else:
                assert all(hi - lo == size for lo, hi in b[:-1]) and size // 4 <= b[-1][1] - b[-1][0] < size + size // 4 + 1, (B, chunk, b[-2:])
    assert pipeline.chunk_bounds(8192) == [(1024 * i, 1024 * (i + 1)) for i in range(8)]
    assert pipeline.chunk_bounds(6144) == [(0, 6144)]          # up to PIPELINE_MIN pairs: one call
    assert pipeline.chunk_bounds(6144 + 1024 + 100)[-1] == (6144, 7268)  # a short remainder joins the last chunk
    assert pipeline.chunk_bounds(8192, 0) == [(0, 8192)]       # chunk 0: never split


def test_camera_records_of_a_batch():
    cam = {"model": "SIMPLE_PINHOLE", "width": 1600, "height": 1200, "params": [800.0, 1.5, -2.5]}
    r = poselib._camera_records(cam, 5)
    assert r.dtype == _capi.CAMERA_DTYPE and r.shape == (5,) and r.flags["C_CONTIGUOUS"]
    assert (r["model_id"] == 0).all() and np.array_equal(r["params"], np.tile([800.0, 1.5, -2.5, 0.0], (5, 1)))
    pin = {"model": "PINHOLE", "width": 10, "height": 10, "params": [700.0, 900.0, 1.0, 2.0]}
    r2 = poselib._camera_records([cam, pin, cam], 3)
    assert list(r2["model_id"]) == [0, 1, 0] and np.array_equal(r2["params"][1], [700.0, 900.0, 1.0, 2.0])
    assert poselib._camera_records(r2, 3) is not None and np.array_equal(poselib._camera_records(r2, 3), r2)
    with pytest.raises(ValueError):
        poselib._camera_records([cam, pin], 3)
    with pytest.raises(ValueError):
        poselib._camera_records(r2, 4)
