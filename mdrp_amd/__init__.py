"""mdrp_amd — MI355X-native RANSAC hot path of RePoseD (kocurvik/mdrp): the 3-point monodepth relative-pose
estimators behind poselib-compatible signatures, computed by hand-written HIP kernels for gfx950.

    import mdrp_amd.poselib as poselib      # drop-in module (single pair and *_batch entry points)
    from mdrp_amd import _capi               # thin ctypes binding of include/mdrp.h

The package has no CPU path: importing is cheap, but any estimator call needs libmdrp_hip.so and a gfx950 GPU.
"""
__all__ = ["poselib", "synth", "build", "dist"]
