"""Where the 5-point solver's time goes: kc_solver_unit stopped after each stage (experiment build with -DMDRP_5PT_STAGES,
MDRP_LIB pointing at it).  Run on the GPU box: python tools/stage_5pt.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "libmdrp_5pt_stages.so")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from mdrp_amd import _capi
    rng = np.random.default_rng(0)
    n = 1 << 19
    x1 = rng.uniform(-1, 1, (n, 5, 2)); x2 = x1 + rng.normal(size=x1.shape) * 0.3
    def unit(x):
        h = np.concatenate([x, np.ones(x.shape[:-1] + (1,))], axis=-1)
        return h / np.linalg.norm(h, axis=-1, keepdims=True)
    a, b = unit(x1), unit(x2)
    h = _capi.Handle(0)
    for _ in range(2):
        out, cnt = h.classic_solver_batch(3, a, b)
    print("stage", os.environ.get("MDRP_5PT_STAGE"), "solutions", int(cnt.sum()), flush=True)
else:
    from mdrp_amd import build
    build.build(force=True, defines=("MDRP_5PT_STAGES",), out=lib)
    for stage in (1, 2, 3, 4, 99):
        env = dict(os.environ, MDRP_LIB=lib, MDRP_5PT_STAGE=str(stage))
        subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
