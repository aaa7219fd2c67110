"""Compile the HIP library in-tree: hipcc --offload-arch=gfx950 -> mdrp_amd/libmdrp_hip.so (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mdrp_capi.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "mdrp_kernels.h"), os.path.join(HERE, "csrc", "mdrp_math.h"),
        os.path.join(HERE, "..", "include", "mdrp.h")]
OUT = os.path.join(HERE, "libmdrp_hip.so")


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    cmd = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=fast", SRC, "-o", OUT + ".tmp"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
