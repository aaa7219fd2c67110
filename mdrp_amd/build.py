"""Compile the HIP library in-tree: hipcc --offload-arch=gfx950 -> mdrp_amd/libmdrp_hip.so (cross-compiles without a GPU).

The library embeds a hash of its source files (returned by mdrp_version()); build() rebuilds whenever the hash inside
the existing .so differs from the tree, so a prebuilt library can never silently be stale.
"""
import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mdrp_capi.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "mdrp_kernels.h"), os.path.join(HERE, "csrc", "mdrp_math.h"),
        os.path.join(HERE, "csrc", "mdrp_classic.h"), os.path.join(HERE, "csrc", "mdrp_classic_math.h"),
        os.path.join(HERE, "csrc", "mdrp_lm.h"), os.path.join(HERE, "csrc", "mdrp_logtab.h"),
        os.path.join(HERE, "..", "include", "mdrp.h")]
OUT = os.path.join(HERE, "libmdrp_hip.so")
_MARK = b"MDRP_SRC_HASH="
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=fast", "-no-hip-rt"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def source_hash():
    """sha256 over the source files (name + content), first 16 hex digits"""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())  # a change of the compile / link flags is a change of the binary
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_hash(path=OUT):
    """the hash embedded in an existing library (read from its bytes: no dlopen, no GPU), or None"""
    try:
        with open(path, "rb") as f:
            m = re.search(_MARK + rb"([0-9a-f]{16})", f.read())
        return m.group(1).decode() if m else None
    except OSError:
        return None


def up_to_date():
    return built_hash() == source_hash()


def build(force=False, verbose=False, defines=(), out=OUT, extra_flags=()):
    if not force and not defines and not extra_flags and out == OUT and up_to_date():
        return out
    # -no-hip-rt: no DT_NEEDED on a particular libamdhip64.  The hip* symbols stay undefined and bind, when the library is loaded,
    # to the ONE HIP runtime the host process already has (PyTorch-ROCm wheels bundle their own; a second runtime instance in the
    # same process cannot share streams or device memory with it).  mdrp_amd/_capi.py makes a runtime globally visible first;
    # a C / C++ host links -lamdhip64 itself.
    cmd = [hipcc(), *FLAGS, *extra_flags,
           f'-DMDRP_SRC_HASH="{source_hash()}"', *[f"-D{d}" for d in defines], SRC, "-o", out + ".tmp"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
