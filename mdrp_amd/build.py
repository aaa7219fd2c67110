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
SRC_TU = os.path.join(HERE, "csrc", "mdrp_tu.hip")  # secondary translation units: one compile per group of kernel instantiations
TU_GROUPS = (1, 2, 3)                                # k_final at 64 lanes | k_final at 256 lanes | 5- / 6- / 7-point baselines (mdrp_instances.h)
DEPS = [SRC, SRC_TU, os.path.join(HERE, "csrc", "mdrp_kernels.h"), os.path.join(HERE, "csrc", "mdrp_math.h"),
        os.path.join(HERE, "csrc", "mdrp_classic.h"), os.path.join(HERE, "csrc", "mdrp_classic_math.h"),
        os.path.join(HERE, "csrc", "mdrp_logtab.h"), os.path.join(HERE, "csrc", "mdrp_instances.h"),
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


def build(force=False, verbose=False, defines=(), out=OUT, extra_flags=(), single=None):
    """single=True: everything in ONE translation unit (mdrp_capi.hip instantiates every kernel itself) — the experiment builds with -D
    switches use it (their device-side globals, e.g. the trace buffers of MDRP_LO_TRACE, exist once).  Default: the split build — mdrp_capi.hip
    and the groups of mdrp_tu.hip compiled in parallel, then linked."""
    if not force and not defines and not extra_flags and out == OUT and up_to_date():
        return out
    if single is None:
        single = bool(defines) or bool(extra_flags)
    # -no-hip-rt: no DT_NEEDED on a particular libamdhip64.  The hip* symbols stay undefined and bind, when the library is loaded,
    # to the ONE HIP runtime the host process already has (PyTorch-ROCm wheels bundle their own; a second runtime instance in the
    # same process cannot share streams or device memory with it).  mdrp_amd/_capi.py makes a runtime globally visible first;
    # a C / C++ host links -lamdhip64 itself.
    common = [*FLAGS, *extra_flags, f'-DMDRP_SRC_HASH="{source_hash()}"', *[f"-D{d}" for d in defines]]
    if verbose:
        common.insert(0, "-Rpass-analysis=kernel-resource-usage")
    if single:
        subprocess.check_call([hipcc(), *common, SRC, "-o", out + ".tmp"])
        os.replace(out + ".tmp", out)
        return out
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    cflags = [f for f in common if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="mdrp_build_") as tmp:
        jobs = [([hipcc(), *cflags, "-DMDRP_SPLIT_TU", "-c", SRC, "-o", os.path.join(tmp, "capi.o")], os.path.join(tmp, "capi.o"))]
        for g in TU_GROUPS:
            obj = os.path.join(tmp, f"tu{g}.o")
            jobs.append(([hipcc(), *cflags, f"-DMDRP_TU={g}", "-c", SRC_TU, "-o", obj], obj))

        def run(job):
            r = subprocess.run(job[0], capture_output=True, text=True)
            return job, r
        with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
            results = list(pool.map(run, jobs))
        for (cmd, obj), r in results:
            if verbose or r.returncode != 0:
                import sys
                sys.stderr.write(r.stderr)
            if r.returncode != 0:
                raise subprocess.CalledProcessError(r.returncode, cmd, r.stdout, r.stderr)
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", *[o for _, o in jobs], "-o", out + ".tmp"])
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
