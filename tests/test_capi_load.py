"""The C-ABI library builds for gfx950 and exports every symbol include/mdrp.h declares (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

from mdrp_amd import _capi, build as b

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    b.build()
    lib = _capi.load_library()  # makes the process's one HIP runtime visible first: the library carries none of its own
    import shutil
    import subprocess
    readelf = shutil.which("readelf") or shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
    assert os.path.exists(readelf), "no readelf: the NEEDED check below would pass vacuously"
    dyn = subprocess.run([readelf, "-d", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "NEEDED" in dyn, dyn  # the dynamic section was really read (libstdc++, libc, ...)
    assert not [ln for ln in dyn.splitlines() if "NEEDED" in ln and "amdhip" in ln]
    hdr = open(os.path.join(ROOT, "include", "mdrp.h")).read()
    macros = set(re.findall(r"#define\s+(mdrp_[a-z_]+)\(", hdr))  # mdrp_create / mdrp_create_on_stream: macros over the versioned entry points
    assert macros == {"mdrp_create", "mdrp_create_on_stream"}
    declared = sorted(set(re.findall(r"\b(mdrp_[a-z_]+)\s*\(", hdr)) - macros)
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_capi.EXPORTS) == declared


def test_struct_layouts_match_header():
    assert C.sizeof(_capi.Model) == 96 and C.sizeof(_capi.Camera) == 40 and C.sizeof(_capi.Result) == 136
    assert C.sizeof(_capi.RansacOpt) == 88 and C.sizeof(_capi.BundleOpt) == 64  # ABI 0.4: + progressive_sampling, max_prosac_iterations, real_focal_check
    assert _capi.RansacOpt.progressive_sampling.offset == 68 and _capi.RansacOpt.max_prosac_iterations.offset == 72 and _capi.RansacOpt.real_focal_check.offset == 80
    hdr0 = open(os.path.join(ROOT, "include", "mdrp.h")).read()
    assert int(re.search(r"#define MDRP_ABI_VERSION (0x[0-9a-fA-F]+)", hdr0).group(1), 16) == _capi.ABI_VERSION == _capi.load_library().mdrp_abi_version()
    # mdrp_stats: every field is 8 bytes; the binding's field list must be the header's, in order
    hdr = open(os.path.join(ROOT, "include", "mdrp.h")).read()
    body = hdr[hdr.index("typedef struct {", hdr.index("fp64 sweep of the hypotheses")):hdr.index("} mdrp_stats;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:double|int64_t)\s+(\w+)\s*;", body)
    assert fields == [f for f, _ in _capi.Stats._fields_], fields
    assert C.sizeof(_capi.Stats) == 8 * len(fields)


def test_hip_build_version_is_exported_and_matches_the_runtime_major():
    """the library binds to the process's HIP runtime at load time; load_library() refuses a different major (here: equal)"""
    lib = _capi.load_library()
    lib.mdrp_hip_build_version.restype = C.c_int
    built = lib.mdrp_hip_build_version()
    v = C.c_int(0)
    assert _capi._hip_runtime.hipRuntimeGetVersion(C.byref(v)) == 0
    assert built // 10_000_000 == v.value // 10_000_000 and built > 0


def test_no_cpu_fallback_in_product():
    """the product package never touches the oracle"""
    for fn in os.listdir(os.path.join(ROOT, "mdrp_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "mdrp_amd", fn)).read()
            assert "oracle" not in src.replace("no CPU fallback", ""), fn


def test_option_dicts_follow_reference_defaults():
    ro = _capi.ransac_opt_from_dict({"lo_iterations": 25, "weight_sampson": 1.0})  # unknown keys ignored (make_pair.py:31-33)
    assert (ro.max_iterations, ro.min_iterations, ro.max_reproj_error, ro.max_epipolar_error, ro.seed) == (100000, 1000, 12.0, 1.0, 0)
    assert (ro.progressive_sampling, ro.max_prosac_iterations, ro.real_focal_check) == (0, 100000, 0)
    ro = _capi.ransac_opt_from_dict({"progressive_sampling": True, "max_prosac_iterations": 5000, "real_focal_check": True})
    assert (ro.progressive_sampling, ro.max_prosac_iterations, ro.real_focal_check) == (1, 5000, 1)  # handed to the library, which refuses them
    bo = _capi.bundle_opt_from_dict({"loss_type": "TRUNCATED_CAUCHY"})
    assert (bo.max_iterations, bo.loss_type, bo.loss_scale, bo.gradient_tol) == (100, 4, 1.0, 1e-10)


def test_library_carries_the_hash_of_its_sources():
    """mdrp_version() names the sources the binary was built from; build() rebuilds on a mismatch (a stale prebuilt
    library must never be mistaken for the tree)"""
    b.build()
    assert b.built_hash() == b.source_hash()
    assert _capi.library_source_hash() == b.source_hash()
    assert _capi.library_version().startswith("mdrp-hip")


def test_c_host_links_and_loads(tmp_path):
    """a plain C host links libmdrp_hip.so next to the ONE HIP runtime it uses (the library itself has no DT_NEEDED
    on one) and reaches the ABI; without a GPU mdrp_create reports MDRP_ERR_NO_DEVICE through the return code, not a crash."""
    import shutil
    import subprocess
    b.build()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if not (shutil.which("gcc") and os.path.exists(os.path.join(rocm, "lib", "libamdhip64.so"))):
        import pytest
        pytest.skip("no gcc / system HIP runtime")
    src = tmp_path / "host.c"
    src.write_text('#include "mdrp.h"\n#include <stdio.h>\nint main(void) {\n'
                   '    printf("%s hip %d\\n", mdrp_version(), mdrp_hip_build_version());\n'
                   '    mdrp_handle *h = 0; int rc = mdrp_create(0, NULL, &h);\n'
                   '    printf("create rc %d\\n", rc);\n'
                   '    if (!rc) { mdrp_stats st; rc = mdrp_last_stats_sized(h, &st, sizeof st); mdrp_destroy(h); }\n'
                   '    if (mdrp_abi_version() != MDRP_ABI_VERSION || sizeof(mdrp_ransac_opt) != 88) return 2;\n'
                   '    return rc == 0 || rc == MDRP_ERR_NO_DEVICE ? 0 : 1;\n}\n')
    exe = tmp_path / "host"
    libdir = os.path.dirname(_capi.LIB_PATH)
    subprocess.run(["gcc", str(src), "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lmdrp_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                    f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "mdrp-hip" in out.stdout and "create rc" in out.stdout, (out.stdout, out.stderr)


def test_handle_creation_refuses_a_host_compiled_against_another_abi():
    """mdrp_ransac_opt grew from 72 to 88 bytes in ABI 0.4 and nothing stopped a host compiled against 0.3 from passing the smaller
    struct.  Since 0.5 the header's mdrp_create / mdrp_create_on_stream are macros over mdrp_create_ / mdrp_create_on_stream_ that hand over the
    host's MDRP_ABI_VERSION and sizeof(mdrp_ransac_opt): a mismatch is MDRP_ERR_INVALID (1) with both versions in the message, before any device
    is touched — so this runs without a GPU."""
    lib = _capi.load_library()
    h = C.c_void_p()
    for fn in (lib.mdrp_create_, lib.mdrp_create_on_stream_):
        assert fn(0, None, C.byref(h), 0x00000003, 72) == 1 and b"recompile" in lib.mdrp_last_error() and not h.value
        assert fn(0, None, C.byref(h), _capi.ABI_VERSION, 72) == 1 and not h.value
    # the matching ABI gets past the check: on this box (no GPU) the next refusal is MDRP_ERR_NO_DEVICE / a HIP error, never INVALID
    rc = lib.mdrp_create_(0, None, C.byref(h), _capi.ABI_VERSION, C.sizeof(_capi.RansacOpt))
    assert rc != 1
    if rc == 0:
        lib.mdrp_destroy(h)
