#!/usr/bin/env python3
"""Register / spill / LDS table of EVERY kernel in the built library, read from the code objects themselves:
the .hip_fatbin section of libmdrp_hip.so is cut into its offload bundles (one per translation unit), the gfx950 code object of each is
written out and its AMDGPU metadata note is parsed (llvm-readelf --notes): VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes per
lane, static LDS, and the wavefronts per SIMD the 512-register file allows.

    python3 tools/kernel_table.py [lib.so] > profiles/rNN_kernel_table.txt        (no GPU needed; a diff of two rounds' tables shows regressions)
As a module: kernel_table(path) -> {demangled kernel name: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, waves_per_simd}}."""
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _tool(name):
    for c in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", name), os.path.join(LLVM, name), shutil.which(name)):
        if c and os.path.exists(c):
            return c
    raise RuntimeError(f"{name} not found")


def code_objects(lib):
    """the gfx950 code objects embedded in the library, as bytes (one per translation unit)"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin")
        subprocess.check_call([_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        data = open(fat, "rb").read()
    out = []
    o = data.find(MAGIC)
    while o >= 0:
        n = struct.unpack_from("<Q", data, o + 24)[0]
        p = o + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                out.append(data[o + off:o + off + size])
        o = data.find(MAGIC, o + 1)
    return out


def kernel_table(lib=None):
    lib = lib or os.path.join(ROOT, "mdrp_amd", "libmdrp_hip.so")
    rows = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([_tool("llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for ln in notes.splitlines():
            m = re.match(r"\s+(?:- )?(\.[a-z_]+):\s+(.*)$", ln)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == ".agpr_count" and ln.lstrip().startswith("- "):
                cur = {"agpr": int(v)}
            elif cur is not None:
                if k == ".name":
                    cur["name"] = v.strip("'\"")
                elif k in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size"):
                    cur[{".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill", ".sgpr_spill_count": "sgpr_spill",
                         ".private_segment_fixed_size": "scratch", ".group_segment_fixed_size": "lds"}[k]] = int(v)
                elif k == ".wavefront_size":
                    if "name" in cur:
                        rows[cur["name"]] = cur
    names = list(rows)
    dem = subprocess.run([shutil.which("c++filt") or _tool("llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for n, d in zip(names, dem):
        r = rows[n]
        d = re.sub(r"^void ", "", d.split("(")[0])
        regs = max(r.get("vgpr", 0), 1)  # gfx950: .vgpr_count is the unified total (arch VGPRs + AGPRs) of the 512-entry file per SIMD lane, granule 8
        r["waves_per_simd"] = min(8, 512 // (((regs + 7) // 8) * 8))
        out[d] = r
    return out


if __name__ == "__main__":
    t = kernel_table(sys.argv[1] if len(sys.argv) > 1 else None)
    print(f"{'kernel':72s} {'VGPR':>4} {'AGPR':>4} {'SGPR':>4} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'LDS':>6} {'waves/SIMD':>10}")
    for k in sorted(t):
        r = t[k]
        print(f"{k[:72]:72s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('sgpr', 0):4d} {r.get('vgpr_spill', 0):6d} {r.get('sgpr_spill', 0):6d} "
              f"{r.get('scratch', 0):7d} {r.get('lds', 0):6d} {r['waves_per_simd']:10d}")
