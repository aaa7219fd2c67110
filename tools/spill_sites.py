#!/usr/bin/env python3
"""WHERE the spilled registers of a kernel are touched: the gfx950 code objects of the built library are
disassembled (llvm-objdump) and, per kernel, the scratch_load / scratch_store instructions are counted in total and INSIDE INNERMOST
LOOPS — the record loops of the LM sweeps, the scoring sweeps, the solvers' Newton loops.  A kernel whose spills sit outside its innermost
loops pays a handful of scratch accesses per LM iteration (tens of thousands of instructions), not per record.

    python3 tools/spill_sites.py [lib.so] [name filter] > profiles/rNN_spill_sites.txt          (no GPU needed)
As a module: spill_sites(path) -> {demangled kernel: {"insts", "scratch", "scratch_in_inner_loops", "scratch_in_sweep_loops" (innermost loops of >= 100
instructions: the per-record bodies), "inner_loops", "largest_inner_loop"}}."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_table as kt  # noqa: E402


SWEEP_MIN = 100


def spill_sites(lib=None, only=None):
    lib = lib or os.path.join(kt.ROOT, "mdrp_amd", "libmdrp_hip.so")
    out = {}
    for co in kt.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([kt._tool("llvm-objdump"), "-d", "--symbolize-operands", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur, body = None, []
        funcs = {}
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(_Z[^>]+)>:$", ln)
            if m:
                cur = m.group(1)
                funcs[cur] = body = []
                continue
            if cur is not None:
                body.append(ln)
        names = list(funcs)
        dem = subprocess.run([shutil.which("c++filt") or kt._tool("llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        for n, d in zip(names, dem):
            d = re.sub(r"^void ", "", d.split("(")[0])
            if only and only not in d:
                continue
            body = funcs[n]
            labels, insts = {}, []
            for ln in body:
                m = re.match(r"^[0-9a-f]+ <(L\d+)>:$", ln)
                if m:
                    labels[m.group(1)] = len(insts)
                elif ln.startswith("\t"):
                    insts.append(ln.split("//")[0].strip())
            loops = []
            for j, ins in enumerate(insts):
                m = re.match(r"s_c?branch\S*\s+(L\d+)", ins)
                if m and m.group(1) in labels and labels[m.group(1)] <= j:
                    loops.append((labels[m.group(1)], j))
            inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
            sc = [j for j, ins in enumerate(insts) if ins.startswith("scratch_")]
            # "sweep" loops: innermost loops of at least SWEEP_MIN instructions — the per-record bodies (400-800 instructions in the LM kernels); the short
            # innermost loops are publish / retry loops that run once per problem (a trigger's result store with its release fence, queue pops)
            sweeps = [l for l in inner if l[1] - l[0] + 1 >= SWEEP_MIN]
            out[d] = {"insts": len(insts), "scratch": len(sc), "scratch_in_inner_loops": sum(1 for j in sc if any(a <= j <= b for a, b in inner)),
                      "scratch_in_sweep_loops": sum(1 for j in sc if any(a <= j <= b for a, b in sweeps)),
                      "inner_loops": len(inner), "largest_inner_loop": max((b - a + 1 for a, b in inner), default=0)}
    return out


if __name__ == "__main__":
    t = spill_sites(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None, sys.argv[2] if len(sys.argv) > 2 else None)
    regs = kt.kernel_table(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None)
    print(f"{'kernel':64s} {'insts':>6} {'VGPR spill':>10} {'scratch ops':>11} {'in innermost loops':>18} {'of them in sweeps':>17} {'innermost loops':>15} {'largest':>8}")
    for k in sorted(t):
        r = t[k]
        if r["scratch"] == 0 and regs.get(k, {}).get("vgpr_spill", 0) == 0:
            continue
        print(f"{k[:64]:64s} {r['insts']:6d} {regs.get(k, {}).get('vgpr_spill', 0):10d} {r['scratch']:11d} {r['scratch_in_inner_loops']:18d} {r['scratch_in_sweep_loops']:17d} {r['inner_loops']:15d} {r['largest_inner_loop']:8d}")
    print(f"({sum(1 for k in t if t[k]['scratch'] == 0)} of {len(t)} kernels have no scratch instruction at all and are not listed)")
