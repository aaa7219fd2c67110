#!/usr/bin/env python3
"""fp64 flop per correspondence of the LM sweeps, read off the ISA of the build (bench.py's fp64 roofline of k_lo / k_final):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast --cuda-device-only -S -o lm.s mdrp_amd/csrc/mdrp_capi.hip
    python3 tools/lm_flops.py lm.s
For every k_lo<KIND, SHIFT, 64> it finds the innermost loops (a label and the backward branch to it), counts their fp64 VALU
instructions (FMA-class = 2 flop: v_fma_f64, v_fmac_f64, v_div_fmas_f64; everything else named *_f64 = 1 flop) and reports
the normal-equation sweep (the largest loop: residuals + Jacobians + J'J) and the cost sweep (the second largest: residuals)."""
import re
import sys

FMA = ("v_fma_f64", "v_fmac_f64", "v_div_fmas_f64")


def flops(lines):
    f = 0
    n = 0
    for ln in lines:
        op = ln.split()[0] if ln.split() else ""
        if op.endswith("_f64") or "_f64_" in op:
            base = op.replace("_e32", "").replace("_e64", "")
            f += 2 if base in FMA else 1
            n += 1
    return f, n


def main(path):
    text = open(path).read().split("\n")
    starts = [(i, m.group(1)) for i, ln in enumerate(text) for m in [re.match(r"^(_ZN4mdrp(?:4k_loILi(\d)ELb(\d)ELi64EE|7k_finalILi(\d)ELb(\d)ELi256EE)\S*):", ln)] if m]
    for i0, name in starts:
        kname = "k_lo" if "4k_lo" in name else "k_final"
        kind, shift = re.match(r"_ZN4mdrp(?:4k_lo|7k_final)ILi(\d)ELb(\d)", name).groups()
        i1 = next(j for j in range(i0, len(text)) if ".end_amdhsa_kernel" in text[j] or text[j].startswith(".Lfunc_end"))
        body = text[i0:i1]
        labels = {m.group(1): j for j, ln in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
        loops = []
        for j, ln in enumerate(body):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] < j:
                loops.append((labels[m.group(1)], j))
        inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
        stats = []
        for a, b in inner:
            lines = [x.strip() for x in body[a:b + 1]]
            # the sweeps read one record per trip (<= 4 loads); the replay loop of the fused tail (walk_pair: ~40 loads and stores per
            # trigger, pow / log expansions) is not an LM sweep
            if sum(1 for x in lines if x.startswith(("global_", "flat_", "scratch_", "buffer_"))) > 12:
                continue
            f, n = flops(lines)
            ballot = any("v_mbcnt" in x or "v_bcnt" in x for x in body[a:b + 1])
            stats.append((f, n, ballot, b - a))
        acc = max(stats)
        cost = max((s for s in stats if 2 * s[0] < acc[0]), default=(0, 0, True, 0))  # the largest loop below half of it: residuals only
        print(f"{kname}<{kind}, {'true' if shift == '1' else 'false'}, {64 if kname == 'k_lo' else 256}>: cost sweep {cost[0]} flop ({cost[1]} fp64 instructions) per correspondence, "
              f"normal equations {acc[0]} flop ({acc[1]} fp64 instructions)")


if __name__ == "__main__":
    main(sys.argv[1])
