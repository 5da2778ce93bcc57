#!/usr/bin/env python3
"""dev: issue-cost profile of the hand-placed loop of fa_fwd_w64_kernel, from the generated code.

Model (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost' / one wave per SIMD): a wave issues in order; an MFMA holds the issue
for 8 cycles and the next MFMA cannot issue before 32 cycles after it; v_exp_f32 & co. 8, other VALU 4, DS / VMEM / scalar 4,
s_nop N: N + 1 (at least 4), waits 0.  A gap (MFMA .. next MFMA) therefore runs max(32, 8 + sum of its fillers); the loop's
predicted cycles are the sum over gaps.  Prints, per kernel, the gaps of the COMMON path of the two-tile loop body (the rare
rescale blocks are skipped) and the prediction, which tools/w64_stamp.py measures.   usage: w64_gaps.py [name-filter] [--flags ..]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from audit_w64 import compile_asm, kernels

TRANS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32")


# calibrated on tools/ubench/gapcost.hip (profiles/r02/gapcost.log): marginal cycles of one more instruction in a gap that is already full
COST = dict(mfma=8, trans=8, valu=4, salu=6, nop=2, wait=2, barrier=6, ds128=12, dstr=10, dma=40)


def cost(op, args):
    if op.startswith("v_mfma"): return COST["mfma"]
    if op.startswith(TRANS): return COST["trans"]
    if op == "s_nop": return COST["nop"] * (int(args[0]) + 1)
    if op.startswith("s_waitcnt"): return COST["wait"]
    if op == "s_barrier": return COST["barrier"]
    if op.startswith("ds_read_b128"): return COST["ds128"]
    if op.startswith("ds_"): return COST["dstr"]
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and "lds" in args[-1]): return COST["dma"]
    if op.startswith("s_"): return COST["salu"]
    return COST["valu"]


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "Li0ELb1"
    extra = sys.argv[sys.argv.index("--flags") + 1].split() if "--flags" in sys.argv else []
    bwd = "--bwd" in sys.argv          # the backward kernels of fa_bwd_w64.hpp (filter e.g. Li64ELi0ELi0E: E = 64, dK/dV, plain)
    text = compile_asm("fa_bwd_bf16.hip" if bwd else "fa_fwd_bf16.hip", extra)
    for name, body, meta in kernels(text, "fa_bwd_w64_kernel" if bwd else "fa_fwd_w64_kernel"):
        if flt not in name: continue
        lines = [l.split(";")[0].strip() for l in body.split("\n")]
        lines = [l for l in lines if l and (not l.startswith(".") or l.endswith(":"))]
        labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
        cands = []
        for i, l in enumerate(lines):
            m = re.match(r"s_c?branch\w* (\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                s = labels[m.group(1)]
                cands.append((s, i, sum(1 for x in lines[s:i] if x.startswith("v_mfma"))))
        # the hand-placed loop = the INNERMOST region with the most MFMAs (the persistent forward wraps it in a loop over blocks)
        inner = [c for c in cands if not any(o is not c and c[0] <= o[0] and o[1] <= c[1] and o[2] >= 16 and (o[0], o[1]) != (c[0], c[1]) for o in cands)]
        s, e, n = max(inner or cands, key=lambda c: c[2])
        # common path: skip forward-branch regions that contain accumulator reads (the rescale blocks)
        path, i = [], s
        while i <= e:
            l = lines[i]
            m = re.match(r"s_cbranch\w* (\S+)", l)
            rare = lambda blk: (any("v_accvgpr_read" in x for x in blk) or sum("v_cndmask" in x for x in blk) >= 24 or     # rescale / mask blocks
                                sum(x.startswith("s_nop 15") for x in blk) >= 4)                                         # the exit fence
            if m and m.group(1) in labels and i < labels[m.group(1)] <= e and rare(lines[i:labels[m.group(1)]]):
                i = labels[m.group(1)]
                continue
            if not l.endswith(":"): path.append(l)
            i += 1
        gaps, cur = [], None
        for l in path:
            op, _, rest = l.partition(" ")
            args = [a.strip() for a in rest.split(",")] if rest else []
            if op.startswith("v_mfma"):
                if cur is not None: gaps.append(cur)
                cur = [8, []]
            elif cur is not None:
                cur[0] += cost(op, args); cur[1].append(op)
        gaps.append(cur)
        tot = sum(max(32, g[0]) for g in gaps)
        print(f"{name}: {len(gaps)} MFMA gaps on the common path of 2 tiles; issue cost sum {sum(g[0] for g in gaps)}; predicted {tot} cycles = {tot / 2:.0f} per tile, {tot / len(gaps):.1f} per MFMA")
        print("  gap costs:", " ".join(str(g[0]) for g in gaps))
        if "--verbose" in sys.argv:
            for k, g in enumerate(gaps):
                if g[0] > 40: print(f"   gap {k}: {g[0]}: {' '.join(x.replace('_e32','').replace('_f32','') for x in g[1])}")


if __name__ == "__main__":
    main()
