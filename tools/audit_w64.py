#!/usr/bin/env python3
"""Static audit of the generated code of the inline-asm MFMA kernels (fa_fwd_w64.hpp, fa_bwd_w64.hpp).

hipcc treats an `asm` MFMA as an ordinary 1-cycle instruction: it pads no wait states behind it, and its register allocator is
free to split the live range of an MFMA result and put a copy (v_accvgpr_mov / v_accvgpr_read / v_mov) directly behind the MFMA
that produced it -- which then reads the accumulator registers the MFMA's last pass has not written yet (found the hard way:
stale registers 13..15 of one O tile, profiles/r02/NOTES.md).  The source keeps such copies away from MFMAs (fences at the three
places where O is read, a fence at the end of a wave's last iteration); this script checks the RESULT, per kernel:

  1. no spill code (scratch, v_writelane / v_readlane) inside the hand-placed loop, at most MAX_PROLOGUE_SCRATCH bytes per lane around it;
  2. every compiler-generated read of an accumulator register (v_accvgpr_read_b32, v_accvgpr_mov_b32 source) that an MFMA wrote is
     separated from the closest preceding MFMA writing that register by a fence (>= 4 x `s_nop 15`) or by >= MIN_GAP instructions;
  3. the same for v_mov_b32 / VALU reads of VGPR tuples written by an MFMA is left to the slot schedule (checked: no v_mov_b32 from
     an MFMA destination within MIN_GAP instructions);
  4. M0 is touched only by the LDS-DMA statements.

usage: audit_w64.py [--only fwd|bwd] [--flags "..."]   exit code 1 on any violation.  Used by tests/test_w64_codegen.py.
"""
import os
import re
import subprocess
import sys
import tempfile

MAX_PROLOGUE_SCRATCH = 128          # bytes per lane a kernel may park around (never inside) its hand-placed loop
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nnop.jl_amd", "csrc")


def compile_asm(src, extra):
    """device assembly of one translation unit.  Cached under the temp dir by a hash of every source in csrc/ and the flags: the audit,
    the schedule model (w64_gaps.py) and the codegen tests all look at the same few compiles (a minute each)."""
    import glob, hashlib
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.h*")) + glob.glob(os.path.join(CSRC, "*.inc")) + [os.path.join(CSRC, src)]):
        hsh.update(f.encode()); hsh.update(open(f, "rb").read())
    hsh.update(" ".join(extra).encode())
    cache = os.path.join(tempfile.gettempdir(), "nnop_asm_cache")
    os.makedirs(cache, exist_ok=True)
    hit = os.path.join(cache, f"{src}.{hsh.hexdigest()[:24]}.s")
    if os.path.exists(hit):
        return open(hit).read()
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False, dir=cache).name
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S",
           os.path.join(CSRC, src), "-o", out] + extra
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.replace(out, hit)
    return text


def prewarm(jobs):
    """compile several (src, extra flags) pairs at once into the cache (the compiles are independent minutes of single-threaded hipcc)"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
        list(ex.map(lambda j: compile_asm(*j), jobs))


def kernels(text, pattern):
    for m in re.finditer(r"^(_ZN4nnop\w*" + pattern + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        meta = re.search(r"\.name:\s+" + re.escape(m.group(1)) + r"\n(.*?)\.wavefront_size", text, re.S)
        yield m.group(1), m.group(2), (meta.group(1) if meta else "")


def regs(tok):
    m = re.match(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    m = re.match(r"([av])(\d+)$", tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(2))
    return None


# Cycle model (deliberately one-sided): time advances by the LEAST an instruction can take on a SIMD that runs one wave
# (MI355X_MICROARCH.md, vector-instruction issue costs: VALU 4, transcendental 8, s_nop N: N + 1, scalar 1, waits 0); an MFMA
# issues when the matrix pipe is free (32 cycles per v_mfma_f32_32x32x16: back-to-back MFMAs of one wave issue at that cadence)
# and its last accumulator registers count as written DONE_LAT cycles after its issue (ISA: 11 wait states of 4 cycles for an
# 8-pass MFMA; 64 leaves a margin).  A reader modelled earlier than that is reported.
MFMA_CYCLES, DONE_LAT = 32, 64


def cost(op, args):
    if op == "s_nop":
        return int(args[0]) + 1
    if op.startswith("s_waitcnt") or op == "s_barrier":
        return 0
    if op.startswith("s_"):
        return 1
    if op in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_log_f32_e32", "v_rsq_f32_e32", "v_sqrt_f32_e32"):
        return 8
    return 4


def reg_list(tok):
    r = regs(tok.split(" ")[0])
    return [] if not r else [(r[0], i) for i in range(r[1], r[2] + 1)]


def audit(name, body, meta):
    """Forward data-flow over the kernel's text with control-flow edges: the state is, per register, the modelled cycles until
    the MFMA result in it has landed (and the cycles until the matrix pipe is free); at a label the states of all incoming
    edges merge to the worst case.  Two passes so that loop back-edges reach their header."""
    errs = []
    lines = [l.split(";")[0].strip() for l in body.split("\n")]
    lines = [l for l in lines if l and not (l.startswith(".") and not l.endswith(":"))]
    # 1. spills: none inside the hand-placed loop (= the innermost backward-branch region with the most MFMAs); the persistent
    # kernels of E = 128 park a few prologue values (early-requested fragments) in scratch AROUND that loop, once per block
    spilled = {}
    for key in (".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count"):
        m = re.search(re.escape(key) + r":\s+(\d+)", meta)
        if m and int(m.group(1)) != 0:
            spilled[key] = int(m.group(1))
    if spilled:
        labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
        cands = []
        for i, l in enumerate(lines):
            m = re.match(r"s_c?branch\w* (\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                s0 = labels[m.group(1)]
                cands.append((s0, i, sum(1 for x in lines[s0:i] if x.startswith("v_mfma"))))
        inner = [c for c in cands if not any((o[0], o[1]) != (c[0], c[1]) and c[0] <= o[0] and o[1] <= c[1] and o[2] >= 16 for o in cands)]
        hot = max(inner or cands or [(0, len(lines), 0)], key=lambda c: c[2])
        in_loop = [l for l in lines[hot[0]:hot[1] + 1] if l.startswith(("scratch_", "v_writelane", "v_readlane"))]
        if in_loop or spilled.get(".private_segment_fixed_size", 0) > MAX_PROLOGUE_SCRATCH:
            errs += [f"{k} = {v}" for k, v in spilled.items()] + [f"spill code inside the hand-placed loop: {l}" for l in in_loop[:8]]
    edge = {}                                    # label -> merged state carried by branches to it

    def merge(a, b):
        out = dict(a)
        for k, v in b.items():
            out[k] = max(out.get(k, 0), v)
        return out

    seen = set()
    for final in (False, True):
        rem, reachable = {}, True                # rem["pipe"]: cycles until the matrix pipe is free
        fresh = {}                               # register -> age (instructions) of a VALU write, for ages 1, 2
        for l in lines:
            if l.endswith(":"):
                lab = l[:-1]
                inc = edge.get(lab)
                if reachable and inc is not None:
                    rem = merge(rem, inc)
                elif inc is not None:
                    rem = dict(inc)
                elif not reachable:
                    rem = {}
                reachable = True
                continue
            if not reachable:
                continue
            op, _, rest = l.partition(" ")
            args = [a.strip() for a in rest.split(",")] if rest else []

            def advance(c):
                for k in list(rem):
                    rem[k] -= c
                    if rem[k] <= 0:
                        del rem[k]

            if op.startswith("v_mfma"):
                # VALU-written register -> MFMA operand needs 2 wait states that hipcc does not know about
                for a in args[1:]:
                    for r in reg_list(a):
                        if r in fresh and final and (l, r) not in seen:
                            seen.add((l, r))
                            errs.append(f"`{l}` reads {r[0]}{r[1]} written by the VALU instruction {fresh[r]} instruction(s) before it")
                fresh = {}
                advance(rem.get("pipe", 0))      # issue blocks until the pipe is free
                rem["pipe"] = MFMA_CYCLES
                for r in reg_list(args[0]):
                    rem[r] = DONE_LAT
                advance(4)
                continue
            is_store = op.startswith(("global_store", "ds_write", "buffer_store", "scratch_store"))
            srcs = args if is_store else args[1:]
            if op.startswith(("v_", "ds_", "global_", "buffer_", "scratch_")):
                for a in srcs:
                    for r in reg_list(a):
                        if r in rem and final and (l, r) not in seen:
                            seen.add((l, r))
                            errs.append(f"`{l}` reads {r[0]}{r[1]} {rem[r]} cycles (modelled) before the MFMA writing it is done")
            if re.search(r"\bm0\b", l) and not (op == "s_mov_b32" and args and args[0] == "m0") and final:
                errs.append(f"M0 used outside the LDS-DMA statements: {l}")
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = args[0]
                edge[tgt] = merge(edge.get(tgt, {}), rem)
                if op == "s_branch":
                    reachable = False
                continue
            if op in ("s_endpgm", "s_setpc_b64"):
                reachable = False
                continue
            # age the VALU-write window (an s_nop N covers N + 1 wait states), then record this instruction's write
            age = int(args[0]) + 1 if op == "s_nop" else 1
            fresh = {r: a + age for r, a in fresh.items() if a + age <= 2}
            if op.startswith("v_") and not op.startswith("v_cmp") and args:
                for r in reg_list(args[0]):
                    fresh[r] = 1 if op != "s_nop" else fresh.get(r, 1)
            advance(cost(op, args))
    return errs


def main():
    extra = []
    if "--flags" in sys.argv:
        extra = sys.argv[sys.argv.index("--flags") + 1].split()
    bad = 0
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None       # "fwd" | "bwd"
    jobs = []
    if only in (None, "fwd"):
        jobs += [("fa_fwd_bf16.hip", "fa_fwd_w64_kernel"), ("fa_fwd_f16.hip", "fa_fwd_w64_kernel")]
    if only in (None, "bwd"):
        jobs += [("fa_bwd_bf16.hip", "fa_bwd_w64_kernel"), ("fa_bwd_f16.hip", "fa_bwd_w64_kernel")]
    for src, pat in jobs:
        text = compile_asm(src, extra)
        n = 0
        for name, body, meta in kernels(text, pat):
            n += 1
            errs = audit(name, body, meta)
            print(f"{src} {name}: {'OK' if not errs else str(len(errs)) + ' violation(s)'}")
            for e in errs[:40]:
                print("    " + e)
            bad += len(errs)
        if n == 0:
            print(f"{src}: no {pat} instantiation found")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
