#!/usr/bin/env python3
"""Generates nnop.jl_amd/csrc/fa_fwd_duo_asm.inc: the phase loop of the two-waves-per-SIMD forward (csrc/fa_fwd_duo.hpp) as ONE
inline-asm statement per mode with FIXED physical registers.

Why generated text: at 256 registers per wave the kernel's long-lived tiles (O 64, Q 32, S / P 64 registers) leave hipcc's allocator
no slack -- given virtual 16-register tuples it moves whole tiles between phases and spills the Q fragments (228-660 bytes of scratch
per lane in every C++ form tried, with or without physical-register constraints on per-phase statements: a value that stays live
behind the copy into its constraint register interferes with that register's own fixed range).  With the whole loop in one statement
hipcc copies each tile into its home register once, before the loop, and allocates nothing inside it.

Register map
  v[0:63]    O^T accumulators  o[z][eb]   = 16 (2 z + eb)
  v[64:95]   Q fragments       q[z][ks]   = 64 + 16 z + 4 ks
  v[96:103]  row sums          l[z]       = 96 + 4 z           (registers 0 / 1 of lanes 0..15 = queries lane, lane + 16)
  v[104:107] row-sum selector
  v[108:110] K fragment addresses for ks = 1..3 (matrix phase)
  v[112:175] score tile / P words  s[z][kb] = 112 + 32 z + 16 kb
  v[176:191] fragment ring (matrix phase) / row-max chains (vector phase)
  v[192:195] m2[0..1] (exponent reference), mt[0..1] (true row max), log2 units
  v[196:203] k_voff0, k_voff1, v_voff (DMA source offsets), k_lane, v_lane (fragment read bases), qlim0, qlim1 (causal limit), 4 h
  v[204:223] temporaries
  v[224:239] the scalar state on entry (copied to s[32:44] by the statement's first instructions), v[240:247] the two buffer descriptors
  s[32:44]   t, H, n_live, kX, kY, kZ, vX, vY, vZ (ring slots of tiles t, t+1, t+2 mod 3), last_off, c2, causal_q0, vbits (LDS)
  s[48:51]   K descriptor      s[52:55] V descriptor      s[56:67] temporaries

usage: gen_duo_asm.py [--check]     (--check: exit 1 if the committed file differs)
"""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "nnop.jl_amd", "csrc", "fa_fwd_duo_asm.inc")

O = lambda z, eb: 16 * (2 * z + eb)
Q = lambda z, ks: 64 + 16 * z + 4 * ks
L = lambda z: 96 + 4 * z
SEL = 104
KA = lambda ks: 107 + ks            # ks = 1..3 -> v108..v110
S = lambda z, kb: 112 + 32 * z + 16 * kb
FR = lambda i: 176 + 4 * i
TMP = 176
M2 = lambda z: 192 + z
MT = lambda z: 194 + z
KVO = lambda j: 196 + j
VVO, KLANE, VLANE = 198, 199, 200
QLIM = lambda z: 201 + z
H4 = 203
KIMG, VIMG = 204, 205
MX = lambda z: 206 + z
NM = lambda z: 208 + z
T0 = 210                             # v210..v223: temporaries
# scalars
ST, SH, SNLIVE, SKX, SKY, SKZ, SVX, SVY, SVZ, SLAST, SC2, SCQ0, SVB = range(32, 45)
KRS, VRS = "s[48:51]", "s[52:55]"
SQK, SPV, SA, SB, SKOFF, SVOFF = 56, 57, 58, 59, 60, 61
SVAL = "s[62:63]"
SM = "s[64:65]"

PF = 3                               # fragments read ahead
KS, KB, EB, NKF, NVF = 4, 2, 2, 8, 8
NJK, NJV = 2, 2
TILE_SHIFT = 13                      # one 64-key tile of E = 64 16-bit elements = 8 KiB


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def sr(i):
    return f"s{i}"


def m_phase(qk, pv):
    """matrix phase: [row sums of P(t-2)] [O += V(t-2)^T P(t-2)^T] [S(t) = K(t) Q^T]; the DMA of K(t+2), V(t) in the gaps"""
    out = []
    stream = ([("V", g) for g in range(NVF)] if pv else []) + ([("K", f) for f in range(NKF)] if qk else [])
    lds_issued = 0
    last_read = {}                   # stream position -> index of its last LDS instruction

    def dma(d):
        if d < NJK:
            if d == 0:
                out.append(f"s_mov_b32 m0, {sr(SKZ)}")
                out.append("s_nop 0")
            out.append(f"buffer_load_dwordx4 {vr(KVO(d))}, {KRS}, {sr(SKOFF)} offen offset:{d * 1024} lds")
        else:
            j = d - NJK
            if j == 0:
                out.append(f"s_mov_b32 m0, {sr(SVX)}")
                out.append("s_nop 0")
            out.append(f"buffer_load_dwordx4 {vr(VVO)}, {VRS}, {sr(SVOFF)} offen offset:{j * 1024} lds")

    def reads(p):
        nonlocal lds_issued
        if p >= len(stream):
            return
        kind, idx = stream[p]
        slot = FR(p % 4)
        if kind == "V":
            kk, eb = divmod(idx, EB)
            out.append(f"ds_read_b64_tr_b16 {vr(slot, 2)}, {vr(VIMG)} offset:{(8 * kk + eb) * 256}")
            out.append(f"ds_read_b64_tr_b16 {vr(slot + 2, 2)}, {vr(VIMG)} offset:{(8 * kk + 4 + eb) * 256}")
            lds_issued += 2
        else:
            kb, ks = divmod(idx, KS)
            addr = vr(KIMG) if ks == 0 else vr(KA(ks))
            out.append(f"ds_read_b128 {vr(slot, 4)}, {addr} offset:{kb * 32 * 128}")
            lds_issued += 1
        last_read[p] = lds_issued - 1

    if not stream:
        for d in range(NJK + NJV):
            dma(d)
        return out
    if qk:
        for ks in range(1, KS):
            out.append(f"v_xor_b32 {vr(KA(ks))}, {ks << 5}, {vr(KIMG)}")
    for p in range(PF):
        reads(p)
    if not qk:                       # no QK^T part to spread the DMA over: issue it up front
        for d in range(NJK + NJV):
            dma(d)
    if pv:
        # row sums first: register operands only -- they cover the latency of the first fragment reads
        for kk in range(2 * KB):
            for z in range(2):
                p0 = S(z, kk >> 1) + 8 * (kk & 1)
                out.append(f"v_mfma_f32_16x16x32_@T@ {vr(L(z), 4)}, {vr(SEL, 4)}, {vr(p0, 4)}, {vr(L(z), 4)}")
    for p, (kind, idx) in enumerate(stream):
        reads(p + PF)
        out.append(f"s_waitcnt lgkmcnt({lds_issued - 1 - last_read[p]})")
        slot = FR(p % 4)
        if kind == "V":
            kk, eb = divmod(idx, EB)
            for z in range(2):
                p0 = S(z, kk >> 1) + 8 * (kk & 1)
                out.append(f"v_mfma_f32_32x32x16_@T@ {vr(O(z, eb), 16)}, {vr(slot, 4)}, {vr(p0, 4)}, {vr(O(z, eb), 16)}")
        else:
            kb, ks = divmod(idx, KS)
            for z in range(2):
                c = "0" if ks == 0 else vr(S(z, kb), 16)
                out.append(f"v_mfma_f32_32x32x16_@T@ {vr(S(z, kb), 16)}, {vr(slot, 4)}, {vr(Q(z, ks), 4)}, {c}")
        if qk and (p & 1) == 1 and p // 2 < NJK + NJV:
            dma(p // 2)
    if qk:
        # the vector phase reads S right behind the barrier: idle out the last MFMAs (the final pass of an MFMA issued behind a busy
        # pipe lands up to 64 cycles after its issue)
        out += ["s_nop 15"] * 4
    return out


def v_rowmax():
    """row max of the raw score tile per query block z: 4 chains of v_max3 per z, folded into MX(z)"""
    out = []
    t = lambda z, c: TMP + 4 * z + c
    for q in range(16):
        kb, i0 = q >> 3, 2 * (q & 7)
        for z in range(2):
            a, b = S(z, kb) + i0, S(z, kb) + i0 + 1
            if q < 4:
                out.append(f"v_max_f32 {vr(t(z, q))}, {vr(a)}, {vr(b)}")
            else:
                out.append(f"v_max3_f32 {vr(t(z, q & 3))}, {vr(t(z, q & 3))}, {vr(a)}, {vr(b)}")
    for z in range(2):
        out.append(f"v_max3_f32 {vr(t(z, 0))}, {vr(t(z, 0))}, {vr(t(z, 1))}, {vr(t(z, 2))}")
    for z in range(2):
        out.append(f"v_max_f32 {vr(MX(z))}, {vr(t(z, 0))}, {vr(t(z, 3))}")
    return out


def v_softmax(l1=4, l2=4):
    """P = exp2(s c2 + nm[z]) in place, then packed in place: the 8 logits of a chunk (16-key step kk, query block z) become 4 words
    in the chunk's first 4 registers.  Software pipeline: fma (n), exp (n - l1), convert of the pair that ends at n - l1 - l2
    (a transcendental's result needs one wait state before a non-transcendental reader: l2 >= 1 gives it)."""
    out = []

    def reg(n):
        c, j = n >> 3, n & 7
        kk, z = c >> 1, c & 1
        return S(z, kk >> 1) + 8 * (kk & 1) + j, z

    for step in range(64 + l1 + l2):
        if step < 64:
            r, z = reg(step)
            out.append(f"v_fma_f32 {vr(r)}, {vr(r)}, {sr(SC2)}, {vr(NM(z))}")
        e = step - l1
        if 0 <= e < 64:
            r, _ = reg(e)
            out.append(f"v_exp_f32 {vr(r)}, {vr(r)}")
        m = step - l1 - l2
        if 0 <= m < 64 and (m & 1):
            r, _ = reg(m)
            dst = r - (m & 7) + ((m & 7) >> 1)
            out.append(f"v_cvt_pk_@T@_f32 {vr(dst)}, {vr(r - 1)}, {vr(r)}")
    return out


def v_mask():
    """masked mode: fetch the validity word of tile t; where the tile needs it (padding / ragged end / causal diagonal) set the
    hidden logits to -inf.  Per (z, kb) ONE 32-bit lane mask: validity bits of the lane's key rows AND the causal prefix."""
    out = []
    out += [f"s_lshl_b32 {sr(SA)}, {sr(ST)}, 3", f"s_add_u32 {sr(SA)}, {sr(SA)}, {sr(SVB)}", f"v_mov_b32 {vr(T0)}, {sr(SA)}",
            f"ds_read_b64 {vr(T0 + 2, 2)}, {vr(T0)}", "s_waitcnt lgkmcnt(0)",
            f"v_readfirstlane_b32 s62, {vr(T0 + 2)}", f"v_readfirstlane_b32 s63, {vr(T0 + 3)}",
            f"s_cmp_lg_u64 {SVAL}, -1", "s_cbranch_scc1 L_domask_%=",
            f"s_lshl_b32 {sr(SA)}, {sr(ST)}, 6", f"s_add_i32 {sr(SA)}, {sr(SA)}, 63", f"s_cmp_gt_i32 {sr(SA)}, {sr(SCQ0)}",
            "s_cbranch_scc0 L_maskdone_%=", "L_domask_%=:"]
    lim, cm, sh, w = T0 + 4, T0 + 5, T0 + 6, T0 + 8        # w: 2 registers
    for z in range(2):
        for kb in range(KB):
            # lim = qlim[z] - (64 t + 32 kb) - 4 h;  cm = lim < 0 ? 0 : 0xffffffff >> (31 - min(lim, 31))
            out += [f"s_lshl_b32 {sr(SA)}, {sr(ST)}, 6", f"s_add_i32 {sr(SA)}, {sr(SA)}, {32 * kb}",
                    f"v_subrev_u32 {vr(lim)}, {sr(SA)}, {vr(QLIM(z))}", f"v_sub_u32 {vr(lim)}, {vr(lim)}, {vr(H4)}",
                    f"v_min_i32 {vr(cm)}, 31, {vr(lim)}", f"v_sub_u32 {vr(cm)}, 31, {vr(cm)}", f"v_lshrrev_b32 {vr(cm)}, {vr(cm)}, -1",
                    f"v_cmp_gt_i32 vcc, 0, {vr(lim)}", f"v_cndmask_b32 {vr(cm)}, {vr(cm)}, 0, vcc",
                    # m = (valid >> (32 kb + 4 h)) & cm
                    f"v_add_u32 {vr(sh)}, {32 * kb}, {vr(H4)}", f"v_lshrrev_b64 {vr(w, 2)}, {vr(sh)}, {SVAL}",
                    f"v_and_b32 {vr(cm)}, {vr(cm)}, {vr(w)}"]
            for i in range(16):
                lr = (i & 3) + 8 * (i >> 2)
                r = S(z, kb) + i
                out += [f"v_and_b32 {vr(sh)}, {hex(1 << lr)}, {vr(cm)}", f"v_cmp_ne_u32 vcc, 0, {vr(sh)}",
                        f"v_cndmask_b32 {vr(r)}, {vr(T0 + 10)}, {vr(r)}, vcc"]
    out.insert(out.index("L_domask_%=:") + 1, f"v_mov_b32 {vr(T0 + 10)}, 0xff800000")
    out.append("L_maskdone_%=:")
    return out


def rescale():
    """rare: a row max outgrew its reference by more than 2^8 (or the row sees its first key): raise the reference, scale what was
    accumulated at the old one (O, l) exactly once.  Before the first tile O and l are zero: scaled by 0, harmless."""
    out = ["s_nop 15"] * 8           # the PV MFMAs of the last matrix phase have written O (fa_fwd_w64.hpp: 128 idle cycles)
    a0, a1, lane = T0, T0 + 1, T0 + 2
    out += [f"v_mbcnt_lo_u32_b32 {vr(lane)}, -1, 0", f"v_mbcnt_hi_u32_b32 {vr(lane)}, -1, {vr(lane)}", f"v_and_b32 {vr(a0)}, 15, {vr(lane)}",
            f"v_lshlrev_b32 {vr(a0)}, 2, {vr(a0)}", f"v_add_u32 {vr(a1)}, 64, {vr(a0)}"]
    thr, mn, al, b0, b1 = T0 + 3, T0 + 4, T0 + 5, T0 + 6, T0 + 7
    for z in range(2):
        out += [f"v_add_f32 {vr(thr)}, 0x41000000, {vr(M2(z))}", f"v_cmp_gt_f32 vcc, {vr(MX(z))}, {vr(thr)}",
                f"v_cndmask_b32 {vr(mn)}, {vr(M2(z))}, {vr(MX(z))}, vcc",
                f"v_sub_f32 {vr(al)}, {vr(M2(z))}, {vr(mn)}", f"v_exp_f32 {vr(al)}, {vr(al)}", "s_nop 0",
                f"v_cndmask_b32 {vr(al)}, 1.0, {vr(al)}, vcc",          # not raised: factor one (also when m2 = mn = -inf)
                f"v_mov_b32 {vr(M2(z))}, {vr(mn)}"]
        for eb in range(EB):
            for i in range(16):
                out.append(f"v_mul_f32 {vr(O(z, eb) + i)}, {vr(O(z, eb) + i)}, {vr(al)}")
        # the sums of queries n and n + 16 sit in registers 0 / 1 of lanes 0..15: fetch their factors from those queries' lanes
        out += [f"ds_bpermute_b32 {vr(b0)}, {vr(a0)}, {vr(al)}", f"ds_bpermute_b32 {vr(b1)}, {vr(a1)}, {vr(al)}", "s_waitcnt lgkmcnt(0)",
                f"v_mul_f32 {vr(L(z))}, {vr(L(z))}, {vr(b0)}", f"v_mul_f32 {vr(L(z) + 1)}, {vr(L(z) + 1)}, {vr(b1)}"]
    return out


def loop(masked):
    """half-steps h = 0 .. n_tiles + 1, one barrier each; group g runs M(t) at h = t for t = g (mod 2) and V(t) at h = t + 1; PV(t)
    happens in M(t + 2).  (Group 1's idle half-step 0 is a barrier in front of this statement.)"""
    # the scalar state arrives in vector registers (a 16-register SGPR tuple as an asm operand does not survive hipcc's copy
    # legalisation: "illegal VGPR to SGPR copy"): v[224:236] -> s[32:44], v[240:247] -> the two buffer descriptors s[48:55]
    out = [f"v_readfirstlane_b32 s{32 + i}, v{224 + i}" for i in range(13)]
    out += [f"v_readfirstlane_b32 s{48 + i}, v{240 + i}" for i in range(8)]
    out += ["L_loop_%=:"]
    # qk = t < n_live;  pv = t >= 2 && t - 2 < n_live
    out += [f"s_cmp_lt_i32 {sr(ST)}, {sr(SNLIVE)}", f"s_cselect_b32 {sr(SQK)}, 1, 0",
            f"s_sub_i32 {sr(SA)}, {sr(ST)}, 2", f"s_cmp_lt_i32 {sr(SA)}, {sr(SNLIVE)}", f"s_cselect_b32 {sr(SPV)}, 1, 0",
            f"s_cmp_lt_i32 {sr(SA)}, 0", f"s_cselect_b32 {sr(SPV)}, 0, {sr(SPV)}"]
    # fragment read bases: K(t) in slot kX, V(t-2) in slot (t + 1) % 3 = vY;  DMA sources: K(t+2), V(t), clamped to the last tile
    out += [f"v_add_u32 {vr(KIMG)}, {sr(SKX)}, {vr(KLANE)}", f"v_add_u32 {vr(VIMG)}, {sr(SVY)}, {vr(VLANE)}",
            f"s_add_i32 {sr(SA)}, {sr(ST)}, 2", f"s_lshl_b32 {sr(SA)}, {sr(SA)}, {TILE_SHIFT}", f"s_min_u32 {sr(SKOFF)}, {sr(SA)}, {sr(SLAST)}",
            f"s_lshl_b32 {sr(SA)}, {sr(ST)}, {TILE_SHIFT}", f"s_min_u32 {sr(SVOFF)}, {sr(SA)}, {sr(SLAST)}"]
    out += [f"s_cmp_eq_u32 {sr(SQK)}, 0", "s_cbranch_scc1 L_noqk_%=", f"s_cmp_eq_u32 {sr(SPV)}, 0", "s_cbranch_scc1 L_mqk_%="]
    out += m_phase(True, True) + ["s_branch L_mdone_%=", "L_mqk_%=:"] + m_phase(True, False) + ["s_branch L_mdone_%=", "L_noqk_%=:"]
    out += [f"s_cmp_eq_u32 {sr(SPV)}, 0", "s_cbranch_scc1 L_mnone_%="] + m_phase(False, True) + ["s_branch L_mdone_%=", "L_mnone_%=:"]
    out += m_phase(False, False) + ["L_mdone_%=:", "s_barrier"]
    out += [f"s_add_i32 {sr(SA)}, {sr(ST)}, 1", f"s_cmp_ge_i32 {sr(SA)}, {sr(SH)}", "s_cbranch_scc1 L_exit_%=",
            f"s_cmp_eq_u32 {sr(SQK)}, 0", "s_cbranch_scc1 L_vdone_%="]
    # ---- vector phase
    if masked:
        out += v_mask()
    out += v_rowmax()
    for z in range(2):               # both lane halves (lane l <-> l ^ 32), log2 units
        out += [f"v_mov_b32 {vr(T0 + z)}, {vr(MX(z))}"]
    out += ["s_nop 1"]               # VALU write -> v_permlane32_swap read
    for z in range(2):
        out += [f"v_permlane32_swap_b32 {vr(MX(z))}, {vr(T0 + z)}"]
    for z in range(2):
        out += [f"v_max_f32 {vr(MX(z))}, {vr(MX(z))}, {vr(T0 + z)}", f"v_mul_f32 {vr(MX(z))}, {sr(SC2)}, {vr(MX(z))}",
                f"v_max_f32 {vr(MT(z))}, {vr(MT(z))}, {vr(MX(z))}", f"v_add_f32 {vr(T0 + 2 + z)}, 0x41000000, {vr(M2(z))}"]
    out += [f"v_cmp_gt_f32 vcc, {vr(MX(0))}, {vr(T0 + 2)}", f"s_mov_b64 {SM}, vcc", f"v_cmp_gt_f32 vcc, {vr(MX(1))}, {vr(T0 + 3)}",
            f"s_or_b64 {SM}, {SM}, vcc", f"s_cmp_lg_u64 {SM}, 0", "s_cbranch_scc1 L_rescale_%=", "L_rescdone_%=:"]
    for z in range(2):               # nm = -m2, or 0 while the row has seen no key (m2 = -inf): P = exp2(-inf) = 0
        out += [f"v_cmp_eq_f32 vcc, 0xff800000, {vr(M2(z))}", f"v_sub_f32 {vr(NM(z))}, 0, {vr(M2(z))}",
                f"v_cndmask_b32 {vr(NM(z))}, {vr(NM(z))}, 0, vcc"]
    out += v_softmax()
    out += ["L_vdone_%=:", "s_waitcnt vmcnt(0)", "s_barrier"]
    # rotate the ring slots by two tiles: (X, Y, Z) <- (Z, X, Y)
    for x, y, z in ((SKX, SKY, SKZ), (SVX, SVY, SVZ)):
        out += [f"s_mov_b32 {sr(SA)}, {sr(x)}", f"s_mov_b32 {sr(x)}, {sr(z)}", f"s_mov_b32 {sr(z)}, {sr(y)}", f"s_mov_b32 {sr(y)}, {sr(SA)}"]
    out += [f"s_add_i32 {sr(ST)}, {sr(ST)}, 2", f"s_cmp_lt_i32 {sr(ST)}, {sr(SH)}", "s_cbranch_scc1 L_loop_%=", "s_branch L_exit_%=",
            "L_rescale_%=:"] + rescale() + ["s_branch L_rescdone_%=", "L_exit_%=:"]
    return out


def as_macro(name, lines):
    body = " \\\n".join(f'    "{ln}\\n\\t"' for ln in lines)
    return f"#define {name}(TS) \\\n{body}\n".replace("@T@", '" TS "')


def render():
    parts = ["// GENERATED by tools/gen_duo_asm.py -- do not edit; tests/test_duo_codegen.py checks that it is up to date.\n"
             "// The phase loop of fa_fwd_duo.hpp with fixed physical registers (register map: the generator's header).\n"
             "// TS: the element type's mnemonic suffix (\"bf16\" / \"f16\").\n"]
    parts.append(as_macro("NNOP_DUO_LOOP_PLAIN", loop(False)))
    parts.append(as_macro("NNOP_DUO_LOOP_MASKED", loop(True)))
    return "\n".join(parts)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}: {text.count(chr(10))} lines")
