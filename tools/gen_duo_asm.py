#!/usr/bin/env python3
"""Generates nnop.jl_amd/csrc/fa_fwd_duo_asm.inc: the phase loop of the two-waves-per-SIMD forward (csrc/fa_fwd_duo.hpp) as ONE
inline-asm statement per mode with FIXED physical registers.

Why generated text: at 256 registers per wave the kernel's long-lived tiles (O 64, Q 32, S 64, P 32 registers) leave hipcc's allocator
no slack -- given virtual 16-register tuples it moves whole tiles between phases and spills the Q fragments (228-660 bytes of scratch
per lane in every C++ form tried, with or without physical-register constraints on per-phase statements: a value that stays live
behind the copy into its constraint register interferes with that register's own fixed range).  With the whole loop in one statement
hipcc copies each tile into its home register once, before the loop, and allocates nothing inside it.

Register map
  v[0:63]    O^T accumulators  o[z][eb]   = 16 (2 z + eb)
  v[64:95]   Q fragments       q[z][ks]   = 64 + 16 z + 4 ks
  v[96:103]  row sums          l[z]       = 96 + 4 z           (registers 0 / 1 of lanes 0..15 = queries lane, lane + 16)
  v[104:107] row-sum selector
  v[108:110] K fragment addresses for ks = 1..3
  v[112:175] score tile S(t)   s[z][kb]   = 112 + 32 z + 16 kb; V(t) packs P(t)^T IN PLACE: the 8 logits of 16-key step kk (registers
             8 (kk & 1) .. + 7 of s[z][kk >> 1]) become 4 operand words in the first 4 of those registers
  v[176:191] fragment ring (matrix phase) / row-max chains (vector phase)
  v[192:195] m2[0..1] (exponent reference), mt[0..1] (true row max of this lane's HALF of the keys: the two lane halves are combined once,
             behind the loop), log2 units;  v[208:211] -m2 and m2 + 8, kept beside them
  v[196:203] k_voff0, k_voff1, v_voff (DMA source offsets), k_lane, v_lane (fragment read bases), qlim0, qlim1 (causal limit), 4 h
  v[204:207], v[212:223] temporaries
  v[224:255] hipcc's (values that live across the loop: the epilogue's addresses); profile builds return five counters in v[224:228]
  scalar operands (allocated by hipcc): t, H, n_live, kA, kB, kC (K ring slots of K(t), K(t+2), the free one), vA, vB, vC (V ring slots
             of V(t-2), V(t), free), last_off, c2, causal_q0, vbits (LDS address of the validity words), the two buffer descriptors
  s[56:67]   temporaries      s[68:76] profile builds

The loop (per wave; group g = wave / 4 owns the kv tiles t = g mod 2; rings of 4 slots, tile t in slot t % 4):
  M(t)  matrix phase   row sums of P(t-2) (8 MFMAs 16x16x32: register operands, they cover the latency of the first fragment reads) |
                       O += V(t-2)^T P(t-2)^T (16 MFMAs) | S(t) = K(t) Q^T (16 MFMAs; S takes the registers of the P words that die
                       with the last PV MFMA).  Nothing but MFMAs and fragment reads: a VALU or LDS-DMA instruction between them
                       stalls the wave's in-order issue and the matrix pipe idles (measured, profiles/r04/duo_ablations.log).
        s_barrier
  V(t)  vector phase   LDS-DMA of K(t+4), V(t+2) into the slots M(t) has just read (behind the barrier: every wave of the group is done
                       with them), spread over the phase | [masked mode, rare: mask S(t)] | row max, test | [rare: raise the reference,
                       rescale O and l] | P(t) = exp2(S(t) c2 - m2) -> operand words
        s_waitcnt vmcnt(4) (the batch issued one iteration ago -- K(t+2), V(t) -- has landed) ; s_barrier
Group 1 runs one phase behind group 0 (a barrier in front of the statement), so each SIMD always holds one wave in M and one in V.

usage: gen_duo_asm.py [--check]     (--check: exit 1 if the committed file differs)
"""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "nnop.jl_amd", "csrc", "fa_fwd_duo_asm.inc")

O = lambda z, eb: 16 * (2 * z + eb)
Q = lambda z, ks: 64 + 16 * z + 4 * ks
L = lambda z: 96 + 4 * z
SEL = 104
KA = lambda ks: 107 + ks if ks < 4 else 96 + ks      # ks = 1..3 -> v108..v110; E = 128: ks = 4..7 -> v100..v103
S = lambda z, kb: 112 + 32 * z + 16 * kb
FR = lambda i: (176 + 4 * i if i < 4 else 224 + 4 * (i - 4)) if NZ == 2 else 144 + 4 * i      # NZ = 1: v[144:175], the z = 1 score tile's
M2 = lambda z: 192 + z
MT = lambda z: 194 + z
KVO = lambda j: 196 + j if E != 128 else (196 if j == 0 else 179 + j)     # E = 128: pieces 1..3 in v[180:182], derived in front of the loop
VVO, KLANE, VLANE = 198, 199, 200
QLIM = lambda z: 201 + z
H4 = 203
KIMG, VIMG = 204, 205
MX = lambda z: 206 + z
NM = lambda z: 208 + z               # -m2, or 0 while m2 = -inf            } kept beside m2: they change only when a reference rises
THR = lambda z: 210 + z              # m2 + 8: the rescale threshold        }
T0 = 212                             # v212..v223: temporaries
PW = lambda kk, z: S(z, kk >> 1) + 8 * (kk & 1)         # P^T operand words of 16-key step kk: in place of the logits
# scalars: operands of the statement, allocated by hipcc (names in fa_fwd_duo.hpp's operand list)
ST, SH, SNLIVE, SKA, SKB, SKC, SVA, SVB, SVC, SLAST, SC2, SCQ0, SVBITS = ("%[st]", "%[sh]", "%[snlive]", "%[ska]", "%[skb]", "%[skc]", "%[sva]",
                                                                          "%[svb]", "%[svc]", "%[slast]", "%[sc2]", "%[scq0]", "%[svbits]")
KRS, VRS = "%[krs]", "%[vrs]"
SQK, SPV, SA, SB, SKOFF, SVOFF = 56, 57, 58, 59, 60, 61
SVAL = "s[62:63]"
SM = "s[64:65]"
SNEED = 77

E = 64
KS, KB, EB, NKF, NVF = 4, 2, 2, 8, 8
NJK, NJV = 2, 2
TILE_SHIFT = 13                      # one 64-key tile of E = 64 16-bit elements = 8 KiB
ROWB = 128                           # bytes of a K row in its LDS image


# where the LDS-DMA batch K(t+4), V(t+2) is issued: "mhead" / "mtail" / "msplit" = first / last thing (half and half) of the matrix phase M(t), into the group's
# FREE ring slots (3 slots per group and ring: what the free one held was read in M(t-2), two barriers back); "v" = spread over the
# vector phase V(t), into the slots M(t) has just read (2 slots per group and ring).  Measured (profiles/r04/duo_ablations.log): the
# vector phase is the longer one, and an LDS-DMA instruction stalls the issuing wave ~60 cycles WITHOUT using the vector port.
DMA_AT = os.environ.get("NNOP_DUO_GEN_DMA", "mtail")
SLOTS = 2 if DMA_AT == "v" else 3                        # ("mmid" / "mmidq": between the MFMA pairs of the matrix phase)
_DMA64, _SLOTS64 = DMA_AT, SLOTS

# row sums of P: "mfma" = 8 v_mfma_f32_16x16x32 per tile in the matrix phase (selector operand), "valu" = 64 v_add_f32 per tile in the
# vector phase (4 chains in v[248:251]).  Measured (profiles/r04/duo_sums.log): see DESIGN.md section 4.1d.
SUMS = os.environ.get("NNOP_DUO_GEN_SUMS", "mfma")
LS = lambda z, c: 248 + 2 * z + c
SYNC = os.environ.get("NNOP_DUO_GEN_SYNC", "one")        # barriers per iteration: "two" (behind every phase) / "one" (group_loop)
_SYNC64 = SYNC
RING = int(os.environ.get("NNOP_DUO_GEN_RING", "4"))      # fragment ring slots (4 registers each): v[176:191] (+ v[224:...] beyond 4)
PF = int(os.environ.get("NNOP_DUO_GEN_PF", "3"))          # fragments read ahead (< RING)
# query blocks of 32 rows per wave.  2: the form described above (64 rows per wave, 256 per workgroup).  1: the SAME loop with every z = 1
# register and instruction left out -- 32 rows per wave, 128 per workgroup: twice the workgroups for problems that cannot fill the chip
# with 256-row blocks (every fragment then feeds ONE MFMA, so the fragment ring is deeper: it takes the registers of the z = 1 score tile).
NZ = 2
_RING2, _PF2 = RING, PF


def set_nz(nz, e=64):
    """the loop's shape: nz 32-row query blocks per wave at embedding dim e.  E = 128 (nz = 1 only: O^T alone is 64 registers): tiles of
    16 KiB, so 2 ring slots per key group and ring (128 KiB of LDS) -- the DMA batch then goes behind the barrier that closes the matrix
    phase whose slots it overwrites, i.e. into the vector phase, with a barrier behind EVERY phase (the matrix phase is the longer one
    there: 36 MFMAs against 32 logits per lane, so the vector-phase wave has the idle issue slots the 8 LDS-DMA pieces cost)."""
    global NZ, RING, PF, E, KS, EB, NKF, NVF, NJK, NJV, TILE_SHIFT, ROWB, DMA_AT, SLOTS, SYNC
    assert e == 64 or (e == 128 and nz == 1) or (e == 32 and nz == 2)
    NZ, E = nz, e
    RING, PF = (_RING2, _PF2) if nz == 2 else (8, 6)
    KS, EB = e // 16, e // 32
    NKF, NVF = KB * KS, 4 * EB
    NJK = NJV = max(1, e // 32)
    TILE_SHIFT, ROWB = {32: (12, 64), 64: (13, 128), 128: (14, 256)}[e]
    DMA_AT, SLOTS, SYNC = (_DMA64, _SLOTS64, _SYNC64) if e != 128 else ("v", 2, "two")

# the scale-and-shift of the logits (s c2 - m2) as v_pk_fma_f32 over register pairs (one instruction per two logits; the scale in the scalar
# pair s[66:67], -m2 broadcast from one register of v[208:209] by op_sel) instead of one v_fma_f32 per logit.  Measured
# (profiles/r04/duo_pkfma.log): 10 % SLOWER (C2 67.5 -> 74.4 us) -- the packed form holds the vector port longer than the two
# instructions it replaces; results identical.  Off.
PKFMA = int(os.environ.get("NNOP_DUO_GEN_PKFMA", "0"))

# timing-only ablations (results WRONG by construction; never committed): NNOP_DUO_GEN_ABL bit mask
#   1 no LDS-DMA in the loop   2 no row-max fillers   4 no exp / fma / convert in the vector phase   8 no MFMAs and no fragment reads
ABL = int(os.environ.get("NNOP_DUO_GEN_ABL", "0"))


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def sr(i):
    return i if isinstance(i, str) else f"s{i}"


def v_rowmax():
    """row max of the raw score tile per query block (4 chains of v_max3 per z) over THIS LANE's 32 keys, log2 units; the running
    maximum mt; SM (and SCC) = some lane's maximum exceeds its reference by more than 2^8 (or its row sees its first key).  The other
    lane half (lane l ^ 32 holds the same query's other 32 keys) is not consulted here: "some lane's half maximum exceeds" is the same
    wave-wide condition as "some lane's row maximum exceeds", and the rare path that acts on it combines the halves itself."""
    t = lambda z, c: 176 + 4 * z + c
    out = []
    for q in range(16):
        kb, i0 = q >> 3, 2 * (q & 7)
        for z in range(NZ):
            a, b = S(z, kb) + i0, S(z, kb) + i0 + 1
            if q < 4:
                out.append(f"v_max_f32 {vr(t(z, q))}, {vr(a)}, {vr(b)}")
            else:
                out.append(f"v_max3_f32 {vr(t(z, q & 3))}, {vr(t(z, q & 3))}, {vr(a)}, {vr(b)}")
    out += [f"v_max3_f32 {vr(t(z, 0))}, {vr(t(z, 0))}, {vr(t(z, 1))}, {vr(t(z, 2))}" for z in range(NZ)]
    out += [f"v_max_f32 {vr(MX(z))}, {vr(t(z, 0))}, {vr(t(z, 3))}" for z in range(NZ)]
    out += [f"v_mul_f32 {vr(MX(z))}, {sr(SC2)}, {vr(MX(z))}" for z in range(NZ)]
    out += [f"v_max_f32 {vr(MT(z))}, {vr(MT(z))}, {vr(MX(z))}" for z in range(NZ)]
    if NZ == 2:
        out += [f"v_cmp_gt_f32 vcc, {vr(MX(0))}, {vr(THR(0))}", f"v_cmp_gt_f32 {SM}, {vr(MX(1))}, {vr(THR(1))}", f"s_or_b64 {SM}, {SM}, vcc"]
    else:
        out += [f"v_cmp_gt_f32 {SM}, {vr(MX(0))}, {vr(THR(0))}", f"s_or_b64 {SM}, {SM}, {SM}"]       # (s_or_b64 sets SCC = result != 0)
    return out


def v_mask(g=0):
    """masked mode: does tile t need its mask (validity word != all ones, or the tile reaches past the wave's first query)?  Rare
    (padding / ragged end / causal diagonal): set the hidden logits of S(t) to -inf.  Per (z, kb) ONE 32-bit lane mask: validity bits
    of the lane's key rows AND the causal prefix.  (The validity word was fetched by the matrix phase into v[212:213].)"""
    out = [f"v_readfirstlane_b32 s62, {vr(T0 + 2)}", f"v_readfirstlane_b32 s63, {vr(T0 + 3)}",
           f"s_cmp_lg_u64 {SVAL}, -1", f"s_cbranch_scc1 L_domask{g}_%=",
           f"s_lshl_b32 {sr(SA)}, {sr(ST)}, 6", f"s_add_i32 {sr(SA)}, {sr(SA)}, 63", f"s_cmp_gt_i32 {sr(SA)}, {sr(SCQ0)}",
           f"s_cbranch_scc0 L_maskdone{g}_%=", f"L_domask{g}_%=:", f"v_mov_b32 {vr(T0 + 10)}, 0xff800000"]
    lim, cm, sh, w = T0 + 4, T0 + 5, T0 + 6, T0 + 8        # w: 2 registers
    for z in range(NZ):
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
    out.append(f"L_maskdone{g}_%=:")
    return out


def dma_piece(d):
    """piece d of the batch K(t+4), V(t+2)"""
    if ABL & 1:
        return []
    out = []
    kdst, vdst = (SKA, SVA) if SLOTS == 2 else (SKC, SVC)
    if d < NJK:
        if d == 0:
            out += [f"s_mov_b32 m0, {sr(kdst)}", "s_nop 0"]
        out.append(f"buffer_load_dwordx4 {vr(KVO(d))}, {KRS}, {sr(SKOFF)} offen offset:{d * 1024} lds")
    else:
        j = d - NJK
        if j == 0:
            out += [f"s_mov_b32 m0, {sr(vdst)}", "s_nop 0"]
        out.append(f"buffer_load_dwordx4 {vr(VVO)}, {VRS}, {sr(SVOFF)} offen offset:{j * 1024} lds")
    return out


def m_phase(qk, pv, masked):
    """matrix phase M(t); qk: tile t exists for this wave, pv: tile t - 2 did"""
    out = []
    stream = ([("V", g) for g in range(NVF)] if pv else []) + ([("K", f) for f in range(NKF)] if qk else [])
    lds_issued = 0
    last_read = {}                   # stream position -> index of its last LDS instruction

    def reads(p):
        nonlocal lds_issued
        if p >= len(stream):
            return
        kind, idx = stream[p]
        slot = FR(p % RING)
        if kind == "V":
            kk, eb = divmod(idx, EB)
            out.append(f"ds_read_b64_tr_b16 {vr(slot, 2)}, {vr(VIMG)} offset:{(4 * kk * EB + eb) * 256}")
            out.append(f"ds_read_b64_tr_b16 {vr(slot + 2, 2)}, {vr(VIMG)} offset:{((4 * kk + 2) * EB + eb) * 256}")
            lds_issued += 2
        else:
            kb, ks = divmod(idx, KS)
            addr = vr(KIMG) if ks == 0 else vr(KA(ks))
            out.append(f"ds_read_b128 {vr(slot, 4)}, {addr} offset:{kb * 32 * ROWB}")
            lds_issued += 1
        last_read[p] = lds_issued - 1

    all_dma = [ln for d in range(NJK + NJV) for ln in dma_piece(d)] if DMA_AT != "v" else []
    if not stream:
        return all_dma
    out.append("@MP1@")
    if DMA_AT == "mhead":
        out += all_dma
    elif DMA_AT == "msplit":
        out += dma_piece(0) + dma_piece(1)
    if qk:
        for ks in range(1, KS):
            out.append(f"v_xor_b32 {vr(KA(ks))}, {ks << 5}, {vr(KIMG)}")
    for p in range(PF):
        reads(p)
    if pv and SUMS == "mfma":
        for kk in range(2 * KB):
            for z in range(NZ):
                out.append(f"v_mfma_f32_16x16x32_@T@ {vr(L(z), 4)}, {vr(SEL, 4)}, {vr(PW(kk, z), 4)}, {vr(L(z), 4)}")
    mid = {}
    if DMA_AT.startswith("mmid") and len(stream) == NKF + NVF:
        # one piece behind every fourth MFMA pair ("mmid") / behind the last four pairs but one ("mmidq"): the wave stalls ~60 cycles per
        # piece with one MFMA in the pipe
        at = (3, 7, 11, 15) if DMA_AT == "mmid" else (9, 11, 13, 15)
        mid = {q: d for d, q in enumerate(at)}
    for p, (kind, idx) in enumerate(stream):
        reads(p + PF)
        out.append(f"s_waitcnt lgkmcnt({lds_issued - 1 - last_read[p]})")
        slot = FR(p % RING)
        if kind == "V":
            kk, eb = divmod(idx, EB)
            for z in range(NZ):
                out.append(f"v_mfma_f32_32x32x16_@T@ {vr(O(z, eb), 16)}, {vr(slot, 4)}, {vr(PW(kk, z), 4)}, {vr(O(z, eb), 16)}")
        else:
            kb, ks = divmod(idx, KS)
            for z in range(NZ):
                c = "0" if ks == 0 else vr(S(z, kb), 16)
                out.append(f"v_mfma_f32_32x32x16_@T@ {vr(S(z, kb), 16)}, {vr(slot, 4)}, {vr(Q(z, ks), 4)}, {c}")
        if p in mid:
            out += dma_piece(mid[p])
    if qk and masked:
        # the validity word of tile t for the vector phase (issued last: nothing in this phase waits for it)
        out += [f"s_lshl_b32 {sr(SA)}, {sr(ST)}, 3", f"s_add_u32 {sr(SA)}, {sr(SA)}, {sr(SVBITS)}", f"v_mov_b32 {vr(T0)}, {sr(SA)}",
                f"ds_read_b64 {vr(T0 + 2, 2)}, {vr(T0)}"]
    if DMA_AT == "mtail" or (DMA_AT.startswith("mmid") and not mid):
        out += all_dma
    elif DMA_AT.startswith("mmid"):
        out += (["s_nop 15"] * 3 if qk else [])
    elif DMA_AT == "msplit":
        out += dma_piece(2) + dma_piece(3) + (["s_nop 15"] * 2 if qk else [])
    elif qk:
        # the vector phase reads S right behind the barrier: idle out the last MFMAs (the final pass of an MFMA issued behind a busy
        # pipe lands up to 64 cycles after its issue)
        out += ["s_nop 15"] * 4
    out.append("@MP0@")
    if ABL & 8:
        out = [ln for ln in out if not (ln.startswith("v_mfma") or ln.startswith("ds_read_b128") or ln.startswith("ds_read_b64_tr") or ln.startswith("s_waitcnt lgkmcnt"))]
    return out


def v_softmax(l1=4, l2=4, dma_at=()):
    """P = exp2(s c2 + nm[z]) in place, then packed in place (register map).  Software pipeline: fma (n), exp (n - l1), convert of the
    pair that ends at n - l1 - l2 (a transcendental's result needs one wait state before a non-transcendental reader: l2 >= 1 gives
    it).  Element n: chunk c = n / 8 = 2 kk + z, element j = n % 8 of it.  dma_at: steps in front of which one LDS-DMA piece goes."""
    out = []
    pieces = list(range(NJK + NJV - len(dma_at), NJK + NJV))

    NEL = 32 * NZ

    def reg(n):
        c, j = n >> 3, n & 7
        kk, z = (c >> 1, c & 1) if NZ == 2 else (c, 0)
        return S(z, kk >> 1) + 8 * (kk & 1) + j, z, kk, j

    for step in range(NEL + l1 + l2):
        if step in dma_at:
            out += dma_piece(pieces.pop(0))
        if step < NEL and PKFMA:
            r, z, _, _ = reg(step)
            if step % 2 == 0:
                sel = "op_sel_hi:[1,0,0]" if z == 0 else "op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                out.append(f"v_pk_fma_f32 {vr(r, 2)}, {vr(r, 2)}, s[66:67], {vr(NM(0), 2)} {sel}")
        elif step < NEL:
            r, z, _, _ = reg(step)
            out.append(f"v_fma_f32 {vr(r)}, {vr(r)}, {sr(SC2)}, {vr(NM(z))}")
        e = step - l1
        if 0 <= e < NEL:
            r, _, _, _ = reg(e)
            out.append(f"v_exp_f32 {vr(r)}, {vr(r)}")
        a = step - l1 - 2
        if SUMS == "valu" and 0 <= a < NEL:
            r, z, _, j = reg(a)
            out.append(f"v_add_f32 {vr(LS(z, j & 1))}, {vr(LS(z, j & 1))}, {vr(r)}")
        m = step - l1 - l2
        if 0 <= m < NEL and (m & 1):
            r, z, kk, j = reg(m)
            out.append(f"v_cvt_pk_@T@_f32 {vr(PW(kk, z) + (j >> 1))}, {vr(r - 1)}, {vr(r)}")
    return out


def rescale():
    """rare: a row max outgrew its reference by more than 2^8 (or the row sees its first key): raise the reference, scale what was
    accumulated at the old one (O, l) exactly once.  Before the first tile O and l are zero: scaled by 0, harmless."""
    out = ["s_nop 15"] * 8           # the PV MFMAs of the last matrix phase have written O (fa_fwd_w64.hpp: 128 idle cycles)
    # the row maxima of both lane halves (lane l <-> l ^ 32): swap the upper half of one copy with the lower half of the other
    out += [f"v_mov_b32 {vr(T0 + 8 + z)}, {vr(MX(z))}" for z in range(NZ)] + ["s_nop 1"]
    out += [f"v_permlane32_swap_b32 {vr(MX(z))}, {vr(T0 + 8 + z)}" for z in range(NZ)]
    out += [f"v_max_f32 {vr(MX(z))}, {vr(MX(z))}, {vr(T0 + 8 + z)}" for z in range(NZ)]
    a0, a1, lane = T0, T0 + 1, T0 + 2
    out += [f"v_mbcnt_lo_u32_b32 {vr(lane)}, -1, 0", f"v_mbcnt_hi_u32_b32 {vr(lane)}, -1, {vr(lane)}", f"v_and_b32 {vr(a0)}, 15, {vr(lane)}",
            f"v_lshlrev_b32 {vr(a0)}, 2, {vr(a0)}", f"v_add_u32 {vr(a1)}, 64, {vr(a0)}"]
    thr, mn, al, b0, b1 = T0 + 3, T0 + 4, T0 + 5, T0 + 6, T0 + 7
    for z in range(NZ):
        out += [f"v_cmp_gt_f32 vcc, {vr(MX(z))}, {vr(THR(z))}",
                f"v_cndmask_b32 {vr(mn)}, {vr(M2(z))}, {vr(MX(z))}, vcc",
                f"v_sub_f32 {vr(al)}, {vr(M2(z))}, {vr(mn)}", f"v_exp_f32 {vr(al)}, {vr(al)}", "s_nop 0",
                f"v_cndmask_b32 {vr(al)}, 1.0, {vr(al)}, vcc",          # not raised: factor one (also when m2 = mn = -inf)
                f"v_mov_b32 {vr(M2(z))}, {vr(mn)}",
                # the values kept beside the reference: -m2 (0 while no key has been seen: P = exp2(-inf) = 0) and the next threshold
                f"v_cmp_eq_f32 vcc, 0xff800000, {vr(mn)}", f"v_sub_f32 {vr(NM(z))}, 0, {vr(mn)}",
                f"v_cndmask_b32 {vr(NM(z))}, {vr(NM(z))}, 0, vcc", f"v_add_f32 {vr(THR(z))}, 0x41000000, {vr(mn)}"]
        for eb in range(EB):
            for i in range(16):
                out.append(f"v_mul_f32 {vr(O(z, eb) + i)}, {vr(O(z, eb) + i)}, {vr(al)}")
        if SUMS == "valu":
            out += [f"v_mul_f32 {vr(LS(z, c))}, {vr(LS(z, c))}, {vr(al)}" for c in range(2)]
            continue
        # the sums of queries n and n + 16 sit in registers 0 / 1 of lanes 0..15: fetch their factors from those queries' lanes
        out += [f"ds_bpermute_b32 {vr(b0)}, {vr(a0)}, {vr(al)}", f"ds_bpermute_b32 {vr(b1)}, {vr(a1)}, {vr(al)}", "s_waitcnt lgkmcnt(0)",
                f"v_mul_f32 {vr(L(z))}, {vr(L(z))}, {vr(b0)}", f"v_mul_f32 {vr(L(z) + 1)}, {vr(L(z) + 1)}, {vr(b1)}"]
    return out


def tick(acc):
    """profile build: add the cycles since the last tick to scalar accumulator `acc` (s_memtime returns through lgkmcnt)"""
    return ["s_memtime s[68:69]", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s76, s68, s70", f"s_add_u32 s{acc}, s{acc}, s76", "s_mov_b32 s70, s68"]


def group_loop(g, masked, prof):
    """the phase loop of key group g (tiles t = g, g + 2, ...).  SYNC == "two": a barrier behind every phase (both groups alike; group 1
    starts one phase late).  SYNC == "one": ONE barrier per iteration -- group 0 behind its vector phase, group 1 behind its matrix
    phase -- so that a wave runs M(t), V(t) back to back and an iteration costs M + V instead of 2 max(M, V) + a second barrier; the
    groups still sit in opposite phases (group 0: M, V | barrier; group 1: M | barrier | V, its first interval holding M(1) only)."""
    L = lambda name: f"L_{name}{g}_%="
    out = [L("loop") + ":"]
    # qk = t < n_live;  pv = t >= 2 && t - 2 < n_live
    out += [f"s_cmp_lt_i32 {sr(ST)}, {sr(SNLIVE)}", f"s_cselect_b32 {sr(SQK)}, 1, 0",
            f"s_sub_i32 {sr(SA)}, {sr(ST)}, 2", f"s_cmp_lt_i32 {sr(SA)}, {sr(SNLIVE)}", f"s_cselect_b32 {sr(SPV)}, 1, 0",
            f"s_cmp_lt_i32 {sr(SA)}, 0", f"s_cselect_b32 {sr(SPV)}, 0, {sr(SPV)}"]
    # fragment read bases: K(t) in slot kA, V(t-2) in slot vA;  DMA sources: K(t+4), V(t+2), clamped to the last tile
    out += [f"v_add_u32 {vr(KIMG)}, {sr(SKA)}, {vr(KLANE)}", f"v_add_u32 {vr(VIMG)}, {sr(SVA)}, {vr(VLANE)}",
            f"s_add_i32 {sr(SA)}, {sr(ST)}, 4", f"s_lshl_b32 {sr(SA)}, {sr(SA)}, {TILE_SHIFT}", f"s_min_u32 {sr(SKOFF)}, {sr(SA)}, {sr(SLAST)}",
            f"s_add_i32 {sr(SA)}, {sr(ST)}, 2", f"s_lshl_b32 {sr(SA)}, {sr(SA)}, {TILE_SHIFT}", f"s_min_u32 {sr(SVOFF)}, {sr(SA)}, {sr(SLAST)}"]
    out += [f"s_cmp_eq_u32 {sr(SQK)}, 0", f"s_cbranch_scc1 {L('noqk')}", f"s_cmp_eq_u32 {sr(SPV)}, 0", f"s_cbranch_scc1 {L('mqk')}"]
    out += m_phase(True, True, masked) + [f"s_branch {L('mdone')}", L("mqk") + ":"] + m_phase(True, False, masked) + [f"s_branch {L('mdone')}", L("noqk") + ":"]
    out += [f"s_cmp_eq_u32 {sr(SPV)}, 0", f"s_cbranch_scc1 {L('mnone')}"] + m_phase(False, True, masked) + [f"s_branch {L('mdone')}", L("mnone") + ":"]
    out += m_phase(False, False, masked) + [L("mdone") + ":"] + (tick(71) if prof else [])
    # the DMA batch issued one iteration ago -- K(t+2), V(t): what M(t+2) reads -- has landed; behind the barrier every wave's has
    sync = (tick(73) if prof else []) + [f"s_waitcnt vmcnt({NJK + NJV})"] + (tick(74) if prof else []) + ["s_barrier"] + (tick(75) if prof else [])
    if SYNC == "two":
        out += ["s_barrier"] + (tick(72) if prof else [])
        out += [f"s_add_i32 {sr(SA)}, {sr(ST)}, 1", f"s_cmp_ge_i32 {sr(SA)}, {sr(SH)}", f"s_cbranch_scc1 {L('exit')}"]
    elif g == 1:
        out += sync
    out += [f"s_cmp_eq_u32 {sr(SQK)}, 0", f"s_cbranch_scc1 {L('vdone')}"]
    # ---- vector phase
    out += ["@VP1@"] + (dma_piece(0) + dma_piece(1) + ["s_nop 15", "s_nop 15"] if DMA_AT == "v" else [])
    if masked:
        out += ["s_waitcnt lgkmcnt(0)"] + v_mask(g)
    out += ([f"s_mov_b64 {SM}, 0", "s_cmp_lg_u32 0, 0"] if ABL & 2 else v_rowmax())
    out += [f"s_cbranch_scc1 {L('rescale')}", L("rescdone") + ":"]          # (s_or_b64 sets SCC = result != 0)
    # ("v": the first two pieces lead the phase, the others go between the element-wise steps)
    spread = ((16, 40) if NZ == 2 else (12, 28)) if NJK + NJV == 4 else (4, 10, 16, 22, 28, 34)
    out += ([] if ABL & 4 else v_softmax(dma_at=spread if DMA_AT == "v" else ())) + ["@VP0@"]
    if DMA_AT == "v":                # a wave without the tile keeps the group's DMA schedule
        out += [f"s_branch {L('vdone2')}", L("vdone") + ":"]
        for d in range(NJK + NJV):
            out += dma_piece(d)
        out += [L("vdone2") + ":"]
    else:
        out += [L("vdone") + ":"]
    if SYNC == "two" or g == 0:
        out += sync
    # t += 2: the group's slots of either ring rotate, (A, B[, C]) <- (B[, C], A)
    for x, y, z in ((SKA, SKB, SKC), (SVA, SVB, SVC)):
        if SLOTS == 2:
            out += [f"s_mov_b32 {sr(SA)}, {sr(x)}", f"s_mov_b32 {sr(x)}, {sr(y)}", f"s_mov_b32 {sr(y)}, {sr(SA)}"]
        else:
            out += [f"s_mov_b32 {sr(SA)}, {sr(x)}", f"s_mov_b32 {sr(x)}, {sr(y)}", f"s_mov_b32 {sr(y)}, {sr(z)}", f"s_mov_b32 {sr(z)}, {sr(SA)}"]
    out += [f"s_add_i32 {sr(ST)}, {sr(ST)}, 2", f"s_cmp_lt_i32 {sr(ST)}, {sr(SH)}", f"s_cbranch_scc1 {L('loop')}"]
    if SYNC == "one" and g == 1:
        # group 0 runs ceil(H / 2) iterations = barriers, group 1 floor(H / 2): one more when H is odd
        out += [f"s_bitcmp1_b32 {sr(SH)}, 0", f"s_cbranch_scc0 {L('exit')}", "s_barrier"]
    out += [f"s_branch {L('exit')}", L("rescale") + ":"] + rescale() + [f"s_branch {L('rescdone')}", L("exit") + ":"]
    return out


def loop(masked, prof=False):
    """the statement: group 0's loop and group 1's loop (SYNC == "two": one loop for both, group 1 enters it behind a barrier of its
    own in front of the statement).  Half-steps h = 0 .. n_tiles + 1: group g runs M(t) at h = t for t = g (mod 2) and V(t) at h = t + 1;
    PV(t) happens in M(t + 2)."""
    out = []
    out += [f"v_mov_b32 {vr(NM(z))}, 0" for z in range(NZ)] + [f"v_mov_b32 {vr(THR(z))}, 0xff800000" for z in range(NZ)]     # m2 = -inf
    if PKFMA:
        out += [f"v_readfirstlane_b32 s66, {sr(SC2)}", f"v_readfirstlane_b32 s67, {sr(SC2)}"]        # (the scale operand arrives in a VGPR)
    if E == 128:
        # DMA source offsets of this wave's K pieces 1..3: piece j covers image rows 4 j .. 4 j + 3 of the wave's 16 and the image XORs a
        # row's 16-byte chunks with row & 15, so offset j = offset 0 with chunk bits 2..3 flipped by j
        out += [f"v_xor_b32 {vr(KVO(j))}, {j << 6}, {vr(KVO(0))}" for j in range(1, NJK)]
    if prof:                         # s71..s75: cycles in M, at the barrier behind it, in V, in the DMA wait, at the barrier behind that
        out += ["s_memtime s[68:69]", "s_waitcnt lgkmcnt(0)", "s_mov_b32 s70, s68"] + [f"s_mov_b32 s{a}, 0" for a in range(71, 76)]
    if SYNC == "two":
        out += group_loop(0, masked, prof)
    else:
        out += [f"s_cmp_lg_u32 {sr(ST)}, 0", "s_cbranch_scc1 L_loop1_%="]      # t = the wave's key group on entry
        out += group_loop(0, masked, prof) + ["s_branch L_end_%="] + group_loop(1, masked, prof) + ["L_end_%=:"]
    if prof:
        out += [f"v_mov_b32 v{224 + i}, s{71 + i}" for i in range(5)]
    return out


def check_stream(lines):
    """Static audit of one generated loop (hipcc pads nothing inside an asm statement, so the streams carry their own waits):
      1. every fragment an MFMA reads has been requested and WAITED for (LDS returns in order: s_waitcnt lgkmcnt(n) retires all but
         the n youngest requests), and no LDS read overwrites a fragment register that is still waiting to be consumed;
      2. a transcendental's result is not read by the very next instruction (one wait state, gfx940+);
      3. v_permlane32_swap is not directly behind a VALU write of its operands (two wait states); s_mov m0 -> LDS-DMA has its s_nop;
      4. behind the last QK^T MFMA of a matrix phase, >= 16 instructions (>= 64 cycles at 4 cycles of issue each) or as many explicit
         idle cycles precede the first vector read of a score register of key block 1 (key block 0 was finished 8 MFMAs earlier): the
         final pass of an MFMA issued behind a busy pipe lands up to 64 cycles after its issue;
      5. the P^T words an MFMA reads were all written (v_cvt_pk) in the vector phase before (structural: same registers as S)."""
    import re
    reg_re = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")

    def regs(tok):
        out = set()
        for m in reg_re.finditer(tok):
            if m.group(1):
                out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
            else:
                out.add(int(m.group(3)))
        return out

    pending = []                     # outstanding LDS requests, oldest first: sets of destination registers
    prev = ""
    since_barrier, idle = None, 0
    for ln in lines:
        op = ln.split()[0] if ln and not ln.endswith(":") and not ln.startswith("@") else ""
        args = ln[len(op):].split(",") if op else []
        if ln.endswith(":") or op in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1"):
            # control flow: every path into a label has drained its requests in these streams (phases end with lgkmcnt(0) or MFMAs that
            # consumed everything); keep the model simple and conservative
            if op == "":
                pending = [] if not ln.startswith("L_mdone") else pending
        if op.startswith("ds_read") or op.startswith("ds_bpermute"):
            dst = regs(args[0])
            for d in pending:
                assert not (d & dst), f"LDS read overwrites a pending fragment: {ln}"
            pending.append(dst)
        elif op == "s_memtime":
            pending.append(set())
        elif op == "s_waitcnt" and "lgkmcnt" in ln:
            n = int(re.search(r"lgkmcnt\((\d+)\)", ln).group(1))
            pending = pending[len(pending) - n:] if n else []
        elif op.startswith("v_mfma"):
            src = regs(args[1]) | regs(args[2])
            for d in pending:
                assert not (d & src), f"MFMA reads a fragment that was not waited for: {ln}"
        elif op.startswith("v_") or op.startswith("buffer_"):
            src = set().union(*[regs(a) for a in args[1:]]) if len(args) > 1 else set()
            for d in pending:
                assert not (d & src), f"vector instruction reads an LDS result that was not waited for: {ln}"
        if prev.startswith("v_exp_f32") and op.startswith("v_") and not op.startswith("v_exp"):
            assert not (regs(prev.split(",")[0]) & set().union(*[regs(a) for a in args[1:]])), f"transcendental result read without a wait state: {ln}"
        if op == "v_permlane32_swap_b32":
            assert prev.startswith("s_nop") or prev.startswith("v_permlane32_swap"), f"v_permlane32_swap directly behind a vector instruction: {ln}"
        if op.startswith("buffer_load") and prev.startswith("s_mov_b32 m0"):
            assert False, f"LDS-DMA directly behind the M0 write: {ln}"
        # rule 4
        if op.startswith("v_mfma_f32_32x32x16") and regs(args[0]) & (set(range(S(0, 1), S(0, 1) + 16)) | set(range(S(1, 1), S(1, 1) + 16))):
            since_barrier, idle = 0, 0
        elif since_barrier is not None and op:
            if op == "s_nop":
                idle += int(ln.split()[1]) + 1
            elif op.startswith("v_") and not op.startswith("v_mfma"):
                kb1 = set(range(S(0, 1), S(0, 1) + 16)) | set(range(S(1, 1), S(1, 1) + 16))
                if set().union(*[regs(a) for a in args[1:]]) & kb1:
                    assert 4 * since_barrier + idle >= 64, f"score registers of key block 1 read {since_barrier} instructions behind the MFMA that writes them: {ln}"
                    since_barrier = None
                else:
                    since_barrier += 1
            elif op.startswith("v_mfma"):
                since_barrier = None
            elif op.startswith("buffer_") or op.startswith("s_"):
                since_barrier += 1
        if op:
            prev = ln
    return True


def as_macro(name, lines):
    body = " \\\n".join(f'    "{ln}\\n\\t"' for ln in lines)
    text = f"#define {name}(TS, MP1, MP0, VP1, VP0) \\\n{body}\n".replace("@T@", '" TS "')
    for k in ("MP1", "MP0", "VP1", "VP0"):      # optional s_setprio at the phase boundaries (experiments; empty strings in the release)
        text = text.replace(f"@{k}@", f'" {k} "')
    return text


def render():
    parts = ["// GENERATED by tools/gen_duo_asm.py -- do not edit; tests/test_duo_codegen.py checks that it is up to date.\n"
             "// The phase loop of fa_fwd_duo.hpp with fixed physical registers (register map: the generator's header).\n"
             "// TS: the element type's mnemonic suffix (\"bf16\" / \"f16\").\n"
             f"#define NNOP_DUO_SLOTS_PER_GROUP {SLOTS}      // ring slots per key group and ring (where the DMA batch is issued decides)\n"
             f"#define NNOP_DUO_VALU_SUMS {1 if SUMS == 'valu' else 0}            // 1: row sums by v_add_f32 in the vector phase (4 chains in v[248:251])\n"
             f"#define NNOP_DUO_SYNC_ONE {1 if SYNC == 'one' else 0}             // 1: one barrier per iteration, no barrier of group 1 in front of the statement\n"]
    for nz, e in ((2, 64), (1, 64), (1, 128), (2, 32)):
        set_nz(nz, e)
        for masked in (False, True):
            check_stream(loop(masked))
    set_nz(2)
    parts.append(as_macro("NNOP_DUO_LOOP_PLAIN", loop(False)))
    parts.append(as_macro("NNOP_DUO_LOOP_MASKED", loop(True)))
    set_nz(1)
    parts.append("// the same loop with ONE 32-row query block per wave (128-row workgroups): the z = 1 registers and instructions left out\n")
    parts.append(as_macro("NNOP_DUO1_LOOP_PLAIN", loop(False)))
    parts.append(as_macro("NNOP_DUO1_LOOP_MASKED", loop(True)))
    set_nz(1, 128)
    parts.append("// E = 128, one 32-row query block per wave: 16 + 16 MFMAs per tile, 2 ring slots per key group, the DMA batch in the vector phase,\n"
                 "// a barrier behind every phase (group 1 enters behind a barrier of its own in front of the statement)\n"
                 f"#define NNOP_DUO128_SLOTS_PER_GROUP {SLOTS}\n#define NNOP_DUO128_SYNC_ONE {1 if SYNC == 'one' else 0}\n")
    parts.append(as_macro("NNOP_DUO128_LOOP_PLAIN", loop(False)))
    parts.append(as_macro("NNOP_DUO128_LOOP_MASKED", loop(True)))
    set_nz(2, 32)
    parts.append("// E = 32, 64-row waves: the E = 64 loop with two contraction steps and one 32-column block of O^T per query block\n")
    parts.append(as_macro("NNOP_DUO32_LOOP_PLAIN", loop(False)))
    parts.append(as_macro("NNOP_DUO32_LOOP_MASKED", loop(True)))
    set_nz(2)
    parts.append("// profile builds (make DEV=1 VAR=-DNNOP_DUO_STAMP=1): the same loops with s_memtime ticks; the five accumulators leave in v[224:228]\n"
                 "#ifdef NNOP_DEV_BUILD")
    parts.append(as_macro("NNOP_DUO_LOOP_PLAIN_PROF", loop(False, True)))
    parts.append(as_macro("NNOP_DUO_LOOP_MASKED_PROF", loop(True, True)))
    parts.append("#endif")
    return "\n".join(parts)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}: {text.count(chr(10))} lines")
