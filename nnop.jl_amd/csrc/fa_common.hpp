// fa_common.hpp -- device-side building blocks shared by the gfx950 flash-attention kernels.
//
// Everything here is written for CDNA4 (MI355X, gfx950) only: 64-lane wavefronts,
// v_mfma_f32_32x32x16_{bf16,f16} / v_mfma_f32_32x32x2_f32, ds_read_b64_tr_b16,
// v_permlane32_swap.  There is no other target and no fallback.
//
// Replaces (does not translate) the reference's LDS-tile scalar GEMM `mma!` and its
// FATileConfig index conventions (src/mma.jl:1-48): the contraction runs on the matrix
// cores with fp32 accumulation, and the tile "configs" become the two LDS image types below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nnop {

typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef float    f32x8  __attribute__((ext_vector_type(8)));
typedef float    f32x4  __attribute__((ext_vector_type(4)));
typedef float    f32x2  __attribute__((ext_vector_type(2)));
typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16   bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16   bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4  __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2  __attribute__((ext_vector_type(2)));
typedef short    s16x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2  __attribute__((ext_vector_type(2)));

#define NNOP_DEV __device__ __forceinline__

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2   = 0.6931471805599453f;

// ----------------------------------------------------------------------------------------
// Element traits.  A "fragment" is what ONE lane feeds to one 16-deep contraction step:
// 8 elements.  For the 16-bit types that is exactly the A/B operand of
// v_mfma_f32_32x32x16 (lane l = (r = l&31, h = l>>5) holds k = 8h + j, j = 0..7).  For fp32
// the same 8 elements feed eight v_mfma_f32_32x32x2_f32 (lane holds k = h of each), i.e.
// MFMA j contracts k in {j, 8 + j}: any k order is legal as long as A and B agree, so both
// dtypes share one data layout.
// ----------------------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    using frag = f32x8;
    static constexpr int kBytes = 4;
};
template <> struct Elem<_Float16> {
    using frag = f16x8;
    static constexpr int kBytes = 2;
};
template <> struct Elem<__bf16> {
    using frag = bf16x8;
    static constexpr int kBytes = 2;
};

template <typename T> NNOP_DEV f32x16 mma16(typename Elem<T>::frag a, typename Elem<T>::frag b, f32x16 c);
template <> NNOP_DEV f32x16 mma16<__bf16>(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> NNOP_DEV f32x16 mma16<_Float16>(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <> NNOP_DEV f32x16 mma16<float>(f32x8 a, f32x8 b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}

// 8 consecutive accumulator registers (8s .. 8s+7) of a 32x32 result X as the fragment of
// contraction step s of the NEXT product (the product then sums over X's ROW index; element
// j of lane half h is X row 16s + 8(j>>2) + 4h + (j&3)).  No lane movement, no LDS.
template <typename T, int S> NNOP_DEV typename Elem<T>::frag acc_frag(const f32x16& x) {
    f32x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = x[8 * S + j];
    if constexpr (sizeof(T) == 4) {
        return t;
    } else {
        return __builtin_convertvector(t, typename Elem<T>::frag);
    }
}

// Row index (within a 32x32 MFMA result) held in accumulator register i by lane half h.
NNOP_DEV constexpr int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

NNOP_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max over both 32-lane halves (lane l <-> l^32), result in every lane.
NNOP_DEV float half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
NNOP_DEV float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Force a register-resident value to have LANDED here: the compiler must insert the s_waitcnt for
// the load that produces it at this point, not at its first use.  Used before a pipelined loop for
// loads issued ahead of it: vmcnt retires in order, so a wait left inside the loop for an OLD load
// would also drain the loop's own prefetch every iteration.
template <typename V> NNOP_DEV void landed(V& v) { asm volatile("" : "+v"(v)); }

template <typename T> NNOP_DEV float to_f32(T x) { return (float)x; }
template <typename T> NNOP_DEV T from_f32(float x) { return (T)x; }

// ----------------------------------------------------------------------------------------
// LDS images of a [ROWS][E] tile (ROWS = keys or queries, E = embedding, E fastest in HBM).
//
// RowImg: row-major, 16-byte chunks XOR-swizzled so that the MFMA "row read" -- lane (r,h)
// takes 8 consecutive embedding elements of row r -- is bank-conflict free for ds_read_b128
// (bank = (addr/4) % 64, serviced in 16-lane groups whose rows are distinct mod 16).
//   chunk' = chunk ^ x(row),  x(row) = row & 15                      if row bytes >= 256
//                             x(row) = (row / rows_per_256B) & (n16-1) otherwise
// Whole rows are written from coalesced 16-byte global loads (8 consecutive lanes write a
// permutation of one contiguous 128-byte span: conflict-free ds_write_b128).
// ----------------------------------------------------------------------------------------
template <typename T, int E> struct RowImg {
    static constexpr int kRowBytes = E * (int)sizeof(T);
    static constexpr int kN16      = kRowBytes / 16;            // 16-byte chunks per row
    static_assert(kRowBytes >= 32, "row must hold one 16-deep contraction step");
    static constexpr int bytes(int rows) { return rows * kRowBytes; }

    NNOP_DEV static int xor_of(int row) {
        if constexpr (kN16 >= 16) return row & 15;
        else return (row / (16 / kN16)) & (kN16 - 1);
    }
    // byte offset of 16-byte chunk `c16` of row `row`
    NNOP_DEV static int off(int row, int c16) { return row * kRowBytes + ((c16 ^ xor_of(row)) << 4); }

    NNOP_DEV static void write16(char* img, int row, int c16, u32x4 v) {
        *reinterpret_cast<u32x4*>(img + off(row, c16)) = v;
    }
    // fragment of contraction step ks (16 embedding elements) for this lane's row.
    // `row` = tile row of this lane, h = lane >> 5.
    NNOP_DEV static typename Elem<T>::frag read_row_frag(const char* img, int row, int h, int ks) {
        if constexpr (sizeof(T) == 2) {
            return *reinterpret_cast<const typename Elem<T>::frag*>(img + off(row, 2 * ks + h));
        } else {
            f32x4 a = *reinterpret_cast<const f32x4*>(img + off(row, 4 * ks + 2 * h));
            f32x4 b = *reinterpret_cast<const f32x4*>(img + off(row, 4 * ks + 2 * h + 1));
            f32x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            return r;
        }
    }
    // fp32 only: the MFMA "column read" (see ColImg) served from THIS swizzled row-major image,
    // so that a tile consumed both ways (backward: Q, dO, K) needs one LDS copy.  Lane (r,h)
    // reads column 32*eb + r of rows 16*kk + 8(j>>2) + 4h + (j&3): 32 lanes touch a permutation
    // of 128 contiguous bytes of one row -> conflict-free ds_read_b32.
    NNOP_DEV static f32x8 read_col_frag_f32(const char* img, int r, int h, int kk, int eb) {
        static_assert(sizeof(T) == 4 || E > 0, "");
        f32x8 out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 16 * kk + 8 * (j >> 2) + 4 * h + (j & 3);
            out[j] = *reinterpret_cast<const float*>(img + off(row, (8 * eb + (r >> 2)) & (kN16 - 1)) + ((r & 3) << 2));
        }
        return out;
    }
};

// ----------------------------------------------------------------------------------------
// ColImg: image for the MFMA "column read" -- lane (r,h) takes, for embedding column
// e = 32*eb + r, the 8 tile rows 16*kk + 8(j>>2) + 4h + (j&3), j = 0..7 (the row order an
// accumulator tile has when it is the other operand, see acc_frag).
//   16-bit: blocks of 4 rows x 32 columns stored contiguously (256 B = one LDS bank row):
//           [row>>2][e>>5][row&3][e&31]; read with two ds_read_b64_tr_b16 (the hardware
//           4x16 transpose), each 32-lane half covering exactly one 256-byte block.
//           E = 16 is padded to 32 columns (upper 16 never stored, results discarded).
//   fp32  : plain row-major; eight ds_read_b32, 32 lanes reading 128 contiguous bytes.
// ----------------------------------------------------------------------------------------
template <typename T, int E> struct ColImg {
    static constexpr int kEP  = (sizeof(T) == 2 && E < 32) ? 32 : E;     // padded columns
    static constexpr int kEB  = kEP / 32;                                  // 32-column blocks
    static constexpr int kRowBytes = kEP * (int)sizeof(T);
    static constexpr int bytes(int rows) { return rows * kRowBytes; }

    NNOP_DEV static int off16(int row, int c16) {   // 16-bit: chunk c16 = 8 columns
        return (((row >> 2) * kEB + (c16 >> 2)) << 8) + ((row & 3) << 6) + ((c16 & 3) << 4);
    }
    NNOP_DEV static void write16(char* img, int row, int c16, u32x4 v) {
        if constexpr (sizeof(T) == 2) *reinterpret_cast<u32x4*>(img + off16(row, c16)) = v;
        else *reinterpret_cast<u32x4*>(img + row * kRowBytes + (c16 << 4)) = v;
    }
    // per-lane constant part of the column-read address (compute once per kernel)
    NNOP_DEV static int lane_base(int lane) {
        if constexpr (sizeof(T) == 2) {
            const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
            return h * kEB * 256 + q * 64 + g1 * 32 + p * 8;
        } else {
            const int h = lane >> 5, r = lane & 31;
            return 4 * h * kRowBytes + r * 4;
        }
    }
    // fragment for 16-row step kk, column block eb.  base = img + lane_base(lane).
    NNOP_DEV static typename Elem<T>::frag read_col_frag(const char* base, int kk, int eb) {
        if constexpr (sizeof(T) == 2) {
            typedef __attribute__((address_space(3))) s16x4* lds_p;
            const char* p0 = base + (((4 * kk) * kEB + eb) << 8);
            const char* p1 = base + (((4 * kk + 2) * kEB + eb) << 8);
            s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
            s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            return __builtin_bit_cast(typename Elem<T>::frag, r);
        } else {
            f32x8 r;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 16 * kk + 8 * (j >> 2) + (j & 3);
                r[j] = *reinterpret_cast<const float*>(base + row * kRowBytes + eb * 128);
            }
            return r;
        }
    }
};

// ----------------------------------------------------------------------------------------
// Stager: the NT threads of a workgroup move a dense [ROWS][E] tile HBM -> registers (16-byte
// coalesced loads, issued early) -> LDS images (written late, after the compute phase that
// hides the load latency).  Rows >= rows_valid are zero-filled.
// ----------------------------------------------------------------------------------------
template <typename T, int E, int ROWS, int NT> struct Stager {
    static constexpr int kN16 = E * (int)sizeof(T) / 16;
    static constexpr int kNCH = ROWS * kN16;
    static constexpr int kNLD = (kNCH + NT - 1) / NT;
    u32x4 reg[kNLD];
    int rows_valid_ = ROWS;      // rows of the held tile that exist; the others are zeroed when the tile is WRITTEN

    // gtile: address of row 0 of the tile; rows_valid: rows that exist, 1 <= rows_valid (may exceed ROWS).
    // Branch-free: the row index is clamped for the load, so no exec-masked region splits the surrounding instruction
    // stream.  Rows past the end are zero-filled by write(), NOT here: a select on the loaded registers at this point
    // would put the wait for the load right behind its issue and expose the HBM / L2 latency on every tile (measured:
    // +20 % per tile in the masked forward, see profiles/r01/NOTES.md).
    NNOP_DEV void load(const void* gtile, int rows_valid, int tid) {
        rows_valid_ = rows_valid;
#pragma unroll
        for (int i = 0; i < kNLD; ++i) {
            int c = tid + i * NT;
            if (kNCH % NT != 0) c = c < kNCH ? c : kNCH - 1;          // surplus lanes re-read the last chunk
            const int row = c / kN16, c16 = c % kN16;
            const int rowc = row < rows_valid ? row : rows_valid - 1;
            reg[i] = *reinterpret_cast<const u32x4*>((const char*)gtile + ((size_t)rowc * kN16 + c16) * 16);
        }
    }
    // same, when the caller guarantees rows_valid >= ROWS (every row exists)
    NNOP_DEV void load_full(const void* gtile, int tid) {
        rows_valid_ = ROWS;
#pragma unroll
        for (int i = 0; i < kNLD; ++i) {
            int c = tid + i * NT;
            if (kNCH % NT != 0) c = c < kNCH ? c : kNCH - 1;
            reg[i] = *reinterpret_cast<const u32x4*>((const char*)gtile + (size_t)c * 16);
        }
    }
    // ZFILL: zero the rows >= rows_valid_ of a tile fetched with load(); false for tiles fetched with load_full()
    template <typename Img, bool ZFILL = true> NNOP_DEV void write(char* img, int tid) const {
#pragma unroll
        for (int i = 0; i < kNLD; ++i) {
            const int c = tid + i * NT;
            if (kNCH % NT == 0 || c < kNCH) {
                u32x4 v = reg[i];
                if constexpr (ZFILL) {
                    const bool ok = c / kN16 < rows_valid_;
                    v[0] = ok ? v[0] : 0u; v[1] = ok ? v[1] : 0u; v[2] = ok ? v[2] : 0u; v[3] = ok ? v[3] : 0u;
                }
                Img::write16(img, c / kN16, c % kN16, v);
            }
        }
    }
};

// ----------------------------------------------------------------------------------------
// Key-padding mask -> LDS.  ONE pass over a batch's mask row [0, nkeys): every thread turns 16 mask bytes
// (one 16-byte load when the row is 16-byte aligned) into a 16-bit piece of the per-64-key validity words
// `words` (so word w covers keys 64w..64w+63) and tracks the last valid key.  Returns that index (-1: none).
// All threads of the workgroup must call it (two barriers inside).
// ----------------------------------------------------------------------------------------
NNOP_DEV int kpad_scan(const uint8_t* __restrict__ mp, int KL, int nkeys, uint64_t* words, int max_words,
                       int* slot, int tid, int nthreads) {
    if (tid == 0) *slot = -1;
    __syncthreads();
    uint16_t* w16 = reinterpret_cast<uint16_t*>(words);
    const bool aligned = (reinterpret_cast<uintptr_t>(mp) & 15) == 0;
    const int nround = (nkeys + 63) & ~63;                 // whole words get written (tail bits = 0)
    int last = -1;
    for (int c = tid; c * 16 < nround; c += nthreads) {
        const int k0 = c * 16;
        uint32_t bits = 0;
        if (aligned && k0 + 16 <= KL) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(mp + k0);
#pragma unroll
            for (int j = 0; j < 16; ++j) bits |= (((v[j >> 2] >> (8 * (j & 3))) & 0xffu) != 0u ? 1u : 0u) << j;
        } else {
            // unaligned row (KL not a multiple of 16) or the row's tail: 16 independent byte loads from clamped
            // addresses, combined afterwards -- one wait, not one per byte
            uint8_t by[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) by[j] = mp[k0 + j < KL ? k0 + j : KL - 1];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (k0 + j < KL && by[j] != 0) bits |= 1u << j;
        }
        if (k0 + 16 > nkeys) bits &= (k0 < nkeys) ? ((1u << (nkeys - k0)) - 1u) : 0u;
        if (c < 4 * max_words) w16[c] = (uint16_t)bits;
        if (bits) last = k0 + 31 - __builtin_clz(bits);
    }
    if (last >= 0) atomicMax(slot, last);
    __syncthreads();
    return *slot;
}
// validity bits of the BK keys of tile t (BK = 32 or 64), wave-uniform, from the words built by kpad_scan
template <int BK> NNOP_DEV uint64_t kpad_tile_bits(const uint64_t* words, int t) {
    const uint64_t w = words[(t * BK) >> 6];
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w >> 32)) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w);
    if constexpr (BK == 64) return u;
    else return (u >> ((t * BK) & 63)) & ((1ull << BK) - 1ull);
}

// XCD-aware, bijective remap of a linear workgroup id: workgroups b and b+8 share an XCD
// (observed round-robin dispatch; a speed assumption only, never correctness), so give each
// XCD one contiguous span of the logical tile order -> neighbouring tiles (same K/V) share
// an L2.
NNOP_DEV int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}
// Same purpose, finer grain: the grid is `n / chunk` chunks of `chunk` consecutive tiles that share their K/V (all
// q-blocks of the q-heads of one (batch, kv-head)); chunks are dealt round-robin to the 8 XCDs, so every XCD gets
// chunks of every batch.  One contiguous eighth per XCD (xcd_remap) is badly unbalanced when the work per batch
// differs (variable sequence lengths: measured 1.4x between XCDs at BASELINE config 4).  Needs (n / chunk) % 8 == 0;
// otherwise falls back to xcd_remap.
NNOP_DEV int xcd_remap_chunked(int id, int n, int chunk) {
    const int n_chunks = n / chunk;
    if (chunk <= 0 || n_chunks * chunk != n || (n_chunks & 7) != 0) return xcd_remap(id, n);
    const int x = id & 7, s = id >> 3;
    return ((s / chunk) * 8 + x) * chunk + (s % chunk);
}

// Pair-bias kernels (gather straight from the reference layout [B][KL][QL][QH], head fastest): the `nh` heads of one (batch, block)
// read the SAME cache lines -- 2 bytes of every 2 nh -- so they go to the same XCD in consecutive dispatch slots (ids x, x + 8, ...
// of that XCD): one L2 then fetches each line once for all heads instead of every head's XCD fetching it again.  Returns the
// linear index in the kernels' usual (batch, head, block) order.  Bijective for every n_units; the last n_units % 8 units keep the
// plain order.
NNOP_DEV int xcd_remap_heads(int id, int n_blk, int nh, int n_units) {
    const int full = n_units & ~7;
    int unit, head;
    if (id < full * nh) {
        const int x = id & 7, s = id >> 3;
        unit = (s / nh) * 8 + x;
        head = s % nh;
    } else {
        const int rel = id - full * nh;
        unit = full + rel / nh;
        head = rel % nh;
    }
    const int b = unit / n_blk, blk = unit - b * n_blk;
    return (b * nh + head) * n_blk + blk;
}

}  // namespace nnop
