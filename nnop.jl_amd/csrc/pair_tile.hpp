// pair_tile.hpp -- the additive pair bias and its gradient in the backward kernels, without 2-byte global accesses.
//
// `pair` / `dpair` are [B][KL][QL][QH] with the HEAD fastest (src/attention.jl:62, src/attention_bwd.jl:123-132): for one
// head, neighbouring (query, key) elements are QH elements apart, so a kernel that works on one (batch, head) can only
// touch them one element per lane and instruction -- 16 loads (+ 16 stores) per 32 x 32 tile and wave, and that instruction
// count, not the bytes, was the whole cost of the pair-bias backward (round 1: dQ kernel 405 us with bias, 40 without).
//
// With scratch space from the caller's workspace (nnop_fa_bwd_workspace_bytes_pair) the backward instead runs
//   1. pair_pack_kernel    pair -> one head-major copy A [B][QH][KLp][QLp] (q contiguous), zero-padded to multiples of 64
//   2. the two kernels in MODE 3: a wave moves its 32 x 32 bias tile with two 16-byte loads per lane (PairTile::fetch, ahead
//      of the MFMAs) into a wave-private LDS tile and picks it up in accumulator layout -- the dQ kernel (lane = query,
//      registers = keys) with the hardware-transposed LDS read (unpack), the dK/dV kernel (lane = key, registers =
//      queries) from its own row of a padded row-major tile (unpack_rows): 2 global loads + 2-4 LDS writes + 4 LDS reads
//      instead of 16 two-byte global loads; the dQ kernel writes dS the reverse way into
//      S [B][QH][QLp][KLp] (PairTile::store: 4 LDS writes + 4 LDS reads + 2 global 16-byte stores instead of 16 stores)
//   3. dpair_unpack_kernel  S -> dpair, writing zeros where no tile was visited (causally hidden or padded keys).
// MODE 2 (direct 2-byte accesses, dpair zero-filled first) stays as the path for callers that pass the small workspace.
#pragma once
#include "fa_common.hpp"

namespace nnop {

__host__ __device__ constexpr int pair_pad(int n) { return (n + 63) & ~63; }

template <typename T> struct PairTile {
    // wave-private LDS tile: 32 x 32 elements; the row-major variant (unpack_rows) pads its rows against bank conflicts
    static constexpr int kRowPad = sizeof(T) == 2 ? 72 : 132;    // bytes per padded row
    static constexpr int kBytes = 32 * kRowPad;
    using Img = ColImg<T, 32>;

    // 32 x 32 tile at `g` (row stride `ld` elements, 16-byte aligned rows): rows = the kernel's accumulator REGISTER axis,
    // columns = its LANE axis.  Two steps so that the kernel can put its MFMAs between them (the loads' latency):
    //   fetch : 16-byte global loads into registers (2 for 16-bit types, 4 for fp32)
    //   unpack: registers -> wave-private LDS tile -> out[i] = tile[acc_row(i, h)][r] for lane (r, h), as fp32
    static constexpr int kRegs = sizeof(T) == 2 ? 2 : 4;
    struct Regs { u32x4 v[kRegs]; };
    NNOP_DEV static Regs fetch(const T* __restrict__ g, size_t ld, int lane) {
        Regs x;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) x.v[j] = *reinterpret_cast<const u32x4*>(g + (size_t)(16 * j + (lane >> 2)) * ld + 8 * (lane & 3));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) x.v[j] = *reinterpret_cast<const u32x4*>(g + (size_t)(8 * j + (lane >> 3)) * ld + 4 * (lane & 7));
        }
        return x;
    }
    NNOP_DEV static void unpack(const Regs& x, char* lds, int lane, float (&out)[16]) {
        const int r = lane & 31, h = lane >> 5;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) Img::write16(lds, 16 * j + (lane >> 2), lane & 3, x.v[j]);
            __builtin_amdgcn_wave_barrier();
            const char* base = lds + Img::lane_base(lane);
            const typename Elem<T>::frag f0 = Img::read_col_frag(base, 0, 0), f1 = Img::read_col_frag(base, 1, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { out[j] = to_f32(f0[j]); out[8 + j] = to_f32(f1[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(lds + (8 * j + (lane >> 3)) * 128 + 16 * (lane & 7)) = x.v[j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 16; ++i) out[i] = *reinterpret_cast<const float*>(lds + acc_row(i, h) * 128 + 4 * r);
        }
        __builtin_amdgcn_wave_barrier();                         // the tile may be overwritten (store) right away
    }

    // Same registers, other orientation: out[i] = tile[r][acc_row(i, h)] -- the lane's own ROW of the tile (rows = the kernel's
    // LANE axis, columns = its register axis).  The dK/dV kernel (lane = key) reads the q-contiguous copy of the bias this
    // way, so that one head-major copy serves both kernels.  Row-major LDS tile with padded rows (72 / 132 bytes: a lane's
    // 8-byte / 4-byte reads then spread over the banks).
    NNOP_DEV static void unpack_rows(const Regs& x, char* lds, int lane, float (&out)[16]) {
        const int r = lane & 31, h = lane >> 5;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                char* d = lds + (16 * j + (lane >> 2)) * kRowPad + 16 * (lane & 3);
                *reinterpret_cast<u32x2*>(d) = u32x2{x.v[j][0], x.v[j][1]};
                *reinterpret_cast<u32x2*>(d + 8) = u32x2{x.v[j][2], x.v[j][3]};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef T t4 __attribute__((ext_vector_type(4)));
                const t4 w = __builtin_bit_cast(t4, *reinterpret_cast<const u32x2*>(lds + r * kRowPad + (8 * g + 4 * h) * 2));
#pragma unroll
                for (int j = 0; j < 4; ++j) out[4 * g + j] = to_f32(w[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* d = reinterpret_cast<float*>(lds + (8 * j + (lane >> 3)) * kRowPad + 16 * (lane & 7));
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = __uint_as_float(x.v[j][e]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 16; ++i) out[i] = *reinterpret_cast<const float*>(lds + r * kRowPad + 4 * acc_row(i, h));
        }
        __builtin_amdgcn_wave_barrier();
    }

    // The reverse for dS: v[i] belongs to (lane axis r, register axis acc_row(i, h)).
    //   16-bit: written to `g` with rows = the LANE axis, columns = the register axis (4 consecutive register-axis elements
    //           of a lane are one 8-byte LDS write); 8-byte units XOR-swizzled per row (4-way instead of 8-way conflicts).
    //   fp32  : written with rows = the REGISTER axis, columns = the lane axis (conflict-free 4-byte LDS writes).
    // Either way two / four coalesced 16-byte global stores per lane.
    NNOP_DEV static void store(T* __restrict__ g, size_t ld, char* lds, int lane, const f32x16& v) {
        const int r = lane & 31, h = lane >> 5;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                typedef T t4 __attribute__((ext_vector_type(4)));
                const f32x4 w = {v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]};
                const u32x2 u = __builtin_bit_cast(u32x2, __builtin_convertvector(w, t4));
                const int unit = (2 * q4 + h) ^ (r & 7);         // columns 8 q4 + 4 h .. + 3
                *reinterpret_cast<u32x2*>(lds + r * 64 + unit * 8) = u;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 16 * j + (lane >> 2), c16 = lane & 3;
                const u32x2 a = *reinterpret_cast<const u32x2*>(lds + row * 64 + ((2 * c16) ^ (row & 7)) * 8);
                const u32x2 b = *reinterpret_cast<const u32x2*>(lds + row * 64 + ((2 * c16 + 1) ^ (row & 7)) * 8);
                *reinterpret_cast<u32x4*>(g + (size_t)row * ld + 8 * c16) = u32x4{a[0], a[1], b[0], b[1]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) *reinterpret_cast<float*>(lds + acc_row(i, h) * 128 + 4 * r) = v[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 8 * j + (lane >> 3), c = lane & 7;
                *reinterpret_cast<u32x4*>(g + (size_t)row * ld + 4 * c) = *reinterpret_cast<const u32x4*>(lds + row * 128 + 16 * c);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // orientation of the dS scratch matrix of one (batch, head): true = [QLp][KLp] (16-bit), false = [KLp][QLp] (fp32)
    static constexpr bool kStoreLaneMajor = sizeof(T) == 2;
};

struct PairPackParams {
    const void* pair;        // [B][KL][QL][QH]
    void* a;                 // [B][QH][KLp][QLp]
    int QL, KL, QH, B, QLp, KLp, causal;
};

// 16 bytes of T gathered from LDS at an element stride / scattered to it
template <typename T> NNOP_DEV u32x4 gather16(const T* base, int stride) {
    constexpr int N = 16 / (int)sizeof(T);
    typedef T tn __attribute__((ext_vector_type(N)));
    tn v;
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = base[j * stride];
    return __builtin_bit_cast(u32x4, v);
}
template <typename T> NNOP_DEV void scatter16(T* base, int stride, u32x4 w) {
    constexpr int N = 16 / (int)sizeof(T);
    typedef T tn __attribute__((ext_vector_type(N)));
    const tn v = __builtin_bit_cast(tn, w);
#pragma unroll
    for (int j = 0; j < N; ++j) base[j * stride] = v[j];
}

// One workgroup: a 32 (keys) x 32 (queries) x QH block of one batch, through LDS.  Global accesses are 16-byte vectors
// (element-wise only where a row of the block is not 16-byte aligned or sticks out of the tensor): |pair| read, 2 |pair| written.
template <typename T>
__global__ __launch_bounds__(256) void pair_pack_kernel(const PairPackParams p) {
    constexpr int N = 16 / (int)sizeof(T);                       // elements per 16-byte vector
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tile = reinterpret_cast<T*>(smem);                        // [32 k][32 q][QH]
    const int tid = threadIdx.x;
    const int nqb = p.QLp / 32, nkb = p.KLp / 32;
    int id = blockIdx.x;
    const int qb = id % nqb; id /= nqb;
    const int kb = id % nkb;
    const int b = id / nkb;
    const int k0 = kb * 32, q0 = qb * 32;
    // causally hidden block (every key behind every query): the kernels either skip it or mask each of its elements with a
    // select, so whatever the scratch holds there is never used
    if (p.causal && k0 > q0 + 31) return;
    const int run = 32 * p.QH;                                   // contiguous elements of one key row of the block
    const T* src = (const T*)p.pair + (((size_t)b * p.KL + k0) * p.QL + q0) * p.QH;
    const bool vec = ((p.QL * p.QH) % N) == 0 && (run % N) == 0 && q0 + 32 <= p.QL && ((uintptr_t)p.pair & 15) == 0;
    if (vec) {
        const int nv = run / N;
        for (int i = tid; i < 32 * nv; i += 256) {
            const int k = i / nv, c = i - k * nv;
            u32x4 v = {0, 0, 0, 0};
            if (k0 + k < p.KL) v = *reinterpret_cast<const u32x4*>(src + (size_t)k * p.QL * p.QH + c * N);
            *reinterpret_cast<u32x4*>(tile + k * run + c * N) = v;
        }
    } else {
        for (int i = tid; i < 32 * run; i += 256) {
            const int k = i / run, e = i - k * run;              // e = q * QH + h
            T v = from_f32<T>(0.f);
            if (k0 + k < p.KL && q0 + e / p.QH < p.QL) v = src[(size_t)k * p.QL * p.QH + e];
            tile[i] = v;
        }
    }
    __syncthreads();
    T* pa = (T*)p.a;
    // 16-byte chunks of the output: rows = keys, chunk = N queries (LDS stride QH); 32 / N chunks per row
    constexpr int CPR = 32 / N;
    for (int i = tid; i < p.QH * 32 * CPR; i += 256) {
        const int c = i % CPR, row = (i / CPR) & 31, hh = i / (CPR * 32);
        *reinterpret_cast<u32x4*>(pa + (((size_t)b * p.QH + hh) * p.KLp + k0 + row) * p.QLp + q0 + N * c) =
            gather16<T>(tile + (row * 32 + N * c) * p.QH + hh, p.QH);
    }
}

struct PairUnpackParams {
    void* dpair;             // [B][KL][QL][QH]
    const void* s;           // dS scratch, [B][QH] x (lane-major ? [QLp][KLp] : [KLp][QLp])
    const uint8_t* kpad;     // [B][KL] or null
    int QL, KL, QH, B, QLp, KLp, causal, lane_major;
};

template <typename T>
__global__ __launch_bounds__(256) void dpair_unpack_kernel(const PairUnpackParams p) {
    constexpr int N = 16 / (int)sizeof(T);
    constexpr int CPR = 32 / N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tile = reinterpret_cast<T*>(smem);                        // [32 k][32 q][QH]: the layout of the output block
    const int tid = threadIdx.x;
    const int nqb = p.QLp / 32, nkb = p.KLp / 32;
    int id = blockIdx.x;
    const int qb = id % nqb; id /= nqb;
    const int kb = id % nkb;
    const int b = id / nkb;
    const int k0 = kb * 32, q0 = qb * 32;
    const int run = 32 * p.QH;
    // tiles no kernel visited hold garbage: causally hidden (k > q for every element); elements are masked again below
    const bool dead_tile = p.causal && k0 > q0 + 31;
    const T* ps = (const T*)p.s;
    if (!dead_tile) {
        for (int i = tid; i < p.QH * 32 * CPR; i += 256) {
            const int c = i % CPR, row = (i / CPR) & 31, hh = i / (CPR * 32);
            const size_t base = ((size_t)b * p.QH + hh) * (size_t)p.QLp * p.KLp;
            if (p.lane_major) {      // row = query, chunk = N keys
                scatter16<T>(tile + (N * c * 32 + row) * p.QH + hh, run, *reinterpret_cast<const u32x4*>(ps + base + (size_t)(q0 + row) * p.KLp + k0 + N * c));
            } else {                 // row = key, chunk = N queries
                scatter16<T>(tile + (row * 32 + N * c) * p.QH + hh, p.QH, *reinterpret_cast<const u32x4*>(ps + base + (size_t)(k0 + row) * p.QLp + q0 + N * c));
            }
        }
    }
    __syncthreads();
    T* dst = (T*)p.dpair + (((size_t)b * p.KL + k0) * p.QL + q0) * p.QH;
    const uint8_t* mp = p.kpad ? p.kpad + (size_t)b * p.KL : nullptr;
    const bool vec = ((p.QL * p.QH) % N) == 0 && (run % N) == 0 && q0 + 32 <= p.QL && ((uintptr_t)p.dpair & 15) == 0 && (N % p.QH == 0 || p.QH % N == 0);
    if (vec) {
        const int nv = run / N;
        for (int i = tid; i < 32 * nv; i += 256) {
            const int k = i / nv, c = i - k * nv;
            if (k0 + k >= p.KL) continue;
            typedef T tn __attribute__((ext_vector_type(N)));
            tn v = *reinterpret_cast<const tn*>(tile + k * run + c * N);
            const bool krow_dead = dead_tile || (mp && mp[k0 + k] == 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int q = q0 + (c * N + j) / p.QH;
                if (krow_dead || (p.causal && k0 + k > q)) v[j] = from_f32<T>(0.f);
            }
            *reinterpret_cast<tn*>(dst + (size_t)k * p.QL * p.QH + c * N) = v;
        }
    } else {
        for (int i = tid; i < 32 * run; i += 256) {
            const int k = i / run, e = i - k * run, q = e / p.QH;
            if (k0 + k >= p.KL || q0 + q >= p.QL) continue;
            bool live = !dead_tile;
            if (p.causal && k0 + k > q0 + q) live = false;
            if (mp && mp[k0 + k] == 0) live = false;
            dst[(size_t)k * p.QL * p.QH + e] = live ? tile[i] : from_f32<T>(0.f);
        }
    }
}

}  // namespace nnop
