// Per-workgroup token-tile helpers shared by the forward / backward / weight-gradient tower kernels.
//
// A workgroup (512 threads = 8 waves, two per SIMD so that the VALU-heavy GELU / dropout epilogues of
// one wave overlap the MFMA and load latency of its partner) owns BM = 16 token rows: on the fused path SPW whole
// samples of N tokens (token mixing couples the N tokens of a sample, channel mixing is row-wise), on the wide
// path (kernels instantiated with NMAX == 0: channel mixing only) any 16 consecutive rows.  It keeps the fp32
// residual stream of those rows in LDS for the whole launch and streams the weights past it.
#pragma once
#include "common.h"
#include "../../include/m2mixer.h"
#include <stddef.h>
#include <string.h>

#ifndef BM
#define BM 16                      // token rows per workgroup of the chain kernels (16 or 32)
#endif
#define MT (BM / 16)
#define NTHREADS 512
#define NWAVES 8
#define TPR (NTHREADS / BM)        // threads per row in the row-wise (LayerNorm) phases: 32 (BM 16) / 16 (BM 32)
#define WPAIR 32                   // rows of a weight-gradient tile: the packed operand images are laid out per 32 rows

static __host__ __device__ __forceinline__ unsigned int m2m_mix32_hd(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// dropout stream key of one site: site = site_base + 4*block + {0 tok hidden,1 tok out,2 ch hidden,3 ch out}
static __host__ __device__ __forceinline__ unsigned int m2m_site_key(unsigned int seed, unsigned int step, unsigned int site) {
    return m2m_mix32_hd(seed ^ m2m_mix32_hd(step * 0x9E3779B9U + site * 0x85EBCA77U + 0x1234567U));
}
static __host__ __device__ __forceinline__ unsigned int m2m_drop_thr(float p) {
    // keep probability quantised to 16 bits; p == 0 -> 65536 (keep all)
    double keep = 1.0 - (double)p;
    long t = (long)(keep * 65536.0 + 0.5);
    if (t < 1) t = 1;
    if (t > 65536) t = 65536;
    return (unsigned int)t;
}
static __device__ __forceinline__ Drop make_drop(bool training, float p, unsigned int seed, unsigned int step, unsigned int site) {
    Drop d;
    d.thr = (training && p > 0.f) ? m2m_drop_thr(p) : 65536u;
    d.scale = 65536.0f / (float)d.thr;
    d.key = m2m_site_key(seed, step, site);
    return d;
}

// ---- dropout of the channel-mixing hidden activation (the hot site) -----------------------------------
// Element (token row m, hidden column c).  One 32-bit word per (m, c >> 5) when p == 0.5 (bit c & 31
// decides), otherwise one word per (m, c >> 1) with a 16-bit threshold compare.  Defined on (m, c) only,
// so forward, backward and the weight-gradient pass -- which hold the elements in different lane
// layouts -- regenerate identical masks.
static __device__ __forceinline__ unsigned int drop_word_half(const Drop& d, unsigned int m, unsigned int cgroup, unsigned int ngroups) {
    return mix32(d.key ^ (m * ngroups + cgroup));
}
static __device__ __forceinline__ bool drop_keep_mc(const Drop& d, unsigned int m, unsigned int c, unsigned int Cp) {
    if (d.thr == 32768u) return (drop_word_half(d, m, c >> 5, Cp >> 5) >> (c & 31)) & 1u;
    return drop_keep(d, m * Cp + c);
}

// Dropout mode of a launch (template parameter of the hot kernels, so the epilogues stay branch-free):
//   DM_NONE  eval / p == 0        DM_HALF  p == 0.5 (one bit per element)        DM_GEN  any other p (16-bit draws)
enum { DM_NONE = 0, DM_HALF = 1, DM_GEN = 2 };
static inline int m2m_drop_mode(int training, float p) {
    if (!training || p <= 0.f) return DM_NONE;
    return m2m_drop_thr(p) == 32768u ? DM_HALF : DM_GEN;
}
// Keep-bits of the ncols (<= 32) elements of a token-site row `bd` (one (sample, channel) column): bit t = keep.
template <int DM>
static __device__ __forceinline__ unsigned int drop_row_bits(const Drop& d, unsigned int bd, int ncols) {
    if (DM == DM_NONE) return 0xFFFFFFFFu;
    if (DM == DM_HALF) return mix32(d.key ^ bd);
    unsigned int bits = 0u;
    for (int t = 0; t < ncols; ++t) bits |= (drop_keep(d, bd * ncols + t) ? 1u : 0u) << t;
    return bits;
}
// Wide path (ncols may exceed 32): keep-bit of column idx of token-site row bd.  Agrees with drop_row_bits for
// ncols <= 32 (one word per row); longer rows take ceil(ncols / 32) consecutive words.
template <int DM>
static __device__ __forceinline__ bool drop_row_keep(const Drop& d, unsigned int bd, unsigned int ncols, unsigned int idx) {
    if (DM == DM_NONE) return true;
    if (DM == DM_HALF) {
        const unsigned int nw = (ncols + 31u) >> 5;
        return (mix32(d.key ^ (bd * nw + (idx >> 5))) >> (idx & 31u)) & 1u;
    }
    return drop_keep(d, bd * ncols + idx);
}
// Keep-bits of the 32 hidden columns [32 q, 32 q + 32) of token row m (channel-hidden site): bit c & 31.
template <int DM>
static __device__ __forceinline__ unsigned int drop_hidden_bits(const Drop& d, unsigned int m, unsigned int q, unsigned int Cp) {
    if (DM == DM_NONE) return 0xFFFFFFFFu;
    if (DM == DM_HALF) return drop_word_half(d, m, q, Cp >> 5);
    unsigned int bits = 0u;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {           // 16 hash words, two 16-bit draws each
        const unsigned int w = mix32(d.key ^ ((m * Cp + 32 * q) / 2 + j));
        bits |= (((w & 0xFFFFu) < d.thr) ? 1u : 0u) << (2 * j);
        bits |= (((w >> 16) < d.thr) ? 1u : 0u) << (2 * j + 1);
    }
    return bits;
}
// one element of the channel-output / generic sites
template <int DM>
static __device__ __forceinline__ bool drop_keep_elem(const Drop& d, unsigned int idx) {
    if (DM == DM_NONE) return true;
    return drop_keep(d, idx);
}

// Byte stride between the operand streams (h_chn / dh_chn) of consecutive column-tile groups (a pair of 16-column tiles in
// bf16, one tile in fp32): npair 2-KiB blocks plus a pad.  Without the pad the stride is a power of two (128 KiB at batch
// 512), every column slice of the weight-gradient launch walks its stream in lockstep, and all of them hit the same L2
// channel at the same time.
#ifndef M2M_HCHN_PAD
#define M2M_HCHN_PAD 2304
#endif
static __host__ __device__ __forceinline__ long m2m_hchn_stride(long npair) { return npair * 2048 + M2M_HCHN_PAD; }

// A tower descriptor with room for 4 blocks only: two of them (plus per-tower arguments) fit the 4 KiB kernel-argument
// limit, so a multi-tower launch can take its descriptors BY VALUE (pointers in kernel arguments are known to be global
// memory; descriptors read from device memory would turn every load into a FLAT access that also ticks lgkmcnt).
// Layout-identical prefix of m2m_tower: copy sizeof(m2m_tower4) bytes of a tower with nblocks <= 4.
#define M2M_GROUP_BLOCKS 4
struct m2m_tower4 {
    int32_t prec, D, N, T, C, Cp, nblocks, has_final_ln;
    float p_drop;
    uint32_t site_base;
    const float* lnf_w; const float* lnf_b; float* g_lnf_w; float* g_lnf_b; float* x_final; float* ws_a; float* ws_b;
    m2m_block blk[M2M_GROUP_BLOCKS];
};
static_assert(offsetof(m2m_tower4, blk) == offsetof(m2m_tower, blk), "m2m_tower4 must be a prefix of m2m_tower");
static inline m2m_tower4 m2m_shrink(const m2m_tower* t) { m2m_tower4 r; memcpy(&r, t, sizeof(r)); return r; }

// Which execution path a tower takes (see include/m2mixer.h): fused = whole samples per workgroup.
static inline bool m2m_is_wide(const m2m_tower* t) { return t->N > 8 || t->D > 128; }
// Form of the channel-mixing weight gradients of a tower at batch B (tower_wgrad.hip).  true: the weight-gradient launch
// recomputes the hidden activation from the packed image of A = LN2(x_mid) that the backward chain leaves in
// m2m_block.h_chn (tower_bwd_body<HREC>), and only dHpre^T is streamed; false: both hidden operands are stored and streamed.
bool m2m_wgrad_recompute(const m2m_tower* t, int B);

template <int D> struct TileGeom {
    static constexpr int XLD = D + 4;          // padded fp32 row stride (floats)
    static constexpr int DT = D / 16;
    static constexpr int EPT = D / TPR;        // elements per thread in row-wise phases
    static_assert(D % 32 == 0 && D >= 32, "hidden_dim must be a multiple of 32");
};

// column of element e (0..EPT-1) of row-thread j (0..TPR-1): float4 chunks interleaved over the TPR
// threads of a row (coalesced segments); fewer than 4 elements per thread -> EPT contiguous columns.
template <int D>
static __device__ __forceinline__ int ln_col(int e, int j) {
    constexpr int EPT = D / TPR;
    if (EPT >= 4) return 4 * TPR * (e >> 2) + 4 * j + (e & 3);
    return EPT * j + e;
}

// row statistics (mean, 1/std with biased variance, eps 1e-5) of one row held by TPR threads.
// src points at the row (LDS or global); invalid rows read as zeros.
template <int D>
static __device__ __forceinline__ void row_stats(const float* src, bool valid, int j, float v[D / TPR], float& mean, float& rstd) {
    constexpr int EPT = D / TPR;
    static_assert(EPT >= 1, "hidden_dim too small for this tile geometry");
    float s = 0.f;
    if (EPT >= 4) {
#pragma unroll
        for (int i = 0; i < EPT / 4; ++i) {
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) q = *reinterpret_cast<const float4*>(src + 4 * TPR * i + 4 * j);
            v[4 * i + 0] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
            s += (q.x + q.y) + (q.z + q.w);
        }
    } else if (EPT == 2) {
        float2 q = make_float2(0.f, 0.f);
        if (valid) q = *reinterpret_cast<const float2*>(src + 2 * j);
        v[0] = q.x; v[EPT - 1] = q.y;
        s = q.x + q.y;
    } else {
        v[0] = valid ? src[j] : 0.f;
        s = v[0];
    }
    s = wave_sum_xor(s, TPR);
    mean = s * (1.0f / D);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) { const float c = v[i] - mean; s2 = __builtin_fmaf(c, c, s2); }
    s2 = wave_sum_xor(s2, TPR);
    const float vv = s2 * (1.0f / D) + 1e-5f;
    rstd = __builtin_amdgcn_rsqf(vv);
    rstd = rstd * (1.5f - 0.5f * vv * rstd * rstd);     // one Newton step: full fp32 accuracy for the parity mode
}

// LayerNorm of every row of the fp32 LDS tile x into the fp32 tile dst (same geometry).
template <int D>
static __device__ __forceinline__ void ln_to_tile(const float* x, float* dst, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int tid) {
    constexpr int XLD = TileGeom<D>::XLD, EPT = D / TPR;
    const int r = tid / TPR, j = tid % TPR;
    float v[EPT], mean, rstd;
    row_stats<D>(x + r * XLD, true, j, v, mean, rstd);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = ln_col<D>(e, j);
        dst[r * XLD + c] = (v[e] - mean) * rstd * gamma[c] + beta[c];
    }
}

// LayerNorm backward for the whole tile.
//   xg      : global saved LN input, row r at xg + r*D (rows >= R treated as zero)
//   up      : LDS tile, upstream gradient wrt the LN output                       [BM][XLD]
//   dxs     : LDS tile, receives (accumulate ? += : =) the gradient wrt the LN input
//   prod    : LDS tile, receives up * xhat (for the gamma gradient)
// then column sums -> atomicAdd into g_w (gamma) / g_b (beta).  Contains two __syncthreads().
// ATOMIC == false: the column sums are STORED to g_w / g_b (a per-workgroup partial-sum slot) instead of added atomically.
template <int D, bool ATOMIC = true>
static __device__ __forceinline__ void ln_backward_tile(const float* xg, int R, const float* up, const float* __restrict__ gamma,
                                                        float* dxs, bool accumulate, float* prod, float* g_w, float* g_b,
                                                        int tid) {
    constexpr int XLD = TileGeom<D>::XLD, EPT = D / TPR;
    const int r = tid / TPR, j = tid % TPR;
    const bool valid = r < R;
    float v[EPT], mean, rstd;
    row_stats<D>(xg + (long)r * D, valid, j, v, mean, rstd);
    float gsum = 0.f, gxsum = 0.f;
    float gv[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = ln_col<D>(e, j);
        const float xh = (v[e] - mean) * rstd;
        const float u = up[r * XLD + c];
        const float gg = u * gamma[c];
        gv[e] = gg;
        v[e] = xh;
        gsum += gg;
        gxsum = __builtin_fmaf(gg, xh, gxsum);
        prod[r * XLD + c] = valid ? u * xh : 0.f;
    }
    gsum = wave_sum_xor(gsum, TPR) * (1.0f / D);
    gxsum = wave_sum_xor(gxsum, TPR) * (1.0f / D);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int c = ln_col<D>(e, j);
        const float dx = rstd * (gv[e] - gsum - v[e] * gxsum);
        if (accumulate) { if (valid) dxs[r * XLD + c] += dx; }
        else dxs[r * XLD + c] = valid ? dx : 0.f;
    }
    __syncthreads();
    _Pragma("unroll 1") for (int d = tid; d < 2 * D; d += NTHREADS) {
        const float* src = d < D ? prod : up;
        const int c = d < D ? d : d - D;
        float s = 0.f;
        for (int rr = 0; rr < R; ++rr) s += src[rr * XLD + c];
        if (ATOMIC) atomicAdd((d < D ? g_w : g_b) + c, s);
        else (d < D ? g_w : g_b)[c] = s;
    }
    __syncthreads();
}

// Build the 16 bytes of one lane slot of a packed block from an fp32 tile in LDS.
//   transposed == false : X[i][k] = tile[i][k]      transposed == true : X[i][k] = tile[k][i]
template <int P>
static __device__ __forceinline__ u32x4_t gather_slot(const float* tile, int xld, int mode, bool transposed, int ib, int kb, int lane) {
    typedef Prec<P> Pr;
    const int i = ib * 16 + (lane & 15), g = lane >> 4;
    float v[Pr::EPL];
#pragma unroll
    for (int e = 0; e < Pr::EPL; ++e) {
        const int k = kb * Pr::KB + Pr::kmap(mode, g, e);
        v[e] = transposed ? tile[k * xld + i] : tile[i * xld + k];
    }
    Frag f;
    if (P == PREC_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) f.u[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f.f[e] = v[e];
    }
    return f.u;
}

// fp32 tile [BM][XLD] -> packed NAT image X[i = m][k = d], blocks ordered [mt][kb]  (dst: LDS or global)
template <int P, int D>
static __device__ __forceinline__ void pack_tile_nat(const float* tile, char* img, int tid) {
    constexpr int KD = D / Prec<P>::KB;
    _Pragma("unroll 1") for (int slot = tid; slot < MT * KD * 64; slot += NTHREADS) {
        const int blk = slot >> 6;
        *reinterpret_cast<u32x4_t*>(img + slot * 16) =
            gather_slot<P>(tile, TileGeom<D>::XLD, PACK_NAT, false, blk / KD, blk % KD, slot & 63);
    }
}
// fp32 tile [BM][XLD] -> this tile's part of the packed CHN image of the TRANSPOSE, X[i = d][k = m], m running
// over a 32-row PAIR of tiles (blocks ordered [kb over the 32 rows][dt]); `pair_img` points at the pair's image,
// `tile_in_pair` = 0/1 says which 16 rows this workgroup owns when BM == 16.
//   bf16 (k-block = 32 rows): BM == 16 fills one 8-byte half of every 16-byte lane slot (elements 4h .. 4h+3).  These
//        images are small (D x 32 elements per pair): the half-slot writes cost nothing measurable, and the 16-byte slot
//        layout lets the weight-gradient kernel stage a block with plain conflict-free 16-byte LDS writes.
//   fp32 (k-block = 16 rows): BM == 16 fills k-block `tile_in_pair` whole.
template <int P, int D>
static __device__ __forceinline__ void pack_tile_chn_t(const float* tile, char* pair_img, int tile_in_pair, int tid) {
    typedef Prec<P> Pr;
    constexpr int DT = D / 16, XLD = TileGeom<D>::XLD;
    if (BM >= Pr::KB) {                      // whole k-blocks
        constexpr int NKM = BM / Pr::KB > 0 ? BM / Pr::KB : 1;
        const int kb0 = tile_in_pair * NKM;
        _Pragma("unroll 1") for (int slot = tid; slot < NKM * DT * 64; slot += NTHREADS) {
            const int blk = slot >> 6;
            *reinterpret_cast<u32x4_t*>(pair_img + (long)kb0 * DT * 1024 + slot * 16) =
                gather_slot<P>(tile, XLD, PACK_CHN, true, blk % DT, blk / DT, slot & 63);
        }
    } else {                                 // bf16, 16 rows: half slots
        _Pragma("unroll 1") for (int slot = tid; slot < DT * 64; slot += NTHREADS) {
            const int dt = slot >> 6, lane = slot & 63, i = dt * 16 + (lane & 15), g = lane >> 4;
            const float v0 = tile[(4 * g + 0) * XLD + i], v1 = tile[(4 * g + 1) * XLD + i];
            const float v2 = tile[(4 * g + 2) * XLD + i], v3 = tile[(4 * g + 3) * XLD + i];
            uint2 o;
            o.x = pack_bf2(v0, v1);
            o.y = pack_bf2(v2, v3);
            *reinterpret_cast<uint2*>(pair_img + slot * 16 + tile_in_pair * 8) = o;
        }
    }
}

// accumulator tiles -> chained operand fragment(s)
template <int P> struct Chain;
template <> struct Chain<PREC_BF16> {
    static constexpr int NF = 1;   // fragments per pair of 16-row accumulator tiles
    static __device__ __forceinline__ void make(const f32x4_t& t0, const f32x4_t& t1, Frag out[1]) {
        out[0].u[0] = pack_bf2(t0[0], t0[1]);
        out[0].u[1] = pack_bf2(t0[2], t0[3]);
        out[0].u[2] = pack_bf2(t1[0], t1[1]);
        out[0].u[3] = pack_bf2(t1[2], t1[3]);
    }
};
template <> struct Chain<PREC_F32> {
    static constexpr int NF = 2;
    static __device__ __forceinline__ void make(const f32x4_t& t0, const f32x4_t& t1, Frag out[2]) {
        out[0].f = t0;
        out[1].f = t1;
    }
};

// Sum the NWAVES per-wave partial [BM][D] accumulator sets through four LDS slabs, deterministically:
// waves 0-3 store, waves 4-7 add, then the caller reads slab0 + slab1 + slab2 + slab3.
// acc[mt][dt][r] = element (row 16 mt + 4 g + r, column 16 dt + il).  The slabs are TRANSPOSED,
// [D][SLD] (column-major tiles), so the four rows a lane holds go out as one 16-byte access.
#define SLD (BM + 4)
template <int D> struct SlabGeom { static constexpr int FLOATS = D * SLD; };
template <int D>
static __device__ __forceinline__ void reduce_waves_to_slabs(f32x4_t (&acc)[MT][D / 16], float* slabs, int wave, int g, int il) {
    constexpr int DT = D / 16;
    float* my = slabs + (wave & 3) * SlabGeom<D>::FLOATS;
    if (wave < 4) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                *reinterpret_cast<f32x4_t*>(my + (dt * 16 + il) * SLD + mt * 16 + 4 * g) = acc[mt][dt];
    }
    __syncthreads();
    if (wave >= 4) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x4_t* p = reinterpret_cast<f32x4_t*>(my + (dt * 16 + il) * SLD + mt * 16 + 4 * g);
                *p = *p + acc[mt][dt];
            }
    }
    __syncthreads();
}
// value of element (row r, column d) after reduce_waves_to_slabs
template <int D>
static __device__ __forceinline__ float slab_sum(const float* slabs, int r, int d) {
    const float* s = slabs + d * SLD + r;
    constexpr int F = SlabGeom<D>::FLOATS;
    return (s[0] + s[F]) + (s[2 * F] + s[3 * F]);
}

// ---- row-wise phases on registers (round 2) ---------------------------------------------------------------------
// Thread (r = tid / TPR, j = tid % TPR) owns the EPT = D / TPR elements ln_col<D>(e, j) of row r.  The helpers below keep a
// phase's values in registers from the LDS / global read to the last write, so that neighbouring phases (sum of the waves'
// partial results -> residual -> LayerNorm -> packed operand image) need no LDS round trip and no barrier between them.
template <int D>
static __device__ __forceinline__ void ld_row(const float* row, int j, float v[D / TPR]) {
    constexpr int EPT = D / TPR;
    if constexpr (EPT >= 4) {
#pragma unroll
        for (int i = 0; i < EPT / 4; ++i) {
            const f32x4_t q = *reinterpret_cast<const f32x4_t*>(row + 4 * TPR * i + 4 * j);
            v[4 * i + 0] = q[0]; v[4 * i + 1] = q[1]; v[4 * i + 2] = q[2]; v[4 * i + 3] = q[3];
        }
    } else if constexpr (EPT == 2) {
        const float2 q = *reinterpret_cast<const float2*>(row + 2 * j);
        v[0] = q.x; v[1] = q.y;
    } else {
        v[0] = row[j];
    }
}
template <int D>
static __device__ __forceinline__ void st_row(float* row, int j, const float v[D / TPR]) {
    constexpr int EPT = D / TPR;
    if constexpr (EPT >= 4) {
#pragma unroll
        for (int i = 0; i < EPT / 4; ++i)
            *reinterpret_cast<f32x4_t*>(row + 4 * TPR * i + 4 * j) = f32x4_t{v[4 * i + 0], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    } else if constexpr (EPT == 2) {
        *reinterpret_cast<float2*>(row + 2 * j) = make_float2(v[0], v[1]);
    } else {
        row[j] = v[0];
    }
}
// mean and 1 / std (biased variance, eps 1e-5) of a row held by its TPR threads
template <int D>
static __device__ __forceinline__ void reg_stats(const float v[D / TPR], float& mean, float& rstd) {
    constexpr int EPT = D / TPR;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) s += v[i];
    s = wave_sum_xor(s, TPR);
    mean = s * (1.0f / D);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) { const float c = v[i] - mean; s2 = __builtin_fmaf(c, c, s2); }
    s2 = wave_sum_xor(s2, TPR);
    const float vv = s2 * (1.0f / D) + 1e-5f;
    rstd = __builtin_amdgcn_rsqf(vv);
    rstd = rstd * (1.5f - 0.5f * vv * rstd * rstd);
}
// this thread's elements of row r -> packed NAT operand image X[i = row][k = column] (blocks [mt][kb], see pack_tile_nat)
template <int P, int D>
static __device__ __forceinline__ void pack_row_nat(char* img, int r, int j, const float y[D / TPR]) {
    typedef Prec<P> Pr;
    constexpr int EPT = D / TPR, KD = D / Pr::KB;
    const int mt = r >> 4, il = r & 15;
    if constexpr (P == PREC_BF16 && EPT >= 4) {
#pragma unroll
        for (int i = 0; i < EPT / 4; ++i) {
            const int c = 4 * TPR * i + 4 * j, kb = c >> 5, kk = c & 31;          // 4 consecutive k inside one lane slot
            uint2 o;
            o.x = pack_bf2(y[4 * i + 0], y[4 * i + 1]);
            o.y = pack_bf2(y[4 * i + 2], y[4 * i + 3]);
            *reinterpret_cast<uint2*>(img + (((mt * KD + kb) * 64 + (kk >> 3) * 16 + il) * 16 + (kk & 7) * 2)) = o;
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int c = ln_col<D>(e, j), kb = c / Pr::KB, kk = c % Pr::KB;
            if constexpr (P == PREC_BF16) {
                *reinterpret_cast<unsigned short*>(img + (((mt * KD + kb) * 64 + (kk >> 3) * 16 + il) * 16 + (kk & 7) * 2)) = f2bf(y[e]);
            } else {
                *reinterpret_cast<float*>(img + (((mt * KD + kb) * 64 + (kk & 3) * 16 + il) * 16 + (kk >> 2) * 4)) = y[e];
            }
        }
    }
}
// accumulator set of a wave -> its row-major fp32 slab [BM][XLD]: element (row 16 mt + 4 g + r, column 16 dt + il)
template <int D>
static __device__ __forceinline__ void acc_to_slab(const f32x4_t (&acc)[MT][D / 16], float* slab, int g, int il) {
    constexpr int XLD = TileGeom<D>::XLD;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(mt * 16 + 4 * g + r) * XLD + dt * 16 + il] = acc[mt][dt][r];
}
template <int D>
static __device__ __forceinline__ void acc_add_slab(const f32x4_t (&acc)[MT][D / 16], float* slab, int g, int il) {
    constexpr int XLD = TileGeom<D>::XLD;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(mt * 16 + 4 * g + r) * XLD + dt * 16 + il] += acc[mt][dt][r];
}
// number of row-major slabs the chain kernels keep: one per wave where LDS allows it (hidden_dim <= 128), else four (waves
// 4-7 add into the slabs of waves 0-3 behind a barrier)
template <int D> struct RowSlabs { static constexpr int N = D <= 128 ? NWAVES : 4; };
// this thread's elements of row r, summed over the slabs
template <int D>
static __device__ __forceinline__ void slab_row_sum(const float* slabs, int r, int j, float v[D / TPR]) {
    constexpr int EPT = D / TPR, XLD = TileGeom<D>::XLD;
    ld_row<D>(slabs + r * XLD, j, v);
#pragma unroll
    for (int w = 1; w < RowSlabs<D>::N; ++w) {
        float u[EPT];
        ld_row<D>(slabs + (w * BM + r) * XLD, j, u);
#pragma unroll
        for (int e = 0; e < EPT; ++e) v[e] += u[e];
    }
}
