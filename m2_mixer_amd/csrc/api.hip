// Library plumbing, operand packing, Adam, and the small probe kernels used by the tests.
#include "tile.h"
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <type_traits>

static thread_local char g_err[512] = "";

void m2m_set_error(const char* msg, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s (%s:%d)", msg, file, line);
}

extern "C" const char* m2m_last_error(void) { return g_err; }
extern "C" int m2m_abi_version(void) { return M2M_ABI_VERSION; }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

extern "C" int64_t m2m_packed_bytes(int prec, int64_t I, int64_t K) {
    const int64_t kb = prec == PREC_BF16 ? 32 : 16;
    return ceil_div(I, 16) * ceil_div(K, kb) * 1024;
}

int m2m_check_tower(const m2m_tower* t, int B) {
    if (!t) { m2m_set_error("null tower", __FILE__, __LINE__); return -1; }
    if (t->prec != PREC_BF16 && t->prec != PREC_F32) { m2m_set_error("bad prec", __FILE__, __LINE__); return -1; }
    if (t->nblocks < 0 || t->nblocks > M2M_MAX_BLOCKS) { m2m_set_error("nblocks out of range", __FILE__, __LINE__); return -1; }
    if (t->D != 32 && t->D != 64 && t->D != 128 && t->D != 256) { m2m_set_error("hidden_dim D must be 32, 64, 128 or 256 in this build", __FILE__, __LINE__); return -1; }
    if (t->N < 1 || t->N > 128) { m2m_set_error("num_patch N must be in [1, 128] in this build", __FILE__, __LINE__); return -1; }
    if (t->T < 1 || t->T > 32) { m2m_set_error("token_dim T must be in [1, 32] in this build", __FILE__, __LINE__); return -1; }
    if (!m2m_is_wide(t) && (t->T % 8) != 0) { m2m_set_error("token_dim T must be a multiple of 8 on the fused path (N <= 8)", __FILE__, __LINE__); return -1; }
    if (t->Cp % 32 != 0 || t->Cp < t->C || t->C < 1) { m2m_set_error("Cp must be C rounded up to a multiple of 32", __FILE__, __LINE__); return -1; }
    if (B < 1) { m2m_set_error("B < 1", __FILE__, __LINE__); return -1; }
    if ((int64_t)B * t->N * (int64_t)t->Cp >= (1LL << 32)) { m2m_set_error("B*N*Cp exceeds the 32-bit dropout counter", __FILE__, __LINE__); return -1; }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// packing: one thread per 16-byte lane slot
// ---------------------------------------------------------------------------------------------------
template <int P>
__global__ void pack_kernel(int mode, int order_k_major, const float* __restrict__ src, long stride_i, long stride_k,
                            long I, long K, char* __restrict__ dst, long nIB, long nKB) {
    typedef Prec<P> Pr;
    const long slot = (long)blockIdx.x * blockDim.x + threadIdx.x;   // global lane slot
    const long nslots = nIB * nKB * 64;
    if (slot >= nslots) return;
    const long blk = slot >> 6;
    const int lane = (int)(slot & 63), g = lane >> 4, il = lane & 15;
    long ib, kb;
    if (order_k_major) { kb = blk / nIB; ib = blk % nIB; } else { ib = blk / nKB; kb = blk % nKB; }
    const long i = ib * 16 + il;
    Frag f;
    f.u = u32x4_t{0u, 0u, 0u, 0u};
    float v[8];
#pragma unroll
    for (int e = 0; e < Pr::EPL; ++e) {
        const long k = kb * Pr::KB + Pr::kmap(mode, g, e);
        v[e] = (i < I && k < K) ? src[i * stride_i + k * stride_k] : 0.f;
    }
    if (P == PREC_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) f.u[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f.f[e] = v[e];
    }
    *reinterpret_cast<u32x4_t*>(dst + slot * 16) = f.u;
}

// I, K: valid extents (reads are guarded); Ip, Kp: extents of the zero-padded image
static int pack_impl(int prec, int mode, int order_k_major, const float* src, int64_t stride_i, int64_t stride_k,
                     int64_t I, int64_t K, int64_t Ip, int64_t Kp, void* dst, void* stream) {
    if (prec != PREC_BF16 && prec != PREC_F32) { m2m_set_error("bad prec", __FILE__, __LINE__); return -1; }
    const long KB = prec == PREC_BF16 ? 32 : 16;
    const long nIB = ceil_div(Ip, 16), nKB = ceil_div(Kp, KB);
    const long nslots = nIB * nKB * 64;
    const int threads = 256;
    const long grid = ceil_div(nslots, threads);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (prec == PREC_BF16)
        hipLaunchKernelGGL(pack_kernel<PREC_BF16>, dim3((unsigned)grid), dim3(threads), 0, st, mode, order_k_major, src,
                           (long)stride_i, (long)stride_k, (long)I, (long)K, (char*)dst, nIB, nKB);
    else
        hipLaunchKernelGGL(pack_kernel<PREC_F32>, dim3((unsigned)grid), dim3(threads), 0, st, mode, order_k_major, src,
                           (long)stride_i, (long)stride_k, (long)I, (long)K, (char*)dst, nIB, nKB);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int m2m_pack(int prec, int mode, int order_k_major, const float* src, int64_t stride_i, int64_t stride_k,
                        int64_t I, int64_t K, void* dst, void* stream) {
    return pack_impl(prec, mode, order_k_major, src, stride_i, stride_k, I, K, I, K, dst, stream);
}

// One packed 16-byte slot: logical operand X[i][k] = src[i * si + kk * sk], i < I, kk < K, image padded to (Ip, Kp);
// same layouts as pack_impl.
template <int P>
__device__ __forceinline__ void pack_slot(const float* src, long si, long sk, long I, long K, long Ip, long Kp, int mode,
                                          int kmajor, char* dst, long slot) {
    typedef Prec<P> Pr;
    const long nIB = Ip / 16, nKB = Kp / Pr::KB;
    if (slot >= nIB * nKB * 64) return;
    const long blk = slot >> 6;
    const int lane = (int)(slot & 63), g = lane >> 4, il = lane & 15;
    long ib, kb;
    if (kmajor) { kb = blk / nIB; ib = blk % nIB; } else { ib = blk / nKB; kb = blk % nKB; }
    const long i = ib * 16 + il;
    float v[8];
#pragma unroll
    for (int e = 0; e < Pr::EPL; ++e) {
        const long kk = kb * Pr::KB + Pr::kmap(mode, g, e);
        v[e] = (i < I && kk < K) ? src[i * si + kk * sk] : 0.f;
    }
    Frag f;
    if (P == PREC_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) f.u[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f.f[e] = v[e];
    }
    *reinterpret_cast<u32x4_t*>(dst + slot * 16) = f.u;
}

// Packed copies of one tower block.  which: 0 w1n, 1 w1tc, 2 w2c, 3 w2tn, 4 ch_b1p.
template <int P, class TW>
__device__ __forceinline__ void pack_block_job(const TW& tw, int block, int which, long slot) {
    const m2m_block& k = tw.blk[block];
    const long D = tw.D, C = tw.C, Cp = tw.Cp;
    if (which == 4) {
        if (slot < Cp) k.ch_b1p[slot] = slot < C ? k.ch_b1[slot] : 0.f;
    } else if (which == 0) pack_slot<P>(k.ch_w1, D, 1, C, D, Cp, D, PACK_NAT, 0, (char*)k.w1n, slot);
    else if (which == 1)   pack_slot<P>(k.ch_w1, 1, D, D, C, D, Cp, PACK_CHN, 1, (char*)k.w1tc, slot);
    else if (which == 2)   pack_slot<P>(k.ch_w2, C, 1, D, C, D, Cp, PACK_CHN, 1, (char*)k.w2c, slot);
    else                   pack_slot<P>(k.ch_w2, 1, C, C, D, Cp, D, PACK_NAT, 0, (char*)k.w2tn, slot);
}

// All packed copies of every block of a tower in ONE launch: blockIdx.y = 5 * block + which.
template <int P>
__global__ void pack_tower_kernel(const m2m_tower tw) {
    pack_block_job<P>(tw, blockIdx.y / 5, blockIdx.y % 5, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

extern "C" int m2m_pack_tower(const m2m_tower* t, void* stream) {
    if (int rc = m2m_check_tower(t, 1)) return rc;
    if (t->nblocks == 0) return 0;
    const long KB = t->prec == PREC_BF16 ? 32 : 16;
    const long nslots = (long)(t->Cp / 16) * (t->D / KB) * 64;
    const long need = nslots > t->Cp ? nslots : t->Cp;
    const dim3 grid((unsigned)ceil_div(need, 256), (unsigned)(5 * t->nblocks));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (t->prec == PREC_BF16) hipLaunchKernelGGL(pack_tower_kernel<PREC_BF16>, grid, dim3(256), 0, st, *t);
    else hipLaunchKernelGGL(pack_tower_kernel<PREC_F32>, grid, dim3(256), 0, st, *t);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

// Every packed copy a model needs after an optimizer step -- up to three towers of <= 4 blocks and two patch
// embeddings -- in ONE launch (the whole repack is ~100 MB of HBM traffic: one launch at the memory roofline instead of
// five launches forked over side streams).  Grid layout: see pack_all_kernel.
#define M2M_PACK_TOWERS 3
#define M2M_PACK_EMBEDS 2
#ifndef M2M_W1TC_SKIP
#define M2M_W1TC_SKIP 1
#endif
#ifndef M2M_PACK_NT_DEFAULT
#define M2M_PACK_NT_DEFAULT 0
#endif
struct PackAllArgs {
    m2m_tower4 tw[M2M_PACK_TOWERS];
    m2m_embed em[M2M_PACK_EMBEDS];
    int nt, ne;
    int tile_end[M2M_PACK_TOWERS];     // running count of (block, 32-column group) tiles up to and including tower t
    int embed_wgs0;                    // workgroups (256 slots each) of embedding 0
    int nt_loads;                      // 1: the fp32 masters are read with non-temporal loads (M2M_PACK_NT)
    int rowtiles[M2M_PACK_TOWERS];     // m2m_adam_pack_all, bf16: > 0 = W2 in 8-row x AP_W-column tiles, this many column chunks per row group
    int skip_w1tc[M2M_PACK_TOWERS];    // 1: nothing reads this tower's w1tc copy (pack_skips_w1tc): a quarter of the re-pack's writes
};
// The W1^T (CHN) copy feeds the third product of the backward chain -- except in the bf16 / hidden_dim 128 instantiation, which
// takes that operand from the W1 fragments it parks in LDS (tower_bwd.hip, W1LDS); the only other reader is the column-split
// path, which needs the tower's slab buffer.  Such towers skip the copy in the whole-model re-pack (8.4 MB of 100 MB on
// M2-Mixer-B).  m2m_pack_tower / m2m_pack (per-tower, tests, the module path) always write all copies.
static inline int pack_skips_w1tc(const m2m_tower* t) {
    return t->prec == PREC_BF16 && t->D == 128 && !m2m_is_wide(t) && t->slabs == nullptr && M2M_W1TC_SKIP;
}
extern "C" int m2m_pack_skips_w1tc(const m2m_tower* t) { return t ? pack_skips_w1tc(t) : 0; }
static_assert(sizeof(PackAllArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// One workgroup = one 32-column group q of one block: W1 rows [32q, 32q + 32) (one contiguous 32 x D chunk) and W2 columns
// [32q, 32q + 32) are read ONCE, coalesced, into LDS and all four packed copies (w1n, w1tc, w2c, w2tn) plus ch_b1p are
// written from there.  (The slot-per-thread kernels above gather every master element twice, the transposed copies with
// 4-byte loads in 64-byte segments: 28 us for the whole model against ~100 MB of unavoidable traffic.)
// Second half of a tile workgroup: the four packed copies (+ ch_b1p is written by the caller) from the LDS tiles
//   t1 [32][D + 1] = W1[32q + r][d],  t2 [D][33] = W2[d][32q + j]   (rows / columns past C are zero)
template <int P, bool DO1 = true, bool DO2 = true>
static __device__ __forceinline__ void pack_emit_tile(const m2m_block& k, int D, int q, const float* t1, const float* t2, bool skip_w1tc = false) {
    typedef Prec<P> Pr;
    const int L1 = D + 1, L2 = 33;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nKB = D / Pr::KB, nIB = D / 16, CB = 32 / Pr::KB;      // k-blocks along d; 16-row blocks along d; c k-blocks per tile
    auto emit = [&](char* dst, long blk, int lane, const float (&v)[8]) {
        Frag f;
        if (P == PREC_BF16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) f.u[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) f.f[e] = v[e];
        }
        *reinterpret_cast<u32x4_t*>(dst + (blk * 64 + lane) * 16) = f.u;
    };
    // NAT copies, X[i = c][k = d]: blocks (ib = 2q + h, kb); w1n from t1[c][d], w2tn from t2[d][c]
    for (int s = tid; s < 2 * nKB * 64; s += nthr) {
        const int lane = s & 63, bl = s >> 6, h = bl / nKB, kb = bl % nKB, g = lane >> 4, il = lane & 15;
        const int r = 16 * h + il;
        float v1[8], v2[8];
#pragma unroll
        for (int e = 0; e < Pr::EPL; ++e) {
            const int d = kb * Pr::KB + Pr::kmap(PACK_NAT, g, e);
            v1[e] = DO1 ? t1[r * L1 + d] : 0.f;
            v2[e] = DO2 ? t2[d * L2 + r] : 0.f;
        }
        const long blk = (long)(2 * q + h) * nKB + kb;
        if (DO1) emit((char*)k.w1n, blk, lane, v1);
        if (DO2) emit((char*)k.w2tn, blk, lane, v2);
    }
    // CHN copies, k-major, X[i = d][k = c]: blocks (kb = CB q + h, ib); w1tc from t1[c][d], w2c from t2[d][c]
    for (int s = tid; s < CB * nIB * 64; s += nthr) {
        const int lane = s & 63, bl = s >> 6, h = bl / nIB, ib = bl % nIB, g = lane >> 4, il = lane & 15;
        const int d = 16 * ib + il;
        float v1[8], v2[8];
#pragma unroll
        for (int e = 0; e < Pr::EPL; ++e) {
            const int j = h * Pr::KB + Pr::kmap(PACK_CHN, g, e);
            v1[e] = DO1 ? t1[j * L1 + d] : 0.f;
            v2[e] = DO2 ? t2[d * L2 + j] : 0.f;
        }
        const long blk = (long)(CB * q + h) * nIB + ib;
        if (DO1 && !skip_w1tc) emit((char*)k.w1tc, blk, lane, v1);
        if (DO2) emit((char*)k.w2c, blk, lane, v2);
    }
}

template <int P, class TW>
static __device__ __forceinline__ void pack_block_tile(const TW& tw, int block, int q, char* smem, bool skip_w1tc = false, bool nt_loads = false) {
    const m2m_block& k = tw.blk[block];
    const int D = tw.D, C = tw.C, L1 = D + 1, L2 = 33;
    float* t1 = reinterpret_cast<float*>(smem);            // [32][D + 1]   W1[32q + r][d]
    float* t2 = t1 + 32 * L1;                               // [D][33]       W2[d][32q + j]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int c0 = 32 * q;
    for (int idx = tid; idx < 32 * (D / 4); idx += nthr) {
        const int r = idx / (D / 4), d4 = (idx % (D / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + r < C) {
            // nt_loads (workgroup-uniform): the masters are read once per step -- past the memory-side cache (see adam_kernel)
            const f32x4_t* src = reinterpret_cast<const f32x4_t*>(k.ch_w1 + (long)(c0 + r) * D + d4);
            const f32x4_t x = nt_loads ? __builtin_nontemporal_load(src) : *src;
            v = make_float4(x[0], x[1], x[2], x[3]);
        }
        float* o = t1 + r * L1 + d4;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    for (int idx = tid; idx < D * 32; idx += nthr) {
        const int d = idx >> 5, j = idx & 31;
        const float* src = k.ch_w2 + (long)d * C + c0 + j;
        t2[d * L2 + j] = c0 + j < C ? (nt_loads ? __builtin_nontemporal_load(src) : *src) : 0.f;
    }
    if (tid < 32) k.ch_b1p[c0 + tid] = c0 + tid < C ? k.ch_b1[c0 + tid] : 0.f;
    __syncthreads();
    pack_emit_tile<P>(k, D, q, t1, t2, skip_w1tc);
}

// blockIdx.x: the towers' (block, column group) tiles first -- tower t owns tile_end[t - 1] .. tile_end[t] -- then the
// embeddings' slots, 256 per workgroup.
template <int P>
__global__ __launch_bounds__(256) void pack_all_kernel(const PackAllArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int id = blockIdx.x;
    if (id < a.tile_end[M2M_PACK_TOWERS - 1]) {
        int t = 0;
        while (id >= a.tile_end[t]) ++t;
        if (t) id -= a.tile_end[t - 1];
        const int nq = a.tw[t].Cp >> 5;
        pack_block_tile<P>(a.tw[t], id / nq, id % nq, smem, a.skip_w1tc[t] != 0, a.nt_loads != 0);
        return;
    }
    id -= a.tile_end[M2M_PACK_TOWERS - 1];
    const int e = id < a.embed_wgs0 ? 0 : 1;
    if (e) id -= a.embed_wgs0;
    const m2m_embed& em = a.em[e];
    const long slot = (long)id * 256 + threadIdx.x;
    if constexpr (P == PREC_BF16) {
        // a slot is eight consecutive k of one row: two 16-byte loads when the row length / alignment allow it -- the four k-groups of
        // a row then share a 128-byte line per instruction; the generic gather below touches 64 lines per 4-byte load instruction
        const long nKB = em.Kp / 32, blk = slot >> 6;
        const int lane = (int)(slot & 63), g = lane >> 4, il = lane & 15;
        const long i = (blk / nKB) * 16 + il, k0 = (blk % nKB) * 32 + 8 * g;
        if (slot < (long)(em.D / 16) * nKB * 64 && (em.K & 3) == 0 && (reinterpret_cast<uintptr_t>(em.w) & 15) == 0 && i < em.D && k0 + 8 <= em.K) {
            const f32x4_t* src = reinterpret_cast<const f32x4_t*>(em.w + i * em.K + k0);
            const f32x4_t x0 = src[0], x1 = src[1];
            *reinterpret_cast<u32x4_t*>((char*)em.wn + slot * 16) =
                u32x4_t{pack_bf2(x0[0], x0[1]), pack_bf2(x0[2], x0[3]), pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3])};
            return;
        }
    }
    pack_slot<P>(em.w, em.K, 1, em.D, em.K, em.D, em.Kp, PACK_NAT, 0, (char*)em.wn, slot);
}

extern "C" int m2m_pack_all(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds,
                            void* stream) {
    if (ntowers < 0 || ntowers > M2M_PACK_TOWERS || nembeds < 0 || nembeds > M2M_PACK_EMBEDS || (ntowers && !towers) ||
        (nembeds && !embeds) || ntowers + nembeds == 0) {
        m2m_set_error("pack_all: up to 3 towers and 2 embeddings", __FILE__, __LINE__);
        return -1;
    }
    PackAllArgs a;
    memset(&a, 0, sizeof(a));
    a.nt = ntowers; a.ne = nembeds;
    static const int pack_nt = [] { const char* e = getenv("M2M_PACK_NT"); return e ? atoi(e) : M2M_PACK_NT_DEFAULT; }();
    a.nt_loads = pack_nt;
    int prec = -1, tiles = 0, maxD = 0;
    for (int i = 0; i < M2M_PACK_TOWERS; ++i) {
        if (i < ntowers) {
            if (int rc = m2m_check_tower(towers[i], 1)) return rc;
            if (towers[i]->nblocks > M2M_GROUP_BLOCKS) { m2m_set_error("pack_all: towers of <= 4 blocks", __FILE__, __LINE__); return -1; }
            if (prec < 0) prec = towers[i]->prec;
            if (towers[i]->prec != prec) { m2m_set_error("pack_all: one precision per launch", __FILE__, __LINE__); return -1; }
            a.tw[i] = m2m_shrink(towers[i]);
            a.skip_w1tc[i] = pack_skips_w1tc(towers[i]);
            tiles += towers[i]->nblocks * (towers[i]->Cp / 32);
            maxD = std::max(maxD, (int)towers[i]->D);
        }
        a.tile_end[i] = tiles;
    }
    int embed_wgs = 0;
    for (int i = 0; i < nembeds; ++i) {
        const m2m_embed* e = embeds[i];
        if (!e || !e->w || !e->wn) { m2m_set_error("pack_all: null embed", __FILE__, __LINE__); return -1; }
        if (prec < 0) prec = e->prec;
        if (e->prec != prec) { m2m_set_error("pack_all: one precision per launch", __FILE__, __LINE__); return -1; }
        const long KB = prec == PREC_BF16 ? 32 : 16;
        if (e->D % 16 || e->Kp % KB || e->Kp < e->K) { m2m_set_error("pack_all: bad embed geometry", __FILE__, __LINE__); return -1; }
        a.em[i] = *e;
        const int wgs = (int)ceil_div((long)(e->D / 16) * (e->Kp / KB) * 64, 256);
        if (i == 0) a.embed_wgs0 = wgs;
        embed_wgs += wgs;
    }
    const size_t lds = (size_t)(32 * (maxD + 1) + maxD * 33) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static size_t attr_lds[2] = {0, 0};
    const int pi = prec == PREC_BF16 ? 0 : 1;
    if (lds > attr_lds[pi]) {
        const void* fn = prec == PREC_BF16 ? reinterpret_cast<const void*>(pack_all_kernel<PREC_BF16>) : reinterpret_cast<const void*>(pack_all_kernel<PREC_F32>);
        M2M_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds[pi] = lds;
    }
    const dim3 grid((unsigned)(tiles + embed_wgs));
    if (prec == PREC_BF16) hipLaunchKernelGGL(pack_all_kernel<PREC_BF16>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(pack_all_kernel<PREC_F32>, grid, dim3(256), lds, st, a);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int m2m_pack_embed(const m2m_embed* e, void* stream) {
    if (!e) { m2m_set_error("null embed", __FILE__, __LINE__); return -1; }
    return pack_impl(e->prec, PACK_NAT, 0, e->w, e->K, 1, e->D, e->K, e->D, e->Kp, e->wn, stream);
}

// ---------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults, no amsgrad): models/avmnist.py:413-415
// ---------------------------------------------------------------------------------------------------
__global__ void adam_bump_kernel(float* state) { state[0] += 1.0f; }

// gscale < 0 requests "consume": after the update the gradient element is cleared, so the next step starts from
// zeroed gradients without a separate fill pass (|gscale| is the scale).
// LOWP: the gradient VALUE comes from a bf16 copy (the all-reduced, compressed gradient of the data-parallel step: no pass
// to widen it back); the fp32 gradient buffer is only cleared.
// Ranges (m2m_adam_step_ranges): inside [lo, lo + n) the gradient is grad[i] + add[i - lo] (a weight-gradient slot) and / or is
// not cleared (keep: the next backward overwrites it).  A workgroup walks 1024-element chunks (256 threads x 16 bytes); the
// range a chunk lies in is a wave-uniform decision, chunks that straddle a range boundary (a handful) go element by element.
struct AdamRanges {
    int n;
    long lo[M2M_MAX_GRAD_RANGES], hi[M2M_MAX_GRAD_RANGES];
    const float* add[M2M_MAX_GRAD_RANGES];
    int keep[M2M_MAX_GRAD_RANGES];
};
struct AdamK { float b1, b2, eps, wd, gscale, step_size, inv_sqrt_bc2; };
static __device__ __forceinline__ void adam_one(const AdamK& k, float g, float& p, float& m, float& v) {
    g *= k.gscale;
    if (k.wd != 0.f) g = __builtin_fmaf(k.wd, p, g);
    m = k.b1 * m + (1.0f - k.b1) * g;
    v = k.b2 * v + (1.0f - k.b2) * g * g;
    const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
    p = p - k.step_size * (m / denom);
}
// NT (bit mask): which streams use non-temporal accesses -- they pass the memory-side cache (Infinity Cache) without allocating,
// so what the chain kernels keep there (weights, the stored operands of the last backward blocks: m2m_handoff_resident_blocks)
// survives the optimizer's 230 MB.  1: exp_avg / exp_avg_sq (read once, written once per step), 2: parameter loads,
// 4: parameter stores (the re-pack then reads the parameters from HBM), 8: gradient loads.
#ifndef M2M_ADAM_NT_DEFAULT
#define M2M_ADAM_NT_DEFAULT 1
#endif
template <bool ON> static __device__ __forceinline__ float ld_maybe_nt(const float* p) { return ON ? __builtin_nontemporal_load(p) : *p; }
template <bool ON> static __device__ __forceinline__ void st_maybe_nt(float* p, float v) { if (ON) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool LOWP, int NT>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ gr, const unsigned short* __restrict__ gb,
                                                   float* __restrict__ m, float* __restrict__ v, long n, const float* __restrict__ state,
                                                   float b1, float b2, float eps, float wd, float gscale_in, const AdamRanges rg) {
    const bool consume = gscale_in < 0.f;
    const float stepf = state[0], lr = state[1];
    const float bc1 = 1.0f - powf(b1, stepf);
    const float bc2 = 1.0f - powf(b2, stepf);
    AdamK k;
    k.b1 = b1; k.b2 = b2; k.eps = eps; k.wd = wd;
    k.gscale = consume ? -gscale_in : gscale_in;
    k.step_size = lr / bc1;
    k.inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    // A workgroup walks 1024-element chunks, four 4-byte elements per thread (256 contiguous bytes per wave instruction: the
    // access shape of the plain grid-stride loop this replaces, which ran at the HBM rate; 16-byte accesses -- one or four per
    // thread -- measured 62 and 85 us for Adam + re-pack against 61).  The range a chunk lies in is a workgroup-uniform decision;
    // the handful of chunks that straddle a range boundary look every element up.
    constexpr int EPT = 4, CH = 256 * EPT;
    const long nchunks = (n + CH - 1) / CH;
    for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const long c0 = ch * CH, c1 = min(c0 + CH, n);
        int cls = -1;                                            // -1 outside every range, r >= 0 wholly inside range r, -2 straddling
        for (int r = 0; r < rg.n; ++r) {
            if (c0 >= rg.lo[r] && c1 <= rg.hi[r]) { cls = r; break; }
            if (c0 < rg.hi[r] && c1 > rg.lo[r]) { cls = -2; break; }
        }
        if (cls != -2) {
            const bool keep = cls >= 0 && rg.keep[cls] != 0;
            const float* add = cls >= 0 ? rg.add[cls] : nullptr;
            const long alo = cls >= 0 ? rg.lo[cls] : 0;
            float g[EPT], pp[EPT], mm[EPT], vv[EPT];
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const long i = min(c0 + q * 256 + (long)threadIdx.x, c1 - 1);       // (clamped: unconditional loads)
                g[q] = LOWP ? __uint_as_float((unsigned int)gb[i] << 16) : ld_maybe_nt<(NT & 8) != 0>(gr + i);
                pp[q] = ld_maybe_nt<(NT & 2) != 0>(p + i); mm[q] = ld_maybe_nt<(NT & 1) != 0>(m + i); vv[q] = ld_maybe_nt<(NT & 1) != 0>(v + i);
            }
            if (add) {                                           // workgroup-uniform
#pragma unroll
                for (int q = 0; q < EPT; ++q) g[q] += add[min(c0 + q * 256 + (long)threadIdx.x, c1 - 1) - alo];
            }
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const long i = c0 + q * 256 + (long)threadIdx.x;
                adam_one(k, g[q], pp[q], mm[q], vv[q]);
                if (i < c1) {
                    if (consume && !keep) gr[i] = 0.f;
                    st_maybe_nt<(NT & 1) != 0>(m + i, mm[q]); st_maybe_nt<(NT & 1) != 0>(v + i, vv[q]); st_maybe_nt<(NT & 4) != 0>(p + i, pp[q]);
                }
            }
        } else {
            for (long e = c0 + threadIdx.x; e < c1; e += 256) {
                float g = LOWP ? __uint_as_float((unsigned int)gb[e] << 16) : gr[e];
                bool keep = false;
                for (int r = 0; r < rg.n; ++r)
                    if (e >= rg.lo[r] && e < rg.hi[r]) {
                        keep = rg.keep[r] != 0;
                        if (rg.add[r]) g += rg.add[r][e - rg.lo[r]];
                    }
                float pp = p[e], mm = m[e], vv = v[e];
                adam_one(k, g, pp, mm, vv);
                if (consume && !keep) gr[e] = 0.f;
                m[e] = mm; v[e] = vv; p[e] = pp;
            }
        }
    }
}

static int adam_launch(float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                       float beta1, float beta2, float eps, float weight_decay, float grad_scale, int bump_step,
                       const m2m_grad_range* ranges, int nranges, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (bump_step) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, st, state);
    if (n <= 0) return 0;
    if (nranges < 0 || nranges > M2M_MAX_GRAD_RANGES || (nranges > 0 && !ranges)) { m2m_set_error("adam_step: bad ranges", __FILE__, __LINE__); return -1; }
    const long head = 0;                                         // (4-byte accesses: no alignment requirement on the segment)
    AdamRanges rg;
    memset(&rg, 0, sizeof(rg));
    rg.n = nranges;
    for (int r = 0; r < nranges; ++r) {
        if (ranges[r].lo < 0 || ranges[r].n < 0 || ranges[r].lo + ranges[r].n > n) { m2m_set_error("adam_step_ranges: range outside the buffers", __FILE__, __LINE__); return -1; }
        rg.lo[r] = (long)ranges[r].lo - head; rg.hi[r] = (long)(ranges[r].lo + ranges[r].n) - head;
        rg.add[r] = ranges[r].add; rg.keep[r] = ranges[r].keep;
    }
    const unsigned short* gb = reinterpret_cast<const unsigned short*>(grad_bf16);
    auto launch = [&](float* p_, float* g_, const unsigned short* gb_, float* m_, float* v_, long n_, const AdamRanges& r_) {
        long grid = ceil_div(n_, 1024);
        if (grid > 2048) grid = 2048;
        static const int nt = [] { const char* e = getenv("M2M_ADAM_NT"); return e ? atoi(e) : M2M_ADAM_NT_DEFAULT; }();
#define M2M_ADAM_GO(LP, N) hipLaunchKernelGGL((adam_kernel<LP, N>), dim3((unsigned)grid), dim3(256), 0, st, p_, g_, gb_, m_, v_, n_, state, beta1, beta2, \
                                              eps, weight_decay, grad_scale, r_)
#define M2M_ADAM_SW(LP) switch (nt) { case 1: M2M_ADAM_GO(LP, 1); break; case 3: M2M_ADAM_GO(LP, 3); break; case 7: M2M_ADAM_GO(LP, 7); break; \
                                      case 15: M2M_ADAM_GO(LP, 15); break; default: M2M_ADAM_GO(LP, 0); break; }
        if (gb_) { M2M_ADAM_SW(true) } else { M2M_ADAM_SW(false) }
#undef M2M_ADAM_SW
#undef M2M_ADAM_GO
    };
    if (head > 0) {
        // scalar head: a 1-chunk launch whose only chunk is shorter than a full chunk goes element by element
        AdamRanges rh = rg;
        for (int r = 0; r < rh.n; ++r) { rh.lo[r] += head; rh.hi[r] += head; }
        launch(param, grad, gb, exp_avg, exp_avg_sq, head, rh);
    }
    if (n - head > 0)
        launch(param + head, grad + head, gb ? gb + head : nullptr, exp_avg + head, exp_avg_sq + head, (long)n - head, rg);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int m2m_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                             float beta1, float beta2, float eps, float weight_decay, float grad_scale, int bump_step,
                             void* stream) {
    return adam_launch(param, grad, nullptr, exp_avg, exp_avg_sq, n, state, beta1, beta2, eps, weight_decay, grad_scale, bump_step, nullptr, 0, stream);
}
extern "C" int m2m_adam_step_bf16(float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  float* state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                  int bump_step, void* stream) {
    if (!grad_bf16) { m2m_set_error("adam_step_bf16: null bf16 gradient", __FILE__, __LINE__); return -1; }
    return adam_launch(param, grad, grad_bf16, exp_avg, exp_avg_sq, n, state, beta1, beta2, eps, weight_decay, grad_scale, bump_step, nullptr, 0, stream);
}
extern "C" int m2m_adam_step_ranges(float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    float* state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                    int bump_step, const m2m_grad_range* ranges, int nranges, void* stream) {
    return adam_launch(param, grad, grad_bf16, exp_avg, exp_avg_sq, n, state, beta1, beta2, eps, weight_decay, grad_scale, bump_step, ranges, nranges, stream);
}

// ---------------------------------------------------------------------------------------------------
// Adam + operand re-pack of a whole model in ONE launch (replaces m2m_adam_step over the flat buffers followed by
// m2m_pack_all): the re-pack no longer re-reads the 33 MB of fp32 masters Adam has just written, and the step loses a launch.
//   tile workgroups  (tower, block, 32-column group q): Adam on W1 rows [32q, 32q + 32), W2 columns [32q, 32q + 32) and
//                    ch_b1[32q ..]; the updated values go to memory AND into the LDS tiles the four packed copies are
//                    emitted from (pack_emit_tile);
//   embed workgroups 256 packed slots each: Adam on the 8 weights of a slot (one 128-byte line per 16-row block row),
//                    then the slot's packed bf16 / fp32 image;
//   flat workgroups  1024 elements each of everything else (LayerNorms, token MLPs, ch_b2, embedding biases, heads).
// The plan (which flat ranges are "everything else", the Adam constants, the flat buffers) is a device-resident struct the
// host builds once (m2m_adam_pack_plan): with the three by-value tower descriptors the kernel arguments are at the 4 KiB limit.
// ---------------------------------------------------------------------------------------------------
#define M2M_AP_MAXSEG 96
struct AdamPackPlan {
    float* p; float* g; const unsigned short* gb; float* m; float* v; const float* state;
    float b1, b2, eps, wd, gscale;
    int nseg;
    long seg_lo[M2M_AP_MAXSEG], seg_hi[M2M_AP_MAXSEG];
    int seg_wg0[M2M_AP_MAXSEG + 1];          // first flat workgroup of each segment (1024 elements per workgroup)
    // gradient ranges (m2m_adam_pack_plan_ranges; the semantics of m2m_adam_step_ranges): inside [lo, hi) the gradient is
    // grad[i] + add[i - lo] (a weight-gradient slot) and / or is not cleared (keep: the next backward overwrites it)
    int nrange;
    long r_lo[M2M_MAX_GRAD_RANGES], r_hi[M2M_MAX_GRAD_RANGES];
    const float* r_add[M2M_MAX_GRAD_RANGES];
    int r_keep[M2M_MAX_GRAD_RANGES];
};
struct AdamConsts { float b1, b2, eps, wd, gscale, step_size, inv_sqrt_bc2; };
static __device__ __forceinline__ AdamConsts adam_consts(const AdamPackPlan& pl) {
    AdamConsts c;
    c.b1 = pl.b1; c.b2 = pl.b2; c.eps = pl.eps; c.wd = pl.wd; c.gscale = pl.gscale;
    const float stepf = pl.state[0], lr = pl.state[1];
    c.step_size = lr / (1.0f - powf(pl.b1, stepf));
    c.inv_sqrt_bc2 = 1.0f / sqrtf(1.0f - powf(pl.b2, stepf));
    return c;
}
// the Adam arithmetic of adam_one (api.hip above) on registers
static __device__ __forceinline__ void adam_math(const AdamConsts& c, float g, float& p, float& m, float& v) {
    g *= c.gscale;
    if (c.wd != 0.f) g = __builtin_fmaf(c.wd, p, g);
    m = c.b1 * m + (1.0f - c.b1) * g;
    v = c.b2 * v + (1.0f - c.b2) * g * g;
    p = p - c.step_size * (m / (sqrtf(v) * c.inv_sqrt_bc2 + c.eps));
}
// The four flat streams + the range of one tensor, as global-address-space pointers with scalar bases (the plan lives in
// device memory: generic pointers read from it would give FLAT accesses and per-load pointer re-reads).
struct AdamStreams {
    M2M_AS1 float* p; M2M_AS1 float* g; const M2M_AS1 unsigned short* gb; M2M_AS1 float* m; M2M_AS1 float* v;
    const M2M_AS1 float* add;      // slot of the tensor at hand (NULL: none), indexed like the flat buffers MINUS add_lo
    long add_lo;
    bool keep;
};
static __device__ __forceinline__ AdamStreams adam_streams(const AdamPackPlan& pl, long flat_off) {
    AdamStreams s;
    s.p = (M2M_AS1 float*)uniform_u64((unsigned long long)pl.p); s.g = (M2M_AS1 float*)uniform_u64((unsigned long long)pl.g);
    s.gb = (const M2M_AS1 unsigned short*)uniform_u64((unsigned long long)pl.gb);
    s.m = (M2M_AS1 float*)uniform_u64((unsigned long long)pl.m); s.v = (M2M_AS1 float*)uniform_u64((unsigned long long)pl.v);
    s.add = nullptr; s.add_lo = 0; s.keep = false;
    for (int r = 0; r < pl.nrange; ++r)
        if (flat_off >= pl.r_lo[r] && flat_off < pl.r_hi[r]) {
            s.add = (const M2M_AS1 float*)uniform_u64((unsigned long long)pl.r_add[r]); s.add_lo = pl.r_lo[r]; s.keep = pl.r_keep[r] != 0;
        }
    return s;
}
// NV float4 groups per thread at flat offsets off[k] (any 4-byte alignment; entries with ok[k] == false are skipped by the
// stores -- their loads are clamped duplicates): EVERY load first, then the arithmetic, then the stores.  The first version of
// this kernel updated element by element through generic pointers (load, store, load ... in series; the stores may alias the
// next loads): 97 us for the model against 45 + 19 us for the flat Adam + m2m_pack_all it was meant to replace.
#ifndef M2M_AP_NT_P
#define M2M_AP_NT_P 0
#endif
// NTMV (compile time -- a run-time choice between a plain and a non-temporal store of the same value is merged into ONE plain
// store by the compiler): exp_avg / exp_avg_sq past the memory-side cache (large models, see adam_kernel's NT)
template <bool LOWP, int NV, bool NTMV = false>
static __device__ __forceinline__ void adam_vec(const AdamStreams& s, const AdamConsts& c, const long (&off)[NV], const bool (&ok)[NV],
                                                f32x4_t (&pn)[NV]) {
    typedef M2M_AS1 f32x4_t* g4_t;
    f32x4_t gv[NV], mv[NV], vv[NV], av[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        // (the big masters too when M2M_AP_NT_P: in the one-launch form nothing re-reads them before the next step's Adam)
        if constexpr (NTMV && M2M_AP_NT_P) pn[k] = __builtin_nontemporal_load((g4_t)(s.p + off[k])); else pn[k] = *(g4_t)(s.p + off[k]);
        if constexpr (NTMV) { mv[k] = __builtin_nontemporal_load((g4_t)(s.m + off[k])); vv[k] = __builtin_nontemporal_load((g4_t)(s.v + off[k])); }
        else { mv[k] = *(g4_t)(s.m + off[k]); vv[k] = *(g4_t)(s.v + off[k]); }
        if (LOWP) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[k][e] = __uint_as_float((unsigned int)s.gb[off[k] + e] << 16);
        } else gv[k] = *(g4_t)(s.g + off[k]);
    }
    if (s.add) {                                        // workgroup-uniform
#pragma unroll
        for (int k = 0; k < NV; ++k) av[k] = *(const g4_t)(const_cast<M2M_AS1 float*>(s.add) + (off[k] - s.add_lo));
#pragma unroll
        for (int k = 0; k < NV; ++k) gv[k] = gv[k] + av[k];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { float pp = pn[k][e], mm = mv[k][e], v1 = vv[k][e]; adam_math(c, gv[k][e], pp, mm, v1); pn[k][e] = pp; mv[k][e] = mm; vv[k][e] = v1; }
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (ok[k]) {
            if constexpr (NTMV && M2M_AP_NT_P) __builtin_nontemporal_store(pn[k], (g4_t)(s.p + off[k])); else *(g4_t)(s.p + off[k]) = pn[k];
            if constexpr (NTMV) { __builtin_nontemporal_store(mv[k], (g4_t)(s.m + off[k])); __builtin_nontemporal_store(vv[k], (g4_t)(s.v + off[k])); }
            else { *(g4_t)(s.m + off[k]) = mv[k]; *(g4_t)(s.v + off[k]) = vv[k]; }
            if (!s.keep) *(g4_t)(s.g + off[k]) = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
}
// one element (the ragged last column group of a tensor, embedding slots, the flat workgroups' range edges)
template <bool LOWP>
static __device__ __forceinline__ float adam_elem(const AdamStreams& s, const AdamConsts& c, long i) {
    float g = LOWP ? __uint_as_float((unsigned int)s.gb[i] << 16) : s.g[i];
    if (s.add) g += s.add[i - s.add_lo];
    float pp = s.p[i], mm = s.m[i], vv = s.v[i];
    adam_math(c, g, pp, mm, vv);
    s.p[i] = pp; s.m[i] = mm; s.v[i] = vv;
    if (!s.keep) s.g[i] = 0.f;
    return pp;
}

// (tower, block, 32-column group q) of hidden_dim DD: Adam on W1 rows [32q, 32q + 32) and W2 columns [32q, 32q + 32) with every
// load in flight together (2 x DD / 32 float4 groups per thread and stream), the updated values into the LDS tiles, ch_b1, then
// the packed copies from the tiles.
template <int P, bool LOWP, int DD, bool NTMV>
static __device__ __forceinline__ void adam_pack_tile(const AdamPackPlan& pl, const AdamConsts& c, const m2m_tower4& tw, int block, int q, char* smem, bool skip_w1tc) {
    const m2m_block& k = tw.blk[block];
    constexpr int D = DD, L1 = DD + 1, L2 = 33, NV = 32 * (DD / 4) / 256;      // float4 groups per thread and tensor (256 threads)
    const int C = tw.C, c0 = 32 * q, tid = threadIdx.x;
    float* t1 = reinterpret_cast<float*>(smem);
    float* t2 = t1 + 32 * L1;
    const long o1 = k.ch_w1 - pl.p, o2 = k.ch_w2 - pl.p, ob = k.ch_b1 - pl.p;       // flat offsets of this block's tensors
    const AdamStreams s1 = adam_streams(pl, o1), s2 = adam_streams(pl, o2), sb = adam_streams(pl, ob);
    const bool full = c0 + 32 <= C;                     // (workgroup-uniform) the whole column group lies inside the tensor
    {
        long off[NV];
        bool ok[NV];
        int rr[NV], dd[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            rr[i] = idx / (D / 4); dd[i] = (idx % (D / 4)) * 4;
            ok[i] = c0 + rr[i] < C;
            off[i] = o1 + (long)min(c0 + rr[i], C - 1) * D + dd[i];
        }
        f32x4_t pn[NV];
        adam_vec<LOWP, NV, NTMV>(s1, c, off, ok, pn);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float* o = t1 + rr[i] * L1 + dd[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ok[i] ? pn[i][e] : 0.f;
        }
    }
    if (full) {
        long off[NV];
        bool ok[NV];
        int rd[NV], jj[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;               // D rows x 8 float4 per row
            rd[i] = idx >> 3; jj[i] = (idx & 7) * 4;
            ok[i] = true;
            off[i] = o2 + (long)rd[i] * C + c0 + jj[i];
        }
        f32x4_t pn[NV];
        adam_vec<LOWP, NV, NTMV>(s2, c, off, ok, pn);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) t2[rd[i] * L2 + jj[i] + e] = pn[i][e];
    } else {
        for (int idx = tid; idx < D * 32; idx += 256) {
            const int d = idx >> 5, j = idx & 31;
            t2[d * L2 + j] = c0 + j < C ? adam_elem<LOWP>(s2, c, o2 + (long)d * C + c0 + j) : 0.f;
        }
    }
    if (tid < 32) k.ch_b1p[c0 + tid] = c0 + tid < C ? adam_elem<LOWP>(sb, c, ob + c0 + tid) : 0.f;
    __syncthreads();
    pack_emit_tile<P>(k, D, q, t1, t2, skip_w1tc);
}

// ---- row-tile form (bf16): W1 and W2 as tiles of their own, each read and written in long contiguous runs -------------------------------
// The (block, 32-column group) tile above touches W2 -- (D, C) row-major, the reference's nn.Linear layout -- in 128-byte segments
// 4 C bytes apart: four streams of DRAM row misses, 3.4 TB/s for the model against 5.3 for the flat Adam.  Here W1 keeps its tile
// (32 rows of W1 are one contiguous 32 x D chunk) and W2 is walked in tiles of 8 rows x AP_W columns (2 KiB runs): a packed NAT slot
// of W2^T is eight consecutive d of one column -- exactly the tile's eight rows --, a packed CHN slot of W2 eight columns of one
// row, so both images come out of the tile (128- and 256-byte runs of 16-byte slots).
#ifndef AP_W
#define AP_W 512
#endif
template <int P, bool LOWP, int DD, bool NTMV>
static __device__ __forceinline__ void adam_pack_w1_tile(const AdamPackPlan& pl, const AdamConsts& c, const m2m_tower4& tw, int block, int q, char* smem, bool skip_w1tc) {
    const m2m_block& k = tw.blk[block];
    constexpr int D = DD, L1 = DD + 1, NV = 32 * (DD / 4) / 256;
    const int C = tw.C, c0 = 32 * q, tid = threadIdx.x;
    float* t1 = reinterpret_cast<float*>(smem);
    const long o1 = k.ch_w1 - pl.p, ob = k.ch_b1 - pl.p;
    const AdamStreams s1 = adam_streams(pl, o1), sb = adam_streams(pl, ob);
    long off[NV];
    bool ok[NV];
    int rr[NV], dd[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        rr[i] = idx / (D / 4); dd[i] = (idx % (D / 4)) * 4;
        ok[i] = c0 + rr[i] < C;
        off[i] = o1 + (long)min(c0 + rr[i], C - 1) * D + dd[i];
    }
    f32x4_t pn[NV];
    adam_vec<LOWP, NV, NTMV>(s1, c, off, ok, pn);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float* o = t1 + rr[i] * L1 + dd[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ok[i] ? pn[i][e] : 0.f;
    }
    if (tid < 32) k.ch_b1p[c0 + tid] = c0 + tid < C ? adam_elem<LOWP>(sb, c, ob + c0 + tid) : 0.f;
    __syncthreads();
    pack_emit_tile<P, true, false>(k, D, q, t1, t1, skip_w1tc);
}
template <bool LOWP, bool NTMV>
static __device__ __forceinline__ void adam_pack_w2_rows(const AdamPackPlan& pl, const AdamConsts& c, const m2m_tower4& tw, int block, int dgrp, int chunk, char* smem) {
    const m2m_block& k = tw.blk[block];
    constexpr int W = AP_W, LD = AP_W + 4, NV = 8 * (AP_W / 4) / 256;
    const int D = tw.D, C = tw.C, Cp = tw.Cp, tid = threadIdx.x;
    const int d0 = 8 * dgrp, c0 = W * chunk;
    float* t = reinterpret_cast<float*>(smem);           // [8][LD]: W2[d0 + r][c0 + j] (columns past C: zero)
    const long o2 = k.ch_w2 - pl.p;
    const AdamStreams s2 = adam_streams(pl, o2);
    if (c0 + W <= C) {                                    // (workgroup-uniform)
        long off[NV];
        bool ok[NV];
        int rr[NV], jj[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;                // 8 rows x W / 4 float4 per row
            rr[i] = idx / (W / 4); jj[i] = (idx % (W / 4)) * 4;
            ok[i] = true;
            off[i] = o2 + (long)(d0 + rr[i]) * C + c0 + jj[i];
        }
        f32x4_t pn[NV];
        adam_vec<LOWP, NV, NTMV>(s2, c, off, ok, pn);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            *reinterpret_cast<f32x4_t*>(t + rr[i] * LD + jj[i]) = pn[i];
    } else {
        for (int idx = tid; idx < 8 * W; idx += 256) {
            const int r = idx / W, j = idx % W;
            t[r * LD + j] = c0 + j < C ? adam_elem<LOWP>(s2, c, o2 + (long)(d0 + r) * C + c0 + j) : 0.f;
        }
    }
    __syncthreads();
    const int nIB = D / 16, nKB = D / 32, ncb = (min(W, Cp - c0)) >> 5;      // 32-column blocks of this chunk
    auto emit = [&](char* dst, long blk, int lane, const float (&v)[8]) {
        *reinterpret_cast<u32x4_t*>(dst + (blk * 64 + lane) * 16) =
            u32x4_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    };
    // w2c: CHN, k-major, X[i = d][k = c]: blocks (kb = c / 32, ib = d / 16), lane (g, il = d % 16): the row's columns 32 kb + {4g..4g+3, 16+4g..}
    for (int sl = tid; sl < 8 * 4 * ncb; sl += 256) {
        const int r = sl & 7, g = (sl >> 3) & 3, kbl = sl >> 5;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t[r * LD + 32 * kbl + 16 * (e >> 2) + 4 * g + (e & 3)];
        emit((char*)k.w2c, (long)(c0 / 32 + kbl) * nIB + d0 / 16, g * 16 + (d0 & 15) + r, v);
    }
    // w2tn: NAT, X[i = c][k = d]: blocks (ib = c / 16, kb = d / 32), lane (g = (d % 32) / 8, il = c % 16): the column's eight rows
    for (int j = tid; j < 32 * ncb; j += 256) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t[e * LD + j];
        const int cc = c0 + j;
        emit((char*)k.w2tn, (long)(cc / 16) * nKB + d0 / 32, ((d0 & 31) >> 3) * 16 + (cc & 15), v);
    }
}

// DK: 0 = towers of any hidden_dim (run-time switch: the kernel's register allocation is then that of the widest instantiation, 254
// VGPRs = two workgroups per CU), else the hidden_dim every tower of the launch has (128: ~100 registers, five workgroups per CU)
template <int P, bool LOWP, bool NTMV, int DK>
__global__ __launch_bounds__(256, (DK == 64 || DK == 128) ? 4 : 1) void adam_pack_all_kernel(const PackAllArgs a, const AdamPackPlan* __restrict__ plan, int embed_wgs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AdamPackPlan& pl = *plan;
    const AdamConsts c = adam_consts(pl);
    int id = blockIdx.x;
    const int tid = threadIdx.x;
    if (id < a.tile_end[M2M_PACK_TOWERS - 1]) {
        int t = 0;
        while (id >= a.tile_end[t]) ++t;
        if (t) id -= a.tile_end[t - 1];
        const m2m_tower4& tw = a.tw[t];
        if constexpr (P == PREC_BF16) {
            if (a.rowtiles[t] > 0) {                     // (workgroup-uniform) row-tile form: W1 tiles, then W2 row tiles, per block
                const int nq = tw.Cp >> 5, nch = a.rowtiles[t], per_block = nq + (tw.D / 8) * nch;
                const int block = id / per_block, r = id % per_block;
                if (r < nq) {
                    if constexpr (DK != 0) adam_pack_w1_tile<P, LOWP, DK, NTMV>(pl, c, tw, block, r, smem, a.skip_w1tc[t] != 0);
                    else
                    switch (tw.D) {
                        case 32:  adam_pack_w1_tile<P, LOWP, 32, NTMV>(pl, c, tw, block, r, smem, a.skip_w1tc[t] != 0); break;
                        case 64:  adam_pack_w1_tile<P, LOWP, 64, NTMV>(pl, c, tw, block, r, smem, a.skip_w1tc[t] != 0); break;
                        case 128: adam_pack_w1_tile<P, LOWP, 128, NTMV>(pl, c, tw, block, r, smem, a.skip_w1tc[t] != 0); break;
                        default:  adam_pack_w1_tile<P, LOWP, 256, NTMV>(pl, c, tw, block, r, smem, a.skip_w1tc[t] != 0); break;
                    }
                } else adam_pack_w2_rows<LOWP, NTMV>(pl, c, tw, block, (r - nq) / nch, (r - nq) % nch, smem);
                return;
            }
        }
        const int nq = tw.Cp >> 5, block = id / nq, q = id % nq;
        if constexpr (DK != 0) adam_pack_tile<P, LOWP, DK, NTMV>(pl, c, tw, block, q, smem, a.skip_w1tc[t] != 0);
        else
        switch (tw.D) {                                  // (workgroup-uniform)
            case 32:  adam_pack_tile<P, LOWP, 32, NTMV>(pl, c, tw, block, q, smem, a.skip_w1tc[t] != 0); break;
            case 64:  adam_pack_tile<P, LOWP, 64, NTMV>(pl, c, tw, block, q, smem, a.skip_w1tc[t] != 0); break;
            case 128: adam_pack_tile<P, LOWP, 128, NTMV>(pl, c, tw, block, q, smem, a.skip_w1tc[t] != 0); break;
            default:  adam_pack_tile<P, LOWP, 256, NTMV>(pl, c, tw, block, q, smem, a.skip_w1tc[t] != 0); break;
        }
        return;
    }
    id -= a.tile_end[M2M_PACK_TOWERS - 1];
    if (id < embed_wgs) {
        typedef Prec<P> Pr;
        const int e = id < a.embed_wgs0 ? 0 : 1;
        if (e) id -= a.embed_wgs0;
        const m2m_embed& em = a.em[e];
        const long nIB = em.D / 16, nKB = em.Kp / Pr::KB;
        const long slot = (long)id * 256 + tid;
        if (slot >= nIB * nKB * 64) return;
        const long blk = slot >> 6;
        const int lane = (int)(slot & 63), g = lane >> 4, il = lane & 15;
        const long ib = blk / nKB, kb = blk % nKB;                 // NAT, i-major: m2m_pack_embed's layout
        const long i = ib * 16 + il, o = em.w - pl.p;
        const AdamStreams se = adam_streams(pl, o);
        // the slot's EPL weights are consecutive in k: all loads first (clamped), then the arithmetic, then the guarded stores
        float gq[8], pq[8], mq[8], vq[8];
        bool okq[8];
        // bf16: the slot's eight weights are eight consecutive k of one row (32 bytes).  16-byte accesses when the row length and the
        // buffers' offsets allow it (workgroup-uniform): the four k-groups of a row then share a 128-byte line per instruction instead
        // of every lane of every instruction touching a line of its own (rows are K floats apart)
        const long k0 = kb * Pr::KB + Pr::kmap(PACK_NAT, g, 0);
        const bool vec = P == PREC_BF16 && !LOWP && (em.K & 3) == 0 && (o & 3) == 0 && i < em.D && k0 + 8 <= em.K &&
                         ((reinterpret_cast<uintptr_t>(se.p) | reinterpret_cast<uintptr_t>(se.g) | reinterpret_cast<uintptr_t>(se.m) |
                           reinterpret_cast<uintptr_t>(se.v)) & 15) == 0;
        if (vec) {
            const long at = o + i * em.K + k0;
            f32x4_t g4[2], p4[2], m4[2], v4[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                g4[h] = *reinterpret_cast<const M2M_AS1 f32x4_t*>(se.g + at + 4 * h); p4[h] = *reinterpret_cast<const M2M_AS1 f32x4_t*>(se.p + at + 4 * h);
                m4[h] = *reinterpret_cast<const M2M_AS1 f32x4_t*>(se.m + at + 4 * h); v4[h] = *reinterpret_cast<const M2M_AS1 f32x4_t*>(se.v + at + 4 * h);
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) { gq[x] = g4[x >> 2][x & 3]; pq[x] = p4[x >> 2][x & 3]; mq[x] = m4[x >> 2][x & 3]; vq[x] = v4[x >> 2][x & 3]; okq[x] = true; }
        } else {
#pragma unroll
        for (int x = 0; x < Pr::EPL; ++x) {
            const long kk = kb * Pr::KB + Pr::kmap(PACK_NAT, g, x);
            okq[x] = i < em.D && kk < em.K;
            const long at = o + min(i, (long)em.D - 1) * em.K + min(kk, (long)em.K - 1);
            gq[x] = LOWP ? __uint_as_float((unsigned int)se.gb[at] << 16) : se.g[at];
            pq[x] = se.p[at]; mq[x] = se.m[at]; vq[x] = se.v[at];
        }
        }
        float v[8];
        if (vec) {
#pragma unroll
            for (int x = 0; x < 8; ++x) { adam_math(c, gq[x], pq[x], mq[x], vq[x]); v[x] = pq[x]; }
            const long at = o + i * em.K + k0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<M2M_AS1 f32x4_t*>(se.p + at + 4 * h) = f32x4_t{pq[4 * h], pq[4 * h + 1], pq[4 * h + 2], pq[4 * h + 3]};
                *reinterpret_cast<M2M_AS1 f32x4_t*>(se.m + at + 4 * h) = f32x4_t{mq[4 * h], mq[4 * h + 1], mq[4 * h + 2], mq[4 * h + 3]};
                *reinterpret_cast<M2M_AS1 f32x4_t*>(se.v + at + 4 * h) = f32x4_t{vq[4 * h], vq[4 * h + 1], vq[4 * h + 2], vq[4 * h + 3]};
                if (!se.keep) *reinterpret_cast<M2M_AS1 f32x4_t*>(se.g + at + 4 * h) = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
        for (int x = 0; x < Pr::EPL; ++x) {
            const long kk = kb * Pr::KB + Pr::kmap(PACK_NAT, g, x);
            adam_math(c, gq[x], pq[x], mq[x], vq[x]);
            v[x] = okq[x] ? pq[x] : 0.f;
            if (okq[x]) {
                const long at = o + i * em.K + kk;
                se.p[at] = pq[x]; se.m[at] = mq[x]; se.v[at] = vq[x];
                if (!se.keep) se.g[at] = 0.f;
            }
        }
        }
        Frag f;
        if (P == PREC_BF16) {
#pragma unroll
            for (int x = 0; x < 4; ++x) f.u[x] = pack_bf2(v[2 * x], v[2 * x + 1]);
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) f.f[x] = v[x];
        }
        *reinterpret_cast<u32x4_t*>((char*)em.wn + slot * 16) = f.u;
        return;
    }
    id -= embed_wgs;
    int sgm = 0;
    while (sgm + 1 < pl.nseg && id >= pl.seg_wg0[sgm + 1]) ++sgm;
    const long lo = pl.seg_lo[sgm] + (long)(id - pl.seg_wg0[sgm]) * 1024, hi = min(pl.seg_hi[sgm], lo + 1024);
    // everything else: 1024 contiguous elements, four per thread, all loads first (the flat Adam's chunk: adam_kernel)
    {
        int cls = -1;                                            // -1 outside every range, r wholly inside range r, -2 straddling
        for (int r = 0; r < pl.nrange; ++r) {
            if (lo >= pl.r_lo[r] && hi <= pl.r_hi[r]) { cls = r; break; }
            if (lo < pl.r_hi[r] && hi > pl.r_lo[r]) { cls = -2; break; }
        }
        if (cls != -2) {
            const AdamStreams sf = adam_streams(pl, cls >= 0 ? pl.r_lo[cls] : -1);
            float gq[4], pq[4], mq[4], vq[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const long i = min(lo + x * 256 + (long)tid, hi - 1);
                gq[x] = LOWP ? __uint_as_float((unsigned int)sf.gb[i] << 16) : sf.g[i];
                if (sf.add) gq[x] += sf.add[i - sf.add_lo];
                pq[x] = sf.p[i];
                if constexpr (NTMV) { mq[x] = __builtin_nontemporal_load(sf.m + i); vq[x] = __builtin_nontemporal_load(sf.v + i); }
                else { mq[x] = sf.m[i]; vq[x] = sf.v[i]; }
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const long i = lo + x * 256 + (long)tid;
                adam_math(c, gq[x], pq[x], mq[x], vq[x]);
                if (i < hi) {
                    sf.p[i] = pq[x];
                    if constexpr (NTMV) { __builtin_nontemporal_store(mq[x], sf.m + i); __builtin_nontemporal_store(vq[x], sf.v + i); }
                    else { sf.m[i] = mq[x]; sf.v[i] = vq[x]; }
                    if (!sf.keep) sf.g[i] = 0.f;
                }
            }
        } else {
            for (long i = lo + tid; i < hi; i += 256) adam_elem<LOWP>(adam_streams(pl, i), c, i);
        }
    }
}

// Fills `plan_host` (sizeof == m2m_adam_pack_plan_bytes()) for the given model; the caller copies it to device memory and
// passes that copy to m2m_adam_pack_all.  grad_bf16 != NULL: gradient values come from that bf16 copy of `grad`.
extern "C" int64_t m2m_adam_pack_plan_bytes(void) { return (int64_t)sizeof(AdamPackPlan); }
extern "C" int m2m_adam_pack_plan_ranges(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds,
                                  float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const float* state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                  const m2m_grad_range* ranges, int nranges, void* plan_host);
extern "C" int m2m_adam_pack_plan(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds,
                                  float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const float* state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                  void* plan_host) {
    return m2m_adam_pack_plan_ranges(towers, ntowers, embeds, nembeds, param, grad, grad_bf16, exp_avg, exp_avg_sq, n, state, beta1, beta2,
                                     eps, weight_decay, grad_scale, nullptr, 0, plan_host);
}
extern "C" int m2m_adam_pack_plan_ranges(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds,
                                  float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const float* state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                  const m2m_grad_range* ranges, int nranges, void* plan_host) {
    if (!plan_host || !param || !grad || !exp_avg || !exp_avg_sq || !state || n <= 0) { m2m_set_error("adam_pack_plan: null argument", __FILE__, __LINE__); return -1; }
    if (nranges < 0 || nranges > M2M_MAX_GRAD_RANGES || (nranges > 0 && !ranges)) { m2m_set_error("adam_pack_plan: bad ranges", __FILE__, __LINE__); return -1; }
    AdamPackPlan pl;
    memset(&pl, 0, sizeof(pl));
    pl.nrange = nranges;
    for (int r = 0; r < nranges; ++r) {
        if (ranges[r].lo < 0 || ranges[r].n < 0 || ranges[r].lo + ranges[r].n > n) { m2m_set_error("adam_pack_plan: range outside the buffers", __FILE__, __LINE__); return -1; }
        pl.r_lo[r] = (long)ranges[r].lo; pl.r_hi[r] = (long)(ranges[r].lo + ranges[r].n); pl.r_add[r] = ranges[r].add; pl.r_keep[r] = ranges[r].keep;
    }
    pl.p = param; pl.g = grad; pl.gb = reinterpret_cast<const unsigned short*>(grad_bf16); pl.m = exp_avg; pl.v = exp_avg_sq; pl.state = state;
    pl.b1 = beta1; pl.b2 = beta2; pl.eps = eps; pl.wd = weight_decay; pl.gscale = grad_scale < 0.f ? -grad_scale : grad_scale;
    // the ranges the tile / embed workgroups own, sorted; the flat workgroups take the complement
    struct R { long lo, hi; };
    R own[3 * M2M_PACK_TOWERS * M2M_GROUP_BLOCKS + M2M_PACK_EMBEDS];
    int no = 0;
    auto add = [&](const float* ptr, long cnt) -> bool {
        const long lo = ptr - param;
        if (lo < 0 || lo + cnt > n) return false;
        own[no].lo = lo; own[no].hi = lo + cnt; ++no;
        return true;
    };
    for (int i = 0; i < ntowers; ++i) {
        const m2m_tower* t = towers[i];
        if (t->nblocks > M2M_GROUP_BLOCKS) { m2m_set_error("adam_pack_plan: towers of <= 4 blocks", __FILE__, __LINE__); return -1; }
        for (int b = 0; b < t->nblocks; ++b) {
            const m2m_block& k = t->blk[b];
            if (!add(k.ch_w1, (long)t->C * t->D) || !add(k.ch_w2, (long)t->C * t->D) || !add(k.ch_b1, t->C)) {
                m2m_set_error("adam_pack_plan: a channel-mixing weight is not inside the flat parameter buffer", __FILE__, __LINE__);
                return -1;
            }
        }
    }
    for (int i = 0; i < nembeds; ++i)
        if (!add(embeds[i]->w, (long)embeds[i]->D * embeds[i]->K)) { m2m_set_error("adam_pack_plan: an embedding weight is not inside the flat parameter buffer", __FILE__, __LINE__); return -1; }
    std::sort(own, own + no, [](const R& x, const R& y) { return x.lo < y.lo; });
    long cur = 0;
    int wg = 0;
    for (int i = 0; i <= no; ++i) {
        const long lo = cur, hi = i < no ? own[i].lo : (long)n;
        if (i < no && own[i].lo < cur) { m2m_set_error("adam_pack_plan: overlapping parameter tensors", __FILE__, __LINE__); return -1; }
        if (hi > lo) {
            if (pl.nseg >= M2M_AP_MAXSEG) { m2m_set_error("adam_pack_plan: too many parameter segments", __FILE__, __LINE__); return -1; }
            pl.seg_lo[pl.nseg] = lo; pl.seg_hi[pl.nseg] = hi; pl.seg_wg0[pl.nseg] = wg;
            wg += (int)ceil_div(hi - lo, 1024);
            ++pl.nseg;
        }
        if (i < no) cur = own[i].hi;
    }
    pl.seg_wg0[pl.nseg] = wg;
    memcpy(plan_host, &pl, sizeof(pl));
    return 0;
}

extern "C" int m2m_adam_pack_all(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds,
                                 const void* plan_dev, const void* plan_host, void* stream) {
    if (ntowers < 1 || ntowers > M2M_PACK_TOWERS || nembeds < 0 || nembeds > M2M_PACK_EMBEDS || !towers || (nembeds && !embeds) ||
        !plan_dev || !plan_host) {
        m2m_set_error("adam_pack_all: up to 3 towers and 2 embeddings, and a plan", __FILE__, __LINE__);
        return -1;
    }
    const AdamPackPlan* ph = reinterpret_cast<const AdamPackPlan*>(plan_host);
    PackAllArgs a;
    memset(&a, 0, sizeof(a));
    a.nt = ntowers; a.ne = nembeds;
    int prec = -1, tiles = 0, maxD = 0;
    bool all_rowtiles = true;
    for (int i = 0; i < M2M_PACK_TOWERS; ++i) {
        if (i < ntowers) {
            if (int rc = m2m_check_tower(towers[i], 1)) return rc;
            if (towers[i]->nblocks > M2M_GROUP_BLOCKS) { m2m_set_error("adam_pack_all: towers of <= 4 blocks", __FILE__, __LINE__); return -1; }
            if (prec < 0) prec = towers[i]->prec;
            if (towers[i]->prec != prec) { m2m_set_error("adam_pack_all: one precision per launch", __FILE__, __LINE__); return -1; }
            a.tw[i] = m2m_shrink(towers[i]);
            a.skip_w1tc[i] = pack_skips_w1tc(towers[i]);
            const char* rt_env = getenv("M2M_AP_ROWTILES");       // (read per call: the tests switch it inside one process)
            const int rowtiles = rt_env ? atoi(rt_env) : 1;
            if (rowtiles && prec == PREC_BF16 && towers[i]->Cp >= AP_W) {      // (narrow towers keep the column-group tiles)
                a.rowtiles[i] = (int)ceil_div((long)towers[i]->Cp, AP_W);
                tiles += towers[i]->nblocks * (towers[i]->Cp / 32 + (towers[i]->D / 8) * a.rowtiles[i]);
            } else {
                tiles += towers[i]->nblocks * (towers[i]->Cp / 32);
                all_rowtiles = false;
            }
            maxD = std::max(maxD, (int)towers[i]->D);
        }
        a.tile_end[i] = tiles;
    }
    int embed_wgs = 0;
    for (int i = 0; i < nembeds; ++i) {
        const m2m_embed* e = embeds[i];
        if (!e || !e->w || !e->wn || e->prec != prec) { m2m_set_error("adam_pack_all: bad embed", __FILE__, __LINE__); return -1; }
        const long KB = prec == PREC_BF16 ? 32 : 16;
        a.em[i] = *e;
        const int wgs = (int)ceil_div((long)(e->D / 16) * (e->Kp / KB) * 64, 256);
        if (i == 0) a.embed_wgs0 = wgs;
        embed_wgs += wgs;
    }
    const int flat_wgs = ph->seg_wg0[ph->nseg];
    // row-tile form everywhere: a workgroup needs the W1 tile OR the W2 row tile (half the LDS: twice the workgroups per CU)
    const size_t lds = all_rowtiles ? std::max((size_t)32 * (maxD + 1), (size_t)8 * (AP_W + 4)) * sizeof(float)
                                    : (size_t)(32 * (maxD + 1) + maxD * 33) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool lowp = ph->gb != nullptr;
    // the two moment streams past the memory-side cache for models it cannot hold anyway (> 4 M parameters: 64+ MB of moments);
    // small models keep them plain (they stay resident from step to step).  M2M_ADAM_NT=0 / 1 forces either.
    static const int nt_env = [] { const char* e = getenv("M2M_ADAM_NT"); return e ? atoi(e) : -1; }();
    long n_own = 0;
    for (int i = 0; i < ntowers; ++i) n_own += 2L * towers[i]->nblocks * towers[i]->C * towers[i]->D;
    const bool ntmv = nt_env >= 0 ? (nt_env & 1) != 0 : n_own > 4000000L;
    const dim3 grid((unsigned)(tiles + embed_wgs + flat_wgs));
    const AdamPackPlan* pd = reinterpret_cast<const AdamPackPlan*>(plan_dev);
    static size_t attr_lds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int dk = towers[0]->D;                                  // one hidden_dim for the whole launch: the instantiation built for it
    for (int i = 1; i < ntowers; ++i) if (towers[i]->D != dk) dk = 0;
    if (dk != 64 && dk != 128 && dk != 256) dk = 0;
#define M2M_APA_GO(PP, LP, NT, slot_)                                                                                                         \
    do {                                                                                                                                      \
        auto kern = dk == 128 ? adam_pack_all_kernel<PP, LP, NT, 128> : dk == 256 ? adam_pack_all_kernel<PP, LP, NT, 256>                     \
                  : dk == 64 ? adam_pack_all_kernel<PP, LP, NT, 64> : adam_pack_all_kernel<PP, LP, NT, 0>;                                   \
        static size_t attr_lds_k[4] = {0, 0, 0, 0};                                                                                           \
        size_t& attr_ref = attr_lds_k[dk == 128 ? 1 : dk == 256 ? 2 : dk == 64 ? 3 : 0];                                                       \
        (void)attr_lds;                                                                                                                       \
        if (lds > attr_ref) {                                                                                                                 \
            M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
            attr_ref = lds;                                                                                                                   \
        }                                                                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, pd, embed_wgs);                                                                 \
    } while (0)
    if (prec == PREC_BF16) {
        if (lowp) { if (ntmv) M2M_APA_GO(PREC_BF16, true, true, 0); else M2M_APA_GO(PREC_BF16, true, false, 1); }
        else { if (ntmv) M2M_APA_GO(PREC_BF16, false, true, 2); else M2M_APA_GO(PREC_BF16, false, false, 3); }
    } else {
        if (lowp) { if (ntmv) M2M_APA_GO(PREC_F32, true, true, 4); else M2M_APA_GO(PREC_F32, true, false, 5); }
        else { if (ntmv) M2M_APA_GO(PREC_F32, false, true, 6); else M2M_APA_GO(PREC_F32, false, false, 7); }
    }
#undef M2M_APA_GO
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

// One tiny launch at the head of a training step instead of three scattered through it (each tiny kernel costs
// 3-9 us on the critical path of a replayed graph): Adam step count += 1, dropout step counter += 1, losses = 0.
__global__ void step_prologue_kernel(float* adam_state, unsigned int* drop_counter, float* losses, int nlosses) {
    const int t = threadIdx.x;
    if (t == 0 && adam_state) adam_state[0] += 1.0f;
    if (t == 1 && drop_counter) *drop_counter += 1u;
    if (losses && t < nlosses) losses[t] = 0.f;
}
extern "C" int m2m_step_prologue(float* adam_state, uint32_t* drop_counter, float* losses, int nlosses, void* stream) {
    if (nlosses < 0 || nlosses > 64) { m2m_set_error("step_prologue: nlosses must be in [0, 64]", __FILE__, __LINE__); return -1; }
    hipLaunchKernelGGL(step_prologue_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), adam_state, drop_counter, losses, nlosses);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ void counter_add_kernel(unsigned int* c, unsigned int d) { *c += d; }
extern "C" int m2m_counter_add(uint32_t* counter, uint32_t delta, void* stream) {
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), counter, delta);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// probes (tests only)
// ---------------------------------------------------------------------------------------------------
__global__ void gelu_probe_kernel(const float* x, float* y, float* dy, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float a, b;
        gelu_grad_f(x[i], a, b);
        y[i] = gelu_f(x[i]);
        dy[i] = b;
        (void)a;
    }
}
extern "C" int m2m_gelu_probe(const float* x, float* y, float* dy, int64_t n, void* stream) {
    hipLaunchKernelGGL(gelu_probe_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, dy, (long)n);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

// Mirrors the kernels' mask functions.  mode 0: generic 16-bit draw per element index;
// mode 1 (token sites at p == 0.5): one word per row (row = sample*D + channel), bit = column;
// mode 2 (channel-hidden site): drop_keep_mc on (row, column).
__global__ void dropout_mask_kernel(unsigned int key, unsigned int thr, long n, unsigned int cols, int mode, uint8_t* mask) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        Drop d; d.key = key; d.thr = thr; d.scale = 1.f;
        bool k;
        if (mode == 1) {
            const unsigned int row = (unsigned int)(i / cols), col = (unsigned int)(i % cols), nw = (cols + 31u) >> 5;
            k = (mix32(key ^ (row * nw + (col >> 5))) >> (col & 31u)) & 1u;
        }
        else if (mode == 2) k = drop_keep_mc(d, (unsigned int)(i / cols), (unsigned int)(i % cols), cols);
        else k = drop_keep(d, (unsigned int)i);
        mask[i] = k ? 1 : 0;
    }
}
extern "C" int m2m_dropout_mask(const m2m_tower* t, int blk, int site, int B, uint32_t seed, uint32_t step, uint8_t* mask, void* stream) {
    if (int rc = m2m_check_tower(t, B)) return rc;
    if (site < 0 || site > 3 || blk < 0 || blk >= t->nblocks) { m2m_set_error("bad site/blk", __FILE__, __LINE__); return -1; }
    // element counts in kernel index order: 0 (B,D,T)  1 (B,D,N)  2 (B*N, Cp)  3 (B*N, D)
    long n = 0;
    if (site == 0) n = (long)B * t->D * t->T;
    if (site == 1) n = (long)B * t->D * t->N;
    if (site == 2) n = (long)B * t->N * t->Cp;
    if (site == 3) n = (long)B * t->N * t->D;
    const unsigned int key = m2m_site_key(seed, step, t->site_base + 4u * blk + site);
    const unsigned int thr = m2m_drop_thr(t->p_drop);
    unsigned int cols = 1;
    int mode = 0;
    if (site == 2) { cols = t->Cp; mode = 2; }
    else if (site == 0 && thr == 32768u) { cols = t->T; mode = 1; }
    else if (site == 1 && thr == 32768u) { cols = t->N; mode = 1; }
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       key, thr, n, cols, mode, mask);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

// C = A B^T with every operand going through the packed layouts; then C2 = C Bc^T with C's
// accumulators chained as the second product's A operand.  One wave per 16 rows of A.
template <int P>
__global__ void gemm_probe_kernel(const char* Ap /*NAT [i][k] k-minor*/, const char* Bp /*NAT [j][k] k-minor*/,
                                  const char* Bcp /*CHN [j2][k=j] k-major*/, int I, int J, int K, int J2, float* C, float* C2) {
    typedef Prec<P> Pr;
    const int lane = threadIdx.x & 63, g = lane >> 4, il = lane & 15;
    const int it = blockIdx.x;                 // 16-row tile of A
    const int nKB = (K + Pr::KB - 1) / Pr::KB;
    const int nJT = (J + 15) / 16;
    const int nJ2T = (J2 + 15) / 16;
    // swapped product: Ct[j][i] = B A^T so that the accumulator (rows j) chains into k = j
    for (int jp = 0; jp < (nJT + 1) / 2; ++jp) {
        f32x4_t acc[2];
        for (int t = 0; t < 2; ++t) {
            acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int jt = 2 * jp + t;
            if (jt < nJT)
                for (int kb = 0; kb < nKB; ++kb) {
                    const Frag b = ld_frag_global(Bp, (long)jt * nKB + kb, lane);
                    const Frag a = ld_frag_global(Ap, (long)it * nKB + kb, lane);
                    Pr::mma(acc[t], b, a);
                }
            // acc[t][r] = C[i = 16 it + il][j = 16 jt + 4g + r]
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + il, j = 16 * jt + 4 * g + r;
                if (jt < nJT && i < I && j < J) C[(long)i * J + j] = acc[t][r];
            }
        }
        if (C2) {
            Frag hf[Chain<P>::NF];
            Chain<P>::make(acc[0], acc[1], hf);
            for (int f = 0; f < Chain<P>::NF; ++f)
                for (int j2t = 0; j2t < nJ2T; ++j2t) {
                    const Frag w = ld_frag_global(Bcp, (long)(jp * Chain<P>::NF + f) * nJ2T + j2t, lane);
                    f32x4_t o = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    Pr::mma(o, hf[f], w);
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * it + 4 * g + r, j2 = 16 * j2t + il;
                        if (i < I && j2 < J2) atomicAdd(&C2[(long)i * J2 + j2], o[r]);
                    }
                }
        }
    }
}

extern "C" int m2m_gemm_probe(int prec, const float* A, const float* Bm, int I, int J, int K, const float* Bc, int J2,
                              float* C, float* C2, void* workspace, void* stream) {
    char* ws = reinterpret_cast<char*>(workspace);
    const int64_t ab = m2m_packed_bytes(prec, I, K), bb = m2m_packed_bytes(prec, J, K);
    char* Ap = ws; char* Bp = ws + ab; char* Bcp = Bp + bb;
    int rc;
    if ((rc = m2m_pack(prec, PACK_NAT, 0, A, K, 1, I, K, Ap, stream))) return rc;
    if ((rc = m2m_pack(prec, PACK_NAT, 0, Bm, K, 1, J, K, Bp, stream))) return rc;
    if (Bc && (rc = m2m_pack(prec, PACK_CHN, 1, Bc, J, 1, J2, J, Bcp, stream))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (C2) M2M_CHECK_HIP(hipMemsetAsync(C2, 0, sizeof(float) * (size_t)I * J2, st));
    const int grid = (I + 15) / 16;
    if (prec == PREC_BF16)
        hipLaunchKernelGGL(gemm_probe_kernel<PREC_BF16>, dim3(grid), dim3(64), 0, st, Ap, Bp, Bc ? Bcp : nullptr, I, J, K, J2, C, Bc ? C2 : nullptr);
    else
        hipLaunchKernelGGL(gemm_probe_kernel<PREC_F32>, dim3(grid), dim3(64), 0, st, Ap, Bp, Bc ? Bcp : nullptr, I, J, K, J2, C, Bc ? C2 : nullptr);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- shader clock under load (include/m2mixer.h: m2m_clock_probe) ---------------------------------------------------------
// Every wave runs the same bounded loop: 32 bf16 MFMAs + a few VALU instructions per trip, the wall clock read once per trip;
// the loop ends when spin_ticks have passed (an exit condition every wave reaches: the 100 MHz counter always advances).
__global__ __launch_bounds__(512) void clock_probe_kernel(unsigned long long* __restrict__ out, unsigned int spin_ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    Frag a, b;
    const unsigned int seed = mix32(threadIdx.x * 2654435761u + blockIdx.x);
    a.u = u32x4_t{0x3F803F80u ^ (seed & 0x00070007u), 0x3F003F00u, 0x3E803E80u ^ ((seed >> 8) & 0x00030003u), 0x3F803F00u};
    b.u = u32x4_t{0x3F003F80u, 0x3E803F00u ^ ((seed >> 16) & 0x00070007u), 0x3F803E80u, 0x3F003F00u};
    f32x4_t acc[4] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    float v = (float)(seed & 1023u) * 1e-3f;
    unsigned long long r1 = r0;
    for (int guard = 0; guard < (1 << 22); ++guard) {                 // (hard bound on top of the time condition)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc[j], 0, 0, 0);
            v = __builtin_fmaf(v, 0.999f, 0.001f);
            v = __builtin_fmaf(v, 1.001f, -0.001f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] * 0.5f;           // keep the sums finite
        r1 = __builtin_amdgcn_s_memrealtime();
        if (r1 - r0 >= (unsigned long long)spin_ticks) break;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float sink = v;
#pragma unroll
    for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = (r1 - r0) | (sink == 12345.678f ? 1ull << 63 : 0ull);     // (the sink keeps the work alive)
    }
}
extern "C" int m2m_clock_probe(uint64_t* out, int nwg, int spin_ticks, void* stream) {
    if (!out || nwg < 1 || nwg > 4096 || spin_ticks < 1) { m2m_set_error("clock_probe: bad arguments", __FILE__, __LINE__); return -1; }
    if (spin_ticks > 1000000) spin_ticks = 1000000;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(nwg), dim3(512), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<unsigned long long*>(out), (unsigned int)spin_ticks);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
