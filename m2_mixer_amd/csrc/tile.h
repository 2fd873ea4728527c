// Per-workgroup token-tile helpers shared by the forward and backward tower kernels.
//
// A workgroup (256 threads = 4 waves, one per SIMD) owns BM = 64 token rows = SPW whole samples of
// N tokens (token mixing couples the N tokens of a sample, channel mixing is row-wise), keeps the
// fp32 residual stream of those rows in LDS for the whole tower and streams the weights past it.
#pragma once
#include "common.h"
#include "../../include/m2mixer.h"

#define BM 64
#define MT (BM / 16)
#define NTHREADS 256

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

template <int D> struct TileGeom {
    static constexpr int XLD = D + 4;          // padded fp32 row stride (floats)
    static constexpr int DT = D / 16;
    static constexpr int CPT = D / 4;          // columns per LayerNorm thread (4 threads per row)
};

// Row statistics of the fp32 tile `x` (BM rows, stride XLD): thread (r = tid>>2, j = tid&3) owns the
// float4 chunks at columns 16*i + 4*j.  Two-pass (mean, then centred variance), biased variance, eps 1e-5.
// v[] receives the thread's raw values.
template <int D>
static __device__ __forceinline__ void row_stats(const float* x, int tid, float v[D / 4], float& mean, float& rstd) {
    constexpr int XLD = TileGeom<D>::XLD;
    const int r = tid >> 2, j = tid & 3;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < D / 16; ++i) {
        const float4 q = *reinterpret_cast<const float4*>(x + r * XLD + 16 * i + 4 * j);
        v[4 * i + 0] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        s += (q.x + q.y) + (q.z + q.w);
    }
    s = wave_sum_xor(s, 4);
    mean = s * (1.0f / D);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < D / 4; ++i) { const float c = v[i] - mean; s2 = __builtin_fmaf(c, c, s2); }
    s2 = wave_sum_xor(s2, 4);
    rstd = __builtin_amdgcn_rsqf(s2 * (1.0f / D) + 1e-5f);
    // one Newton step: v_rsq_f32 is ~1 ulp, the parity mode wants full fp32
    const float vv = s2 * (1.0f / D) + 1e-5f;
    rstd = rstd * (1.5f - 0.5f * vv * rstd * rstd);
}

// column index of element e (0..D/4-1) of thread j
static __device__ __forceinline__ int ln_col(int e, int j) { return 16 * (e >> 2) + 4 * j + (e & 3); }

// LayerNorm of every row of x into the fp32 tile `dst` (same geometry).
template <int D>
static __device__ __forceinline__ void ln_to_tile(const float* x, float* dst, const float* gamma, const float* beta, int tid) {
    constexpr int XLD = TileGeom<D>::XLD;
    float v[D / 4], mean, rstd;
    row_stats<D>(x, tid, v, mean, rstd);
    const int r = tid >> 2, j = tid & 3;
#pragma unroll
    for (int i = 0; i < D / 16; ++i) {
        const int c = 16 * i + 4 * j;
        const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = (v[4 * i + 0] - mean) * rstd * gm.x + bt.x;
        o.y = (v[4 * i + 1] - mean) * rstd * gm.y + bt.y;
        o.z = (v[4 * i + 2] - mean) * rstd * gm.z + bt.z;
        o.w = (v[4 * i + 3] - mean) * rstd * gm.w + bt.w;
        *reinterpret_cast<float4*>(dst + r * XLD + c) = o;
    }
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
#pragma unroll 2
    for (int slot = tid; slot < MT * KD * 64; slot += NTHREADS) {
        const int blk = slot >> 6;
        *reinterpret_cast<u32x4_t*>(img + slot * 16) =
            gather_slot<P>(tile, TileGeom<D>::XLD, PACK_NAT, false, blk / KD, blk % KD, slot & 63);
    }
}
// fp32 tile [BM][XLD] -> packed CHN image of the TRANSPOSE, X[i = d][k = m], blocks ordered [kb over m][dt]
template <int P, int D>
static __device__ __forceinline__ void pack_tile_chn_t(const float* tile, char* img, int tid) {
    constexpr int DT = D / 16, NKM = BM / Prec<P>::KB;
#pragma unroll 2
    for (int slot = tid; slot < NKM * DT * 64; slot += NTHREADS) {
        const int blk = slot >> 6;
        *reinterpret_cast<u32x4_t*>(img + slot * 16) =
            gather_slot<P>(tile, TileGeom<D>::XLD, PACK_CHN, true, blk % DT, blk / DT, slot & 63);
    }
}

// copy a packed 64-row tile image (bytes) between LDS and global, 16 bytes per thread step
static __device__ __forceinline__ void copy16(char* dst, const char* src, int bytes, int tid) {
    for (int o = tid * 16; o < bytes; o += NTHREADS * 16)
        *reinterpret_cast<u32x4_t*>(dst + o) = *reinterpret_cast<const u32x4_t*>(src + o);
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
