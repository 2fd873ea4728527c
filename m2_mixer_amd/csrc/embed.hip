// Patch embedding (modules/mixer.py:143-146) and the plain input projection of MLPMixerNoPatching
// (modules/mixer.py:171,180), forward and weight gradient.
//
//   x0[m][d] = sum_k patch[m][k] W[d][k] + b[d],   m = b*N + (gy*GW + gx),  k = c*ph*pw + py*pw + px
//
// The conv with stride == kernel is an unfold + GEMM; the unfold is done on the fly while staging the
// input (read once, coalesced along image rows) into LDS -- address = rowbase[m] + koff[k] from two
// small LDS tables, so no integer division in the streaming loop -- and the weight fragments come
// straight from the packed NAT copy in global memory.
#include "tile.h"

#define EBM 32             // token rows per workgroup
#define EMT (EBM / 16)
#define EMB_KS 128          // k extent staged per step (floats)
#define EMB_LD (EMB_KS + 4)
#define EMB_KMAX 4096       // largest padded K the offset table holds (AV-MNIST audio 3136, MM-IMDb 3072)

struct PatchGeom {
    int Cin, H, W, ph, pw, GW, N, K;
};
// offset of element k of a patch relative to the patch origin; -1 beyond K
static __device__ __forceinline__ int patch_koff(const PatchGeom& pg, int k) {
    if (k >= pg.K) return -1;
    const int c = k / (pg.ph * pg.pw), rem = k % (pg.ph * pg.pw);
    const int py = rem / pg.pw, px = rem % pg.pw;
    return (c * pg.H + py) * pg.W + px;
}
// offset of the origin of token row m's patch; -1 beyond M
static __device__ __forceinline__ long patch_rowbase(const PatchGeom& pg, long m, long M) {
    if (m >= M) return -1;
    const long b = m / pg.N;
    const int n = (int)(m % pg.N);
    const int gy = n / pg.GW, gx = n % pg.GW;
    return (b * pg.Cin * pg.H + gy * pg.ph) * (long)pg.W + gx * pg.pw;
}

template <int P, int D, int RB>
__global__ __launch_bounds__(NTHREADS) void embed_fwd_kernel(const m2m_embed em, const float* __restrict__ in, long M, int N,
                                                             float* __restrict__ x0) {
    typedef Prec<P> Pr;
    constexpr int DT = D / 16, KSB = EMB_KS / Pr::KB;     // k-blocks per stage
    constexpr int DPW = (DT + NWAVES - 1) / NWAVES;        // d-tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);          // [RB][EMB_LD] fp32
    char* img = smem + RB * EMB_LD * 4;                    // packed NAT [mt][kb] of the stage
    int* koff = reinterpret_cast<int*>(img + RB * EMB_KS * Pr::ESZ);   // [Kp rounded up to EMB_KS]
    long* rbase = reinterpret_cast<long*>(koff + EMB_KMAX);            // [RB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    PatchGeom pg{em.Cin, em.H, em.W, em.ph, em.pw, em.W / em.pw, N, em.K};
    const long m0 = (long)blockIdx.x * RB;
    const int nKB = em.Kp / Pr::KB;
    const int kext = (em.Kp + EMB_KS - 1) / EMB_KS * EMB_KS;
    for (int k = tid; k < kext; k += NTHREADS) koff[k] = patch_koff(pg, k);
    if (tid < RB) rbase[tid] = patch_rowbase(pg, m0 + tid, M);

    f32x4_t acc[(RB / 16)][DPW];
#pragma unroll
    for (int mt = 0; mt < (RB / 16); ++mt)
#pragma unroll
        for (int j = 0; j < DPW; ++j) acc[mt][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Software pipeline over EMB_KS-wide stages: the global loads of stage s+1 (this thread's patch elements and
    // this wave's weight fragments) are issued before stage s is packed and multiplied, so their latency hides
    // behind the LDS work and the MFMAs.  One workgroup owns its rows for the whole K: deterministic, no atomics.
    constexpr int EPT = RB * EMB_KS / NTHREADS;            // patch elements per thread per stage
    float pre[EPT];
    Frag wpre[DPW][KSB];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int idx = i * NTHREADS + tid;
            const int r = idx / EMB_KS, kk = idx % EMB_KS;
            const long rb = rbase[r];
            const int ko = koff[k0 + kk];
            pre[i] = (rb >= 0 && ko >= 0) ? in[rb + ko] : 0.f;
        }
        const int kb0 = k0 / Pr::KB;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int dt = wave + NWAVES * j;
#pragma unroll
            for (int kb = 0; kb < KSB; ++kb) {
                wpre[j][kb].u = u32x4_t{0u, 0u, 0u, 0u};
                if (dt < DT && kb0 + kb < nKB) wpre[j][kb] = ld_frag_global(em.wn, (long)dt * nKB + kb0 + kb, lane);
            }
        }
    };
    __syncthreads();                                         // offset tables ready
    load_stage(0);
    for (int k0 = 0; k0 < em.Kp; k0 += EMB_KS) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int idx = i * NTHREADS + tid;
            tile[(idx / EMB_KS) * EMB_LD + idx % EMB_KS] = pre[i];
        }
        Frag wcur[DPW][KSB];
#pragma unroll
        for (int j = 0; j < DPW; ++j)
#pragma unroll
            for (int kb = 0; kb < KSB; ++kb) wcur[j][kb] = wpre[j][kb];
        __syncthreads();
        if (k0 + EMB_KS < em.Kp) load_stage(k0 + EMB_KS);
        for (int slot = tid; slot < (RB / 16) * KSB * 64; slot += NTHREADS) {
            const int blk = slot >> 6;
            *reinterpret_cast<u32x4_t*>(img + slot * 16) =
                gather_slot<P>(tile, EMB_LD, PACK_NAT, false, blk / KSB, blk % KSB, slot & 63);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int dt = wave + NWAVES * j;
            if (dt < DT) {
#pragma unroll
                for (int kb = 0; kb < KSB; ++kb) {
#pragma unroll
                    for (int mt = 0; mt < (RB / 16); ++mt) {
                        const Frag a = ld_frag_lds(img, mt * KSB + kb, lane);
                        Pr::mma(acc[mt][j], a, wcur[j][kb]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int dt = wave + NWAVES * j;
        if (dt < DT) {
            const int d = 16 * dt + il;
            const float bv = em.b[d];
#pragma unroll
            for (int mt = 0; mt < (RB / 16); ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long m = m0 + 16 * mt + 4 * g + r;
                    if (m < M) x0[m * D + d] = acc[mt][j][r] + bv;
                }
        }
    }
}

// g_w[d][k] += sum_m dx0[m][d] patch[m][k];  workgroup = 64 k columns x one group of row tiles.
template <int P, int D>
__global__ __launch_bounds__(NTHREADS) void embed_wgrad_kernel(const m2m_embed em, const float* __restrict__ in,
                                                               const float* __restrict__ dx0, long M, int N, int tiles_per_group) {
    typedef Prec<P> Pr;
    constexpr int DT = D / 16, NKM = EBM / Pr::KB, XLD = TileGeom<D>::XLD;
    constexpr int KC = 64, KCT = KC / 16, PLD = KC + 4;
    constexpr int DPW = (DT + NWAVES - 1) / NWAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dxt = reinterpret_cast<float*>(smem);                  // [EBM][XLD]   dx0 tile
    float* pt = dxt + EBM * XLD;                                    // [EBM][PLD]   patch tile
    char* aimg = reinterpret_cast<char*>(pt + EBM * PLD);           // NAT X[i=d][k=m]  blocks [dt][kbm]
    char* bimg = aimg + EBM * D * Pr::ESZ;                          // NAT X[i=kk][k=m] blocks [kt][kbm]
    int* koff = reinterpret_cast<int*>(bimg + EBM * KC * Pr::ESZ);  // [KC]
    long* rbase = reinterpret_cast<long*>(koff + KC);              // [EBM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    PatchGeom pg{em.Cin, em.H, em.W, em.ph, em.pw, em.W / em.pw, N, em.K};
    const int k0 = blockIdx.x * KC;
    if (tid < KC) koff[tid] = patch_koff(pg, k0 + tid);

    f32x4_t acc[DPW][KCT];
#pragma unroll
    for (int j = 0; j < DPW; ++j)
#pragma unroll
        for (int kt = 0; kt < KCT; ++kt) acc[j][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                               // bias gradient (k-chunk 0 only), thread d

    const long ntiles = (M + EBM - 1) / EBM;
    const long t_begin = (long)blockIdx.y * tiles_per_group;
    const long t_end = min(ntiles, t_begin + tiles_per_group);
    for (long tl = t_begin; tl < t_end; ++tl) {
        const long m0 = tl * EBM;
        __syncthreads();
        if (tid < EBM) rbase[tid] = patch_rowbase(pg, m0 + tid, M);
        for (int idx = tid; idx < EBM * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < M) v = *reinterpret_cast<const float4*>(dx0 + (m0 + r) * D + c);
            *reinterpret_cast<float4*>(dxt + r * XLD + c) = v;
        }
        __syncthreads();
        for (int idx = tid; idx < EBM * KC; idx += NTHREADS) {
            const int r = idx / KC, kk = idx % KC;
            const long rb = rbase[r];
            const int ko = koff[kk];
            pt[r * PLD + kk] = (rb >= 0 && ko >= 0) ? in[rb + ko] : 0.f;
        }
        if (blockIdx.x == 0 && tid < D) {
            float s = 0.f;
            for (int r = 0; r < EBM; ++r) s += dxt[r * XLD + tid];
            bsum += s;
        }
        __syncthreads();
        for (int slot = tid; slot < DT * NKM * 64; slot += NTHREADS) {
            const int blk = slot >> 6;
            *reinterpret_cast<u32x4_t*>(aimg + slot * 16) =
                gather_slot<P>(dxt, XLD, PACK_NAT, true, blk / NKM, blk % NKM, slot & 63);
        }
        for (int slot = tid; slot < KCT * NKM * 64; slot += NTHREADS) {
            const int blk = slot >> 6;
            *reinterpret_cast<u32x4_t*>(bimg + slot * 16) =
                gather_slot<P>(pt, PLD, PACK_NAT, true, blk / NKM, blk % NKM, slot & 63);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int dt = wave + NWAVES * j;
            if (dt < DT) {
#pragma unroll
                for (int kbm = 0; kbm < NKM; ++kbm) {
                    const Frag a = ld_frag_lds(aimg, dt * NKM + kbm, lane);
#pragma unroll
                    for (int kt = 0; kt < KCT; ++kt) {
                        const Frag b = ld_frag_lds(bimg, kt * NKM + kbm, lane);
                        Pr::mma(acc[j][kt], a, b);
                    }
                }
            }
        }
    }
    const bool single = gridDim.y == 1;
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int dt = wave + NWAVES * j;
        if (dt < DT) {
#pragma unroll
            for (int kt = 0; kt < KCT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * dt + 4 * g + r, k = k0 + 16 * kt + il;
                    if (k < em.K) {
                        float* p = em.g_w + (long)d * em.K + k;
                        if (single) *p += acc[j][kt][r]; else atomicAdd(p, acc[j][kt][r]);
                    }
                }
        }
    }
    if (blockIdx.x == 0 && tid < D) { if (single) em.g_b[tid] += bsum; else atomicAdd(em.g_b + tid, bsum); }
}

static int check_embed(const m2m_embed* e, int B) {
    if (!e || B < 1) { m2m_set_error("embed: bad argument", __FILE__, __LINE__); return -1; }
    if (e->H % e->ph || e->W % e->pw) { m2m_set_error("embed: image not divisible by patch", __FILE__, __LINE__); return -1; }
    if (e->K != e->Cin * e->ph * e->pw) { m2m_set_error("embed: K != Cin*ph*pw", __FILE__, __LINE__); return -1; }
    const int KB = e->prec == PREC_BF16 ? 32 : 16;
    if (e->Kp % KB || e->Kp < e->K) { m2m_set_error("embed: Kp must be K rounded up to the k-block", __FILE__, __LINE__); return -1; }
    if (e->Kp > EMB_KMAX - EMB_KS) { m2m_set_error("embed: patch too large (Cin*ph*pw must be <= 3968)", __FILE__, __LINE__); return -1; }
    return 0;
}

template <int P, int D>
static int launch_embed_fwd(const m2m_embed* e, const float* in, int B, float* x0, hipStream_t st) {
    const int N = (e->H / e->ph) * (e->W / e->pw);
    const long M = (long)B * N;
    // 16 rows per workgroup: the audio embedding (2048 rows at batch 512, 50 KB of input per sample) sits at the head of
    // the step's critical path, and 64 workgroups of 32 rows left three quarters of the chip idle
    constexpr int RB = 16;
    const size_t lds = (size_t)RB * EMB_LD * 4 + (size_t)RB * EMB_KS * Prec<P>::ESZ + EMB_KMAX * 4 + RB * 8;
    auto kern = embed_fwd_kernel<P, D, RB>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)((M + RB - 1) / RB)), dim3(NTHREADS), lds, st, *e, in, M, N, x0);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int D>
static int launch_embed_wgrad(const m2m_embed* e, const float* in, const float* dx0, int B, hipStream_t st) {
    const int N = (e->H / e->ph) * (e->W / e->pw);
    const long M = (long)B * N;
    const int nchunks = (e->K + 63) / 64;
    const long ntiles = (M + EBM - 1) / EBM;
    long groups = (128 + nchunks - 1) / nchunks;           // ~128 workgroups; row groups add with atomics
    if (groups > ntiles / 4) groups = ntiles / 4;
    if (groups < 1) groups = 1;
    const int tpg = (int)((ntiles + groups - 1) / groups);
    groups = (ntiles + tpg - 1) / tpg;
    const size_t lds = (size_t)EBM * TileGeom<D>::XLD * 4 + (size_t)EBM * 68 * 4 + (size_t)EBM * D * Prec<P>::ESZ +
                       (size_t)EBM * 64 * Prec<P>::ESZ + 64 * 4 + EBM * 8;
    auto kern = embed_wgrad_kernel<P, D>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nchunks, (unsigned)groups), dim3(NTHREADS), lds, st, *e, in, dx0, M, N, tpg);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int m2m_embed_forward(const m2m_embed* e, const float* input, int B, float* x0, void* stream) {
    if (int rc = check_embed(e, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_EF_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_fwd<PP, DD>(e, input, B, x0, st);
    M2M_EF_CASE(PREC_BF16, 32) M2M_EF_CASE(PREC_BF16, 64) M2M_EF_CASE(PREC_BF16, 128) M2M_EF_CASE(PREC_BF16, 256)
    M2M_EF_CASE(PREC_F32, 32) M2M_EF_CASE(PREC_F32, 64) M2M_EF_CASE(PREC_F32, 128) M2M_EF_CASE(PREC_F32, 256)
#undef M2M_EF_CASE
    m2m_set_error("embed_forward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_embed_wgrad(const m2m_embed* e, const float* input, const float* d_x0, int B, void* stream) {
    if (int rc = check_embed(e, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_EW_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_wgrad<PP, DD>(e, input, d_x0, B, st);
    M2M_EW_CASE(PREC_BF16, 32) M2M_EW_CASE(PREC_BF16, 64) M2M_EW_CASE(PREC_BF16, 128) M2M_EW_CASE(PREC_BF16, 256)
    M2M_EW_CASE(PREC_F32, 32) M2M_EW_CASE(PREC_F32, 64) M2M_EW_CASE(PREC_F32, 128) M2M_EW_CASE(PREC_F32, 256)
#undef M2M_EW_CASE
    m2m_set_error("embed_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}
