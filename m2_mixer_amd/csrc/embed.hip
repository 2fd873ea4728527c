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
#include "embed_wgrad.h"

#ifndef EMB_KS
#define EMB_KS 128          // k extent staged per step (floats)
#endif
#define EMB_LD (EMB_KS + 4)
#define EMB_KMAX 4096       // largest padded K the offset table holds (AV-MNIST audio 3136, MM-IMDb 3072)

template <int P, int D, int RB>
static __device__ __forceinline__ void embed_fwd_body(const m2m_embed& em, const float* __restrict__ in, long M, int N,
                                                      float* __restrict__ x0, int wg, char* smem) {
    typedef Prec<P> Pr;
    constexpr int DT = D / 16, KSB = EMB_KS / Pr::KB;     // k-blocks per stage
    constexpr int DPW = (DT + NWAVES - 1) / NWAVES;        // d-tiles per wave
    float* tile = reinterpret_cast<float*>(smem);          // [RB][EMB_LD] fp32
    char* img = smem + RB * EMB_LD * 4;                    // packed NAT [mt][kb] of the stage
    int* koff = reinterpret_cast<int*>(img + RB * EMB_KS * Pr::ESZ);   // [Kp rounded up to EMB_KS]
    long* rbase = reinterpret_cast<long*>(koff + EMB_KMAX);            // [RB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    PatchGeom pg{em.Cin, em.H, em.W, em.ph, em.pw, em.W / em.pw, N, em.K};
    const long m0 = (long)wg * RB;
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

template <int P, int D, int RB>
__global__ __launch_bounds__(NTHREADS) void embed_fwd_kernel(const m2m_embed em, const float* __restrict__ in, long M, int N,
                                                             float* __restrict__ x0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    embed_fwd_body<P, D, RB>(em, in, M, N, x0, blockIdx.x, smem);
}

// Both patch embeddings of a two-tower model in ONE launch (no fork / join of a second stream at the head of the step):
// workgroups [0, nwg0) serve embedding 0 -- the one with the larger patch, dispatched first -- the rest embedding 1.
struct EmbedFwdGroupArgs {
    m2m_embed em[2];
    const float* in[2];
    float* x0[2];
    long M[2];
    int N[2], nwg0;
};
template <int P, int D, int RB>
__global__ __launch_bounds__(NTHREADS) void embed_fwd_group_kernel(const EmbedFwdGroupArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int e = (int)blockIdx.x < a.nwg0 ? 0 : 1;
    embed_fwd_body<P, D, RB>(a.em[e], a.in[e], a.M[e], a.N[e], a.x0[e], e ? blockIdx.x - a.nwg0 : blockIdx.x, smem);
}

template <int P, int D>
__global__ __launch_bounds__(NTHREADS) void embed_wgrad_kernel(const m2m_embed em, const float* __restrict__ in,
                                                               const float* __restrict__ dx0, long M, int N, int tiles_per_group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    embed_wgrad_body<P, D, NTHREADS>(em, in, dx0, M, N, tiles_per_group, blockIdx.x, blockIdx.y, gridDim.y == 1, smem);
}

template <int P, int D>
__global__ __launch_bounds__(NTHREADS) void embed_wgrad_group_kernel(const EmbedWgradGroupArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    embed_wgrad_group_body<P, D, NTHREADS>(a, blockIdx.x, smem);
}

int m2m_check_embed(const m2m_embed* e, int B) {
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
static int launch_embed_fwd_group(const m2m_embed* const* es, const float* const* ins, float* const* x0s, int B, hipStream_t st) {
    constexpr int RB = 16;
    EmbedFwdGroupArgs a;
    memset(&a, 0, sizeof(a));
    const int first = es[1]->Kp > es[0]->Kp ? 1 : 0;
    int total = 0;
    for (int k = 0; k < 2; ++k) {
        const int i = k == 0 ? first : 1 - first;
        const int N = (es[i]->H / es[i]->ph) * (es[i]->W / es[i]->pw);
        a.em[k] = *es[i]; a.in[k] = ins[i]; a.x0[k] = x0s[i]; a.N[k] = N; a.M[k] = (long)B * N;
        const int nwg = (int)((a.M[k] + RB - 1) / RB);
        if (k == 0) a.nwg0 = nwg;
        total += nwg;
    }
    const size_t lds = (size_t)RB * EMB_LD * 4 + (size_t)RB * EMB_KS * Prec<P>::ESZ + EMB_KMAX * 4 + RB * 8;
    auto kern = embed_fwd_group_kernel<P, D, RB>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NTHREADS), lds, st, a);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int P, int D>
static int launch_embed_wgrad(const m2m_embed* e, const float* in, const float* dx0, int B, hipStream_t st) {
    const EmbedWgradPlan pl = embed_wgrad_plan(e, B, 256);
    const size_t lds = embed_wgrad_lds<D, P>();
    auto kern = embed_wgrad_kernel<P, D>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)pl.nchunks, (unsigned)pl.groups), dim3(NTHREADS), lds, st, *e, in, dx0, pl.M, pl.N, pl.tpg);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int D>
static int launch_embed_wgrad_group(const m2m_embed* const* es, const float* const* ins, const float* const* dx0s, int B, hipStream_t st) {
    EmbedWgradGroupArgs a;
    const int total = embed_wgrad_group_args(a, es, ins, dx0s, B, 256);
    const size_t lds = embed_wgrad_lds<D, P>();
    auto kern = embed_wgrad_group_kernel<P, D>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NTHREADS), lds, st, a);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int m2m_embed_forward(const m2m_embed* e, const float* input, int B, float* x0, void* stream) {
    if (int rc = m2m_check_embed(e, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_EF_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_fwd<PP, DD>(e, input, B, x0, st);
    M2M_EF_CASE(PREC_BF16, 32) M2M_EF_CASE(PREC_BF16, 64) M2M_EF_CASE(PREC_BF16, 128) M2M_EF_CASE(PREC_BF16, 256)
    M2M_EF_CASE(PREC_F32, 32) M2M_EF_CASE(PREC_F32, 64) M2M_EF_CASE(PREC_F32, 128) M2M_EF_CASE(PREC_F32, 256)
#undef M2M_EF_CASE
    m2m_set_error("embed_forward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_embed_wgrad(const m2m_embed* e, const float* input, const float* d_x0, int B, void* stream) {
    if (int rc = m2m_check_embed(e, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_EW_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_wgrad<PP, DD>(e, input, d_x0, B, st);
    M2M_EW_CASE(PREC_BF16, 32) M2M_EW_CASE(PREC_BF16, 64) M2M_EW_CASE(PREC_BF16, 128) M2M_EW_CASE(PREC_BF16, 256)
    M2M_EW_CASE(PREC_F32, 32) M2M_EW_CASE(PREC_F32, 64) M2M_EW_CASE(PREC_F32, 128) M2M_EW_CASE(PREC_F32, 256)
#undef M2M_EW_CASE
    m2m_set_error("embed_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_embeds_wgrad(const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s, int nembeds,
                                int B, void* stream) {
    if (!embeds || !inputs || !d_x0s || nembeds != EMB_GROUP) { m2m_set_error("embeds_wgrad: exactly two embeddings", __FILE__, __LINE__); return -1; }
    for (int i = 0; i < nembeds; ++i) {
        if (int rc = m2m_check_embed(embeds[i], B)) return rc;
        if (embeds[i]->prec != embeds[0]->prec || embeds[i]->D != embeds[0]->D) {
            m2m_set_error("embeds_wgrad: the embeddings of one launch must share precision and hidden_dim", __FILE__, __LINE__);
            return -1;
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const m2m_embed* e = embeds[0];
#define M2M_EWG_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_wgrad_group<PP, DD>(embeds, inputs, d_x0s, B, st);
    M2M_EWG_CASE(PREC_BF16, 32) M2M_EWG_CASE(PREC_BF16, 64) M2M_EWG_CASE(PREC_BF16, 128) M2M_EWG_CASE(PREC_BF16, 256)
    M2M_EWG_CASE(PREC_F32, 32) M2M_EWG_CASE(PREC_F32, 64) M2M_EWG_CASE(PREC_F32, 128) M2M_EWG_CASE(PREC_F32, 256)
#undef M2M_EWG_CASE
    m2m_set_error("embeds_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_embeds_forward(const m2m_embed* const* embeds, const float* const* inputs, float* const* x0s, int nembeds, int B,
                                  void* stream) {
    if (!embeds || !inputs || !x0s || nembeds != 2) { m2m_set_error("embeds_forward: exactly two embeddings", __FILE__, __LINE__); return -1; }
    for (int i = 0; i < nembeds; ++i) {
        if (int rc = m2m_check_embed(embeds[i], B)) return rc;
        if (embeds[i]->prec != embeds[0]->prec || embeds[i]->D != embeds[0]->D) {
            m2m_set_error("embeds_forward: the embeddings of one launch must share precision and hidden_dim", __FILE__, __LINE__);
            return -1;
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const m2m_embed* e = embeds[0];
#define M2M_EFG_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_fwd_group<PP, DD>(embeds, inputs, x0s, B, st);
    M2M_EFG_CASE(PREC_BF16, 32) M2M_EFG_CASE(PREC_BF16, 64) M2M_EFG_CASE(PREC_BF16, 128) M2M_EFG_CASE(PREC_BF16, 256)
    M2M_EFG_CASE(PREC_F32, 32) M2M_EFG_CASE(PREC_F32, 64) M2M_EFG_CASE(PREC_F32, 128) M2M_EFG_CASE(PREC_F32, 256)
#undef M2M_EFG_CASE
    m2m_set_error("embeds_forward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}
