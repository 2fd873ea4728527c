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

// Fast path of the forward (bf16, patch rows of a multiple of 8 pixels, 16-byte aligned image rows: the AV-MNIST audio
// spectrogram, 56 x 56 patches of a 112 x 112 image).  A packed NAT slot is 8 consecutive k of one token row = 8
// consecutive pixels of one patch row, so every thread loads its slot's 32 bytes straight from the image, converts and
// writes the 16-byte slot: no fp32 staging tile, no offset tables, one barrier per 256-wide stage (the packed stage is
// double-buffered), and a register ring keeps EMB_FDEPTH stages of loads in flight (the generic path has one 128-wide stage
// in flight and spends its time on per-element LDS table lookups).  Audio embedding at batch 512: 25 -> 19 us, of which
// ~7 us are fixed (launch, first loads, epilogue), ~5 us the 25.7 MB of input at the HBM roofline and ~6 us the packed weight
// streamed from L2 by every workgroup (measured by removing either stream).
#define EMB_FKS 256
#define EMB_FDEPTH 3
template <int D>
static __device__ __forceinline__ void embed_fwd_fast_body(const m2m_embed& em, const float* __restrict__ in, long M, int N,
                                                           float* __restrict__ x0, int wg, int split, int nsplit, char* smem) {
    typedef Prec<PREC_BF16> Pr;
    constexpr int KSB = EMB_FKS / 32, DT = D / 16, DPW = (DT + NWAVES - 1) / NWAVES, IMG_B = 16 * EMB_FKS * 2;
    static_assert(KSB == NWAVES, "one k-block of the stage per wave");
    char* img = smem;                                       // [2][16 rows x EMB_FKS] packed NAT, blocks [kb]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    PatchGeom pg{em.Cin, em.H, em.W, em.ph, em.pw, em.W / em.pw, N, em.K};
    const long m0 = (long)wg * 16;
    const long rb = patch_rowbase(pg, m0 + il, M);          // this thread's token row (slot row il), -1 beyond M
    const int nKB = em.Kp / 32;
    // k-split: `nsplit` workgroups share a row tile, each contracting a contiguous range of stages into its own partial
    // output (x0 points at this split's part; the tower forward adds the parts).  The loop is bound by streaming the
    // packed weight (D x Kp bf16 per workgroup, ~32 B/clk per CU): splitting K halves that stream per workgroup and fills
    // the chip (batch 512: 128 row tiles on 256 CUs).
    const int nst_all = (em.Kp + EMB_FKS - 1) / EMB_FKS;
    const int per = (nst_all + nsplit - 1) / nsplit;
    const int st_begin = split * per, nst = min(nst_all, st_begin + per);

    f32x4_t acc[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    struct Pre {
        f32x4_t p0, p1;
        Frag w[DPW][KSB];
    };
    // Every load is unconditional (indices clamped into range; a stage past the end re-reads the last one and its patch
    // slot is zeroed at use), so the waits in the loop are counted.
    auto load = [&](Pre& p, int st) {
        const int k = min(st * EMB_FKS + wave * 32 + 8 * g, em.K - 8);
        const float* src = in + (rb >= 0 ? rb : 0) + patch_koff(pg, k);
        p.p0 = *reinterpret_cast<const f32x4_t*>(src);
        p.p1 = *reinterpret_cast<const f32x4_t*>(src + 4);
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int dt = min(wave + NWAVES * j, DT - 1);
#pragma unroll
            for (int kb = 0; kb < KSB; ++kb) p.w[j][kb] = ld_frag_global(em.wn, (long)dt * nKB + min(st * KSB + kb, nKB - 1), lane);
        }
    };
    auto step = [&](Pre& p, int st) {
        const bool valid = rb >= 0 && st < nst && st * EMB_FKS + wave * 32 + 8 * g < em.K;
        Frag f;
        f.u[0] = pack_bf2(p.p0[0], p.p0[1]); f.u[1] = pack_bf2(p.p0[2], p.p0[3]);
        f.u[2] = pack_bf2(p.p1[0], p.p1[1]); f.u[3] = pack_bf2(p.p1[2], p.p1[3]);
        if (!valid) f.u = u32x4_t{0u, 0u, 0u, 0u};
        char* cur = img + ((st - st_begin) & 1) * IMG_B;
        *reinterpret_cast<u32x4_t*>(cur + tid * 16) = f.u;    // block kb = wave, lane
        __syncthreads();                                        // (also: everyone is done with the stage before last)
#pragma unroll
        for (int kb = 0; kb < KSB; ++kb) {
            const Frag a = ld_frag_lds(cur, kb, lane);
#pragma unroll
            for (int j = 0; j < DPW; ++j)
                if (wave + NWAVES * j < DT) Pr::mma(acc[j], a, p.w[j][kb]);
        }
        load(p, st + EMB_FDEPTH);
    };
    Pre ring[EMB_FDEPTH];
#pragma unroll
    for (int d = 0; d < EMB_FDEPTH; ++d) load(ring[d], st_begin + d);
    for (int st = st_begin; st < nst; st += EMB_FDEPTH) {
#pragma unroll
        for (int d = 0; d < EMB_FDEPTH; ++d) step(ring[d], st + d);
    }
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int dt = wave + NWAVES * j;
        if (dt < DT) {
            const int d = 16 * dt + il;
            const float bv = split == 0 ? em.b[d] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = m0 + 4 * g + r;
                if (m < M) x0[m * D + d] = acc[j][r] + bv;
            }
        }
    }
}
// host side: may this embedding take the fast path?
static inline bool embed_fwd_fast_ok(const m2m_embed* e, const float* in) {
    return e->prec == PREC_BF16 && e->pw % 8 == 0 && e->W % 4 == 0 && e->K >= 8 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
}

template <int P, int D, int RB>
__global__ __launch_bounds__(NTHREADS) void embed_fwd_kernel(const m2m_embed em, const float* __restrict__ in, long M, int N,
                                                             float* __restrict__ x0, int fast, const m2m_step_head head) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x == 0) {                              // the step prologue rides here (m2m_embed_forward_head): see m2m_step_head
        const int t = threadIdx.x;
        if (t == 0 && head.adam_state) head.adam_state[0] += 1.0f;
        if (t == 1 && head.drop_counter) *head.drop_counter += 1u;
        if (head.losses && t < head.nlosses) head.losses[t] = 0.f;
    }
    if (P == PREC_BF16 && fast) embed_fwd_fast_body<D>(em, in, M, N, x0, blockIdx.x, 0, 1, smem);
    else embed_fwd_body<P, D, RB>(em, in, M, N, x0, blockIdx.x, smem);
}

// Both patch embeddings of a two-tower model in ONE launch (no fork / join of a second stream at the head of the step):
// workgroups [0, nwg0) serve embedding 0 -- the one with the larger patch, dispatched first -- the rest embedding 1.
struct EmbedFwdGroupArgs {
    m2m_embed em[2];
    const float* in[2];
    float* x0[2];
    long M[2];
    int N[2], nwg0, fast[2], nsplit[2];
    long part_stride[2];       // floats between the k-split partial outputs
    // head of a training step folded into this launch (the first one of the step; nothing in it reads these):
    // adam_state[0] += 1, *drop_counter += 1, losses[0..nlosses) = 0 -- what m2m_step_prologue does in a launch of its own
    float* adam_state;
    unsigned int* drop_counter;
    float* losses;
    int nlosses, prologue;
};
template <int P, int D, int RB>
__global__ __launch_bounds__(NTHREADS) void embed_fwd_group_kernel(const EmbedFwdGroupArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.prologue && blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t == 0 && a.adam_state) a.adam_state[0] += 1.0f;
        if (t == 1 && a.drop_counter) *a.drop_counter += 1u;
        if (a.losses && t < a.nlosses) a.losses[t] = 0.f;
    }
    const int e = (int)blockIdx.x < a.nwg0 ? 0 : 1;
    const int wg = e ? blockIdx.x - a.nwg0 : blockIdx.x;
    if (P == PREC_BF16 && a.fast[e]) {
        const int ns = a.nsplit[e], split = wg % ns;
        embed_fwd_fast_body<D>(a.em[e], a.in[e], a.M[e], a.N[e], a.x0[e] + split * a.part_stride[e], wg / ns, split, ns, smem);
    }
    else {
        // generic path asked for partial sums (an input the fast path cannot take, e.g. not 16-byte aligned): split 0 computes
        // the whole sum into part 0, the other splits' workgroups clear their rows of their part
        const int ns = a.nsplit[e], split = wg % ns, tile = wg / ns;
        if (split == 0) embed_fwd_body<P, D, RB>(a.em[e], a.in[e], a.M[e], a.N[e], a.x0[e], tile, smem);
        else {
            float* part = a.x0[e] + split * a.part_stride[e];
            const long m0 = (long)tile * RB;
            for (int idx = threadIdx.x; idx < RB * D; idx += NTHREADS) {
                const long m = m0 + idx / D;
                if (m < a.M[e]) part[m * D + idx % D] = 0.f;
            }
        }
    }
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
static int launch_embed_fwd(const m2m_embed* e, const float* in, int B, float* x0, hipStream_t st, const m2m_step_head* head = nullptr) {
    m2m_step_head hd;
    memset(&hd, 0, sizeof(hd));
    if (head) {
        if (head->nlosses < 0 || head->nlosses > 64) { m2m_set_error("embed_forward: step head nlosses must be in [0, 64]", __FILE__, __LINE__); return -1; }
        hd = *head;
    }
    const int N = (e->H / e->ph) * (e->W / e->pw);
    const long M = (long)B * N;
    // 16 rows per workgroup: the audio embedding (2048 rows at batch 512, 50 KB of input per sample) sits at the head of
    // the step's critical path, and 64 workgroups of 32 rows left three quarters of the chip idle
    constexpr int RB = 16;
    const size_t lds = (size_t)RB * EMB_LD * 4 + (size_t)RB * EMB_KS * Prec<P>::ESZ + EMB_KMAX * 4 + RB * 8;
    auto kern = embed_fwd_kernel<P, D, RB>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)((M + RB - 1) / RB)), dim3(NTHREADS), lds, st, *e, in, M, N, x0, (int)embed_fwd_fast_ok(e, in), hd);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int D>
static int launch_embed_fwd_group(const m2m_embed* const* es, const float* const* ins, float* const* x0s, const int* nsplits,
                                  const int64_t* part_strides, const m2m_step_head* head, int B, hipStream_t st) {
    constexpr int RB = 16;
    EmbedFwdGroupArgs a;
    memset(&a, 0, sizeof(a));
    if (head) {
        if (head->nlosses < 0 || head->nlosses > 64) { m2m_set_error("embeds_forward: step head nlosses must be in [0, 64]", __FILE__, __LINE__); return -1; }
        a.prologue = 1; a.adam_state = head->adam_state; a.drop_counter = head->drop_counter; a.losses = head->losses; a.nlosses = head->nlosses;
    }
    const int first = es[1]->Kp > es[0]->Kp ? 1 : 0;
    int total = 0;
    for (int k = 0; k < 2; ++k) {
        const int i = k == 0 ? first : 1 - first;
        const int N = (es[i]->H / es[i]->ph) * (es[i]->W / es[i]->pw);
        a.em[k] = *es[i]; a.in[k] = ins[i]; a.x0[k] = x0s[i]; a.N[k] = N; a.M[k] = (long)B * N;
        a.fast[k] = embed_fwd_fast_ok(es[i], ins[i]);
        a.nsplit[k] = nsplits ? nsplits[i] : 1;
        a.part_stride[k] = part_strides ? (long)part_strides[i] : 0;
        if (a.nsplit[k] < 1 || a.nsplit[k] > 4 || (a.nsplit[k] > 1 && a.part_stride[k] < a.M[k] * (long)D)) {
            m2m_set_error("embeds_forward: 1..4 k-splits, part stride >= B*N*D", __FILE__, __LINE__);
            return -1;
        }
        const int nwg = (int)((a.M[k] + RB - 1) / RB) * a.nsplit[k];
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
    return m2m_embed_forward_head(e, input, B, x0, nullptr, stream);
}
extern "C" int m2m_embed_forward_head(const m2m_embed* e, const float* input, int B, float* x0, const m2m_step_head* head, void* stream) {
    if (int rc = m2m_check_embed(e, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_EF_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_fwd<PP, DD>(e, input, B, x0, st, head);
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

// How many k-splits m2m_embeds_forward should be given for this embedding: 2 when the fast path applies and K is long
// enough for the weight stream to dominate (the audio spectrogram patches), else 1.
extern "C" int m2m_embed_fwd_splits(const m2m_embed* e) {
    if (!e) return 1;
    return (embed_fwd_fast_ok(e, nullptr) && e->Kp >= 4 * EMB_FKS) ? 2 : 1;
}

extern "C" int m2m_embeds_forward(const m2m_embed* const* embeds, const float* const* inputs, float* const* x0s, const int* nsplits,
                                  const int64_t* part_strides, int nembeds, int B, const m2m_step_head* head, void* stream) {
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
#define M2M_EFG_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_fwd_group<PP, DD>(embeds, inputs, x0s, nsplits, part_strides, head, B, st);
    M2M_EFG_CASE(PREC_BF16, 32) M2M_EFG_CASE(PREC_BF16, 64) M2M_EFG_CASE(PREC_BF16, 128) M2M_EFG_CASE(PREC_BF16, 256)
    M2M_EFG_CASE(PREC_F32, 32) M2M_EFG_CASE(PREC_F32, 64) M2M_EFG_CASE(PREC_F32, 128) M2M_EFG_CASE(PREC_F32, 256)
#undef M2M_EFG_CASE
    m2m_set_error("embeds_forward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}
