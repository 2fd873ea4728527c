// Patch-embedding forward bodies (modules/mixer.py:143-146): one workgroup of NTHREADS threads computes 16 token rows x all D
// channels.  Shared by the embedding launches (embed.hip) and by the tower forward launch that carries the embedding of its own
// rows as a prologue (tower_fwd.hip: m2m_towers_forward_embeds).
#pragma once
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
#ifndef EMB_FDEPTH
#define EMB_FDEPTH 2        // (3 stages in flight need 146 VGPRs: one workgroup per CU; see embed_fwd_group_kernel)
#endif
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

