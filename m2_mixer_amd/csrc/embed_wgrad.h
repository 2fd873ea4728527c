// Patch-embedding weight gradient: the workgroup body, shared by embed.hip (its own launches, 512 threads) and
// tower_wgrad.hip (as extra workgroups of the merged weight-gradient launch, 256 threads).
#pragma once
#include "tile.h"
#include <stdlib.h>

#define EBM 32             // token rows per tile

int m2m_check_embed(const m2m_embed* e, int B);    // embed.hip

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

// g_w[d][k] += sum_m dx0[m][d] patch[m][k];  workgroup = 64 k columns (chunk) x one group of 32-row tiles.
// Software-pipelined: the next tile's dx0 rows and patch elements are loaded into registers (12 per thread) while the
// current tile is packed and multiplied, so the loop runs at the LDS/MFMA rate instead of one exposed memory latency per
// tile (2.8 -> ~1 us per tile).
template <int P, int D, int NT>
static __device__ __forceinline__ void embed_wgrad_body(const m2m_embed& em, const float* __restrict__ in,
                                                        const float* __restrict__ dx0, long M, int N, int tiles_per_group,
                                                        int chunk, int group, bool single, char* smem) {
    typedef Prec<P> Pr;
    constexpr int DT = D / 16, NKM = EBM / Pr::KB, XLD = TileGeom<D>::XLD;
    constexpr int KC = 64, KCT = KC / 16, PLD = KC + 4;
    constexpr int DPW = (DT + (NT / 64) - 1) / (NT / 64);
    constexpr int NDX = (EBM * (D / 4) + NT - 1) / NT;    // float4 pieces of the dx0 tile per thread
    constexpr int NPT = EBM * KC / NT;                          // patch elements per thread (rows wave + 8 i, column tid % 64)
    static_assert(NT % KC == 0 && EBM % (NT / KC) == 0, "patch tile mapping");
    float* dxt = reinterpret_cast<float*>(smem);                  // [EBM][XLD]   dx0 tile
    float* pt = dxt + EBM * XLD;                                    // [EBM][PLD]   patch tile
    char* aimg = reinterpret_cast<char*>(pt + EBM * PLD);           // NAT X[i=d][k=m]  blocks [dt][kbm]
    char* bimg = aimg + EBM * D * Pr::ESZ;                          // NAT X[i=kk][k=m] blocks [kt][kbm]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    PatchGeom pg{em.Cin, em.H, em.W, em.ph, em.pw, em.W / em.pw, N, em.K};
    const int k0 = chunk * KC;
    const int ko = patch_koff(pg, k0 + (tid % KC));                 // this thread's patch column, fixed for the whole loop

    f32x4_t acc[DPW][KCT];
#pragma unroll
    for (int j = 0; j < DPW; ++j)
#pragma unroll
        for (int kt = 0; kt < KCT; ++kt) acc[j][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                               // bias gradient (k-chunk 0 only), thread d

    const long ntiles = (M + EBM - 1) / EBM;
    const long t_begin = (long)group * tiles_per_group;
    const long t_end = min(ntiles, t_begin + tiles_per_group);
    float4 dxr[NDX];
    float ptr_[NPT];
    auto load_tile = [&](long tl) {
        const long m0 = tl * EBM;
#pragma unroll
        for (int i = 0; i < NDX; ++i) {
            const int idx = i * NT + tid;
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            dxr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < EBM * (D / 4) && m0 + r < M) dxr[i] = *reinterpret_cast<const float4*>(dx0 + (m0 + r) * D + c);
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const long rb = patch_rowbase(pg, m0 + tid / KC + (NT / KC) * i, M);
            ptr_[i] = (rb >= 0 && ko >= 0) ? in[rb + ko] : 0.f;
        }
    };
    if (t_begin < t_end) load_tile(t_begin);
    for (long tl = t_begin; tl < t_end; ++tl) {
        __syncthreads();                                            // the previous tile's MFMAs are done with the LDS images
#pragma unroll
        for (int i = 0; i < NDX; ++i) {
            const int idx = i * NT + tid;
            if (idx < EBM * (D / 4)) *reinterpret_cast<float4*>(dxt + (idx / (D / 4)) * XLD + (idx % (D / 4)) * 4) = dxr[i];
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) pt[(tid / KC + (NT / KC) * i) * PLD + tid % KC] = ptr_[i];
        if (tl + 1 < t_end) load_tile(tl + 1);                      // in flight behind the packing and the MFMAs below
        __syncthreads();
        if (chunk == 0 && tid < D) {
            float s = 0.f;
            for (int r = 0; r < EBM; ++r) s += dxt[r * XLD + tid];
            bsum += s;
        }
        for (int slot = tid; slot < DT * NKM * 64; slot += NT) {
            const int blk = slot >> 6;
            *reinterpret_cast<u32x4_t*>(aimg + slot * 16) =
                gather_slot<P>(dxt, XLD, PACK_NAT, true, blk / NKM, blk % NKM, slot & 63);
        }
        for (int slot = tid; slot < KCT * NKM * 64; slot += NT) {
            const int blk = slot >> 6;
            *reinterpret_cast<u32x4_t*>(bimg + slot * 16) =
                gather_slot<P>(pt, PLD, PACK_NAT, true, blk / NKM, blk % NKM, slot & 63);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int dt = wave + (NT / 64) * j;
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
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int dt = wave + (NT / 64) * j;
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
    if (chunk == 0 && tid < D) { if (single) em.g_b[tid] += bsum; else atomicAdd(em.g_b + tid, bsum); }
}

// ---- fast form (bf16) ---------------------------------------------------------------------------------------------------
// g_w[d][k] += sum_m dx0[m][d] patch[m][k] with a SINGLE OWNER per output: a workgroup owns EFK = 16 pixel columns k over ALL
// token rows; its waves split the 32-row pairs among themselves, each accumulating a full [D x 16] partial in MFMA
// accumulators, ONE reduction through LDS at the end, plain "+=".  No atomics (the row-group form above adds 9.6 MB of partial
// sums with float atomics on M2-Mixer-B: 28-30 us as a launch of its own against < 10 us for 26 MB of input and 1.7 GFLOP),
// no LDS staging: the first operand comes as the packed image of d_x0^T the tower backward leaves (m2m_tower.dx0_chn,
// [32-row pair][d tile][lane] 16 B, chained k order), the second one is gathered straight from the input -- lane (g, il)
// needs pixel k0 + il of the 8 token rows its chained k positions name: 8 four-byte loads, 16 consecutive pixels per lane
// group and row.  Reference: the weight gradient of MLPMixer.to_patch_embedding (modules/mixer.py:143-146) under autograd.
#define EFK 32
template <int D, int NT> static constexpr size_t embed_wgrad_fast_red_bytes() { return (size_t)(NT / 64) * (EFK / 16) * (D / 16) * 64 * 16; }
bool m2m_split_eligible(const m2m_tower* t, int B, int training);    // split_api.hip (its backward does not write the image)
template <int D, int NT>
static __device__ __forceinline__ void embed_wgrad_fast_body(const m2m_embed& em, const float* __restrict__ in,
                                                             const char* __restrict__ dx0_chn, long M, int N, int npairs, int rpt,
                                                             int chunk, bool vec2, char* smem) {
    typedef Prec<PREC_BF16> Pr;
    constexpr int DT = D / 16, NW = NT / 64, NB = EFK / 16;
    static_assert(NB == 2, "two 16-column fragments per workgroup (the 8-byte gather pairs them)");
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, il = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PatchGeom pg{em.Cin, em.H, em.W, em.ph, em.pw, em.W / em.pw, N, em.K};
    const int k0 = chunk * EFK;
    int ko[NB];                                                  // this lane's pixels (one per 16-column fragment); -1 beyond K
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) ko[jb] = patch_koff(pg, vec2 ? k0 + 2 * il + jb : k0 + 16 * jb + il);
    // vec2 (workgroup-uniform; every patch row, token origin and sample stride even, input 8-byte aligned): lane il owns the
    // pixel PAIR (2 il, 2 il + 1) -- column il of fragment jb is pixel k0 + 2 il + jb -- and fetches it with ONE 8-byte load
    // origin of token n's patch inside a sample, by table (LDS): the per-row address is sample * stride + tokoff[n] + ko
    int* tokoff = reinterpret_cast<int*>(smem + embed_wgrad_fast_red_bytes<D, NT>());
    for (int n = tid; n < N; n += NT) tokoff[n] = (n / pg.GW) * pg.ph * pg.W + (n % pg.GW) * pg.pw;
    __syncthreads();
    const unsigned int sstride = (unsigned int)(em.Cin * em.H * em.W), uN = (unsigned int)N, uM = (unsigned int)M;
    const unsigned int magic = 0xFFFFFFFFu / uN;                 // floor: the quotient below is exact or one too small

    f32x4_t acc[NB][DT];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[jb][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    struct Tile { Frag a[DT]; float v[NB][8]; unsigned int okbits; };
    auto load = [&](Tile& t, int p) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) t.a[dt] = ld_frag_global(dx0_chn, (long)p * DT + dt, lane);
        // element e of the lane's second-operand fragment: token slot 16 (e >> 2) + 4 g + (e & 3) of the pair (chained k order).
        // ALL addresses first (branch-free: invalid rows read row 0 and are zeroed afterwards), then ALL loads: written as
        // `ok ? in[..] : 0` hipcc put every load into its own basic block -- division, LDS lookup, two loads, vmcnt(0), eight
        // times in series per pair (50 us for the launch).  The empty asm keeps the loads from being sunk back under the test.
        unsigned int rb[8];
        bool okr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 4 * g + (e & 3);
            const unsigned int row = (unsigned int)((2 * p + (e >> 2)) * rpt + r);
            okr[e] = r < rpt && row < uM;
            const unsigned int rowc = okr[e] ? row : 0u;
            unsigned int q = __umulhi(rowc, magic);              // rowc / N by multiplication + one fix-up
            unsigned int n = rowc - q * uN;
            if (n >= uN) { n -= uN; q += 1u; }
            rb[e] = q * sstride + (unsigned int)tokoff[n];
        }
        if (vec2) {
            const unsigned int k2 = (unsigned int)max(ko[0], 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float2 x = *reinterpret_cast<const float2*>(in + rb[e] + k2);
                t.v[0][e] = x.x; t.v[1][e] = x.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) t.v[jb][e] = in[rb[e] + (unsigned int)max(ko[jb], 0)];
        }
        t.okbits = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) t.okbits |= (okr[e] ? 1u : 0u) << e;
    };
    auto mac = [&](const Tile& t) {
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            Frag b;                                              // (invalid rows / pixels beyond K: zeroed HERE, at the use)
            float z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = ((t.okbits >> e) & 1u) && ko[jb] >= 0 ? t.v[jb][e] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) b.u[e] = pack_bf2(z[2 * e], z[2 * e + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) Pr::mma(acc[jb][dt], t.a[dt], b);
        }
    };
    // two pairs in flight per wave (register double buffer, unrolled by two so that no slot is copied; three spilled ~20 registers)
    Tile t0, t1;
    const int p0 = wave;
    if (p0 < npairs) load(t0, p0);
    for (int p = p0; p < npairs; p += 2 * NW) {
        if (p + NW < npairs) load(t1, p + NW);
        mac(t0);
        if (p + NW < npairs) {
            if (p + 2 * NW < npairs) load(t0, p + 2 * NW);
            mac(t1);
        }
    }
    // ---- reduction over the waves: [wave][jb][dt][lane] 16 B, then wave w finishes (jb, dt) tiles w, w + NW, ... ----
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) red[((wave * NB + jb) * DT + dt) * 64 + lane] = acc[jb][dt];
    __syncthreads();
    for (int q = wave; q < NB * DT; q += NW) {
        const int jb = q / DT, dt = q % DT;
        f32x4_t s = red[q * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) s = s + red[(w * NB * DT + q) * 64 + lane];
        const int k = vec2 ? k0 + 2 * il + jb : k0 + 16 * jb + il;
        if (k < em.K) {
            const bool store = (em.wgrad_flags & M2M_WGRAD_OVERWRITE) != 0;      // "=": no read of the old values
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* qd = em.g_w + (long)(16 * dt + 4 * g + r) * em.K + k;      // element (d = 16 dt + 4 g + r, k)
                *qd = store ? s[r] : *qd + s[r];
            }
        }
    }
    if (chunk == 0) {
        // (workgroup-uniform) the bias gradient = row sums of d_x0^T: a second walk over the image with an all-ones second
        // operand -- one workgroup's extra ~3 us beside ~100 others.  (The image is bf16: the sums carry its rounding, like the
        // weight gradient; the row-group form summed the fp32 d_x0.)
        Frag ones;
        ones.u = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
        f32x4_t accb[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) accb[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int p = wave; p < npairs; p += NW) {
            Frag a[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) a[dt] = ld_frag_global(dx0_chn, (long)p * DT + dt, lane);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) Pr::mma(accb[dt], a[dt], ones);
        }
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) red[(wave * DT + dt) * 64 + lane] = accb[dt];
        __syncthreads();
        for (int dt = wave; dt < DT; dt += NW) {
            f32x4_t sb = red[dt * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) sb = sb + red[(w * DT + dt) * 64 + lane];
            if (il == 0) {                                        // every column of accb holds the row sums: column 0 writes
#pragma unroll
                for (int r = 0; r < 4; ++r) em.g_b[16 * dt + 4 * g + r] += sb[r];
            }
        }
    }
}
template <int D, int NT> static constexpr size_t embed_wgrad_fast_lds() { return embed_wgrad_fast_red_bytes<D, NT>() + 128 * sizeof(int); }

// Both patch embeddings of a two-tower model behind one launch: workgroups [0, nwg(0)) serve embedding 0, the rest 1.
#define EMB_GROUP 2
struct EmbedWgradGroupArgs {
    m2m_embed em[EMB_GROUP];
    const float* in[EMB_GROUP];
    const float* dx0[EMB_GROUP];
    long M[EMB_GROUP];
    int N[EMB_GROUP], tpg[EMB_GROUP], nchunks[EMB_GROUP], groups[EMB_GROUP];
    // fast form (fast != 0): nchunks = ceil(K / EFK), groups = 1
    const char* dx0_chn[EMB_GROUP];
    int npairs[EMB_GROUP], rpt[EMB_GROUP], vec2[EMB_GROUP], fast;
};
template <int P, int D, int NT>
static __device__ __forceinline__ void embed_wgrad_group_body(const EmbedWgradGroupArgs& a, int id, char* smem) {
    const int n0 = a.nchunks[0] * a.groups[0];
    const int e = id < n0 ? 0 : 1;
    if (e) id -= n0;
    if constexpr (P == PREC_BF16) {
        if (a.fast) {
            embed_wgrad_fast_body<D, NT>(a.em[e], a.in[e], a.dx0_chn[e], a.M[e], a.N[e], a.npairs[e], a.rpt[e], id, a.vec2[e] != 0, smem);
            return;
        }
    }
    if constexpr (NT % 64 == 0 && EBM % (NT / 64) == 0)           // (the row-group form's tile mapping: 256 or 512 threads)
        embed_wgrad_body<P, D, NT>(a.em[e], a.in[e], a.dx0[e], a.M[e], a.N[e], a.tpg[e], id % a.nchunks[e], id / a.nchunks[e],
                                   a.groups[e] == 1, smem);
}

struct EmbedWgradPlan { long M; int N, nchunks, groups, tpg; };
static EmbedWgradPlan embed_wgrad_plan(const m2m_embed* e, int B, int target_wgs) {
    EmbedWgradPlan pl;
    pl.N = (e->H / e->ph) * (e->W / e->pw);
    pl.M = (long)B * pl.N;
    pl.nchunks = (e->K + 63) / 64;
    const long ntiles = (pl.M + EBM - 1) / EBM;
    long groups = (target_wgs + pl.nchunks - 1) / pl.nchunks;  // row groups add with atomics
    if (groups > ntiles / 4) groups = ntiles / 4;
    if (groups < 1) groups = 1;
    pl.tpg = (int)((ntiles + groups - 1) / groups);
    pl.groups = (int)((ntiles + pl.tpg - 1) / pl.tpg);
    return pl;
}
template <int D, int P> static constexpr size_t embed_wgrad_lds() {   // independent of the thread count
    return (size_t)EBM * TileGeom<D>::XLD * 4 + (size_t)EBM * 68 * 4 + (size_t)EBM * D * Prec<P>::ESZ + (size_t)EBM * 64 * Prec<P>::ESZ;
}


// The fast form's arguments: embedding i takes its d_x0^T image from tower tw[i] (geometry of the image = the tower's chain
// tiles at batch B).  Returns the number of workgroups, 0 if the fast form does not apply (missing image, fp32, wide tower).
static inline int embed_wgrad_group_args_fast(EmbedWgradGroupArgs& a, const m2m_embed* const* es, const float* const* ins,
                                              const m2m_tower* const* tw, int B) {
    memset(&a, 0, sizeof(a));
    int total = 0;
    for (int i = 0; i < EMB_GROUP; ++i) {
        const m2m_embed* e = es[i];
        const m2m_tower* t = tw ? tw[i] : nullptr;
        const int N = (e->H / e->ph) * (e->W / e->pw);
        if (!t || !t->dx0_chn || e->prec != PREC_BF16 || t->prec != PREC_BF16 || m2m_is_wide(t) || t->N != N || t->D != e->D) return 0;
        if (m2m_split_eligible(t, B, 1)) return 0;               // (the split path's backward does not write the image)
        if (N > 128 || e->D > 128 || (long)B * e->Cin * e->H * e->W >= (1L << 31)) return 0;   // (hidden_dim 256: the ring spills)
        const int SPW = BM / t->N, nchain = (B + SPW - 1) / SPW;
        a.em[i] = *e; a.in[i] = ins[i]; a.dx0_chn[i] = (const char*)t->dx0_chn;
        a.M[i] = (long)B * N; a.N[i] = N; a.npairs[i] = (nchain * BM + WPAIR - 1) / WPAIR; a.rpt[i] = SPW * t->N;
        a.nchunks[i] = (e->K + EFK - 1) / EFK; a.groups[i] = 1; a.tpg[i] = 0;
        // 8-byte gathers: every address the kernel forms is even (patch rows, token origins, channel / sample strides, K)
        static const int vec2_on = [] { const char* v = getenv("M2M_EMBED_VEC2"); return v ? atoi(v) : 1; }();
        a.vec2[i] = (vec2_on && e->pw % 2 == 0 && e->W % 2 == 0 && ((long)e->H * e->W) % 2 == 0 && e->K % 2 == 0 && ((uintptr_t)ins[i] & 7) == 0) ? 1 : 0;
        total += a.nchunks[i];
    }
    a.fast = 1;
    return total;
}

// Fills the group arguments (embedding with more row tiles per workgroup first); returns the number of workgroups.
static inline int embed_wgrad_group_args(EmbedWgradGroupArgs& a, const m2m_embed* const* es, const float* const* ins,
                                         const float* const* dx0s, int B, int target_wgs, int n = EMB_GROUP) {
    memset(&a, 0, sizeof(a));
    if (n == 1) {                                     // one embedding (MIMIC-H's input projection): slot 1 stays empty (0 workgroups)
        const EmbedWgradPlan p0 = embed_wgrad_plan(es[0], B, target_wgs);
        a.em[0] = *es[0]; a.in[0] = ins[0]; a.dx0[0] = dx0s[0];
        a.M[0] = p0.M; a.N[0] = p0.N; a.tpg[0] = p0.tpg; a.nchunks[0] = p0.nchunks; a.groups[0] = p0.groups;
        a.em[1] = *es[0];
        return p0.nchunks * p0.groups;
    }
    EmbedWgradPlan pl[EMB_GROUP];
    for (int i = 0; i < EMB_GROUP; ++i) pl[i] = embed_wgrad_plan(es[i], B, target_wgs);
    const int first = pl[1].tpg > pl[0].tpg ? 1 : 0;
    int total = 0;
    for (int k = 0; k < EMB_GROUP; ++k) {
        const int i = k == 0 ? first : 1 - first;
        a.em[k] = *es[i]; a.in[k] = ins[i]; a.dx0[k] = dx0s[i];
        a.M[k] = pl[i].M; a.N[k] = pl[i].N; a.tpg[k] = pl[i].tpg; a.nchunks[k] = pl[i].nchunks; a.groups[k] = pl[i].groups;
        total += pl[i].nchunks * pl[i].groups;
    }
    return total;
}
