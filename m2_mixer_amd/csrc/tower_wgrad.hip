// Channel-mixing weight gradients of every block of a tower: g_ch_w1, g_ch_b1, g_ch_w2.
//
//   dW1[c][d] = sum_m dHpre[m][c] A[m][d]      dW2[d][c] = sum_m dYd[m][d] Hact[m][c]      db1[c] = sum_m dHpre[m][c]
//
// The contraction runs over ALL token rows, so the roles flip relative to the forward/backward chain:
// a workgroup owns 128 hidden columns of one block (each of its 8 waves 16; the wave's W1 / W2^T
// fragments and its 16x128 slices of dW1 and dW2^T stay in registers: 2 waves per SIMD need <= 256 VGPRs) and streams the 32-row operand
// tiles that tower_bwd.hip wrote (A, A^T, dYd, dYd^T, already in packed MFMA order) through a
// double-buffered LDS stage (global loads of tile t+1 are issued before tile t is computed and written
// to LDS after it).  Per tile it recomputes Hpre = A W1^T + b1 and dHact = dYd W2 (hidden activations
// are never stored), applies GELU / GELU' / dropout on the accumulators and chains them (k = row
// index) into the two products.  With one row group (the default when the launch already has enough
// workgroups) every result element has a single owner and is written without atomics.
#include "tile.h"
#include <stdlib.h>

#define WBM 32            // rows per streamed tile (= two 16-row tiles of the chain kernels when BM == 16)
#define WMT (WBM / 16)

TIMER_DECL(g_tm_wg);
TIMER_READER(m2m_debug_timers_wgrad, g_tm_wg)

template <int P, int D>
static constexpr int wgrad_stages() {
    return (2 * 4 * WBM * D * Prec<P>::ESZ + (Act<P>::USES_TABLE ? GELU_TAB_N * 16 : 0)) <= 160 * 1024 ? 2 : 1;
}

template <int P, int D, int DM>
__global__ __launch_bounds__(NTHREADS) void tower_wgrad_kernel(const m2m_tower tw, int ntiles, int tiles_per_group,
                                                               int rows_per_tile, unsigned int seed, unsigned int step_host,
                                                               const unsigned int* __restrict__ step_dev) {
    typedef Prec<P> Pr;
    constexpr int DT = D / 16, KD = D / Pr::KB, NF = Chain<P>::NF;
    constexpr int IMG_B = WBM * D * Pr::ESZ;
    constexpr int STAGE_B = 4 * IMG_B;                      // A | dYd | A^T | dYd^T of one tile
    constexpr int NLD = STAGE_B / (NTHREADS * 16);          // 16-byte pieces per thread per tile
    constexpr int NST = wgrad_stages<P, D>();               // 2: double-buffered LDS stage; 1 when two would not fit (fp32, D = 256)
    static_assert(STAGE_B % (NTHREADS * 16) == 0, "tile stage must split evenly over the threads");
    static_assert(IMG_B % (NTHREADS * 16) == 0 || (NTHREADS * 16) % IMG_B == 0, "piece never straddles two images");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    gtab_t* gtab = reinterpret_cast<gtab_t*>(smem + NST * STAGE_B);   // [GELU_TAB_N] (bf16 mode only)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    const m2m_block& bk = tw.blk[blockIdx.y];
    const int Cp = tw.Cp, C = tw.C;
    const int npairs = Cp >> 5;
    const int ct = blockIdx.x * NWAVES + wave;              // this wave's 16-column tile
    const bool active = ct < (Cp >> 4);
    const int q = ct >> 1, tq = ct & 1;                     // pair / half of the pair (dropout word addressing)
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    const unsigned int site = tw.site_base + 4u * blockIdx.y;
    const Drop dr_ch = make_drop(true, tw.p_drop, seed, step, site + 2);

    Frag w1f[KD], w2f[KD];
    f32x4_t dw1[DT], dw2[DT];
    float db1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) {
        w1f[kb].u = u32x4_t{0u, 0u, 0u, 0u};
        w2f[kb].u = u32x4_t{0u, 0u, 0u, 0u};
        if (active) {
            w1f[kb] = ld_frag_global(bk.w1n, (long)ct * KD + kb, lane);
            w2f[kb] = ld_frag_global(bk.w2tn, (long)ct * KD + kb, lane);
        }
    }
    const float bias = active ? bk.ch_b1p[16 * ct + il] : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        dw1[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dw2[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // the four packed images of a tile, as NLD 16-byte pieces per thread
    const char* src_a = reinterpret_cast<const char*>(bk.a_nat);
    const char* src_dy = reinterpret_cast<const char*>(bk.dy_nat);
    const char* src_at = reinterpret_cast<const char*>(bk.at_chn);
    const char* src_dyt = reinterpret_cast<const char*>(bk.dyt_chn);
    u32x4_t pre[NLD];
#define STAGE_LOAD(tile_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                              \
        const int o = (i * NTHREADS + tid) * 16;                                                   \
        const int img = o / IMG_B, oo = o % IMG_B;                                                 \
        const char* sp = img == 0 ? src_a : (img == 1 ? src_dy : (img == 2 ? src_at : src_dyt));   \
        pre[i] = *reinterpret_cast<const u32x4_t*>(sp + (long)(tile_) * IMG_B + oo);               \
    }
#define STAGE_STORE(buf_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i)                                                \
        *reinterpret_cast<u32x4_t*>((buf_) + (i * NTHREADS + tid) * 16) = pre[i];

    TIMER_START();
    if (Act<P>::USES_TABLE) gelu_tab_fill(gtab, tid, NTHREADS);
    const int t_begin = blockIdx.z * tiles_per_group;
    const int t_end = min(ntiles, t_begin + tiles_per_group);
    if (t_begin < t_end) {
        STAGE_LOAD(t_begin)
        STAGE_STORE(smem)
    }
    __syncthreads();
    for (int tile = t_begin; tile < t_end; ++tile) {
        char* cur = smem + (NST == 2 ? ((tile - t_begin) & 1) * STAGE_B : 0);
        char* nxt = smem + (NST == 2 ? (((tile - t_begin) & 1) ^ 1) * STAGE_B : 0);
        const bool more = tile + 1 < t_end;
        if (more) { STAGE_LOAD(tile + 1) }                   // in flight during this tile's math
        const char* a_nat = cur;
        const char* dy_nat = cur + IMG_B;
        const char* at_chn = cur + 2 * IMG_B;
        const char* dyt_chn = cur + 3 * IMG_B;
        TIMER_MARK(g_tm_wg, 0);
        if (active) {
            // the tile's two 16-row sub-tiles chain into one k-block (bf16) / two (fp32)
            f32x4_t hact[2], dhp[2];                        // [row sub-tile]
#pragma unroll
            for (int u = 0; u < WMT; ++u) {
                f32x4_t hacc = f32x4_t{bias, bias, bias, bias};
                f32x4_t gacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KD; ++kb) {
                    const Frag a = ld_frag_lds(a_nat, u * KD + kb, lane);
                    const Frag dy = ld_frag_lds(dy_nat, u * KD + kb, lane);
                    Pr::mma(hacc, a, w1f[kb]);
                    Pr::mma(gacc, dy, w2f[kb]);
                }
                // accumulator element r: row m = 16 u + 4 g + r, column c = 16 ct + il
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // sub-tile u of this tile is chain tile (tile * SUB + u / (BM/16)); its rows are sample-aligned
                    const unsigned int m = (unsigned int)(tile * (WBM / BM) + (16 * u) / BM) * rows_per_tile + (16 * u) % BM + 4 * g + r;
                    float gl, dgl;
                    Act<P>::gelu_grad(gtab, hacc[r], gl, dgl);
                    const bool keep = (drop_hidden_bits<DM>(dr_ch, m, q, Cp) >> (16 * tq + il)) & 1u;
                    const float hv = keep ? gl * dr_ch.scale : 0.f;
                    const float dv = keep ? gacc[r] * dgl * dr_ch.scale : 0.f;
                    hact[u][r] = hv;
                    dhp[u][r] = dv;
                    db1 += dv;
                }
            }
            TIMER_MARK(g_tm_wg, 1);
            {
                Frag hf[NF], df[NF];
                Chain<P>::make(hact[0], hact[1], hf);
                Chain<P>::make(dhp[0], dhp[1], df);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const Frag at = ld_frag_lds(at_chn, f * DT + dt, lane);
                        const Frag dyt = ld_frag_lds(dyt_chn, f * DT + dt, lane);
                        Pr::mma(dw1[dt], df[f], at);
                        Pr::mma(dw2[dt], hf[f], dyt);
                    }
                }
            }
            TIMER_MARK(g_tm_wg, 2);
        }
        if (NST == 1) __syncthreads();                       // single stage: everyone is done reading before the overwrite
        if (more) { STAGE_STORE(nxt) }
        __syncthreads();
        TIMER_MARK(g_tm_wg, 3);
    }
#undef STAGE_LOAD
#undef STAGE_STORE

    if (!active) return;
    const bool single = gridDim.z == 1;
    // ---- results: dw1[dt][r] = dW1[c = 16ct + 4g + r][d = 16dt + il]; dw2 likewise = dW2[d][c] ----
    {
        const int c0 = 16 * ct + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + il;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (c0 + r < C) {
                    float* p1 = bk.g_ch_w1 + (long)(c0 + r) * D + d;
                    if (single) *p1 += dw1[dt][r]; else atomicAdd(p1, dw1[dt][r]);
                }
            }
            float* p2 = bk.g_ch_w2 + (long)d * C + c0;      // four consecutive c of row d
            if (single && c0 + 3 < C && (C & 3) == 0) {
                float4 o = *reinterpret_cast<float4*>(p2);
                o.x += dw2[dt][0]; o.y += dw2[dt][1]; o.z += dw2[dt][2]; o.w += dw2[dt][3];
                *reinterpret_cast<float4*>(p2) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + r < C) { if (single) p2[r] += dw2[dt][r]; else atomicAdd(p2 + r, dw2[dt][r]); }
            }
        }
        float s = db1;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const int c = 16 * ct + il;
        if (g == 0 && c < C) { if (single) bk.g_ch_b1[c] += s; else atomicAdd(bk.g_ch_b1 + c, s); }
    }
    TIMER_MARK(g_tm_wg, 4);        // result write-out
}

template <int P, int D, int DM>
static int launch_wgrad_dm(const m2m_tower* t, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const bool wide = m2m_is_wide(t);                           // wide path: chain tiles are any BM consecutive rows
    const int SPW = wide ? 1 : BM / t->N;                       // samples per chain tile
    const int nchain = wide ? (int)(((long)B * t->N + BM - 1) / BM) : (B + SPW - 1) / SPW;   // chain tiles (BM rows each)
    const int ntiles = (nchain * BM + WBM - 1) / WBM;           // streamed tiles (WBM rows each)
    const int nsl = ((t->Cp >> 4) + NWAVES - 1) / NWAVES;      // 128-column slices
    // Row groups trade parallelism against float-atomic traffic (every extra group re-adds the whole slice) and,
    // measured on M2-Mixer-B, against fitting the three towers' launches (100 + 100 + 50 workgroups) on the 256 CUs
    // in ONE round: one group (single owner per element, no atomics) whenever the launch has >= 32 workgroups.
    int groups = (32 + nsl * t->nblocks - 1) / (nsl * t->nblocks);
    // Cutting the launch with few, long workgroups (the fusion tower: 50 x 128 tiles) into row groups so that it
    // back-fills free CUs was measured and LOST (atomic traffic + contention: 612k -> 593k samples/s at 32-tile
    // groups); kept as an opt-in knob.
    if (nsl * t->nblocks <= 64 && ntiles > 64) {
        int tgt = 0;
        if (const char* e = getenv("M2M_WGRAD_SMALL_TILES")) tgt = atoi(e);
        if (tgt > 0) groups = (ntiles + tgt - 1) / tgt;
    }
    if (const char* e = getenv("M2M_WGRAD_GROUPS")) groups = atoi(e);
    if (groups < 1) groups = 1;
    int tpg = (ntiles + groups - 1) / groups;
    if (tpg < 4) tpg = 4;
    if (tpg > ntiles) tpg = ntiles;
    groups = (ntiles + tpg - 1) / tpg;
    const size_t lds = (size_t)wgrad_stages<P, D>() * 4 * WBM * D * Prec<P>::ESZ + GELU_TAB_N * 16;
    const int rows_per_tile = wide ? BM : SPW * t->N;
    auto kern = tower_wgrad_kernel<P, D, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(nsl, t->nblocks, groups), dim3(NTHREADS), lds, st, *t, ntiles, tpg, rows_per_tile, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int P, int D>
static int launch_wgrad(const m2m_tower* t, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    switch (m2m_drop_mode(1, t->p_drop)) {
        case DM_NONE: return launch_wgrad_dm<P, D, DM_NONE>(t, B, seed, step, step_dev, st);
        case DM_HALF: return launch_wgrad_dm<P, D, DM_HALF>(t, B, seed, step, step_dev, st);
        default:      return launch_wgrad_dm<P, D, DM_GEN>(t, B, seed, step, step_dev, st);
    }
}

int m2m_check_tower(const m2m_tower* t, int B);

extern "C" int m2m_tower_wgrad(const m2m_tower* t, int B, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (int rc = m2m_check_tower(t, B)) return rc;
    if (t->nblocks == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_WG_CASE(PP, DD) if (t->prec == PP && t->D == DD) return launch_wgrad<PP, DD>(t, B, seed, step, step_dev, st);
    M2M_WG_CASE(PREC_BF16, 32) M2M_WG_CASE(PREC_BF16, 64) M2M_WG_CASE(PREC_BF16, 128) M2M_WG_CASE(PREC_BF16, 256)
    M2M_WG_CASE(PREC_F32, 32) M2M_WG_CASE(PREC_F32, 64) M2M_WG_CASE(PREC_F32, 128) M2M_WG_CASE(PREC_F32, 256)
#undef M2M_WG_CASE
    m2m_set_error("tower_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}
