// Column-split channel-mixing launches of the split path (see split.h).
//
// Reference semantics: the channel_mix half of MixerBlock.forward (modules/mixer.py:37-40, :45) and its backward.
//
// Workgroup = SP_ROWS (128) token rows x the hidden-column units [u0, u1) of split s; 8 waves = 4 row quarters (32 rows = two
// 16-row MFMA tiles, so every weight fragment read from LDS feeds two MFMAs) x 2 column units per chunk.  Per chunk the
// weights of two 32-column units are copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: the packed blocks are 64 lanes x
// 16 B, lane-linear, exactly the DMA's image), double-buffered: the DMA of chunk c + 1 is in flight while chunk c computes
// (counted vmcnt, raw s_barrier -- a __syncthreads() would drain the DMA).  The activations (A, dYd fragments of the wave's
// 32 rows) stay in registers for the whole launch.
#include "split.h"

TIMER_DECL(g_tm_scf);
TIMER_READER(m2m_debug_timers_scf, g_tm_scf)
TIMER_DECL(g_tm_scb);
TIMER_READER(m2m_debug_timers_scb, g_tm_scb)

namespace {

typedef Prec<PREC_BF16> Pr;

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to the wave-uniform LDS byte address ldst (+ lane x
// 16).  Written as inline asm on purpose: hipcc orders every later LDS read of an object it cannot tell apart from the DMA's
// destination behind an `s_waitcnt vmcnt(0)` (it did so for the GELU table here), which drains the chunk in flight and
// serialises the loop on the DMA latency.  An asm DMA is outside its bookkeeping; completion is counted by hand (the
// `s_waitcnt vmcnt(N)` + s_barrier pairs in the chunk loops).  M0 (the DMA's LDS base) is saved and restored.
static __device__ __forceinline__ void glds16(const char* gsrc, unsigned int ldst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
}
// LDS byte address of a pointer into the dynamic shared array
static __device__ __forceinline__ unsigned int lds_addr(const void* p) {
    return (unsigned int)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}

template <int D>
struct ChainGeom {
    static constexpr int KD = D / 32, DT = D / 16;
    static constexpr int NB = 2 * KD;                       // packed 1 KiB blocks of one unit of one weight copy (== DT)
    static constexpr int UNIT = NB * 1024;                  // bytes
    static_assert(DT == NB, "block counts of the NAT and CHN copies of a unit agree");
    static constexpr int OUT_LD = D + 4;                    // fp32 row stride of the output tile in LDS
    static constexpr int OUT_BYTES = SP_ROWS * OUT_LD * 4;
};

// ---------------------------------------------------------------------------------------------------------------------------
// forward:  Yslab[s] = dropout(gelu(A W1[u0:u1]^T + b1)) W2[:, u0:u1]^T          (bias b2, output dropout, residual: next mix launch)
// ---------------------------------------------------------------------------------------------------------------------------
template <int D>
static size_t chain_fwd_lds() {
    typedef ChainGeom<D> G;
    const size_t bufs = 2 * (size_t)(2 * 2 * G::UNIT);     // 2 buffers x 2 units x (W1, W2)
    const size_t body = bufs > (size_t)G::OUT_BYTES ? bufs : (size_t)G::OUT_BYTES;
    return body + SP_MAX_UNITS_PER_SPLIT * 32 * sizeof(float) + GELU_TAB_N * sizeof(float2);
}

template <int D, int DM>
__global__ __launch_bounds__(SP_THREADS, 2) void split_chain_fwd_kernel(const SplitChainArgs a, int training, unsigned int seed,
                                                                        unsigned int step_host,
                                                                        const unsigned int* __restrict__ step_dev) {
    typedef ChainGeom<D> G;
    constexpr int KD = G::KD, DT = G::DT, NB = G::NB, UNIT = G::UNIT;
    constexpr int CHUNK = 2 * 2 * UNIT;                     // bytes of one LDS buffer: [W1 unit 0 | W1 unit 1 | W2 unit 0 | W2 unit 1]
    constexpr int BODY = 2 * CHUNK > G::OUT_BYTES ? 2 * CHUNK : G::OUT_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wbuf = smem;
    float* biasl = reinterpret_cast<float*>(smem + BODY);
    float2* gtab = reinterpret_cast<float2*>(biasl + SP_MAX_UNITS_PER_SPLIT * 32);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, il = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rq = wave & 3, cu = wave >> 2;                // row quarter, column unit inside a chunk
    // workgroup -> (split, tower, row tile).  Consecutive ids go to consecutive XCDs (id % 8): with 8 splits every XCD's L2
    // holds ONE column slice of each tower's weights.
    const int id = blockIdx.x, s = id % a.nsplit, j = id / a.nsplit, ti = j % a.ntow, rt = j / a.ntow;
    const SplitChainTower& tw = a.t[ti];
    if ((long)rt * SP_ROWS >= tw.M) return;
    TIMER_WG_BEGIN();
    const int u0 = (s * tw.nunits) / a.nsplit, u1 = ((s + 1) * tw.nunits) / a.nsplit;
    const int nch = (u1 - u0 + 1) >> 1;
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    const Drop dr = make_drop(training, tw.p_drop, seed, step, tw.site);
    TIMER_START();

    // ---- this wave's A fragments (32 rows x D), resident for the whole launch; rows >= M are zero ----
    Frag afr[2][KD];
    const int ntile16 = (tw.M + 15) >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int tile = rt * (SP_ROWS / 16) + rq * 2 + mt;
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) {
            afr[mt][kb].u = u32x4_t{0u, 0u, 0u, 0u};
            if (tile < ntile16) afr[mt][kb] = ld_frag_global(tw.a_nat, (long)tile * KD + kb, lane);
        }
    }
    // LDS-DMA of chunk c into buffer b: 4 NB blocks, NB / 2 per wave (units past u1 re-read the last unit: harmless,
    // and every wave issues the same number of DMAs, which keeps the vmcnt bookkeeping uniform)
    const unsigned int wbase = lds_addr(wbuf);
    auto stage = [&](int c, int b) {
#pragma unroll
        for (int k = 0; k < NB / 2; ++k) {
            const int jb = wave * (NB / 2) + k;             // 0 .. 4 NB - 1
            const int mat = jb / (2 * NB), un = (jb / NB) & 1, blk = jb % NB;
            int u = u0 + 2 * c + un;
            u = u < u1 ? u : u1 - 1;
            const char* src = (mat ? tw.w2c : tw.w1n) + (long)u * UNIT + blk * 1024 + lane * 16;
            glds16(src, __builtin_amdgcn_readfirstlane(wbase + b * CHUNK + jb * 1024));
        }
    };
    stage(0, 0);
    for (int i = tid; i < (u1 - u0) * 32; i += SP_THREADS) biasl[i] = tw.b1p[u0 * 32 + i];
    for (int i = tid; i < GELU_TAB_N; i += SP_THREADS) {
        const gtab_t e = pwl_cell(i, dr.scale);
        gtab[i] = make_float2(e[0], e[1]);
    }
    // the A fragments must have arrived BEFORE the chunk loop: left to itself hipcc waits for them at their first use, inside
    // the loop, with a vmcnt(0) that also drains the (to it invisible) DMA of the next chunk on every iteration
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) asm volatile("" : "+v"(afr[mt][kb].u));
    __syncthreads();                                        // table + bias visible (chunk 0's DMA is waited for in the loop)
    TIMER_MARK(g_tm_scf, 0);   // prologue

    f32x4_t yacc[2][DT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) yacc[mt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const unsigned int row_base = (unsigned int)(rt * SP_ROWS + rq * 32);
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) {
            stage(c + 1, (c + 1) & 1);
            if (NB / 2 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (NB / 2 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (NB / 2 == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                       // every wave's share of chunk c has landed (and the bias / table writes)
        TIMER_MARK(g_tm_scf, 1);   // DMA issue + wait + barrier
        const int unit = u0 + 2 * c + cu;
        if (unit < u1) {
            const char* w1 = wbuf + (c & 1) * CHUNK + cu * UNIT;
            const char* w2 = wbuf + (c & 1) * CHUNK + 2 * UNIT + cu * UNIT;
            f32x4_t hacc[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4_t bias = *reinterpret_cast<const f32x4_t*>(biasl + (unit - u0) * 32 + 16 * t + 4 * g);
                hacc[0][t] = bias;
                hacc[1][t] = bias;
            }
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
                const Frag w0 = ld_frag_lds(w1, kb, lane), w1f = ld_frag_lds(w1, KD + kb, lane);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    Pr::mma(hacc[mt][0], w0, afr[mt][kb]);
                    Pr::mma(hacc[mt][1], w1f, afr[mt][kb]);
                }
            }
            // GELU + dropout on the accumulators (row c = 32 unit + 16 t + 4 g + r, column m = il)
            Frag hf[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned int m = row_base + 16 * mt + il;
                const unsigned int word = drop_hidden_bits<DM>(dr, m, (unsigned int)unit, (unsigned int)tw.Cp) >> (4 * g);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = hacc[mt][t][r];
                        const float2 e = gtab[pwl_index(x)];
                        const float v = __builtin_fmaf(e.y, x, e.x);
                        hacc[mt][t][r] = DM == DM_NONE ? v : __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, v) & bit_to_mask(word, 16 * t + r));
                    }
                }
                Chain<PREC_BF16>::make(hacc[mt][0], hacc[mt][1], &hf[mt]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const Frag w = ld_frag_lds(w2, dt, lane);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) Pr::mma(yacc[mt][dt], hf[mt], w);
            }
        }
        TIMER_MARK(g_tm_scf, 2);   // compute
        __builtin_amdgcn_s_barrier();                       // buffer c & 1 is free for chunk c + 2
        TIMER_MARK(g_tm_scf, 3);   // trailing barrier
    }

    // ---- sum the two column-unit waves of every row quarter through LDS, then coalesced stores into slab s ----
    float* outt = reinterpret_cast<float*>(smem);            // [SP_ROWS][OUT_LD], aliases the weight buffers (all reads are behind the last barrier)
    constexpr int LD = G::OUT_LD;
    if (cu == 1) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) outt[(rq * 32 + 16 * mt + 4 * g + r) * LD + 16 * dt + il] = yacc[mt][dt][r];
    }
    __syncthreads();
    if (cu == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) outt[(rq * 32 + 16 * mt + 4 * g + r) * LD + 16 * dt + il] += yacc[mt][dt][r];
    }
    __syncthreads();
    float* slab = tw.slabs + (long)s * tw.M * D;
    const int rows = min(SP_ROWS, tw.M - rt * SP_ROWS);
    _Pragma("unroll 1") for (int idx = tid; idx < rows * (D / 4); idx += SP_THREADS) {
        const int r = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
        *reinterpret_cast<f32x4_t*>(slab + ((long)rt * SP_ROWS + r) * D + c4) = *reinterpret_cast<const f32x4_t*>(outt + r * LD + c4);
    }
    TIMER_MARK(g_tm_scf, 4);   // epilogue
    TIMER_WG_END(g_tm_scf);
}


// ---------------------------------------------------------------------------------------------------------------------------
// backward:  Hpre = A W1[u0:u1]^T + b1 (recomputed),  dHact = dYd W2[:, u0:u1],  dHpre = dHact * mask * gelu'(Hpre),
//            dAslab[s] = dHpre W1[u0:u1];  Hact^T and dHpre^T go out once, in operand precision and layout, for the
//            weight-gradient launch (tower_wgrad.hip; transposed through an identity MFMA as in tower_bwd.hip)
// ---------------------------------------------------------------------------------------------------------------------------
template <int D>
static size_t chain_bwd_lds() {
    typedef ChainGeom<D> G;
    const size_t bufs = 2 * (size_t)(2 * 3 * G::UNIT);     // 2 buffers x 2 units x (W1, W2^T, W1^T)
    const size_t body = bufs > (size_t)G::OUT_BYTES ? bufs : (size_t)G::OUT_BYTES;
    return body + SP_MAX_UNITS_PER_SPLIT * 32 * sizeof(float) + GELU_TAB_N * sizeof(gtab_t);
}

template <int D, int DM>
__global__ __launch_bounds__(SP_THREADS, 2) void split_chain_bwd_kernel(const SplitChainArgs a, unsigned int seed, unsigned int step_host,
                                                                        const unsigned int* __restrict__ step_dev) {
    typedef ChainGeom<D> G;
    constexpr int KD = G::KD, DT = G::DT, NB = G::NB, UNIT = G::UNIT;
    constexpr int CHUNK = 2 * 3 * UNIT;                     // [W1 u0 | W1 u1 | W2^T u0 | W2^T u1 | W1^T u0 | W1^T u1]
    constexpr int BODY = 2 * CHUNK > G::OUT_BYTES ? 2 * CHUNK : G::OUT_BYTES;
    constexpr int NDMA = 6 * NB / 8;                        // LDS-DMA instructions per wave and chunk
    static_assert(6 * NB % 8 == 0, "chunk blocks divide over the eight waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wbuf = smem;
    float* biasl = reinterpret_cast<float*>(smem + BODY);
    gtab_t* gtab = reinterpret_cast<gtab_t*>(biasl + SP_MAX_UNITS_PER_SPLIT * 32);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, il = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rq = wave & 3, cu = wave >> 2;
    const int id = blockIdx.x, s = id % a.nsplit, j = id / a.nsplit, ti = j % a.ntow, rt = j / a.ntow;
    const SplitChainTower& tw = a.t[ti];
    if ((long)rt * SP_ROWS >= tw.M) return;
    const int u0 = (s * tw.nunits) / a.nsplit, u1 = ((s + 1) * tw.nunits) / a.nsplit;
    const int nch = (u1 - u0 + 1) >> 1;
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    const Drop dr = make_drop(true, tw.p_drop, seed, step, tw.site);

    Frag afr[2][KD], dyfr[2][KD];
    const int ntile16 = (tw.M + 15) >> 4;
    const int tile0 = rt * (SP_ROWS / 16) + rq * 2;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) {
            afr[mt][kb].u = u32x4_t{0u, 0u, 0u, 0u};
            dyfr[mt][kb].u = u32x4_t{0u, 0u, 0u, 0u};
            if (tile0 + mt < ntile16) {
                afr[mt][kb] = ld_frag_global(tw.a_nat, (long)(tile0 + mt) * KD + kb, lane);
                dyfr[mt][kb] = ld_frag_global(tw.dy_nat, (long)(tile0 + mt) * KD + kb, lane);
            }
        }
    }
    const unsigned int wbase = lds_addr(wbuf);
    auto stage = [&](int c, int b) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int jb = wave * NDMA + k;                 // 0 .. 6 NB - 1
            const int mat = jb / (2 * NB), un = (jb / NB) & 1, blk = jb % NB;
            int u = u0 + 2 * c + un;
            u = u < u1 ? u : u1 - 1;
            const char* base = mat == 0 ? tw.w1n : (mat == 1 ? tw.w2tn : tw.w1tc);
            glds16(base + (long)u * UNIT + blk * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(wbase + b * CHUNK + jb * 1024));
        }
    };
    stage(0, 0);
    for (int i = tid; i < (u1 - u0) * 32; i += SP_THREADS) biasl[i] = tw.b1p[u0 * 32 + i];
    gelu_tab_fill(gtab, dr.scale, tid, SP_THREADS);
    // operand fragments in registers before the chunk loop (see the forward kernel)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) {
            asm volatile("" : "+v"(afr[mt][kb].u));
            asm volatile("" : "+v"(dyfr[mt][kb].u));
        }
    __syncthreads();                                        // table + bias visible (chunk 0's DMA is waited for in the loop)

    f32x4_t dacc[2][DT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dacc[mt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // identity block of the transposing MFMA: lane (g, il) is non-zero iff g == il >> 2, at element il & 3
    const unsigned int id_sel = (g == (il >> 2)) ? ((il & 1) ? 0x3F800000u : 0x00003F80u) : 0u;
    const unsigned int id_a = (il & 2) ? 0u : id_sel, id_b = (il & 2) ? id_sel : 0u;
    const unsigned int row_base = (unsigned int)(rt * SP_ROWS + rq * 32);
    const long npair = (ntile16 + 1) >> 1, pair = rt * (SP_ROWS / 32) + rq;
    const bool rows_here = tile0 < ntile16;                 // this wave's 32-row pair exists (its operand streams are allocated)

    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) {
            stage(c + 1, (c + 1) & 1);
            if (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (NDMA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const int unit = u0 + 2 * c + cu;
        if (unit < u1 && rows_here) {
            const char* w1 = wbuf + (c & 1) * CHUNK + cu * UNIT;
            const char* w2t = wbuf + (c & 1) * CHUNK + 2 * UNIT + cu * UNIT;
            const char* w1t = wbuf + (c & 1) * CHUNK + 4 * UNIT + cu * UNIT;
            f32x4_t hacc[2][2], gacc[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4_t bias = *reinterpret_cast<const f32x4_t*>(biasl + (unit - u0) * 32 + 16 * t + 4 * g);
                hacc[0][t] = bias;
                hacc[1][t] = bias;
                gacc[0][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                gacc[1][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
                const Frag wa = ld_frag_lds(w1, kb, lane), wb = ld_frag_lds(w1, KD + kb, lane);
                const Frag va = ld_frag_lds(w2t, kb, lane), vb = ld_frag_lds(w2t, KD + kb, lane);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    Pr::mma(hacc[mt][0], wa, afr[mt][kb]);
                    Pr::mma(hacc[mt][1], wb, afr[mt][kb]);
                    Pr::mma(gacc[mt][0], va, dyfr[mt][kb]);
                    Pr::mma(gacc[mt][1], vb, dyfr[mt][kb]);
                }
            }
            Frag hf[2], af[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned int m = row_base + 16 * mt + il;
                const unsigned int word = drop_hidden_bits<DM>(dr, m, (unsigned int)unit, (unsigned int)tw.Cp) >> (4 * g);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = hacc[mt][t][r];
                        const gtab_t e = gtab[pwl_index(x)];
                        const float gl = __builtin_fmaf(e[1], x, e[0]);                   // gelu(x) * scale
                        const float dg = gacc[mt][t][r] * __builtin_fmaf(e[3], x, e[2]);  // dHact * gelu'(x) * scale
                        if (DM == DM_NONE) { hacc[mt][t][r] = gl; gacc[mt][t][r] = dg; }
                        else {
                            const unsigned int mk = bit_to_mask(word, 16 * t + r);
                            hacc[mt][t][r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, gl) & mk);
                            gacc[mt][t][r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, dg) & mk);
                        }
                    }
                }
                Chain<PREC_BF16>::make(gacc[mt][0], gacc[mt][1], &hf[mt]);     // dHpre: operand of dA += dHpre W1
                Chain<PREC_BF16>::make(hacc[mt][0], hacc[mt][1], &af[mt]);     // Hact : only stored, for the weight gradients
            }
            // operand streams of the weight-gradient launch (layout and reasoning: tower_bwd.hip)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4_t od[2], oa[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    Frag idf;
                    idf.u = t == 0 ? u32x4_t{id_a, id_b, 0u, 0u} : u32x4_t{0u, 0u, id_a, id_b};
                    od[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    oa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    Pr::mma(od[t], hf[mt], idf);
                    Pr::mma(oa[t], af[mt], idf);
                }
                const long off = (long)unit * m2m_hchn_stride(npair) + (pair * 2 + mt) * 1024 + lane * 16;
                __builtin_nontemporal_store(u32x4_t{pack_bf2(od[0][0], od[0][1]), pack_bf2(od[0][2], od[0][3]),
                                                    pack_bf2(od[1][0], od[1][1]), pack_bf2(od[1][2], od[1][3])},
                                            reinterpret_cast<u32x4_t*>(tw.dh_chn + off));
                __builtin_nontemporal_store(u32x4_t{pack_bf2(oa[0][0], oa[0][1]), pack_bf2(oa[0][2], oa[0][3]),
                                                    pack_bf2(oa[1][0], oa[1][1]), pack_bf2(oa[1][2], oa[1][3])},
                                            reinterpret_cast<u32x4_t*>(tw.h_chn + off));
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const Frag w = ld_frag_lds(w1t, dt, lane);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) Pr::mma(dacc[mt][dt], hf[mt], w);
            }
        }
        __builtin_amdgcn_s_barrier();
    }

    float* outt = reinterpret_cast<float*>(smem);
    constexpr int LD = G::OUT_LD;
    if (cu == 1) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) outt[(rq * 32 + 16 * mt + 4 * g + r) * LD + 16 * dt + il] = dacc[mt][dt][r];
    }
    __syncthreads();
    if (cu == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) outt[(rq * 32 + 16 * mt + 4 * g + r) * LD + 16 * dt + il] += dacc[mt][dt][r];
    }
    __syncthreads();
    float* slab = tw.slabs + (long)s * tw.M * D;
    const int rows = min(SP_ROWS, tw.M - rt * SP_ROWS);
    _Pragma("unroll 1") for (int idx = tid; idx < rows * (D / 4); idx += SP_THREADS) {
        const int r = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
        *reinterpret_cast<f32x4_t*>(slab + ((long)rt * SP_ROWS + r) * D + c4) = *reinterpret_cast<const f32x4_t*>(outt + r * LD + c4);
    }
}

}  // namespace

template <int D, int DM>
static int launch_chain_fwd_dm(const SplitChainArgs& a, int training, unsigned int seed, unsigned int step,
                               const unsigned int* step_dev, hipStream_t st) {
    const size_t lds = chain_fwd_lds<D>();
    auto kern = split_chain_fwd_kernel<D, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.nsplit * a.ntow * a.max_rt), dim3(SP_THREADS), lds, st, a, training, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

int m2m_split_chain_forward(const SplitChainArgs& a, int D, int training, float p_drop, unsigned int seed, unsigned int step,
                            const unsigned int* step_dev, hipStream_t st) {
    if (D != 128) { m2m_set_error("split path: hidden_dim 128 only in this build", __FILE__, __LINE__); return -1; }
    for (int i = 0; i < a.ntow; ++i) {
        const int per = (a.t[i].nunits + a.nsplit - 1) / a.nsplit;
        if (per > SP_MAX_UNITS_PER_SPLIT || a.t[i].nunits < a.nsplit) { m2m_set_error("split path: channel_dim out of range for the column split", __FILE__, __LINE__); return -1; }
    }
    switch (m2m_drop_mode(training, p_drop)) {
        case DM_NONE: return launch_chain_fwd_dm<128, DM_NONE>(a, training, seed, step, step_dev, st);
        case DM_HALF: return launch_chain_fwd_dm<128, DM_HALF>(a, training, seed, step, step_dev, st);
        default:      return launch_chain_fwd_dm<128, DM_GEN>(a, training, seed, step, step_dev, st);
    }
}

template <int D, int DM>
static int launch_chain_bwd_dm(const SplitChainArgs& a, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const size_t lds = chain_bwd_lds<D>();
    auto kern = split_chain_bwd_kernel<D, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.nsplit * a.ntow * a.max_rt), dim3(SP_THREADS), lds, st, a, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

int m2m_split_chain_backward(const SplitChainArgs& a, int D, float p_drop, unsigned int seed, unsigned int step,
                             const unsigned int* step_dev, hipStream_t st) {
    if (D != 128) { m2m_set_error("split path: hidden_dim 128 only in this build", __FILE__, __LINE__); return -1; }
    switch (m2m_drop_mode(1, p_drop)) {
        case DM_NONE: return launch_chain_bwd_dm<128, DM_NONE>(a, seed, step, step_dev, st);
        case DM_HALF: return launch_chain_bwd_dm<128, DM_HALF>(a, seed, step, step_dev, st);
        default:      return launch_chain_bwd_dm<128, DM_GEN>(a, seed, step, step_dev, st);
    }
}
