// Forward of a stack of MixerBlocks (+ final LayerNorm) -- one launch per tower.
//
// Reference semantics: MixerBlock.forward (modules/mixer.py:42-47) applied num_mixers times, then
// layer_norm (modules/mixer.py:128-131, :158-161, :182-185).
//
// Workgroup = BM = 16 token rows (whole samples), resident in LDS as fp32 for the entire tower; 8 waves:
//   token mixing   LN1 -> per (sample, channel) MLP over the N tokens on the VALU (N, T are tiny);
//                  the token weights sit zero-padded in LDS (broadcast reads, no bounds branches)
//   channel mixing LN2 -> packed operand image in LDS; each wave takes 32 hidden columns at a time:
//                  H^T[c][m] = W1 A^T (MFMA, W1 fragments straight from global in packed order, the next
//                  step's fragments prefetched under the epilogue), bias + erf-GELU + dropout on the
//                  accumulators, which then ARE the A operand of Y[m][d] += H W2^T (chained k order)
//                  -- the hidden activation never leaves registers.
//   the eight waves' partial Y are summed through LDS slabs, then bias + dropout + residual.
#include "tile.h"
#include "token_mfma.h"
#include "embed_fwd.h"
#include <algorithm>

// LDS budget of the forward chain kernel (bytes): FIXED + nblocks * PB * 4
template <int P, int D, int NMAX> struct FwdLds {
    static constexpr bool TOK = NMAX > 0;
    static constexpr int NM = TOK ? NMAX : 1, TW_LD = 2 * NM + 4, XLD = D + 4;
    static constexpr int PB = 5 * D + (TOK ? 32 * TW_LD + 8 : 0);          // floats of one block's small parameters
    static constexpr size_t FIXED = (size_t)BM * XLD * sizeof(float) * (1 + (TOK ? 1 : 0) + RowSlabs<D>::N) +
                                    (size_t)BM * D * Prec<P>::ESZ + (GELU_TAB_N + 2) * 8;
    // + keep-words of the token sites (two per column of the workgroup's SPW = BM / N samples) + hidden bias of one block
    static size_t bytes(int nblocks, int N, int Cp) {
        return FIXED + (TOK ? 2 * (size_t)(BM / N) * D * sizeof(unsigned int) : 0) + (size_t)nblocks * PB * sizeof(float) +
               (size_t)Cp * sizeof(float) + 16;
    }
};
#ifdef M2M_TIMERS
#define M2M_LDS_MAX (163840 - 1024)     // the diagnostic build keeps its timer slots in static LDS
#else
#define M2M_LDS_MAX 163840
#endif

#ifndef M2M_FWD_TICKETS
#define M2M_FWD_TICKETS 0     // measured: -0.6 % step time, at the price of run-to-run different forward sums (off)
#endif
TIMER_DECL(g_tm_fwd);
TIMER_READER(m2m_debug_timers_fwd, g_tm_fwd)

// One workgroup's share of a tower forward: token tile `wg`.  TW is m2m_tower (single-tower launch) or m2m_tower4 (the
// by-value descriptors of a multi-tower launch).
template <class TW, int P, int D, int NMAX, int DM>
static __device__ __forceinline__ void tower_fwd_body(const TW& tw, const float* __restrict__ x0, long x0_ss, int x0_parts,
                                                      long x0_pstride, int B,
                                                      float* __restrict__ out, long out_ss, float* __restrict__ pooled,
                                                      int training, unsigned int seed, unsigned int step_host,
                                                      const unsigned int* __restrict__ step_dev, int wg, char* smem) {
    typedef Prec<P> Pr;
    typedef TileGeom<D> G;
    constexpr int XLD = G::XLD, DT = G::DT, KD = D / Pr::KB, NF = Chain<P>::NF;
    constexpr bool TOK = NMAX > 0;                               // false: wide path, channel mixing only (rows independent)
    constexpr int NM = TOK ? NMAX : 1;
    constexpr int TW_LD = 2 * NM + 4;                            // token-weight row: W1 | W2^T | b1 | pad

    // LDS: residual stream | LN1 output (token path) | one row-major partial-Y slab per wave (RowSlabs) | packed A image |
    //      GELU table | keep-words of the token sites | the small parameters of EVERY block (loaded once: no phase of the
    //      block loop waits for a global load of a parameter)
    typedef FwdLds<P, D, NMAX> L;
    float* xs = reinterpret_cast<float*>(smem);                  // residual stream  [BM][XLD]
    float* ub = xs + BM * XLD;                                    // LN1 output       [BM][XLD] (token path only)
    float* slabs = ub + (TOK ? BM * XLD : 0);                     // [RowSlabs<D>::N][BM][XLD]
    char* at = reinterpret_cast<char*>(slabs + RowSlabs<D>::N * BM * XLD);   // packed A image   BM*D*ESZ bytes
    gtab2_t* gtab = reinterpret_cast<gtab2_t*>(at + BM * D * Pr::ESZ);        // [GELU_TAB_N] x {a, b} (bf16 mode only)
    unsigned int* wth = reinterpret_cast<unsigned int*>(gtab + GELU_TAB_N + 2);   // [SPW * D] keep-words of the token sites, one per column
    const int SPW_ = TOK ? BM / tw.N : 0;
    unsigned int* wto = wth + SPW_ * D;                           //   (bf16 mode with dropout only: token_mfma.h)
    float* par = reinterpret_cast<float*>(wto + SPW_ * D);        // [nblocks][L::PB]
    float* bias_s = par + tw.nblocks * L::PB;                     // [Cp] hidden bias (padded layout) of the block in flight
    unsigned int* qctr = reinterpret_cast<unsigned int*>(bias_s + tw.Cp);      // ticket counter of the column loop
    constexpr int PB = L::PB, O_LN1W = 0, O_LN1B = D, O_LN2W = 2 * D, O_LN2B = 3 * D, O_CHB2 = 4 * D, O_TOKW = 5 * D,
                  O_TOKB2 = 5 * D + 32 * TW_LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    const int N = tw.N, T = tw.T, Cp = tw.Cp;
    const int SPW = TOK ? BM / N : 0;
    const int s0 = wg * SPW;
    const int ns = TOK ? min(SPW, B - s0) : 0;
    const long row0 = TOK ? (long)s0 * N : (long)wg * BM;           // first global token row of this tile
    const int R = TOK ? ns * N : (int)min((long)BM, (long)B * N - row0);
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    const int rr = tid / TPR, rj = tid % TPR;                       // this thread's row and slot in the row-wise phases
    constexpr int EPT = D / TPR;

    TIMER_LSTART();
    constexpr int MAXB = (int)(sizeof(tw.blk) / sizeof(tw.blk[0]));       // blocks the descriptor type can hold
    // ---- prologue.  EVERY global load of the launch's start is requested before the first LDS write (input tile, hidden bias
    //      of block 0, the small parameters of every block); the GELU table is computed while they fly.  Written as four
    //      load-then-store loops this was four memory round trips at the head of each launch.  The block index of the
    //      parameter loads stays wave-uniform (a per-thread index into the by-value descriptor would turn every later
    //      descriptor read into a vector load). ----
    constexpr int XI = (BM * (D / 4) + NTHREADS - 1) / NTHREADS, BPT = 8, TI = TOK ? (32 * TW_LD + NTHREADS - 1) / NTHREADS : 1;
    float4 xv[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int idx = tid + k * NTHREADS, r = idx / (D / 4), c = (idx % (D / 4)) * 4;
        xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < BM * (D / 4) && r < R) {
            const long gr = row0 + r;
            const float* src = x0 + (gr / N) * x0_ss + (gr % N) * D + c;
            xv[k] = *reinterpret_cast<const float4*>(src);
            for (int p = 1; p < x0_parts; ++p) {             // k-split partial sums of the patch embedding (m2m_embeds_forward)
                const float4 u = *reinterpret_cast<const float4*>(src + p * x0_pstride);
                xv[k].x += u.x; xv[k].y += u.y; xv[k].z += u.z; xv[k].w += u.w;
            }
        }
    }
    float nb0[BPT];                                                 // Cp <= BPT * NTHREADS (checked by the host)
    const M2M_AS1 float* b1p0 = reinterpret_cast<const M2M_AS1 float*>(to_gptr(tw.blk[0].ch_b1p));   // scalar, read once (see tower_bwd.hip)
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        nb0[k] = 0.f;
        if (tid + k * NTHREADS < Cp) nb0[k] = b1p0[tid + k * NTHREADS];
    }
    float pv[MAXB][5], tv[TI][MAXB], b2v[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
#pragma unroll
        for (int q = 0; q < 5; ++q) pv[b][q] = 0.f;
        b2v[b] = 0.f;
        if (b < tw.nblocks) {
            const m2m_block& bk = tw.blk[b];
            if (tid < D) {
                if (TOK) { pv[b][0] = bk.ln1_w[tid]; pv[b][1] = bk.ln1_b[tid]; }
                pv[b][2] = bk.ln2_w[tid]; pv[b][3] = bk.ln2_b[tid]; pv[b][4] = bk.ch_b2[tid];
            }
            if (TOK && tid < N) b2v[b] = bk.tok_b2[tid];
        }
#pragma unroll
        for (int k = 0; k < TI; ++k) {
            // zero-padded token weights (rows t >= T and columns n >= N are zero):
            //   tokw[t][0..NMAX) = W1[t][n]   tokw[t][NMAX..2NMAX) = W2[n][t]   tokw[t][2NMAX] = b1[t]
            tv[k][b] = 0.f;
            const int idx = tid + k * NTHREADS, t = idx / TW_LD, j = idx % TW_LD;
            if (TOK && b < tw.nblocks && idx < 32 * TW_LD && t < T) {
                // one load through a per-thread choice among three SCALAR pointers (a per-thread choice of the descriptor field
                // made hipcc load the pointer itself per thread: pointer load, vmcnt(0), data load, once per block in series)
                const m2m_block& bk = tw.blk[b];
                const M2M_AS1 float* w1 = reinterpret_cast<const M2M_AS1 float*>(to_gptr(bk.tok_w1));
                const M2M_AS1 float* w2 = reinterpret_cast<const M2M_AS1 float*>(to_gptr(bk.tok_w2));
                const M2M_AS1 float* b1 = reinterpret_cast<const M2M_AS1 float*>(to_gptr(bk.tok_b1));
                const bool v1 = j < NMAX, v2 = !v1 && j < 2 * NMAX;
                const M2M_AS1 float* src = v1 ? w1 + (t * N + j) : (v2 ? w2 + ((j - NMAX) * T + t) : b1 + t);
                const bool ok = v1 ? j < N : (v2 ? j - NMAX < N : j == 2 * NMAX);
                if (ok) tv[k][b] = *src;
            }
        }
    }
    if (Act<P>::USES_TABLE) gelu_tab2_fill(gtab, make_drop(training, tw.p_drop, 0u, 0u, 0u).scale, tid, NTHREADS);
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int idx = tid + k * NTHREADS, r = idx / (D / 4), c = (idx % (D / 4)) * 4;
        if (idx < BM * (D / 4)) *reinterpret_cast<float4*>(xs + r * XLD + c) = xv[k];
    }
#pragma unroll
    for (int k = 0; k < BPT; ++k)
        if (tid + k * NTHREADS < Cp) bias_s[tid + k * NTHREADS] = nb0[k];
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
        if (b < tw.nblocks) {
            float* pb = par + b * PB;
            if (tid < D) {
                if (TOK) { pb[O_LN1W + tid] = pv[b][0]; pb[O_LN1B + tid] = pv[b][1]; }
                pb[O_LN2W + tid] = pv[b][2]; pb[O_LN2B + tid] = pv[b][3]; pb[O_CHB2 + tid] = pv[b][4];
            }
            if constexpr (TOK) {
#pragma unroll
                for (int k = 0; k < TI; ++k)
                    if (tid + k * NTHREADS < 32 * TW_LD) pb[O_TOKW + tid + k * NTHREADS] = tv[k][b];
                if (tid < 8) pb[O_TOKB2 + tid] = b2v[b];
            }
        }
    __syncthreads();

    // Block input of block b, held by its row thread: save it for the backward pass, LayerNorm-1 -> ub, and the keep-words
    // of the block's token sites.  Runs once from LDS for block 0 and afterwards inside the phase that produces the value.
    auto block_input = [&](int b, const float (&x)[EPT]) {
        const m2m_block& bk = tw.blk[b];
        const float* pb = par + b * PB;
        if (training && rr < R) st_row<D>(bk.x_in + (row0 + rr) * D, rj, x);
        float mean, rstd, y[EPT];
        reg_stats<D>(x, mean, rstd);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int c = ln_col<D>(e, rj);
            y[e] = (x[e] - mean) * rstd * pb[O_LN1W + c] + pb[O_LN1B + c];
        }
        st_row<D>(ub + rr * XLD, rj, y);
        if constexpr (P == PREC_BF16) {
            const unsigned int site = tw.site_base + 4u * b;
            token_keep_words<D, DM>(wth, wto, make_drop(training, tw.p_drop, seed, step, site + 0),
                                    make_drop(training, tw.p_drop, seed, step, site + 1), s0, ns, SPW, N, T, tid);
        }
    };
    if constexpr (TOK) {
        float x[EPT];
        ld_row<D>(xs + rr * XLD, rj, x);
        block_input(0, x);
        __syncthreads();
    }

    for (int b = 0; b < tw.nblocks; ++b) {
        const m2m_block& bk = tw.blk[b];
        const float* pb = par + b * PB;
        const unsigned int site = tw.site_base + 4u * b;
        const Drop dr_th = make_drop(training, tw.p_drop, seed, step, site + 0);
        const Drop dr_to = make_drop(training, tw.p_drop, seed, step, site + 1);
        const Drop dr_ch = make_drop(training, tw.p_drop, seed, step, site + 2);
        const Drop dr_co = make_drop(training, tw.p_drop, seed, step, site + 3);
        TIMER_LMARK(0);   // block input: save / LN1 (in the previous block's last phase from block 1 on)
        if (tid == 0) *qctr = NWAVES;      // tickets of the column loop (barriers lie between here and the loop)

        if constexpr (TOK) {
        // ---- token mixing (modules/mixer.py:30-35): bf16 mode on the matrix pipe (token_mfma.h), fp32 mode one thread
        //      per (sample, channel) column ----
        const float* tokw = pb + O_TOKW;
        const float* tokb2 = pb + O_TOKB2;
        if constexpr (P == PREC_BF16) {
            token_fwd_mfma<D, NMAX, DM>(ub, xs, tokw, tokb2, gtab, wth, wto, N, ns, dr_th.scale, dr_to.scale, wave, lane);
        } else
        _Pragma("unroll 1") for (int p = tid; p < ns * D; p += NTHREADS) {
            const int sl = p / D, d = p % D;
            const unsigned int bd = (unsigned int)(s0 + sl) * D + d;
            float un[NMAX], o[NMAX];
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                un[n] = (n < N) ? ub[(sl * N + n) * XLD + d] : 0.f;
                o[n] = tokb2[n];
            }
            // keep-bits of this column's T hidden units / N outputs (all ones when dropout is off): no branches below
            const unsigned int wth1 = drop_row_bits<DM>(dr_th, bd, T);
            const unsigned int wto1 = drop_row_bits<DM>(dr_to, bd, N);
#pragma unroll 4
            for (int t = 0; t < T; ++t) {
                const float* wr = tokw + t * TW_LD;
                float h = wr[2 * NMAX];
#pragma unroll
                for (int n = 0; n < NMAX; ++n) h = __builtin_fmaf(wr[n], un[n], h);
                h = Act<P>::gelu_scaled(gtab, h, dr_th.scale);
                h = DM == DM_NONE ? h : mask_f(h, bit_to_mask(wth1, t));
#pragma unroll
                for (int n = 0; n < NMAX; ++n) o[n] = __builtin_fmaf(wr[NMAX + n], h, o[n]);
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                if (n < N) {
                    xs[(sl * N + n) * XLD + d] += ((wto1 >> n) & 1u) ? o[n] * dr_to.scale : 0.f;
                }
            }
        }
        __syncthreads();
        TIMER_LMARK(1);   // token mixing
        }   // TOK

        // ---- save x_mid, LN2 -> packed operand image, all on the row thread's registers (wide path: the input IS the
        //      saved x_mid) ----
        {
            float x[EPT], y[EPT], mean, rstd;
            ld_row<D>(xs + rr * XLD, rj, x);
            if (training && TOK && rr < R) st_row<D>(bk.x_mid + (row0 + rr) * D, rj, x);
            reg_stats<D>(x, mean, rstd);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int c = ln_col<D>(e, rj);
                y[e] = (x[e] - mean) * rstd * pb[O_LN2W + c] + pb[O_LN2B + c];
            }
            pack_row_nat<P, D>(at, rr, rj, y);
        }
        __syncthreads();
        TIMER_LMARK(2);   // save x_mid, LN2, pack

        // ---- channel mixing (modules/mixer.py:37-40), each wave owns 32 hidden columns per step ----
        f32x4_t yacc[MT][DT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) yacc[mt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        const int npairs = Cp >> 5;
        Frag w1f[2][KD];
        if (wave < npairs) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kb = 0; kb < KD; ++kb) w1f[t][kb] = ld_frag_global(bk.w1n, (long)(2 * wave + t) * KD + kb, lane);
        }
        // bf16 training: steps by ticket, as in tower_bwd.hip (inference and the fp32 parity mode keep the static split:
        // reproducible sums)
        const bool tickets = P == PREC_BF16 && M2M_FWD_TICKETS && training;
        // stagger of the younger half (waves 4-7 share their SIMDs with waves 0-3): see tower_bwd.hip
#ifndef M2M_FWD_STAGGER
#define M2M_FWD_STAGGER 0
#endif
        if (M2M_FWD_STAGGER > 0 && P == PREC_BF16 && __builtin_amdgcn_readfirstlane(wave) >= NWAVES / 2) __builtin_amdgcn_s_sleep(M2M_FWD_STAGGER);
        for (int q = __builtin_amdgcn_readfirstlane(wave); q < npairs;) {
            unsigned int ticket = 0u;
            if (tickets && lane == 0) ticket = atomicAdd(qctr, 1u);
            // this step's W2 fragments: in flight during GEMM1 + epilogue
            Frag w2f[NF][DT];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) w2f[f][dt] = ld_frag_global(bk.w2c, (long)(q * NF + f) * DT + dt, lane);
            f32x4_t bias[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const f32x4_t*>(bias_s + 32 * q + 16 * t + 4 * g);

            f32x4_t hacc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                hacc[mt][0] = bias[0];
                hacc[mt][1] = bias[1];
            }
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const Frag a = ld_frag_lds(at, mt * KD + kb, lane);
                    Pr::mma(hacc[mt][0], w1f[0][kb], a);
                    Pr::mma(hacc[mt][1], w1f[1][kb], a);
                }
            }
            // prefetch the next step's W1 fragments under the epilogue (scheduling barrier: do not hoist the
            // loads above the MFMAs that still read the current fragments)
            __builtin_amdgcn_sched_barrier(0);
            const int qn = tickets ? (int)__builtin_amdgcn_readfirstlane(ticket) : q + NWAVES;
            if (qn < npairs) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kb = 0; kb < KD; ++kb)
                        w1f[t][kb] = ld_frag_global(bk.w1n, (long)(2 * qn + t) * KD + kb, lane);
            }
            // bias is already in; GELU + dropout on the accumulators (row c = 32q + 16t + 4g + r, column m = il)
            Frag hf[MT][NF];
#ifndef M2M_FWD_STAGED
#define M2M_FWD_STAGED 1
#endif
            if constexpr (M2M_FWD_STAGED && Act<P>::USES_TABLE && MT == 1) {
                // the eight table look-ups as one batch (indices, then all reads in flight, then the fmas): left to itself the
                // compiler issues them 1 + 3 + 1 + 3 with a full LDS wait after each group
                const unsigned int m = (unsigned int)(row0 + il);
                const unsigned int word = drop_hidden_bits<DM>(dr_ch, m, q, Cp) >> (4 * g);
                unsigned int idx[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) idx[t][r] = pwl_index(hacc[0][t][r]);
                __builtin_amdgcn_sched_barrier(0);
                gtab2_t e[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[t][r] = gtab[idx[t][r]];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = __builtin_fmaf(e[t][r][1], hacc[0][t][r], e[t][r][0]);
                        hacc[0][t][r] = DM == DM_NONE ? v : mask_f(v, bit_to_mask(word, 16 * t + r));
                    }
                Chain<P>::make(hacc[0][0], hacc[0][1], hf[0]);
            } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned int m = (unsigned int)(row0 + mt * 16 + il);
                const unsigned int word = drop_hidden_bits<DM>(dr_ch, m, q, Cp) >> (4 * g);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = Act<P>::gelu_scaled(gtab, hacc[mt][t][r], dr_ch.scale);
                        hacc[mt][t][r] = DM == DM_NONE ? v : mask_f(v, bit_to_mask(word, 16 * t + r));
                    }
                }
                Chain<P>::make(hacc[mt][0], hacc[mt][1], hf[mt]);
            }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) Pr::mma(yacc[mt][dt], hf[mt][f], w2f[f][dt]);
            q = qn;
        }
        TIMER_LMARK(3);   // hidden-column loop (wave 0)

        // ---- the waves' partial Y -> row-major slabs (nothing else lives there: no barrier in front) ----
        if (RowSlabs<D>::N == NWAVES || wave < RowSlabs<D>::N) acc_to_slab<D>(yacc, slabs + (wave % RowSlabs<D>::N) * BM * XLD, g, il);
        __syncthreads();
        if (RowSlabs<D>::N < NWAVES) {
            if (wave >= RowSlabs<D>::N) acc_add_slab<D>(yacc, slabs + (wave % RowSlabs<D>::N) * BM * XLD, g, il);
            __syncthreads();
        }
        TIMER_LMARK(4);   // slabs
        // ---- sum over the waves (deterministic order) + bias, dropout, residual -> the next block's input, which its row
        //      thread hands straight to that block's first phase ----
        {
            // the next block's hidden bias: requested first, written to LDS at the end of the phase (every wave is out of the
            // column loop that read this block's)
            constexpr int BPT = 8;                                  // Cp <= BPT * NTHREADS (checked by the host)
            float nb[BPT];
            const bool more = b + 1 < tw.nblocks;
            const M2M_AS1 float* b1pn = reinterpret_cast<const M2M_AS1 float*>(to_gptr(tw.blk[more ? b + 1 : b].ch_b1p));   // scalar, read once
#pragma unroll
            for (int k = 0; k < BPT; ++k) {
                nb[k] = 0.f;
                if (more && tid + k * NTHREADS < Cp) nb[k] = b1pn[tid + k * NTHREADS];
            }
            float v[EPT], x[EPT];
            slab_row_sum<D>(slabs, rr, rj, v);
            ld_row<D>(xs + rr * XLD, rj, x);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int c = ln_col<D>(e, rj);
                float o = v[e] + pb[O_CHB2 + c];
                o = drop_keep_elem<DM>(dr_co, (unsigned int)(row0 + rr) * D + c) ? o * dr_co.scale : 0.f;
                x[e] = rr < R ? x[e] + o : x[e];
            }
            st_row<D>(xs + rr * XLD, rj, x);
            if constexpr (TOK) {
            if (b + 1 < tw.nblocks) block_input(b + 1, x);
        }
#pragma unroll
            for (int k = 0; k < BPT; ++k)
                if (more && tid + k * NTHREADS < Cp) bias_s[tid + k * NTHREADS] = nb[k];
        }
        __syncthreads();
        TIMER_LMARK(5);   // wave sum + bias/dropout/residual (+ next block's save / LN1)
    }

    // ---- final LayerNorm (modules/mixer.py:131,161,185), output + token mean ----
    if (training && tw.x_final) {
        _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            *reinterpret_cast<float4*>(tw.x_final + (row0 + r) * D + c) = *reinterpret_cast<const float4*>(xs + r * XLD + c);
        }
    }
    const float* res = xs;
    if (tw.has_final_ln) {
        ln_to_tile<D>(xs, slabs, tw.lnf_w, tw.lnf_b, tid);     // slab 0 is free after the last block
        res = slabs;
        __syncthreads();
    }
    _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
        const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
        const long gr = row0 + r;
        *reinterpret_cast<float4*>(out + (gr / N) * out_ss + (gr % N) * D + c) =
            *reinterpret_cast<const float4*>(res + r * XLD + c);
    }
    if (TOK && pooled) {
        const float inv = 1.0f / (float)N;
        _Pragma("unroll 1") for (int p = tid; p < ns * D; p += NTHREADS) {
            const int sl = p / D, d = p % D;
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += res[(sl * N + n) * XLD + d];
            pooled[(long)(s0 + sl) * D + d] = s * inv;
        }
    }
    TIMER_LMARK(6);       // final LN, output, pooled
    TIMER_LFLUSH(g_tm_fwd);
}

template <int P, int D, int NMAX, int DM>
__global__ __launch_bounds__(NTHREADS) void tower_fwd_kernel(const m2m_tower tw, const float* __restrict__ x0,
                                                             long x0_ss, int B, float* __restrict__ out, long out_ss,
                                                             float* __restrict__ pooled, int training,
                                                             unsigned int seed, unsigned int step_host,
                                                             const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tower_fwd_body<m2m_tower, P, D, NMAX, DM>(tw, x0, x0_ss, 1, 0, B, out, out_ss, pooled, training, seed, step_host, step_dev,
                                               blockIdx.x, smem);
}

// Two towers side by side in ONE launch (blockIdx.y = tower): the image and audio towers need half the chip each, and as
// one launch on one stream they cost no cross-queue fork / join in the replayed graph.
struct FwdGroupArgs {
    m2m_tower4 tw[2];
    const float* x0[2];
    long x0_ss[2];
    int x0_parts[2];
    long x0_pstride[2];
    float* out[2];
    long out_ss[2];
    float* pooled[2];
    int ntiles[2];
    // Optional (m2m_towers_forward_embeds): the patch embedding of a tower's OWN 16 token rows as the prologue of its workgroups
    // (embed[i] != 0: x0[i] is the (B N, D) scratch the rows pass through), and the head of the training step (losses = 0,
    // Adam step count += 1) in workgroup 0.  The embedding launch of its own cost ~7 us of fixed time on the step's critical
    // path (launch, first loads, epilogue) for ~14 us of streaming.
    m2m_embed em[2];
    const float* ein[2];
    int embed[2], efast[2];
    float* head_losses;
    float* head_adam_state;
    int head_nlosses;
};
static_assert(sizeof(FwdGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");
template <int P, int D, int NMAX, int DM>
__global__ __launch_bounds__(NTHREADS) void tower_fwd_group_kernel(const FwdGroupArgs a, int B, int training, unsigned int seed,
                                                                   unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware mapping: workgroups are dealt to the 8 XCDs round-robin (id % 8), each XCD has its own 4 MB L2.  Tower 0
    // takes XCDs 0-3, tower 1 XCDs 4-7, so an L2 caches ONE tower's weights (2.25 MB per block in the backward chain; both
    // towers' 4.5 MB would not fit) and each tower's weights are fetched by four L2s instead of eight.
    const int id = blockIdx.x, xcd = id & 7, t = xcd >> 2;
    const int wg = (id >> 3) * 4 + (xcd & 3);
    if constexpr (NMAX > 0) {
        if (blockIdx.x == 0) {                             // head of the training step (nothing in this launch reads these)
            const int tt = threadIdx.x;
            if (tt == 0 && a.head_adam_state) a.head_adam_state[0] += 1.0f;
            if (a.head_losses && tt < a.head_nlosses) a.head_losses[tt] = 0.f;
        }
    }
    if (wg >= a.ntiles[t]) return;
    if constexpr (NMAX > 0) {
        if (a.embed[t]) {
            // this workgroup's rows [16 wg, 16 wg + 16) of the embedding output (whole samples: 16 % N == 0, checked by the host),
            // through the embedding bodies' own LDS use (nothing of the tower lives there yet) into the scratch the tower reads
            const int N = a.tw[t].N;
            float* x0w = const_cast<float*>(a.x0[t]);
            if (P == PREC_BF16 && a.efast[t]) embed_fwd_fast_body<D>(a.em[t], a.ein[t], (long)B * N, N, x0w, wg, 0, 1, smem);
            else embed_fwd_body<P, D, BM>(a.em[t], a.ein[t], (long)B * N, N, x0w, wg, smem);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's rows are in L2 ...
            __syncthreads();                                      // ... and so are the other waves' (same CU: read back through L2)
        }
    }
    tower_fwd_body<m2m_tower4, P, D, NMAX, DM>(a.tw[t], a.x0[t], a.x0_ss[t], a.x0_parts[t], a.x0_pstride[t], B, a.out[t], a.out_ss[t], a.pooled[t], training, seed,
                                                step_host, step_dev, wg, smem);
}

#ifdef M2M_ISA_PROBE
// ISA probe (scripts/isa_probe.sh): only the instantiations of the benchmark's two forward launches, no host code
template __global__ void tower_fwd_group_kernel<PREC_BF16, 128, 4, DM_HALF>(const FwdGroupArgs, int, int, unsigned int, unsigned int, const unsigned int*);
template __global__ void tower_fwd_kernel<PREC_BF16, 128, 8, DM_HALF>(const m2m_tower, const float*, long, int, float*, long, float*, int, unsigned int, unsigned int, const unsigned int*);
#else
template <int P, int D, int NMAX>
static size_t fwd_lds_bytes(int nblocks, int N, int Cp) { return FwdLds<P, D, NMAX>::bytes(nblocks, N, Cp); }

template <int P, int D, int NMAX, int DM>
static int launch_fwd_dm(const m2m_tower* t, const float* x0, long x0_ss, int B, float* out, long out_ss, float* pooled,
                      int training, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const int SPW = NMAX > 0 ? BM / t->N : 1;
    const int grid = NMAX > 0 ? (B + SPW - 1) / SPW : (int)(((long)B * t->N + BM - 1) / BM);
    const size_t lds = fwd_lds_bytes<P, D, NMAX>(t->nblocks, t->N, t->Cp);
    if (lds > M2M_LDS_MAX || t->Cp > 8 * NTHREADS) { m2m_set_error("tower_forward: blocks x channel_dim exceed the workgroup's LDS", __FILE__, __LINE__); return -1; }
    auto kern = tower_fwd_kernel<P, D, NMAX, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, M2M_LDS_MAX));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, *t, x0, x0_ss, B, out, out_ss, pooled, training, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int P, int D, int NMAX>
static int launch_fwd(const m2m_tower* t, const float* x0, long x0_ss, int B, float* out, long out_ss, float* pooled,
                      int training, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    switch (m2m_drop_mode(training, t->p_drop)) {
        case DM_NONE: return launch_fwd_dm<P, D, NMAX, DM_NONE>(t, x0, x0_ss, B, out, out_ss, pooled, training, seed, step, step_dev, st);
        case DM_HALF: return launch_fwd_dm<P, D, NMAX, DM_HALF>(t, x0, x0_ss, B, out, out_ss, pooled, training, seed, step, step_dev, st);
        default:      return launch_fwd_dm<P, D, NMAX, DM_GEN>(t, x0, x0_ss, B, out, out_ss, pooled, training, seed, step, step_dev, st);
    }
}

int m2m_check_tower(const m2m_tower* t, int B);
bool m2m_split_eligible(const m2m_tower* t, int B, int training);
bool m2m_split_can_group(const m2m_tower* a, const m2m_tower* b);
int m2m_split_forward(const m2m_tower* const* towers, const m2m_tower_io* io, int ntow, int B, int training, unsigned int seed,
                      unsigned int step, const unsigned int* step_dev, hipStream_t st);
int m2m_forward_wide(const m2m_tower* t, const float* x0, long x0_ss, int B, float* out, long out_ss, float* pooled,
                     int training, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st);

// Channel-mixing half of ONE block (+ final LayerNorm if the view has it) over B*N independent rows: the wide path's
// per-block launch.  `view` is a one-block copy of the tower (token parameters unused).
int m2m_chain_forward_rows(const m2m_tower* t, const float* x0, long x0_ss, int B, float* out, long out_ss, int training,
                           unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
#define M2M_FWDR_CASE(PP, DD) \
    if (t->prec == PP && t->D == DD) return launch_fwd<PP, DD, 0>(t, x0, x0_ss, B, out, out_ss, nullptr, training, seed, step, step_dev, st);
    M2M_FWDR_CASE(PREC_BF16, 32) M2M_FWDR_CASE(PREC_BF16, 64) M2M_FWDR_CASE(PREC_BF16, 128) M2M_FWDR_CASE(PREC_BF16, 256)
    M2M_FWDR_CASE(PREC_F32, 32) M2M_FWDR_CASE(PREC_F32, 64) M2M_FWDR_CASE(PREC_F32, 128) M2M_FWDR_CASE(PREC_F32, 256)
#undef M2M_FWDR_CASE
    m2m_set_error("tower_forward (wide): unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

template <int P, int D, int NMAX, int DM>
static int launch_fwd_group_dm(const FwdGroupArgs& a, int B, int training, unsigned int seed, unsigned int step,
                               const unsigned int* step_dev, hipStream_t st) {
    const size_t lds = std::max(fwd_lds_bytes<P, D, NMAX>(a.tw[0].nblocks, a.tw[0].N, a.tw[0].Cp),
                                fwd_lds_bytes<P, D, NMAX>(a.tw[1].nblocks, a.tw[1].N, a.tw[1].Cp));
    if (lds > M2M_LDS_MAX || a.tw[0].Cp > 8 * NTHREADS || a.tw[1].Cp > 8 * NTHREADS) {
        m2m_set_error("towers_forward: blocks x channel_dim exceed the workgroup's LDS", __FILE__, __LINE__);
        return -1;
    }
    auto kern = tower_fwd_group_kernel<P, D, NMAX, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, M2M_LDS_MAX));
        attr_done = true;
    }
    const int mx = a.ntiles[0] > a.ntiles[1] ? a.ntiles[0] : a.ntiles[1];
    const int grid = 8 * ((mx + 3) / 4);                     // see the XCD-aware mapping in the kernel
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, a, B, training, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int D, int NMAX>
static int launch_fwd_group(const FwdGroupArgs& a, int B, int training, unsigned int seed, unsigned int step,
                            const unsigned int* step_dev, hipStream_t st) {
    switch (m2m_drop_mode(training, a.tw[0].p_drop)) {
        case DM_NONE: return launch_fwd_group_dm<P, D, NMAX, DM_NONE>(a, B, training, seed, step, step_dev, st);
        case DM_HALF: return launch_fwd_group_dm<P, D, NMAX, DM_HALF>(a, B, training, seed, step, step_dev, st);
        default:      return launch_fwd_group_dm<P, D, NMAX, DM_GEN>(a, B, training, seed, step, step_dev, st);
    }
}

// Channel-mixing halves (wide path) of two towers' blocks in one launch: v[i] = block views (token_wide.hip: m2m_forward_wide_group).
int m2m_chain_forward_rows_group(const m2m_tower* const* v, const float* const* x0, const long* x0_ss, int B, float* const* out,
                                 const long* out_ss, int training, unsigned int seed, unsigned int step, const unsigned int* step_dev,
                                 hipStream_t st) {
    FwdGroupArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 2; ++i) {
        a.tw[i] = m2m_shrink(v[i]);
        a.x0[i] = x0[i]; a.x0_ss[i] = x0_ss[i]; a.x0_parts[i] = 1; a.x0_pstride[i] = 0;
        a.out[i] = out[i]; a.out_ss[i] = out_ss[i]; a.pooled[i] = nullptr;
        a.ntiles[i] = (int)(((long)B * v[i]->N + BM - 1) / BM);
    }
    const m2m_tower* t = v[0];
    if (t->D == 256 && t->prec == PREC_BF16) return launch_fwd_group<PREC_BF16, 256, 0>(a, B, training, seed, step, step_dev, st);
    if (t->D == 256 && t->prec == PREC_F32) return launch_fwd_group<PREC_F32, 256, 0>(a, B, training, seed, step, step_dev, st);
    m2m_set_error("towers_forward (wide): hidden_dim 256 only", __FILE__, __LINE__);
    return -1;
}
bool m2m_can_group_wide(const m2m_tower* a, const m2m_tower* b, int B);       // token_wide.hip
int m2m_forward_wide_group(const m2m_tower* const* tw, const m2m_tower_io* io, int B, int training, unsigned int seed,
                           unsigned int step, const unsigned int* step_dev, hipStream_t st);

// True when two towers can share one chain launch: both on the fused path, same kernel instantiation, <= 4 blocks each.
bool m2m_can_group(const m2m_tower* a, const m2m_tower* b) {
    if (m2m_is_wide(a) || m2m_is_wide(b)) return false;
    if (a->prec != b->prec || a->D != b->D || a->p_drop != b->p_drop) return false;
    if ((a->N <= 4) != (b->N <= 4) || (a->T % 16 == 0) != (b->T % 16 == 0)) return false;
    return a->nblocks <= M2M_GROUP_BLOCKS && b->nblocks <= M2M_GROUP_BLOCKS;
}

extern "C" int m2m_towers_can_group(const m2m_tower* a, const m2m_tower* b, int B) {
    if (!a || !b || B < 1) return 0;
    return (m2m_can_group(a, b) || m2m_can_group_wide(a, b, B)) ? 1 : 0;
}

int m2m_check_embed(const m2m_embed* e, int B);          // embed.hip
// 1: m2m_towers_forward_embeds takes these towers / embeddings at batch B (fused-path pair, whole samples per 16-row tile,
// embedding and tower of one hidden_dim / precision); 0: run m2m_embeds_forward + m2m_towers_forward
extern "C" int m2m_towers_forward_embeds_ok(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int B) {
    if (!towers || !embeds || ntowers != 2 || B < 1) return 0;
    for (int i = 0; i < 2; ++i) {
        if (!towers[i] || m2m_check_tower(towers[i], B) != 0) return 0;
        if (m2m_is_wide(towers[i]) || BM % towers[i]->N != 0 || m2m_split_eligible(towers[i], B, 1)) return 0;
        const m2m_embed* e = embeds[i];
        if (!e) continue;
        if (m2m_check_embed(e, B) != 0 || e->prec != towers[i]->prec || e->D != towers[i]->D) return 0;
        if ((e->H / e->ph) * (e->W / e->pw) != towers[i]->N) return 0;
    }
    return m2m_can_group(towers[0], towers[1]) ? 1 : 0;
}
static int towers_forward_impl(const m2m_tower* const* towers, const m2m_tower_io* io, int ntowers, const m2m_embed* const* embeds,
                               const float* const* inputs, const m2m_step_head* head, int B, int training,
                               uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);
extern "C" int m2m_towers_forward(const m2m_tower* const* towers, const m2m_tower_io* io, int ntowers, int B, int training,
                                  uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    return towers_forward_impl(towers, io, ntowers, nullptr, nullptr, nullptr, B, training, seed, step, step_dev, stream);
}
extern "C" int m2m_towers_forward_embeds(const m2m_tower* const* towers, const m2m_tower_io* io, int ntowers,
                                         const m2m_embed* const* embeds, const float* const* inputs, const m2m_step_head* head,
                                         int B, int training, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (!embeds || !inputs) { m2m_set_error("towers_forward_embeds: embeds / inputs are required", __FILE__, __LINE__); return -1; }
    if (!m2m_towers_forward_embeds_ok(towers, ntowers, embeds, B)) {
        m2m_set_error("towers_forward_embeds: unsupported towers / embeddings (see m2m_towers_forward_embeds_ok)", __FILE__, __LINE__);
        return -1;
    }
    if (head && head->drop_counter) {
        m2m_set_error("towers_forward_embeds: this launch READS the dropout counter; advance it behind the step's last reader (m2m_towers_wgrad_tail)", __FILE__, __LINE__);
        return -1;
    }
    if (head && (head->nlosses < 0 || head->nlosses > 64)) { m2m_set_error("towers_forward_embeds: step head nlosses must be in [0, 64]", __FILE__, __LINE__); return -1; }
    for (int i = 0; i < 2; ++i)
        if (embeds[i] && (!inputs[i] || !io[i].x0 || io[i].x0_sample_stride != (int64_t)towers[i]->N * towers[i]->D)) {
            m2m_set_error("towers_forward_embeds: an embedded tower needs its input and a dense (B N, D) x0 scratch", __FILE__, __LINE__);
            return -1;
        }
    return towers_forward_impl(towers, io, ntowers, embeds, inputs, head, B, training, seed, step, step_dev, stream);
}
static int towers_forward_impl(const m2m_tower* const* towers, const m2m_tower_io* io, int ntowers, const m2m_embed* const* embeds,
                               const float* const* inputs, const m2m_step_head* head, int B, int training,
                               uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (!towers || !io || ntowers != 2) { m2m_set_error("towers_forward: exactly two towers per launch", __FILE__, __LINE__); return -1; }
    for (int i = 0; i < 2; ++i)
        if (int rc = m2m_check_tower(towers[i], B)) return rc;
    // large batches: per-block launches with column-split channel mixing (csrc/split.h)
    if (m2m_split_eligible(towers[0], B, training) && m2m_split_eligible(towers[1], B, training) &&
        m2m_split_can_group(towers[0], towers[1]))
        return m2m_split_forward(towers, io, 2, B, training, seed, step, step_dev, reinterpret_cast<hipStream_t>(stream));
    if (m2m_can_group_wide(towers[0], towers[1], B))
        return m2m_forward_wide_group(towers, io, B, training, seed, step, step_dev, reinterpret_cast<hipStream_t>(stream));
    if (!m2m_can_group(towers[0], towers[1])) {
        m2m_set_error("towers_forward: the two towers do not share a kernel instantiation (fused path, precision, hidden_dim, "
                      "dropout, token class, <= 4 blocks): launch them separately", __FILE__, __LINE__);
        return -1;
    }
    FwdGroupArgs a;
    memset(&a, 0, sizeof(a));
    if (head) { a.head_losses = head->losses; a.head_adam_state = head->adam_state; a.head_nlosses = head->nlosses; }
    for (int i = 0; i < 2; ++i) {
        a.tw[i] = m2m_shrink(towers[i]);
        a.x0[i] = io[i].x0; a.x0_ss[i] = (long)io[i].x0_sample_stride;
        if (embeds && embeds[i]) {
            a.embed[i] = 1; a.em[i] = *embeds[i]; a.ein[i] = inputs[i];
            a.efast[i] = embed_fwd_fast_ok(embeds[i], inputs[i]) ? 1 : 0;
        }
        a.x0_parts[i] = (io[i].x0_parts > 1 && !a.embed[i]) ? io[i].x0_parts : 1;
        a.x0_pstride[i] = (long)io[i].x0_part_stride;
        if (a.x0_parts[i] > 4) { m2m_set_error("towers_forward: at most 4 input parts", __FILE__, __LINE__); return -1; }
        a.out[i] = io[i].out; a.out_ss[i] = (long)io[i].out_sample_stride;
        a.pooled[i] = io[i].pooled;
        const int SPW = BM / towers[i]->N;
        a.ntiles[i] = (B + SPW - 1) / SPW;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const m2m_tower* t = towers[0];
#define M2M_FWDG_CASE(PP, DD) \
    if (t->prec == PP && t->D == DD) return t->N <= 4 ? launch_fwd_group<PP, DD, 4>(a, B, training, seed, step, step_dev, st) \
                                                      : launch_fwd_group<PP, DD, 8>(a, B, training, seed, step, step_dev, st);
    M2M_FWDG_CASE(PREC_BF16, 32) M2M_FWDG_CASE(PREC_BF16, 64) M2M_FWDG_CASE(PREC_BF16, 128)
    M2M_FWDG_CASE(PREC_F32, 32) M2M_FWDG_CASE(PREC_F32, 64) M2M_FWDG_CASE(PREC_F32, 128)
#undef M2M_FWDG_CASE
    m2m_set_error("towers_forward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_tower_forward(const m2m_tower* t, const float* x0, int64_t x0_ss, int B, float* out, int64_t out_ss,
                                 float* pooled, int training, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (int rc = m2m_check_tower(t, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (m2m_is_wide(t)) return m2m_forward_wide(t, x0, x0_ss, B, out, out_ss, pooled, training, seed, step, step_dev, st);
    if (m2m_split_eligible(t, B, training)) {
        m2m_tower_io io1;
        io1.x0 = x0; io1.x0_sample_stride = x0_ss; io1.out = out; io1.out_sample_stride = out_ss; io1.pooled = pooled;
        io1.x0_parts = 1; io1.x0_part_stride = 0;
        return m2m_split_forward(&t, &io1, 1, B, training, seed, step, step_dev, st);
    }
#define M2M_FWD_CASE(PP, DD) \
    if (t->prec == PP && t->D == DD) return t->N <= 4 ? launch_fwd<PP, DD, 4>(t, x0, x0_ss, B, out, out_ss, pooled, training, seed, step, step_dev, st) \
                                                      : launch_fwd<PP, DD, 8>(t, x0, x0_ss, B, out, out_ss, pooled, training, seed, step, step_dev, st);
    M2M_FWD_CASE(PREC_BF16, 32) M2M_FWD_CASE(PREC_BF16, 64) M2M_FWD_CASE(PREC_BF16, 128)
    M2M_FWD_CASE(PREC_F32, 32) M2M_FWD_CASE(PREC_F32, 64) M2M_FWD_CASE(PREC_F32, 128)
#undef M2M_FWD_CASE
    m2m_set_error("tower_forward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}
#endif   // M2M_ISA_PROBE
