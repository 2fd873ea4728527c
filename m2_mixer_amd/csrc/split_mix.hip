// Mix launches of the split path (see split.h): everything of a MixerBlock that is NOT the two channel-mixing GEMMs, one
// workgroup per BM = 16 token rows (whole samples), 512 threads.
//
// Reference semantics: MixerBlock.forward (modules/mixer.py:42-47) around the channel MLP -- LayerNorm, the token-mixing MLP
// over the N tokens of a sample (:30-35), the residual adds, the dropout of the channel MLP's output (:18), the tower's final
// LayerNorm (:131, :161, :185) -- and the backward of all of it.
//
//   forward  mix(b):  x = [b == 0: tower input | x_mid(b-1) + dropout(sum_s Yslab_s + b2(b-1))]
//                     -> x_in(b); LN1 -> token MLP -> +residual -> x_mid(b); LN2 -> bf16 operand images a_nat(b) / at_chn(b)
//            final:   x = x_mid(last) + dropout(sum_s Yslab_s + b2) -> x_final; LayerNorm -> out, token mean
//   backward mix(b):  g = [b == last: upstream through the final LayerNorm | carry + LN2'(sum_s dAslab_s) of block b+1, then the
//                     token-mixing backward of block b+1]; carry = g; dYd(b) = dropout mask x g -> db2, operand images
//                     dy_nat(b) / dyt_chn(b);   after block 0's chain launch: the same without a next block -> d_x0
#include "split.h"

TIMER_DECL(g_tm_smf);
TIMER_READER(m2m_debug_timers_smf, g_tm_smf)
TIMER_DECL(g_tm_smb);
TIMER_READER(m2m_debug_timers_smb, g_tm_smb)

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------
template <int D, int NMAX>
static size_t mix_fwd_lds() {
    return (size_t)(2 * BM * TileGeom<D>::XLD) * sizeof(float) + (32 * (2 * NMAX + 4) + 8) * sizeof(float) + GELU_TAB_N * 16;
}

template <int D, int NMAX, int DM>
__global__ __launch_bounds__(NTHREADS) void split_mix_fwd_kernel(const SplitMixArgs a, int training, unsigned int seed,
                                                                 unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    typedef TileGeom<D> G;
    constexpr int P = PREC_BF16, XLD = G::XLD, TW_LD = 2 * NMAX + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);                   // residual stream [BM][XLD]
    float* ub = xs + BM * XLD;                                     // scratch tile
    float* tokw = ub + BM * XLD;                                   // [32][TW_LD]
    float* tokb2 = tokw + 32 * TW_LD;                              // [NMAX]
    gtab_t* gtab = reinterpret_cast<gtab_t*>(tokb2 + 8);

    const int ti = blockIdx.y;
    const SplitMixTower& tw = a.t[ti];
    const int wg = blockIdx.x;
    if (wg >= tw.ntiles) return;
    const int tid = threadIdx.x;
    const int N = tw.N, T = tw.T, B = tw.B;
    const int SPW = BM / N, s0 = wg * SPW, ns = min(SPW, B - s0);
    const long row0 = (long)s0 * N;
    const int R = ns * N;
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    const bool has_block = tw.blk.ln1_w != nullptr;
    TIMER_START();
    if (has_block) gelu_tab_fill(gtab, make_drop(training, tw.p_drop, 0u, 0u, 0u).scale, tid, NTHREADS);

    // ---- the residual stream entering this launch (rows >= R are zero) ----
    if (tw.xprev == nullptr) {
        _Pragma("unroll 1") for (int idx = tid; idx < BM * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R) {
                const long gr = row0 + r;
                const float* src = tw.x0 + (gr / N) * tw.x0_ss + (gr % N) * D + c;
                v = *reinterpret_cast<const float4*>(src);
                for (int p = 1; p < tw.x0_parts; ++p) {
                    const float4 u = *reinterpret_cast<const float4*>(src + p * tw.x0_pstride);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
            }
            *reinterpret_cast<float4*>(xs + r * XLD + c) = v;
        }
    } else {
        // x = x_mid(prev) + dropout(sum of the column-split partial results + b2)      (modules/mixer.py:17-18, :45)
        const Drop dr_co = make_drop(training, tw.p_drop, seed, step, tw.site_prev_out);
        _Pragma("unroll 1") for (int idx = tid; idx < BM * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R) {
                const long off = (row0 + r) * D + c;
                float4 y = *reinterpret_cast<const float4*>(tw.b2prev + c);
                // all slab loads in flight together (a runtime-length loop issues them one L2 round trip after another);
                // slots past nslab re-read the last slab and are discarded
                float4 u[SP_MAX_SPLITS];
#pragma unroll
                for (int s = 0; s < SP_MAX_SPLITS; ++s)
                    u[s] = *reinterpret_cast<const float4*>(tw.slabs + (long)min(s, tw.nslab - 1) * tw.slab_stride + off);
#pragma unroll
                for (int s = 0; s < SP_MAX_SPLITS; ++s) {
                    const float k = s < tw.nslab ? 1.f : 0.f;
                    y.x = __builtin_fmaf(k, u[s].x, y.x); y.y = __builtin_fmaf(k, u[s].y, y.y);
                    y.z = __builtin_fmaf(k, u[s].z, y.z); y.w = __builtin_fmaf(k, u[s].w, y.w);
                }
                const unsigned int e0 = (unsigned int)off;
                y.x = drop_keep_elem<DM>(dr_co, e0 + 0) ? y.x * dr_co.scale : 0.f;
                y.y = drop_keep_elem<DM>(dr_co, e0 + 1) ? y.y * dr_co.scale : 0.f;
                y.z = drop_keep_elem<DM>(dr_co, e0 + 2) ? y.z * dr_co.scale : 0.f;
                y.w = drop_keep_elem<DM>(dr_co, e0 + 3) ? y.w * dr_co.scale : 0.f;
                v = *reinterpret_cast<const float4*>(tw.xprev + off);
                v.x += y.x; v.y += y.y; v.z += y.z; v.w += y.w;
            }
            *reinterpret_cast<float4*>(xs + r * XLD + c) = v;
        }
    }
    __syncthreads();
    TIMER_MARK(g_tm_smf, 0);   // table + input (slabs)

    if (has_block) {
        const m2m_block& bk = tw.blk;
        const Drop dr_th = make_drop(training, tw.p_drop, seed, step, tw.site + 0);
        const Drop dr_to = make_drop(training, tw.p_drop, seed, step, tw.site + 1);
        // ---- save the block input; LN1 -> ub; token-MLP weights -> LDS, zero-padded to NMAX tokens ----
        if (tw.x_in) {
            _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
                const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
                *reinterpret_cast<float4*>(tw.x_in + (row0 + r) * D + c) = *reinterpret_cast<const float4*>(xs + r * XLD + c);
            }
        }
        _Pragma("unroll 1") for (int idx = tid; idx < T * TW_LD; idx += NTHREADS) {
            const int t = idx / TW_LD, j = idx % TW_LD;
            float v = 0.f;
            if (j < NMAX) { if (j < N) v = bk.tok_w1[t * N + j]; }
            else if (j < 2 * NMAX) { if (j - NMAX < N) v = bk.tok_w2[(j - NMAX) * T + t]; }
            else if (j == 2 * NMAX) v = bk.tok_b1[t];
            tokw[idx] = v;
        }
        if (tid < NMAX) tokb2[tid] = tid < N ? bk.tok_b2[tid] : 0.f;
        ln_to_tile<D>(xs, ub, bk.ln1_w, bk.ln1_b, tid);
        __syncthreads();
        TIMER_MARK(g_tm_smf, 1);   // x_in, token weights, LN1
        // ---- token mixing: one thread per (sample, channel) column (modules/mixer.py:30-35) ----
        _Pragma("unroll 1") for (int p = tid; p < ns * D; p += NTHREADS) {
            const int sl = p / D, d = p % D;
            const unsigned int bd = (unsigned int)(s0 + sl) * D + d;
            float un[NMAX], o[NMAX];
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                un[n] = (n < N) ? ub[(sl * N + n) * XLD + d] : 0.f;
                o[n] = tokb2[n];
            }
            const unsigned int wth = drop_row_bits<DM>(dr_th, bd, T);
            const unsigned int wto = drop_row_bits<DM>(dr_to, bd, N);
#pragma unroll 4
            for (int t = 0; t < T; ++t) {
                const float* wr = tokw + t * TW_LD;
                float h = wr[2 * NMAX];
#pragma unroll
                for (int n = 0; n < NMAX; ++n) h = __builtin_fmaf(wr[n], un[n], h);
                h = Act<P>::gelu_scaled(gtab, h, dr_th.scale);
                h = DM == DM_NONE ? h : mask_f(h, bit_to_mask(wth, t));
#pragma unroll
                for (int n = 0; n < NMAX; ++n) o[n] = __builtin_fmaf(wr[NMAX + n], h, o[n]);
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n)
                if (n < N) xs[(sl * N + n) * XLD + d] += ((wto >> n) & 1u) ? o[n] * dr_to.scale : 0.f;
        }
        __syncthreads();
        TIMER_MARK(g_tm_smf, 2);   // token mixing
        // ---- x_mid (saved activation and the carry to the next mix launch); LN2 -> operand images ----
        _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            *reinterpret_cast<float4*>(tw.x_mid + (row0 + r) * D + c) = *reinterpret_cast<const float4*>(xs + r * XLD + c);
        }
        ln_to_tile<D>(xs, ub, bk.ln2_w, bk.ln2_b, tid);
        __syncthreads();
        if (R < BM) {                                            // rows past the batch: zero operands
            _Pragma("unroll 1") for (int idx = tid; idx < (BM - R) * D; idx += NTHREADS) ub[(R + idx / D) * XLD + idx % D] = 0.f;
            __syncthreads();
        }
        constexpr int KD = D / Prec<P>::KB;
        TIMER_MARK(g_tm_smf, 3);   // x_mid, LN2
        pack_tile_nat<P, D>(ub, tw.a_nat + (long)wg * KD * 1024, tid);
        if (tw.at_chn) {
            constexpr int TPP = WPAIR / BM;
            pack_tile_chn_t<P, D>(ub, tw.at_chn + (long)(wg / TPP) * (WPAIR * D * Prec<P>::ESZ), wg % TPP, tid);
        }
        TIMER_MARK(g_tm_smf, 4);   // operand images
        return;
    }

    // ---- final LayerNorm (modules/mixer.py:131,161,185), output + token mean ----
    if (tw.x_final) {
        _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            *reinterpret_cast<float4*>(tw.x_final + (row0 + r) * D + c) = *reinterpret_cast<const float4*>(xs + r * XLD + c);
        }
    }
    const float* res = xs;
    if (tw.lnf_w) {
        ln_to_tile<D>(xs, ub, tw.lnf_w, tw.lnf_b, tid);
        res = ub;
        __syncthreads();
    }
    _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
        const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
        const long gr = row0 + r;
        *reinterpret_cast<float4*>(tw.out + (gr / N) * tw.out_ss + (gr % N) * D + c) = *reinterpret_cast<const float4*>(res + r * XLD + c);
    }
    if (tw.pooled) {
        const float inv = 1.0f / (float)N;
        _Pragma("unroll 1") for (int p = tid; p < ns * D; p += NTHREADS) {
            const int sl = p / D, d = p % D;
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += res[(sl * N + n) * XLD + d];
            tw.pooled[(long)(s0 + sl) * D + d] = s * inv;
        }
    }
}

}  // namespace

template <int D, int NMAX, int DM>
static int launch_mix_fwd_dm(const SplitMixArgs& a, int training, unsigned int seed, unsigned int step, const unsigned int* step_dev,
                             hipStream_t st) {
    const size_t lds = mix_fwd_lds<D, NMAX>();
    auto kern = split_mix_fwd_kernel<D, NMAX, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    int mx = 0;
    for (int i = 0; i < a.ntow; ++i) mx = a.t[i].ntiles > mx ? a.t[i].ntiles : mx;
    hipLaunchKernelGGL(kern, dim3(mx, a.ntow), dim3(NTHREADS), lds, st, a, training, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

int m2m_split_mix_forward(const SplitMixArgs& a, int D, int training, float p_drop, unsigned int seed, unsigned int step,
                          const unsigned int* step_dev, hipStream_t st) {
    if (D != 128) { m2m_set_error("split path: hidden_dim 128 only in this build", __FILE__, __LINE__); return -1; }
    const int dm = m2m_drop_mode(training, p_drop);
    const bool n4 = a.t[0].N <= 4;
#define M2M_MIXF(NM, DMV) return launch_mix_fwd_dm<128, NM, DMV>(a, training, seed, step, step_dev, st)
    if (n4) { if (dm == DM_NONE) M2M_MIXF(4, DM_NONE); if (dm == DM_HALF) M2M_MIXF(4, DM_HALF); M2M_MIXF(4, DM_GEN); }
    if (dm == DM_NONE) M2M_MIXF(8, DM_NONE);
    if (dm == DM_HALF) M2M_MIXF(8, DM_HALF);
    M2M_MIXF(8, DM_GEN);
#undef M2M_MIXF
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

template <int D, int NMAX, int TG>
static size_t mix_bwd_lds() {
    const size_t tile_b = (size_t)BM * TileGeom<D>::XLD * sizeof(float);
    return 4 * tile_b + BM * sizeof(float) + GELU_TAB_N * 16 + (size_t)NWAVES * ((32 / TG) * (1 + 2 * NMAX) + NMAX) * TG * sizeof(float) +
           32 * (2 * NMAX + 4) * sizeof(float);
}

template <int D, int NMAX, int TG, int DM>
__global__ __launch_bounds__(NTHREADS) void split_mix_bwd_kernel(const SplitMixBwdArgs a, unsigned int seed, unsigned int step_host,
                                                                 const unsigned int* __restrict__ step_dev) {
    typedef TileGeom<D> G;
    constexpr int P = PREC_BF16, XLD = G::XLD, TILE_F = BM * XLD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dxs = reinterpret_cast<float*>(smem);        // gradient stream
    float* ub = dxs + TILE_F;                            // scratch
    float* xh = ub + TILE_F;                             // scratch
    float* dasum = xh + TILE_F;                          // summed dA of the upper block
    float* rstd_s = dasum + TILE_F;                      // [BM]
    gtab_t* gtab = reinterpret_cast<gtab_t*>(rstd_s + BM);
    constexpr int RED_LD = ((32 / TG) * (1 + 2 * NMAX) + NMAX) * TG;
    float* red = reinterpret_cast<float*>(gtab + GELU_TAB_N);        // [NWAVES][RED_LD]
    constexpr int TW_LD = 2 * NMAX + 4;
    float* tokw = red + NWAVES * RED_LD;                             // [32][TW_LD]

    const SplitMixBwdTower& tw = a.t[blockIdx.y];
    const int wg = blockIdx.x;
    if (wg >= tw.ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = tw.N, T = tw.T, B = tw.B;
    const int SPW = BM / N, s0 = wg * SPW, ns = min(SPW, B - s0);
    const long row0 = (long)s0 * N;
    const int R = ns * N;
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    // this workgroup's partial-sum slot: the small gradients are STORED here and summed over the workgroups by the reduction
    // launch (split_small_grads_kernel) -- 256 workgroups adding to the same ~900 addresses ran at the contended-atomic rate
    float* mypart = tw.part + (long)wg * SPP_STRIDE;
    TIMER_START();

    if (!tw.has_upper) {
        // ---- upstream gradient of the tower output (+ token-mean head), through the final LayerNorm ----
        const float invN = 1.0f / (float)N;
        _Pragma("unroll 1") for (int idx = tid; idx < BM * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R) {
                const long gr = row0 + r, gs = gr / N;
                if (tw.d_out) v = *reinterpret_cast<const float4*>(tw.d_out + gs * tw.d_out_ss + (gr % N) * D + c);
                if (tw.d_pooled) {
                    const float4 p = *reinterpret_cast<const float4*>(tw.d_pooled + gs * D + c);
                    v.x += p.x * invN; v.y += p.y * invN; v.z += p.z * invN; v.w += p.w * invN;
                }
            }
            *reinterpret_cast<float4*>((tw.lnf_w ? ub : dxs) + r * XLD + c) = v;
        }
        __syncthreads();
        if (tw.lnf_w) ln_backward_tile<D, false>(tw.x_final + row0 * D, R, ub, tw.lnf_w, dxs, false, xh, mypart + SPP_LNF(D), mypart + SPP_LNF(D) + D, tid);
    } else {
        const m2m_block& bk = tw.up;
        if (Act<P>::USES_TABLE) gelu_tab_fill(gtab, make_drop(true, tw.p_drop, 0u, 0u, 0u).scale, tid, NTHREADS);
        const Drop dr_th = make_drop(true, tw.p_drop, seed, step, tw.site_up + 0);
        const Drop dr_to = make_drop(true, tw.p_drop, seed, step, tw.site_up + 1);
        // ---- gradient stream carried from the previous launch; dA = sum of the column-split partial results ----
        _Pragma("unroll 1") for (int idx = tid; idx < BM * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f), y = v;
            if (r < R) {
                const long off = (row0 + r) * D + c;
                v = *reinterpret_cast<const float4*>(tw.carry + off);
                float4 u[SP_MAX_SPLITS];
#pragma unroll
                for (int s = 0; s < SP_MAX_SPLITS; ++s)
                    u[s] = *reinterpret_cast<const float4*>(tw.slabs + (long)min(s, tw.nslab - 1) * tw.slab_stride + off);
#pragma unroll
                for (int s = 0; s < SP_MAX_SPLITS; ++s) {
                    const float k = s < tw.nslab ? 1.f : 0.f;
                    y.x = __builtin_fmaf(k, u[s].x, y.x); y.y = __builtin_fmaf(k, u[s].y, y.y);
                    y.z = __builtin_fmaf(k, u[s].z, y.z); y.w = __builtin_fmaf(k, u[s].w, y.w);
                }
            }
            *reinterpret_cast<float4*>(dxs + r * XLD + c) = v;
            *reinterpret_cast<float4*>(dasum + r * XLD + c) = y;
        }
        __syncthreads();
        TIMER_MARK(g_tm_smb, 0);   // table, carry + slabs
        // ---- LayerNorm-2 backward; dx_mid = dY + LN2'(dA) ----
        ln_backward_tile<D, false>(bk.x_mid + row0 * D, R, dasum, bk.ln2_w, dxs, true, xh, mypart + SPP_LN2, mypart + SPP_LN2 + D, tid);

        TIMER_MARK(g_tm_smb, 1);   // LN2 backward
        // ================= token mixing backward (as tower_bwd.hip) =================
        _Pragma("unroll 1") for (int idx = tid; idx < 32 * TW_LD; idx += NTHREADS) {
            const int t = idx / TW_LD, j = idx % TW_LD;
            float v = 0.f;
            if (t < T) {
                if (j < NMAX) { if (j < N) v = bk.tok_w1[t * N + j]; }
                else if (j < 2 * NMAX) { if (j - NMAX < N) v = bk.tok_w2[(j - NMAX) * T + t]; }
                else if (j == 2 * NMAX) v = bk.tok_b1[t];
            }
            tokw[idx] = v;
        }
        {
            const int r = tid / TPR, j = tid % TPR;
            float v[D / TPR], mean, rstd;
            row_stats<D>(bk.x_in + (row0 + r) * D, r < R, j, v, mean, rstd);
            if (j == 0) rstd_s[r] = rstd;
#pragma unroll
            for (int e = 0; e < D / TPR; ++e) {
                const int c = ln_col<D>(e, j);
                const float xhv = (v[e] - mean) * rstd;
                xh[r * XLD + c] = xhv;
                ub[r * XLD + c] = xhv * bk.ln1_w[c] + bk.ln1_b[c];
            }
        }
        __syncthreads();
        TIMER_MARK(g_tm_smb, 2);   // token weights, LN1 recompute
        {
            constexpr int TTMAX = 32 / TG;                 // hidden units per lane (T <= 32)
            const int tg = tid % TG, pl = tid / TG;        // TG lanes share a column and split its T hidden units
            const int TT = T / TG;
            float w1r[TTMAX][NMAX], w2r[NMAX][TTMAX], b1r[TTMAX];
            float aw1[TTMAX][NMAX], aw2[NMAX][TTMAX], ab1[TTMAX], ab2[NMAX];
#pragma unroll
            for (int tt = 0; tt < TTMAX; ++tt) {
                const float* wr = tokw + ((tg * TT + tt) & 31) * TW_LD;
                b1r[tt] = (tt < TT) ? wr[2 * NMAX] : 0.f;
                ab1[tt] = 0.f;
#pragma unroll
                for (int n = 0; n < NMAX; ++n) {
                    w1r[tt][n] = (tt < TT) ? wr[n] : 0.f;
                    w2r[n][tt] = (tt < TT) ? wr[NMAX + n] : 0.f;
                    aw1[tt][n] = 0.f;
                    aw2[n][tt] = 0.f;
                }
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n) ab2[n] = 0.f;

            const int npairs_tok = ns * D;
            constexpr int PL = NTHREADS / TG;             // columns handled concurrently
            const int iters = (npairs_tok + PL - 1) / PL;
            for (int it = 0; it < iters; ++it) {
                const int p = it * PL + pl;
                const bool pv = p < npairs_tok;          // keep all lanes in the shuffles below
                const int sl = pv ? p / D : 0, d = pv ? p % D : 0;
                const unsigned int bd = (unsigned int)(s0 + sl) * D + d;
                float un[NMAX], dv[NMAX], du[NMAX];
                const unsigned int wth = drop_row_bits<DM>(dr_th, bd, T);
                const unsigned int wto = drop_row_bits<DM>(dr_to, bd, N);
#pragma unroll
                for (int n = 0; n < NMAX; ++n) {
                    un[n] = 0.f; dv[n] = 0.f; du[n] = 0.f;
                    if (pv && n < N) {
                        un[n] = ub[(sl * N + n) * XLD + d];
                        const float v = dxs[(sl * N + n) * XLD + d] * dr_to.scale;
                        dv[n] = ((wto >> n) & 1u) ? v : 0.f;
                    }
                }
#pragma unroll
                for (int tt = 0; tt < TTMAX; ++tt) {
                    const int t = tg * TT + tt;
                    float h = b1r[tt], dh = 0.f;
#pragma unroll
                    for (int n = 0; n < NMAX; ++n) {
                        h = __builtin_fmaf(w1r[tt][n], un[n], h);
                        dh = __builtin_fmaf(w2r[n][tt], dv[n], dh);
                    }
                    float gl, dgl;                                   // both carry the dropout scale
                    Act<P>::gelu_grad_scaled(gtab, h, dr_th.scale, gl, dgl);
                    const bool keep = (wth >> (t & 31)) & 1u;
                    const float hact = keep ? gl : 0.f;
                    const float dhp = (keep && pv) ? dh * dgl : 0.f;
                    ab1[tt] += dhp;
#pragma unroll
                    for (int n = 0; n < NMAX; ++n) {
                        aw2[n][tt] = __builtin_fmaf(dv[n], hact, aw2[n][tt]);
                        aw1[tt][n] = __builtin_fmaf(dhp, un[n], aw1[tt][n]);
                        du[n] = __builtin_fmaf(dhp, w1r[tt][n], du[n]);
                    }
                }
#pragma unroll
                for (int n = 0; n < NMAX; ++n) {
                    if (n < N) {
                        const float s = wave_sum_xor(du[n], TG);     // over the TG lanes that share the column
                        if (pv && tg == 0) {
                            ub[(sl * N + n) * XLD + d] = s;
                            ab2[n] += dv[n];
                        }
                    }
                }
            }
            TIMER_MARK(g_tm_smb, 3);   // token pair loop
            // token-weight gradients: cross-lane sums, per-wave LDS slots, one partial-sum store per value and workgroup
            constexpr int KS = 1 + 2 * NMAX;
            float* myred = red + wave * RED_LD + tg;
#pragma unroll
            for (int tt = 0; tt < TTMAX; ++tt) {
                const float s = lane_class_sum(ab1[tt], TG);
                if (lane < TG) myred[(tt * KS) * TG] = s;
#pragma unroll
                for (int n = 0; n < NMAX; ++n) {
                    const float aa = lane_class_sum(aw1[tt][n], TG), cc = lane_class_sum(aw2[n][tt], TG);
                    if (lane < TG) {
                        myred[(tt * KS + 1 + n) * TG] = aa;
                        myred[(tt * KS + 1 + NMAX + n) * TG] = cc;
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                const float s = lane_class_sum(ab2[n], TG);
                if (lane < TG) myred[(TTMAX * KS + n) * TG] = s;
            }
            __syncthreads();
            const int nred = 2 * T * N + T + N;
            for (int i = tid; i < nred; i += NTHREADS) {               // slot order: W1 (T N) | W2 (N T) | b1 (T) | b2 (N)
                int t, k;
                if (i < T * N)              { t = i / N; k = 1 + i % N; }
                else if (i < 2 * T * N)     { const int j = i - T * N; t = j % T; k = 1 + NMAX + j / T; }
                else if (i < 2 * T * N + T) { t = i - 2 * T * N; k = 0; }
                else                        { t = -1; k = i - 2 * T * N - T; }
                const int slot = t < 0 ? (TTMAX * KS + k) * TG : ((t % TT) * KS + k) * TG + t / TT;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVES; ++w) v += red[w * RED_LD + slot];
                mypart[SPP_TOK(D) + i] = v;
            }
        }
        __syncthreads();
        TIMER_MARK(g_tm_smb, 4);   // token-gradient reduction
        // LayerNorm-1 backward: dx_in = dx_mid + LN1'(dU); gamma / beta gradients
        {
            const int r = tid / TPR, j = tid % TPR;
            const bool valid = r < R;
            const float rstd = rstd_s[r];
            float gv[D / TPR], xv[D / TPR];
            float gsum = 0.f, gxsum = 0.f;
#pragma unroll
            for (int e = 0; e < D / TPR; ++e) {
                const int c = ln_col<D>(e, j);
                const float u = ub[r * XLD + c];
                const float xhv = xh[r * XLD + c];
                const float gg = u * bk.ln1_w[c];
                gv[e] = gg; xv[e] = xhv;
                gsum += gg;
                gxsum = __builtin_fmaf(gg, xhv, gxsum);
                xh[r * XLD + c] = valid ? u * xhv : 0.f;
            }
            gsum = wave_sum_xor(gsum, TPR) * (1.0f / D);
            gxsum = wave_sum_xor(gxsum, TPR) * (1.0f / D);
            if (valid) {
#pragma unroll
                for (int e = 0; e < D / TPR; ++e) {
                    const int c = ln_col<D>(e, j);
                    dxs[r * XLD + c] += rstd * (gv[e] - gsum - xv[e] * gxsum);
                }
            }
        }
        __syncthreads();
        _Pragma("unroll 1") for (int d = tid; d < 2 * D; d += NTHREADS) {
            const float* src = d < D ? xh : ub;
            const int c = d < D ? d : d - D;
            float s = 0.f;
            for (int rr = 0; rr < R; ++rr) s += src[rr * XLD + c];
            mypart[SPP_LN1(D) + d] = s;
        }
        __syncthreads();
        TIMER_MARK(g_tm_smb, 5);   // LN1 backward
    }

    if (!tw.has_lower) {
        // ---- gradient wrt the tower input ----
        _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            const long gr = row0 + r;
            *reinterpret_cast<float4*>(tw.d_x0 + (gr / N) * tw.d_x0_ss + (gr % N) * D + c) = *reinterpret_cast<const float4*>(dxs + r * XLD + c);
        }
        return;
    }
    // ---- the lower block: carry the stream, dYd = dY * mask_out, its column sums (db2), operand images ----
    const Drop dr_co = make_drop(true, tw.p_drop, seed, step, tw.site_lower_out);
    _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
        const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
        *reinterpret_cast<float4*>(tw.carry + (row0 + r) * D + c) = *reinterpret_cast<const float4*>(dxs + r * XLD + c);
    }
    _Pragma("unroll 1") for (int idx = tid; idx < BM * D; idx += NTHREADS) {
        const int r = idx / D, d = idx % D;
        float v = dxs[r * XLD + d];
        v = drop_keep_elem<DM>(dr_co, (unsigned int)(row0 + r) * D + d) ? v * dr_co.scale : 0.f;
        ub[r * XLD + d] = (r < R) ? v : 0.f;
    }
    __syncthreads();
    _Pragma("unroll 1") for (int d = tid; d < D; d += NTHREADS) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += ub[r * XLD + d];
        mypart[SPP_B2(D) + d] = s;
    }
    constexpr int KD = D / Prec<P>::KB, TPP = WPAIR / BM;
    pack_tile_nat<P, D>(ub, tw.dy_nat + (long)wg * KD * 1024, tid);
    pack_tile_chn_t<P, D>(ub, tw.dyt_chn + (long)(wg / TPP) * (WPAIR * D * Prec<P>::ESZ), wg % TPP, tid);
    TIMER_MARK(g_tm_smb, 6);   // carry, dYd, db2, operand images
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// sum of the per-workgroup partial sums of one backward pass into the gradient buffers (+=), deterministic order
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SPR_COLS * SPR_GROUPS) void split_small_grads_kernel(const SplitReduceArgs a) {
    __shared__ float red[SPR_GROUPS * (SPR_COLS + 1)];
    split_small_grads_body<SPR_COLS * SPR_GROUPS>(a.t[blockIdx.z], blockIdx.x, blockIdx.y, red);
}

int m2m_split_small_grads(const SplitReduceArgs& a, hipStream_t st) {
    int nl = 0;
    for (int i = 0; i < a.ntow; ++i) nl = a.t[i].nlaunch > nl ? a.t[i].nlaunch : nl;
    hipLaunchKernelGGL(split_small_grads_kernel, dim3(SPR_NBX, nl, a.ntow), dim3(SPR_COLS * SPR_GROUPS), 0, st, a);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int D, int NMAX, int TG, int DM>
static int launch_mix_bwd_dm(const SplitMixBwdArgs& a, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const size_t lds = mix_bwd_lds<D, NMAX, TG>();
    auto kern = split_mix_bwd_kernel<D, NMAX, TG, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    int mx = 0;
    for (int i = 0; i < a.ntow; ++i) mx = a.t[i].ntiles > mx ? a.t[i].ntiles : mx;
    hipLaunchKernelGGL(kern, dim3(mx, a.ntow), dim3(NTHREADS), lds, st, a, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

int m2m_split_mix_backward(const SplitMixBwdArgs& a, int D, float p_drop, unsigned int seed, unsigned int step,
                           const unsigned int* step_dev, hipStream_t st) {
    if (D != 128) { m2m_set_error("split path: hidden_dim 128 only in this build", __FILE__, __LINE__); return -1; }
    const int dm = m2m_drop_mode(1, p_drop);
    const int N = a.t[0].N, T = a.t[0].T;
#define M2M_MIXB(NM, TGV, DMV) return launch_mix_bwd_dm<128, NM, TGV, DMV>(a, seed, step, step_dev, st)
#define M2M_MIXB_DM(NM, TGV) { if (dm == DM_NONE) M2M_MIXB(NM, TGV, DM_NONE); if (dm == DM_HALF) M2M_MIXB(NM, TGV, DM_HALF); M2M_MIXB(NM, TGV, DM_GEN); }
    if (N <= 4) M2M_MIXB_DM(4, 8)
    if (T % 16 == 0) M2M_MIXB_DM(8, 16)
    M2M_MIXB_DM(8, 8)
#undef M2M_MIXB_DM
#undef M2M_MIXB
}
