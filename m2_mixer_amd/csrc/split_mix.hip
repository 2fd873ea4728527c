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
    if (has_block) gelu_tab_fill(gtab, tid, NTHREADS);

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
                for (int s = 0; s < tw.nslab; ++s) {
                    const float4 u = *reinterpret_cast<const float4*>(tw.slabs + s * tw.slab_stride + off);
                    y.x += u.x; y.y += u.y; y.z += u.z; y.w += u.w;
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
                h = Act<P>::gelu(gtab, h) * dr_th.scale;
                h = ((wth >> t) & 1u) ? h : 0.f;
#pragma unroll
                for (int n = 0; n < NMAX; ++n) o[n] = __builtin_fmaf(wr[NMAX + n], h, o[n]);
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n)
                if (n < N) xs[(sl * N + n) * XLD + d] += ((wto >> n) & 1u) ? o[n] * dr_to.scale : 0.f;
        }
        __syncthreads();
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
        pack_tile_nat<P, D>(ub, tw.a_nat + (long)wg * KD * 1024, tid);
        if (tw.at_chn) {
            constexpr int TPP = WPAIR / BM;
            pack_tile_chn_t<P, D>(ub, tw.at_chn + (long)(wg / TPP) * (WPAIR * D * Prec<P>::ESZ), wg % TPP, tid);
        }
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
