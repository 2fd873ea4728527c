// The MLP tower's small-batch bodies (reference: modules/mlp.py:4-27), shared by mlp.hip (launches of their own) and
// token_wide.hip (the MLP as extra workgroups of a token-mixing launch: m2m_mlp_forward_ride / m2m_mlp_backward_ride).
#pragma once
#include "tile.h"

// Workgroup shape (template parameters S = samples per workgroup, T = threads): the kernels are pure latency (dependent LDS
// reads), a workgroup's time is proportional to its samples, and every workgroup adds its weight-gradient partials with
// atomics: 8 samples x 128 threads spreads the cfg batch (128) over 16 CUs; 32 x 256 keeps the atomics of a large batch down.
#define MLP_MAXW 128       // widest layer (LDS: activations + one layer's weights)

// index -> (row, column) of a row-major [rows][cols] array without a runtime division when cols is a power of two (the
// MIMIC widths are 64): 16 divisions per thread and layer were a visible share of these latency-bound kernels
struct Div {
    int d, sh; bool p2;
    __device__ __forceinline__ explicit Div(int dd) : d(dd), sh(31 - __clz(dd)), p2((dd & (dd - 1)) == 0) {}
    __device__ __forceinline__ int q(int i) const { return p2 ? i >> sh : i / d; }
    __device__ __forceinline__ int r(int i) const { return p2 ? i & (d - 1) : i % d; }
};
// MLP_U: outputs a thread accumulates at a time (independent fma chains) = S * 64 / T
#define MLP_SU 16          // global loads a thread keeps in flight in the staging loops (one L2 round trip per batch, not per element)

static __device__ __forceinline__ Drop mlp_drop(const m2m_mlp& m, int layer, int training, unsigned int seed, unsigned int step) {
    return make_drop(training != 0, m.p_drop, seed, step, m.site_base + (unsigned int)layer);
}

// loads a thread keeps in flight in the MFMA bodies' staging batches: the first batch covers a 64 x 64 layer
template <int T> struct MlpSU { static constexpr int value = (MLP_MAXW * MLP_MAXW / 4 + T - 1) / T; };

// ---- small batches: the same arithmetic on the matrix pipe ------------------------------------------------------------
// At the cfg batch (128) the VALU kernels above are a chain of dependent LDS reads on 4-16 workgroups: 50 us forward and
// 67 us backward, ON the critical path of the MIMIC step (the time tower runs beside them and finishes first).  Here a
// workgroup owns ONE 16-sample tile and every product is v_mfma_f32_16x16x4_f32: exact fp32, and the same k-ordered fmaf chain
// as the loops above (bitwise the same forward values).  4 waves; a wave takes the 16-column output tiles jt = wave, wave + 4, ...
#define MLPM_T 256
#define MLPM_S 16
static __device__ __forceinline__ f32x4_t mlp_mfma4(float a, float b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// A k-ordered chain of 16x16x4 products over n4 (a multiple of 4) values of k, the operands of FOUR steps requested together: the
// trip count is a run-time value, so the plain loop was one exposed LDS round trip per product (16 per 64-wide layer).  Same
// products in the same order: bit-identical sums.
template <class FA, class FB>
static __device__ __forceinline__ f32x4_t mlp_chain(int n4, FA fa, FB fb, f32x4_t acc) {
    for (int k0 = 0; k0 < n4; k0 += 16) {
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(k0 + 4 * u, n4 - 4);
            a[u] = fa(k);
            b[u] = fb(k);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + 4 * u < n4) acc = mlp_mfma4(a[u], b[u], acc);      // (uniform)
    }
    return acc;
}
// T: threads of the workgroup that runs the body (256 in the launches of mlp.hip; the token-mixing launches that carry the MLP as
// extra workgroups -- token_wide.hip, m2m_mlp_forward_ride -- have 1024).  wg: 16-sample tile.
template <int T>
static __device__ __forceinline__ void mlp_fwd_mfma_body(const m2m_mlp& m, const float* __restrict__ x, int B, float* __restrict__ out,
                                                         long out_ss, float* __restrict__ out2, int training, unsigned int seed,
                                                         unsigned int step_host, const unsigned int* __restrict__ step_dev, int wg, float* sm) {
    constexpr int SU = MlpSU<T>::value;
    constexpr int LD = MLP_MAXW + 1;
    float* a0 = sm;                                   // [16][LD] activations of the current layer (columns k >= din up to the next multiple of 4: zero)
    float* a1 = a0 + MLPM_S * LD;
    float* wt = a1 + MLPM_S * LD;                     // [din4][LD] transposed weights wt[k][j] = W[j][k]; rows k >= din: zero
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    const int s0 = wg * MLPM_S;
    const int ns = min(MLPM_S, B - s0);
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    {
        const int din = m.dims[0], din4 = (din + 3) & ~3;
        const Div d4(din4);                                // (runtime divisions were a third of these latency-bound kernels)
        for (int i = tid; i < MLPM_S * din4; i += T) {
            const int s = d4.q(i), k = d4.r(i);
            a0[s * LD + k] = (s < ns && k < din) ? x[(long)(s0 + s) * din + k] : 0.f;
        }
    }
    float* cur = a0;
    float* nxt = a1;
    // the first batch of a layer's weights is requested one layer ahead (its round trip hides behind the previous layer's
    // products; a 64 x 64 layer is one batch)
    float wv[SU];
    auto issue = [&](int l) {
        const int din = m.dims[l], dout = m.dims[l + 1], din4 = (din + 3) & ~3;
        const float* __restrict__ w = m.w[l];
        const Div d4(din4);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = tid + u * T, j = d4.q(i), k = d4.r(i);
            wv[u] = (i < din4 * dout && k < din) ? w[j * din + k] : 0.f;
        }
    };
    issue(0);
    for (int l = 0; l < m.nlayers; ++l) {
        const int din = m.dims[l], dout = m.dims[l + 1], din4 = (din + 3) & ~3, dout4 = (dout + 3) & ~3;
        const bool hidden = l < m.nlayers - m.has_out;
        const Drop dr = mlp_drop(m, l, training && hidden, seed, step);
        __syncthreads();                              // the previous layer is done with wt; cur is complete
        {
            const float* __restrict__ w = m.w[l];
            const Div d4(din4);
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int i = tid + u * T;
                if (i < din4 * dout) wt[d4.r(i) * LD + d4.q(i)] = wv[u];
            }
            if (l + 1 < m.nlayers) issue(l + 1);
            for (int i0 = tid + SU * T; i0 < din4 * dout; i0 += SU * T) {
                float v[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = i0 + u * T, j = d4.q(i), k = d4.r(i);
                    v[u] = (i < din4 * dout && k < din) ? w[j * din + k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = i0 + u * T;
                    if (i < din4 * dout) wt[d4.r(i) * LD + d4.q(i)] = v[u];
                }
            }
        }
        __syncthreads();
        for (int jt = wave; jt * 16 < dout; jt += T / 64) {
            const int j = jt * 16 + il;
            const bool jv = j < dout;
            const float bj = jv ? m.b[l][j] : 0.f;
            f32x4_t acc = f32x4_t{bj, bj, bj, bj};
            const int jc = jv ? j : 0;
            acc = mlp_chain(din4, [&](int k) { return cur[il * LD + k + g]; },                  // A[i = sample il][k]
                            [&](int k) { const float b = wt[(k + g) * LD + jc]; return jv ? b : 0.f; }, acc);   // B[k][j = output]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 4 * g + r;
                if (jv) {
                    float a = acc[r];
                    if (hidden) {
                        a = a > 0.f ? a : 0.f;
                        a = drop_keep(dr, (unsigned int)(s0 + s) * dout + j) ? a * dr.scale : 0.f;
                        if (training && s < ns) m.act[l][(long)(s0 + s) * dout + j] = a;
                    }
                    nxt[s * LD + j] = a;
                }
            }
        }
        for (int i = tid; i < MLPM_S * (dout4 - dout); i += T)          // zero padding of the next layer's k
            nxt[(i / (dout4 - dout)) * LD + dout + i % (dout4 - dout)] = 0.f;
        float* t = cur; cur = nxt; nxt = t;
    }
    __syncthreads();
    const int dl = m.dims[m.nlayers];
    const Div ddl(dl);
    for (int i = tid; i < ns * dl; i += T) {
        const int s = ddl.q(i), j = ddl.r(i);
        const float v = cur[s * LD + j];
        out[(long)(s0 + s) * out_ss + j] = v;
        if (out2) out2[(long)(s0 + s) * dl + j] = v;
    }
}

template <int T>
static __device__ __forceinline__ void mlp_bwd_mfma_body(const m2m_mlp& m, const float* __restrict__ x, int B,
                                                         const float* __restrict__ d_out, long d_out_ss,
                                                         const float* __restrict__ d_out2, int wg, float* sm) {
    constexpr int SU = MlpSU<T>::value;
    constexpr int LD = MLP_MAXW + 1;
    float* g0 = sm;                                   // dz: gradient wrt the current layer's pre-dropout output [16][LD]
    float* g1 = g0 + MLPM_S * LD;                     // gradient wrt its input
    float* ain = g1 + MLPM_S * LD;                    // the layer's input activations [16][LD]
    float* wl = ain + MLPM_S * LD;                    // weights [dout4][LD], rows j >= dout: zero
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    const int s0 = wg * MLPM_S;
    const int ns = min(MLPM_S, B - s0);
    const float scale = 65536.0f / (float)m2m_drop_thr(m.p_drop);
    const int dl = m.dims[m.nlayers];
    const Div ddl(dl);
    for (int i = tid; i < MLPM_S * dl; i += T) {
        const int s = ddl.q(i), j = ddl.r(i);
        float v = 0.f;
        if (s < ns) {
            if (d_out) v = d_out[(long)(s0 + s) * d_out_ss + j];
            if (d_out2) v += d_out2[(long)(s0 + s) * dl + j];
        }
        g0[s * LD + j] = v;
    }
    float* gc = g0;
    float* gn = g1;
    // Everything a layer needs from global memory (its weights, its input activations, its output activations for the ReLU /
    // dropout mask) is independent of the gradient stream: requested in ONE batch per layer, and the next layer's batch is
    // requested before this layer's products, so that its round trip hides behind them.  (Three dependent round trips per layer
    // were most of this kernel's 26 us at the MIMIC cfg batch.)
    constexpr int AU = (MLPM_S * MLP_MAXW + T - 1) / T;     // activation elements per thread (16 samples x up to 128 columns)
    float wv[SU], iv[AU], ov[AU];
    auto issue = [&](int l) {
        const int din = m.dims[l], dout = m.dims[l + 1], dout4 = (dout + 3) & ~3;
        const bool hidden = l < m.nlayers - m.has_out;
        const float* __restrict__ inp = l == 0 ? x : m.act[l - 1];
        const float* __restrict__ w = m.w[l];
        const float* __restrict__ actl = hidden ? m.act[l] : nullptr;
        const Div ddi(din), ddo4(dout4);
#pragma unroll
        for (int u = 0; u < SU; ++u) { const int i = tid + u * T; wv[u] = i < dout * din ? w[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int i = tid + u * T, sI = ddi.q(i);
            iv[u] = (i < MLPM_S * din && sI < ns) ? inp[(long)(s0 + sI) * din + ddi.r(i)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int i = tid + u * T, sI = ddo4.q(i), j = ddo4.r(i);
            ov[u] = (hidden && i < MLPM_S * dout4 && sI < ns && j < dout) ? actl[(long)(s0 + sI) * dout + j] : 1.f;
        }
    };
    issue(m.nlayers - 1);
    for (int l = m.nlayers - 1; l >= 0; --l) {
        const int din = m.dims[l], dout = m.dims[l + 1], dout4 = (dout + 3) & ~3;
        const bool hidden = l < m.nlayers - m.has_out;
        __syncthreads();                              // gc complete; the previous layer is done with wl / ain
        const Div ddi(din), ddo4(dout4);
        {
            const float* __restrict__ w = m.w[l];
#pragma unroll
            for (int u = 0; u < SU; ++u) { const int i = tid + u * T; if (i < dout4 * din) wl[ddi.q(i) * LD + ddi.r(i)] = wv[u]; }
            for (int i0 = tid + SU * T; i0 < dout4 * din; i0 += SU * T) {     // (layers wider than 64 x 64)
                float v[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) { const int i = i0 + u * T; v[u] = i < dout * din ? w[i] : 0.f; }
#pragma unroll
                for (int u = 0; u < SU; ++u) { const int i = i0 + u * T; if (i < dout4 * din) wl[ddi.q(i) * LD + ddi.r(i)] = v[u]; }
            }
#pragma unroll
            for (int u = 0; u < AU; ++u) { const int i = tid + u * T; if (i < MLPM_S * din) ain[ddi.q(i) * LD + ddi.r(i)] = iv[u]; }
        }
        // through Dropout and ReLU; columns j in [dout, dout4): zero (k padding of d_in)
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int i = tid + u * T;
            if (i < MLPM_S * dout4) {
                const int j = ddo4.r(i);
                float* gp = gc + ddo4.q(i) * LD + j;
                if (j >= dout) *gp = 0.f;
                else if (hidden) *gp = ov[u] != 0.f ? *gp * scale : 0.f;
            }
        }
        if (l > 0) issue(l - 1);
        __syncthreads();
        // dW[j][k] += sum_s dz[s][j] in[s][k]: tiles (jt, kt) of 16 x 16, the 16 samples are the contraction
        const int njt = (dout + 15) >> 4, nkt = (din + 15) >> 4;
        for (int t = wave; t < njt * nkt; t += T / 64) {
            const int jt = t / nkt, kt = t % nkt;
            const int ja = min(jt * 16 + il, dout - 1), kb = min(kt * 16 + il, din - 1);
            f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < MLPM_S; s4 += 4) acc = mlp_mfma4(gc[(s4 + g) * LD + ja], ain[(s4 + g) * LD + kb], acc);
            const int k = kt * 16 + il;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + 4 * g + r;
                if (j < dout && k < din) atomicAdd(m.g_w[l] + j * din + k, acc[r]);
            }
        }
        for (int j = tid; j < dout; j += T) {
            float a = 0.f;
            for (int s = 0; s < MLPM_S; ++s) a += gc[s * LD + j];
            atomicAdd(m.g_b[l] + j, a);
        }
        if (l > 0) {                                  // d_in[s][k] = sum_j dz[s][j] W[j][k]
            for (int kt = wave; kt * 16 < din; kt += T / 64) {
                const int k = kt * 16 + il;
                const bool kv = k < din;
                f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const int kc = kv ? k : 0;
                acc = mlp_chain(dout4, [&](int j) { return gc[il * LD + j + g]; },              // A[i = sample il][j]
                                [&](int j) { const float b = wl[(j + g) * LD + kc]; return kv ? b : 0.f; }, acc);   // B[j][k]
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kv) gn[(4 * g + r) * LD + k] = acc[r];
            }
        }
        float* t = gc; gc = gn; gn = t;
    }
}

