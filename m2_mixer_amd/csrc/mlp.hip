// Plain MLP tower: num_blocks x (Linear -> ReLU -> Dropout) + optional output Linear  (reference: modules/mlp.py:4-27;
// the MIMIC `static` modality, models/mimic.py:98: 5 -> 64 -> 64 -> 64, ~8.5 kMAC per sample).
//
// The work is tiny and the widths are <= 128, so both passes are exact-fp32 VALU kernels: a workgroup owns
// MLP_S samples, keeps the activations of the current layer in LDS and the layer's weights (transposed, padded)
// beside them.  Backward needs no mask regeneration: the saved layer output is relu(z) * keep * scale, which is
// non-zero exactly where the gradient passes, so dz = d_out * scale * [out != 0].
#include "mlp_body.h"

template <int MLP_S, int MLP_T>
__global__ __launch_bounds__(MLP_T) void mlp_fwd_kernel(const m2m_mlp m, const float* __restrict__ x, int B, float* __restrict__ out,
                                                        long out_ss, float* __restrict__ out2, int training, unsigned int seed,
                                                        unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    constexpr int MLP_U = MLP_S * 64 / MLP_T;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a0 = sm;                                   // [MLP_S][MLP_MAXW + 1]
    float* a1 = a0 + MLP_S * (MLP_MAXW + 1);
    float* wt = a1 + MLP_S * (MLP_MAXW + 1);          // [din][dout + 1]  (transposed weights of the current layer)
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * MLP_S;
    const int ns = min(MLP_S, B - s0);
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    constexpr int LD = MLP_MAXW + 1;

    for (int i = tid; i < MLP_S * m.dims[0]; i += MLP_T) {
        const int s = i / m.dims[0], k = i % m.dims[0];
        a0[s * LD + k] = s < ns ? x[(long)(s0 + s) * m.dims[0] + k] : 0.f;
    }
    float* cur = a0;
    float* nxt = a1;
    for (int l = 0; l < m.nlayers; ++l) {
        const int din = m.dims[l], dout = m.dims[l + 1];
        const bool hidden = l < m.nlayers - m.has_out;
        const Drop dr = mlp_drop(m, l, training && hidden, seed, step);
        __syncthreads();
        {
            const Div dv(din);
            const float* __restrict__ w = m.w[l];
            for (int i0 = tid; i0 < din * dout; i0 += MLP_SU * MLP_T) {
                float v[MLP_SU];
#pragma unroll
                for (int u = 0; u < MLP_SU; ++u) { const int i = i0 + u * MLP_T; v[u] = i < din * dout ? w[i] : 0.f; }
#pragma unroll
                for (int u = 0; u < MLP_SU; ++u) { const int i = i0 + u * MLP_T; if (i < din * dout) wt[dv.r(i) * (dout + 1) + dv.q(i)] = v[u]; }
            }
        }
        __syncthreads();
        // MLP_U outputs per thread at a time: one dependent fma chain per output left every LDS read latency exposed
        // (~100 cycles per step of a 64-step chain, 24 chains per thread: most of this kernel's 50 us)
        const Div dvo(dout);
        for (int i0 = tid; i0 < MLP_S * dout; i0 += MLP_U * MLP_T) {
            int so[MLP_U], jo[MLP_U];
            float acc[MLP_U];
#pragma unroll
            for (int u = 0; u < MLP_U; ++u) {
                const int i = min(i0 + u * MLP_T, MLP_S * dout - 1);
                so[u] = dvo.q(i) * LD; jo[u] = dvo.r(i);
                acc[u] = m.b[l][jo[u]];
            }
#pragma unroll 8
            for (int k = 0; k < din; ++k) {
#pragma unroll
                for (int u = 0; u < MLP_U; ++u) acc[u] = __builtin_fmaf(cur[so[u] + k], wt[k * (dout + 1) + jo[u]], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < MLP_U; ++u) {
                const int i = i0 + u * MLP_T;
                if (i < MLP_S * dout) {
                    const int s = so[u] / LD, j = jo[u];
                    float a = acc[u];
                    if (hidden) {
                        a = a > 0.f ? a : 0.f;
                        a = drop_keep(dr, (unsigned int)(s0 + s) * dout + j) ? a * dr.scale : 0.f;
                        if (training && s < ns) m.act[l][(long)(s0 + s) * dout + j] = a;
                    }
                    nxt[so[u] + j] = a;
                }
            }
        }
        float* t = cur; cur = nxt; nxt = t;
    }
    __syncthreads();
    const int dl = m.dims[m.nlayers];
    for (int i = tid; i < ns * dl; i += MLP_T) {
        const int s = i / dl, j = i % dl;
        const float v = cur[s * LD + j];
        out[(long)(s0 + s) * out_ss + j] = v;
        if (out2) out2[(long)(s0 + s) * dl + j] = v;
    }
}

template <int MLP_S, int MLP_T>
__global__ __launch_bounds__(MLP_T) void mlp_bwd_kernel(const m2m_mlp m, const float* __restrict__ x, int B,
                                                        const float* __restrict__ d_out, long d_out_ss,
                                                        const float* __restrict__ d_out2) {
    constexpr int MLP_U = MLP_S * 64 / MLP_T;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int LD = MLP_MAXW + 1;
    float* g0 = sm;                                   // gradient wrt the current layer's output [MLP_S][LD]
    float* g1 = g0 + MLP_S * LD;                      // gradient wrt its input
    float* ain = g1 + MLP_S * LD;                     // the layer's input activations [MLP_S][LD]
    float* wl = ain + MLP_S * LD;                     // weights [dout][din + 1]
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * MLP_S;
    const int ns = min(MLP_S, B - s0);
    const float scale = 65536.0f / (float)m2m_drop_thr(m.p_drop);

    const int dl = m.dims[m.nlayers];
    for (int i = tid; i < MLP_S * dl; i += MLP_T) {
        const int s = i / dl, j = i % dl;
        float v = 0.f;
        if (s < ns) {
            if (d_out) v = d_out[(long)(s0 + s) * d_out_ss + j];
            if (d_out2) v += d_out2[(long)(s0 + s) * dl + j];
        }
        g0[s * LD + j] = v;
    }
    float* gc = g0;
    float* gn = g1;
    for (int l = m.nlayers - 1; l >= 0; --l) {
        const int din = m.dims[l], dout = m.dims[l + 1];
        const bool hidden = l < m.nlayers - m.has_out;
        const float* inp = l == 0 ? x : m.act[l - 1];
        __syncthreads();
        const Div dvi(din), dvo(dout);
        {
            const float* __restrict__ w = m.w[l];
            for (int i0 = tid; i0 < dout * din; i0 += MLP_SU * MLP_T) {
                float v[MLP_SU];
#pragma unroll
                for (int u = 0; u < MLP_SU; ++u) { const int i = i0 + u * MLP_T; v[u] = i < dout * din ? w[i] : 0.f; }
#pragma unroll
                for (int u = 0; u < MLP_SU; ++u) { const int i = i0 + u * MLP_T; if (i < dout * din) wl[dvi.q(i) * (din + 1) + dvi.r(i)] = v[u]; }
            }
            for (int i0 = tid; i0 < MLP_S * din; i0 += MLP_SU * MLP_T) {
                float v[MLP_SU];
#pragma unroll
                for (int u = 0; u < MLP_SU; ++u) {
                    const int i = i0 + u * MLP_T, sI = dvi.q(i);
                    v[u] = (i < MLP_S * din && sI < ns) ? inp[(long)(s0 + sI) * din + dvi.r(i)] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < MLP_SU; ++u) { const int i = i0 + u * MLP_T; if (i < MLP_S * din) ain[dvi.q(i) * LD + dvi.r(i)] = v[u]; }
            }
        }
        if (hidden) {                                  // through Dropout and ReLU
            const float* __restrict__ actl = m.act[l];
            for (int i0 = tid; i0 < MLP_S * dout; i0 += MLP_SU * MLP_T) {
                float o[MLP_SU];
#pragma unroll
                for (int u = 0; u < MLP_SU; ++u) {
                    const int i = i0 + u * MLP_T, sI = dvo.q(i);
                    o[u] = (i < MLP_S * dout && sI < ns) ? actl[(long)(s0 + sI) * dout + dvo.r(i)] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < MLP_SU; ++u) {
                    const int i = i0 + u * MLP_T;
                    if (i < MLP_S * dout) { float* gp = gc + dvo.q(i) * LD + dvo.r(i); *gp = o[u] != 0.f ? *gp * scale : 0.f; }
                }
            }
        }
        __syncthreads();
        for (int i0 = tid; i0 < dout * din; i0 += MLP_U * MLP_T) {      // dW[j][k] += sum_s dz[s][j] in[s][k]
            int jo[MLP_U], ko[MLP_U];
            float a[MLP_U];
#pragma unroll
            for (int u = 0; u < MLP_U; ++u) {
                const int i = min(i0 + u * MLP_T, dout * din - 1);
                jo[u] = dvi.q(i); ko[u] = dvi.r(i); a[u] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < MLP_S; ++s) {
#pragma unroll
                for (int u = 0; u < MLP_U; ++u) a[u] = __builtin_fmaf(gc[s * LD + jo[u]], ain[s * LD + ko[u]], a[u]);
            }
#pragma unroll
            for (int u = 0; u < MLP_U; ++u) { const int i = i0 + u * MLP_T; if (i < dout * din) atomicAdd(m.g_w[l] + i, a[u]); }
        }
        for (int j = tid; j < dout; j += MLP_T) {
            float a = 0.f;
            for (int s = 0; s < MLP_S; ++s) a += gc[s * LD + j];
            atomicAdd(m.g_b[l] + j, a);
        }
        if (l > 0) {
            for (int i0 = tid; i0 < MLP_S * din; i0 += MLP_U * MLP_T) {  // d_in[s][k] = sum_j dz[s][j] W[j][k]
                int so[MLP_U], ko[MLP_U];
                float a[MLP_U];
#pragma unroll
                for (int u = 0; u < MLP_U; ++u) {
                    const int i = min(i0 + u * MLP_T, MLP_S * din - 1);
                    so[u] = dvi.q(i) * LD; ko[u] = dvi.r(i); a[u] = 0.f;
                }
#pragma unroll 8
                for (int j = 0; j < dout; ++j) {
#pragma unroll
                    for (int u = 0; u < MLP_U; ++u) a[u] = __builtin_fmaf(gc[so[u] + j], wl[j * (din + 1) + ko[u]], a[u]);
                }
#pragma unroll
                for (int u = 0; u < MLP_U; ++u) { const int i = i0 + u * MLP_T; if (i < MLP_S * din) gn[so[u] + ko[u]] = a[u]; }
            }
        }
        float* t = gc; gc = gn; gn = t;
    }
}

// ---- small batches: the MFMA bodies of mlp_body.h as launches of their own ----
__global__ __launch_bounds__(MLPM_T) void mlp_fwd_mfma_kernel(const m2m_mlp m, const float* __restrict__ x, int B, float* __restrict__ out,
                                                              long out_ss, float* __restrict__ out2, int training, unsigned int seed,
                                                              unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    mlp_fwd_mfma_body<MLPM_T>(m, x, B, out, out_ss, out2, training, seed, step_host, step_dev, blockIdx.x, sm);
}
__global__ __launch_bounds__(MLPM_T) void mlp_bwd_mfma_kernel(const m2m_mlp m, const float* __restrict__ x, int B,
                                                              const float* __restrict__ d_out, long d_out_ss,
                                                              const float* __restrict__ d_out2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    mlp_bwd_mfma_body<MLPM_T>(m, x, B, d_out, d_out_ss, d_out2, blockIdx.x, sm);
}

static int check_mlp(const m2m_mlp* m, int B) {
    if (!m || B < 1) { m2m_set_error("mlp: bad argument", __FILE__, __LINE__); return -1; }
    if (m->nlayers < 1 || m->nlayers > M2M_MLP_MAX_LAYERS) { m2m_set_error("mlp: nlayers out of range", __FILE__, __LINE__); return -1; }
    for (int i = 0; i <= m->nlayers; ++i)
        if (m->dims[i] < 1 || m->dims[i] > MLP_MAXW) { m2m_set_error("mlp: layer widths must be in [1, 128]", __FILE__, __LINE__); return -1; }
    return 0;
}

template <int S, int T>
static int launch_mlp_fwd(const m2m_mlp* m, const float* x, int B, float* out, long out_ss, float* out_dense, int training,
                          unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)2 * S * (MLP_MAXW + 1) + (size_t)MLP_MAXW * (MLP_MAXW + 1));
    auto kern = mlp_fwd_kernel<S, T>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((B + S - 1) / S), dim3(T), lds, st, *m, x, B, out, out_ss, out_dense, training, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int S, int T>
static int launch_mlp_bwd(const m2m_mlp* m, const float* x, int B, const float* d_out, long d_out_ss, const float* d_out_dense,
                          hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)3 * S * (MLP_MAXW + 1) + (size_t)MLP_MAXW * (MLP_MAXW + 1));
    auto kern = mlp_bwd_kernel<S, T>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((B + S - 1) / S), dim3(T), lds, st, *m, x, B, d_out, d_out_ss, d_out_dense);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
#define MLP_SMALL_BATCH 2048      // up to here the MFMA kernels (one 16-sample tile per workgroup)

extern "C" int m2m_mlp_forward(const m2m_mlp* m, const float* x, int B, float* out, int64_t out_sample_stride, float* out_dense,
                               int training, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (int rc = check_mlp(m, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (B <= MLP_SMALL_BATCH) {
        const size_t lds = sizeof(float) * ((size_t)2 * MLPM_S + MLP_MAXW) * (MLP_MAXW + 1);
        static bool done = false;
        if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
        hipLaunchKernelGGL(mlp_fwd_mfma_kernel, dim3((B + MLPM_S - 1) / MLPM_S), dim3(MLPM_T), lds, st, *m, x, B, out, (long)out_sample_stride,
                           out_dense, training, seed, step, step_dev);
        M2M_CHECK_HIP(hipGetLastError());
        return 0;
    }
    return launch_mlp_fwd<32, 256>(m, x, B, out, (long)out_sample_stride, out_dense, training, seed, step, step_dev, st);
}

extern "C" int m2m_mlp_backward(const m2m_mlp* m, const float* x, int B, const float* d_out, int64_t d_out_sample_stride,
                                const float* d_out_dense, void* stream) {
    if (int rc = check_mlp(m, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (B <= MLP_SMALL_BATCH) {
        const size_t lds = sizeof(float) * ((size_t)3 * MLPM_S + MLP_MAXW) * (MLP_MAXW + 1);
        static bool done = false;
        if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
        hipLaunchKernelGGL(mlp_bwd_mfma_kernel, dim3((B + MLPM_S - 1) / MLPM_S), dim3(MLPM_T), lds, st, *m, x, B, d_out, (long)d_out_sample_stride, d_out_dense);
        M2M_CHECK_HIP(hipGetLastError());
        return 0;
    }
    return launch_mlp_bwd<32, 256>(m, x, B, d_out, (long)d_out_sample_stride, d_out_dense, st);
}
