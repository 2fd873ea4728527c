// Classification heads + multi-head cross-entropy, forward and backward in one launch.
//
// Reference: models/avmnist.py:271-298 -- classifier_image/audio = nn.Linear on tokens.mean(dim=1),
// classifier_fusion = StandardClassifier (mean over tokens -> Linear, modules/classification.py:89-90),
// three nn.CrossEntropyLoss() (mean), loss = (w Lf + ow Li + ow La) * 3, preds = softmax(.).argmax(1).
// The token means ("pooled") are produced by the tower forward kernel and consumed here -- or, for towers whose forward does not
// own whole samples (the wide path), computed here from the tower output (m2m_head.tokens).
// BCE = true is the MM-IMDb variant (models/mmimdb.py:47-50, :115-133): nn.BCEWithLogitsLoss(pos_weight) with mean
// reduction over all B*K elements per head, preds = sigmoid(logits) > 0.5 per label.
#include "tile.h"

#ifndef HEAD_S
#define HEAD_S 16       // samples per workgroup (small: the launch sits between forward and backward on the critical path)
#endif
#define HEAD_MAXK 32
#define HEAD_MAXH 4

struct HeadArgs {
    m2m_head h[HEAD_MAXH];
};

// S: samples per workgroup (HEAD_S, or 4 at small batch: at the MM-IMDb cfg batch 32 samples x 3 heads were 6 workgroups).
// DD: hidden_dim (compile time: the dot-product loops unroll and their LDS reads are requested in batches).  Round 4: the
// kernel was a chain of ~300 dependent LDS round trips (runtime loop bounds: nothing unrolled, every ds_read followed by
// s_waitcnt lgkmcnt(0)) -- 13.3 us for 4 MFLOP on the critical path between forward and backward.  Now: 16-byte LDS accesses on
// rows padded to D + 4, four lanes x eight float4 per logit, the softmax on 32 lanes per sample (cross-lane max / sum), the
// gradient phases with the sample / class loops unrolled.  Summation orders changed (logits: four interleaved partial sums of
// float4 chunks; loss terms per workgroup in sample order as before): parity with the oracle is unaffected (fp32, ~1e-7).
template <bool BCE, int S, int DD>
__global__ __launch_bounds__(NTHREADS) void heads_kernel(const HeadArgs ha, const void* __restrict__ labels_v,
                                                         const float* __restrict__ pos_weight, int B, int D_rt,
                                                         int K, float* __restrict__ logits_out, float* __restrict__ losses,
                                                         int32_t* __restrict__ preds, int nheads) {
    const int64_t* labels = reinterpret_cast<const int64_t*>(labels_v);       // CE: class index (B)
    const float* targets = reinterpret_cast<const float*>(labels_v);          // BCE: multi-hot (B, K)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = DD, DL = DD + 4, D4 = DD / 4;
    float* pl = reinterpret_cast<float*>(smem);          // pooled tile [S][DL]
    float* wl = pl + S * DL;                              // weights     [HEAD_MAXK][DL]
    float* lg = wl + HEAD_MAXK * DL;                      // logits / dlogits [S][HEAD_MAXK]
    float* red = lg + S * HEAD_MAXK;                      // [S] loss terms
    float* tm = red + S;                                  // [S][HEAD_MAXK] per-element loss terms (BCE)

    const int tid = threadIdx.x;
    const int hI = blockIdx.y;
    const m2m_head& hd = ha.h[hI];
    const int s0 = blockIdx.x * S;
    const int ns = min(S, B - s0);

    // ---- pooled rows and head weights -> LDS (16-byte accesses; every global load requested before the first LDS write) ----
    {
        constexpr int NP = (S * D4 + NTHREADS - 1) / NTHREADS, NW = (HEAD_MAXK * D4 + NTHREADS - 1) / NTHREADS;
        f32x4_t pv[NP], wv[NW];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + i * NTHREADS, sI = idx / D4, c = (idx % D4) * 4;
            pv[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (idx < S * D4 && sI < ns) {
                if (hd.tokens) {
                    // the head pools the tower output itself: x.mean(dim=1), tokens in order, eight loads in flight per round trip
                    // (the order and the 1 / N product of the towers' token-mean launch this replaces: same bits)
                    const float* col = hd.tokens + (long)(s0 + sI) * hd.tok_sample_stride + c;
                    const int N = hd.ntok;
                    f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    for (int n0 = 0; n0 < N; n0 += 8) {
                        f32x4_t t8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) t8[j] = *reinterpret_cast<const f32x4_t*>(col + (long)min(n0 + j, N - 1) * D);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (n0 + j < N) a += t8[j];
                    }
                    pv[i] = a * (1.0f / (float)N);
                } else pv[i] = *reinterpret_cast<const f32x4_t*>(hd.pooled + (long)(s0 + sI) * D + c);
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = tid + i * NTHREADS;
            wv[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (idx < K * D4) wv[i] = *reinterpret_cast<const f32x4_t*>(hd.w + (long)idx * 4);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + i * NTHREADS;
            if (idx < S * D4) *reinterpret_cast<f32x4_t*>(pl + (idx / D4) * DL + (idx % D4) * 4) = pv[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = tid + i * NTHREADS;
            if (idx < HEAD_MAXK * D4) *reinterpret_cast<f32x4_t*>(wl + (idx / D4) * DL + (idx % D4) * 4) = wv[i];     // rows >= K: zeros
        }
    }
    __syncthreads();
    // ---- logits: 4 adjacent lanes split each D-long dot product, float4 chunks interleaved over the lanes ----
    for (int idx = tid >> 2; idx < S * K; idx += NTHREADS / 4) {
        const int sI = idx / K, k = idx % K, part = tid & 3;
        constexpr int NC = D4 / 4;                        // float4 chunks per lane
        f32x4_t a4[NC], b4[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            a4[c] = *reinterpret_cast<const f32x4_t*>(pl + sI * DL + 4 * (part + 4 * c));
            b4[c] = *reinterpret_cast<const f32x4_t*>(wl + k * DL + 4 * (part + 4 * c));
        }
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) a = __builtin_fmaf(a4[c][e], b4[c][e], a);
        a = wave_sum_xor(a, 4) + hd.b[k];
        if (part == 0) {
            lg[sI * HEAD_MAXK + k] = a;
            if (sI < ns) logits_out[((long)hI * B + s0 + sI) * K + k] = a;
        }
    }
    __syncthreads();
    // ---- loss / prediction / dlogits ----
    if (BCE) {
        // every (sample, label) element is independent -- one thread each
        const float scale = hd.weight / ((float)B * (float)K);
        for (int idx = tid; idx < S * K; idx += NTHREADS) {
            const int sI = idx / K, k = idx % K;
            float term = 0.f, dl = 0.f;
            if (sI < ns) {
                const float x = lg[sI * HEAD_MAXK + k], y = targets[(long)(s0 + sI) * K + k], pw = pos_weight[k];
                // log sigmoid(x) = min(x, 0) - log1p(exp(-|x|));  log(1 - sigmoid(x)) = log sigmoid(x) - x
                const float ls = __builtin_fminf(x, 0.f) - log1pf(__expf(-__builtin_fabsf(x)));
                term = -(pw * y * ls + (1.f - y) * (ls - x));
                const float sg = 1.0f / (1.0f + __expf(-x));
                dl = scale * ((1.f - y) * sg - pw * y * (1.f - sg));
                preds[((long)hI * B + s0 + sI) * K + k] = x > 0.f ? 1 : 0;
            }
            lg[sI * HEAD_MAXK + k] = dl;
            tm[sI * HEAD_MAXK + k] = term;
        }
        __syncthreads();
        if (tid < S) {
            float term = 0.f;
            for (int k = 0; k < K; ++k) term += tm[tid * HEAD_MAXK + k];      // label order, as the single-thread loop summed
            red[tid] = term / (float)K;
        }
    } else if (tid < S * 32) {
        // cross-entropy: 32 lanes per sample, lane k holds logit k (K <= HEAD_MAXK = 32); max / argmax / sum across the lanes.
        // (first index of the maximum, like torch.argmax on distinct values; the sums run over a butterfly instead of k = 0..K-1:
        // last-bit differences in a printed loss, none in the predictions)
        const int sI = tid >> 5, k = tid & 31;
        const bool live = k < K;
        const float v = live ? lg[sI * HEAD_MAXK + k] : -3.0e38f;
        float mx = v;
        int am = live ? k : 0x7fffffff;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(mx, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            if (ov > mx || (ov == mx && oa < am)) { mx = ov; am = oa; }
        }
        const float ex = live ? __expf(v - mx) : 0.f;
        float se = ex;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) se += __shfl_xor(se, o, 64);
        const int y = sI < ns ? (int)labels[s0 + sI] : 0;
        const float scale = hd.weight / (float)B;
        const float dl = sI < ns ? scale * (ex * (1.0f / se) - (k == y ? 1.f : 0.f)) : 0.f;
        const float vy = __shfl(v, (tid & 32) + y, 64);       // the label's logit (lane y of this sample's 32 lanes)
        if (live) lg[sI * HEAD_MAXK + k] = dl;
        if (k == 0) {
            red[sI] = sI < ns ? (__logf(se) + mx) - vy : 0.f;
            if (sI < ns) preds[(long)hI * B + s0 + sI] = am;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int sI = 0; sI < S; ++sI) t += red[sI];
        t /= (float)B;
        atomicAdd(losses + hI, t);
        atomicAdd(losses + nheads, t * hd.weight);
    }
    if (hd.d_pooled) {
        // d_pooled[s][d..d+3] = sum_k dl[s][k] w[k][d..d+3]: thread = (sample, float4 of d)
        for (int idx = tid; idx < ns * D4; idx += NTHREADS) {
            const int sI = idx / D4, c = (idx % D4) * 4;
            f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 4 <= K; k += 4) {
                f32x4_t w4[4];
                float d4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { w4[j] = *reinterpret_cast<const f32x4_t*>(wl + (k + j) * DL + c); d4[j] = lg[sI * HEAD_MAXK + k + j]; }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(d4[j], w4[j][e], a[e]);
            }
            for (; k < K; ++k) {
                const f32x4_t w4 = *reinterpret_cast<const f32x4_t*>(wl + k * DL + c);
                const float dk = lg[sI * HEAD_MAXK + k];
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(dk, w4[e], a[e]);
            }
            *reinterpret_cast<f32x4_t*>(hd.d_pooled + (long)(s0 + sI) * D + c) = a;
        }
        // g_part: this workgroup's sums go to its slot (plain stores; m2m_towers_wgrad_heads adds the slots in a fixed order)
        // g_w[k][d..d+3] = sum_s dl[s][k] pooled[s][d..d+3]: thread = (class, float4 of d), the sample loop unrolled
        float* slot = hd.g_part ? hd.g_part + (long)blockIdx.x * M2M_SPLIT_GPART : nullptr;
        for (int idx = tid; idx < K * D4; idx += NTHREADS) {
            const int k = idx / D4, c = (idx % D4) * 4;
            f32x4_t p4[S];
            float d1[S];
#pragma unroll
            for (int sI = 0; sI < S; ++sI) { p4[sI] = *reinterpret_cast<const f32x4_t*>(pl + sI * DL + c); d1[sI] = lg[sI * HEAD_MAXK + k]; }
            f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sI = 0; sI < S; ++sI)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(d1[sI], p4[sI][e], a[e]);
            if (slot) *reinterpret_cast<f32x4_t*>(slot + (long)k * D + c) = a;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(hd.g_w + (long)k * D + c + e, a[e]);
            }
        }
        if (tid < K) {
            float a = 0.f;
#pragma unroll
            for (int sI = 0; sI < S; ++sI) a += lg[sI * HEAD_MAXK + tid];
            if (slot) slot[K * D + tid] = a;
            else atomicAdd(hd.g_b + tid, a);
        }
        if (slot && tid < 2) slot[K * D + K + tid] = 0.f;           // (the slot layout's loss entries: the losses stay atomics)
    }
}

// losses are accumulated with atomics; they are cleared by a kernel (not a memset node: a 16-byte
// hipMemsetAsync node was observed to write stale bytes on hipGraph replay).
__global__ void zero_floats_kernel(float* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f;
}

template <bool BCE, int S, int DD>
static int launch_heads_sd(const HeadArgs& ha, int nheads, const void* labels, const float* pos_weight, int B, int K,
                           float* logits, float* losses, int32_t* preds, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)S * (DD + 4) + (size_t)HEAD_MAXK * (DD + 4) + 2 * S * HEAD_MAXK + S);
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(heads_kernel<BCE, S, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
    hipLaunchKernelGGL((heads_kernel<BCE, S, DD>), dim3((B + S - 1) / S, nheads), dim3(NTHREADS), lds, st, ha, labels, pos_weight, B, DD, K, logits, losses, preds, nheads);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <bool BCE, int S>
static int launch_heads_s(const HeadArgs& ha, int nheads, const void* labels, const float* pos_weight, int B, int D, int K,
                          float* logits, float* losses, int32_t* preds, hipStream_t st) {
    switch (D) {
        case 32:  return launch_heads_sd<BCE, S, 32>(ha, nheads, labels, pos_weight, B, K, logits, losses, preds, st);
        case 64:  return launch_heads_sd<BCE, S, 64>(ha, nheads, labels, pos_weight, B, K, logits, losses, preds, st);
        case 128: return launch_heads_sd<BCE, S, 128>(ha, nheads, labels, pos_weight, B, K, logits, losses, preds, st);
        case 256: return launch_heads_sd<BCE, S, 256>(ha, nheads, labels, pos_weight, B, K, logits, losses, preds, st);
    }
    m2m_set_error("heads: hidden_dim must be 32, 64, 128 or 256", __FILE__, __LINE__);
    return -1;
}

static int heads_samples_per_wg(int B) { return B <= 64 ? 4 : HEAD_S; }
template <bool BCE>
static int launch_heads(const m2m_head* heads, int nheads, const void* labels, const float* pos_weight, int B, int D, int K,
                        float* logits, float* losses, int32_t* preds, int zero_losses, void* stream) {
    if (!heads || nheads < 1 || nheads > HEAD_MAXH || K < 2 || K > HEAD_MAXK || D < 1 || D > 256 || B < 1) {
        m2m_set_error("heads: unsupported (nheads<=4, K<=32, D<=256)", __FILE__, __LINE__);
        return -1;
    }
    HeadArgs ha;
    for (int i = 0; i < nheads; ++i) {
        if (heads[i].tokens ? (heads[i].ntok < 1 || heads[i].tok_sample_stride < (int64_t)heads[i].ntok * D || (heads[i].tok_sample_stride & 3) ||
                               (reinterpret_cast<uintptr_t>(heads[i].tokens) & 15))
                            : !heads[i].pooled) {
            m2m_set_error("heads: a head needs pooled, or 16-byte aligned tokens with ntok >= 1 and a sample stride >= ntok * D (multiple of 4)", __FILE__, __LINE__);
            return -1;
        }
        if (heads[i].g_part && (BCE || (long)K * D + K + 2 > M2M_SPLIT_GPART)) {
            m2m_set_error("heads: g_part needs cross-entropy heads with K*D + K + 2 <= M2M_SPLIT_GPART", __FILE__, __LINE__);
            return -1;
        }
    }
    for (int i = 0; i < nheads; ++i) ha.h[i] = heads[i];
    for (int i = nheads; i < HEAD_MAXH; ++i) ha.h[i] = heads[0];
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (zero_losses) hipLaunchKernelGGL(zero_floats_kernel, dim3(1), dim3(64), 0, st, losses, nheads + 1);
    if (heads_samples_per_wg(B) == 4) return launch_heads_s<BCE, 4>(ha, nheads, labels, pos_weight, B, D, K, logits, losses, preds, st);
    return launch_heads_s<BCE, HEAD_S>(ha, nheads, labels, pos_weight, B, D, K, logits, losses, preds, st);
}

extern "C" int m2m_heads_part_tiles(int B) { const int S = heads_samples_per_wg(B); return B < 1 ? 0 : (B + S - 1) / S; }

extern "C" int m2m_heads_ce(const m2m_head* heads, int nheads, const int64_t* labels, int B, int D, int K, float* logits,
                            float* losses, int32_t* preds, int zero_losses, void* stream) {
    return launch_heads<false>(heads, nheads, labels, nullptr, B, D, K, logits, losses, preds, zero_losses, stream);
}

extern "C" int m2m_heads_bce(const m2m_head* heads, int nheads, const float* targets, const float* pos_weight, int B, int D, int K,
                             float* logits, float* losses, int32_t* preds, int zero_losses, void* stream) {
    if (!targets || !pos_weight) { m2m_set_error("heads_bce: targets and pos_weight are required", __FILE__, __LINE__); return -1; }
    return launch_heads<true>(heads, nheads, targets, pos_weight, B, D, K, logits, losses, preds, zero_losses, stream);
}
