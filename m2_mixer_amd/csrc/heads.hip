// Classification heads + multi-head cross-entropy, forward and backward in one launch.
//
// Reference: models/avmnist.py:271-298 -- classifier_image/audio = nn.Linear on tokens.mean(dim=1),
// classifier_fusion = StandardClassifier (mean over tokens -> Linear, modules/classification.py:89-90),
// three nn.CrossEntropyLoss() (mean), loss = (w Lf + ow Li + ow La) * 3, preds = softmax(.).argmax(1).
// The token means ("pooled") are produced by the tower forward kernel; this kernel consumes them.
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

// S: samples per workgroup (HEAD_S, or 4 at small batch: the kernel is a chain of dependent LDS loops whose lengths go with S,
// and at the MM-IMDb cfg batch 32 samples x 3 heads were 6 workgroups)
template <bool BCE, int S>
__global__ __launch_bounds__(NTHREADS) void heads_kernel(const HeadArgs ha, const void* __restrict__ labels_v,
                                                         const float* __restrict__ pos_weight, int B, int D,
                                                         int K, float* __restrict__ logits_out, float* __restrict__ losses,
                                                         int32_t* __restrict__ preds, int nheads) {
    const int64_t* labels = reinterpret_cast<const int64_t*>(labels_v);       // CE: class index (B)
    const float* targets = reinterpret_cast<const float*>(labels_v);          // BCE: multi-hot (B, K)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int DL = D + 1;
    float* pl = reinterpret_cast<float*>(smem);          // pooled tile [S][DL]
    float* wl = pl + S * DL;                         // weights     [K][DL]
    float* lg = wl + HEAD_MAXK * DL;                      // logits / dlogits [S][HEAD_MAXK]
    float* red = lg + S * HEAD_MAXK;                 // [S] loss terms
    float* tm = red + S;                             // [S][HEAD_MAXK] per-element loss terms (BCE)

    const int tid = threadIdx.x;
    const int hI = blockIdx.y;
    const m2m_head& hd = ha.h[hI];
    const int s0 = blockIdx.x * S;
    const int ns = min(S, B - s0);

    for (int idx = tid; idx < S * D; idx += NTHREADS) {
        const int s = idx / D, d = idx % D;
        pl[s * DL + d] = s < ns ? hd.pooled[(long)(s0 + s) * D + d] : 0.f;
    }
    for (int idx = tid; idx < K * D; idx += NTHREADS) wl[(idx / D) * DL + idx % D] = hd.w[idx];
    __syncthreads();
    // logits: 4 adjacent lanes split each D-long dot product
    for (int idx = tid >> 2; idx < S * K; idx += NTHREADS / 4) {
        const int s = idx / K, k = idx % K, part = tid & 3;
        float a = 0.f;
        for (int d = part; d < D; d += 4) a = __builtin_fmaf(pl[s * DL + d], wl[k * DL + d], a);
        a = wave_sum_xor(a, 4) + hd.b[k];
        if (part == 0) {
            lg[s * HEAD_MAXK + k] = a;
            if (s < ns) logits_out[((long)hI * B + s0 + s) * K + k] = a;
        }
    }
    __syncthreads();
    // loss / prediction / dlogits.  BCE: every (sample, label) element is independent -- one thread each (the transcendental
    // functions of K = 23 labels in one thread's loop made this the longest phase of the MM-IMDb heads); CE: softmax needs
    // the row, one thread per sample.
    if (BCE) {
        const float scale = hd.weight / ((float)B * (float)K);
        for (int idx = tid; idx < S * K; idx += NTHREADS) {
            const int s = idx / K, k = idx % K;
            float term = 0.f, dl = 0.f;
            if (s < ns) {
                const float x = lg[s * HEAD_MAXK + k], y = targets[(long)(s0 + s) * K + k], pw = pos_weight[k];
                // log sigmoid(x) = min(x, 0) - log1p(exp(-|x|));  log(1 - sigmoid(x)) = log sigmoid(x) - x
                const float ls = __builtin_fminf(x, 0.f) - log1pf(__expf(-__builtin_fabsf(x)));
                term = -(pw * y * ls + (1.f - y) * (ls - x));
                const float sg = 1.0f / (1.0f + __expf(-x));
                dl = scale * ((1.f - y) * sg - pw * y * (1.f - sg));
                preds[((long)hI * B + s0 + s) * K + k] = x > 0.f ? 1 : 0;
            }
            lg[s * HEAD_MAXK + k] = dl;
            tm[s * HEAD_MAXK + k] = term;
        }
        __syncthreads();
        if (tid < S) {
            float term = 0.f;
            for (int k = 0; k < K; ++k) term += tm[tid * HEAD_MAXK + k];      // label order, as the single-thread loop summed
            red[tid] = term / (float)K;
        }
    } else if (tid < S) {
        float term = 0.f;
        if (tid < ns) {
            const int s = tid;
            const int y = (int)labels[s0 + s];
            float mx = lg[s * HEAD_MAXK];
            int am = 0;
            for (int k = 1; k < K; ++k) { const float v = lg[s * HEAD_MAXK + k]; if (v > mx) { mx = v; am = k; } }
            float se = 0.f;
            for (int k = 0; k < K; ++k) se += __expf(lg[s * HEAD_MAXK + k] - mx);
            const float lse = __logf(se) + mx;
            term = lse - lg[s * HEAD_MAXK + y];
            const float scale = hd.weight / (float)B;
            const float inv = 1.0f / se;
            for (int k = 0; k < K; ++k) {
                const float p = __expf(lg[s * HEAD_MAXK + k] - mx) * inv;
                lg[s * HEAD_MAXK + k] = scale * (p - (k == y ? 1.f : 0.f));
            }
            preds[(long)hI * B + s0 + s] = am;
        } else {
            for (int k = 0; k < K; ++k) lg[tid * HEAD_MAXK + k] = 0.f;
        }
        red[tid] = term;
    }
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int s = 0; s < S; ++s) t += red[s];
        t /= (float)B;
        atomicAdd(losses + hI, t);
        atomicAdd(losses + nheads, t * hd.weight);
    }
    if (hd.d_pooled) {
        for (int idx = tid; idx < ns * D; idx += NTHREADS) {
            const int s = idx / D, d = idx % D;
            float a = 0.f;
            for (int k = 0; k < K; ++k) a = __builtin_fmaf(lg[s * HEAD_MAXK + k], wl[k * DL + d], a);
            hd.d_pooled[(long)(s0 + s) * D + d] = a;
        }
        // g_part: this workgroup's sums go to its slot (plain stores; m2m_towers_wgrad_heads adds the slots in a fixed order)
        float* slot = hd.g_part ? hd.g_part + (long)blockIdx.x * M2M_SPLIT_GPART : nullptr;
        for (int idx = tid; idx < K * D; idx += NTHREADS) {
            const int k = idx / D, d = idx % D;
            float a = 0.f;
            for (int s = 0; s < S; ++s) a = __builtin_fmaf(lg[s * HEAD_MAXK + k], pl[s * DL + d], a);
            if (slot) slot[idx] = a;
            else atomicAdd(hd.g_w + idx, a);
        }
        if (tid < K) {
            float a = 0.f;
            for (int s = 0; s < S; ++s) a += lg[s * HEAD_MAXK + tid];
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

template <bool BCE, int S>
static int launch_heads_s(const HeadArgs& ha, int nheads, const void* labels, const float* pos_weight, int B, int D, int K,
                          float* logits, float* losses, int32_t* preds, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)S * (D + 1) + (size_t)HEAD_MAXK * (D + 1) + 2 * S * HEAD_MAXK + S);
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(heads_kernel<BCE, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
    hipLaunchKernelGGL((heads_kernel<BCE, S>), dim3((B + S - 1) / S, nheads), dim3(NTHREADS), lds, st, ha, labels, pos_weight, B, D, K, logits, losses, preds, nheads);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
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
