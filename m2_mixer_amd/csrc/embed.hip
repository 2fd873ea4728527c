// Patch embedding (modules/mixer.py:143-146) and the plain input projection of MLPMixerNoPatching
// (modules/mixer.py:171,180), forward and weight gradient.
//
//   x0[m][d] = sum_k patch[m][k] W[d][k] + b[d],   m = b*N + (gy*GW + gx),  k = c*ph*pw + py*pw + px
//
// The conv with stride == kernel is an unfold + GEMM; the unfold is done on the fly while staging the
// input (read once, coalesced along image rows) into LDS -- address = rowbase[m] + koff[k] from two
// small LDS tables, so no integer division in the streaming loop -- and the weight fragments come
// straight from the packed NAT copy in global memory.
#include "tile.h"
#include "embed_wgrad.h"

#include "embed_fwd.h"

template <int P, int D, int RB>
__global__ __launch_bounds__(NTHREADS) void embed_fwd_kernel(const m2m_embed em, const float* __restrict__ in, long M, int N,
                                                             float* __restrict__ x0, int fast, const m2m_step_head head) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x == 0) {                              // the step prologue rides here (m2m_embed_forward_head): see m2m_step_head
        const int t = threadIdx.x;
        if (t == 0 && head.adam_state) head.adam_state[0] += 1.0f;
        if (t == 1 && head.drop_counter) *head.drop_counter += 1u;
        if (head.losses && t < head.nlosses) head.losses[t] = 0.f;
    }
    if (P == PREC_BF16 && fast) embed_fwd_fast_body<D>(em, in, M, N, x0, blockIdx.x, 0, 1, smem);
    else embed_fwd_body<P, D, RB>(em, in, M, N, x0, blockIdx.x, smem);
}

// Both patch embeddings of a two-tower model in ONE launch (no fork / join of a second stream at the head of the step):
// workgroups [0, nwg0) serve embedding 0 -- the one with the larger patch, dispatched first -- the rest embedding 1.
struct EmbedFwdGroupArgs {
    m2m_embed em[2];
    const float* in[2];
    float* x0[2];
    long M[2];
    int N[2], nwg0, fast[2], nsplit[2];
    long part_stride[2];       // floats between the k-split partial outputs
    // head of a training step folded into this launch (the first one of the step; nothing in it reads these):
    // adam_state[0] += 1, *drop_counter += 1, losses[0..nlosses) = 0 -- what m2m_step_prologue does in a launch of its own
    float* adam_state;
    unsigned int* drop_counter;
    float* losses;
    int nlosses, prologue;
};
// bf16, hidden_dim <= 128: built for TWO workgroups per CU (4 waves per SIMD, <= 128 VGPRs; the fast body's ring holds two stages):
// M2-Mixer-B's launch is 384 workgroups -- 256 of the audio embedding, 128 short ones of the image embedding -- and with one
// workgroup per CU it ran as a full round plus a half-empty one (21.3 -> 16 us, profiles/r04_ab_results.txt r5a).
template <int P, int D, int RB>
__global__ __launch_bounds__(NTHREADS, (P == PREC_BF16 && D <= 128) ? 4 : 1) void embed_fwd_group_kernel(const EmbedFwdGroupArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.prologue && blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t == 0 && a.adam_state) a.adam_state[0] += 1.0f;
        if (t == 1 && a.drop_counter) *a.drop_counter += 1u;
        if (a.losses && t < a.nlosses) a.losses[t] = 0.f;
    }
    const int e = (int)blockIdx.x < a.nwg0 ? 0 : 1;
    const int wg = e ? blockIdx.x - a.nwg0 : blockIdx.x;
    if (P == PREC_BF16 && a.fast[e]) {
        const int ns = a.nsplit[e], split = wg % ns;
        embed_fwd_fast_body<D>(a.em[e], a.in[e], a.M[e], a.N[e], a.x0[e] + split * a.part_stride[e], wg / ns, split, ns, smem);
    }
    else {
        // generic path asked for partial sums (an input the fast path cannot take, e.g. not 16-byte aligned): split 0 computes
        // the whole sum into part 0, the other splits' workgroups clear their rows of their part
        const int ns = a.nsplit[e], split = wg % ns, tile = wg / ns;
        if (split == 0) embed_fwd_body<P, D, RB>(a.em[e], a.in[e], a.M[e], a.N[e], a.x0[e], tile, smem);
        else {
            float* part = a.x0[e] + split * a.part_stride[e];
            const long m0 = (long)tile * RB;
            for (int idx = threadIdx.x; idx < RB * D; idx += NTHREADS) {
                const long m = m0 + idx / D;
                if (m < a.M[e]) part[m * D + idx % D] = 0.f;
            }
        }
    }
}

template <int P, int D>
__global__ __launch_bounds__(NTHREADS) void embed_wgrad_kernel(const m2m_embed em, const float* __restrict__ in,
                                                               const float* __restrict__ dx0, long M, int N, int tiles_per_group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    embed_wgrad_body<P, D, NTHREADS>(em, in, dx0, M, N, tiles_per_group, blockIdx.x, blockIdx.y, gridDim.y == 1, smem);
}

template <int P, int D>
__global__ __launch_bounds__(NTHREADS) void embed_wgrad_group_kernel(const EmbedWgradGroupArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    embed_wgrad_group_body<P, D, NTHREADS>(a, blockIdx.x, smem);
}

int m2m_check_embed(const m2m_embed* e, int B) {
    if (!e || B < 1) { m2m_set_error("embed: bad argument", __FILE__, __LINE__); return -1; }
    if (e->H % e->ph || e->W % e->pw) { m2m_set_error("embed: image not divisible by patch", __FILE__, __LINE__); return -1; }
    if (e->K != e->Cin * e->ph * e->pw) { m2m_set_error("embed: K != Cin*ph*pw", __FILE__, __LINE__); return -1; }
    const int KB = e->prec == PREC_BF16 ? 32 : 16;
    if (e->Kp % KB || e->Kp < e->K) { m2m_set_error("embed: Kp must be K rounded up to the k-block", __FILE__, __LINE__); return -1; }
    if (e->Kp > EMB_KMAX - EMB_KS) { m2m_set_error("embed: patch too large (Cin*ph*pw must be <= 3968)", __FILE__, __LINE__); return -1; }
    return 0;
}

template <int P, int D>
static int launch_embed_fwd(const m2m_embed* e, const float* in, int B, float* x0, hipStream_t st, const m2m_step_head* head = nullptr) {
    m2m_step_head hd;
    memset(&hd, 0, sizeof(hd));
    if (head) {
        if (head->nlosses < 0 || head->nlosses > 64) { m2m_set_error("embed_forward: step head nlosses must be in [0, 64]", __FILE__, __LINE__); return -1; }
        hd = *head;
    }
    const int N = (e->H / e->ph) * (e->W / e->pw);
    const long M = (long)B * N;
    // 16 rows per workgroup: the audio embedding (2048 rows at batch 512, 50 KB of input per sample) sits at the head of
    // the step's critical path, and 64 workgroups of 32 rows left three quarters of the chip idle
    constexpr int RB = 16;
    const size_t lds = (size_t)RB * EMB_LD * 4 + (size_t)RB * EMB_KS * Prec<P>::ESZ + EMB_KMAX * 4 + RB * 8;
    auto kern = embed_fwd_kernel<P, D, RB>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)((M + RB - 1) / RB)), dim3(NTHREADS), lds, st, *e, in, M, N, x0, (int)embed_fwd_fast_ok(e, in), hd);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int D>
static int launch_embed_fwd_group(const m2m_embed* const* es, const float* const* ins, float* const* x0s, const int* nsplits,
                                  const int64_t* part_strides, const m2m_step_head* head, int B, hipStream_t st) {
    constexpr int RB = 16;
    EmbedFwdGroupArgs a;
    memset(&a, 0, sizeof(a));
    if (head) {
        if (head->nlosses < 0 || head->nlosses > 64) { m2m_set_error("embeds_forward: step head nlosses must be in [0, 64]", __FILE__, __LINE__); return -1; }
        a.prologue = 1; a.adam_state = head->adam_state; a.drop_counter = head->drop_counter; a.losses = head->losses; a.nlosses = head->nlosses;
    }
    const int first = es[1]->Kp > es[0]->Kp ? 1 : 0;
    int total = 0;
    for (int k = 0; k < 2; ++k) {
        const int i = k == 0 ? first : 1 - first;
        const int N = (es[i]->H / es[i]->ph) * (es[i]->W / es[i]->pw);
        a.em[k] = *es[i]; a.in[k] = ins[i]; a.x0[k] = x0s[i]; a.N[k] = N; a.M[k] = (long)B * N;
        a.fast[k] = embed_fwd_fast_ok(es[i], ins[i]);
        a.nsplit[k] = nsplits ? nsplits[i] : 1;
        a.part_stride[k] = part_strides ? (long)part_strides[i] : 0;
        if (a.nsplit[k] < 1 || a.nsplit[k] > 4 || (a.nsplit[k] > 1 && a.part_stride[k] < a.M[k] * (long)D)) {
            m2m_set_error("embeds_forward: 1..4 k-splits, part stride >= B*N*D", __FILE__, __LINE__);
            return -1;
        }
        const int nwg = (int)((a.M[k] + RB - 1) / RB) * a.nsplit[k];
        if (k == 0) a.nwg0 = nwg;
        total += nwg;
    }
    const size_t lds = (size_t)RB * EMB_LD * 4 + (size_t)RB * EMB_KS * Prec<P>::ESZ + EMB_KMAX * 4 + RB * 8;
    auto kern = embed_fwd_group_kernel<P, D, RB>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NTHREADS), lds, st, a);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int P, int D>
static int launch_embed_wgrad(const m2m_embed* e, const float* in, const float* dx0, int B, hipStream_t st) {
    const EmbedWgradPlan pl = embed_wgrad_plan(e, B, 256);
    const size_t lds = embed_wgrad_lds<D, P>();
    auto kern = embed_wgrad_kernel<P, D>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)pl.nchunks, (unsigned)pl.groups), dim3(NTHREADS), lds, st, *e, in, dx0, pl.M, pl.N, pl.tpg);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int D>
static int launch_embed_wgrad_group(const m2m_embed* const* es, const float* const* ins, const float* const* dx0s, int B, hipStream_t st) {
    EmbedWgradGroupArgs a;
    const int total = embed_wgrad_group_args(a, es, ins, dx0s, B, 256);
    const size_t lds = embed_wgrad_lds<D, P>();
    auto kern = embed_wgrad_group_kernel<P, D>;
    static bool done = false;
    if (!done) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NTHREADS), lds, st, a);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int m2m_embed_forward(const m2m_embed* e, const float* input, int B, float* x0, void* stream) {
    return m2m_embed_forward_head(e, input, B, x0, nullptr, stream);
}
extern "C" int m2m_embed_forward_head(const m2m_embed* e, const float* input, int B, float* x0, const m2m_step_head* head, void* stream) {
    if (int rc = m2m_check_embed(e, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_EF_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_fwd<PP, DD>(e, input, B, x0, st, head);
    M2M_EF_CASE(PREC_BF16, 32) M2M_EF_CASE(PREC_BF16, 64) M2M_EF_CASE(PREC_BF16, 128) M2M_EF_CASE(PREC_BF16, 256)
    M2M_EF_CASE(PREC_F32, 32) M2M_EF_CASE(PREC_F32, 64) M2M_EF_CASE(PREC_F32, 128) M2M_EF_CASE(PREC_F32, 256)
#undef M2M_EF_CASE
    m2m_set_error("embed_forward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_embed_wgrad(const m2m_embed* e, const float* input, const float* d_x0, int B, void* stream) {
    if (int rc = m2m_check_embed(e, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_EW_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_wgrad<PP, DD>(e, input, d_x0, B, st);
    M2M_EW_CASE(PREC_BF16, 32) M2M_EW_CASE(PREC_BF16, 64) M2M_EW_CASE(PREC_BF16, 128) M2M_EW_CASE(PREC_BF16, 256)
    M2M_EW_CASE(PREC_F32, 32) M2M_EW_CASE(PREC_F32, 64) M2M_EW_CASE(PREC_F32, 128) M2M_EW_CASE(PREC_F32, 256)
#undef M2M_EW_CASE
    m2m_set_error("embed_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_embeds_wgrad(const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s, int nembeds,
                                int B, void* stream) {
    if (!embeds || !inputs || !d_x0s || nembeds != EMB_GROUP) { m2m_set_error("embeds_wgrad: exactly two embeddings", __FILE__, __LINE__); return -1; }
    for (int i = 0; i < nembeds; ++i) {
        if (int rc = m2m_check_embed(embeds[i], B)) return rc;
        if (embeds[i]->prec != embeds[0]->prec || embeds[i]->D != embeds[0]->D) {
            m2m_set_error("embeds_wgrad: the embeddings of one launch must share precision and hidden_dim", __FILE__, __LINE__);
            return -1;
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const m2m_embed* e = embeds[0];
#define M2M_EWG_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_wgrad_group<PP, DD>(embeds, inputs, d_x0s, B, st);
    M2M_EWG_CASE(PREC_BF16, 32) M2M_EWG_CASE(PREC_BF16, 64) M2M_EWG_CASE(PREC_BF16, 128) M2M_EWG_CASE(PREC_BF16, 256)
    M2M_EWG_CASE(PREC_F32, 32) M2M_EWG_CASE(PREC_F32, 64) M2M_EWG_CASE(PREC_F32, 128) M2M_EWG_CASE(PREC_F32, 256)
#undef M2M_EWG_CASE
    m2m_set_error("embeds_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

// How many k-splits m2m_embeds_forward should be given for this embedding: 2 when the fast path applies and K is long
// enough for the weight stream to dominate (the audio spectrogram patches), else 1.
extern "C" int m2m_embed_fwd_splits(const m2m_embed* e) {
    if (!e) return 1;
    static const int forced = [] { const char* v = getenv("M2M_EMBED_SPLITS"); return v ? atoi(v) : 0; }();    // diagnostic (1..4)
    const bool pays = embed_fwd_fast_ok(e, nullptr) && e->Kp >= 4 * EMB_FKS;
    if (pays && forced >= 1 && forced <= 4) return forced;
    return pays ? 2 : 1;
}

extern "C" int m2m_embeds_forward(const m2m_embed* const* embeds, const float* const* inputs, float* const* x0s, const int* nsplits,
                                  const int64_t* part_strides, int nembeds, int B, const m2m_step_head* head, void* stream) {
    if (!embeds || !inputs || !x0s || nembeds != 2) { m2m_set_error("embeds_forward: exactly two embeddings", __FILE__, __LINE__); return -1; }
    for (int i = 0; i < nembeds; ++i) {
        if (int rc = m2m_check_embed(embeds[i], B)) return rc;
        if (embeds[i]->prec != embeds[0]->prec || embeds[i]->D != embeds[0]->D) {
            m2m_set_error("embeds_forward: the embeddings of one launch must share precision and hidden_dim", __FILE__, __LINE__);
            return -1;
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const m2m_embed* e = embeds[0];
#define M2M_EFG_CASE(PP, DD) if (e->prec == PP && e->D == DD) return launch_embed_fwd_group<PP, DD>(embeds, inputs, x0s, nsplits, part_strides, head, B, st);
    M2M_EFG_CASE(PREC_BF16, 32) M2M_EFG_CASE(PREC_BF16, 64) M2M_EFG_CASE(PREC_BF16, 128) M2M_EFG_CASE(PREC_BF16, 256)
    M2M_EFG_CASE(PREC_F32, 32) M2M_EFG_CASE(PREC_F32, 64) M2M_EFG_CASE(PREC_F32, 128) M2M_EFG_CASE(PREC_F32, 256)
#undef M2M_EFG_CASE
    m2m_set_error("embeds_forward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}
