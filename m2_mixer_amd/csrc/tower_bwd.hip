// Backward of a stack of MixerBlocks (+ final LayerNorm): the data-gradient chain -- one launch per tower.
//
// Same tiling as the forward: a workgroup owns BM = 16 token rows (whole samples) and walks the blocks in
// reverse with the fp32 gradient stream of those rows resident in LDS.  Per block:
//   channel mixing:  dYd = dY * mask_out;  A = LN2(x_mid) recomputed from the saved x_mid;
//                    per 32 hidden columns and wave:  Hpre^T = W1 A^T + b1,  dHact^T = W2^T dYd^T   (MFMA)
//                    dHpre = dHact * mask * gelu'(Hpre) on the accumulators, chained straight into
//                    dA += dHpre W1 (MFMA).  LayerNorm backward, residual add.
//   token mixing:    recompute LN1(x_in) and the token MLP per (sample, channel) on the VALU; eight lanes
//                    share a column and split the T hidden units, so the token-weight gradients are
//                    accumulated in registers and reduced once per block.
// Small parameter gradients (LayerNorm, token MLP, ch_b2) are added to the fp32 gradient buffers with
// float atomics.  The channel-mixing WEIGHT gradients need a reduction over all rows and are left to
// tower_wgrad.hip; this kernel writes the operands it contracts: A^T, dYd^T per token tile and, per hidden column
// tile, Hact^T and dHpre^T (the hidden activation and its gradient leave the chip ONCE, in operand precision, here).
#include "tile.h"
#include "token_mfma.h"
#include "split.h"
int m2m_split_small_grads(const SplitReduceArgs& a, hipStream_t st);     // split_mix.hip: sum of the per-workgroup slots into the gradients
#include <algorithm>

// The classification heads + multi-head cross-entropy computed in the PROLOGUE of the fusion tower's backward launch (template
// flag HEADS) instead of a launch of their own between forward and backward (heads.hip: 13.5 us of pure latency per step for
// ~4 MFLOP).  A workgroup owns whole samples, and a sample's three heads need nothing but its own token means (reference:
// models/avmnist.py:271-298 -- classifier_image / classifier_audio on tokens.mean(1), StandardClassifier on the fused tokens,
// three CrossEntropyLoss(mean), loss = sum_h weight_h L_h): logits, predictions, loss terms and the gradients wrt the token means
// come out per workgroup; the head weight gradients and the loss sums go to the workgroup's partial-sum slots (one slot set per
// head behind the blocks' sets: K D + K + 2 <= M2M_SPLIT_GPART floats) and are added up by the reduction launch that follows
// this one anyway (m2m_split_small_grads) -- no atomics.  The gradient wrt THIS tower's token mean stays in LDS; the other
// towers' go to global memory for their backward launch.
#define BH_MAXH 3
struct BwdHeads {
    m2m_head h[BH_MAXH];
    const int64_t* labels;
    float* logits;                 // (nheads, B, K)
    int32_t* preds;                // (nheads, B)
    int nheads, K, own;            // own: index of the head that sits on this tower's token mean
};

// LDS budget of the backward chain kernel (bytes): FIXED + nblocks * PB * 4
template <int P, int D, int NMAX, int TG> struct BwdLds {
    static constexpr bool TOK = NMAX > 0;
    static constexpr int NM = TOK ? NMAX : 1, TW_LD = 2 * NM + 4, XLD = D + 4;
    static constexpr int RED_LD0 = ((32 / TG) * (1 + 2 * NM) + NM) * TG;      // token-grad slots per wave (VALU form, fp32 mode)
    static constexpr int RED_LD = RED_LD0 > TokRed<NM>::LD ? RED_LD0 : TokRed<NM>::LD;
    static constexpr int PB = 4 * D + (TOK ? 32 * TW_LD : 0);                  // floats of one block's small parameters
    static constexpr size_t FIXED = (size_t)BM * XLD * sizeof(float) * (1 + RowSlabs<D>::N + 2) + 2 * (size_t)BM * D * Prec<P>::ESZ +
                                    GELU_TAB_N * sizeof(gtabB_t) + (TOK ? (size_t)NWAVES * RED_LD * sizeof(float) : 0);
    // + keep-words of the token-hidden site (one per column of the workgroup's SPW = BM / N samples) + hidden bias of one block
    static size_t bytes(int nblocks, int N, int Cp) {
        return FIXED + (TOK ? (size_t)(BM / N) * D * sizeof(unsigned int) : 0) + (size_t)nblocks * PB * sizeof(float) +
               (size_t)Cp * sizeof(float) + 16 + 8 * (M2M_MAX_BLOCKS + 1 + BH_MAXH);
    }
};
#ifdef M2M_TIMERS
#define M2M_LDS_MAX (163840 - 1024)     // the diagnostic build keeps its timer slots in static LDS
#else
#define M2M_LDS_MAX 163840
#endif

// the small parameter gradients' float atomics (M2M_ABL_NOATOM: timing ablation, drops them)
#ifdef M2M_ABL_NOATOM
#define M2M_SMALL_ATOMIC(p, v) asm volatile("" :: "v"(p), "v"(v))
#else
#define M2M_SMALL_ATOMIC(p, v) atomicAdd(p, v)
#endif
TIMER_DECL(g_tm_bwd);
// occupancy the scheduler plans for (see the kernels' declarations)
#ifndef M2M_BWD_KATTR
#define M2M_BWD_KATTR
#endif
__device__ int g_bwd_static_split = 1;      // 1: static split of the bf16 column loop (default), 0: ticket counter (M2M_BWD_TICKETS=1)
// copies the environment's choice to the device once per process (before the first backward launch on any stream)
static int bwd_split_mode_init(hipStream_t st) {
    static bool done = false;
    if (done) return 0;
    const char* e = getenv("M2M_BWD_TICKETS");
    if (e && atoi(e) != 0) {
        const int v = 0;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cs);
        if (cs != hipStreamCaptureStatusNone) { m2m_set_error("tower_backward: first launch with M2M_BWD_TICKETS=1 inside a stream capture (run one eager step first)", __FILE__, __LINE__); return -1; }
        M2M_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_static_split), &v, sizeof(v)));
    }
    done = true;
    return 0;
}
TIMER_READER(m2m_debug_timers_bwd, g_tm_bwd)

// One workgroup's share of a tower backward: token tile `wg` of `nwg`.  TW is m2m_tower (single-tower launch) or
// m2m_tower4 (the by-value descriptors of a two-tower launch).
// HREC: the weight-gradient launch recomputes the hidden activation itself (tower_wgrad_rc.h): this kernel then stores the packed
// NAT image of A = LN2(x_mid) where Hact^T would have gone (m2m_block.h_chn) and keeps only the dHpre^T stream -- half the
// operand spill, two of the four transposing MFMAs and one of the two streaming stores per step less.
template <class TW, int P, int D, int NMAX, int TG, int DM, bool PART = false, bool HREC = false, bool HEADS = false>
static __device__ __forceinline__ void tower_bwd_body(const TW& tw, int B, const float* __restrict__ d_out, long d_out_ss,
                                                      const float* __restrict__ d_pooled, float* __restrict__ d_x0, long d_x0_ss,
                                                      unsigned int seed, unsigned int step_host,
                                                      const unsigned int* __restrict__ step_dev, int wg, int nwg, char* smem,
                                                      float* __restrict__ part = nullptr, char* dx0_chn = nullptr,
                                                      const BwdHeads* hd = nullptr) {
    typedef Prec<P> Pr;
    typedef TileGeom<D> G;
    typedef BwdLds<P, D, NMAX, TG> L;
    constexpr int XLD = G::XLD, DT = G::DT, KD = D / Pr::KB, NF = Chain<P>::NF;
    constexpr bool TOK = NMAX > 0;                      // false: wide path, channel mixing only (rows independent)
    constexpr int NM = TOK ? NMAX : 1;
    constexpr int TILE_F = BM * XLD;                    // floats in one fp32 tile
    constexpr int IMG_B = BM * D * Pr::ESZ;             // bytes of one packed BM-row image
    constexpr int EPT = D / TPR, NSL = RowSlabs<D>::N;
    constexpr int TW_LD = 2 * NM + 4, RED_LD = L::RED_LD, PB = L::PB;
    constexpr int O_LN1W = 0, O_LN1B = D, O_LN2W = 2 * D, O_LN2B = 3 * D, O_TOKW = 4 * D;

    // LDS: gradient stream | one row-major partial-dA slab per wave (RowSlabs); after the row threads have summed them, slabs
    //      0-3 are reused in place as fp32 tiles (each thread rewrites only the elements it alone has read) | dYd / A tiles
    //      (sources of the transposed operand copies) | packed A / dYd images | GELU table | token-gradient partial sums |
    //      keep-words | the small parameters of every block (loaded once)
    float* dxs = reinterpret_cast<float*>(smem);        // gradient stream
    float* slabs = dxs + TILE_F;
    float* t_prod = slabs;                               // LayerNorm gamma-gradient products (column-summed one phase later)
    float* t_up = slabs + TILE_F;                        // LayerNorm beta-gradient source
    float* ub = slabs + 2 * TILE_F;                      // token path: U = LN1 output in, dU out
    float* dov = slabs + 3 * TILE_F;                     // token path: dO' = dropout'(dx_mid)
    float* tdy = slabs + NSL * TILE_F;                   // dYd (fp32)
    float* ta = tdy + TILE_F;                            // A = LN2(x_mid) (fp32)
    char* at = reinterpret_cast<char*>(ta + TILE_F);
    char* dyp = at + IMG_B;
    gtabB_t* gtab = reinterpret_cast<gtabB_t*>(dyp + IMG_B);     // [GELU_TAB_N] (bf16 mode only)
    float* red = reinterpret_cast<float*>(gtab + GELU_TAB_N);    // [NWAVES][RED_LD] (token path)
    unsigned int* wth = reinterpret_cast<unsigned int*>(red + (TOK ? NWAVES * RED_LD : 0));   // [BM * D] (token path)
    float* par = reinterpret_cast<float*>(wth + (TOK ? (BM / tw.N) * D : 0));                 // [nblocks][PB]
    float* bias_s = par + tw.nblocks * PB;                                                    // [Cp] hidden bias of the block in flight
    unsigned int* qctr = reinterpret_cast<unsigned int*>(bias_s + tw.Cp);                      // ticket counter of the column loop
    // this workgroup's partial-sum slots (one per slot set; 0: atomics) -- kept in LDS, not in registers: nothing of the
    // small-gradient bookkeeping stays live across the hidden-column loop (as registers it cost the loop 2-10 VGPR spills)
    unsigned long long* slotp = reinterpret_cast<unsigned long long*>(qctr + 4);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, il = lane & 15;
    const int wave = tid >> 6;
    const int N = tw.N, T = tw.T, Cp = tw.Cp;
    const int SPW = TOK ? BM / N : 0;
    const int s0 = wg * SPW;
    const int ns = TOK ? min(SPW, B - s0) : 0;
    const long row0 = TOK ? (long)s0 * N : (long)wg * BM;
    const int R = TOK ? ns * N : (int)min((long)BM, (long)B * N - row0);
    constexpr int TPP = WPAIR / BM;                                         // chain tiles per 32-row pair
    const long pair_off = (long)(wg / TPP) * (WPAIR * D * Pr::ESZ); // CHN images: per 32-row pair
    const int tile_in_pair = wg % TPP;
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);

    TIMER_LSTART();
    if (PART && threadIdx.x <= (unsigned)tw.nblocks + (HEADS ? BH_MAXH : 0))
        slotp[threadIdx.x] = (unsigned long long)(part + ((long)threadIdx.x * nwg + wg) * SPP_STRIDE);
    // ---- classification heads of this workgroup's samples (HEADS): see BwdHeads ----
    float* hdp = slabs + 3 * 4 * D + BH_MAXH * 32 * (D + 1) + BH_MAXH * 4 * 32;     // [SPW][D] d(loss) / d(this tower's token mean)
    if constexpr (HEADS && PART && TOK) {
        const int nh = hd->nheads, K = hd->K, DL = D + 1;
        float* hpool = slabs;                                   // [nh][4][D]
        float* hwt = hpool + 3 * 4 * D;                         // [nh][K][D + 1]
        float* hlog = hwt + BH_MAXH * 32 * DL;                  // [nh][4][32] logits, then dlogits
        for (int i = tid; i < nh * SPW * D; i += NTHREADS) {
            const int h = i / (SPW * D), sl = (i / D) % SPW, d = i % D;
            hpool[(h * 4 + sl) * D + d] = sl < ns ? hd->h[h].pooled[(long)(s0 + sl) * D + d] : 0.f;
        }
        for (int i = tid; i < nh * K * D; i += NTHREADS) {
            const int h = i / (K * D), k = (i / D) % K, d = i % D;
            hwt[(h * 32 + k) * DL + d] = hd->h[h].w[k * D + d];
        }
        __syncthreads();
        // logits: 8 adjacent lanes split each D-long dot product
        for (int i = tid >> 3; i < nh * SPW * K; i += NTHREADS / 8) {
            const int h = i / (SPW * K), sl = (i / K) % SPW, k = i % K, part8 = tid & 7;
            float a = 0.f;
            for (int d = part8; d < D; d += 8) a = __builtin_fmaf(hpool[(h * 4 + sl) * D + d], hwt[(h * 32 + k) * DL + d], a);
            a = wave_sum_xor(a, 8) + hd->h[h].b[k];
            if (part8 == 0) {
                hlog[(h * 4 + sl) * 32 + k] = a;
                if (sl < ns) hd->logits[((long)h * B + s0 + sl) * K + k] = a;
            }
        }
        __syncthreads();
        // softmax cross-entropy per (head, sample): loss term, prediction, dlogits in place
        float term = 0.f;
        if (tid < nh * SPW) {
            const int h = tid / SPW, sl = tid % SPW;
            float* lg = hlog + (h * 4 + sl) * 32;
            if (sl < ns) {
                const int y = (int)hd->labels[s0 + sl];
                float mx = lg[0];
                int am = 0;
                for (int k = 1; k < K; ++k) { const float v = lg[k]; if (v > mx) { mx = v; am = k; } }
                float se = 0.f;
                for (int k = 0; k < K; ++k) se += __expf(lg[k] - mx);
                term = (__logf(se) + mx - lg[y]) / (float)B;
                const float scale = hd->h[h].weight / (float)B, inv = 1.0f / se;
                for (int k = 0; k < K; ++k) lg[k] = scale * (__expf(lg[k] - mx) * inv - (k == y ? 1.f : 0.f));
                hd->preds[(long)h * B + s0 + sl] = am;
            } else {
                for (int k = 0; k < K; ++k) lg[k] = 0.f;
            }
            hlog[BH_MAXH * 4 * 32 - 16 + tid] = term;          // (rows [2][3][20..31] of the last head's tile are never logits: K <= 11)
        }
        __syncthreads();
        // gradients wrt the token means; head weight / bias gradients and loss sums -> this workgroup's slots
        for (int i = tid; i < nh * ns * D; i += NTHREADS) {
            const int h = i / (ns * D), sl = (i / D) % ns, d = i % D;
            float a = 0.f;
            for (int k = 0; k < K; ++k) a = __builtin_fmaf(hlog[(h * 4 + sl) * 32 + k], hwt[(h * 32 + k) * DL + d], a);
            if (h == hd->own) hdp[sl * D + d] = a;
            else hd->h[h].d_pooled[(long)(s0 + sl) * D + d] = a;
        }
        for (int i = tid; i < nh * (K * D + K + 2); i += NTHREADS) {
            const int h = i / (K * D + K + 2), e = i % (K * D + K + 2);
            float* sl_h = reinterpret_cast<float*>(slotp[tw.nblocks + 1 + h]);
            float a = 0.f;
            if (e < K * D) {
                const int k = e / D, d = e % D;
                for (int sl = 0; sl < SPW; ++sl) a = __builtin_fmaf(hlog[(h * 4 + sl) * 32 + k], hpool[(h * 4 + sl) * D + d], a);
            } else if (e < K * D + K) {
                for (int sl = 0; sl < SPW; ++sl) a += hlog[(h * 4 + sl) * 32 + (e - K * D)];
            } else {
                for (int sl = 0; sl < SPW; ++sl) a += hlog[BH_MAXH * 4 * 32 - 16 + h * SPW + sl];
                if (e == K * D + K + 1) a *= hd->h[h].weight;                 // this head's share of the total loss
            }
            sl_h[e] = a;
        }
        __syncthreads();
    }
    constexpr int MAXB = (int)(sizeof(tw.blk) / sizeof(tw.blk[0]));       // blocks the descriptor type can hold
    // ---- prologue.  EVERY global load of the launch's start is requested before the first LDS write: the upstream gradient,
    //      the last block's x_mid rows and hidden bias (used by the first phase of the block loop), the small parameters of
    //      every block (block index wave-uniform: a per-thread index into the by-value descriptor would turn every later
    //      descriptor read into a vector load); the GELU table is computed while they fly. ----
    constexpr int XI = (BM * (D / 4) + NTHREADS - 1) / NTHREADS, BPT = 8, TI = TOK ? (32 * TW_LD + NTHREADS - 1) / NTHREADS : 1;
    float4 uv[XI];
    {
        const float invN = 1.0f / (float)N;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int idx = tid + k * NTHREADS, r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            uv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BM * (D / 4) && r < R) {
                const long gr = row0 + r, gs = gr / N;
                if (d_out) uv[k] = *reinterpret_cast<const float4*>(d_out + gs * d_out_ss + (gr % N) * D + c);
                if constexpr (HEADS && PART && TOK) {
                    const float4 p = *reinterpret_cast<const float4*>(hdp + (gs - s0) * D + c);
                    uv[k].x += p.x * invN; uv[k].y += p.y * invN; uv[k].z += p.z * invN; uv[k].w += p.w * invN;
                } else if (d_pooled) {
                    const float4 p = *reinterpret_cast<const float4*>(d_pooled + gs * D + c);
                    uv[k].x += p.x * invN; uv[k].y += p.y * invN; uv[k].z += p.z * invN; uv[k].w += p.w * invN;
                }
            }
        }
    }
    float xm[EPT];                                                  // x_mid rows of the block about to be processed
    float nb[BPT];                                                  // this thread's share of that block's hidden bias (Cp <= BPT * NTHREADS)
    {
        const int r = tid / TPR, j = tid % TPR;
#pragma unroll
        for (int e = 0; e < EPT; ++e) xm[e] = 0.f;
        if (r < R) ld_row<D>(tw.blk[tw.nblocks - 1].x_mid + (row0 + r) * D, j, xm);
        // (the pointer as a scalar, read ONCE: left inside the guarded loads hipcc re-read it from the descriptor in front of
        // every one of them -- pointer load, vmcnt(0), data load, eight times in series)
        const M2M_AS1 float* b1p = reinterpret_cast<const M2M_AS1 float*>(to_gptr(tw.blk[tw.nblocks - 1].ch_b1p));
#pragma unroll
        for (int k = 0; k < BPT; ++k) {
            nb[k] = 0.f;
            if (tid + k * NTHREADS < Cp) nb[k] = b1p[tid + k * NTHREADS];
        }
    }
    float pv[MAXB][4], tv[TI][MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[b][q] = 0.f;
        if (b < tw.nblocks && tid < D) {
            const m2m_block& bk = tw.blk[b];
            if (TOK) { pv[b][0] = bk.ln1_w[tid]; pv[b][1] = bk.ln1_b[tid]; }
            pv[b][2] = bk.ln2_w[tid]; pv[b][3] = bk.ln2_b[tid];
        }
#pragma unroll
        for (int k = 0; k < TI; ++k) {
            // zero-padded token weights: tokw[t][0..NMAX) = W1[t][n]  tokw[t][NMAX..2NMAX) = W2[n][t]  tokw[t][2NMAX] = b1[t]  (t < 32)
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
    if (ActB<P>::USES_TABLE) gelu_tabB_fill(gtab, make_drop(true, tw.p_drop, 0u, 0u, 0u).scale, tid, NTHREADS);
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
        if (b < tw.nblocks) {
            float* pb = par + b * PB;
            if (tid < D) {
                if (TOK) { pb[O_LN1W + tid] = pv[b][0]; pb[O_LN1B + tid] = pv[b][1]; }
                pb[O_LN2W + tid] = pv[b][2]; pb[O_LN2B + tid] = pv[b][3];
            }
            if constexpr (TOK) {
#pragma unroll
                for (int k = 0; k < TI; ++k)
                    if (tid + k * NTHREADS < 32 * TW_LD) pb[O_TOKW + tid + k * NTHREADS] = tv[k][b];
            }
        }
    // ---- upstream gradient of the tower output ----
    {
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int idx = tid + k * NTHREADS, r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            if (idx < BM * (D / 4)) *reinterpret_cast<float4*>((tw.has_final_ln ? tdy : dxs) + r * XLD + c) = uv[k];
        }
        __syncthreads();
        if (tw.has_final_ln)
        {
            if constexpr (PART) {
                float* sl = reinterpret_cast<float*>(slotp[0]);     // slot set 0: the final LayerNorm (written before the barrier above)
                ln_backward_tile<D, false>(tw.x_final + row0 * D, R, tdy, tw.lnf_w, dxs, false, ta, sl + SPP_LNF(D), sl + SPP_LNF(D) + D, tid);
            } else
            ln_backward_tile<D>(tw.x_final + row0 * D, R, tdy, tw.lnf_w, dxs, false, ta, tw.g_lnf_w, tw.g_lnf_b, tid);
        }
    }

    TIMER_LMARK(0);       // parameters, upstream + final LN backward
    int pend = -1;        // block whose LayerNorm-1 parameter gradients are still to be column-summed (from t_prod / ub)
    // column sums of two tiles -> global atomics (gamma gradient from `pw`, beta gradient from `pb_`); rows >= R hold zeros
    auto colsums = [&](const float* pw, const float* pb_, float* gw, float* gb, int t0, bool store = false) {
        _Pragma("unroll 1") for (int d = t0; d < 2 * D; d += NTHREADS) {
            const float* src = d < D ? pw : pb_;
            const int c = d < D ? d : d - D;
            float s_ = 0.f;
#pragma unroll 4
            for (int r = 0; r < BM; ++r) s_ += src[r * XLD + c];
            if (store) (d < D ? gw : gb)[c] = s_;                 // gw / gb point into this workgroup's slot
            else M2M_SMALL_ATOMIC((d < D ? gw : gb) + c, s_);
        }
    };
    // Small parameter gradients without atomics (`part` != NULL): every workgroup STORES its partial sums into its own slot
    // (split.h's layout; slot set 0 = the final LayerNorm, slot set nblocks - b = block b) and one reduction launch adds the
    // slots to the gradients in a fixed order.  256 workgroups adding to the same ~3000 addresses ran at the contended-atomic
    // rate: 16 us of the step's two backward launches (timing ablation M2M_ABL_NOATOM), and order-dependent sums.
    auto slot_of = [&](int b) -> float* { return PART ? reinterpret_cast<float*>(slotp[tw.nblocks - b]) : nullptr; };
    for (int b = tw.nblocks - 1; b >= 0; --b) {
        const m2m_block& bk = tw.blk[b];
        const float* pb = par + b * PB;
        const unsigned int site = tw.site_base + 4u * b;
        const Drop dr_th = make_drop(true, tw.p_drop, seed, step, site + 0);
        const Drop dr_to = make_drop(true, tw.p_drop, seed, step, site + 1);
        const Drop dr_ch = make_drop(true, tw.p_drop, seed, step, site + 2);
        const Drop dr_co = make_drop(true, tw.p_drop, seed, step, site + 3);

        // Per-phase copies of the thread index behind an opaque asm: the phases' index arithmetic is then recomputed where it
        // is used instead of being hoisted out of the block loop and kept live across the hidden-column loop, which needs
        // every register it can get (the hoisted values were spilled to scratch and reloaded at L2 latency in each phase).
        int tb1 = tid;
        asm volatile("" : "+v"(tb1));
        // ================= channel mixing backward =================
        // (X1) on the row thread's registers: dYd = dY * mask_out and A = LN2(x_mid), each to its fp32 tile (source of the
        //      transposed operand copy and of the ch_b2 gradient) and straight into its packed NAT image.  The previous
        //      block's LayerNorm-1 parameter gradients are column-summed beside it (other tiles).
        {
            // this block's hidden bias -> LDS: requested a phase ago (with the x_mid rows), written at the end of this phase
            // (the column loop reads it from there: a global load of it at the top of every step was waited for at once, and
            // with it -- the memory counter is in order -- everything requested before it)
            const int r = tb1 / TPR, j = tb1 % TPR;
            float dy[EPT], a[EPT], mean, rstd;
            ld_row<D>(dxs + r * XLD, j, dy);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int c = ln_col<D>(e, j);
                const float v = drop_keep_elem<DM>(dr_co, (unsigned int)(row0 + r) * D + c) ? dy[e] * dr_co.scale : 0.f;
                dy[e] = r < R ? v : 0.f;
            }
            st_row<D>(tdy + r * XLD, j, dy);
            pack_row_nat<P, D>(dyp, r, j, dy);
            reg_stats<D>(xm, mean, rstd);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int c = ln_col<D>(e, j);
                a[e] = (xm[e] - mean) * rstd * pb[O_LN2W + c] + pb[O_LN2B + c];
            }
            st_row<D>(ta + r * XLD, j, a);
            pack_row_nat<P, D>(at, r, j, a);
            if (TOK && pend >= 0) {
                float* sl = slot_of(pend);
                if constexpr (PART) colsums(t_prod, ub, sl + SPP_LN1(D), sl + SPP_LN1(D) + D, tb1, true);
                else colsums(t_prod, ub, tw.blk[pend].g_ln1_w, tw.blk[pend].g_ln1_b, tb1);
            }
            if (tb1 == 0) *qctr = NWAVES;
#pragma unroll
            for (int k = 0; k < BPT; ++k)
                if (tb1 + k * NTHREADS < Cp) bias_s[tb1 + k * NTHREADS] = nb[k];
        }
        __syncthreads();
        TIMER_LMARK(1);   // X1: dYd, A, packed images (+ previous block's LN1 column sums)
        // (X2) ch_b2 gradient (column sums of dYd) and the transposed (CHN) operand copies for the weight gradients: global
        //      writes from tiles nothing rewrites before the next block -- no barrier between here and the column loop
        // Every descriptor pointer of this phase and of the column loop, read in ONE batch into scalars: read where they are
        // used, each cost its own round trip (pointer load, vmcnt(0), use -- the descriptor is addressed through VGPRs once the
        // block index is a loop counter), four in series between the barrier above and the loop's first weight loads.
        gptr_w_t p_gb2 = to_gptr_w(bk.g_ch_b2), p_dyt = to_gptr_w(bk.dyt_chn), p_atc = to_gptr_w(bk.at_chn);
        gptr_t p_w1n = to_gptr(bk.w1n), p_w2tn = to_gptr(bk.w2tn), p_w1tc = to_gptr(bk.w1tc);
        gptr_w_t p_dh = to_gptr_w(bk.dh_chn), p_h = to_gptr_w(bk.h_chn);
        asm volatile("" : "+s"(p_gb2), "+s"(p_dyt), "+s"(p_atc), "+s"(p_w1n), "+s"(p_w2tn), "+s"(p_w1tc), "+s"(p_dh), "+s"(p_h));
        // The first step's weight fragments: requested HERE, in front of this phase's atomics and operand stores.  (a) their
        // latency lies beside the phase instead of at the top of the loop; (b) the loop's waits: memory operations retire in
        // order, and inside the loop two operand stores follow each step's 16 prefetch loads, so waiting for the last load may
        // leave 2 operations in flight -- but the waitcnt pass merges the loop's entry edge with its back edge, and with the
        // first loads issued LAST before the loop it had to wait for vmcnt(0) in every step, i.e. for the previous step's
        // non-temporal stores to be written back (measured without the stores: -6 us per launch).
        constexpr bool HOLD0 = D <= 128;
        const unsigned int lane16_0 = (unsigned int)lane * 16u;
        Frag w1f[2][HOLD0 ? D / Pr::KB : 1], w2f[2][HOLD0 ? D / Pr::KB : 1];
        if (HOLD0 && wave < (Cp >> 5)) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kb = 0; kb < D / Pr::KB; ++kb) {
                    w1f[t][kb] = ld_frag_global_u(p_w1n, (long)(2 * wave + t) * (D / Pr::KB) + kb, lane16_0);
                    w2f[t][kb] = ld_frag_global_u(p_w2tn, (long)(2 * wave + t) * (D / Pr::KB) + kb, lane16_0);
                }
        }
        float* const sl_b2 = slot_of(b);
        _Pragma("unroll 1") for (int d = tb1; d < D; d += NTHREADS) {
            float s_ = 0.f;
#pragma unroll 4
            for (int r = 0; r < BM; ++r) s_ += tdy[r * XLD + d];
            if constexpr (PART) sl_b2[SPP_B2(D) + d] = s_;
            else M2M_SMALL_ATOMIC(reinterpret_cast<float*>((char*)p_gb2) + d, s_);
        }
        pack_tile_chn_t<P, D>(tdy, (char*)p_dyt + pair_off, tile_in_pair, tb1);
        pack_tile_chn_t<P, D>(ta, (char*)p_atc + pair_off, tile_in_pair, tb1);
        if constexpr (HREC) {
            // the packed NAT image of A (this workgroup's 16 rows: [kb][lane] 16 B), straight from LDS: the first operand of
            // the weight-gradient launch's recompute of Hpre = A W1^T
            if (tb1 < IMG_B / 16)
                *reinterpret_cast<M2M_AS1 u32x4_t*>(p_h + (long)wg * IMG_B + tb1 * 16) = *reinterpret_cast<const u32x4_t*>(at + tb1 * 16);
        }

        // (C3) hidden-column loop
        f32x4_t dacc[MT][DT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dacc[mt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int npairs = Cp >> 5;
        // The loop's six streams as scalar (SGPR) base pointers behind an opaque asm: under register pressure hipcc otherwise
        // re-reads them from the descriptor INSIDE the loop (two global_load_dwordx2 + vmcnt(0) per step, which drains the
        // weight prefetch in flight).
        asm volatile("" : "+s"(p_w1n), "+s"(p_w2tn), "+s"(p_w1tc), "+s"(p_dh), "+s"(p_h));
        const unsigned int lane16 = (unsigned int)lane * 16u;
        // identity block of the transposing MFMA (bf16): lane (g, il) is non-zero iff g == il >> 2, at element il & 3
        const unsigned int id_sel = (g == (il >> 2)) ? ((il & 1) ? 0x3F800000u : 0x00003F80u) : 0u;
        const unsigned int id_a = (il & 2) ? 0u : id_sel, id_b = (il & 2) ? id_sel : 0u;
        // hidden_dim <= 128: the W1 / W2^T fragments of a whole step (2 x 2 x KD) live in registers and the next step's are
        // requested during this one's epilogue.  hidden_dim 256: that is 128 registers next to 64 accumulators and the 64 of
        // the third product -- the kernel spilled ~200 -- and such towers have few steps per wave (C = 512: two), so there the
        // first two products stream their fragments two k-blocks at a time, nothing held across steps.
        constexpr bool HOLD = D <= 128;
        // bf16, hidden_dim <= 128: the third product takes its W1 operand (k = hidden column) from THIS step's W1 fragments,
        // parked in the wave's LDS slot as 32 rows of 256 bytes and read back transposed (ds_read_b64_tr_b16): the W1^T copy is
        // not streamed at all (a third less traffic on the CU's vector-memory path, which paces this loop) and its 32
        // registers are free.  Rows are XOR-swizzled by 16-byte chunk (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) so that
        // both the 16-byte fragment writes and the 8-byte transposed reads spread over the banks.  The slot lies inside the
        // wave's own slab (dead until the wave leaves the loop).
#ifndef M2M_W1LDS
#define M2M_W1LDS 1
#endif
#ifndef M2M_TICKETS
#define M2M_TICKETS 1
#endif
        constexpr bool W1LDS = M2M_W1LDS && D == 128 && P == PREC_BF16;      // (256-byte rows: hidden_dim 128)
        char* w1slot = reinterpret_cast<char*>(slabs + wave * TILE_F);
        // physical chunk = logical chunk ^ swz(row), swz(r) = (r & 7) | ((r & 1) << 3): its low three bits are a bijection of
        // r & 7 (the 8 lanes one ds_write_b128 cycle serves are 8 consecutive rows at one logical chunk: 8 distinct chunks mod
        // 128 bytes) and so are its high three bits (the 32 lanes one ds_read_b64_tr_b16 cycle serves are 8 consecutive rows x
        // both chunks of one 32-byte granule: 16 distinct chunks = all 64 banks).  The first swizzle of this slot
        // ([il1 il0 il3 il2]) was two-way conflicted on both sides: the parked W1 was half of the kernel's LDS-array cycles,
        // half of those conflicts (SQ_LDS_BANK_CONFLICT with / without the slot).
        const int swz_w = (il & 7) | ((il & 1) << 3);                              // writer: row 16 t + il, chunk 4 kb + g
        char* w1_wr = w1slot + 256 * il + 16 * (g ^ (swz_w & 3));
        const int rrow = 4 * g + (il >> 2);                                        // reader: row 16 t + 4 g + (il >> 2)
        const int swz_r = (rrow & 7) | ((rrow & 1) << 3);
        const char* w1_rd = w1slot + 256 * rrow + 8 * (il & 1) + 16 * (((il >> 1) & 1) ^ (swz_r & 1));
        // bf16 training: the 32-column steps are handed out by a ticket counter in LDS (the first NWAVES statically).  The loop
        // is issue-bound and the waves do not run at the same pace (the older ones win the arbitration): with a static split
        // the fastest wave waited ~4 us per block at the barrier behind the loop.  The next ticket is drawn at the top of a step
        // and used at its prefetch point.  fp32 (parity) mode keeps the static split: reproducible summation order.
        // Default since round 3: the STATIC split (g_bwd_static_split = 1): the column sums of dA have a fixed order.  The ticket
        // counter (M2M_BWD_TICKETS=1 in the environment, read before the first launch) gained 2.6 % in round 2; in round 3, with
        // the small-gradient atomics gone, three A/B runs on three boxes showed no difference.
        constexpr bool TICKETS_CT = P == PREC_BF16 && M2M_TICKETS;
        const bool TICKETS = TICKETS_CT && __builtin_amdgcn_readfirstlane(g_bwd_static_split) == 0;
        // The two waves of a SIMD (w and w + 4) run the same program and leave the barrier before the loop together: their MFMA
        // clusters and their VALU epilogues would collide.  M2M_STAGGER delays waves 4-7 by about half a step (s_sleep counts 64
        // cycles) so that one wave's matrix work lies beside its partner's vector work; with tickets the delayed waves simply
        // draw fewer steps, the stagger costs nothing at the end.  M2M_PRIO_STATIC: the younger half at priority 1 for the
        // loop; M2M_PRIO_FLIP: priority 1 around each step's MFMA cluster.
#ifndef M2M_STAGGER
#define M2M_STAGGER 0
#endif
#ifndef M2M_PRIO_STATIC
#define M2M_PRIO_STATIC 0
#endif
#ifndef M2M_PRIO_FLIP
#define M2M_PRIO_FLIP 0
#endif
        if (TICKETS && (M2M_STAGGER > 0 || M2M_PRIO_STATIC) && __builtin_amdgcn_readfirstlane(wave) >= NWAVES / 2) {
            if (M2M_PRIO_STATIC) __builtin_amdgcn_s_setprio(1);
            if (M2M_STAGGER > 0) __builtin_amdgcn_s_sleep(M2M_STAGGER);
        }
        TIMER_CRESET();
        // ---- staged form of the step (round 4; bf16, hidden_dim 128, table activation) -----------------------------------
        // The ISA of the plain form below showed every one of a step's 16 table look-ups (ds_read_b64) and every one of the third
        // product's 8 transposed operand reads followed at once by s_waitcnt lgkmcnt(0): 24 fully exposed LDS round trips per
        // step although ~80 VGPRs were unused in the loop (the pre-RA scheduler sees the region at its pressure limit and sinks
        // every load to its use), plus a burst of 16 weight loads that blocks the in-order wave until the texture addresser has
        // taken them (in-kernel stamps, scripts/bwd_loop_stamps.py: of ~3400 cycles per step 990 went to waiting for the weights,
        // 490 to issuing the prefetch, 700 to the third product's 8 MFMAs).  Here the step is written as ISSUE and CONSUME stages
        // separated by scheduling barriers, so that nothing is moved back to its use: the weight prefetch in two halves around
        // the table reads, all 16 table reads in flight together, the third product's 16 transposed reads and the next step's
        // bias requested before the transposing MFMAs / packs / stores and consumed after them.  Arithmetic and summation order
        // are those of the plain form (bit-identical results).
#ifndef M2M_BWD_STAGED
#define M2M_BWD_STAGED 1
#endif
        constexpr bool STAGED = M2M_BWD_STAGED && HOLD && W1LDS && MT == 1 && ActB<P>::USES_TABLE && M2M_BWD_HTAB && NF == 1;
        f32x4_t bias_n[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};      // hidden bias of the step about to run
        if constexpr (STAGED) {
            const int q0 = min(TICKETS ? (int)__builtin_amdgcn_readfirstlane(wave) : wave, npairs - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) bias_n[t] = *reinterpret_cast<const f32x4_t*>(bias_s + 32 * q0 + 16 * t + 4 * g);
        }
        // (the step index as a scalar in both split modes: the weight streams are then addressed scalar base + lane offset and the
        //  "more steps" test is a scalar branch; as a per-lane value it cost 64-bit vector address arithmetic per load group)
        for (int q = __builtin_amdgcn_readfirstlane(wave); q < npairs;) {
            unsigned int ticket = 0u;
            if (TICKETS && lane == 0) ticket = atomicAdd(qctr, 1u);
            if constexpr (STAGED) {
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
                // (A) this step's weights have landed: park W1 for the third product, products 1 and 2
                f32x4_t hacc[2] = {bias_n[0], bias_n[1]};
                f32x4_t gacc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
                TIMER_CMARK(8);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kb = 0; kb < KD; ++kb)
                        *reinterpret_cast<u32x4_t*>(w1_wr + 4096 * t + 16 * ((4 * kb) ^ (swz_w & 12))) = w1f[t][kb].u;
                // The next step's fragments are requested k-block by k-block right behind the four MFMAs that read this step's (an MFMA
                // takes its operands when it issues; the load lands hundreds of cycles later): the texture addresser takes 16 cycles
                // per 1 KiB load, the matrix pipe 16 per MFMA -- the loads' issue (which blocks the in-order wave while the
                // addresser's queue is full: 490 cycles as one burst behind the products) lies under the products, and the prefetch
                // distance grows by the length of this stage.  MEASURED (three interleaved repetitions in one process): no gain,
                // 0.5074-0.5120 against 0.5066-0.5079 ms per step for the burst in two halves (stages B1 / B3) -- with the
                // round trips gone the loop runs at ~72 % of the texture addresser's rate (18 KiB per wave and step at 64 B/clk:
                // 2300 of ~3200 cycles per round) and the waves queue behind each other whatever the order.  Off.
#ifndef M2M_BWD_PF_INTERLEAVE
#define M2M_BWD_PF_INTERLEAVE 0
#endif
                const int qn = __builtin_amdgcn_readfirstlane(TICKETS ? (int)ticket : q + NWAVES);
                const bool more = qn < npairs;
#pragma unroll
                for (int kb = 0; kb < KD; ++kb) {
                    const Frag a = ld_frag_lds(at, kb, lane);
                    const Frag dy = ld_frag_lds(dyp, kb, lane);
                    Pr::mma(hacc[0], w1f[0][kb], a);
                    Pr::mma(hacc[1], w1f[1][kb], a);
                    Pr::mma(gacc[0], w2f[0][kb], dy);
                    Pr::mma(gacc[1], w2f[1][kb], dy);
                    if (M2M_BWD_PF_INTERLEAVE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) {
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                w1f[t][kb] = ld_frag_global_u(p_w1n, (long)(2 * qn + t) * KD + kb, lane16);
                                w2f[t][kb] = ld_frag_global_u(p_w2tn, (long)(2 * qn + t) * KD + kb, lane16);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                TIMER_CMARK(9);
                // (B1) first half of the next step's weights (W1: the park writes need it first) + this step's keep-word: both
                //      independent of the products still in the matrix pipe
                if (!M2M_BWD_PF_INTERLEAVE && more) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int kb = 0; kb < KD; ++kb) w1f[t][kb] = ld_frag_global_u(p_w1n, (long)(2 * qn + t) * KD + kb, lane16);
                }
                const unsigned int word = drop_hidden_bits<DM>(dr_ch, (unsigned int)(row0 + il), q, Cp) >> (4 * g);
                __builtin_amdgcn_sched_barrier(0);
                // (B2) the 16 table cells of this lane's hidden elements: indices, then all reads in flight together
                gtabB_t e[2][4];
                {
                    unsigned int idx[2][4];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            idx[t][r] = pwl_index(hacc[t][r]);
                            if (DM != DM_NONE) idx[t][r] &= bit_to_mask(word, 16 * t + r);     // dropped: cell 0 = zeros
                        }
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) e[t][r] = gtab[idx[t][r]];
                }
                __builtin_amdgcn_sched_barrier(0);
                // (B3) second half of the weight prefetch: its issue lies beside the table reads' latency
                if (!M2M_BWD_PF_INTERLEAVE && more) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int kb = 0; kb < KD; ++kb) w2f[t][kb] = ld_frag_global_u(p_w2tn, (long)(2 * qn + t) * KD + kb, lane16);
                }
                __builtin_amdgcn_sched_barrier(0);
                TIMER_CMARK(10);
                // (C) gelu / gelu' (both carry the dropout scale), dHpre = dHact * gelu', chained operand fragments
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = hacc[t][r];
                        const float gl = __builtin_fmaf((float)e[t][r][1], x, (float)e[t][r][0]);
                        const float dgl = __builtin_fmaf((float)e[t][r][3], x, (float)e[t][r][2]);
                        gacc[t][r] *= dgl;
                        hacc[t][r] = gl;
                    }
                Frag hf, af;
                {
                    Frag tmp[1];
                    Chain<P>::make(gacc[0], gacc[1], tmp);
                    hf = tmp[0];
                    if constexpr (!HREC) { Chain<P>::make(hacc[0], hacc[1], tmp); af = tmp[0]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                TIMER_CMARK(11);
                // (D0) requests consumed two stages on: the third product's W1 operand (transposed reads of the parked fragments)
                //      and the next step's hidden bias
                Frag wf[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* pr = w1_rd + 16 * ((2 * dt) ^ (swz_r & 14));
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(pr));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(pr + 4096));
                    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    wf[dt].u = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
                }
                {
                    const int qb = more ? qn : q;
#pragma unroll
                    for (int t = 0; t < 2; ++t) bias_n[t] = *reinterpret_cast<const f32x4_t*>(bias_s + 32 * qb + 16 * t + 4 * g);
                }
                __builtin_amdgcn_sched_barrier(0);
                // (D) operands of the weight-gradient pass: dHpre^T and Hact^T through the identity MFMA (see the plain form)
                {
                    const int u = tile_in_pair;
                    const long npair = (nwg + TPP - 1) / TPP, pair = wg / TPP;
                    const s16x4 id16 = __builtin_bit_cast(s16x4, u32x2{id_a, id_b});
                    f32x4_t od[2], oa[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        od[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, u32x2{hf.u[2 * t], hf.u[2 * t + 1]}), id16,
                                                                          f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        if constexpr (!HREC)
                            oa[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, u32x2{af.u[2 * t], af.u[2 * t + 1]}), id16,
                                                                              f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    }
                    const long off = (long)q * m2m_hchn_stride(npair) + (pair * 2 + u) * 1024 + lane * 16;
                    const u32x4_t sd = u32x4_t{pack_bf2(od[0][0], od[0][1]), pack_bf2(od[0][2], od[0][3]),
                                               pack_bf2(od[1][0], od[1][1]), pack_bf2(od[1][2], od[1][3])};
                    __builtin_nontemporal_store(sd, reinterpret_cast<M2M_AS1 u32x4_t*>(p_dh + off));
                    if constexpr (!HREC) {
                        const u32x4_t sa = u32x4_t{pack_bf2(oa[0][0], oa[0][1]), pack_bf2(oa[0][2], oa[0][3]),
                                                   pack_bf2(oa[1][0], oa[1][1]), pack_bf2(oa[1][2], oa[1][3])};
                        __builtin_nontemporal_store(sa, reinterpret_cast<M2M_AS1 u32x4_t*>(p_h + off));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                TIMER_CMARK(12);
                // (E) third product: dA += dHpre W1
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) Pr::mma(dacc[0][dt], hf, wf[dt]);
                TIMER_CMARK(13);
                q = qn;
                continue;
            }
            f32x4_t bias[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const f32x4_t*>(bias_s + 32 * q + 16 * t + 4 * g);
            f32x4_t hacc[MT][2], gacc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                hacc[mt][0] = bias[0];
                hacc[mt][1] = bias[1];
                gacc[mt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                gacc[mt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            TIMER_CMARK(8);    // loop step head: ticket, bias, accumulator init (+ previous step's tail)
            if constexpr (HOLD) {
            if (M2M_PRIO_FLIP) __builtin_amdgcn_s_setprio(1);
            if constexpr (W1LDS) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kb = 0; kb < KD; ++kb)
                        *reinterpret_cast<u32x4_t*>(w1_wr + 4096 * t + 16 * ((4 * kb) ^ (swz_w & 12))) = w1f[t][kb].u;
            }
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const Frag a = ld_frag_lds(at, mt * KD + kb, lane);
                    const Frag dy = ld_frag_lds(dyp, mt * KD + kb, lane);
                    Pr::mma(hacc[mt][0], w1f[0][kb], a);
                    Pr::mma(hacc[mt][1], w1f[1][kb], a);
                    Pr::mma(gacc[mt][0], w2f[0][kb], dy);
                    Pr::mma(gacc[mt][1], w2f[1][kb], dy);
                }
            }
            } else {
                constexpr int KG = 2;
#pragma unroll
                for (int k0 = 0; k0 < KD; k0 += KG) {
                    Frag u1[2][KG], u2[2][KG];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int kk = 0; kk < KG; ++kk) {
                            u1[t][kk] = ld_frag_global_u(p_w1n, (long)(2 * q + t) * KD + k0 + kk, lane16);
                            u2[t][kk] = ld_frag_global_u(p_w2tn, (long)(2 * q + t) * KD + k0 + kk, lane16);
                        }
#pragma unroll
                    for (int kk = 0; kk < KG; ++kk) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const Frag a = ld_frag_lds(at, mt * KD + k0 + kk, lane);
                            const Frag dy = ld_frag_lds(dyp, mt * KD + k0 + kk, lane);
                            Pr::mma(hacc[mt][0], u1[0][kk], a);
                            Pr::mma(hacc[mt][1], u1[1][kk], a);
                            Pr::mma(gacc[mt][0], u2[0][kk], dy);
                            Pr::mma(gacc[mt][1], u2[1][kk], dy);
                        }
                    }
                }
            }
            // this step's W1^T fragments (third product) and the next step's W1 / W2^T fragments: in flight
            // during the epilogue.  The scheduling barrier keeps the compiler from hoisting these loads above
            // the MFMAs that still read the current fragments (which would double the live registers).
            __builtin_amdgcn_sched_barrier(0);
            TIMER_CMARK(9);    // wait for this step's weights, W1 park writes, A / dYd fragment reads, products 1-2 issued
            if (M2M_PRIO_FLIP) __builtin_amdgcn_s_setprio(0);
            Frag w3f[W1LDS ? 1 : NF][W1LDS ? 1 : DT];
            if constexpr (!W1LDS) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) w3f[f][dt] = ld_frag_global_u(p_w1tc, (long)(q * NF + f) * DT + dt, lane16);
            }
            const int qn = __builtin_amdgcn_readfirstlane(TICKETS ? (int)ticket : q + NWAVES);
#ifdef M2M_ABL_NOPF
            if (false) {
#else
            if (HOLD && qn < npairs) {
#endif
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kb = 0; kb < KD; ++kb) {
                        w1f[t][kb] = ld_frag_global_u(p_w1n, (long)(2 * qn + t) * KD + kb, lane16);
                        w2f[t][kb] = ld_frag_global_u(p_w2tn, (long)(2 * qn + t) * KD + kb, lane16);
                    }
            }
            TIMER_CMARK(10);   // prefetch issue (next step's weights)
            Frag hf[MT][NF], af[MT][NF];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned int m = (unsigned int)(row0 + mt * 16 + il);
                const unsigned int word = drop_hidden_bits<DM>(dr_ch, m, q, Cp) >> (4 * g);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float gl, dgl;                                   // both carry the dropout scale
#ifdef M2M_ABL_NOGELU
                        gacc[mt][t][r] *= 0.5f; hacc[mt][t][r] = mask_f(hacc[mt][t][r], bit_to_mask(word, 16 * t + r)); continue;
#endif
                        if constexpr (DM != DM_NONE && ActB<P>::USES_TABLE && M2M_BWD_HTAB) {
                            gelu_grad_tabh_masked(gtab, hacc[mt][t][r], bit_to_mask(word, 16 * t + r), gl, dgl);
                            gacc[mt][t][r] *= dgl;
                            hacc[mt][t][r] = gl;
                            continue;
                        }
                        ActB<P>::gelu_grad_scaled(gtab, hacc[mt][t][r], dr_ch.scale, gl, dgl);
                        const float v = gacc[mt][t][r] * dgl;
                        if (DM == DM_NONE) { gacc[mt][t][r] = v; hacc[mt][t][r] = gl; }
                        else {
                            const unsigned int mk = bit_to_mask(word, 16 * t + r);
                            gacc[mt][t][r] = mask_f(v, mk);
                            hacc[mt][t][r] = mask_f(gl, mk);
                        }
                    }
                }
                Chain<P>::make(gacc[mt][0], gacc[mt][1], hf[mt]);     // dHpre: operand of dA += dHpre W1
                if constexpr (!HREC) Chain<P>::make(hacc[mt][0], hacc[mt][1], af[mt]);     // Hact : only stored, for the weight gradients
            }
            // Operands of the weight-gradient pass: dHpre^T and Hact^T with k = token row.  The accumulators hold
            // [c in registers][m across lanes]; an MFMA against an identity block turns them ([m][c] as the A operand,
            // chained k order) into [m in registers][c across lanes] = exactly the layout that pass consumes, exact in
            // the operand precision.  The MFMA pipe is mostly idle here, so the transpose is nearly free.
            TIMER_CMARK(11);   // epilogue: products 1-2 complete, keep-words, table, chained fragments
#ifndef M2M_ABL_NOTR
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int u = BM == 16 ? tile_in_pair : mt;        // 16-row half of the 32-row pair
                const long npair = (nwg + TPP - 1) / TPP, pair = wg / TPP;
                f32x4_t od[2], oa[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    od[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    oa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if constexpr (P == PREC_BF16) {
                        // half t of the chained fragment = accumulator tile t, rows 4g + j: the natural k order of the
                        // 16x16x16 form, so ONE two-register identity serves both halves (the 16x16x32 form needs two
                        // four-register selectors, and the column loop has no registers to spare)
                        typedef short s16x4 __attribute__((ext_vector_type(4)));
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        const s16x4 id16 = __builtin_bit_cast(s16x4, u32x2{id_a, id_b});
                        od[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, u32x2{hf[mt][0].u[2 * t], hf[mt][0].u[2 * t + 1]}), id16, od[t], 0, 0, 0);
                        if constexpr (!HREC) oa[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, u32x2{af[mt][0].u[2 * t], af[mt][0].u[2 * t + 1]}), id16, oa[t], 0, 0, 0);
                    } else {
                        Frag id;
                        id.f = f32x4_t{4 * g + 0 == il ? 1.f : 0.f, 4 * g + 1 == il ? 1.f : 0.f, 4 * g + 2 == il ? 1.f : 0.f, 4 * g + 3 == il ? 1.f : 0.f};
                        Pr::mma(od[t], hf[mt][t], id);
                        if constexpr (!HREC) Pr::mma(oa[t], af[mt][t], id);
                    }
                    // od[t][r] = dHpre[m = 16u + 4g + r][c = 32q + 16t + il]
                }
                // streamed out once and read once by the weight-gradient pass: non-temporal, so that the 100 MB per
                // launch do not evict the weights the other workgroups of this XCD keep re-reading from its L2
#ifdef M2M_ABL_NOSTORE
                asm volatile("" :: "v"(od[0]), "v"(od[1]), "v"(oa[0]), "v"(oa[1]));
                continue;
#endif
                if (P == PREC_BF16) {
                    // [column-tile pair q][32-row pair][16-row half u][lane][tile 2q: 4 bf16 | tile 2q+1: 4 bf16]: one full
                    // 16-byte-per-lane store per operand and step (1 KiB contiguous), and the weight-gradient wave that owns
                    // both column tiles reads it back with one 16-byte load per half
                    const long off = (long)q * m2m_hchn_stride(npair) + (pair * 2 + u) * 1024 + lane * 16;
#ifndef M2M_ST_NT
#define M2M_ST_NT 1
#endif
                    const u32x4_t sd = u32x4_t{pack_bf2(od[0][0], od[0][1]), pack_bf2(od[0][2], od[0][3]),
                                               pack_bf2(od[1][0], od[1][1]), pack_bf2(od[1][2], od[1][3])};
                    if (M2M_ST_NT) __builtin_nontemporal_store(sd, reinterpret_cast<M2M_AS1 u32x4_t*>(p_dh + off));
                    else *reinterpret_cast<M2M_AS1 u32x4_t*>(p_dh + off) = sd;
                    if constexpr (!HREC) {
                        const u32x4_t sa = u32x4_t{pack_bf2(oa[0][0], oa[0][1]), pack_bf2(oa[0][2], oa[0][3]),
                                                   pack_bf2(oa[1][0], oa[1][1]), pack_bf2(oa[1][2], oa[1][3])};
                        if (M2M_ST_NT) __builtin_nontemporal_store(sa, reinterpret_cast<M2M_AS1 u32x4_t*>(p_h + off));
                        else *reinterpret_cast<M2M_AS1 u32x4_t*>(p_h + off) = sa;
                    }
                } else {
                    // fp32: a 16-row half is a whole k-block: [column tile][32-row pair][half][lane][16 bytes]
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const long off = (long)(2 * q + t) * m2m_hchn_stride(npair) + (pair * 2 + u) * 1024 + lane * 16;
                        __builtin_nontemporal_store(od[t], reinterpret_cast<M2M_AS1 f32x4_t*>(p_dh + off));
                        if constexpr (!HREC) __builtin_nontemporal_store(oa[t], reinterpret_cast<M2M_AS1 f32x4_t*>(p_h + off));
                    }
                }
            }
#endif   // M2M_ABL_NOTR
            TIMER_CMARK(12);   // transposing MFMAs, packs, operand stores
#ifdef M2M_ABL_NOP3
            asm volatile("" :: "v"(hf[0][0].u));
            q = qn;
            continue;
#endif
            if constexpr (W1LDS) {
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    // B[k = hidden column, chained order][n = 16 dt + il]: rows 4g .. 4g+3 of tile 0, then of tile 1
                    const char* pr = w1_rd + 16 * ((2 * dt) ^ (swz_r & 14));
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(pr));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(pr + 4096));
                    Frag wf;
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    wf.u = u32x4_t{l2[0], l2[1], h2[0], h2[1]};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) Pr::mma(dacc[mt][dt], hf[mt][0], wf);
                }
            } else {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) Pr::mma(dacc[mt][dt], hf[mt][f], w3f[f][dt]);
            }
            TIMER_CMARK(13);   // third product (transposed reads of the parked W1 + MFMAs)
            q = qn;
        }
        if (TICKETS && M2M_PRIO_STATIC) __builtin_amdgcn_s_setprio(0);
        TIMER_CMARK(14);   // (loop exit)
        TIMER_LMARK(2);   // C3 hidden-column loop (wave 0)
        int tb2 = tid;
        asm volatile("" : "+v"(tb2));
        const int r2 = tb2 / TPR, j2 = tb2 % TPR;
        // rows of x_mid (LayerNorm-2 backward) and x_in (LayerNorm-1 recompute): requested now, used behind the next barrier
        float xm2[EPT], xi[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) { xm2[e] = 0.f; xi[e] = 0.f; }
        if (r2 < R) {
            ld_row<D>(bk.x_mid + (row0 + r2) * D, j2, xm2);
            if (TOK) ld_row<D>(bk.x_in + (row0 + r2) * D, j2, xi);
        }
        // (C4) the waves' partial dA -> row-major slabs (nothing else lives there: no barrier in front)
        if (NSL == NWAVES || wave < NSL) acc_to_slab<D>(dacc, slabs + (wave % NSL) * TILE_F, g, il);
        __syncthreads();
        if (NSL < NWAVES) {
            if (wave >= NSL) acc_add_slab<D>(dacc, slabs + (wave % NSL) * TILE_F, g, il);
            __syncthreads();
        }
        TIMER_LMARK(3);   // C4: slabs (wave 0 waits here for the other waves' loops)
        // (R1) on the row thread's registers: dA = sum of the slabs; LayerNorm-2 backward, dx_mid = dY + LN2'(dA); sources of
        //      the LN2 parameter gradients; token path: LayerNorm-1 recompute (U -> ub, xhat / rstd stay in registers for
        //      R3), dO' = dropout'(dx_mid) and the keep-words of the hidden site
        float xh1[EPT], rstd1 = 0.f;
        {
            float dA[EPT], dx[EPT], mean, rstd;
            slab_row_sum<D>(slabs, r2, j2, dA);
            reg_stats<D>(xm2, mean, rstd);
            float gg[EPT], xh2[EPT], gsum = 0.f, gxsum = 0.f;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int c = ln_col<D>(e, j2);
                xh2[e] = (xm2[e] - mean) * rstd;
                gg[e] = dA[e] * pb[O_LN2W + c];
                gsum += gg[e];
                gxsum = __builtin_fmaf(gg[e], xh2[e], gxsum);
            }
            gsum = wave_sum_xor(gsum, TPR) * (1.0f / D);
            gxsum = wave_sum_xor(gxsum, TPR) * (1.0f / D);
            ld_row<D>(dxs + r2 * XLD, j2, dx);
            float pr[EPT], up[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const bool valid = r2 < R;
                if (valid) dx[e] += rstd * (gg[e] - gsum - xh2[e] * gxsum);
                pr[e] = valid ? dA[e] * xh2[e] : 0.f;
                up[e] = valid ? dA[e] : 0.f;
            }
            st_row<D>(dxs + r2 * XLD, j2, dx);
            st_row<D>(t_prod + r2 * XLD, j2, pr);
            st_row<D>(t_up + r2 * XLD, j2, up);
            if constexpr (TOK) {
                float u[EPT], mean1;
                reg_stats<D>(xi, mean1, rstd1);
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const int c = ln_col<D>(e, j2);
                    xh1[e] = (xi[e] - mean1) * rstd1;
                    u[e] = xh1[e] * pb[O_LN1W + c] + pb[O_LN1B + c];
                }
                st_row<D>(ub + r2 * XLD, j2, u);
                if constexpr (P == PREC_BF16) {
                    float o[EPT];
                    const int sl = r2 / N, n = r2 % N;
#pragma unroll
                    for (int e = 0; e < EPT; ++e) {
                        const int c = ln_col<D>(e, j2);
                        const bool keep = r2 < R && drop_row_keep<DM>(dr_to, (unsigned int)(s0 + sl) * D + c, (unsigned int)N, (unsigned int)n);
                        o[e] = keep ? dx[e] * dr_to.scale : 0.f;
                    }
                    st_row<D>(dov + r2 * XLD, j2, o);
                    if (DM != DM_NONE) {
                        _Pragma("unroll 1") for (int p = tb2; p < SPW * D; p += NTHREADS) {
                            const int slp = p / D;
                            wth[p] = slp < ns ? drop_row_bits<DM>(dr_th, (unsigned int)(s0 + slp) * D + p % D, T) : 0u;
                        }
                    }
                }
            }
        }
        __syncthreads();
        TIMER_LMARK(4);   // R1: slab sum, LN2 backward, LN1 recompute, token operands
        // (R2) LayerNorm-2 parameter gradients (column sums of t_prod / t_up) beside the token MLP backward
        int tb3 = tid;
        asm volatile("" : "+v"(tb3));
        const int lane3 = tb3 & 63;
        if constexpr (!TOK) {
            { float* sl = slot_of(b); if constexpr (PART) colsums(t_prod, t_up, sl + SPP_LN2, sl + SPP_LN2 + D, tb3, true); else colsums(t_prod, t_up, bk.g_ln2_w, bk.g_ln2_b, tb3); }
        } else {
        // ================= token mixing backward =================
        const float* tokw = pb + O_TOKW;
        if constexpr (P == PREC_BF16) {
            token_bwd_mfma<D, NM, DM>(ub, dov, tokw, gtab, wth, red, N, ns, dr_th.scale, wave, lane3);
            { float* sl = slot_of(b); if constexpr (PART) colsums(t_prod, t_up, sl + SPP_LN2, sl + SPP_LN2 + D, tb3, true); else colsums(t_prod, t_up, bk.g_ln2_w, bk.g_ln2_b, tb3); }
            __syncthreads();
            TIMER_LMARK(5);   // R2: token MLP backward (MFMA form) + LN2 column sums
            // sum the waves' partial token-weight gradients (layout: TokRed), ONE global atomic per value per workgroup
            const int nred = 2 * T * N + T + N;
            for (int i = tb3; i < nred; i += NTHREADS) {
                int slot;
                float* dst;
                if (i < T * N)              { slot = (i % N) * 32 + i / N; dst = bk.g_tok_w1 + i; }
                else if (i < 2 * T * N)     { const int j = i - T * N; slot = (NM + 1 + j / T) * 32 + j % T; dst = bk.g_tok_w2 + j; }
                else if (i < 2 * T * N + T) { const int t = i - 2 * T * N; slot = N * 32 + t; dst = bk.g_tok_b1 + t; }
                else                        { const int n = i - 2 * T * N - T; slot = 2 * (NM + 1) * 32 + n; dst = bk.g_tok_b2 + n; }
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVES; ++w) v += red[w * TokRed<NM>::LD + slot];
                float* sl = slot_of(b);
                if constexpr (PART) sl[SPP_TOK(D) + i] = v;
                else M2M_SMALL_ATOMIC(dst, v);
            }
        } else {
            { float* sl = slot_of(b); if constexpr (PART) colsums(t_prod, t_up, sl + SPP_LN2, sl + SPP_LN2 + D, tb3, true); else colsums(t_prod, t_up, bk.g_ln2_w, bk.g_ln2_b, tb3); }
            constexpr int TTMAX = 32 / TG;                 // hidden units per lane (T <= 32)
            const int tg = tb3 % TG, pl = tb3 / TG;        // TG lanes share a column and split its T hidden units
            const int TT = T / TG;
            float w1r[TTMAX][NMAX], w2r[NMAX][TTMAX], b1r[TTMAX];
            float aw1[TTMAX][NMAX], aw2[NMAX][TTMAX], ab1[TTMAX], ab2[NMAX];
            // this lane's hidden units from the zero-padded LDS copy of the token weights (rows t >= T and
            // columns n >= N are zero, so unused slots contribute nothing and need no guards)
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
                const unsigned int wth = drop_row_bits<DM>(dr_th, bd, T);     // keep-bits (all ones when dropout is off)
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
                // all TTMAX slots are computed unconditionally (weights of unused slots are zero, so they add
                // nothing): straight-line code lets the TTMAX independent chains overlap their LDS latencies
#pragma unroll
                for (int tt = 0; tt < TTMAX; ++tt) {
                    {
                        const int t = tg * TT + tt;
                        float h = b1r[tt], dh = 0.f;
#pragma unroll
                        for (int n = 0; n < NMAX; ++n) {
                            h = __builtin_fmaf(w1r[tt][n], un[n], h);
                            dh = __builtin_fmaf(w2r[n][tt], dv[n], dh);
                        }
                        float gl, dgl;                                   // both carry the dropout scale
                        ActB<P>::gelu_grad_scaled(gtab, h, dr_th.scale, gl, dgl);
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
            // reduce the token-weight gradients: over the columns a wave handles concurrently (VALU cross-lane
            // sums), over the 8 waves (per-wave LDS slots written in REGISTER order -- slot (k, tg) at k*TG + tg
            // from one base address, so no per-value address arithmetic stays live), then ONE global atomic
            // per value per workgroup (the same ~300 addresses are hit by every workgroup of the launch).
            //   k = tt*KS: db1[t] | tt*KS + 1 + n: dW1[t][n] | tt*KS + 1 + NMAX + n: dW2[n][t] | TTMAX*KS + n: db2[n]
            constexpr int KS = 1 + 2 * NMAX;
            float* myred = red + wave * RED_LD + tg;          // lanes >= TG of a wave never store
#pragma unroll
            for (int tt = 0; tt < TTMAX; ++tt) {
                const float s = lane_class_sum(ab1[tt], TG);
                if (lane < TG) myred[(tt * KS) * TG] = s;
#pragma unroll
                for (int n = 0; n < NMAX; ++n) {
                    const float a = lane_class_sum(aw1[tt][n], TG), c = lane_class_sum(aw2[n][tt], TG);
                    if (lane < TG) {
                        myred[(tt * KS + 1 + n) * TG] = a;
                        myred[(tt * KS + 1 + NMAX + n) * TG] = c;
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                const float s = lane_class_sum(ab2[n], TG);   // non-zero on tg == 0 lanes only
                if (lane < TG) myred[(TTMAX * KS + n) * TG] = s;
            }
            __syncthreads();
            const int nred = 2 * T * N + T + N;
            for (int i = tb3; i < nred; i += NTHREADS) {
                int t, k;
                float* dst;
                if (i < T * N)              { t = i / N; k = 1 + i % N; dst = bk.g_tok_w1 + i; }
                else if (i < 2 * T * N)     { const int j = i - T * N; t = j % T; k = 1 + NMAX + j / T; dst = bk.g_tok_w2 + j; }
                else if (i < 2 * T * N + T) { t = i - 2 * T * N; k = 0; dst = bk.g_tok_b1 + t; }
                else                        { t = -1; k = i - 2 * T * N - T; dst = bk.g_tok_b2 + k; }
                const int slot = t < 0 ? (TTMAX * KS + k) * TG : ((t % TT) * KS + k) * TG + t / TT;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVES; ++w) v += red[w * RED_LD + slot];
                float* sl = slot_of(b);
                if constexpr (PART) sl[SPP_TOK(D) + i] = v;
                else M2M_SMALL_ATOMIC(dst, v);
            }
        }
        // (R3) LayerNorm-1 backward on the row thread's registers: dx_in = dx_mid + LN1'(dU); the sources of the LN1 parameter
        //      gradients go to t_prod (dU * xhat) and stay in ub (dU): column-summed during the next block's first phase
        {
            const int r = tb3 / TPR, j = tb3 % TPR;
            float du[EPT], dx[EPT], gg[EPT], gsum = 0.f, gxsum = 0.f;
            ld_row<D>(ub + r * XLD, j, du);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int c = ln_col<D>(e, j);
                gg[e] = du[e] * pb[O_LN1W + c];
                gsum += gg[e];
                gxsum = __builtin_fmaf(gg[e], xh1[e], gxsum);
            }
            gsum = wave_sum_xor(gsum, TPR) * (1.0f / D);
            gxsum = wave_sum_xor(gxsum, TPR) * (1.0f / D);
            ld_row<D>(dxs + r * XLD, j, dx);
            float pr[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const bool valid = r < R;
                if (valid) dx[e] += rstd1 * (gg[e] - gsum - xh1[e] * gxsum);
                pr[e] = valid ? du[e] * xh1[e] : 0.f;
                if (!valid) du[e] = 0.f;
            }
            st_row<D>(dxs + r * XLD, j, dx);
            st_row<D>(t_prod + r * XLD, j, pr);
            st_row<D>(ub + r * XLD, j, du);
            pend = b;
        }
        }   // TOK
        // the next block's x_mid rows: requested here, used in its first phase
        if (b > 0) {
            const int r = tb3 / TPR, j = tb3 % TPR;
#pragma unroll
            for (int e = 0; e < EPT; ++e) xm[e] = 0.f;
            if (r < R) ld_row<D>(tw.blk[b - 1].x_mid + (row0 + r) * D, j, xm);
            const M2M_AS1 float* b1p = reinterpret_cast<const M2M_AS1 float*>(to_gptr(tw.blk[b - 1].ch_b1p));   // scalar, read once
#pragma unroll
            for (int k = 0; k < BPT; ++k) {
                nb[k] = 0.f;
                if (tb3 + k * NTHREADS < Cp) nb[k] = b1p[tb3 + k * NTHREADS];
            }
        }
        __syncthreads();
        TIMER_LMARK(6);   // R3: token-gradient atomics, LN1 backward
    }
    if (TOK && pend >= 0) {
        float* sl = slot_of(pend);
        if constexpr (PART) colsums(t_prod, ub, sl + SPP_LN1(D), sl + SPP_LN1(D) + D, tid, true);
        else colsums(t_prod, ub, tw.blk[pend].g_ln1_w, tw.blk[pend].g_ln1_b, tid);
    }

    // ---- gradient wrt the tower input ----
    _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
        const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
        const long gr = row0 + r;
        *reinterpret_cast<float4*>(d_x0 + (gr / N) * d_x0_ss + (gr % N) * D + c) =
            *reinterpret_cast<const float4*>(dxs + r * XLD + c);
    }
    // ... and, on request, as packed operand blocks (d_x0^T, k = token row): the first operand of the patch-embedding weight
    // gradient (embed_wgrad.h, fast form); rows >= R of the tile are zero
    if (dx0_chn) pack_tile_chn_t<P, D>(dxs, dx0_chn + pair_off, tile_in_pair, tid);
    TIMER_LFLUSH(g_tm_bwd);
}

template <int P, int D, int NMAX, int TG, int DM, bool PART = false, bool HREC = false>
__global__ __launch_bounds__(NTHREADS) M2M_BWD_KATTR void tower_bwd_kernel(const m2m_tower tw, int B, const float* __restrict__ d_out,
                                                             long d_out_ss, const float* __restrict__ d_pooled,
                                                             float* __restrict__ d_x0, long d_x0_ss, unsigned int seed,
                                                             unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tower_bwd_body<m2m_tower, P, D, NMAX, TG, DM, PART, HREC>(tw, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step_host, step_dev,
                                                               blockIdx.x, gridDim.x, smem, PART ? tw.gpart : nullptr, (char*)tw.dx0_chn);
}
// the same with the classification heads in its prologue (BwdHeads); slot form only
template <int P, int D, int NMAX, int TG, int DM, bool HREC>
__global__ __launch_bounds__(NTHREADS) M2M_BWD_KATTR void tower_bwd_heads_kernel(const m2m_tower tw, const BwdHeads hd, int B, float* __restrict__ d_x0,
                                                                   long d_x0_ss, unsigned int seed, unsigned int step_host,
                                                                   const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tower_bwd_body<m2m_tower, P, D, NMAX, TG, DM, true, HREC, true>(tw, B, nullptr, 0, nullptr, d_x0, d_x0_ss, seed, step_host, step_dev,
                                                                     blockIdx.x, gridDim.x, smem, tw.gpart, (char*)tw.dx0_chn, &hd);
}

// Two towers side by side in ONE launch (blockIdx.y = tower), see tower_fwd.hip.
struct BwdGroupArgs {
    m2m_tower4 tw[2];
    const float* d_out[2];
    long d_out_ss[2];
    const float* d_pooled[2];
    float* d_x0[2];
    long d_x0_ss[2];
    int ntiles[2];
    float* part[2];               // per-workgroup partial-sum slots of the small gradients (m2m_tower.gpart) or NULL: atomics
    char* dx0_chn[2];             // packed image of d_x0^T (m2m_tower.dx0_chn) or NULL
};
static_assert(sizeof(BwdGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");

// Small parameter gradients (LayerNorms, token MLP, ch_b2) through per-workgroup slots + one reduction launch instead of float
// atomics: fused-class towers whose descriptor carries the slot buffer (m2m_tower.gpart: the runtime allocates it for bf16,
// hidden_dim 128).  Timing ablation without these atomics (M2M_ABL_NOATOM): fusion backward -9 us, two-tower backward -7 us --
// 256 (128) workgroups add to the same ~3000 addresses.  With slots the single-tower launch (the fusion tower: 256-way
// contention) nets -3 us including its reduction launch, and its small gradients no longer depend on the order of the atomics.
// M2M_SMALL_PART=0 keeps the atomics everywhere.
static bool m2m_small_part(const m2m_tower* t) {
    static const int off = [] { const char* e = getenv("M2M_SMALL_PART"); return e && e[0] == '0'; }();
    return !off && t->gpart != nullptr && !m2m_is_wide(t) && t->prec == PREC_BF16 && t->D == 128 && t->nblocks >= 1 &&
           2 * t->T * t->N + t->T + t->N <= SPP_TOK_MAX;
}
static void m2m_small_part_reduce_args(SplitReduceTower& x, const m2m_tower* t, int nwg, const BwdHeads* hd = nullptr, float* losses = nullptr) {
    memset(&x, 0, sizeof(x));
    x.part = t->gpart; x.ntiles = nwg; x.nlaunch = t->nblocks + 1; x.D = t->D; x.N = t->N; x.T = t->T;
    if (hd) {                                                // slot sets nblocks + 1 + h: head h (BwdHeads)
        x.head_set0 = t->nblocks + 1; x.nheads = hd->nheads; x.K = hd->K; x.losses = losses;
        for (int h = 0; h < hd->nheads; ++h) { x.g_hw[h] = hd->h[h].g_w; x.g_hb[h] = hd->h[h].g_b; }
        x.nlaunch += hd->nheads;
    }
    if (t->has_final_ln) { x.g_lnf_w = t->g_lnf_w; x.g_lnf_b = t->g_lnf_b; }
    for (int b = 0; b < t->nblocks; ++b) {                   // slot set nblocks - b holds block b (tower_bwd_body::slot_of)
        const int L = t->nblocks - b;
        const m2m_block& k = t->blk[b];
        x.g_ln2_w[L] = k.g_ln2_w; x.g_ln2_b[L] = k.g_ln2_b;
        x.g_tok_w1[L] = k.g_tok_w1; x.g_tok_w2[L] = k.g_tok_w2; x.g_tok_b1[L] = k.g_tok_b1; x.g_tok_b2[L] = k.g_tok_b2;
        x.g_ln1_w[L] = k.g_ln1_w; x.g_ln1_b[L] = k.g_ln1_b;
        x.g_b2[L] = k.g_ch_b2;
    }
}
bool m2m_split_eligible(const m2m_tower* t, int B, int training);
// see split.h: the slot reduction of the fused single-tower backward launch (launch_bwd_dm) of tower t at batch B
bool m2m_small_part_deferred(SplitReduceTower& x, const m2m_tower* t, int B) {
    if (!t || m2m_is_wide(t) || t->N < 1 || t->N > 8 || !m2m_small_part(t) || m2m_split_eligible(t, B, 1)) return false;
    const int SPW = BM / t->N;
    m2m_small_part_reduce_args(x, t, (B + SPW - 1) / SPW);
    return true;
}
template <int P, int D, int NMAX, int TG, int DM, bool PART = false, bool HREC = false>
__global__ __launch_bounds__(NTHREADS) M2M_BWD_KATTR void tower_bwd_group_kernel(const BwdGroupArgs a, int B, unsigned int seed,
                                                                   unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware mapping: workgroups are dealt to the 8 XCDs round-robin (id % 8), each XCD has its own 4 MB L2.  Tower 0
    // takes XCDs 0-3, tower 1 XCDs 4-7, so an L2 caches ONE tower's weights (2.25 MB per block in the backward chain; both
    // towers' 4.5 MB would not fit) and each tower's weights are fetched by four L2s instead of eight.
    const int id = blockIdx.x, xcd = id & 7, t = xcd >> 2;
    const int wg = (id >> 3) * 4 + (xcd & 3);
    if (wg >= a.ntiles[t]) return;
    tower_bwd_body<m2m_tower4, P, D, NMAX, TG, DM, PART, HREC>(a.tw[t], B, a.d_out[t], a.d_out_ss[t], a.d_pooled[t], a.d_x0[t], a.d_x0_ss[t],
                                                                seed, step_host, step_dev, wg, a.ntiles[t], smem, PART ? a.part[t] : nullptr, a.dx0_chn[t]);
}

#ifdef M2M_ISA_PROBE
// ISA probe (hipcc -S -DM2M_ISA_PROBE): only the instantiations of the benchmark's two backward launches, no host code
template __global__ void tower_bwd_group_kernel<PREC_BF16, 128, 4, 8, DM_HALF, true, false>(const BwdGroupArgs, int, unsigned int, unsigned int, const unsigned int*);
template __global__ void tower_bwd_kernel<PREC_BF16, 128, 8, 16, DM_HALF, true, false>(const m2m_tower, int, const float*, long, const float*, float*, long, unsigned int, unsigned int, const unsigned int*);
#else
template <int P, int D, int NMAX, int TG>
static size_t bwd_lds_bytes(int nblocks, int N, int Cp) { return BwdLds<P, D, NMAX, TG>::bytes(nblocks, N, Cp); }

template <int P, int D, int NMAX, int TG, int DM>
static int launch_bwd_group_dm(const BwdGroupArgs& a, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev,
                               hipStream_t st, bool hrec_req) {
    const size_t lds = std::max(bwd_lds_bytes<P, D, NMAX, TG>(a.tw[0].nblocks, a.tw[0].N, a.tw[0].Cp),
                                bwd_lds_bytes<P, D, NMAX, TG>(a.tw[1].nblocks, a.tw[1].N, a.tw[1].Cp));
    if (lds > M2M_LDS_MAX || a.tw[0].Cp > 8 * NTHREADS || a.tw[1].Cp > 8 * NTHREADS) {
        m2m_set_error("towers_backward: blocks x channel_dim exceed the workgroup's LDS", __FILE__, __LINE__);
        return -1;
    }
    constexpr bool CAN_PART = P == PREC_BF16 && D == 128 && NMAX > 0;      // (the slot form is built where the runtime allocates slots)
    const bool part = CAN_PART && a.part[0] && a.part[1];
    constexpr bool CAN_HREC = P == PREC_BF16 && D == 128 && DM != DM_GEN;   // (m2m_wgrad_recompute's instantiations)
    const bool hrec = CAN_HREC && hrec_req;
    if (hrec_req && !CAN_HREC) { m2m_set_error("towers_backward: recompute form requested for an instantiation without it", __FILE__, __LINE__); return -1; }
    auto kern = tower_bwd_group_kernel<P, D, NMAX, TG, DM, false, false>;
    if constexpr (CAN_PART) { if (part) kern = tower_bwd_group_kernel<P, D, NMAX, TG, DM, true, false>; }
    if constexpr (CAN_HREC) {
        if (hrec) kern = tower_bwd_group_kernel<P, D, NMAX, TG, DM, false, true>;
        if constexpr (CAN_PART) { if (hrec && part) kern = tower_bwd_group_kernel<P, D, NMAX, TG, DM, true, true>; }
    }
    static bool attr_done[4] = {false, false, false, false};
    if (!attr_done[part + 2 * hrec]) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, M2M_LDS_MAX));
        attr_done[part + 2 * hrec] = true;
    }
    const int mx = a.ntiles[0] > a.ntiles[1] ? a.ntiles[0] : a.ntiles[1];
    const int grid = 8 * ((mx + 3) / 4);                     // see the XCD-aware mapping in the kernel
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, a, B, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int D, int NMAX, int TG>
static int launch_bwd_group(const BwdGroupArgs& a, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev,
                            hipStream_t st, bool hrec) {
    switch (m2m_drop_mode(1, a.tw[0].p_drop)) {
        case DM_NONE: return launch_bwd_group_dm<P, D, NMAX, TG, DM_NONE>(a, B, seed, step, step_dev, st, hrec);
        case DM_HALF: return launch_bwd_group_dm<P, D, NMAX, TG, DM_HALF>(a, B, seed, step, step_dev, st, hrec);
        default:      return launch_bwd_group_dm<P, D, NMAX, TG, DM_GEN>(a, B, seed, step, step_dev, st, hrec);
    }
}

template <int P, int D, int NMAX, int TG, int DM>
static int launch_bwd_dm(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                      long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const int SPW = NMAX > 0 ? BM / t->N : 1;
    const int grid = NMAX > 0 ? (B + SPW - 1) / SPW : (int)(((long)B * t->N + BM - 1) / BM);
    const size_t lds = bwd_lds_bytes<P, D, NMAX, TG>(t->nblocks, t->N, t->Cp);
    if (lds > M2M_LDS_MAX || t->Cp > 8 * NTHREADS) { m2m_set_error("tower_backward: blocks x channel_dim exceed the workgroup's LDS", __FILE__, __LINE__); return -1; }
    constexpr bool CAN_PART = P == PREC_BF16 && D == 128 && NMAX > 0;
    const bool part = CAN_PART && m2m_small_part(t);
    constexpr bool CAN_HREC = P == PREC_BF16 && D == 128 && DM != DM_GEN;
    const bool hrec = CAN_HREC && m2m_wgrad_recompute(t, B);
    auto kern = tower_bwd_kernel<P, D, NMAX, TG, DM, false, false>;
    if constexpr (CAN_PART) { if (part) kern = tower_bwd_kernel<P, D, NMAX, TG, DM, true, false>; }
    if constexpr (CAN_HREC) {
        if (hrec) kern = tower_bwd_kernel<P, D, NMAX, TG, DM, false, true>;
        if constexpr (CAN_PART) { if (hrec && part) kern = tower_bwd_kernel<P, D, NMAX, TG, DM, true, true>; }
    }
    static bool attr_done[4] = {false, false, false, false};
    if (!attr_done[part + 2 * hrec]) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, M2M_LDS_MAX));
        attr_done[part + 2 * hrec] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, *t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    if (part && !(t->wgrad_flags & M2M_WGRAD_REDUCES_SMALL)) {       // (flagged: the next weight-gradient launch reduces the slots)
        SplitReduceArgs r;
        memset(&r, 0, sizeof(r));
        r.ntow = 1;
        m2m_small_part_reduce_args(r.t[0], t, grid);
        return m2m_split_small_grads(r, st);
    }
    return 0;
}

template <int P, int D, int NMAX, int TG, int DM>
static int launch_bwd_heads_dm(const m2m_tower* t, int B, const BwdHeads& hd, float* losses, float* d_x0, long d_x0_ss, unsigned int seed,
                               unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const int SPW = BM / t->N;
    const int grid = (B + SPW - 1) / SPW;
    const size_t lds = bwd_lds_bytes<P, D, NMAX, TG>(t->nblocks, t->N, t->Cp);
    if (lds > M2M_LDS_MAX || t->Cp > 8 * NTHREADS) { m2m_set_error("tower_backward_heads: blocks x channel_dim exceed the workgroup's LDS", __FILE__, __LINE__); return -1; }
    constexpr bool CAN_HREC = DM != DM_GEN;
    const bool hrec = CAN_HREC && m2m_wgrad_recompute(t, B);
    auto kern = tower_bwd_heads_kernel<P, D, NMAX, TG, DM, false>;
    if constexpr (CAN_HREC) { if (hrec) kern = tower_bwd_heads_kernel<P, D, NMAX, TG, DM, true>; }
    static bool attr_done[2] = {false, false};
    if (!attr_done[hrec]) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, M2M_LDS_MAX));
        attr_done[hrec] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, *t, hd, B, d_x0, d_x0_ss, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    SplitReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.ntow = 1;
    m2m_small_part_reduce_args(r.t[0], t, grid, &hd, losses);
    return m2m_split_small_grads(r, st);
}

template <int P, int D, int NMAX, int TG>
static int launch_bwd(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                      long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    switch (m2m_drop_mode(1, t->p_drop)) {
        case DM_NONE: return launch_bwd_dm<P, D, NMAX, TG, DM_NONE>(t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st);
        case DM_HALF: return launch_bwd_dm<P, D, NMAX, TG, DM_HALF>(t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st);
        default:      return launch_bwd_dm<P, D, NMAX, TG, DM_GEN>(t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st);
    }
}

int m2m_check_tower(const m2m_tower* t, int B);
bool m2m_wgrad_recompute(const m2m_tower* t, int B);     // tower_wgrad.hip
int m2m_backward_wide(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                      long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st);

// Backward of the channel-mixing half of ONE block (+ final LayerNorm if the view has it) over B*N independent rows.
int m2m_chain_backward_rows(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                            long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    if (int rc = bwd_split_mode_init(st)) return rc;
#define M2M_BWDR_CASE(PP, DD) \
    if (t->prec == PP && t->D == DD) return launch_bwd<PP, DD, 0, 8>(t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st);
    M2M_BWDR_CASE(PREC_BF16, 32) M2M_BWDR_CASE(PREC_BF16, 64) M2M_BWDR_CASE(PREC_BF16, 128) M2M_BWDR_CASE(PREC_BF16, 256)
    M2M_BWDR_CASE(PREC_F32, 32) M2M_BWDR_CASE(PREC_F32, 64) M2M_BWDR_CASE(PREC_F32, 128) M2M_BWDR_CASE(PREC_F32, 256)
#undef M2M_BWDR_CASE
    m2m_set_error("tower_backward (wide): unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

bool m2m_can_group(const m2m_tower* a, const m2m_tower* b);
bool m2m_split_eligible(const m2m_tower* t, int B, int training);
bool m2m_split_can_group(const m2m_tower* a, const m2m_tower* b);
int m2m_split_backward(const m2m_tower* const* towers, const m2m_tower_gio* io, int ntow, int B, unsigned int seed, unsigned int step,
                       const unsigned int* step_dev, hipStream_t st);

// Channel-mixing halves (wide path) of two towers' blocks in one launch (token_wide.hip: m2m_backward_wide_group).
int m2m_chain_backward_rows_group(const m2m_tower* const* v, int B, const float* const* d_out, const long* d_out_ss,
                                  const float* const* d_pooled, float* const* d_x0, const long* d_x0_ss, unsigned int seed,
                                  unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    if (int rc = bwd_split_mode_init(st)) return rc;
    BwdGroupArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < 2; ++i) {
        a.tw[i] = m2m_shrink(v[i]);
        a.d_out[i] = d_out[i]; a.d_out_ss[i] = d_out_ss[i]; a.d_pooled[i] = d_pooled[i];
        a.d_x0[i] = d_x0[i]; a.d_x0_ss[i] = d_x0_ss[i];
        a.ntiles[i] = (int)(((long)B * v[i]->N + BM - 1) / BM);
    }
    const m2m_tower* t = v[0];
    if (t->D == 256 && t->prec == PREC_BF16) return launch_bwd_group<PREC_BF16, 256, 0, 8>(a, B, seed, step, step_dev, st, false);
    if (t->D == 256 && t->prec == PREC_F32) return launch_bwd_group<PREC_F32, 256, 0, 8>(a, B, seed, step, step_dev, st, false);
    m2m_set_error("towers_backward (wide): hidden_dim 256 only", __FILE__, __LINE__);
    return -1;
}
bool m2m_can_group_wide(const m2m_tower* a, const m2m_tower* b, int B);       // token_wide.hip
int m2m_backward_wide_group(const m2m_tower* const* tw, const m2m_tower_gio* io, int B, unsigned int seed, unsigned int step,
                            const unsigned int* step_dev, hipStream_t st);

extern "C" int m2m_towers_backward(const m2m_tower* const* towers, const m2m_tower_gio* io, int ntowers, int B, uint32_t seed,
                                   uint32_t step, const uint32_t* step_dev, void* stream) {
    if (!towers || !io || ntowers != 2) { m2m_set_error("towers_backward: exactly two towers per launch", __FILE__, __LINE__); return -1; }
    if (int rc = bwd_split_mode_init(reinterpret_cast<hipStream_t>(stream))) return rc;
    for (int i = 0; i < 2; ++i)
        if (int rc = m2m_check_tower(towers[i], B)) return rc;
    // the forward of this step took the split path under the same conditions (csrc/split.h)
    if (m2m_split_eligible(towers[0], B, 1) && m2m_split_eligible(towers[1], B, 1) && m2m_split_can_group(towers[0], towers[1]))
        return m2m_split_backward(towers, io, 2, B, seed, step, step_dev, reinterpret_cast<hipStream_t>(stream));
    // (mixed eligibility -- only reachable through M2M_SPLIT / M2M_SPLIT_MIN_ROWS -- takes the fused launch below; a tower flagged
    // M2M_WGRAD_REDUCES_SMALL that IS split-eligible would then have its slot reduction skipped by the weight-gradient launch
    // (m2m_small_part_deferred decides per tower): refused instead of losing its LayerNorm / token-MLP / ch_b2 gradients)
    for (int i = 0; i < 2; ++i)
        if ((towers[i]->wgrad_flags & M2M_WGRAD_REDUCES_SMALL) && m2m_split_eligible(towers[i], B, 1)) {
            m2m_set_error("towers_backward: a tower with M2M_WGRAD_REDUCES_SMALL is eligible for the column-split path but its partner is not: launch them separately", __FILE__, __LINE__);
            return -1;
        }
    if (m2m_can_group_wide(towers[0], towers[1], B))
        return m2m_backward_wide_group(towers, io, B, seed, step, step_dev, reinterpret_cast<hipStream_t>(stream));
    if (!m2m_can_group(towers[0], towers[1])) {
        m2m_set_error("towers_backward: the two towers do not share a kernel instantiation: launch them separately", __FILE__, __LINE__);
        return -1;
    }
    BwdGroupArgs a;
    for (int i = 0; i < 2; ++i) {
        a.tw[i] = m2m_shrink(towers[i]);
        a.d_out[i] = io[i].d_out; a.d_out_ss[i] = (long)io[i].d_out_sample_stride;
        a.d_pooled[i] = io[i].d_pooled;
        a.d_x0[i] = io[i].d_x0; a.d_x0_ss[i] = (long)io[i].d_x0_sample_stride;
        a.dx0_chn[i] = (char*)towers[i]->dx0_chn;
        const int SPW = BM / towers[i]->N;
        a.ntiles[i] = (B + SPW - 1) / SPW;
    }
    // (two-tower launch: measured a net LOSS -- the slot form of this instantiation spills 7 registers around the column loop
    // and its reduction launch costs what the 128-way contended atomics cost -- so it stays opt-in: M2M_SMALL_PART=2)
    // With M2M_WGRAD_GROUP_SLOTS + M2M_WGRAD_REDUCES_SMALL on both towers the reduction rides in the weight-gradient launch.
    static const bool group_part = [] { const char* e = getenv("M2M_SMALL_PART"); return e && e[0] == '2'; }();
    const bool flagged = (towers[0]->wgrad_flags & M2M_WGRAD_GROUP_SLOTS) && (towers[1]->wgrad_flags & M2M_WGRAD_GROUP_SLOTS);
    const bool part = (group_part || flagged) && m2m_small_part(towers[0]) && m2m_small_part(towers[1]);
    const bool defer = flagged && (towers[0]->wgrad_flags & M2M_WGRAD_REDUCES_SMALL) && (towers[1]->wgrad_flags & M2M_WGRAD_REDUCES_SMALL);
    for (int i = 0; i < 2; ++i) {
        const int f = towers[i]->wgrad_flags;
        // (the weight-gradient launch decides from the flags alone whether a tower's slots hold this step's sums)
        if (((f & M2M_WGRAD_GROUP_SLOTS) && !part) || ((f & M2M_WGRAD_REDUCES_SMALL) && !defer)) {
            m2m_set_error("towers_backward: M2M_WGRAD_GROUP_SLOTS / M2M_WGRAD_REDUCES_SMALL must be set on both towers, with slot buffers (gpart)", __FILE__, __LINE__);
            return -1;
        }
    }
    for (int i = 0; i < 2; ++i) a.part[i] = part ? towers[i]->gpart : nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const m2m_tower* t = towers[0];
    // both towers share precision, hidden_dim and dropout (m2m_can_group): the weight-gradient form follows from those and
    // from the split-path eligibility, which sent both towers down the split path above or neither
    const bool hrec = m2m_wgrad_recompute(towers[0], B) && m2m_wgrad_recompute(towers[1], B);
    if (hrec != (m2m_wgrad_recompute(towers[0], B) || m2m_wgrad_recompute(towers[1], B))) {
        m2m_set_error("towers_backward: the two towers disagree on the weight-gradient form: launch them separately", __FILE__, __LINE__);
        return -1;
    }
    auto finish = [&](int rc) -> int {
        if (rc || !part || defer) return rc;
        SplitReduceArgs r;
        memset(&r, 0, sizeof(r));
        r.ntow = 2;
        for (int i = 0; i < 2; ++i) m2m_small_part_reduce_args(r.t[i], towers[i], a.ntiles[i]);
        return m2m_split_small_grads(r, st);
    };
#define M2M_BWDG_CASE(PP, DD) \
    if (t->prec == PP && t->D == DD) {                                                                          \
        if (t->N <= 4) return finish(launch_bwd_group<PP, DD, 4, 8>(a, B, seed, step, step_dev, st, hrec));           \
        if (t->T % 16 == 0) return finish(launch_bwd_group<PP, DD, 8, 16>(a, B, seed, step, step_dev, st, hrec));     \
        return finish(launch_bwd_group<PP, DD, 8, 8>(a, B, seed, step, step_dev, st, hrec));                          \
    }
    M2M_BWDG_CASE(PREC_BF16, 32) M2M_BWDG_CASE(PREC_BF16, 64) M2M_BWDG_CASE(PREC_BF16, 128)
    M2M_BWDG_CASE(PREC_F32, 32) M2M_BWDG_CASE(PREC_F32, 64) M2M_BWDG_CASE(PREC_F32, 128)
#undef M2M_BWDG_CASE
    m2m_set_error("towers_backward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

// ---- backward of a tower whose launch also computes the model's classification heads (BwdHeads) ----
// 1 if m2m_tower_backward_heads takes this tower / these heads; 0: use m2m_heads_ce + m2m_tower_backward.
extern "C" int m2m_tower_backward_heads_ok(const m2m_tower* t, int B, int nheads, int K) {
    if (!t || m2m_check_tower(t, B) != 0) return 0;
    // OFF by default (M2M_FUSED_HEADS=1 enables it).  Measured on M2-Mixer-B, batch 512, three interleaved repetitions in one
    // process: fusion backward + heads in one launch 80.3-80.9 us against 68.2-69.3 + 13.6-14.1 us as two launches -- the heads'
    // latency chain (token means and head weights from L2 -> logits -> softmax -> gradients -> 15 KB of slot stores per
    // workgroup) costs ~12 us in front of the backward's own prologue, what the separate launch cost; step time unchanged.
    static const int on = [] { const char* e = getenv("M2M_FUSED_HEADS"); return e && e[0] == '1'; }();
    if (!on || !m2m_small_part(t) || m2m_split_eligible(t, B, 1)) return 0;
    if (nheads < 1 || nheads > BH_MAXH || K < 2 || K * t->D + K + 2 > SPP_STRIDE || K > 16) return 0;
    return 1;
}
extern "C" int m2m_tower_backward_heads(const m2m_tower* t, int B, const m2m_head* heads, int nheads, int own, const int64_t* labels,
                                        int K, float* logits, float* losses, int32_t* preds, float* d_x0, int64_t d_x0_ss,
                                        uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (!m2m_tower_backward_heads_ok(t, B, nheads, K)) { m2m_set_error("tower_backward_heads: unsupported tower / heads (see m2m_tower_backward_heads_ok)", __FILE__, __LINE__); return -1; }
    if (!heads || !labels || !logits || !losses || !preds || !d_x0 || own < 0 || own >= nheads) { m2m_set_error("tower_backward_heads: null argument", __FILE__, __LINE__); return -1; }
    if (int rc = bwd_split_mode_init(reinterpret_cast<hipStream_t>(stream))) return rc;
    BwdHeads hd;
    memset(&hd, 0, sizeof(hd));
    for (int h = 0; h < nheads; ++h) {
        hd.h[h] = heads[h];
        if (!heads[h].pooled || !heads[h].w || !heads[h].b || !heads[h].g_w || !heads[h].g_b || (h != own && !heads[h].d_pooled)) {
            m2m_set_error("tower_backward_heads: incomplete head", __FILE__, __LINE__);
            return -1;
        }
    }
    hd.labels = labels; hd.logits = logits; hd.preds = preds; hd.nheads = nheads; hd.K = K; hd.own = own;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int dm = m2m_drop_mode(1, t->p_drop);
#define M2M_BWDH_CASE(NM, TGG) \
    { if (dm == DM_NONE) return launch_bwd_heads_dm<PREC_BF16, 128, NM, TGG, DM_NONE>(t, B, hd, losses, d_x0, d_x0_ss, seed, step, step_dev, st); \
      if (dm == DM_HALF) return launch_bwd_heads_dm<PREC_BF16, 128, NM, TGG, DM_HALF>(t, B, hd, losses, d_x0, d_x0_ss, seed, step, step_dev, st); \
      return launch_bwd_heads_dm<PREC_BF16, 128, NM, TGG, DM_GEN>(t, B, hd, losses, d_x0, d_x0_ss, seed, step, step_dev, st); }
    if (t->N <= 4) M2M_BWDH_CASE(4, 8)
    if (t->T % 16 == 0) M2M_BWDH_CASE(8, 16)
    M2M_BWDH_CASE(8, 8)
#undef M2M_BWDH_CASE
}

extern "C" int m2m_tower_backward(const m2m_tower* t, int B, const float* d_out, int64_t d_out_ss, const float* d_pooled,
                                  float* d_x0, int64_t d_x0_ss, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (int rc = m2m_check_tower(t, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (int rc = bwd_split_mode_init(st)) return rc;
    if (m2m_is_wide(t)) return m2m_backward_wide(t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st);
    if (m2m_split_eligible(t, B, 1)) {
        m2m_tower_gio io1;
        io1.d_out = d_out; io1.d_out_sample_stride = d_out_ss; io1.d_pooled = d_pooled; io1.d_x0 = d_x0; io1.d_x0_sample_stride = d_x0_ss;
        return m2m_split_backward(&t, &io1, 1, B, seed, step, step_dev, st);
    }
#define M2M_BWD_CASE(PP, DD) \
    if (t->prec == PP && t->D == DD) {                                                                                          \
        if (t->N <= 4) return launch_bwd<PP, DD, 4, 8>(t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st);   \
        if (t->T % 16 == 0) return launch_bwd<PP, DD, 8, 16>(t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st); \
        return launch_bwd<PP, DD, 8, 8>(t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st);                \
    }
    M2M_BWD_CASE(PREC_BF16, 32) M2M_BWD_CASE(PREC_BF16, 64) M2M_BWD_CASE(PREC_BF16, 128)
    M2M_BWD_CASE(PREC_F32, 32) M2M_BWD_CASE(PREC_F32, 64) M2M_BWD_CASE(PREC_F32, 128)
#undef M2M_BWD_CASE
    m2m_set_error("tower_backward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}
#endif   // M2M_ISA_PROBE
