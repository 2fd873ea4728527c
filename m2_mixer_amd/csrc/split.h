// The "split" execution path of the fused-class towers (N <= 8, bf16) at large batches.
//
// Why it exists.  The fused path (tower_fwd.hip / tower_bwd.hip) keeps the 16 token rows of a workgroup on chip for the whole
// tower, so EVERY workgroup streams EVERY block's channel-mixing weights from L2: FLOP per weight byte = rows per workgroup =
// 16, and a CU takes in ~31 B/clk from L2 -- at batch 512 that caps the chain kernels near 10 % of the MFMA peak whatever the
// instruction mix.  Here a block is two launches:
//
//   mix   launch (split_mix.hip)    one workgroup per 16 token rows (whole samples): sums the column-split partial results of
//                                   the previous channel launch (+ bias, output dropout, residual), token mixing, LayerNorm-2,
//                                   and writes the bf16 MFMA operand image of its rows (forward); the mirror image backward.
//   chain launch (split_chain.hip)  one workgroup per (128 token rows, 1/S of the hidden columns): GEMM1 -> bias + GELU +
//                                   dropout on the accumulators -> GEMM2, the weights of its column slice staged through LDS
//                                   by LDS-DMA and shared by eight waves; FLOP per weight byte = 128.  Its partial [128 x D]
//                                   result goes to slab `s` of the tower's slab buffer with plain stores (no atomics: the sum
//                                   over the S slabs is the next mix launch's first step).
//
// The hidden activation still never leaves the chip in the forward pass; the launch boundary is the grid-wide
// synchronisation (~1.5 us, against ~5 us for an in-kernel grid barrier on this part).
#pragma once
#include "tile.h"

#define SP_ROWS 128               // token rows of a chain workgroup (8 row tiles of 16)
#define SP_THREADS 512
#define SP_MAX_SPLITS 8           // column splits (= slabs) a launch may use
#define SP_UNIT 32                // hidden columns one wave handles at a time (= one dropout hash word per row)
#define SP_MAX_UNITS_PER_SPLIT 64 // bias staging in LDS (C up to 8 x 64 x 32 = 16384 at 8 splits)

// ---- arguments -------------------------------------------------------------------------------------------------------------
// One tower's share of a chain launch (block `b` of the tower).
struct SplitChainTower {
    const char* a_nat;            // LN2(x_mid) as packed NAT blocks [16-row tile][k-block]
    const char* dy_nat;           // backward: dYd, same layout
    const char* w1n; const char* w2c; const char* w2tn; const char* w1tc;
    const float* b1p;
    float* slabs;                 // [nsplit][M][D] fp32 partial results
    char* h_chn; char* dh_chn;    // backward: operand streams of the weight-gradient launch (layout: tower_bwd.hip)
    int M;                        // token rows
    int nunits;                   // Cp / 32
    int Cp;
    unsigned int site;            // dropout site of the channel-hidden activation of this block
    float p_drop;
};
struct SplitChainArgs {
    SplitChainTower t[2];
    int ntow, nsplit, max_rt;     // grid = nsplit * ntow * max_rt
};

// One tower's share of a mix launch.
struct SplitMixTower {
    // ---- input of the residual stream ----
    const float* x0; long x0_ss; int x0_parts; long x0_pstride;      // first block: the tower input (sum of x0_parts buffers)
    const float* xprev;           // later: x_mid of the previous block ...
    const float* slabs; int nslab; long slab_stride;                  // ... + dropout(sum of the slabs + b2prev)
    const float* b2prev;
    unsigned int site_prev_out;   // dropout site (channel output) of the previous block
    // ---- this block (NULL ln1_w: no block, only the final LayerNorm) ----
    m2m_block blk;
    float* x_in; float* x_mid;    // saved activations (x_in NULL in eval; x_mid is also the carry to the next mix launch)
    char* a_nat; char* at_chn;    // operand images written for the chain / weight-gradient launches (at_chn NULL in eval)
    unsigned int site;            // site_base + 4 * block
    // ---- final LayerNorm + output (out == NULL: not the last launch) ----
    const float* lnf_w; const float* lnf_b; float* x_final;
    float* out; long out_ss; float* pooled;
    int N, T, B;
    float p_drop;
    int ntiles;
};
struct SplitMixArgs { SplitMixTower t[2]; int ntow; };

// One tower's share of a backward mix launch (between the chain launches of block `upper` = b + 1 and `lower` = b).
struct SplitMixBwdTower {
    // ---- no upper block (first launch): gradient of the tower output through the final LayerNorm ----
    const float* d_out; long d_out_ss; const float* d_pooled;
    const float* lnf_w; float* g_lnf_w; float* g_lnf_b; const float* x_final;      // lnf_w NULL: no final LayerNorm
    // ---- upper block: its chain launch left dA in the slabs; finish its backward (LN2, token mixing, LN1) ----
    int has_upper;
    m2m_block up;
    unsigned int site_up;         // site_base + 4 * upper
    const float* slabs; int nslab; long slab_stride;
    // ---- lower block: operands of its chain / weight-gradient launches ----
    int has_lower;
    unsigned int site_lower_out;  // site_base + 4 * lower + 3
    float* g_ch_b2_lower;
    char* dy_nat; char* dyt_chn;
    float* part;                  // this launch's slice of m2m_tower.gpart: [ntiles][SPP_STRIDE] partial sums of the small gradients
    float* carry;                 // (M, D) fp32 gradient stream between the launches (read when has_upper, written when has_lower)
    float* d_x0; long d_x0_ss;    // no lower block (last launch): gradient wrt the tower input
    int N, T, B;
    float p_drop;
    int ntiles;
};
struct SplitMixBwdArgs { SplitMixBwdTower t[2]; int ntow; };

// Layout of one workgroup's partial-sum slot (floats), D = hidden_dim (<= 128), token MLP up to T = 32, N = 8:
//   [LN2 gamma D | LN2 beta D | token MLP: W1 (T N), W2 (N T), b1 (T), b2 (N) | LN1 gamma D | LN1 beta D | ch_b2 D | final LN gamma D, beta D]
#define SPP_LN2 0
#define SPP_TOK(D) (2 * (D))
#define SPP_TOK_MAX 576
#define SPP_LN1(D) (2 * (D) + SPP_TOK_MAX)
#define SPP_B2(D) (4 * (D) + SPP_TOK_MAX)
#define SPP_LNF(D) (5 * (D) + SPP_TOK_MAX)
#define SPP_STRIDE M2M_SPLIT_GPART
static_assert(7 * 128 + SPP_TOK_MAX == SPP_STRIDE, "partial-sum slot layout");

// Reduction of the partial sums of one backward pass into the gradient buffers: launch L (0 = the first mix launch: final
// LayerNorm + the last block's ch_b2; L = k: the LayerNorm / token gradients of block nb - k and ch_b2 of block nb - k - 1).
struct SplitReduceTower {
    const float* part;            // [nlaunch][ntiles][SPP_STRIDE]
    int ntiles, nlaunch, D, N, T;
    // destinations per launch (NULL: section not produced by that launch)
    float* g_ln2_w[M2M_MAX_BLOCKS + 1]; float* g_ln2_b[M2M_MAX_BLOCKS + 1];
    float* g_tok_w1[M2M_MAX_BLOCKS + 1]; float* g_tok_w2[M2M_MAX_BLOCKS + 1]; float* g_tok_b1[M2M_MAX_BLOCKS + 1]; float* g_tok_b2[M2M_MAX_BLOCKS + 1];
    float* g_ln1_w[M2M_MAX_BLOCKS + 1]; float* g_ln1_b[M2M_MAX_BLOCKS + 1];
    float* g_b2[M2M_MAX_BLOCKS + 1];
    float* g_lnf_w; float* g_lnf_b;               // launch 0
    // slot sets head_set0 + h (h < nheads): classification head h of tower_bwd.hip's BwdHeads -- [dW (K, D) | db (K) | loss | weighted loss]
    int head_set0, nheads, K;
    float* g_hw[3]; float* g_hb[3];
    float* losses;                                // [nheads + 1]: per head, then the total
};
#define SPR_MAX_TOWERS 4       // three towers' small gradients + the classification heads
struct SplitReduceArgs { SplitReduceTower t[SPR_MAX_TOWERS]; int ntow; };

// ---- reduction of the per-workgroup slots (device body shared by split_mix.hip's launch and the weight-gradient launch) -----
#define SPR_COLS 32        // slot entries per workgroup
#define SPR_GROUPS 32      // workgroup-tile groups summed in parallel, then through LDS (1024-thread launch; NT / 32 in general)
// One workgroup of NT threads: entries [bx * 32, bx * 32 + 32) of slot set L of tower `tw`.  red: [NT / 32][33] floats of LDS.
template <int NT>
static __device__ __forceinline__ void split_small_grads_body(const SplitReduceTower& tw, int bx, int L, float* red) {
    constexpr int NG = NT / SPR_COLS, RLD = SPR_COLS + 1;
    if (L >= tw.nlaunch) return;                                  // (uniform)
    const int D = tw.D, N = tw.N, T = tw.T;
    const int col = threadIdx.x % SPR_COLS, grp = threadIdx.x / SPR_COLS;
    const int e = bx * SPR_COLS + col;
    // slot entry -> destination
    float* dst = nullptr;
    bool shared_dst = false;
    if (tw.nheads > 0 && L >= tw.head_set0) {                    // a classification head's slot set
        const int h = L - tw.head_set0, KD = tw.K * D;
        if (e < KD) dst = tw.g_hw[h] + e;
        else if (e < KD + tw.K) dst = tw.g_hb[h] + (e - KD);
        else if (e == KD + tw.K) dst = tw.losses ? tw.losses + h : nullptr;                 // (losses NULL: the slots carry no losses)
        else if (e == KD + tw.K + 1) { dst = tw.losses ? tw.losses + tw.nheads : nullptr; shared_dst = true; }   // the total: every head's set adds to it
    } else if (e < SPP_STRIDE) {
        const int ntok = 2 * T * N + T + N;
        if (e < D) dst = tw.g_ln2_w[L] ? tw.g_ln2_w[L] + e : nullptr;
        else if (e < 2 * D) dst = tw.g_ln2_b[L] ? tw.g_ln2_b[L] + (e - D) : nullptr;
        else if (e < SPP_TOK(D) + ntok) {
            const int i = e - SPP_TOK(D);
            if (tw.g_tok_w1[L]) {
                if (i < T * N) dst = tw.g_tok_w1[L] + i;
                else if (i < 2 * T * N) dst = tw.g_tok_w2[L] + (i - T * N);
                else if (i < 2 * T * N + T) dst = tw.g_tok_b1[L] + (i - 2 * T * N);
                else dst = tw.g_tok_b2[L] + (i - 2 * T * N - T);
            }
        }
        else if (e < SPP_LN1(D)) dst = nullptr;
        else if (e < SPP_LN1(D) + D) dst = tw.g_ln1_w[L] ? tw.g_ln1_w[L] + (e - SPP_LN1(D)) : nullptr;
        else if (e < SPP_LN1(D) + 2 * D) dst = tw.g_ln1_b[L] ? tw.g_ln1_b[L] + (e - SPP_LN1(D) - D) : nullptr;
        else if (e < SPP_B2(D) + D) dst = tw.g_b2[L] ? tw.g_b2[L] + (e - SPP_B2(D)) : nullptr;
        else if (e < SPP_LNF(D) + D) dst = (L == 0 && tw.g_lnf_w) ? tw.g_lnf_w + (e - SPP_LNF(D)) : nullptr;
        else if (e < SPP_LNF(D) + 2 * D) dst = (L == 0 && tw.g_lnf_b) ? tw.g_lnf_b + (e - SPP_LNF(D) - D) : nullptr;
    }
    float s = 0.f;
    if (dst && grp < NG) {
        const float* p = tw.part + (long)L * tw.ntiles * SPP_STRIDE + e;
#pragma unroll 8
        for (int w = grp; w < tw.ntiles; w += NG) s += p[(long)w * SPP_STRIDE];
    }
    if (grp < NG) red[grp * RLD + col] = s;
    __syncthreads();
    if (grp == 0 && dst) {
        float v = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) v += red[g2 * RLD + col];
        if (shared_dst) atomicAdd(dst, v);
        else *dst += v;
    }
}
#define SPR_NBX ((SPP_STRIDE + SPR_COLS - 1) / SPR_COLS)        // workgroups per (tower, slot set)

// tower_bwd.hip: fills `x` for the slot reduction of tower t's fused single-tower backward launch at batch B; returns false if
// that launch does not use slots.  (The caller reduces: immediately, or inside the next weight-gradient launch when the tower's
// wgrad_flags carry M2M_WGRAD_REDUCES_SMALL.)
bool m2m_small_part_deferred(SplitReduceTower& x, const m2m_tower* t, int B);
int m2m_split_small_grads(const SplitReduceArgs& a, hipStream_t st);
