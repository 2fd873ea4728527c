/* libm2mixer -- C ABI of the MI355X-native M2-Mixer training hot path.
 *
 * The reference (bezirganyan/m2-mixer) is pure Python: it has NO FFI for this path.  Its
 * "operator interface" is the nn.Module protocol of modules/mixer.py.  Each entry point below
 * names the reference code it replaces (paths relative to the reference checkout); the Python
 * binding a maintainer would add lives in m2_mixer_amd/_lib.py and is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed / torch CUDA storage) unless noted;
 *   - tensors are dense row-major fp32 unless noted; "packed" operands are built by m2m_pack_*;
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous on that stream,
 *     allocate nothing and never synchronise (they can be captured into a hipGraph);
 *   - return value: 0 ok; -1 unsupported shape / bad argument (nothing launched, see
 *     m2m_last_error()); -2 HIP runtime error.  There is no CPU fallback anywhere.
 */
#ifndef M2MIXER_H
#define M2MIXER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M2M_ABI_VERSION 17
#define M2M_MAX_BLOCKS 8      /* MixerBlocks per m2m_tower; longer towers are chained by the caller */
#define M2M_ROWS_PER_WG 16    /* token rows one workgroup keeps on chip */

#define M2M_SPLIT_GPART 1472  /* floats per workgroup and launch in m2m_tower.gpart (7 x 128 + 576)                       */

#define M2M_PREC_BF16 0       /* bf16 operands, fp32 accumulate, fp32 residual stream / LayerNorm */
#define M2M_PREC_F32 1        /* exact fp32 MFMA (parity mode: matches the reference CPU path to ~1e-5) */

/* One MixerBlock (reference: modules/mixer.py:25-47).  Parameter names = reference state-dict keys. */
typedef struct m2m_block {
    /* fp32 master parameters */
    const float* ln1_w;   /* token_mix.0.weight            (D)    */
    const float* ln1_b;   /* token_mix.0.bias              (D)    */
    const float* tok_w1;  /* token_mix.2.net.0.weight      (T, N) */
    const float* tok_b1;  /* token_mix.2.net.0.bias        (T)    */
    const float* tok_w2;  /* token_mix.2.net.3.weight      (N, T) */
    const float* tok_b2;  /* token_mix.2.net.3.bias        (N)    */
    const float* ln2_w;   /* channel_mix.0.weight          (D)    */
    const float* ln2_b;   /* channel_mix.0.bias            (D)    */
    const float* ch_w1;   /* channel_mix.1.net.0.weight    (C, D) */
    const float* ch_b1;   /* channel_mix.1.net.0.bias      (C)    */
    const float* ch_w2;   /* channel_mix.1.net.3.weight    (D, C) */
    const float* ch_b2;   /* channel_mix.1.net.3.bias      (D)    */
    /* packed MFMA-operand copies of ch_w1 / ch_w2 in the tower's precision (m2m_pack_tower) */
    void* w1n;            /* NAT  [i=c][k=d] = W1[c][d]   forward GEMM1, recompute in backward        */
    void* w2c;            /* CHN  [i=d][k=c] = W2[d][c]   forward GEMM2                               */
    void* w2tn;           /* NAT  [i=c][k=d] = W2[d][c]   backward dH = dY W2                         */
    void* w1tc;           /* CHN  [i=d][k=c] = W1[c][d]   backward dA = dH W1                         */
    float* ch_b1p;        /* ch_b1 zero-padded to Cp                                                  */
    /* fp32 gradients (accumulated with +=; the caller zeroes them), same shapes as the parameters */
    float* g_ln1_w; float* g_ln1_b; float* g_tok_w1; float* g_tok_b1; float* g_tok_w2; float* g_tok_b2;
    float* g_ln2_w; float* g_ln2_b; float* g_ch_w1; float* g_ch_b1; float* g_ch_w2; float* g_ch_b2;
    /* activations saved by forward(training) for backward: (B*N, D) fp32 each */
    float* x_in;          /* block input                                */
    float* x_mid;         /* after the token-mixing residual            */
    /* packed operands written by m2m_tower_backward for m2m_tower_wgrad, laid out per 32-row pair of tiles
     * (k = token row m inside the pair, CHN order): */
    void* at_chn;         /* LN2(x_mid)^T                    [pair][dt][lane]   i = d, k = m           (rows x D elements)  */
    void* dyt_chn;        /* d(channel MLP out)^T            [pair][dt][lane]   i = d, k = m                                */
    void* h_chn;          /* hidden activation^T (after GELU + dropout), i = c, k = m, rows x Cp elements.  bf16:
                           * [ct / 2][pair][16-row half][lane][tile 2q: 4 elem | tile 2q+1: 4 elem]; fp32: [ct][pair][half][lane].
                           * Recompute form (m2m_wgrad_form() == 1): the SAME buffer holds the packed NAT image of LN2(x_mid),
                           * [16-row tile][k-block][lane] 16 B -- the weight-gradient launch recomputes the hidden activation */
    void* dh_chn;         /* gradient wrt the hidden pre-activation ^T, same layout                                          */
} m2m_block;

/* A stack of MixerBlocks + optional final LayerNorm: the body of MLPMixer / FusionMixer /
 * MLPMixerNoPatching (reference: modules/mixer.py:125-132, :158-162, :182-186).
 * Two execution paths behind the same entry points:
 *   fused (N <= 8, D <= 128): one launch walks all blocks with the token rows of whole samples resident on chip;
 *   wide  (N <= 128, D <= 256; MIMIC N = 24/25, MM-IMDb N = 40/80, D = 256): per block one token-mixing launch
 *         (a workgroup owns 64 (sample, channel) columns) and one channel-mixing launch (16 token rows per
 *         workgroup, same MFMA code as the fused path), the residual stream passing through x_in / x_mid / ws_*. */
typedef struct m2m_tower {
    int32_t prec;          /* M2M_PREC_*                                       */
    int32_t D, N, T, C;    /* hidden_dim, num_patch, token_dim, channel_dim    */
    int32_t Cp;            /* C rounded up to a multiple of 32 (packed operands are zero-padded) */
    int32_t nblocks;       /* <= M2M_MAX_BLOCKS                                */
    int32_t has_final_ln;  /* 1: apply layer_norm after the blocks             */
    float p_drop;          /* nn.Dropout p of the four dropout sites of every block */
    uint32_t site_base;    /* distinguishes this tower's dropout streams       */
    const float* lnf_w;    /* layer_norm.weight (D) */
    const float* lnf_b;    /* layer_norm.bias   (D) */
    float* g_lnf_w;
    float* g_lnf_b;
    float* x_final;        /* saved input of the final LayerNorm (B*N, D) */
    float* ws_a;           /* two (B*N, D) fp32 workspaces; required only on the wide path (N > 8 or D > 128), where  */
    float* ws_b;           /* token mixing and channel mixing are separate launches and hand the stream over in HBM */
    m2m_block blk[M2M_MAX_BLOCKS];
    /* ---- split path (optional; fused-class towers in bf16 at large batches, see csrc/split.h) -------------------------
     * With these buffers present and B*N large enough, m2m_tower_forward / m2m_towers_forward and their backward counterparts run every block as two launches:
     * a per-sample launch (token mixing, LayerNorms, residual) and a channel-mixing launch whose workgroups own 128 token
     * rows x 1/nsplit of the hidden columns, so each CU streams 1/nsplit of the weights for 8x the rows.  Results are the
     * same up to fp32 summation order.  slabs == NULL (or nsplit < 2) keeps the one-launch-per-tower path. */
    float* slabs;          /* nsplit x (B*N, D) fp32: partial results of the column-split launches                       */
    int32_t nsplit;        /* slabs the buffer has room for (the library uses up to 8)                                     */
    int32_t wgrad_flags;   /* M2M_WGRAD_* (weight-gradient launches)                                                        */
    float* xres;           /* (B*N, D) fp32: residual / gradient stream carried between the launches                       */
    float* gpart;          /* (nblocks + 1 [+ 3: m2m_tower_backward_heads]) x ceil(B / (16 / N)) x M2M_SPLIT_GPART floats: per-workgroup partial sums of the small
                            * gradients (LayerNorms, token MLP, ch_b2), stored plainly and summed by one reduction launch --
                            * deterministic, and free of the same-address atomics of 256 workgroups                          */
    void* a_nat[M2M_MAX_BLOCKS];   /* per block: LN2(x_mid) as packed NAT blocks [16-row tile][k-block], rows padded to 16 */
    void* dy_nat[M2M_MAX_BLOCKS];  /* per block: d(channel MLP out) after its dropout mask, same layout                    */
    /* ---- weight-gradient slot (optional) ---------------------------------------------------------------------------------
     * per block 2 C D + C floats laid out [dW1 (C, D) | db1 (C) | dW2 (D, C)] (NULL: none).  With it -- and g_ch_w1 / g_ch_b1 /
     * g_ch_w2 of every block lying back to back in that order, as in a flat gradient buffer -- m2m_towers_wgrad may split a
     * tower with twice the rows of its neighbours into TWO row groups: group 0 writes the gradient, group 1 stores its partial
     * sums here with plain stores (no atomics, no zero fill).  The caller adds the slot to the gradient: inside the optimizer
     * (m2m_adam_step_ranges) or with m2m_wgrad_fold; m2m_wgrad_slot_groups says whether a launch will use it. */
    float* wslot[M2M_MAX_BLOCKS];
    /* ---- packed image of the gradient wrt the tower input (optional) ------------------------------------------------------
     * With it m2m_tower_backward / m2m_towers_backward also leave d_x0^T as packed operand blocks, laid out like at_chn
     * ([32-row pair][d tile][lane] 16 B, k = token row in chained order): the first operand of the patch-embedding weight
     * gradient, which then needs neither LDS staging nor atomics (m2m_towers_wgrad, `embed_towers`).
     * Precision note: in that single-owner form the embedding BIAS gradient is summed from this image too, i.e. from d_x0
     * rounded to bf16 (2^-9 relative per summand), where the row-group form sums the fp32 d_x0; observed difference <= 3e-3 of
     * the tensor's max (tests/test_gpu_parity.py, "bf16 embedding bias grad"); the reference has no fixture for it. */
    void* dx0_chn;
} m2m_tower;
#define M2M_WGRAD_OVERWRITE 1 /* wgrad_flags: g_ch_w1 / g_ch_b1 / g_ch_w2 are WRITTEN ("="), not accumulated ("+="): the caller
                               * neither zeroes nor accumulates them (the fused engines: one backward per optimizer step) */
#define M2M_WGRAD_GROUP_SLOTS 4  /* wgrad_flags: in the two-tower backward launch (m2m_towers_backward) this tower's small parameter
                               * gradients go through per-workgroup slots (m2m_tower.gpart) instead of float atomics -- both
                               * towers of the launch must carry the flag.  With M2M_WGRAD_REDUCES_SMALL the reduction rides in
                               * the weight-gradient launch (free); without it, it is a launch of its own (a measured net loss). */
#define M2M_WGRAD_REDUCES_SMALL 2 /* wgrad_flags: every m2m_tower_backward of this tower is followed by an m2m_towers_wgrad /
                               * m2m_tower_wgrad that includes it (same batch).  Where the backward launch collects its small
                               * parameter gradients (LayerNorms, token MLP, ch_b2) in per-workgroup slots (m2m_tower.gpart), their
                               * reduction then rides in that weight-gradient launch instead of being a launch of its own; the
                               * small gradients are complete only after it. */

/* Patch embedding = Conv2d(Cin, D, (ph,pw), stride=(ph,pw)) + 'b c h w -> b (h w) c'
 * (reference: modules/mixer.py:143-146) or, with H = N, ph = 1, pw = W = K, the plain
 * Linear(K, D) of MLPMixerNoPatching.proj (modules/mixer.py:171,180). */
typedef struct m2m_embed {
    int32_t prec;
    int32_t Cin, H, W, ph, pw;  /* input (B, Cin, H, W); patch (ph, pw) */
    int32_t D;                   /* output channels */
    int32_t K;                   /* Cin*ph*pw */
    int32_t Kp;                  /* K rounded up to the packed k-block (32 bf16 / 16 fp32) */
    const float* w;              /* to_patch_embedding.0.weight (D, Cin, ph, pw) == (D, K) */
    const float* b;              /* to_patch_embedding.0.bias   (D) */
    void* wn;                    /* packed NAT [i=d][k] */
    float* g_w;
    float* g_b;
    int32_t wgrad_flags;         /* M2M_WGRAD_OVERWRITE: the single-owner weight-gradient form (m2m_towers_wgrad, embed_towers) WRITES */
    int32_t reserved;            /* g_w ("="; g_b stays "+="): the caller neither zeroes nor accumulates it                          */
} m2m_embed;

/* ---- library ------------------------------------------------------------------------------------ */
int m2m_abi_version(void);
const char* m2m_last_error(void);         /* thread-local, host string */
/* bytes of one packed copy of an (I x K) operand in precision `prec` */
int64_t m2m_packed_bytes(int prec, int64_t I, int64_t K);

/* ---- operand packing (after every optimizer step) ------------------------------------------------- */
/* Generic: dst = packed image (mode 0 NAT / 1 CHN) of X[i][k] = src[i*stride_i + k*stride_k], i<I, k<K,
 * zero padded to (ceil16(I), ceil KB(K)).  order_k_major: 0 -> block(ib,kb) at ib*nKB+kb, 1 -> kb*nIB+ib. */
int m2m_pack(int prec, int mode, int order_k_major, const float* src, int64_t stride_i, int64_t stride_k,
             int64_t I, int64_t K, void* dst, void* stream);
int m2m_pack_tower(const m2m_tower* t, void* stream);     /* w1n, w2c, w2tn, w1tc, ch_b1p of every block */
int m2m_pack_embed(const m2m_embed* e, void* stream);
/* Everything above for a whole model in ONE launch: up to 3 towers (<= 4 blocks each) and up to 2 embeddings of one
 * precision.  What the engines run after the fused Adam (replaces the per-module repack implied by
 * torch.optim.Adam.step updating the weights the next forward reads, models/avmnist.py:412-414). */
int m2m_pack_all(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds, void* stream);
/* 1: m2m_pack_all / m2m_adam_pack_all leave this tower's w1tc copies unwritten, because nothing reads them (bf16, hidden_dim 128:
 * the backward chain takes that operand from the W1 fragments it parks in LDS; no slab buffer: not on the column-split path).
 * m2m_pack_tower always writes every copy. */
int m2m_pack_skips_w1tc(const m2m_tower* t);

/* ---- forward ------------------------------------------------------------------------------------ */
/* x0 (B*N, D) = patches(input) W^T + b.   Replaces MLPMixer.to_patch_embedding / MLPMixerNoPatching.proj. */
int m2m_embed_forward(const m2m_embed* e, const float* input, int B, float* x0, void* stream);
/* The two patch embeddings of a two-tower model (same precision and D) in ONE launch.  nsplits[i] (NULL: all 1) > 1
 * splits embedding i's contraction over K across that many workgroups per row tile; split s writes its partial sum
 * (split 0 includes the bias) to x0s[i] + s * part_strides[i] floats, and the consumer adds the parts
 * (m2m_tower_io.x0_parts).  m2m_embed_fwd_splits says how many splits pay off for an embedding (1 or 2); an embedding or
 * input the split kernel cannot take (fp32 mode, unaligned input) still fills all requested parts (sum in part 0, zeros after). */
/* head != NULL: the launch also performs m2m_step_prologue(head->...) -- the embeddings are the first launch of a training
 * step and read none of those values, so the step needs no separate prologue launch. */
typedef struct m2m_step_head {
    float* adam_state; uint32_t* drop_counter; float* losses; int32_t nlosses;
} m2m_step_head;
int m2m_embeds_forward(const m2m_embed* const* embeds, const float* const* inputs, float* const* x0s, const int* nsplits,
                       const int64_t* part_strides, int nembeds, int B, const m2m_step_head* head, void* stream);
int m2m_embed_fwd_splits(const m2m_embed* e);
/* m2m_embed_forward carrying the step prologue (head may be NULL): for models with ONE embedding whose launch is the first of the
 * training step and reads none of the head's values (MIMIC-H: the time tower's input projection). */
int m2m_embed_forward_head(const m2m_embed* e, const float* input, int B, float* x0, const m2m_step_head* head, void* stream);

/* Blocks + final LayerNorm over a (B, N, D) input.  Replaces the `for mixer_block in self.mixer_blocks`
 * loop + self.layer_norm of MLPMixer/FusionMixer/MLPMixerNoPatching.forward (modules/mixer.py:125-132).
 *   x0, x0_sample_stride : input tokens; sample b starts at x0 + b*x0_sample_stride (elements)
 *   out, out_sample_stride: output tokens (lets two towers write the halves of one fused buffer,
 *                           which is ConcatFusion(dim=1), modules/fusion.py:116-117)
 *   pooled (B, D) or NULL : mean over tokens of the output (x.mean(dim=1), models/avmnist.py:271-272,
 *                           modules/classification.py:89-90)
 *   training              : 1 -> dropout active and activations saved
 *   seed, step, step_dev  : dropout stream = f(seed, step + (step_dev ? *step_dev : 0), site, element); step_dev is a
 *                           device counter so that a captured hipGraph draws fresh masks on every replay
 *                           (m2m_counter_add bumps it); backward / wgrad must be given the same values */
int m2m_tower_forward(const m2m_tower* t, const float* x0, int64_t x0_sample_stride, int B,
                      float* out, int64_t out_sample_stride, float* pooled,
                      int training, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);

/* Two towers (e.g. the image and the audio tower, which run side by side and need half the chip each) in ONE launch
 * instead of two launches on two streams: no cross-queue fork / join in a replayed graph.  Both towers must be on the
 * fused path, share precision, hidden_dim, dropout and token class, and have at most 4 blocks; otherwise -1 (launch them
 * separately).  Arguments per tower as for m2m_tower_forward. */
/* 1: m2m_towers_forward / m2m_towers_backward take this pair at batch B in one launch per phase (fused-path pairs as described
 * above; wide pairs -- N > 8 or D > 128 -- with hidden_dim 256, token_dim <= 16, equal block counts, at most 256 column blocks:
 * MM-IMDb's two modality towers at its cfg batch: every launch of the pair is one launch on the caller's stream). */
int m2m_towers_can_group(const m2m_tower* a, const m2m_tower* b, int B);
typedef struct m2m_tower_io {
    const float* x0; int64_t x0_sample_stride;
    float* out; int64_t out_sample_stride;
    float* pooled;
    int32_t x0_parts;            /* 0 / 1: x0 is the input.  2..4: the input is the sum of x0_parts buffers, part p at */
    int64_t x0_part_stride;      /*        x0 + p * x0_part_stride floats (k-split partial sums of m2m_embeds_forward) */
} m2m_tower_io;
int m2m_towers_forward(const m2m_tower* const* towers, const m2m_tower_io* io, int ntowers, int B, int training,
                       uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);
/* The same launch with the patch embeddings inside it (round 4; reference: models/avmnist.py:259-260 -- image_mixer(image),
 * audio_mixer(audio): MLPMixer.forward = to_patch_embedding, then the blocks, modules/mixer.py:155-162).  embeds[i] != NULL:
 * tower i's input tokens are the patch embedding of inputs[i], computed by the tower's own workgroups for their own 16 token
 * rows (io[i].x0 is the dense (B N, D) scratch the rows pass through; x0_parts is ignored) -- the embedding launch of its own
 * costs ~7 us of fixed time at the head of the step.  head (may be NULL): losses[0 .. nlosses) = 0 and adam_state[0] += 1 in
 * workgroup 0; head->drop_counter must be NULL, because this launch READS the dropout counter (see m2m_towers_wgrad_tail).
 * m2m_towers_forward_embeds_ok: 1 if the pair / embeddings are taken (fused-path pair, 16 % N == 0, same precision and
 * hidden_dim, not on the column-split path), else 0: use m2m_embeds_forward + m2m_towers_forward. */
int m2m_towers_forward_embeds_ok(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int B);
int m2m_towers_forward_embeds(const m2m_tower* const* towers, const m2m_tower_io* io, int ntowers,
                              const m2m_embed* const* embeds, const float* const* inputs, const m2m_step_head* head,
                              int B, int training, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);

/* ---- backward ----------------------------------------------------------------------------------- */
/* Reverse pass through final LayerNorm + blocks.
 *   d_out / d_out_sample_stride : gradient wrt the tower output tokens, or NULL
 *   d_pooled (B, D)             : gradient wrt `pooled`, or NULL (adds d_pooled/N to every token)
 *   d_x0 / d_x0_sample_stride   : gradient wrt the tower input tokens (written)
 * Accumulates the LayerNorm, token-mixing and ch_b2 gradients; writes the packed operands (at_chn, dyt_chn, h_chn,
 * dh_chn) that m2m_tower_wgrad contracts into the channel-mixing weight gradients. */
int m2m_tower_backward(const m2m_tower* t, int B, const float* d_out, int64_t d_out_sample_stride,
                       const float* d_pooled, float* d_x0, int64_t d_x0_sample_stride,
                       uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);
/* The two-tower form, arguments per tower as for m2m_tower_backward (same grouping rules as m2m_towers_forward). */
typedef struct m2m_tower_gio {
    const float* d_out; int64_t d_out_sample_stride;
    const float* d_pooled;
    float* d_x0; int64_t d_x0_sample_stride;
} m2m_tower_gio;
int m2m_towers_backward(const m2m_tower* const* towers, const m2m_tower_gio* io, int ntowers, int B,
                        uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);
/* g_ch_w1, g_ch_b1, g_ch_w2 of every block: two contractions over all token rows (reference arithmetic:
 * modules/mixer.py:37-40 under autograd).  Two forms, chosen by the library per tower (m2m_wgrad_form):
 *   0  stored operands: streams the bf16 (fp32 in parity mode) Hact^T / dHpre^T m2m_tower_backward stored;
 *   1  recompute (bf16, hidden_dim 128, dropout off or p = 0.5): recomputes Hact from the packed image of LN2(x_mid) the
 *      backward left in h_chn and streams only dHpre^T; needs the dropout stream: seed / step / step_dev as in the forward. */
int m2m_tower_wgrad(const m2m_tower* t, int B, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);
int m2m_wgrad_form(const m2m_tower* t, int B);
/* The same for up to 4 towers (same precision and hidden_dim) in ONE launch: the towers of a model finish their backward
 * chains together, and one launch lets the hardware balance all their workgroups over the chip.  `dev_towers[i]` is a
 * device-resident byte copy of *towers[i] (kernel arguments are limited to 4 KiB; the caller refreshes the copy whenever
 * a pointer in the descriptor changes, outside any graph capture).
 * nembeds = 2: the same launch also computes m2m_embed_wgrad for the model's two patch embeddings (inputs[i], d_x0s[i]
 * as there) in extra workgroups that back-fill the CUs the tower workgroups leave idle; nembeds = 1 (ABI 17): one embedding
 * (MIMIC-H's input projection; row-group form); nembeds = 0: towers only. */
int m2m_towers_wgrad(const m2m_tower* const* towers, const m2m_tower* const* dev_towers, int ntowers,
                     const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s,
                     const m2m_tower* const* embed_towers, int nembeds,
                     int B, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);
                     /* embed_towers (NULL or one per embedding): the tower embedding i feeds.  When its dx0_chn image is
                      * present (bf16) the embedding gradients take the single-owner form: a workgroup owns 16 pixel columns
                      * over ALL token rows, operands straight from the image, plain "+=" -- no atomics. */
/* row groups m2m_tower_wgrad / m2m_towers_wgrad would give this tower at batch B without M2M_WGRAD_OVERWRITE (1: every gradient
 * element has a single owner; more: the groups add with float atomics -- such a tower must not set M2M_WGRAD_OVERWRITE) */
int m2m_wgrad_groups(const m2m_tower* t, int B);
/* 1: m2m_towers_wgrad with these embeddings / embed_towers at batch B uses the single-owner embedding form (which honours
 * m2m_embed.wgrad_flags); 0: the row-group form, "+=" with atomics onto a zeroed gradient */
int m2m_embeds_wgrad_form(const m2m_embed* const* embeds, const m2m_tower* const* embed_towers, int nembeds, int B);
/* bit i set: m2m_towers_wgrad on these towers at batch B leaves the second row group of tower i in its wslot */
int m2m_wgrad_slot_groups(const m2m_tower* const* towers, int ntowers, int B);
/* g_ch_w1 / g_ch_b1 / g_ch_w2 of every block += the tower's wslot (complete gradients before a data-parallel exchange) */
int m2m_wgrad_fold(const m2m_tower* t, void* stream);
/* g_w += d_x0^T patches(input), g_b += column sums of d_x0. */
int m2m_embed_wgrad(const m2m_embed* e, const float* input, const float* d_x0, int B, void* stream);
int m2m_embeds_wgrad(const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s, int nembeds, int B,
                     void* stream);                       /* both embeddings of a two-tower model in one launch */

/* ---- heads + multi-head loss (models/avmnist.py:271-298, models/mimic.py:106-121) ------------------ */
/* For each of nheads heads h: logits_h = pooled_h W_h^T + b_h (K classes), CrossEntropyLoss (mean),
 * total loss = sum_h head_weight[h] * loss_h.  Writes logits (nheads, B, K), losses (nheads+1: per head,
 * then total), preds (nheads, B) int32 argmax; and, if d_pooled != NULL, the gradients
 * d_pooled_h (B, D_h) and += g_w_h, g_b_h.  All heads share D here (true for every BASELINE config). */
typedef struct m2m_head {
    const float* pooled;  /* (B, D)  */
    const float* w;       /* (K, D)  */
    const float* b;       /* (K)     */
    float* g_w;
    float* g_b;
    float* d_pooled;      /* (B, D) or NULL */
    float weight;         /* coefficient of this head's mean loss in the total */
    float* g_part;        /* NULL: g_w / g_b += with float atomics (one add per workgroup: the order, hence the last bits, vary
                           * from run to run).  Else (m2m_heads_ce, K*D + K + 2 <= M2M_SPLIT_GPART): every workgroup STORES its
                           * sums at g_part + workgroup * M2M_SPLIT_GPART ([dW (K, D) | db (K) | 0 | 0]; m2m_heads_part_tiles(B)
                           * workgroups; head h's buffer must follow head h - 1's) and the caller hands the heads to
                           * m2m_towers_wgrad_heads, whose launch adds them to g_w / g_b in a fixed order. */
    const float* tokens;  /* NULL: the head reads `pooled`.  Else (ABI 17; m2m_heads_ce / m2m_heads_bce only) the head computes
                           * x.mean(dim=1) itself from the tower output tokens (B, ntok, D) -- sample stride tok_sample_stride floats,
                           * tokens in order, the same sum as the towers' token-mean launch -- and `pooled` is not read: a wide
                           * tower (MM-IMDb, MIMIC-H) called with pooled == NULL then skips that launch. */
    int64_t tok_sample_stride;
    int32_t ntok;
} m2m_head;
int m2m_heads_part_tiles(int B);      /* workgroups per head of m2m_heads_ce at batch B (the slots g_part needs) */
int m2m_heads_ce(const m2m_head* heads, int nheads, const int64_t* labels, int B, int D, int K,
                 float* logits, float* losses, int32_t* preds, int zero_losses, void* stream);
                 /* losses are accumulated over workgroups: zero_losses != 0 clears them first (one more tiny launch);
                  * 0 when the caller already did (m2m_step_prologue) */

/* BCEWithLogitsLoss(pos_weight) variant (models/mmimdb.py:47-50, :115-133): targets (B, K) float multi-hot,
 * pos_weight (K); per-head loss = mean over all B*K elements; preds (nheads, B, K) int32 = sigmoid(logits) > 0.5. */
int m2m_heads_bce(const m2m_head* heads, int nheads, const float* targets, const float* pos_weight, int B, int D, int K,
                  float* logits, float* losses, int32_t* preds, int zero_losses, void* stream);

/* m2m_towers_wgrad (above), which also adds the per-workgroup partial sums of the classification heads' weight gradients (m2m_head.g_part,
 * written by m2m_heads_ce at the same batch) to g_w / g_b: heads == NULL or nheads == 0: exactly m2m_towers_wgrad. */
int m2m_towers_wgrad_heads(const m2m_tower* const* towers, const m2m_tower* const* dev_towers, int ntowers,
                           const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s,
                           const m2m_tower* const* embed_towers, int nembeds,
                           int B, uint32_t seed, uint32_t step, const uint32_t* step_dev,
                           const m2m_head* heads, int nheads, int K, void* stream);

/* m2m_towers_wgrad_heads that also advances a device counter at its head: *bump_counter += 1 (NULL: none).  For a training step
 * whose FIRST launch reads the dropout counter (m2m_towers_forward_embeds): the counter then holds "steps completed", every
 * launch of the step is given step = 1, and this launch -- behind the last reader -- advances it.  Refused for the recompute
 * form of the weight gradients (it reads the counter itself). */
int m2m_towers_wgrad_tail(const m2m_tower* const* towers, const m2m_tower* const* dev_towers, int ntowers,
                          const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s,
                          const m2m_tower* const* embed_towers, int nembeds,
                          int B, uint32_t seed, uint32_t step, const uint32_t* step_dev,
                          const m2m_head* heads, int nheads, int K, uint32_t* bump_counter, void* stream);

/* m2m_tower_backward of the tower whose token mean carries head `own`, with the model's classification heads + multi-head
 * cross-entropy (m2m_heads_ce: models/avmnist.py:271-298) computed in the launch's prologue instead of a launch of their own:
 * a workgroup owns whole samples and a sample's heads need only its own token means.  heads[h]: pooled / w / b / g_w / g_b /
 * weight as for m2m_heads_ce; d_pooled of the heads other than `own` is WRITTEN (input of those towers' backward), the one of
 * `own` stays on chip.  logits (nheads, B, K), preds (nheads, B); losses (nheads + 1) must be zero on entry (m2m_step_prologue)
 * and, like the head weight gradients, are complete after the call (they go through the tower's gpart slots: the tower needs
 * room for nblocks + 1 + nheads slot sets).  m2m_tower_backward_heads_ok: 1 if this tower / these heads are taken (bf16,
 * hidden_dim 128, fused path with gpart, K * D + K + 2 <= M2M_SPLIT_GPART); otherwise use m2m_heads_ce + m2m_tower_backward. */
int m2m_tower_backward_heads_ok(const m2m_tower* t, int B, int nheads, int K);
int m2m_tower_backward_heads(const m2m_tower* t, int B, const m2m_head* heads, int nheads, int own, const int64_t* labels,
                             int K, float* logits, float* losses, int32_t* preds, float* d_x0, int64_t d_x0_sample_stride,
                             uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);

/* ---- plain MLP tower (modules/mlp.py:4-27; the MIMIC `static` modality, models/mimic.py:98) ------- */
#define M2M_MLP_MAX_LAYERS 4
/* num_blocks x (Linear -> ReLU -> Dropout(p)) followed, if has_out, by an output Linear.  nlayers counts every Linear;
 * dims[i] -> dims[i+1] is layer i (reference state-dict keys module_list.{3*i}.weight/bias, output layer at 3*num_blocks). */
typedef struct m2m_mlp {
    int32_t nlayers;
    int32_t has_out;                         /* 1: the last Linear has no ReLU / Dropout after it */
    int32_t dims[M2M_MLP_MAX_LAYERS + 1];    /* widths, each <= 128 */
    float p_drop;
    uint32_t site_base;                      /* dropout stream of layer i = site_base + i */
    const float* w[M2M_MLP_MAX_LAYERS];      /* (dims[i+1], dims[i]) */
    const float* b[M2M_MLP_MAX_LAYERS];
    float* g_w[M2M_MLP_MAX_LAYERS];          /* gradients, accumulated with += */
    float* g_b[M2M_MLP_MAX_LAYERS];
    float* act[M2M_MLP_MAX_LAYERS];          /* saved outputs of the hidden layers (B, dims[i+1]) for backward */
} m2m_mlp;
/* out: sample b at out + b*out_sample_stride (lets the result be token 0 of a fused buffer, models/mimic.py:102);
 * out_dense (B, dims[nlayers]) or NULL: a second dense copy (input of the modality's own head, models/mimic.py:106). */
int m2m_mlp_forward(const m2m_mlp* m, const float* x, int B, float* out, int64_t out_sample_stride, float* out_dense,
                    int training, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream);
/* gradient wrt the output = d_out (strided, or NULL) + d_out_dense (or NULL); accumulates g_w / g_b.  The input is data. */
int m2m_mlp_backward(const m2m_mlp* m, const float* x, int B, const float* d_out, int64_t d_out_sample_stride,
                     const float* d_out_dense, void* stream);
/* The MLP as EXTRA WORKGROUPS of a wide tower's token-mixing launch (ABI 17; models/mimic.py:98-106: the static MLP is independent of
 * the time tower that runs beside it).  m2m_mlp_forward_ride / m2m_mlp_backward_ride only RECORD the call (same arguments as
 * above, no stream; batch <= 2048; one pending call per host thread); the NEXT m2m_tower_forward / m2m_tower_backward of a wide
 * tower on this thread whose token-mixing launch is small (<= 256 workgroups, token_dim <= 16) carries the MLP's workgroups in
 * its first such launch -- the MLP then runs on that call's stream, concurrently with the token mixing, instead of as a launch
 * of its own in front of it.  m2m_mlp_ride_flush(stream) launches a recorded call nothing carried (returns 0 if none is
 * pending): call it before anything consumes the MLP's results.  Results are those of m2m_mlp_forward / m2m_mlp_backward (same code). */
int m2m_mlp_forward_ride(const m2m_mlp* m, const float* x, int B, float* out, int64_t out_sample_stride, float* out_dense,
                         int training, uint32_t seed, uint32_t step, const uint32_t* step_dev);
int m2m_mlp_backward_ride(const m2m_mlp* m, const float* x, int B, const float* d_out, int64_t d_out_sample_stride,
                          const float* d_out_dense);
int m2m_mlp_ride_flush(void* stream);

/* ---- optimizer (torch.optim.Adam as configured at models/avmnist.py:413-415) ---------------------- */
/* state: device float[4] = {step (as float count), lr, unused, unused}; the kernel reads lr and the
 * step count from it, so a captured graph can be replayed while the host edits lr.  With bump_step != 0
 * state[0] is incremented on the stream first; a step may be applied as several calls over disjoint
 * segments of the flat buffers (bump once, n == 0 allowed), e.g. per tower as its gradients complete. */
int m2m_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float* state, float beta1, float beta2, float eps, float weight_decay,
                  float grad_scale, int bump_step, void* stream);
                  /* grad_scale < 0: scale by |grad_scale| and clear grad afterwards */
/* The same with the gradient VALUE read from a bf16 copy of `grad` (same indexing) -- the compressed, all-reduced gradient
 * of the data-parallel step (DDP's bf16_compress_hook), consumed without a pass to widen it; `grad` is only cleared. */
int m2m_adam_step_bf16(float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float* state, float beta1, float beta2, float eps, float weight_decay,
                       float grad_scale, int bump_step, void* stream);

/* m2m_adam_step / m2m_adam_step_bf16 with up to M2M_MAX_GRAD_RANGES special index ranges [lo, lo + n) of the flat buffers:
 *   add  != NULL : the gradient of element i is grad[i] + add[i - lo] (a weight-gradient slot, m2m_tower.wslot; add + (i - lo)
 *                  must be 16-byte aligned whenever i is a multiple of 4: allocate the slot with lo % 4 floats of lead padding);
 *   keep != 0    : grad[i] is not cleared (the next backward overwrites it: M2M_WGRAD_OVERWRITE).
 * `ranges` is a HOST array (copied into the kernel arguments); ranges must not overlap.  grad_bf16 may be NULL. */
#define M2M_MAX_GRAD_RANGES 16
typedef struct m2m_grad_range { int64_t lo, n; const float* add; int32_t keep; int32_t reserved; } m2m_grad_range;
int m2m_adam_step_ranges(float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                         float* state, float beta1, float beta2, float eps, float weight_decay, float grad_scale, int bump_step,
                         const m2m_grad_range* ranges, int nranges, void* stream);

/* Adam + m2m_pack_all in ONE launch: the workgroups that update a channel-mixing weight tile (or an embedding weight) emit its
 * packed operand copies from the values they have just computed, so the re-pack does not re-read the fp32 masters; everything
 * else (LayerNorms, token MLPs, biases, heads) is updated by plain flat workgroups of the same launch.  All parameters must
 * live in ONE flat buffer (param / grad / exp_avg / exp_avg_sq: same indexing, n elements); gradients are consumed (cleared).
 * m2m_adam_pack_plan fills a host struct of m2m_adam_pack_plan_bytes() bytes describing the model (which flat ranges belong to
 * tiles, the Adam constants; |grad_scale| is applied; grad_bf16 != NULL: take the gradient values from that bf16 copy); the
 * caller keeps a device copy of it and passes both to m2m_adam_pack_all.  Replaces optimizer.step() + zero_grad()
 * (models/avmnist.py:413-415) and the implied refresh of the weights the next forward reads. */
int64_t m2m_adam_pack_plan_bytes(void);
int m2m_adam_pack_plan(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds,
                       float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const float* state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                       void* plan_host);
/* the same with up to M2M_MAX_GRAD_RANGES special index ranges (the semantics of m2m_adam_step_ranges: a weight-gradient slot to
 * add, and / or gradients the next backward overwrites and this launch therefore leaves uncleared); ranges: HOST array */
int m2m_adam_pack_plan_ranges(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds,
                              float* param, float* grad, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                              const float* state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                              const m2m_grad_range* ranges, int nranges, void* plan_host);
int m2m_adam_pack_all(const m2m_tower* const* towers, int ntowers, const m2m_embed* const* embeds, int nembeds,
                      const void* plan_dev, const void* plan_host, void* stream);

/* Head of a training step, ONE launch: adam_state[0] += 1 (the step count m2m_adam_step reads), *drop_counter += 1
 * (the step_dev of the tower calls), losses[0 .. nlosses) = 0.  Any pointer may be NULL. */
int m2m_step_prologue(float* adam_state, uint32_t* drop_counter, float* losses, int nlosses, void* stream);

/* *counter += delta on the stream (device uint32). */
int m2m_counter_add(uint32_t* counter, uint32_t delta, void* stream);

/* ---- test hooks ----------------------------------------------------------------------------------- */
/* keep-mask (uint8, 1 keep) the kernels use for dropout site `site` (0 tok hidden, 1 tok out,
 * 2 channel hidden, 3 channel out) of block `blk` of tower t at (seed, step): rows x cols elements with
 * the index convention of the kernels (see DESIGN.md "Dropout"). */
int m2m_dropout_mask(const m2m_tower* t, int blk, int site, int B, uint32_t seed, uint32_t step,
                     uint8_t* mask, void* stream);
/* y = gelu(x), dy = gelu'(x) elementwise with the device implementation (n floats). */
int m2m_gelu_probe(const float* x, float* y, float* dy, int64_t n, void* stream);
/* C (I x J, fp32) = A (I x K) * B(J x K)^T through pack + MFMA, mode_b 0: B packed NAT, plain product;
 * 1: chained: C = (A B^T) is fed as operand of a second product with Bc (J2 x J): C2 = C * Bc^T.
 * Exercises the fragment layouts end to end. */
int m2m_gemm_probe(int prec, const float* A, const float* Bm, int I, int J, int K,
                   const float* Bc, int J2, float* C, float* C2, void* workspace, void* stream);

/* Shader clock under load (bench.py records it next to the timings: boxes and power states differ by several percent).
 * `nwg` workgroups of 512 threads (one per CU at nwg = 256) spin on MFMA + VALU work for `spin_ticks` ticks of the 100 MHz
 * wall clock (s_memrealtime) and store {shader cycles (s_memtime), wall ticks} per workgroup: out[2 * w], out[2 * w + 1]
 * (device memory, 2 * nwg uint64).  MHz = cycles / ticks * 100.  spin_ticks is capped at 1 000 000 (10 ms). */
int m2m_clock_probe(uint64_t* out, int nwg, int spin_ticks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M2MIXER_H */
