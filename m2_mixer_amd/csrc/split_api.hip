// Host side of the split path: decides when a tower takes it and issues its launch sequence (see split.h).  Entered from
// m2m_tower(s)_forward / _backward; nothing here is a new C-ABI entry point.
#include "split.h"
#include <stdlib.h>

int m2m_split_mix_forward(const SplitMixArgs& a, int D, int training, float p_drop, unsigned int seed, unsigned int step,
                          const unsigned int* step_dev, hipStream_t st);
int m2m_split_chain_forward(const SplitChainArgs& a, int D, int training, float p_drop, unsigned int seed, unsigned int step,
                            const unsigned int* step_dev, hipStream_t st);
int m2m_split_mix_backward(const SplitMixBwdArgs& a, int D, float p_drop, unsigned int seed, unsigned int step,
                           const unsigned int* step_dev, hipStream_t st);
int m2m_split_chain_backward(const SplitChainArgs& a, int D, float p_drop, unsigned int seed, unsigned int step,
                             const unsigned int* step_dev, hipStream_t st);
int m2m_split_small_grads(const SplitReduceArgs& a, hipStream_t st);

// Rows (B * N) from which the split path pays: below it the chain launches cannot fill the chip (one workgroup per 128 rows
// and column split) and the fused one-launch tower wins.  M2M_SPLIT=0 / 1 forces the choice (diagnostics, A/B tests).
static int split_min_rows() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("M2M_SPLIT_MIN_ROWS");
        v = e ? atoi(e) : (1 << 30);       // off by default until the measured step beats the fused path (profiles/); M2M_SPLIT=1 forces it
    }
    return v;
}
static int split_mode() {                       // -1 auto, 0 never, 1 whenever the buffers are there
    const char* e = getenv("M2M_SPLIT");
    return e ? atoi(e) : -1;
}

static int split_count(const m2m_tower* t) {
    int s = t->nsplit < SP_MAX_SPLITS ? t->nsplit : SP_MAX_SPLITS;
    const int nunits = t->Cp / 32;
    while (s > 1 && nunits / s < 2) --s;        // at least two 32-column units (one LDS chunk) per split
    return s;
}

bool m2m_split_eligible(const m2m_tower* t, int B, int training) {
    if (split_mode() == 0) return false;
    if (t->prec != PREC_BF16 || m2m_is_wide(t) || t->D != 128 || t->nblocks < 1) return false;
    if (!t->slabs || !t->xres || !t->gpart || split_count(t) < 2) return false;
    for (int b = 0; b < t->nblocks; ++b) {
        if (!t->a_nat[b]) return false;
        if (training && (!t->dy_nat[b] || !t->blk[b].x_in || !t->blk[b].x_mid || !t->blk[b].at_chn)) return false;
    }
    if ((t->Cp / 32 + split_count(t) - 1) / split_count(t) > SP_MAX_UNITS_PER_SPLIT) return false;
    if (split_mode() == 1) return true;
    return (long)B * t->N >= split_min_rows();
}

// Towers that can share the launches of the split path: same instantiation of the mix kernels and the same split count.
bool m2m_split_can_group(const m2m_tower* a, const m2m_tower* b) {
    return a->D == b->D && a->p_drop == b->p_drop && (a->N <= 4) == (b->N <= 4) && a->nblocks == b->nblocks &&
           split_count(a) == split_count(b);
}

static void chain_tower(SplitChainTower& c, const m2m_tower* t, int b, int B) {
    const m2m_block& k = t->blk[b];
    c.a_nat = (const char*)t->a_nat[b];
    c.dy_nat = (const char*)t->dy_nat[b];
    c.w1n = (const char*)k.w1n; c.w2c = (const char*)k.w2c; c.w2tn = (const char*)k.w2tn; c.w1tc = (const char*)k.w1tc;
    c.b1p = k.ch_b1p;
    c.slabs = t->slabs;
    c.h_chn = (char*)k.h_chn; c.dh_chn = (char*)k.dh_chn;
    c.M = B * t->N;
    c.nunits = t->Cp / 32;
    c.Cp = t->Cp;
    c.site = t->site_base + 4u * b + 2u;
    c.p_drop = t->p_drop;
}

// Forward of `ntow` (1 or 2) towers of equal depth: per block a mix launch and a chain launch, then the final mix launch.
int m2m_split_forward(const m2m_tower* const* towers, const m2m_tower_io* io, int ntow, int B, int training, unsigned int seed,
                      unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const m2m_tower* t0 = towers[0];
    const int nb = t0->nblocks, S = split_count(t0);
    for (int b = 0; b <= nb; ++b) {
        SplitMixArgs m;
        memset(&m, 0, sizeof(m));
        m.ntow = ntow;
        for (int i = 0; i < ntow; ++i) {
            const m2m_tower* t = towers[i];
            SplitMixTower& x = m.t[i];
            const long M = (long)B * t->N;
            x.N = t->N; x.T = t->T; x.B = B; x.p_drop = t->p_drop;
            const int SPW = BM / t->N;
            x.ntiles = (B + SPW - 1) / SPW;
            if (b == 0) {
                x.x0 = io[i].x0; x.x0_ss = (long)io[i].x0_sample_stride;
                x.x0_parts = io[i].x0_parts > 1 ? io[i].x0_parts : 1;
                x.x0_pstride = (long)io[i].x0_part_stride;
            } else {
                x.xprev = training ? t->blk[b - 1].x_mid : t->xres;
                x.slabs = t->slabs; x.nslab = S; x.slab_stride = M * t->D;
                x.b2prev = t->blk[b - 1].ch_b2;
                x.site_prev_out = t->site_base + 4u * (b - 1) + 3u;
            }
            if (b < nb) {
                x.blk = t->blk[b];
                x.x_in = training ? t->blk[b].x_in : nullptr;
                x.x_mid = training ? t->blk[b].x_mid : t->xres;      // eval: the carry is rewritten in place (a workgroup owns its rows)
                x.a_nat = (char*)t->a_nat[b];
                x.at_chn = training ? (char*)t->blk[b].at_chn : nullptr;
                x.site = t->site_base + 4u * b;
            } else {
                x.lnf_w = t->has_final_ln ? t->lnf_w : nullptr;
                x.lnf_b = t->lnf_b;
                x.x_final = training ? t->x_final : nullptr;
                x.out = io[i].out; x.out_ss = (long)io[i].out_sample_stride; x.pooled = io[i].pooled;
            }
        }
        if (int rc = m2m_split_mix_forward(m, t0->D, training, t0->p_drop, seed, step, step_dev, st)) return rc;
        if (b == nb) break;
        SplitChainArgs c;
        memset(&c, 0, sizeof(c));
        c.ntow = ntow; c.nsplit = S;
        for (int i = 0; i < ntow; ++i) {
            chain_tower(c.t[i], towers[i], b, B);
            const int nrt = (c.t[i].M + SP_ROWS - 1) / SP_ROWS;
            c.max_rt = nrt > c.max_rt ? nrt : c.max_rt;
        }
        if (int rc = m2m_split_chain_forward(c, t0->D, training, t0->p_drop, seed, step, step_dev, st)) return rc;
    }
    return 0;
}

// Backward of `ntow` (1 or 2) towers: mix launch (upstream / the upper block's LayerNorm + token backward, the lower block's
// dYd operands), chain launch of the lower block, ... and a last mix launch that finishes block 0 and writes d_x0.
int m2m_split_backward(const m2m_tower* const* towers, const m2m_tower_gio* io, int ntow, int B, unsigned int seed, unsigned int step,
                       const unsigned int* step_dev, hipStream_t st) {
    const m2m_tower* t0 = towers[0];
    const int nb = t0->nblocks, S = split_count(t0);
    for (int lower = nb - 1; lower >= -1; --lower) {
        const int upper = lower + 1;
        SplitMixBwdArgs m;
        memset(&m, 0, sizeof(m));
        m.ntow = ntow;
        for (int i = 0; i < ntow; ++i) {
            const m2m_tower* t = towers[i];
            SplitMixBwdTower& x = m.t[i];
            const long M = (long)B * t->N;
            x.N = t->N; x.T = t->T; x.B = B; x.p_drop = t->p_drop;
            const int SPW = BM / t->N;
            x.ntiles = (B + SPW - 1) / SPW;
            x.carry = t->xres;
            x.part = t->gpart + (long)(nb - 1 - lower) * x.ntiles * SPP_STRIDE;
            if (upper == nb) {
                x.d_out = io[i].d_out; x.d_out_ss = (long)io[i].d_out_sample_stride; x.d_pooled = io[i].d_pooled;
                x.lnf_w = t->has_final_ln ? t->lnf_w : nullptr;
                x.g_lnf_w = t->g_lnf_w; x.g_lnf_b = t->g_lnf_b; x.x_final = t->x_final;
            } else {
                x.has_upper = 1;
                x.up = t->blk[upper];
                x.site_up = t->site_base + 4u * upper;
                x.slabs = t->slabs; x.nslab = S; x.slab_stride = M * t->D;
            }
            if (lower >= 0) {
                x.has_lower = 1;
                x.site_lower_out = t->site_base + 4u * lower + 3u;
                x.g_ch_b2_lower = t->blk[lower].g_ch_b2;
                x.dy_nat = (char*)t->dy_nat[lower];
                x.dyt_chn = (char*)t->blk[lower].dyt_chn;
            } else {
                x.d_x0 = io[i].d_x0; x.d_x0_ss = (long)io[i].d_x0_sample_stride;
            }
        }
        if (int rc = m2m_split_mix_backward(m, t0->D, t0->p_drop, seed, step, step_dev, st)) return rc;
        if (lower < 0) break;
        SplitChainArgs c;
        memset(&c, 0, sizeof(c));
        c.ntow = ntow; c.nsplit = S;
        for (int i = 0; i < ntow; ++i) {
            chain_tower(c.t[i], towers[i], lower, B);
            const int nrt = (c.t[i].M + SP_ROWS - 1) / SP_ROWS;
            c.max_rt = nrt > c.max_rt ? nrt : c.max_rt;
        }
        if (int rc = m2m_split_chain_backward(c, t0->D, t0->p_drop, seed, step, step_dev, st)) return rc;
    }
    // the small gradients: sum the per-workgroup partial sums of the nb + 1 mix launches into the gradient buffers
    SplitReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.ntow = ntow;
    for (int i = 0; i < ntow; ++i) {
        const m2m_tower* t = towers[i];
        SplitReduceTower& x = r.t[i];
        const int SPW = BM / t->N;
        x.part = t->gpart; x.ntiles = (B + SPW - 1) / SPW; x.nlaunch = nb + 1; x.D = t->D; x.N = t->N; x.T = t->T;
        if (t->has_final_ln) { x.g_lnf_w = t->g_lnf_w; x.g_lnf_b = t->g_lnf_b; }
        for (int L = 0; L <= nb; ++L) {
            const int lower = nb - 1 - L, upper = lower + 1;
            if (upper < nb) {
                const m2m_block& k = t->blk[upper];
                x.g_ln2_w[L] = k.g_ln2_w; x.g_ln2_b[L] = k.g_ln2_b;
                x.g_tok_w1[L] = k.g_tok_w1; x.g_tok_w2[L] = k.g_tok_w2; x.g_tok_b1[L] = k.g_tok_b1; x.g_tok_b2[L] = k.g_tok_b2;
                x.g_ln1_w[L] = k.g_ln1_w; x.g_ln1_b[L] = k.g_ln1_b;
            }
            if (lower >= 0) x.g_b2[L] = t->blk[lower].g_ch_b2;
        }
    }
    return m2m_split_small_grads(r, st);
}
