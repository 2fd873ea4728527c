// Token-mixing MLP of the fused path on the matrix pipe (bf16 mode).
//
// Reference: MixerBlock.token_mix, modules/mixer.py:30-35 -- per (sample, channel) column a tiny MLP over the N tokens:
//   h[t] = b1[t] + sum_n W1[t][n] u[n],  g = dropout(gelu(h)),  o[n] = b2[n] + sum_t W2[n][t] g[t],  x[n] += dropout(o[n]).
// The VALU form (one thread per column, tower_fwd.hip / tower_bwd.hip, kept for the fp32 parity mode) spends ~17 VALU
// instructions per (column, hidden unit) forward and ~45 backward, most of them the multiply-adds; here those run as MFMAs
// on 16-column tiles and the VALU keeps GELU / dropout / packing only:
//   * products over the N <= 8 tokens (K = 4 or 8): v_mfma_f32_16x16x4_f32, exact fp32 (the same fmaf chain);
//   * products over the T <= 32 hidden units or over 32 columns: v_mfma_f32_16x16x32_bf16 with the accumulators of the
//     previous product as operand (chained k order, common.h), fp32 accumulate -- the precision of the channel mixing.
// Layout vocabulary: an accumulator tile has its COLUMN index on the 16 lanes (il) and its rows in (g, r): row 4g + r.
#pragma once
#include "tile.h"

static __device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ f32x4_t mfma32(const Frag& a, const Frag& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
}
static __device__ __forceinline__ Frag chain_bf16(const f32x4_t& t0, const f32x4_t& t1) {
    Frag f;
    f.u[0] = pack_bf2(t0[0], t0[1]);
    f.u[1] = pack_bf2(t0[2], t0[3]);
    f.u[2] = pack_bf2(t1[0], t1[1]);
    f.u[3] = pack_bf2(t1[2], t1[3]);
    return f;
}
// k index of element j of lane group g in the chained order
static __device__ __forceinline__ int chn_k(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }

// Keep-words of the token sites of this workgroup's columns -> LDS (one thread per column; invalid columns: 0).
//   wth[p] bit t: hidden unit t of column p = sl * D + d is kept;  wto[p] bit n: output token n is kept.
template <int D, int DM>
static __device__ __forceinline__ void token_keep_words(unsigned int* wth, unsigned int* wto, const Drop& dr_th, const Drop& dr_to,
                                                        int s0, int ns, int spw, int N, int T, int tid) {
    if (DM == DM_NONE) return;
    _Pragma("unroll 1") for (int p = tid; p < spw * D; p += NTHREADS) {
        const int sl = p / D, d = p % D;
        const unsigned int bd = (unsigned int)(s0 + sl) * D + d;
        const bool v = sl < ns;
        wth[p] = v ? drop_row_bits<DM>(dr_th, bd, T) : 0u;
        wto[p] = v ? drop_row_bits<DM>(dr_to, bd, N) : 0u;
    }
}

// ---- forward -------------------------------------------------------------------------------------------------------
//   ub    [BM][XLD]  LN1 output U (rows sl * N + n)
//   xs    [BM][XLD]  residual stream, receives += dropout(o)
//   tokw  [32][2 NMAX + 4] zero-padded token weights: W1[t][n] | W2[n][t] | b1[t]   (rows t >= T and columns n >= N zero)
//   tokb2 [8] zero-padded b2
// Per 16-column tile (16 channels of one sample), one wave:  H^T[t][col] (two 16-row tiles: t = 4g + r + 16 tt) by the f32
// MFMA (A = W1, B = U), GELU / dropout on the accumulators, which then are the B operand (k = t) of O[n][col] = W2 G.
template <int D, int NMAX, int DM>
static __device__ __forceinline__ void token_fwd_mfma(const float* ub, float* xs, const float* tokw, const float* tokb2,
                                                      const gtab2_t* gtab, const unsigned int* wth, const unsigned int* wto,
                                                      int N, int ns, float scale_th, float scale_to, int wave, int lane) {
    constexpr int XLD = TileGeom<D>::XLD, TW_LD = 2 * NMAX + 4, KS = NMAX / 4, CT = D / 16;
    const int g = lane >> 4, il = lane & 15;
    // operands that depend on the block only
    float w1a[2][KS];
    f32x4_t b1v[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w1a[tt][ks] = tokw[(il + 16 * tt) * TW_LD + g + 4 * ks];      // A[i = t][k = n]
#pragma unroll
        for (int r = 0; r < 4; ++r) b1v[tt][r] = tokw[(4 * g + r + 16 * tt) * TW_LD + 2 * NMAX];
    }
    Frag w2a;                                                                                         // A[i = n][k = t chained]
    {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = tokw[chn_k(g, j) * TW_LD + NMAX + (il & (NMAX - 1))];
            v[j] = il < NMAX ? w : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) w2a.u[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    }
    f32x4_t b2v;
#pragma unroll
    for (int r = 0; r < 4; ++r) b2v[r] = (4 * g + r < NMAX) ? tokb2[(4 * g + r) & 7] : 0.f;

    // TU tiles per pass, stage by stage, so that the LDS round trips (operand, table, read-modify-write) of the tiles overlap
    constexpr int TU = NMAX == 4 ? 4 : 2;
    const int ntile = ns * CT;
    for (int p0 = wave; p0 < ntile; p0 += NWAVES * TU) {
        int sl[TU], d0[TU];
        bool pv[TU];
        f32x4_t hacc[TU][2];
        unsigned int word[TU], wo[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int p = p0 + u * NWAVES;
            pv[u] = p < ntile;
            const int pc = pv[u] ? p : p0;
            sl[u] = pc / CT;
            d0[u] = (pc % CT) * 16;
            float bu[KS];                                                                             // B[k = n][j = col]
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int n = g + 4 * ks;
                const float v = ub[(sl[u] * N + (n < N ? n : 0)) * XLD + d0[u] + il];
                bu[ks] = n < N ? v : 0.f;
            }
            word[u] = wo[u] = 0xFFFFFFFFu;
            if (DM != DM_NONE) {
                word[u] = wth[sl[u] * D + d0[u] + il] >> (4 * g);
                wo[u] = wto[sl[u] * D + d0[u] + il];
            }
            hacc[u][0] = b1v[0];
            hacc[u][1] = b1v[1];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                hacc[u][0] = mfma4(w1a[0][ks], bu[ks], hacc[u][0]);
                hacc[u][1] = mfma4(w1a[1][ks], bu[ks], hacc[u][1]);
            }
        }
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = ActTokF::gelu_scaled(gtab, hacc[u][tt][r], scale_th);
                    hacc[u][tt][r] = DM == DM_NONE ? v : mask_f(v, bit_to_mask(word[u], 16 * tt + r));
                }
        f32x4_t o[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) o[u] = mfma32(w2a, chain_bf16(hacc[u][0], hacc[u][1]), b2v);     // rows n, columns col
        // residual update: ALL old values first (unconditional reads, row clamped into the sample), then the guarded stores.
        // Written as one guarded "*px += ..." per element, every read-modify-write sat in its own basic block behind
        // s_waitcnt lgkmcnt(0): TU x 4 = 16 LDS round trips in series per pass.
        float xo[TU][4];
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 4 * g + r;
                xo[u][r] = xs[(sl[u] * N + (n < N ? n : N - 1)) * XLD + d0[u] + il];
            }
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 4 * g + r;
                if (pv[u] && n < N)
                    xs[(sl[u] * N + n) * XLD + d0[u] + il] = xo[u][r] + (((wo[u] >> n) & 1u) ? o[u][r] * scale_to : 0.f);
            }
    }
}

// ---- backward -------------------------------------------------------------------------------------------------------
//   ub   [BM][XLD]  in: U = LN1 output;  out: dU (gradient wrt the LN1 output), same rows / columns
//   dov  [BM][XLD]  dO' = dropout'(dx): upstream gradient of the token-mixing output, mask and scale applied
//   red  [NWAVES][TOK_RED_LD<NMAX>] per-wave partial sums of the token-weight gradients (summed by the caller):
//          [n <= NMAX][32]  dW1[t][n] at n * 32 + t, row n = N holds db1[t]
//          [n <  NMAX][32]  dW2[n][t] at (NMAX + 1 + n) * 32 + t
//          [NMAX]           db2[n]    at 2 * (NMAX + 1) * 32 + n      (rows / entries beyond N are not written)
// Per PAIR of 16-column tiles (32 channels of one sample), one wave, with the COLUMNS in the accumulator rows:
//   H[col][t], dG[col][t] (f32 MFMA, A = U / dO', B = W1 / W2), dH = dG gelu'(H) mask, G = gelu(H) mask on the VALU;
//   dW1 += U dH, db1 (a row of ones in the U operand), dW2 += dO' G, db2 (a ones operand): bf16 MFMAs over the 32 columns;
//   dH^T through an identity MFMA (tower_bwd.hip does the same for the channel-mixing operands), then dU = W1^T dH.
template <int NMAX> struct TokRed { static constexpr int LD = 2 * (NMAX + 1) * 32 + NMAX; };

template <int D, int NMAX, int DM>
static __device__ __forceinline__ void token_bwd_mfma(float* ub, const float* dov, const float* tokw, const gtabB_t* gtab,
                                                      const unsigned int* wth, float* red, int N, int ns, float scale_th,
                                                      int wave, int lane) {
    constexpr int XLD = TileGeom<D>::XLD, TW_LD = 2 * NMAX + 4, KS = NMAX / 4, CP = D / 32;
    const int g = lane >> 4, il = lane & 15;
    float w1b[2][KS], w2b[2][KS], b1s[2];                                    // B[k = n][j = t]
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            w1b[tt][ks] = tokw[(il + 16 * tt) * TW_LD + g + 4 * ks];
            w2b[tt][ks] = tokw[(il + 16 * tt) * TW_LD + NMAX + g + 4 * ks];
        }
        b1s[tt] = tokw[(il + 16 * tt) * TW_LD + 2 * NMAX];
    }
    Frag w1t;                                                                // A[i = n][k = t chained] = W1[t][n]
    {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = tokw[chn_k(g, j) * TW_LD + (il & (NMAX - 1))];
            v[j] = il < NMAX ? w : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) w1t.u[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
    }
    // identity selectors of the transposing MFMA: B[k = col chained][j = col'] for col' in the first / second 16 columns
    Frag idf[2];
    {
        const unsigned int sel = (g == (il >> 2)) ? ((il & 1) ? 0x3F800000u : 0x00003F80u) : 0u;
        const unsigned int a = (il & 2) ? 0u : sel, b = (il & 2) ? sel : 0u;
        idf[0].u = u32x4_t{a, b, 0u, 0u};
        idf[1].u = u32x4_t{0u, 0u, a, b};
    }
    Frag ones;
    ones.u = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    f32x4_t dw1[2], dw2[2], db2 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) dw1[tt] = dw2[tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // PU pairs per pass, stage by stage: one pair alone is a chain of ~10 dependent LDS / MFMA / VALU round trips, and the two
    // waves of a SIMD do not cover it (measured: 3.2 us per pair against 0.8 us of instruction issue)
#ifndef M2M_TOK_PU
#define M2M_TOK_PU 1
#endif
    constexpr int PU = NMAX == 4 ? M2M_TOK_PU : 1;      // (pairs per pass; NMAX 4: a 16-row tile holds 4 samples = 16 pairs, two per wave.  2 was measured in rounds 2 and 4 (-DM2M_TOK_PU=2): the 64 extra accumulator registers spill around the phase (58 VGPRs), two-tower launch 165 -> 194 us / 120 -> 145 us)
    const int npair = ns * CP;
    for (int p0 = wave; p0 < npair; p0 += NWAVES * PU) {
        int sl[PU], d0[PU];
        bool pv[PU];
        f32x4_t H[PU][2][2], dG[PU][2][2];                                   // [pair][column tile][hidden tile]
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int p = p0 + u * NWAVES;
            pv[u] = p < npair;
            const int pc = pv[u] ? p : p0;
            sl[u] = pc / CP;
            d0[u] = (pc % CP) * 32;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float au[KS], ao[KS];                                        // A[i = col][k = n]
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int n = g + 4 * ks;
                    const int off = (sl[u] * N + (n < N ? n : 0)) * XLD + d0[u] + 16 * ct + il;
                    const float uu = ub[off], o = dov[off];
                    au[ks] = n < N ? uu : 0.f;
                    ao[ks] = n < N ? o : 0.f;
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    H[u][ct][tt] = f32x4_t{b1s[tt], b1s[tt], b1s[tt], b1s[tt]};
                    dG[u][ct][tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        H[u][ct][tt] = mfma4(au[ks], w1b[tt][ks], H[u][ct][tt]);
                        dG[u][ct][tt] = mfma4(ao[ks], w2b[tt][ks], dG[u][ct][tt]);
                    }
                }
            }
        }
        // dH = dG gelu'(H) keep scale, G = gelu(H) keep scale: element (column 16 ct + 4 g + r, hidden unit il + 16 tt)
        // Staged form (table activation): the 16 cells of a pair are looked up TOGETHER -- indices, all reads, a scheduling barrier,
        // then the arithmetic.  Written element by element (below), hipcc followed every ds_read_b64 with s_waitcnt lgkmcnt(0):
        // 16 exposed LDS round trips per pair, half of this phase's time (the backward kernel sits at its register limit and the
        // scheduler sinks every load to its use; tower_bwd.hip's column loop has the same cure).  The keep-mask is folded into
        // the index (a dropped element reads cell 0 = zeros), as in the column loop.
        constexpr bool TOK_STAGED = ActB<PREC_BF16>::USES_TABLE && M2M_BWD_HTAB && !(M2M_TOK_FORMULA & 2);
        if constexpr (TOK_STAGED) {
#pragma unroll
            for (int u = 0; u < PU; ++u)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {                             // (8 cells at a time: 16 more registers, no spills)
                    u32x4_t w4 = u32x4_t{~0u, ~0u, ~0u, ~0u};
                    if (DM != DM_NONE) w4 = *reinterpret_cast<const u32x4_t*>(wth + sl[u] * D + d0[u] + 16 * ct + 4 * g);
                    gtabB_t e[2][4];
                    {
                        unsigned int idx[2][4];
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                idx[tt][r] = pwl_index(H[u][ct][tt][r]);
                                if (DM != DM_NONE) idx[tt][r] &= (unsigned int)(((int)(w4[r] << (31 - il - 16 * tt))) >> 31);
                            }
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) e[tt][r] = gtab[idx[tt][r]];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x = H[u][ct][tt][r];
                            const float gl = __builtin_fmaf((float)e[tt][r][1], x, (float)e[tt][r][0]);
                            const float dgl = __builtin_fmaf((float)e[tt][r][3], x, (float)e[tt][r][2]);
                            dG[u][ct][tt][r] *= dgl;
                            H[u][ct][tt][r] = gl;
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
        for (int u = 0; u < PU; ++u)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                u32x4_t w4 = u32x4_t{~0u, ~0u, ~0u, ~0u};
                if (DM != DM_NONE) w4 = *reinterpret_cast<const u32x4_t*>(wth + sl[u] * D + d0[u] + 16 * ct + 4 * g);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float gl, dgl;
                        ActTokB::gelu_grad_scaled(gtab, H[u][ct][tt][r], scale_th, gl, dgl);
                        const float v = dG[u][ct][tt][r] * dgl;
                        if (DM == DM_NONE) { dG[u][ct][tt][r] = v; H[u][ct][tt][r] = gl; }
                        else {
                            const unsigned int mk = (unsigned int)(((int)(w4[r] << (31 - il - 16 * tt))) >> 31);
                            dG[u][ct][tt][r] = mask_f(v, mk);
                            H[u][ct][tt][r] = mask_f(gl, mk);
                        }
                    }
            }
        }
        Frag hB[PU][2];                                                      // k = the pair's 32 columns (chained), j = t
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            Frag gB[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                hB[u][tt] = chain_bf16(dG[u][0][tt], dG[u][1][tt]);
                gB[tt] = chain_bf16(H[u][0][tt], H[u][1][tt]);
            }
            // A operands of the weight gradients: lane (il = n, g) holds columns d0 + chn_k(g, j); row N of U is all ones (db1)
            Frag ua, oa;
            {
                const int row = sl[u] * N + (il < N ? il : 0);
                const f32x4_t u0 = *reinterpret_cast<const f32x4_t*>(ub + row * XLD + d0[u] + 4 * g);
                const f32x4_t u1 = *reinterpret_cast<const f32x4_t*>(ub + row * XLD + d0[u] + 16 + 4 * g);
                const f32x4_t o0 = *reinterpret_cast<const f32x4_t*>(dov + row * XLD + d0[u] + 4 * g);
                const f32x4_t o1 = *reinterpret_cast<const f32x4_t*>(dov + row * XLD + d0[u] + 16 + 4 * g);
                ua = chain_bf16(u0, u1);
                oa = chain_bf16(o0, o1);
                if (il >= N || !pv[u]) {
                    const unsigned int f = (il == N && pv[u]) ? 0x3F803F80u : 0u;
                    ua.u = u32x4_t{f, f, f, f};
                    oa.u = u32x4_t{0u, 0u, 0u, 0u};
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                dw1[tt] = mfma32(ua, hB[u][tt], dw1[tt]);
                dw2[tt] = mfma32(oa, gB[tt], dw2[tt]);
            }
            db2 = mfma32(oa, ones, db2);
        }
        // dU[n][col'] = sum_t W1[t][n] dH[col'][t]: transpose dH (rows t, columns col'), then chain
#pragma unroll
        for (int u = 0; u < PU; ++u)
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                const f32x4_t z = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const f32x4_t t0 = mfma32(hB[u][0], idf[cp], z), t1 = mfma32(hB[u][1], idf[cp], z);
                const Frag hT = chain_bf16(t0, t1);
                const f32x4_t du = mfma32(w1t, hT, z);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 4 * g + r;
                    if (pv[u] && n < N) ub[(sl[u] * N + n) * XLD + d0[u] + 16 * cp + il] = du[r];
                }
            }
    }
    float* my = red + wave * TokRed<NMAX>::LD;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * g + r;
            if (n <= N) my[n * 32 + il + 16 * tt] = dw1[tt][r];
            if (n < N) my[(NMAX + 1 + n) * 32 + il + 16 * tt] = dw2[tt][r];
        }
    if (il == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < N) my[2 * (NMAX + 1) * 32 + 4 * g + r] = db2[r];
    }
}
