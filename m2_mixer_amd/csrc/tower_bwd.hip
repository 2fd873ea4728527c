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
#include <algorithm>

TIMER_DECL(g_tm_bwd);
TIMER_READER(m2m_debug_timers_bwd, g_tm_bwd)

// One workgroup's share of a tower backward: token tile `wg` of `nwg`.  TW is m2m_tower (single-tower launch) or
// m2m_tower4 (the by-value descriptors of a two-tower launch).
template <class TW, int P, int D, int NMAX, int TG, int DM>
static __device__ __forceinline__ void tower_bwd_body(const TW& tw, int B, const float* __restrict__ d_out, long d_out_ss,
                                                      const float* __restrict__ d_pooled, float* __restrict__ d_x0, long d_x0_ss,
                                                      unsigned int seed, unsigned int step_host,
                                                      const unsigned int* __restrict__ step_dev, int wg, int nwg, char* smem) {
    typedef Prec<P> Pr;
    typedef TileGeom<D> G;
    constexpr int XLD = G::XLD, DT = G::DT, KD = D / Pr::KB, NF = Chain<P>::NF;
    constexpr bool TOK = NMAX > 0;                      // false: wide path, channel mixing only (rows independent)
    constexpr int NM = TOK ? NMAX : 1;
    constexpr int TILE_F = BM * XLD;                    // floats in one fp32 tile
    constexpr int IMG_B = BM * D * Pr::ESZ;             // bytes of one packed BM-row image

    constexpr int SF = SlabGeom<D>::FLOATS;             // floats in one (transposed) reduction slab, >= TILE_F
    static_assert(IMG_B <= SF * 4 && TILE_F <= SF, "packed image / fp32 tile must fit the slab it aliases");
    float* dxs = reinterpret_cast<float*>(smem);        // gradient stream
    float* slabs = dxs + TILE_F;                         // 4 slabs: S0 = ub (scratch), S1 = xh (scratch),
    float* ub = slabs;                                   //   S2 / S3 hold the packed A / dYd images during the
    float* xh = slabs + SF;                              //   hidden-column loop, then all four receive the waves'
    char* at = reinterpret_cast<char*>(slabs + 2 * SF);  //   partial dA
    char* dyp = reinterpret_cast<char*>(slabs + 3 * SF);
    float* rstd_s = slabs + 4 * SF;                      // [BM]
    float* dasum = rstd_s + BM;                          // [BM][XLD] summed dA (row-major)
    gtab_t* gtab = reinterpret_cast<gtab_t*>(dasum + TILE_F);   // [GELU_TAB_N] (bf16 mode only)
    constexpr int RED_LD0 = ((32 / TG) * (1 + 2 * NM) + NM) * TG;      // token-grad slots per wave (VALU form)
    constexpr int RED_LD = RED_LD0 > TokRed<NM>::LD ? RED_LD0 : TokRed<NM>::LD;
    float* red = reinterpret_cast<float*>(gtab + GELU_TAB_N);    // [NWAVES][RED_LD]
    constexpr int TW_LD = 2 * NM + 4;
    float* tokw = red + NWAVES * RED_LD;                         // [32][TW_LD] zero-padded token-MLP weights
    unsigned int* wth = reinterpret_cast<unsigned int*>(tokw + 32 * TW_LD);   // [BM * D] keep-words of the token-hidden site (bf16 mode)
    float* dov = dasum;                                          // dO' tile of the MFMA token path (dasum is dead by then)

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, il = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar loop control in the column loop
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
    if (Act<P>::USES_TABLE) gelu_tab_fill(gtab, make_drop(true, tw.p_drop, 0u, 0u, 0u).scale, tid, NTHREADS);
    // ---- upstream gradient of the tower output ----
    {
        const float invN = 1.0f / (float)N;
        _Pragma("unroll 1") for (int idx = tid; idx < BM * (D / 4); idx += NTHREADS) {
            const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R) {
                const long gr = row0 + r, gs = gr / N;
                if (d_out) v = *reinterpret_cast<const float4*>(d_out + gs * d_out_ss + (gr % N) * D + c);
                if (d_pooled) {
                    const float4 p = *reinterpret_cast<const float4*>(d_pooled + gs * D + c);
                    v.x += p.x * invN; v.y += p.y * invN; v.z += p.z * invN; v.w += p.w * invN;
                }
            }
            *reinterpret_cast<float4*>((tw.has_final_ln ? ub : dxs) + r * XLD + c) = v;
        }
        __syncthreads();
        if (tw.has_final_ln)
            ln_backward_tile<D>(tw.x_final + row0 * D, R, ub, tw.lnf_w, dxs, false, xh, tw.g_lnf_w, tw.g_lnf_b, tid);
    }

    TIMER_LMARK(0);       // upstream + final LN backward
    for (int b = tw.nblocks - 1; b >= 0; --b) {
        const m2m_block& bk = tw.blk[b];
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
        // (C1) dYd = dY * mask_out -> fp32 temp (ub)
        _Pragma("unroll 1") for (int idx = tb1; idx < BM * D; idx += NTHREADS) {
            const int r = idx / D, d = idx % D;
            float v = dxs[r * XLD + d];
            v = drop_keep_elem<DM>(dr_co, (unsigned int)(row0 + r) * D + d) ? v * dr_co.scale : 0.f;
            ub[r * XLD + d] = (r < R) ? v : 0.f;
        }
        __syncthreads();
        TIMER_LMARK(8);   // C1a: dYd
        // ch_b2 gradient: column sums of dYd
        _Pragma("unroll 1") for (int d = tb1; d < D; d += NTHREADS) {
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += ub[r * XLD + d];
            atomicAdd(bk.g_ch_b2 + d, s);
        }
        // pack dYd: NAT [m][d] (LDS image + global copy) and CHN [d][m] (global, for the weight gradients)
        pack_tile_nat<P, D>(ub, dyp, tb1);
        pack_tile_chn_t<P, D>(ub, reinterpret_cast<char*>(bk.dyt_chn) + pair_off, tile_in_pair, tb1);
        __syncthreads();
        TIMER_LMARK(9);   // C1b: b2 sums, packs
        // (C2) A = LN2(x_mid) -> fp32 tile (ub) -> packed images
        {
            const int r = tb1 / TPR, j = tb1 % TPR;
            float v[D / TPR], mean, rstd;
            row_stats<D>(bk.x_mid + (row0 + r) * D, r < R, j, v, mean, rstd);
#pragma unroll
            for (int e = 0; e < D / TPR; ++e) {
                const int c = ln_col<D>(e, j);
                ub[r * XLD + c] = (v[e] - mean) * rstd * bk.ln2_w[c] + bk.ln2_b[c];
            }
        }
        __syncthreads();
        TIMER_LMARK(10);  // C2a: LN2 recompute
        pack_tile_nat<P, D>(ub, at, tb1);
        pack_tile_chn_t<P, D>(ub, reinterpret_cast<char*>(bk.at_chn) + pair_off, tile_in_pair, tb1);
        __syncthreads();

        TIMER_LMARK(1);   // C1 + C2: dYd, A, packing, global copies
        // (C3) hidden-column loop
        f32x4_t dacc[MT][DT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dacc[mt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int npairs = Cp >> 5;
        // identity block of the transposing MFMA (bf16): lane (g, il) is non-zero iff g == il >> 2, at element il & 3
        const unsigned int id_sel = (g == (il >> 2)) ? ((il & 1) ? 0x3F800000u : 0x00003F80u) : 0u;
        const unsigned int id_a = (il & 2) ? 0u : id_sel, id_b = (il & 2) ? id_sel : 0u;
        // hidden_dim <= 128: the W1 / W2^T fragments of a whole step (2 x 2 x KD) live in registers and the next step's are
        // requested during this one's epilogue.  hidden_dim 256: that is 128 registers next to 64 accumulators and the 64 of
        // the third product -- the kernel spilled ~200 -- and such towers have few steps per wave (C = 512: two), so there the
        // first two products stream their fragments two k-blocks at a time, nothing held across steps.
        constexpr bool HOLD = D <= 128;
        Frag w1f[2][HOLD ? KD : 1], w2f[2][HOLD ? KD : 1];
        if (HOLD && wave < npairs) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kb = 0; kb < KD; ++kb) {
                    w1f[t][kb] = ld_frag_global(bk.w1n, (long)(2 * wave + t) * KD + kb, lane);
                    w2f[t][kb] = ld_frag_global(bk.w2tn, (long)(2 * wave + t) * KD + kb, lane);
                }
        }
        for (int q = wave; q < npairs; q += NWAVES) {
            f32x4_t bias[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const f32x4_t*>(bk.ch_b1p + 32 * q + 16 * t + 4 * g);
            f32x4_t hacc[MT][2], gacc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                hacc[mt][0] = bias[0];
                hacc[mt][1] = bias[1];
                gacc[mt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                gacc[mt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (HOLD) {
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
                            u1[t][kk] = ld_frag_global(bk.w1n, (long)(2 * q + t) * KD + k0 + kk, lane);
                            u2[t][kk] = ld_frag_global(bk.w2tn, (long)(2 * q + t) * KD + k0 + kk, lane);
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
            Frag w3f[NF][DT];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) w3f[f][dt] = ld_frag_global(bk.w1tc, (long)(q * NF + f) * DT + dt, lane);
            if (HOLD && q + NWAVES < npairs) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kb = 0; kb < KD; ++kb) {
                        w1f[t][kb] = ld_frag_global(bk.w1n, (long)(2 * (q + NWAVES) + t) * KD + kb, lane);
                        w2f[t][kb] = ld_frag_global(bk.w2tn, (long)(2 * (q + NWAVES) + t) * KD + kb, lane);
                    }
            }
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
                        Act<P>::gelu_grad_scaled(gtab, hacc[mt][t][r], dr_ch.scale, gl, dgl);
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
                Chain<P>::make(hacc[mt][0], hacc[mt][1], af[mt]);     // Hact : only stored, for the weight gradients
            }
            // Operands of the weight-gradient pass: dHpre^T and Hact^T with k = token row.  The accumulators hold
            // [c in registers][m across lanes]; an MFMA against an identity block turns them ([m][c] as the A operand,
            // chained k order) into [m in registers][c across lanes] = exactly the layout that pass consumes, exact in
            // the operand precision.  The MFMA pipe is mostly idle here, so the transpose is nearly free.
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
                        oa[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, u32x2{af[mt][0].u[2 * t], af[mt][0].u[2 * t + 1]}), id16, oa[t], 0, 0, 0);
                    } else {
                        Frag id;
                        id.f = f32x4_t{4 * g + 0 == il ? 1.f : 0.f, 4 * g + 1 == il ? 1.f : 0.f, 4 * g + 2 == il ? 1.f : 0.f, 4 * g + 3 == il ? 1.f : 0.f};
                        Pr::mma(od[t], hf[mt][t], id);
                        Pr::mma(oa[t], af[mt][t], id);
                    }
                    // od[t][r] = dHpre[m = 16u + 4g + r][c = 32q + 16t + il]
                }
                // streamed out once and read once by the weight-gradient pass: non-temporal, so that the 100 MB per
                // launch do not evict the weights the other workgroups of this XCD keep re-reading from its L2
                if (P == PREC_BF16) {
                    // [column-tile pair q][32-row pair][16-row half u][lane][tile 2q: 4 bf16 | tile 2q+1: 4 bf16]: one full
                    // 16-byte-per-lane store per operand and step (1 KiB contiguous), and the weight-gradient wave that owns
                    // both column tiles reads it back with one 16-byte load per half
                    const long off = (long)q * m2m_hchn_stride(npair) + (pair * 2 + u) * 1024 + lane * 16;
                    __builtin_nontemporal_store(u32x4_t{pack_bf2(od[0][0], od[0][1]), pack_bf2(od[0][2], od[0][3]),
                                                        pack_bf2(od[1][0], od[1][1]), pack_bf2(od[1][2], od[1][3])},
                                                reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(bk.dh_chn) + off));
                    __builtin_nontemporal_store(u32x4_t{pack_bf2(oa[0][0], oa[0][1]), pack_bf2(oa[0][2], oa[0][3]),
                                                        pack_bf2(oa[1][0], oa[1][1]), pack_bf2(oa[1][2], oa[1][3])},
                                                reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(bk.h_chn) + off));
                } else {
                    // fp32: a 16-row half is a whole k-block: [column tile][32-row pair][half][lane][16 bytes]
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const long off = (long)(2 * q + t) * m2m_hchn_stride(npair) + (pair * 2 + u) * 1024 + lane * 16;
                        __builtin_nontemporal_store(od[t], reinterpret_cast<f32x4_t*>(reinterpret_cast<char*>(bk.dh_chn) + off));
                        __builtin_nontemporal_store(oa[t], reinterpret_cast<f32x4_t*>(reinterpret_cast<char*>(bk.h_chn) + off));
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) Pr::mma(dacc[mt][dt], hf[mt][f], w3f[f][dt]);
        }
        TIMER_LMARK(2);   // C3 hidden-column loop (wave 0)
        __syncthreads();   // every wave is done reading the packed images that xh aliases
        TIMER_LMARK(11);  // wave 0 waiting for the other waves' loops
        int tb2 = tid;
        asm volatile("" : "+v"(tb2));
        // (C4) dA = sum of the eight waves' partials (through the four slabs) -> ub
        reduce_waves_to_slabs<D>(dacc, slabs, wave, g, il);
        TIMER_LMARK(12);  // C4a: slabs
        _Pragma("unroll 1") for (int idx = tb2; idx < BM * D; idx += NTHREADS) {
            const int d = idx / BM, r = idx % BM;
            dasum[r * XLD + d] = slab_sum<D>(slabs, r, d);
        }
        __syncthreads();
        TIMER_LMARK(13);  // C4b: slab sum
        // (C5) LayerNorm-2 backward; dx_mid = dY + LN2'(dA)
        ln_backward_tile<D>(bk.x_mid + row0 * D, R, dasum, bk.ln2_w, dxs, true, xh, bk.g_ln2_w, bk.g_ln2_b, tb2);

        TIMER_LMARK(3);   // C4 + C5: reduction, LN2 backward
        if constexpr (TOK) {
        // ================= token mixing backward =================
        int tb3 = tid;
        asm volatile("" : "+v"(tb3));
        const int lane3 = tb3 & 63;
        // (T1) xhat1 -> xh, U = LN1(x_in) -> ub, rstd -> rstd_s; token-MLP weights -> LDS, zero-padded:
        //   tokw[t][0..NMAX) = W1[t][n]   tokw[t][NMAX..2NMAX) = W2[n][t]   tokw[t][2NMAX] = b1[t]   (t < 32)
        _Pragma("unroll 1") for (int idx = tb3; idx < 32 * TW_LD; idx += NTHREADS) {
            const int t = idx / TW_LD, j = idx % TW_LD;
            float v = 0.f;
            if (t < T) {
                if (j < NMAX) { if (j < N) v = bk.tok_w1[t * N + j]; }
                else if (j < 2 * NMAX) { if (j - NMAX < N) v = bk.tok_w2[(j - NMAX) * T + t]; }
                else if (j == 2 * NMAX) v = bk.tok_b1[t];
            }
            tokw[idx] = v;
        }
        {
            const int r = tb3 / TPR, j = tb3 % TPR;
            float v[D / TPR], mean, rstd;
            row_stats<D>(bk.x_in + (row0 + r) * D, r < R, j, v, mean, rstd);
            if (j == 0) rstd_s[r] = rstd;
#pragma unroll
            for (int e = 0; e < D / TPR; ++e) {
                const int c = ln_col<D>(e, j);
                const float xhv = (v[e] - mean) * rstd;
                xh[r * XLD + c] = xhv;
                ub[r * XLD + c] = xhv * bk.ln1_w[c] + bk.ln1_b[c];
            }
        }
        if constexpr (P == PREC_BF16) {
            // operands of the MFMA token path (token_mfma.h): keep-words of the hidden site, dO' = dropout'(dx_mid)
            _Pragma("unroll 1") for (int p = tb3; p < SPW * D; p += NTHREADS) {
                const int sl = p / D, d = p % D;
                const bool v = sl < ns;
                const unsigned int bd = (unsigned int)(s0 + sl) * D + d;
                unsigned int wto = 0xFFFFFFFFu;
                if (DM != DM_NONE) {
                    wth[p] = v ? drop_row_bits<DM>(dr_th, bd, T) : 0u;
                    wto = drop_row_bits<DM>(dr_to, bd, N);
                }
                if (v) {
                    for (int n = 0; n < N; ++n) {
                        const float x = dxs[(sl * N + n) * XLD + d] * dr_to.scale;
                        dov[(sl * N + n) * XLD + d] = ((wto >> n) & 1u) ? x : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        TIMER_LMARK(14);  // T0: token weights, LN1 recompute, operands of the MFMA token path
        if constexpr (P == PREC_BF16) {
            token_bwd_mfma<D, NM, DM>(ub, dov, tokw, gtab, wth, red, N, ns, dr_th.scale, wave, lane3);
            __syncthreads();
            TIMER_LMARK(4);   // T1: token MLP backward (MFMA form)
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
                atomicAdd(dst, v);
            }
        } else {
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
                        Act<P>::gelu_grad_scaled(gtab, h, dr_th.scale, gl, dgl);
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
            TIMER_LMARK(4);   // T1: LN1 recompute + token MLP backward pair loop
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
            TIMER_LMARK(6);   // T1b: token-grad cross-lane sums + per-wave LDS slots
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
                atomicAdd(dst, v);
            }
        }
        __syncthreads();
        TIMER_LMARK(7);       // T1c: global atomics of the token grads
        // (T2) LayerNorm-1 backward: dx_in = dx_mid + LN1'(dU); gamma/beta gradients
        {
            const int r = tb3 / TPR, j = tb3 % TPR;
            const bool valid = r < R;
            const float rstd = rstd_s[r];
            float gv[D / TPR], xv[D / TPR];
            float gsum = 0.f, gxsum = 0.f;
#pragma unroll
            for (int e = 0; e < D / TPR; ++e) {
                const int c = ln_col<D>(e, j);
                const float u = ub[r * XLD + c];
                const float xhv = xh[r * XLD + c];
                const float gg = u * bk.ln1_w[c];
                gv[e] = gg; xv[e] = xhv;
                gsum += gg;
                gxsum = __builtin_fmaf(gg, xhv, gxsum);
                xh[r * XLD + c] = valid ? u * xhv : 0.f;       // product tile for the gamma gradient
            }
            gsum = wave_sum_xor(gsum, TPR) * (1.0f / D);
            gxsum = wave_sum_xor(gxsum, TPR) * (1.0f / D);
            if (valid) {
#pragma unroll
                for (int e = 0; e < D / TPR; ++e) {
                    const int c = ln_col<D>(e, j);
                    dxs[r * XLD + c] += rstd * (gv[e] - gsum - xv[e] * gxsum);
                }
            }
        }
        __syncthreads();
        _Pragma("unroll 1") for (int d = tb3; d < 2 * D; d += NTHREADS) {
            const float* src = d < D ? xh : ub;
            const int c = d < D ? d : d - D;
            float s = 0.f;
            for (int rr = 0; rr < R; ++rr) s += src[rr * XLD + c];
            atomicAdd((d < D ? bk.g_ln1_w : bk.g_ln1_b) + c, s);
        }
        __syncthreads();
        TIMER_LMARK(5);   // T2: LN1 backward
        }   // TOK
    }

    // ---- gradient wrt the tower input ----
    _Pragma("unroll 1") for (int idx = tid; idx < R * (D / 4); idx += NTHREADS) {
        const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
        const long gr = row0 + r;
        *reinterpret_cast<float4*>(d_x0 + (gr / N) * d_x0_ss + (gr % N) * D + c) =
            *reinterpret_cast<const float4*>(dxs + r * XLD + c);
    }
    TIMER_LFLUSH(g_tm_bwd);
}

template <int P, int D, int NMAX, int TG, int DM>
__global__ __launch_bounds__(NTHREADS) void tower_bwd_kernel(const m2m_tower tw, int B, const float* __restrict__ d_out,
                                                             long d_out_ss, const float* __restrict__ d_pooled,
                                                             float* __restrict__ d_x0, long d_x0_ss, unsigned int seed,
                                                             unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tower_bwd_body<m2m_tower, P, D, NMAX, TG, DM>(tw, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step_host, step_dev,
                                                   blockIdx.x, gridDim.x, smem);
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
};
static_assert(sizeof(BwdGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");
template <int P, int D, int NMAX, int TG, int DM>
__global__ __launch_bounds__(NTHREADS) void tower_bwd_group_kernel(const BwdGroupArgs a, int B, unsigned int seed,
                                                                   unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware mapping: workgroups are dealt to the 8 XCDs round-robin (id % 8), each XCD has its own 4 MB L2.  Tower 0
    // takes XCDs 0-3, tower 1 XCDs 4-7, so an L2 caches ONE tower's weights (2.25 MB per block in the backward chain; both
    // towers' 4.5 MB would not fit) and each tower's weights are fetched by four L2s instead of eight.
    const int id = blockIdx.x, xcd = id & 7, t = xcd >> 2;
    const int wg = (id >> 3) * 4 + (xcd & 3);
    if (wg >= a.ntiles[t]) return;
    tower_bwd_body<m2m_tower4, P, D, NMAX, TG, DM>(a.tw[t], B, a.d_out[t], a.d_out_ss[t], a.d_pooled[t], a.d_x0[t], a.d_x0_ss[t],
                                                    seed, step_host, step_dev, wg, a.ntiles[t], smem);
}

template <int P, int D, int NMAX, int TG>
static size_t bwd_lds_bytes() {
    constexpr int NM = NMAX > 0 ? NMAX : 1;
    const size_t tile_b = (size_t)BM * TileGeom<D>::XLD * sizeof(float);
    return 2 * tile_b + 4 * SlabGeom<D>::FLOATS * sizeof(float) + BM * sizeof(float) + GELU_TAB_N * 16 +
           (size_t)NWAVES * std::max(((32 / TG) * (1 + 2 * NM) + NM) * TG, TokRed<NM>::LD) * sizeof(float) + 32 * (2 * NM + 4) * sizeof(float) +
           (size_t)BM * D * sizeof(unsigned int);
}

template <int P, int D, int NMAX, int TG, int DM>
static int launch_bwd_group_dm(const BwdGroupArgs& a, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev,
                               hipStream_t st) {
    const size_t lds = bwd_lds_bytes<P, D, NMAX, TG>();
    auto kern = tower_bwd_group_kernel<P, D, NMAX, TG, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int mx = a.ntiles[0] > a.ntiles[1] ? a.ntiles[0] : a.ntiles[1];
    const int grid = 8 * ((mx + 3) / 4);                     // see the XCD-aware mapping in the kernel
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, a, B, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int D, int NMAX, int TG>
static int launch_bwd_group(const BwdGroupArgs& a, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev,
                            hipStream_t st) {
    switch (m2m_drop_mode(1, a.tw[0].p_drop)) {
        case DM_NONE: return launch_bwd_group_dm<P, D, NMAX, TG, DM_NONE>(a, B, seed, step, step_dev, st);
        case DM_HALF: return launch_bwd_group_dm<P, D, NMAX, TG, DM_HALF>(a, B, seed, step, step_dev, st);
        default:      return launch_bwd_group_dm<P, D, NMAX, TG, DM_GEN>(a, B, seed, step, step_dev, st);
    }
}

template <int P, int D, int NMAX, int TG, int DM>
static int launch_bwd_dm(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                      long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const int SPW = NMAX > 0 ? BM / t->N : 1;
    const int grid = NMAX > 0 ? (B + SPW - 1) / SPW : (int)(((long)B * t->N + BM - 1) / BM);
    const size_t lds = bwd_lds_bytes<P, D, NMAX, TG>();
    auto kern = tower_bwd_kernel<P, D, NMAX, TG, DM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, *t, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
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
int m2m_backward_wide(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                      long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st);

// Backward of the channel-mixing half of ONE block (+ final LayerNorm if the view has it) over B*N independent rows.
int m2m_chain_backward_rows(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                            long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
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

extern "C" int m2m_towers_backward(const m2m_tower* const* towers, const m2m_tower_gio* io, int ntowers, int B, uint32_t seed,
                                   uint32_t step, const uint32_t* step_dev, void* stream) {
    if (!towers || !io || ntowers != 2) { m2m_set_error("towers_backward: exactly two towers per launch", __FILE__, __LINE__); return -1; }
    for (int i = 0; i < 2; ++i)
        if (int rc = m2m_check_tower(towers[i], B)) return rc;
    // the forward of this step took the split path under the same conditions (csrc/split.h)
    if (m2m_split_eligible(towers[0], B, 1) && m2m_split_eligible(towers[1], B, 1) && m2m_split_can_group(towers[0], towers[1]))
        return m2m_split_backward(towers, io, 2, B, seed, step, step_dev, reinterpret_cast<hipStream_t>(stream));
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
        const int SPW = BM / towers[i]->N;
        a.ntiles[i] = (B + SPW - 1) / SPW;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const m2m_tower* t = towers[0];
#define M2M_BWDG_CASE(PP, DD) \
    if (t->prec == PP && t->D == DD) {                                                                          \
        if (t->N <= 4) return launch_bwd_group<PP, DD, 4, 8>(a, B, seed, step, step_dev, st);                   \
        if (t->T % 16 == 0) return launch_bwd_group<PP, DD, 8, 16>(a, B, seed, step, step_dev, st);             \
        return launch_bwd_group<PP, DD, 8, 8>(a, B, seed, step, step_dev, st);                                  \
    }
    M2M_BWDG_CASE(PREC_BF16, 32) M2M_BWDG_CASE(PREC_BF16, 64) M2M_BWDG_CASE(PREC_BF16, 128)
    M2M_BWDG_CASE(PREC_F32, 32) M2M_BWDG_CASE(PREC_F32, 64) M2M_BWDG_CASE(PREC_F32, 128)
#undef M2M_BWDG_CASE
    m2m_set_error("towers_backward: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_tower_backward(const m2m_tower* t, int B, const float* d_out, int64_t d_out_ss, const float* d_pooled,
                                  float* d_x0, int64_t d_x0_ss, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (int rc = m2m_check_tower(t, B)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
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
