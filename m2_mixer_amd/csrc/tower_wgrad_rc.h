// Channel-mixing weight gradients, RECOMPUTE form (bf16, hidden_dim 128): the hidden activation never leaves the chip.
//
//   dW1[c][d] = sum_m dHpre[m][c] A[m][d]      dW2[d][c] = sum_m dYd[m][d] Hact[m][c]      db1[c] = sum_m dHpre[m][c]
//   Hact[m][c] = dropout(gelu(A[m][:] W1[c][:] + b1[c]))                    (reference: modules/mixer.py:13-19, :37-40)
//
// The stored-operand form (tower_wgrad.hip) streams Hact^T and dHpre^T from HBM: 2 x rows x C bf16 per block, 300 MB per
// step of M2-Mixer-B at batch 512, written by the backward chains and read back here -- the launch ran at the HBM rate
// (559 MB at 4.9 TB/s) and the backward chains paid ~13 us for the transposes and stores.  Here a wave keeps the W1 rows of
// its 32 hidden columns in registers (the `w1n` fragments: 32 VGPRs), takes the packed NAT image of A = LN2(x_mid) -- 8 KB
// per 32 token rows, L2-resident, written by tower_bwd_body<HREC> where Hact^T used to go -- and recomputes
//       Hpre[m][c] = A[m][:] W1[c][:]      (16 MFMAs per 32-row step: A as the FIRST operand, so the accumulators hold
//                                           [m = 4g + r][c = il], which IS the chained operand layout "c = il, k = m" of the
//                                           gradient products -- no transpose, no LDS round trip)
// then bias / GELU (the forward's table) / keep-bit on the accumulators, and the same 32 + 32 gradient MFMAs as before.
// Only dHpre^T is still streamed (it needs dYd W2 as well: recomputing it too costs 16 more MFMAs and the GELU' epilogue
// per step, more than the 150 MB it saves are worth once the launch is no longer HBM-bound).
//
// Staging: the three shared 8 KB images of a step (A^T and dYd^T in chained k order for the gradient products, A in natural
// order for the recompute) go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging), double-buffered, ONE
// barrier per step; the wave's own dHpre^T fragments (2 KB per step, an HBM stream) go by LDS-DMA too, into a private ring of
// RC_NR slots, RC_NR - 1 steps ahead.  vmcnt bookkeeping: every memory operation of the loop is an inline-asm DMA, invisible to
// hipcc; per step a wave issues 6 image DMAs then 2 ring DMAs, so at the top of a step "all but the 2 youngest" (s_waitcnt
// vmcnt(2)) == this step's images have landed (own share; the barrier covers the other waves') and so has its ring slot,
// requested RC_NR - 1 steps ago.
#pragma once
#include "tile.h"

#ifndef RC_WAVES
#define RC_WAVES 4
#endif
#define RC_THREADS (RC_WAVES * 64)
#ifndef RC_NR
#define RC_NR 3           // LDS ring slots of a wave's dHpre^T stream: RC_NR - 1 steps in flight (3: 76 KB per workgroup, two per CU)
#endif
#ifndef RC_LA
#define RC_LA 3           // LDS fragments read ahead of their MFMAs
#endif
#ifndef RC_DB1_MFMA
#define RC_DB1_MFMA 0     // 1: db1 through an all-ones MFMA operand (12 VGPRs more), 0: v_dot2c_f32_bf16 on the fragments
#endif

template <int D> struct RcGeom {
    static constexpr int KD = D / 32, DT = D / 16;
    static constexpr int IMG_B = 32 * D * 2;                 // one 32-row bf16 image
    static constexpr int STAGE_B = 3 * IMG_B;                // A^T (CHN) | dYd^T (CHN) | A (NAT)
    static constexpr int PPI = IMG_B / 1024;                 // 1 KiB DMA pieces per image
    static constexpr int PPW = PPI / RC_WAVES;               // pieces per image and wave
    static constexpr int TAB_B = GELU_TAB_N * (int)sizeof(gtab2_t);
    static constexpr int TR_B = RC_WAVES * 16 * (D + 4) * 4; // write-out transposes (alias the stage)
    static constexpr int BODY_B = 2 * STAGE_B > TR_B ? 2 * STAGE_B : TR_B;
    static constexpr int RING_B = RC_WAVES * RC_NR * 2048;   // per wave RC_NR slots of its own dHpre^T fragments (2 KiB per step)
    static constexpr int LDS_B = BODY_B + TAB_B + RING_B;
    static constexpr int COLS = RC_WAVES * 32;
};

// LDS-DMA with a scalar base and a 32-bit per-lane offset (cf. glds16_g in common.h)
static __device__ __forceinline__ void glds16_sv(unsigned long long sbase, unsigned int voff, unsigned int ldst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldst) : "memory");
}
// the same, non-temporal: a stream that is read exactly once must not evict the images the other column slices re-read from L2
static __device__ __forceinline__ void glds16_sv_nt(unsigned long long sbase, unsigned int voff, unsigned int ldst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldst) : "memory");
}

struct WgOut {            // where a workgroup's results go
    float* w1; float* w2; float* b1;
    int mode;             // WG_OUT_*
};

// Shared by both forms.  The accumulators dw1[j][dt][r] = dW1[c = 16 (ct0 + j) + 4g + r][d = 16 dt + il], dw2 likewise =
// dW2[d][c]; the caller has passed a barrier since the last read of the LDS stage.
//   WG_OUT_ADD    "+=" onto the caller's gradient by its single owner (one row group): ALL loads of the old values first (they
//                 overlap each other; one dependent load-add-store at a time cost 40 % of the kernel), then the stores.
//   WG_OUT_STORE  "=" by a single owner: no loads at all.
//   WG_OUT_ATOMIC "+=" by several row groups: no-return float atomics onto the zeroed gradient (two groups: a + b == b + a,
//                 still bit-deterministic).  Device-scope float atomics sustain ~1 TB/s: fine for small launches, ruinous for a
//                 whole model's gradients.
// The accumulators hold one matrix index across lanes and the other in registers, the wrong way round for dW1's row-major
// layout (4-byte accesses in 4 to 64 segments per instruction cost 17 us per workgroup), so dW1's tiles are transposed through
// the (now free) LDS stage, each wave in its own part (one wave's LDS accesses complete in order: no further barrier); dW2
// has four consecutive c per lane already (16-byte accesses).
template <int D, int CPW>
static __device__ __forceinline__ void wgrad_write_w(f32x4_t (&dw1)[CPW][D / 16], f32x4_t (&dw2)[CPW][D / 16], const WgOut& out,
                                                     int ct0, int nct, int C, char* smem, int wave, int lane) {
    constexpr int DT = D / 16;
    const int g = lane >> 4, il = lane & 15, mode = out.mode;
    // the pointers in the global address space for the plain loads / stores (generic pointers from a descriptor in memory
    // give FLAT accesses, which wait on vmcnt AND lgkmcnt)
    typedef M2M_AS1 float* gf_t;
    typedef M2M_AS1 f32x4_t* gf4_t;
    float* const o_w1 = out.w1;
    float* const o_w2 = out.w2;
    const gf_t g_w1 = (gf_t)o_w1, g_w2 = (gf_t)o_w2;
    if (mode != WG_OUT_ATOMIC) {
        constexpr int TLD = D + 4;                           // padded row (floats): the four g-groups land in different banks
        float* tr = reinterpret_cast<float*>(smem) + wave * 16 * TLD;
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const int ct = ct0 + j;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) tr[(4 * g + r) * TLD + 16 * dt + il] = dw1[j][dt][r];
            if (ct < nct) {
                constexpr int PER = 16 * D / (64 * 4);       // float4 pieces per lane
                f32x4_t v[PER], old[PER];
                // the old values: UNCONDITIONAL loads (row clamped into the tensor), all requested before the first is used.
                // Guarded per piece, each load sat in its own basic block with a vmcnt(0) behind it.
                if (mode == WG_OUT_ADD) {
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int idx = i * 64 + lane, row = idx / (D / 4), c4 = idx % (D / 4);
                        const int c = min(16 * ct + row, C - 1);
                        old[i] = *(gf4_t)(g_w1 + (long)c * D + 4 * c4);
                    }
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int idx = i * 64 + lane, row = idx / (D / 4), c4 = idx % (D / 4);
                    v[i] = *reinterpret_cast<const f32x4_t*>(tr + row * TLD + 4 * c4);
                }
                if (mode == WG_OUT_ADD) {
#pragma unroll
                    for (int i = 0; i < PER; ++i) v[i] = v[i] + old[i];
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int idx = i * 64 + lane, row = idx / (D / 4), c4 = idx % (D / 4);
                    const int c = 16 * ct + row;
                    if (c < C) *(gf4_t)(g_w1 + (long)c * D + 4 * c4) = v[i];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            if (ct0 + j >= nct) continue;
            const int c0 = 16 * (ct0 + j) + 4 * g;
            const bool vec = c0 + 3 < C && (C & 3) == 0;
            if ((C & 3) == 0 && 16 * (ct0 + j) + 16 <= C) {   // wave-uniform: the whole column tile is inside the tensor
                if (mode == WG_OUT_ADD) {
                    f32x4_t o2[DT];                           // unconditional, batched (see above)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) o2[dt] = *(gf4_t)(g_w2 + (long)(16 * dt + il) * C + c0);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) dw2[j][dt] = dw2[j][dt] + o2[dt];
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) *(gf4_t)(g_w2 + (long)(16 * dt + il) * C + c0) = dw2[j][dt];
                continue;
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = 16 * dt + il;
                float* p2 = o_w2 + (long)d * C + c0;          // four consecutive c of row d
                if (vec) {
                    if (mode == WG_OUT_ADD) dw2[j][dt] = dw2[j][dt] + *reinterpret_cast<const f32x4_t*>(p2);
                    *(gf4_t)(g_w2 + (long)d * C + c0) = dw2[j][dt];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c0 + r < C) p2[r] = (mode == WG_OUT_ADD ? p2[r] : 0.f) + dw2[j][dt][r];
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            if (ct0 + j >= nct) continue;
            const int c0 = 16 * (ct0 + j) + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = 16 * dt + il;                  // 16 consecutive d of one row in 16 lanes
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + r < C) atomicAdd(o_w1 + (long)(c0 + r) * D + d, dw1[j][dt][r]);
            }
        }
        // dW2[d][c]: each wave transposes its (d x 16 CPW) slice, CHT d-tiles at a time, and adds 16 CPW consecutive
        // columns of a row per 16 CPW lanes.
        constexpr int W = 16 * CPW, TLD2 = W + 1, RPI = 64 / W;     // columns per wave, padded LDS row, rows per instruction
        constexpr int CAP = 16 * (D + 4) / (16 * TLD2);             // d-tiles the wave's LDS part holds
        constexpr int CHT = CAP >= 4 ? 4 : (CAP >= 2 ? 2 : 1);
        static_assert(DT % CHT == 0 && CAP >= 1, "dW2 transpose chunks");
        float* tr = reinterpret_cast<float*>(smem) + wave * 16 * (D + 4);
        const int cw = 16 * ct0 + (lane % W);                // this lane's column in the transposed read
#pragma unroll
        for (int ch = 0; ch < DT / CHT; ++ch) {
#pragma unroll
            for (int dtl = 0; dtl < CHT; ++dtl)
#pragma unroll
                for (int j = 0; j < CPW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tr[(16 * dtl + il) * TLD2 + 16 * j + 4 * g + r] = dw2[j][CHT * ch + dtl][r];
#pragma unroll
            for (int i = 0; i < 16 * CHT / RPI; ++i) {
                const int row = RPI * i + lane / W;
                const float v = tr[row * TLD2 + lane % W];
                if (cw < C) atomicAdd(o_w2 + (long)(16 * CHT * ch + row) * C + cw, v);
            }
        }
    }
}

template <int D, int DM>
static __device__ __forceinline__ void wgrad_rc_body(const m2m_block& bk, const WgOut& out, int Cp, int C, int slice, int group,
                                                     int ntiles, int tiles_per_group, int rows_per_t16, unsigned int drop_key,
                                                     float drop_scale, char* smem) {
    typedef Prec<PREC_BF16> Pr;
    typedef RcGeom<D> G;
    constexpr int KD = G::KD, DT = G::DT, IMG_B = G::IMG_B, STAGE_B = G::STAGE_B, PPW = G::PPW, CPW = 2;
    static_assert(DM == DM_NONE || DM == DM_HALF, "general-p dropout keeps the stored-operand form");
    static_assert(G::PPI % RC_WAVES == 0, "every wave issues the same number of DMAs (vmcnt bookkeeping)");

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, il = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nct = Cp >> 4;
    const int q = slice * RC_WAVES + wave;                  // this wave's 32-column group (the dropout word index)
    const int ct0 = 2 * q;
    const int t_begin = group * tiles_per_group;
    const int t_end = min(ntiles, t_begin + tiles_per_group);
    if (t_begin >= t_end) return;
    TIMER_WG_BEGIN();
    TIMER_LSTART();
    int ctl[CPW];                                           // column tiles past the end (last slice) shadow the last one
#pragma unroll
    for (int j = 0; j < CPW; ++j) ctl[j] = min(ct0 + j, nct - 1);

    gtab2_t* gtab = reinterpret_cast<gtab2_t*>(smem + (2 * STAGE_B > G::TR_B ? 2 * STAGE_B : G::TR_B));
    gelu_tab2_fill(gtab, drop_scale, tid, RC_THREADS);

    // ---- per-launch constants of the wave: its W1 rows (second operand of the recompute) and hidden bias ----
    Frag w1f[CPW][KD];
    float b1v[CPW];
    {
        const gptr_t p_w1n = to_gptr(bk.w1n);
        const M2M_AS1 float* b1p = reinterpret_cast<const M2M_AS1 float*>(to_gptr(bk.ch_b1p));
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) w1f[j][kb] = ld_frag_global_u(p_w1n, (long)ctl[j] * KD + kb, (unsigned int)lane * 16u);
            b1v[j] = b1p[16 * ctl[j] + il];
        }
    }

    // ---- streams ----
    const unsigned long long s_at = uniform_u64((unsigned long long)bk.at_chn), s_dyt = uniform_u64((unsigned long long)bk.dyt_chn),
                             s_an = uniform_u64((unsigned long long)bk.h_chn);       // (HREC: h_chn holds the NAT image of A)
    const unsigned int sbase = lds_addr_of(smem);
    const unsigned int voff = (unsigned int)(wave * PPW) * 1024u + (unsigned int)lane * 16u;
    auto stage = [&](int tile, int buf) {                   // this wave's share of a step's three images: 3 x PPW DMAs
        const unsigned long long o = (unsigned long long)tile * IMG_B;
#pragma unroll
        for (int img = 0; img < 3; ++img) {
            const unsigned long long src = (img == 0 ? s_at : (img == 1 ? s_dyt : s_an)) + o;
#pragma unroll
            for (int h = 0; h < PPW; ++h)
                glds16_sv(src, voff + h * 1024u,
                          __builtin_amdgcn_readfirstlane(sbase + buf * STAGE_B + img * IMG_B + (wave * PPW + h) * 1024));
        }
    };
    // dHpre^T fragments of the wave's column-tile pair: [pair q][32-row tile][16-row half][lane][tile 2q: 8 B | tile 2q+1: 8 B]
    const unsigned long long s_dh = uniform_u64((unsigned long long)bk.dh_chn + (unsigned long long)(ctl[0] >> 1) * m2m_hchn_stride(ntiles));
    const unsigned int lane16 = (unsigned int)lane * 16u;
    const unsigned int ring_lds = sbase + G::BODY_B + G::TAB_B + (unsigned int)wave * (RC_NR * 2048);
    const char* ring_ptr = smem + G::BODY_B + G::TAB_B + wave * (RC_NR * 2048);
    auto ring_load = [&](int tile, int slot) {               // 2 DMAs: the two 16-row halves
        const unsigned long long src = s_dh + (unsigned long long)tile * 2048;
        glds16_sv_nt(src, lane16, __builtin_amdgcn_readfirstlane(ring_lds + slot * 2048));
        glds16_sv_nt(src, lane16 + 1024u, __builtin_amdgcn_readfirstlane(ring_lds + slot * 2048 + 1024));
    };

    f32x4_t dw1[CPW][DT], dw2[CPW][DT];
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dw1[j][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dw2[j][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#if RC_DB1_MFMA
    f32x4_t db1[CPW] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    Frag ones;
    ones.u = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
#else
    float db1[CPW] = {0.f, 0.f};                            // this lane's rows only; summed over the four lane groups at the end
#endif

    const unsigned int ngroups32 = (unsigned int)(Cp >> 5);
    // (the keep-word of hidden-column group q for token row m is mix32(key ^ (m * ngroups32 + q)): tile.h, drop_word_half)

    // One 32-row step.  The LDS fragment reads run RC_LA fragments ahead of the MFMAs that consume them (left to hipcc every read
    // was waited for at once: ds_read, lgkmcnt(0), two MFMAs, ...), and the GELU epilogue of accumulator row (mt, r) is written
    // between the dW1 MFMAs of d-tile 4 mt + r -- those do not depend on the recompute -- so that a wave alone on its SIMD
    // overlaps its own vector work with its own matrix work.
    auto step = [&](int slot, int tile, int buf) {
        const char* cur = smem + buf * STAGE_B;
        constexpr int LA = RC_LA;
        Frag df[CPW];
        {
            const u32x4_t d0 = *reinterpret_cast<const u32x4_t*>(ring_ptr + slot * 2048 + lane * 16);
            const u32x4_t d1 = *reinterpret_cast<const u32x4_t*>(ring_ptr + slot * 2048 + 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < CPW; ++j) df[j].u = u32x4_t{d0[2 * j], d0[2 * j + 1], d1[2 * j], d1[2 * j + 1]};
        }
        // ---- recompute Hpre (bias in the accumulators): fragment q = (kb, mt) ----
        f32x4_t hacc[2][CPW];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < CPW; ++j) hacc[mt][j] = f32x4_t{b1v[j], b1v[j], b1v[j], b1v[j]};
        {
            constexpr int NQ = 2 * KD, L = LA < NQ ? LA : NQ;
            Frag aq[L];
#pragma unroll
            for (int q = 0; q < L; ++q) aq[q] = ld_frag_lds(cur + 2 * IMG_B, (q & 1) * KD + (q >> 1), lane);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const Frag a = aq[q % L];
                if (q + L < NQ) aq[q % L] = ld_frag_lds(cur + 2 * IMG_B, ((q + L) & 1) * KD + ((q + L) >> 1), lane);
#pragma unroll
                for (int j = 0; j < CPW; ++j) Pr::mma(hacc[q & 1][j], a, w1f[j][q >> 1]);
            }
        }
        // ---- db1 (this lane's rows) ----
#if RC_DB1_MFMA
#pragma unroll
        for (int j = 0; j < CPW; ++j) Pr::mma(db1[j], df[j], ones);
#else
        {
            // plain unpack + add
#pragma unroll
            for (int j = 0; j < CPW; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned int u = df[j].u[e];
                    db1[j] += __builtin_bit_cast(float, u << 16) + __builtin_bit_cast(float, u & 0xFFFF0000u);
                }
        }
#endif
        // ---- dW1 += dHpre^T A, with the GELU + dropout epilogue of the recompute between its MFMAs (they do not depend on it).
        //      Element (m = 32 tile + 16 mt + 4g + r, c = 32 q + 16 j + il).  Every LDS round trip of the epilogue is issued as a
        //      BATCH and consumed one MFMA group later: written row by row (keep-word -> index -> table read -> fma, each step
        //      waiting for the previous one) the 8 rows cost 16 exposed LDS latencies per step -- 1.1 us of a 1.5 us step.
        //      Keep-words without LDS: lane (g, n) computes the word of row (mt = n >> 2, r = n & 3) of ITS lane group g (n < 8),
        //      and a DPP row broadcast (row_newbcast:n, VALU) hands it to the group's 16 lanes, shifted to the lane's bit. ----
        unsigned int wsh[8];
        if (DM == DM_HALF) {
            const unsigned int m = (unsigned int)((2 * tile + ((il >> 2) & 1)) * rows_per_t16 + 4 * g + (il & 3));
            const unsigned int myword = mix32(drop_key ^ (m * ngroups32 + (unsigned int)q));
#define RC_BCAST(n) wsh[n] = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)myword, 0x150 + n, 0xF, 0xF, false) >> il;
            RC_BCAST(0) RC_BCAST(1) RC_BCAST(2) RC_BCAST(3) RC_BCAST(4) RC_BCAST(5) RC_BCAST(6) RC_BCAST(7)
#undef RC_BCAST
        }
        {
            constexpr int L = LA < DT ? LA : DT;
            Frag tq[L];
#pragma unroll
            for (int dt = 0; dt < L; ++dt) tq[dt] = ld_frag_lds(cur, dt, lane);
            gtab2_t te[4][CPW];                              // table entries of one 16-row tile in flight
            auto lookup = [&](int mt) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int j = 0; j < CPW; ++j) {
                        unsigned int idx = pwl_index(hacc[mt][j][rr]);
                        if (DM == DM_HALF) idx &= (unsigned int)(((int)(wsh[4 * mt + rr] << (31 - 16 * j))) >> 31);   // dropped: cell 0 = {0, 0}
                        te[rr][j] = gtab[idx];
                    }
            };
            auto apply = [&](int mt) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int j = 0; j < CPW; ++j) hacc[mt][j][rr] = __builtin_fmaf(te[rr][j][1], hacc[mt][j][rr], te[rr][j][0]);
            };
            lookup(0);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const Frag at = tq[dt % L];
                if (dt + L < DT) tq[dt % L] = ld_frag_lds(cur, dt + L, lane);
#pragma unroll
                for (int j = 0; j < CPW; ++j) Pr::mma(dw1[j][dt], df[j], at);
                if (dt == DT / 2 - 1) { apply(0); lookup(1); }
            }
            apply(1);
        }
        Frag hf[CPW];
#pragma unroll
        for (int j = 0; j < CPW; ++j) Chain<PREC_BF16>::make(hacc[0][j], hacc[1][j], &hf[j]);
        // ---- dW2^T += Hact^T dYd ----
        {
            constexpr int L = LA < DT ? LA : DT;
            Frag yq[L];
#pragma unroll
            for (int dt = 0; dt < L; ++dt) yq[dt] = ld_frag_lds(cur + IMG_B, dt, lane);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const Frag dyt = yq[dt % L];
                if (dt + L < DT) yq[dt % L] = ld_frag_lds(cur + IMG_B, dt + L, lane);
#pragma unroll
                for (int j = 0; j < CPW; ++j) Pr::mma(dw2[j][dt], hf[j], dyt);
            }
        }
    };

    // ---- pipeline ----
    // issue order per step i (tile T_i): [wait: images of T_i + ring slot of T_i landed] barrier | image DMAs of T_{i+1} (3 PPW) |
    // ring DMAs of T_{i+RC_NR-1} (2) | compute(T_i).  One barrier per step: stage buffer (i + 1) & 1 was last read in step i - 1,
    // which every wave has left when it arrives at the barrier of step i; the ring is private to the wave, and the slot refilled
    // in step i is the one step i - 1 consumed.  All DMAs are unconditional (tile clamped): a fixed number per step keeps the
    // counts.
    static_assert(RC_NR >= 2, "ring slots");
    stage(t_begin, 0);
#pragma unroll
    for (int k = 0; k < RC_NR - 1; ++k) ring_load(min(t_begin + k, t_end - 1), k);
    __syncthreads();                                        // GELU table visible (drains the prologue DMAs once: fine)
    TIMER_LMARK(0);       // prologue: W1 fragments, table, first DMAs
    int it = 0;
    for (int tile = t_begin; tile < t_end; ++tile, ++it) {
        // everything but the youngest two DMAs (the ring slot of a later step) has arrived
        if (RC_NR >= 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TIMER_LMARK(1);   // wait for this step's DMAs
        __builtin_amdgcn_s_barrier();
        TIMER_LMARK(2);   // barrier
        stage(min(tile + 1, t_end - 1), (it + 1) & 1);
        ring_load(min(tile + RC_NR - 1, t_end - 1), (it + RC_NR - 1) % RC_NR);
        TIMER_LMARK(3);   // DMA issue
        step(it % RC_NR, tile, it & 1);
        TIMER_LMARK(4);   // compute
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the redundant tail DMAs must not land in the transposes below
    __syncthreads();                                        // every wave is done with the stage

    // ---- results ----
#if !RC_DB1_MFMA
    // db1: the four lane groups' partial sums of column c = 16 (ct0 + j) + il, added in straight-line code right behind the loop
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        db1[j] += __shfl_xor(db1[j], 16, 64);
        db1[j] += __shfl_xor(db1[j], 32, 64);
    }
#endif
    wgrad_write_w<D, CPW>(dw1, dw2, out, ct0, nct, C, smem, wave, lane);
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        if (ct0 + j >= nct) continue;
#if RC_DB1_MFMA
        // db1[j][r]: column c = 16 (ct0 + j) + 4g + r, identical in all 16 lanes il
        if (il == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * (ct0 + j) + 4 * g + r;
                if (c >= C) continue;
                if (out.mode == WG_OUT_ATOMIC) atomicAdd(out.b1 + c, db1[j][r]);
                else if (out.mode == WG_OUT_ADD) out.b1[c] += db1[j][r];
                else out.b1[c] = db1[j][r];
            }
        }
#else
        const int c = 16 * (ct0 + j) + il;
        const float s = db1[j];
        if (g == 0 && c < C) {
            if (out.mode == WG_OUT_ATOMIC) atomicAdd(out.b1 + c, s);
            else if (out.mode == WG_OUT_ADD) out.b1[c] += s;
            else out.b1[c] = s;
        }
#endif
    }
    TIMER_LMARK(5);       // write-out
    TIMER_LFLUSH(g_tm_wg);
    TIMER_WG_END(g_tm_wg);
}


// ---- stored-operand form with every stream on LDS-DMA (bf16, hidden_dim 128) ------------------------------------------------
// The same contraction as wgrad_body (tower_wgrad.hip: both hidden operands stored by the backward chain and streamed here), but
// NOTHING of a step goes through VGPRs on its way in: the two shared images (A^T, dYd^T: 16 KB per 32 token rows) are staged
// global -> LDS by LDS-DMA, double-buffered, one barrier per step, and each wave's own Hact^T / dHpre^T fragments (4 KB per step:
// an HBM stream, read once) go by LDS-DMA into a private ring of DR_NR slots, DR_NR - 1 steps ahead.  The register-staged form
// holds ONE step of loads in flight per wave (32 VGPRs: there is no room for a second) -- ~40 KB per CU, which at the ~2 us an
// HBM access takes under load caps the CU near 20 GB/s: the phase timers showed 27 us of LDS + MFMA work per workgroup inside
// ~110 us of waiting for loads and barriers.  Here three steps (12 KB per wave, 60 KB per CU) are in flight and cost no
// registers; the price is the DMA issue (~75 cycles per 1 KB piece, 8 pieces per wave and step).
// vmcnt bookkeeping (every memory operation of the loop is an inline-asm DMA, invisible to hipcc): per step a wave issues 4 image
// DMAs then 4 ring DMAs; at the top of a step "all but the 4 youngest" have landed = this step's images (own share; the barrier
// covers the other waves') and its ring slot.
#ifndef DR_NR
#define DR_NR 4
#endif
template <int D, int NW> struct DrGeom {
    static constexpr int DT = D / 16;
    static constexpr int IMG_B = 32 * D * 2;                 // one 32-row bf16 image
    static constexpr int STAGE_B = 2 * IMG_B;                // A^T | dYd^T
    static constexpr int NPIECE = STAGE_B / 1024;            // 1 KiB DMA pieces per step
    static constexpr int PPW = (NPIECE + NW - 1) / NW;       // pieces per wave (the last wave repeats the final piece)
    static constexpr int TR_B = NW * 16 * (D + 4) * 4;       // write-out transposes (alias the stage)
    static constexpr int BODY_B = 2 * STAGE_B > TR_B ? 2 * STAGE_B : TR_B;
    static constexpr int RING_B = NW * DR_NR * 4096;
    static constexpr int LDS_B = BODY_B + RING_B;
    static constexpr int COLS = NW * 32;
};

template <int D, int NW>
static __device__ __forceinline__ void wgrad_dma_body(const m2m_block& bk, const WgOut& out, int Cp, int C, int slice, int group,
                                                      int ntiles, int tiles_per_group, char* smem) {
    typedef Prec<PREC_BF16> Pr;
    typedef DrGeom<D, NW> G;
    constexpr int DT = G::DT, IMG_B = G::IMG_B, STAGE_B = G::STAGE_B, PPW = G::PPW, CPW = 2;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, il = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nct = Cp >> 4;
    const int q = slice * NW + wave;                        // this wave's 32-column group
    const int ct0 = 2 * q;
    const int t_begin = group * tiles_per_group;
    const int t_end = min(ntiles, t_begin + tiles_per_group);
    if (t_begin >= t_end) return;
    TIMER_WG_BEGIN();
    TIMER_LSTART();
    const int qc = min(q, (nct >> 1) - 1);                  // column groups past the end (last slice) shadow the last one

    const unsigned long long s_at = uniform_u64((unsigned long long)bk.at_chn), s_dyt = uniform_u64((unsigned long long)bk.dyt_chn);
    const unsigned long long hoff = (unsigned long long)qc * m2m_hchn_stride(ntiles);
    const unsigned long long s_h = uniform_u64((unsigned long long)bk.h_chn + hoff), s_dh = uniform_u64((unsigned long long)bk.dh_chn + hoff);
    const unsigned int sbase = lds_addr_of(smem);
    const unsigned int lane16 = (unsigned int)lane * 16u;
    auto stage = [&](int tile, int buf) {                   // this wave's PPW pieces of the step's two images
        const unsigned long long o = (unsigned long long)tile * IMG_B;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int piece = min(wave * PPW + k, G::NPIECE - 1);             // wave-uniform; [A^T: 0 .. IMG_B/1024) | dYd^T
            const int img = piece >= IMG_B / 1024 ? 1 : 0, off = (piece - img * (IMG_B / 1024)) * 1024;
            glds16_sv((img ? s_dyt : s_at) + o, lane16 + (unsigned int)off,
                      __builtin_amdgcn_readfirstlane(sbase + buf * STAGE_B + piece * 1024));
        }
    };
    // the wave's own fragments: [pair q][32-row tile][16-row half][lane][tile 2q: 8 B | tile 2q+1: 8 B], Hact^T and dHpre^T
    const unsigned int ring_lds = sbase + G::BODY_B + (unsigned int)wave * (DR_NR * 4096);
    const char* ring_ptr = smem + G::BODY_B + wave * (DR_NR * 4096);
    auto ring_load = [&](int tile, int slot) {              // 4 DMAs: h half 0, h half 1, dh half 0, dh half 1
        const unsigned long long o = (unsigned long long)tile * 2048;
        const unsigned int dst = ring_lds + slot * 4096;
        glds16_sv_nt(s_h + o, lane16, __builtin_amdgcn_readfirstlane(dst));
        glds16_sv_nt(s_h + o, lane16 + 1024u, __builtin_amdgcn_readfirstlane(dst + 1024));
        glds16_sv_nt(s_dh + o, lane16, __builtin_amdgcn_readfirstlane(dst + 2048));
        glds16_sv_nt(s_dh + o, lane16 + 1024u, __builtin_amdgcn_readfirstlane(dst + 3072));
    };

    f32x4_t dw1[CPW][DT], dw2[CPW][DT], db1[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        db1[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dw1[j][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dw2[j][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    Frag ones;
    ones.u = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    auto step = [&](int slot, int buf) {
        const char* cur = smem + buf * STAGE_B;
        const char* rs = ring_ptr + slot * 4096 + lane * 16;
        Frag hf[CPW], df[CPW];
        {
            const u32x4_t h0 = *reinterpret_cast<const u32x4_t*>(rs), h1 = *reinterpret_cast<const u32x4_t*>(rs + 1024);
            const u32x4_t d0 = *reinterpret_cast<const u32x4_t*>(rs + 2048), d1 = *reinterpret_cast<const u32x4_t*>(rs + 3072);
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                hf[j].u = u32x4_t{h0[2 * j], h0[2 * j + 1], h1[2 * j], h1[2 * j + 1]};
                df[j].u = u32x4_t{d0[2 * j], d0[2 * j + 1], d1[2 * j], d1[2 * j + 1]};
            }
        }
        constexpr int LA = RC_LA < DT ? RC_LA : DT;
        Frag aq[LA], yq[LA];
#pragma unroll
        for (int dt = 0; dt < LA; ++dt) { aq[dt] = ld_frag_lds(cur, dt, lane); yq[dt] = ld_frag_lds(cur + IMG_B, dt, lane); }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const Frag at = aq[dt % LA], dyt = yq[dt % LA];
            if (dt + LA < DT) { aq[dt % LA] = ld_frag_lds(cur, dt + LA, lane); yq[dt % LA] = ld_frag_lds(cur + IMG_B, dt + LA, lane); }
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                Pr::mma(dw1[j][dt], df[j], at);
                Pr::mma(dw2[j][dt], hf[j], dyt);
            }
        }
#pragma unroll
        for (int j = 0; j < CPW; ++j) Pr::mma(db1[j], df[j], ones);      // every column = sum over the tile's rows of dHpre[.][c]
    };

    static_assert(DR_NR >= 3, "ring slots: the vmcnt(4) below assumes the slot of a step was requested at least two steps ago");
    stage(t_begin, 0);
#pragma unroll
    for (int k = 0; k < DR_NR - 1; ++k) ring_load(min(t_begin + k, t_end - 1), k);
    int it = 0;
    TIMER_LMARK(0);       // prologue
    for (int tile = t_begin; tile < t_end; ++tile, ++it) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // (first step: the prologue's ring DMAs of the later slots stay in flight)
        TIMER_LMARK(1);   // wait for this step's DMAs
        __builtin_amdgcn_s_barrier();
        TIMER_LMARK(2);   // barrier
        stage(min(tile + 1, t_end - 1), (it + 1) & 1);
        ring_load(min(tile + DR_NR - 1, t_end - 1), (it + DR_NR - 1) % DR_NR);
        TIMER_LMARK(3);   // DMA issue
        step(it % DR_NR, it & 1);
        TIMER_LMARK(4);   // LDS reads + MFMAs
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the redundant tail DMAs must not land in the transposes below
    __syncthreads();
    wgrad_write_w<D, CPW>(dw1, dw2, out, ct0, nct, C, smem, wave, lane);
    // db1[j][r]: column c = 16 (ct0 + j) + 4g + r, identical in all 16 lanes il: lane il == 0 writes
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        if (ct0 + j >= nct || il != 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * (ct0 + j) + 4 * g + r;
            if (c >= C) continue;
            if (out.mode == WG_OUT_ATOMIC) atomicAdd(out.b1 + c, db1[j][r]);
            else if (out.mode == WG_OUT_ADD) out.b1[c] += db1[j][r];
            else out.b1[c] = db1[j][r];
        }
    }
    TIMER_LMARK(5);       // write-out
    TIMER_LFLUSH(g_tm_wg);
    TIMER_WG_END(g_tm_wg);
}
