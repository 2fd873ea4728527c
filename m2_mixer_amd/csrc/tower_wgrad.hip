// Channel-mixing weight gradients of every block of a tower: g_ch_w1, g_ch_b1, g_ch_w2.
//
//   dW1[c][d] = sum_m dHpre[m][c] A[m][d]      dW2[d][c] = sum_m dYd[m][d] Hact[m][c]      db1[c] = sum_m dHpre[m][c]
//
// The contraction runs over ALL token rows, so the roles flip relative to the forward/backward chain: a workgroup of
// FOUR waves owns 128 hidden columns of one block (each wave 32 = two 16-column tiles; the wave's 32x128 slices of dW1
// and dW2^T stay in registers) and streams the operands tower_bwd.hip stored, 32 token rows per step:
//   A^T, dYd^T   (D x 32, shared by the whole workgroup)  -> double-buffered LDS stage; each ds_read_b128 fragment feeds
//                                                            both column tiles of the wave (two MFMAs per LDS read);
//   dHpre^T, Hact^T (this wave's 32 x 32)                  -> straight from global into registers.
// Nothing is recomputed.  Two such workgroups share a CU (<= 256 VGPRs, 32 KiB LDS each): while one sits at its
// per-tile barrier or waits for loads the other computes, and together with the register ring (WG_DEPTH tiles of
// loads in flight per workgroup, counted s_waitcnt) that covers the memory latency.
// With one row group every result element has a single owner and is written without atomics; two groups add onto the
// zeroed gradient with float atomics (a + b == b + a: still bit-deterministic).
#include "tile.h"
#include "embed_wgrad.h"
#include "split.h"
#include <stdlib.h>
#include <string.h>
bool m2m_split_eligible(const m2m_tower* t, int B, int training);     // split_api.hip

// address-space qualifier for pointers known to be global memory (device pass only; the host pass just parses)
#if defined(__HIP_DEVICE_COMPILE__)
#define M2M_GLOBAL_AS __attribute__((address_space(1)))
#else
#define M2M_GLOBAL_AS
#endif

#define WBM 32            // rows per streamed tile (= two 16-row tiles of the chain kernels when BM == 16)
#ifndef WG_RING_MAX
#define WG_RING_MAX 24    // VGPR budget of one tile in flight that still allows a second one (register ring depth 2)
#endif
#define WG_WAVES 4        // waves of a tower workgroup, except:
#ifndef WG_WAVES_WIDE
#define WG_WAVES_WIDE 5   // bf16, hidden_dim 128 (the stored-operand form of M2-Mixer-B): 160 hidden columns per workgroup -> 240
#endif                    // workgroups for the model, ONE per CU, all of the same length (towers alone: 102 -> 80 us)
#ifndef WG_LA
#define WG_LA 4           // LDS fragments read ahead of their MFMAs
#endif
#ifndef WG_MINWAVES
#define WG_MINWAVES 2
#endif
#define WG_OUT_ATOMIC 0   // how a workgroup hands over its results (wgrad_write_w)
#define WG_OUT_ADD 1
#define WG_OUT_STORE 2    // "=": single owner, the old values are not read (m2m_tower.wgrad_flags & M2M_WGRAD_OVERWRITE, and the
                          //      second row group of a tower with a partial-gradient slot, m2m_tower.wslot)

TIMER_DECL(g_tm_wg);
TIMER_READER(m2m_debug_timers_wgrad, g_tm_wg)

#include "tower_wgrad_rc.h"   // WgOut, wgrad_write_w, the recompute form (bf16, hidden_dim 128)

// Round 4 experiment, bf16 / hidden_dim 128 (-DM2M_WG48=1; OFF): FOUR waves x THREE 16-column tiles = 192 hidden columns per
// workgroup, one wave per SIMD with up to 512 registers and a deeper register ring.  Idea: the five-wave form puts two waves on
// SIMD 0 (2 x 36 MFMAs per 32-row step = 0.55 us against 0.28 us on the other SIMDs); here every SIMD issues 54 MFMAs per step
// and the 196 workgroups of M2-Mixer-B leave 60 CUs to the embedding workgroups.  MEASURED (parity-green, two interleaved
// repetitions in one process): merged launch 132-134 us (embedding workgroups first) / 116 us (last) against 105 us for the
// five-wave form, ring depth 1 or 2 alike -- the step is not paced by SIMD 0's MFMAs but by the latency chain stage write ->
// barrier -> LDS fragment reads -> MFMAs, which a lone wave per SIMD hides worse than five waves on four SIMDs do; the L2's
// memory-side queue shows ~1200 cycles per read for this launch (profiles/r04_ea_read_latency.txt: HBM, not Infinity Cache).
#ifndef M2M_WG48
#define M2M_WG48 0
#endif
#ifndef WG48_DEPTH
#define WG48_DEPTH 2
#endif
template <int P, int D> struct WgradGeom {
    static constexpr int NF = Chain<P>::NF;
    static constexpr bool W48 = M2M_WG48 && P == PREC_BF16 && D == 128;
    static constexpr int WAVES = W48 ? 4 : ((P == PREC_BF16 && D == 128) ? WG_WAVES_WIDE : WG_WAVES);
    static constexpr int THREADS = WAVES * 64;
#ifdef WG_CPW_FORCE
    static constexpr int CPW = WG_CPW_FORCE;
#else
    static constexpr int CPW = W48 ? 3 : ((P == PREC_BF16 && D <= 128) ? 2 : 1);     // 16-column tiles per wave
#endif
    static constexpr int IMG_B = WBM * D * Prec<P>::ESZ;
    static constexpr int STAGE_B = 2 * IMG_B;                                          // A^T | dYd^T of one tile
    static constexpr int NLD = (STAGE_B + THREADS * 16 - 1) / (THREADS * 16);   // 16-byte pieces per thread per tile
    // token tiles per step (= per barrier).  Measured on M2-Mixer-B: 4 tiles per step with one workgroup per CU (the next
    // step's 128 KiB of loads in flight in up to 512 VGPRs) was SLOWER (230 vs 160 us for the three towers) than one
    // tile per step with two workgroups per CU, so 1 is the default; the knob stays for other shapes.
#ifdef WG_TPS_FORCE
    static constexpr int TPS = (P == PREC_BF16 && D <= 128) ? WG_TPS_FORCE : 1;
#else
    static constexpr int TPS = 1;
#endif
    static constexpr int RING_REGS = NLD * 4 + CPW * 2 * NF * 4;                      // VGPRs of one tile in flight
    static constexpr int DEPTH = W48 ? WG48_DEPTH : ((TPS == 1 && RING_REGS <= WG_RING_MAX) ? 2 : 1);      // steps of loads in flight
    static constexpr int COLS = WAVES * CPW * 16;                                   // hidden columns per workgroup
    static constexpr int TR_B = WAVES * 16 * (D + 4) * 4;                          // dW1 write-out transpose: 16 x (D + 4) floats per wave
    static constexpr int LDS_B = 2 * TPS * STAGE_B > TR_B ? 2 * TPS * STAGE_B : TR_B;  // dynamic LDS of the kernel
    static constexpr int MINWAVES = (W48 || TPS > 2) ? 1 : 2;                          // waves per SIMD the kernel is built for
    // the wave's own Hact^T / dHpre^T fragments (the HBM stream: read once, ~1200 cycles per read under load) requested TWO
    // steps ahead, the shared stage (L2-resident images) one step ahead: 16 registers more than a ring of one step
#ifndef M2M_WG_HD2
#define M2M_WG_HD2 1
#endif
    static constexpr bool HD2 = M2M_WG_HD2 && !W48 && P == PREC_BF16 && D == 128 && TPS == 1 && DEPTH == 1;
    static_assert(TPS == 1 || DEPTH == 1, "multi-tile steps use a ring of one step");
};

// One workgroup's share: column slice `slice` of block `bk`, token tiles [group * tiles_per_group, ...).
template <int P, int D>
static __device__ __forceinline__ void wgrad_body(const m2m_block& bk, const WgOut& out, int Cp, int C, int slice, int group,
                                                  int ntiles, int tiles_per_group, char* smem) {
    typedef Prec<P> Pr;
    typedef WgradGeom<P, D> G;
    constexpr int DT = D / 16, NF = G::NF, CPW = G::CPW, IMG_B = G::IMG_B, STAGE_B = G::STAGE_B, NLD = G::NLD, DEPTH = G::DEPTH,
                  TPS = G::TPS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    const int nct = Cp >> 4;
    const int ct0 = (slice * G::WAVES + wave) * CPW;        // this wave's first 16-column tile

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
    if (P == PREC_BF16) ones.u = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    else ones.f = f32x4_t{1.f, 1.f, 1.f, 1.f};

    // Explicit global address space: in the multi-tower launch the descriptor is read from memory, so the compiler
    // cannot prove where its pointers point and would emit FLAT loads -- which also count on lgkmcnt, so that every wait
    // for an LDS fragment would drain the global prefetch too.
    typedef const M2M_GLOBAL_AS char* gptr_t;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    const gptr_t src_at = (gptr_t)bk.at_chn, src_dyt = (gptr_t)bk.dyt_chn, src_h = (gptr_t)bk.h_chn, src_dh = (gptr_t)bk.dh_chn;
    // One tile's worth of loads in flight: this thread's pieces of the shared stage + this wave's own fragments.
    struct Pre {
        u32x4_t st[TPS][NLD];
        Frag h[TPS][CPW][NF], d[TPS][CPW][NF];
    };
    // Every load below is unconditional (indices clamped into range, a few redundant loads at the tail): with a fixed
    // number of loads per step the compiler can place counted s_waitcnt vmcnt(N) and really keep DEPTH tiles in flight;
    // any guard around a load makes it fall back to vmcnt(0) at the top of the loop.
    TIMER_START();
    int ctl[CPW];                                           // column tiles past the end (last slice) shadow the last one
#pragma unroll
    for (int j = 0; j < CPW; ++j) ctl[j] = min(ct0 + j, nct - 1);
    auto tile_load = [&](Pre& p, int tile0, int t_end) {
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            const int tile = min(tile0 + u, t_end - 1);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int o = min((i * G::THREADS + tid) * 16, STAGE_B - 16);
                const gptr_t sp = o < IMG_B ? src_at : src_dyt;
                p.st[u][i] = *(const M2M_GLOBAL_AS u32x4_t*)(sp + (long)tile * IMG_B + (o < IMG_B ? o : o - IMG_B));
            }
            if (P == PREC_BF16) {
                // [column-tile pair][32-row pair][16-row half][lane][tile 2q: 8 B | tile 2q+1: 8 B]  (written by tower_bwd.hip)
                // read exactly once: non-temporal, so the stream does not evict the A^T / dYd^T tiles that the other column
                // slices of this block re-read from L2
                const long blk = (long)(ctl[0] >> 1) * m2m_hchn_stride(ntiles) + (long)tile * 2048 + lane * 16;
                if (CPW == 3) {
                    // three tiles = a whole column-tile pair + one half of the neighbouring pair (wave-uniform: which half, and on
                    // which side, follows from the parity of the wave's first tile): 16-byte loads for the pair, 8-byte loads for
                    // the half slot (the other 8 bytes of those slots go to the neighbouring wave of this workgroup)
                    typedef const M2M_GLOBAL_AS u32x4_t* g4_t;
                    typedef const M2M_GLOBAL_AS u32x2_t* g2_t;
                    const bool odd = (__builtin_amdgcn_readfirstlane(ct0) & 1) != 0;
                    const int plast = (nct >> 1) - 1;
                    const int pfull = min(odd ? (ct0 + 1) >> 1 : ct0 >> 1, plast), phalf = min(odd ? ct0 >> 1 : (ct0 + 2) >> 1, plast);
                    const long strd = m2m_hchn_stride(ntiles);
                    const long bf = (long)pfull * strd + (long)tile * 2048 + lane * 16;
                    const long bh = (long)phalf * strd + (long)tile * 2048 + lane * 16 + (odd ? 8 : 0);
                    const u32x4_t h0 = __builtin_nontemporal_load((g4_t)(src_h + bf)), h1 = __builtin_nontemporal_load((g4_t)(src_h + bf + 1024));
                    const u32x4_t d0 = __builtin_nontemporal_load((g4_t)(src_dh + bf)), d1 = __builtin_nontemporal_load((g4_t)(src_dh + bf + 1024));
                    const u32x2_t hh0 = __builtin_nontemporal_load((g2_t)(src_h + bh)), hh1 = __builtin_nontemporal_load((g2_t)(src_h + bh + 1024));
                    const u32x2_t dd0 = __builtin_nontemporal_load((g2_t)(src_dh + bh)), dd1 = __builtin_nontemporal_load((g2_t)(src_dh + bh + 1024));
                    const u32x4_t hf0 = u32x4_t{h0[0], h0[1], h1[0], h1[1]}, hf1 = u32x4_t{h0[2], h0[3], h1[2], h1[3]};
                    const u32x4_t df0 = u32x4_t{d0[0], d0[1], d1[0], d1[1]}, df1 = u32x4_t{d0[2], d0[3], d1[2], d1[3]};
                    const u32x4_t hhf = u32x4_t{hh0[0], hh0[1], hh1[0], hh1[1]}, dhf = u32x4_t{dd0[0], dd0[1], dd1[0], dd1[1]};
                    p.h[u][0][0].u = odd ? hhf : hf0;  p.d[u][0][0].u = odd ? dhf : df0;
                    p.h[u][1 % CPW][0].u = odd ? hf0 : hf1;  p.d[u][1 % CPW][0].u = odd ? df0 : df1;
                    p.h[u][2 % CPW][0].u = odd ? hf1 : hhf;  p.d[u][2 % CPW][0].u = odd ? df1 : dhf;
                } else if (CPW == 2) {                       // the wave owns both tiles of the pair: one 16-byte load per half
                    typedef const M2M_GLOBAL_AS u32x4_t* g4_t;
                    const u32x4_t h0 = __builtin_nontemporal_load((g4_t)(src_h + blk)), h1 = __builtin_nontemporal_load((g4_t)(src_h + blk + 1024));
                    const u32x4_t d0 = __builtin_nontemporal_load((g4_t)(src_dh + blk)), d1 = __builtin_nontemporal_load((g4_t)(src_dh + blk + 1024));
#pragma unroll
                    for (int j = 0; j < CPW; ++j) {
                        p.h[u][j][0].u = u32x4_t{h0[2 * j], h0[2 * j + 1], h1[2 * j], h1[2 * j + 1]};
                        p.d[u][j][0].u = u32x4_t{d0[2 * j], d0[2 * j + 1], d1[2 * j], d1[2 * j + 1]};
                    }
                } else {                                     // one tile of the pair: its 8-byte half of every lane slot
                    typedef const M2M_GLOBAL_AS u32x2_t* g2_t;
                    const long o = blk + (ctl[0] & 1) * 8;
                    const u32x2_t h0 = __builtin_nontemporal_load((g2_t)(src_h + o)), h1 = __builtin_nontemporal_load((g2_t)(src_h + o + 1024));
                    const u32x2_t d0 = __builtin_nontemporal_load((g2_t)(src_dh + o)), d1 = __builtin_nontemporal_load((g2_t)(src_dh + o + 1024));
                    p.h[u][0][0].u = u32x4_t{h0[0], h0[1], h1[0], h1[1]};
                    p.d[u][0][0].u = u32x4_t{d0[0], d0[1], d1[0], d1[1]};
                }
            } else {
                typedef const M2M_GLOBAL_AS u32x4_t* g4_t;
#pragma unroll
                for (int j = 0; j < CPW; ++j) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) {           // fp32: [column tile][32-row pair][half = k-block f][lane][16 B]
                        const long blk = (long)ctl[j] * m2m_hchn_stride(ntiles) + ((long)tile * NF + f) * 1024 + lane * 16;
                        p.h[u][j][f].u = __builtin_nontemporal_load((g4_t)(src_h + blk));
                        p.d[u][j][f].u = __builtin_nontemporal_load((g4_t)(src_dh + blk));
                    }
                }
            }
        }
    };
    // Consume the step (TPS tiles from `tile0`) held in p, then refill p with the step DEPTH ahead.  One barrier per step:
    // the stage written in step i (buffer i & 1) was last read in step i - 2, and every wave has passed the barrier of
    // step i - 1 since.
    auto step = [&](Pre& p, int tile0, int t_end, int i) {
        char* buf = smem + (i & 1) * (TPS * STAGE_B);
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            char* cur = buf + u * STAGE_B;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int o = (k * G::THREADS + tid) * 16;
                if (o < STAGE_B) *reinterpret_cast<u32x4_t*>(cur + o) = p.st[u][k];
            }
        }
        Frag hf[TPS][CPW][NF], df[TPS][CPW][NF];
#pragma unroll
        for (int u = 0; u < TPS; ++u)
#pragma unroll
            for (int j = 0; j < CPW; ++j)
#pragma unroll
                for (int f = 0; f < NF; ++f) { hf[u][j][f] = p.h[u][j][f]; df[u][j][f] = p.d[u][j][f]; }
        TIMER_MARK(g_tm_wg, 0);    // wait for this step's loads + stage write
        tile_load(p, min(tile0 + DEPTH * TPS, t_end - 1), t_end);
        TIMER_MARK(g_tm_wg, 1);    // issue of the refill loads
        __syncthreads();
        TIMER_MARK(g_tm_wg, 2);    // barrier
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            if (TPS > 1 && tile0 + u >= t_end) break;
            const char* cur = buf + u * STAGE_B;
            // The shared operands come from LDS one 1-KiB fragment per (f, dt); with one wave per SIMD nothing else hides
            // the ds_read latency, so the reads run WG_LA fragments ahead of the MFMAs that consume them (measured: the
            // compiler's own distance of one made the loop LDS-latency-bound at 0.83 us per tile).
            constexpr int NQ = NF * DT;
            constexpr int LA = WG_LA < NQ ? WG_LA : NQ;
            Frag aq[LA], dq[LA];
#pragma unroll
            for (int q = 0; q < LA; ++q) {
                aq[q] = ld_frag_lds(cur, q, lane);
                dq[q] = ld_frag_lds(cur + IMG_B, q, lane);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int f = q / DT, dt = q % DT;
                const Frag at = aq[q % LA], dyt = dq[q % LA];
                if (q + LA < NQ) {
                    aq[q % LA] = ld_frag_lds(cur, q + LA, lane);
                    dq[q % LA] = ld_frag_lds(cur + IMG_B, q + LA, lane);
                }
#pragma unroll
                for (int j = 0; j < CPW; ++j) {
                    Pr::mma(dw1[j][dt], df[u][j][f], at);
                    Pr::mma(dw2[j][dt], hf[u][j][f], dyt);
                }
                if (dt == DT - 1) {
#pragma unroll
                    for (int j = 0; j < CPW; ++j) Pr::mma(db1[j], df[u][j][f], ones);   // every column = sum over the tile's rows of dHpre[.][c]
                }
            }
        }
        TIMER_MARK(g_tm_wg, 3);    // LDS reads + MFMAs
    };

    const int t_begin = group * tiles_per_group;
    const int t_end = min(ntiles, t_begin + tiles_per_group);
    if (t_begin >= t_end) return;
    if constexpr (G::HD2) {
        // ---- stage one step ahead, own fragments two steps ahead (all loads unconditional, tile clamped) ----
        static_assert(CPW == 2 && NF == 1, "the two-tile bf16 form");
        typedef const M2M_GLOBAL_AS u32x4_t* g4_t;
        struct HD { u32x4_t h0, h1, d0, d1; };
        const long hbase = (long)(ctl[0] >> 1) * m2m_hchn_stride(ntiles) + lane * 16;
        auto load_hd = [&](HD& x, int tile) {
            const long blk = hbase + (long)tile * 2048;
            x.h0 = __builtin_nontemporal_load((g4_t)(src_h + blk)); x.h1 = __builtin_nontemporal_load((g4_t)(src_h + blk + 1024));
            x.d0 = __builtin_nontemporal_load((g4_t)(src_dh + blk)); x.d1 = __builtin_nontemporal_load((g4_t)(src_dh + blk + 1024));
        };
        u32x4_t st[NLD];
        auto load_st = [&](int tile) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int o = min((i * G::THREADS + tid) * 16, STAGE_B - 16);
                const gptr_t sp = o < IMG_B ? src_at : src_dyt;
                st[i] = *(const M2M_GLOBAL_AS u32x4_t*)(sp + (long)tile * IMG_B + (o < IMG_B ? o : o - IMG_B));
            }
        };
        auto step2 = [&](HD& x, int tile, int i) {
            char* cur = smem + (i & 1) * STAGE_B;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int o = (k * G::THREADS + tid) * 16;
                if (o < STAGE_B) *reinterpret_cast<u32x4_t*>(cur + o) = st[k];
            }
            Frag hf[CPW], df[CPW];
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                hf[j].u = u32x4_t{x.h0[2 * j], x.h0[2 * j + 1], x.h1[2 * j], x.h1[2 * j + 1]};
                df[j].u = u32x4_t{x.d0[2 * j], x.d0[2 * j + 1], x.d1[2 * j], x.d1[2 * j + 1]};
            }
            // the next step's stage FIRST, then the fragments of the step after next: at the next step's top only the four
            // youngest loads (those fragments) may still be in flight
            load_st(min(tile + 1, t_end - 1));
            load_hd(x, min(tile + 2, t_end - 1));
            __syncthreads();
            constexpr int NQ = DT;
            constexpr int LA = WG_LA < NQ ? WG_LA : NQ;
            Frag aq[LA], dq[LA];
#pragma unroll
            for (int q = 0; q < LA; ++q) { aq[q] = ld_frag_lds(cur, q, lane); dq[q] = ld_frag_lds(cur + IMG_B, q, lane); }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const Frag at = aq[q % LA], dyt = dq[q % LA];
                if (q + LA < NQ) { aq[q % LA] = ld_frag_lds(cur, q + LA, lane); dq[q % LA] = ld_frag_lds(cur + IMG_B, q + LA, lane); }
#pragma unroll
                for (int j = 0; j < CPW; ++j) {
                    Pr::mma(dw1[j][q], df[j], at);
                    Pr::mma(dw2[j][q], hf[j], dyt);
                }
                if (q == NQ - 1) {
#pragma unroll
                    for (int j = 0; j < CPW; ++j) Pr::mma(db1[j], df[j], ones);
                }
            }
        };
        HD ring[2];
        load_hd(ring[0], t_begin);
        load_st(t_begin);
        load_hd(ring[1], min(t_begin + 1, t_end - 1));
        int tile = t_begin, it = 0;
        for (; tile + 2 <= t_end; tile += 2) {
            step2(ring[0], tile, it++);
            step2(ring[1], tile + 1, it++);
        }
        if (tile < t_end) step2(ring[0], tile, it++);
    } else {
    Pre p[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) tile_load(p[k], min(t_begin + k * TPS, t_end - 1), t_end);
    int tile = t_begin, it = 0;
    if (DEPTH == 1) {
        for (; tile < t_end; tile += TPS) step(p[0], tile, t_end, it++);
    } else {
        for (; tile + DEPTH <= t_end; tile += DEPTH) {       // full trips: straight-line, fixed load count
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) step(p[k], tile + k, t_end, it++);
        }
#pragma unroll
        for (int k = 0; k + 1 < DEPTH; ++k)
            if (tile + k < t_end) step(p[k], tile + k, t_end, it++);
    }
    }

    __syncthreads();                                         // every wave is done reading the stage
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
    TIMER_MARK(g_tm_wg, 4);        // result write-out
}

// Where workgroup (block b, row group `group` of `ngroups`) of tower tw puts its results.
//   one group                : the caller's gradient, "+=" (or "=" with M2M_WGRAD_OVERWRITE)
//   two groups, slot mode    : group 0 as above; group 1 stores its partial sums into the tower's partial-gradient slot
//                              (m2m_tower.wslot[b], laid out [dW1 | db1 | dW2] like the flat gradient), which the optimizer
//                              adds (m2m_adam_step_ranges) or m2m_wgrad_fold folds in
//   otherwise                : float atomics onto the (zeroed) gradient
template <class TW>
static __device__ __forceinline__ WgOut wgrad_out(const TW& tw, int b, int group, int ngroups, int slot_mode) {
    const m2m_block& bk = tw.blk[b];
    WgOut o;
    o.w1 = bk.g_ch_w1; o.w2 = bk.g_ch_w2; o.b1 = bk.g_ch_b1;
    const bool overwrite = (tw.wgrad_flags & M2M_WGRAD_OVERWRITE) != 0;
    if (ngroups == 1) o.mode = overwrite ? WG_OUT_STORE : WG_OUT_ADD;
    else if (slot_mode) {
        if (group == 0) o.mode = overwrite ? WG_OUT_STORE : WG_OUT_ADD;
        else {
            const long cd = (long)tw.C * tw.D;
            float* s = tw.wslot[b];
            o.w1 = s; o.b1 = s + cd; o.w2 = s + cd + tw.C;
            o.mode = WG_OUT_STORE;
        }
    } else o.mode = WG_OUT_ATOMIC;
    return o;
}

// the stored-operand form with every stream on LDS-DMA (tower_wgrad_rc.h: wgrad_dma_body) exists for these instantiations
// Compile-time choice (both loops in one kernel spilled 60 registers); OFF by default (-DM2M_WGRAD_DMA=1 builds it).  Measured on
// M2-Mixer-B, batch 512, A/B in one process: merged launch 108.3 / 107.2 us with the DMA loop against 107.3 / 106.3 us with the
// register-staged one; the three towers alone 90 against 82 us.  Three steps of prefetch instead of one change nothing: the loop is
// not waiting for its operand streams any more (in-kernel timers, scripts/rc_timers.py: per 32-row step 0.46 us LDS reads + MFMAs,
// 0.24 us DMA issue -- 8 pieces per wave at ~75 cycles each --, 0.18 us barrier, 0.14 us DMA wait; write-out 13 us), and with five
// waves per workgroup SIMD 0 carries two of them: 2 x 36 MFMAs x 16 cycles = 0.55 us per step is the floor of this tiling.
#ifndef M2M_WGRAD_DMA
#define M2M_WGRAD_DMA 0
#endif
template <int P, int D, int RCDM> struct WgradHasDma { static constexpr bool value = M2M_WGRAD_DMA && RCDM < 0 && P == PREC_BF16 && D == 128; };

// RCDM: -1 = stored-operand form; DM_NONE / DM_HALF = recompute form with that dropout mode (bf16, hidden_dim 128 only)
template <int P, int D, int RCDM> struct WgradKernelGeom {
    static constexpr bool RC = RCDM >= 0;
    static constexpr int LDS_PLAIN = RC ? RcGeom<D>::LDS_B : WgradGeom<P, D>::LDS_B;
    static constexpr int LDS_DMA = DrGeom<D, WgradGeom<P, D>::WAVES>::LDS_B;
    static constexpr int LDS_B = WgradHasDma<P, D, RCDM>::value ? LDS_DMA : LDS_PLAIN;
    static constexpr int COLS = RC ? RcGeom<D>::COLS : WgradGeom<P, D>::COLS;
    static constexpr int MINWAVES = RC ? 2 : WgradGeom<P, D>::MINWAVES;
    static constexpr int THREADS = RC ? RC_THREADS : WgradGeom<P, D>::THREADS;
};

template <int P, int D, int RCDM, class TW>
static __device__ __forceinline__ void wgrad_dispatch(const TW& tw, int b, int slice, int group, int ngroups, int slot_mode, int ntiles,
                                                      int tpg, int rows_per_t16, unsigned int seed, unsigned int step, int dma, char* smem) {
    const WgOut out = wgrad_out(tw, b, group, ngroups, slot_mode);
    if constexpr (RCDM >= 0) {
        const Drop dr = make_drop(true, tw.p_drop, seed, step, tw.site_base + 4u * (unsigned int)b + 2u);
        wgrad_rc_body<D, RCDM>(tw.blk[b], out, tw.Cp, tw.C, slice, group, ntiles, tpg, rows_per_t16, dr.key, dr.scale, smem);
    } else {
        if constexpr (WgradHasDma<P, D, RCDM>::value)
            wgrad_dma_body<D, WgradGeom<P, D>::WAVES>(tw.blk[b], out, tw.Cp, tw.C, slice, group, ntiles, tpg, smem);
        else
            wgrad_body<P, D>(tw.blk[b], out, tw.Cp, tw.C, slice, group, ntiles, tpg, smem);
    }
}

template <int P, int D, int RCDM>
__global__ __launch_bounds__((WgradKernelGeom<P, D, RCDM>::THREADS), (WgradKernelGeom<P, D, RCDM>::MINWAVES)) void tower_wgrad_kernel(
    const m2m_tower tw, int ntiles, int tiles_per_group, int slot_mode, int rows_per_t16, unsigned int seed, unsigned int step_host,
    const unsigned int* __restrict__ step_dev, int dma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    wgrad_dispatch<P, D, RCDM>(tw, (int)blockIdx.y, (int)blockIdx.x, (int)blockIdx.z, (int)gridDim.z, slot_mode, ntiles, tiles_per_group,
                               rows_per_t16, seed, step, dma, smem);
}

// Several towers in ONE launch (job = (tower, block)): the three towers of a model finish their backward chains at about the
// same time, and one launch lets the hardware dispatcher balance their ~300 workgroups over the chip instead of three
// launches on three queues of a replayed graph racing (and sometimes serialising) each other.
// The descriptors are device-resident copies (kernel arguments are limited to 4 KiB, one m2m_tower is 2.4 KiB).
#define WG_MAX_TOWERS 4
#define WG_MAX_JOBS 32
struct WgradGroupArgs {
    const m2m_tower* tw[WG_MAX_TOWERS];
    int ntiles[WG_MAX_TOWERS], tpg[WG_MAX_TOWERS], groups[WG_MAX_TOWERS], nsl[WG_MAX_TOWERS], slot[WG_MAX_TOWERS], rpt[WG_MAX_TOWERS];
    unsigned char job_tower[WG_MAX_JOBS], job_block[WG_MAX_JOBS];
    short job_start[WG_MAX_JOBS + 1];                 // first linear workgroup index of job j (its workgroups: group-major, slice fastest)
    short xcd_start[8], xcd_len[8];                   // XCD x runs linear indices [xcd_start[x], xcd_start[x] + xcd_len[x])
    int njobs, n_tower_wgs;                           // n_tower_wgs = 8 x the longest XCD chunk (ids beyond a chunk return at once)
    int dma;                                          // stored-operand form: every stream on LDS-DMA (wgrad_dma_body)
    int n_embed_first, n_embed_pad;                   // embedding workgroups dispatched FIRST: ids [0, n_embed_first), padded to a
                                                      // multiple of 8 (n_embed_pad) so that tower ids keep their XCD (id % 8)
    int n_reduce, reduce_sets;                         // slot-reduction workgroups at the END of the grid: SPR_NBX x reduce_sets x ra.ntow
    unsigned int seed, step_host;
    const unsigned int* step_dev;
    unsigned int* bump_counter;                        // m2m_towers_wgrad_tail: *bump_counter += 1 (the step's dropout counter, behind its last reader)
};
// The grid is one-dimensional.  Tower workgroups first, XCD-aware: the hardware deals consecutive workgroup ids to the 8 XCDs
// round-robin (id % 8), each XCD has its own 4 MB L2, and what a workgroup re-reads -- the 32-row operand images of ITS
// (tower, block) job, 1.5 MB per job and row group on M2-Mixer-B, shared by the job's 24 column slices -- should come from
// that L2.  With slices dealt out in id order every XCD touched every job (18 MB of images through each 4 MB L2: the images
// came from the Infinity Cache at ~33 GB/s per CU and paced the loop at ~2 us per step).  So the jobs' workgroups are laid
// out in ONE linear list, job-major, and XCD x takes a contiguous eighth of it: an L2 serves two or three jobs.
// Then -- dispatched last, back-filling the CUs whose tower workgroup has finished -- the workgroups of the model's two
// patch-embedding weight gradients (embed_wgrad.h); a second launch beside this one costs a fork and a join in the replayed
// graph (~10 us each) and slows this kernel by contending for the same CUs.
#ifndef M2M_WG_KATTR
#define M2M_WG_KATTR
#endif
template <int P, int D, int RCDM>
__global__ __launch_bounds__((WgradKernelGeom<P, D, RCDM>::THREADS), (WgradKernelGeom<P, D, RCDM>::MINWAVES)) M2M_WG_KATTR void tower_wgrad_group_kernel(const WgradGroupArgs a,
                                                                                                          const EmbedWgradGroupArgs ea,
                                                                                                          const SplitReduceArgs ra) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int id = blockIdx.x;
    if (a.bump_counter && blockIdx.x == 0 && threadIdx.x == 0) *a.bump_counter += 1u;     // (nothing in this launch reads it: stored-operand form)
#ifndef M2M_WG_PROBE_TOWERS_ONLY        // (ISA probe: the tower path's own register need)
    // The slot reduction of a preceding fused backward launch (M2M_WGRAD_REDUCES_SMALL): ~140 short workgroups at the end of the
    // grid -- they run on the CUs the one-per-CU tower workgroups leave free, long before those finish.
    if (a.n_reduce && id >= (int)gridDim.x - a.n_reduce) {
        const int rid = id - ((int)gridDim.x - a.n_reduce);
        const int bx = rid % SPR_NBX, L = (rid / SPR_NBX) % a.reduce_sets, z = rid / (SPR_NBX * a.reduce_sets);
        split_small_grads_body<WgradKernelGeom<P, D, RCDM>::THREADS>(ra.t[z], bx, L, reinterpret_cast<float*>(smem));
        return;
    }
    if (id < a.n_embed_pad) {
        if (id < a.n_embed_first) embed_wgrad_group_body<P, D, WgradKernelGeom<P, D, RCDM>::THREADS>(ea, id, smem);
        return;
    }
    id -= a.n_embed_pad;
    if (id >= a.n_tower_wgs) {
        embed_wgrad_group_body<P, D, WgradKernelGeom<P, D, RCDM>::THREADS>(ea, id - a.n_tower_wgs, smem);
        return;
    }
#endif
    const int xcd = id & 7, idx = id >> 3;
    if (idx >= a.xcd_len[xcd]) return;
    const int lin = a.xcd_start[xcd] + idx;
    int job = 0;
    while (job + 1 < a.njobs && lin >= a.job_start[job + 1]) ++job;
    const int t = a.job_tower[job], rel = lin - a.job_start[job];
    const int slice = rel % a.nsl[t], group = rel / a.nsl[t];
    const m2m_tower& tw = *a.tw[t];
    const unsigned int step = a.step_host + (a.step_dev ? *a.step_dev : 0u);
    wgrad_dispatch<P, D, RCDM>(tw, (int)a.job_block[job], slice, group, a.groups[t], a.slot[t], a.ntiles[t], a.tpg[t], a.rpt[t], a.seed,
                               step, a.dma, smem);
}

static_assert(sizeof(WgradGroupArgs) + sizeof(EmbedWgradGroupArgs) + sizeof(SplitReduceArgs) <= 3840, "kernel arguments are limited to 4 KiB");

#ifdef M2M_ISA_PROBE
// ISA probe (scripts/isa_probe.sh wgrad): only the benchmark's weight-gradient instantiation, no host code
template __global__ void tower_wgrad_group_kernel<PREC_BF16, 128, -1>(const WgradGroupArgs, const EmbedWgradGroupArgs, const SplitReduceArgs);
template __global__ void tower_wgrad_group_kernel<PREC_BF16, 128, DM_HALF>(const WgradGroupArgs, const EmbedWgradGroupArgs, const SplitReduceArgs);
#else
// The single-owner embedding gradients as a launch of their own (256 threads, in front of the tower launch on the same stream):
// beside five-wave tower workgroups (one per CU, SIMD 0 full) an embedding workgroup finds room only on the 16 CUs the 240 tower
// workgroups leave free, and ~100 latency-bound workgroups queued on 16 CUs outlast the towers by ~20 us.
template <int D>
__global__ __launch_bounds__(256, 2) void embed_wgrad_fast_group_kernel(const EmbedWgradGroupArgs ea) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    embed_wgrad_group_body<PREC_BF16, D, 256>(ea, blockIdx.x, smem);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int wgrad_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
// The recompute form: bf16, hidden_dim 128, dropout off or p == 0.5 (the one-bit keep stream), and not a tower the split path
// takes (its chain launches store both hidden operands).  OFF by default (M2M_WGRAD_RECOMP=1 enables it): measured on
// M2-Mixer-B, batch 512, the merged launch takes 134-142 us against 111-115 us for the stored-operand form -- the launch
// leaves the HBM bound (150 MB instead of 300 MB of hidden operands) but a wave's step grows from 36 MFMAs + two loads to 48
// MFMAs + ~150 VALU + 40 LDS reads + 8 LDS-DMA pieces (~75 cycles of issue EACH), and with 292 workgroups on 256 CUs most
// SIMDs hold ONE wave, so nothing overlaps that issue stream (in-kernel timers, scripts/rc_timers.py: 1.05 us of compute,
// 0.29 us of DMA issue, 0.25 us of DMA wait per 32-row step).  DESIGN.md section 4e.
bool m2m_wgrad_recompute(const m2m_tower* t, int B) {
    static const int on = wgrad_env("M2M_WGRAD_RECOMP", 0);
    if (!on || t->prec != PREC_BF16 || t->D != 128 || t->nblocks < 1 || t->Cp < 64) return false;
    if (m2m_drop_mode(1, t->p_drop) == DM_GEN) return false;
    if (m2m_split_eligible(t, B, 1)) return false;
    return true;
}
extern "C" int m2m_wgrad_form(const m2m_tower* t, int B) { return t && m2m_wgrad_recompute(t, B) ? 1 : 0; }

// The partial-gradient slot can stand in for a second row group when the three channel-mixing gradients of every block lie
// back to back ([dW1 | db1 | dW2], the flat engines' layout), because the slot is folded in as ONE range per block.
static bool wgrad_slot_usable(const m2m_tower* t) {
    static const int on = wgrad_env("M2M_WGRAD_SLOT", 1);
    if (!on || t->nblocks < 1) return false;
    for (int b = 0; b < t->nblocks; ++b) {
        const m2m_block& k = t->blk[b];
        if (!t->wslot[b]) return false;
        if (k.g_ch_b1 != k.g_ch_w1 + (long)t->C * t->D || k.g_ch_w2 != k.g_ch_b1 + t->C) return false;
    }
    return true;
}

// M2M_WGRAD_DMA=0: the register-staged stored-operand loop (A/B)
static int wgrad_use_dma() { static const int on = wgrad_env("M2M_WGRAD_DMA", 1); return on; }

struct WgradPlan { int ntiles, nsl, groups, tpg, rpt, slot; };
// honour_overwrite == false: the plan the tower would get if its gradient were zeroed and accumulated (m2m_wgrad_groups)
static WgradPlan wgrad_plan(const m2m_tower* t, int B, int cols, bool honour_overwrite = true) {
    const bool wide = m2m_is_wide(t);                           // wide path: chain tiles are any BM consecutive rows
    const int SPW = wide ? 1 : BM / t->N;                       // samples per chain tile
    const int nchain = wide ? (int)(((long)B * t->N + BM - 1) / BM) : (B + SPW - 1) / SPW;   // chain tiles (BM rows each)
    const int ntiles = (nchain * BM + WBM - 1) / WBM;           // streamed tiles (WBM rows each)
    const int nsl = (t->Cp + cols - 1) / cols;                  // column slices (128 or 64 columns)
    // A workgroup's time is linear in its number of 32-row steps, so the launch time is set by its longest workgroup: rows
    // are split into groups until a workgroup has at most 64 steps (2048 rows) or the launch reaches ~128 workgroups.
    // Groups beyond the first add their partial results with float atomics onto the zeroed gradient (two groups: a + b ==
    // b + a, bit-deterministic; more -- only tiny launches -- are not), which an overwriting caller (M2M_WGRAD_OVERWRITE:
    // the gradient is NOT zero on entry) cannot have: one group then, or two through the slot (wgrad_group_plans).
    int groups = 1;
    const int wgs = nsl * t->nblocks;
    while (wgs * groups < 128 && (ntiles + groups - 1) / groups > 64) ++groups;
    if (groups < (32 + wgs - 1) / wgs) groups = (32 + wgs - 1) / wgs;
    if (const char* e = getenv("M2M_WGRAD_GROUPS")) groups = atoi(e);   // diagnostic override (scripts/wgrad_probe.py sweeps it); unset in production
    if (const char* e = getenv("M2M_WGRAD_LONG_GROUPS")) { if (ntiles > 64) groups = atoi(e); }   // diagnostic: towers with more than 64 steps only
    if (honour_overwrite && (t->wgrad_flags & M2M_WGRAD_OVERWRITE)) groups = 1;
    if (groups < 1) groups = 1;
    int tpg = (ntiles + groups - 1) / groups;
    if (tpg < 4) tpg = 4;
    if (tpg > ntiles) tpg = ntiles;
    groups = (ntiles + tpg - 1) / tpg;
    WgradPlan pl;
    pl.ntiles = ntiles; pl.nsl = nsl; pl.groups = groups; pl.tpg = tpg; pl.slot = 0;
    pl.rpt = wide ? BM : SPW * t->N;                            // token rows per chain tile (the dropout row index of the recompute form)
    return pl;
}
static int wgrad_cols(const m2m_tower* t, int B) {
    if (m2m_wgrad_recompute(t, B)) return RcGeom<128>::COLS;
    if (t->prec == PREC_BF16 && t->D == 128) return WgradGeom<PREC_BF16, 128>::COLS;
    return t->prec == PREC_BF16 && t->D < 128 ? WG_WAVES * 32 : WG_WAVES * 16;
}
extern "C" int m2m_wgrad_groups(const m2m_tower* t, int B) {
    if (!t || t->nblocks < 1) return 1;
    return wgrad_plan(t, B, wgrad_cols(t, B), false).groups;
}

template <int P, int D, int RCDM>
static int launch_wgrad(const m2m_tower* t, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    typedef WgradKernelGeom<P, D, RCDM> KG;
    const WgradPlan pl = wgrad_plan(t, B, KG::COLS);
    const size_t lds = (size_t)KG::LDS_B;
    auto kern = tower_wgrad_kernel<P, D, RCDM>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.nsl, t->nblocks, pl.groups), dim3(KG::THREADS), lds, st, *t, pl.ntiles, pl.tpg, 0, pl.rpt, seed, step,
                       step_dev, wgrad_use_dma());
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

// Plans of a multi-tower launch: per tower as above; then, for a tower with a usable partial-gradient slot (m2m_tower.wslot),
//   * two row groups that would add with atomics become two groups through the slot (plain stores; the fusion tower of
//     M2-Mixer-B: 12.6 MB of float atomics at ~1 TB/s sat at the end of the launch);
//   * one group whose workgroups would run at least twice as many steps as the shortest tower's is split in two the same way
//     (its workgroups alone on their CUs would set the launch time).
static void wgrad_group_plans(const m2m_tower* const* host, int n, int B, int cols, WgradPlan* pl) {
    int min_tpg = 1 << 30;
    WgradPlan nat[WG_MAX_TOWERS];
    for (int i = 0; i < n; ++i) {
        pl[i] = wgrad_plan(host[i], B, cols);
        nat[i] = wgrad_plan(host[i], B, cols, false);
        if (nat[i].tpg < min_tpg) min_tpg = nat[i].tpg;
    }
    for (int i = 0; i < n; ++i) {
        if (n < 2 || !wgrad_slot_usable(host[i])) continue;
        if (nat[i].groups == 2) { pl[i] = nat[i]; pl[i].slot = 1; }
        else if (nat[i].groups == 1 && nat[i].tpg >= 2 * min_tpg && nat[i].tpg >= 32) { pl[i] = nat[i]; pl[i].groups = 2; pl[i].tpg = (pl[i].ntiles + 1) / 2; pl[i].slot = 1; }
    }
}

template <int P, int D, int RCDM>
static int launch_wgrad_group(const m2m_tower* const* host, const m2m_tower* const* dev, int n, const m2m_embed* const* embeds,
                              const float* const* inputs, const float* const* d_x0s, const m2m_tower* const* embed_towers,
                              int nembeds, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st,
                              const SplitReduceTower* heads_reduce = nullptr, unsigned int* bump_counter = nullptr) {
    typedef WgradKernelGeom<P, D, RCDM> KG;
    WgradGroupArgs a;
    memset(&a, 0, sizeof(a));
    int njobs = 0, total = 0;
    // longest workgroups first (most token tiles per workgroup): they are dispatched first and the short ones back-fill
    int order[WG_MAX_TOWERS];
    WgradPlan pl[WG_MAX_TOWERS];
    wgrad_group_plans(host, n, B, KG::COLS, pl);
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (pl[order[j]].tpg > pl[order[i]].tpg) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int k = 0; k < n; ++k) {
        const int i = order[k];
        a.tw[i] = dev[i];
        a.ntiles[i] = pl[i].ntiles; a.tpg[i] = pl[i].tpg; a.groups[i] = pl[i].groups; a.nsl[i] = pl[i].nsl; a.slot[i] = pl[i].slot;
        a.rpt[i] = pl[i].rpt;
        for (int b = 0; b < host[i]->nblocks; ++b) {
            if (njobs >= WG_MAX_JOBS) { m2m_set_error("towers_wgrad: more than 32 (tower, block) jobs", __FILE__, __LINE__); return -1; }
            a.job_tower[njobs] = (unsigned char)i;
            a.job_block[njobs] = (unsigned char)b;
            a.job_start[njobs] = (short)total;
            total += pl[i].nsl * pl[i].groups;
            ++njobs;
        }
    }
    if (njobs == 0 && nembeds == 0) return 0;
    if (total > 32000) { m2m_set_error("towers_wgrad: too many workgroups for one launch", __FILE__, __LINE__); return -1; }
    a.job_start[njobs] = (short)total;
    int max_len = 0;
    for (int x = 0, at = 0; x < 8; ++x) {
        const int len = total / 8 + (x < total % 8 ? 1 : 0);
        a.xcd_start[x] = (short)at; a.xcd_len[x] = (short)len;
        at += len;
        if (len > max_len) max_len = len;
    }
    a.njobs = njobs; a.n_tower_wgs = 8 * max_len;
    a.seed = seed; a.step_host = step; a.step_dev = step_dev;
    a.bump_counter = bump_counter;
    a.dma = wgrad_use_dma();
    EmbedWgradGroupArgs ea;
    memset(&ea, 0, sizeof(ea));
    // The patch-embedding gradients ride in the same launch, dispatched last.  Fast form (single owner, bf16, needs the d_x0^T
    // images of the towers the embeddings feed): ~200 short workgroups.  Otherwise the row-group form (~2 workgroups per CU,
    // float atomics) -- which needs 256-thread workgroups: an instantiation with wider tower workgroups launches it separately.
    int n_embed_wgs = 0;
    size_t lds_e = 0;
    bool embeds_separately = false;
    if (nembeds) {
        n_embed_wgs = (P == PREC_BF16 && nembeds == EMB_GROUP) ? embed_wgrad_group_args_fast(ea, embeds, inputs, embed_towers, B) : 0;
        if (n_embed_wgs) lds_e = embed_wgrad_fast_lds<D, KG::THREADS>();
        else if (KG::THREADS == 256) { static const int ewgs = wgrad_env("M2M_EMBED_WGS", D >= 256 ? 128 : 512);   /* every row group adds 64 x D floats with atomics: at D = 256 (MM-IMDb, batch 32) 512 workgroups were 34 MB of them */ n_embed_wgs = embed_wgrad_group_args(ea, embeds, inputs, d_x0s, B, ewgs, nembeds); lds_e = embed_wgrad_lds<D, P>(); }
        else embeds_separately = true;
    }
    if (embeds_separately) {
        if (nembeds == 1) { if (int rc = m2m_embed_wgrad(embeds[0], inputs[0], d_x0s[0], B, (void*)st)) return rc; }
        else if (int rc = m2m_embeds_wgrad(embeds, inputs, d_x0s, nembeds, B, (void*)st)) return rc;
    }
    if constexpr (P == PREC_BF16) {
        static const int merged = wgrad_env("M2M_EMBED_MERGED", 1);
        if (n_embed_wgs && ea.fast && !merged && KG::THREADS != 256) {
            const size_t le = embed_wgrad_fast_lds<D, 256>();
            auto ek = embed_wgrad_fast_group_kernel<D>;
            static bool eattr = false;
            if (!eattr) { M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ek), hipFuncAttributeMaxDynamicSharedMemorySize, (int)le)); eattr = true; }
            hipLaunchKernelGGL(ek, dim3((unsigned)n_embed_wgs), dim3(256), le, st, ea);
            M2M_CHECK_HIP(hipGetLastError());
            n_embed_wgs = 0; lds_e = 0;
        }
    }
    const size_t lds_t = (size_t)KG::LDS_B;
    const size_t lds = lds_t > lds_e ? lds_t : lds_e;
    auto kern = tower_wgrad_group_kernel<P, D, RCDM>;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    // Fast-form embedding workgroups go FIRST (every CU is free then: ~100 of them run beside the first tower workgroups and
    // are gone after ~15 us; dispatched last they queue on the 16 CUs that 240 one-per-CU tower workgroups leave free and the
    // launch's end depends on where those land: 101-121 us measured from box to box, against a steady ~95 us).
    static const int embed_first = wgrad_env("M2M_EMBED_FIRST", 1);
    if (ea.fast && n_embed_wgs && embed_first) { a.n_embed_first = n_embed_wgs; a.n_embed_pad = (n_embed_wgs + 7) & ~7; n_embed_wgs = 0; }
    // slot reductions deferred to this launch (towers flagged M2M_WGRAD_REDUCES_SMALL whose backward used slots): up to three
    // ride here, more get the reduction launch of their own
    SplitReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    for (int i = 0; i < n; ++i) {
        if (!(host[i]->wgrad_flags & M2M_WGRAD_REDUCES_SMALL)) continue;
        SplitReduceTower x;
        if (!m2m_small_part_deferred(x, host[i], B)) continue;
        if (ra.ntow < SPR_MAX_TOWERS && KG::THREADS % SPR_COLS == 0) { ra.t[ra.ntow++] = x; continue; }
        SplitReduceArgs one;
        memset(&one, 0, sizeof(one));
        one.t[0] = x; one.ntow = 1;
        if (int rc = m2m_split_small_grads(one, st)) return rc;
    }
    if (heads_reduce) {                                            // the classification heads' weight-gradient slots (m2m_head.g_part)
        if (ra.ntow < SPR_MAX_TOWERS && KG::THREADS % SPR_COLS == 0) ra.t[ra.ntow++] = *heads_reduce;
        else {
            SplitReduceArgs one;
            memset(&one, 0, sizeof(one));
            one.t[0] = *heads_reduce; one.ntow = 1;
            if (int rc = m2m_split_small_grads(one, st)) return rc;
        }
    }
    for (int i = 0; i < ra.ntow; ++i) a.reduce_sets = ra.t[i].nlaunch > a.reduce_sets ? ra.t[i].nlaunch : a.reduce_sets;
    a.n_reduce = SPR_NBX * a.reduce_sets * ra.ntow;
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.n_embed_pad + a.n_tower_wgs + n_embed_wgs + a.n_reduce)), dim3(KG::THREADS), lds, st, a, ea, ra);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

int m2m_check_tower(const m2m_tower* t, int B);

// 1 if m2m_towers_wgrad on these towers at batch B leaves part of tower i's channel-mixing gradients in its slot (bit i of the
// result): the caller must then add the slot (m2m_adam_step_ranges / m2m_wgrad_fold) before using the gradient.
extern "C" int m2m_wgrad_slot_groups(const m2m_tower* const* towers, int ntowers, int B) {
    if (!towers || ntowers < 1 || ntowers > WG_MAX_TOWERS) return 0;
    WgradPlan pl[WG_MAX_TOWERS];
    wgrad_group_plans(towers, ntowers, B, wgrad_cols(towers[0], B), pl);
    int bits = 0;
    for (int i = 0; i < ntowers; ++i) bits |= pl[i].slot << i;
    return bits;
}

extern "C" int m2m_tower_wgrad(const m2m_tower* t, int B, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (int rc = m2m_check_tower(t, B)) return rc;
    if (t->nblocks == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (t->wgrad_flags & M2M_WGRAD_REDUCES_SMALL) {              // the deferred slot reduction of this tower's backward: its own launch here
        SplitReduceArgs one;
        memset(&one, 0, sizeof(one));
        if (m2m_small_part_deferred(one.t[0], t, B)) {
            one.ntow = 1;
            if (int rc = m2m_split_small_grads(one, st)) return rc;
        }
    }
    if (m2m_wgrad_recompute(t, B)) {
        if (m2m_drop_mode(1, t->p_drop) == DM_HALF) return launch_wgrad<PREC_BF16, 128, DM_HALF>(t, B, seed, step, step_dev, st);
        return launch_wgrad<PREC_BF16, 128, DM_NONE>(t, B, seed, step, step_dev, st);
    }
#define M2M_WG_CASE(PP, DD) if (t->prec == PP && t->D == DD) return launch_wgrad<PP, DD, -1>(t, B, seed, step, step_dev, st);
    M2M_WG_CASE(PREC_BF16, 32) M2M_WG_CASE(PREC_BF16, 64) M2M_WG_CASE(PREC_BF16, 128) M2M_WG_CASE(PREC_BF16, 256)
    M2M_WG_CASE(PREC_F32, 32) M2M_WG_CASE(PREC_F32, 64) M2M_WG_CASE(PREC_F32, 128) M2M_WG_CASE(PREC_F32, 256)
#undef M2M_WG_CASE
    m2m_set_error("tower_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_heads_part_tiles(int B);
static int towers_wgrad_impl(const m2m_tower* const* towers, const m2m_tower* const* dev_towers, int ntowers,
                             const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s,
                             const m2m_tower* const* embed_towers, int nembeds,
                             int B, uint32_t seed, uint32_t step, const uint32_t* step_dev,
                             const m2m_head* heads, int nheads, int K, void* stream, uint32_t* bump_counter = nullptr) {
    SplitReduceTower hr;
    const SplitReduceTower* heads_reduce = nullptr;
    if (heads && nheads > 0) {
        if (nheads > 3 || K < 2 || !towers || ntowers < 1 || !towers[0]) { m2m_set_error("towers_wgrad_heads: 1..3 heads, K >= 2", __FILE__, __LINE__); return -1; }
        const int D = towers[0]->D, tiles = m2m_heads_part_tiles(B);
        if ((long)K * D + K + 2 > SPP_STRIDE) { m2m_set_error("towers_wgrad_heads: K*D + K + 2 exceeds the slot", __FILE__, __LINE__); return -1; }
        memset(&hr, 0, sizeof(hr));
        for (int h = 0; h < nheads; ++h) {
            if (!heads[h].g_part || !heads[h].g_w || !heads[h].g_b || heads[h].g_part != heads[0].g_part + (long)h * tiles * SPP_STRIDE) {
                m2m_set_error("towers_wgrad_heads: every head needs g_w, g_b and consecutive g_part buffers", __FILE__, __LINE__);
                return -1;
            }
            hr.g_hw[h] = heads[h].g_w; hr.g_hb[h] = heads[h].g_b;
        }
        hr.part = heads[0].g_part; hr.ntiles = tiles; hr.nlaunch = nheads; hr.D = D; hr.head_set0 = 0; hr.nheads = nheads; hr.K = K;
        hr.losses = nullptr;                                      // (the slots' loss entries are zero: see m2m_head.g_part)
        heads_reduce = &hr;
    }
    if (!towers || !dev_towers || ntowers < 1 || ntowers > WG_MAX_TOWERS) { m2m_set_error("towers_wgrad: 1..4 towers", __FILE__, __LINE__); return -1; }
    if (nembeds != 0 && (nembeds < 1 || nembeds > EMB_GROUP || !embeds || !inputs || !d_x0s)) {
        m2m_set_error("towers_wgrad: at most two patch embeddings", __FILE__, __LINE__);
        return -1;
    }
    for (int i = 0; i < ntowers; ++i) {
        if (int rc = m2m_check_tower(towers[i], B)) return rc;
        if (!dev_towers[i]) { m2m_set_error("towers_wgrad: missing device-resident descriptor", __FILE__, __LINE__); return -1; }
        if (towers[i]->prec != towers[0]->prec || towers[i]->D != towers[0]->D) {
            m2m_set_error("towers_wgrad: the towers of one launch must share precision and hidden_dim", __FILE__, __LINE__);
            return -1;
        }
        if (m2m_wgrad_recompute(towers[i], B) != m2m_wgrad_recompute(towers[0], B) ||
            (m2m_wgrad_recompute(towers[0], B) && m2m_drop_mode(1, towers[i]->p_drop) != m2m_drop_mode(1, towers[0]->p_drop))) {
            m2m_set_error("towers_wgrad: the towers of one launch must share the weight-gradient form and the dropout mode", __FILE__, __LINE__);
            return -1;
        }
    }
    for (int i = 0; i < nembeds; ++i) {
        if (int rc = m2m_check_embed(embeds[i], B)) return rc;
        if (!inputs[i] || !d_x0s[i]) { m2m_set_error("towers_wgrad: null embedding input", __FILE__, __LINE__); return -1; }
        if (embeds[i]->prec != towers[0]->prec || embeds[i]->D != towers[0]->D) {
            m2m_set_error("towers_wgrad: embeddings must share the towers' precision and hidden_dim", __FILE__, __LINE__);
            return -1;
        }
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const m2m_tower* t = towers[0];
    if (bump_counter && m2m_wgrad_recompute(t, B)) {
        m2m_set_error("towers_wgrad_tail: the recompute form READS the dropout counter in this launch; advance it with m2m_counter_add", __FILE__, __LINE__);
        return -1;
    }
    if (m2m_wgrad_recompute(t, B)) {
        if (m2m_drop_mode(1, t->p_drop) == DM_HALF)
            return launch_wgrad_group<PREC_BF16, 128, DM_HALF>(towers, dev_towers, ntowers, embeds, inputs, d_x0s, embed_towers, nembeds, B, seed, step, step_dev, st, heads_reduce);
        return launch_wgrad_group<PREC_BF16, 128, DM_NONE>(towers, dev_towers, ntowers, embeds, inputs, d_x0s, embed_towers, nembeds, B, seed, step, step_dev, st, heads_reduce);
    }
#define M2M_WGG_CASE(PP, DD) if (t->prec == PP && t->D == DD) return launch_wgrad_group<PP, DD, -1>(towers, dev_towers, ntowers, embeds, inputs, d_x0s, embed_towers, nembeds, B, seed, step, step_dev, st, heads_reduce, bump_counter);
    M2M_WGG_CASE(PREC_BF16, 32) M2M_WGG_CASE(PREC_BF16, 64) M2M_WGG_CASE(PREC_BF16, 128) M2M_WGG_CASE(PREC_BF16, 256)
    M2M_WGG_CASE(PREC_F32, 32) M2M_WGG_CASE(PREC_F32, 64) M2M_WGG_CASE(PREC_F32, 128) M2M_WGG_CASE(PREC_F32, 256)
#undef M2M_WGG_CASE
    m2m_set_error("towers_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}

extern "C" int m2m_towers_wgrad(const m2m_tower* const* towers, const m2m_tower* const* dev_towers, int ntowers,
                                const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s,
                                const m2m_tower* const* embed_towers, int nembeds,
                                int B, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    return towers_wgrad_impl(towers, dev_towers, ntowers, embeds, inputs, d_x0s, embed_towers, nembeds, B, seed, step, step_dev, nullptr, 0, 0, stream);
}
extern "C" int m2m_towers_wgrad_heads(const m2m_tower* const* towers, const m2m_tower* const* dev_towers, int ntowers,
                                      const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s,
                                      const m2m_tower* const* embed_towers, int nembeds,
                                      int B, uint32_t seed, uint32_t step, const uint32_t* step_dev,
                                      const m2m_head* heads, int nheads, int K, void* stream) {
    return towers_wgrad_impl(towers, dev_towers, ntowers, embeds, inputs, d_x0s, embed_towers, nembeds, B, seed, step, step_dev, heads, nheads, K, stream);
}

extern "C" int m2m_towers_wgrad_tail(const m2m_tower* const* towers, const m2m_tower* const* dev_towers, int ntowers,
                                     const m2m_embed* const* embeds, const float* const* inputs, const float* const* d_x0s,
                                     const m2m_tower* const* embed_towers, int nembeds,
                                     int B, uint32_t seed, uint32_t step, const uint32_t* step_dev,
                                     const m2m_head* heads, int nheads, int K, uint32_t* bump_counter, void* stream) {
    return towers_wgrad_impl(towers, dev_towers, ntowers, embeds, inputs, d_x0s, embed_towers, nembeds, B, seed, step, step_dev, heads, nheads, K, stream, bump_counter);
}

// 1: m2m_towers_wgrad(..., embeds, ..., embed_towers, ...) at batch B computes the embedding gradients in the single-owner form
// (and honours m2m_embed.wgrad_flags); 0: the row-group form ("+=" with atomics onto a zeroed gradient).
extern "C" int m2m_embeds_wgrad_form(const m2m_embed* const* embeds, const m2m_tower* const* embed_towers, int nembeds, int B) {
    if (!embeds || !embed_towers || nembeds != EMB_GROUP) return 0;
    EmbedWgradGroupArgs ea;
    const float* ins[EMB_GROUP] = {nullptr, nullptr};
    return embed_wgrad_group_args_fast(ea, embeds, ins, embed_towers, B) > 0 ? 1 : 0;
}

// g_ch_w1 / g_ch_b1 / g_ch_w2 += the tower's partial-gradient slot (blocks with contiguous channel gradients; see
// m2m_tower.wslot).  For callers that need the complete gradient before the optimizer (the data-parallel exchange).
__global__ void wgrad_fold_kernel(float* __restrict__ g, const float* __restrict__ s, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] += s[i];
}
extern "C" int m2m_wgrad_fold(const m2m_tower* t, void* stream) {
    if (!t) { m2m_set_error("wgrad_fold: null tower", __FILE__, __LINE__); return -1; }
    if (!wgrad_slot_usable(t)) { m2m_set_error("wgrad_fold: channel gradients are not contiguous", __FILE__, __LINE__); return -1; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long n = 2L * t->C * t->D + t->C;
    for (int b = 0; b < t->nblocks; ++b) {
        hipLaunchKernelGGL(wgrad_fold_kernel, dim3(512), dim3(256), 0, st, t->blk[b].g_ch_w1, t->wslot[b], n);
        M2M_CHECK_HIP(hipGetLastError());
    }
    return 0;
}
#endif   // M2M_ISA_PROBE
