// Wide path of a tower (N > 8 tokens per sample or D > 128: MIMIC N = 24 / 25, MM-IMDb N = 40 / 80, D = 256).
//
// A sample no longer fits one 16-row workgroup tile, so each MixerBlock (modules/mixer.py:42-47) runs as
//   forward : token_fwd (x -> x_mid = x + token_mix(x))         then the channel-mixing launch of tower_fwd.hip
//   backward: the channel-mixing launch of tower_bwd.hip         then token_bwd_cols + ln1_bwd_rows
// with the fp32 residual / gradient stream handed over through x_in / x_mid / ws_a / ws_b.
//
// Token mixing couples the N tokens of one (sample, channel) COLUMN and nothing else, so the token kernels give
// every lane one column: a workgroup is ONE wave = 64 columns (64 channels of a sample; two samples when D = 32).
// LayerNorm-1 needs row statistics over all D channels: the wave first computes mean / rstd of its samples' N rows
// (workgroups sharing a sample repeat that; the rows come from L2), then each lane walks its column.
// The token MLP (N x T, T <= 32) runs on the VALU with the weights broadcast from LDS; h[t] lives in registers.
#include "tile.h"
#include "mlp_body.h"

#define TW_COLS 64                 // columns (lanes) per workgroup
#define TW_TMAX 32                 // token_dim upper bound; the kernels are instantiated for TM = 16 and 32 hidden units (h[] registers,
                                   // LDS weight rows): token_dim <= 16 (MIMIC, MM-IMDb) does half the work in half the LDS
#define TW_LDW (TW_COLS + 1)       // padded row stride of the per-column LDS tiles
#define TW_UB 8                    // tokens whose global loads are issued together
// Waves per workgroup (template parameter NW: 4, 8 or 16): all own the same 64 columns, each a share of the tokens.  A launch is
// (samples x D / 64) workgroups of dependent latency chains (row statistics, strided column walks, LDS sums): at the cfg batches
// (MM-IMDb 32, MIMIC-H 128: 128 workgroups) four waves leave half of the chip's SIMDs empty and every chain four times as long
// as it need be -- small launches take 16 waves (8 when token_dim > 16), large ones 4 (more workgroups per CU instead).
// NC: tokens per chunk of the parameter-gradient reduction (backward): 16 at 4 waves, 8 above (more chunks than waves otherwise)
template <int NW> struct TokNC { static constexpr int value = NW > 4 ? 8 : 16; };

int m2m_chain_forward_rows(const m2m_tower* t, const float* x0, long x0_ss, int B, float* out, long out_ss, int training,
                           unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st);
int m2m_chain_backward_rows(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                            long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st);

// ---- shared prologue: token weights -> LDS, LayerNorm-1 row statistics of this workgroup's samples -------------
//   w1s[n][t] = W1[t][n], w2s[n][t] = W2[n][t] (t padded to 32 with zeros), b1s[32], b2s[N]
//   stats[(sl * N + n) * 2 + {0, 1}] = mean, rstd of row n of local sample sl
struct TokGeom {
    int chunks;      // 64-column chunks per sample (D / 64, at least 1)
    int spw;         // samples per workgroup (2 when D == 32)
};
static __host__ __device__ __forceinline__ TokGeom tok_geom(int D) {
    TokGeom g;
    g.chunks = D >= TW_COLS ? D / TW_COLS : 1;
    g.spw = D >= TW_COLS ? 1 : TW_COLS / D;
    return g;
}
static __host__ __device__ __forceinline__ size_t tok_lds_floats(int N, int spw, int TM) { return (size_t)2 * N * TM + TM + N + 2 * spw * N; }

static __device__ __forceinline__ void tok_stage_weights(const m2m_block& bk, int N, int T, int TM, float* w1s, float* w2s, float* b1s,
                                                         float* b2s, int lane, int nthreads = TW_COLS) {
    for (int i = lane; i < N * TM; i += nthreads) {
        const int n = i / TM, t = i % TM;
        w1s[i] = t < T ? bk.tok_w1[t * N + n] : 0.f;
        w2s[i] = t < T ? bk.tok_w2[n * T + t] : 0.f;
    }
    if (lane < TM) b1s[lane] = lane < T ? bk.tok_b1[lane] : 0.f;
    for (int n = lane; n < N; n += nthreads) b2s[n] = bk.tok_b2[n];
}
// one wave: statistics of `rows` rows; row r lives at src + (r / N) * ss + (r % N) * D  (r counted from sample s_first)
// (rows r_begin, r_begin + r_step, ...: the waves of a workgroup share the rows)
static __device__ __forceinline__ void tok_row_stats(const float* __restrict__ src, long ss, int s_first, int rows, int N, int D,
                                                     float* stats, int lane, int r_begin = 0, int r_step = 1) {
    // four rows per batch: all their loads are requested before the first reduction (one memory round trip per batch, not
    // per row -- this single wave has nothing else to hide it behind)
    constexpr int RB = 4;
    for (int r0 = r_begin; r0 < rows; r0 += RB * r_step) {
        float v[RB][4];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int r = min(r0 + j * r_step, rows - 1);
            const float* row = src + (long)(s_first + r / N) * ss + (long)(r % N) * D;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + 64 * i;
                v[j][i] = c < D ? row[c] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            float s = (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
            s = wave_sum_xor(s, 64);
            const float mean = s / (float)D;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + 64 * i;
                const float dlt = c < D ? v[j][i] - mean : 0.f;
                s2 = __builtin_fmaf(dlt, dlt, s2);
            }
            s2 = wave_sum_xor(s2, 64);
            const float vv = s2 / (float)D + 1e-5f;
            float rstd = __builtin_amdgcn_rsqf(vv);
            rstd = rstd * (1.5f - 0.5f * vv * rstd * rstd);
            if (lane == 0 && r0 + j * r_step < rows) { stats[2 * (r0 + j * r_step)] = mean; stats[2 * (r0 + j * r_step) + 1] = rstd; }
        }
    }
}

// ---- forward: x_mid = x + Dropout(W2 Dropout(GELU(W1 LN1(x)^T + b1)) + b2)^T  (modules/mixer.py:30-35, :43) ------
template <int P, int DM, int TM, int NW, class TW>
static __device__ __forceinline__ void token_fwd_body(const TW& tw, int b, int bx, const float* __restrict__ src, long src_ss,
                                                      int B, float* __restrict__ x_mid, float* __restrict__ save_x_in,
                                                      int training, unsigned int seed, unsigned int step_host,
                                                      const unsigned int* __restrict__ step_dev, float* smf) {
    const int N = tw.N, T = tw.T, D = tw.D;
    const TokGeom tg = tok_geom(D);
    float* w1s = smf;
    float* w2s = w1s + N * TM;
    float* b1s = w2s + N * TM;
    float* b2s = b1s + TM;
    float* stats = b2s + N;
    float* hp = smf + ((tok_lds_floats(N, tg.spw, TM) + 3) & ~(size_t)3);   // [NW][TM][64] partial hidden pre-activations
    float* ha = hp + NW * TM * TW_COLS;                               // [TM][64]           hidden activations

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const m2m_block& bk = tw.blk[b];
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    const unsigned int site = tw.site_base + 4u * b;
    const Drop dr_th = make_drop(training, tw.p_drop, seed, step, site + 0);
    const Drop dr_to = make_drop(training, tw.p_drop, seed, step, site + 1);

    const int s_first = (bx / tg.chunks) * tg.spw;
    const int chunk = bx % tg.chunks;
    const int ns = min(tg.spw, B - s_first);
    tok_stage_weights(bk, N, T, TM, w1s, w2s, b1s, b2s, tid, (NW * 64));
    tok_row_stats(src, src_ss, s_first, ns * N, N, D, stats, lane, wave, NW);
    __syncthreads();

    const int sl = D >= TW_COLS ? 0 : lane / D;
    const int d = D >= TW_COLS ? chunk * TW_COLS + lane : lane % D;
    const bool pv = sl < ns;                                 // (no early exit: every thread reaches the barriers below)
    const int s = s_first + (pv ? sl : 0);
    const unsigned int bd = (unsigned int)s * D + d;
    const float gam = bk.ln1_w[d], bet = bk.ln1_b[d];
    const float* col = src + (long)s * src_ss + d;
    const float* st = stats + 2 * (pv ? sl : 0) * N;
    // this wave's tokens: a quarter of the N tokens of the column (the four waves work on the same 64 columns)
    const int NQ = (N + NW - 1) / NW;
    const int nb = wave * NQ, ne = min(N, nb + NQ);

    float h[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) h[t] = wave == 0 ? b1s[t] : 0.f;
    // The column's values are whole rows apart in memory: requested TW_UB at a time, so that the wave waits for one memory
    // round trip per TW_UB tokens instead of one per token.
    for (int n0 = nb; n0 < ne; n0 += TW_UB) {
        float xv[TW_UB];
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) xv[j] = pv ? col[(long)min(n0 + j, ne - 1) * D] : 0.f;
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) {
            const int n = n0 + j;
            if (n < ne) {
                const float u = (xv[j] - st[2 * n]) * st[2 * n + 1] * gam + bet;
                const float4* wr = reinterpret_cast<const float4*>(w1s + n * TM);
#pragma unroll
                for (int t4 = 0; t4 < TM / 4; ++t4) {
                    const float4 w = wr[t4];
                    h[4 * t4 + 0] = __builtin_fmaf(w.x, u, h[4 * t4 + 0]);
                    h[4 * t4 + 1] = __builtin_fmaf(w.y, u, h[4 * t4 + 1]);
                    h[4 * t4 + 2] = __builtin_fmaf(w.z, u, h[4 * t4 + 2]);
                    h[4 * t4 + 3] = __builtin_fmaf(w.w, u, h[4 * t4 + 3]);
                }
            }
        }
    }
    // the waves' partial sums meet in LDS; each wave finishes a quarter of the hidden units (GELU, dropout) for all
#pragma unroll
    for (int t = 0; t < TM; ++t) hp[(wave * TM + t) * TW_COLS + lane] = h[t];
    __syncthreads();
#pragma unroll
    for (int tq = 0; tq < TM / NW; ++tq) {
        const int t = wave * (TM / NW) + tq;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += hp[(w * TM + t) * TW_COLS + lane];
        float v = 0.f;
        if (t < T) {
            v = gelu_f(a) * dr_th.scale;
            v = drop_row_keep<DM>(dr_th, bd, T, t) ? v : 0.f;
        }
        ha[t * TW_COLS + lane] = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TM; ++t) h[t] = ha[t * TW_COLS + lane];
    for (int n0 = nb; n0 < ne; n0 += TW_UB) {
        float xv[TW_UB];
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) xv[j] = pv ? col[(long)min(n0 + j, ne - 1) * D] : 0.f;
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) {
            const int n = n0 + j;
            if (n < ne && pv) {
                const float4* wr = reinterpret_cast<const float4*>(w2s + n * TM);
                float o0 = b2s[n], o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
                for (int t4 = 0; t4 < TM / 4; ++t4) {
                    const float4 w = wr[t4];
                    o0 = __builtin_fmaf(w.x, h[4 * t4 + 0], o0);
                    o1 = __builtin_fmaf(w.y, h[4 * t4 + 1], o1);
                    o2 = __builtin_fmaf(w.z, h[4 * t4 + 2], o2);
                    o3 = __builtin_fmaf(w.w, h[4 * t4 + 3], o3);
                }
                const float o = (o0 + o1) + (o2 + o3);
                const long off = ((long)s * N + n) * D + d;
                if (save_x_in) save_x_in[off] = xv[j];
                x_mid[off] = xv[j] + (drop_row_keep<DM>(dr_to, bd, N, n) ? o * dr_to.scale : 0.f);
            }
        }
    }
}

template <int P, int DM, int TM, int NW>
__global__ __launch_bounds__(NW * 64) void token_fwd_kernel(const m2m_tower tw, int b, const float* __restrict__ src, long src_ss,
                                                               int B, float* __restrict__ x_mid, float* __restrict__ save_x_in,
                                                               int training, unsigned int seed, unsigned int step_host,
                                                               const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    token_fwd_body<P, DM, TM, NW>(tw, b, blockIdx.x, src, src_ss, B, x_mid, save_x_in, training, seed, step_host, step_dev, smf);
}
// ---- the MLP tower as extra workgroups of a token-mixing launch (MIMIC-H: static features beside the time tower) -------------------
// At the cfg batch the MLP's two launches (8 workgroups, a chain of dependent layers: 14 / 22 us) are independent of the time
// tower's launches that run before / after them on the same queue, and those launches leave most of the chip idle.  A second
// stream costs a fork and a join in the replayed graph (~5 us per edge: measured); here the MLP's
// workgroups ride in the time tower's token-mixing launch instead: m2m_mlp_forward_ride / m2m_mlp_backward_ride record the call,
// the next eligible token launch (one tower, 16 waves per workgroup) carries it as its FIRST workgroups, m2m_mlp_ride_flush
// launches whatever was not picked up.  Same bodies (mlp_body.h), same results.
struct MlpRideFwd { m2m_mlp m; const float* x; int B; float* out; long out_ss; float* out2; int training; unsigned int seed, step; const unsigned int* step_dev; };
struct MlpRideBwd { m2m_mlp m; const float* x; int B; const float* d_out; long d_out_ss; const float* d_out2; };
template <int P, int DM, int TM>
__global__ __launch_bounds__(1024) void token_fwd_ride_kernel(const m2m_tower tw, int b, const float* __restrict__ src, long src_ss,
                                                              int B, float* __restrict__ x_mid, float* __restrict__ save_x_in,
                                                              int training, unsigned int seed, unsigned int step_host,
                                                              const unsigned int* __restrict__ step_dev, int n_ride, const MlpRideFwd r) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    if ((int)blockIdx.x < n_ride) {
        mlp_fwd_mfma_body<1024>(r.m, r.x, r.B, r.out, r.out_ss, r.out2, r.training, r.seed, r.step, r.step_dev, blockIdx.x, smf);
        return;
    }
    token_fwd_body<P, DM, TM, 16>(tw, b, blockIdx.x - n_ride, src, src_ss, B, x_mid, save_x_in, training, seed, step_host, step_dev, smf);
}
// Two towers (same token_dim class, hidden_dim, dropout) in one launch: blockIdx.y = tower.  One launch on one stream instead
// of two launches on two queues: no fork / join edges in the replayed graph (~5 us each).
struct TokGroupArgs {
    m2m_tower4 tw[2];
    const float* src[2]; long src_ss[2];     // forward: block input rows
    float* dst[2];                            // forward: x_mid; backward: dU
    float* save[2];                           // forward: where to save the block input (or NULL)
    const float* g_mid[2];                    // backward: gradient wrt x_mid
    int nblk[2];                              // column blocks (= workgroups) of each tower
};
static_assert(sizeof(TokGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");
template <int P, int DM, int TM, int NW>
__global__ __launch_bounds__(NW * 64) void token_fwd_group_kernel(const TokGroupArgs a, int b, int B, int training, unsigned int seed,
                                                                     unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int t = blockIdx.y;
    if ((int)blockIdx.x >= a.nblk[t]) return;
    token_fwd_body<P, DM, TM, NW>(a.tw[t], b, blockIdx.x, a.src[t], a.src_ss[t], B, a.dst[t], a.save[t], training, seed, step_host, step_dev, smf);
}

// ---- backward, column part: dU (gradient wrt LN1 output) + token-MLP parameter gradients --------------------------
//   g_mid : gradient wrt x_mid (dense rows);  x_in : saved block input (dense rows);  du_out : receives dU (dense rows)
template <int P, int DM, int TM, int NW, class TW>
static __device__ __forceinline__ void token_bwd_cols_body(const TW& tw, int b, int bx, const float* __restrict__ g_mid, int B,
                                                           float* __restrict__ du_out, unsigned int seed,
                                                           unsigned int step_host, const unsigned int* __restrict__ step_dev, int iters,
                                                           float* smf) {
    const int N = tw.N, T = tw.T, D = tw.D;
    const TokGeom tg = tok_geom(D);
    float* w1s = smf;
    float* w2s = w1s + N * TM;
    float* b1s = w2s + N * TM;
    float* b2s = b1s + TM;
    float* stats = b2s + N;
    float* hs = smf + ((tok_lds_floats(N, tg.spw, TM) + 3) & ~(size_t)3);   // [TM][TW_LDW] hidden activation (after dropout)
    float* dhs = hs + TM * TW_LDW;                              // [TM][TW_LDW] gradient wrt the hidden pre-activation
    constexpr int NC = TokNC<NW>::value;
    float* part = dhs + TM * TW_LDW;                            // per wave: [2][TM][64] partial h | dh, later its
                                                                //           [2][NC][TW_LDW] chunk of U | dV  (whichever is larger)
    constexpr int PART_F = 2 * TM * TW_COLS > 2 * NC * TW_LDW ? 2 * TM * TW_COLS : 2 * NC * TW_LDW;
    // parameter-gradient accumulators over the `iters` column blocks this workgroup walks: every value has one owner (a lane
    // of a wave), so they are plain LDS read-modify-writes; the float atomics -- the same few hundred addresses for every
    // workgroup of the launch -- happen once per workgroup at the end
    float* gw1s = part + NW * PART_F;                      // [T][N]
    float* gw2s = gw1s + N * TM;                                // [N][T]
    float* gb1s = gw2s + N * TM;                                // [TM]
    float* gb2s = gb1s + TM;                                    // [N]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const m2m_block& bk = tw.blk[b];
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    const unsigned int site = tw.site_base + 4u * b;
    const Drop dr_th = make_drop(true, tw.p_drop, seed, step, site + 0);
    const Drop dr_to = make_drop(true, tw.p_drop, seed, step, site + 1);

    tok_stage_weights(bk, N, T, TM, w1s, w2s, b1s, b2s, tid, (NW * 64));
    for (int i = tid; i < 2 * N * TM + TM + N; i += (NW * 64)) gw1s[i] = 0.f;
    const int nblk = ((B + tg.spw - 1) / tg.spw) * tg.chunks;   // column blocks of the launch
    for (int it = 0; it < iters; ++it) {
    const int blk = bx * iters + it;
    if (blk >= nblk) break;                                      // (uniform)
    const int s_first = (blk / tg.chunks) * tg.spw;
    const int chunk = blk % tg.chunks;
    const int ns = min(tg.spw, B - s_first);
    tok_row_stats(bk.x_in, (long)N * D, s_first, ns * N, N, D, stats, lane, wave, NW);
    __syncthreads();

    const int sl = D >= TW_COLS ? 0 : lane / D;
    const int d = D >= TW_COLS ? chunk * TW_COLS + lane : lane % D;
    const bool pv = sl < ns;
    const int s = s_first + (pv ? sl : 0);
    const unsigned int bd = (unsigned int)s * D + d;
    const float gam = bk.ln1_w[d], bet = bk.ln1_b[d];
    const long col0 = (long)s * N * D + d;
    const float* st = stats + 2 * (pv ? sl : 0) * N;
    // this wave's tokens: a quarter of the N tokens of the column (the four waves work on the same 64 columns)
    const int NQ = (N + NW - 1) / NW;
    const int nb = wave * NQ, ne = min(N, nb + NQ);

    // LN1 output and masked upstream gradient of tokens n0 .. n0 + TW_UB - 1 (clamped to < nmax) of this lane's column:
    // all their loads are requested before the first is used
    auto u_dv8 = [&](int n0, int nmax, float (&u)[TW_UB], float (&dv)[TW_UB]) {
        float xv[TW_UB], gv[TW_UB];
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) {
            const long o = col0 + (long)min(n0 + j, nmax - 1) * D;
            xv[j] = pv ? bk.x_in[o] : 0.f;
            gv[j] = pv ? g_mid[o] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) {
            const int n = min(n0 + j, nmax - 1);
            u[j] = pv ? (xv[j] - st[2 * n]) * st[2 * n + 1] * gam + bet : 0.f;
            dv[j] = (pv && drop_row_keep<DM>(dr_to, bd, N, n)) ? gv[j] * dr_to.scale : 0.f;
        }
    };

    float h[TM], dh[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) { h[t] = wave == 0 ? b1s[t] : 0.f; dh[t] = 0.f; }
    for (int n0 = nb; n0 < ne; n0 += TW_UB) {
        float u8[TW_UB], dv8[TW_UB];
        u_dv8(n0, ne, u8, dv8);
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) {
            const int n = n0 + j;
            if (n >= ne) break;
            const float u = u8[j], dv = dv8[j];
            const float4* wr1 = reinterpret_cast<const float4*>(w1s + n * TM);
            const float4* wr2 = reinterpret_cast<const float4*>(w2s + n * TM);
#pragma unroll
            for (int t4 = 0; t4 < TM / 4; ++t4) {
                const float4 a = wr1[t4], c = wr2[t4];
                h[4 * t4 + 0] = __builtin_fmaf(a.x, u, h[4 * t4 + 0]);
                h[4 * t4 + 1] = __builtin_fmaf(a.y, u, h[4 * t4 + 1]);
                h[4 * t4 + 2] = __builtin_fmaf(a.z, u, h[4 * t4 + 2]);
                h[4 * t4 + 3] = __builtin_fmaf(a.w, u, h[4 * t4 + 3]);
                dh[4 * t4 + 0] = __builtin_fmaf(c.x, dv, dh[4 * t4 + 0]);
                dh[4 * t4 + 1] = __builtin_fmaf(c.y, dv, dh[4 * t4 + 1]);
                dh[4 * t4 + 2] = __builtin_fmaf(c.z, dv, dh[4 * t4 + 2]);
                dh[4 * t4 + 3] = __builtin_fmaf(c.w, dv, dh[4 * t4 + 3]);
            }
        }
    }
    // the waves' partial sums meet in LDS; each wave finishes a quarter of the hidden units: activation and gradient wrt the
    // pre-activation -> hs / dhs tiles (kept for the parameter gradients); then every wave reads all of dHpre back
    {
        float* mine = part + wave * PART_F;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            mine[t * TW_COLS + lane] = h[t];
            mine[(TM + t) * TW_COLS + lane] = dh[t];
        }
    }
    __syncthreads();
#pragma unroll
    for (int tq = 0; tq < TM / NW; ++tq) {
        const int t = wave * (TM / NW) + tq;
        float hsum = 0.f, dsum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            hsum += part[w * PART_F + t * TW_COLS + lane];
            dsum += part[w * PART_F + (TM + t) * TW_COLS + lane];
        }
        float hact = 0.f, dhp = 0.f;
        if (t < T) {
            float gl, dgl;
            gelu_grad_f(hsum, gl, dgl);
            const bool keep = pv && drop_row_keep<DM>(dr_th, bd, T, t);
            hact = keep ? gl * dr_th.scale : 0.f;
            dhp = keep ? dsum * dr_th.scale * dgl : 0.f;
        }
        hs[t * TW_LDW + lane] = hact;
        dhs[t * TW_LDW + lane] = dhp;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TM; ++t) dh[t] = dhs[t * TW_LDW + lane];
    // dU[n] = sum_t W1[t][n] dHpre[t], this wave's tokens
    if (pv) {
        for (int n = nb; n < ne; ++n) {
            const float4* wr = reinterpret_cast<const float4*>(w1s + n * TM);
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
            for (int t4 = 0; t4 < TM / 4; ++t4) {
                const float4 w = wr[t4];
                o0 = __builtin_fmaf(w.x, dh[4 * t4 + 0], o0);
                o1 = __builtin_fmaf(w.y, dh[4 * t4 + 1], o1);
                o2 = __builtin_fmaf(w.z, dh[4 * t4 + 2], o2);
                o3 = __builtin_fmaf(w.w, dh[4 * t4 + 3], o3);
            }
            du_out[col0 + (long)n * D] = (o0 + o1) + (o2 + o3);
        }
    }
    // parameter gradients: sums over the workgroup's 64 columns (fixed order), then one float atomic per value per
    // workgroup.  U and dV go through LDS one chunk of NC tokens at a time (recomputed from x_in / g_mid, which are
    // L2-resident by now), chunk c by wave c % NW in its own part of LDS -- the partial sums above are dead: every wave
    // passed the second barrier only after all had read them -- so the loop needs no further workgroup barrier.
    // (Measured and dropped: the two column sums as fp32 16x16x4 MFMAs fed from the LDS tiles -- 30 % slower at MIMIC's batch
    // 8192, where 8192 workgroups then reach their atomics on the same ~800 addresses together.)
    if (wave == 0 && lane < T) {
        const float* gr = dhs + lane * TW_LDW;
        float a = 0.f;
        for (int k = 0; k < TW_COLS; ++k) a += gr[k];
        gb1s[lane] += a;
    }
    float* us = part + wave * PART_F;                           // [NC][TW_LDW]  LN1 output of each column, one chunk of tokens
    float* dvs = us + NC * TW_LDW;                           // [NC][TW_LDW]  masked upstream gradient, same chunk
    static_assert(NC % TW_UB == 0, "a chunk is a whole number of load batches");
    for (int n0 = wave * NC; n0 < N; n0 += NW * NC) {
        const int nc = min(NC, N - n0);
        for (int j0 = 0; j0 < nc; j0 += TW_UB) {
            float u8[TW_UB], dv8[TW_UB];
            u_dv8(n0 + j0, N, u8, dv8);
#pragma unroll
            for (int j = 0; j < TW_UB; ++j) {
                if (j0 + j < nc) {
                    us[(j0 + j) * TW_LDW + lane] = u8[j];
                    dvs[(j0 + j) * TW_LDW + lane] = dv8[j];
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): this wave's LDS writes are done (wave-private tile)
        __builtin_amdgcn_wave_barrier();
        for (int p = lane; p < nc * T; p += TW_COLS) {
            const int j = p / T, t = p % T, n = n0 + j;
            const float* ur = us + j * TW_LDW;
            const float* vr = dvs + j * TW_LDW;
            const float* hr = hs + t * TW_LDW;
            const float* gr = dhs + t * TW_LDW;
            float a = 0.f, c = 0.f;
#pragma unroll 8
            for (int k = 0; k < TW_COLS; ++k) {
                a = __builtin_fmaf(gr[k], ur[k], a);         // dW1[t][n] += dHpre[t] U[n]
                c = __builtin_fmaf(vr[k], hr[k], c);         // dW2[n][t] += dV[n] Hact[t]
            }
            gw1s[t * N + n] += a;
            gw2s[n * T + t] += c;
        }
        if (lane < nc) {
            const float* vr = dvs + lane * TW_LDW;
            float a = 0.f;
            for (int k = 0; k < TW_COLS; ++k) a += vr[k];
            gb2s[n0 + lane] += a;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // reads done before the next chunk overwrites the tile
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                             // the next column block overwrites stats / hs / dhs
    }
    __syncthreads();
    for (int i = tid; i < T * N; i += (NW * 64)) {
        atomicAdd(bk.g_tok_w1 + i, gw1s[i]);
        atomicAdd(bk.g_tok_w2 + i, gw2s[i]);
    }
    if (tid < T) atomicAdd(bk.g_tok_b1 + tid, gb1s[tid]);
    for (int n = tid; n < N; n += (NW * 64)) atomicAdd(bk.g_tok_b2 + n, gb2s[n]);
}

template <int P, int DM, int TM, int NW>
__global__ __launch_bounds__(NW * 64) void token_bwd_cols_kernel(const m2m_tower tw, int b, const float* __restrict__ g_mid, int B,
                                                                    float* __restrict__ du_out, unsigned int seed,
                                                                    unsigned int step_host, const unsigned int* __restrict__ step_dev, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    token_bwd_cols_body<P, DM, TM, NW>(tw, b, blockIdx.x, g_mid, B, du_out, seed, step_host, step_dev, iters, smf);
}
template <int P, int DM, int TM>
__global__ __launch_bounds__(1024) void token_bwd_cols_ride_kernel(const m2m_tower tw, int b, const float* __restrict__ g_mid, int B,
                                                                   float* __restrict__ du_out, unsigned int seed, unsigned int step_host,
                                                                   const unsigned int* __restrict__ step_dev, int iters, int n_ride,
                                                                   const MlpRideBwd r) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    if ((int)blockIdx.x < n_ride) {
        mlp_bwd_mfma_body<1024>(r.m, r.x, r.B, r.d_out, r.d_out_ss, r.d_out2, blockIdx.x, smf);
        return;
    }
    token_bwd_cols_body<P, DM, TM, 16>(tw, b, blockIdx.x - n_ride, g_mid, B, du_out, seed, step_host, step_dev, iters, smf);
}
template <int P, int DM, int TM, int NW>
__global__ __launch_bounds__(NW * 64) void token_bwd_cols_group_kernel(const TokGroupArgs a, int b, int B, unsigned int seed,
                                                                          unsigned int step_host, const unsigned int* __restrict__ step_dev) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int t = blockIdx.y;
    if ((int)blockIdx.x >= a.nblk[t]) return;
    token_bwd_cols_body<P, DM, TM, NW>(a.tw[t], b, blockIdx.x, a.g_mid[t], B, a.dst[t], seed, step_host, step_dev, 1, smf);
}

// ---- backward, row part: dx_in = g_mid + LN1'(dU); gamma / beta gradients ------------------------------------------
// NWV waves, each walks rows (row = wave, wave + NWV, ...) of the workgroup's rows, two at a time (all six streams of a pair
// requested before the first reduction); a lane holds columns lane + 64 i.  The gamma / beta sums of a workgroup meet in LDS
// and leave as one float atomic per column: every workgroup of the launch adds to the same 2 D addresses, and same-address
// atomics queue in L2 (~30 ns each) -- at the cfg batches that queue WAS this kernel (320-400 workgroups of 8 rows: 8-17 us
// for 3 MB of traffic), so small launches take 16 waves and 32 rows per workgroup (a quarter of the atomics, same rows per wave).
#define LN_ROWS 32
// `dst` may alias `du` (dense, in place): a row's dU is read before its result is written.
struct Ln1Args {
    const float* x_in; const float* g_mid; const float* du; const float* gamma;
    long rows; int N, D; float* dst; long dst_ss; float* g_w; float* g_b;
};
template <int NWV>
static __device__ __forceinline__ void ln1_bwd_rows_body(const float* __restrict__ x_in, const float* __restrict__ g_mid,
                                                         const float* du, const float* __restrict__ gamma,
                                                         long rows, int N, int D, float* dst, long dst_ss,
                                                         float* __restrict__ g_w, float* __restrict__ g_b, int rows_per_wg, int bx,
                                                         float (*acc_w)[256], float (*acc_b)[256]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gw[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f}, gm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gm[i] = (lane + 64 * i) < D ? gamma[lane + 64 * i] : 0.f;
    const long r0 = (long)bx * rows_per_wg;
    constexpr int RB = 2;
    for (int rr = wave; rr < rows_per_wg; rr += RB * NWV) {
        float v[RB][4], u[RB][4], gmid[RB][4];
        bool ok[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {                      // all streams of the rows requested together
            const long r = r0 + rr + j * NWV;
            ok[j] = rr + j * NWV < rows_per_wg && r < rows;
            const long rc = ok[j] ? r : r0;                 // (clamped: the loads stay unconditional, the result is dropped)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + 64 * i;
                v[j][i] = c < D ? x_in[rc * D + c] : 0.f;
                u[j][i] = c < D ? du[rc * D + c] : 0.f;
                gmid[j][i] = c < D ? g_mid[rc * D + c] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const long r = r0 + rr + j * NWV;
            float s = (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
            s = wave_sum_xor(s, 64);
            const float mean = s / (float)D;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + 64 * i;
                v[j][i] = c < D ? v[j][i] - mean : 0.f;
                s2 = __builtin_fmaf(v[j][i], v[j][i], s2);
            }
            s2 = wave_sum_xor(s2, 64);
            const float vv = s2 / (float)D + 1e-5f;
            float rstd = __builtin_amdgcn_rsqf(vv);
            rstd = rstd * (1.5f - 0.5f * vv * rstd * rstd);
            float gsum = 0.f, gxsum = 0.f, gg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[j][i] *= rstd;                              // xhat
                gg[i] = u[j][i] * gm[i];
                gsum += gg[i];
                gxsum = __builtin_fmaf(gg[i], v[j][i], gxsum);
                if (ok[j]) {
                    gw[i] = __builtin_fmaf(u[j][i], v[j][i], gw[i]);
                    gb[i] += u[j][i];
                }
            }
            gsum = wave_sum_xor(gsum, 64) / (float)D;
            gxsum = wave_sum_xor(gxsum, 64) / (float)D;
            if (ok[j]) {
                float* orow = dst + (r / N) * dst_ss + (r % N) * D;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = lane + 64 * i;
                    if (c < D) orow[c] = gmid[j][i] + rstd * (gg[i] - gsum - v[j][i] * gxsum);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc_w[wave][lane + 64 * i] = gw[i]; acc_b[wave][lane + 64 * i] = gb[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += NWV * 64) {
        float sw = 0.f, sb = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { sw += acc_w[w][c]; sb += acc_b[w][c]; }
        atomicAdd(g_w + c, sw);
        atomicAdd(g_b + c, sb);
    }
}

template <int NWV>
__global__ __launch_bounds__(NWV * 64) void ln1_bwd_rows_kernel(const float* __restrict__ x_in, const float* __restrict__ g_mid,
                                                                const float* du, const float* __restrict__ gamma,
                                                                long rows, int N, int D, float* dst, long dst_ss,
                                                                float* __restrict__ g_w, float* __restrict__ g_b, int rows_per_wg) {
    __shared__ float acc_w[NWV][256], acc_b[NWV][256];
    ln1_bwd_rows_body<NWV>(x_in, g_mid, du, gamma, rows, N, D, dst, dst_ss, g_w, g_b, rows_per_wg, blockIdx.x, acc_w, acc_b);
}
struct Ln1GroupArgs { Ln1Args t[2]; };
template <int NWV>
__global__ __launch_bounds__(NWV * 64) void ln1_bwd_rows_group_kernel(const Ln1GroupArgs a, int rows_per_wg) {   // blockIdx.y = tower
    __shared__ float acc_w[NWV][256], acc_b[NWV][256];
    const Ln1Args& x = a.t[blockIdx.y];
    if ((long)blockIdx.x * rows_per_wg >= x.rows) return;
    ln1_bwd_rows_body<NWV>(x.x_in, x.g_mid, x.du, x.gamma, x.rows, x.N, x.D, x.dst, x.dst_ss, x.g_w, x.g_b, rows_per_wg, blockIdx.x, acc_w, acc_b);
}

// pooled[s][d] = mean over the N tokens of out[s]  (x.mean(dim=1), models/avmnist.py:271-272)
__global__ void token_mean_kernel(const float* __restrict__ out, long out_ss, int B, int N, int D, float* __restrict__ pooled) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * D) return;
    const long s = i / D, d = i % D;
    const float* col = out + s * out_ss + d;
    float a = 0.f;
    for (int n0 = 0; n0 < N; n0 += TW_UB) {                 // TW_UB loads in flight per round trip (same summation order)
        float v[TW_UB];
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) v[j] = col[(long)min(n0 + j, N - 1) * D];
#pragma unroll
        for (int j = 0; j < TW_UB; ++j)
            if (n0 + j < N) a += v[j];
    }
    pooled[i] = a * (1.0f / (float)N);
}

struct MeanGroupArgs { const float* out[2]; long out_ss[2]; int N[2]; float* pooled[2]; };
__global__ void token_mean_group_kernel(const MeanGroupArgs a, int B, int D) {                                    // blockIdx.y = tower
    const int t = blockIdx.y;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * D || !a.pooled[t]) return;
    const long s = i / D, d = i % D;
    const float* col = a.out[t] + s * a.out_ss[t] + d;
    const int N = a.N[t];
    float acc = 0.f;
    for (int n0 = 0; n0 < N; n0 += TW_UB) {
        float v[TW_UB];
#pragma unroll
        for (int j = 0; j < TW_UB; ++j) v[j] = col[(long)min(n0 + j, N - 1) * D];
#pragma unroll
        for (int j = 0; j < TW_UB; ++j)
            if (n0 + j < N) acc += v[j];
    }
    a.pooled[t][i] = acc * (1.0f / (float)N);
}

// ---- host side ------------------------------------------------------------------------------------------------------
static m2m_tower block_view(const m2m_tower* t, int b) {
    m2m_tower v = *t;
    v.nblocks = 1;
    v.blk[0] = t->blk[b];
    v.site_base = t->site_base + 4u * (unsigned int)b;
    v.has_final_ln = (b == t->nblocks - 1) ? t->has_final_ln : 0;
    v.dx0_chn = nullptr;        // (the chain launch of a view does not produce the tower-input gradient: token mixing follows)
    return v;
}

// waves per workgroup of a token launch with `nblk` column blocks (see TokNC above)
static int tok_waves(int nblk, int TM) {
    static const int forced = [] { const char* e = getenv("M2M_TOKEN_WAVES"); return e ? atoi(e) : 0; }();   // diagnostic (A/B): 4, 8 or 16
    int nw = nblk <= 256 ? 16 : (nblk <= 512 ? 8 : 4);
    if (forced == 4 || forced == 8 || forced == 16) nw = forced;
    if (TM > 16 && nw > 8) nw = 8;                               // 32 hidden units per lane: registers and LDS of 16 waves do not fit
    return nw;
}

// the recorded MLP call (per host thread; kind 0 none, 1 forward, 2 backward)
struct MlpRidePending { int kind; MlpRideFwd f; MlpRideBwd b; };
static thread_local MlpRidePending g_ride = {};
static size_t mlp_ride_lds(bool bwd) { return sizeof(float) * ((size_t)(bwd ? 3 : 2) * MLPM_S + MLP_MAXW) * (MLP_MAXW + 1); }

template <int P, int DM, int TM, int NW>
static int launch_token_fwd_nw(const m2m_tower* t, int b, const float* src, long src_ss, int B, float* x_mid, float* save_x_in,
                            int training, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const TokGeom g = tok_geom(t->D);
    const int grid = ((B + g.spw - 1) / g.spw) * g.chunks;
    const size_t lds = ((tok_lds_floats(t->N, g.spw, TM) + 3) & ~(size_t)3) * sizeof(float) +
                       (size_t)(NW + 1) * TM * TW_COLS * sizeof(float);
    if constexpr (NW == 16 && TM <= 16) {
        if (g_ride.kind == 1 && grid <= 256) {
            const MlpRideFwd r = g_ride.f;
            g_ride.kind = 0;
            const int n_ride = (r.B + MLPM_S - 1) / MLPM_S;
            const size_t rl = std::max(lds, mlp_ride_lds(false));
            auto rk = token_fwd_ride_kernel<P, DM, TM>;
            static size_t ride_attr = 0;
            if (rl > ride_attr) {
                M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
                ride_attr = rl;
            }
            hipLaunchKernelGGL(rk, dim3(grid + n_ride), dim3(1024), rl, st, *t, b, src, src_ss, B, x_mid, save_x_in, training, seed, step, step_dev,
                               n_ride, r);
            M2M_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    auto kern = token_fwd_kernel<P, DM, TM, NW>;
    static size_t attr_lds = 48 * 1024;
    if (lds > attr_lds) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, *t, b, src, src_ss, B, x_mid, save_x_in, training, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int DM, int TM>
static int launch_token_fwd(const m2m_tower* t, int b, const float* src, long src_ss, int B, float* x_mid, float* save_x_in,
                            int training, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const TokGeom g = tok_geom(t->D);
    const int nw = tok_waves(((B + g.spw - 1) / g.spw) * g.chunks, TM);
    if constexpr (TM <= 16) { if (nw == 16) return launch_token_fwd_nw<P, DM, TM, 16>(t, b, src, src_ss, B, x_mid, save_x_in, training, seed, step, step_dev, st); }
    if (nw >= 8) return launch_token_fwd_nw<P, DM, TM, 8>(t, b, src, src_ss, B, x_mid, save_x_in, training, seed, step, step_dev, st);
    return launch_token_fwd_nw<P, DM, TM, 4>(t, b, src, src_ss, B, x_mid, save_x_in, training, seed, step, step_dev, st);
}
template <int P, int DM, int TM, int NW>
static int launch_token_bwd_nw(const m2m_tower* t, int b, const float* g_mid, int B, float* du, unsigned int seed, unsigned int step,
                            const unsigned int* step_dev, hipStream_t st) {
    const TokGeom g = tok_geom(t->D);
    const int nblk = ((B + g.spw - 1) / g.spw) * g.chunks;
    // column blocks per workgroup: one until the launch has more than ~1024 workgroups (4 per CU), then enough to stay there
    // -- the token-weight gradients end in float atomics on the same few hundred addresses from every workgroup
    const int iters = nblk > 1024 ? (nblk + 1023) / 1024 : 1;
    const int grid = (nblk + iters - 1) / iters;
    constexpr int NC = TokNC<NW>::value;
    const size_t part_f = 2 * TM * TW_COLS > 2 * NC * TW_LDW ? 2 * TM * TW_COLS : 2 * NC * TW_LDW;
    const size_t lds = ((tok_lds_floats(t->N, g.spw, TM) + 3) & ~(size_t)3) * sizeof(float) +
                       ((size_t)2 * TM * TW_LDW + NW * part_f + 2 * t->N * TM + TM + t->N) * sizeof(float);
    if (lds > 160 * 1024) { m2m_set_error("token backward: tokens x token_dim exceed the workgroup's LDS", __FILE__, __LINE__); return -1; }
    if constexpr (NW == 16 && TM <= 16) {
        if (g_ride.kind == 2 && grid <= 256) {
            const MlpRideBwd r = g_ride.b;
            g_ride.kind = 0;
            const int n_ride = (r.B + MLPM_S - 1) / MLPM_S;
            const size_t rl = std::max(lds, mlp_ride_lds(true));
            auto rk = token_bwd_cols_ride_kernel<P, DM, TM>;
            static size_t ride_attr = 0;
            if (rl > ride_attr) {
                M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
                ride_attr = rl;
            }
            hipLaunchKernelGGL(rk, dim3(grid + n_ride), dim3(1024), rl, st, *t, b, g_mid, B, du, seed, step, step_dev, iters, n_ride, r);
            M2M_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    auto kern = token_bwd_cols_kernel<P, DM, TM, NW>;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, *t, b, g_mid, B, du, seed, step, step_dev, iters);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int DM, int TM>
static int launch_token_bwd(const m2m_tower* t, int b, const float* g_mid, int B, float* du, unsigned int seed, unsigned int step,
                            const unsigned int* step_dev, hipStream_t st) {
    const TokGeom g = tok_geom(t->D);
    const int nw = tok_waves(((B + g.spw - 1) / g.spw) * g.chunks, TM);
    if constexpr (TM <= 16) { if (nw == 16) return launch_token_bwd_nw<P, DM, TM, 16>(t, b, g_mid, B, du, seed, step, step_dev, st); }
    if (nw >= 8) return launch_token_bwd_nw<P, DM, TM, 8>(t, b, g_mid, B, du, seed, step, step_dev, st);
    return launch_token_bwd_nw<P, DM, TM, 4>(t, b, g_mid, B, du, seed, step, step_dev, st);
}
#define M2M_TOK_DISPATCH_TM(FN, TM_, training_, ...)                                                \
    do {                                                                                            \
        const int dm_ = m2m_drop_mode(training_, t->p_drop);                                       \
        if (t->prec == PREC_BF16) {                                                                 \
            if (dm_ == DM_NONE) return FN<PREC_BF16, DM_NONE, TM_>(__VA_ARGS__);                   \
            if (dm_ == DM_HALF) return FN<PREC_BF16, DM_HALF, TM_>(__VA_ARGS__);                   \
            return FN<PREC_BF16, DM_GEN, TM_>(__VA_ARGS__);                                        \
        }                                                                                           \
        if (dm_ == DM_NONE) return FN<PREC_F32, DM_NONE, TM_>(__VA_ARGS__);                        \
        if (dm_ == DM_HALF) return FN<PREC_F32, DM_HALF, TM_>(__VA_ARGS__);                        \
        return FN<PREC_F32, DM_GEN, TM_>(__VA_ARGS__);                                             \
    } while (0)
#define M2M_TOK_DISPATCH(FN, training_, ...)                                                        \
    do {                                                                                            \
        if (t->T <= 16) M2M_TOK_DISPATCH_TM(FN, 16, training_, __VA_ARGS__);                       \
        M2M_TOK_DISPATCH_TM(FN, 32, training_, __VA_ARGS__);                                       \
    } while (0)

static int token_fwd(const m2m_tower* t, int b, const float* src, long src_ss, int B, float* x_mid, float* save_x_in, int training,
                     unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    M2M_TOK_DISPATCH(launch_token_fwd, training, t, b, src, src_ss, B, x_mid, save_x_in, training, seed, step, step_dev, st);
}
static int token_bwd(const m2m_tower* t, int b, const float* g_mid, int B, float* du, unsigned int seed, unsigned int step,
                     const unsigned int* step_dev, hipStream_t st) {
    M2M_TOK_DISPATCH(launch_token_bwd, 1, t, b, g_mid, B, du, seed, step, step_dev, st);
}

// ---- two wide towers per launch (m2m_towers_forward / _backward) ---------------------------------------------------------
// Conditions (m2m_can_group_wide): same precision, hidden_dim 256, dropout, token_dim class and block count (<= 4), small launches
// (one column block per workgroup).  MM-IMDb's image and text towers; every launch of the pair is then ONE launch on the main
// stream: the two-queue form paid two fork / join pairs per step (~5 us per edge in a replayed graph) and staggered launches.
int m2m_chain_forward_rows_group(const m2m_tower* const* v, const float* const* x0, const long* x0_ss, int B, float* const* out,
                                 const long* out_ss, int training, unsigned int seed, unsigned int step, const unsigned int* step_dev,
                                 hipStream_t st);                                                        // tower_fwd.hip
int m2m_chain_backward_rows_group(const m2m_tower* const* v, int B, const float* const* d_out, const long* d_out_ss,
                                  const float* const* d_pooled, float* const* d_x0, const long* d_x0_ss, unsigned int seed,
                                  unsigned int step, const unsigned int* step_dev, hipStream_t st);     // tower_bwd.hip
bool m2m_can_group_wide(const m2m_tower* a, const m2m_tower* b, int B) {
    static const int off = [] { const char* e = getenv("M2M_WIDE_GROUP"); return e && e[0] == '0'; }();   // diagnostic (A/B)
    if (off || !m2m_is_wide(a) || !m2m_is_wide(b)) return false;
    if (a->prec != b->prec || a->D != b->D || a->D != 256 || a->p_drop != b->p_drop) return false;
    if ((a->T <= 16) != (b->T <= 16) || a->T > 16) return false;        // (16 waves per workgroup: token_dim <= 16)
    if (a->nblocks != b->nblocks || a->nblocks < 1 || a->nblocks > M2M_GROUP_BLOCKS) return false;
    const TokGeom g = tok_geom(a->D);
    return ((B + g.spw - 1) / g.spw) * g.chunks <= 256;                  // the 16-wave, one-block-per-workgroup regime
}

template <int P, int DM, int TM>
static int launch_token_fwd_group(const TokGroupArgs& a, int b, int B, int N, int training, unsigned int seed, unsigned int step,
                                  const unsigned int* step_dev, hipStream_t st) {
    constexpr int NW = 16;
    const TokGeom g = tok_geom(a.tw[0].D);
    const size_t lds = ((tok_lds_floats(N, g.spw, TM) + 3) & ~(size_t)3) * sizeof(float) + (size_t)(NW + 1) * TM * TW_COLS * sizeof(float);
    auto kern = token_fwd_group_kernel<P, DM, TM, NW>;
    static size_t attr_lds = 48 * 1024;
    if (lds > attr_lds) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int mx = a.nblk[0] > a.nblk[1] ? a.nblk[0] : a.nblk[1];
    hipLaunchKernelGGL(kern, dim3(mx, 2), dim3(NW * 64), lds, st, a, b, B, training, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
template <int P, int DM, int TM>
static int launch_token_bwd_group(const TokGroupArgs& a, int b, int B, int N, int training_unused, unsigned int seed, unsigned int step,
                                  const unsigned int* step_dev, hipStream_t st) {
    constexpr int NW = 16, NC = TokNC<NW>::value;
    const TokGeom g = tok_geom(a.tw[0].D);
    const size_t part_f = 2 * TM * TW_COLS > 2 * NC * TW_LDW ? 2 * TM * TW_COLS : 2 * NC * TW_LDW;
    const size_t lds = ((tok_lds_floats(N, g.spw, TM) + 3) & ~(size_t)3) * sizeof(float) +
                       ((size_t)2 * TM * TW_LDW + NW * part_f + 2 * N * TM + TM + N) * sizeof(float);
    if (lds > 160 * 1024) { m2m_set_error("token backward (group): tokens x token_dim exceed the workgroup's LDS", __FILE__, __LINE__); return -1; }
    auto kern = token_bwd_cols_group_kernel<P, DM, TM, NW>;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int mx = a.nblk[0] > a.nblk[1] ? a.nblk[0] : a.nblk[1];
    hipLaunchKernelGGL(kern, dim3(mx, 2), dim3(NW * 64), lds, st, a, b, B, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}
// (both towers: token_dim <= 16 -- m2m_can_group_wide)
#define M2M_TOKG_DISPATCH(FN, training_, p_drop_, prec_, ...)                                       \
    do {                                                                                            \
        const int dm_ = m2m_drop_mode(training_, p_drop_);                                         \
        if (prec_ == PREC_BF16) {                                                                   \
            if (dm_ == DM_NONE) return FN<PREC_BF16, DM_NONE, 16>(__VA_ARGS__);                    \
            if (dm_ == DM_HALF) return FN<PREC_BF16, DM_HALF, 16>(__VA_ARGS__);                    \
            return FN<PREC_BF16, DM_GEN, 16>(__VA_ARGS__);                                         \
        }                                                                                           \
        if (dm_ == DM_NONE) return FN<PREC_F32, DM_NONE, 16>(__VA_ARGS__);                         \
        if (dm_ == DM_HALF) return FN<PREC_F32, DM_HALF, 16>(__VA_ARGS__);                         \
        return FN<PREC_F32, DM_GEN, 16>(__VA_ARGS__);                                              \
    } while (0)
static int token_fwd_group(const TokGroupArgs& a, int b, int B, int N, int training, unsigned int seed, unsigned int step,
                           const unsigned int* step_dev, hipStream_t st) {
    M2M_TOKG_DISPATCH(launch_token_fwd_group, training, a.tw[0].p_drop, a.tw[0].prec, a, b, B, N, training, seed, step, step_dev, st);
}
static int token_bwd_group(const TokGroupArgs& a, int b, int B, int N, unsigned int seed, unsigned int step,
                           const unsigned int* step_dev, hipStream_t st) {
    M2M_TOKG_DISPATCH(launch_token_bwd_group, 1, a.tw[0].p_drop, a.tw[0].prec, a, b, B, N, 1, seed, step, step_dev, st);
}

int m2m_forward_wide_group(const m2m_tower* const* tw, const m2m_tower_io* io, int B, int training, unsigned int seed,
                           unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    if (!tw[0]->ws_a || !tw[0]->ws_b || !tw[1]->ws_a || !tw[1]->ws_b) { m2m_set_error("wide path (N > 8 or D > 128) needs the ws_a / ws_b workspaces", __FILE__, __LINE__); return -1; }
    const int nb = tw[0]->nblocks;
    const TokGeom g = tok_geom(tw[0]->D);
    const int nblk = ((B + g.spw - 1) / g.spw) * g.chunks;
    const int Nmax = tw[0]->N > tw[1]->N ? tw[0]->N : tw[1]->N;
    const float* src[2]; long src_ss[2]; long dense[2];
    for (int i = 0; i < 2; ++i) {
        if (io[i].x0_parts > 1) { m2m_set_error("towers_forward (wide): k-split inputs are not supported", __FILE__, __LINE__); return -1; }
        src[i] = io[i].x0; src_ss[i] = (long)io[i].x0_sample_stride; dense[i] = (long)tw[i]->N * tw[i]->D;
    }
    for (int b = 0; b < nb; ++b) {
        const bool last = b == nb - 1;
        TokGroupArgs a;
        memset(&a, 0, sizeof(a));
        float* mid[2]; float* nxt[2]; long nxt_ss[2];
        m2m_tower v[2]; const m2m_tower* vp[2] = {&v[0], &v[1]};
        for (int i = 0; i < 2; ++i) {
            const m2m_block& bk = tw[i]->blk[b];
            mid[i] = training ? bk.x_mid : tw[i]->ws_a;
            a.tw[i] = m2m_shrink(tw[i]);
            a.src[i] = src[i]; a.src_ss[i] = src_ss[i]; a.dst[i] = mid[i];
            a.save[i] = (training && src[i] != bk.x_in) ? bk.x_in : nullptr;
            a.nblk[i] = nblk;
            nxt[i] = last ? io[i].out : (training ? tw[i]->blk[b + 1].x_in : tw[i]->ws_b);
            nxt_ss[i] = last ? (long)io[i].out_sample_stride : dense[i];
            v[i] = block_view(tw[i], b);
        }
        if (int rc = token_fwd_group(a, b, B, Nmax, training, seed, step, step_dev, st)) return rc;
        const float* midc[2] = {mid[0], mid[1]};
        if (int rc = m2m_chain_forward_rows_group(vp, midc, dense, B, nxt, nxt_ss, training, seed, step, step_dev, st)) return rc;
        for (int i = 0; i < 2; ++i) { src[i] = nxt[i]; src_ss[i] = nxt_ss[i]; }
    }
    if (io[0].pooled || io[1].pooled) {
        MeanGroupArgs m;
        for (int i = 0; i < 2; ++i) { m.out[i] = io[i].out; m.out_ss[i] = (long)io[i].out_sample_stride; m.N[i] = tw[i]->N; m.pooled[i] = io[i].pooled; }
        const long n = (long)B * tw[0]->D;
        hipLaunchKernelGGL(token_mean_group_kernel, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, st, m, B, tw[0]->D);
        M2M_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

int m2m_backward_wide_group(const m2m_tower* const* tw, const m2m_tower_gio* io, int B, unsigned int seed, unsigned int step,
                            const unsigned int* step_dev, hipStream_t st) {
    if (!tw[0]->ws_a || !tw[0]->ws_b || !tw[1]->ws_a || !tw[1]->ws_b) { m2m_set_error("wide path (N > 8 or D > 128) needs the ws_a / ws_b workspaces", __FILE__, __LINE__); return -1; }
    const int nb = tw[0]->nblocks;
    const TokGeom g = tok_geom(tw[0]->D);
    const int nblk = ((B + g.spw - 1) / g.spw) * g.chunks;
    const int Nmax = tw[0]->N > tw[1]->N ? tw[0]->N : tw[1]->N;
    const float* up[2]; long up_ss[2]; const float* up_pooled[2]; long dense[2]; long rows[2];
    for (int i = 0; i < 2; ++i) {
        up[i] = io[i].d_out; up_ss[i] = (long)io[i].d_out_sample_stride; up_pooled[i] = io[i].d_pooled;
        dense[i] = (long)tw[i]->N * tw[i]->D; rows[i] = (long)B * tw[i]->N;
    }
    for (int b = nb - 1; b >= 0; --b) {
        m2m_tower v[2]; const m2m_tower* vp[2] = {&v[0], &v[1]};
        float* gmid[2];
        for (int i = 0; i < 2; ++i) { v[i] = block_view(tw[i], b); gmid[i] = tw[i]->ws_b; }
        // gradient wrt x_mid -> ws_b
        if (int rc = m2m_chain_backward_rows_group(vp, B, up, up_ss, up_pooled, gmid, dense, seed, step, step_dev, st)) return rc;
        // dU -> ws_a, token parameter gradients
        TokGroupArgs a;
        memset(&a, 0, sizeof(a));
        for (int i = 0; i < 2; ++i) { a.tw[i] = m2m_shrink(tw[i]); a.g_mid[i] = tw[i]->ws_b; a.dst[i] = tw[i]->ws_a; a.nblk[i] = nblk; }
        if (int rc = token_bwd_group(a, b, B, Nmax, seed, step, step_dev, st)) return rc;
        // gradient wrt the block input -> ws_a in place (row-local), or the caller's buffer for the first block
        Ln1GroupArgs l;
        long maxrows = 0;
        for (int i = 0; i < 2; ++i) {
            const m2m_block& bk = tw[i]->blk[b];
            Ln1Args& x = l.t[i];
            x.x_in = bk.x_in; x.g_mid = tw[i]->ws_b; x.du = tw[i]->ws_a; x.gamma = bk.ln1_w; x.rows = rows[i]; x.N = tw[i]->N; x.D = tw[i]->D;
            x.dst = b > 0 ? tw[i]->ws_a : io[i].d_x0; x.dst_ss = b > 0 ? dense[i] : (long)io[i].d_x0_sample_stride;
            x.g_w = bk.g_ln1_w; x.g_b = bk.g_ln1_b;
            maxrows = rows[i] > maxrows ? rows[i] : maxrows;
        }
        hipLaunchKernelGGL(ln1_bwd_rows_group_kernel<16>, dim3((unsigned)((maxrows + LN_ROWS - 1) / LN_ROWS), 2), dim3(1024), 0, st, l, LN_ROWS);
        M2M_CHECK_HIP(hipGetLastError());
        for (int i = 0; i < 2; ++i) { up[i] = tw[i]->ws_a; up_ss[i] = dense[i]; up_pooled[i] = nullptr; }
    }
    return 0;
}

int m2m_forward_wide(const m2m_tower* t, const float* x0, long x0_ss, int B, float* out, long out_ss, float* pooled,
                     int training, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const long dense = (long)t->N * t->D;
    if (!t->ws_a || !t->ws_b) { m2m_set_error("wide path (N > 8 or D > 128) needs the ws_a / ws_b workspaces", __FILE__, __LINE__); return -1; }
    if (t->nblocks == 0) {      // final LayerNorm only
        m2m_tower v = *t;
        if (int rc = m2m_chain_forward_rows(&v, x0, x0_ss, B, out, out_ss, training, seed, step, step_dev, st)) return rc;
    }
    const float* src = x0;
    long src_ss = x0_ss;
    for (int b = 0; b < t->nblocks; ++b) {
        const m2m_block& bk = t->blk[b];
        const bool last = b == t->nblocks - 1;
        // training: the saved activations ARE the stream (x_in[b] -> x_mid[b] -> x_in[b+1]); eval: the two workspaces
        float* mid = training ? bk.x_mid : t->ws_a;
        float* save = (training && src != bk.x_in) ? bk.x_in : nullptr;
        if (int rc = token_fwd(t, b, src, src_ss, B, mid, save, training, seed, step, step_dev, st)) return rc;
        float* nxt = last ? out : (training ? t->blk[b + 1].x_in : t->ws_b);
        const long nxt_ss = last ? out_ss : dense;
        const m2m_tower v = block_view(t, b);
        if (int rc = m2m_chain_forward_rows(&v, mid, dense, B, nxt, nxt_ss, training, seed, step, step_dev, st)) return rc;
        src = nxt;
        src_ss = nxt_ss;
    }
    if (pooled) {
        const long n = (long)B * t->D;
        hipLaunchKernelGGL(token_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, out_ss, B, t->N, t->D, pooled);
        M2M_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

int m2m_backward_wide(const m2m_tower* t, int B, const float* d_out, long d_out_ss, const float* d_pooled, float* d_x0,
                      long d_x0_ss, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const long dense = (long)t->N * t->D;
    const long rows = (long)B * t->N;
    if (!t->ws_a || !t->ws_b) { m2m_set_error("wide path (N > 8 or D > 128) needs the ws_a / ws_b workspaces", __FILE__, __LINE__); return -1; }
    if (t->nblocks == 0) {
        m2m_tower v = *t;
        return m2m_chain_backward_rows(&v, B, d_out, d_out_ss, d_pooled, d_x0, d_x0_ss, seed, step, step_dev, st);
    }
    const float* up = d_out;
    long up_ss = d_out_ss;
    const float* up_pooled = d_pooled;
    for (int b = t->nblocks - 1; b >= 0; --b) {
        const m2m_block& bk = t->blk[b];
        const m2m_tower v = block_view(t, b);
        // gradient wrt x_mid -> ws_b
        if (int rc = m2m_chain_backward_rows(&v, B, up, up_ss, up_pooled, t->ws_b, dense, seed, step, step_dev, st)) return rc;
        // dU -> ws_a, token parameter gradients
        if (int rc = token_bwd(t, b, t->ws_b, B, t->ws_a, seed, step, step_dev, st)) return rc;
        // gradient wrt the block input -> ws_a in place (row-local), or the caller's buffer for the first block
        float* dst = b > 0 ? t->ws_a : d_x0;
        const long dst_ss = b > 0 ? dense : d_x0_ss;
        // rows per workgroup: 32 (8 per wave) when there are plenty, 8 at small batch so that the chip is not left to 80
        // workgroups, 128 when there are so many that the gamma / beta atomics of thousands of workgroups queue up
        static const int ln_small = [] { const char* e = getenv("M2M_LN1_SMALL"); return e ? atoi(e) : 1; }();   // diagnostic (A/B)
        if (rows < 16384 && ln_small) {                            // 16 waves x 2 rows
            hipLaunchKernelGGL(ln1_bwd_rows_kernel<16>, dim3((unsigned)((rows + LN_ROWS - 1) / LN_ROWS)), dim3(1024), 0, st, bk.x_in, t->ws_b,
                               t->ws_a, bk.ln1_w, rows, t->N, t->D, dst, dst_ss, bk.g_ln1_w, bk.g_ln1_b, LN_ROWS);
        } else {
            const int rpw = rows >= 131072 ? 4 * LN_ROWS : (rows >= 16384 ? LN_ROWS : 8);   // (large: fewer workgroups on the same 2 D atomics)
            hipLaunchKernelGGL(ln1_bwd_rows_kernel<4>, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(256), 0, st, bk.x_in, t->ws_b,
                               t->ws_a, bk.ln1_w, rows, t->N, t->D, dst, dst_ss, bk.g_ln1_w, bk.g_ln1_b, rpw);
        }
        M2M_CHECK_HIP(hipGetLastError());
        up = t->ws_a;
        up_ss = dense;
        up_pooled = nullptr;
    }
    return 0;
}

// ---- the MLP tower riding in a token-mixing launch (see token_fwd_ride_kernel) --------------------------------------------------------
static int mlp_ride_ok(const m2m_mlp* m, int B) {
    if (!m || B < 1 || B > 2048 || m->nlayers < 1 || m->nlayers > M2M_MLP_MAX_LAYERS) return 0;
    for (int i = 0; i <= m->nlayers; ++i)
        if (m->dims[i] < 1 || m->dims[i] > MLP_MAXW) return 0;
    return 1;
}
extern "C" int m2m_mlp_forward_ride(const m2m_mlp* m, const float* x, int B, float* out, int64_t out_sample_stride, float* out_dense,
                                    int training, uint32_t seed, uint32_t step, const uint32_t* step_dev) {
    if (!mlp_ride_ok(m, B) || !x || !out) { m2m_set_error("mlp_forward_ride: bad argument (batch <= 2048, widths <= 128)", __FILE__, __LINE__); return -1; }
    if (g_ride.kind) { m2m_set_error("mlp_forward_ride: another MLP call is still pending (m2m_mlp_ride_flush)", __FILE__, __LINE__); return -1; }
    g_ride.f = MlpRideFwd{*m, x, B, out, (long)out_sample_stride, out_dense, training, seed, step, step_dev};
    g_ride.kind = 1;
    return 0;
}
extern "C" int m2m_mlp_backward_ride(const m2m_mlp* m, const float* x, int B, const float* d_out, int64_t d_out_sample_stride,
                                     const float* d_out_dense) {
    if (!mlp_ride_ok(m, B) || !x) { m2m_set_error("mlp_backward_ride: bad argument (batch <= 2048, widths <= 128)", __FILE__, __LINE__); return -1; }
    if (g_ride.kind) { m2m_set_error("mlp_backward_ride: another MLP call is still pending (m2m_mlp_ride_flush)", __FILE__, __LINE__); return -1; }
    g_ride.b = MlpRideBwd{*m, x, B, d_out, (long)d_out_sample_stride, d_out_dense};
    g_ride.kind = 2;
    return 0;
}
extern "C" int m2m_mlp_ride_flush(void* stream) {
    const int kind = g_ride.kind;
    g_ride.kind = 0;
    if (kind == 1) {
        const MlpRideFwd& r = g_ride.f;
        return m2m_mlp_forward(&r.m, r.x, r.B, r.out, r.out_ss, r.out2, r.training, r.seed, r.step, r.step_dev, stream);
    }
    if (kind == 2) {
        const MlpRideBwd& r = g_ride.b;
        return m2m_mlp_backward(&r.m, r.x, r.B, r.d_out, r.d_out_ss, r.d_out2, stream);
    }
    return 0;
}
