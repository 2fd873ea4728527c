// Channel-mixing weight gradients of every block of a tower: g_ch_w1, g_ch_b1, g_ch_w2.
//
//   dW1[c][d] = sum_m dHpre[m][c] A[m][d]      dW2[d][c] = sum_m dYd[m][d] Hact[m][c]      db1[c] = sum_m dHpre[m][c]
//
// The contraction runs over ALL token rows, so the roles flip relative to the forward/backward chain:
// a workgroup owns 128 hidden columns of one block (each wave 32, its W1 / W2^T fragments and its
// 32x128 slices of dW1 and dW2^T stay in registers) and streams the 64-row operand tiles that
// tower_bwd.hip wrote (A, A^T, dYd, dYd^T, already in packed MFMA order) through LDS.  Per tile it
// recomputes Hpre = A W1^T + b1 and dHact = dYd W2 (hidden activations are never stored), applies
// GELU / GELU' / dropout on the accumulators and chains them (k = row index) into the two products.
// Row groups (grid.z) that share a column slice add their partial results with float atomics.
#include "tile.h"

template <int P, int D>
__global__ __launch_bounds__(NTHREADS) void tower_wgrad_kernel(const m2m_tower tw, int ntiles, int tiles_per_group,
                                                               int rows_per_tile, unsigned int seed, unsigned int step_host,
                                                               const unsigned int* __restrict__ step_dev) {
    typedef Prec<P> Pr;
    constexpr int DT = D / 16, KD = D / Pr::KB, NF = Chain<P>::NF;
    constexpr int IMG_B = BM * D * Pr::ESZ;
    constexpr int NKM = BM / Pr::KB;                 // k-blocks over the 64 rows of a tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_nat = smem;
    char* dy_nat = smem + IMG_B;
    char* at_chn = smem + 2 * IMG_B;
    char* dyt_chn = smem + 3 * IMG_B;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, il = lane & 15;
    const m2m_block& bk = tw.blk[blockIdx.y];
    const int Cp = tw.Cp, C = tw.C;
    const int q = blockIdx.x * 4 + wave;             // this wave's pair of 16-column tiles
    const bool active = q < (Cp >> 5);
    const unsigned int step = step_host + (step_dev ? *step_dev : 0u);
    const unsigned int site = tw.site_base + 4u * blockIdx.y;
    const Drop dr_ch = make_drop(true, tw.p_drop, seed, step, site + 2);
    const bool dropping = dr_ch.thr < 65536u;

    Frag w1f[2][KD], w2f[2][KD];
    float bias[2];
    f32x4_t dw1[2][DT], dw2[2][DT];
    float db1[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) {
            w1f[t][kb].u = u32x4_t{0u, 0u, 0u, 0u};
            w2f[t][kb].u = u32x4_t{0u, 0u, 0u, 0u};
            if (active) {
                w1f[t][kb] = ld_frag_global(bk.w1n, (long)(2 * q + t) * KD + kb, lane);
                w2f[t][kb] = ld_frag_global(bk.w2tn, (long)(2 * q + t) * KD + kb, lane);
            }
        }
        bias[t] = active ? bk.ch_b1p[32 * q + 16 * t + il] : 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            dw1[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dw2[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }

    const int t_begin = blockIdx.z * tiles_per_group;
    const int t_end = min(ntiles, t_begin + tiles_per_group);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const long off = (long)tile * IMG_B;
        __syncthreads();                               // previous tile fully consumed
        copy16(a_nat, reinterpret_cast<const char*>(bk.a_nat) + off, IMG_B, tid);
        copy16(dy_nat, reinterpret_cast<const char*>(bk.dy_nat) + off, IMG_B, tid);
        copy16(at_chn, reinterpret_cast<const char*>(bk.at_chn) + off, IMG_B, tid);
        copy16(dyt_chn, reinterpret_cast<const char*>(bk.dyt_chn) + off, IMG_B, tid);
        __syncthreads();
        if (!active) continue;
        const unsigned int mrow0 = (unsigned int)tile * rows_per_tile;   // global token row of the tile's row 0

        // two 16-row tiles at a time: their accumulators chain into one k-block (bf16) / two (fp32)
#pragma unroll
        for (int mb = 0; mb < MT / 2; ++mb) {
            f32x4_t hact[2][2], dhp[2][2];             // [row tile in pair][column tile]
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int mt = 2 * mb + u;
                f32x4_t hacc[2] = {f32x4_t{bias[0], bias[0], bias[0], bias[0]}, f32x4_t{bias[1], bias[1], bias[1], bias[1]}};
                f32x4_t gacc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kb = 0; kb < KD; ++kb) {
                    const Frag a = ld_frag_lds(a_nat, mt * KD + kb, lane);
                    const Frag dy = ld_frag_lds(dy_nat, mt * KD + kb, lane);
                    Pr::mma(hacc[0], a, w1f[0][kb]);
                    Pr::mma(hacc[1], a, w1f[1][kb]);
                    Pr::mma(gacc[0], dy, w2f[0][kb]);
                    Pr::mma(gacc[1], dy, w2f[1][kb]);
                }
                // accumulator element r: row m = 16 mt + 4 g + r, column c = 32 q + 16 t + il
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float gl, dgl;
                        gelu_grad_f(hacc[t][r], gl, dgl);
                        float hv = gl, dv = gacc[t][r] * dgl;
                        if (dropping) {
                            const unsigned int m = mrow0 + 16 * mt + 4 * g + r;
                            const bool keep = drop_keep(dr_ch, m * (unsigned int)Cp + 32 * q + 16 * t + il);
                            hv = keep ? hv * dr_ch.scale : 0.f;
                            dv = keep ? dv * dr_ch.scale : 0.f;
                        }
                        hact[u][t][r] = hv;
                        dhp[u][t][r] = dv;
                        db1[t] += dv;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag hf[NF], df[NF];
                Chain<P>::make(hact[0][t], hact[1][t], hf);
                Chain<P>::make(dhp[0][t], dhp[1][t], df);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int kbm = mb * NF + f;       // k-block over the tile's rows
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const Frag at = ld_frag_lds(at_chn, kbm * DT + dt, lane);
                        const Frag dyt = ld_frag_lds(dyt_chn, kbm * DT + dt, lane);
                        Pr::mma(dw1[t][dt], df[f], at);
                        Pr::mma(dw2[t][dt], hf[f], dyt);
                    }
                }
            }
        }
    }

    if (!active) return;
    const bool single = gridDim.z == 1;
    // ---- results: dw1[t][dt][r] = dW1[c = 32q + 16t + 4g + r][d = 16dt + il]; dw2 likewise = dW2[d][c] ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 32 * q + 16 * t + 4 * g + r, d = 16 * dt + il;
                if (c < C) {
                    if (single) {       // sole owner of these elements: plain read-modify-write
                        bk.g_ch_w1[(long)c * D + d] += dw1[t][dt][r];
                        bk.g_ch_w2[(long)d * C + c] += dw2[t][dt][r];
                    } else {
                        atomicAdd(bk.g_ch_w1 + (long)c * D + d, dw1[t][dt][r]);
                        atomicAdd(bk.g_ch_w2 + (long)d * C + c, dw2[t][dt][r]);
                    }
                }
            }
        float s = db1[t];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const int c = 32 * q + 16 * t + il;
        if (g == 0 && c < C) { if (single) bk.g_ch_b1[c] += s; else atomicAdd(bk.g_ch_b1 + c, s); }
    }
}

template <int P, int D>
static int launch_wgrad(const m2m_tower* t, int B, unsigned int seed, unsigned int step, const unsigned int* step_dev, hipStream_t st) {
    const int SPW = BM / t->N;
    const int ntiles = (B + SPW - 1) / SPW;
    const int nsl = ((t->Cp >> 5) + 3) / 4;
    // Row groups trade parallelism against float-atomic traffic (every group re-adds the whole slice):
    // aim at ~192 workgroups per launch, at least 4 tiles per group.
    int groups = (192 + (nsl * t->nblocks) / 2) / (nsl * t->nblocks);
    if (groups < 1) groups = 1;
    int tpg = (ntiles + groups - 1) / groups;
    if (tpg < 4) tpg = 4;
    if (tpg > ntiles) tpg = ntiles;
    groups = (ntiles + tpg - 1) / tpg;
    const size_t lds = (size_t)4 * BM * D * Prec<P>::ESZ;
    auto kern = tower_wgrad_kernel<P, D>;
    static bool attr_done = false;
    if (!attr_done) {
        M2M_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(nsl, t->nblocks, groups), dim3(NTHREADS), lds, st, *t, ntiles, tpg, SPW * t->N, seed, step, step_dev);
    M2M_CHECK_HIP(hipGetLastError());
    return 0;
}

int m2m_check_tower(const m2m_tower* t, int B);

extern "C" int m2m_tower_wgrad(const m2m_tower* t, int B, uint32_t seed, uint32_t step, const uint32_t* step_dev, void* stream) {
    if (int rc = m2m_check_tower(t, B)) return rc;
    if (t->nblocks == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define M2M_WG_CASE(PP, DD) if (t->prec == PP && t->D == DD) return launch_wgrad<PP, DD>(t, B, seed, step, step_dev, st);
    M2M_WG_CASE(PREC_BF16, 32) M2M_WG_CASE(PREC_BF16, 64) M2M_WG_CASE(PREC_BF16, 128)
    M2M_WG_CASE(PREC_F32, 32) M2M_WG_CASE(PREC_F32, 64) M2M_WG_CASE(PREC_F32, 128)
#undef M2M_WG_CASE
    m2m_set_error("tower_wgrad: unsupported (prec, D)", __FILE__, __LINE__);
    return -1;
}
