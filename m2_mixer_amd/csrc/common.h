// Device-side building blocks shared by every kernel of libm2mixer (gfx950 / CDNA4 only).
//
//  * Prec<P>: the arithmetic a kernel is instantiated for.
//      P = 0  bf16 operands, fp32 accumulate  (v_mfma_f32_16x16x32_bf16)
//      P = 1  fp32 operands, fp32 accumulate  (v_mfma_f32_16x16x4_f32, exact fp32; the parity mode)
//  * "packed block": 16 (row-or-column index i) x KB (contraction index k) operand elements stored as
//    64 lanes x 16 bytes, lane-major, so that ONE 16-byte access per lane (global_load_dwordx4 or
//    ds_read_b128, both fully contiguous over the wave) yields ready-to-issue MFMA operand registers.
//    Two k orders exist:
//      NAT  natural: the order the MFMA defines           (bf16: k = 8g + e   ; fp32 step e: k = 4e + g)
//      CHN  chained: the order in which a previous MFMA's accumulator rows sit in a lane's registers,
//           so an accumulator tile becomes the next MFMA's operand with no lane movement
//                                                         (bf16: k = 16(e>>2) + 4g + (e&3) ; fp32: k = 4g + e)
//    with g = lane >> 4, i = lane & 15, e = element index inside the lane's 16 bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define M2M_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

enum { PREC_BF16 = 0, PREC_F32 = 1 };
enum { PACK_NAT = 0, PACK_CHN = 1 };

// 16-byte operand fragment, viewed either way.
union Frag {
    u32x4_t u;
    f32x4_t f;
    bf16x8_t h;
};

template <int P> struct Prec;

template <> struct Prec<PREC_BF16> {
    typedef __bf16 elem_t;
    static constexpr int KB = 32;        // k extent of one packed block
    static constexpr int ESZ = 2;
    static constexpr int EPL = 8;        // elements per lane in a block
    // acc(16x16) += A(16 x 32) * B(32 x 16); a supplies rows, b supplies columns
    static __device__ __forceinline__ void mma(f32x4_t& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ int kmap(int mode, int g, int e) {
        return mode == PACK_NAT ? 8 * g + e : 16 * (e >> 2) + 4 * g + (e & 3);
    }
};

template <> struct Prec<PREC_F32> {
    typedef float elem_t;
    static constexpr int KB = 16;
    static constexpr int ESZ = 4;
    static constexpr int EPL = 4;
    static __device__ __forceinline__ void mma(f32x4_t& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[0], b.f[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[1], b.f[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[2], b.f[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[3], b.f[3], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ int kmap(int mode, int g, int e) {
        return mode == PACK_NAT ? 4 * e + g : 4 * g + e;
    }
};

// ---- fragment <-> memory -------------------------------------------------------------------------
static __device__ __forceinline__ Frag ld_frag_global(const void* base, long block, int lane) {
    Frag f;
    f.u = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(base) + block * 1024 + lane * 16);
    return f;
}
// Global-memory pointer type for values that pass through an opaque asm (which strips the address-space inference: a plain
// pointer would come back as FLAT, and flat loads also tick lgkmcnt).
#if defined(__HIP_DEVICE_COMPILE__)
#define M2M_AS1 __attribute__((address_space(1)))
#else
#define M2M_AS1                                   /* the host pass only parses device code */
#endif
typedef const M2M_AS1 char* gptr_t;
typedef M2M_AS1 char* gptr_w_t;
// (the value is wave-uniform by construction -- a descriptor field -- and is made so for the compiler too: readfirstlane of
//  both halves, free when it already sits in SGPRs)
static __device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
static __device__ __forceinline__ gptr_t to_gptr(const void* p) { return (gptr_t)uniform_u64((unsigned long long)p); }
static __device__ __forceinline__ gptr_w_t to_gptr_w(void* p) { return (gptr_w_t)uniform_u64((unsigned long long)p); }
// fragment load with a wave-uniform block index: scalar base + one 32-bit per-lane offset shared by every load of the loop
static __device__ __forceinline__ Frag ld_frag_global_u(gptr_t base, long block_uniform, unsigned int lane16) {
    Frag f;
    f.u = *reinterpret_cast<const M2M_AS1 u32x4_t*>(base + block_uniform * 1024 + lane16);
    return f;
}
// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to the wave-uniform LDS byte address ldst (+ lane x
// 16).  Inline asm on purpose: outside hipcc's memory-counter bookkeeping, so that completion can be counted by hand (counted
// s_waitcnt vmcnt(N) + barrier) and several tiles stay in flight.  M0 (the DMA's LDS base) is saved and restored.
static __device__ __forceinline__ void glds16_g(gptr_t gsrc, unsigned int ldst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
}
// LDS byte address of a pointer into the dynamic shared array
static __device__ __forceinline__ unsigned int lds_addr_of(const void* p) {
    return (unsigned int)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
static __device__ __forceinline__ Frag ld_frag_lds(const char* base, int block, int lane) {
    Frag f;
    f.u = *reinterpret_cast<const u32x4_t*>(base + block * 1024 + lane * 16);
    return f;
}

static __device__ __forceinline__ unsigned short f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
}
// two floats -> one register of two bf16 (lo in bits 0-15): as a two-element vector conversion this is ONE v_cvt_pk_bf16_f32;
// written as two scalar casts combined with shift / or, hipcc often converted each value alone (cvt_pk with a dummy second
// source) and merged them with v_lshlrev + v_or_b32_sdwa: 4 instructions per pair in the hot loops
static __device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
static __device__ __forceinline__ float bf2f(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned int)b << 16);
}

// ---- math -----------------------------------------------------------------------------------------
// erf(x) as an odd rational polynomial on [-4, 4] (|err| < 5e-7, checked against scipy in
// tests/test_host_cpu.py::test_erf_rational_coefficients); no exp, one reciprocal.  Coefficients: the float erf rational used by Eigen.
static __device__ __forceinline__ float erf_fast(float x) {
    x = __builtin_fminf(__builtin_fmaxf(x, -4.0f), 4.0f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = __builtin_fmaf(x2, p, 2.77068142495902e-08f);
    p = __builtin_fmaf(x2, p, -2.10102402082508e-06f);
    p = __builtin_fmaf(x2, p, -5.69250639462346e-05f);
    p = __builtin_fmaf(x2, p, -7.34990630326855e-04f);
    p = __builtin_fmaf(x2, p, -2.95459980854025e-03f);
    p = __builtin_fmaf(x2, p, -1.60960333262415e-02f);
    p = x * p;
    float q = -1.45660718464996e-05f;
    q = __builtin_fmaf(x2, q, -2.13374055278905e-04f);
    q = __builtin_fmaf(x2, q, -1.68282697438203e-03f);
    q = __builtin_fmaf(x2, q, -7.37332916720468e-03f);
    q = __builtin_fmaf(x2, q, -1.42647390514189e-02f);
    return p * __builtin_amdgcn_rcpf(q);
}

static __device__ __forceinline__ float gelu_f(float x) {
    return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f));
}
// gelu(x) and d gelu / dx = Phi(x) + x phi(x)
static __device__ __forceinline__ void gelu_grad_f(float x, float& g, float& dg) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    g = x * cdf;
    dg = __builtin_fmaf(x, pdf, cdf);
}

// ---- activation dispatch ---------------------------------------------------------------------------------
// bf16 mode: gelu(x) ~ a_i + b_i x and gelu'(x) ~ c_i + d_i x on cell i of 512 cells over [-6, 6) (|err| < 6e-5 / 9e-5, an
// order of magnitude below bf16 resolution); cell 0 is x < -6 (0, 0 | 0, 0), cell 513 is x >= 6 (0, 1 | 1, 0).  One fma
// builds the index, one fma per function evaluates it, and the dropout scale 1 / (1 - p) of the launch is folded into the
// table: ~6 VALU + one ds_read per element for gelu, ~7 for gelu and gelu' (the interpolating table this replaced: 11 / 15;
// the rational erf + exp: ~35).  fp32 (parity) mode: the accurate functions above.
#define PWL_N 512
#define PWL_XMAX 6.0f
#define GELU_TAB_N (PWL_N + 2)                                 // table entries (16 bytes each)
typedef __attribute__((ext_vector_type(4))) float gtab_t;     // {a, b, c, d}
static __device__ __forceinline__ unsigned int pwl_index(float x) {
    const float t = __builtin_fmaf(x, PWL_N / (2.0f * PWL_XMAX), 0.5f * PWL_N + 1.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cvt_u32_f32 saturates (t < 0 and NaN -> 0): written as the instruction, the C cast needs a v_max_f32 in front of it to
    // be defined for negative t -- one VALU instruction per hidden element in loops that are VALU-issue-bound
    unsigned int i;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(i) : "v"(t));
#else
    const unsigned int i = (unsigned int)__builtin_fmaxf(t, 0.0f);
#endif
    return i < (unsigned int)(PWL_N + 1) ? i : (unsigned int)(PWL_N + 1);
}
static __device__ __forceinline__ gtab_t pwl_cell(int i, float scale) {
    if (i <= 0) return gtab_t{0.f, 0.f, 0.f, 0.f};
    if (i >= PWL_N + 1) return gtab_t{0.f, scale, scale, 0.f};
    const float h = 2.0f * PWL_XMAX / PWL_N;
    const float x0 = -PWL_XMAX + h * (i - 1), x1 = x0 + h;
    float g0, d0, g1, d1;
    gelu_grad_f(x0, g0, d0);
    gelu_grad_f(x1, g1, d1);
    const float b = (g1 - g0) / h, d = (d1 - d0) / h;
    return gtab_t{(g0 - b * x0) * scale, b * scale, (d0 - d * x0) * scale, d * scale};
}
static __device__ __forceinline__ void gelu_tab_fill(gtab_t* tab, float scale, int tid, int nthreads) {
    for (int k = tid; k < GELU_TAB_N; k += nthreads) tab[k] = pwl_cell(k, scale);
}
// gelu(x) * scale and gelu'(x) * scale, `scale` being the value the table was filled with (bf16) / applied here (fp32)
// Closed form (ActB below): Phi(x) through the Abramowitz-Stegun 7.1.26 erfc polynomial (|err| <
// 1e-7), one v_exp + one v_rcp + ~13 plain VALU for gelu AND gelu' (they share exp(-x^2/2)); no LDS traffic: the table costs
// one ds_read_b128 at a random address per element, ~16 LDS cycles per wave-instruction with its bank conflicts.
static __device__ __forceinline__ void gelu_grad_as(float x, float& g, float& dg) {
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);       // exp(-x^2 / 2)
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x), 0.23164189213f, 1.0f));   // 1 / (1 + p |x| / sqrt 2)
    float p = 0.5f * 1.061405429f;
    p = __builtin_fmaf(p, t, 0.5f * -1.453152027f);
    p = __builtin_fmaf(p, t, 0.5f * 1.421413741f);
    p = __builtin_fmaf(p, t, 0.5f * -0.284496736f);
    p = __builtin_fmaf(p, t, 0.5f * 0.254829592f);
    const float q = p * t * e;                                                      // Phi(-|x|)
    const float cdf = x >= 0.f ? 1.0f - q : q;
    g = x * cdf;
    dg = __builtin_fmaf(x * e, 0.3989422804014327f, cdf);
}
// Forward-only table: {a, b} per cell (8 bytes).  The forward never needs gelu', and an 8-byte read at a random address
// conflicts far less than a 16-byte one (two 32-lane groups over 64 banks instead of four 16-lane groups of four banks each).
typedef __attribute__((ext_vector_type(2))) float gtab2_t;
static __device__ __forceinline__ void gelu_tab2_fill(gtab2_t* tab, float scale, int tid, int nthreads) {
    for (int k = tid; k < GELU_TAB_N; k += nthreads) { const gtab_t c = pwl_cell(k, scale); tab[k] = gtab2_t{c[0], c[1]}; }
}
// Backward table: {a, b, c, d} as four fp16 values in 8 bytes.  The backward needs gelu AND gelu' of every hidden element; the
// 16-byte fp32 entry costs a ds_read_b128 at a random address (four 16-lane groups of four banks each: ~3 LDS cycles per
// group with its conflicts), the 8-byte entry a ds_read_b64 (two 32-lane groups of two banks).  fp16 coefficients carry a
// relative error of 2^-11 = 4.9e-4, an eighth of bf16 resolution -- the values are rounded to bf16 right after (the operand of
// the next product / of the weight gradients); both fmas take the fp16 operands directly (v_fma_mix_f32).  Coefficients are
// O(1) x the dropout scale, far inside fp16 range.
typedef _Float16 gtabh_t __attribute__((ext_vector_type(4)));       // {a, b, c, d}
static __device__ __forceinline__ void gelu_tabh_fill(gtabh_t* tab, float scale, int tid, int nthreads) {
    for (int k = tid; k < GELU_TAB_N; k += nthreads) {
        const gtab_t c = pwl_cell(k, scale);
        tab[k] = gtabh_t{(_Float16)c[0], (_Float16)c[1], (_Float16)c[2], (_Float16)c[3]};
    }
}
// keep-mask folded into the index: a dropped element (mk == 0) reads cell 0 = {0, 0, 0, 0}, so gelu and gelu' both come out as
// 0 with ONE v_and instead of one per value
static __device__ __forceinline__ void gelu_grad_tabh_masked(const gtabh_t* tab, float x, unsigned int mk, float& g, float& dg) {
    const gtabh_t e = tab[pwl_index(x) & mk];
    g = __builtin_fmaf((float)e[1], x, (float)e[0]);
    dg = __builtin_fmaf((float)e[3], x, (float)e[2]);
}
static __device__ __forceinline__ void gelu_grad_tabh(const gtabh_t* tab, float x, float& g, float& dg) {
    const gtabh_t e = tab[pwl_index(x)];
    g = __builtin_fmaf((float)e[1], x, (float)e[0]);
    dg = __builtin_fmaf((float)e[3], x, (float)e[2]);
}
template <int P> struct Act;
template <> struct Act<PREC_BF16> {
    static constexpr bool USES_TABLE = true;
    static __device__ __forceinline__ float gelu_scaled(const gtab2_t* tab, float x, float) {
        const gtab2_t e = tab[pwl_index(x)];
        return __builtin_fmaf(e[1], x, e[0]);
    }
    static __device__ __forceinline__ float gelu_scaled(const gtab_t* tab, float x, float) {
        const gtab_t e = tab[pwl_index(x)];
        return __builtin_fmaf(e[1], x, e[0]);
    }
    static __device__ __forceinline__ void gelu_grad_scaled(const gtab_t* tab, float x, float, float& g, float& dg) {
        const gtab_t e = tab[pwl_index(x)];
        g = __builtin_fmaf(e[1], x, e[0]);
        dg = __builtin_fmaf(e[3], x, e[2]);
    }
    static __device__ __forceinline__ void gelu_grad_scaled(const gtabh_t* tab, float x, float, float& g, float& dg) { gelu_grad_tabh(tab, x, g, dg); }
};
template <> struct Act<PREC_F32> {
    static constexpr bool USES_TABLE = false;
    static __device__ __forceinline__ float gelu_scaled(const gtab_t*, float x, float scale) { return gelu_f(x) * scale; }
    static __device__ __forceinline__ float gelu_scaled(const gtab2_t*, float x, float scale) { return gelu_f(x) * scale; }
    static __device__ __forceinline__ void gelu_grad_scaled(const gtab_t*, float x, float scale, float& g, float& dg) {
        gelu_grad_f(x, g, dg);
        g *= scale;
        dg *= scale;
    }
    static __device__ __forceinline__ void gelu_grad_scaled(const gtabh_t*, float x, float scale, float& g, float& dg) {
        gelu_grad_scaled(static_cast<const gtab_t*>(nullptr), x, scale, g, dg);
    }
};
// The activation as the BACKWARD kernels evaluate it: the table too (M2M_BWD_FORMULA=1 selects the closed form).  Measured in
// one process both ways: while the backward column loop still carried ~70 wasted packing instructions per step the closed form
// won by 3-4 % (less LDS traffic); once those were gone the loop was VALU-issue-bound and the table's fewer instructions win
// by 2 % on the step (932k vs 915k samples/s).  The forward chains always preferred the table (closed form: +4 % time).
#ifndef M2M_BWD_FORMULA
#define M2M_BWD_FORMULA 0
#endif
template <int P> struct ActB : Act<P> {};
#if M2M_BWD_FORMULA
template <> struct ActB<PREC_BF16> {
    static constexpr bool USES_TABLE = false;
    static __device__ __forceinline__ void gelu_grad_scaled(const gtab_t*, float x, float scale, float& g, float& dg) {
        gelu_grad_as(x, g, dg);
        g *= scale;
        dg *= scale;
    }
    static __device__ __forceinline__ void gelu_grad_scaled(const gtabh_t*, float x, float scale, float& g, float& dg) {
        gelu_grad_scaled(static_cast<const gtab_t*>(nullptr), x, scale, g, dg);
    }
};
#endif
// Table type of the backward chain kernel (tower_bwd.hip, token_mfma.h's backward): the packed fp16 table (M2M_BWD_HTAB=0: fp32).
#ifndef M2M_BWD_HTAB
#define M2M_BWD_HTAB 1
#endif
#if M2M_BWD_HTAB
typedef gtabh_t gtabB_t;
static __device__ __forceinline__ void gelu_tabB_fill(gtabB_t* tab, float scale, int tid, int nthreads) { gelu_tabh_fill(tab, scale, tid, nthreads); }
#else
typedef gtab_t gtabB_t;
static __device__ __forceinline__ void gelu_tabB_fill(gtabB_t* tab, float scale, int tid, int nthreads) { gelu_tab_fill(tab, scale, tid, nthreads); }
#endif
// The activation inside the token-mixing MFMA phases (token_mfma.h): 1 = closed form, 0 = table.  M2M_TOK_FORMULA bit 0: forward,
// bit 1: backward.
#ifndef M2M_TOK_FORMULA
#define M2M_TOK_FORMULA 0
#endif
struct ActTokF {
    static __device__ __forceinline__ float gelu_scaled(const gtab2_t* tab, float x, float scale) {
        if (M2M_TOK_FORMULA & 1) { float g, dg; gelu_grad_as(x, g, dg); return g * scale; }
        return Act<PREC_BF16>::gelu_scaled(tab, x, scale);
    }
};
struct ActTokB {
    template <class TAB>
    static __device__ __forceinline__ void gelu_grad_scaled(const TAB* tab, float x, float scale, float& g, float& dg) {
        if (M2M_TOK_FORMULA & 2) { gelu_grad_as(x, g, dg); g *= scale; dg *= scale; }
        else ActB<PREC_BF16>::gelu_grad_scaled(tab, x, scale, g, dg);
    }
};
// 0 / ~0 from bit k of w (v_bfe_i32): keep-masks are applied with one AND
static __device__ __forceinline__ unsigned int bit_to_mask(unsigned int w, int k) {
    return (unsigned int)(((int)(w << (31 - k))) >> 31);
}
static __device__ __forceinline__ float mask_f(float v, unsigned int m) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, v) & m);
}

// ---- dropout: counter-based, stateless ---------------------------------------------------------------
// keep(element) = 16 bits of mix32(key ^ word) < thr16, two elements per 32-bit word.  The same
// function is evaluated by forward, backward and the weight-gradient pass, so no mask is ever stored.
// key = site key derived on the host (m2m_dropout_key); thr16 = round((1-p) * 65536).
static __device__ __forceinline__ unsigned int mix32(unsigned int x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
struct Drop {
    unsigned int key;
    unsigned int thr;   // keep iff 16-bit draw < thr ; thr == 65536 keeps everything
    float scale;        // 65536 / thr
};
static __device__ __forceinline__ bool drop_keep(const Drop& d, unsigned int idx) {
    const unsigned int w = mix32(d.key ^ (idx >> 1));
    const unsigned int r = (idx & 1) ? (w >> 16) : (w & 0xFFFFu);
    return r < d.thr;
}
// ---- wave helpers -----------------------------------------------------------------------------------
// Sum over groups of `width` adjacent lanes (width in {4, 8, 16}: DPP, stays in the VALU; 32, 64: the last
// steps go through ds_bpermute).  Every lane of a group ends up with the group's sum.
template <int CTRL>
static __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
static __device__ __forceinline__ float wave_sum_xor(float v, int width) {
    v += dpp_mov<0xB1>(v);                    // quad_perm [1,0,3,2]   : lane ^ 1
    v += dpp_mov<0x4E>(v);                    // quad_perm [2,3,0,1]   : lane ^ 2
    if (width >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror       : pairs the two quads of 8 lanes
    if (width >= 16) v += dpp_mov<0x140>(v);  // row_mirror            : pairs the two halves of 16 lanes
    if (width >= 32) v += __shfl_xor(v, 16, 64);
    if (width >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// Sum over the lanes that agree in (lane % stride), stride in {8, 16, 32}; every lane gets its class's sum.
// All steps are VALU cross-lane operations (DPP row rotate, v_permlane16_swap, v_permlane32_swap): no LDS.
static __device__ __forceinline__ float lane_class_sum(float v, int stride) {
    if (stride <= 8) v += dpp_mov<0x128>(v);                       // row_ror:8  : lane ^ 8 within each row of 16
    if (stride <= 16) {
        const unsigned int u = __builtin_bit_cast(unsigned int, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned int)r[0]) + __builtin_bit_cast(float, (unsigned int)r[1]);
    }
    {
        const unsigned int u = __builtin_bit_cast(unsigned int, v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned int)r[0]) + __builtin_bit_cast(float, (unsigned int)r[1]);
    }
    return v;
}

#define M2M_CHECK_HIP(expr)                                                      \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) { m2m_set_error(hipGetErrorString(_e), __FILE__, __LINE__); return -2; } \
    } while (0)

void m2m_set_error(const char* msg, const char* file, int line);

// ---- optional phase timers (diagnostic build only: make TIMERS=1 -> libm2mixer_timers.so) -------------
// Workgroup 0 / thread 0 accumulates the 100 MHz wall clock spent between consecutive marks of a launch.
#ifdef M2M_TIMERS
#define TIMER_DECL(sym) static __device__ unsigned long long sym[32]
#define TIMER_START() unsigned long long _tm_last = __builtin_amdgcn_s_memrealtime()
// The mark is taken by the WHOLE first wave of workgroup 0 behind a wave-uniform (scalar) branch, lanes other than 0 adding
// zero: an exec-masked single-lane read-modify-write here was miscompiled once the kernels grew (the store's zero offset
// register was materialised under a different exec mask and lane 0 wrote through garbage: "write access to a read-only page").
#define TIMER_MARK(sym, i)                                                          \
    do {                                                                            \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 &&                \
            __builtin_amdgcn_readfirstlane((int)threadIdx.x) == 0) {                \
            const unsigned long long _n = __builtin_amdgcn_s_memrealtime();         \
            atomicAdd(&sym[i], (threadIdx.x & 63) == 0 ? _n - _tm_last : 0ULL);     \
            _tm_last = _n;                                                          \
        }                                                                           \
    } while (0)
// Cheap flavour for kernels with many marks (the tower chains): thread 0 of workgroup 0 accumulates into LDS, one flush of
// the 32 slots at the end of the kernel.  (A mark of the flavour above costs ~2.5 us: its atomic is waited for by the next
// vmcnt(0) of the wave.)
#define TIMER_LSTART()                                                              \
    __shared__ unsigned long long _tm_lds[32];                                      \
    if (threadIdx.x < 32) _tm_lds[threadIdx.x] = 0ULL;                              \
    unsigned long long _tm_last = __builtin_amdgcn_s_memrealtime()
#define TIMER_LMARK(i)                                                              \
    do {                                                                            \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { \
            const unsigned long long _n = __builtin_amdgcn_s_memrealtime();         \
            _tm_lds[i] += _n - _tm_last;                                            \
            _tm_last = _n;                                                          \
        }                                                                           \
    } while (0)
#define TIMER_LFLUSH(sym)                                                           \
    do {                                                                            \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 32) \
            atomicAdd(&sym[threadIdx.x], _tm_lds[threadIdx.x]);                     \
    } while (0)
// The same in SHADER CYCLES (s_memtime) for marks inside a hot loop: TIMER_CRESET() in front of the loop, TIMER_CMARK(i) at the
// stage boundaries; slot i accumulates cycles (the reader's us conversion does not apply: scripts/bwd_loop_stamps.py).
#define TIMER_CRESET() unsigned long long _tm_lastc = __builtin_amdgcn_s_memtime()
#define TIMER_CMARK(i)                                                              \
    do {                                                                            \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { \
            const unsigned long long _n = __builtin_amdgcn_s_memtime();             \
            _tm_lds[i] += _n - _tm_lastc;                                           \
            _tm_lastc = _n;                                                         \
        }                                                                           \
    } while (0)
// every workgroup: [16] = earliest start, [17] = latest end (100 MHz wall clock), [18] = sum of workgroup durations, [19] = count,
// [20] = latest start, [21] = earliest end
#define TIMER_WG_BEGIN() const unsigned long long _tm_wg0 = __builtin_amdgcn_s_memrealtime()
#define TIMER_WG_END(sym)                                                           \
    do {                                                                            \
        if (threadIdx.x == 0) {                                                     \
            const unsigned long long _n = __builtin_amdgcn_s_memrealtime();         \
            atomicMin(&sym[16], _tm_wg0);                                           \
            atomicMax(&sym[17], _n);                                                \
            atomicMax(&sym[20], _tm_wg0);                                           \
            atomicMin(&sym[21], _n);                                                \
            atomicAdd(&sym[18], _n - _tm_wg0);                                      \
            atomicAdd(&sym[19], 1ULL);                                              \
        }                                                                           \
    } while (0)
#define TIMER_READER(name, sym)                                                     \
    extern "C" int name(unsigned long long* out, int reset) {                       \
        unsigned long long z[32] = {0};                                             \
        z[16] = ~0ULL; z[21] = ~0ULL;                                               \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sym), sizeof(z)) != hipSuccess) return -2; \
        if (reset && hipMemcpyToSymbol(HIP_SYMBOL(sym), z, sizeof(z)) != hipSuccess) return -2; \
        return 0;                                                                   \
    }
#else
#define TIMER_DECL(sym)
#define TIMER_START()
#define TIMER_MARK(sym, i)
#define TIMER_LSTART()
#define TIMER_LMARK(i)
#define TIMER_LFLUSH(sym)
#define TIMER_CRESET()
#define TIMER_CMARK(i)
#define TIMER_READER(name, sym)
#define TIMER_WG_BEGIN()
#define TIMER_WG_END(sym)
#endif
