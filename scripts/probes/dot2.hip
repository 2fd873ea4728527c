// probe: semantics of v_dot2c_f32_bf16 and lane_class_sum(.., 16) on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "../../m2_mixer_amd/csrc/common.h"
void m2m_set_error(const char*, const char*, int) {}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned int* a, float* o, float* o2) {
    unsigned int u = a[threadIdx.x];
    const bf16x2_t one2 = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
    float acc = 0.f;
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, u), one2, acc, false);
    o[threadIdx.x] = acc;
    o2[threadIdx.x] = lane_class_sum((float)threadIdx.x, 16);
}
int main() {
    unsigned int h[64]; float ref[64];
    for (int i = 0; i < 64; ++i) {
        float lo = 0.001f * (i + 1), hi = -0.0003f * (i + 3);
        unsigned int ul, uh; memcpy(&ul, &lo, 4); memcpy(&uh, &hi, 4);
        ul >>= 16; uh >>= 16;
        h[i] = ul | (uh << 16);
        unsigned int a = ul << 16, b = uh << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
        ref[i] = fa + fb;
    }
    unsigned int* d; float *o, *o2;
    hipMalloc(&d, 256); hipMalloc(&o, 256); hipMalloc(&o2, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, o2);
    float r[64], r2[64];
    hipMemcpy(r, o, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 256, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0;
    for (int i = 0; i < 64; ++i) {
        if (fabsf(r[i] - ref[i]) > 1e-7f) { if (bad < 4) printf("dot2 lane %d got %g want %g\n", i, r[i], ref[i]); ++bad; }
        float want = (float)((i % 16) * 4 + 16 + 32 + 48);
        if (r2[i] != want) { if (bad2 < 4) printf("class_sum lane %d got %g want %g\n", i, r2[i], want); ++bad2; }
    }
    printf("dot2 bad %d, class_sum bad %d\n", bad, bad2);
    return 0;
}
