// ubench5.hip — does the bf16 MFMA pipe run beside the scan's VALU tail (pk_fma, v_sqrt, v_bfi, v_min3)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// KIND 0: NM mfma only; 1: NM mfma + tail for 16 results; 2: tail only
template <int KIND, int NM>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = seed * j;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + j); }
    float q[16], best = 1e30f;
    for (int j = 0; j < 16; ++j) q[j] = seed + j + threadIdx.x * 0.001f;
    unsigned ub = 0x7f800000u;
    for (int it = 0; it < iters; ++it) {
        if (KIND != 2) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        }
        if (KIND != 0) {
            // the per-tile VALU tail on 16 (b, q) results: disc = b*b - q (packed), sqrt, copysign, subtract, min
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                f2 bb = {q[j], q[j + 1]}, qq = {q[j + 1], q[j]};
                f2 disc = __builtin_elementwise_fma(bb, bb, -qq);
                float s0 = __builtin_amdgcn_sqrtf(disc.x), s1 = __builtin_amdgcn_sqrtf(disc.y);
                f2 sx = {__builtin_copysignf(s0, qq.x), __builtin_copysignf(s1, qq.y)};
                f2 t = bb - sx;
                unsigned k0 = __float_as_uint(t.x), k1 = __float_as_uint(t.y);
                ub = min(ub, min(k0, k1));
                asm volatile("" : "+v"(q[j]), "+v"(q[j + 1]));
            }
        }
    }
    float s = __uint_as_float(ub) + best;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int NM>
void run(const char* name, int w, float* dout) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    hipLaunchKernelGGL((k<KIND, NM>), dim3(256 * w), dim3(256), 0, 0, dout, 100, 1.0f);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, NM>), dim3(256 * w), dim3(256), 0, 0, dout, iters, 1.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s waves/SIMD=%d ms=%7.3f cycles(2.4GHz) per iteration per SIMD = %7.1f\n", name, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * w));
}
int main() {
    float* dout; CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 2, 4}) {
        run<0, 4>("4 x mfma_32x32x16_bf16", w, dout);
        run<2, 0>("tail on 16 results", w, dout);
        run<1, 4>("4 mfma + tail", w, dout);
        run<0, 8>("8 x mfma", w, dout);
        run<1, 8>("8 mfma + tail", w, dout);
    }
    return 0;
}
