// ubench3.hip — cycles per sphere test (wave64) for candidate codings of the scan loop, VALU only
// (sphere data already in registers; no memory in the loop).  4 waves per SIMD, 256 CUs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

struct Sph {
    float x[8], y[8], z[8], w[8];
};

// V1: scalar (non-packed) ops, sphere in SGPRs
__device__ __forceinline__ unsigned test_v1(float cx, float cy, float cz, float cw, float ox, float oy, float oz,
                                            float dx, float dy, float dz, float nod) {
    float ocx, ocy, ocz, b, q, t;
    asm volatile(
        "v_sub_f32 %0, %6, %10\n"
        "v_sub_f32 %1, %7, %11\n"
        "v_sub_f32 %2, %8, %12\n"
        "v_fma_f32 %3, %6, %13, %16\n"
        "v_fma_f32 %3, %7, %14, %3\n"
        "v_fma_f32 %3, %8, %15, %3\n"
        "v_fma_f32 %4, %0, %0, %9\n"
        "v_fma_f32 %4, %1, %1, %4\n"
        "v_fma_f32 %4, %2, %2, %4\n"
        "v_fma_f32 %5, %3, %3, -%4\n"
        "v_sqrt_f32 %5, %5\n"
        "v_bfi_b32 %5, %17, %5, %4\n"
        "v_sub_f32 %5, %3, %5\n"
        : "=&v"(ocx), "=&v"(ocy), "=&v"(ocz), "=&v"(b), "=&v"(q), "=&v"(t)
        : "s"(cx), "s"(cy), "s"(cz), "s"(cw), "v"(ox), "v"(oy), "v"(oz), "v"(dx), "v"(dy), "v"(dz), "v"(nod), "s"(0x7fffffff));
    return __float_as_uint(t);
}
// V2: scalar ops, sphere in VGPRs (what an LDS broadcast gives)
__device__ __forceinline__ unsigned test_v2(float cx, float cy, float cz, float cw, float ox, float oy, float oz,
                                            float dx, float dy, float dz, float nod) {
    float ocx, ocy, ocz, b, q, t;
    asm volatile(
        "v_sub_f32 %0, %6, %10\n"
        "v_sub_f32 %1, %7, %11\n"
        "v_sub_f32 %2, %8, %12\n"
        "v_fma_f32 %3, %6, %13, %16\n"
        "v_fma_f32 %3, %7, %14, %3\n"
        "v_fma_f32 %3, %8, %15, %3\n"
        "v_fma_f32 %4, %0, %0, %9\n"
        "v_fma_f32 %4, %1, %1, %4\n"
        "v_fma_f32 %4, %2, %2, %4\n"
        "v_fma_f32 %5, %3, %3, -%4\n"
        "v_sqrt_f32 %5, %5\n"
        "v_bfi_b32 %5, %17, %5, %4\n"
        "v_sub_f32 %5, %3, %5\n"
        : "=&v"(ocx), "=&v"(ocy), "=&v"(ocz), "=&v"(b), "=&v"(q), "=&v"(t)
        : "v"(cx), "v"(cy), "v"(cz), "v"(cw), "v"(ox), "v"(oy), "v"(oz), "v"(dx), "v"(dy), "v"(dz), "v"(nod), "s"(0x7fffffff));
    return __float_as_uint(t);
}
// V5: as V1 without sqrt/bfi/sub (how much is the tail?)
__device__ __forceinline__ unsigned test_v5(float cx, float cy, float cz, float cw, float ox, float oy, float oz,
                                            float dx, float dy, float dz, float nod) {
    float ocx, ocy, ocz, b, q, t;
    asm volatile(
        "v_sub_f32 %0, %6, %10\n"
        "v_sub_f32 %1, %7, %11\n"
        "v_sub_f32 %2, %8, %12\n"
        "v_fma_f32 %3, %6, %13, %16\n"
        "v_fma_f32 %3, %7, %14, %3\n"
        "v_fma_f32 %3, %8, %15, %3\n"
        "v_fma_f32 %4, %0, %0, %9\n"
        "v_fma_f32 %4, %1, %1, %4\n"
        "v_fma_f32 %4, %2, %2, %4\n"
        "v_fma_f32 %5, %3, %3, -%4\n"
        : "=&v"(ocx), "=&v"(ocy), "=&v"(ocz), "=&v"(b), "=&v"(q), "=&v"(t)
        : "s"(cx), "s"(cy), "s"(cz), "s"(cw), "v"(ox), "v"(oy), "v"(oz), "v"(dx), "v"(dy), "v"(dz), "v"(nod), "s"(0x7fffffff));
    return __float_as_uint(t);
}
// V3: two rays per lane, packed math written with 2-vectors (compiler emits v_pk_*; the sphere scalar is
// splatted into an SGPR pair outside the loop)
__device__ __forceinline__ void test_v3(float cx, float cy, float cz, float cw, f2 ox, f2 oy, f2 oz, f2 dx, f2 dy,
                                        f2 dz, f2 nod, unsigned& ta, unsigned& tb) {
    const f2 X = {cx, cx}, Y = {cy, cy}, Z = {cz, cz}, W = {cw, cw};
    const f2 ocx = X - ox, ocy = Y - oy, ocz = Z - oz;
    const f2 b = __builtin_elementwise_fma(Z, dz, __builtin_elementwise_fma(Y, dy, __builtin_elementwise_fma(X, dx, nod)));
    const f2 q = __builtin_elementwise_fma(ocz, ocz, __builtin_elementwise_fma(ocy, ocy, __builtin_elementwise_fma(ocx, ocx, W)));
    const f2 disc = __builtin_elementwise_fma(b, b, -q);
    const float s0 = __builtin_amdgcn_sqrtf(disc.x), s1 = __builtin_amdgcn_sqrtf(disc.y);
    ta = __float_as_uint(b.x - __builtin_copysignf(s0, q.x));
    tb = __float_as_uint(b.y - __builtin_copysignf(s1, q.y));
}

template <int V>
__global__ __launch_bounds__(256) void k(const Sph sp, float* out, int iters, float seed) {
    const float fx = seed + threadIdx.x * 0.01f;
    float ox = fx, oy = fx * 0.5f, oz = fx * 0.25f, dx = 0.6f, dy = 0.48f, dz = 0.64f, nod = -fx;
    unsigned best = 0x7f800000u, best2 = 0x7f800000u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            unsigned ta, tb;
            if (V == 1) {
                ta = test_v1(sp.x[j], sp.y[j], sp.z[j], sp.w[j], ox, oy, oz, dx, dy, dz, nod);
                tb = test_v1(sp.x[j + 1], sp.y[j + 1], sp.z[j + 1], sp.w[j + 1], ox, oy, oz, dx, dy, dz, nod);
            } else if (V == 5) {
                ta = test_v5(sp.x[j], sp.y[j], sp.z[j], sp.w[j], ox, oy, oz, dx, dy, dz, nod);
                tb = test_v5(sp.x[j + 1], sp.y[j + 1], sp.z[j + 1], sp.w[j + 1], ox, oy, oz, dx, dy, dz, nod);
            } else {
                // force the sphere into VGPRs (v_mov hoisted out of the loop by the compiler)
                float vx = sp.x[j], vy = sp.y[j], vz = sp.z[j], vw = sp.w[j];
                float ux = sp.x[j + 1], uy = sp.y[j + 1], uz = sp.z[j + 1], uw = sp.w[j + 1];
                asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz), "+v"(vw), "+v"(ux), "+v"(uy), "+v"(uz), "+v"(uw));
                ta = test_v2(vx, vy, vz, vw, ox, oy, oz, dx, dy, dz, nod);
                tb = test_v2(ux, uy, uz, uw, ox, oy, oz, dx, dy, dz, nod);
            }
            best = min(best, min(ta, tb));
        }
        ox += 1e-6f;  // keep the loop body live
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(best ^ best2);
}

// two rays per lane
__global__ __launch_bounds__(256) void k3(const Sph sp, float* out, int iters, float seed) {
    const float fx = seed + threadIdx.x * 0.01f;
    f2 ox = {fx, fx + 1}, oy = {fx * 0.5f, fx}, oz = {fx * 0.25f, fx * 2}, dx = {0.6f, 0.1f}, dy = {0.48f, 0.2f},
       dz = {0.64f, 0.3f}, nod = {-fx, fx};
    unsigned besta = 0x7f800000u, bestb = 0x7f800000u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned ta, tb;
            test_v3(sp.x[j], sp.y[j], sp.z[j], sp.w[j], ox, oy, oz, dx, dy, dz, nod, ta, tb);
            besta = min(besta, ta);
            bestb = min(bestb, tb);
        }
        ox.x += 1e-6f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(besta ^ bestb);
}

template <class F>
void run(const char* name, int tests_per_iter, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    launch(1000);
    CHECK(hipEventRecord(e0));
    launch(iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int w = 4;
    double tests = (double)iters * tests_per_iter * w;  // wave-tests (x rays per lane) per SIMD
    printf("%-44s ms=%8.3f  cycles(2.4GHz) per wave64 sphere-ray-test = %6.2f\n", name, ms, ms * 1e-3 * 2.4e9 / tests);
    fflush(stdout);
}

int main() {
    float* dout;
    CHECK(hipMalloc(&dout, 256 * 4 * 256 * sizeof(float)));
    Sph sp;
    for (int i = 0; i < 8; ++i) { sp.x[i] = i * 1.5f; sp.y[i] = 0.2f; sp.z[i] = -i * 0.7f; sp.w[i] = -0.04f; }
    dim3 g(256 * 4), b(256);
    run("V1 scalar ops, sphere in SGPRs", 8, [&](int it) { hipLaunchKernelGGL(k<1>, g, b, 0, 0, sp, dout, it, 1.0f); });
    run("V2 scalar ops, sphere in VGPRs", 8, [&](int it) { hipLaunchKernelGGL(k<2>, g, b, 0, 0, sp, dout, it, 1.0f); });
    run("V5 = V1 without sqrt/bfi/sub", 8, [&](int it) { hipLaunchKernelGGL(k<5>, g, b, 0, 0, sp, dout, it, 1.0f); });
    run("V3 packed, 2 rays/lane, sphere in SGPRs", 16, [&](int it) { hipLaunchKernelGGL(k3, g, b, 0, 0, sp, dout, it, 1.0f); });
    return 0;
}
