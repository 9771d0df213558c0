// ubench2.hip — does the f32 MFMA pipe run beside the VALU on gfx950, and what do the remaining
// candidate scan instructions cost?  Same harness idea as ubench.hip: m workgroups of 256 threads per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 mfma only; 1 mfma + NV pk_fma; 2 NV pk_fma only; 3 mfma + NV v_sqrt; 4 NV sqrt only
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    f32x2 p[16];
    float q[16];
    for (int i = 0; i < 16; ++i) { p[i] = f32x2{seed + i, seed - i}; q[i] = seed + i + threadIdx.x; }
    const float a = seed + threadIdx.x, b = seed * 0.5f;
    f32x2 pb = {b, a}, pc = {a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0 || KIND == 1 || KIND == 3) {
                f32x16& acc = (r == 0) ? acc0 : (r == 1) ? acc1 : (r == 2) ? acc2 : acc3;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (KIND == 1 || KIND == 2) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 15]) : "v"(pb), "v"(pc));
            }
            if (KIND == 3 || KIND == 4) {
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(q[i & 15]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i] + p[i].x + p[i].y + q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// single-instruction rates: 0 v_fmac_f32 (VOP2), 1 v_fma_f32 with SGPR src, 2 v_mul_f32, 3 v_pk_mul_f32,
// 4 v_cndmask_b32, 5 v_cmp_lt_f32, 6 v_min_u32, 7 v_and_b32, 8 v_add_f32, 9 v_max_f32, 10 v_rsq_f32
template <int KIND>
__global__ __launch_bounds__(256) void k1(float* out, int iters, float seed) {
    float a[16];
    f32x2 p[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1.f}; }
    float b = seed * 0.5f, c = seed * 0.25f;
    f32x2 pb = {b, c};
    const float sc = seed * 3.f;  // uniform -> SGPR
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(sc));
                if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if (KIND == 5) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (KIND == 6) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 9) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 10) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 11) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 12) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(pb));
            }
        }
    }
    float s = b + c;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* dout;

template <class F>
double time_ms(F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(200);
    CHECK(hipEventRecord(e0));
    launch(4000);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int KIND, int NV>
void run(const char* name, int w) {
    double ms = time_ms([&](int it) { hipLaunchKernelGGL((k<KIND, NV>), dim3(256 * w), dim3(256), 0, 0, dout, it, 1.0f); });
    // per wave: 4000 iters x 4 groups; report ns per group per wave-slot and implied cycles at 2.4 GHz
    double groups = 4000.0 * 4 * w;  // groups issued per SIMD
    double cyc = ms * 1e-3 * 2.4e9 / groups;
    printf("%-34s NV=%2d waves/SIMD=%d  ms=%8.3f  cycles(2.4GHz)/group/SIMD=%7.1f\n", name, NV, w, ms, cyc);
    fflush(stdout);
}
template <int KIND>
void run1(const char* name, int w) {
    double ms = time_ms([&](int it) { hipLaunchKernelGGL((k1<KIND>), dim3(256 * w), dim3(256), 0, 0, dout, it, 1.0f); });
    double inst = 4000.0 * 64 * w;
    printf("%-34s waves/SIMD=%d  ms=%8.3f  cycles(2.4GHz)/inst/SIMD=%6.2f\n", name, w, ms, ms * 1e-3 * 2.4e9 / inst);
    fflush(stdout);
}

int main() {
    CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 2, 4}) {
        run<0, 0>("mfma_f32_32x32x2 only", w);
        run<2, 8>("8 pk_fma only", w);
        run<1, 8>("mfma + 8 pk_fma", w);
        run<2, 16>("16 pk_fma only", w);
        run<1, 16>("mfma + 16 pk_fma", w);
        run<4, 8>("8 sqrt only", w);
        run<3, 8>("mfma + 8 sqrt", w);
        run<3, 4>("mfma + 4 sqrt", w);
    }
    for (int w : {2, 8}) {
        run1<0>("v_fmac_f32 (VOP2)", w);
        run1<1>("v_fma_f32 v,v,s", w);
        run1<9>("v_fma_f32 a,a,a,b (2 distinct)", w);
        run1<11>("v_fma_f32 a,a,b,b (2 distinct)", w);
        run1<2>("v_mul_f32", w);
        run1<3>("v_pk_mul_f32", w);
        run1<12>("v_pk_add_f32 neg", w);
        run1<4>("v_cndmask_b32", w);
        run1<5>("v_cmp_lt_f32", w);
        run1<6>("v_min_u32", w);
        run1<7>("v_and_b32", w);
        run1<8>("v_add_f32", w);
        run1<10>("v_rsq_f32", w);
    }
    return 0;
}
