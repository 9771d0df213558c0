// ubench.hip — gfx950 VALU issue-rate microbenchmarks that decide how the sphere scan is written.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench tools/ubench.hip ; run on the GPU box.
// Each kernel runs `iters` x 64 instructions of one kind per wave over 16 independent registers;
// grid = 256 CUs x m workgroups of 256 threads -> m waves per SIMD.  Reports wave-instructions
// per SIMD-cycle from s_memtime deltas (in-kernel, so DVFS does not matter) and wall GFLOP/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef float float2_ __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind { FMA = 0, PKFMA, SQRT, MIN3, BFI, SUB, PKADD, MIXSCAN, LDSFMA, PKFMA_OPSEL, N_KINDS };
static const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_sqrt_f32", "v_min3_u32", "v_bfi_b32",
                               "v_sub_f32", "v_pk_add_f32", "scan-mix(10fma+sqrt+bfi+add+min)", "ds_read_b128+12fma",
                               "v_pk_fma_f32 op_sel bcast"};
// flops per lane per instruction (for the GFLOP/s column)
static const double kFlops[] = {2, 4, 1, 0, 0, 1, 2, 0, 0, 4};

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
    __shared__ float4 sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = make_float4(seed, seed * 2, seed * 3, seed * 4);
    __syncthreads();
    float a[16];
    float2_ p[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = float2_{a[i], a[i] + 1.f}; }
    float b = seed * 0.5f, c = seed * 0.25f;
    float2_ pb = {b, c}, pc = {c, b};
    unsigned u = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (KIND == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP16(X)
#undef X
            } else if (KIND == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                REP16(X)
#undef X
            } else if (KIND == PKFMA_OPSEL) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pb), "v"(pc));
                REP16(X)
#undef X
            } else if (KIND == SQRT) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                REP16(X)
#undef X
            } else if (KIND == MIN3) {
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP16(X)
#undef X
            } else if (KIND == BFI) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP16(X)
#undef X
            } else if (KIND == SUB) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                REP16(X)
#undef X
            } else if (KIND == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                REP16(X)
#undef X
            } else if (KIND == MIXSCAN) {
                // 4 independent copies of the per-sphere instruction mix: 3 sub, mul, 6 fma, sqrt, bfi, add, min
#define X(i)                                                                                                   \
    asm volatile(                                                                                              \
        "v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %5, %1\n v_sub_f32 %2, %4, %2\n"                                 \
        "v_mul_f32 %3, %0, %4\n v_fma_f32 %3, %1, %5, %3\n v_fma_f32 %3, %2, %4, %3\n"                         \
        "v_fma_f32 %0, %0, %0, %5\n v_fma_f32 %0, %1, %1, %0\n v_fma_f32 %0, %2, %2, %0\n"                     \
        "v_fma_f32 %1, %3, %3, -%0\n v_sqrt_f32 %1, %1\n v_bfi_b32 %1, %6, %1, %0\n v_sub_f32 %1, %3, %1\n"    \
        "v_min_u32 %2, %2, %1\n v_mov_b32 %0, %1\n v_mov_b32 %3, %2\n"                                         \
        : "+v"(a[4 * i]), "+v"(a[4 * i + 1]), "+v"(a[4 * i + 2]), "+v"(a[4 * i + 3])                           \
        : "v"(b), "v"(c), "s"(0x7fffffff));
                X(0) X(1) X(2) X(3)
#undef X
            } else if (KIND == LDSFMA) {
                // one broadcast ds_read_b128 per 12 FMAs (the scan's LDS : VALU ratio at 1 ray per lane)
                float4 s = sh[(it + rep) & 63];
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s.x), "v"(s.y));
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
#undef X
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = b + c + (float)u;
    for (int i = 0; i < 16; ++i) acc += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
void run(int waves_per_simd, int iters, float* dout, long long* dcyc) {
    const int grid = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, dout, dcyc, iters / 8, 1.0f);  // warm-up
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, dout, dcyc, iters, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc(grid * 4);
    CHECK(hipMemcpy(cyc.data(), dcyc, cyc.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (long long c : cyc) mean += (double)c;
    mean /= cyc.size();
    int per_iter = 64;
    if (KIND == MIXSCAN) per_iter = 4 * 4 * 16;
    if (KIND == LDSFMA) per_iter = 4 * 12;
    const double inst = (double)iters * per_iter;  // wave-instructions per wave
    // s_memtime ticks at 100 MHz on this part if readcyclecounter maps to s_memrealtime; print both views
    const double simd_rate = waves_per_simd * inst / mean;
    const double lanes = (double)grid * 256;
    const double gflops = lanes * inst * kFlops[KIND] / (ms * 1e6);
    printf("%-36s waves/SIMD=%d  ms=%8.3f  ticks/wave=%12.0f  inst/tick/SIMD=%7.4f  wave-inst/s/SIMD=%8.3f G  TFLOP/s=%7.2f\n",
           kNames[KIND], waves_per_simd, ms, mean, simd_rate, (double)waves_per_simd * inst / (ms * 1e6), gflops / 1e3);
    fflush(stdout);
}

int main() {
    float* dout;
    long long* dcyc;
    CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&dcyc, 256 * 8 * 4 * sizeof(long long)));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        run<FMA>(w, iters, dout, dcyc);
        run<PKFMA>(w, iters, dout, dcyc);
        run<PKFMA_OPSEL>(w, iters, dout, dcyc);
        run<SQRT>(w, iters, dout, dcyc);
        run<MIN3>(w, iters, dout, dcyc);
        run<BFI>(w, iters, dout, dcyc);
        run<SUB>(w, iters, dout, dcyc);
        run<PKADD>(w, iters, dout, dcyc);
        run<MIXSCAN>(w, iters / 4, dout, dcyc);
        run<LDSFMA>(w, iters, dout, dcyc);
    }
    return 0;
}
