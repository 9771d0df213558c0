// ubench6.hip — issue cost (cycles per wave64 instruction per SIMD) of the exact instruction forms of the scan loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define R8(x) x x x x x x x x
template <int V, int W>
__global__ __launch_bounds__(256, W) void k(float* out, int iters, float s0, float s1) {
    f2 a = {threadIdx.x * 0.01f, 1.0f}, b = {0.5f, 0.25f}, c = {0.1f, 0.2f}, d = {0.3f, 0.4f}, e = {1, 2}, f = {3, 4}, g = {5, 6}, h = {7, 8};
    f2 sp = {s0, s1};
    int i0 = threadIdx.x, i1 = 3, i2 = 4;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {  // independent pk_fma, VGPR only (4 accumulators)
            asm volatile(R8("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        } else if (V == 1) {  // pk_fma with SGPR broadcast as addend (op_sel_hi:[1,1,0])
            asm volatile(R8("v_pk_fma_f32 %0, %4, %5, %6 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %1, %4, %5, %6 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %2, %4, %5, %6 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %3, %4, %5, %6 op_sel_hi:[1,1,0]\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(sp));
        } else if (V == 2) {  // pk_add sgpr - vgpr  (oc = c - o)
            asm volatile(R8("v_pk_add_f32 %0, %6, %4 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %6, %5 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %6, %4 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %6, %5 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(sp));
        } else if (V == 3) {  // dependent pk_fma chain (1 accumulator)
            asm volatile(R8("v_pk_fma_f32 %0, %0, %5, %4\n v_pk_fma_f32 %0, %0, %5, %4\n v_pk_fma_f32 %0, %0, %5, %4\n v_pk_fma_f32 %0, %0, %5, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        } else if (V == 4) {  // pk_fma a*a+c (same source twice), neg on c
            asm volatile(R8("v_pk_fma_f32 %0, %4, %4, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %1, %5, %5, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %2, %4, %4, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %3, %5, %5, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        } else if (V == 5) {  // v_max_i32 / v_max3_i32
            asm volatile(R8("v_max_i32 %0, %1, %2\n v_max3_i32 %0, %0, %1, %2\n v_max_i32 %0, %1, %2\n v_max3_i32 %0, %0, %1, %2\n")
                         : "+v"(i0) : "v"(i1), "v"(i2));
        } else if (V == 6) {  // plain v_fma_f32 independent
            asm volatile(R8("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n")
                         : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x));
        } else if (V == 7) {  // 2-deep dependent pk (alternating two chains)
            asm volatile(R8("v_pk_fma_f32 %0, %0, %5, %4\n v_pk_fma_f32 %1, %1, %5, %4\n v_pk_fma_f32 %0, %0, %5, %4\n v_pk_fma_f32 %1, %1, %5, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        } else if (V == 9) {  // v_fma_f32 with SGPR addend, independent
            asm volatile(R8("v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %4, %5, %6\n v_fma_f32 %2, %4, %5, %6\n v_fma_f32 %3, %4, %5, %6\n")
                         : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x), "s"(s0));
        } else if (V == 10) {  // v_fma_f32 dependent chain
            asm volatile(R8("v_fma_f32 %0, %0, %5, %4\n v_fma_f32 %0, %0, %5, %4\n v_fma_f32 %0, %0, %5, %4\n v_fma_f32 %0, %0, %5, %4\n")
                         : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x));
        } else if (V == 11) {  // v_sub_f32 SGPR - VGPR (VOP2), independent
            asm volatile(R8("v_sub_f32 %0, %6, %4\n v_sub_f32 %1, %6, %5\n v_sub_f32 %2, %6, %4\n v_sub_f32 %3, %6, %5\n")
                         : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x), "s"(s0));
        } else if (V == 12) {  // the scalar sphere test as the compiler would interleave it: 2 spheres, SGPR operands
            asm volatile(R8("v_sub_f32 %0, %6, %4\n v_sub_f32 %1, %7, %4\n v_fma_f32 %2, %0, %0, %6\n v_fma_f32 %3, %1, %1, %7\n")
                         : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x), "s"(s0), "s"(s1));
        } else if (V == 13) {  // two-deep dependent v_fma_f32 (two alternating chains)
            asm volatile(R8("v_fma_f32 %0, %0, %5, %4\n v_fma_f32 %1, %1, %5, %4\n v_fma_f32 %0, %0, %5, %4\n v_fma_f32 %1, %1, %5, %4\n")
                         : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x));
        } else if (V == 14) {  // four alternating chains
            asm volatile(R8("v_fma_f32 %0, %0, %5, %4\n v_fma_f32 %1, %1, %5, %4\n v_fma_f32 %2, %2, %5, %4\n v_fma_f32 %3, %3, %5, %4\n")
                         : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x));
        } else if (V == 8) {  // pk_mul vgpr*vgpr
            asm volatile(R8("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %4, %5\n v_pk_mul_f32 %2, %4, %5\n v_pk_mul_f32 %3, %4, %5\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y + i0;
}

template <int V, int W>
void run(const char* name, float* dout) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    hipLaunchKernelGGL((k<V, W>), dim3(256 * W), dim3(256), 0, 0, dout, 100, 1.0f, 2.0f);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<V, W>), dim3(256 * W), dim3(256), 0, 0, dout, iters, 1.0f, 2.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s W=%d  cycles(2.4GHz)/instr/SIMD = %5.2f\n", name, W, ms * 1e-3 * 2.4e9 / ((double)iters * 32 * W));
    fflush(stdout);
}
#define RUNW(V, name) run<V, 1>(name, dout); run<V, 2>(name, dout); run<V, 4>(name, dout); run<V, 6>(name, dout); run<V, 8>(name, dout);
int main() {
    float* dout; CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
    RUNW(6, "v_fma_f32 independent");
    RUNW(0, "v_pk_fma_f32 independent, VGPR");
    RUNW(8, "v_pk_mul_f32 independent, VGPR");
    RUNW(1, "v_pk_fma_f32 SGPR addend op_sel_hi");
    RUNW(2, "v_pk_add_f32 SGPR - VGPR, neg");
    RUNW(4, "v_pk_fma_f32 a*a - c");
    RUNW(3, "v_pk_fma_f32 dependent chain");
    RUNW(7, "v_pk_fma_f32 two alternating chains");
    RUNW(5, "v_max_i32 / v_max3_i32");
    RUNW(9, "v_fma_f32 SGPR addend, independent");
    RUNW(10, "v_fma_f32 dependent chain");
    RUNW(13, "v_fma_f32 two alternating chains");
    RUNW(14, "v_fma_f32 four alternating chains");
    RUNW(11, "v_sub_f32 SGPR - VGPR");
    RUNW(12, "v_sub/v_fma mix, SGPR operands");
    return 0;
}
