// ubench4.hip — the REAL scan_spheres() of rt_device.h (scalar loads + packed math) in isolation:
// cycles per sphere pair-test per wave-pass, for a 488-sphere list, at a chosen waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../raytracing_amd/csrc/rt_device.h"
using namespace rtd;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int W>
__global__ __launch_bounds__(256, W) void k(const DevScene S, float* out, int reps) {
    const float fx = threadIdx.x * 0.01f + blockIdx.x * 1e-4f;
    Pair p;
    p.ox = f2{13.f + fx, 12.f - fx}; p.oy = f2{2.f, 2.1f}; p.oz = f2{3.f, 2.9f};
    p.dx = f2{-0.96f, -0.95f}; p.dy = f2{-0.15f, -0.1f}; p.dz = f2{-0.22f, -0.29f};
    p.time = f2{0.f, 0.f};
    uint32_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        uint32_t ba = kInfBits, bb = kInfBits;
        int sa = -1, sb = -1;
        scan_spheres(S.lists, p, ba, bb, sa, sb);
        acc ^= ba ^ bb ^ (uint32_t)sa ^ (uint32_t)sb;
        p.ox.x += 1e-5f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = __uint_as_float(acc);
}

template <int W>
void run(const DevScene& S, float* dout, int n) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 200;
    hipLaunchKernelGGL(k<W>, dim3(256 * W), dim3(256), 0, 0, S, dout, 10);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<W>, dim3(256 * W), dim3(256), 0, 0, S, dout, reps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double tests = (double)reps * n * W;  // pair-tests per SIMD
    printf("waves/SIMD=%d n=%d  ms=%7.3f  cycles(2.4GHz) per sphere pair-test per SIMD = %6.1f  (=> %4.1f%% of fp32 peak)\n", W, n, ms,
           ms * 1e-3 * 2.4e9 / tests, 100.0 * 17 * 128 / 64 / (ms * 1e-3 * 2.4e9 / tests));
}

int main() {
    for (int n : {64, 488}) {
        std::vector<float4> st(n);
        for (int i = 0; i < n; ++i) st[i] = make_float4((i % 22) - 11 + 0.3f, 0.2f, (i / 22) - 11 + 0.4f, -0.04f);
        DevScene S{};
        float4* d; CHECK(hipMalloc(&d, n * sizeof(float4)));
        CHECK(hipMemcpy(d, st.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        S.lists.stat = d; S.lists.ns_pad = n; S.lists.nm_pad = 0; S.lists.mov_a = d; S.lists.mov_b = d;
        float* dout; CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
        run<1>(S, dout, n); run<2>(S, dout, n); run<3>(S, dout, n); run<4>(S, dout, n);
    }
    return 0;
}
