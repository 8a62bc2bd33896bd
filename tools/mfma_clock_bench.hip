// Diagnostic (not product code): which fp32 MFMA shape sustains the higher wall-clock rate on this device, and at what
// shader clock?  Bare MFMA loops on random operands held in registers, 8 waves per CU (two per SIMD), 96 accumulator
// registers per wave (the F(4,3) kernel's budget).  The in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6), median over workgroups.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_clock_bench.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, real; };

template <int SHAPE>   // 0: 32x32x2 f32, 1: 16x16x4 f32
__global__ __launch_bounds__(512, 2) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, Stamp* stamps, int iters) {
    const int tid = threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(blockIdx.x * 512 + tid) * 16 + i];
        b[i] = in[(blockIdx.x * 512 + tid) * 16 + 8 + i];
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float res = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 6; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + k) & 7], acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) res += acc[k][r];
    } else {
        f32x4 acc[24];
#pragma unroll
        for (int k = 0; k < 24; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)   // same FLOPs per iteration: 4 x 24 x 2048 x ... = 8 x 6 x 4096
#pragma unroll
                for (int k = 0; k < 24; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + k) & 7], b[(2 * j + k) & 7], acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 24; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) res += acc[k][r];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 512 + tid] = res;
    if (tid == 0) { stamps[blockIdx.x].cyc = c1 - c0; stamps[blockIdx.x].real = r1 - r0; }
}

template <int SHAPE>
static void run(const char* name, const float* in, float* out, Stamp* stamps, int blocks, int iters, int launches) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(512), 0, 0, in, out, stamps, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(512), 0, 0, in, out, stamps, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks);
    hipMemcpy(h.data(), stamps, blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (auto& s : h) if (s.real) ghz.push_back((double)s.cyc / (double)s.real * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flops = (double)launches * blocks * 8 * (double)iters * 48 * 4096.0;
    printf("%-12s %8.3f ms/launch  %7.1f TFLOP/s  in-kernel clock %.3f GHz (median of %zu workgroups)\n", name, ms / launches,
           flops / (ms * 1e-3) / 1e12, ghz.empty() ? 0.0 : ghz[ghz.size() / 2], ghz.size());
}

int main(int argc, char** argv) {
    const int blocks = 256, iters = argc > 1 ? atoi(argv[1]) : 2000, launches = argc > 2 ? atoi(argv[2]) : 300;
    float *in, *out;
    Stamp* stamps;
    hipMalloc(&in, (size_t)blocks * 512 * 16 * 4);
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipMalloc(&stamps, blocks * sizeof(Stamp));
    std::vector<float> h((size_t)blocks * 512 * 16);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("32x32x2 f32", in, out, stamps, blocks, iters, launches);
        run<1>("16x16x4 f32", in, out, stamps, blocks, iters, launches);
    }
    return 0;
}
