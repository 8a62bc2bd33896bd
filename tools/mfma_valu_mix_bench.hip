// Diagnostic (not product code): what do vector instructions BETWEEN fp32 MFMAs cost the matrix pipe?  Every fp32 MFMA kernel of
// the step sits at 0.63-0.75 of the pipe (profiles/r06_pmc_conv_fp32.md), the fewer vector instructions per MFMA the higher; the
// compute skeleton of the layer1 experiment (no memory traffic at all) at 0.66 (profiles/r06_layer1_experiments.txt).  This loop
// isolates it: v_mfma_f32_32x32x2_f32 on six accumulators round-robin (the F(4,3) kernels' pattern) with K v_fma_f32 between two
// MFMAs, operands in registers, 1 or 2 waves per SIMD; DEP = the MFMA's A operand is the result of the vector instruction in
// front of it (an operand transformed on the fly); LDS = one ds_read_b128 per four MFMAs feeding the B operand.
// Reports cycles per MFMA per SIMD (s_memtime; floor 64), TFLOP/s by wall and the in-kernel clock.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_mix_bench.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long cyc, real; };

#define FENCE __builtin_amdgcn_sched_barrier(0)

template <int K, bool DEP, bool LDS, int NT>
__global__ __launch_bounds__(NT, NT / 256) void mix_loop(const float* __restrict__ in, float* __restrict__ out, Stamp* stamps, int iters) {
    __shared__ float4 sB[1024];
    const int tid = threadIdx.x;
    float a[8], b[8], x[6];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(blockIdx.x * NT + tid) * 16 + i];
        b[i] = in[(blockIdx.x * NT + tid) * 16 + 8 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = a[i] * 0.5f;
    for (int i = tid; i < 1024; i += NT) sB[i] = make_float4(a[0], a[1], b[0], b[1]);
    __syncthreads();
    f32x16 acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const float4* lb = sB + (tid & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 u = make_float4(b[j], b[(j + 1) & 7], b[(j + 2) & 7], b[(j + 3) & 7]);
            float4 u2 = u;
            if (LDS) { u = lb[((it + j) & 7) * 64]; u2 = lb[((it + j + 3) & 7) * 64 + 512]; FENCE; }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float av = a[(j + k) & 7];
#pragma unroll
                for (int v = 0; v < K; ++v) {
                    // independent vector work (an operand transform): x[(k+v)%6] = fma(x, c, a)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(k + v) % 6]) : "v"(b[v & 7]), "v"(a[(v + 1) & 7]));
                }
                if (DEP && K > 0) av = x[(k + K - 1) % 6];
                FENCE;
                const float bv = k < 4 ? (k == 0 ? u.x : k == 1 ? u.y : k == 2 ? u.z : u.w) : (k == 4 ? u2.x : u2.y);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
                FENCE;
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float res = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) res += acc[k][r];
#pragma unroll
    for (int i = 0; i < 6; ++i) res += x[i];
    out[blockIdx.x * NT + tid] = res;
    if (tid == 0) { stamps[blockIdx.x].cyc = c1 - c0; stamps[blockIdx.x].real = r1 - r0; }
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
// the same FLOPs and the same vector instructions per FLOP on v_mfma_f32_16x16x4_f32 (two MFMAs of 32 cycles per K vector instructions)
template <int K, int NT>
__global__ __launch_bounds__(NT, NT / 256) void mix_loop16(const float* __restrict__ in, float* __restrict__ out, Stamp* stamps, int iters) {
    const int tid = threadIdx.x;
    float a[8], b[8], x[6];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(blockIdx.x * NT + tid) * 16 + i];
        b[i] = in[(blockIdx.x * NT + tid) * 16 + 8 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = a[i] * 0.5f;
    f32x4v acc[24];
#pragma unroll
    for (int k = 0; k < 24; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[k][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
#pragma unroll
                for (int v = 0; v < K; ++v)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(k + v) % 6]) : "v"(b[v & 7]), "v"(a[(v + 1) & 7]));
                FENCE;
                acc[2 * k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + k) & 7], b[(2 * j + k) & 7], acc[2 * k], 0, 0, 0);
                acc[2 * k + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + k + 1) & 7], b[(2 * j + k) & 7], acc[2 * k + 1], 0, 0, 0);
                FENCE;
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float res = 0.f;
#pragma unroll
    for (int k = 0; k < 24; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) res += acc[k][r];
#pragma unroll
    for (int i = 0; i < 6; ++i) res += x[i];
    out[blockIdx.x * NT + tid] = res;
    if (tid == 0) { stamps[blockIdx.x].cyc = c1 - c0; stamps[blockIdx.x].real = r1 - r0; }
}

template <int K, int NT>
static void run16(const float* in, float* out, Stamp* stamps, int blocks, int iters, int launches) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mix_loop16<K, NT>), dim3(blocks), dim3(NT), 0, 0, in, out, stamps, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((mix_loop16<K, NT>), dim3(blocks), dim3(NT), 0, 0, in, out, stamps, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks);
    hipMemcpy(h.data(), stamps, blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (auto& s : h) if (s.real) { ghz.push_back((double)s.cyc / (double)s.real * 0.1); cyc.push_back((double)s.cyc); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const int wps = NT / 256;
    const double pairs_per_simd = (double)iters * 48 * wps;   // a pair of 16x16x4 = the FLOPs of one 32x32x2
    const double flops = (double)launches * blocks * (NT / 64) * (double)iters * 48 * 4096.0;
    printf("16x16x4 K=%d per MFMA pair, %d wave/SIMD | %7.1f cycles per pair per SIMD (floor 64) | %7.1f TFLOP/s | clock %.3f GHz\n", K, wps,
           cyc[cyc.size() / 2] / pairs_per_simd, flops / (ms * 1e-3) / 1e12, ghz[ghz.size() / 2]);
    fflush(stdout);
}

template <int K, bool DEP, bool LDS, int NT>
static void run(const float* in, float* out, Stamp* stamps, int blocks, int iters, int launches) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mix_loop<K, DEP, LDS, NT>), dim3(blocks), dim3(NT), 0, 0, in, out, stamps, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((mix_loop<K, DEP, LDS, NT>), dim3(blocks), dim3(NT), 0, 0, in, out, stamps, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks);
    hipMemcpy(h.data(), stamps, blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (auto& s : h) if (s.real) { ghz.push_back((double)s.cyc / (double)s.real * 0.1); cyc.push_back((double)s.cyc); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const int wps = NT / 256;                                   // waves per SIMD
    const double mfma_per_simd = (double)iters * 48 * wps;
    const double flops = (double)launches * blocks * (NT / 64) * (double)iters * 48 * 4096.0;
    printf("K=%d %s %s %d wave/SIMD | %7.1f cycles per MFMA per SIMD (floor 64) | %7.1f TFLOP/s | clock %.3f GHz | %.3f ms\n", K,
           DEP ? "DEP  " : "indep", LDS ? "LDS" : "reg", wps, cyc[cyc.size() / 2] / mfma_per_simd, flops / (ms * 1e-3) / 1e12 / launches * launches,
           ghz[ghz.size() / 2], ms / launches);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int blocks = 256, iters = argc > 1 ? atoi(argv[1]) : 1000, launches = argc > 2 ? atoi(argv[2]) : 100;
    float *in, *out;
    Stamp* stamps;
    hipMalloc(&in, (size_t)blocks * 512 * 16 * 4);
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipMalloc(&stamps, blocks * sizeof(Stamp));
    std::vector<float> h((size_t)blocks * 512 * 16);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
#define ROW(K_) run<K_, false, false, 512>(in, out, stamps, blocks, iters, launches);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(6) ROW(8) ROW(12)
#undef ROW
    run<2, true, false, 512>(in, out, stamps, blocks, iters, launches);
    run<4, true, false, 512>(in, out, stamps, blocks, iters, launches);
    run<0, false, true, 512>(in, out, stamps, blocks, iters, launches);
    run<3, false, true, 512>(in, out, stamps, blocks, iters, launches);
    run<3, true, true, 512>(in, out, stamps, blocks, iters, launches);
    run<0, false, false, 256>(in, out, stamps, blocks, iters, launches);
    run<3, false, false, 256>(in, out, stamps, blocks, iters, launches);
    run<6, false, false, 256>(in, out, stamps, blocks, iters, launches);
    run<3, true, true, 256>(in, out, stamps, blocks, iters, launches);
    run16<0, 512>(in, out, stamps, blocks, iters, launches);
    run16<1, 512>(in, out, stamps, blocks, iters, launches);
    run16<2, 512>(in, out, stamps, blocks, iters, launches);
    run16<3, 512>(in, out, stamps, blocks, iters, launches);
    run16<4, 512>(in, out, stamps, blocks, iters, launches);
    run16<6, 512>(in, out, stamps, blocks, iters, launches);
    run16<3, 256>(in, out, stamps, blocks, iters, launches);
    return 0;
}
