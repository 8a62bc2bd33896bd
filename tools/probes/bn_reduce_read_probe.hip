// Probe (GPU box): how fast can the two tensors of a BatchNorm-backward reduction be READ under three traversal orders?
//   A  one workgroup per (channel, sample slab): rows of L floats strided by C*L (what bn_bwd_reduce_kernel does)
//   B  flat grid-stride sweep, 16 B per lane (what bn_bwd_apply_kernel does; no per-channel result: upper bound)
//   C  one workgroup per contiguous chunk of whole rows, a wave per row (the candidate: per-channel sums stay possible)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/bn_reduce_read_probe.hip -o /tmp/bnprobe && /tmp/bnprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void kA(const float* dy, const float* x, int N, int C, int L, float* out) {
    const int c = blockIdx.x, S = gridDim.y, per = (N + S - 1) / S, n0 = blockIdx.y * per, n1 = min(N, n0 + per);
    float s = 0.f;
    const int items = (n1 - n0) * L;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int n = n0 + it / L, l = it - (it / L) * L;
        const size_t e = ((size_t)n * C + c) * L + l;
        s += dy[e] * x[e];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) out[((size_t)blockIdx.y * gridDim.x + c) * 4 + (threadIdx.x >> 6)] = s;   // (no atomics: they would dominate)
}
__global__ void kB(const float4* dy, const float4* x, size_t nv, float* out) {
    float s = 0.f;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (size_t)gridDim.x * blockDim.x) {
        const float4 a = dy[v], b = x[v];
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}
template <int RU>   // rows in flight per wave
__global__ void kC(const float* dy, const float* x, long rows, int C, int L, int rpb, float* out) {
    __shared__ float acc[4][512];
    for (int i = threadIdx.x; i < 4 * 512; i += 256) (&acc[0][0])[i] = 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r0 = (long)blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    for (long r = r0 + wave * RU; r < r1; r += 4 * RU) {
        float s[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            s[u] = 0.f;
            if (r + u < r1) {
                const size_t base = (size_t)(r + u) * L;
                for (int l = lane; l < L; l += 64) s[u] += dy[base + l] * x[base + l];
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            float v = s[u];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0 && r + u < r1) acc[wave][(r + u) % C] += v;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) out[(size_t)blockIdx.x * C + c] = acc[0][c] + acc[1][c] + acc[2][c] + acc[3][c];
}

int main() {
    const int shapes[4][3] = {{1024, 64, 500}, {1024, 128, 250}, {1024, 256, 125}, {1024, 512, 63}};
    const int NB = 3;                                  // rotate through buffer pairs (> 256 MB Infinity Cache in total)
    for (auto& sh : shapes) {
        const int N = sh[0], C = sh[1], L = sh[2];
        const size_t n = (size_t)N * C * L;
        std::vector<float*> dy(NB), x(NB);
        for (int i = 0; i < NB; ++i) { CK(hipMalloc(&dy[i], n * 4)); CK(hipMalloc(&x[i], n * 4)); CK(hipMemset(dy[i], 0, n * 4)); CK(hipMemset(x[i], 0, n * 4)); }
        float* out; CK(hipMalloc(&out, (size_t)4096 * 512 * 4 * 2));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](auto launch, const char* name) {
            for (int i = 0; i < NB; ++i) launch(i);
            hipEventRecord(e0);
            const int reps = 12;
            for (int i = 0; i < reps; ++i) launch(i % NB);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %-44s %7.1f us  %5.2f TB/s\n", name, ms / reps * 1e3, 2.0 * n * 4 / (ms / reps * 1e-3) / 1e12);
        };
        printf("N=%d C=%d L=%d (%.0f MB per tensor)\n", N, C, L, n * 4 / 1e6);
        const int S = 2048 / C;
        timeit([&](int i) { hipLaunchKernelGGL(kA, dim3(C, S), dim3(256), 0, 0, dy[i], x[i], N, C, L, out); }, "A channel-per-workgroup (current)");
        if (n % 4 == 0) {
            timeit([&](int i) { hipLaunchKernelGGL(kB, dim3(2048), dim3(256), 0, 0, (const float4*)dy[i], (const float4*)x[i], n / 4, out); }, "B flat float4 sweep, 2048 blocks (upper bound)");
            timeit([&](int i) { hipLaunchKernelGGL(kB, dim3(8192), dim3(256), 0, 0, (const float4*)dy[i], (const float4*)x[i], n / 4, out); }, "B flat float4 sweep, 8192 blocks");
        }
        const long rows = (long)N * C;
        for (int G : {1024, 2048, 4096}) {
            const int rpb = (int)((rows + G - 1) / G);
            char nm[64];
            snprintf(nm, 64, "C chunk-per-block, 1 row/wave, G=%d", G);
            timeit([&](int i) { hipLaunchKernelGGL(kC<1>, dim3(G), dim3(256), 0, 0, dy[i], x[i], rows, C, L, rpb, out); }, nm);
            snprintf(nm, 64, "C chunk-per-block, 4 rows/wave, G=%d", G);
            timeit([&](int i) { hipLaunchKernelGGL(kC<4>, dim3(G), dim3(256), 0, 0, dy[i], x[i], rows, C, L, rpb, out); }, nm);
        }
        for (int i = 0; i < NB; ++i) { hipFree(dy[i]); hipFree(x[i]); }
        hipFree(out);
    }
    return 0;
}
