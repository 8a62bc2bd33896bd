// Probe: does global_load_lds_dwordx4 accept a source address that is only 4-byte aligned (gfx950)?
// Prints the number of mismatching floats for source offsets 0..3 floats.  hipcc -O3 --offload-arch=gfx950 tools/dma_align_probe.hip -o /tmp/dmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(const float* src, float* out, int off) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const int lane = threadIdx.x;
    const float* g = src + off + 4 * lane;   // 16 B per lane, base only 4-byte aligned when off % 4 != 0
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    for (int off = 0; off < 4; ++off) {
        hipMemset(o, 0, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, off);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(off + i);
        printf("offset %d floats: %s, %d of 256 wrong (first values %g %g %g %g)\n", off, hipGetErrorString(e), bad, r[0], r[1], r[2], r[3]);
    }
    return 0;
}
