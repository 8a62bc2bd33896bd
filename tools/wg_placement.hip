// Diagnostic: where do the workgroups of a 512-workgroup launch (256 threads, ~62 KB LDS: two per CU) land?
// Prints, for a few CUs, the workgroup ids that shared them.  hipcc -O3 --offload-arch=gfx950 tools/wg_placement.hip -o /tmp/wgp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
    __shared__ float pad[15500];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pad[threadIdx.x] = (float)hw;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (unsigned)pad[1] * 0u; }
}
int main() {
    const int G = 512;
    unsigned* d; hipMalloc(&d, G * 8);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * G);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int i = 0; i < G; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back(i);
    }
    printf("%zu distinct (xcc, se, sh, cu) slots for %d workgroups\n", cu.size(), G);
    int n = 0;
    for (auto& kv : cu) {
        if (n++ % 16 == 0) { printf("slot %05x:", kv.first); for (int w : kv.second) printf(" %d", w); printf("\n"); }
    }
    return 0;
}
