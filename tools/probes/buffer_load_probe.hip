// Probe (GPU box): what does a 16-byte raw buffer load return (a) at a 4-byte-aligned (not 16-byte-aligned) offset, (b) when it
// straddles the end of the buffer (num_records)?  Decides whether the Winograd input staging can use one dwordx4 per quad.
// build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_load_probe.hip -o /tmp/blp && /tmp/blp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int W3>
__global__ void probe(const float* src, int nbytes, const int* offs, float* out) {
    const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, W3);
    const int t = threadIdx.x;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)offs[t], 0, 0);
    // (NOT __builtin_bit_cast(float, v.y): this clang reads element 0 for every swizzle operand of __builtin_bit_cast)
    const unsigned a = v.x, b = v.y, c = v.z, e = v.w;
    out[4 * t + 0] = __uint_as_float(a); out[4 * t + 1] = __uint_as_float(b);
    out[4 * t + 2] = __uint_as_float(c); out[4 * t + 3] = __uint_as_float(e);
}
int main() {
    const int n = 64;                       // floats in the buffer: values 1..64; guard floats 1001.. behind it
    float h[n + 8]; for (int i = 0; i < n + 8; ++i) h[i] = i < n ? (float)(i + 1) : (float)(1001 + i - n);
    float* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    int ho[64]; int k = 0;
    for (int o : {0, 4, 8, 12, 20, 36, 4 * (n - 4), 4 * (n - 3), 4 * (n - 2), 4 * (n - 1), 4 * n, -4}) ho[k++] = o;
    for (; k < 64; ++k) ho[k] = 0;
    int* dofs; hipMalloc(&dofs, sizeof(ho)); hipMemcpy(dofs, ho, sizeof(ho), hipMemcpyHostToDevice);
    float* dout; hipMalloc(&dout, 256 * 4); 
    float out[256];
    for (int variant = 0; variant < 3; ++variant) {
        if (variant == 0) probe<0x00020000><<<1, 64>>>(d, n * 4, dofs, dout);
        if (variant == 1) probe<0x00027000><<<1, 64>>>(d, n * 4, dofs, dout);
        if (variant == 2) probe<0x00027FAC><<<1, 64>>>(d, n * 4, dofs, dout);
        hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
        printf("descriptor word 3 variant %d\n", variant);
        for (int t = 0; t < 12; ++t) printf("offset %4d bytes -> %7.1f %7.1f %7.1f %7.1f\n", ho[t], out[4 * t], out[4 * t + 1], out[4 * t + 2], out[4 * t + 3]);
    }
    return 0;
}
