// Helpers shared by the convolution kernels (conv.hip: direct implicit GEMM; conv_wino.hip: Winograd forms).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));  // one 32x32 fp32 MFMA accumulator block per lane

namespace {

constexpr int kNumCU = 256;  // MI355X

// Byte offsets are < 2^31 (launcher-checked); OR-ing bit 31 in pushes a lane beyond num_records, where a raw buffer
// load returns 0.  The offset is then made opaque so the compiler cannot turn the flag back into control flow around
// the load (it otherwise splits the block per condition and drains vmcnt between the pieces).
__device__ __forceinline__ unsigned oob_if(unsigned byte_off, bool invalid) {
    unsigned off = byte_off | ((unsigned)invalid << 31);
    asm volatile("" : "+v"(off));
    return off;
}

__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

}  // namespace
