// Helpers shared by the convolution kernels (conv.hip: direct implicit GEMM; conv_wino.hip: Winograd forms).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));  // one 32x32 fp32 MFMA accumulator block per lane

namespace ssecg_detail {
// K split of small convolution launches (conv.hip; also used by conv_wino4.hip).  The split kernels write S partial results
// with the plain epilogue into ``part`` (S planes laid out like the output tensor); this pass adds them in split order (fixed:
// reproducible), applies the launch's epilogue ([* scale[m]] [+ shift[m]] [+ residual] [ReLU]) and, for a train-mode forward,
// emits the per-channel BatchNorm partial sums of what it stored into rows [0, rows_used) of ``stats`` ([stats_parts][M][2];
// the remaining rows are zeroed).  Output element (n, m, j) lives at (n*M + m)*Lrow + j*ostride + ooff (stride-2 data-gradient
// phases write every other position).  residual == out is allowed (an accumulating data gradient).
int launch_split_finish(const float* part, int S, size_t plane, float* out, int N, int M, int Ldst, int Lrow, int ostride, int ooff,
                        const float* scale, const float* shift, const float* residual, int relu, float* stats, int stats_parts,
                        hipStream_t st);
// largest power of two S <= 8 such that S workgroup columns of ``tiles`` tiles still fit the chip's ``slots`` and every split keeps
// whole stages of ``cgran`` input channels (>= 2 of them); 1 = no split.  Only launches of at most 64 position tiles are split
// (small batches: <= 128 student windows at this network's lengths) - a launch of the 512-window batch never is, whatever its
// channel-tile count (tests/test_fullsize_gpu.py rests on that).
inline int pick_ksplit(long long tiles, int pos_tiles, int slots, int C, int cgran) {
    int s = 1;
    if (pos_tiles > 64) return 1;
    while (s < 8 && tiles * (2 * s) <= slots && (C / (2 * s)) % cgran == 0 && C / (2 * s) >= 2 * cgran) s *= 2;
    return s;
}
}  // namespace ssecg_detail

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

// The fused part of the conv epilogues: 16 output rows (channels rbase, rbase+2, ...) of one 32-position round, read back from the
// wave's transpose tile T with the lane on the position axis: [* scale[row]] [+ shift[row]] [+ residual] [ReLU] -> out.
// Every global operand of the round - the per-row scale / shift and the residual (for an accumulating data gradient the gradient
// already in ``out``: residual == out) - is loaded BEFORE the first store.  Inside the store loop each load would have to wait
// for the store in front of it (the compiler cannot rule out that they alias; for residual == out they do, but only at the same
// element of the same lane): 16 dependent global round trips per round, MEASURED +65...+140 us on every launch with a residual
// (tools/kernel_sequence.sh, profiles/r04_residual_epilogue.txt).  Same arithmetic, operation for operation, as the loop it replaces.
// ``lds_ss`` (optional): the workgroup's (scale, shift) pairs staged in LDS, indexed by row - m_lo (LDS reads do not queue behind
// global stores at all; the F(4,3) kernel's teacher-pass launches gained 10-15 us each from this).
__device__ __forceinline__ void epilogue_rows_fused(const float* T, int lhi, int l31, bool pok, int rbase, int M, unsigned o,
                                                    unsigned ostep, const float* scale, const float* shift, const float* residual,
                                                    int relu, float* out, const float2* lds_ss = nullptr, int m_lo = 0) {
    float sc[16], sh[16], res[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        const int row = rbase + 2 * k2;
        const bool ok = pok && row < M;
        if (lds_ss != nullptr) {
            const float2 ss = lds_ss[row - m_lo];
            sc[k2] = ss.x; sh[k2] = ss.y;
        } else {
            sc[k2] = (scale != nullptr && ok) ? scale[row] : 1.f;
            sh[k2] = (shift != nullptr && ok) ? shift[row] : 0.f;
        }
        res[k2] = (residual != nullptr && ok) ? residual[o + (unsigned)k2 * ostep] : 0.f;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        const int row = rbase + 2 * k2;
        float v = T[(2 * k2 + lhi) * 33 + l31];
        if (pok && row < M) {
            if (scale != nullptr) v *= sc[k2];
            if (shift != nullptr) v += sh[k2];
            if (residual != nullptr) v += res[k2];
            if (relu) v = fmaxf(v, 0.f);
            out[o] = v;
        }
        o += ostep;
    }
}

}  // namespace
