// Shared declarations of the bf16 conv kernels (amp.hip, amp_ws.hip).  Reference path: the student forward under
// torch.cuda.amp.autocast (src/algorithms/fixmatch.py:97) - nn.Conv1d of src/models/backbones/resnet.py:55-72.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace ssecg_amp {

constexpr int kNumCU = 256;

__device__ __forceinline__ unsigned pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// LDS-DMA from inline asm: invisible to hipcc's wait-count bookkeeping, which otherwise drains the DMA queue (vmcnt(0)) before
// the first LDS read that follows a DMA in the same basic block.  The caller counts completions itself (s_waitcnt vmcnt(N)).
// Each lane's 16 bytes land at lds_byte_addr_wave_uniform + lane * 16.
__device__ __forceinline__ void lds_dma16_asm(const u32x4* gsrc, unsigned lds_byte_addr_wave_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr_wave_uniform) : "memory");
}

// Launch descriptor of the weights-stationary kernel (amp_ws.hip); filled by ssecg_amp_conv (amp.hip).
struct WsP {
    const u32x4* W;      // operand [(Csrc/16)*KS][2][M] 16-byte vectors (ssecg_amp_weight_operand_multi)
    const u32x4* src;    // blocked (N, Csrc/8, L)
    u32x4* out;          // blocked (N, M/8, Lrow): output position l of a row at l * ostride + ooff
    float* stats;        // [rows][M][2] or null
    int N, M, Csrc, L;   // L = output positions per sample (Ldst)
    int Lsrc, Lrow, ostride, ooff;
    int P, numPT;        // flattened output positions N*L, position tiles
    int Psrc;            // flattened source positions N*Lsrc
    int tapoff[3], tmin; // source position of (output l, tap t) = gmul * l + tapoff[t]; tmin = the smallest offset
    int MG, rows;        // channel groups of a position tile; position lanes (= statistics rows written)
    unsigned magic, magic_src;   // min(floor(2^32 / L) + 1, 2^32 - 1) for L and Lsrc: division as a multiply-high (divmod_pos)
    unsigned out_bytes;
};

// -> number of statistics rows (> 0) if the weights-stationary kernel takes this convolution, else 0
int ws_rows(int N, int Csrc, int Lsrc, int M, int Ldst, int ntaps, int gmul, int tapoff0, int tapoff1, int tapoff2, int Lrow,
            int ostride, int ooff, bool accumulate, bool want_stats);
int ws_launch(const void* src, const void* w_operand, void* out, int N, int Csrc, int Lsrc, int M, int Ldst, int ntaps, int gmul,
              int tapoff0, int tapoff1, int tapoff2, int Lrow, int ostride, int ooff, float* stats, hipStream_t st);

}  // namespace ssecg_amp
