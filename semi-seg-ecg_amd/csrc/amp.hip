// Reduced-precision variant of the conv path (reference: `use_amp: true`, the student forward under
// torch.cuda.amp.autocast, src/algorithms/fixmatch.py:97): bf16 STORAGE of every activation between the stem and the
// classifier, bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulation) for every conv of the body, fp32 master weights,
// fp32 BatchNorm statistics, fp32 loss.
//
// HBM layout of a bf16 activation: "blocked" (N, C/8, L, 8) - eight channels of one position are one 16-byte vector.
// That is exactly the MFMA operand fragment of an implicit-GEMM convolution over channels (lane = position, 8
// consecutive k = 8 channels), so forward and data-gradient kernels load their X fragments straight from HBM/L2 with
// one 16-byte load per lane and k-step, tap shifts are whole-vector shifts (no alignment problem), and nothing is ever
// transposed; the accumulator tile D[channel][position] puts 4 consecutive channels of one position in one lane, i.e.
// 8-byte stores that pair up into full 16-byte vectors across the two lane halves.  Only the weight gradient contracts
// over POSITIONS and needs the transposed view; it gets it from ds_read_b64_tr_b16 on an LDS image of the tiles.
//
// The teacher / eval passes are outside autocast in the reference and stay on the fp32 kernels (conv.hip, conv_wino.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "amp_common.h"
#include "ssecg.h"

using namespace ssecg_amp;
typedef unsigned u32x2v __attribute__((__vector_size__(8)));   // operand type of the raw buffer b64 store builtin

// bit 31 pushes a lane's byte offset beyond num_records (the buffer store is dropped, the load returns 0); made opaque so that the
// flag is not turned back into control flow around the access (as conv_common.h::oob_if)
__device__ __forceinline__ unsigned amp_oob_if(unsigned byte_off, bool invalid) {
    unsigned off = byte_off | ((unsigned)invalid << 31);
    asm volatile("" : "+v"(off));
    return off;
}

namespace {

__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
    v.x = pack2(f[0], f[1]); v.y = pack2(f[2], f[3]); v.z = pack2(f[4], f[5]); v.w = pack2(f[6], f[7]);
    return v;
}

// ------------------------------------------------------------------------------------------------ layout converters
__global__ __launch_bounds__(256) void cvt_planar_to_blocked_kernel(const float* __restrict__ x, u32x4* __restrict__ y,
                                                                    int N, int C, int L) {
    const int CB = C >> 3;
    const size_t total = (size_t)N * CB * L;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t row = idx / L;
        const int l = (int)(idx - row * L);
        const int cb = (int)(row % CB);
        const size_t n = row / CB;
        const float* px = x + (n * C + 8 * cb) * (size_t)L + l;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = px[(size_t)j * L];
        y[idx] = pack8(f);
    }
}

__global__ __launch_bounds__(256) void cvt_blocked_to_planar_kernel(const u32x4* __restrict__ x, float* __restrict__ y,
                                                                    int N, int C, int L) {
    const int CB = C >> 3;
    const size_t total = (size_t)N * CB * L;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t row = idx / L;
        const int l = (int)(idx - row * L);
        const int cb = (int)(row % CB);
        const size_t n = row / CB;
        float f[8];
        unpack8(x[idx], f);
        float* py = y + (n * C + 8 * cb) * (size_t)L + l;
#pragma unroll
        for (int j = 0; j < 8; ++j) py[(size_t)j * L] = f[j];
    }
}

// ------------------------------------------------------------------------------------------------ weight operands
// fp32 master weights (Cout, Cin, K) -> bf16 MFMA A-operands, one launch for every conv of the model.
//   operand[(cc*KSe + tt)][h][m][j],  cc = 16-channel chunk of the contracted axis, tt = index into the tap list,
//   h = lane half, m = output row, j = 0..7:
//     forward  (transposed = 0): m = co, value = w[co][16cc + 8h + j][tap[tt]]
//     data grad (transposed = 1): m = ci, value = w[16cc + 8h + j][ci][tap[tt]]
// table rows (int64): { w*, operand*, Cout, Cin, K, transposed, KSe, tap0 | tap1<<8 | tap2<<16 }
__global__ __launch_bounds__(256) void weight_operand_multi_kernel(const int64_t* __restrict__ table) {
    const int64_t* row = table + 8 * (size_t)blockIdx.y;
    const float* w = reinterpret_cast<const float*>(row[0]);
    u32x4* op = reinterpret_cast<u32x4*>(row[1]);
    const int Cout = (int)row[2], Cin = (int)row[3], K = (int)row[4], tr = (int)row[5], KSe = (int)row[6];
    const int taps = (int)row[7];
    const int M = tr ? Cin : Cout, Ck = tr ? Cout : Cin;   // output rows, contracted channels
    const int total = (Ck / 16) * KSe * 2 * M;             // 16-byte vectors
    for (int v = blockIdx.x * 256 + threadIdx.x; v < total; v += gridDim.x * 256) {
        const int m = v % M;
        const int h = (v / M) & 1;
        const int s = v / (2 * M);
        const int tt = s % KSe, cc = s / KSe;
        const int t = (taps >> (8 * tt)) & 0xff;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ck = 16 * cc + 8 * h + j;
            f[j] = tr ? w[((size_t)ck * Cin + m) * K + t] : w[((size_t)m * Cin + ck) * K + t];
        }
        op[v] = pack8(f);
    }
}

// ------------------------------------------------------------------------------------------------ conv forward / dgrad
struct ConvB {
    const u32x4* W;      // operand (see above): [S][2][M] 16-byte vectors, S = (Csrc/16)*KS
    const u32x4* src;    // blocked (N, Csrc/8, Lsrc)
    u32x4* out;          // blocked (N, M/8, Lrow)
    const u32x4* accum;  // blocked like out, added before rounding (data-gradient residual branch), or null
    float* stats;        // [gridDim.x][M][2] per-channel {sum, sum of squares} of the ROUNDED output, or null
    int N, M, Csrc, Lsrc, Ldst;
    int gmul, tapoff[3];       // source index of (output position l, tap t) = l*gmul + tapoff[t]
    int Lrow, ostride, ooff;   // output position l is stored at l*ostride + ooff of a row of Lrow vectors
    int P, numPT;
};

constexpr int kSC = 24;   // k-steps of weights staged in LDS at a time (48 KB): a multiple of every tap count 1, 2, 3

// Workgroup = 4 waves side by side over 256 positions x 64 output channels; a wave owns 64 positions x 64 channels
// (2 x 2 MFMA tiles of 32 x 32).  D[channel][position] += W[channel][k] * X[k][position], k-step = 16 input channels of
// one tap: A fragment (weights) from LDS, B fragment (8 channels of one position) one 16-byte global load.
template <int KS, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_b16_kernel(ConvB p) {
    constexpr int TM = 2, TN = 2;
    __shared__ u32x4 Wl[kSC * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 64;
    const int S = (p.Csrc >> 4) * KS;
    const int CBs = p.Csrc >> 3, CBo = p.M >> 3;
    const bool single = S <= kSC;   // the whole weight slab of this channel tile fits: staged once per workgroup
    bool staged = false;

    float st_s[STATS ? TM : 1][16], st_q[STATS ? TM : 1][16];
    if (STATS) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { st_s[i][e] = 0.f; st_q[i][e] = 0.f; }
    }

    for (int pt = blockIdx.x; pt < p.numPT; pt += gridDim.x) {
        const int p0 = pt * 256 + wave * 64;
        const u32x4* xp[TN][KS];
        unsigned xm[TN][KS];
        bool pok[TN];
        size_t obase[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pos = p0 + j * 32 + r;
            pok[j] = pos < p.P;
            const int n = pok[j] ? pos / p.Ldst : 0;
            const int l = pos - n * p.Ldst;
            obase[j] = (size_t)n * CBo * p.Lrow + (size_t)(l * p.ostride + p.ooff);
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const int ls = l * p.gmul + p.tapoff[t];
                const bool v = pok[j] && (unsigned)ls < (unsigned)p.Lsrc;
                xp[j][t] = p.src + ((size_t)n * CBs + h) * p.Lsrc + (v ? ls : 0);
                xm[j][t] = v ? 0xffffffffu : 0u;
            }
        }
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // K loop over 16-channel chunks (KS k-steps each).  The X fragments of chunk c+1 are requested from HBM/L2 before
        // the MFMAs of chunk c (two register sets, loop unrolled by two); weights are restaged every kSC k-steps.
        constexpr int CPS = kSC / KS;            // chunks per staged weight group
        const int nchunks = S / KS;
        // (MEASURED: a third register set - two chunks in flight ahead of the MFMAs - was 10-15 % SLOWER on every layer shape:
        // 255 VGPRs with spills; the 64/128-channel layers already run within 1.3x of a plain elementwise pass over the
        // same bytes, i.e. they are HBM-bound, not latency-bound)
        u32x4 bA[KS][TN], bB[KS][TN];
        auto xload = [&](u32x4 (&bf)[KS][TN], int c) {
            const size_t xoff = (size_t)c * 2 * p.Lsrc;   // 16 channels = 2 blocks further
#pragma unroll
            for (int t = 0; t < KS; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[t][j] = xp[j][t][xoff];
        };
        auto compute = [&](u32x4 (&bf)[KS][TN], int c) {
            if (c % CPS == 0 && !(single && staged)) {
                const int s0 = c * KS;
                const int sc = (S - s0) < kSC ? (S - s0) : kSC;
                __syncthreads();   // the previous group's readers are done
                for (int v = tid; v < sc * 128; v += 256) {
                    const int sh = v >> 6, row = v & 63;   // sh = local k-step * 2 + half
                    Wl[v] = p.W[((size_t)(s0 * 2 + sh)) * p.M + m0 + row];
                }
                __syncthreads();
                staged = true;
            }
            const int sl = (c % CPS) * KS;
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    u32x4 v = bf[t][j];
                    v.x &= xm[j][t]; v.y &= xm[j][t]; v.z &= xm[j][t]; v.w &= xm[j][t];
                    b[j] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = __builtin_bit_cast(bf16x8, Wl[((sl + t) * 2 + h) * 64 + 32 * i + r]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        };
        xload(bA, 0);
        for (int c = 0; c < nchunks; c += 2) {
            if (c + 1 < nchunks) xload(bB, c + 1);
            compute(bA, c);
            if (c + 2 < nchunks) xload(bA, c + 2);
            if (c + 1 < nchunks) compute(bB, c + 1);
        }

        // epilogue: accumulator register 4q+e of tile (i, j) = channel m0 + 32i + 8q + 4h + e at position j*32 + r
        u32x2 av[STATS ? 1 : TM][STATS ? 1 : TN][4];
        const bool has_acc = !STATS && p.accum != nullptr;   // (statistics and accumulation never come together)
        if (!STATS && has_acc) {   // all 16 loads in flight together
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const size_t o = (pok[j] ? obase[j] : 0) + (size_t)((m0 >> 3) + 4 * i + q) * p.Lrow;
                        av[STATS ? 0 : i][STATS ? 0 : j][q] = *(reinterpret_cast<const u32x2*>(p.accum + o) + h);
                    }
            // claim all 16 loads HERE (one wait while nothing but loads is in flight): left to the uses in the store loop below, the
            // compiler waits vmcnt(0) in front of every one of them - i.e. for the acknowledgement of the store before it
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        asm volatile("" : "+v"(av[STATS ? 0 : i][STATS ? 0 : j][q].x), "+v"(av[STATS ? 0 : i][STATS ? 0 : j][q].y));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = acc[i][j][4 * q + 0], v1 = acc[i][j][4 * q + 1], v2 = acc[i][j][4 * q + 2], v3 = acc[i][j][4 * q + 3];
                    const size_t o = obase[j] + (size_t)((m0 >> 3) + 4 * i + q) * p.Lrow;   // 16-byte vector index
                    u32x2* dst = reinterpret_cast<u32x2*>(p.out + o) + h;
                    if (!STATS && has_acc) {   // autograd's bf16 sum of two STORED branch gradients: this branch is rounded first
                        const u32x2 a2 = av[STATS ? 0 : i][STATS ? 0 : j][q];
                        const unsigned r01 = pack2(v0, v1), r23 = pack2(v2, v3);
                        v0 = bf_lo(r01) + bf_lo(a2.x); v1 = bf_hi(r01) + bf_hi(a2.x);
                        v2 = bf_lo(r23) + bf_lo(a2.y); v3 = bf_hi(r23) + bf_hi(a2.y);
                    }
                    u32x2 pk;
                    pk.x = pack2(v0, v1); pk.y = pack2(v2, v3);
                    if (pok[j]) *dst = pk;
                    if (STATS) {   // statistics of what was stored (positions beyond P hold exact zeros)
                        const float w0 = bf_lo(pk.x), w1 = bf_hi(pk.x), w2 = bf_lo(pk.y), w3 = bf_hi(pk.y);
                        st_s[i][4 * q + 0] += w0; st_q[i][4 * q + 0] = fmaf(w0, w0, st_q[i][4 * q + 0]);
                        st_s[i][4 * q + 1] += w1; st_q[i][4 * q + 1] = fmaf(w1, w1, st_q[i][4 * q + 1]);
                        st_s[i][4 * q + 2] += w2; st_q[i][4 * q + 2] = fmaf(w2, w2, st_q[i][4 * q + 2]);
                        st_s[i][4 * q + 3] += w3; st_q[i][4 * q + 3] = fmaf(w3, w3, st_q[i][4 * q + 3]);
                    }
                }
    }

    if (STATS) {
        // lanes of one half hold the same 32 channels at 32 different positions: butterfly over the 32 lanes, then the
        // four waves through LDS (the staging buffer is free once every wave has left the tile loop)
        float* red = reinterpret_cast<float*>(Wl);   // [4 waves][64 channels][2]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float s = st_s[i][e], q = st_q[i][e];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
                if (r == 0) {
                    const int ch = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                    red[(wave * 64 + ch) * 2 + 0] = s;
                    red[(wave * 64 + ch) * 2 + 1] = q;
                }
            }
        __syncthreads();
        if (tid < 64) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[(w * 64 + tid) * 2]; q += red[(w * 64 + tid) * 2 + 1]; }
            float* dst = p.stats + ((size_t)blockIdx.x * p.M + m0 + tid) * 2;
            dst[0] = s; dst[1] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3-tap stride-1 convs
// The 14 three-tap stride-1 pad-1 convolutions and their data gradients (the same kernel on the transposed operand) are 83 %
// of the student's MACs.  A bf16 MFMA retires 16 input channels x 32 x 32 outputs in 32 cycles, so a 32-channel stage of a
// 64 x 64 wave tile lasts ~0.35 us - far less than a global-memory round trip.  conv_b16_kernel above (X fragments straight
// from global memory, one register set ahead) therefore waits for memory every stage; here both operands travel by LDS-DMA
// (global_load_lds_dwordx4: no registers, asynchronous) into a THREE-stage LDS ring, each stage requested two stages before
// it is consumed, with counted s_waitcnt vmcnt + raw s_barrier (a __syncthreads would drain the DMA queue).
//   workgroup = 8 waves (4 position groups x 2 channel groups): 256 positions x 128 output channels, positions flattened over
//   samples (short rows waste nothing); a stage = 32 input channels: X [4 blocks][320 slots] (slot i = flattened source
//   position P0 - 1 + i; slots 258.. are never read) + W [12 k-step halves][128 channels], 16-byte vectors, 45 KB; 44 DMA
//   pieces of 1 KB per stage, 6 per wave (4 harmless duplicates keep the per-wave count uniform for the vmcnt arithmetic).
//   A tap that would cross a sample boundary (or the tensor's ends, where the staged slot holds clamped garbage) reads the
//   row's zero slot instead: the choice is a per-lane LDS offset computed once per tile (masking every fragment with four
//   v_and cost as much vector issue as the MFMAs themselves: 168 VALU per 24 MFMAs per wave in the PMC run).
constexpr int kXS = 321;                         // slots per block row: 320 written by DMA + one that stays ZERO
constexpr int kStgX = 4 * kXS;                   // vectors
constexpr int kStgV = kStgX + 12 * 128;          // vectors per stage (45 KB)

__device__ __forceinline__ void lds_dma16(const u32x4* gsrc, u32x4* lds_wave_base) {
    // each lane's 16 bytes land at lds_wave_base + lane * 16 (wave-uniform base in M0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <bool STATS>
__global__ __launch_bounds__(512, 2) void conv_b16s1_kernel(ConvB p) {
    __shared__ u32x4 ring[3 * kStgV];            // the ONLY LDS object: 135 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave & 3, wm = wave >> 2;
    const int m0 = blockIdx.y * 128;
    const int CBs = p.Csrc >> 3, CBo = p.M >> 3, L = p.Lsrc;
    const int nst = p.Csrc >> 5;

    float st_s[STATS ? 2 : 1][16], st_q[STATS ? 2 : 1][16];
    if (STATS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { st_s[i][e] = 0.f; st_q[i][e] = 0.f; }
    }
    if (tid < 12) {   // the zero slot of every block row of every stage (never written again)
        u32x4 z = {0u, 0u, 0u, 0u};
        ring[(tid >> 2) * kStgV + (tid & 3) * kXS + 320] = z;
    }
    __syncthreads();

    // ---- DMA bookkeeping, reduced to one 64-bit add per piece and stage (a first version recomputed the piece table every
    // stage: 128 scalar + 70 vector instructions per 24 MFMAs - the wave's issue slots, not the matrix pipe, set the pace).
    // Piece k of a wave (k = 0..5): k < 3 -> weight piece wave + 8k of 24; k >= 3 -> input piece (wave + 8(k-3)) mod 20 of 20
    // (block = piece / 5, 64-slot part = piece % 5; the four repeats rewrite the same bytes).
    const u32x4* wsrc[3];   // stage 0 source of the weight pieces (advance: 12 * M vectors per stage)
    int wdst[3], xdst[3], xblk[3], xpart[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int qw = wave + 8 * k;
        wsrc[k] = p.W + (size_t)(qw >> 1) * p.M + m0 + (qw & 1) * 64 + lane;
        wdst[k] = kStgX + (qw >> 1) * 128 + (qw & 1) * 64;
        int qx = wave + 8 * k;
        qx = qx >= 20 ? qx - 20 : qx;
        xblk[k] = qx / 5;
        xpart[k] = qx - 5 * xblk[k];
        xdst[k] = xblk[k] * kXS + xpart[k] * 64;
    }
    const size_t wstep = (size_t)12 * p.M, xstep = (size_t)4 * L;
    // source vector (stage 0) of the lane's slot of input piece k of the tile starting at flattened position P0; slots outside
    // the tensor are clamped to a valid address (their data is never read: zero-slot addressing below)
    auto piece_src = [&](int P0, int k) -> const u32x4* {
        int sp = P0 - 1 + xpart[k] * 64 + lane;
        sp = sp < 0 ? 0 : (sp >= p.P ? p.P - 1 : sp);
        const int n = sp / L;
        return p.src + ((size_t)n * CBs + xblk[k]) * L + (sp - n * L);
    };
    auto issue = [&](int c, int bufv, const u32x4* x0, const u32x4* x1, const u32x4* x2) {
#if defined(SSECG_ABLB_NODMA)      // timing experiment: nothing requested (the stage is multiplied from whatever the ring holds)
        return;
#endif
        u32x4* stg = ring + bufv;
#if defined(SSECG_ABLB_NODMAW)     // timing experiment: inputs only
        lds_dma16(x0 + c * xstep, stg + xdst[0]);
        lds_dma16(x1 + c * xstep, stg + xdst[1]);
        lds_dma16(x2 + c * xstep, stg + xdst[2]);
        lds_dma16(x0 + c * xstep, stg + xdst[0]);
        lds_dma16(x1 + c * xstep, stg + xdst[1]);
        lds_dma16(x2 + c * xstep, stg + xdst[2]);
        return;
#endif
        lds_dma16(wsrc[0] + c * wstep, stg + wdst[0]);
        lds_dma16(wsrc[1] + c * wstep, stg + wdst[1]);
        lds_dma16(wsrc[2] + c * wstep, stg + wdst[2]);
        lds_dma16(x0 + c * xstep, stg + xdst[0]);
        lds_dma16(x1 + c * xstep, stg + xdst[1]);
        lds_dma16(x2 + c * xstep, stg + xdst[2]);
    };

    // The stages of successive tiles form ONE stream through the ring: while a tile's last two stages are multiplied, the first
    // two stages of the workgroup's next tile are already travelling (only the epilogue is not overlapped with MFMAs).
    int pt = blockIdx.x;
    const u32x4 *c0 = p.src, *c1 = p.src, *c2 = p.src;   // current tile's input piece sources
    int bcur = 0, bn1 = kStgV, bn2 = 2 * kStgV;             // ring buffers (vector offsets): now, +1 stage, +2 stages
    if (pt < p.numPT) {
        c0 = piece_src(pt * 256, 0); c1 = piece_src(pt * 256, 1); c2 = piece_src(pt * 256, 2);
        issue(0, bcur, c0, c1, c2);
        issue(1, bn1, c0, c1, c2);
    }
    // per-lane fragment offsets inside a stage (vectors): weights: rows of this wave's channel group; inputs: see below
    const int wfo = kStgX + h * 128 + wm * 64 + r;
    bool after_epi = false;   // (uniform) an epilogue's stores may still be in flight
    const auto outR = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.N * CBo * p.Lrow * 16), 0x00020000);
    for (; pt < p.numPT; pt += gridDim.x) {
        const int P0 = pt * 256;
        const int ptn = pt + gridDim.x;
        const bool has_next = ptn < p.numPT;
        const u32x4 *n0 = p.src, *n1 = p.src, *n2 = p.src;
        if (has_next) { n0 = piece_src(ptn * 256, 0); n1 = piece_src(ptn * 256, 1); n2 = piece_src(ptn * 256, 2); }
        int xfo[2][3];   // input fragment offset (vectors from the stage base, block 0) per position tile and tap
        bool pok[2];
        size_t obase[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pos = P0 + wp * 64 + j * 32 + r;
            pok[j] = pos < p.P;
            const int n = pok[j] ? pos / L : 0;
            const int l = pos - n * L;
            obase[j] = (size_t)n * CBo * p.Lrow + (size_t)l;
#pragma unroll
            for (int t = 0; t < 3; ++t)   // a tap outside the sample reads the row's zero slot
                xfo[j][t] = h * kXS + ((pok[j] && (unsigned)(l + p.tapoff[t]) < (unsigned)L) ? wp * 64 + r + j * 32 + 1 + p.tapoff[t] : 320);
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int c = 0; c < nst; ++c) {
            // this stage has landed when at most the operations requested AFTER its pieces are outstanding: the six pieces of the
            // following stage and - for the first two stages of a tile that follows an epilogue, whose pieces were requested before
            // that epilogue - its 16 output stores (issued unconditionally, below, so that the count is exact).  Round 3 waited
            // vmcnt(6) here, i.e. for the write acknowledgement of every store of the previous tile before the first MFMA of the next.
            {
                const bool more = has_next || c + 1 < nst;
                const bool st = after_epi && c < 2;
                if (more && st) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
                else if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (st) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();            // ... for every wave's pieces; and the previous stage's readers are done
            if (c + 2 < nst) issue(c + 2, bn2, c0, c1, c2);          // into the buffer the previous stage was read from
            else if (has_next) issue(c + 2 - nst, bn2, n0, n1, n2);
            const u32x4* sb = ring + bcur;
            const u32x4* ww = sb + wfo;
            // six groups (16-channel half x tap) of 4 MFMAs; the fragments of group g+1 are read from LDS before the MFMAs of
            // group g are issued (two fragment sets); every offset below is a compile-time constant on a per-lane base
            u32x4 fa[2][2], fb[2][2];
            auto frag = [&](int g, int set) {
                const int cc = g / 3, t = g - 3 * cc;
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[set][j] = (sb + xfo[j][t])[2 * cc * kXS];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[set][i] = ww[(g * 2) * 128 + 32 * i];
            };
            frag(0, 0);
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const int set = g & 1;
                if (g + 1 < 6) frag(g + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the reads of group g+1 ahead of the MFMAs of group g
                bf16x8 a[2], b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(bf16x8, fb[set][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, fa[set][i]);
#if defined(SSECG_ABLB_NOMFMA)   // timing experiment: fragments read, nothing multiplied
                asm volatile("" :: "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
#else
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            const int tb = bcur; bcur = bn1; bn1 = bn2; bn2 = tb;   // rotate the ring
        }
        c0 = n0; c1 = n1; c2 = n2;


        // epilogue (as conv_b16_kernel): register 4q+e of tile (i, j) = channel m0 + 64 wm + 32i + 8q + 4h + e at position j*32 + r
        const int mb = (m0 >> 3) + 8 * wm;
        u32x2 av[STATS ? 1 : 2][STATS ? 1 : 2][4];
        const bool has_acc = !STATS && p.accum != nullptr;
        if (!STATS && has_acc) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const size_t o = (pok[j] ? obase[j] : 0) + (size_t)(mb + 4 * i + q) * p.Lrow;
                        av[STATS ? 0 : i][STATS ? 0 : j][q] = *(reinterpret_cast<const u32x2*>(p.accum + o) + h);
                    }
            // claim all 16 loads HERE (one wait while nothing but loads is in flight): left to the uses in the store loop below, the
            // compiler waits vmcnt(0) in front of every one of them - i.e. for the acknowledgement of the store before it
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        asm volatile("" : "+v"(av[STATS ? 0 : i][STATS ? 0 : j][q].x), "+v"(av[STATS ? 0 : i][STATS ? 0 : j][q].y));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = acc[i][j][4 * q + 0], v1 = acc[i][j][4 * q + 1], v2 = acc[i][j][4 * q + 2], v3 = acc[i][j][4 * q + 3];
                    const size_t o = obase[j] + (size_t)(mb + 4 * i + q) * p.Lrow;
                    u32x2* dst = reinterpret_cast<u32x2*>(p.out + o) + h;
                    if (!STATS && has_acc) {   // autograd's bf16 sum of two STORED branch gradients: this branch is rounded first
                        const u32x2 a2 = av[STATS ? 0 : i][STATS ? 0 : j][q];
                        const unsigned r01 = pack2(v0, v1), r23 = pack2(v2, v3);
                        v0 = bf_lo(r01) + bf_lo(a2.x); v1 = bf_hi(r01) + bf_hi(a2.x);
                        v2 = bf_lo(r23) + bf_lo(a2.y); v3 = bf_hi(r23) + bf_hi(a2.y);
                    }
                    u32x2 pk;
                    pk.x = pack2(v0, v1); pk.y = pack2(v2, v3);
#if defined(SSECG_ABLB_NOSTORE)   // timing experiment: results packed, not written
                    asm volatile("" :: "v"(pk.x), "v"(pk.y), "v"(dst));
#else
                    // always issued (an out-of-range offset where the position lies outside the tensor): the stage waits count it
                    (void)dst;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, pk), outR,
                                                          amp_oob_if((unsigned)(obase[j] * 16u) + 8u * (unsigned)h, !pok[j]),
                                                          (unsigned)((mb + 4 * i + q) * p.Lrow) * 16u, 0);
#endif
                    if (STATS) {
                        const float w0 = bf_lo(pk.x), w1 = bf_hi(pk.x), w2 = bf_lo(pk.y), w3 = bf_hi(pk.y);
                        st_s[i][4 * q + 0] += w0; st_q[i][4 * q + 0] = fmaf(w0, w0, st_q[i][4 * q + 0]);
                        st_s[i][4 * q + 1] += w1; st_q[i][4 * q + 1] = fmaf(w1, w1, st_q[i][4 * q + 1]);
                        st_s[i][4 * q + 2] += w2; st_q[i][4 * q + 2] = fmaf(w2, w2, st_q[i][4 * q + 2]);
                        st_s[i][4 * q + 3] += w3; st_q[i][4 * q + 3] = fmaf(w3, w3, st_q[i][4 * q + 3]);
                    }
                }
        after_epi = true;
    }

    if (STATS) {
        float* red = reinterpret_cast<float*>(ring);   // [4 position groups][128 channels][2]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float s = st_s[i][e], q = st_q[i][e];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
                if (r == 0) {
                    const int ch = 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                    red[(wp * 128 + ch) * 2 + 0] = s;
                    red[(wp * 128 + ch) * 2 + 1] = q;
                }
            }
        __syncthreads();
        if (tid < 128) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[(w * 128 + tid) * 2]; q += red[(w * 128 + tid) * 2 + 1]; }
            float* dst = p.stats + ((size_t)blockIdx.x * p.M + m0 + tid) * 2;
            dst[0] = s; dst[1] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm on blocked bf16
// y = [relu]( x * a[c] + b[c] [+ residual] ),  a = gamma*invstd, b = beta - mean*a   (fp32 arithmetic, rounded once; with a
// residual the affine result is rounded before the add, as the reference's bf16 BatchNorm output is - round 5)
__global__ __launch_bounds__(256) void bn_apply_fwd_b16_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int N, int C,
                                                               int L, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const u32x4* __restrict__ residual,
                                                               int relu, unsigned char* __restrict__ mask_bytes,
                                                               const float* __restrict__ rmean, const float* __restrict__ rinvstd,
                                                               const float* __restrict__ rgamma, const float* __restrict__ rbeta) {
    __shared__ __attribute__((aligned(16))) float sa[512], sb[512], sra[512], srb[512];
    // rmean != NULL (round 6): the residual is the RAW output of the block's 1x1 downsample convolution; ITS BatchNorm is applied
    // while it is read and rounded to bf16 first, as the stored identity tensor was - which is then never written
    const bool resbn = rmean != nullptr;
    for (int c = threadIdx.x; c < C; c += 256) {
        if (resbn) {
            const float ra = rgamma[c] * rinvstd[c];
            sra[c] = ra;
            srb[c] = rbeta[c] - rmean[c] * ra;
        }
        if (mean != nullptr) {
            const float a = gamma[c] * invstd[c];
            sa[c] = a;
            sb[c] = beta[c] - mean[c] * a;
        } else {   // eval mode (ABI 11): gamma / beta are the folded scale / shift of the running statistics (ssecg_bn_fold)
            sa[c] = gamma[c];
            sb[c] = beta[c];
        }
    }
    __syncthreads();
    // flat vector index -> (row = n*CB + cb, l) WITHOUT a division per element (a 64-bit divide by a runtime L is ~100
    // instructions - more than the arithmetic of the 8 elements): one 32-bit divide up front, then carries
    const unsigned CB = (unsigned)(C >> 3), Lu = (unsigned)L;
    const unsigned total = (unsigned)N * CB * Lu;   // < 2^31 (launcher-checked)
    const unsigned S = gridDim.x * 256u;
    const unsigned dq = S / Lu, dr = S - dq * Lu, dcb = dq % CB;
    unsigned idx = blockIdx.x * 256u + threadIdx.x;
    unsigned row = idx / Lu, l = idx - row * Lu, cb = row % CB;
    for (; idx < total; idx += S) {
        float f[8], g[8];
        unpack8(x[idx], f);
        if (residual != nullptr) {
            unpack8(residual[idx], g);
            if (resbn) {
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = bf_lo(pack2(fmaf(g[j], sra[8 * cb + j], srb[8 * cb + j]), 0.f));
            }
        }
        const float4 a0 = *reinterpret_cast<const float4*>(sa + 8 * cb), a1 = *reinterpret_cast<const float4*>(sa + 8 * cb + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(sb + 8 * cb), b1 = *reinterpret_cast<const float4*>(sb + 8 * cb + 4);
        const float aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = fmaf(f[j], aa[j], bb[j]);
            if (residual != nullptr) v = bf_lo(pack2(v, 0.f)) + g[j];   // autocast stores BatchNorm's bf16 output, then ``out += identity``
            if (relu) v = fmaxf(v, 0.f);
            f[j] = v;
        }
        y[idx] = pack8(f);
        if (mask_bytes != nullptr) {   // the ReLU mask of this vector's 8 channels, for the backward passes (1/16 of y's bytes)
            unsigned m = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) m |= (f[j] > 0.f ? 1u : 0u) << j;
            mask_bytes[idx] = (unsigned char)m;
        }
        l += dr;
        const unsigned carry = l >= Lu ? 1u : 0u;
        l -= carry * Lu;
        cb += dcb + carry;
        cb -= cb >= CB ? CB : 0u;
    }
}

__device__ __forceinline__ bool bf_pos(unsigned short bits) { return (bits & 0x8000u) == 0 && (bits & 0x7fffu) != 0; }

// mode 0: no ReLU (dz = dy); 1: ReLU mask from the saved output y; 2: mask recomputed from x (z = x*a + b > 0);
// 3: `y` points to the byte-per-vector mask bn_apply_fwd_b16_kernel wrote (bit j = channel 8*cb + j passed)
// partial[blockIdx.x][C][2] = { sum dz, sum dz * xhat } over this workgroup's positions of channel block blockIdx.y
// PAIR (round 6): a SECOND BatchNorm behind the same masked gradient - the 1x1 downsample branch beside the block's bn2 (modes 1 / 3: the
// block's final ReLU mask) - reduced / applied in the same pass: dy and the mask are read once for the two (csrc/elementwise.hip:
// BnPair).  The sums and the apply arithmetic carry explicit roundings so that pair and single launches agree bit for bit.
struct BnPairB {
    const u32x4* x;        // the second BatchNorm's input (the raw 1x1 output, blocked bf16), or nullptr
    const float* mean;
    const float* invstd;
    const float* gamma;    // apply only
    const double* sums;    // apply only
    float* partial;        // reduce: its partial rows
    u32x4* dx;             // apply: its input gradient
};

template <bool APPLY, bool PAIR = false>
__global__ __launch_bounds__(256) void bn_bwd_b16_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ y,
                                                         const u32x4* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int mode, int N, int C, int L,
                                                         float* __restrict__ partial, const double* __restrict__ sums,
                                                         double count, u32x4* __restrict__ dx, u32x4* __restrict__ dz_out,
                                                         BnPairB pb = BnPairB{}) {
    const int cb = blockIdx.y, CB = C >> 3;
    float mu[8], is[8], a[8], b[8], k1[8], k2[8];
    float mu2[PAIR ? 8 : 1], is2[PAIR ? 8 : 1], a2[PAIR ? 8 : 1], k1b[PAIR ? 8 : 1], k2b[PAIR ? 8 : 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 8 * cb + j;
        mu[j] = mean[c]; is[j] = invstd[c];
        a[j] = gamma[c] * is[j];
        b[j] = (beta != nullptr ? beta[c] : 0.f) - mu[j] * a[j];   // (mode 2 only: as the forward pass forms it)
        if (APPLY) {
            k1[j] = (float)(sums[2 * c] / count);
            k2[j] = (float)(sums[2 * c + 1] / count);
        }
        if (PAIR) {
            mu2[j] = pb.mean[c]; is2[j] = pb.invstd[c];
            if (APPLY) {
                a2[j] = __fmul_rn(pb.gamma[c], is2[j]);
                k1b[j] = (float)(pb.sums[2 * c] / count);
                k2b[j] = (float)(pb.sums[2 * c + 1] / count);
            }
        }
    }
    float s1[8], s2[8], s2b[PAIR ? 8 : 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; if (PAIR) s2b[j] = 0.f; }
    // position index -> (n, l) by carries, not by a 64-bit divide per element (see bn_apply_fwd_b16_kernel)
    const unsigned NL = (unsigned)N * (unsigned)L, Lu = (unsigned)L;
    const unsigned S = gridDim.x * 256u;
    const unsigned dq = S / Lu, dr = S - dq * Lu;
    unsigned pidx = blockIdx.x * 256u + threadIdx.x;
    unsigned n = pidx / Lu, l = pidx - n * Lu;
    for (; pidx < NL; pidx += S, n += dq, l += dr) {
        if (l >= Lu) { l -= Lu; ++n; }
        const unsigned off = (n * (unsigned)CB + (unsigned)cb) * Lu + l;
        float g[8], xv[8];
        unpack8(dy[off], g);
        const u32x4 xr = x[off];
        unpack8(xr, xv);
        if (mode == 1) {
            const u32x4 yr = y[off];
            const unsigned w[4] = {yr.x, yr.y, yr.z, yr.w};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!bf_pos((unsigned short)(w[j >> 1] >> (16 * (j & 1))))) g[j] = 0.f;
        } else if (mode == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!(fmaf(xv[j], a[j], b[j]) > 0.f)) g[j] = 0.f;
        } else if (mode == 3) {
            const unsigned m = reinterpret_cast<const unsigned char*>(y)[off];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (!((m >> j) & 1u)) g[j] = 0.f;
        }
        float xb[8];
        if (PAIR) unpack8(pb.x[off], xb);
        if (APPLY) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = __fmul_rn(__fsub_rn(xv[j], mu[j]), is[j]);
                o[j] = __fmul_rn(a[j], fmaf(-xh, k2[j], __fsub_rn(g[j], k1[j])));
            }
            dx[off] = pack8(o);
            if (dz_out != nullptr) dz_out[off] = pack8(g);
            if (PAIR) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = __fmul_rn(__fsub_rn(xb[j], mu2[j]), is2[j]);
                    o[j] = __fmul_rn(a2[j], fmaf(-xh, k2b[j], __fsub_rn(g[j], k1b[j])));
                }
                pb.dx[off] = pack8(o);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = __fmul_rn(__fsub_rn(xv[j], mu[j]), is[j]);
                s1[j] = __fadd_rn(s1[j], g[j]);
                s2[j] = fmaf(g[j], xh, s2[j]);
                if (PAIR) s2b[j] = fmaf(g[j], __fmul_rn(__fsub_rn(xb[j], mu2[j]), is2[j]), s2b[j]);
            }
        }
    }
    if (!APPLY) {
        __shared__ float red[4][16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
        }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[wv][2 * j] = s1[j]; red[wv][2 * j + 1] = s2[j]; }
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            partial[((size_t)blockIdx.x * C + 8 * cb + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)] = v;
        }
        if (PAIR) {   // the second BatchNorm's rows: the same sum of dz (already reduced over the wave in s1), its own sum of dz * xhat
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s2b[j] += __shfl_xor(s2b[j], o, 64);
            }
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { red[wv][2 * j] = s1[j]; red[wv][2 * j + 1] = s2b[j]; }
            }
            __syncthreads();
            if (threadIdx.x < 16) {
                const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
                pb.partial[((size_t)blockIdx.x * C + 8 * cb + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dw[co][ci][t] = sum_{n,l} dy[n][co][l] * x[n][ci][l*stride + t - pad]: both operands are needed with POSITIONS on the
// contracted axis.  A stage = 32 output positions of one sample; the dy tile [32 pos][128 co] and the x tile
// [rows][128 ci] are copied to LDS as they lie in HBM (one 16-byte vector = 8 channels of a position) in the XOR-swizzled
// 256-byte-row image of cdna_hip_programming.md T10(b); ds_read_b64_tr_b16 then hands every lane 4 consecutive
// positions of ITS channel - the MFMA operand, transposed for free.  Workgroup = 4 waves as 2 x 2 over a 128 (co) x 128
// (ci) tile, all taps; position slabs are summed in a fixed order by wgrad_b16_reduce_kernel (reproducible).
struct WgB {
    const u32x4* dy;   // blocked (N, Cout/8, Ldy)
    const u32x4* x;    // blocked (N, Cin/8, Lx)
    float* ws;         // [Z][Cout][KS*Cin] slabs (column = t*Cin + ci)
    int N, Cout, Cin, Ldy, Lx, stride, pad;
    int MT, JT, Z, stages_per_sample, total_stages, stages_per_slab;
};

__device__ __forceinline__ int img_off(int row, int ch) {   // byte offset of 16-byte chunk ch (0..15) of a 256-byte row
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

constexpr int kSP = 64;   // output positions per stage (4 MFMA k-steps)

template <int KS, int TMW, int TJW>   // workgroup tile = (64*TMW co) x (64*TJW ci): 2 x 2 waves of TMW x TJW MFMA tiles each
__global__ __launch_bounds__(256, 1) void conv_wgrad_b16_kernel(WgB p) {
    constexpr int XR = 2 * kSP + 2;                      // x rows per stage: kSP*stride + KS - 1 <= 130
    constexpr int IMG = (kSP + XR) * 256;                // one stage's dy + x images (256-byte rows)
    constexpr int NDY = (kSP * 8 * TMW + 255) / 256;     // 16-byte vectors per thread per stage
    constexpr int NX = (XR * 8 * TJW + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * IMG];   // double-buffered (2 x 48.5 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tiles = p.MT * p.JT;
    // XCD-aware order as in the fp32 kernel: all tiles of one slab on one XCD (workgroup b runs on XCD b % 8)
    const int slot = blockIdx.x >> 3;
    const int zslab = (slot / tiles) * 8 + (blockIdx.x & 7);
    if (zslab >= p.Z) return;
    const int tile = slot % tiles;
    const int m0 = (tile / p.JT) * (64 * TMW), j0 = (tile % p.JT) * (64 * TJW);
    const int CBo = p.Cout >> 3, CBi = p.Cin >> 3;
    const int xrows = kSP * p.stride + KS - 1;

    f32x16 acc[TMW][TJW][KS];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TJW; ++j)
#pragma unroll
            for (int t = 0; t < KS; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][t][e] = 0.f;

    // transposed-read addressing: the 16-lane group g = lane >> 4 covers channels 16*(g & 1) .. +15 of a 32-channel tile
    // and the lane half h = g >> 1 the positions 8h .. 8h+7 of a 16-position k-step; lane 4q+pp of the group supplies
    // the address of row (position) q, columns 4pp..4pp+3: chunk (pp >> 1), byte 8*(pp & 1)
    const int gl = lane & 15, q4 = gl >> 2, pp = gl & 3;
    const int chunk_in_tile = 2 * ((lane >> 4) & 1) + (pp >> 1);   // 16-byte chunk within the 32-channel tile (0..3)

    // staging: global -> registers (requested before the MFMAs of the current stage) -> LDS image of the other buffer
    // (the loaded values are NOT touched here: any use would make the compiler wait for each load right after issuing it;
    // the zero-fill of padding / out-of-range rows is applied when the registers are written to LDS, after the MFMAs)
    u32x4 rdy[NDY], rx[NX];
    unsigned okbits = 0;
    auto gload = [&](int s) {
        const int n = s / p.stages_per_sample;
        const int l0 = (s - n * p.stages_per_sample) * kSP;
        okbits = 0;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int v = tid + 256 * k;
            const int row = v / (8 * TMW), ch = v % (8 * TMW);
            const int l = l0 + row;
            const bool ok = (v < kSP * 8 * TMW) && l < p.Ldy;
            rdy[k] = p.dy[((size_t)n * CBo + (m0 >> 3) + ch) * p.Ldy + (ok ? l : 0)];
            okbits |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int v = tid + 256 * k;
            const int row = v / (8 * TJW), ch = v % (8 * TJW);
            const int lx = l0 * p.stride - p.pad + row;
            const bool ok = (row < xrows) && (unsigned)lx < (unsigned)p.Lx;
            rx[k] = p.x[((size_t)n * CBi + (j0 >> 3) + ch) * p.Lx + (ok ? lx : 0)];
            okbits |= (ok ? 1u : 0u) << (16 + k);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* const dyI = sm + buf * IMG;
        unsigned char* const xI = dyI + kSP * 256;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int v = tid + 256 * k;
            const unsigned m = ((okbits >> k) & 1u) ? 0xffffffffu : 0u;
            u32x4 val = rdy[k];
            val.x &= m; val.y &= m; val.z &= m; val.w &= m;
            if (v < kSP * 8 * TMW) *reinterpret_cast<u32x4*>(dyI + img_off(v / (8 * TMW), v % (8 * TMW))) = val;
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int v = tid + 256 * k;
            const unsigned m = ((okbits >> (16 + k)) & 1u) ? 0xffffffffu : 0u;
            u32x4 val = rx[k];
            val.x &= m; val.y &= m; val.z &= m; val.w &= m;
            if (v / (8 * TJW) < xrows) *reinterpret_cast<u32x4*>(xI + img_off(v / (8 * TJW), v % (8 * TJW))) = val;
        }
    };

    const int s_begin = zslab * p.stages_per_slab;
    int s_end = s_begin + p.stages_per_slab;
    if (s_end > p.total_stages) s_end = p.total_stages;
    if (s_begin < s_end) {
        gload(s_begin);
        lstore(0);
    }
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) gload(s + 1);
        const unsigned char* const dyI = sm + buf * IMG;
        const unsigned char* const xI = dyI + kSP * 256;
#pragma unroll
        for (int ks = 0; ks < kSP / 16; ++ks) {
            // A: dy[co][pos], positions 16ks + 8h + (0..7)
            bf16x8 a[TMW];
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int ch = (wm * 32 * TMW + i * 32) / 8 + chunk_in_tile;
                const int row0 = 16 * ks + 8 * h + q4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(dyI + img_off(row0, ch) + 8 * (pp & 1)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(dyI + img_off(row0 + 4, ch) + 8 * (pp & 1)));
                u32x4 v;
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                v.x = l2.x; v.y = l2.y; v.z = h2.x; v.w = h2.y;
                a[i] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                bf16x8 b[TJW];
#pragma unroll
                for (int j = 0; j < TJW; ++j) {
                    const int ch = (wj * 32 * TJW + j * 32) / 8 + chunk_in_tile;
                    const int k0 = 16 * ks + 8 * h + q4;              // output position within the stage
                    const int row0 = k0 * p.stride + t;                // x row of (position, tap)
                    const int row1 = (k0 + 4) * p.stride + t;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(xI + img_off(row0, ch) + 8 * (pp & 1)));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(xI + img_off(row1, ch) + 8 * (pp & 1)));
                    u32x4 v;
                    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    v.x = l2.x; v.y = l2.y; v.z = h2.x; v.w = h2.y;
                    b[j] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < TJW; ++j)
                        acc[i][j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j][t], 0, 0, 0);
            }
        }
        if (s + 1 < s_end) lstore(buf ^ 1);
        __syncthreads();   // the other buffer is complete, and every wave is done reading this one
    }
    // slab store: ws[z][co][t*Cin + ci]; accumulator: column (lane & 31) = ci, rows = co
    const int J = KS * p.Cin;
    float* ws = p.ws + (size_t)zslab * p.Cout * J;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TJW; ++j)
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const int ci = j0 + wj * 32 * TJW + j * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = m0 + wm * 32 * TMW + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    ws[(size_t)co * J + t * p.Cin + ci] = acc[i][j][t][e];
                }
            }
}

// The same weight gradient for the 3-tap stride-1 convolutions with 128 x 128 channel tiles (layers 2-4 and the head),
// operands by LDS-DMA into a FOUR-stage ring (the kernel above keeps one 64-position stage in registers ahead: 48 MFMAs
// = 0.7 us of cover for a global-memory round trip).  The swizzled image is produced by the DMA itself: LDS slot (row,
// chunk') of a 1 KB piece (4 rows x 16 chunks) receives the global chunk chunk' ^ swz(row) - the per-lane SOURCE address
// carries the permutation, the destination stays linear.  Rows outside the sample (the stage's tail, the convolution's
// padding) load 16 zero bytes from a constant in device memory.  33 pieces per stage (16 dy + 17 x), 9 per wave (3 harmless
// repeats); counted s_waitcnt vmcnt + raw s_barrier as in conv_b16s1_kernel.
__device__ const u32x4 g_zero16 = {0u, 0u, 0u, 0u};

template <int KS, int STRIDE, int TMW, int TJW>
__global__ __launch_bounds__(256, 1) void conv_wgrad_b16s1_kernel(WgB p) {
    constexpr int XROWS = kSP * STRIDE + 4;                // multiple of 4; rows 0 .. kSP*STRIDE + KS - 2 are read
    constexpr int NPD = kSP / 4, NPX = XROWS / 4;          // 1 KB DMA pieces per stage: dy image, x image
    constexpr int NP = NPD + NPX, PPW = (NP + 3) / 4;      // pieces per wave and stage (the last few repeat piece 0..)
    constexpr int IMG = (kSP + XROWS) * 256;               // 33 KB (stride 1) / 49 KB (stride 2) per stage
    constexpr int RING = STRIDE == 1 ? 4 : 3;
    static_assert(4 * PPW - NP < NP && PPW * (RING - 1) < 64, "piece arithmetic");
    __shared__ __attribute__((aligned(16))) unsigned char sm[RING * IMG];   // the ONLY LDS object (132 / 147 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tiles = p.MT * p.JT;
    const int slot = blockIdx.x >> 3;
    const int zslab = (slot / tiles) * 8 + (blockIdx.x & 7);
    if (zslab >= p.Z) return;
    const int tile = slot % tiles;
    const int m0 = (tile / p.JT) * (64 * TMW), j0 = (tile % p.JT) * (64 * TJW);
    const int CBo = p.Cout >> 3, CBi = p.Cin >> 3;

    f32x16 acc[TMW][TJW][KS];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TJW; ++j)
#pragma unroll
            for (int t = 0; t < KS; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][t][e] = 0.f;

    const int gl = lane & 15, q4 = gl >> 2, pp = gl & 3;
    const int chunk_in_tile = 2 * ((lane >> 4) & 1) + (pp >> 1);

    // DMA pieces of this wave: piece q = wave + 4k (k < PPW), q >= NP repeats q - NP; q < NPD: dy rows 4q..4q+3; else x rows
    // 4(q-NPD)..+3.  Per lane: row-in-piece rl, chunk position cp; source chunk = cp ^ (rl << 2) ^ (piece & 3); a chunk
    // beyond the tile's channel blocks (64-channel tiles use half of every 256-byte row) or a row outside the sample loads zeros.
    const int rl = lane >> 4, cp = lane & 15;
    const int chA = cp ^ (rl << 2);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm;
    auto issue = [&](int s, int buf) {
        const int n = s / p.stages_per_sample;
        const int l0 = (s - n * p.stages_per_sample) * kSP;
        const u32x4* dyb = p.dy + ((size_t)n * CBo + (m0 >> 3)) * p.Ldy;
        const u32x4* xb = p.x + ((size_t)n * CBi + (j0 >> 3)) * p.Lx;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            int q = wave + 4 * k;
            q = q >= NP ? q - NP : q;
            const bool is_dy = q < NPD;
            const int i = is_dy ? q : q - NPD;                                   // piece index inside its image
            const int ch = chA ^ (i & 3);
            constexpr int PAD = KS == 3 ? 1 : 0;                                 // (launcher-checked)
            const int l = (is_dy ? l0 : l0 * STRIDE - PAD) + 4 * i + rl;         // position of this lane's row
            const int Lr = is_dy ? p.Ldy : p.Lx;
            const u32x4* src = (is_dy ? dyb : xb) + (size_t)ch * Lr + l;
            bool ok = (unsigned)l < (unsigned)Lr;
            if (TMW == 1 || TJW == 1) ok = ok && ch < 8 * (is_dy ? TMW : TJW);   // (128-channel tiles use every chunk)
            src = ok ? src : &g_zero16;
            lds_dma16_asm(src, __builtin_amdgcn_readfirstlane(lds0 + buf * IMG + (is_dy ? 0 : kSP * 256) + i * 1024));
        }
    };

    const int s_begin = zslab * p.stages_per_slab;
    int s_end = s_begin + p.stages_per_slab;
    if (s_end > p.total_stages) s_end = p.total_stages;
    const int nst = s_end - s_begin;
#pragma unroll
    for (int a = 0; a < RING - 1; ++a)
        if (a < nst) issue(s_begin + a, a);
    for (int c = 0; c < nst; ++c) {
        // stage c has landed when at most the pieces of the stages requested after it are outstanding (PPW per stage)
        const int ahead = nst - 1 - c < RING - 2 ? nst - 1 - c : RING - 2;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // ... for every wave's pieces; and stage c-1's readers are done
        if (c + RING - 1 < nst) issue(s_begin + c + RING - 1, (c + RING - 1) % RING);   // into stage c-1's buffer
        const unsigned char* const dyI = sm + (c % RING) * IMG;
        const unsigned char* const xI = dyI + kSP * 256;
#pragma unroll
        for (int ks = 0; ks < kSP / 16; ++ks) {
            bf16x8 a[TMW];
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int ch = (wm * 32 * TMW + i * 32) / 8 + chunk_in_tile;
                const int row0 = 16 * ks + 8 * h + q4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(dyI + img_off(row0, ch) + 8 * (pp & 1)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(dyI + img_off(row0 + 4, ch) + 8 * (pp & 1)));
                u32x4 v;
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                v.x = l2.x; v.y = l2.y; v.z = h2.x; v.w = h2.y;
                a[i] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                bf16x8 b[TJW];
#pragma unroll
                for (int j = 0; j < TJW; ++j) {
                    const int ch = (wj * 32 * TJW + j * 32) / 8 + chunk_in_tile;
                    const int k0 = 16 * ks + 8 * h + q4;               // output position within the stage
                    const int row0 = k0 * STRIDE + t;                  // x row of (position, tap)
                    const int row1 = (k0 + 4) * STRIDE + t;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(xI + img_off(row0, ch) + 8 * (pp & 1)));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(xI + img_off(row1, ch) + 8 * (pp & 1)));
                    u32x4 v;
                    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    v.x = l2.x; v.y = l2.y; v.z = h2.x; v.w = h2.y;
                    b[j] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < TJW; ++j)
                        acc[i][j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j][t], 0, 0, 0);
            }
        }
    }
    const int J = KS * p.Cin;
    float* ws = p.ws + (size_t)zslab * p.Cout * J;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TJW; ++j)
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const int ci = j0 + wj * 32 * TJW + j * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = m0 + wm * 32 * TMW + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    ws[(size_t)co * J + t * p.Cin + ci] = acc[i][j][t][e];
                }
            }
}

// dw[co][ci][t] = sum_z ws[z][co][t*Cin + ci]: 64 elements per workgroup x ZL slab lanes (lane zl sums slabs zl, zl+ZL, ...
// with two loads in flight), combined in a fixed order through LDS -> reproducible.  ZL = 16 where a tile has many slabs (the 64- and
// 128-channel layers: 128 - 256 slabs were 32 dependent round trips per thread with 4 lanes)
template <int ZL>
__global__ __launch_bounds__(64 * ZL) void wgrad_b16_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Z, int Cout,
                                                                  int Cin, int KS) {
    __shared__ float part[ZL][64];
    const size_t total = (size_t)Cout * Cin * KS;
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < total; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + el;   // index into ws rows: co*(KS*Cin) + t*Cin + ci
        float s0 = 0.f, s1 = 0.f;
        if (e < total) {
            int z = zl;
            for (; z + ZL < Z; z += 2 * ZL) {
                s0 += ws[(size_t)z * total + e];
                s1 += ws[(size_t)(z + ZL) * total + e];
            }
            if (z < Z) s0 += ws[(size_t)z * total + e];
        }
        part[zl][el] = s0 + s1;
        __syncthreads();
        if (zl == 0 && e < total) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < ZL; q += 4) v += (part[q][el] + part[q + 1][el]) + (part[q + 2][el] + part[q + 3][el]);
            const int ci = (int)(e % Cin);
            const size_t rest = e / Cin;
            const int t = (int)(rest % KS);
            const size_t co = rest / KS;
            dw[(co * Cin + ci) * KS + t] = v;
        }
        __syncthreads();
    }
}

inline int grid_for(size_t total, int per_block = 256, int cap = kNumCU * 8) {
    size_t b = (total + per_block - 1) / per_block;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

int ssecg_amp_planar_to_blocked(const float* x, void* y, int N, int C, int L, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || (C & 7) || L <= 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(cvt_planar_to_blocked_kernel, dim3(grid_for((size_t)N * (C >> 3) * L)), dim3(256), 0, (hipStream_t)stream, x,
                       (u32x4*)y, N, C, L);
    return (int)hipGetLastError();
}

int ssecg_amp_blocked_to_planar(const void* x, float* y, int N, int C, int L, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || (C & 7) || L <= 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(cvt_blocked_to_planar_kernel, dim3(grid_for((size_t)N * (C >> 3) * L)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)x, y, N, C, L);
    return (int)hipGetLastError();
}

int ssecg_amp_weight_operand_multi(const int64_t* table, int ntensors, int max_vectors, void* stream) {
    if (!table || ntensors <= 0 || max_vectors <= 0) return SSECG_E_INVAL;
    int gx = (max_vectors + 255) / 256;
    if (gx > 512) gx = 512;   // (64 until round 4: the gathers are latency-bound, the layer4 operands alone are 384 blocks of work)
    hipLaunchKernelGGL(weight_operand_multi_kernel, dim3(gx, ntensors), dim3(256), 0, (hipStream_t)stream, table);
    return (int)hipGetLastError();
}

static int ring_parts(int N, int Ldst, int M) {
    const long long P = (long long)N * Ldst;
    const int numPT = (int)((P + 255) / 256);
    const int MT = M / 64;
    int g = ((kNumCU * 2) / MT) & ~7;   // a multiple of 8: the MT channel tiles of one position tile share an XCD
    if (g < 8) g = 8;
    return numPT < g ? numPT : g;
}

int ssecg_amp_conv_parts(int N, int Csrc, int Lsrc, int M, int Ldst, int ntaps, int gmul, int tapoff0, int tapoff1, int tapoff2,
                         int Lrow, int ostride, int ooff) {
    if (N <= 0 || Ldst <= 0 || M <= 0 || (M & 63)) return SSECG_E_INVAL;
    const int ws = ws_rows(N, Csrc, Lsrc, M, Ldst, ntaps, gmul, tapoff0, tapoff1, tapoff2, Lrow, ostride, ooff, false, true);
    return ws > 0 ? ws : ring_parts(N, Ldst, M);
}

int ssecg_amp_conv(const void* src, const void* w_operand, void* out, int N, int Csrc, int Lsrc, int M, int Ldst, int ntaps,
                   int gmul, int tapoff0, int tapoff1, int tapoff2, int Lrow, int ostride, int ooff, const void* accumulate,
                   float* stats, int stats_parts, void* stream) {
    if (!src || !w_operand || !out || N <= 0 || Csrc <= 0 || (Csrc & 15) || M <= 0 || (M & 63) || Lsrc <= 0 || Ldst <= 0 ||
        ntaps < 1 || ntaps > 3 || gmul < 1 || Lrow <= 0 || ostride < 1 || ooff < 0 || (long long)(Ldst - 1) * ostride + ooff >= Lrow)
        return SSECG_E_INVAL;
    if ((long long)N * Ldst >= (1ll << 31)) return SSECG_E_INVAL;
    if (stats != nullptr && accumulate != nullptr) return SSECG_E_INVAL;
    ConvB p;
    p.W = (const u32x4*)w_operand; p.src = (const u32x4*)src; p.out = (u32x4*)out; p.accum = (const u32x4*)accumulate;
    p.stats = stats;
    p.N = N; p.M = M; p.Csrc = Csrc; p.Lsrc = Lsrc; p.Ldst = Ldst;
    p.gmul = gmul; p.tapoff[0] = tapoff0; p.tapoff[1] = tapoff1; p.tapoff[2] = tapoff2;
    p.Lrow = Lrow; p.ostride = ostride; p.ooff = ooff;
    p.P = N * Ldst; p.numPT = (p.P + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    // weights-stationary kernel (amp_ws.hip): the 3-tap stride-1 convs, the stride-2 convs and the 1x1 downsample with its gradient
    const int ws = ws_rows(N, Csrc, Lsrc, M, Ldst, ntaps, gmul, tapoff0, tapoff1, tapoff2, Lrow, ostride, ooff, accumulate != nullptr,
                            stats != nullptr);
    if (ws > 0) {
        if (stats != nullptr && stats_parts != ws) return SSECG_E_WORKSPACE;   // every row handed over is written: exact count
        return ws_launch(src, w_operand, out, N, Csrc, Lsrc, M, Ldst, ntaps, gmul, tapoff0, tapoff1, tapoff2, Lrow, ostride, ooff, stats, st);
    }
    const int G = ring_parts(N, Ldst, M);
    if (stats != nullptr && stats_parts != G) return SSECG_E_WORKSPACE;
    const bool taps_s1 = (tapoff0 == -1 && tapoff1 == 0 && tapoff2 == 1) || (tapoff0 == 1 && tapoff1 == 0 && tapoff2 == -1);
    if (ntaps == 3 && gmul == 1 && ostride == 1 && ooff == 0 && taps_s1 && Lsrc == Ldst && Lrow == Ldst &&
        (Csrc & 31) == 0 && Csrc >= 64 && M % 128 == 0 && (long long)N * (Csrc >> 3) * Lsrc < (1ll << 31)) {
        // LDS-DMA kernel for the 3-tap stride-1 convolutions; G workgroup columns keep the partial-row contract
        dim3 grid(G, M / 128), block(512);
        if (stats != nullptr) hipLaunchKernelGGL((conv_b16s1_kernel<true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_b16s1_kernel<false>), grid, block, 0, st, p);
        return (int)hipGetLastError();
    }
    dim3 grid(G, M / 64), block(256);
#define SSECG_AMP_LAUNCH(KS_)                                                                         \
    do {                                                                                              \
        if (stats != nullptr) hipLaunchKernelGGL((conv_b16_kernel<KS_, true>), grid, block, 0, st, p); \
        else hipLaunchKernelGGL((conv_b16_kernel<KS_, false>), grid, block, 0, st, p);                 \
    } while (0)
    if (ntaps == 3) SSECG_AMP_LAUNCH(3);
    else if (ntaps == 2) SSECG_AMP_LAUNCH(2);
    else SSECG_AMP_LAUNCH(1);
#undef SSECG_AMP_LAUNCH
    return (int)hipGetLastError();
}

int ssecg_amp_bn_apply_fwd(const void* x, void* y, int N, int C, int L, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, const void* residual, int relu, unsigned char* mask_bytes,
                           void* stream) {
    return ssecg_amp_bn_apply_fwd_resbn(x, y, N, C, L, mean, invstd, gamma, beta, residual, nullptr, nullptr, nullptr, nullptr, relu,
                                        mask_bytes, stream);
}

int ssecg_amp_bn_apply_fwd_resbn(const void* x, void* y, int N, int C, int L, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, const void* residual, const float* res_mean,
                                 const float* res_invstd, const float* res_gamma, const float* res_beta, int relu,
                                 unsigned char* mask_bytes, void* stream) {
    if (!x || !y || !gamma || !beta || N <= 0 || C <= 0 || (C & 7) || C > 512 || L <= 0) return SSECG_E_INVAL;
    if ((mean == nullptr) != (invstd == nullptr)) return SSECG_E_INVAL;
    const bool resbn = res_mean != nullptr;
    if (resbn != (res_invstd != nullptr) || resbn != (res_gamma != nullptr) || resbn != (res_beta != nullptr)) return SSECG_E_INVAL;
    if (resbn && residual == nullptr) return SSECG_E_INVAL;
    if ((long long)N * (C >> 3) * L >= (1ll << 31)) return SSECG_E_INVAL;
    if (mask_bytes != nullptr && !relu) return SSECG_E_INVAL;
    hipLaunchKernelGGL(bn_apply_fwd_b16_kernel, dim3(grid_for((size_t)N * (C >> 3) * L)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)x, (u32x4*)y, N, C, L, mean, invstd, gamma, beta, (const u32x4*)residual, relu, mask_bytes,
                       res_mean, res_invstd, res_gamma, res_beta);
    return (int)hipGetLastError();
}

int ssecg_amp_bn_bwd_parts(int N, int C, int L) {
    if (N <= 0 || C <= 0 || (C & 7) || L <= 0) return SSECG_E_INVAL;
    const long long NL = (long long)N * L;
    long long gx = (NL + 256 * 8 - 1) / (256 * 8);
    const long long cap = (kNumCU * 8) / (C >> 3) > 0 ? (kNumCU * 8) / (C >> 3) : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    return (int)gx;
}

int ssecg_amp_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, int mode, int N, int C, int L, float* partial, void* stream) {
    if (!dy || !x || !mean || !invstd || !partial || N <= 0 || C <= 0 || (C & 7) || L <= 0 || mode < 0 || mode > 3 ||
        ((mode == 1 || mode == 3) && !y) || (mode == 2 && (!gamma || !beta)))
        return SSECG_E_INVAL;
    if ((long long)N * (C >> 3) * L >= (1ll << 31)) return SSECG_E_INVAL;
    const int gx = ssecg_amp_bn_bwd_parts(N, C, L);
    hipLaunchKernelGGL((bn_bwd_b16_kernel<false>), dim3(gx, C >> 3), dim3(256), 0, (hipStream_t)stream, (const u32x4*)dy,
                       (const u32x4*)y, (const u32x4*)x, mean, invstd, gamma ? gamma : invstd, beta, mode, N, C, L, partial,
                       (const double*)nullptr, 1.0, (u32x4*)nullptr, (u32x4*)nullptr);
    return (int)hipGetLastError();
}

int ssecg_amp_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int mode, const double* sums, double count, int N, int C, int L,
                           void* dx, void* dz, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !sums || !dx || N <= 0 || C <= 0 || (C & 7) || L <= 0 || mode < 0 || mode > 3 ||
        ((mode == 1 || mode == 3) && !y) || (mode == 2 && !beta) || !(count > 0.0))
        return SSECG_E_INVAL;
    if ((long long)N * (C >> 3) * L >= (1ll << 31)) return SSECG_E_INVAL;
    const int gx = ssecg_amp_bn_bwd_parts(N, C, L);
    hipLaunchKernelGGL((bn_bwd_b16_kernel<true>), dim3(gx, C >> 3), dim3(256), 0, (hipStream_t)stream, (const u32x4*)dy,
                       (const u32x4*)y, (const u32x4*)x, mean, invstd, gamma, beta, mode, N, C, L, (float*)nullptr, sums, count,
                       (u32x4*)dx, (u32x4*)dz);
    return (int)hipGetLastError();
}

int ssecg_amp_bn_bwd_reduce_pair(const void* dy, const void* y, int mode, const void* x, const float* mean, const float* invstd,
                                 const void* x2, const float* mean2, const float* invstd2, int N, int C, int L, float* partial,
                                 float* partial2, void* stream) {
    if (!dy || !y || !x || !mean || !invstd || !x2 || !mean2 || !invstd2 || !partial || !partial2 || N <= 0 || C <= 0 || (C & 7) || L <= 0 ||
        (mode != 1 && mode != 3))
        return SSECG_E_INVAL;
    if ((long long)N * (C >> 3) * L >= (1ll << 31)) return SSECG_E_INVAL;
    BnPairB pb{};
    pb.x = (const u32x4*)x2; pb.mean = mean2; pb.invstd = invstd2; pb.partial = partial2;
    const int gx = ssecg_amp_bn_bwd_parts(N, C, L);
    hipLaunchKernelGGL((bn_bwd_b16_kernel<false, true>), dim3(gx, C >> 3), dim3(256), 0, (hipStream_t)stream, (const u32x4*)dy,
                       (const u32x4*)y, (const u32x4*)x, mean, invstd, invstd, (const float*)nullptr, mode, N, C, L, partial,
                       (const double*)nullptr, 1.0, (u32x4*)nullptr, (u32x4*)nullptr, pb);
    return (int)hipGetLastError();
}

int ssecg_amp_bn_bwd_apply_pair(const void* dy, const void* y, int mode, const void* x, const float* mean, const float* invstd,
                                const float* gamma, const double* sums, const void* x2, const float* mean2, const float* invstd2,
                                const float* gamma2, const double* sums2, double count, int N, int C, int L, void* dx, void* dx2,
                                void* stream) {
    if (!dy || !y || !x || !mean || !invstd || !gamma || !sums || !x2 || !mean2 || !invstd2 || !gamma2 || !sums2 || !dx || !dx2 || N <= 0 ||
        C <= 0 || (C & 7) || L <= 0 || (mode != 1 && mode != 3) || !(count > 0.0))
        return SSECG_E_INVAL;
    if ((long long)N * (C >> 3) * L >= (1ll << 31)) return SSECG_E_INVAL;
    BnPairB pb{};
    pb.x = (const u32x4*)x2; pb.mean = mean2; pb.invstd = invstd2; pb.gamma = gamma2; pb.sums = sums2; pb.dx = (u32x4*)dx2;
    const int gx = ssecg_amp_bn_bwd_parts(N, C, L);
    hipLaunchKernelGGL((bn_bwd_b16_kernel<true, true>), dim3(gx, C >> 3), dim3(256), 0, (hipStream_t)stream, (const u32x4*)dy,
                       (const u32x4*)y, (const u32x4*)x, mean, invstd, gamma, (const float*)nullptr, mode, N, C, L, (float*)nullptr, sums,
                       count, (u32x4*)dx, (u32x4*)nullptr, pb);
    return (int)hipGetLastError();
}

static void launch_wgrad_b16_reduce(const float* ws, float* dw, int Z, int Cout, int Cin, int K, hipStream_t st) {
    const dim3 grid(grid_for((size_t)Cout * Cin * K, 64, 2048));
    if (Z >= 64) hipLaunchKernelGGL(wgrad_b16_reduce_kernel<16>, grid, dim3(1024), 0, st, ws, dw, Z, Cout, Cin, K);
    else hipLaunchKernelGGL(wgrad_b16_reduce_kernel<4>, grid, dim3(256), 0, st, ws, dw, Z, Cout, Cin, K);
}

static int wg_geometry(int N, int Cin, int Lx, int Cout, int Ldy, WgB* p) {
    p->MT = Cout / ((Cout & 127) ? 64 : 128); p->JT = Cin / ((Cin & 127) ? 64 : 128);
    p->stages_per_sample = (Ldy + kSP - 1) / kSP;
    p->total_stages = N * p->stages_per_sample;
    // slabs: enough workgroups to fill the chip (one per CU), at least 8 stages each
    const int tiles = p->MT * p->JT;
    int Z = (kNumCU + tiles - 1) / tiles;
    Z = (Z + 7) & ~7;
    int per = (p->total_stages + Z - 1) / Z;
    if (per < 4) per = 4;
    Z = (p->total_stages + per - 1) / per;
    p->Z = Z; p->stages_per_slab = per;
    return 0;
}

int ssecg_amp_wgrad_supported(int N, int Cin, int Lx, int Cout, int Ldy, int K, int stride, int pad) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Lx <= 0 || Ldy <= 0) return 0;
    if ((Cin & 63) || (Cout & 63)) return 0;
    if (!((K == 3 && pad == 1) || (K == 1 && pad == 0))) return 0;
    if (stride != 1 && stride != 2) return 0;
    if (Ldy != (Lx + 2 * pad - K) / stride + 1) return 0;
    return 1;
}

size_t ssecg_amp_wgrad_workspace(int N, int Cin, int Lx, int Cout, int Ldy, int K) {
    WgB p;
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Ldy <= 0 || (Cin & 63) || (Cout & 63)) return 0;
    wg_geometry(N, Cin, Lx, Cout, Ldy, &p);
    return (size_t)p.Z * Cout * Cin * K * sizeof(float);
}

int ssecg_amp_wgrad(const void* dy, const void* x, float* dw, int N, int Cin, int Lx, int Cout, int Ldy, int K, int stride, int pad,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !x || !dw || !workspace || !ssecg_amp_wgrad_supported(N, Cin, Lx, Cout, Ldy, K, stride, pad)) return SSECG_E_INVAL;
    WgB p;
    wg_geometry(N, Cin, Lx, Cout, Ldy, &p);
    if (workspace_bytes < (size_t)p.Z * Cout * Cin * K * sizeof(float)) return SSECG_E_WORKSPACE;
    p.dy = (const u32x4*)dy; p.x = (const u32x4*)x; p.ws = (float*)workspace;
    p.N = N; p.Cout = Cout; p.Cin = Cin; p.Ldy = Ldy; p.Lx = Lx; p.stride = stride; p.pad = pad;
    const int tiles = p.MT * p.JT;
    const int groups = (p.Z + 7) / 8;
    dim3 grid(groups * tiles * 8), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int tm = (Cout & 127) ? 1 : 2, tj = (Cin & 127) ? 1 : 2;
    {
        // LDS-DMA ring kernels for the (taps, stride, tile) combinations of the network; anything else: register-staged kernel
        bool ring = true;
        if (K == 3 && stride == 1 && tm == 2 && tj == 2) hipLaunchKernelGGL((conv_wgrad_b16s1_kernel<3, 1, 2, 2>), grid, block, 0, st, p);
        else if (K == 3 && stride == 1 && tm == 1 && tj == 1) hipLaunchKernelGGL((conv_wgrad_b16s1_kernel<3, 1, 1, 1>), grid, block, 0, st, p);
        else if (K == 3 && stride == 2 && tm == 2 && tj == 2) hipLaunchKernelGGL((conv_wgrad_b16s1_kernel<3, 2, 2, 2>), grid, block, 0, st, p);
        else if (K == 3 && stride == 2 && tm == 2 && tj == 1) hipLaunchKernelGGL((conv_wgrad_b16s1_kernel<3, 2, 2, 1>), grid, block, 0, st, p);
        else if (K == 1 && stride == 2 && tm == 2 && tj == 2) hipLaunchKernelGGL((conv_wgrad_b16s1_kernel<1, 2, 2, 2>), grid, block, 0, st, p);
        else if (K == 1 && stride == 2 && tm == 2 && tj == 1) hipLaunchKernelGGL((conv_wgrad_b16s1_kernel<1, 2, 2, 1>), grid, block, 0, st, p);
        else ring = false;
        if (ring) {
            launch_wgrad_b16_reduce(p.ws, dw, p.Z, Cout, Cin, K, st);
            return (int)hipGetLastError();
        }
    }
#define SSECG_WGB(KS_)                                                                                     \
    do {                                                                                                   \
        if (tm == 2 && tj == 2) hipLaunchKernelGGL((conv_wgrad_b16_kernel<KS_, 2, 2>), grid, block, 0, st, p);      \
        else if (tm == 2) hipLaunchKernelGGL((conv_wgrad_b16_kernel<KS_, 2, 1>), grid, block, 0, st, p);            \
        else if (tj == 2) hipLaunchKernelGGL((conv_wgrad_b16_kernel<KS_, 1, 2>), grid, block, 0, st, p);            \
        else hipLaunchKernelGGL((conv_wgrad_b16_kernel<KS_, 1, 1>), grid, block, 0, st, p);                         \
    } while (0)
    if (K == 3) SSECG_WGB(3);
    else SSECG_WGB(1);
#undef SSECG_WGB
    launch_wgrad_b16_reduce(p.ws, dw, p.Z, Cout, Cin, K, st);
    return (int)hipGetLastError();
}

}  // extern "C"
