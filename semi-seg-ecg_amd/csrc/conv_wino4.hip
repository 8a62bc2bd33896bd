// 3-tap, stride-1, pad-1 Conv1d (forward and data gradient) in Winograd F(4,3) form on the gfx950 fp32 matrix pipe.
//
// F(4,3) produces FOUR neighbouring outputs from six inputs with 6 multiplications per (out-channel, in-channel)
// instead of 12 (direct) or 8 (the F(2,3) kernels of conv_wino.hip): half the MFMA work of the direct form.
//     d_i = src[c][4j-1+i], i = 0..5        (interpolation points 0, +-1, +-2, inf)
//     v   = ( 4d0 - 5d2 + d4,  -4d1 - 4d2 + d3 + d4,  4d1 - 4d2 - d3 + d4,  -2d1 - d2 + 2d3 + d4,  2d1 - d2 - 2d3 + d4,
//             4d1 - 5d3 + d5 )
//     u   = ( g0/4,  -(g0+g1+g2)/6,  -(g0-g1+g2)/6,  g0/24 + g1/12 + g2/6,  g0/24 - g1/12 + g2/6,  g2 )
//     m_k[co][j] = sum_c u_k[co][c] * v_k[c][j]                                    (six independent GEMMs over quads)
//     y[4j]   = m0 + (m1+m2) +   (m3+m4)          y[4j+1] = (m1-m2) + 2(m3-m4)
//     y[4j+2] =      (m1+m2) + 4 (m3+m4)          y[4j+3] = (m1-m2) + 8(m3-m4) + m5
// fp32 error against an fp64 convolution (He-initialised weights, post-ReLU inputs, C = 64..512, measured on the host,
// tools/wino_numerics.py): relative L2 4e-7..8e-7, max 1.4e-6..1.9e-6 of the output scale - 2.5x the F(2,3) form, 4-7x
// the direct form, 10x inside the 2e-5 kernel bar and 50x inside the 1e-4 model bar.
//
// A wave owns a 32-channel x 32-quad block of all six planes (6 x 16 accumulator registers), so the output transform is
// in-register and the epilogue (BN statistics with the ragged tail masked, folded scale/shift, residual, ReLU,
// LDS-transposed coalesced stores) follows the F(2,3) kernel.  96 accumulators do not fit four waves per SIMD, so the
// workgroup is 8 waves (two per SIMD, one workgroup per CU): 128 channels x 64 quads.  K advances 16 input
// channels per LDS stage (48 MFMAs per wave), double-buffered; LDS operands in the stage order [c/8][plane][half][m][c%4].
// The weight operand in global memory is the tap-major re-layout of the RAW taps (round 3): the six planes are formed by the
// staging threads, which halves the operand's bytes (measured: -31 MB of HBM reads per launch, time unchanged).
// MEASURED alternative (round 2): a 16-wave variant that splits the six planes of a block between two waves (48
// accumulators each, four waves per SIMD) and trades the partial output transforms through LDS after the K loop ran at the
// same speed as this kernel (layer4 shape 0.549 vs 0.530 ms) - the limit is not the number of waves - and was dropped.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ssecg.h"
#include "conv_common.h"

namespace {

typedef unsigned u32x4v __attribute__((__vector_size__(16)));

constexpr int kKC = 8;   // input channels per stage

#if defined(SSECG_ABL4_CLOCK)   // diagnostic build only: shader cycles and 100 MHz ticks of every workgroup's lifetime
__device__ unsigned long long g_w4_stamps[2 * 4096];
#endif

struct Wino4P {
    const float* U;    // tap-major re-laid weights R [C/8][3][2][M][4] (ssecg_conv1d_wino4_weight_multi)
    const float* src;  // (N, C, L)
    float* out;        // (N, M, L)
    int M, C, L, Lq, Q, numQT;
    unsigned src_bytes, out_bytes;
    // K split (small batches: fewer tiles than CUs): workgroup column blockIdx.z contracts channels [z*Cz, (z+1)*Cz) into its own
    // partial output out + z*out_split (plain epilogue); wino4_split_finish_kernel sums the partials and applies the epilogue.
    // No split: Cz = C, out_split = 0, gridDim.z = 1.
    int Cz;
    size_t out_split;
    const float* scale;
    const float* shift;
    const float* residual;
    int relu;
    float* stats;
    const float* in_scale;   // gathered input = relu(src*in_scale[c] + in_shift[c]) (fused producer BN + ReLU), C <= 512
    const float* in_shift;
};

// (Round 3 folded the upstream BatchNorm-backward reduction into this kernel's data-gradient epilogue - template parameter NRED,
// ssecg_conv1d_wino4_dgrad_bnred; bit-identical dx, 13 launches fewer, and 1.1 ms/step SLOWER: the single 154 KB workgroup of a CU
// cannot overlap the extra epilogue reads with anything.  Removed in round 4: tools/experiments/r04_bn_reduce_in_dgrad.patch.)
template <int WM, int WN, bool AFF>   // AFF: the producer's BatchNorm + ReLU is applied to the gathered input
__global__ __launch_bounds__(512, 2) void conv_wino4_kernel(Wino4P p) {   // (compile-time: no branch inside the MFMA blocks)
    static_assert(WM * WN == 8, "8 waves");
    constexpr int NT = 512;
    constexpr int BM = 32 * WM, BNQ = 32 * WN;
    constexpr int SUB = 2;                         // 8-channel sub-stages per LDS stage (one barrier per 16 channels)
    constexpr int U_ITEMS = SUB * 2 * BM;          // (sub-stage, channel half, output channel) weight items: one per thread,
    static_assert(U_ITEMS <= NT && U_ITEMS % 64 == 0, "whole waves stage the weights");   // on the first U_ITEMS / 64 waves
    constexpr int VIT = BNQ / 64;                  // (channel, quad) items per thread per sub-stage (1 or 2)
    constexpr int U_SUB = 6 * 8 * BM, V_SUB = 6 * 8 * BNQ;   // floats per sub-stage
    constexpr int U_STAGE = SUB * U_SUB, V_STAGE = SUB * V_SUB;
    constexpr int T_FLOATS = 8 * 32 * 33;
    constexpr int SMEM_FLOATS = 2 * (U_STAGE + V_STAGE) > T_FLOATS ? 2 * (U_STAGE + V_STAGE) : T_FLOATS;   // 144 KB
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    __shared__ int sRem[BNQ];     // valid outputs (0..4) of each quad of the tile
    __shared__ float2 sAff[AFF ? 512 : 1];
    __shared__ float2 sEp[BM];    // the epilogue's per-channel (scale, shift) of this workgroup's rows: LDS reads do not queue behind global stores
    float* const Us0 = smem;
    float* const Vs0 = smem + 2 * U_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int mt = blockIdx.y, first = blockIdx.x, step = gridDim.x;
    const int m0 = mt * BM;
    const int nstages = p.Cz / (kKC * SUB);
    const int kz = blockIdx.z;
    float* const outp = p.out + (size_t)kz * p.out_split;

    // V staging: lane = (ch4 = lane & 3, pq = lane >> 2); wave -> (channel half vg, quad group)
    const int ch4 = lane & 3, pq = lane >> 2;
    const int vg = wave & 1;
    const int vq0 = (wave >> 1) * 16 + pq;   // + 64 * it
    const unsigned src_skip = (unsigned)kz * (unsigned)p.Cz * (unsigned)p.L * 4u;   // bytes: this split's first channel (0 without a split)
    const auto srcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src) + (size_t)kz * p.Cz * p.L, 0,
                                                        (int)(p.src_bytes - src_skip), 0x00020000);
    const unsigned sub_step = (unsigned)(kKC * p.L) * 4u;      // bytes: 8 channels further
    const unsigned chan_step = SUB * sub_step;
    // weights: the tap-major re-layout R [c/8][tap][(c%8)/4][M][c%4] of the raw taps (HALF the bytes of the six transformed
    // planes): a thread loads the three taps of (sub-stage uu, channel half uh, output channel um) as three float4 (4 input
    // channels), forms the six planes in registers (9 VALU per channel) and writes six float4 to the LDS stage.  MEASURED
    // (round 2 ablation, layer4 shape): the weight operand's global loads alone cost 13 % of the kernel; with the taps
    // transformed here they are halved, and the per-step transform launch becomes a plain re-layout.
    const float4* const Rg = reinterpret_cast<const float4*>(p.U) + (size_t)kz * (p.Cz / 8) * 6 * p.M;
    const int uu = tid / (2 * BM), uh = (tid / BM) & 1, um = tid % BM;
    const bool u_thread = U_ITEMS == NT || tid < U_ITEMS;   // (wave-uniform)

#if defined(SSECG_ABL4_CLOCK)
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), real0 = __builtin_amdgcn_s_memrealtime();
#endif
    float st_sum = 0.f, st_sq = 0.f;
    constexpr bool in_aff = AFF;
    if (in_aff) {
        for (int c = tid; c < p.C; c += NT) sAff[c] = make_float2(p.in_scale[c], p.in_shift[c]);
    }
    if (tid < BM) sEp[tid] = make_float2(p.scale != nullptr ? p.scale[m0 + tid] : 1.f, p.shift != nullptr ? p.shift[m0 + tid] : 0.f);
    __syncthreads();
    // One LDS stage = 16 input channels = 48 MFMAs per wave (~1.5 us at the sustained clock): long enough to cover the
    // latency of the next stage's global loads, which are requested at its start.  MEASURED (layer4 shape, 8-channel
    // stages, tools/ablate_wino4.sh): full 0.554 ms, no global loads 0.445, no LDS stores 0.476, no MFMAs 0.308; two
    // register sets (loads two 8-channel stages ahead, loop unrolled by two) spilled 150-200 VGPRs.
    // Input rows of a (channel, quad) item: ONE 16-byte load of positions 4j .. 4j+3 (d1..d4; any 4-byte alignment - dwordx4 buffer
    // loads are range-checked per dword and need no 16-byte alignment, tools/probes/buffer_load_probe.hip) + two dword loads for the
    // neighbours 4j-1 and 4j+4 (d0, d5): 3 vector-memory instructions per item instead of 6.  voff = {vector offset, left, right}
    // (bit 31 = out of range -> the load returns 0); vcnt = how many of the vector's four positions lie inside the row (the last
    // quad of a row whose length is not a multiple of 4 reads into the next row: masked after the load).
    unsigned voff[VIT][3];
    int vcnt[VIT];
    float4 rw[3];
    float rd[SUB][VIT][6];
    auto tile_offsets = [&](int q0) {
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int q = q0 + vq0 + 64 * it;
            const bool q_ok = q < p.Q;
            const int n = q_ok ? q / p.Lq : 0;
            const int jq = q - n * p.Lq;
            const unsigned row = ((unsigned)n * (unsigned)p.C + (unsigned)(4 * vg + ch4)) * (unsigned)p.L;
            const int l1 = 4 * jq;
            voff[it][0] = oob_if((row + (unsigned)l1) * 4u, !q_ok);
            voff[it][1] = oob_if((row + (unsigned)(l1 - 1)) * 4u, !(q_ok && jq > 0));
            voff[it][2] = oob_if((row + (unsigned)(l1 + 4)) * 4u, !(q_ok && l1 + 4 < p.L));
            const int left = p.L - l1;                      // positions of the row from 4j on
            vcnt[it] = q_ok ? (left > 4 ? 4 : left) : 0;
        }
    };
    // The next stage's global loads are issued in pieces between the MFMA groups of the current stage, not as one burst:
    // a wave issues in order, so 18 back-to-back vector loads on all 8 waves (which run in lockstep after the barrier) park
    // every wave behind the address unit's queue and the matrix pipe idles meanwhile (same-box A/B: -5..-8 %).
    auto load_v = [&](int u, unsigned soff) {
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
#if defined(SSECG_ABL4_NOLOAD) || defined(SSECG_ABL4_NOLOADV)
#pragma unroll
            for (int i = 0; i < 6; ++i) rd[u][it][i] = 0.5f;
#else
            const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(srcR, voff[it][0], soff + u * sub_step, 0);
            const unsigned v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];   // (no __builtin_bit_cast on vector elements: clang reads element 0)
            rd[u][it][1] = __uint_as_float(v0);
            rd[u][it][2] = __uint_as_float(v1);
            rd[u][it][3] = __uint_as_float(v2);
            rd[u][it][4] = __uint_as_float(v3);
            rd[u][it][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srcR, voff[it][1], soff + u * sub_step, 0));
            rd[u][it][5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srcR, voff[it][2], soff + u * sub_step, 0));
#endif
        }
    };
    auto load_u = [&](int s) {
        if (!u_thread) return;
#if defined(SSECG_ABL4_NOLOAD) || defined(SSECG_ABL4_NOLOADU)
        rw[0] = make_float4(1.f, 2.f, 0.5f, 0.25f); rw[1] = rw[0]; rw[2] = rw[0];
#else
        const float4* g = Rg + ((size_t)((s * SUB + uu) * 3) * 2 + uh) * p.M + m0 + um;
        const size_t ts = (size_t)2 * p.M;
        rw[0] = g[0]; rw[1] = g[ts]; rw[2] = g[2 * ts];
#endif
    };
    auto load_stage = [&](int s, unsigned soff) {
#pragma unroll
        for (int u = 0; u < SUB; ++u) load_v(u, soff);
        load_u(s);
    };
    auto store_v = [&](int u, int buf, int chan0) {
        // the row's tail: vector positions beyond the row end hold the NEXT row's first values -> exactly 0 (only when L % 4 != 0)
        if (p.L & 3) {
#pragma unroll
            for (int it = 0; it < VIT; ++it) {
                rd[u][it][2] = vcnt[it] > 1 ? rd[u][it][2] : 0.f;
                rd[u][it][3] = vcnt[it] > 2 ? rd[u][it][3] : 0.f;
                rd[u][it][4] = vcnt[it] > 3 ? rd[u][it][4] : 0.f;
            }
        }
        if (in_aff) {
            const float2 ab = sAff[kz * p.Cz + chan0 + 8 * u + 4 * vg + ch4];   // (kz * Cz: this K split's first channel)
#pragma unroll
            for (int it = 0; it < VIT; ++it)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float a = fmaxf(fmaf(rd[u][it][i], ab.x, ab.y), 0.f);
                    const bool inside = i == 0 ? (int)voff[it][1] >= 0 : (i == 5 ? (int)voff[it][2] >= 0 : vcnt[it] > i - 1);
                    rd[u][it][i] = inside ? a : 0.f;   // padding / out of range: stays exactly 0
                }
        }
#if defined(SSECG_ABL4_NOSTORE)   // timing experiment: loads waited for, nothing written to LDS
        for (int it = 0; it < VIT; ++it) for (int i = 0; i < 6; ++i) asm volatile("" :: "v"(rd[u][it][i]));
        return;
#endif
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            float* v = Vs0 + buf * V_STAGE + u * V_SUB + (vg * BNQ + vq0 + 64 * it) * 4 + ch4;
            const float d0 = rd[u][it][0], d1 = rd[u][it][1], d2 = rd[u][it][2], d3 = rd[u][it][3], d4 = rd[u][it][4], d5 = rd[u][it][5];
            const float a = d4 - 4.f * d2, b = d3 - 4.f * d1;       // shared sub-expressions of v1 / v2
            const float c = d4 - d2, e = 2.f * (d3 - d1);           // ... of v3 / v4
            constexpr int PS = 2 * BNQ * 4;                         // floats between planes
            v[0 * PS] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
            v[1 * PS] = a + b;
            v[2 * PS] = a - b;
            v[3 * PS] = c + e;
            v[4 * PS] = c - e;
            v[5 * PS] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
        }
    };
    auto store_u = [&](int buf) {
        if (!u_thread) return;
#if defined(SSECG_ABL4_NOSTORE)
        asm volatile("" :: "v"(rw[0].x), "v"(rw[1].y), "v"(rw[2].z));
        return;
#endif
        // u = (g0/4, -(g0+g1+g2)/6, -(g0-g1+g2)/6, g0/24 + g1/12 + g2/6, g0/24 - g1/12 + g2/6, g2): the arithmetic of the
        // former per-step transform kernel, operation for operation (bit-identical operands)
        float4 u0, u1, u2, u3, u4;
#define W4_U(X)                                                                 \
        {                                                                       \
            const float g0 = rw[0].X, g1 = rw[1].X, g2 = rw[2].X;               \
            const float s02 = g0 + g2;                                          \
            const float t = fmaf(g2, 4.0f, g0) * (1.0f / 24.0f);                \
            u0.X = g0 * 0.25f;                                                  \
            u1.X = (s02 + g1) * (-1.0f / 6.0f);                                 \
            u2.X = (s02 - g1) * (-1.0f / 6.0f);                                 \
            u3.X = fmaf(g1, 1.0f / 12.0f, t);                                   \
            u4.X = fmaf(g1, -1.0f / 12.0f, t);                                  \
        }
        W4_U(x) W4_U(y) W4_U(z) W4_U(w)
#undef W4_U
        float4* dst = reinterpret_cast<float4*>(Us0 + buf * U_STAGE + uu * U_SUB) + uh * BM + um;   // + plane * 2 * BM
        dst[0 * 2 * BM] = u0;
        dst[1 * 2 * BM] = u1;
        dst[2 * 2 * BM] = u2;
        dst[3 * 2 * BM] = u3;
        dst[4 * 2 * BM] = u4;
        dst[5 * 2 * BM] = rw[2];
    };
    auto store_stage = [&](int buf, int chan0) {
#pragma unroll
        for (int u = 0; u < SUB; ++u) store_v(u, buf, chan0);
        store_u(buf);
    };
    if (first < p.numQT) {
        tile_offsets(first * BNQ);
        load_stage(0, 0u);
    }
    for (int qt = first; qt < p.numQT; qt += step) {
        const int q0 = qt * BNQ;
        if (tid < BNQ) {
            const int q = q0 + tid;
            int rem = 0;
            if (q < p.Q) {
                const int jq = q % p.Lq;
                rem = p.L - 4 * jq;
                rem = rem > 4 ? 4 : rem;
            }
            sRem[tid] = rem;
        }

        f32x16 acc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

        // A sub-stage = three groups of two planes (8 MFMAs each).  The fragments of a group are requested from LDS BEFORE the
        // MFMAs of the previous group are issued; the next stage's LDS stores (transform + 18 ds_write) sit in the middle of
        // the second sub-stage: an MFMA occupies the pipe for 64 cycles and the wave issues the VALU / DS work in its shadow.
#define W4_READ(base_u, base_v, kk, U0, V0, U1, V1)                                          \
        const float4 U0 = *reinterpret_cast<const float4*>((base_u) + (kk) * 2 * BM * 4);          \
        const float4 V0 = *reinterpret_cast<const float4*>((base_v) + (kk) * 2 * BNQ * 4);         \
        const float4 U1 = *reinterpret_cast<const float4*>((base_u) + ((kk) + 1) * 2 * BM * 4);    \
        const float4 V1 = *reinterpret_cast<const float4*>((base_v) + ((kk) + 1) * 2 * BNQ * 4);
#if defined(SSECG_ABL4_NOMFMA)   // timing experiment: fragments read, no MFMA
#define W4_MMA(kk, U0, V0, U1, V1) asm volatile("" :: "v"(U0.x), "v"(V0.x), "v"(U1.w), "v"(V1.w), "v"(U0.y), "v"(V0.z));
#else
#define W4_MMA(kk, U0, V0, U1, V1)                                                                         \
        acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(V0.x, U0.x, acc[kk], 0, 0, 0);                       \
        acc[(kk) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V1.x, U1.x, acc[(kk) + 1], 0, 0, 0);           \
        acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(V0.y, U0.y, acc[kk], 0, 0, 0);                       \
        acc[(kk) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V1.y, U1.y, acc[(kk) + 1], 0, 0, 0);           \
        acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(V0.z, U0.z, acc[kk], 0, 0, 0);                       \
        acc[(kk) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V1.z, U1.z, acc[(kk) + 1], 0, 0, 0);           \
        acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(V0.w, U0.w, acc[kk], 0, 0, 0);                       \
        acc[(kk) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V1.w, U1.w, acc[(kk) + 1], 0, 0, 0);
#endif

        // The stage body is expanded twice - inside the loop (with the next stage's loads and LDS stores, unconditionally)
        // and once for the last stage (without) - so that the stores sit in the SAME basic block as the MFMAs around them and
        // the scheduler can interleave them; as an `if (more)` block they were a scheduling barrier.  MEASURED: giving waves
        // 4-7 an earlier store position (a stagger between the two waves of a SIMD) was 3 % slower.
        // The loop is rotated across the barrier: the last MFMA group of stage s is issued AFTER the barrier and after the
        // first two fragment groups of stage s+1 have been requested, so the matrix pipe has 8 MFMAs (512 cycles) of work
        // while those LDS reads are in flight (PMC before: pipe 65 % busy, waves parked at waitcnt / barrier 26 % of their time).
#if defined(SSECG_ABL4_NOREAD)   // timing experiment: fragments never re-read from LDS
#define W4_RD(DST, base_u, base_v, kk) asm volatile("" : "+v"(DST##u0.x), "+v"(DST##v0.x), "+v"(DST##u1.x), "+v"(DST##v1.x));
#else
#define W4_RD(DST, base_u, base_v, kk)                                                       \
        DST##u0 = *reinterpret_cast<const float4*>((base_u) + (kk) * 2 * BM * 4);                  \
        DST##v0 = *reinterpret_cast<const float4*>((base_v) + (kk) * 2 * BNQ * 4);                 \
        DST##u1 = *reinterpret_cast<const float4*>((base_u) + ((kk) + 1) * 2 * BM * 4);            \
        DST##v1 = *reinterpret_cast<const float4*>((base_v) + ((kk) + 1) * 2 * BNQ * 4);
#endif
#define W4_M(kk, SRC) W4_MMA(kk, SRC##u0, SRC##v0, SRC##u1, SRC##v1)
#define W4_FENCE __builtin_amdgcn_sched_barrier(0);
        float4 Au0, Av0, Au1, Av1, Bu0, Bv0, Bu1, Bv1, Cu0, Cv0, Cu1, Cv1, Du0, Dv0, Du1, Dv1, Eu0, Ev0, Eu1, Ev1, Fu0, Fv0, Fu1, Fv1;
#define W4_BODY(G1, G2, G3, G4, G5, TAIL)                                                    \
        {                                                                                    \
            const int buf = s & 1;                                                           \
            soff += chan_step;                                                               \
            const float* us = Us0 + buf * U_STAGE + (lhi * BM + wm * 32 + l31) * 4;            \
            const float* vs = Vs0 + buf * V_STAGE + (lhi * BNQ + wn * 32 + l31) * 4;           \
            W4_FENCE G1 W4_M(0, A) W4_FENCE                                                  \
            W4_RD(C, us, vs, 4) W4_FENCE                                                     \
            G2 W4_M(2, B) W4_FENCE                                                           \
            W4_RD(D, us + U_SUB, vs + V_SUB, 0) W4_FENCE                                     \
            G3 W4_M(4, C) W4_FENCE                                                           \
            W4_RD(E, us + U_SUB, vs + V_SUB, 2)                                              \
            W4_RD(F, us + U_SUB, vs + V_SUB, 4) W4_FENCE                                     \
            G4 W4_M(0, D) W4_FENCE                                                           \
            G5 W4_M(2, E)                                                                    \
            TAIL                                                                             \
        }

        unsigned soff = 0;
        __syncthreads();   // the previous tile's readers are done with the LDS buffers (and sRem is written)
        store_stage(0, 0);
        __syncthreads();
        {
            const float* us = Us0 + (lhi * BM + wm * 32 + l31) * 4;
            const float* vs = Vs0 + (lhi * BNQ + wn * 32 + l31) * 4;
            W4_RD(A, us, vs, 0)
            W4_RD(B, us, vs, 2)
        }
        // Schedule of one stage s (six groups of 8 MFMAs; the barrier sits between the fifth and the sixth): the input rows of
        // stage s+1, first half, were requested during the sixth group of stage s-1 (an out-of-range stage reads zeros or the
        // next sample's rows - never used); group 1 requests the second half, groups 2 and 3 the transformed weights; group 4
        // transforms + stores the inputs, group 5 stores the weights.
        load_v(0, chan_step);
        int s = 0;
        for (; s + 1 < nstages; ++s)
            W4_BODY(load_v(1, soff);, load_u(s + 1);, ,
                    store_v(0, buf ^ 1, (s + 1) * kKC * SUB); store_v(1, buf ^ 1, (s + 1) * kKC * SUB);,
                    store_u(buf ^ 1);,
                    __syncthreads();
                    const float* usn = Us0 + (buf ^ 1) * U_STAGE + (lhi * BM + wm * 32 + l31) * 4;
                    const float* vsn = Vs0 + (buf ^ 1) * V_STAGE + (lhi * BNQ + wn * 32 + l31) * 4;
                    W4_RD(A, usn, vsn, 0)
                    W4_RD(B, usn, vsn, 2)
                    W4_FENCE
                    load_v(0, soff + chan_step);
                    W4_M(4, F))
        W4_BODY(, , , , , W4_M(4, F) __syncthreads();)
#undef W4_BODY
#undef W4_RD
#undef W4_M
#undef W4_FENCE
#undef W4_READ
#undef W4_MMA

        if (qt + step < p.numQT) {   // next tile's first stage: in flight during this tile's epilogue
            tile_offsets((qt + step) * BNQ);
            load_stage(0, 0u);
        }

        // ---------------- epilogue: output transform, statistics, stores ----------------
        // acc[e] <- y[4j+e]   (register r = quad row (r&3) + 8(r>>2) + 4*lhi of the wave's 32)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0_ = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            acc[0][r] = (m0_ + s12) + s34;
            acc[1][r] = fmaf(2.f, d34, d12);
            acc[2][r] = fmaf(4.f, s34, s12);
            acc[3][r] = fmaf(8.f, d34, d12) + m5;
        }
        if (p.stats != nullptr) {
            float s = 0.f, q = 0.f;
            if (p.L & 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rem = sRem[wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = e < rem ? acc[e][r] : 0.f;
                        s += y;
                        q = fmaf(y, y, q);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = acc[e][r];
                        s += y;
                        q = fmaf(y, y, q);
                    }
            }
            st_sum += s;
            st_sq += q;
        }
        {
            int opq = 0;
            asm volatile("" : "+s"(opq));
            float* T = smem + wave * (32 * 33) + opq;
            const bool plain = p.scale == nullptr && p.shift == nullptr && p.residual == nullptr && !p.relu;
            const int rbase = m0 + wm * 32 + lhi + opq;
            const unsigned ostep = 2u * (unsigned)p.L;
            // round h = quads 8h..8h+7 of the wave's block = 32 consecutive positions per sample row (registers 4h..4h+3)
            auto round_of = [&](int h, bool& pok) -> unsigned {
                const int q = q0 + wn * 32 + 8 * h + (l31 >> 2);
                const bool q_ok = q < p.Q;
                const int n = q_ok ? q / p.Lq : 0;
                const int l = 4 * (q - n * p.Lq) + (l31 & 3);
                pok = q_ok && l < p.L;
                return ((unsigned)n * (unsigned)p.M + (unsigned)rbase) * (unsigned)p.L + (unsigned)l;
            };
            // Fused epilogue ([* scale] [+ shift] [+ residual] [ReLU]): the per-row scale / shift come from LDS (sEp), the
            // residual rows of round h+1 while round h is transposed and stored (two register sets) - a load issued inside the
            // store loop waits for the store in front of it (possible alias; profiles/r04_residual_epilogue.txt)
            // (raw buffer loads: ONE offset register per round - the row step travels in the scalar offset - and an out-of-range
            // offset instead of a branch for positions outside the tensor)
            const auto outR = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)p.out_bytes, 0x00020000);
            float res[2][16];
            bool pok_n = false;
            unsigned o_n = 0;
            const auto resR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0, (int)p.out_bytes, 0x00020000);
            auto load_res = [&](float (&dst)[16], unsigned o, bool pok) {
                const unsigned off = oob_if(o * 4u, !pok);
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2)
                    dst[k2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(resR, off, (unsigned)k2 * ostep * 4u, 0));
            };
            if (!plain) {
                o_n = round_of(0, pok_n);
                if (p.residual != nullptr) load_res(res[0], o_n, pok_n);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int qrow = rr + 4 * lhi;   // quad within the round (0..7)
#pragma unroll
                    for (int e = 0; e < 4; ++e) T[l31 * 33 + 4 * qrow + e] = acc[e][4 * h + rr];
                }
                asm volatile("" ::: "memory");
                if (plain) {   // (buffer stores: an out-of-range offset instead of a branch per row for positions outside the tensor)
                    bool pok;
                    const unsigned o = round_of(h, pok);
                    const unsigned off = oob_if(o * 4u, !pok);
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, T[(2 * k2 + lhi) * 33 + l31]), outR, off,
                                                              (unsigned)k2 * ostep * 4u, 0);
                } else {
                    const bool pok = pok_n;
                    unsigned o = o_n;
                    if (h + 1 < 4) {
                        o_n = round_of(h + 1, pok_n);
                        if (p.residual != nullptr) load_res(res[(h + 1) & 1], o_n, pok_n);
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) {
                        float v = T[(2 * k2 + lhi) * 33 + l31];
                        const float2 ss = sEp[wm * 32 + lhi + 2 * k2];
                        if (pok) {
                            if (p.scale != nullptr) v *= ss.x;
                            if (p.shift != nullptr) v += ss.y;
                            if (p.residual != nullptr) v += res[h & 1][k2];
                            if (p.relu) v = fmaxf(v, 0.f);
                            outp[o] = v;
                        }
                        o += ostep;
                    }
                }
            }
        }
        __syncthreads();   // the next tile's staging overwrites the transpose tiles (and sRem)
    }

#if defined(SSECG_ABL4_CLOCK)
    if (tid == 0) {
        const int b = (blockIdx.y * gridDim.x + blockIdx.x) & 4095;
        g_w4_stamps[2 * b] = __builtin_amdgcn_s_memtime() - clk0;
        g_w4_stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - real0;
    }
#endif
    if (p.stats != nullptr) {
        float* red = smem;   // [WN][BM][2]
        const float s = st_sum + __shfl_xor(st_sum, 32, 64);
        const float q = st_sq + __shfl_xor(st_sq, 32, 64);
        if (lhi == 0) {
            const int r = wm * 32 + l31;
            red[(wn * BM + r) * 2 + 0] = s;
            red[(wn * BM + r) * 2 + 1] = q;
        }
        __syncthreads();
        if (tid < BM) {
            float ss = 0.f, qq = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { ss += red[(w * BM + tid) * 2]; qq += red[(w * BM + tid) * 2 + 1]; }
            float* dst = p.stats + ((size_t)first * p.M + m0 + tid) * 2;
            dst[0] = ss;
            dst[1] = qq;
        }
    }
}


// the three taps of one (m, c) pair in the staging order of conv_wino4_kernel: r layout [c/8][tap][(c%8)/4][M][c%4]
__device__ __forceinline__ void put_u4(float* __restrict__ r, int M, int m, int c, float g0, float g1, float g2) {
    const size_t base = ((((size_t)(c >> 3) * 3) * 2 + ((c >> 2) & 1)) * M + m) * 4 + (c & 3);
    const size_t ts = (size_t)2 * M * 4;
    r[base] = g0;
    r[base + ts] = g1;
    r[base + 2 * ts] = g2;
}

// all registered weights in one launch: row = {w, u_fwd, u_transposed, Cout, Cin}; blockIdx.y = tensor
__global__ void wino4_weight_multi_kernel(const int64_t* __restrict__ table) {
    const int64_t* row = table + 5 * (size_t)blockIdx.y;
    const float* w = reinterpret_cast<const float*>(row[0]);
    float* uf = reinterpret_cast<float*>(row[1]);
    float* ut = reinterpret_cast<float*>(row[2]);
    const int Cout = (int)row[3], Cin = (int)row[4];
    const int total = Cout * Cin;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int ci = e % Cin, co = e / Cin;
        const float g0 = w[(size_t)e * 3], g1 = w[(size_t)e * 3 + 1], g2 = w[(size_t)e * 3 + 2];
        if (uf) put_u4(uf, Cout, co, ci, g0, g1, g2);   // forward operand: m = co, c = ci
        if (ut) put_u4(ut, Cin, ci, co, g2, g1, g0);    // data-gradient operand: m = ci, c = co, taps flipped
    }
}


struct W4Cfg { int BM, BNQ, numQT, MT, G; };

inline W4Cfg pick_wino4(int M, long long Q) {
    W4Cfg c;
    c.BM = M % 128 == 0 ? 128 : 64;      // 64 output channels (layer1): two channel waves x four quad waves
    c.BNQ = c.BM == 128 ? 64 : 128;
    c.numQT = (int)((Q + c.BNQ - 1) / c.BNQ);
    c.MT = M / c.BM;
    int g = (kNumCU / c.MT) & ~7;   // one workgroup per CU; the channel tiles of one quad tile share an XCD (G % 8 == 0)
    if (g < 8) g = 8;
    c.G = c.numQT < g ? c.numQT : g;
    return c;
}

// K split for launches with fewer tiles than CUs (small batches): up to 8 workgroup columns, each contracting whole 16-channel
// stages (at least two); 1 = no split.  The partial planes are summed, the epilogue applied and the BatchNorm sums taken by
// ssecg_detail::launch_split_finish (conv.hip).
inline int pick_wino4_split(int M, int C, long long Q) {
    const W4Cfg c = pick_wino4(M, Q);
    return ssecg_detail::pick_ksplit((long long)c.numQT * c.MT, c.numQT, kNumCU, C, 2 * kKC);
}

inline bool wino4_shape_ok(int N, int C, int L, int M) {
    if (N <= 0 || C <= 0 || L <= 0 || M <= 0) return false;
    if (C % (2 * kKC) != 0 || M % 64 != 0) return false;   // 16 input channels per LDS stage, 128 (or 64) output channels per workgroup
    const long long Q = (long long)N * ((L + 3) / 4);
    if (Q > 0x7fffffffLL) return false;
    return (size_t)N * C * L * 4 < 0x7fffff00ull && (size_t)N * M * L * 4 < 0x7fffff00ull;
}

}  // namespace

extern "C" {

#if defined(SSECG_ABL4_CLOCK)
// median in-kernel clock (GHz) of the last F(4,3) launch's first n workgroups
double ssecg_debug_w4_clock(int n) {
    static unsigned long long h[2 * 4096];
    if (n > 4096) n = 4096;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_w4_stamps), sizeof(h)) != hipSuccess) return -1.0;
    double v[4096];
    int m = 0;
    for (int i = 0; i < n; ++i) if (h[2 * i + 1]) v[m++] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    if (!m) return 0.0;
    for (int i = 1; i < m; ++i) { double x = v[i]; int j = i - 1; while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; } v[j + 1] = x; }
    return v[m / 2];
}
#endif

int ssecg_conv1d_wino4_supported(int N, int C, int L, int M) { return wino4_shape_ok(N, C, L, M) ? 1 : 0; }

int ssecg_conv1d_wino4_parts(int N, int L, int M) {
    if (N <= 0 || L <= 0 || M <= 0 || M % 64 != 0) return SSECG_E_INVAL;
    return pick_wino4(M, (long long)N * ((L + 3) / 4)).G;
}

int ssecg_conv1d_wino4_weight_multi(const int64_t* table, int ntensors, int max_elems, void* stream) {
    if (!table || ntensors <= 0 || max_elems <= 0) return SSECG_E_INVAL;
    int bx = (max_elems + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(wino4_weight_multi_kernel, dim3(bx, ntensors), dim3(256), 0, (hipStream_t)stream, table);
    return (int)hipGetLastError();
}

int ssecg_conv1d_wino4_split(int N, int C, int L, int M) {
    if (!wino4_shape_ok(N, C, L, M)) return SSECG_E_INVAL;
    return pick_wino4_split(M, C, (long long)N * ((L + 3) / 4));
}

int ssecg_conv1d_wino4(const float* src, const float* u, float* out, int N, int C, int L, int M, const float* scale,
                       const float* shift, const float* residual, int relu, float* stats_partial, int stats_parts,
                       const float* in_scale, const float* in_shift, float* split_ws, size_t split_ws_bytes, void* stream) {
    if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale != nullptr && C > 512)) return SSECG_E_INVAL;
    if (scale != nullptr && shift == nullptr) return SSECG_E_INVAL;
    if (!src || !u || !out || !wino4_shape_ok(N, C, L, M) || (((uintptr_t)u) & 15) != 0) return SSECG_E_INVAL;
    const int Lq = (L + 3) / 4;
    const long long Q = (long long)N * Lq;
    const W4Cfg c = pick_wino4(M, Q);
    const bool will_split = split_ws != nullptr && pick_wino4_split(M, C, Q) > 1;   // (the finishing pass writes every statistics row)
    if (stats_partial != nullptr && !will_split) {
        if (stats_parts < c.G) return SSECG_E_WORKSPACE;
        if (stats_parts > c.G) {
            const hipError_t e = hipMemsetAsync(stats_partial + (size_t)c.G * M * 2, 0,
                                                (size_t)(stats_parts - c.G) * M * 2 * sizeof(float), (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    Wino4P p;
    p.U = u; p.src = src; p.out = out;
    p.M = M; p.C = C; p.L = L; p.Lq = Lq; p.Q = (int)Q; p.numQT = c.numQT;
    p.src_bytes = (unsigned)((size_t)N * C * L * 4);
    p.out_bytes = (unsigned)((size_t)N * M * L * 4);
    p.out_bytes = (unsigned)((size_t)N * M * L * 4);
    p.Cz = C; p.out_split = 0;
    p.scale = scale; p.shift = shift; p.residual = residual; p.relu = relu; p.stats = stats_partial;
    p.in_scale = in_scale; p.in_shift = in_shift;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(c.G, c.MT), block(512);
    // small launches (fewer tiles than CUs): K split over blockIdx.z into the caller's workspace; the finishing pass adds the
    // partial planes, applies the epilogue and - for a train-mode forward - emits the BatchNorm partial sums (round 5: rounds 3-4
    // split only launches without statistics / fused input BN)
    const int S = split_ws != nullptr ? pick_wino4_split(M, C, Q) : 1;
    if (S > 1) {
        const size_t plane = (size_t)N * M * L;
        if (split_ws_bytes < (size_t)S * plane * sizeof(float)) return SSECG_E_WORKSPACE;
        p.Cz = C / S; p.out_split = plane; p.out = split_ws;
        p.scale = nullptr; p.shift = nullptr; p.residual = nullptr; p.relu = 0; p.stats = nullptr;
        grid.z = S;
        if (c.BM == 128) {
            if (in_scale != nullptr) hipLaunchKernelGGL((conv_wino4_kernel<4, 2, true>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((conv_wino4_kernel<4, 2, false>), grid, block, 0, st, p);
        } else {
            if (in_scale != nullptr) hipLaunchKernelGGL((conv_wino4_kernel<2, 4, true>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((conv_wino4_kernel<2, 4, false>), grid, block, 0, st, p);
        }
        return ssecg_detail::launch_split_finish(split_ws, S, plane, out, N, M, L, L, 1, 0, scale, shift, residual, relu, stats_partial,
                                                 stats_parts, st);
    }
    if (c.BM == 128) {
        if (in_scale != nullptr) hipLaunchKernelGGL((conv_wino4_kernel<4, 2, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_wino4_kernel<4, 2, false>), grid, block, 0, st, p);
    } else {
        if (in_scale != nullptr) hipLaunchKernelGGL((conv_wino4_kernel<2, 4, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_wino4_kernel<2, 4, false>), grid, block, 0, st, p);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
