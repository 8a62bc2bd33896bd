// 3-tap, stride-1, pad-1 Conv1d (forward and data gradient) in Winograd F(2,3) form on the gfx950 fp32 matrix pipe.
//
// 83 % of the network's MACs sit in such convolutions (14 of the 17 k=3 convs).  F(2,3) produces two
// neighbouring outputs from four inputs with 4 multiplications per (out-channel, in-channel) instead of 6:
//     d_i = src[c][2j-1+i], i = 0..3            v = (d0-d2, d1+d2, d2-d1, d1-d3)
//     u   = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)            m_k[co][j] = sum_c u_k[co][c] * v_k[c][j]
//     y[2j] = m0 + m1 + m2          y[2j+1] = m1 - m2 - m3
// i.e. FOUR independent GEMMs of depth Cin over the PAIR axis q = n*ceil(L/2) + j.  A wave owns a 32-channel x 32-pair
// block of all four (4 x 16 accumulator registers), so the output transform is in-register arithmetic and everything
// after it (BN statistics, folded-BN scale/shift, residual, ReLU, coalesced stores through a per-wave LDS transpose)
// is the epilogue of the direct kernel (conv.hip).
//
// Staging.  K advances 8 input channels per LDS stage.  The transformed weights are stored by
// ssecg_conv1d_wino_weight in the stage order [c/8][k][half][m][c%4], so a stage is one contiguous block: 16-byte
// coalesced global loads, 16-byte LDS stores.  The input transform is applied on the fly: a thread owns one
// (channel, pair), issues its four d_i loads as raw buffer loads whose offsets (padding / range verdict in bit 31 ->
// the load returns 0) are computed once per tile, and writes v_0..v_3 to the four planes.  Fragments are read with
// ds_read_b128: lane half h holds channels 4h..4h+3 of the stage for both operands, one read feeds four MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ssecg.h"
#include "conv_common.h"

#if defined(SSECG_WINO_TRACE)
// debug build only (tools/trace_wino.py): s_memtime stamps of wave 0 of workgroup 0, 5 per K stage of its first tile
__device__ unsigned long long g_wino_trace[8192];
#define WINO_STAMP(slot)                                                                   \
    do {                                                                                   \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && qt == first && s < 1600) \
            g_wino_trace[s * 5 + (slot)] = __builtin_readcyclecounter();                   \
    } while (0)
#else
#define WINO_STAMP(slot) do {} while (0)
#endif

namespace {

constexpr int kWinoKC = 8;    // input channels per stage

struct WinoP {
    const float* U;    // [C/8][4][2][M][4]
    const float* src;  // (N, C, L)
    float* out;        // (N, M, L)
    int M, C, L, Lh, Q, numQT;
    unsigned src_bytes;
    const float* scale;
    const float* shift;
    const float* residual;
    int relu;
    float* stats;
    // the gathered input is relu(src*in_scale[c] + in_shift[c]) - the producer's BatchNorm + ReLU applied while staging, so
    // that activation is never materialised (padding stays exactly 0); C <= 512
    const float* in_scale;
    const float* in_shift;
};

template <int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 4) void conv_wino_kernel(WinoP p) {
    constexpr int NT = WM * WN * 64;   // 8 waves (two workgroups per CU) or 16 waves (one workgroup per CU)
    constexpr int BM = 32 * WM, BNP = 32 * WN;
    static_assert(NT == 512 || NT == 1024, "8 or 16 waves");
    constexpr int UF4 = (8 * BM + NT - 1) / NT;   // float4 of transformed weights per thread per stage
    constexpr bool U_ALL = (8 * BM) % NT == 0;    // every thread stages weights (else only the first 8*BM)
    constexpr int VIT = 8 * BNP / NT;             // (channel, pair) items per thread per stage (1 or 2)
    static_assert(VIT >= 1 && (8 * BNP) % NT == 0, "V staging shape");
    constexpr int U_STAGE = 8 * BM * 4, V_STAGE = 8 * BNP * 4;  // floats
    constexpr int T_FLOATS = (NT / 64) * 32 * 33;
    // LDS ring of SLOTS 8-channel stages (2 = the classic double buffer).  MEASURED: a 4-slot ring with one workgroup
    // barrier per TWO stages (32 MFMAs per wave per barrier) ran at the same speed on every layer shape (layer4: 154.4 vs
    // 154.7 TF), and reading all eight fragments of a stage before its 16 MFMAs needs 32 fragment registers -> spills.
    // Giving the four waves of a SIMD different phase orders (MFMAs first / loads first / loads+store first) so that the
    // pipe is never without a burst lost 6 % (layer4 146 vs 155 TF): the arbitration already interleaves the bursts.
    // Ablations on the 16-wave kernel (layer4, same box): full 0.70 ms; no input-transform arithmetic 0.695; loads waited
    // for but nothing stored to LDS 0.578; no loads at all 0.528 - the LDS stores (3 instructions per wave per stage) are
    // the expensive part of staging.  Moving them to the START of the next iteration ("write after the barrier, re-issue
    // the loads at once") measured 3 % slower (148 vs 153 TF), a double-buffered weight-gradient kernel the same as the
    // single-buffered one.  A plane-innermost LDS image (one 16-byte store per staged item, four accumulator chains per
    // fragment pair, rows padded to 20 floats) was correct and 2-3 % slower on every layer shape.
    constexpr int SLOTS = 2;
    constexpr int SMEM_FLOATS = SLOTS * (U_STAGE + V_STAGE) > T_FLOATS ? SLOTS * (U_STAGE + V_STAGE) : T_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    __shared__ float sMask[BNP];
    __shared__ float2 sAff[512];  // per-input-channel {scale, shift} of the fused producer BN
    float* const Us0 = smem;
    float* const Vs0 = smem + SLOTS * U_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    // Work assignment: grid (pair-tile slots, channel tiles).  Workgroup ids are dealt round-robin to the 8 XCDs, so the
    // channel tiles of one pair tile (ids x, x + G, x + 2G, ...; G % 8 == 0) land on the SAME XCD and run at the same time:
    // the input tile is fetched from HBM once and shared through that XCD's L2.  MEASURED (PMC FETCH_SIZE, N = 1024): the
    // opposite placement (one channel tile per XCD, so each L2 keeps a single 1 MB weight slice) read 563 MB instead of
    // 267 MB per launch on the layer4 shape (281 vs 193 MB on layer3) and was not faster (0.717 vs 0.697 ms).
    const int mt = blockIdx.y, first = blockIdx.x, step = gridDim.x;
    const int m0 = mt * BM;
    const int nstages = p.C / kWinoKC;

    // V staging: item e = tid + it*NT: lane = (ch4 = lane & 3, pq = lane >> 2); e >> 6 = (half g = low bit, pair group)
    const int ch4 = lane & 3, pq = lane >> 2;
    const int vg = wave & 1;
    const int vq0 = (wave >> 1) * 16 + pq;            // + (NT / 8) * it
    constexpr int VQ_STEP = NT / 8;                   // (NT/64 waves / 2 halves) * 16 pairs
    const auto srcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)p.src_bytes, 0x00020000);
    const unsigned chan_step = (unsigned)(kWinoKC * p.L) * 4u;
    // U staging: float4 index e = tid + it * NT -> (kg = e / BM, m = e % BM); global float4 index (s*8 + kg)*M + m0 + m
    const float4* const Ug = reinterpret_cast<const float4*>(p.U);

    float st_sum = 0.f, st_sq = 0.f;
    const bool in_aff = p.in_scale != nullptr;
    if (in_aff) {
        for (int c = tid; c < p.C; c += NT) sAff[c] = make_float2(p.in_scale[c], p.in_shift[c]);
        __syncthreads();
    }

    // staging state lives across tiles: the first stage of the NEXT tile is requested before the epilogue of the current
    // one (its registers are free there), so a tile does not start with an exposed global-memory latency - on the
    // 8-stage 64-channel tiles that latency was a quarter of the tile time
    unsigned voff[VIT][4];
    float ru[UF4][4];  // scalars (not a float4 array): stays in registers across the lambdas
    float rd[VIT][4];
    auto tile_offsets = [&](int q0) {
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int q = q0 + vq0 + VQ_STEP * it;
            const bool q_ok = q < p.Q;
#if defined(SSECG_ABL_VHOT)   // timing experiment: every tile gathers from sample 0 (cache-resident input)
            const int n = 0;
            const int jh = q_ok ? q % p.Lh : 0;
#else
            const int n = q_ok ? q / p.Lh : 0;
            const int jh = q - n * p.Lh;
#endif
            const unsigned row = ((unsigned)n * (unsigned)p.C + (unsigned)(4 * vg + ch4)) * (unsigned)p.L;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int l = 2 * jh - 1 + i;
                voff[it][i] = oob_if((row + (unsigned)l) * 4u, !(q_ok && (unsigned)l < (unsigned)p.L));
            }
        }
    };
    auto load_stage = [&](int s, unsigned soff) {
#if !defined(SSECG_ABL_NOU)
#pragma unroll
        for (int it = 0; it < UF4; ++it) {
            const int e = tid + it * NT;
            if (U_ALL || e < 8 * BM) {
                const int kg = e / BM, m = e % BM;
                const float4 t4 = Ug[((size_t)s * 8 + kg) * p.M + m0 + m];
                ru[it][0] = t4.x; ru[it][1] = t4.y; ru[it][2] = t4.z; ru[it][3] = t4.w;
            }
        }
#endif
#if !defined(SSECG_ABL_NOV)
#pragma unroll
        for (int it = 0; it < VIT; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rd[it][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srcR, voff[it][i], soff, 0));
#endif
    };
    auto store_stage = [&](int buf, int chan0) {   // chan0 = first input channel of the stage held in ru / rd
        if (in_aff) {
            const float2 ab = sAff[chan0 + 4 * vg + ch4];
#pragma unroll
            for (int it = 0; it < VIT; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a = fmaxf(fmaf(rd[it][i], ab.x, ab.y), 0.f);
                    rd[it][i] = (int)voff[it][i] < 0 ? 0.f : a;   // bit 31 = padding / out of range: stays exactly 0
                }
        }
#pragma unroll
        for (int it = 0; it < UF4; ++it)
            if (U_ALL || tid + it * NT < 8 * BM) {
#if defined(SSECG_ABL_NOSTORE)
                asm volatile("" :: "v"(ru[it][0]), "v"(ru[it][1]), "v"(ru[it][2]), "v"(ru[it][3]));
#else
                reinterpret_cast<float4*>(Us0 + buf * U_STAGE)[tid + it * NT] = make_float4(ru[it][0], ru[it][1], ru[it][2], ru[it][3]);
#endif
            }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            float* v = Vs0 + buf * V_STAGE + (vg * BNP + vq0 + VQ_STEP * it) * 4 + ch4;
#if defined(SSECG_ABL_NOXF)      // timing experiment: no input-transform arithmetic
            v[0 * 2 * BNP * 4] = rd[it][0];
            v[1 * 2 * BNP * 4] = rd[it][1];
            v[2 * 2 * BNP * 4] = rd[it][2];
            v[3 * 2 * BNP * 4] = rd[it][3];
#elif defined(SSECG_ABL_NOSTORE) // timing experiment: loads waited for, nothing written to LDS
            asm volatile("" :: "v"(rd[it][0]), "v"(rd[it][1]), "v"(rd[it][2]), "v"(rd[it][3]));
            (void)v;
#else
            v[0 * 2 * BNP * 4] = rd[it][0] - rd[it][2];
            v[1 * 2 * BNP * 4] = rd[it][1] + rd[it][2];
            v[2 * 2 * BNP * 4] = rd[it][2] - rd[it][1];
            v[3 * 2 * BNP * 4] = rd[it][1] - rd[it][3];
#endif
        }
    };
    if (first < p.numQT) {
        tile_offsets(first * BNP);
        load_stage(0, 0u);
    }
    for (int qt = first; qt < p.numQT; qt += step) {
        const int q0 = qt * BNP;
        if ((p.L & 1) && p.stats != nullptr && tid < BNP) {  // odd rows: the last pair's second output does not exist
            const int q = q0 + tid;
            const int jh = q < p.Q ? q % p.Lh : 0;
            sMask[tid] = (2 * jh + 1 < p.L) ? 1.f : 0.f;
        }

        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

        auto mfma_stage = [&](int buf) {
            const float* us = Us0 + buf * U_STAGE + (lhi * BM + wm * 32 + l31) * 4;
            const float* vs = Vs0 + buf * V_STAGE + (lhi * BNP + wn * 32 + l31) * 4;
            // two transform planes at a time: 16 fragment registers live, 8 MFMAs alternating between two accumulators
#pragma unroll
            for (int kk = 0; kk < 4; kk += 2) {
                if (kk) __builtin_amdgcn_sched_barrier(0);
#if defined(SSECG_ABL_NOLDS)
                float4 u0 = make_float4(1.f, 2.f, 3.f, 4.f), v0 = u0, u1 = u0, v1 = u0;
                asm volatile("" : "+v"(u0.x), "+v"(v0.x), "+v"(u1.x), "+v"(v1.x));
#else
                const float4 u0 = *reinterpret_cast<const float4*>(us + kk * 2 * BM * 4);
                const float4 v0 = *reinterpret_cast<const float4*>(vs + kk * 2 * BNP * 4);
                const float4 u1 = *reinterpret_cast<const float4*>(us + (kk + 1) * 2 * BM * 4);
                const float4 v1 = *reinterpret_cast<const float4*>(vs + (kk + 1) * 2 * BNP * 4);
#endif
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.x, u0.x, acc[kk], 0, 0, 0);
                acc[kk + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.x, u1.x, acc[kk + 1], 0, 0, 0);
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.y, u0.y, acc[kk], 0, 0, 0);
                acc[kk + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.y, u1.y, acc[kk + 1], 0, 0, 0);
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.z, u0.z, acc[kk], 0, 0, 0);
                acc[kk + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.z, u1.z, acc[kk + 1], 0, 0, 0);
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.w, u0.w, acc[kk], 0, 0, 0);
                acc[kk + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.w, u1.w, acc[kk + 1], 0, 0, 0);
            }
        };

        // Measured alternatives (round 1, layer4 shape): issuing the next loads after the MFMAs (just before the barrier)
        // instead of before them changed nothing for C >= 128 and cost 10 % on the 8-stage 64-channel tiles.  PMC: the
        // matrix pipe is 75 % busy at the clock the chip actually sustains under this load (~2.1 GHz, not 2.4), TA 26 %
        // busy, no LDS bank conflicts; L2 hit rate 65 % (the misses are the compulsory input stream).
        unsigned soff = 0;
        __syncthreads();  // the previous tile's readers are done with the LDS buffers
        store_stage(0, 0);   // stage 0 was requested before the previous tile's epilogue (or before the loop)
        if (SLOTS == 4) {
            soff += chan_step;
            if (nstages > 1) { load_stage(1, soff); store_stage(1, kWinoKC); }
        }
        __syncthreads();
        constexpr int AHEAD = SLOTS / 2;   // stages between a stage's LDS store and its use
        for (int s = 0; s < nstages; ++s) {
            const bool more = (s + AHEAD) < nstages;
            soff += chan_step;
            WINO_STAMP(0);
#if !defined(SSECG_ABL_NOLOAD)
            if (more) load_stage(s + AHEAD, soff);
#endif
            WINO_STAMP(1);
#if !defined(SSECG_ABL_NOMFMA)
            mfma_stage(s % SLOTS);
#endif
            WINO_STAMP(2);
#if !defined(SSECG_ABL_NOLOAD)
            if (more) store_stage((s + AHEAD) % SLOTS, (s + AHEAD) * kWinoKC);
#endif
            WINO_STAMP(3);
            if (SLOTS == 2 || (s & 1) || s + 1 == nstages) __syncthreads();
            WINO_STAMP(4);
        }

        if (qt + step < p.numQT) {   // next tile's first stage: in flight during this tile's epilogue
            tile_offsets((qt + step) * BNP);
            load_stage(0, 0u);
        }

        // ---------------- epilogue: output transform, statistics, stores ----------------
        // acc[0] <- y[2j] = m0 + m1 + m2,  acc[1] <- y[2j+1] = m1 - m2 - m3   (register r = pair row (r&3)+8(r>>2)+4*lhi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a0 = acc[0][r], a1 = acc[1][r], a2 = acc[2][r], a3 = acc[3][r];
            acc[0][r] = (a0 + a1) + a2;
            acc[1][r] = (a1 - a2) - a3;
        }
        if (p.stats != nullptr) {
            float s = 0.f, q = 0.f;
            if (p.L & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y0 = acc[0][r];
                    const float y1 = acc[1][r] * sMask[wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
                    s += y0 + y1;
                    q = fmaf(y0, y0, q);
                    q = fmaf(y1, y1, q);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y0 = acc[0][r], y1 = acc[1][r];
                    s += y0 + y1;
                    q = fmaf(y0, y0, q);
                    q = fmaf(y1, y1, q);
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
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // pairs 16h..16h+15 of the wave's block = 32 consecutive positions per sample row
                asm volatile("" ::: "memory");
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * h + rr;
                    const int rowl = (rr & 3) + 8 * (rr >> 2) + 4 * lhi;  // pair within the half (0..15)
                    T[l31 * 33 + 2 * rowl + 0] = acc[0][r];
                    T[l31 * 33 + 2 * rowl + 1] = acc[1][r];
                }
                asm volatile("" ::: "memory");
                const int q = q0 + wn * 32 + 16 * h + (l31 >> 1);
                const bool q_ok = q < p.Q;
                const int n = q_ok ? q / p.Lh : 0;
                const int l = 2 * (q - n * p.Lh) + (l31 & 1);
                const bool pok = q_ok && l < p.L;
                const int rbase = m0 + wm * 32 + lhi + opq;
                unsigned o = ((unsigned)n * (unsigned)p.M + (unsigned)rbase) * (unsigned)p.L + (unsigned)l;
                const unsigned ostep = 2u * (unsigned)p.L;
                if (plain) {
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) {
                        const float v = T[(2 * k2 + lhi) * 33 + l31];
                        if (pok) p.out[o] = v;
                        o += ostep;
                    }
                } else {
                    epilogue_rows_fused(T, lhi, l31, pok, rbase, 0x7fffffff, o, ostep, p.scale, p.shift, p.residual, p.relu, p.out);
                }
            }
        }
        __syncthreads();  // the next tile's staging overwrites the transpose tiles (and sMask)
    }

    if (p.stats != nullptr) {
        float* red = smem;  // [WN][BM][2]
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

// u[c/8][k][(c%8)/4][m][c%4] from the taps g[m][c][0..2] = w[m*sm + c*sc + t] (flip: t -> 2 - t)
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int M, int C, int sm, int sc, int flip) {
    const int total = M * C;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int m = e % M, c = e / M;
        const float* g = w + (size_t)m * sm + (size_t)c * sc;
        const float g0 = flip ? g[2] : g[0], g1 = g[1], g2 = flip ? g[0] : g[2];
        const size_t base = ((((size_t)(c >> 3) * 4) * 2 + ((c >> 2) & 1)) * M + m) * 4 + (c & 3);
        const size_t kstep = (size_t)2 * M * 4;
        u[base] = g0;
        u[base + kstep] = ((g0 + g1) + g2) * 0.5f;
        u[base + 2 * kstep] = ((g0 - g1) + g2) * 0.5f;
        u[base + 3 * kstep] = g2;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions in Winograd form (the transpose of F(2,3)).  Per output pair (e0, e1) =
// dy[2j], dy[2j+1] and the four inputs d0..d3 = x[2j-1 .. 2j+2]:
//     E = (e0, e0+e1, e0-e1, -e1)       V = (d0-d2, d1+d2, d2-d1, d1-d3)       M_k[co][ci] = sum_{n,j} E_k * V_k
//     dw[.,.,0] = M0 + (M1+M2)/2      dw[.,.,1] = (M1-M2)/2      dw[.,.,2] = (M1+M2)/2 + M3
// - four GEMMs [Cout x Cin] whose depth is the number of PAIRS (half the positions): 4 instead of 6 multiplications per
// (co, ci, pair).  One 16-wave workgroup per CU owns a 128 x 128 tile of all four planes (a wave: 32 x 32 x 4 planes)
// over one slab of pairs; slabs are summed (fixed order) and recombined by wino_wgrad_reduce_kernel.  Both operands
// are transformed on the fly while staging 16 pairs per LDS stage; fragments are 16-byte LDS reads (4 pairs).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWgKP = 16;                        // pairs per stage
#ifndef SSECG_WG_PAD
#define SSECG_WG_PAD 8
#endif
// floats per (plane, pair-quad) block: 128 rows x 4 pairs + a pad that spreads a staging store over the 32 LDS banks: a half-wave
// writes (quad q = 0..3) x (2 rows) x (4 pairs) and lands in bank 8 q + 4 row + pair.  MEASURED (PMC, round 3): with the former
// pad of 16 (chosen for 64 banks) quads 0 / 2 and 1 / 3 collided - SQ_LDS_BANK_CONFLICT was 25 % of the LDS-active cycles.
constexpr int kWgBlk = 128 * 4 + SSECG_WG_PAD;

struct WinoWgP {
    const float* dy;  // (N, Cout, L)
    const float* x;   // (N, Cin, L)
    float* ws;        // [Z][4][Cout][Cin]
    unsigned dy_bytes, x_bytes;
    int Cout, Cin, L, Lh, MT, JT, Z;
    long long Q, chunk;  // pairs; pairs per slab (multiple of kWgKP)
    // the x operand is relu(x*x_scale[ci] + x_shift[ci]) (fused producer BN + ReLU), or nullptr
    const float* x_scale;
    const float* x_shift;
};

__global__ __launch_bounds__(1024, 4) void conv_wino_wgrad_kernel(WinoWgP p) {
    constexpr int PLANE = 4 * kWgBlk;  // floats per transform plane (4 pair-quads)
    __shared__ __attribute__((aligned(16))) float sE[4 * PLANE];
    __shared__ __attribute__((aligned(16))) float sV[4 * PLANE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wj = wave & 3;
    const int l31 = lane & 31, lhi = lane >> 5;
    // XCD-aware slab order (see conv_wgrad_kernel): all tiles of one pair slab run on one XCD and share its L2
    const int tiles = p.MT * p.JT;
    const int slot = blockIdx.x >> 3;
    const int zslab = (slot / tiles) * 8 + (blockIdx.x & 7);
    if (zslab >= p.Z) return;
    const int tile = slot % tiles;
    const int j0 = (tile % p.JT) * 128, m0 = (tile / p.JT) * 128;
    const long long kbeg = (long long)zslab * p.chunk;
    long long kend = kbeg + p.chunk;
    if (kend > p.Q) kend = p.Q;
    const int nstages = kend > kbeg ? (int)((kend - kbeg + kWgKP - 1) / kWgKP) : 0;

    // staging: lane = (pair pp = lane & 15, row-in-group rl = lane >> 4); item it: row = it*64 + wave*4 + rl
    const int pp = lane & 15, rl = lane >> 4;
    const auto dyR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
    const auto xR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const int wrow = __builtin_amdgcn_readfirstlane(wave) * 4;
    const unsigned soffE0 = (unsigned)((m0 + wrow) * p.L) * 4u, soffE1 = (unsigned)((m0 + 64 + wrow) * p.L) * 4u;
    const unsigned soffV0 = (unsigned)((j0 + wrow) * p.L) * 4u, soffV1 = (unsigned)((j0 + 64 + wrow) * p.L) * 4u;
    // this lane's pair in the stage being loaded: (n, jh), advanced by kWgKP per stage without a division
    long long q = kbeg + pp;
    int n = (int)(q / p.Lh), jh = (int)(q - (long long)n * p.Lh);

    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    float re[2][2], rd[2][4];
    const bool x_aff = p.x_scale != nullptr;
    float xsc[2] = {1.f, 1.f}, xsh[2] = {0.f, 0.f};
    if (x_aff) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            xsc[it] = p.x_scale[j0 + it * 64 + wave * 4 + rl];
            xsh[it] = p.x_shift[j0 + it * 64 + wave * 4 + rl];
        }
    }
    unsigned vvalid = 0;  // validity of the four x samples of the stage held in rd (bit i)
    unsigned vo[4];
    auto load_stage_e = [&]() {
        const bool ok = q < kend;
        const int l0 = 2 * jh;
        const unsigned eb = ((unsigned)n * (unsigned)p.Cout + (unsigned)rl) * (unsigned)p.L + (unsigned)l0;
        const unsigned vb = ((unsigned)n * (unsigned)p.Cin + (unsigned)rl) * (unsigned)p.L + (unsigned)l0;
        const unsigned e0o = oob_if(eb * 4u, !ok), e1o = oob_if((eb + 1u) * 4u, !(ok && l0 + 1 < p.L));
        vvalid = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool v_ok = ok && (unsigned)(l0 - 1 + i) < (unsigned)p.L;
            vo[i] = oob_if((vb + (unsigned)(i - 1)) * 4u, !v_ok);
            vvalid |= (unsigned)v_ok << i;
        }
        re[0][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyR, e0o, soffE0, 0));
        re[0][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyR, e1o, soffE0, 0));
        re[1][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyR, e0o, soffE1, 0));
        re[1][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyR, e1o, soffE1, 0));
        q += kWgKP;
        jh += kWgKP;
        while (jh >= p.Lh) { jh -= p.Lh; ++n; }
    };
    // the x loads of the next stage are requested between the two halves of the stage's MFMAs, not in one burst with the
    // dy loads (16 lock-stepped waves otherwise queue behind the address unit while the matrix pipe idles)
    auto load_stage_v = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rd[0][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xR, vo[i], soffV0, 0));
            rd[1][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xR, vo[i], soffV1, 0));
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int o = (pp >> 2) * kWgBlk + (it * 64 + wave * 4 + rl) * 4 + (pp & 3);
            if (x_aff) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a = fmaxf(fmaf(rd[it][i], xsc[it], xsh[it]), 0.f);
                    rd[it][i] = ((vvalid >> i) & 1u) ? a : 0.f;
                }
            }
            const float e0 = re[it][0], e1 = re[it][1];
            sE[0 * PLANE + o] = e0;
            sE[1 * PLANE + o] = e0 + e1;
            sE[2 * PLANE + o] = e0 - e1;
            sE[3 * PLANE + o] = -e1;
            sV[0 * PLANE + o] = rd[it][0] - rd[it][2];
            sV[1 * PLANE + o] = rd[it][1] + rd[it][2];
            sV[2 * PLANE + o] = rd[it][2] - rd[it][1];
            sV[3 * PLANE + o] = rd[it][1] - rd[it][3];
        }
    };

    if (nstages > 0) { load_stage_e(); load_stage_v(); }
    for (int s = 0; s < nstages; ++s) {
        __syncthreads();  // readers of the previous stage are done
        store_stage();
        __syncthreads();
        const bool more = s + 1 < nstages;
        if (more) load_stage_e();
        const float* es = sE + (wm * 32 + l31) * 4;
        const float* vs = sV + (wj * 32 + l31) * 4;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && more) load_stage_v();
            const int qo = (2 * r + lhi) * kWgBlk;  // lane half h takes pair quad 2r + h: pairs 4(2r+h) .. +3
#pragma unroll
            for (int kk = 0; kk < 4; kk += 2) {
                const float4 e0 = *reinterpret_cast<const float4*>(es + kk * PLANE + qo);
                const float4 v0 = *reinterpret_cast<const float4*>(vs + kk * PLANE + qo);
                const float4 e1 = *reinterpret_cast<const float4*>(es + (kk + 1) * PLANE + qo);
                const float4 v1 = *reinterpret_cast<const float4*>(vs + (kk + 1) * PLANE + qo);
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0.x, v0.x, acc[kk], 0, 0, 0);
                acc[kk + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1.x, v1.x, acc[kk + 1], 0, 0, 0);
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0.y, v0.y, acc[kk], 0, 0, 0);
                acc[kk + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1.y, v1.y, acc[kk + 1], 0, 0, 0);
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0.z, v0.z, acc[kk], 0, 0, 0);
                acc[kk + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1.z, v1.z, acc[kk + 1], 0, 0, 0);
                acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0.w, v0.w, acc[kk], 0, 0, 0);
                acc[kk + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1.w, v1.w, acc[kk + 1], 0, 0, 0);
            }
        }
    }

    // slab store: ws[z][k][co][ci]; accumulator row (register) = co, column (lane) = ci
    const size_t plane = (size_t)p.Cout * p.Cin;
    float* ws = p.ws + (size_t)zslab * 4 * plane;
    const int col = j0 + wj * 32 + l31;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            ws[k * plane + (size_t)row * p.Cin + col] = acc[k][r];
        }
}

// dw[co][ci][0..2] from the slab sums of the four planes (fixed summation order -> reproducible)
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Z, int Cout,
                                                                int Cin) {
    // 64 elements per workgroup x 4 slab lanes: wave zl sums the slabs zl, zl+4, ... of its 64 elements (coalesced), the
    // four partial sums are combined in a fixed order through LDS.  (One thread per element left the 128-channel layers
    // with 64 workgroups for a 67 MB read.)
    __shared__ float part[4][4][64];
    const size_t plane = (size_t)Cout * Cin;
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < plane; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + el;
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        if (e < plane)
            for (int z = zl; z < Z; z += 4) {
                const float* w = ws + (size_t)z * 4 * plane + e;
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] += w[k * plane];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) part[zl][k][el] = m[k];
        __syncthreads();
        if (zl == 0 && e < plane) {
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = (part[0][k][el] + part[1][k][el]) + (part[2][k][el] + part[3][k][el]);
            const float hs = (m[1] + m[2]) * 0.5f, hd = (m[1] - m[2]) * 0.5f;
            dw[e * 3 + 0] = m[0] + hs;
            dw[e * 3 + 1] = hd;
            dw[e * 3 + 2] = hs + m[3];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same weight gradient as the transpose of F(4,3) ("F(3,4)": three taps from four neighbouring output gradients and six
// inputs with 6 multiplications per (co, ci) instead of 12; 3/4 of the F(2,3) form's).  Per output quad e0..e3 = dy[4j .. 4j+3]
// and the six inputs d0..d5 = x[4j-1 .. 4j+4]:
//     E = (e0/4, -(e0+e1+e2+e3)/6, (-e0+e1-e2+e3)/6, e0/24+e1/12+e2/6+e3/3, e0/24-e1/12+e2/6-e3/3, e3)
//     V = (4d0-5d2+d4, -4d1-4d2+d3+d4, 4d1-4d2-d3+d4, -2d1-d2+2d3+d4, 2d1-d2-2d3+d4, 4d1-5d3+d5)      M_k[co][ci] = sum_{n,j} E_k V_k
//     dw[.,.,0] = M0+M1+M2+M3+M4      dw[.,.,1] = (M1-M2) + 2 (M3-M4)      dw[.,.,2] = (M1+M2) + 4 (M3+M4) + M5
// Six GEMMs [Cout x Cin] whose depth is the number of QUADS.  fp32 error against an fp64 gradient: 2e-6 of the tensor's scale on
// 2048-quad chains (1.6 - 2.2 x the F(2,3) form's, simulated with the kernel's accumulation order before it was built).
// An 8-wave workgroup owns a 64 (co) x 64 (ci) tile of all six planes - waves 0-3 planes 0-2, waves 4-7 planes 3-5, a wave
// 32 x 32 x 3 planes (48 accumulator registers) - and TWO workgroups share a CU: while one transforms and stores its next stage
// between its two barriers the other keeps the matrix pipe busy (one 16-wave workgroup per CU, 128 x 64: 0.60 of the pipe's peak;
// its double-buffered form spilled into the loop and was slower still; so was a spill-free double-buffered form of THIS tile with
// 8-position items and one barrier per 12 MFMAs: 576 against 503 us on the 512-channel layer; staggering the two workgroups of a
// CU by up to a stage changed nothing - they drift apart by themselves).  A stage is 4 GROUPS of 16 positions of one row (samples
// are padded to whole groups); every thread owns one (row, group) of one operand: 4 unaligned 16-byte buffer loads (+ 2 dwords
// of halo for x), the four quads' transforms in registers, one 16-byte LDS store per plane - the layout the fragment reads want.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kW4Blk = 64 * 4 + 8;   // floats per (plane, group) block: 64 rows x 4 quads + the bank-spreading pad (264 % 32 == 8)
typedef unsigned u32x4w __attribute__((__vector_size__(16)));
typedef float f32x4w __attribute__((ext_vector_type(4)));   // one LLVM <4 x float>: a 16-byte LDS access that is never scalarised

struct WinoWg4P {
    const float* dy;  // (N, Cout, L)
    const float* x;   // (N, Cin, L)
    float* ws;        // [Z][3][Cout][Cin]
    unsigned dy_bytes, x_bytes;
    int Cout, Cin, L, Lg, MT, JT, Z;
    long long Q, chunk;  // groups (N * Lg); groups per slab (multiple of 4)
    const float* x_scale;
    const float* x_shift;
};

__global__ __launch_bounds__(512, 4) void conv_wino_wgrad4_kernel(WinoWg4P p) {
    constexpr int BLK = kW4Blk / 4;      // a (plane, group) block in 16-byte units
    constexpr int PL = 4 * BLK;          // a transform plane (4 groups)
    __shared__ f32x4w smem4[2 * 6 * PL]; // 2 x 25344 B: two workgroups per CU
    f32x4w* const sE = smem4;
    f32x4w* const sV = smem4 + 6 * PL;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave >> 2, wm = (wave >> 1) & 1, wj = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tiles = p.MT * p.JT;
    const int slot = blockIdx.x >> 3;
    const int zslab = (slot / tiles) * 8 + (blockIdx.x & 7);   // XCD-aware slab order, as conv_wino_wgrad_kernel
    if (zslab >= p.Z) return;
    const int tile = slot % tiles;
    const int j0c = (tile % p.JT) * 64, m0 = (tile / p.JT) * 64;
    const long long kbeg = (long long)zslab * p.chunk;
    long long kend = kbeg + p.chunk;
    if (kend > p.Q) kend = p.Q;
    const int nstages = kend > kbeg ? (int)((kend - kbeg + 3) / 4) : 0;

    // staging roles (wave-uniform): waves 0-3 the dy operand, waves 4-7 the x operand; item = (row, group) = (item >> 2, item & 3)
    const bool roleE = wave < 4;
    const int item = tid & 255;
    const int sg = item & 3, srow = item >> 2;
    const int chan = roleE ? m0 + srow : j0c + srow;
    const int nch = roleE ? p.Cout : p.Cin;
    const auto R = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(roleE ? p.dy : p.x), 0, (int)(roleE ? p.dy_bytes : p.x_bytes), 0x00020000);
    const int cnt = (int)(kend - kbeg);   // groups of this slab
    int rel = sg;                         // this lane's group of the stage being loaded, relative to the slab
    int n = (int)((kbeg + sg) / p.Lg), gi = (int)((kbeg + sg) - (long long)n * p.Lg);
    float xsc = 1.f, xsh = 0.f;
    const bool x_aff = p.x_scale != nullptr;
    if (!roleE && x_aff) { xsc = p.x_scale[chan]; xsh = p.x_shift[chan]; }

    f32x16 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    float raw[18];     // positions j0-1 .. j0+16 of the item being loaded (raw[0] and raw[17]: the x operand only)
    unsigned cur_off = 0;
    int lim = 0;       // valid positions from j0 on (0: the whole item lies outside the slab)
    bool left = false; // position j0 - 1 lies inside the row
    auto ld4 = [&](int u) {
        const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(R, cur_off + 16u * (unsigned)u, 0, 0);
        const unsigned v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
        raw[1 + 4 * u] = __uint_as_float(v0); raw[2 + 4 * u] = __uint_as_float(v1);
        raw[3 + 4 * u] = __uint_as_float(v2); raw[4 + 4 * u] = __uint_as_float(v3);
    };
    auto load_a = [&]() {   // first half of the item's loads
        const bool ok = rel < cnt;
        const int j0 = 16 * gi;
        lim = ok ? p.L - j0 : 0;
        left = ok && j0 > 0;
        cur_off = oob_if((((unsigned)n * (unsigned)nch + (unsigned)chan) * (unsigned)p.L + (unsigned)j0) * 4u, !ok);
        ld4(0); ld4(1);
        if (!roleE) raw[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R, cur_off - 4u, 0, 0));
        rel += 4;
        gi += 4;
        while (gi >= p.Lg) { gi -= p.Lg; ++n; }
    };
    auto load_b = [&]() {
        ld4(2); ld4(3);
        if (!roleE) raw[17] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R, cur_off + 64u, 0, 0));
    };
    auto store_stage = [&]() {
        const int nv = lim;
        float o[6][4];
        if (roleE) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float e0 = 4 * qd + 0 < nv ? raw[1 + 4 * qd] : 0.f, e1 = 4 * qd + 1 < nv ? raw[2 + 4 * qd] : 0.f;
                const float e2 = 4 * qd + 2 < nv ? raw[3 + 4 * qd] : 0.f, e3 = 4 * qd + 3 < nv ? raw[4 + 4 * qd] : 0.f;
                const float a = e0 + e2, b = e1 + e3;
                const float c = e0 * (1.f / 24.f) + e2 * (1.f / 6.f), d = e1 * (1.f / 12.f) + e3 * (1.f / 3.f);
                o[0][qd] = e0 * 0.25f;
                o[1][qd] = (a + b) * (-1.f / 6.f);
                o[2][qd] = (b - a) * (1.f / 6.f);
                o[3][qd] = c + d;
                o[4][qd] = c - d;
                o[5][qd] = e3;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 18; ++i) {   // position j0 - 1 + i: inside the row?  (padding is applied AFTER the fused activation)
                float a = raw[i];
                if (x_aff) a = fmaxf(fmaf(a, xsc, xsh), 0.f);
                const bool ok = i == 0 ? left : (i - 1 < nv);
                raw[i] = ok ? a : 0.f;
            }
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float d0 = raw[4 * qd], d1 = raw[4 * qd + 1], d2 = raw[4 * qd + 2], d3 = raw[4 * qd + 3], d4 = raw[4 * qd + 4],
                            d5 = raw[4 * qd + 5];
                const float s42 = d4 - 4.f * d2, s31 = d3 - 4.f * d1, t42 = d4 - d2, t31 = 2.f * (d3 - d1);
                o[0][qd] = (4.f * d0 - 5.f * d2) + d4;
                o[1][qd] = s42 + s31;
                o[2][qd] = s42 - s31;
                o[3][qd] = t42 + t31;
                o[4][qd] = t42 - t31;
                o[5][qd] = (4.f * d1 - 5.f * d3) + d5;
            }
        }
        f32x4w* dst = (roleE ? sE : sV) + sg * BLK + srow;
#pragma unroll
        for (int k = 0; k < 6; ++k) dst[k * PL] = f32x4w{o[k][0], o[k][1], o[k][2], o[k][3]};
    };

    if (nstages > 0) { load_a(); load_b(); }
    const f32x4w* es = sE + 3 * pg * PL + (wm * 32 + l31);
    const f32x4w* vs = sV + 3 * pg * PL + (wj * 32 + l31);
    for (int s = 0; s < nstages; ++s) {
        __syncthreads();  // readers of the previous stage are done
        store_stage();
        __syncthreads();
        const bool more = s + 1 < nstages;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (more) { if (r == 0) load_a(); else load_b(); }
            const int qo = (2 * r + lhi) * BLK;   // lane half h takes group 2r + h: its four quads
            const f32x4w e0 = es[qo], v0 = vs[qo];
            const f32x4w e1 = es[PL + qo], v1 = vs[PL + qo];
            const f32x4w e2 = es[2 * PL + qo], v2 = vs[2 * PL + qo];
            __builtin_amdgcn_s_setprio(1);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0.x, v0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1.x, v1.x, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(e2.x, v2.x, acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0.y, v0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1.y, v1.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(e2.y, v2.y, acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0.z, v0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1.z, v1.z, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(e2.z, v2.z, acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0.w, v0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1.w, v1.w, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(e2.w, v2.w, acc[2], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }

    // Output transform per slab (it is linear: the slab sums of the three taps are what the reduction needs), the two plane groups
    // combined through LDS: HALF the workspace traffic of six planes per slab.  taps = (M0+M1+M2, M1-M2, M1+M2) + (M3+M4, 2(M3-M4),
    // 4(M3+M4)+M5).  ws[z][tap][co][ci]; accumulator row (register) = co, column (lane) = ci
    float* const X = reinterpret_cast<float*>(smem4) + (wave & 3) * (3 * 16 * 64);   // [tap][register][lane] of the wave pair
    __syncthreads();   // the last stage's fragment reads are done
    if (pg == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s34 = acc[0][r] + acc[1][r], d34 = acc[0][r] - acc[1][r];
            X[(0 * 16 + r) * 64 + lane] = s34;
            X[(1 * 16 + r) * 64 + lane] = 2.f * d34;
            X[(2 * 16 + r) * 64 + lane] = 4.f * s34 + acc[2][r];
        }
    }
    __syncthreads();
    if (pg == 0) {
        const size_t plane = (size_t)p.Cout * p.Cin;
        float* ws = p.ws + (size_t)zslab * 3 * plane;
        const int col = j0c + wj * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
            float* o = ws + (size_t)row * p.Cin + col;
            o[0] = (acc[0][r] + s12) + X[(0 * 16 + r) * 64 + lane];
            o[plane] = d12 + X[(1 * 16 + r) * 64 + lane];
            o[2 * plane] = s12 + X[(2 * 16 + r) * 64 + lane];
        }
    }
}

// dw[co][ci][0..2] = the slab sums of the three tap planes in a fixed order (reproducible): 64 elements x ZL slab lanes per workgroup
template <int ZL>
__global__ __launch_bounds__(64 * ZL) void wino_wgrad4_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Z, int Cout,
                                                                    int Cin) {
    __shared__ float part[ZL][3][64];
    const size_t plane = (size_t)Cout * Cin;
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < plane; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + el;
        float m[3] = {0.f, 0.f, 0.f};
        if (e < plane)
            for (int z = zl; z < Z; z += ZL) {
                const float* w = ws + (size_t)z * 3 * plane + e;
#pragma unroll
                for (int k = 0; k < 3; ++k) m[k] += w[k * plane];
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) part[zl][k][el] = m[k];
        __syncthreads();
        if (zl < 3 && e < plane) {   // wave t finishes tap t
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < ZL; q += 4) v += (part[q][zl][el] + part[q + 1][zl][el]) + (part[q + 2][zl][el] + part[q + 3][zl][el]);
            dw[e * 3 + zl] = v;
        }
        __syncthreads();
    }
}

struct WinoWgCfg { int MT, JT, Z; long long chunk; };

inline WinoWgCfg pick_wino_wgrad4(int Cout, int Cin, long long Q) {   // Q = groups of 16 positions
    WinoWgCfg c;
    c.MT = Cout / 64; c.JT = Cin / 64;
    const int tiles = c.MT * c.JT;
    int z = 2 * kNumCU / tiles;       // two 8-wave workgroups per CU
    if (z > Q / 16) z = (int)(Q / 16);   // ... of at least four stages each (a slab costs 3 x 16 KB of workspace traffic per tile)
    z = (z / 8) * 8;                  // whole groups of 8 slabs (one per XCD)
    if (z < 8) z = 8;
    long long chunk = (Q + z - 1) / z;
    chunk = ((chunk + 3) / 4) * 4;
    c.Z = (int)((Q + chunk - 1) / chunk);
    c.chunk = chunk;
    return c;
}


inline WinoWgCfg pick_wino_wgrad(int Cout, int Cin, long long Q) {
    WinoWgCfg c;
    c.MT = Cout / 128; c.JT = Cin / 128;
    const int tiles = c.MT * c.JT;
    int z = kNumCU / tiles;           // one 16-wave workgroup per CU
    z = (z / 8) * 8;                  // whole groups of 8 slabs (one per XCD)
    if (z < 8) z = 8;
    long long chunk = (Q + z - 1) / z;
    chunk = ((chunk + kWgKP - 1) / kWgKP) * kWgKP;
    c.Z = (int)((Q + chunk - 1) / chunk);
    c.chunk = chunk;
    return c;
}

inline bool wino_wgrad_ok(int N, int Cin, int L, int Cout, int gran = 128) {
    if (N <= 0 || Cin <= 0 || L <= 0 || Cout <= 0 || Cin % gran != 0 || Cout % gran != 0) return false;
    return (size_t)N * Cin * L * 4 < 0x7fffff00ull && (size_t)N * Cout * L * 4 < 0x7fffff00ull;
}

// all registered weights in one launch: row = {w, u_fwd, u_transposed, Cout, Cin}; blockIdx.y = tensor
__global__ void wino_weight_multi_kernel(const int64_t* __restrict__ table) {
    const int64_t* row = table + 5 * (size_t)blockIdx.y;
    const float* w = reinterpret_cast<const float*>(row[0]);
    float* uf = reinterpret_cast<float*>(row[1]);
    float* ut = reinterpret_cast<float*>(row[2]);
    const int Cout = (int)row[3], Cin = (int)row[4];
    const int total = Cout * Cin;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int ci = e % Cin, co = e / Cin;
        const float g0 = w[(size_t)e * 3], g1 = w[(size_t)e * 3 + 1], g2 = w[(size_t)e * 3 + 2];
        const float s1 = ((g0 + g1) + g2) * 0.5f, s2 = ((g0 - g1) + g2) * 0.5f;
        if (uf) {   // forward operand: m = co, c = ci
            const size_t b = ((((size_t)(ci >> 3) * 4) * 2 + ((ci >> 2) & 1)) * Cout + co) * 4 + (ci & 3), ks = (size_t)2 * Cout * 4;
            uf[b] = g0; uf[b + ks] = s1; uf[b + 2 * ks] = s2; uf[b + 3 * ks] = g2;
        }
        if (ut) {   // data-gradient operand: m = ci, c = co, taps flipped (g0 <-> g2; s1 is symmetric, s2 too)
            const size_t b = ((((size_t)(co >> 3) * 4) * 2 + ((co >> 2) & 1)) * Cin + ci) * 4 + (co & 3), ks = (size_t)2 * Cin * 4;
            ut[b] = g2; ut[b + ks] = ((g2 + g1) + g0) * 0.5f; ut[b + 2 * ks] = ((g2 - g1) + g0) * 0.5f; ut[b + 3 * ks] = g0;
        }
    }
}

struct WinoCfg { int NT, BM, BNP, numQT, MT, G; };

inline WinoCfg pick_wino(int M, long long Q) {
    WinoCfg c;
    c.NT = 1024;   // 16-wave workgroups (one per CU) halve the weight traffic per MFMA ...
    // small problems (few windows per launch): the 16-wave tiles cannot give every CU a workgroup - use the 8-wave tiles
    // (twice as many workgroups, two per CU)
    if (c.NT == 1024) {
        const long long tiles16 = ((Q + (M % 128 == 0 ? 128 : 256) - 1) / (M % 128 == 0 ? 128 : 256)) * (M / (M % 128 == 0 ? 128 : 64));
        if (tiles16 < kNumCU) c.NT = 512;
    }
    const int pairs = c.NT == 1024 ? 2 : 1;
    if (M % 128 == 0) { c.BM = 128; c.BNP = 64 * pairs; }
    else { c.BM = 64; c.BNP = 128 * pairs; }
    c.numQT = (int)((Q + c.BNP - 1) / c.BNP);
    c.MT = M / c.BM;
    int g = (kNumCU * (c.NT == 1024 ? 1 : 2)) / c.MT;
    if (g < 1) g = 1;
    c.G = c.numQT < g ? c.numQT : g;
    return c;
}

inline bool wino_shape_ok(int N, int C, int L, int M) {
    if (N <= 0 || C <= 0 || L <= 0 || M <= 0) return false;
    if (C % kWinoKC != 0 || M % 64 != 0) return false;
    const long long Q = (long long)N * ((L + 1) / 2);
    if (Q > 0x7fffffffLL) return false;
    return (size_t)N * C * L * 4 < 0x7fffff00ull && (size_t)N * M * L * 4 < 0x7fffff00ull;
}

}  // namespace

extern "C" {

#if defined(SSECG_WINO_TRACE)
int ssecg_debug_wino_trace(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_trace), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

int ssecg_conv1d_wino_supported(int N, int C, int L, int M) { return wino_shape_ok(N, C, L, M) ? 1 : 0; }

int ssecg_conv1d_wino_parts(int N, int L, int M) {
    if (N <= 0 || L <= 0 || M <= 0 || M % 64 != 0) return SSECG_E_INVAL;
    return pick_wino(M, (long long)N * ((L + 1) / 2)).G;
}

int ssecg_conv1d_wino_weight(const float* w, float* u, int Cout, int Cin, int transposed, void* stream) {
    if (!w || !u || Cout <= 0 || Cin <= 0) return SSECG_E_INVAL;
    const int M = transposed ? Cin : Cout, C = transposed ? Cout : Cin;
    if (C % kWinoKC != 0) return SSECG_E_INVAL;
    const int sm = transposed ? 3 : Cin * 3, sc = transposed ? Cin * 3 : 3;
    int blocks = (M * C + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, u, M, C, sm, sc, transposed ? 1 : 0);
    return (int)hipGetLastError();
}

int ssecg_conv1d_wino_weight_multi(const int64_t* table, int ntensors, int max_elems, void* stream) {
    if (!table || ntensors <= 0 || max_elems <= 0) return SSECG_E_INVAL;
    int bx = (max_elems + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(wino_weight_multi_kernel, dim3(bx, ntensors), dim3(256), 0, (hipStream_t)stream, table);
    return (int)hipGetLastError();
}

int ssecg_conv1d_wino(const float* src, const float* u, float* out, int N, int C, int L, int M, const float* scale,
                      const float* shift, const float* residual, int relu, float* stats_partial, int stats_parts,
                      const float* in_scale, const float* in_shift, void* stream) {
    if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale != nullptr && C > 512)) return SSECG_E_INVAL;
    if (!src || !u || !out || !wino_shape_ok(N, C, L, M) || (((uintptr_t)u) & 15) != 0) return SSECG_E_INVAL;
    const int Lh = (L + 1) / 2;
    const long long Q = (long long)N * Lh;
    const WinoCfg c = pick_wino(M, Q);
    if (stats_partial != nullptr) {
        if (stats_parts < c.G) return SSECG_E_WORKSPACE;
        if (stats_parts > c.G) {
            const hipError_t e = hipMemsetAsync(stats_partial + (size_t)c.G * M * 2, 0,
                                                (size_t)(stats_parts - c.G) * M * 2 * sizeof(float), (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    WinoP p;
    p.U = u; p.src = src; p.out = out;
    p.M = M; p.C = C; p.L = L; p.Lh = Lh; p.Q = (int)Q; p.numQT = c.numQT;
    p.src_bytes = (unsigned)((size_t)N * C * L * 4);
    p.scale = scale; p.shift = shift; p.residual = residual; p.relu = relu; p.stats = stats_partial;
    p.in_scale = in_scale; p.in_shift = in_shift;
    dim3 grid(c.G, c.MT), block(c.NT);
    hipStream_t st = (hipStream_t)stream;
    if (c.NT == 1024) {
        if (c.BM == 128) hipLaunchKernelGGL((conv_wino_kernel<4, 4>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_wino_kernel<2, 8>), grid, block, 0, st, p);
    } else {
        if (c.BM == 128) hipLaunchKernelGGL((conv_wino_kernel<4, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_wino_kernel<2, 4>), grid, block, 0, st, p);
    }
    return (int)hipGetLastError();
}

int ssecg_conv1d_wino_wgrad_supported(int N, int Cin, int L, int Cout) { return wino_wgrad_ok(N, Cin, L, Cout) ? 1 : 0; }

int ssecg_conv1d_wino_wgrad4_supported(int N, int Cin, int L, int Cout) { return wino_wgrad_ok(N, Cin, L, Cout, 64) ? 1 : 0; }

size_t ssecg_conv1d_wino_wgrad4_workspace(int N, int Cin, int L, int Cout) {
    if (!wino_wgrad_ok(N, Cin, L, Cout, 64)) return 0;
    const WinoWgCfg c = pick_wino_wgrad4(Cout, Cin, (long long)N * ((L + 15) / 16));
    return (size_t)c.Z * 3 * Cout * Cin * sizeof(float);
}

int ssecg_conv1d_wino_wgrad4(const float* dy, const float* x, float* dw, int N, int Cin, int L, int Cout, void* workspace,
                             size_t workspace_bytes, const float* x_scale, const float* x_shift, void* stream) {
    if ((x_scale == nullptr) != (x_shift == nullptr)) return SSECG_E_INVAL;
    if (!dy || !x || !dw || !workspace || !wino_wgrad_ok(N, Cin, L, Cout, 64)) return SSECG_E_INVAL;
    const int Lg = (L + 15) / 16;
    const long long Q = (long long)N * Lg;
    const WinoWgCfg c = pick_wino_wgrad4(Cout, Cin, Q);
    if (workspace_bytes < (size_t)c.Z * 3 * Cout * Cin * sizeof(float)) return SSECG_E_WORKSPACE;
    WinoWg4P p;
    p.dy = dy; p.x = x; p.ws = (float*)workspace;
    p.dy_bytes = (unsigned)((size_t)N * Cout * L * 4); p.x_bytes = (unsigned)((size_t)N * Cin * L * 4);
    p.Cout = Cout; p.Cin = Cin; p.L = L; p.Lg = Lg; p.MT = c.MT; p.JT = c.JT; p.Z = c.Z; p.Q = Q; p.chunk = c.chunk;
    p.x_scale = x_scale; p.x_shift = x_shift;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = c.MT * c.JT;
    const int groups = (c.Z + 7) / 8;
    hipLaunchKernelGGL(conv_wino_wgrad4_kernel, dim3(groups * tiles * 8), dim3(512), 0, st, p);
    const int plane = Cout * Cin, blocks = (plane + 63) / 64;
    if (blocks < 2 * kNumCU && c.Z >= 64)   // few elements, many slabs (the 64- and 128-channel layers): 16 slab lanes per workgroup
        hipLaunchKernelGGL(wino_wgrad4_reduce_kernel<16>, dim3(blocks), dim3(1024), 0, st, p.ws, dw, c.Z, Cout, Cin);
    else
        hipLaunchKernelGGL(wino_wgrad4_reduce_kernel<4>, dim3(blocks), dim3(256), 0, st, p.ws, dw, c.Z, Cout, Cin);
    return (int)hipGetLastError();
}


size_t ssecg_conv1d_wino_wgrad_workspace(int N, int Cin, int L, int Cout) {
    if (!wino_wgrad_ok(N, Cin, L, Cout)) return 0;
    const WinoWgCfg c = pick_wino_wgrad(Cout, Cin, (long long)N * ((L + 1) / 2));
    return (size_t)c.Z * 4 * Cout * Cin * sizeof(float);
}

int ssecg_conv1d_wino_wgrad(const float* dy, const float* x, float* dw, int N, int Cin, int L, int Cout, void* workspace,
                            size_t workspace_bytes, const float* x_scale, const float* x_shift, void* stream) {
    if ((x_scale == nullptr) != (x_shift == nullptr)) return SSECG_E_INVAL;
    if (!dy || !x || !dw || !workspace || !wino_wgrad_ok(N, Cin, L, Cout)) return SSECG_E_INVAL;
    const int Lh = (L + 1) / 2;
    const long long Q = (long long)N * Lh;
    const WinoWgCfg c = pick_wino_wgrad(Cout, Cin, Q);
    if (workspace_bytes < (size_t)c.Z * 4 * Cout * Cin * sizeof(float)) return SSECG_E_WORKSPACE;
    WinoWgP p;
    p.dy = dy; p.x = x; p.ws = (float*)workspace;
    p.dy_bytes = (unsigned)((size_t)N * Cout * L * 4); p.x_bytes = (unsigned)((size_t)N * Cin * L * 4);
    p.Cout = Cout; p.Cin = Cin; p.L = L; p.Lh = Lh; p.MT = c.MT; p.JT = c.JT; p.Z = c.Z; p.Q = Q; p.chunk = c.chunk;
    p.x_scale = x_scale; p.x_shift = x_shift;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = c.MT * c.JT;
    const int groups = (c.Z + 7) / 8;
    hipLaunchKernelGGL(conv_wino_wgrad_kernel, dim3(groups * tiles * 8), dim3(1024), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const size_t plane = (size_t)Cout * Cin;
    int blocks = (int)((plane + 63) / 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, c.Z, Cout, Cin);
    return (int)hipGetLastError();
}

}  // extern "C"
