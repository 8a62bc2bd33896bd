// The ResNet stem convolution - Conv1d(C -> 64, k = 7, stride 2, pad 3), C = 1..16 input leads - as its own kernels
// (reference: src/models/backbones/resnet.py:245-257, 354-355).
//
// Why not the implicit-GEMM kernels of conv.hip: with K = 7*C <= 112 the whole weight matrix (<= 28 KB) and the input
// rows of a 256-position tile (<= 34 KB) fit in LDS together, so nothing is gathered from global memory inside the K loop
// and nothing is re-read: the generic kernel ran this shape at 0.37 of its roofline (round-1 profile), its weight
// gradient at 0.34.  The stride-2 gather becomes unit-stride LDS reads by staging the input rows de-interleaved:
//     xe[i] = x[2 (j0 - 3 + i)],  xo[i] = x[2 (j0 - 3 + i) + 1]          (zero outside [0, L): the conv's padding)
//     tap t of output j0 + jl reads x[2 (j0 + jl) + t - 3] = (t odd ? xe : xo)[jl + (t + 2 + (t & 1)) / 2]
// Positions ride the MFMA row axis and output channels the lanes (as in conv.hip), so per-channel BatchNorm sums are
// in-lane.  v_mfma_f32_32x32x2_f32; a wave owns 64 positions x 64 channels (4 accumulator blocks).
//   stem_fwd_kernel<false>: conv output (N, 64, Lout) + per-workgroup BN partial sums (train mode)
//   stem_fwd_kernel<true> : eval mode - folded BN scale/shift + ReLU + MaxPool1d(3, 2, 1) fused: only the pooled
//                           activation (N, 64, Lp) is written; the 262 MB conv output never exists
//   stem_wgrad_kernel     : dW[m][c][t] = sum_{n,j} dc[n][m][j] x[n][c][2j + t - 3]; positions are the MFMA depth,
//                           D[(c,t)][m] accumulates over the workgroup's tiles, fixed-order slab reduction (reproducible)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ssecg.h"
#include "conv_common.h"

namespace {

typedef unsigned u32x4s __attribute__((__vector_size__(16)));

constexpr int kSM = 64;           // output channels of the stem
constexpr int kSMaxC = 16;        // input leads supported
constexpr int kSKmax = 7 * kSMaxC;
constexpr int kSTile = 256;       // conv positions per workgroup tile (forward)
constexpr int kSXP = 264;         // floats per de-interleaved input row (>= kSTile/1 + 5, forward)
constexpr int kSTP = 65;          // pitch of the per-wave transpose tile

struct StemP {
    const float* x;      // (N, C, L)
    const float* w;      // (64, C, 7)
    float* out;          // train: (N, 64, Lout); eval: pooled (N, 64, Lp)
    float* stats;        // [gridDim.x][64][2] or nullptr
    const float* scale;  // eval: folded BN
    const float* shift;
    int N, C, L, Lout, Lp, K, KP, tps, numTiles;
    unsigned x_bytes;
    // Two-source input (round 4): samples [0, N1) come from x, [N1, N) from x2 - the student batch of the semi-supervised plugins is
    // (labelled, strongly augmented unlabelled), which the reference concatenates (src/algorithms/fixmatch.py:98-100) before its stem
    // reads it once; reading the two tensors where they lie saves that copy.  x2 == nullptr: N1 = N.
    const float* x2;
    int N1;
    unsigned x2_bytes;
    // use_amp (round 5): the reference under autocast runs the stem convolution on 16-bit operands and stores its output in 16 bit
    // (src/algorithms/fixmatch.py:97; op table profiles/r05_cpu_autocast_op_table.txt).  lp != 0: input samples and weights are
    // rounded to bf16 while they are staged into LDS and the convolution output is rounded to bf16 before the BatchNorm sums
    // and the store (values stay in fp32 containers: products of two bf16 numbers are exact in the fp32 MFMA, accumulation fp32).
    int lp;
    int out16;  // lp == 2: the (bf16-valued) output is STORED as bf16, planar (N, 64, Lout) 16-bit values; Lout % 8 == 0, aligned base
    int vec4;   // 16-byte output stores possible (Lout % 4 == 0, aligned base)
    int xvec;   // 16-byte input loads possible (L % 4 == 0, aligned base): the staged window starts on a multiple of 4 samples
};

// round to nearest-even bf16, returned in an fp32 container (v_cvt_pk_bf16_f32)
__device__ __forceinline__ float rbf16(float v) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ t = {v, 0.f};
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_)) << 16);
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {   // v_cvt_pk_bf16_f32
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_));
}

// LDS float offset of the (c, t) row for output position jl = 0 (see the header)
__device__ __forceinline__ int stem_ct_off(int c, int t, int xp) {
    return (2 * c + 1 - (t & 1)) * xp + ((t + 2 + (t & 1)) >> 1);
}
__device__ __forceinline__ int stem_row_off(int k, int K, int xp) {   // depth order k = 7 c + t (weight gradient rows)
    const int kk = k < K ? k : 0;   // rows beyond K are never stored; the operand only has to be in range
    const int c = kk / 7;
    return stem_ct_off(c, kk - 7 * c, xp);
}

// Forward depth order: leads in PAIRS, k' = 14 cp + kl with (c, t) = (2 cp + (kl >= 7), kl mod 7); an MFMA k-step j of a
// pair holds kl = 2 j + lane-half, so every LDS offset inside the K loop is a per-lane constant computed once (only j = 3
// mixes the two leads).  A missing odd lead is a block of zero weights.
constexpr int kSXR = (kSMaxC * 2 * (kSTile + 5) + 255) / 256;   // staged input samples per thread (<= 33)

template <bool EVAL, bool XV, bool OUT16 = false>   // XV: 16-byte input staging (p.xvec); OUT16: bf16 output (p.out16; compile time:
// as a run-time branch it cost the fp32 launches 10 registers and 5 us)
__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(StemP p) {
    __shared__ float Ws[kSKmax * kSM];                 // [k'][m]
    __shared__ float xs[2 * kSMaxC * kSXP];            // [c][even|odd][kSXP]; reused as 4 per-wave [32][65] transpose tiles
    __shared__ float sHalo[kSM];
    static_assert(4 * 32 * kSTP <= 2 * kSMaxC * kSXP, "transpose tiles alias the input rows");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int npairs = (p.C + 1) >> 1;

    for (int e = tid; e < npairs * 14 * kSM; e += 256) {
        const int k = e >> 6, m = e & 63;
        const int cp = k / 14, kl = k - 14 * cp;
        const int c = 2 * cp + (kl >= 7), t = kl >= 7 ? kl - 7 : kl;
        const float wv = c < p.C ? p.w[(m * p.C + c) * 7 + t] : 0.f;
        Ws[e] = p.lp ? rbf16(wv) : wv;
    }
    int offj[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int kl = 2 * j + lhi;
        offj[j] = stem_ct_off(kl >= 7, kl >= 7 ? kl - 7 : kl, kSXP);
    }
    float st_sum[2] = {0.f, 0.f}, st_sq[2] = {0.f, 0.f};

    // The next tile's input samples are requested right after the current tile's rows are in LDS and travel during its
    // MFMAs and epilogue (a serial load -> LDS-write loop made the first version latency-bound: 280 us vs 214 generic).
    constexpr int per = 2 * (kSTile + 5);   // 522 input samples per lead and tile
    const int total = 2 * npairs * per;   // the missing second lead of an odd count is staged as zeros (its weights are zero,
    float rx[XV ? 1 : kSXR];               // but 0 x stale LDS garbage must not be NaN)
    const auto xR1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const auto xR2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2 != nullptr ? p.x2 : p.x), 0,
                                                        (int)(p.x2 != nullptr ? p.x2_bytes : p.x_bytes), 0x00020000);
    // Two staging forms.  Rows whose length is a multiple of 4 (and an aligned base - the shipped L = 2000): the window of a lead
    // starts at sample 2 j0 - 8 (a multiple of 4: j0 is a multiple of 256), so a thread issues 16-byte buffer loads - 131 per lead
    // instead of 522 dword loads, 7 vector-memory instructions per thread and tile instead of 25 for 12 leads - and a vector lies
    // inside the row or outside it as a whole (padding = out-of-range offset -> zeros).  Otherwise: one dword per sample.
    constexpr int kVecPerLead = (per + 2 + 3) / 4;                         // 131: samples 2 j0 - 8 ... 2 j0 - 8 + 523
    constexpr int kSXV = (2 * kSMaxC / 2 * kVecPerLead + 255) / 256;       // vectors per thread (<= 9 for 16 leads)
    const int totalv = 2 * npairs * kVecPerLead;
    u32x4s rv[XV ? kSXV : 1];
    auto load_x = [&](int tile) {
        const int n = tile / p.tps, j0 = (tile - n * p.tps) * kSTile;
        const int g0 = 2 * (j0 - 3);
        const bool second = n >= p.N1;                                      // (uniform) which source tensor holds this sample
        const auto xR = second ? xR2 : xR1;
        const unsigned row0 = (unsigned)(second ? n - p.N1 : n) * (unsigned)p.C * (unsigned)p.L;
        if (XV) {
#pragma unroll
            for (int u = 0; u < kSXV; ++u) {
                const int e = tid + 256 * u;
                const int c = e / kVecPerLead, v = e - c * kVecPerLead;
                const int g = g0 - 2 + 4 * v;
                const bool ok = e < totalv && c < p.C && (unsigned)g < (unsigned)p.L;   // whole vector inside the row, or zeros
                rv[u] = __builtin_amdgcn_raw_buffer_load_b128(xR, oob_if((row0 + (unsigned)(c * p.L + g)) * 4u, !ok), 0, 0);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < kSXR; ++u) {
            const int e = tid + 256 * u;
            const int c = e / per, i = e - c * per;
            const int g = g0 + i;
            const bool ok = e < total && c < p.C && (unsigned)g < (unsigned)p.L;   // else: padding / missing lead -> 0
#if defined(SSECG_ABLS_NOLOAD)   // timing experiment
            rx[u] = ok ? 0.25f : 0.f;
#else
            rx[u] = buf_load_f32(xR, oob_if((row0 + (unsigned)(c * p.L + g)) * 4u, !ok));
#endif
        }
    };
    auto store_x = [&]() {
        if (XV) {
#pragma unroll
            for (int u = 0; u < kSXV; ++u) {
                const int e = tid + 256 * u;
                const int c = e / kVecPerLead, v = e - c * kVecPerLead;
                if (e < totalv) {
                    // vector v holds window samples i = 4v - 2 ... 4v + 1: (even row, odd row) x (index 2v - 1, 2v)
                    const unsigned a0 = rv[u][0], a1 = rv[u][1], a2 = rv[u][2], a3 = rv[u][3];
                    float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
                    if (p.lp) { f0 = rbf16(f0); f1 = rbf16(f1); f2 = rbf16(f2); f3 = rbf16(f3); }
                    float* xe = xs + (2 * c) * kSXP + 2 * v;
                    float* xo = xe + kSXP;
                    if (v > 0) { xe[-1] = f0; xo[-1] = f1; }
                    xe[0] = f2; xo[0] = f3;
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < kSXR; ++u) {
            const int e = tid + 256 * u;
            const int c = e / per, i = e - c * per;
            if (e < total) xs[(2 * c + (i & 1)) * kSXP + (i >> 1)] = p.lp ? rbf16(rx[u]) : rx[u];
        }
    };
    if (blockIdx.x < p.numTiles) load_x(blockIdx.x);

    // folded-BatchNorm coefficients of this lane's channels: read ONCE.  A load issued inside the tile loop sits behind the previous
    // tile's output stores in the in-order vmcnt queue - waiting for it means waiting for their write acknowledgements
    // (profiles/r04_residual_epilogue.txt)
    float esc[2] = {1.f, 1.f}, esh[2] = {0.f, 0.f}, hsc = 1.f, hsh = 0.f;
    if (EVAL) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) { esc[cb] = p.scale[32 * cb + l31]; esh[cb] = p.shift[32 * cb + l31]; }
        hsc = p.scale[lane]; hsh = p.shift[lane];
    }
    for (int tile = blockIdx.x; tile < p.numTiles; tile += gridDim.x) {
        const int n = tile / p.tps, j0 = (tile - n * p.tps) * kSTile;
        __syncthreads();   // previous tile's transpose readers are done; Ws is written (first tile)
        store_x();
        __syncthreads();
        if (tile + gridDim.x < p.numTiles) load_x(tile + gridDim.x);   // travels during the MFMAs and the epilogue

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        const float* xw = xs + wave * 64 + l31;
        const float* ww = Ws + lhi * kSM + l31;
        for (int cp = 0; cp < npairs; ++cp) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float x0 = xw[offj[j]], x1 = xw[offj[j] + 32];
                const float w0 = ww[2 * j * kSM], w1 = ww[2 * j * kSM + 32];
#if defined(SSECG_ABLS_NOMFMA)   // timing experiment
                asm volatile("" :: "v"(x0), "v"(x1), "v"(w0), "v"(w1));
                continue;
#endif
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, w0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, w0, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, w1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, w1, acc[1][1], 0, 0, 0);
            }
            xw += 4 * kSXP;
            ww += 14 * kSM;
        }
        if (EVAL && wave == 0) {
            // conv position j0 - 1 (the pooling window of the tile's first output reaches one position back): 64 dot products
            float h = 0.f;
            if (j0 > 0) {
                for (int c = 0; c < p.C; ++c)
#pragma unroll
                    for (int t = 0; t < 7; ++t)
                        h = fmaf(Ws[((c >> 1) * 14 + (c & 1) * 7 + t) * kSM + lane], xs[stem_ct_off(c, t, kSXP) - 1], h);
                h = fmaf(h, hsc, hsh);
                h = h < 0.f ? 0.f : h;          // ReLU that keeps a NaN (fmaxf would return 0), as torch and the generic path do
            }
            sHalo[lane] = h;
        }
        __syncthreads();   // every wave is done reading xs: it becomes the transpose tiles

        float* T = xs + wave * (32 * kSTP);
        const int pw = j0 + wave * 64;   // first conv position of this wave
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float sc = esc[cb], sh = esh[cb];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pl = 32 * pb + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const bool ok = pw + pl < p.Lout;
                    float v = acc[pb][cb][r];
                    if (EVAL) { v = fmaf(v, sc, sh); v = v < 0.f ? 0.f : v; }   // NaN-keeping ReLU
                    else if (p.lp) v = rbf16(v);                                // use_amp: the stored (and summed) output is 16 bit
                    v = ok ? v : 0.f;
                    if (!EVAL) { s += v; q = fmaf(v, v, q); }
                    T[l31 * kSTP + pl] = v;
                }
            if (!EVAL) { st_sum[cb] += s; st_sq[cb] += q; }
            if (EVAL) {
                __syncthreads();   // the wave to the left holds this wave's halo column
                const int chh = lane >> 5, ql = lane & 31;
                const int qg = (pw >> 1) + ql;
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const int ch = 2 * i + chh;
                    float a;
                    if (ql > 0) a = T[ch * kSTP + 2 * ql - 1];
                    else a = wave > 0 ? (T - 32 * kSTP)[ch * kSTP + 63] : sHalo[32 * cb + ch];
                    const float b = T[ch * kSTP + 2 * ql], c = T[ch * kSTP + 2 * ql + 1];
                    float m = a;                       // first maximum wins, a NaN in the window propagates (nn.MaxPool1d)
                    if (b > m || b != b) m = b;
                    if (c > m || c != c) m = c;
                    if (qg < p.Lp) p.out[((size_t)n * kSM + 32 * cb + ch) * p.Lp + qg] = m;
                }
                __syncthreads();   // before the next channel block overwrites the tiles
            } else {
                asm volatile("" ::: "memory");
                // 64 dword stores per wave and tile were the largest part of the non-MFMA time (ablation: -46 us of 187 without
                // them - store ISSUE, not bandwidth): 16-byte stores, four channel rows x 16 quads per instruction
                if (OUT16) {   // 16 bytes = 8 positions per lane: eight channel rows x 8 octets per instruction, HALF the stores
                    const int rq = lane >> 3, oc = lane & 7;
                    const bool ok = pw + 8 * oc < p.Lout;   // Lout % 8 == 0: an octet is inside or outside as a whole
                    uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + ((size_t)n * kSM + 32 * cb + rq) * p.Lout + pw + 8 * oc;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float* t = T + (8 * i + rq) * kSTP + 8 * oc;
                        u32x4s v;
                        v[0] = pack_bf16x2(t[0], t[1]); v[1] = pack_bf16x2(t[2], t[3]);
                        v[2] = pack_bf16x2(t[4], t[5]); v[3] = pack_bf16x2(t[6], t[7]);
                        if (ok) *reinterpret_cast<u32x4s*>(o + (size_t)(8 * i) * p.Lout) = v;
                    }
                } else if (p.vec4) {
                    const int rq = lane >> 4, qd = lane & 15;
                    const bool ok = pw + 4 * qd < p.Lout;   // Lout % 4 == 0: a quad is inside or outside as a whole
                    float* o = p.out + ((size_t)n * kSM + 32 * cb + rq) * p.Lout + pw + 4 * qd;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float* t = T + (4 * i + rq) * kSTP + 4 * qd;
                        const float4 v = make_float4(t[0], t[1], t[2], t[3]);
#if defined(SSECG_ABLS_NOSTORE)
                        asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
                        if (ok) *reinterpret_cast<float4*>(o + (size_t)(4 * i) * p.Lout) = v;
#endif
                    }
                } else {
                    const bool ok = pw + lane < p.Lout;
                    float* o = p.out + ((size_t)n * kSM + 32 * cb) * p.Lout + pw + lane;
#pragma unroll 8
                    for (int ch = 0; ch < 32; ++ch) {
                        const float v = T[ch * kSTP + lane];
#if defined(SSECG_ABLS_NOSTORE)   // timing experiment
                        asm volatile("" :: "v"(v));
#else
                        if (ok) o[(size_t)ch * p.Lout] = v;
#endif
                    }
                }
                asm volatile("" ::: "memory");
            }
        }
    }

    if (!EVAL && p.stats != nullptr) {
        __syncthreads();
        float* red = xs;   // [wave][64][2]
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float s = st_sum[cb] + __shfl_xor(st_sum[cb], 32, 64);
            const float q = st_sq[cb] + __shfl_xor(st_sq[cb], 32, 64);
            if (lhi == 0) {
                red[(wave * kSM + 32 * cb + l31) * 2] = s;
                red[(wave * kSM + 32 * cb + l31) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < kSM) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[(w * kSM + tid) * 2]; q += red[(w * kSM + tid) * 2 + 1]; }
            p.stats[((size_t)blockIdx.x * kSM + tid) * 2] = s;
            p.stats[((size_t)blockIdx.x * kSM + tid) * 2 + 1] = q;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWTile = 128;        // positions per stage
constexpr int kWXP = 136;          // de-interleaved input row (>= kWTile + 5)
constexpr int kWDP = kWTile + 1;   // pitch of a dc row (odd: lanes = channels read conflict-free)

struct StemWgP {
    const float* dc;   // (N, 64, Lout)
    const float* x;    // (N, C, L)
    float* ws;         // [gridDim.x][KR][64], KR = 32 * ceil(K / 32)
    int N, C, L, Lout, K, KR, tps, numTiles;
    unsigned x_bytes, dc_bytes;
    const float* x2;   // two-source input as StemP: samples [N1, N) come from x2 (nullptr: N1 = N)
    int N1;
    unsigned x2_bytes;
    int lp;            // use_amp: both operands rounded to bf16 while staged (autocast's 16-bit conv backward reads 16-bit x and dc)
    int dc16;          // lp == 2: dc is STORED as bf16 (planar; written by ssecg_bn_relu_maxpool_bwd_apply with lp == 2); XV launches only
};

constexpr int kWXR = (kSMaxC * 2 * (kWTile + 5) + 255) / 256;   // staged input samples per thread (<= 17)

template <int NRB, bool XV, bool DC16 = false>   // row blocks of (c, t): ceil(7 C / 32) = 1..4; XV: 16-byte staging loads (rows of x and dc are
// 16-byte aligned); DC16: dc is stored as bf16 (compile time: as a run-time branch the staged vectors went to scratch, 120 -> 158 us)
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(StemWgP p) {
    __shared__ float xs[2 * kSMaxC * kWXP];   // 17 KB
    __shared__ float ds[kSM * kWDP];          // 33 KB; reused for the cross-wave reduction
    static_assert(4 * 32 * kSM <= kSM * kWDP, "reduction buffer aliases the dc tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    constexpr int nrb = NRB;

    int offA[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) offA[rb] = stem_row_off(32 * rb + l31, p.K, kWXP);

    f32x16 acc[NRB][2];
#pragma unroll
    for (int a = 0; a < NRB; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Next tile's operands travel in registers during this tile's MFMAs.  Every load is a raw buffer load whose offset
    // carries the range verdict in bit 31 (-> 0): predicated plain loads compiled to a branch + vmcnt(0) per load and the
    // first version spent 66 % of its wave time waiting.
    constexpr int per = 2 * (kWTile + 5);
    const int total = p.C * per;
    const auto xR1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const auto xR2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2 != nullptr ? p.x2 : p.x), 0,
                                                        (int)(p.x2 != nullptr ? p.x2_bytes : p.x_bytes), 0x00020000);
    const auto dR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dc), 0, (int)p.dc_bytes, 0x00020000);
    // XV (rows of x and dc start on 16-byte boundaries: L % 4 == 0, Lout % 4 == 0 - the shipped L = 2000): 16-byte loads - the
    // input window of a lead starts at sample 2 j0 - 8 (a multiple of 4), 67 vectors per lead instead of 266 dwords; the dc tile
    // is 64 rows x 32 quads: 13 vector-memory instructions per thread and tile instead of 49 (as stem_fwd_kernel)
    constexpr int kVecPerLead = (per + 2 + 3) / 4;                      // 67
    constexpr int kWXV = (kSMaxC * kVecPerLead + 255) / 256;            // <= 5
    const int totalv = p.C * kVecPerLead;
    float rx[XV ? 1 : kWXR];
    float rd[XV ? 1 : 32];
    u32x4s rxv[XV ? kWXV : 1], rdv[XV ? 8 : 1];
    auto load_tile = [&](int tile) {
        const int n = tile / p.tps, j0 = (tile - n * p.tps) * kWTile;
        const int g0 = 2 * (j0 - 3);
        const bool second = n >= p.N1;                                      // (uniform) which source tensor holds this sample
        const auto xR = second ? xR2 : xR1;
        const unsigned row0 = (unsigned)(second ? n - p.N1 : n) * (unsigned)p.C * (unsigned)p.L;
        if (XV) {
#pragma unroll
            for (int u = 0; u < kWXV; ++u) {
                const int e = tid + 256 * u;
                const int c = e / kVecPerLead, v = e - c * kVecPerLead;
                const int g = g0 - 2 + 4 * v;
                const bool ok = e < totalv && (unsigned)g < (unsigned)p.L;
                rxv[u] = __builtin_amdgcn_raw_buffer_load_b128(xR, oob_if((row0 + (unsigned)(c * p.L + g)) * 4u, !ok), 0, 0);
            }
            if (DC16) {   // 16 bytes = 8 positions: row m = e / 16, octet e % 16 - four loads per thread instead of eight
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = tid + 256 * u;
                    const int m = e >> 4, jo = j0 + 8 * (e & 15);
                    const unsigned off = (((unsigned)n * kSM + (unsigned)m) * (unsigned)p.Lout + (unsigned)jo) * 2u;
                    rdv[u] = __builtin_amdgcn_raw_buffer_load_b128(dR, oob_if(off, !(jo < p.Lout)), 0, 0);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * u;          // row m = e / 32, quad e % 32
                const int m = e >> 5, jq = j0 + 4 * (e & 31);
                const unsigned off = (((unsigned)n * kSM + (unsigned)m) * (unsigned)p.Lout + (unsigned)jq) * 4u;
                rdv[u] = __builtin_amdgcn_raw_buffer_load_b128(dR, oob_if(off, !(jq < p.Lout)), 0, 0);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < kWXR; ++u) {
            const int e = tid + 256 * u;
            const int c = e / per, i = e - c * per;
            const int g = g0 + i;
            const bool ok = e < total && (unsigned)g < (unsigned)p.L;
            rx[u] = buf_load_f32(xR, oob_if((row0 + (unsigned)(c * p.L + g)) * 4u, !ok));
        }
        // dc tile: thread -> (row m = 2 u + tid / 128, position tid % 128): 256 contiguous bytes per wave instruction,
        // conflict-free LDS rows
        const int j = tid & 127;
        const unsigned d0 = ((unsigned)n * kSM + (unsigned)(tid >> 7)) * (unsigned)p.Lout + (unsigned)(j0 + j);
        const unsigned doff = oob_if(d0 * 4u, !(j0 + j < p.Lout));
        const unsigned rstep = 2u * (unsigned)p.Lout * 4u;
#pragma unroll
        for (int u = 0; u < 32; ++u)
            rd[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dR, doff, u * rstep, 0));
    };
    auto store_tile = [&]() {
        if (XV) {
#pragma unroll
            for (int u = 0; u < kWXV; ++u) {
                const int e = tid + 256 * u;
                const int c = e / kVecPerLead, v = e - c * kVecPerLead;
                if (e < totalv) {   // vector v = window samples i = 4v - 2 ... 4v + 1: (even row, odd row) x (index 2v - 1, 2v)
                    const unsigned a0 = rxv[u][0], a1 = rxv[u][1], a2 = rxv[u][2], a3 = rxv[u][3];
                    float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
                    if (p.lp) { f0 = rbf16(f0); f1 = rbf16(f1); f2 = rbf16(f2); f3 = rbf16(f3); }
                    float* xe = xs + (2 * c) * kWXP + 2 * v;
                    float* xo = xe + kWXP;
                    if (v > 0) { xe[-1] = f0; xo[-1] = f1; }
                    xe[0] = f2; xo[0] = f3;
                }
            }
            if (DC16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = tid + 256 * u;
                    float* d = ds + (e >> 4) * kWDP + 8 * (e & 15);
                    const unsigned a0 = rdv[u][0], a1 = rdv[u][1], a2 = rdv[u][2], a3 = rdv[u][3];
                    d[0] = __uint_as_float(a0 << 16); d[1] = __uint_as_float(a0 & 0xffff0000u);
                    d[2] = __uint_as_float(a1 << 16); d[3] = __uint_as_float(a1 & 0xffff0000u);
                    d[4] = __uint_as_float(a2 << 16); d[5] = __uint_as_float(a2 & 0xffff0000u);
                    d[6] = __uint_as_float(a3 << 16); d[7] = __uint_as_float(a3 & 0xffff0000u);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + 256 * u;
                float* d = ds + (e >> 5) * kWDP + 4 * (e & 31);
                const unsigned a0 = rdv[u][0], a1 = rdv[u][1], a2 = rdv[u][2], a3 = rdv[u][3];
                float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
                if (p.lp) { f0 = rbf16(f0); f1 = rbf16(f1); f2 = rbf16(f2); f3 = rbf16(f3); }
                d[0] = f0; d[1] = f1; d[2] = f2; d[3] = f3;
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < kWXR; ++u) {
            const int e = tid + 256 * u;
            const int c = e / per, i = e - c * per;
            if (e < total) xs[(2 * c + (i & 1)) * kWXP + (i >> 1)] = p.lp ? rbf16(rx[u]) : rx[u];
        }
        float* d = ds + (tid >> 7) * kWDP + (tid & 127);
#pragma unroll
        for (int u = 0; u < 32; ++u) d[2 * u * kWDP] = p.lp ? rbf16(rd[u]) : rd[u];
    };
    if (blockIdx.x < p.numTiles) load_tile(blockIdx.x);

    for (int tile = blockIdx.x; tile < p.numTiles; tile += gridDim.x) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + gridDim.x < p.numTiles) load_tile(tile + gridDim.x);
        const float* xw = xs + wave * 32 + lhi;
        const float* dw = ds + l31 * kWDP + wave * 32 + lhi;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const float d0 = dw[2 * ks], d1 = dw[2 * ks + 32 * kWDP];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const float xv = xw[offA[rb] + 2 * ks];
                acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, d0, acc[rb][0], 0, 0, 0);
                acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, d1, acc[rb][1], 0, 0, 0);
            }
        }
    }

    // cross-wave sum (fixed order) of D[(c,t) row][m] and the slab store
    float* red = ds;   // [wave][32 rows][64]
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                red[(wave * 32 + row) * kSM + 32 * mb + l31] = acc[rb][mb][r];
            }
        __syncthreads();
        for (int e = tid; e < 32 * kSM; e += 256) {
            const float v = (red[e] + red[32 * kSM + e]) + (red[2 * 32 * kSM + e] + red[3 * 32 * kSM + e]);
            p.ws[((size_t)blockIdx.x * p.KR + 32 * rb) * kSM + e] = v;
        }
    }
}

// dw[m][k] = sum over slabs of ws[z][k][m]; 16 slab lanes per element, combined in a fixed order
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Z, int K, int KR) {
    __shared__ float part[16][17];
    const int el = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;   // e = k * 64 + m
    float s0 = 0.f, s1 = 0.f;
    if (e < K * kSM) {
        int z = zl;
        for (; z + 16 < Z; z += 32) {
            s0 += ws[(size_t)z * KR * kSM + e];
            s1 += ws[(size_t)(z + 16) * KR * kSM + e];
        }
        if (z < Z) s0 += ws[(size_t)z * KR * kSM + e];
    }
    part[zl][el] = s0 + s1;
    __syncthreads();
    if (zl == 0 && e < K * kSM) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][el];
        const int k = e >> 6, m = e & 63;
        dw[(size_t)m * K + k] = t;
    }
}

inline bool stem_ok(int N, int C, int L) {
    if (N <= 0 || C <= 0 || C > kSMaxC || L < 1) return false;
    const long long Lout = (L - 1) / 2 + 1;
    return (size_t)N * C * L * 4 < 0x7fffff00ull && (size_t)N * kSM * Lout * 4 < 0x7fffff00ull && (long long)N * ((Lout + kWTile - 1) / kWTile) < 0x7fffffffLL;   // buffer loads
}

inline int stem_fwd_grid(int N, int L) {
    const int Lout = (L - 1) / 2 + 1;
    const long long tiles = (long long)N * ((Lout + kSTile - 1) / kSTile);
    return (int)(tiles < 2 * kNumCU ? tiles : 2 * kNumCU);
}

inline int stem_wg_grid(int N, int L) {
    const int Lout = (L - 1) / 2 + 1;
    const long long tiles = (long long)N * ((Lout + kWTile - 1) / kWTile);
    return (int)(tiles < 2 * kNumCU ? tiles : 2 * kNumCU);
}

}  // namespace

extern "C" {

int ssecg_stem_supported(int N, int C, int L) { return stem_ok(N, C, L) ? 1 : 0; }

int ssecg_stem_c16_supported(int N, int C, int L) {
    return (stem_ok(N, C, L) && L % 4 == 0 && ((L - 1) / 2 + 1) % 8 == 0) ? 1 : 0;
}

int ssecg_stem_parts(int N, int L) {
    if (N <= 0 || L < 1) return SSECG_E_INVAL;
    return stem_fwd_grid(N, L);
}

static int stem_launch(const float* x, const float* w, float* out, int N, int C, int L, float* stats, const float* scale,
                       const float* shift, bool eval, void* stream, const float* x2 = nullptr, int n1 = 0, int lp = 0) {
    StemP p;
    p.x = x; p.w = w; p.out = out; p.stats = stats; p.scale = scale; p.shift = shift; p.lp = eval ? 0 : lp;
    p.out16 = (!eval && lp == 2) ? 1 : 0;
    p.N = N; p.C = C; p.L = L;
    p.x2 = x2; p.N1 = x2 != nullptr ? n1 : N;
    p.x_bytes = (unsigned)((size_t)p.N1 * C * L * 4);
    p.x2_bytes = (unsigned)((size_t)(N - p.N1) * C * L * 4);
    p.Lout = (L - 1) / 2 + 1;
    p.Lp = (p.Lout - 1) / 2 + 1;
    p.K = 7 * C; p.KP = (p.K + 1) & ~1;
    p.tps = (p.Lout + kSTile - 1) / kSTile;
    p.numTiles = N * p.tps;
    const int grid = stem_fwd_grid(N, L);
    p.vec4 = (p.Lout % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    p.xvec = (L % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(x2) & 15) == 0);
    if (eval) {
        if (p.xvec) hipLaunchKernelGGL((stem_fwd_kernel<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((stem_fwd_kernel<true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    } else if (p.out16) {
        if (p.xvec) hipLaunchKernelGGL((stem_fwd_kernel<false, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((stem_fwd_kernel<false, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (p.xvec) hipLaunchKernelGGL((stem_fwd_kernel<false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((stem_fwd_kernel<false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}

int ssecg_stem_fwd2(const float* x, const float* x2, int n1, const float* w, void* c, int N, int C, int L, float* stats_partial,
                    int stats_parts, int lp, void* stream) {
    if (!x || !w || !c || !stem_ok(N, C, L) || lp < 0 || lp > 2) return SSECG_E_INVAL;
    if (x2 != nullptr && (n1 <= 0 || n1 >= N)) return SSECG_E_INVAL;
    if (lp == 2 && ((((L - 1) / 2 + 1) % 8) != 0 || (reinterpret_cast<uintptr_t>(c) & 15) != 0)) return SSECG_E_INVAL;
    if (stats_partial != nullptr) {
        const int g = stem_fwd_grid(N, L);
        if (stats_parts < g) return SSECG_E_WORKSPACE;
        if (stats_parts > g) {
            const hipError_t e = hipMemsetAsync(stats_partial + (size_t)g * kSM * 2, 0, (size_t)(stats_parts - g) * kSM * 2 * sizeof(float),
                                                (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    return stem_launch(x, w, reinterpret_cast<float*>(c), N, C, L, stats_partial, nullptr, nullptr, false, stream, x2, n1, lp);
}

int ssecg_stem_fwd(const float* x, const float* w, float* c, int N, int C, int L, float* stats_partial, int stats_parts,
                   void* stream) {
    return ssecg_stem_fwd2(x, nullptr, 0, w, c, N, C, L, stats_partial, stats_parts, 0, stream);
}

int ssecg_stem_fwd_eval_pool(const float* x, const float* w, const float* scale, const float* shift, float* pooled, int N, int C,
                             int L, void* stream) {
    if (!x || !w || !scale || !shift || !pooled || !stem_ok(N, C, L)) return SSECG_E_INVAL;
    return stem_launch(x, w, pooled, N, C, L, nullptr, scale, shift, true, stream);
}

size_t ssecg_stem_wgrad_workspace(int N, int C, int L) {
    if (!stem_ok(N, C, L)) return 0;
    const int KR = 32 * ((7 * C + 31) / 32);
    return (size_t)stem_wg_grid(N, L) * KR * kSM * sizeof(float);
}

int ssecg_stem_wgrad2(const void* dc, const float* x, const float* x2, int n1, float* dw, int N, int C, int L, void* workspace,
                      size_t workspace_bytes, int lp, void* stream);

int ssecg_stem_wgrad(const float* dc, const float* x, float* dw, int N, int C, int L, void* workspace, size_t workspace_bytes,
                     void* stream) {
    return ssecg_stem_wgrad2(dc, x, nullptr, 0, dw, N, C, L, workspace, workspace_bytes, 0, stream);
}

int ssecg_stem_wgrad2(const void* dc, const float* x, const float* x2, int n1, float* dw, int N, int C, int L, void* workspace,
                      size_t workspace_bytes, int lp, void* stream) {
    if (!dc || !x || !dw || !workspace || !stem_ok(N, C, L) || lp < 0 || lp > 2) return SSECG_E_INVAL;
    if (x2 != nullptr && (n1 <= 0 || n1 >= N)) return SSECG_E_INVAL;
    if (workspace_bytes < ssecg_stem_wgrad_workspace(N, C, L)) return SSECG_E_WORKSPACE;
    StemWgP p;
    p.dc = reinterpret_cast<const float*>(dc); p.x = x; p.ws = (float*)workspace; p.lp = lp; p.dc16 = lp == 2 ? 1 : 0;
    p.N = N; p.C = C; p.L = L;
    p.x2 = x2; p.N1 = x2 != nullptr ? n1 : N;
    p.x_bytes = (unsigned)((size_t)p.N1 * C * L * 4);
    p.x2_bytes = (unsigned)((size_t)(N - p.N1) * C * L * 4);
    p.Lout = (L - 1) / 2 + 1;
    p.K = 7 * C; p.KR = 32 * ((p.K + 31) / 32);
    p.tps = (p.Lout + kWTile - 1) / kWTile;
    p.numTiles = N * p.tps;
    const int grid = stem_wg_grid(N, L);
    hipStream_t st = (hipStream_t)stream;
    p.dc_bytes = (unsigned)((size_t)N * kSM * p.Lout * (p.dc16 ? 2 : 4));
    if (p.dc16 && !((L % 4 == 0) && (p.Lout % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(p.x2) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.dc) & 15) == 0)))
        return SSECG_E_INVAL;   // bf16-stored dc: the 16-byte-load launches only
    const bool xv = (L % 4 == 0) && (p.Lout % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(p.x2) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.dc) & 15) == 0);
#define SSECG_STEM_WG(R_)                                                                                            \
    if (p.dc16) hipLaunchKernelGGL((stem_wgrad_kernel<R_, true, true>), dim3(grid), dim3(256), 0, st, p);            \
    else if (xv) hipLaunchKernelGGL((stem_wgrad_kernel<R_, true>), dim3(grid), dim3(256), 0, st, p);                 \
    else hipLaunchKernelGGL((stem_wgrad_kernel<R_, false>), dim3(grid), dim3(256), 0, st, p)
    switch (p.KR / 32) {
        case 1: SSECG_STEM_WG(1); break;
        case 2: SSECG_STEM_WG(2); break;
        case 3: SSECG_STEM_WG(3); break;
        default: SSECG_STEM_WG(4); break;
    }
#undef SSECG_STEM_WG
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((p.K * kSM + 15) / 16), dim3(256), 0, st, (const float*)workspace, dw, grid, p.K, p.KR);
    return (int)hipGetLastError();
}

}  // extern "C"
