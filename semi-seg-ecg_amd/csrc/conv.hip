// Conv1d forward / data-gradient / weight-gradient as implicit GEMMs on the gfx950
// fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, 64 FLOP/clk/SIMD).
//
// Layout: PyTorch-contiguous (N, C, L), L fastest.  The GEMM "column" axis is the
// flattened position axis p = n*L + l (so short rows such as L = 63 waste nothing);
// the im2col matrix is never materialised - each workgroup gathers its K-slab of
// the input straight into LDS with zero padding applied in the gather.
//
//   fwd  : out[n,m,l]  = sum_{c,t} A[m][c*KS+t] * src[n,c,l*stride + t*dil - pad]      A = w   [Cout][Cin*KS]
//   dgrad: out[n,m,l]  = sum_{c,t} A[m][c*KS+t] * src[n,c,(l + pad - t*dil)/stride]    A = w^T [Cin][Cout*KS]
//   wgrad: dw[co,ci,t] = sum_{n,l} dy[n,co,l] * x[n,ci,l*stride + t*dil - pad]         split over p, slab-reduced
//
// Tiling: 256 threads = 4 wavefronts (64 lanes); each wave owns TMxTN tiles of
// 32x32 accumulators (16 VGPRs each); K advances 16 per LDS stage (8 MFMA k-steps),
// double-buffered with the next stage's global loads in flight during the MFMAs.
// Workgroups loop over position tiles (grid.x is sized to the chip, not the problem)
// so the per-channel BatchNorm statistics fused into the forward epilogue leave
// one partial row per (workgroup, wave-column) instead of one per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ssecg.h"
#include "conv_common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 256;
constexpr int kBK = 16;
struct ConvP {
    const float* A;
    const float* src;
    float* out;
    int N, M, Csrc, Lsrc, Ldst, Ktot;
    int stride, pad, dil;
    // fast kernel: source index of (position l, tap t) = l*gmul + tapoff[t]; output element l of a row lives at
    // l*ostride + ooff in a row of Lrow floats (stride-2 data gradients are computed as two phase launches)
    int gmul, tapoff[3], Lrow, ostride, ooff;
    int P, numPT;
    unsigned a_bytes, src_bytes;  // operand sizes for the buffer descriptors (< 2^31, checked by the launcher)
    int a_vec;    // A rows may be read with 16-byte loads
    int out_vec;  // out / residual rows may be accessed 16 bytes at a time (Ldst % 4 == 0, aligned bases)
    const float* scale;
    const float* shift;
    const float* residual;
    int relu;
    float* stats;
    // fast kernel only: the gathered input is relu(src*in_scale[c] + in_shift[c]) - the producer's BatchNorm + ReLU
    // applied on the fly, so that activation is never materialised (padding stays exactly 0)
    const float* in_scale;
    const float* in_shift;
    // fast kernel only - K split of small launches (fewer tiles than workgroup slots): workgroup column blockIdx.z contracts stages
    // [z*nst, (z+1)*nst) into its own partial plane out + z*out_split with the plain epilogue; split_finish_kernel sums the planes
    // and applies the epilogue / emits the BatchNorm sums.  No split: nst = Ktot / 16, out_split = 0, gridDim.z = 1.
    int nst;
    size_t out_split;
};

// MODE 0 = forward gather, 1 = data-gradient gather.
// Workgroups per CU the kernels are register-budgeted for (waves per SIMD, since a workgroup puts one wave on each).
// The launcher sizes grid.x so that all workgroups are co-resident and each loops over its share of the tiles.
#ifndef SSECG_IGEMM_WG_PER_CU
#define SSECG_IGEMM_WG_PER_CU 3
#endif
constexpr int kIgemmWgPerCu = SSECG_IGEMM_WG_PER_CU;

template <int BM, int BN, int WM, int WN, int KS, int MODE, bool AVEC>
__global__ __launch_bounds__(kThreads, (BM == 64 && KS == 1 && kIgemmWgPerCu == 3) ? 2 : kIgemmWgPerCu) void conv_igemm_kernel(ConvP p) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    constexpr int APITCH = kBK + 1;
    constexpr int AE = BM * kBK / kThreads;  // A floats per thread per stage
    constexpr int BE = BN * kBK / kThreads;  // B floats per thread per stage
    constexpr int BROWSTEP = kThreads / BN == 0 ? 1 : kThreads / BN;
    static_assert(AE >= 2 && AE <= 8, "A staging shape");
    static_assert(!AVEC || AE >= 4, "vector A staging needs 4 floats per thread");
    static_assert(BN == 128 || BN == 256, "B staging shape");

    __shared__ float As[2][BM * APITCH];
    __shared__ float Bs[2][kBK * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int m0 = blockIdx.y * BM;
    const int nstages = (p.Ktot + kBK - 1) / kBK;

    // A staging: thread owns AE consecutive k of one row
    const int a_row = (tid * AE) / kBK;
    const int a_col = (tid * AE) % kBK;
    const float* a_ptr = p.A + (size_t)((m0 + a_row) < p.M ? (m0 + a_row) : m0) * p.Ktot + a_col;
    const auto srcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)p.src_bytes, 0x00020000);

    // B staging: thread owns one column j, rows b_r0 + i*BROWSTEP
    const int b_col = tid % BN;
    const int b_r0 = tid / BN;

    float st_sum[TM], st_sq[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { st_sum[i] = 0.f; st_sq[i] = 0.f; }

    for (int pt = blockIdx.x; pt < p.numPT; pt += gridDim.x) {
        const int p0 = pt * BN;
        // this thread's gather column
        const int pc = p0 + b_col;
        const bool col_ok = pc < p.P;
        const int gn = col_ok ? pc / p.Ldst : 0;
        const int gl = pc - gn * p.Ldst;
        const int gbase = (MODE == 0) ? gl * p.stride - p.pad : gl + p.pad;
        const unsigned src_off = (unsigned)gn * (unsigned)(p.Csrc * p.Lsrc);  // element offset of sample gn

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        float ra[AE], rb[BE];

        // Operand loads go through buffer descriptors: a lane whose element is padding / out of range uses an
        // offset beyond num_records and the hardware returns 0 - no branch, no select, nothing that would make the
        // compiler wait on the load before the MFMA block (a conditional load is branched around and drained).
        auto load_stage = [&](int s) {
            const int k0 = s * kBK;
#if defined(SSECG_ABL_NOLOAD)
            for (int q = 0; q < AE; ++q) ra[q] = 1.0f + (float)k0;
            for (int i = 0; i < BE; ++i) rb[i] = 0.5f;
            return;
#endif
            // ---- A (weights) ----
            // plain loads from CLAMPED addresses, no zero-fill needed: a k beyond Ktot meets a B row that is exactly 0
            // (and re-reads a weight of the same row, so non-finite weights poison nothing new), and rows beyond M
            // are never stored.  (__builtin_amdgcn_raw_buffer_load_b128 mis-compiles to a splat dword on ROCm 7.2.)
            if (AVEC) {
#pragma unroll
                for (int q = 0; q < AE / 4; ++q) {
                    const int kk = k0 + a_col + 4 * q;  // Ktot % 4 == 0: the four are in range together
                    // beyond Ktot: re-read the row's first four weights (a_ptr already includes a_col, so the clamp must
                    // go back to the ROW start - a_ptr + 0 can lie past the end of the tensor when a_col >= Ktot)
                    const float4 v = *reinterpret_cast<const float4*>(a_ptr + (kk < p.Ktot ? k0 + 4 * q : -a_col));
                    ra[4 * q + 0] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < AE; ++q) ra[q] = a_ptr[(k0 + a_col + q) < p.Ktot ? k0 + q : -a_col];  // clamp to the row start
            }
            // ---- B (gathered input) ----
#pragma unroll
            for (int i = 0; i < BE; ++i) {
                const int kk = k0 + b_r0 + i * BROWSTEP;
                const int c = kk / KS;
                const int t = kk - c * KS;
                int sidx;
                bool ok = col_ok && (kk < p.Ktot);
                if (MODE == 0) {
                    sidx = gbase + t * p.dil;
                } else {
                    const int num = gbase - t * p.dil;
                    if (p.stride == 2) {
                        ok = ok && ((num & 1) == 0);
                        sidx = num >> 1;
                    } else {
                        sidx = num;
                    }
                }
                ok = ok && ((unsigned)sidx < (unsigned)p.Lsrc);
                rb[i] = buf_load_f32(srcR, oob_if((src_off + (unsigned)(c * p.Lsrc + sidx)) * 4u, !ok));
            }
        };
        auto store_stage = [&](int buf) {
#pragma unroll
            for (int q = 0; q < AE; ++q) As[buf][a_row * APITCH + a_col + q] = ra[q];
#pragma unroll
            for (int i = 0; i < BE; ++i) Bs[buf][(b_r0 + i * BROWSTEP) * BN + b_col] = rb[i];
        };

        load_stage(0);
        __syncthreads();  // previous tile's readers are done with both buffers
        store_stage(0);
        __syncthreads();

        for (int s = 0; s < nstages; ++s) {
            const int buf = s & 1;
            if (s + 1 < nstages) load_stage(s + 1);
            const float* as = &As[buf][(wm * TM * 32 + l31) * APITCH + lhi];
            const float* bs = &Bs[buf][lhi * BN + wn * TN * 32 + l31];
#pragma unroll
            for (int ks = 0; ks < kBK / 2; ++ks) {
                // D[pos][ch] += X[pos][k] * W[k][ch]: positions ride the MFMA row axis (registers),
                // channels the lane axis, so per-channel reductions and parameters are lane-local
                float wv[TM], xv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) wv[i] = as[i * 32 * APITCH + 2 * ks];
#pragma unroll
                for (int j = 0; j < TN; ++j) xv[j] = bs[2 * ks * BN + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#if defined(SSECG_ABL_NOMFMA)
                        { asm volatile("" :: "v"(xv[j]), "v"(wv[i])); }
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[j], wv[i], acc[i][j], 0, 0, 0);
#endif
            }
            if (s + 1 < nstages) store_stage(buf ^ 1);
            __syncthreads();
        }

        // ---------------- epilogue ----------------
        // accumulator layout (32x32 tile): channel = lane & 31, position = (r&3) + 8*(r>>2) + 4*(lane>>5)
        if (p.stats != nullptr) {
            // raw conv output statistics; columns beyond P and rows beyond M hold exact zeros
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];
                        s += v;
                        q = fmaf(v, v, q);
                    }
                st_sum[i] += s;
                st_sq[i] += q;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * TM * 32 + i * 32 + l31;  // output channel of this lane
            const bool rok = row < p.M;
            const float sc = (rok && p.scale != nullptr) ? p.scale[row] : 1.f;
            const float sh = (rok && p.shift != nullptr) ? p.shift[row] : 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int pbase = p0 + wn * TN * 32 + j * 32 + 4 * lhi;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int pp = pbase + 8 * rq;  // 4 consecutive positions pp..pp+3, pp % 4 == 0
                    if (rok && pp < p.P) {
                        int n = pp / p.Ldst;
                        int l = pp - n * p.Ldst;
                        float v0 = acc[i][j][4 * rq + 0] * sc + sh;
                        float v1 = acc[i][j][4 * rq + 1] * sc + sh;
                        float v2 = acc[i][j][4 * rq + 2] * sc + sh;
                        float v3 = acc[i][j][4 * rq + 3] * sc + sh;
                        if (p.out_vec) {  // Ldst % 4 == 0: the four share a row and are 16-byte aligned
                            const size_t o = ((size_t)n * p.M + row) * p.Ldst + l;
                            if (p.residual != nullptr) {
                                const float4 rv = *reinterpret_cast<const float4*>(p.residual + o);
                                v0 += rv.x; v1 += rv.y; v2 += rv.z; v3 += rv.w;
                            }
                            if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                            *reinterpret_cast<float4*>(p.out + o) = make_float4(v0, v1, v2, v3);
                        } else {
                            float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (pp + e < p.P) {
                                    const size_t o = ((size_t)n * p.M + row) * p.Ldst + l;
                                    float v = vv[e];
                                    if (p.residual != nullptr) v += p.residual[o];
                                    if (p.relu) v = fmaxf(v, 0.f);
                                    p.out[o] = v;
                                }
                                if (++l == p.Ldst) { l = 0; ++n; }
                            }
                        }
                    }
                }
                asm volatile("" ::: "memory");
            }
        }
    }

    if (p.stats != nullptr) {
        // combine the WN wave-columns through LDS (the tile loop is over; As is free) -> ONE partial row per workgroup
        float* red = &As[0][0];  // [WN][BM][2]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s = st_sum[i] + __shfl_xor(st_sum[i], 32, 64);
            const float q = st_sq[i] + __shfl_xor(st_sq[i], 32, 64);
            if (lhi == 0) {
                const int r = wm * TM * 32 + i * 32 + l31;
                red[(wn * BM + r) * 2 + 0] = s;
                red[(wn * BM + r) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BM && (m0 + tid) < p.M) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { s += red[(w * BM + tid) * 2]; q += red[(w * BM + tid) * 2 + 1]; }
            float* dst = p.stats + ((size_t)blockIdx.x * p.M + m0 + tid) * 2;
            dst[0] = s;
            dst[1] = q;
        }
    }
}


// ---------------------------------------------------------------------------
// Fast path (every conv of the ResNet body and head: Csrc % 16 == 0, kernel size 1 or 3, aligned weights).
//
// What the generic kernel above spends its time on is not the MFMAs but the ~200 VALU instructions of gather
// address arithmetic per 32 MFMAs per wave (ablation on the layer4 shape: MFMA+LDS skeleton 0.83 ms, staging
// alone 0.49 ms, together 1.12 ms).  Here the K order c*KS + t repeats with period 16*KS, so the (channel, tap)
// of every gather row is the same in every "super-stage" of KS stages except for a channel advance of 16 -
// which is a SCALAR offset.  Each thread therefore computes its KS*BE gather offsets (with the padding /
// range / stride-parity verdict folded into bit 31) ONCE per position tile; in the K loop a B load is a single
// buffer_load with a precomputed VGPR offset and an SGPR soffset, and an A load is base(SGPR)+offset(VGPR).
// 8 waves per workgroup, 2 workgroups per CU (<= 128 VGPRs): the tile counts of this network at B = 512
// (1008 / 2000 tiles) then fill the 512 slots in whole rounds.
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int KS>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * 64) / 128) void conv_igemm_fast_kernel(ConvP p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    constexpr int APITCH = kBK + 1;
    constexpr int AE = BM * kBK / NT;  // A floats per thread per stage (1, 2, 4 or 8)
    constexpr int BE = BN * kBK / NT;  // gathered B floats per thread per stage
    constexpr int BROWSTEP = NT / BN;
    constexpr int AV = AE >= 4 ? 4 : AE;  // vector width of an A load
    static_assert(BE >= 1 && AE >= 1 && AE <= 8 && NT % BN == 0, "staging shape");
    // Column-base mode (BN == NT: every thread gathers one column and all 16 rows of a stage).  Row i of stage u is
    // then the same (channel, tap) for every thread - compile-time constants after unrolling - so a gather offset
    // is colbase[tap] (3 VGPRs per tile, padding verdict in bit 31) + a scalar channel offset: no offset table.
    constexpr bool CB = (BN == NT);

    // one LDS array: [A stage 0 | A stage 1 | B stage 0 | B stage 1]; the epilogue reuses it as per-wave 32x33 tiles
    constexpr int A_STAGE = BM * APITCH, B_STAGE = kBK * BN;
    static_assert(2 * A_STAGE + 2 * B_STAGE >= (NT / 64) * 32 * 33, "epilogue transpose tiles must fit");
    __shared__ float smem[2 * A_STAGE + 2 * B_STAGE];
    __shared__ float2 sAff[512];  // per-input-channel {scale, shift} of the fused producer BN (Csrc <= 512)
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * A_STAGE;
    __shared__ float2 sEp[BM];    // the epilogue's per-channel (scale, shift) of this workgroup's rows (1, 0 beyond M / without a fold)
    const bool in_aff = p.in_scale != nullptr;
    if (in_aff) {
        for (int c = threadIdx.x; c < p.Csrc; c += NT) sAff[c] = make_float2(p.in_scale[c], p.in_shift[c]);
    }
    for (int r = threadIdx.x; r < BM; r += NT) {
        const int row = blockIdx.y * BM + r;
        sEp[r] = make_float2((p.scale != nullptr && row < p.M) ? p.scale[row] : 1.f, (p.shift != nullptr && row < p.M) ? p.shift[row] : 0.f);
    }
    __syncthreads();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int m0 = blockIdx.y * BM;
    const int nstages = p.nst;  // stages of this K split (all of Ktot / 16 without one): a multiple of KS (launcher-checked)
    const int kz = blockIdx.z;
    float* const outp = p.out + (size_t)kz * p.out_split;

    const int a_row = (tid * AE) / kBK;
    const int a_col = (tid * AE) % kBK;
    // byte offset of this thread's A segment at k0 = 0; rows beyond M are clamped to a valid row (never stored)
    const unsigned a_boff = ((unsigned)((m0 + a_row) < p.M ? (m0 + a_row) : m0) * (unsigned)p.Ktot + (unsigned)a_col) * 4u;
    const int b_col = tid % BN;
    const int b_r0 = tid / BN;
    const auto srcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)p.src_bytes, 0x00020000);
    const unsigned chan_step = (unsigned)(kBK * p.Lsrc) * 4u;  // bytes: one super-stage advances 16 channels
    const unsigned row_bytes = (unsigned)p.Lsrc * 4u;

    float st_sum[TM], st_sq[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { st_sum[i] = 0.f; st_sq[i] = 0.f; }

    for (int pt = blockIdx.x; pt < p.numPT; pt += gridDim.x) {
        const int p0 = pt * BN;
        // ---- per-tile gather offsets: voff[u][i] for stage-in-super-stage u and row i (or colbase[t] in CB mode) ----
        unsigned voff[CB ? 1 : KS][CB ? 1 : BE];
        unsigned colbase[3];
        {
            const int pc = p0 + b_col;
            const bool col_ok = pc < p.P;
            const int gn = col_ok ? pc / p.Ldst : 0;
            const int gl = pc - gn * p.Ldst;
            const unsigned src_off = (unsigned)gn * (unsigned)(p.Csrc * p.Lsrc);
            const int gbase = gl * p.gmul;
            if (CB) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int sidx = gbase + p.tapoff[t < KS ? t : 0];
                    colbase[t] = oob_if((src_off + (unsigned)sidx) * 4u, !(col_ok && (unsigned)sidx < (unsigned)p.Lsrc));
                }
            } else {
#pragma unroll
                for (int u = 0; u < KS; ++u)
#pragma unroll
                    for (int i = 0; i < BE; ++i) {
                        const int kl = u * kBK + b_r0 + i * BROWSTEP;
                        const int c = kl / KS;
                        const int t = kl - c * KS;
                        const int sidx = gbase + (t == 0 ? p.tapoff[0] : (t == 1 ? p.tapoff[1] : p.tapoff[2]));
                        const bool ok = col_ok && ((unsigned)sidx < (unsigned)p.Lsrc);
                        voff[u][i] = oob_if((src_off + (unsigned)(c * p.Lsrc + sidx)) * 4u, !ok);
                    }
            }
        }

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        float ra[AE], rb[BE];
        // stage s = S*KS + u;  a_base = weights at k0 = 16*s (uniform), soff = channel advance of super-stage S
        auto load_stage = [&](const float* a_base, int u, unsigned soff) {
            const char* ab = reinterpret_cast<const char*>(a_base) + a_boff;
#pragma unroll
            for (int q = 0; q < AE / AV; ++q) {
                if (AV == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(ab + 16 * q);
                    ra[4 * q + 0] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w;
                } else if (AV == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(ab + 8 * q);
                    ra[2 * q + 0] = v.x; ra[2 * q + 1] = v.y;
                } else {
                    ra[q] = *reinterpret_cast<const float*>(ab + 4 * q);
                }
            }
#pragma unroll
            for (int i = 0; i < BE; ++i) {
                if (CB) {
                    const int kl = u * kBK + i;  // u and i are literals after unrolling: c and t fold to constants
                    const int c = kl / KS;
                    const int t = kl - c * KS;
                    const unsigned vb = t == 0 ? colbase[0] : (t == 1 ? colbase[1] : colbase[2]);
                    rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srcR, vb, soff + (unsigned)c * row_bytes, 0));
                } else {
                    rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srcR, voff[u][i], soff, 0));
                }
            }
        };
        // u / chan0: position in the super-stage and first channel (16*S) of the stage held in ra/rb
        auto store_stage = [&](int buf, int u, int chan0) {
#pragma unroll
            for (int q = 0; q < AE; ++q) As0[buf * A_STAGE + a_row * APITCH + a_col + q] = ra[q];
            if (in_aff) {
#pragma unroll
                for (int i = 0; i < BE; ++i) {
                    const int kl = u * kBK + (CB ? 0 : b_r0) + i * BROWSTEP;
                    const int c = kl / KS;
                    const int t = kl - c * KS;
                    const unsigned vo = CB ? (t == 0 ? colbase[0] : (t == 1 ? colbase[1] : colbase[2])) : voff[CB ? 0 : u][CB ? 0 : i];
                    const float2 ab = sAff[chan0 + c];
                    const float v = fmaxf(fmaf(rb[i], ab.x, ab.y), 0.f);
                    rb[i] = (int)vo < 0 ? 0.f : v;  // bit 31 = padding / out of range: stays exactly 0
                }
            }
#pragma unroll
            for (int i = 0; i < BE; ++i) Bs0[buf * B_STAGE + (b_r0 + i * BROWSTEP) * BN + b_col] = rb[i];
        };
        auto mfma_stage = [&](int buf) {
            const float* as = &As0[buf * A_STAGE + (wm * TM * 32 + l31) * APITCH + lhi];
            const float* bs = &Bs0[buf * B_STAGE + lhi * BN + wn * TN * 32 + l31];
#pragma unroll
            for (int ks = 0; ks < kBK / 2; ++ks) {
                // keep the scheduler from hoisting all 8 k-steps' fragment reads to the top (register pressure)
                if (ks == kBK / 4) __builtin_amdgcn_sched_barrier(0);
                float wv[TM], xv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) wv[i] = as[i * 32 * APITCH + 2 * ks];
#pragma unroll
                for (int j = 0; j < TN; ++j) xv[j] = bs[2 * ks * BN + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[j], wv[i], acc[i][j], 0, 0, 0);
            }
        };

        const float* a_base = p.A + (size_t)kz * nstages * kBK;
        unsigned soff = (unsigned)kz * (unsigned)(nstages / KS) * chan_step;
        int chan0 = kz * (nstages / KS) * kBK;
        load_stage(a_base, 0, soff);
        __syncthreads();  // the previous tile's readers are done with both LDS buffers
        store_stage(0, 0, chan0);
        __syncthreads();
        int buf = 0;
        for (int s = 0; s < nstages; s += KS) {
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const bool more = (s + u + 1) < nstages;
                a_base += kBK;
                if (more) {
                    if (u + 1 < KS) load_stage(a_base, u + 1, soff);
                    else load_stage(a_base, 0, soff + chan_step);
                }
                mfma_stage(buf);
                if (more) {
                    if (u + 1 < KS) store_stage(buf ^ 1, u + 1, chan0);
                    else store_stage(buf ^ 1, 0, chan0 + kBK);
                }
                __syncthreads();
                buf ^= 1;
            }
            soff += chan_step;
            chan0 += kBK;
        }

        // ---------------- epilogue (same as the generic kernel) ----------------
        if (p.stats != nullptr) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];
                        s += v;
                        q = fmaf(v, v, q);
                    }
                st_sum[i] += s;
                st_sq[i] += q;
            }
        }
        // Output through a per-wave LDS transpose.  In the accumulators a lane owns one CHANNEL (rows of the output
        // tensor) and 16 positions; storing from there makes every store instruction touch 64 different rows (64
        // cache lines, 4 B each) - for the short-K layers that cost more than the MFMA loop.  Each 32x32 sub-tile is
        // therefore written to LDS [channel][position] and read back with the lane on the POSITION axis: a store
        // instruction then writes two 128-byte row segments, and residual reads are coalesced the same way.
        // (same-wave LDS accesses execute in order; the K loop ended on a workgroup barrier, so the staging buffers
        // are free)
        {
            // an opaque zero keeps the epilogue's address arithmetic from being hoisted out of the tile loop, where it
            // would be live (and spilled) across the whole K loop
            int opq = 0;
            asm volatile("" : "+s"(opq));
            float* T = smem + wave * (32 * 33) + opq;
            const bool plain = p.scale == nullptr && p.shift == nullptr && p.residual == nullptr && !p.relu;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int pos = p0 + wn * TN * 32 + j * 32 + l31;
                const bool pok = pos < p.P;
                const int n = pok ? pos / p.Ldst : 0;
                const int l = pos - n * p.Ldst;
                // 32-bit element offsets (every tensor is < 2 GiB, launcher-checked): base pointer stays scalar
                const unsigned obase = (unsigned)n * (unsigned)(p.M * p.Lrow) + (unsigned)(l * p.ostride + p.ooff);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * lhi] = acc[i][j][r];
                    asm volatile("" ::: "memory");
                    const int rbase = m0 + wm * TM * 32 + i * 32 + lhi + opq;
                    unsigned o = obase + (unsigned)(rbase * p.Lrow);
                    const unsigned ostep = 2u * (unsigned)p.Lrow;
                    if (plain) {  // train-mode forward / plain data gradient: nothing but the store
#pragma unroll
                        for (int k2 = 0; k2 < 16; ++k2) {
                            const float v = T[(2 * k2 + lhi) * 33 + l31];
                            if (pok && (rbase + 2 * k2) < p.M) outp[o] = v;
                            o += ostep;
                        }
                    } else {
                        epilogue_rows_fused(T, lhi, l31, pok, rbase, p.M, o, ostep, p.scale, p.shift, p.residual, p.relu, outp, sEp, m0);
                    }
                }
            }
        }
        __syncthreads();  // the next tile's staging overwrites the transpose tiles
    }

    if (p.stats != nullptr) {
        float* red = smem;  // [WN][BM][2]  (BM*WN*2 floats <= BM*17 for WN <= 8)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s = st_sum[i] + __shfl_xor(st_sum[i], 32, 64);
            const float q = st_sq[i] + __shfl_xor(st_sq[i], 32, 64);
            if (lhi == 0) {
                const int r = wm * TM * 32 + i * 32 + l31;
                red[(wn * BM + r) * 2 + 0] = s;
                red[(wn * BM + r) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BM && (m0 + tid) < p.M) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { s += red[(w * BM + tid) * 2]; q += red[(w * BM + tid) * 2 + 1]; }
            float* dst = p.stats + ((size_t)blockIdx.x * p.M + m0 + tid) * 2;
            dst[0] = s;
            dst[1] = q;
        }
    }
}

// ---------------------------------------------------------------------------
// K split: the finishing pass (see conv_common.h::launch_split_finish)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_finish_kernel(const float* __restrict__ part, int S, size_t plane, float* out, int N,
                                                            int M, int Ldst, int Lrow, int ostride, int ooff,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* residual, int relu, float* __restrict__ stats, int R) {
    const int m = blockIdx.x, r = blockIdx.y;      // gridDim.y > R only with statistics: the surplus rows are written as zeros
    if (r >= R) {
        if (threadIdx.x < 2) stats[((size_t)r * M + m) * 2 + threadIdx.x] = 0.f;
        return;
    }
    const int per = (N + R - 1) / R;
    const int n0 = r * per, n1 = min(N, n0 + per);
    const float sc = scale != nullptr ? scale[m] : 1.f, sh = shift != nullptr ? shift[m] : 0.f;
    float s = 0.f, q = 0.f;
    const int items = n1 > n0 ? (n1 - n0) * Ldst : 0;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int dn = it / Ldst, j = it - dn * Ldst;
        const size_t e = ((size_t)(n0 + dn) * M + m) * (size_t)Lrow + (size_t)j * ostride + ooff;
        float v = part[e];
        for (int z = 1; z < S; ++z) v += part[(size_t)z * plane + e];      // split order: fixed, reproducible
        if (scale != nullptr) v *= sc;
        if (shift != nullptr) v += sh;
        if (residual != nullptr) v += residual[e];
        if (relu) v = fmaxf(v, 0.f);
        out[e] = v;
        s += v;
        q = fmaf(v, v, q);
    }
    if (stats != nullptr) {
        __shared__ float red[2][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float* dst = stats + ((size_t)r * M + m) * 2;
            dst[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            dst[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}

// The same without statistics on a dense output (the eval pass, stride-1 data gradients): a flat sweep of the plane, 16 bytes per
// lane where the plane allows - rows of 63 / 125 floats are then not split between workgroups (no partial cache lines re-fetched
// by another XCD), which is what the per-channel grid above pays for its in-block statistics.
template <bool VEC>
__global__ __launch_bounds__(256) void split_finish_flat_kernel(const float* __restrict__ part, int S, size_t plane, float* out, int M,
                                                                 int L, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* residual, int relu) {
    constexpr int W = VEC ? 4 : 1;
    const size_t nv = plane / W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (size_t)gridDim.x * 256) {
        const size_t e = v * W;
        float a[W];
        if (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(part + e);
            a[0] = t.x; a[W > 1 ? 1 : 0] = t.y; a[W > 2 ? 2 : 0] = t.z; a[W > 3 ? 3 : 0] = t.w;
        } else {
            a[0] = part[e];
        }
        for (int z = 1; z < S; ++z) {
            if (VEC) {
                const float4 t = *reinterpret_cast<const float4*>(part + (size_t)z * plane + e);
                a[0] += t.x; a[W > 1 ? 1 : 0] += t.y; a[W > 2 ? 2 : 0] += t.z; a[W > 3 ? 3 : 0] += t.w;
            } else {
                a[0] += part[(size_t)z * plane + e];
            }
        }
        float r[W];
        if (residual != nullptr) {
            if (VEC) {
                const float4 t = *reinterpret_cast<const float4*>(residual + e);
                r[0] = t.x; r[W > 1 ? 1 : 0] = t.y; r[W > 2 ? 2 : 0] = t.z; r[W > 3 ? 3 : 0] = t.w;
            } else {
                r[0] = residual[e];
            }
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            float x = a[i];
            if (scale != nullptr || shift != nullptr) {
                const int m = (int)(((e + i) / (size_t)L) % (size_t)M);
                if (scale != nullptr) x *= scale[m];
                if (shift != nullptr) x += shift[m];
            }
            if (residual != nullptr) x += r[i];
            if (relu) x = fmaxf(x, 0.f);
            a[i] = x;
        }
        if (VEC) *reinterpret_cast<float4*>(out + e) = make_float4(a[0], a[W > 1 ? 1 : 0], a[W > 2 ? 2 : 0], a[W > 3 ? 3 : 0]);
        else out[e] = a[0];
    }
}

}  // namespace

namespace ssecg_detail {
int launch_split_finish(const float* part, int S, size_t plane, float* out, int N, int M, int Ldst, int Lrow, int ostride, int ooff,
                        const float* scale, const float* shift, const float* residual, int relu, float* stats, int stats_parts,
                        hipStream_t st) {
    if (stats == nullptr && ostride == 1 && ooff == 0 && Ldst == Lrow) {
        const bool vec = plane % 4 == 0 && (((uintptr_t)part | (uintptr_t)out | (uintptr_t)residual) & 15) == 0;
        const size_t want = (plane / (vec ? 4 : 1) + 255) / 256;
        const dim3 grid((unsigned)(want < 2048 ? want : 2048));
        if (vec) hipLaunchKernelGGL(split_finish_flat_kernel<true>, grid, dim3(256), 0, st, part, S, plane, out, M, Ldst, scale, shift, residual, relu);
        else hipLaunchKernelGGL(split_finish_flat_kernel<false>, grid, dim3(256), 0, st, part, S, plane, out, M, Ldst, scale, shift, residual, relu);
        return (int)hipGetLastError();
    }
    int R = 1024 / M;                       // ~1024 workgroups
    if (R < 1) R = 1;
    if (R > N) R = N;
    int rows = R;
    if (stats != nullptr) {
        if (stats_parts < 1) return SSECG_E_WORKSPACE;
        if (R > stats_parts) R = stats_parts;
        rows = stats_parts;                 // rows beyond R must read as zero: written by the surplus workgroups, no memset launch
    }
    hipLaunchKernelGGL(split_finish_kernel, dim3(M, rows), dim3(256), 0, st, part, S, plane, out, N, M, Ldst, Lrow, ostride, ooff, scale,
                       shift, residual, relu, stats, R);
    return (int)hipGetLastError();
}
}  // namespace ssecg_detail

namespace {

// ---------------------------------------------------------------------------
// tile configuration shared by the launcher and ssecg_conv1d_stats_parts
// ---------------------------------------------------------------------------
struct TileCfg { int BM, BN, numPT, MT, G; bool fast; };
#ifndef SSECG_FAST128_BN
#define SSECG_FAST128_BN 128
#endif
constexpr int kFast128BN = SSECG_FAST128_BN;  // position-tile width of the 128-channel fast config (128 or 256)

// fast path eligibility (see conv_igemm_fast_kernel); a_vec = weights 16-byte aligned and Ktot % 4 == 0
inline bool fast_ok(int M, int Csrc, int KS, bool a_vec) {
    return a_vec && M > 32 && (Csrc % kBK == 0) && (KS >= 1 && KS <= 3);
}

inline TileCfg pick_cfg(int M, long long P, bool fast) {
    TileCfg c;
    c.fast = fast;
    int slots;
    if (fast) {
        if (M > 128) { c.BM = 256; c.BN = 128; }
        else if (M > 64) { c.BM = 128; c.BN = kFast128BN; }
        else { c.BM = 64; c.BN = 512; }
        slots = kNumCU * 2;  // 8-wave workgroups, 2 per CU
    } else {
        if (M > 64) { c.BM = 128; c.BN = 128; }
        else if (M > 32) { c.BM = 64; c.BN = 256; }
        else { c.BM = 32; c.BN = 256; }
        slots = kNumCU * kIgemmWgPerCu;
    }
    c.numPT = (int)((P + c.BN - 1) / c.BN);
    c.MT = (M + c.BM - 1) / c.BM;
    // one co-resident wave of workgroups shared by the MT channel tiles; each workgroup strides over the position
    // tiles (a tail round costs a whole tile-time, so never launch more than one wave)
    int g = slots / c.MT;
    if (g < 1) g = 1;
    c.G = c.numPT < g ? c.numPT : g;
    return c;
}

// K split of a fast-path launch: S > 1 when the launch has fewer tiles than workgroup slots, the caller gave a workspace and the
// contraction divides into S runs of whole 16-channel super-stages (>= 2 each)
inline int pick_igemm_split(const ConvP& p, int KS, const TileCfg& c) {
    if (!c.fast || p.Ktot != p.Csrc * KS) return 1;
    return ssecg_detail::pick_ksplit((long long)c.numPT * c.MT, c.numPT, kNumCU * 2, p.Csrc, kBK);
}

template <int MODE>
int launch_igemm(const ConvP& p0, int KS, const TileCfg& c, hipStream_t st, float* split_ws = nullptr, size_t split_ws_bytes = 0,
                 int stats_parts = 0) {
    ConvP p = p0;
    p.nst = p.Ktot / kBK;
    p.out_split = 0;
    if (c.fast) {
        dim3 grid(c.G, c.MT), block(512);
        const int S = split_ws != nullptr ? pick_igemm_split(p, KS, c) : 1;
        const size_t plane = (size_t)p.N * p.M * p.Lrow;
        if (S > 1) {
            if (split_ws_bytes < (size_t)S * plane * sizeof(float)) return SSECG_E_WORKSPACE;
            p.nst /= S; p.out_split = plane; p.out = split_ws;
            p.scale = nullptr; p.shift = nullptr; p.residual = nullptr; p.relu = 0; p.stats = nullptr;
            grid.z = S;
        }
#define SSECG_FAST(BM_, BN_, WM_, WN_)                                                                               \
    do {                                                                                                             \
        if (KS == 3) hipLaunchKernelGGL((conv_igemm_fast_kernel<BM_, BN_, WM_, WN_, 3>), grid, block, 0, st, p);      \
        else if (KS == 2) hipLaunchKernelGGL((conv_igemm_fast_kernel<BM_, BN_, WM_, WN_, 2>), grid, block, 0, st, p); \
        else hipLaunchKernelGGL((conv_igemm_fast_kernel<BM_, BN_, WM_, WN_, 1>), grid, block, 0, st, p);             \
    } while (0)
        if (c.BM == 256) SSECG_FAST(256, 128, 4, 2);
        else if (c.BM == 128 && c.BN == 256) SSECG_FAST(128, 256, 2, 4);
        else if (c.BM == 128) SSECG_FAST(128, 128, 2, 4);
        else SSECG_FAST(64, 512, 1, 8);
#undef SSECG_FAST
        if (S > 1)
            return ssecg_detail::launch_split_finish(split_ws, S, plane, p0.out, p0.N, p0.M, p0.Ldst, p0.Lrow, p0.ostride, p0.ooff,
                                                     p0.scale, p0.shift, p0.residual, p0.relu, p0.stats, stats_parts, st);
        return (int)hipGetLastError();
    }
    dim3 grid(c.G, c.MT), block(kThreads);
#define SSECG_LAUNCH(BM_, BN_, WM_, WN_, KS_)                                                        \
    do {                                                                                             \
        if (p.a_vec && (BM_) >= 64)                                                                  \
            hipLaunchKernelGGL((conv_igemm_kernel<BM_, BN_, WM_, WN_, KS_, MODE, ((BM_) >= 64)>), grid, block, 0, st, p); \
        else                                                                                         \
            hipLaunchKernelGGL((conv_igemm_kernel<BM_, BN_, WM_, WN_, KS_, MODE, false>), grid, block, 0, st, p);         \
    } while (0)
#define SSECG_BY_KS(BM_, BN_, WM_, WN_)                                                              \
    switch (KS) {                                                                                    \
        case 1: SSECG_LAUNCH(BM_, BN_, WM_, WN_, 1); break;                                          \
        case 3: SSECG_LAUNCH(BM_, BN_, WM_, WN_, 3); break;                                          \
        case 7: SSECG_LAUNCH(BM_, BN_, WM_, WN_, 7); break;                                          \
        default: return SSECG_E_INVAL;                                                               \
    }
    if (c.BM == 128) { SSECG_BY_KS(128, 128, 2, 2) }
    else if (c.BM == 64) { SSECG_BY_KS(64, 256, 1, 4) }
    else { SSECG_BY_KS(32, 256, 1, 4) }
#undef SSECG_BY_KS
#undef SSECG_LAUNCH
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------
constexpr int kBKP = 32;  // positions per LDS stage

struct WgradP {
    const float* dy;
    const float* x;
    float* ws;
    unsigned dy_bytes, x_bytes;
    const float* x_scale;  // FAST only: the x operand is relu(x*x_scale[ci] + x_shift[ci]) (fused producer BN + ReLU)
    const float* x_shift;
    int MT, JT, Z;  // tiles over Cout, tiles over (tap, Cin), position slabs
    int N, Cout, Ldy, Csrc, Lx, KS, J;
    int stride, pad, dil;
    long long P;
    long long chunk;  // positions per split (multiple of kBKP)
};

// FAST (every conv of the body/head: Csrc % BJ == 0, Cout even): a tile's BJ columns then share ONE tap, the rows a
// wave stages are wave-uniform, and a load is buffer_load(voffset = this lane's position base, soffset = row) -
// no per-load address arithmetic (the generic path spends ~250 VALU per 64 MFMAs per wave on it).
template <int BM, int BJ, int WM, int WJ, bool FAST>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(WgradP p) {
    static_assert(WM * WJ == 4, "4 waves");
    constexpr int TM = BM / (32 * WM);
    constexpr int TJ = BJ / (32 * WJ);
    constexpr int PITCH = kBKP + 1;
    constexpr int AR = BM / 8;  // A rows per thread per stage
    constexpr int BR = BJ / 8;

    __shared__ float As[BM * PITCH];
    __shared__ float Bs[BJ * PITCH];
    __shared__ int rowOff[BJ];  // ci*Lx           (or -1 when j >= J)
    __shared__ int rowTap[BJ];  // t*dil - pad
    __shared__ float2 sAffW[FAST ? BJ : 1];  // {scale, shift} of this tile's BJ input channels

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WJ;
    const int wj = wave % WJ;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (round-robin dispatch, a speed assumption only).  All
    // MT*JT tiles of one position slab are given consecutive slots of ONE XCD, so the dy / x rows of that slab are
    // fetched into that XCD's L2 once and shared by the tiles, instead of once per tile from HBM (measured before
    // this mapping: 2.2 GB of fabric reads per launch for two 132 MB operands).
    const int tiles = p.MT * p.JT;
    const int slot = blockIdx.x >> 3;
    const int zslab = (slot / tiles) * 8 + (blockIdx.x & 7);
    if (zslab >= p.Z) return;
    const int tile = slot % tiles;
    const int j0 = (tile % p.JT) * BJ;
    const int m0 = (tile / p.JT) * BM;
    const long long kbeg = (long long)zslab * p.chunk;
    long long kend = kbeg + p.chunk;
    if (kend > p.P) kend = p.P;
    const int nstages = kend > kbeg ? (int)((kend - kbeg + kBKP - 1) / kBKP) : 0;

    for (int j = tid; j < BJ; j += kThreads) {
        const int jj = j0 + j;
        if (jj < p.J) {
            const int t = jj / p.Csrc;
            const int ci = jj - t * p.Csrc;
            rowOff[j] = ci * p.Lx;
            rowTap[j] = t * p.dil - p.pad;
        } else {
            rowOff[j] = -1;
            rowTap[j] = 0;
        }
    }
    __syncthreads();

    const int ppos = tid & 31;
    const int rg = tid >> 5;  // 0..7
    const bool x_aff = FAST && p.x_scale != nullptr;
    if (x_aff) {
        const int cbase = j0 - (j0 / p.Csrc) * p.Csrc;
        for (int j = tid; j < BJ; j += kThreads) sAffW[j] = make_float2(p.x_scale[cbase + j], p.x_shift[cbase + j]);
        __syncthreads();
    }
    bool b_ok = false;  // FAST: validity of this lane's x column in the stage held in rb

    f32x16 acc[TM][TJ];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[AR], rb[BR];
    const auto dyR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
    const auto xR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    // wave-uniform row parts (FAST): rows of this wave are (2*wave + lhi) + 8*i; the lhi part rides in the lane base
    const int wave2 = __builtin_amdgcn_readfirstlane(wave) * 2;
    const int tapF = (j0 / p.Csrc) * p.dil - p.pad;      // the tile's single tap (FAST)
    const int ci0F = j0 - (j0 / p.Csrc) * p.Csrc + wave2;  // first input channel staged by this wave (FAST)
    int rowA0 = m0 + wave2;                                 // first dy row staged by this wave, clamped into range
    if (rowA0 > p.Cout - 2) rowA0 = p.Cout - 2;
    // buffer loads: padding / out-of-range lanes use an offset beyond num_records and read 0 (no branch, no select)
    auto load_stage = [&](int s) {
        const long long pp = kbeg + (long long)s * kBKP + ppos;
        const bool ok = pp < kend;
        const int n = ok ? (int)(pp / p.Ldy) : 0;
        const int l = ok ? (int)(pp - (long long)n * p.Ldy) : 0;
        if (FAST) {
            const int sidx = l * p.stride + tapF;
            const unsigned baseA = oob_if(((unsigned)n * (unsigned)(p.Cout * p.Ldy) + (unsigned)(l + lhi * p.Ldy)) * 4u, !ok);
            b_ok = ok && (unsigned)sidx < (unsigned)p.Lx;
            const unsigned baseB = oob_if(((unsigned)n * (unsigned)(p.Csrc * p.Lx) + (unsigned)(sidx + lhi * p.Lx)) * 4u, !b_ok);
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                int row = rowA0 + 8 * i;  // rows beyond Cout re-read a valid row: their accumulators are never stored
                if (row > p.Cout - 2) row = p.Cout - 2;
                ra[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyR, baseA, row * p.Ldy * 4, 0));
            }
#pragma unroll
            for (int i = 0; i < BR; ++i)
                rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xR, baseB, (ci0F + 8 * i) * p.Lx * 4, 0));
            return;
        }
        const unsigned dy_off = (unsigned)n * (unsigned)(p.Cout * p.Ldy) + (unsigned)l;
        const unsigned x_off = (unsigned)n * (unsigned)(p.Csrc * p.Lx);
        const int lx0 = l * p.stride;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int co = m0 + rg + 8 * i;
            ra[i] = buf_load_f32(dyR, oob_if((dy_off + (unsigned)(co * p.Ldy)) * 4u, !(ok && co < p.Cout)));
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int j = rg + 8 * i;
            const int roff = rowOff[j];
            const int sidx = lx0 + rowTap[j];
            const bool okb = ok && (roff >= 0) && ((unsigned)sidx < (unsigned)p.Lx);
            rb[i] = buf_load_f32(xR, oob_if((x_off + (unsigned)(roff + sidx)) * 4u, !okb));
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) As[(rg + 8 * i) * PITCH + ppos] = ra[i];
        if (x_aff) {
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const float2 ab = sAffW[rg + 8 * i];
                rb[i] = b_ok ? fmaxf(fmaf(rb[i], ab.x, ab.y), 0.f) : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) Bs[(rg + 8 * i) * PITCH + ppos] = rb[i];
    };

    if (nstages > 0) load_stage(0);
    for (int s = 0; s < nstages; ++s) {
        __syncthreads();  // readers of the previous stage are done
        store_stage();
        __syncthreads();
        if (s + 1 < nstages) load_stage(s + 1);
        const float* as = &As[(wm * TM * 32 + l31) * PITCH + lhi];
        const float* bs = &Bs[(wj * TJ * 32 + l31) * PITCH + lhi];
#pragma unroll
        for (int ks = 0; ks < kBKP / 2; ++ks) {
            float a[TM], b[TJ];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[i * 32 * PITCH + 2 * ks];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = bs[j * 32 * PITCH + 2 * ks];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // slab store: ws[z][co][j]
    float* ws = p.ws + (size_t)zslab * p.Cout * p.J;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int col = j0 + wj * TJ * 32 + j * 32 + l31;
        if (col >= p.J) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row < p.Cout) ws[(size_t)row * p.J + col] = acc[i][j][r];
            }
    }
}

// dw[co][ci][t] = sum_z ws[z][co][t*Csrc + ci]   (fixed order -> reproducible).  64 elements per workgroup x 4 slab
// lanes (wave zl sums slabs zl, zl+4, ...), combined in a fixed order through LDS: enough workgroups for small weights.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, float* dw, int Z, int Cout, int Csrc, int KS) {
    __shared__ float part[4][64];
    const int J = Csrc * KS;
    const size_t total = (size_t)Cout * J;
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    for (size_t e0 = (size_t)blockIdx.x * 64; e0 < total; e0 += (size_t)gridDim.x * 64) {
        const size_t e = e0 + el;
        float s0 = 0.f, s1 = 0.f;  // two loads in flight; the summation order stays fixed
        if (e < total) {
            int z = zl;
            for (; z + 4 < Z; z += 8) {
                s0 += ws[(size_t)z * total + e];
                s1 += ws[(size_t)(z + 4) * total + e];
            }
            if (z < Z) s0 += ws[(size_t)z * total + e];
        }
        part[zl][el] = s0 + s1;
        __syncthreads();
        if (zl == 0 && e < total) {
            const int co = (int)(e / J);
            const int j = (int)(e - (size_t)co * J);
            const int t = j / Csrc;
            const int ci = j - t * Csrc;
            dw[((size_t)co * Csrc + ci) * KS + t] = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
        }
        __syncthreads();
    }
}

struct WgradCfg { int BM, BJ, MT, JT, Z; long long chunk; };

inline WgradCfg pick_wgrad(int Cout, int Csrc, int KS, long long P) {
    WgradCfg c;
    const int J = Csrc * KS;
    c.BM = (Cout > 64) ? 128 : 64;
    c.BJ = (J >= 128 && (Csrc % 128 == 0 || J > 256)) ? 128 : 64;
    if (c.BM == 128 && c.BJ == 64) { /* supported */ }
    if (c.BM == 64 && c.BJ == 128) c.BJ = 64;
    c.MT = (Cout + c.BM - 1) / c.BM;
    c.JT = (J + c.BJ - 1) / c.BJ;
    constexpr int wg_per_cu = 4;   // one co-resident wave of workgroups: 4 per CU fit (<= 128 registers)
    long long z = (kNumCU * wg_per_cu) / (c.MT * c.JT);
    const long long zmax = (P + 255) / 256;  // at least 8 stages per split
    if (z > zmax) z = zmax;
    if (z < 1) z = 1;
    if (z > 8) z = z / 8 * 8;  // whole groups of 8 slabs: one slab group per XCD round (see conv_wgrad_kernel)
    long long chunk = (P + z - 1) / z;
    chunk = (chunk + kBKP - 1) / kBKP * kBKP;
    z = (P + chunk - 1) / chunk;
    c.Z = (int)z;
    c.chunk = chunk;
    return c;
}

// wt[ci][co][t] = w[co][ci][t].  For a stride-2 conv with 3 taps the data gradient splits by output parity
// (even inputs see only tap 1, odd inputs taps 0 and 2), so the operand is packed as two matrices:
//   phase 0: [ci][co]     = w[co][ci][1]              at wt
//   phase 1: [ci][co][2]  = { w[co][ci][0], w[co][ci][2] }   at wt + Cin*Cout
__global__ void transpose_weight_kernel(const float* w, float* wt, int Cout, int Cin, int KS, int phased) {
    const size_t total = (size_t)Cout * Cin * KS;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(e % KS);
        const size_t r = e / KS;
        const int co = (int)(r % Cout);
        const int ci = (int)(r / Cout);
        const float v = w[((size_t)co * Cin + ci) * KS + t];
        if (!phased) wt[e] = v;  // e indexes wt[ci][co][t]
        else if (t == 1) wt[(size_t)ci * Cout + co] = v;
        else wt[(size_t)Cin * Cout + ((size_t)ci * Cout + co) * 2 + (t >> 1)] = v;
    }
}

// a 3-tap stride-2 data gradient runs as two parity phases over a packed operand when the fast kernel applies
inline bool dgrad_phased(int Cin, int Cout, int ksize, int stride, const void* wt) {
    return stride == 2 && ksize == 3 && Cin > 32 && (Cout % kBK == 0) && (((uintptr_t)wt & 15) == 0);
}

// operands are addressed with 32-bit byte offsets through buffer descriptors: each must stay below 2 GiB
inline bool fits_descriptor(size_t a_elems, size_t b_elems) {
    return a_elems * 4 < 0x7fffff00ull && b_elems * 4 < 0x7fffff00ull;
}

inline bool bad_conv_shape(int N, int Cin, int Lin, int Cout, int Lout, int k, int s, int pad, int dil) {
    if (N <= 0 || Cin <= 0 || Lin <= 0 || Cout <= 0 || Lout <= 0) return true;
    if (!(k == 1 || k == 3 || k == 7)) return true;
    if (!(s == 1 || s == 2) || pad < 0 || dil < 1) return true;
    const int expect = (Lin + 2 * pad - dil * (k - 1) - 1) / s + 1;
    return expect != Lout;
}

}  // namespace

extern "C" {

int ssecg_conv1d_stats_parts(int N, int Cin, int Cout, int Lout, int ksize) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0 || ksize <= 0) return SSECG_E_INVAL;
    // rows needed by whichever kernel the launcher ends up choosing (alignment of w is not known here)
    const TileCfg a = pick_cfg(Cout, (long long)N * Lout, false);
    const TileCfg b = pick_cfg(Cout, (long long)N * Lout, true);
    return a.G > b.G ? a.G : b.G;
}

int ssecg_conv1d_fwd(const float* x, const float* w, float* y, int N, int Cin, int Lin, int Cout, int Lout,
                     int ksize, int stride, int pad, int dil, const float* scale, const float* shift,
                     const float* residual, int relu, float* stats_partial, int stats_parts, const float* in_scale,
                     const float* in_shift, float* split_ws, size_t split_ws_bytes, void* stream) {
    if (!x || !w || !y || bad_conv_shape(N, Cin, Lin, Cout, Lout, ksize, stride, pad, dil)) return SSECG_E_INVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return SSECG_E_INVAL;
    const long long P = (long long)N * Lout;
    if (P > 0x7fffffffLL) return SSECG_E_INVAL;
    const bool a_vec = ((Cin * ksize) % 4 == 0) && (((uintptr_t)w & 15) == 0);
    const TileCfg c = pick_cfg(Cout, P, fast_ok(Cout, Cin, ksize, a_vec));
    if (in_scale != nullptr && !(c.fast && Cin <= 512)) return SSECG_E_INVAL;  // fused input BN: fast kernel only
    ConvP probe; probe.Csrc = Cin; probe.Ktot = Cin * ksize;
    const bool will_split = split_ws != nullptr && pick_igemm_split(probe, ksize, c) > 1;   // (the finishing pass writes every row)
    if (stats_partial != nullptr && !will_split) {
        if (stats_parts < c.G) return SSECG_E_WORKSPACE;
        if (stats_parts > c.G) {  // rows no workgroup writes must read as zero
            const hipError_t e = hipMemsetAsync(stats_partial + (size_t)c.G * Cout * 2, 0,
                                                (size_t)(stats_parts - c.G) * Cout * 2 * sizeof(float), (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    ConvP p;
    p.A = w; p.src = x; p.out = y;
    p.N = N; p.M = Cout; p.Csrc = Cin; p.Lsrc = Lin; p.Ldst = Lout; p.Ktot = Cin * ksize;
    p.stride = stride; p.pad = pad; p.dil = dil;
    p.P = (int)P; p.numPT = c.numPT;
    p.a_vec = a_vec;
    if (!fits_descriptor((size_t)Cout * p.Ktot, (size_t)N * Cin * Lin)) return SSECG_E_INVAL;
    p.a_bytes = (unsigned)((size_t)Cout * p.Ktot * 4); p.src_bytes = (unsigned)((size_t)N * Cin * Lin * 4);
    p.gmul = stride; p.tapoff[0] = -pad; p.tapoff[1] = dil - pad; p.tapoff[2] = 2 * dil - pad;
    p.Lrow = Lout; p.ostride = 1; p.ooff = 0;
    p.scale = scale; p.shift = shift; p.residual = residual; p.relu = relu; p.stats = stats_partial;
    p.in_scale = in_scale; p.in_shift = in_shift;
    p.out_vec = (Lout % 4 == 0) && (((uintptr_t)y & 15) == 0) && (residual == nullptr || ((uintptr_t)residual & 15) == 0);
    return launch_igemm<0>(p, ksize, c, (hipStream_t)stream, split_ws, split_ws_bytes, stats_parts);
}

// bytes of split workspace ssecg_conv1d_fwd / ssecg_conv1d_dgrad can use for this shape (0: the launch is not split).  An upper
// bound: it assumes the aligned fast path; a launch that takes another kernel ignores the workspace.
size_t ssecg_conv1d_fwd_split_workspace(int N, int Cin, int Lin, int Cout, int Lout, int ksize) {
    (void)Lin;
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0 || ksize <= 0) return 0;
    const TileCfg c = pick_cfg(Cout, (long long)N * Lout, fast_ok(Cout, Cin, ksize, true));
    ConvP p; p.Csrc = Cin; p.Ktot = Cin * ksize;
    const int S = pick_igemm_split(p, ksize, c);
    return S > 1 ? (size_t)S * N * Cout * Lout * sizeof(float) : 0;
}

size_t ssecg_conv1d_dgrad_split_workspace(int N, int Cin, int Lin, int Cout, int Lout, int ksize, int stride) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0 || Lin <= 0 || ksize <= 0) return 0;
    int S = 1;
    ConvP p; p.Csrc = Cout;
    if (stride == 2 && (ksize == 3 || ksize == 1)) {       // phase launches: 1 / 2 taps onto every other input position
        for (int phase = 0; phase < (ksize == 3 ? 2 : 1); ++phase) {
            const int Lq = phase == 0 ? (Lin + 1) / 2 : Lin / 2, ks = phase == 0 ? 1 : 2;
            if (Lq == 0) continue;
            const TileCfg c = pick_cfg(Cin, (long long)N * Lq, true);
            p.Ktot = Cout * ks;
            const int s2 = pick_igemm_split(p, ks, c);
            S = s2 > S ? s2 : S;
        }
    } else if (stride == 1) {
        const TileCfg c = pick_cfg(Cin, (long long)N * Lin, fast_ok(Cin, Cout, ksize, true));
        p.Ktot = Cout * ksize;
        S = pick_igemm_split(p, ksize, c);
    }
    return S > 1 ? (size_t)S * N * Cin * Lin * sizeof(float) : 0;
}

int ssecg_conv1d_transpose_weight(const float* w, float* wt, int Cout, int Cin, int ksize, int stride, void* stream) {
    if (!w || !wt || Cout <= 0 || Cin <= 0 || ksize <= 0 || stride < 1) return SSECG_E_INVAL;
    const size_t total = (size_t)Cout * Cin * ksize;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(transpose_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin, ksize,
                       dgrad_phased(Cin, Cout, ksize, stride, wt) ? 1 : 0);
    return (int)hipGetLastError();
}

int ssecg_conv1d_dgrad(const float* dy, const float* wt, float* dx, int N, int Cin, int Lin, int Cout, int Lout,
                       int ksize, int stride, int pad, int dil, const float* accumulate, float* split_ws, size_t split_ws_bytes,
                       void* stream) {
    if (!dy || !wt || !dx || bad_conv_shape(N, Cin, Lin, Cout, Lout, ksize, stride, pad, dil)) return SSECG_E_INVAL;
    if ((long long)N * Lin > 0x7fffffffLL) return SSECG_E_INVAL;
    if (!fits_descriptor((size_t)Cin * Cout * ksize, (size_t)N * Cout * Lout)) return SSECG_E_INVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool aligned = (((uintptr_t)wt & 15) == 0);
    ConvP p;
    p.src = dy; p.out = dx;
    p.N = N; p.M = Cin; p.Csrc = Cout; p.Lsrc = Lout;
    p.stride = stride; p.pad = pad; p.dil = dil;
    p.src_bytes = (unsigned)((size_t)N * Cout * Lout * 4);
    p.scale = nullptr; p.shift = nullptr; p.residual = accumulate; p.relu = 0; p.stats = nullptr;
    p.in_scale = nullptr; p.in_shift = nullptr;

    if (dgrad_phased(Cin, Cout, ksize, stride, wt)) {
        // two phase launches over the packed operand (see transpose_weight_kernel): no MFMA is spent on the taps
        // that cannot reach a given output parity.
        if (pad != 1 || dil != 1) return SSECG_E_INVAL;  // the packing assumes the ResNet's k3 s2 p1 geometry
        for (int phase = 0; phase < 2; ++phase) {
            const int Lq = phase == 0 ? (Lin + 1) / 2 : Lin / 2;  // outputs m = 2q + phase
            if (Lq == 0) continue;
            const int ks = phase == 0 ? 1 : 2;
            p.A = phase == 0 ? wt : wt + (size_t)Cin * Cout;
            p.Ktot = Cout * ks;
            p.a_bytes = (unsigned)((size_t)Cin * p.Ktot * 4);
            p.a_vec = 1;
            p.Ldst = Lq; p.Lrow = Lin; p.ostride = 2; p.ooff = phase;
            p.gmul = 1;
            if (phase == 0) { p.tapoff[0] = 0; p.tapoff[1] = 0; p.tapoff[2] = 0; }      // tap 1: dy[q]
            else { p.tapoff[0] = 1; p.tapoff[1] = 0; p.tapoff[2] = 0; }                 // tap 0: dy[q+1], tap 2: dy[q]
            const long long P = (long long)N * Lq;
            const TileCfg c = pick_cfg(Cin, P, true);
            p.P = (int)P; p.numPT = c.numPT; p.out_vec = 0;
            const int e = launch_igemm<1>(p, ks, c, st, split_ws, split_ws_bytes);
            if (e) return e;
        }
        return 0;
    }
    if (stride == 2 && ksize == 1 && pad == 0 && (accumulate == nullptr || accumulate == dx) && aligned && Cin > 32 && Cout % kBK == 0) {
        // 1x1 stride-2 (downsample branch): only even inputs receive gradient.  accumulate == dx (round 4): the gradient is ADDED in
        // place at the even positions of a tensor that already holds the main branch's gradient - no zero fill of dx, and the main
        // branch's two phase launches have nothing to accumulate (BasicBlockFn.backward)
        if (accumulate == nullptr) {
            hipError_t e = hipMemsetAsync(dx, 0, (size_t)N * Cin * Lin * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
        p.A = wt; p.Ktot = Cout; p.a_bytes = (unsigned)((size_t)Cin * Cout * 4); p.a_vec = 1;
        const int Lq = (Lin + 1) / 2;
        p.Ldst = Lq; p.Lrow = Lin; p.ostride = 2; p.ooff = 0; p.gmul = 1;
        p.tapoff[0] = p.tapoff[1] = p.tapoff[2] = 0;
        const long long P = (long long)N * Lq;
        const TileCfg c = pick_cfg(Cin, P, true);
        p.P = (int)P; p.numPT = c.numPT; p.out_vec = 0;
        return launch_igemm<1>(p, 1, c, st, split_ws, split_ws_bytes);
    }
    const long long P = (long long)N * Lin;
    const bool a_vec = ((Cout * ksize) % 4 == 0) && aligned;
    const TileCfg c = pick_cfg(Cin, P, stride == 1 && fast_ok(Cin, Cout, ksize, a_vec));
    p.A = wt; p.Ldst = Lin; p.Ktot = Cout * ksize;
    p.P = (int)P; p.numPT = c.numPT;
    p.a_vec = a_vec;
    p.a_bytes = (unsigned)((size_t)Cin * p.Ktot * 4);
    p.gmul = 1; p.tapoff[0] = pad; p.tapoff[1] = pad - dil; p.tapoff[2] = pad - 2 * dil;
    p.Lrow = Lin; p.ostride = 1; p.ooff = 0;
    p.out_vec = (Lin % 4 == 0) && (((uintptr_t)dx & 15) == 0) && (accumulate == nullptr || ((uintptr_t)accumulate & 15) == 0);
    return launch_igemm<1>(p, ksize, c, st, split_ws, split_ws_bytes);
}

size_t ssecg_conv1d_wgrad_workspace(int N, int Cin, int Lin, int Cout, int Lout, int ksize) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0 || ksize <= 0) return 0;
    (void)Lin;
    const WgradCfg c = pick_wgrad(Cout, Cin, ksize, (long long)N * Lout);
    return (size_t)c.Z * Cout * Cin * ksize * sizeof(float);
}

int ssecg_conv1d_wgrad(const float* dy, const float* x, float* dw, int N, int Cin, int Lin, int Cout, int Lout,
                       int ksize, int stride, int pad, int dil, void* workspace, size_t workspace_bytes,
                       const float* x_scale, const float* x_shift, void* stream) {
    if (!dy || !x || !dw || !workspace || bad_conv_shape(N, Cin, Lin, Cout, Lout, ksize, stride, pad, dil))
        return SSECG_E_INVAL;
    if ((x_scale == nullptr) != (x_shift == nullptr)) return SSECG_E_INVAL;
    if (!fits_descriptor((size_t)N * Cout * Lout, (size_t)N * Cin * Lin)) return SSECG_E_INVAL;
    const long long P = (long long)N * Lout;
    const WgradCfg c = pick_wgrad(Cout, Cin, ksize, P);
    const size_t need = (size_t)c.Z * Cout * Cin * ksize * sizeof(float);
    if (workspace_bytes < need) return SSECG_E_WORKSPACE;
    WgradP p;
    p.dy = dy; p.x = x; p.ws = (float*)workspace;
    p.dy_bytes = (unsigned)((size_t)N * Cout * Lout * 4); p.x_bytes = (unsigned)((size_t)N * Cin * Lin * 4);
    p.N = N; p.Cout = Cout; p.Ldy = Lout; p.Csrc = Cin; p.Lx = Lin; p.KS = ksize; p.J = Cin * ksize;
    p.stride = stride; p.pad = pad; p.dil = dil; p.P = P; p.chunk = c.chunk;
    hipStream_t st = (hipStream_t)stream;
    p.MT = c.MT; p.JT = c.JT; p.Z = c.Z;
    dim3 grid((unsigned)(c.MT * c.JT) * (unsigned)((c.Z + 7) / 8 * 8)), block(kThreads);
    const bool fast = (Cin % c.BJ == 0) && (Cout % 2 == 0) && Cout >= 2;
    if (x_scale != nullptr && !fast) return SSECG_E_INVAL;  // fused input BN: fast kernel only
    p.x_scale = x_scale; p.x_shift = x_shift;
#define SSECG_WG(BM_, BJ_)                                                                                     \
    do {                                                                                                       \
        if (fast) hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BJ_, 2, 2, true>), grid, block, 0, st, p);         \
        else hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BJ_, 2, 2, false>), grid, block, 0, st, p);             \
    } while (0)
    if (c.BM == 128 && c.BJ == 128) SSECG_WG(128, 128);
    else if (c.BM == 128 && c.BJ == 64) SSECG_WG(128, 64);
    else SSECG_WG(64, 64);
#undef SSECG_WG
    int e = (int)hipGetLastError();
    if (e) return e;
    const size_t total = (size_t)Cout * Cin * ksize;
    int blocks = (int)((total + 63) / 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, c.Z, Cout, Cin, ksize);
    return (int)hipGetLastError();
}

}  // extern "C"
