// HBM-bound kernels of the hot path: BatchNorm statistics / apply / backward,
// MaxPool1d, linear interpolation, dropout.  All tensors are (N, C, L) fp32,
// L fastest; a "row" is one (n, c) pair.  Streaming kernels move 16 B per lane
// when L % 4 == 0 (rows then never straddle a float4) and fall back to 4 B per
// lane for the odd lengths (125, 63).  Reductions are wave-shuffle (64 lanes) ->
// LDS -> one partial row per workgroup, summed later in a fixed order (fp64), so
// results are bitwise reproducible run to run.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "ssecg.h"
#include "amp_common.h"

namespace {

constexpr int kT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of two values; result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b) {
    __shared__ float red[2][kT / 64];
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = a; red[1][w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int i = 0; i < kT / 64; ++i) { sa += red[0][i]; sb += red[1][i]; }
        a = sa; b = sb;
    }
}

inline int grid_for(size_t work_items, int per_block, int cap = 4096) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------ BN statistics
// 32 channels x 32 part-lanes per workgroup; fp64 accumulation in a fixed order (bitwise reproducible).
// FINALIZE: also turn the sums into mean / invstd and update the running statistics (single-GPU BatchNorm);
// otherwise leave the sums for the caller to all-reduce (SyncBatchNorm) and optionally emit dgamma / dbeta.
template <bool FINALIZE>
__global__ __launch_bounds__(1024) void bn_reduce_partials_kernel(const float* partial, int parts, int C, double* sums,
                                                                 float* dgamma, float* dbeta, double count, float eps,
                                                                 float momentum, float* mean, float* invstd,
                                                                 float* rmean, float* rvar, const float* gamma,
                                                                 const float* beta, float* aff_scale, float* aff_shift) {
    __shared__ double sh[2][32][33];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        // eight rows requested before the first is added (the adds stay in row order: the sums are the same bit for bit); with one
        // load in flight a lane paid a global-memory round trip per row - 42 launches of 4.6-7.3 us per step (round 6)
        int part = pl;
        for (; part + 7 * 32 < parts; part += 8 * 32) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float2*>(partial)[(size_t)(part + 32 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
        }
        for (; part < parts; part += 32) {
            const float2 v = reinterpret_cast<const float2*>(partial)[(size_t)part * C + c];
            s += (double)v.x;
            q += (double)v.y;
        }
    }
    sh[0][pl][cl] = s;
    sh[1][pl][cl] = q;
    __syncthreads();
    if (pl == 0 && c < C) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) { ts += sh[0][i][cl]; tq += sh[1][i][cl]; }
        if (sums != nullptr) { sums[2 * c] = ts; sums[2 * c + 1] = tq; }
        if (dgamma != nullptr) { dbeta[c] = (float)ts; dgamma[c] = (float)tq; }
        if (FINALIZE) {
            const double m = ts / count;
            double var = tq / count - m * m;
            if (var < 0.0) var = 0.0;
            mean[c] = (float)m;
            invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (aff_scale != nullptr) {  // A = invstd*gamma, B = fma(-mean, A, beta): what bn_apply_fwd would use
                const float A = invstd[c] * gamma[c];
                aff_scale[c] = A;
                aff_shift[c] = fmaf(-mean[c], A, beta[c]);
            }
            if (rmean != nullptr) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                rmean[c] = momentum * (float)m + (1.f - momentum) * rmean[c];
                rvar[c] = momentum * (float)unb + (1.f - momentum) * rvar[c];
            }
        }
    }
}

__global__ void bn_finalize_kernel(const double* sums, int C, double count, float eps, float momentum,
                                   float* mean, float* invstd, float* rmean, float* rvar, const float* gamma,
                                   const float* beta, float* aff_scale, float* aff_shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[2 * c] / count;
    double var = sums[2 * c + 1] / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (aff_scale != nullptr) {
        const float A = invstd[c] * gamma[c];
        aff_scale[c] = A;
        aff_shift[c] = fmaf(-mean[c], A, beta[c]);
    }
    if (rmean != nullptr) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = momentum * (float)m + (1.f - momentum) * rmean[c];
        rvar[c] = momentum * (float)unb + (1.f - momentum) * rvar[c];
    }
}

__global__ void bn_fold_kernel(const float* g, const float* b, const float* rm, const float* rv, int C, float eps,
                               float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.0f / sqrtf(rv[c] + eps);
    const float a = g[c] * inv;
    scale[c] = a;
    shift[c] = b[c] - rm[c] * a;
}

// every BatchNorm of a model in one launch: table rows { gamma*, beta*, running_mean*, running_var*, scale*, shift*, C,
// eps (float bits) }; blockIdx.y = layer
__global__ void bn_fold_multi_kernel(const int64_t* __restrict__ table) {
    const int64_t* row = table + 8 * (size_t)blockIdx.y;
    const int C = (int)row[6];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* g = reinterpret_cast<const float*>(row[0]);
    const float* b = reinterpret_cast<const float*>(row[1]);
    const float* rm = reinterpret_cast<const float*>(row[2]);
    const float* rv = reinterpret_cast<const float*>(row[3]);
    const float eps = __int_as_float((int)row[7]);
    const float inv = 1.0f / sqrtf(rv[c] + eps);
    const float a = g[c] * inv;
    reinterpret_cast<float*>(row[4])[c] = a;
    reinterpret_cast<float*>(row[5])[c] = b[c] - rm[c] * a;
}

// ------------------------------------------------------------------ BN apply (forward)
// Streaming kernels walk the FLATTENED tensor 16 bytes per lane (total % 4 == 0 and 16-byte aligned bases, which
// holds for every activation of the network); when L % 4 != 0 a float4 may straddle two rows, i.e. two channels:
// element i of the vector belongs to channel c0 or its successor.
struct Chan2 { int c0, c1, split; };  // elements [0, split) -> c0, [split, 4) -> c1
__device__ __forceinline__ Chan2 chan_of(size_t e, int C, int L) {
    const size_t row = e / L;
    const int l = (int)(e - row * L);
    Chan2 r;
    r.c0 = (int)(row % C);
    r.c1 = r.c0 + 1 == C ? 0 : r.c0 + 1;
    r.split = L - l;  // >= 4 when the vector stays inside the row
    return r;
}

// RESBN: the residual is the RAW convolution output of the block's 1x1 downsample branch and ITS BatchNorm (rstat = { mean, invstd,
// gamma, beta } of that branch) is applied while it is read - the normalised identity tensor of a downsample block is never written
// (round 6: one store + one load of the block's largest tensor less; the same fp32 operations, bit for bit).
struct ResBN { const float *mean, *invstd, *gamma, *beta; };

template <bool VEC, bool RESBN>
__global__ void bn_apply_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int C, int L,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ res, int relu, uint8_t* __restrict__ mask_bits, ResBN rs) {
    constexpr int W = VEC ? 4 : 1;
    const size_t nvec = total / W;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        const size_t e = v * W;
        if (VEC) {
            const Chan2 ch = chan_of(e, C, L);
            // A = invstd*gamma, B = fma(-mean, A, beta): the backward kernels recompute exactly these for the ReLU mask
            const float a0 = invstd[ch.c0] * gamma[ch.c0], b0 = fmaf(-mean[ch.c0], a0, beta[ch.c0]);
            const float a1 = invstd[ch.c1] * gamma[ch.c1], b1 = fmaf(-mean[ch.c1], a1, beta[ch.c1]);
            const float4 xv = reinterpret_cast<const float4*>(x)[v];
            float4 o;
            o.x = fmaf(xv.x, a0, b0);  // fmaf: the backward kernels recompute this value for the ReLU mask
            o.y = ch.split > 1 ? fmaf(xv.y, a0, b0) : fmaf(xv.y, a1, b1);
            o.z = ch.split > 2 ? fmaf(xv.z, a0, b0) : fmaf(xv.z, a1, b1);
            o.w = ch.split > 3 ? fmaf(xv.w, a0, b0) : fmaf(xv.w, a1, b1);
            if (res != nullptr) {
                float4 r = reinterpret_cast<const float4*>(res)[v];
                if (RESBN) {
                    const float ra0 = rs.invstd[ch.c0] * rs.gamma[ch.c0], rb0 = fmaf(-rs.mean[ch.c0], ra0, rs.beta[ch.c0]);
                    const float ra1 = rs.invstd[ch.c1] * rs.gamma[ch.c1], rb1 = fmaf(-rs.mean[ch.c1], ra1, rs.beta[ch.c1]);
                    r.x = fmaf(r.x, ra0, rb0);
                    r.y = ch.split > 1 ? fmaf(r.y, ra0, rb0) : fmaf(r.y, ra1, rb1);
                    r.z = ch.split > 2 ? fmaf(r.z, ra0, rb0) : fmaf(r.z, ra1, rb1);
                    r.w = ch.split > 3 ? fmaf(r.w, ra0, rb0) : fmaf(r.w, ra1, rb1);
                }
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            reinterpret_cast<float4*>(y)[v] = o;
            if (mask_bits != nullptr) {
                // packed ReLU mask for the backward passes: bit (e & 7) of byte (e >> 3) = (y[e] > 0).  A lane holds four
                // elements = one nibble; the odd lane's nibble goes to its even neighbour (nvec is even: launcher), which
                // stores the byte - 1/32 of the bytes the backward passes would read from y otherwise.
                const unsigned nib = (o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u);
                const unsigned hi = (unsigned)__shfl_down((int)nib, 1, 64);
                if ((v & 1) == 0) mask_bits[v >> 1] = (uint8_t)(nib | (hi << 4));
            }
        } else {
            const int c = (int)((e / L) % C);
            const float a = invstd[c] * gamma[c];
            const float b = fmaf(-mean[c], a, beta[c]);
            float o = fmaf(x[e], a, b);
            if (res != nullptr) {
                float r = res[e];
                if (RESBN) {
                    const float ra = rs.invstd[c] * rs.gamma[c];
                    r = fmaf(r, ra, fmaf(-rs.mean[c], ra, rs.beta[c]));
                }
                o += r;
            }
            if (relu) o = fmaxf(o, 0.f);
            y[e] = o;
        }
    }
}

// ------------------------------------------------------------------ BN backward
// grid (C, S): workgroup (c, s) reduces channel c over samples [n0, n1).
// ReLU mask: from the saved activation y when given; RECOMP: recomputed as (x*A + B > 0) from the BN input (a BN
// whose output went straight through a ReLU with no residual) - one tensor less to read.
// PAIR (round 6): a SECOND BatchNorm whose incoming gradient is the same masked dy - the 1x1 downsample branch of a block beside its
// bn2 (resnet.py:64-70: out = relu(bn2(..) + bn_d(conv1x1(x))): both see dz = dout * [out > 0]) - is reduced (and, below, applied) in
// the same pass: dy and the mask are read once for the two.  Per thread the same loads, products and sums in the same order as the
// two single launches, so the partial rows are the same bit for bit.
// The backward sums and the apply pass are written with EXPLICIT roundings (fmaf / __fmul_rn / __fadd_rn): left to the compiler, the
// fused-multiply-add contraction of one and the same source expression came out differently in the PAIR and the single instantiations
// (partial rows 2e-6 apart, one output ulp in the apply pass) - pinned, the two are the same bit for bit.
__device__ __forceinline__ float xhat_of(float x, float mu, float is) { return __fmul_rn(__fsub_rn(x, mu), is); }
__device__ __forceinline__ float dot4(float d0, float t0, float d1, float t1, float d2, float t2, float d3, float t3) {
    return fmaf(d0, t0, fmaf(d1, t1, fmaf(d2, t2, __fmul_rn(d3, t3))));
}
__device__ __forceinline__ float sum4(float a, float b, float c, float d) { return __fadd_rn(__fadd_rn(a, b), __fadd_rn(c, d)); }
__device__ __forceinline__ float affine3(float A, float d, float B, float x, float D) { return fmaf(A, d, fmaf(B, x, D)); }

struct BnPair {
    const float* x;        // the second BatchNorm's input (the raw 1x1 output), or nullptr
    const float* mean;
    const float* invstd;
    const float* gamma;    // apply only
    const double* sums;    // apply only
    float* out;            // reduce: its partial rows [S][C][2]; apply: its dx
};

template <bool VEC, bool RECOMP, bool PAIR = false>
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                     const float* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, int N, int C, int L, float* partial,
                                     const uint8_t* __restrict__ mask_bits, BnPair pb = BnPair{}) {
    constexpr int W = VEC ? 4 : 1;
    const int c = blockIdx.x;
    const int S = gridDim.y;
    const int per = (N + S - 1) / S;
    const int n0 = blockIdx.y * per;
    const int n1 = min(N, n0 + per);
    const float mu = mean[c], is = invstd[c];
    float A = 0.f, B = 0.f;
    if (RECOMP) { A = is * gamma[c]; B = fmaf(-mu, A, beta[c]); }
    float s1 = 0.f, s2 = 0.f, s2b = 0.f;
    float mub = 0.f, isb = 0.f;
    if (PAIR) { mub = pb.mean[c]; isb = pb.invstd[c]; }
    const int LW = L / W;
    const int items = (n1 > n0) ? (n1 - n0) * LW : 0;
#pragma unroll 4
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int n = n0 + it / LW;
        const int lw = it - (it / LW) * LW;
        const size_t e = ((size_t)n * C + c) * L + (size_t)lw * W;
        if (VEC) {
            float4 d = *reinterpret_cast<const float4*>(dy + e);
            const float4 xv = *reinterpret_cast<const float4*>(x + e);
            float4 xb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PAIR) xb = *reinterpret_cast<const float4*>(pb.x + e);
            if (RECOMP) {
                d.x = fmaf(xv.x, A, B) > 0.f ? d.x : 0.f; d.y = fmaf(xv.y, A, B) > 0.f ? d.y : 0.f;
                d.z = fmaf(xv.z, A, B) > 0.f ? d.z : 0.f; d.w = fmaf(xv.w, A, B) > 0.f ? d.w : 0.f;
            } else if (mask_bits != nullptr) {   // packed mask (bn_apply_fwd_kernel): e % 4 == 0 -> one nibble
                const unsigned nib = (unsigned)mask_bits[e >> 3] >> (unsigned)(e & 4);
                d.x = (nib & 1u) ? d.x : 0.f; d.y = (nib & 2u) ? d.y : 0.f;
                d.z = (nib & 4u) ? d.z : 0.f; d.w = (nib & 8u) ? d.w : 0.f;
            } else if (y != nullptr) {
                const float4 yv = *reinterpret_cast<const float4*>(y + e);
                d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f;
                d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
            }
            s1 = __fadd_rn(s1, sum4(d.x, d.y, d.z, d.w));
            s2 = __fadd_rn(s2, dot4(d.x, xhat_of(xv.x, mu, is), d.y, xhat_of(xv.y, mu, is), d.z, xhat_of(xv.z, mu, is), d.w, xhat_of(xv.w, mu, is)));
            if (PAIR)
                s2b = __fadd_rn(s2b, dot4(d.x, xhat_of(xb.x, mub, isb), d.y, xhat_of(xb.y, mub, isb), d.z, xhat_of(xb.z, mub, isb), d.w,
                                          xhat_of(xb.w, mub, isb)));
        } else {
            float d = dy[e];
            const float xv = x[e];
            if (RECOMP) d = fmaf(xv, A, B) > 0.f ? d : 0.f;
            else if (mask_bits != nullptr) d = ((unsigned)mask_bits[e >> 3] >> (unsigned)(e & 7)) & 1u ? d : 0.f;
            else if (y != nullptr) d = y[e] > 0.f ? d : 0.f;
            s1 += d;
            s2 += d * ((xv - mu) * is);
        }
    }
    float s1b = s1;
    block_sum2(s1, s2);
    if (threadIdx.x == 0) {
        partial[((size_t)blockIdx.y * C + c) * 2] = s1;
        partial[((size_t)blockIdx.y * C + c) * 2 + 1] = s2;
    }
    if (PAIR) {
        __syncthreads();       // thread 0 has read the first pair of sums out of block_sum2's scratch
        block_sum2(s1b, s2b);
        if (threadIdx.x == 0) {
            pb.out[((size_t)blockIdx.y * C + c) * 2] = s1b;
            pb.out[((size_t)blockIdx.y * C + c) * 2 + 1] = s2b;
        }
    }
}

// Rows whose length is not a multiple of 4 (L = 250, 125, 63: layers 2-4 and the head at the benchmark's window length) start at
// arbitrary 4-byte alignment, so the flat float4 path above does not apply - and the 4-byte path (one element per lane, an integer
// division per element) read layer2 at 2.2 TB/s, layer3 at 1.1, layer4 at 0.6: 16 of the 20 launches of a FixMatch step, 0.99 of
// their 1.21 ms (round 6, profiles/r05_bench_kernel_stats.md re-read).  Here a row is read with 16-byte RAW BUFFER loads - a dwordx4
// buffer load needs 4-byte alignment only and is range-checked per dword - by LWP = 2^k >= ceil(L / 4) lanes,
// 256 / LWP rows per workgroup iteration, no division in the loop; the last vector of a row runs into the next row (or past the
// tensor: zeros) and is masked by count.  MASK: 0 none / recomputed, 1 the saved activation, 2 the packed bits of bn_apply_fwd.
typedef unsigned u32x4e __attribute__((__vector_size__(16)));

template <bool RECOMP, int MASK, bool PAIR = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_rows_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int N, int C, int L, float* partial,
                                                                 const uint8_t* __restrict__ mask_bits, int lwp_shift, unsigned nbytes,
                                                                 BnPair pb = BnPair{}) {
    const int c = blockIdx.x;
    const int S = gridDim.y;
    const int per = (N + S - 1) / S;
    const int n0 = blockIdx.y * per;
    const int n1 = min(N, n0 + per);
    const float mu = mean[c], is = invstd[c];
    float A = 0.f, B = 0.f;
    if (RECOMP) { A = is * gamma[c]; B = fmaf(-mu, A, beta[c]); }
    const int LW = (L + 3) >> 2;
    const int lw = threadIdx.x & ((1 << lwp_shift) - 1);
    const int rsub = threadIdx.x >> lwp_shift, rpi = 256 >> lwp_shift;
    const bool lane_ok = lw < LW;
    const int cnt = lane_ok ? min(4, L - 4 * lw) : 0;          // valid elements of this lane's vector
    const auto dyR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, (int)nbytes, 0x00020000);
    const auto xR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)nbytes, 0x00020000);
    const auto yR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MASK == 1 ? y : x), 0, (int)nbytes, 0x00020000);
    const auto mR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(MASK == 2 ? mask_bits : reinterpret_cast<const uint8_t*>(x)), 0,
                                                      (int)((nbytes / 4 + 7) / 8), 0x00020000);
    const auto xbR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PAIR ? pb.x : x), 0, (int)nbytes, 0x00020000);
    float s1 = 0.f, s2 = 0.f, s2b = 0.f;
    float mub = 0.f, isb = 0.f;
    if (PAIR) { mub = pb.mean[c]; isb = pb.invstd[c]; }
#pragma unroll 4
    for (int n = n0 + rsub; n < n1; n += rpi) {
        const unsigned e0 = ((unsigned)n * (unsigned)C + (unsigned)c) * (unsigned)L + 4u * (unsigned)lw;
        const unsigned off = lane_ok ? e0 * 4u : 0x80000000u;      // beyond num_records: the loads return zeros
        const u32x4e dv = __builtin_amdgcn_raw_buffer_load_b128(dyR, off, 0, 0);
        const u32x4e xv4 = __builtin_amdgcn_raw_buffer_load_b128(xR, off, 0, 0);
        float xb[4] = {0.f, 0.f, 0.f, 0.f};
        if (PAIR) {
            const u32x4e bv = __builtin_amdgcn_raw_buffer_load_b128(xbR, off, 0, 0);
            const unsigned b0_ = bv[0], b1_ = bv[1], b2_ = bv[2], b3_ = bv[3];
            xb[0] = __uint_as_float(b0_); xb[1] = __uint_as_float(b1_); xb[2] = __uint_as_float(b2_); xb[3] = __uint_as_float(b3_);
        }
        const unsigned d0 = dv[0], d1 = dv[1], d2 = dv[2], d3 = dv[3];       // (no bit_cast on vector elements: clang reads element 0)
        const unsigned x0 = xv4[0], x1 = xv4[1], x2 = xv4[2], x3 = xv4[3];
        float d[4] = {__uint_as_float(d0), __uint_as_float(d1), __uint_as_float(d2), __uint_as_float(d3)};
        const float xv[4] = {__uint_as_float(x0), __uint_as_float(x1), __uint_as_float(x2), __uint_as_float(x3)};
        unsigned keep = (1u << cnt) - 1u;                           // bit j: element j belongs to this row
        if (RECOMP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) keep &= ~((fmaf(xv[j], A, B) > 0.f ? 0u : 1u) << j);
        } else if (MASK == 2) {   // bits e0 .. e0+3 of the packed mask: byte e0 >> 3 and, when they straddle it, the next one
            const unsigned b0 = __builtin_amdgcn_raw_buffer_load_b8(mR, lane_ok ? (e0 >> 3) : 0x80000000u, 0, 0);
            const unsigned b1 = __builtin_amdgcn_raw_buffer_load_b8(mR, lane_ok ? (e0 >> 3) + 1u : 0x80000000u, 0, 0);
            keep &= ((b0 | (b1 << 8)) >> (e0 & 7u)) & 0xfu;
        } else if (MASK == 1) {
            const u32x4e yv = __builtin_amdgcn_raw_buffer_load_b128(yR, off, 0, 0);
            const unsigned y0 = yv[0], y1 = yv[1], y2 = yv[2], y3 = yv[3];
            const float yy[4] = {__uint_as_float(y0), __uint_as_float(y1), __uint_as_float(y2), __uint_as_float(y3)};
#pragma unroll
            for (int j = 0; j < 4; ++j) keep &= ~((yy[j] > 0.f ? 0u : 1u) << j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = ((keep >> j) & 1u) ? d[j] : 0.f;
        s1 = __fadd_rn(s1, sum4(d[0], d[1], d[2], d[3]));
        s2 = __fadd_rn(s2, dot4(d[0], xhat_of(xv[0], mu, is), d[1], xhat_of(xv[1], mu, is), d[2], xhat_of(xv[2], mu, is), d[3], xhat_of(xv[3], mu, is)));
        if (PAIR)
            s2b = __fadd_rn(s2b, dot4(d[0], xhat_of(xb[0], mub, isb), d[1], xhat_of(xb[1], mub, isb), d[2], xhat_of(xb[2], mub, isb), d[3],
                                      xhat_of(xb[3], mub, isb)));
    }
    float s1b = s1;
    block_sum2(s1, s2);
    if (threadIdx.x == 0) {
        partial[((size_t)blockIdx.y * C + c) * 2] = s1;
        partial[((size_t)blockIdx.y * C + c) * 2 + 1] = s2;
    }
    if (PAIR) {
        __syncthreads();
        block_sum2(s1b, s2b);
        if (threadIdx.x == 0) {
            pb.out[((size_t)blockIdx.y * C + c) * 2] = s1b;
            pb.out[((size_t)blockIdx.y * C + c) * 2 + 1] = s2b;
        }
    }
}

template <bool VEC, bool RECOMP, bool PAIR = false>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                    const float* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta,
                                    const double* __restrict__ sums, double inv_count, size_t total, int C, int L,
                                    float* __restrict__ dx, float* __restrict__ dz_out,
                                    const uint8_t* __restrict__ mask_bits, BnPair pb = BnPair{}) {
    constexpr int W = VEC ? 4 : 1;
    const size_t nvec = total / W;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        const size_t e = v * W;
        // dx = k1*(dz - m1 - xhat*m2) = A*dz + B*x + D with A = k1, B = -k1*is*m2, D = k1*(mu*is*m2 - m1)
        if (VEC) {
            const Chan2 ch = chan_of(e, C, L);
            float A[2], Bc[2], D[2], FA[2], FB[2], A2[2], B2[2], D2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = k ? ch.c1 : ch.c0;
                const float is = invstd[c], mu = mean[c];
                const float k1 = __fmul_rn(gamma[c], is);
                const float m1 = (float)(sums[2 * c] * inv_count);
                const float m2 = (float)(sums[2 * c + 1] * inv_count);
                A[k] = k1; Bc[k] = __fmul_rn(__fmul_rn(-k1, is), m2); D[k] = __fmul_rn(k1, fmaf(__fmul_rn(mu, is), m2, -m1));
                if (RECOMP) { FA[k] = is * gamma[c]; FB[k] = fmaf(-mu, FA[k], beta[c]); }
                if (PAIR) {
                    const float isb = pb.invstd[c], mub = pb.mean[c];
                    const float k1b = __fmul_rn(pb.gamma[c], isb);
                    const float m1b = (float)(pb.sums[2 * c] * inv_count);
                    const float m2b = (float)(pb.sums[2 * c + 1] * inv_count);
                    A2[k] = k1b; B2[k] = __fmul_rn(__fmul_rn(-k1b, isb), m2b); D2[k] = __fmul_rn(k1b, fmaf(__fmul_rn(mub, isb), m2b, -m1b));
                }
            }
            float4 d = reinterpret_cast<const float4*>(dy)[v];
            const float4 xv = reinterpret_cast<const float4*>(x)[v];
            if (RECOMP) {
                const int q1 = ch.split > 1 ? 0 : 1, q2 = ch.split > 2 ? 0 : 1, q3 = ch.split > 3 ? 0 : 1;
                d.x = fmaf(xv.x, FA[0], FB[0]) > 0.f ? d.x : 0.f;
                d.y = fmaf(xv.y, FA[q1], FB[q1]) > 0.f ? d.y : 0.f;
                d.z = fmaf(xv.z, FA[q2], FB[q2]) > 0.f ? d.z : 0.f;
                d.w = fmaf(xv.w, FA[q3], FB[q3]) > 0.f ? d.w : 0.f;
            } else if (mask_bits != nullptr) {
                const unsigned nib = (unsigned)mask_bits[v >> 1] >> (unsigned)((v & 1) * 4);
                d.x = (nib & 1u) ? d.x : 0.f; d.y = (nib & 2u) ? d.y : 0.f;
                d.z = (nib & 4u) ? d.z : 0.f; d.w = (nib & 8u) ? d.w : 0.f;
            } else if (y != nullptr) {
                const float4 yv = reinterpret_cast<const float4*>(y)[v];
                d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f;
                d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
            }
            if (dz_out != nullptr) reinterpret_cast<float4*>(dz_out)[v] = d;
            const int k1i = ch.split > 1 ? 0 : 1, k2i = ch.split > 2 ? 0 : 1, k3i = ch.split > 3 ? 0 : 1;
            float4 o;
            o.x = affine3(A[0], d.x, Bc[0], xv.x, D[0]);
            o.y = affine3(A[k1i], d.y, Bc[k1i], xv.y, D[k1i]);
            o.z = affine3(A[k2i], d.z, Bc[k2i], xv.z, D[k2i]);
            o.w = affine3(A[k3i], d.w, Bc[k3i], xv.w, D[k3i]);
            reinterpret_cast<float4*>(dx)[v] = o;
            if (PAIR) {
                const float4 xb = reinterpret_cast<const float4*>(pb.x)[v];
                float4 ob;
                ob.x = affine3(A2[0], d.x, B2[0], xb.x, D2[0]);
                ob.y = affine3(A2[k1i], d.y, B2[k1i], xb.y, D2[k1i]);
                ob.z = affine3(A2[k2i], d.z, B2[k2i], xb.z, D2[k2i]);
                ob.w = affine3(A2[k3i], d.w, B2[k3i], xb.w, D2[k3i]);
                reinterpret_cast<float4*>(pb.out)[v] = ob;
            }
        } else {
            const int c = (int)((e / L) % C);
            const float is = invstd[c], mu = mean[c];
            const float k1 = gamma[c] * is;
            const float m1 = (float)(sums[2 * c] * inv_count);
            const float m2 = (float)(sums[2 * c + 1] * inv_count);
            float d = dy[e];
            if (RECOMP) { const float fa = is * gamma[c]; d = fmaf(x[e], fa, fmaf(-mu, fa, beta[c])) > 0.f ? d : 0.f; }
            else if (mask_bits != nullptr) d = ((unsigned)mask_bits[e >> 3] >> (unsigned)(e & 7)) & 1u ? d : 0.f;
            else if (y != nullptr) d = y[e] > 0.f ? d : 0.f;
            if (dz_out != nullptr) dz_out[e] = d;
            dx[e] = k1 * (d - m1 - (x[e] - mu) * is * m2);
        }
    }
}

__global__ void bn_param_grads_kernel(const double* sums, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)sums[2 * c];
    dgamma[c] = (float)sums[2 * c + 1];
}

// out[c] = sum over (n, l) of x[n][c][l] (the classifier's bias gradient: C = 4 channels, 1 MB).  Round 1-3: ONE workgroup per channel
// (4 workgroups on the whole chip: 33 us).  Now up to kChS sample slabs per channel write their sums into CALLER-OWNED scratch and a
// second tiny launch adds them in slab order - a fixed order, so the result is reproducible.  (Round 4 kept the slab sums and an
// arrival ticket in device globals: two streams at once would have shared them, and a launch that died left the ticket armed -
// ADVICE r4.  No global state now; without scratch the kernel runs one workgroup per channel.)
constexpr int kChS = 32;

__global__ void channel_sum_kernel(const float* __restrict__ x, int N, int C, int L, float* out) {
    const int c = blockIdx.x, sl = blockIdx.y, S = gridDim.y;
    const int per = (N + S - 1) / S;
    const int n0 = sl * per, n1 = min(N, n0 + per);
    // four independent partial sums per thread (fixed assignment -> reproducible): the loop is latency-bound otherwise
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, z = 0.f;
    const int items = n1 > n0 ? (n1 - n0) * L : 0;
    auto at = [&](int it) { const int n = n0 + it / L; return x[((size_t)n * C + c) * L + (it - (it / L) * L)]; };
    int it = threadIdx.x;
    for (; it + 3 * (int)blockDim.x < items; it += 4 * blockDim.x) {
        s0 += at(it); s1 += at(it + blockDim.x); s2 += at(it + 2 * blockDim.x); s3 += at(it + 3 * blockDim.x);
    }
    for (; it < items; it += blockDim.x) s0 += at(it);
    float s = (s0 + s1) + (s2 + s3);
    block_sum2(s, z);
    if (threadIdx.x == 0) out[c * S + sl] = s;      // S == 1: the result itself; else slab sum sl of channel c
}

__global__ void channel_sum_finish_kernel(const float* __restrict__ part, int C, int S, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float t = 0.f;
    for (int i = 0; i < S; ++i) t += part[c * S + i];
    out[c] = t;
}

// ------------------------------------------------------------------ MaxPool1d
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int Lin,
                                   int Lout, int k, int s, int pad) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / Lout;
        const int o = (int)(e - row * Lout);
        const float* xr = x + row * Lin;
        const int st = o * s - pad;
        float m = -INFINITY;
        for (int t = 0; t < k; ++t) {
            const int i = st + t;
            if ((unsigned)i < (unsigned)Lin) {
                const float v = xr[i];
                if (v > m || v != v) m = v;
            }
        }
        y[e] = m;
    }
}

// dx[i] = sum over windows containing i whose FIRST maximum is at i of dy[w]
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                   float* __restrict__ dx, size_t total, int Lin, int Lout, int k, int s, int pad) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / Lin;
        const int i = (int)(e - row * Lin);
        const float* xr = x + row * Lin;
        const float* dr = dy + row * Lout;
        int wlo = i + pad - k + 1;
        wlo = wlo <= 0 ? 0 : (wlo + s - 1) / s;
        int whi = (i + pad) / s;
        if (whi > Lout - 1) whi = Lout - 1;
        float g = 0.f;
        for (int w = wlo; w <= whi; ++w) {
            const int st = w * s - pad;
            float m = -INFINITY;
            int am = -1;
            for (int t = 0; t < k; ++t) {
                const int q = st + t;
                if ((unsigned)q < (unsigned)Lin) {
                    const float v = xr[q];
                    if (v > m || v != v) { m = v; am = q; }
                }
            }
            if (am == i) g += dr[w];
        }
        dx[e] = g;
    }
}

// ------------------------------------------------------------------ stem: BN + ReLU + MaxPool fused
// The stem's activation a = relu(bn(c)) (N,64,1000) is only ever consumed by the max-pool; materialising it costs a
// 262 MB write + read forward and three more passes backward.  Forward pools straight from the conv output c;
// backward recomputes a in registers (5 neighbours per element) to route the pooled gradient (first maximum wins, as
// ATen) and to apply the ReLU mask, feeding the BatchNorm backward sums / apply directly.
struct AffineCh { float A, B; };
__device__ __forceinline__ AffineCh affine_of(int c, const float* mean, const float* invstd, const float* g, const float* b) {
    AffineCh r;
    if (mean != nullptr) { r.A = invstd[c] * g[c]; r.B = b[c] - mean[c] * r.A; }
    else { r.A = g[c]; r.B = b[c]; }  // eval mode: g = folded scale, b = folded shift
    return r;
}

__global__ void bn_relu_maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int C,
                                           int Lin, int Lout, int k, int s, int pad, const float* mean,
                                           const float* invstd, const float* g, const float* b) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / Lout;
        const int o = (int)(e - row * Lout);
        const AffineCh af = affine_of((int)(row % C), mean, invstd, g, b);
        const float* xr = x + row * Lin;
        const int st = o * s - pad;
        float m = -INFINITY;
        for (int t = 0; t < k; ++t) {
            const int i = st + t;
            if ((unsigned)i < (unsigned)Lin) {
                const float v = fmaxf(xr[i] * af.A + af.B, 0.f);
                if (v > m || v != v) m = v;
            }
        }
        y[e] = m;
    }
}

// k = 3, stride 2, pad 1, Lin % 4 == 0 (the stem: 1000 -> 500): one thread = one aligned input quad x[4q..4q+3] (16-byte
// load) + the element before it -> two pooled outputs (8-byte store).  The generic kernel above reads every input 1.5 times
// with stride-2 dword loads and divides 64-bit indices per output: 2.4 TB/s.
__global__ void stem_pool_fwd_quad_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned nquads, int C, int Lin,
                                          const float* mean, const float* invstd, const float* g, const float* b) {
    const unsigned LQ = (unsigned)Lin >> 2;
    for (unsigned v = blockIdx.x * blockDim.x + threadIdx.x; v < nquads; v += gridDim.x * blockDim.x) {
        const unsigned row = v / LQ, q = v - row * LQ;
        const AffineCh af = affine_of((int)(row % (unsigned)C), mean, invstd, g, b);
        const float* xr = x + (size_t)row * Lin;
        const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * q);
        const float a0 = fmaxf(xv.x * af.A + af.B, 0.f), a1 = fmaxf(xv.y * af.A + af.B, 0.f);
        const float a2 = fmaxf(xv.z * af.A + af.B, 0.f), a3 = fmaxf(xv.w * af.A + af.B, 0.f);
        // first-maximum-wins / NaN-propagating comparison chain of the generic kernel, window order (i-1, i, i+1)
        float m0 = -INFINITY;
        if (q > 0) { const float p = fmaxf(xr[4 * q - 1] * af.A + af.B, 0.f); if (p > m0 || p != p) m0 = p; }
        if (a0 > m0 || a0 != a0) m0 = a0;
        if (a1 > m0 || a1 != a1) m0 = a1;
        float m1 = -INFINITY;
        if (a1 > m1 || a1 != a1) m1 = a1;
        if (a2 > m1 || a2 != a2) m1 = a2;
        if (a3 > m1 || a3 != a3) m1 = a3;
        *reinterpret_cast<float2*>(y + (size_t)row * (Lin >> 1) + 2 * q) = make_float2(m0, m1);
    }
}

// The activation the pooled gradient is routed on.  LP (use_amp with the 16-bit stem): the reference under autocast pools the
// bf16 output of BatchNorm, so ties between ROUNDED neighbours go to the first of them; route on the rounded value.
template <bool LP>
__device__ __forceinline__ float pool_act(float x, AffineCh af) {
    const float v = fmaxf(x * af.A + af.B, 0.f);
    return LP ? ssecg_amp::bf_lo(ssecg_amp::pack2(v, 0.f)) : v;
}

// gradient reaching a[i] = relu(bn(x[i])) from the pooled gradient, times the ReLU mask
template <bool LP>
__device__ __forceinline__ float pooled_dz(const float* __restrict__ xr, const float* __restrict__ dr, int i, int Lin,
                                           int Lout, int k, int s, int pad, AffineCh af) {
    const float ai = pool_act<LP>(xr[i], af);
    if (!(ai > 0.f)) return 0.f;
    int wlo = i + pad - k + 1;
    wlo = wlo <= 0 ? 0 : (wlo + s - 1) / s;
    int whi = (i + pad) / s;
    if (whi > Lout - 1) whi = Lout - 1;
    float g = 0.f;
    for (int w = wlo; w <= whi; ++w) {
        const int st = w * s - pad;
        float m = -INFINITY;
        int am = -1;
        for (int t = 0; t < k; ++t) {
            const int q = st + t;
            if ((unsigned)q < (unsigned)Lin) {
                const float v = pool_act<LP>(xr[q], af);
                if (v > m || v != v) { m = v; am = q; }
            }
        }
        if (am == i) g += dr[w];
    }
    return g;
}

// k=3, s=2, pad=1 and Lin % 4 == 0 (the ResNet stem): one thread produces dz for 4 consecutive elements 4q..4q+3
// from 7 recomputed activations and 3 pooled gradients (windows 2q, 2q+1, 2q+2), instead of 2 windows per element.
// The stem's conv output c under use_amp is bf16-valued (csrc/stem.hip rounds it): X16 stores it as bf16 (planar, 2 bytes) - half
// the bytes of every pass over it; values and results are the fp32-container path's, bit for bit.
template <bool X16>
__device__ __forceinline__ float4 c_quad(const void* __restrict__ x, size_t e) {   // elements e .. e+3 (e % 4 == 0, row starts 16-byte aligned)
    if (X16) {
        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(x) + e);
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + e);
}
template <bool X16>
__device__ __forceinline__ float c_one(const void* __restrict__ x, size_t e) {
    if (X16) return __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(x)[e] << 16);
    return reinterpret_cast<const float*>(x)[e];
}

template <bool LP, bool X16>
__device__ __forceinline__ float4 pooled_dz_quad_vals(const void* __restrict__ x, size_t r0, float d0, float d1, float d2, int q, int Lin,
                                                      AffineCh af) {   // r0: the row's first element; d0..d2: pooled gradients of windows 2q, 2q+1, 2q+2 (0 beyond the row)
    const int i0 = 4 * q;
    const float4 xv = c_quad<X16>(x, r0 + i0);
    float a[7];  // activations at i0-1 .. i0+5 (-inf outside the row: never the maximum)
    a[0] = i0 > 0 ? pool_act<LP>(c_one<X16>(x, r0 + i0 - 1), af) : -INFINITY;
    a[1] = pool_act<LP>(xv.x, af);
    a[2] = pool_act<LP>(xv.y, af);
    a[3] = pool_act<LP>(xv.z, af);
    a[4] = pool_act<LP>(xv.w, af);
    a[5] = (i0 + 4) < Lin ? pool_act<LP>(c_one<X16>(x, r0 + i0 + 4), af) : -INFINITY;
    a[6] = (i0 + 5) < Lin ? pool_act<LP>(c_one<X16>(x, r0 + i0 + 5), af) : -INFINITY;
    // first maximum wins (strict >), scanning left to right; index = position in the 3-window
    auto argmax3 = [](float l, float c, float r) { int am = 0; float m = l; if (c > m) { m = c; am = 1; } if (r > m) am = 2; return am; };
    const int am0 = argmax3(a[0], a[1], a[2]);  // window 2q   over i0-1, i0,   i0+1
    const int am1 = argmax3(a[2], a[3], a[4]);  // window 2q+1 over i0+1, i0+2, i0+3
    const int am2 = argmax3(a[4], a[5], a[6]);  // window 2q+2 over i0+3, i0+4, i0+5
    float4 dz;
    dz.x = (am0 == 1 && a[1] > 0.f) ? d0 : 0.f;
    dz.y = a[2] > 0.f ? ((am0 == 2 ? d0 : 0.f) + (am1 == 0 ? d1 : 0.f)) : 0.f;
    dz.z = (am1 == 1 && a[3] > 0.f) ? d1 : 0.f;
    dz.w = a[4] > 0.f ? ((am1 == 2 ? d1 : 0.f) + (am2 == 0 ? d2 : 0.f)) : 0.f;
    return dz;
}

template <bool LP, bool X16>
__device__ __forceinline__ float4 pooled_dz_quad(const void* __restrict__ x, size_t r0, const float* __restrict__ dr, int q, int Lin,
                                                 int Lout, AffineCh af) {
    const int w0 = 2 * q;
    const float d0 = dr[w0];
    const float d1 = (w0 + 1) < Lout ? dr[w0 + 1] : 0.f;
    const float d2 = (w0 + 2) < Lout ? dr[w0 + 2] : 0.f;
    return pooled_dz_quad_vals<LP, X16>(x, r0, d0, d1, d2, q, Lin, af);
}

template <bool LP, bool X16>
__global__ void stem_pool_bwd_reduce_quad_kernel(const float* __restrict__ dy, const void* __restrict__ x,
                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                 const float* __restrict__ g, const float* __restrict__ b, int N, int C,
                                                 int Lin, int Lout, float* partial) {
    const int c = blockIdx.x;
    const int S = gridDim.y;
    const int per = (N + S - 1) / S;
    const int n0 = blockIdx.y * per;
    const int n1 = min(N, n0 + per);
    const AffineCh af = affine_of(c, mean, invstd, g, b);
    const float mu = mean[c], is = invstd[c];
    float s1 = 0.f, s2 = 0.f;
    const int LQ = Lin / 4;
    const int items = (n1 > n0) ? (n1 - n0) * LQ : 0;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int n = n0 + it / LQ;
        const int q = it - (it / LQ) * LQ;
        const size_t row = (size_t)n * C + c;
        const float4 d = pooled_dz_quad<LP, X16>(x, row * Lin, dy + row * Lout, q, Lin, Lout, af);
        const float4 xv = c_quad<X16>(x, row * Lin + 4 * q);
        s1 += (d.x + d.y) + (d.z + d.w);
        s2 += d.x * ((xv.x - mu) * is) + d.y * ((xv.y - mu) * is) + d.z * ((xv.z - mu) * is) + d.w * ((xv.w - mu) * is);
    }
    block_sum2(s1, s2);
    if (threadIdx.x == 0) {
        partial[((size_t)blockIdx.y * C + c) * 2] = s1;
        partial[((size_t)blockIdx.y * C + c) * 2 + 1] = s2;
    }
}

// X16: dx (the gradient of the conv output, which autocast's 16-bit BatchNorm backward stores in 16 bit and the stem's weight gradient
// rounds while staging anyway) is written as bf16 too
template <bool LP, bool X16>
__global__ void stem_pool_bwd_apply_quad_kernel(const float* __restrict__ dy, const void* __restrict__ x,
                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                const float* __restrict__ g, const float* __restrict__ b,
                                                const double* __restrict__ sums, double inv_count, size_t nquads, int C,
                                                int Lin, int Lout, void* __restrict__ dx) {
    const int LQ = Lin / 4;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nquads; v += (size_t)gridDim.x * blockDim.x) {
        const size_t row = v / LQ;
        const int q = (int)(v - row * LQ);
        const int c = (int)(row % C);
        const AffineCh af = affine_of(c, mean, invstd, g, b);
        const float4 d = pooled_dz_quad<LP, X16>(x, row * Lin, dy + row * Lout, q, Lin, Lout, af);
        const float4 xv = c_quad<X16>(x, row * Lin + 4 * q);
        const float is = invstd[c], mu = mean[c];
        const float k1 = g[c] * is;
        const float m1 = (float)(sums[2 * c] * inv_count);
        const float m2 = (float)(sums[2 * c + 1] * inv_count);
        float4 o;
        o.x = k1 * (d.x - m1 - (xv.x - mu) * is * m2);
        o.y = k1 * (d.y - m1 - (xv.y - mu) * is * m2);
        o.z = k1 * (d.z - m1 - (xv.z - mu) * is * m2);
        o.w = k1 * (d.w - m1 - (xv.w - mu) * is * m2);
        if (X16) reinterpret_cast<uint2*>(dx)[v] = make_uint2(ssecg_amp::pack2(o.x, o.y), ssecg_amp::pack2(o.z, o.w));
        else reinterpret_cast<float4*>(dx)[v] = o;
    }
}

// ---- The stem's BN + ReLU + MaxPool(3, 2, 1) writing the POOLED activation in the blocked bf16 layout of the reduced-precision
// student pass (N, C/8, Lp, 8): one thread = 8 channels x one input quad, so a pooled position of 8 channels is one 16-byte vector.
// Same arithmetic, comparison chains and rounding as stem_pool_fwd_quad_kernel + cvt_planar_to_blocked - bit-identical - without
// the fp32 pooled tensor (-131 MB written and read per step): 85 us against 72 + 40.  (The backward counterparts - pooled gradient
// read in the blocked layout, 8 channels per thread - were built, bit-identical, and measured NO faster than cvt_blocked_to_planar +
// the fp32 kernels (145 + 144 us against 35 + 107 + 146): 146 registers / occupancy 3 for the reduction; not kept.)
template <bool X16>
__global__ void stem_pool_fwd_b16_kernel(const void* __restrict__ x, u32x4* __restrict__ yb, unsigned nitems, int C, int Lin,
                                         const float* mean, const float* invstd, const float* g, const float* b) {
    const unsigned LQ = (unsigned)Lin >> 2, CB = (unsigned)C >> 3, Lp = (unsigned)Lin >> 1;
    for (unsigned v = blockIdx.x * blockDim.x + threadIdx.x; v < nitems; v += gridDim.x * blockDim.x) {
        const unsigned rowb = v / LQ, q = v - rowb * LQ;        // rowb = n * CB + cb
        const unsigned n = rowb / CB, cb = rowb - n * CB;
        float m0[8], m1[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const int c = (int)(8 * cb) + ch;
            const AffineCh af = affine_of(c, mean, invstd, g, b);
            const size_t r0 = ((size_t)n * C + c) * Lin;
            const float4 xv = c_quad<X16>(x, r0 + 4 * q);
            const float a0 = fmaxf(xv.x * af.A + af.B, 0.f), a1 = fmaxf(xv.y * af.A + af.B, 0.f);
            const float a2 = fmaxf(xv.z * af.A + af.B, 0.f), a3 = fmaxf(xv.w * af.A + af.B, 0.f);
            float t0 = -INFINITY;
            if (q > 0) { const float p = fmaxf(c_one<X16>(x, r0 + 4 * q - 1) * af.A + af.B, 0.f); if (p > t0 || p != p) t0 = p; }
            if (a0 > t0 || a0 != a0) t0 = a0;
            if (a1 > t0 || a1 != a1) t0 = a1;
            float t1 = -INFINITY;
            if (a1 > t1 || a1 != a1) t1 = a1;
            if (a2 > t1 || a2 != a2) t1 = a2;
            if (a3 > t1 || a3 != a3) t1 = a3;
            m0[ch] = t0; m1[ch] = t1;
        }
        u32x4 o0, o1;
        o0.x = ssecg_amp::pack2(m0[0], m0[1]); o0.y = ssecg_amp::pack2(m0[2], m0[3]); o0.z = ssecg_amp::pack2(m0[4], m0[5]); o0.w = ssecg_amp::pack2(m0[6], m0[7]);
        o1.x = ssecg_amp::pack2(m1[0], m1[1]); o1.y = ssecg_amp::pack2(m1[2], m1[3]); o1.z = ssecg_amp::pack2(m1[4], m1[5]); o1.w = ssecg_amp::pack2(m1[6], m1[7]);
        u32x4* dst = yb + (size_t)rowb * Lp + 2 * q;
        dst[0] = o0; dst[1] = o1;
    }
}

// grid (C, S) as bn_bwd_reduce_kernel
template <bool LP>
__global__ void bn_relu_maxpool_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  const float* __restrict__ g, const float* __restrict__ b, int N, int C,
                                                  int Lin, int Lout, int k, int s, int pad, float* partial) {
    const int c = blockIdx.x;
    const int S = gridDim.y;
    const int per = (N + S - 1) / S;
    const int n0 = blockIdx.y * per;
    const int n1 = min(N, n0 + per);
    const AffineCh af = affine_of(c, mean, invstd, g, b);
    const float mu = mean[c], is = invstd[c];
    float s1 = 0.f, s2 = 0.f;
    const int items = (n1 > n0) ? (n1 - n0) * Lin : 0;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int n = n0 + it / Lin;
        const int i = it - (it / Lin) * Lin;
        const size_t row = (size_t)n * C + c;
        const float* xr = x + row * Lin;
        const float d = pooled_dz<LP>(xr, dy + row * Lout, i, Lin, Lout, k, s, pad, af);
        s1 += d;
        s2 += d * ((xr[i] - mu) * is);
    }
    block_sum2(s1, s2);
    if (threadIdx.x == 0) {
        partial[((size_t)blockIdx.y * C + c) * 2] = s1;
        partial[((size_t)blockIdx.y * C + c) * 2 + 1] = s2;
    }
}

template <bool LP>
__global__ void bn_relu_maxpool_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                 const float* __restrict__ g, const float* __restrict__ b,
                                                 const double* __restrict__ sums, double inv_count, size_t total, int C,
                                                 int Lin, int Lout, int k, int s, int pad, float* __restrict__ dx) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / Lin;
        const int i = (int)(e - row * Lin);
        const int c = (int)(row % C);
        const AffineCh af = affine_of(c, mean, invstd, g, b);
        const float* xr = x + row * Lin;
        const float d = pooled_dz<LP>(xr, dy + row * Lout, i, Lin, Lout, k, s, pad, af);
        const float is = invstd[c], mu = mean[c];
        const float k1 = g[c] * is;
        const float m1 = (float)(sums[2 * c] * inv_count);
        const float m2 = (float)(sums[2 * c + 1] * inv_count);
        dx[e] = k1 * (d - m1 - (xr[i] - mu) * is * m2);
    }
}

// ------------------------------------------------------------------ linear interpolation
struct Interp { int i0, i1; float l0, l1; };

__device__ __forceinline__ Interp interp_src(int o, int Lin, float scale, int align) {
    float src;
    if (align) {
        src = __fmul_rn(scale, (float)o);
    } else {
        src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)o, 0.5f)), 0.5f);
        if (src < 0.f) src = 0.f;
    }
    Interp r;
    r.i0 = min((int)src, Lin - 1);
    float l1 = __fsub_rn(src, (float)r.i0);
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    r.i1 = r.i0 + (r.i0 < Lin - 1 ? 1 : 0);
    r.l1 = l1;
    r.l0 = __fsub_rn(1.f, l1);
    return r;
}

__global__ void interp_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int Lin,
                                  int Lout, float scale, int align) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / Lout;
        const int o = (int)(e - row * Lout);
        const Interp s = interp_src(o, Lin, scale, align);
        const float* xr = x + row * Lin;
        y[e] = __fadd_rn(__fmul_rn(s.l0, xr[s.i0]), __fmul_rn(s.l1, xr[s.i1]));
    }
}

// gather form of the adjoint: input i collects every output whose i0 or i1 is i (fixed order).
// One workgroup per output-gradient row: the row is read ONCE, coalesced, into LDS and the Lin gathers run from there
// (every input touches ~2*Lout/Lin outputs; done from global memory the 33 MB tensor was fetched 24 times over).
constexpr int kInterpMaxLout = 8192;
__device__ __forceinline__ float interp_bwd_one(const float* dr, int i, int Lin, int Lout, float scale, float inv_scale, int align) {
    // conservative output window: src in (i-1, i+1)  ->  o in ((i-1)+0.5)/scale-0.5 .. ((i+1)+0.5)/scale-0.5
    int lo, hi;
    if (align) {
        lo = (int)floorf(((float)i - 1.f) * inv_scale) - 2;
        hi = (int)ceilf(((float)i + 1.f) * inv_scale) + 2;
    } else {
        lo = (int)floorf(((float)i - 0.5f) * inv_scale - 0.5f) - 2;
        hi = (int)ceilf(((float)i + 1.5f) * inv_scale - 0.5f) + 2;
    }
    if (i == 0) lo = 0;  // clamped sources all land on index 0
    if (lo < 0) lo = 0;
    if (hi > Lout - 1 || i == Lin - 1) hi = Lout - 1;
    float g = 0.f;
    for (int o = lo; o <= hi; ++o) {
        const Interp s = interp_src(o, Lin, scale, align);
        const float d = dr[o];
        if (s.i0 == i) g += s.l0 * d;
        if (s.i1 == i) g += s.l1 * d;
    }
    return g;
}

__global__ void interp_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t total, int Lin,
                                  int Lout, float scale, float inv_scale, int align) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / Lin;
        dx[e] = interp_bwd_one(dy + row * Lout, (int)(e - row * Lin), Lin, Lout, scale, inv_scale, align);
    }
}

__global__ void interp_bwd_rows_kernel(const float* __restrict__ dy, float* __restrict__ dx, int rows, int Lin, int Lout,
                                       float scale, float inv_scale, int align) {
    extern __shared__ float srow[];
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* dr = dy + (size_t)row * Lout;
        for (int o = threadIdx.x; o < Lout; o += blockDim.x) srow[o] = dr[o];
        __syncthreads();
        for (int i = threadIdx.x; i < Lin; i += blockDim.x)
            dx[(size_t)row * Lin + i] = interp_bwd_one(srow, i, Lin, Lout, scale, inv_scale, align);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ dropout
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask,
                                   size_t n, float p, float scale, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (seed_dev != nullptr) seed = seed_dev[0];   // captured launch (HIP graph): this step's seed lives on the device
    const uint64_t key = splitmix64(seed);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const uint64_t bits = splitmix64(key ^ (e * 0x2545F4914F6CDD1Dull));
        const float u = (float)(bits >> 40) * (1.0f / 16777216.0f);
        const uint8_t keep = u >= p ? 1 : 0;
        mask[e] = keep;
        y[e] = keep ? x[e] * scale : 0.f;
    }
}

__global__ void mask_scale_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                  float* __restrict__ y, size_t n, float scale) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        y[e] = mask[e] ? x[e] * scale : 0.f;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" {

int ssecg_bn_reduce_partials(const float* partial, int parts, int C, double* sums, float* dgamma, float* dbeta,
                             void* stream) {
    if (!partial || !sums || parts <= 0 || C <= 0 || ((dgamma == nullptr) != (dbeta == nullptr))) return SSECG_E_INVAL;
    hipLaunchKernelGGL(bn_reduce_partials_kernel<false>, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, partial,
                       parts, C, sums, dgamma, dbeta, 0.0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr);
    return (int)hipGetLastError();
}

int ssecg_bn_stats_finalize(const float* partial, int parts, int C, double count, float eps, float momentum, float* mean,
                            float* invstd, float* running_mean, float* running_var, const float* gamma, const float* beta,
                            float* aff_scale, float* aff_shift, void* stream) {
    if (!partial || !mean || !invstd || parts <= 0 || C <= 0 || count <= 0.0) return SSECG_E_INVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SSECG_E_INVAL;
    if ((aff_scale == nullptr) != (aff_shift == nullptr) || (aff_scale != nullptr && (!gamma || !beta))) return SSECG_E_INVAL;
    hipLaunchKernelGGL(bn_reduce_partials_kernel<true>, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, partial,
                       parts, C, nullptr, nullptr, nullptr, count, eps, momentum, mean, invstd, running_mean, running_var,
                       gamma, beta, aff_scale, aff_shift);
    return (int)hipGetLastError();
}

int ssecg_bn_finalize(const double* sums, int C, double count, float eps, float momentum, float* mean, float* invstd,
                      float* running_mean, float* running_var, const float* gamma, const float* beta, float* aff_scale,
                      float* aff_shift, void* stream) {
    if (!sums || !mean || !invstd || C <= 0 || count <= 0.0) return SSECG_E_INVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SSECG_E_INVAL;
    if ((aff_scale == nullptr) != (aff_shift == nullptr) || (aff_scale != nullptr && (!gamma || !beta))) return SSECG_E_INVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, C, count, eps,
                       momentum, mean, invstd, running_mean, running_var, gamma, beta, aff_scale, aff_shift);
    return (int)hipGetLastError();
}

int ssecg_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, int C,
                  float eps, float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, C, eps, scale, shift);
    return (int)hipGetLastError();
}

int ssecg_bn_fold_multi(const int64_t* table, int nlayers, int max_channels, void* stream) {
    if (!table || nlayers <= 0 || max_channels <= 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(bn_fold_multi_kernel, dim3((max_channels + 63) / 64, nlayers), dim3(64), 0, (hipStream_t)stream, table);
    return (int)hipGetLastError();
}

int ssecg_bn_mask_supported(int N, int C, int L) {
    if (N <= 0 || C <= 0 || L <= 0) return SSECG_E_INVAL;
    return (((size_t)N * C * L) % 8 == 0 && L >= 4) ? 1 : 0;
}

int ssecg_bn_apply_fwd(const float* x, float* y, int N, int C, int L, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, const float* residual, int relu, unsigned char* mask_bits,
                       void* stream) {
    return ssecg_bn_apply_fwd_resbn(x, y, N, C, L, mean, invstd, gamma, beta, residual, nullptr, nullptr, nullptr, nullptr, relu,
                                    mask_bits, stream);
}

int ssecg_bn_apply_fwd_resbn(const float* x, float* y, int N, int C, int L, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, const float* residual, const float* res_mean,
                             const float* res_invstd, const float* res_gamma, const float* res_beta, int relu,
                             unsigned char* mask_bits, void* stream) {
    if (!x || !y || !mean || !invstd || !gamma || !beta || N <= 0 || C <= 0 || L <= 0) return SSECG_E_INVAL;
    const bool resbn = res_mean != nullptr;
    if (resbn != (res_invstd != nullptr) || resbn != (res_gamma != nullptr) || resbn != (res_beta != nullptr)) return SSECG_E_INVAL;
    if (resbn && residual == nullptr) return SSECG_E_INVAL;
    const ResBN rs{res_mean, res_invstd, res_gamma, res_beta};
    const size_t total = (size_t)N * C * L;
    const bool vec = (total % 4 == 0) && L >= 4 && aligned16(x) && aligned16(y) && (residual == nullptr || aligned16(residual));
    // the packed mask is written by the vector kernel, a byte per pair of lanes
    if (mask_bits != nullptr && !(vec && relu && ssecg_bn_mask_supported(N, C, L) == 1)) return SSECG_E_INVAL;
    hipStream_t st = (hipStream_t)stream;
#define SSECG_APPLY(V_, R_, G_)                                                                                           \
    hipLaunchKernelGGL((bn_apply_fwd_kernel<V_, R_>), dim3(G_), dim3(kT), 0, st, x, y, total, C, L, mean, invstd, gamma, beta, \
                       residual, relu, mask_bits, rs)
    if (vec) {
        const int g = grid_for(total / 4, kT * 2, 8192);
        if (resbn) SSECG_APPLY(true, true, g); else SSECG_APPLY(true, false, g);
    } else {
        const int g = grid_for(total, kT * 4, 8192);
        if (resbn) SSECG_APPLY(false, true, g); else SSECG_APPLY(false, false, g);
    }
#undef SSECG_APPLY
    return (int)hipGetLastError();
}

int ssecg_bn_bwd_parts(int N, int C, int L) {
    if (N <= 0 || C <= 0 || L <= 0) return SSECG_E_INVAL;
    int s = 2048 / C;
    if (s < 1) s = 1;
    if (s > N) s = N;
    return s;
}

int ssecg_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, int relu_recompute, int N, int C, int L, float* partial,
                        const unsigned char* mask_bits, void* stream) {
    if (!dy || !x || !mean || !invstd || !partial || N <= 0 || C <= 0 || L <= 0) return SSECG_E_INVAL;
    if (relu_recompute && (!gamma || !beta || y != nullptr)) return SSECG_E_INVAL;
    if (mask_bits != nullptr && (y != nullptr || relu_recompute || ssecg_bn_mask_supported(N, C, L) != 1)) return SSECG_E_INVAL;
    const int S = ssecg_bn_bwd_parts(N, C, L);
    const bool vec = (L % 4 == 0) && aligned16(dy) && aligned16(x) && (y == nullptr || aligned16(y));
    hipStream_t st = (hipStream_t)stream;
    // rows at arbitrary 4-byte alignment (L % 4 != 0): 16-byte raw buffer loads, up to 256 lanes per row, 32-bit byte offsets
    const size_t nbytes = (size_t)N * C * L * 4;
    if (!vec && L > 4 && L <= 1024 && nbytes < 0x7fffff00ull && getenv("SSECG_BN_ROWS") == nullptr) {
        int sh = 0;
        while ((1 << sh) < (L + 3) / 4) ++sh;
#define SSECG_ROWS(R_, M_) hipLaunchKernelGGL((bn_bwd_reduce_rows_kernel<R_, M_>), dim3(C, S), dim3(kT), 0, st, dy, y, x, mean, invstd, gamma, beta, N, C, L, partial, mask_bits, sh, (unsigned)nbytes)
        if (relu_recompute) SSECG_ROWS(true, 0);
        else if (mask_bits != nullptr) SSECG_ROWS(false, 2);
        else if (y != nullptr) SSECG_ROWS(false, 1);
        else SSECG_ROWS(false, 0);
#undef SSECG_ROWS
        return (int)hipGetLastError();
    }
#define SSECG_RED(V_, R_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<V_, R_>), dim3(C, S), dim3(kT), 0, st, dy, y, x, mean, invstd, gamma, beta, N, C, L, partial, mask_bits)
    if (vec) { if (relu_recompute) SSECG_RED(true, true); else SSECG_RED(true, false); }
    else { if (relu_recompute) SSECG_RED(false, true); else SSECG_RED(false, false); }
#undef SSECG_RED
    return (int)hipGetLastError();
}

int ssecg_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int relu_recompute, const double* sums, double count, int N,
                       int C, int L, float* dx, float* dz_out, const unsigned char* mask_bits, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !sums || !dx || N <= 0 || C <= 0 || L <= 0 || count <= 0.0)
        return SSECG_E_INVAL;
    if (relu_recompute && (!beta || y != nullptr)) return SSECG_E_INVAL;
    if (mask_bits != nullptr && (y != nullptr || relu_recompute || ssecg_bn_mask_supported(N, C, L) != 1)) return SSECG_E_INVAL;
    const size_t total = (size_t)N * C * L;
    const bool vec = (total % 4 == 0) && L >= 4 && aligned16(dy) && aligned16(x) && aligned16(dx) &&
                     (y == nullptr || aligned16(y)) && (dz_out == nullptr || aligned16(dz_out));
    hipStream_t st = (hipStream_t)stream;
    const double inv = 1.0 / count;
#define SSECG_APP(V_, R_, G_)                                                                                         \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<V_, R_>), dim3(G_), dim3(kT), 0, st, dy, y, x, mean, invstd, gamma, beta, sums, \
                       inv, total, C, L, dx, dz_out, mask_bits)
    if (vec) {
        const int gsz = grid_for(total / 4, kT * 2, 8192);
        if (relu_recompute) SSECG_APP(true, true, gsz); else SSECG_APP(true, false, gsz);
    } else {
        const int gsz = grid_for(total, kT * 4, 8192);
        if (relu_recompute) SSECG_APP(false, true, gsz); else SSECG_APP(false, false, gsz);
    }
#undef SSECG_APP
    return (int)hipGetLastError();
}

// ---- two BatchNorms behind ONE masked gradient (a downsample block's bn2 and the BatchNorm of its 1x1 branch) in one pass each
int ssecg_bn_bwd_pair_supported(int N, int C, int L) {
    if (N <= 0 || C <= 0 || L <= 0) return 0;
    const size_t total = (size_t)N * C * L;
    if (total % 4 != 0 || L < 4) return 0;                                   // the flat float4 apply pass
    if (L % 4 == 0) return 1;                                                // aligned float4 reduction
    return (L > 4 && L <= 1024 && total * 4 < 0x7fffff00ull) ? 1 : 0;        // rows kernel (16-byte raw buffer loads)
}

int ssecg_bn_bwd_reduce_pair(const float* dy, const float* y, const unsigned char* mask_bits, const float* x, const float* mean,
                             const float* invstd, const float* x2, const float* mean2, const float* invstd2, int N, int C, int L,
                             float* partial, float* partial2, void* stream) {
    if (!dy || !x || !mean || !invstd || !x2 || !mean2 || !invstd2 || !partial || !partial2) return SSECG_E_INVAL;
    if ((y == nullptr) == (mask_bits == nullptr)) return SSECG_E_INVAL;     // exactly one mask source: the block's final ReLU
    if (!ssecg_bn_bwd_pair_supported(N, C, L)) return SSECG_E_INVAL;
    if (mask_bits != nullptr && ssecg_bn_mask_supported(N, C, L) != 1) return SSECG_E_INVAL;
    if (!aligned16(dy) || !aligned16(x) || !aligned16(x2) || (y != nullptr && !aligned16(y))) return SSECG_E_INVAL;
    const int S = ssecg_bn_bwd_parts(N, C, L);
    hipStream_t st = (hipStream_t)stream;
    BnPair pb{};
    pb.x = x2; pb.mean = mean2; pb.invstd = invstd2; pb.out = partial2;
    if (L % 4 == 0) {
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, false, true>), dim3(C, S), dim3(kT), 0, st, dy, y, x, mean, invstd,
                           (const float*)nullptr, (const float*)nullptr, N, C, L, partial, mask_bits, pb);
    } else {
        const size_t nbytes = (size_t)N * C * L * 4;
        int sh = 0;
        while ((1 << sh) < (L + 3) / 4) ++sh;
        if (mask_bits != nullptr)
            hipLaunchKernelGGL((bn_bwd_reduce_rows_kernel<false, 2, true>), dim3(C, S), dim3(kT), 0, st, dy, y, x, mean, invstd,
                               (const float*)nullptr, (const float*)nullptr, N, C, L, partial, mask_bits, sh, (unsigned)nbytes, pb);
        else
            hipLaunchKernelGGL((bn_bwd_reduce_rows_kernel<false, 1, true>), dim3(C, S), dim3(kT), 0, st, dy, y, x, mean, invstd,
                               (const float*)nullptr, (const float*)nullptr, N, C, L, partial, mask_bits, sh, (unsigned)nbytes, pb);
    }
    return (int)hipGetLastError();
}

int ssecg_bn_bwd_apply_pair(const float* dy, const float* y, const unsigned char* mask_bits, const float* x, const float* mean,
                            const float* invstd, const float* gamma, const double* sums, const float* x2, const float* mean2,
                            const float* invstd2, const float* gamma2, const double* sums2, double count, int N, int C, int L,
                            float* dx, float* dx2, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !sums || !x2 || !mean2 || !invstd2 || !gamma2 || !sums2 || !dx || !dx2 ||
        !(count > 0.0))
        return SSECG_E_INVAL;
    if ((y == nullptr) == (mask_bits == nullptr)) return SSECG_E_INVAL;
    if (!ssecg_bn_bwd_pair_supported(N, C, L)) return SSECG_E_INVAL;
    if (mask_bits != nullptr && ssecg_bn_mask_supported(N, C, L) != 1) return SSECG_E_INVAL;
    if (!aligned16(dy) || !aligned16(x) || !aligned16(x2) || !aligned16(dx) || !aligned16(dx2) || (y != nullptr && !aligned16(y)))
        return SSECG_E_INVAL;
    const size_t total = (size_t)N * C * L;
    BnPair pb{};
    pb.x = x2; pb.mean = mean2; pb.invstd = invstd2; pb.gamma = gamma2; pb.sums = sums2; pb.out = dx2;
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false, true>), dim3(grid_for(total / 4, kT * 2, 8192)), dim3(kT), 0, (hipStream_t)stream,
                       dy, y, x, mean, invstd, gamma, (const float*)nullptr, sums, 1.0 / count, total, C, L, dx, (float*)nullptr,
                       mask_bits, pb);
    return (int)hipGetLastError();
}

int ssecg_bn_param_grads(const double* sums, int C, float* dgamma, float* dbeta, void* stream) {
    if (!sums || !dgamma || !dbeta || C <= 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, C, dgamma, dbeta);
    return (int)hipGetLastError();
}

int ssecg_channel_sum(const float* x, int N, int C, int L, float* out, float* scratch, size_t scratch_bytes, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || L <= 0 || (long long)N * L > 0x7fffffffLL) return SSECG_E_INVAL;
    const int S = N < kChS ? N : kChS;
    if (scratch == nullptr || S == 1) {
        hipLaunchKernelGGL(channel_sum_kernel, dim3(C, 1), dim3(kT), 0, (hipStream_t)stream, x, N, C, L, out);
        return (int)hipGetLastError();
    }
    if (scratch_bytes < (size_t)C * S * sizeof(float)) return SSECG_E_WORKSPACE;
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, S), dim3(kT), 0, (hipStream_t)stream, x, N, C, L, scratch);
    hipLaunchKernelGGL(channel_sum_finish_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, scratch, C, S, out);
    return (int)hipGetLastError();
}

int ssecg_maxpool1d_fwd(const float* x, float* y, int rows, int Lin, int Lout, int ksize, int stride, int pad, void* stream) {
    if (!x || !y || rows <= 0 || Lin <= 0 || Lout <= 0 || ksize <= 0 || stride <= 0 || pad < 0 || 2 * pad > ksize)
        return SSECG_E_INVAL;
    if ((Lin + 2 * pad - ksize) / stride + 1 != Lout) return SSECG_E_INVAL;
    const size_t total = (size_t)rows * Lout;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total, kT * 4, 8192)), dim3(kT), 0, (hipStream_t)stream, x, y, total, Lin,
                       Lout, ksize, stride, pad);
    return (int)hipGetLastError();
}

int ssecg_maxpool1d_bwd(const float* x, const float* dy, float* dx, int rows, int Lin, int Lout, int ksize, int stride,
                        int pad, void* stream) {
    if (!x || !dy || !dx || rows <= 0 || Lin <= 0 || Lout <= 0 || ksize <= 0 || stride <= 0 || pad < 0 || 2 * pad > ksize)
        return SSECG_E_INVAL;
    if ((Lin + 2 * pad - ksize) / stride + 1 != Lout) return SSECG_E_INVAL;
    const size_t total = (size_t)rows * Lin;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, kT * 4, 8192)), dim3(kT), 0, (hipStream_t)stream, x, dy, dx, total,
                       Lin, Lout, ksize, stride, pad);
    return (int)hipGetLastError();
}

static bool bad_pool(int rows, int Lin, int Lout, int k, int s, int pad) {
    return rows <= 0 || Lin <= 0 || Lout <= 0 || k <= 0 || s <= 0 || pad < 0 || 2 * pad > k || (Lin + 2 * pad - k) / s + 1 != Lout;
}

int ssecg_bn_relu_maxpool_fwd(const float* x, float* y, int N, int C, int Lin, int Lout, int ksize, int stride, int pad,
                              const float* mean, const float* invstd, const float* gamma, const float* beta, void* stream) {
    if (!x || !y || !gamma || !beta || N <= 0 || C <= 0 || bad_pool(N * C, Lin, Lout, ksize, stride, pad)) return SSECG_E_INVAL;
    if ((mean == nullptr) != (invstd == nullptr)) return SSECG_E_INVAL;
    const size_t total = (size_t)N * C * Lout;
    if (ksize == 3 && stride == 2 && pad == 1 && Lin % 4 == 0 && Lout == Lin / 2 && aligned16(x) && aligned16(y) &&
        (size_t)N * C * Lin / 4 < 0x7fffffffull)
        hipLaunchKernelGGL(stem_pool_fwd_quad_kernel, dim3(grid_for((size_t)N * C * Lin / 4, kT * 2, 8192)), dim3(kT), 0,
                           (hipStream_t)stream, x, y, (unsigned)((size_t)N * C * Lin / 4), C, Lin, mean, invstd, gamma, beta);
    else
        hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(grid_for(total, kT * 4, 8192)), dim3(kT), 0, (hipStream_t)stream, x, y,
                           total, C, Lin, Lout, ksize, stride, pad, mean, invstd, gamma, beta);
    return (int)hipGetLastError();
}

int ssecg_bn_relu_maxpool_bwd_reduce(const float* dy, const void* x, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, int N, int C, int Lin, int Lout, int ksize,
                                     int stride, int pad, float* partial, int lp, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !beta || !partial || N <= 0 || C <= 0 ||
        bad_pool(N * C, Lin, Lout, ksize, stride, pad) || (long long)N * Lin > 0x7fffffffLL || lp < 0 || lp > 2)
        return SSECG_E_INVAL;
    const int S = ssecg_bn_bwd_parts(N, C, Lin);
    const bool quad = ksize == 3 && stride == 2 && pad == 1 && Lin % 4 == 0 && aligned16(x);
    if (lp == 2 && !quad) return SSECG_E_INVAL;   // bf16 storage of the conv output: the stem's own shape only
    const float* xf = reinterpret_cast<const float*>(x);
    auto kg = lp ? bn_relu_maxpool_bwd_reduce_kernel<true> : bn_relu_maxpool_bwd_reduce_kernel<false>;
    hipStream_t st = (hipStream_t)stream;
    if (!quad)
        hipLaunchKernelGGL(kg, dim3(C, S), dim3(kT), 0, st, dy, xf, mean, invstd, gamma, beta, N, C, Lin, Lout, ksize, stride, pad, partial);
    else if (lp == 2)
        hipLaunchKernelGGL((stem_pool_bwd_reduce_quad_kernel<true, true>), dim3(C, S), dim3(kT), 0, st, dy, x, mean, invstd, gamma, beta, N, C,
                           Lin, Lout, partial);
    else if (lp == 1)
        hipLaunchKernelGGL((stem_pool_bwd_reduce_quad_kernel<true, false>), dim3(C, S), dim3(kT), 0, st, dy, x, mean, invstd, gamma, beta, N, C,
                           Lin, Lout, partial);
    else
        hipLaunchKernelGGL((stem_pool_bwd_reduce_quad_kernel<false, false>), dim3(C, S), dim3(kT), 0, st, dy, x, mean, invstd, gamma, beta, N, C,
                           Lin, Lout, partial);
    return (int)hipGetLastError();
}

int ssecg_bn_relu_maxpool_bwd_apply(const float* dy, const void* x, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, const double* sums, double count, int N, int C,
                                    int Lin, int Lout, int ksize, int stride, int pad, void* dx, int lp, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !beta || !sums || !dx || N <= 0 || C <= 0 || count <= 0.0 ||
        bad_pool(N * C, Lin, Lout, ksize, stride, pad) || lp < 0 || lp > 2)
        return SSECG_E_INVAL;
    const size_t total = (size_t)N * C * Lin;
    const bool quad = ksize == 3 && stride == 2 && pad == 1 && Lin % 4 == 0 && aligned16(x) && aligned16(dx);
    if (lp == 2 && !quad) return SSECG_E_INVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gq(grid_for(total / 4, kT * 2, 8192));
    if (!quad) {
        auto kg = lp ? bn_relu_maxpool_bwd_apply_kernel<true> : bn_relu_maxpool_bwd_apply_kernel<false>;
        hipLaunchKernelGGL(kg, dim3(grid_for(total, kT * 4, 8192)), dim3(kT), 0, st, dy, reinterpret_cast<const float*>(x), mean, invstd, gamma,
                           beta, sums, 1.0 / count, total, C, Lin, Lout, ksize, stride, pad, reinterpret_cast<float*>(dx));
    } else if (lp == 2)
        hipLaunchKernelGGL((stem_pool_bwd_apply_quad_kernel<true, true>), gq, dim3(kT), 0, st, dy, x, mean, invstd, gamma, beta, sums,
                           1.0 / count, total / 4, C, Lin, Lout, dx);
    else if (lp == 1)
        hipLaunchKernelGGL((stem_pool_bwd_apply_quad_kernel<true, false>), gq, dim3(kT), 0, st, dy, x, mean, invstd, gamma, beta, sums,
                           1.0 / count, total / 4, C, Lin, Lout, dx);
    else
        hipLaunchKernelGGL((stem_pool_bwd_apply_quad_kernel<false, false>), gq, dim3(kT), 0, st, dy, x, mean, invstd, gamma, beta, sums,
                           1.0 / count, total / 4, C, Lin, Lout, dx);
    return (int)hipGetLastError();
}

static float interp_scale(int Lin, int Lout, int align) {
    if (align) return Lout > 1 ? (float)(Lin - 1) / (float)(Lout - 1) : 0.f;
    return (float)Lin / (float)Lout;
}

int ssecg_interp_linear_fwd(const float* x, float* y, int rows, int Lin, int Lout, int align_corners, void* stream) {
    if (!x || !y || rows <= 0 || Lin <= 0 || Lout <= 0) return SSECG_E_INVAL;
    const size_t total = (size_t)rows * Lout;
    hipLaunchKernelGGL(interp_fwd_kernel, dim3(grid_for(total, kT * 4, 8192)), dim3(kT), 0, (hipStream_t)stream, x, y, total, Lin,
                       Lout, interp_scale(Lin, Lout, align_corners), align_corners);
    return (int)hipGetLastError();
}

int ssecg_interp_linear_bwd(const float* dy, float* dx, int rows, int Lin, int Lout, int align_corners, void* stream) {
    if (!dy || !dx || rows <= 0 || Lin <= 0 || Lout <= 0) return SSECG_E_INVAL;
    const size_t total = (size_t)rows * Lin;
    const float sc = interp_scale(Lin, Lout, align_corners);
    const float inv = sc > 0.f ? 1.0f / sc : (float)Lout;
    if (Lout <= kInterpMaxLout && Lout >= 2 * Lin) {
        const int grid = rows < 8192 ? rows : 8192;
        hipLaunchKernelGGL(interp_bwd_rows_kernel, dim3(grid), dim3(kT), (size_t)Lout * sizeof(float), (hipStream_t)stream, dy, dx, rows,
                           Lin, Lout, sc, inv, align_corners);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(interp_bwd_kernel, dim3(grid_for(total, kT, 8192)), dim3(kT), 0, (hipStream_t)stream, dy, dx, total, Lin, Lout,
                       sc, inv, align_corners);
    return (int)hipGetLastError();
}

int ssecg_dropout_fwd(const float* x, float* y, uint8_t* mask, size_t n, float p, uint64_t seed, const uint64_t* seed_dev,
                      void* stream) {
    if (!x || !y || !mask || n == 0 || !(p >= 0.f && p < 1.f)) return SSECG_E_INVAL;
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for(n, kT * 4, 4096)), dim3(kT), 0, (hipStream_t)stream, x, y, mask, n, p,
                       1.0f / (1.0f - p), seed, seed_dev);
    return (int)hipGetLastError();
}

int ssecg_mask_scale(const float* x, const uint8_t* mask, float* y, size_t n, float scale, void* stream) {
    if (!x || !y || !mask || n == 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(mask_scale_kernel, dim3(grid_for(n, kT * 4, 4096)), dim3(kT), 0, (hipStream_t)stream, x, mask, y, n, scale);
    return (int)hipGetLastError();
}

int ssecg_amp_stem_pool_supported(int N, int C, int Lin) {
    return (N > 0 && C > 0 && C % 8 == 0 && Lin >= 4 && Lin % 4 == 0 && (size_t)N * C * Lin / 32 < 0x7fffffffull) ? 1 : 0;
}

int ssecg_amp_stem_pool_fwd(const void* x, void* yb, int N, int C, int Lin, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, int x16, void* stream) {
    if (!x || !yb || !gamma || !beta || !ssecg_amp_stem_pool_supported(N, C, Lin) || !aligned16(x) || !aligned16(yb)) return SSECG_E_INVAL;
    if ((mean == nullptr) != (invstd == nullptr)) return SSECG_E_INVAL;
    const size_t items = (size_t)N * (C / 8) * (Lin / 4);
    auto k = x16 ? stem_pool_fwd_b16_kernel<true> : stem_pool_fwd_b16_kernel<false>;
    hipLaunchKernelGGL(k, dim3(grid_for(items, kT, 8192)), dim3(kT), 0, (hipStream_t)stream, x, (u32x4*)yb, (unsigned)items, C, Lin, mean,
                       invstd, gamma, beta);
    return (int)hipGetLastError();
}

}  // extern "C"
