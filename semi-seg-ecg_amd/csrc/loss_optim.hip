// Pseudo-label head, cross-entropy losses (forward + gradient in one pass over the
// logits) and the multi-tensor AdamW / EMA updates.  Logits are (N, K, L) fp32 with
// K = num_classes small (4 for ECG delineation); one lane owns one (n, l) position
// and walks the K class planes, so every load is a coalesced row segment.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "ssecg.h"

namespace {

constexpr int kT = 256;
constexpr int kMaxClasses = 32;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

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

inline int parts_for(long long positions) {
    long long b = (positions + kT - 1) / kT;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

template <int K>  // K = 0 -> runtime class count
__global__ void softmax_conf_argmax_kernel(const float* __restrict__ logits, int N, int Kr, int L,
                                           float* __restrict__ conf, int64_t* __restrict__ mask,
                                           float* __restrict__ prob) {
    const int KK = K ? K : Kr;
    const size_t P = (size_t)N * L;
    for (size_t pidx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pidx < P; pidx += (size_t)gridDim.x * blockDim.x) {
        const size_t n = pidx / L;
        const int l = (int)(pidx - n * L);
        const float* x = logits + n * KK * L + l;
        float v[K ? K : kMaxClasses];
        float m = -INFINITY;
        int am = 0;
#pragma unroll
        for (int c = 0; c < (K ? K : kMaxClasses); ++c) {
            if (c < KK) {
                v[c] = x[(size_t)c * L];
                if (v[c] > m) { m = v[c]; am = c; }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < (K ? K : kMaxClasses); ++c)
            if (c < KK) { v[c] = expf(v[c] - m); s += v[c]; }
        const float inv = 1.0f / s;
        if (conf) conf[pidx] = inv;  // exp(0) / sum
        if (mask) mask[pidx] = am;
        if (prob) {
            float* pr = prob + n * KK * L + l;
#pragma unroll
            for (int c = 0; c < (K ? K : kMaxClasses); ++c)
                if (c < KK) pr[(size_t)c * L] = v[c] * inv;
        }
    }
}

template <int K, bool SOFT>
__global__ void ce_fwd_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                  const float* __restrict__ conf, float thresh, const float* __restrict__ prob,
                                  int N, int Kr, int L, float gscale, float* __restrict__ dlogits,
                                  float* __restrict__ partial) {
    const int KK = K ? K : Kr;
    const size_t P = (size_t)N * L;
    float lsum = 0.f, wsum = 0.f;
    for (size_t pidx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pidx < P; pidx += (size_t)gridDim.x * blockDim.x) {
        const size_t n = pidx / L;
        const int l = (int)(pidx - n * L);
        const size_t base = n * KK * L + l;
        float v[K ? K : kMaxClasses];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < (K ? K : kMaxClasses); ++c)
            if (c < KK) { v[c] = logits[base + (size_t)c * L]; m = fmaxf(m, v[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < (K ? K : kMaxClasses); ++c)
            if (c < KK) s += expf(v[c] - m);
        const float lse = m + logf(s);
        if (SOFT) {
            float psum = 0.f, loss = 0.f;
            float pr[K ? K : kMaxClasses];
#pragma unroll
            for (int c = 0; c < (K ? K : kMaxClasses); ++c)
                if (c < KK) {
                    pr[c] = prob[base + (size_t)c * L];
                    psum += pr[c];
                    loss -= pr[c] * (v[c] - lse);
                }
            lsum += loss;
            wsum += 1.f;
#pragma unroll
            for (int c = 0; c < (K ? K : kMaxClasses); ++c)
                if (c < KK) dlogits[base + (size_t)c * L] = (expf(v[c] - lse) * psum - pr[c]) * gscale;
        } else {
            const int t = (int)target[pidx];
            const float w = (conf == nullptr || conf[pidx] >= thresh) ? 1.f : 0.f;
            float xt = 0.f;
#pragma unroll
            for (int c = 0; c < (K ? K : kMaxClasses); ++c)
                if (c < KK && c == t) xt = v[c];
            lsum += (lse - xt) * w;
            wsum += w;
            const float g = w * gscale;
#pragma unroll
            for (int c = 0; c < (K ? K : kMaxClasses); ++c)
                if (c < KK) dlogits[base + (size_t)c * L] = (expf(v[c] - lse) - (c == t ? 1.f : 0.f)) * g;
        }
    }
    block_sum2(lsum, wsum);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = lsum;
        partial[2 * blockIdx.x + 1] = wsum;
    }
}

__global__ void sum_partials_kernel(const float* partial, int parts, int width, float scale, float* out) {
    __shared__ double sh[kT];
    for (int k = 0; k < width; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < parts; i += blockDim.x) s += (double)partial[(size_t)i * width + k];
        sh[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < kT; ++i) t += sh[i];
            out[k] = (float)(t * (double)scale);
        }
        __syncthreads();
    }
}

// The tail of the two-term semi-supervised losses (FixMatch: src/algorithms/fixmatch.py:102-118; MeanTeacher: mean_teacher.py:103-117)
// in one launch: out = { 0.5 (a + b), 0.5 (a + b), a, b, w } with a = sx_scale * sum px[:,0] (supervised term), b = su_scale * sum
// pu[:,0] (unsupervised term), w = su_scale * sum pu[:,1] (FixMatch's mask ratio).  The sums are sum_partials_kernel's, operation for
// operation (the same fp64 order), the combination is torch's fp32 (sx + su) * 0.5 - it replaces two sum_partials launches and three
// torch launches (add, mul, stack) with bit-identical results; out[0] is the differentiable loss, out[1:] the logged statistics.
__global__ void loss_pair_finish_kernel(const float* px, int nx, const float* pu, int nu, float sx_scale, float su_scale, float* out) {
    __shared__ double sh[kT];
    float r[3];
    for (int k = 0; k < 3; ++k) {
        const float* p = k == 0 ? px : pu;
        const int parts = k == 0 ? nx : nu, col = k == 2 ? 1 : 0;
        double s = 0.0;
        for (int i = threadIdx.x; i < parts; i += blockDim.x) s += (double)p[(size_t)i * 2 + col];
        sh[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < kT; ++i) t += sh[i];
            r[k] = (float)(t * (double)(k == 0 ? sx_scale : su_scale));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float loss = (r[0] + r[1]) * 0.5f;
        out[0] = loss; out[1] = loss; out[2] = r[0]; out[3] = r[1]; out[4] = r[2];
    }
}

// ------------------------------------------------------------------ AdamW / EMA
constexpr int kChunk = kT * 8;

__global__ void adamw_multi_kernel(const int64_t* __restrict__ table, float one_minus_b1, float b2, float one_minus_b2,
                                   float eps, float decay_mul, float step_size, float bc2_sqrt,
                                   const float* __restrict__ skip, float* __restrict__ skipped, double lr, double beta1,
                                   double beta2, int step, const double* __restrict__ coef) {
    if (coef != nullptr) {   // captured launch (HIP graph): this step's scalars live on the device, formed by the host exactly
        decay_mul = (float)coef[0]; step_size = (float)coef[1]; bc2_sqrt = (float)coef[2];   // as ssecg_adamw_multi forms them
        lr = coef[3]; step = (int)coef[4];
    }
    if (skip != nullptr && skip[0] != 0.f) {         // GradScaler.step: non-finite gradients -> the update is skipped
        // ... and it does not count as an optimizer step either (GradScaler.step never calls optimizer.step): the owner's
        // device-side counter of skipped launches.  Only this thread writes it and nobody reads it in a skipped launch.
        if (skipped != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) skipped[0] += 1.f;
        return;
    }
    if (skipped != nullptr) {
        const float sk = skipped[0];
        if (sk != 0.f) {   // the host counted sk launches that never happened: bias corrections of the TRUE step count
            const double t = (double)step - (double)sk;
            step_size = (float)(lr / (1.0 - pow(beta1, t)));
            bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
        }
    }
    const int64_t* row = table + 5 * (size_t)blockIdx.y;
    float* p = reinterpret_cast<float*>(row[0]);
    const float* g = reinterpret_cast<const float*>(row[1]);
    float* m = reinterpret_cast<float*>(row[2]);
    float* v = reinterpret_cast<float*>(row[3]);
    const int64_t n = row[4];
    const int64_t start = (int64_t)blockIdx.x * kChunk;
    if (start >= n) return;
    const int64_t end = start + kChunk < n ? start + kChunk : n;
    for (int64_t i = start + threadIdx.x; i < end; i += kT) {
        const float gi = g[i];
        float pi = p[i] * decay_mul;
        float mi = m[i];
        mi = mi + (gi - mi) * one_minus_b1;
        float vi = v[i] * b2;
        vi = vi + one_minus_b2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// torch.optim.SGD (src/utils/optimizer.py:15-26): g += wd*p; buf = g on the first step, else momentum*buf + g; p -= lr*buf
__global__ void sgd_multi_kernel(const int64_t* __restrict__ table, float lr, float momentum, float weight_decay,
                                 int first_step, const float* __restrict__ skip, const double* __restrict__ lr_dev) {
    if (skip != nullptr && skip[0] != 0.f) return;
    if (lr_dev != nullptr) lr = (float)lr_dev[0];   // captured launch: this step's learning rate lives on the device
    const int64_t* row = table + 4 * (size_t)blockIdx.y;
    float* p = reinterpret_cast<float*>(row[0]);
    const float* g = reinterpret_cast<const float*>(row[1]);
    float* buf = reinterpret_cast<float*>(row[2]);   // may be null when momentum == 0
    const int64_t n = row[3];
    const int64_t start = (int64_t)blockIdx.x * kChunk;
    if (start >= n) return;
    const int64_t end = start + kChunk < n ? start + kChunk : n;
    for (int64_t i = start + threadIdx.x; i < end; i += kT) {
        const float pi = p[i];
        float gi = g[i];
        if (weight_decay != 0.f) gi = gi + weight_decay * pi;
        if (buf != nullptr) {
            const float bi = first_step ? gi : momentum * buf[i] + gi;
            buf[i] = bi;
            gi = bi;
        }
        p[i] = pi - lr * gi;
    }
}

// Global L2 norm of all gradients (src/utils/misc.py:265-278) + GradScaler bookkeeping (src/utils/misc.py:236-256).
// Stage 1: per (chunk, tensor) sum of squares and count of non-finite elements.
__global__ void grad_sumsq_multi_kernel(const int64_t* __restrict__ table, int words, int grad_col, int numel_col,
                                        float* __restrict__ partial) {
    const int64_t* row = table + (size_t)words * blockIdx.y;
    const float* g = reinterpret_cast<const float*>(row[grad_col]);
    const int64_t n = row[numel_col];
    const int64_t start = (int64_t)blockIdx.x * kChunk;
    float ss = 0.f, bad = 0.f;
    if (start < n) {
        const int64_t end = start + kChunk < n ? start + kChunk : n;
        for (int64_t i = start + threadIdx.x; i < end; i += kT) {
            const float gi = g[i];
            ss += gi * gi;
            bad += (isfinite(gi) ? 0.f : 1.f);
        }
    }
    block_sum2(ss, bad);
    if (threadIdx.x == 0) {
        const size_t o = 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        partial[o] = ss;
        partial[o + 1] = bad;
    }
}

// Stage 2 (one workgroup): out = {norm, found_inf}; if `scaler` is given, torch.cuda.amp.GradScaler.update():
// scaler = {scale, growth_tracker, skipped_steps}: found_inf -> scale *= backoff, tracker = 0, skipped += 1; otherwise
// tracker += 1 and at growth_interval scale *= growth (kept only if finite), tracker = 0.
// (1024 threads, four loads in flight per thread: with 256 threads and one load per iteration the 25 k partials of the shipped model
// were 98 dependent round trips - 28 us for a 200 KB read)
__global__ __launch_bounds__(1024) void grad_norm_finalize_kernel(const float* __restrict__ partial, int parts, float* __restrict__ out,
                                                                  float* __restrict__ scaler, float growth, float backoff, int interval) {
    constexpr int NT = 1024;
    __shared__ double sh[2][NT];
    const float2* p2 = reinterpret_cast<const float2*>(partial);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * NT < parts; i += 4 * NT) {
        const float2 v0 = p2[i], v1 = p2[i + NT], v2 = p2[i + 2 * NT], v3 = p2[i + 3 * NT];
        s0 += (double)v0.x; b0 += (double)v0.y; s1 += (double)v1.x; b1 += (double)v1.y;
        s2 += (double)v2.x; b2 += (double)v2.y; s3 += (double)v3.x; b3 += (double)v3.y;
    }
    for (; i < parts; i += NT) { const float2 v = p2[i]; s0 += (double)v.x; b0 += (double)v.y; }
    sh[0][threadIdx.x] = (s0 + s1) + (s2 + s3); sh[1][threadIdx.x] = (b0 + b1) + (b2 + b3);
    __syncthreads();
    if (threadIdx.x < 32) {   // fixed order: 32 lanes x 32 consecutive entries, then lane 0 over the 32 lane sums
        double t = 0.0, b = 0.0;
        for (int k = 0; k < 32; ++k) { t += sh[0][32 * threadIdx.x + k]; b += sh[1][32 * threadIdx.x + k]; }
        sh[0][32 * threadIdx.x] = t; sh[1][32 * threadIdx.x] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, b = 0.0;
        for (int k = 0; k < 32; ++k) { t += sh[0][32 * k]; b += sh[1][32 * k]; }
        const float found = b > 0.0 ? 1.f : 0.f;
        out[0] = (float)sqrt(t);
        out[1] = found;
        if (scaler != nullptr) {
            if (found != 0.f) {
                scaler[0] = scaler[0] * backoff;
                scaler[1] = 0.f;
                scaler[2] = scaler[2] + 1.f;
            } else {
                const float succ = scaler[1] + 1.f;
                if ((int)succ == interval) {
                    const float ns = scaler[0] * growth;
                    if (isfinite(ns)) scaler[0] = ns;
                    scaler[1] = 0.f;
                } else {
                    scaler[1] = succ;
                }
            }
        }
    }
}

// torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (norm + 1e-6)), norm read from the device
__global__ void grad_clip_multi_kernel(const int64_t* __restrict__ table, int words, int grad_col, int numel_col,
                                       const float* __restrict__ norm, float max_norm) {
    const float coef = max_norm / (norm[0] + 1e-6f);
    if (!(coef < 1.f)) return;
    const int64_t* row = table + (size_t)words * blockIdx.y;
    float* g = reinterpret_cast<float*>(row[grad_col]);
    const int64_t n = row[numel_col];
    const int64_t start = (int64_t)blockIdx.x * kChunk;
    if (start >= n) return;
    const int64_t end = start + kChunk < n ? start + kChunk : n;
    for (int64_t i = start + threadIdx.x; i < end; i += kT) g[i] *= coef;
}

// Gradient shard staging of the data-parallel step (ssecg/parallel.py): dst[row.offset + i] = src[i] * scale for every tensor of a
// reduction bucket in ONE launch (rows {src*, element offset in the bucket, numel}; src == nullptr: the slot is zero-filled - a
// parameter that received no gradient on this rank; src may alias its destination - gradients accumulated in place over micro-steps).
// This is torch DDP's per-parameter ``mul_out(bucket_view, grad, 1 / world)`` (65 launches per step) as one kernel per bucket.
__global__ void pack_scaled_multi_kernel(const int64_t* __restrict__ table, float* __restrict__ dst, float scale) {
    const int64_t* row = table + 3 * (size_t)blockIdx.y;
    const float* s = reinterpret_cast<const float*>(row[0]);
    float* d = dst + row[1];
    const int64_t n = row[2];
    const int64_t start = (int64_t)blockIdx.x * kChunk;
    if (start >= n) return;
    const int64_t end = start + kChunk < n ? start + kChunk : n;
    if (s == nullptr) {
        for (int64_t i = start + threadIdx.x; i < end; i += kT) d[i] = 0.f;
    } else {
        for (int64_t i = start + threadIdx.x; i < end; i += kT) d[i] = s[i] * scale;
    }
}

__global__ void ema_multi_kernel(const int64_t* __restrict__ table, float decay, float one_minus) {
    const int64_t* row = table + 4 * (size_t)blockIdx.y;
    float* t = reinterpret_cast<float*>(row[0]);
    const int64_t n = row[2];
    const bool is_int = row[3] != 0;
    const int64_t start = (int64_t)blockIdx.x * kChunk;
    if (start >= n) return;
    const int64_t end = start + kChunk < n ? start + kChunk : n;
    if (is_int) {
        const int64_t* s = reinterpret_cast<const int64_t*>(row[1]);
        for (int64_t i = start + threadIdx.x; i < end; i += kT) t[i] = t[i] * decay + (float)s[i] * one_minus;
    } else {
        const float* s = reinterpret_cast<const float*>(row[1]);
        for (int64_t i = start + threadIdx.x; i < end; i += kT) t[i] = t[i] * decay + s[i] * one_minus;
    }
}

// Per-record confusion counts: counts[n][t][p] = #{l : target[n,l] == t, pred[n,l] == p}.  One workgroup per record,
// one private LDS histogram per wave (K*K bins), merged in fixed wave order -> deterministic integer result.
__global__ void seg_confusion_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ target, int K, int L,
                                     int32_t* __restrict__ counts) {
    extern __shared__ int32_t bins[];  // [kT/64][K*K]
    const int KK = K * K;
    for (int i = threadIdx.x; i < KK * (kT / 64); i += kT) bins[i] = 0;
    __syncthreads();
    int32_t* mine = bins + (threadIdx.x >> 6) * KK;
    const size_t base = (size_t)blockIdx.x * L;
    for (int l = threadIdx.x; l < L; l += kT) {
        const int64_t t = target[base + l], p = pred[base + l];
        if (t >= 0 && t < K && p >= 0 && p < K) atomicAdd(&mine[(int)t * K + (int)p], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KK; i += kT) {
        int32_t v = 0;
#pragma unroll
        for (int w = 0; w < kT / 64; ++w) v += bins[w * KK + i];
        counts[(size_t)blockIdx.x * KK + i] = v;
    }
}

}  // namespace

extern "C" {

int ssecg_abi_version(void) { return SSECG_ABI_VERSION; }
const char* ssecg_build_arch(void) { return "gfx950"; }

int ssecg_softmax_conf_argmax(const float* logits, int N, int K, int L, float* conf, int64_t* mask, float* prob,
                              void* stream) {
    if (!logits || N <= 0 || K <= 0 || K > kMaxClasses || L <= 0 || (!conf && !mask && !prob)) return SSECG_E_INVAL;
    const int grid = parts_for((long long)N * L);
    hipStream_t st = (hipStream_t)stream;
    if (K == 4)
        hipLaunchKernelGGL(softmax_conf_argmax_kernel<4>, dim3(grid), dim3(kT), 0, st, logits, N, K, L, conf, mask, prob);
    else
        hipLaunchKernelGGL(softmax_conf_argmax_kernel<0>, dim3(grid), dim3(kT), 0, st, logits, N, K, L, conf, mask, prob);
    return (int)hipGetLastError();
}

int ssecg_ce_parts(int N, int L) {
    if (N <= 0 || L <= 0) return SSECG_E_INVAL;
    return parts_for((long long)N * L);
}

int ssecg_ce_hard_fwd_bwd(const float* logits, const int64_t* target, const float* conf, float thresh, int N, int K,
                          int L, float grad_scale, float* dlogits, float* partial, void* stream) {
    if (!logits || !target || !dlogits || !partial || N <= 0 || K <= 0 || K > kMaxClasses || L <= 0) return SSECG_E_INVAL;
    const int grid = parts_for((long long)N * L);
    hipStream_t st = (hipStream_t)stream;
    if (K == 4)
        hipLaunchKernelGGL((ce_fwd_bwd_kernel<4, false>), dim3(grid), dim3(kT), 0, st, logits, target, conf, thresh, nullptr, N, K, L,
                           grad_scale, dlogits, partial);
    else
        hipLaunchKernelGGL((ce_fwd_bwd_kernel<0, false>), dim3(grid), dim3(kT), 0, st, logits, target, conf, thresh, nullptr, N, K, L,
                           grad_scale, dlogits, partial);
    return (int)hipGetLastError();
}

int ssecg_ce_soft_fwd_bwd(const float* logits, const float* prob, int N, int K, int L, float grad_scale, float* dlogits,
                          float* partial, void* stream) {
    if (!logits || !prob || !dlogits || !partial || N <= 0 || K <= 0 || K > kMaxClasses || L <= 0) return SSECG_E_INVAL;
    const int grid = parts_for((long long)N * L);
    hipStream_t st = (hipStream_t)stream;
    if (K == 4)
        hipLaunchKernelGGL((ce_fwd_bwd_kernel<4, true>), dim3(grid), dim3(kT), 0, st, logits, nullptr, nullptr, 0.f, prob, N, K, L,
                           grad_scale, dlogits, partial);
    else
        hipLaunchKernelGGL((ce_fwd_bwd_kernel<0, true>), dim3(grid), dim3(kT), 0, st, logits, nullptr, nullptr, 0.f, prob, N, K, L,
                           grad_scale, dlogits, partial);
    return (int)hipGetLastError();
}

int ssecg_seg_confusion(const int64_t* pred, const int64_t* target, int N, int K, int L, int32_t* counts, void* stream) {
    if (!pred || !target || !counts || N <= 0 || K <= 0 || K > kMaxClasses || L <= 0) return SSECG_E_INVAL;
    const size_t lds = (size_t)(kT / 64) * K * K * sizeof(int32_t);
    hipLaunchKernelGGL(seg_confusion_kernel, dim3(N), dim3(kT), lds, (hipStream_t)stream, pred, target, K, L, counts);
    return (int)hipGetLastError();
}

int ssecg_sum_partials(const float* partial, int parts, int width, float scale, float* out, void* stream) {
    if (!partial || !out || parts <= 0 || width <= 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, partial, parts, width, scale, out);
    return (int)hipGetLastError();
}

int ssecg_loss_pair_finish(const float* px, int nx, const float* pu, int nu, float sx_scale, float su_scale, float* out5, void* stream) {
    if (!px || !pu || !out5 || nx <= 0 || nu <= 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(loss_pair_finish_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, px, nx, pu, nu, sx_scale, su_scale, out5);
    return (int)hipGetLastError();
}

int ssecg_adamw_coefficients(double lr, double beta1, double beta2, double weight_decay, int step, double* out5) {
    if (!out5 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return SSECG_E_INVAL;
    // scalars formed in double exactly as torch.optim.AdamW's Python-side arithmetic (rounded once to fp32 by the launch)
    const double bias_correction1 = 1.0 - pow(beta1, (double)step);
    out5[0] = 1.0 - lr * weight_decay;                    // decay_mul
    out5[1] = lr / bias_correction1;                      // step_size
    out5[2] = sqrt(1.0 - pow(beta2, (double)step));       // sqrt(bias_correction2)
    out5[3] = lr;
    out5[4] = (double)step;
    return 0;
}

int ssecg_adamw_multi(const int64_t* table, int ntensors, int64_t max_numel, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int step, const float* skip_flag, float* skipped_count,
                      const double* coef_dev, void* stream) {
    double c[5];
    if (!table || ntensors <= 0 || max_numel <= 0 || ssecg_adamw_coefficients(lr, beta1, beta2, weight_decay, step, c) != 0)
        return SSECG_E_INVAL;
    const int chunks = (int)((max_numel + kChunk - 1) / kChunk);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(chunks, ntensors), dim3(kT), 0, (hipStream_t)stream, table,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)c[0], (float)c[1],
                       (float)c[2], skip_flag, skipped_count, lr, beta1, beta2, step, coef_dev);
    return (int)hipGetLastError();
}

int ssecg_sgd_multi(const int64_t* table, int ntensors, int64_t max_numel, double lr, double momentum, double weight_decay,
                    int first_step, const float* skip_flag, const double* lr_dev, void* stream) {
    if (!table || ntensors <= 0 || max_numel <= 0 || momentum < 0.0) return SSECG_E_INVAL;
    const int chunks = (int)((max_numel + kChunk - 1) / kChunk);
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(chunks, ntensors), dim3(kT), 0, (hipStream_t)stream, table, (float)lr,
                       (float)momentum, (float)weight_decay, first_step, skip_flag, lr_dev);
    return (int)hipGetLastError();
}

size_t ssecg_grad_norm_workspace(int ntensors, int64_t max_numel) {
    if (ntensors <= 0 || max_numel <= 0) return 0;
    const size_t chunks = (size_t)((max_numel + kChunk - 1) / kChunk);
    return chunks * (size_t)ntensors * 2 * sizeof(float);
}

int ssecg_grad_norm_multi(const int64_t* table, int ntensors, int words, int grad_col, int numel_col, int64_t max_numel,
                          float* workspace, size_t workspace_bytes, float* out, float* scaler_state, double growth_factor,
                          double backoff_factor, int growth_interval, void* stream) {
    if (!table || !workspace || !out || ntensors <= 0 || max_numel <= 0 || words <= 0 || grad_col < 0 || grad_col >= words ||
        numel_col < 0 || numel_col >= words)
        return SSECG_E_INVAL;
    if (workspace_bytes < ssecg_grad_norm_workspace(ntensors, max_numel)) return SSECG_E_WORKSPACE;
    const int chunks = (int)((max_numel + kChunk - 1) / kChunk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_sumsq_multi_kernel, dim3(chunks, ntensors), dim3(kT), 0, st, table, words, grad_col, numel_col, workspace);
    hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(1024), 0, st, workspace, chunks * ntensors, out, scaler_state,
                       (float)growth_factor, (float)backoff_factor, growth_interval);
    return (int)hipGetLastError();
}

int ssecg_grad_clip_multi(const int64_t* table, int ntensors, int words, int grad_col, int numel_col, int64_t max_numel,
                          const float* norm, double max_norm, void* stream) {
    if (!table || !norm || ntensors <= 0 || max_numel <= 0 || words <= 0 || grad_col < 0 || grad_col >= words ||
        numel_col < 0 || numel_col >= words || !(max_norm > 0.0))
        return SSECG_E_INVAL;
    const int chunks = (int)((max_numel + kChunk - 1) / kChunk);
    hipLaunchKernelGGL(grad_clip_multi_kernel, dim3(chunks, ntensors), dim3(kT), 0, (hipStream_t)stream, table, words, grad_col,
                       numel_col, norm, (float)max_norm);
    return (int)hipGetLastError();
}

int ssecg_ema_multi(const int64_t* table, int ntensors, int64_t max_numel, double decay, void* stream) {
    if (!table || ntensors <= 0 || max_numel <= 0) return SSECG_E_INVAL;
    const int chunks = (int)((max_numel + kChunk - 1) / kChunk);
    const float one_minus = (float)(1.0 - decay);
    hipLaunchKernelGGL(ema_multi_kernel, dim3(chunks, ntensors), dim3(kT), 0, (hipStream_t)stream, table, (float)decay, one_minus);
    return (int)hipGetLastError();
}

int ssecg_pack_scaled_multi(const int64_t* table, int ntensors, int64_t max_numel, float* dst, double scale, void* stream) {
    if (!table || !dst || ntensors <= 0 || max_numel <= 0) return SSECG_E_INVAL;
    const int chunks = (int)((max_numel + kChunk - 1) / kChunk);
    hipLaunchKernelGGL(pack_scaled_multi_kernel, dim3(chunks, ntensors), dim3(kT), 0, (hipStream_t)stream, table, dst, (float)scale);
    return (int)hipGetLastError();
}

}  // extern "C"
