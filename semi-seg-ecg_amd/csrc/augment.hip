// On-device record pipeline of the unlabelled loader: RandAugment over {AmplitudeScaling,
// AdaptivePowerlineNoise, RandomPartialWhiteNoise, RandomPartialSineNoise} followed by per-record standardisation
// (src/utils/transforms.py:290-310, 340-351, 480-562, 628-657; call order src/utils/semi_dataset.py:235-244).
//
// HBM-bound byte work: one workgroup owns one (record, lead) row, keeps it in LDS as fp64 (the reference pipeline is
// float64 until ToTensor), applies the record's planned ops in order - the 5th/95th percentiles AdaptivePowerlineNoise
// needs come from an in-LDS bitonic sort of a copy of the row - and writes the row once.  The random DECISIONS arrive
// as a per-record plan (ssecg.h); the two noise fields are either given (parity tests replay the reference's draws) or
// generated in the kernel from a counter-based generator (splitmix64 -> Box-Muller, same law as ssecg/synth.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "ssecg.h"

namespace {

constexpr int kT = 256;
constexpr int kPlanW = SSECG_AUG_PLAN_WIDTH;
constexpr double kTwoPi = 2.0 * 3.141592653589793;   // numpy: 2 * np.pi

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ssecg/synth.py::_key / uniform / normal
__device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint64_t stream) {
    return splitmix64(splitmix64(seed) ^ (stream * 0xD1342543DE82EF95ull));
}
__device__ __forceinline__ double u01(uint64_t key, uint64_t i) {
    const uint64_t bits = splitmix64(i * 0x2545F4914F6CDD1Dull + key) >> 11;
    return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double std_normal(uint64_t k1, uint64_t k2, uint64_t i) {
    return sqrt(-2.0 * log(u01(k1, i))) * cos(kTwoPi * u01(k2, i));
}

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // red may still be read from the previous call
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < kT / 64; ++i) s += red[i];
    return s;
}

// np.percentile(.., method="linear") on the sorted row: virtual index (n-1)*q, numpy's _lerp
__device__ __forceinline__ double percentile_sorted(const double* a, int n, double q) {
    const double h = (double)(n - 1) * q;
    const int lo = (int)floor(h);
    const int hi = lo + 1 < n ? lo + 1 : n - 1;
    const double t = h - (double)lo, d = a[hi] - a[lo];
    return t < 0.5 ? a[lo] + d * t : a[hi] - d * (1.0 - t);
}

struct AugP {
    const float* x;
    float* y;
    const int32_t* plan;
    const float* scales;
    const float* white;
    int C, L, Lp2;
    double sigma, fs, amplitude, sine_freq;
    uint64_t seed;
};

__global__ __launch_bounds__(kT) void strong_augment_rows_kernel(AugP p) {
    extern __shared__ double sm[];
    double* row = sm;            // [L]
    double* srt = sm + p.L;      // [Lp2]
    const int r = blockIdx.x, b = r / p.C;
    const int32_t* pl = p.plan + (size_t)b * kPlanW;
    const size_t base = (size_t)r * p.L;
    for (int l = threadIdx.x; l < p.L; l += kT) row[l] = (double)p.x[base + l];
    __syncthreads();
    const int layers = pl[10], applied = pl[4];
    for (int k = 0; k < layers; ++k) {
        if (!((applied >> k) & 1)) continue;   // RandomApply did not fire
        const int op = pl[k];                  // uniform over the workgroup
        if (op == SSECG_AUG_AMPLITUDE_SCALING) {
            const uint64_t k1 = stream_key(p.seed, 2), k2 = stream_key(p.seed, 3);   // synth.normal(seed, stream=1)
            for (int l = threadIdx.x; l < p.L; l += kT) {
                const double s = p.scales ? (double)p.scales[base + l] : 1.0 + p.sigma * std_normal(k1, k2, base + l);
                row[l] *= s;
            }
        } else if (op == SSECG_AUG_POWERLINE) {
            for (int l = threadIdx.x; l < p.Lp2; l += kT) srt[l] = l < p.L ? row[l] : INFINITY;
            __syncthreads();
            for (int kk = 2; kk <= p.Lp2; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    for (int i = threadIdx.x; i < p.Lp2; i += kT) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const double a = srt[i], c = srt[ixj];
                            if ((a > c) == ((i & kk) == 0)) { srt[i] = c; srt[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            const double amp = (percentile_sorted(srt, p.L, 0.95) - percentile_sorted(srt, p.L, 0.05)) / 2.0;
            const double w = kTwoPi * (double)pl[5];
            for (int l = threadIdx.x; l < p.L; l += kT) row[l] += amp * sin(w * ((double)l / p.fs));
        } else if (op == SSECG_AUG_PARTIAL_WHITE) {
            const int cnt = pl[6], st = pl[7];
            const uint64_t k1 = stream_key(p.seed, 4), k2 = stream_key(p.seed, 5);   // synth.normal(seed, stream=2)
            for (int j = threadIdx.x; j < cnt; j += kT) {
                const double w = p.white ? (double)p.white[base + j] : std_normal(k1, k2, base + j);
                row[st + j] += p.amplitude * w;
            }
        } else if (op == SSECG_AUG_PARTIAL_SINE) {
            const int cnt = pl[8], st = pl[9];
            for (int j = threadIdx.x; j < cnt; j += kT)
                row[st + j] += p.amplitude * sin(kTwoPi * ((double)j / (double)p.L) / p.sine_freq);
        }
        __syncthreads();
    }
    for (int l = threadIdx.x; l < p.L; l += kT) p.y[base + l] = (float)row[l];
}

// y = (x - mean) / std per record over n = C*L elements (population std, two passes in fp64 like numpy); 0 if std == 0
__global__ __launch_bounds__(kT) void standardize_kernel(const float* x, float* y, int n) {
    __shared__ double red[kT / 64];
    const float* xr = x + (size_t)blockIdx.x * n;
    float* yr = y + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kT) s += (double)xr[i];
    const double mean = block_sum(s, red) / (double)n;
    double q = 0.0;
    for (int i = threadIdx.x; i < n; i += kT) {
        const double d = (double)xr[i] - mean;
        q += d * d;
    }
    const double sd = sqrt(block_sum(q, red) / (double)n);
    for (int i = threadIdx.x; i < n; i += kT) yr[i] = sd != 0.0 ? (float)(((double)xr[i] - mean) / sd) : 0.f;
}

}  // namespace

extern "C" {

int ssecg_strong_augment(const float* x, float* y, const int32_t* plan, const float* scales, const float* white, int B,
                         int C, int L, double sigma, double fs, double amplitude, double sine_freq, uint64_t seed,
                         void* stream) {
    if (!x || !y || !plan || B <= 0 || C <= 0 || L <= 0 || L > 4096 || !(fs > 0.0) || !(sine_freq > 0.0)) return SSECG_E_INVAL;
    int lp2 = 1;
    while (lp2 < L) lp2 <<= 1;
    AugP p{x, y, plan, scales, white, C, L, lp2, sigma, fs, amplitude, sine_freq, seed};
    const size_t lds = (size_t)(L + lp2) * sizeof(double);   // <= 64 KiB
    hipLaunchKernelGGL(strong_augment_rows_kernel, dim3(B * C), dim3(kT), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

int ssecg_standardize(const float* x, float* y, int B, int n, void* stream) {
    if (!x || !y || B <= 0 || n <= 0) return SSECG_E_INVAL;
    hipLaunchKernelGGL(standardize_kernel, dim3(B), dim3(kT), 0, (hipStream_t)stream, x, y, n);
    return (int)hipGetLastError();
}

}  // extern "C"
