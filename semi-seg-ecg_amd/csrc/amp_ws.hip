// Weights-stationary bf16 convolution for the 3-tap / 1-tap stride-1 convolutions of the student pass and their data
// gradients (reference: nn.Conv1d of src/models/backbones/resnet.py:55-72 under torch.cuda.amp.autocast,
// src/algorithms/fixmatch.py:97).
//
// Why (profiles/r03_ablation_b16s1.txt re-read in round 4): every ResNet stage moves the same 65.5 MB in and 65.5 MB out at
// N = 1024 windows, so layers 1-3 are HBM-bound (floor 16.4 us at 8 TB/s) - but the ring kernel of amp.hip re-streamed the
// weight operand of every K stage from L2 for every position tile: 24 KB of weights next to 20 KB of activations per stage, all
// through the same per-CU load path (~25 GB/s per CU from HBM, ~70 from L2: MI355X_MICROARCH.md, 'Indexed rows').  Here a wave
// keeps ITS 32 output rows x the whole contraction (Csrc x taps <= 768) in registers for the life of the workgroup:
//   weights: CK*KS k-steps x 4 VGPRs <= 192 registers per lane, loaded once (one wave per SIMD: 512 registers per lane);
//   activations: the only stream - LDS-DMA (global_load_lds_dwordx4) into a ring of R stages of SB channel blocks x (PT + 2)
//     positions, requested AHEAD = R - 1 stages (~100 KB per CU) before they are multiplied, across tile boundaries;
//   MFMA: D[32 ch][128 pos] per wave = 4 accumulator tiles; one ds_read_b128 B fragment per v_mfma_f32_32x32x16_bf16
//     (128 B/clk per CU, half the LDS rate); a tap shift is a +-16-byte LDS offset, a tap that would leave its sample reads the
//     row's zero slot (per-lane offset chosen once per tile);
//   output: a tile's rounded results are held in 32 registers and stored DURING the next tile's stages (2-4 16-byte stores per
//     stage, lane halves exchanged by v_permlane32_swap so every store is a whole 8-channel vector) - no store burst, and
//     every stage issues the same number of vector-memory operations, which makes the counted s_waitcnt vmcnt(N) exact.
// Workgroup = 4 waves: MW channel groups of 32 rows x PW = 4 / MW position groups of 128 (M = 64: 2 x 2, M % 128 == 0: 4 x 1);
// workgroups are persistent (one per CU), the MG = M / (32 MW) channel groups of one position tile run on one XCD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "amp_common.h"
#include "ssecg.h"

namespace ssecg_amp {

__device__ const u32x4 g_ws_zero16 = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void divmod_pos(int sp, int L, unsigned magic, int& n, int& l) {
    // 0 <= sp < 2^24 (launcher-checked), magic = min(floor(2^32 / L) + 1, 2^32 - 1): the estimate is n or n + 1 (n - 1 for L = 1)
    n = (int)__umulhi((unsigned)sp, magic);
    l = sp - n * L;
    const int up = l >= L ? 1 : 0;
    n += up; l -= up * L;
    const int dn = l < 0 ? 1 : 0;
    n -= dn; l += dn * L;
}

// The two vector-memory operations of the tile loop, from inline asm WITHOUT a memory clobber: each is exactly one operation of
// the hand-counted vmcnt stream, volatile asm statements keep their order among themselves (and against the per-stage wait, which
// carries the stage's only compiler fence), and the compiler stays free to schedule LDS fragment reads and MFMAs around them.
__device__ __forceinline__ void buffer_store16(u32x4 v, unsigned voff, u32x4 rsrc, unsigned soff) {
    // a lane whose offset is outside the buffer stores nothing (bit 31 set: first tile, positions beyond the tensor)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rsrc), "s"(soff));
}
__device__ __forceinline__ void lds_dma16_nc(const u32x4* gsrc, unsigned lds_byte_addr_wave_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr_wave_uniform));
}

// Diagnostic build (-DSSECG_WS_STAMP, tools/stamp_ws.sh): s_memtime stamps around the segments of a stage, summed per wave in
// scalar registers; wave 0 of every workgroup writes its sums over its statistics row (a timing build: results are not used).
#if defined(SSECG_WS_STAMP)
#define WS_STAMP(t)                                                                                  \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
#else
#define WS_STAMP(t) do { } while (0)
#endif

// GM = input stride (source position of output l, tap t: GM * l + tapoff[t]); KS = taps; WPAD = extra window slots for source
// rows that are LONGER than GM * (output row): every sample boundary inside a tile then widens the tile's source window by
// Lsrc - GM * Ldst positions (the odd-position phase of a stride-2 data gradient: 63 source, 62 output positions).  The
// 3-tap stride-1 same-length instances need none (WPAD 0: the window is exactly PT + 2).
template <int CK, int KS, int GM, int MW, bool STATS>
struct WsCfg {
    static constexpr int WPAD = (GM == 1 && KS == 3) ? 0 : 8;
    static constexpr int PW = 4 / MW, PT = 128 * PW;
    static constexpr int ROWV = GM * (PT - 1) + KS + WPAD + 1;        // slots: the source window of PT outputs + one ZERO slot
    static constexpr int CBS = 2 * CK;                                 // 8-channel blocks of the source
    static constexpr int SB = (PW == 1 && GM == 1) ? 8 : 4;            // blocks per stage (~16.7 KB)
    static constexpr int NSTG = CBS / SB;                              // stages per tile
    static constexpr int SV = SB * ROWV, NPC = (SV + 63) / 64, PPW = (NPC + 3) / 4, SVB = NPC * 64;
    static constexpr int NST = 8;                                      // 16-byte stores per tile and wave
    static constexpr int SPS = NST / NSTG;                             // ... issued per stage
    static constexpr int OPS = PPW + SPS;                              // vector-memory operations per wave and stage, always
    static constexpr int AH0 = 1 + 63 / OPS;                           // vmcnt is a 6-bit counter
#if !defined(WS_AHEAD_MAX)
#define WS_AHEAD_MAX 7                                                 // (tools/ablate_ws.sh varies it)
#endif
    static constexpr int AHEAD = AH0 < WS_AHEAD_MAX ? AH0 : WS_AHEAD_MAX;
    static constexpr int R = AHEAD + 1;
    static constexpr int LDSV = R * SVB + 64;                          // + a 1 KB target for the count-keeping dummy DMAs
    static_assert(NSTG * SB == CBS && (SB & 1) == 0 && NST % NSTG == 0, "stage geometry");
    static_assert((AHEAD - 1) * OPS <= 63 && LDSV * 16 <= 160 * 1024, "ring geometry");
};

template <int CK, int KS, int GM, int MW, bool STATS>
__global__ __launch_bounds__(256, 1) void conv_b16ws_kernel(WsP p) {
    using C = WsCfg<CK, KS, GM, MW, STATS>;
    constexpr int PW = C::PW, PT = C::PT, ROWV = C::ROWV, CBS = C::CBS, SB = C::SB, NSTG = C::NSTG, SV = C::SV, NPC = C::NPC,
                  PPW = C::PPW, SVB = C::SVB, SPS = C::SPS, OPS = C::OPS, AHEAD = C::AHEAD, R = C::R;
    __shared__ u32x4 lds[C::LDSV];               // the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave % MW, wp = wave / MW;
    // workgroup -> (channel group g, position lane pslot): b and b + 8 share an XCD, so the MG groups of a position tile do
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int g = bi % p.MG, pslot = (bi / p.MG) * 8 + xcd;
    const int m0w = g * (32 * MW) + wm * 32;     // this wave's 32 output channels
    const int Ld = p.L, Ls = p.Lsrc, CBo = p.M >> 3;   // output / source row lengths
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned dummy_dst = lds0 + (unsigned)(R * SVB) * 16u;

#if defined(SSECG_WS_STAMP)
    unsigned long long tP0 = 0, tP1 = 0, tP2 = 0, tP3 = 0, rt0 = 0, rt1 = 0;
    WS_STAMP(tP0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
#endif
    // ---- the weights of this wave: k-step s = chunk * KS + tap, lane (r, h) holds W[m0w + r][16 chunk + 8h .. +7][tap]
    bf16x8 a[CK * KS];
#pragma unroll
    for (int s = 0; s < CK * KS; ++s) a[s] = __builtin_bit_cast(bf16x8, p.W[(size_t)(s * 2 + h) * p.M + m0w + r]);
    // (these loads are waited for behind the prologue's requests, below: their latency overlaps the ring's fill)

#if defined(SSECG_WS_STAMP)
    WS_STAMP(tP1);
#endif
    float st_s[STATS ? 16 : 1], st_q[STATS ? 16 : 1];
    if (STATS) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { st_s[e] = 0.f; st_q[e] = 0.f; }
    }

    // ---- DMA pieces of this wave: piece k = wave + 4i covers stage vectors 64k .. 64k+63 (vector v = row * ROWV + slot).
    // The source pointer of a piece (stage 0 of a position tile) needs a division by L: computed ONCE per target tile
    // (`piece_base`, when the stream of requests enters a new tile); later stages of that tile add SB * L vectors each.
    int prow[PPW], pslt[PPW];
    bool plane[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int v = (wave + 4 * i) * 64 + lane;
        prow[i] = v / ROWV;
        pslt[i] = v - prow[i] * ROWV;
        plane[i] = v < SV && pslt[i] < ROWV - 1 && wave + 4 * i < NPC;   // the row's last slot is its ZERO slot: loaded from the zero constant
    }
    // flattened source position of slot 0 of the tile whose first output is flattened position P0 (wave-uniform)
    auto window_start = [&](int P0, bool ok) -> int {
        int n0, l0;
        divmod_pos(ok ? P0 : 0, Ld, p.magic, n0, l0);
        return __builtin_amdgcn_readfirstlane(n0 * Ls + GM * l0 + p.tmin);
    };
    const u32x4* pb[PPW];   // stage-0 source of each piece of the tile the request stream is in (zero constant: nothing to load)
    bool pbz[PPW];
    auto piece_base = [&](int pt2) {
        const bool tile_ok = pt2 < p.numPT;
        const int S0 = window_start(pt2 * PT, tile_ok);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int sp = S0 + pslt[i];                     // flattened SOURCE position of this lane's slot
            const bool ok = tile_ok && plane[i] && (unsigned)sp < (unsigned)p.Psrc;
            int n, l;
            divmod_pos(ok ? sp : 0, Ls, p.magic_src, n, l);
#if defined(SSECG_ABLW_NODMA)     // timing experiment: every piece loads the zero constant (same instructions, no HBM reads)
            pb[i] = &g_ws_zero16; pbz[i] = true; (void)n; (void)l;
#else
            pb[i] = ok ? p.src + (size_t)((n * CBS + prow[i]) * Ls + l) : &g_ws_zero16;
            pbz[i] = !ok;
#endif
        }
    };
    // piece i of stage sg2 of that tile into ring slot `slot`; exactly one vector-memory operation, whatever the lane loads
    auto issue_piece = [&](int i, int sg2, int slot) {
        const int k = wave + 4 * i;
        const u32x4* src = pbz[i] ? pb[i] : pb[i] + (size_t)(sg2 * SB) * Ls;
        // (a wave with fewer than PPW real pieces sends zeros to the dummy target: the per-stage operation count stays uniform)
        const unsigned dst = k < NPC ? lds0 + (unsigned)(slot * SVB + k * 64) * 16u : dummy_dst;
        lds_dma16_nc(src, __builtin_amdgcn_readfirstlane(dst));
    };

    u32x4 rsrc;   // buffer descriptor of the output: base, stride 0, bytes, raw 32-bit data format
    {
        const uint64_t ob = (uint64_t)(uintptr_t)p.out;
        rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)ob);
        rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(ob >> 32) & 0xffffu);
        rsrc.z = __builtin_amdgcn_readfirstlane(p.out_bytes);
        rsrc.w = 0x00020000u;
    }

    // ---- prologue: the first AHEAD stages of this workgroup's stream (PPW pieces each; no store slots yet)
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) {
        if (s % NSTG == 0) piece_base(pslot + (s / NSTG) * p.rows);
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(i, s % NSTG, s);
    }
    // The compiler must finish ITS wait-count bookkeeping for the weight loads here: a wait of its own inside the tile loop would
    // count the hand-issued operations of the ring as well and drain them.  An opaque use forces its waits to this point; they
    // over-wait (the counter holds the prologue's requests too, all younger than the weight loads) - harmless, once per kernel.
#pragma unroll
    for (int s = 0; s < CK * KS; ++s) asm volatile("" : "+v"(a[s]));

    u32x4 held[8];          // the previous tile's rounded outputs, one 16-byte vector per (position tile j, block pair)
    unsigned held_off[4];   // byte offsets; bit 31 set = outside the buffer = nothing stored (first tile, positions beyond P)
#pragma unroll
    for (int i = 0; i < 8; ++i) held[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) held_off[j] = 0x80000000u;
    int slot_cur = 0;       // ring slot of the current stage
    int gstage = 0;         // stages this workgroup has consumed
    auto wait_warmup = [&](int k) {   // k = groups with store slots inside the window, k < AHEAD - 1 (wave-uniform)
        static_assert(AHEAD <= 7, "wait_warmup enumerates k = 0 .. 5");
        switch (k) {
            case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PPW + 0 * SPS) : "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PPW + 1 * SPS) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PPW + 2 * SPS) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PPW + 3 * SPS) : "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PPW + 4 * SPS) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PPW + 5 * SPS) : "memory"); break;
        }
    };
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, sum_wait = 0, sum_bar = 0, sum_body = 0, sum_setup = 0, sum_epi = 0, tK0 = 0, tK1 = 0;
    (void)tA; (void)tB; (void)tC; (void)tD; (void)tE; (void)tK0; (void)tK1; (void)sum_wait; (void)sum_bar; (void)sum_body; (void)sum_setup; (void)sum_epi;
    WS_STAMP(tK0);

    for (int pt = pslot; pt < p.numPT; pt += p.rows) {
        WS_STAMP(tE);
        const int P0 = pt * PT;
        const int S0 = window_start(P0, true);
        int xoff[4][KS];    // B fragment offset (vectors from the stage base, chunk 0) per position tile and tap
        unsigned cur_off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = P0 + wp * 128 + j * 32 + r;
            const bool pok = pos < p.P;
            int n, l;
            divmod_pos(pok ? pos : 0, Ld, p.magic, n, l);
            // lanes h = 0 store the even block of a pair, lanes h = 1 the odd one; output l sits at l * ostride + ooff of its row
#if defined(SSECG_ABLW_NOSTORE)   // timing experiment: every store is out of range (issued and counted, nothing written)
            cur_off[j] = 0x80000000u;
#else
            cur_off[j] = pok ? (unsigned)(((n * CBo + (m0w >> 3) + h) * p.Lrow + l * p.ostride + p.ooff)) * 16u : 0x80000000u;
#endif
            const int sbase = n * Ls + GM * l - S0;          // slot of tap offset 0
#pragma unroll
            for (int t = 0; t < KS; ++t)   // a tap outside its sample reads the row's zero slot
                xoff[j][t] = h * ROWV + ((pok && (unsigned)(GM * l + p.tapoff[t]) < (unsigned)Ls) ? sbase + p.tapoff[t] : ROWV - 1);
        }
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll
        for (int sg = 0; sg < NSTG; ++sg) {
            WS_STAMP(tA);
            if (sg == 0) sum_setup += tA - tE;
            // this stage's pieces were issued AHEAD groups ago: AHEAD - 1 whole groups are younger - OPS operations each, except
            // the prologue's groups (PPW each), which the first AHEAD - 1 stages of a workgroup still have in their window
            if (gstage >= AHEAD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * OPS) : "memory");
            else wait_warmup(gstage);
            ++gstage;
            WS_STAMP(tB);
            __builtin_amdgcn_s_barrier();   // every wave's pieces have landed; the previous stage's readers are done
            asm volatile("" ::: "memory");  // (the compiler's only fence of the stage: LDS reads stay below the barrier)
            WS_STAMP(tC);
            // Issue group of this stage = SPS stores of the previous tile's results + the PPW pieces of stage + AHEAD into the slot
            // the barrier has just freed, spread over the MFMA groups (the asm statements carry no memory clobber, so fragment
            // reads and MFMAs are scheduled around them; they keep their own order, and all precede the next stage's wait).
            const int tsg = (sg + AHEAD) % NSTG;
            const int tslot = slot_cur == 0 ? R - 1 : slot_cur - 1;
            if (tsg == 0) piece_base(pt + ((sg + AHEAD) / NSTG) * p.rows);
            const u32x4* sb = lds + slot_cur * SVB;
            constexpr int NG = (SB / 2) * KS;   // MFMA groups of the stage: (16-channel chunk, tap) x 4 position tiles
            // B fragments of group gi + FD are read from LDS before the MFMAs of group gi are issued (FD + 1 register sets; left to
            // itself hipcc funnels every fragment through ONE set: ds_read -> lgkmcnt(0) -> MFMA, the whole LDS latency per MFMA)
            constexpr int FD = 2;
            u32x4 fb[FD + 1][4];
            auto frag = [&](int gi, int set) {
                const int cc = gi / KS, t = gi - cc * KS;
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[set][j] = (sb + xoff[j][t])[cc * 2 * ROWV];
            };
#pragma unroll
            for (int gi = 0; gi < FD && gi < NG; ++gi) frag(gi, gi % (FD + 1));
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (gi + FD < NG) frag(gi + FD, (gi + FD) % (FD + 1));
#pragma unroll
                for (int o = 0; o < OPS; ++o)
                    if (o * NG / OPS == gi || (gi == NG - 1 && o * NG / OPS >= NG)) {
                        if (o < SPS) {
                            const int idx = sg * SPS + o;
                            buffer_store16(held[idx], held_off[idx >> 1], rsrc, (unsigned)((idx & 1) * 2 * p.Lrow) * 16u);
                        } else {
                            issue_piece(o - SPS, tsg, tslot);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);   // keep the reads of group gi + FD ahead of the MFMAs of group gi
                const int s = sg * NG + gi;           // k-step = (chunk, tap) in operand order
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, fb[gi % (FD + 1)][j]);
#if defined(SSECG_ABLW_NOMFMA)    // timing experiment: fragments read, nothing multiplied
                    asm volatile("" :: "v"(b), "v"(a[s]));
#else
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b, acc[j], 0, 0, 0);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            slot_cur = slot_cur + 1 == R ? 0 : slot_cur + 1;
            WS_STAMP(tD);
            sum_wait += tB - tA; sum_bar += tC - tB; sum_body += tD - tC;
        }

        // tile end: register 4q+e of tile j = channel m0w + 8q + 4h + e at position j*32 + r.  Round, take the statistics of
        // what will be stored, exchange lane halves so that a lane holds a whole 8-channel vector, hold until the next tile.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                unsigned pk[2][2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int q = 2 * qq + b;
                    pk[b][0] = pack2(acc[j][4 * q + 0], acc[j][4 * q + 1]);
                    pk[b][1] = pack2(acc[j][4 * q + 2], acc[j][4 * q + 3]);
                    if (STATS) {
                        const float w0 = bf_lo(pk[b][0]), w1 = bf_hi(pk[b][0]), w2 = bf_lo(pk[b][1]), w3 = bf_hi(pk[b][1]);
                        st_s[4 * q + 0] += w0; st_q[4 * q + 0] = fmaf(w0, w0, st_q[4 * q + 0]);
                        st_s[4 * q + 1] += w1; st_q[4 * q + 1] = fmaf(w1, w1, st_q[4 * q + 1]);
                        st_s[4 * q + 2] += w2; st_q[4 * q + 2] = fmaf(w2, w2, st_q[4 * q + 2]);
                        st_s[4 * q + 3] += w3; st_q[4 * q + 3] = fmaf(w3, w3, st_q[4 * q + 3]);
                    }
                }
                // lanes 0-31 end up with block 2qq (their own channels 0-3 + the upper half's channels 4-7), lanes 32-63 with
                // block 2qq+1 (cdna_hip_programming.md T21)
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                u32x4 v;
                v.x = s0[0]; v.y = s1[0]; v.z = s0[1]; v.w = s1[1];
                held[j * 2 + qq] = v;
            }
            held_off[j] = cur_off[j];
        }
        WS_STAMP(tA);
        sum_epi += tA - tD;
    }
    WS_STAMP(tK1);

#pragma unroll
    for (int idx = 0; idx < 8; ++idx) buffer_store16(held[idx], held_off[idx >> 1], rsrc, (unsigned)((idx & 1) * 2 * p.Lrow) * 16u);
#if defined(SSECG_WS_STAMP)
    WS_STAMP(tP2);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this wave is in flight into the ring any more
#if defined(SSECG_WS_STAMP)
    WS_STAMP(tP3);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
#endif

#if defined(SSECG_WS_STAMP)
    if (STATS) {
        __syncthreads();
        if (tid == 0) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(p.stats + (size_t)pslot * p.M * 2 + (size_t)g * 64);
            d[0] = tK1 - tK0; d[1] = sum_wait; d[2] = sum_bar; d[3] = sum_body; d[4] = sum_setup; d[5] = sum_epi;
            d[6] = tP1 - tP0; d[7] = tK0 - tP1; d[8] = tP2 - tK1; d[9] = tP3 - tP2; d[10] = tP3 - tP0; d[11] = rt1 - rt0;
        }
        return;
    }
#endif
    if (STATS) {
        // Lane (r, h) holds 32 partial sums {sum, sum of squares} x 16 channels (8q + 4h + e) over ITS positions r, r + 32, ...:
        // the sum over the 32 lanes r goes through LDS, transposed (a butterfly of 160 dependent cross-lane shuffles took ~4 us
        // at one wave per SIMD): each wave writes [32 values][64 lanes + 1 pad], then lane x = (value v = x & 31, half h' = x >> 5)
        // adds the 32 lanes of its half in a fixed order; position groups (PW) are combined by the first 32 MW threads.
        float* red = reinterpret_cast<float*>(lds);   // [4 waves][32 values][65] floats = 33 KB, then [PW][32 MW channels][2]
        __syncthreads();                              // every wave is done with the ring
        float* mine = red + wave * (32 * 65);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            mine[(2 * e + 0) * 65 + lane] = st_s[e];
            mine[(2 * e + 1) * 65 + lane] = st_q[e];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): this wave's own LDS writes (a wave reads only its own block)
        const int v = lane & 31, hh = lane >> 5;
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tot += mine[v * 65 + hh * 32 + k];
        __syncthreads();                              // all blocks consumed: the buffer is reused for the per-workgroup table
        float* tab = red;                             // [PW][32 MW][2]
        {
            const int e = v >> 1, sq = v & 1;
            const int ch = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            tab[(wp * 32 * MW + ch) * 2 + sq] = tot;
        }
        __syncthreads();
        if (tid < 32 * MW) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < PW; ++w) { s += tab[(w * 32 * MW + tid) * 2]; q += tab[(w * 32 * MW + tid) * 2 + 1]; }
            float* dst = p.stats + ((size_t)pslot * p.M + g * (32 * MW) + tid) * 2;
            dst[0] = s; dst[1] = q;
        }
    }
}

static int ws_geometry(int N, int L, int M, int* MGo, int* rowso, int* numPTo, int* gridO) {
    const int MW = (M % 128 == 0) ? 4 : 2;
    const int MG = M / (32 * MW);
    const int PT = 128 * (4 / MW);
    const long long P = (long long)N * L;
    const int numPT = (int)((P + PT - 1) / PT);
    int lanes = (kNumCU / (8 * MG)) * 8;          // position lanes: a multiple of 8 (XCD round-robin)
    if (lanes < 8) lanes = 8;
    const int need = ((numPT + 7) / 8) * 8;
    if (lanes > need) lanes = need;
    *MGo = MG; *rowso = lanes; *numPTo = numPT; *gridO = lanes * MG;
    return MW;
}

static unsigned magic_for(int L) {
    const unsigned long long m = (1ull << 32) / (unsigned long long)L + 1ull;
    return m > 0xffffffffull ? 0xffffffffu : (unsigned)m;
}

// template instances that exist: (source channels / 16, taps, input stride, waves along the channel axis)
static bool ws_instance(int CK, int KS, int GM, int MW) {
    if (KS == 3 && GM == 1) return CK == 4 || CK == 8 || CK == 16;                          // stride-1 body convs, both wave layouts
    if (KS == 3 && GM == 2) return MW == 4 && (CK == 4 || CK == 8 || CK == 16);             // stride-2 first conv of a stage
    if (KS == 1 && GM == 2) return MW == 4 && (CK == 4 || CK == 8 || CK == 16);             // 1x1 stride-2 downsample
    if (KS == 1 && GM == 1) return (MW == 2 && CK == 8) || (MW == 4 && (CK == 16 || CK == 32));   // its data gradient (and the even phase below)
    if (KS == 2 && GM == 1) return (MW == 2 && CK == 8) || (MW == 4 && (CK == 16 || CK == 32));   // odd-position phase of a 3-tap stride-2 data gradient
    return false;
}

int ws_rows(int N, int Csrc, int Lsrc, int M, int Ldst, int ntaps, int gmul, int tapoff0, int tapoff1, int tapoff2, int Lrow,
            int ostride, int ooff, bool accumulate, bool want_stats) {
    // SSECG_AMP_WS: 0 = never, 1 = wherever the kernel applies, unset = where it measured faster than the kernels of amp.hip
    // (profiles/r04_ws_conv_bench.txt: every forward; plain 3-tap stride-1 data gradients only at 256 source channels - without
    // the statistics epilogue the 8-wave ring kernel hides its issue stalls better at 64 / 128 channels)
    const char* ev = getenv("SSECG_AMP_WS");   // read per call (two calls per convolution): tests switch it at run time
    const int mode = ev ? atoi(ev) : -1;
    if (mode == 0 || accumulate) return 0;
    if (ntaps < 1 || ntaps > 3 || (gmul != 1 && gmul != 2) || M % 64 != 0 || M > 2048 || (Csrc & 15)) return 0;
    const int MW = (M % 128 == 0) ? 4 : 2;
    if (!ws_instance(Csrc / 16, ntaps, gmul, MW)) return 0;
    if (mode != 1 && !want_stats && ntaps == 3 && gmul == 1 && Csrc < 256) return 0;
    const int taps[3] = {tapoff0, tapoff1, tapoff2};
    int tmin = taps[0], tmax = taps[0];
    for (int t = 1; t < ntaps; ++t) { tmin = taps[t] < tmin ? taps[t] : tmin; tmax = taps[t] > tmax ? taps[t] : tmax; }
    if (tmax - tmin > ntaps - 1) return 0;                                 // the window holds GM * (PT - 1) + KS positions
    if (ostride < 1 || ooff < 0 || (long long)(Ldst - 1) * ostride + ooff >= Lrow) return 0;
    // every output's taps lie inside [-(KS), Lsrc + KS): anything else is not one of this network's convolutions
    if (tmin < -3 || (long long)gmul * (Ldst - 1) + tmax > Lsrc + 2) return 0;
    // the source rows may be longer than gmul * (output row): each sample boundary inside a tile widens its window by the
    // difference; the instances reserve 8 slots (none for the 3-tap stride-1 form, which needs Lsrc == Ldst)
    const int PT = 128 * (4 / MW);
    const long long slack = (long long)Lsrc - (long long)gmul * Ldst;
    const int wpad = (gmul == 1 && ntaps == 3) ? 0 : 8;
    if (slack > 0 && ((PT + Ldst - 1) / Ldst) * slack > wpad) return 0;
    if (gmul == 1 && ntaps == 3 && (Lsrc != Ldst || Lrow != Ldst || ostride != 1)) return 0;
    const long long P = (long long)N * Ldst, Ps = (long long)N * Lsrc;
    if (P >= (1ll << 24) || Ps + 1024 >= (1ll << 24)) return 0;           // multiply-high division is exact below 2^24 (divmod_pos)
    if ((long long)N * M * Lrow * 2 >= (1ll << 31)) return 0;             // 32-bit buffer offsets, bit 31 = "no store"
    if ((long long)N * Csrc * Lsrc * 2 >= (1ll << 31)) return 0;
    int MG, rows, numPT, grid;
    ws_geometry(N, Ldst, M, &MG, &rows, &numPT, &grid);
    return rows;
}

int ws_launch(const void* src, const void* w_operand, void* out, int N, int Csrc, int Lsrc, int M, int Ldst, int ntaps, int gmul,
              int tapoff0, int tapoff1, int tapoff2, int Lrow, int ostride, int ooff, float* stats, hipStream_t st) {
    WsP p;
    p.W = (const u32x4*)w_operand; p.src = (const u32x4*)src; p.out = (u32x4*)out; p.stats = stats;
    p.N = N; p.M = M; p.Csrc = Csrc; p.L = Ldst; p.Lsrc = Lsrc; p.Lrow = Lrow; p.ostride = ostride; p.ooff = ooff;
    p.P = N * Ldst; p.Psrc = N * Lsrc;
    p.tapoff[0] = tapoff0; p.tapoff[1] = ntaps > 1 ? tapoff1 : tapoff0; p.tapoff[2] = ntaps > 2 ? tapoff2 : tapoff0;
    p.tmin = p.tapoff[0];
    for (int t = 1; t < ntaps; ++t) p.tmin = p.tapoff[t] < p.tmin ? p.tapoff[t] : p.tmin;
    int grid;
    const int MW = ws_geometry(N, Ldst, M, &p.MG, &p.rows, &p.numPT, &grid);
    p.magic = magic_for(Ldst); p.magic_src = magic_for(Lsrc);
    p.out_bytes = (unsigned)((long long)N * M * Lrow * 2);
    dim3 gd(grid), bk(256);
    const int CK = Csrc / 16;
    const bool S = stats != nullptr;
#define SSECG_WS(CK_, KS_, GM_, MW_)                                                                                          \
    if (CK == CK_ && ntaps == KS_ && gmul == GM_ && MW == MW_) {                                                              \
        if (S) hipLaunchKernelGGL((conv_b16ws_kernel<CK_, KS_, GM_, MW_, true>), gd, bk, 0, st, p);                            \
        else hipLaunchKernelGGL((conv_b16ws_kernel<CK_, KS_, GM_, MW_, false>), gd, bk, 0, st, p);                             \
        return (int)hipGetLastError();                                                                                        \
    }
    SSECG_WS(4, 3, 1, 4) SSECG_WS(8, 3, 1, 4) SSECG_WS(16, 3, 1, 4) SSECG_WS(4, 3, 1, 2) SSECG_WS(8, 3, 1, 2) SSECG_WS(16, 3, 1, 2)
    SSECG_WS(4, 3, 2, 4) SSECG_WS(8, 3, 2, 4) SSECG_WS(16, 3, 2, 4)
    SSECG_WS(4, 1, 2, 4) SSECG_WS(8, 1, 2, 4) SSECG_WS(16, 1, 2, 4)
    SSECG_WS(8, 1, 1, 2) SSECG_WS(16, 1, 1, 4) SSECG_WS(32, 1, 1, 4)
    SSECG_WS(8, 2, 1, 2) SSECG_WS(16, 2, 1, 4) SSECG_WS(32, 2, 1, 4)
#undef SSECG_WS
    return SSECG_E_INVAL;   // ws_rows() said yes: unreachable
}

}  // namespace ssecg_amp
