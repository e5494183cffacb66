// vt_igemm_span5.hip -- persistent input-span convolution (bf16) with a dedicated LDS-DMA loader wave, for the
// MFMA-bound stride-1 3x3 layers: every ConvNormAct 3x3 stride-1 forward conv of the Darknet / CSPDarknet /
// VoVNet stages (reference components.py:26-35, darknet.py:23-24, vovnet.py:41-44) and their stride-1 data
// gradients.
//
// Why: ablation builds of vt_igemm_span.hip (tools/exp_span.sh, 128->128 3x3 @28x28 and a 4x longer K, batch chosen
// for exactly two rounds) show that a step costs the SUM of its parts, not their maximum:
//     scalar bookkeeping + barrier 0.23 us  +  LDS-DMA issue 0.31 us  +  MFMAs 0.40 us  =  0.94 us per step.
// The DMA term is the per-CU global->LDS path running at its limit (20 KiB per step and CU in 0.31 us = 67 GB/s
// per CU): every wave issues its share right after the barrier and stays blocked until the path has accepted it,
// so no wave feeds the matrix pipe meanwhile.  Here the waves that compute never touch vector memory:
//   * a workgroup = 4 compute waves (2 x 2 over a 32*FM-row x 128-column tile, FM <= 7) + 1 LOADER wave;
//     two workgroups per CU.  The loader issues every LDS-DMA (input span pieces, filter slices), keeps three
//     filter slices and the next chunk's span in flight, retires them with exact counted vmcnt waits and is the
//     only wave that ever blocks on the memory path; it also builds the next tile's row tables.
//   * one workgroup barrier per (chunk, tap) step carries both hand-offs: before it the loader has waited for
//     the step's slice (and everything older), the compute waves hold the previous step's fragments in registers
//     (their slot may be overwritten after it).
//   * PERSISTENT: each workgroup owns a contiguous range of 32-row units of one XCD's share of the flat pixel
//     index and cuts it into tiles of 4..7 units, so every CU gets the same number of rows (no 1.75-round tail);
//     the loader does not stop at tile boundaries, so the next tile's first span and slices land while the
//     compute waves store the current tile.
//   * swapped MFMA operands (filter rows = MFMA rows, pixels = MFMA columns): a lane ends up with 2 x 8
//     CONSECUTIVE output channels of one pixel and stores them straight from the accumulators (no LDS staging).
// LDS images, swizzles and the summation order are those of vt_igemm_span.hip: outputs are bit-identical to it.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

constexpr int kFMX = 6;           // row fragments (16 rows) per compute wave, at most
constexpr int kNSB = 4;           // filter-slice ring slots (3 slices in flight)
constexpr int kBSlot = 128 * 64;  // bytes per filter slice: 128 filter rows x 32 channels
constexpr int kBMX = 32 * kFMX;   // rows of the tallest tile

// dev diagnostics (VT_SPAN5_ABL bit 16): wall-clock stamps (100 MHz) per workgroup; never read by the kernel itself
__device__ unsigned long long vt_span5_stamps[1024 * 16];
#define VT_S5_STAMP(k)                                                                                         \
    do {                                                                                                       \
        if ((a.debug & 16) && lane == 0 && blockIdx.x < 1024) vt_span5_stamps[blockIdx.x * 16 + (k)] = wall_clock64(); \
    } while (0)

struct S5Args {
    IgemmArgs p;
    int dmin, halo;  // span row of tap t = (eh*W + ew) - dmin, in [0, halo]
    int units;       // ceil(M / 32)
    int upx;         // units per XCD
    int rslots;      // row slots per XCD (workgroups per XCD / tiles_n)
    int npc;         // span pieces (16 rows x 64 B) per chunk: ceil((32*kFMX + halo) / 16)
    int ppt;         // pieces issued per tap at taps 0..5: ceil(npc / 6)
    unsigned hw_magic, w_magic;  // ceil(2^32 / (H*W)), ceil(2^32 / W): quotients by multiply-high (+ one correction)
    int dtap[9];     // span row of every tap
    int eh[9], ew[9];  // its (row, column) offset (h0 + dh[t], w0 + dw[t]): scalar loads, the int8 tables compile to vector loads
    int debug;       // dev ablations (VT_SPAN5_ABL): 1 loader issues no DMA in the loop, 2 no MFMA/reads, 4 no vmcnt wait, 8 no row tables
};

__device__ __forceinline__ int swz4(int g) { return (0x1320 >> ((g & 3) * 4)) & 3; }

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long)hi << 32) | lo);
}
// LDS-DMA, 16 B per lane: LDS address = M0 + lane*16, global address = sbase + voff (or the per-lane address)
__device__ __forceinline__ void glds_s(unsigned voff, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(uniform_ptr(sbase)) : "memory");  // 5 wait states: VALU-written SGPR base
}
__device__ __forceinline__ void glds_v(unsigned long gsrc) {
    asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gsrc) : "memory");
}
__device__ __forceinline__ void set_m0(unsigned v) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(__builtin_amdgcn_readfirstlane(v)) : "memory");
}
__device__ __forceinline__ unsigned get_m0() {
    unsigned v;
    asm volatile("s_mov_b32 %0, m0" : "=s"(v)::"memory");
    return v;
}
template <int N>
__device__ __forceinline__ void vmw() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// counted wait with a run-time, wave-uniform count (vmcnt takes an immediate): binary decision tree, 0..63
__device__ __forceinline__ void vm_wait_dyn(int n) {
#define VT_W4(b)                                  \
    switch (n - (b)) {                            \
        case 0: vmw<(b) + 0>(); break;            \
        case 1: vmw<(b) + 1>(); break;            \
        case 2: vmw<(b) + 2>(); break;            \
        default: vmw<(b) + 3>(); break;           \
    }
    if (n <= 0) { vmw<0>(); return; }
    if (n >= 60) { vmw<60>(); return; }
    if (n < 32) {
        if (n < 16) {
            if (n < 8) { if (n < 4) { VT_W4(0) } else { VT_W4(4) } }
            else { if (n < 12) { VT_W4(8) } else { VT_W4(12) } }
        } else {
            if (n < 24) { if (n < 20) { VT_W4(16) } else { VT_W4(20) } }
            else { if (n < 28) { VT_W4(24) } else { VT_W4(28) } }
        }
    } else {
        if (n < 48) {
            if (n < 40) { if (n < 36) { VT_W4(32) } else { VT_W4(36) } }
            else { if (n < 44) { VT_W4(40) } else { VT_W4(44) } }
        } else {
            if (n < 56) { if (n < 52) { VT_W4(48) } else { VT_W4(52) } }
            else { VT_W4(56) }
        }
    }
#undef VT_W4
}

// LDS map (bytes): [row masks 2 x kBMX x 4][row output pixel 2 x kBMX x 4][filter ring kNSB x 8 KiB][zero 64]
//                  [span slot 0][span slot 1]
struct L5 {
    static constexpr int kMask = 0;
    static constexpr int kPo = kMask + 2 * kBMX * 4;
    static constexpr int kB = kPo + 2 * kBMX * 4;
    static constexpr int kZero = kB + kNSB * kBSlot;
    static constexpr int kA = kZero + 64;
    __host__ __device__ static constexpr int bytes(int npc) { return kA + 2 * npc * 1024; }
};

template <int T>
using I_ = std::integral_constant<int, T>;

typedef const __attribute__((address_space(4))) S5Args* ArgsPtr;
__device__ __forceinline__ ArgsPtr fresh_args() {
    ArgsPtr q = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);  // nothing migrates across a step boundary (keeps live ranges inside a step)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// 5 waves: 0..3 compute, 4 loader.  Two workgroups per CU = 10 waves: one SIMD of the CU hosts three of them,
// hence at most 168 registers per lane.
template <int MODE>  // epilogue: 0 plain (+ residual), 1 BatchNorm statistics, 2 affine (+ ReLU, + residual)
__global__ void __launch_bounds__(320, 3) span5_kernel(const S5Args a) {
    const IgemmArgs& p = a.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* sMask = (unsigned*)(smem + L5::kMask);
    int* sPo = (int*)(smem + L5::kPo);
    const char* sBb = smem + L5::kB;
    const char* sZb = smem + L5::kZero;
    const char* sAb = smem + L5::kA;
    const int aslot_bytes = a.npc * 1024;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's share: a contiguous range of 32-row units of one XCD, one filter column tile ----
    const int bid = blockIdx.x, xcd = bid & 7, l = bid >> 3;
    const int tn = l % p.tiles_n, rs = l / p.tiles_n;
    const int ux0 = xcd * a.upx, ux1 = min(a.units, ux0 + a.upx);
    const int nx = max(0, ux1 - ux0);
    const int ua = __builtin_amdgcn_readfirstlane(ux0 + (int)((unsigned)rs * (unsigned)nx / (unsigned)a.rslots));
    const int ub = __builtin_amdgcn_readfirstlane(ux0 + (int)((unsigned)(rs + 1) * (unsigned)nx / (unsigned)a.rslots));
    const int nun = ub - ua;
    if (nun <= 0) return;
    const int ntile = __builtin_amdgcn_readfirstlane((nun + kFMX - 1) / kFMX);
    const int tbase = __builtin_amdgcn_readfirstlane(nun / ntile);
    const int textra = nun - tbase * ntile;
    // tile k: units [ua + k*tbase + min(k, textra), +tbase + (k < textra)); heights differ by at most one unit
#define VT_TILE_U0(k) (ua + (k)*tbase + min((k), textra))
#define VT_TILE_F(k) (tbase + ((k) < textra ? 1 : 0))

    const int nchunks = __builtin_amdgcn_readfirstlane(p.Cin / 32);
    const int nsteps = nchunks * 9;

    if (wave == 4) {
        // =========================== loader wave ===================================================
        VT_S5_STAMP(0);
        const char* xg = (const char*)p.x;
        const char* wg = (const char*)p.w;
        const unsigned a_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + L5::kA);
        const unsigned b_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + L5::kB);
        const unsigned m0_keep = get_m0();
        const long M = p.M;
        const long ldx2 = (long)p.ldx * 2;
        const int cin2 = p.Cin * 2;
        if (lane < 4) ((unsigned*)(smem + L5::kZero))[lane] = 0u;

        // span piece = 16 rows x 64 B: lane owns row (lane>>2), source chunk (lane&3)^swz4(lane>>4)
        const int cjA = (lane & 3) ^ swz4(lane >> 4);
        const unsigned a_vo = (unsigned)(((lane >> 2) * p.ldx + cjA * 8) * 2);
        // filter slice = 8 pieces of 16 rows; piece q, row n = 16q + (lane>>2); the fragment reads address row n
        // with chunk position kq ^ swz4(n>>3), so the source chunk is (lane&3) ^ swz4(2q + (lane>>5)).
        // Rows past Cout (N tail) are clamped: their outputs are never stored.
        unsigned b_voff[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = min(tn * 128 + 16 * q + (lane >> 2), p.Cout - 1);
            const int cj = (lane & 3) ^ swz4(2 * q + (lane >> 5));
            b_voff[q] = (unsigned)(((long)n * p.ldw + cj * 8) * 2);
        }
        int issued = 0;  // vector-memory instructions issued so far (all of them LDS-DMA)

        auto issue_slice = [&](int slot, int ic, int T) {
            const char* sb = wg + (long)ic * 64 + (long)T * cin2;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                set_m0(b_base + (unsigned)(slot * kBSlot + q * 1024));
                glds_s(b_voff[q], sb);
            }
            issued += 8;
        };
        // pieces [p0, p1) of the span that starts at pixel row prow (may be < 0 / run past M at the two ends of the
        // tensor), channel chunk byte offset cb, into span slot sl.  Rows outside the tensor are clamped: they are
        // padding rows of every tap that could read them, so the fragment reads take the zero block instead.
        auto issue_pieces = [&](int sl, long prow, int cb, int p0, int p1) {
            for (int pc = p0; pc < p1; ++pc) {
                const long r0 = prow + pc * 16;
                set_m0(a_base + (unsigned)(sl * aslot_bytes + pc * 1024));
                if (r0 >= 0 && r0 + 16 <= M) {
                    glds_s(a_vo, xg + r0 * ldx2 + cb);
                } else {
                    long pix = r0 + (lane >> 2);
                    pix = pix < 0 ? 0 : (pix >= M ? M - 1 : pix);
                    glds_v((unsigned long)xg + (unsigned long)(pix * ldx2 + cb + cjA * 16));
                }
            }
            issued += max(0, p1 - p0);
        };
        // rows [r_lo, r_hi) of the tables of the tile that starts at pixel m0t (into table half par): which taps
        // stay inside the image, where the row is stored
        auto row_tables = [&](int par, long m0t, int r_lo, int r_hi) {
            ArgsPtr Q = fresh_args();
            const int W_ = Q->p.Wi, H_ = Q->p.Hi, HW_ = H_ * W_;
            for (int r = r_lo + lane; r < r_hi; r += 64) {
                const long m = m0t + r;
                unsigned bits = 0;
                int po = 0;
                if (m < M) {
                    int b = (int)__umulhi((unsigned)m, Q->hw_magic);
                    int rem = (int)m - b * HW_;
                    if (rem < 0) rem += HW_, --b;
                    int oi = (int)__umulhi((unsigned)rem, Q->w_magic);
                    int oj = rem - oi * W_;
                    if (oj < 0) oj += W_, --oi;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        if ((unsigned)(oi + Q->eh[t]) < (unsigned)H_ && (unsigned)(oj + Q->ew[t]) < (unsigned)W_) bits |= 1u << t;
                    }
                    po = Q->p.dense_out ? (int)m : (b * Q->p.oH + (oi * Q->p.oHs + Q->p.oh0)) * Q->p.oW + (oj * Q->p.oWs + Q->p.ow0);
                }
                sMask[par * kBMX + r] = bits;
                sPo[par * kBMX + r] = po;
            }
        };

        // ---- prologue: first tile's span chunk 0, slices 0..2, its row tables -------------------------
        const long G = (long)ntile * nsteps;  // steps of this workgroup
        long m0_cur = (long)VT_TILE_U0(0) * 32;
        issue_pieces(0, m0_cur + a.dmin, 0, 0, a.npc);
        int h0 = 0, h1 = 0, h2 = 0;  // `issued` right after slices g, g+1, g+2 went out
        int sic = 0, sT = 0;         // (chunk, tap) of the next slice to issue; slices repeat per tile
        long sg = 0;                 // its global step
        auto next_slice = [&]() {
            if (sg < G) {
                issue_slice((int)(sg & 3), sic, sT);
                ++sg;
                if (++sT == 9) {
                    sT = 0;
                    if (++sic == nchunks) sic = 0;
                }
            }
            h0 = h1, h1 = h2, h2 = issued;
        };
        next_slice();
        next_slice();
        next_slice();
        row_tables(0, m0_cur, 0, kBMX);
        VT_S5_STAMP(1);

        int acur = 0;  // span slot of the chunk being computed
        for (int k = 0; k < ntile; ++k) {
            const bool has_next = k + 1 < ntile;
            const long m0_nxt = has_next ? (long)VT_TILE_U0(k + 1) * 32 : 0;
            for (int ic = 0; ic < nchunks; ++ic) {
                const bool lastc = ic + 1 == nchunks;
                const bool nextc = !lastc || has_next;                       // a chunk follows this one
                const long prow_t = (lastc ? m0_nxt : m0_cur) + a.dmin;      // first span row of that chunk
                const int cb_t = lastc ? 0 : (ic + 1) * 64;
                for (int T = 0; T < 9; ++T) {
                    // slice of this step (and everything older: the span of its chunk) must have landed
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // table writes visible past the barrier
                    if (!(a.debug & 4)) vm_wait_dyn(issued - h0);
                    wg_barrier();
                    // the compute waves hold the previous step's fragments: its ring slot, and at tap 0 the other
                    // span slot, may be overwritten
                    if (nextc && T < 6 && !(a.debug & 1)) issue_pieces(acur ^ 1, prow_t, cb_t, T * a.ppt, min(a.npc, (T + 1) * a.ppt));
                    if (!(a.debug & 1)) next_slice();
                    // the next tile's row tables, in four parts behind taps 5..8 of this tile's first chunk (the
                    // half they go to was last read by the previous tile's epilogue, which every compute wave left
                    // before this tile's first barrier)
                    if (ic == 0 && T >= 5 && has_next && !(a.debug & 8)) row_tables((k + 1) & 1, m0_nxt, (T - 5) * 64, min(kBMX, (T - 4) * 64));
                }
                acur ^= 1;
            }
            m0_cur = m0_nxt;
        }
        VT_S5_STAMP(2);
        vmw<0>();
        set_m0(m0_keep);
        return;
    }

    // =============================== compute waves ==================================================
    const int wm = wave >> 1, wn = wave & 1;
    const int q4 = lane >> 4, c16 = lane & 15;
    // this lane's output channels: ch(h, e8) = tn*128 + wn*64 + h*32 + q4*8 + e8, h = 0,1, e8 = 0..7
    const int ch0 = tn * 128 + wn * 64 + q4 * 8;
    // filter fragment j of this lane: MFMA row r = c16 -> slice row n_j = wn*64 + (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3);
    // (n_j >> 3) & 3 = r >> 2 for every j, so the four fragments share one swizzle term and differ by constants
    const int nb0 = wn * 64 + (c16 >> 2) * 8 + (c16 & 3);
    const int b_lane = (nb0 * 4 + (q4 ^ swz4(c16 >> 2))) * 16;  // byte offset inside a slice; j adds {0,256,2048,2304}
    int bcur = 0, acur = 0;

    for (int k = 0; k < ntile; ++k) {
        const int par = k & 1;
        const int f_cur = VT_TILE_F(k);
        const long m0_cur = (long)VT_TILE_U0(k) * 32;
        const int fm = max(f_cur, 4);           // row fragments per wave in this tile (4..7)
        const int rows_tile = 32 * f_cur;       // rows this tile owns (stores / statistics)

        auto run = [&](auto FMc) {
            constexpr int FM = decltype(FMc)::value;
            int wrow = wm * 16 * FM + c16;  // this lane's row inside the tile, fragment 0
            asm volatile("" : "+v"(wrow));    // (opaque: nothing derived from it is hoisted out of the tile loop)
            wg_barrier();                         // step 0 of this tile: its tables, span and first slice are in LDS
            if (wave == 0 && k < 4) VT_S5_STAMP(4 + 3 * k);
            // tap masks of this lane's FM rows, 9 bits each, three rows per register
            unsigned mw[(FM + 2) / 3];
#pragma unroll
            for (int i = 0; i < (FM + 2) / 3; ++i) mw[i] = 0u;
#pragma unroll
            for (int i = 0; i < FM; ++i) mw[i / 3] |= (sMask[par * kBMX + wrow + i * 16] & 0x1ffu) << ((i % 3) * 9);
            f32x4 acc[FM][4];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

            for (int ic = 0; ic < nchunks; ++ic) {
                bcur = __builtin_amdgcn_readfirstlane(bcur);
                acur = __builtin_amdgcn_readfirstlane(acur);
                const unsigned a_rd = (unsigned)(acur * aslot_bytes);
                auto step = [&](auto Tc) {
                    constexpr int T = decltype(Tc)::value;
                    if (!(T == 0)) wg_barrier();
                    else if (ic != 0) wg_barrier();
                    if (a.debug & 2) { bcur = (bcur + 1) & 3; return; }
                    const int srow = wrow + fresh_args()->dtap[T];  // scalar load per step: nine live SGPRs fewer
                    const char* A = sAb + (a_rd + (unsigned)((srow * 4 + (q4 ^ swz4(srow >> 2))) * 16));
                    const char* Bt = sBb + ((bcur << 13) + b_lane);
                    // filter fragments live two at a time (registers: 168 per lane with three waves on a SIMD):
                    // fragment j+2 is read into j's registers as soon as j's seven MFMAs have been issued
                    uint4 af[FM], bfa, bfb;
                    bfa = *(const uint4*)(Bt);
                    bfb = *(const uint4*)(Bt + 256);
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        // (sZb - i*1024) + i*1024 == the zero block: the constant stays in the offset field
                        const char* src = ((mw[i / 3] >> ((i % 3) * 9 + T)) & 1u) ? A : sZb - i * 1024;
                        af[i] = *(const uint4*)(src + i * 1024);
                    }
#define VT_MMA_COL(bfrag, j)                                                                              \
    _Pragma("unroll") for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(   \
        __builtin_bit_cast(bf16x8, bfrag), __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0)
                    __builtin_amdgcn_sched_barrier(0);
                    VT_MMA_COL(bfa, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    bfa = *(const uint4*)(Bt + 2048);
                    VT_MMA_COL(bfb, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    bfb = *(const uint4*)(Bt + 2304);
                    VT_MMA_COL(bfa, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    VT_MMA_COL(bfb, 3);
#undef VT_MMA_COL
                    bcur = (bcur + 1) & 3;
                };
                step(I_<0>{});
                step(I_<1>{});
                step(I_<2>{});
                step(I_<3>{});
                step(I_<4>{});
                step(I_<5>{});
                step(I_<6>{});
                step(I_<7>{});
                step(I_<8>{});
                acur ^= 1;
            }

            if (wave == 0 && k < 4) VT_S5_STAMP(5 + 3 * k);
            // ---- epilogue: two 16-byte stores per row fragment, straight from the accumulators ----------
            ArgsPtr Q = fresh_args();
            constexpr bool affine = MODE == 2, stats = MODE == 1;
            const bool relu = MODE == 2 && (Q->p.flags & VT_CONV_RELU);
            const bool has_res = MODE != 1 && (Q->p.flags & VT_CONV_RESIDUAL) != 0;
            const int Cout_ = Q->p.Cout, M_ = Q->p.M, ldy_ = Q->p.ldy, ldr_ = Q->p.ldr;
            const bool dense_ = Q->p.dense_out;
            bf16_t* __restrict__ yg = (bf16_t*)Q->p.y;
            const bf16_t* __restrict__ rg = (const bf16_t*)Q->p.res;
            const float* scale_ = Q->p.scale;
            const float* shift_ = Q->p.shift;
            float* stats_ = Q->p.stats;
            const int rep = (int)((m0_cur / 32) % kStatReplicas);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // (opaque: keeps the per-lane 64-bit output / statistics addresses from being hoisted out of the tile
                //  loop, where they would sit in scratch across the whole step loop -- any scratch use at all costs
                //  this kernel its second workgroup per CU)
                int n = ch0 + h * 32;
                asm volatile("" : "+v"(n));
                float s1[8], s2[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[e] = 0.f, s2[e] = 0.f;
                float sc[8], sf[8];
                if (affine) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int ne = min(n + e, Cout_ - 1);
                        sc[e] = scale_ ? scale_[ne] : 1.f;
                        sf[e] = shift_[ne];
                    }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int tr = wrow + i * 16;  // row inside the tile
                    const int mrow = (int)m0_cur + tr;  // < 2^31 (checked by the dispatcher)
                    const bool row_ok = tr < rows_tile && mrow < M_;
                    const long po = dense_ ? mrow : sPo[par * kBMX + tr];
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = acc[i][2 * h + (e >> 2)][e & 3];
                        if (affine) t = fmaf(t, sc[e], sf[e]);
                        if (relu) t = fmaxf(t, 0.f);
                        v[e] = t;
                    }
                    uint4 out = VecIO<bf16_t>::pack(v);
                    if (row_ok && n < Cout_) {
                        if (stats) {
                            float r8[8];
                            VecIO<bf16_t>::unpack(out, r8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                s1[e] += r8[e];
                                s2[e] = fmaf(r8[e], r8[e], s2[e]);
                            }
                        }
                        if (has_res) {
                            const uint4 rr = *(const uint4*)(rg + (po * ldr_ + n));
                            float fv[8], fr[8];
                            VecIO<bf16_t>::unpack(out, fv);
                            VecIO<bf16_t>::unpack(rr, fr);
#pragma unroll
                            for (int e = 0; e < 8; ++e) fv[e] += fr[e];
                            out = VecIO<bf16_t>::pack(fv);
                        }
                        *(uint4*)(yg + (po * ldy_ + n)) = out;
                    }
                }
                if (stats) {
                    // sum over the 16 pixel lanes (same q4): butterfly, then lanes c16 = 0..7 keep channel e = c16
                    float u = 0.f, v = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x1 = s1[e], x2 = s2[e];
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) {
                            x1 += __shfl_xor(x1, o, 64);
                            x2 += __shfl_xor(x2, o, 64);
                        }
                        u = c16 == e ? x1 : u;
                        v = c16 == e ? x2 : v;
                    }
                    const int nn = n + c16;
                    if (c16 < 8 && nn < Cout_) {
                        vt_stat_add(stats_, ((long)rep * 2 + 0) * Cout_ + nn, u);
                        vt_stat_add(stats_, ((long)rep * 2 + 1) * Cout_ + nn, v);
                    }
                }
            }
            if (wave == 0 && k < 4) VT_S5_STAMP(6 + 3 * k);
        };
        switch (fm) {
            case 4: run(I_<4>{}); break;
            case 5: run(I_<5>{}); break;
            case 6: if constexpr (kFMX > 6) { run(I_<6>{}); break; }
            default: run(I_<kFMX>{}); break;
        }
    }
#undef VT_TILE_U0
#undef VT_TILE_F
}

}  // namespace

// returns -1 when this kernel does not apply (the caller then tries the other span kernels)
int vt_span5_dispatch(IgemmArgs& a0, int dtype, void* stream) {
    const int enabled = getenv("VT_SPAN5") ? atoi(getenv("VT_SPAN5")) : 0;  // TODO static once settled
    if (!enabled || dtype != VT_BF16) return -1;
    if (a0.sh != 1 || a0.sw != 1 || a0.Ho != a0.Hi || a0.Wo != a0.Wi) return -1;
    if (a0.Cin % 32 != 0 || a0.ntaps != 9 || a0.Cout < 64) return -1;
    if ((long)a0.M + 2L * a0.Wi * VT_MAX_TAPS > 0x7fffffffL) return -1;
    if ((long)a0.B * a0.oH * a0.oW > 0x7fffffffL) return -1;
    if ((unsigned long)a0.M * a0.ldx * 2 >= 0xffff0000ul) return -1;
    if ((unsigned long)a0.Cout * a0.ldw * 2 >= 0xffff0000ul) return -1;
    int dmin = 1 << 30, dmax = -(1 << 30);
    for (int t = 0; t < a0.ntaps; ++t) {
        const int d = (a0.h0 + a0.dh[t]) * a0.Wi + (a0.w0 + a0.dw[t]);
        dmin = d < dmin ? d : dmin;
        dmax = d > dmax ? d : dmax;
    }
    S5Args a;
    a.p = a0;
    IgemmArgs& p = a.p;
    a.dmin = dmin;
    a.halo = dmax - dmin;
    a.debug = getenv("VT_SPAN5_ABL") ? atoi(getenv("VT_SPAN5_ABL")) : 0;
    for (int t = 0; t < 9; ++t) {
        a.eh[t] = a0.h0 + a0.dh[t];
        a.ew[t] = a0.w0 + a0.dw[t];
        a.dtap[t] = a.eh[t] * a0.Wi + a.ew[t] - dmin;
    }
    p.tiles_n = (p.Cout + 127) / 128;
    const int wgs_cu = getenv("VT_SPAN5_WGS") ? atoi(getenv("VT_SPAN5_WGS")) : 2;  // dev: workgroups per CU
    const int g8 = 32 * wgs_cu;  // workgroups per XCD: 32 CUs x 2
    if (g8 % p.tiles_n != 0) return -1;
    // MFMA-bound layers only: enough rows to give every workgroup at least 4 units
    if ((long)p.M * p.tiles_n < 512L * 32 * 4) return -1;
    a.rslots = g8 / p.tiles_n;
    a.units = (p.M + 31) / 32;
    a.upx = (a.units + 7) / 8;
    a.npc = (32 * kFMX + a.halo + 15) / 16;
    a.ppt = (a.npc + 5) / 6;
    const int smem = L5::bytes(a.npc);
    if (smem * 2 > 160 * 1024) return -1;
    const unsigned HW = (unsigned)(p.Hi * p.Wi), W = (unsigned)p.Wi;
    a.hw_magic = (unsigned)((0x100000000ull + HW - 1) / HW);
    a.w_magic = (unsigned)((0x100000000ull + W - 1) / W);
    if (HW == 1 || W == 1) return -1;
    const int mode = (p.flags & VT_CONV_STATS) ? 1 : ((p.flags & VT_CONV_AFFINE) ? 2 : 0);
    if (mode == 1 && (p.flags & (VT_CONV_AFFINE | VT_CONV_RELU | VT_CONV_RESIDUAL))) return -1;
    if (mode == 0 && (p.flags & VT_CONV_RELU)) return -1;
    auto kern = mode == 1 ? span5_kernel<1> : (mode == 2 ? span5_kernel<2> : span5_kernel<0>);
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(span5)");
        if (rc != VT_OK) return rc;
    }
    if (getenv("VT_SPAN5_DEBUG")) {
        static bool once = false;
        if (!once) {
            once = true;
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 320, smem);
            fprintf(stderr, "[span5] occupancy query: %d workgroups of 320 threads per CU at %d B LDS; npc %d ppt %d\n", nb, smem, a.npc, a.ppt);
        }
    }
    vt_note_kernel("span5_kernel<bf16,4+1 waves,2wg/cu>");
    hipLaunchKernelGGL(kern, dim3(8 * g8), dim3(320), smem, (hipStream_t)stream, a);
    VT_CHECK_LAUNCH("vt_conv_igemm(span5)");
    if (a.debug & 16) {
        static int calls = 0;
        if (++calls == 12) {  // a warm launch
            (void)hipStreamSynchronize((hipStream_t)stream);
            static unsigned long long h[1024 * 16];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(vt_span5_stamps), sizeof(h));
            const int nb = 8 * g8 < 1024 ? 8 * g8 : 1024;
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < nb; ++b) t0 = h[b * 16] < t0 ? h[b * 16] : t0;
            double avg[16] = {0};
            for (int b = 0; b < nb; ++b)
                for (int k = 0; k < 16; ++k) avg[k] += (double)(h[b * 16 + k] - t0) * 0.01 / nb;
            fprintf(stderr, "[span5 stamps, us from the first workgroup's start, mean over %d WGs] loader start %.1f prologue done %.1f loop done %.1f |"
                            " tile0: first barrier %.1f loop end %.1f epilogue end %.1f | tile1: %.1f %.1f %.1f | tile2: %.1f %.1f %.1f\n",
                    nb, avg[0], avg[1], avg[2], avg[4], avg[5], avg[6], avg[7], avg[8], avg[9], avg[10], avg[11], avg[12]);
            {
                std::vector<double> st, en;
                for (int b = 0; b < nb; ++b) st.push_back((h[b * 16] - t0) * 0.01), en.push_back((h[b * 16 + 2] - t0) * 0.01);
                std::sort(st.begin(), st.end());
                std::sort(en.begin(), en.end());
                fprintf(stderr, "[span5 stamps] start times (us), sorted, every 32nd WG:");
                for (int b = 0; b < nb; b += 32) fprintf(stderr, " %.1f", st[b]);
                fprintf(stderr, "\n[span5 stamps] loader end times (us), sorted, every 32nd WG:");
                for (int b = 0; b < nb; b += 32) fprintf(stderr, " %.1f", en[b]);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "[span5 stamps] WG 0: %.1f %.1f %.1f | %.1f %.1f %.1f | %.1f %.1f %.1f\n", (h[0]-t0)*0.01, (h[1]-t0)*0.01, (h[2]-t0)*0.01,
                    (h[4]-t0)*0.01, (h[5]-t0)*0.01, (h[6]-t0)*0.01, (h[7]-t0)*0.01, (h[8]-t0)*0.01, (h[9]-t0)*0.01);
        }
    }
    return VT_OK;
}
