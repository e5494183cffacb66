// vt_igemm_span3.hip -- persistent, software-pipelined input-span convolution (bf16) for the
// MFMA-bound stride-1 3x3 layers: every ConvNormAct 3x3 stride-1 forward conv of the Darknet /
// CSPDarknet / VoVNet stages (reference components.py:26-35, darknet.py:23-24, vovnet.py:41-44)
// and their stride-1 data gradients.
//
// Same GEMM view and LDS images as vt_igemm_span.hip (one input span per 32-channel chunk shared by
// all taps, one 128 x 64 B filter slice per (chunk, tap) step, XOR-swizzled through the DMA source
// address).  What is different, each item measured against that kernel on 128->128 3x3 @28x28 B=256
// (tools/exp_span.sh ablations: of its 85 us, 45 remained with neither MFMA nor in-loop DMA):
//   * PERSISTENT workgroups: the grid is one (WM=4: 8 waves) or two (WM=2: 4 waves) workgroups per
//     CU; each owns a contiguous range of 16*WM-row units of the flat pixel index and cuts it into
//     tiles of 4..7 units (tile height 16*WM*FM rows, FM per tile), so every CU gets the same number
//     of rows (no 1.75-round tail) and the LDS-DMA of the next tile's first span / filter slices is
//     in flight while the current tile finishes and stores.
//   * the MFMA operands are swapped (filter rows = MFMA rows, pixels = MFMA columns): a lane ends up
//     with 2 x 8 CONSECUTIVE output channels of one pixel, i.e. two 16-byte NHWC stores straight from
//     the accumulators -- no LDS staging, no 2-byte LDS writes, no barrier in the epilogue.
//   * fragment reads are double buffered in registers: step s+1's twelve ds_read_b128 are issued
//     before step s's MFMAs, so the matrix pipe never waits for LDS; three filter slices in flight.
//   * the filter-slice DMA of a wave is ONE M0 write + instructions that differ in their immediate
//     offset only.
// Applies to: bf16, ntaps >= 3, Cin % 32 == 0, tiles_n (= ceil(Cout/128)) dividing the workgroups of
// an XCD.  Everything else stays on vt_igemm_span.hip / vt_igemm.hip.
#include <stdlib.h>

#include <type_traits>

#include "vt_common.h"
#include "vt_igemm_args.h"

// diagnostic builds only (tools/build_diag.sh): -DVT_SPAN3_ABLATE=<bits>  1: no MFMA, 2: no LDS-DMA inside the
// step loop, 4: no fragment reads, 8: no epilogue stores.  Results are wrong by construction; only time is read.
#ifndef VT_SPAN3_ABLATE
#define VT_SPAN3_ABLATE 0
#endif

namespace {

constexpr int kFMX = 7;    // row fragments (16 rows) per wave, at most
constexpr int kNSB = 4;    // filter-slice ring slots
constexpr int kBSlot = 128 * 64;  // bytes per filter slice

__device__ __attribute__((aligned(16))) unsigned int vt_span3_zero16[4];

struct S3Args {
    IgemmArgs p;
    int dmin, halo;    // span row of tap t = (eh*W + ew) - dmin, in [0, halo]
    int units;         // ceil(M / (16*WM))
    int upx;           // units per XCD
    int rslots;        // row slots per XCD (workgroups per XCD / tiles_n)
    int aslot_rows;    // rows of one span slot (multiple of 16)
    int fast_dma;
};

__device__ __forceinline__ int swz4(int g) { return (0x1320 >> ((g & 3) * 4)) & 3; }

// LDS-DMA, 16 B per lane: LDS address = M0 + imm + lane*16, global address = sbase + voff + imm
// (operands that are wave-uniform by construction go through readfirstlane: the "s" constraint does not
// make the compiler's divergence analysis agree, and the instruction is a no-op when it already does)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long)hi << 32) | lo);
}
template <int IMM>
__device__ __forceinline__ void glds_s(unsigned voff, const void* sbase) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(uniform_ptr(sbase)), "n"(IMM) : "memory");
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
__device__ __forceinline__ void vm_wait3() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void vm_wait3_dyn(int n) {
    if (n <= 0) vm_wait3<0>();
    else if (n == 1) vm_wait3<1>();
    else if (n == 2) vm_wait3<2>();
    else if (n == 3) vm_wait3<3>();
    else if (n == 4) vm_wait3<4>();
    else if (n == 5) vm_wait3<5>();
    else if (n == 6) vm_wait3<6>();
    else if (n == 7) vm_wait3<7>();
    else if (n == 8) vm_wait3<8>();
    else if (n == 9) vm_wait3<9>();
    else if (n == 10) vm_wait3<10>();
    else if (n == 11) vm_wait3<11>();
    else vm_wait3<12>();
}

struct Frags {
    uint4 a[kFMX];
    uint4 b[4];
};

// LDS map (bytes): [tap row offsets 32 x 4][row masks 2 x BMX x 4][row output pixel 2 x BMX x 4]
//                  [filter ring kNSB x 8 KiB][zero 64][span slot 0][span slot 1]
template <int WM>
struct L3 {
    static constexpr int BMX = 16 * kFMX * WM;
    static constexpr int kTap = 0;
    static constexpr int kMask = 128;
    static constexpr int kPo = kMask + 2 * BMX * 4;
    static constexpr int kB = kPo + 2 * BMX * 4;
    static constexpr int kZero = kB + kNSB * kBSlot;
    static constexpr int kA = kZero + 64;
    __host__ __device__ static constexpr int bytes(int aslot_rows) { return kA + 2 * aslot_rows * 64; }
};

template <int T>
using I_ = std::integral_constant<int, T>;

// Kernel arguments read where they are used (prologue, row tables, epilogue) instead of living in scalar
// registers across the step loop: an opaque copy of the kernarg pointer makes every such read a fresh s_load.
typedef const __attribute__((address_space(4))) S3Args* ArgsPtr;
__device__ __forceinline__ ArgsPtr fresh_args() {
    ArgsPtr q = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

template <int WM>
__global__ void __launch_bounds__(128 * WM, 2) span3_kernel(const S3Args a) {
    constexpr int NW = 2 * WM, NT = 64 * NW;
    constexpr int UNIT = 16 * WM;
    constexpr int ITB = 8 / NW;  // filter DMA instructions per wave per step (8 x 1 KiB per slice)
    using L = L3<WM>;
    const IgemmArgs& p = a.p;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* sMask = (unsigned*)(smem + L::kMask);
    int* sPo = (int*)(smem + L::kPo);
    const char* sBb = smem + L::kB;
    const char* sZb = smem + L::kZero;
    const char* sAb = smem + L::kA;
    const int aslot_bytes = a.aslot_rows * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- this workgroup's share: a contiguous range of row units of one XCD, one filter column tile
    const int bid = blockIdx.x, xcd = bid & 7, l = bid >> 3;
    const int tn = l % p.tiles_n, rs = l / p.tiles_n;
    const int ux0 = xcd * a.upx, ux1 = min(a.units, ux0 + a.upx);
    const int nx = max(0, ux1 - ux0);
    // (32-bit arithmetic and readfirstlane: a 64-bit division is expanded on the vector ALU and would drag
    //  every loop counter derived from it into vector registers)
    const int ua = __builtin_amdgcn_readfirstlane(ux0 + (int)((unsigned)rs * (unsigned)nx / (unsigned)a.rslots));
    const int ub = __builtin_amdgcn_readfirstlane(ux0 + (int)((unsigned)(rs + 1) * (unsigned)nx / (unsigned)a.rslots));
    const int nun = ub - ua;
    if (nun <= 0) return;
    const int ntile = __builtin_amdgcn_readfirstlane((nun + kFMX - 1) / kFMX);
    const int tbase = __builtin_amdgcn_readfirstlane(nun / ntile);
    const int textra = nun - tbase * ntile;
    // tile k: units [ua + k*tbase + min(k, textra), +tbase + (k < textra))
#define VT_TILE_U0(k) (ua + (k)*tbase + min((k), textra))
#define VT_TILE_F(k) (tbase + ((k) < textra ? 1 : 0))

    const int W = p.Wi, H = p.Hi, HW = H * W;
    const int nchunks = p.Cin / 32;
    const int cin2 = p.Cin * 2;
    const char* xg = (const char*)p.x;
    const char* wg = (const char*)p.w;
    const unsigned a_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + L::kA);
    const unsigned b_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + L::kB);
    const unsigned m0_keep = get_m0();

    if (tid < 4) ((unsigned*)(smem + L::kZero))[tid] = 0u;

    // ---- DMA geometry -------------------------------------------------------------------------
    // span: a piece = 16 rows x 64 B; lane owns row (lane>>2), source chunk (lane&3)^swz4(lane>>4)
    const int cjA = (lane & 3) ^ swz4(lane >> 4);
    const unsigned a_vo = (unsigned)(((lane >> 2) * p.ldx + cjA * 8) * 2);
    const long ldx2 = (long)p.ldx * 2;
    // filter slice: 8 pieces of 16 rows; piece q = wave*ITB + i; row n = 16q + (lane>>2); the fragment
    // reads address row n with chunk position kq ^ swz4(n>>3), so the source chunk is
    // (lane&3) ^ swz4(2q + (lane>>5)).  Rows past Cout (N tail) are clamped: their outputs are never stored.
    unsigned b_voff[ITB];
#pragma unroll
    for (int i = 0; i < ITB; ++i) {
        const int q = wave * ITB + i;
        const int n = min(tn * 128 + 16 * q + (lane >> 2), p.Cout - 1);
        const int cj = (lane & 3) ^ swz4(2 * q + (lane >> 5));
        b_voff[i] = (unsigned)(((long)n * p.ldw + cj * 8) * 2);
    }

    // filter slice (chunk byte offset cb_, tap T3) into ring slot `slot`
#define VT_ISSUE_B3(slot, cb_, T3)                                                              \
    do {                                                                                        \
        const char* sb = wg + (long)(cb_) + (long)((T3)*cin2);                                  \
        set_m0(b_base + (unsigned)((slot)*kBSlot + wave * ITB * 1024));                         \
        glds_s<0>(b_voff[0], sb);                                                               \
        /* the immediate moves the LDS AND the global address: take it back out of the base */  \
        if constexpr (ITB == 2) glds_s<1024>(b_voff[1], sb - 1024);                             \
    } while (0)

    // piece pc_ of the span that starts at pixel row prow_ (may be < 0 / run past M at the two ends of the
    // tensor), channel chunk byte offset cb_, into span slot sl_.  Rows outside the tensor are clamped:
    // they are padding rows of every tap that could read them, so the fragment reads take the zero block.
#define VT_ISSUE_A_PIECE(sl_, prow_, cb_, pc_)                                                  \
    do {                                                                                        \
        const long r0 = (long)(prow_) + (pc_)*16;                                               \
        set_m0(a_base + (unsigned)((sl_)*aslot_bytes + (pc_)*1024));                            \
        if (r0 >= 0 && r0 + 16 <= (long)p.M) {                                                  \
            glds_s<0>(a_vo, xg + r0 * ldx2 + (cb_));                                            \
        } else {                                                                                \
            long pix = r0 + (lane >> 2);                                                        \
            pix = pix < 0 ? 0 : (pix >= (long)p.M ? (long)p.M - 1 : pix);                       \
            glds_v((unsigned long)xg + (unsigned long)(pix * ldx2 + (cb_) + cjA * 16));         \
        }                                                                                       \
    } while (0)

    // per output row of a tile (into table half par_): which taps stay inside the image, where the row goes
#define VT_ROW_TABLES(par_, m0_, rows_)                                                          \
    do {                                                                                         \
        ArgsPtr Q = fresh_args();                                                                \
        const int W_ = Q->p.Wi, H_ = Q->p.Hi, HW_ = H_ * W_, M_ = Q->p.M;                         \
        for (int r = tid; r < (rows_); r += NT) {                                                \
            const long m = (long)(m0_) + r;                                                      \
            unsigned bits = 0;                                                                   \
            int po = 0;                                                                          \
            if (m < M_) {                                                                        \
                const int b = (int)(m / HW_);                                                    \
                const int rem = (int)(m - (long)b * HW_);                                        \
                const int oi = rem / W_, oj = rem - oi * W_;                                     \
                _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                  \
                    const int eh = Q->p.h0 + Q->p.dh[t], ew = Q->p.w0 + Q->p.dw[t];              \
                    if ((unsigned)(oi + eh) < (unsigned)H_ && (unsigned)(oj + ew) < (unsigned)W_) bits |= 1u << t; \
                }                                                                                \
                po = Q->p.dense_out ? (int)m : (b * Q->p.oH + (oi * Q->p.oHs + Q->p.oh0)) * Q->p.oW + (oj * Q->p.oWs + Q->p.ow0); \
            }                                                                                    \
            sMask[(par_)*L::BMX + r] = bits;                                                     \
            sPo[(par_)*L::BMX + r] = po;                                                         \
        }                                                                                        \
    } while (0)

    // ---- prologue: first tile's span chunk 0 and filter slices 0..2, its row tables ----------------------
    int f_cur = VT_TILE_F(0);
    long m0_cur = (long)VT_TILE_U0(0) * UNIT;
    {
        const int npc = (16 * WM * max(f_cur, 4) + a.halo + 15) / 16;
        for (int pc = wave; pc < npc; pc += NW) VT_ISSUE_A_PIECE(0, m0_cur + a.dmin, 0, pc);
        VT_ISSUE_B3(0, 0, 0);
        VT_ISSUE_B3(1, 0, 1);
        VT_ISSUE_B3(2, 0, 2);
    }
    VT_ROW_TABLES(0, m0_cur, 16 * WM * max(f_cur, 4));
    int bcur = 0, acur = 0;
    const int nsteps = nchunks * 9;
    int gleft = ntile * nsteps - 1;  // filter slices after the current step, over all this workgroup's tiles

    const int q4 = lane >> 4, c16 = lane & 15;
    // this lane's output channels: ch(h, e8) = tn*128 + wn*64 + h*32 + q4*8 + e8, h = 0,1, e8 = 0..7
    const int ch0 = tn * 128 + wn * 64 + q4 * 8;
    // filter fragment j of this lane: MFMA row r = c16 -> slice row n_j = wn*64 + (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3);
    // (n_j >> 3) & 3 = r >> 2 for every j, so the four fragments share one swizzle term and differ by constants
    const int nb0 = wn * 64 + (c16 >> 2) * 8 + (c16 & 3);
    const int b_lane = (nb0 * 4 + (q4 ^ swz4(c16 >> 2))) * 16;  // byte offset inside a slice; j adds {0,256,2048,2304}

    for (int k = 0; k < ntile; ++k) {
        const int par = k & 1;
        const int fm = max(f_cur, 4);               // row fragments per wave in this tile (4..7)
        const int rows_tile = 16 * WM * f_cur;      // rows this tile owns (stores / statistics)
        const bool has_next = k + 1 < ntile;
        const int f_nxt = has_next ? VT_TILE_F(k + 1) : 0;
        const long m0_nxt = has_next ? (long)VT_TILE_U0(k + 1) * UNIT : 0;
        const int npc_cur = (16 * WM * fm + a.halo + 15) / 16;
        const int npc_nxt = (16 * WM * max(f_nxt, 4) + a.halo + 15) / 16;

        // the tables of the next tile go into the half the previous tile's epilogue may still be reading
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (has_next) VT_ROW_TABLES(par ^ 1, m0_nxt, 16 * WM * max(f_nxt, 4));

        auto run = [&](auto FMc) {
            constexpr int FM = decltype(FMc)::value;
            const int wrow = wm * 16 * FM + c16;  // this lane's row inside the tile, fragment 0
            unsigned fmask[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) fmask[i] = sMask[par * L::BMX + wrow + i * 16];
            // byte offset of this lane's fragment-0 row inside a span slot, per tap
            unsigned a_off[9];
            {
                ArgsPtr Q = fresh_args();
                const int W_ = Q->p.Wi, h0_ = Q->p.h0, w0_ = Q->p.w0, dmin_ = Q->dmin;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int srow = wrow + (h0_ + Q->p.dh[t]) * W_ + (w0_ + Q->p.dw[t]) - dmin_;
                    a_off[t] = (unsigned)((srow * 4 + (q4 ^ swz4(srow >> 2))) * 16);
                }
            }
            if (false)
            for (int t = 0; t < 9; ++t) {
                const int srow = wrow;
                a_off[t] = (unsigned)((srow * 4 + (q4 ^ swz4(srow >> 2))) * 16);
            }
            f32x4 acc[FM][4];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

            // One step = (chunk ic, tap T).  Per-wave DMA issue order inside a step: [span piece], [filter slice
            // of step s+3].  The span pieces of the NEXT chunk (this tile's, or chunk 0 of the next tile) go out
            // at taps 0..np_w-1 (np_w <= 7), piece index wave + NW*T; a chunk is first read 9 - T >= 3 steps
            // later, so by then it is always older than the filter slice being waited for.
            for (int ic = 0; ic < nchunks; ++ic) {
                // (loop-carried wave-uniform state: keep it on the scalar side)
                gleft = __builtin_amdgcn_readfirstlane(gleft);
                bcur = __builtin_amdgcn_readfirstlane(bcur);
                acur = __builtin_amdgcn_readfirstlane(acur);
                const bool lastc = ic + 1 == nchunks;
                const bool nextc = !lastc || has_next;
                const int npc_t = lastc ? npc_nxt : npc_cur;
                const int np_w = nextc ? min(7, max(0, (npc_t - wave + NW - 1) / NW)) : 0;  // pieces this wave issues
                const long prow_t = (lastc ? m0_nxt : m0_cur) + a.dmin;                       // first span row of the target
                const char* a_src = xg + (prow_t + wave * 16) * ldx2 + (lastc ? 0 : (ic + 1) * 64);
                const long a_stride = (long)NW * 16 * ldx2;
                unsigned a_m0 = a_base + (unsigned)((acur ^ 1) * aslot_bytes + wave * 1024);
                // pieces [pc_lo, pc_hi) lie inside the tensor (all of them, except at the two ends of the tensor)
                const int pc_lo = prow_t >= 0 ? 0 : (int)((-prow_t + 15) / 16);
                const int pc_hi = (int)min((long)npc_t, ((long)p.M - prow_t) / 16);
                const char* wb_cur = wg + (long)ic * 64;
                const char* wb_nxt = wg + (long)(lastc ? 0 : ic + 1) * 64;
                const unsigned a_rd = (unsigned)(acur * aslot_bytes);

                auto step = [&](auto Tc) {
                    constexpr int T = decltype(Tc)::value;
                    // slice s (issued three steps ago) must have landed; younger: what steps s-2 and s-1 issued
                    if constexpr ((VT_SPAN3_ABLATE & 2) != 0) {
                        vm_wait3<0>();
                    } else if (gleft < 2) {
                        vm_wait3<0>();  // the last two steps of the workgroup
                    } else {
                        constexpr int P1 = T - 1, P2 = T - 2;  // taps of the two previous steps (negative: none)
                        const int cnt = ((P1 >= 0 && P1 < np_w) ? 1 : 0) + ((P2 >= 0 && P2 < np_w) ? 1 : 0);
                        if (cnt == 2) vm_wait3<2 * ITB + 2>();
                        else if (cnt == 1) vm_wait3<2 * ITB + 1>();
                        else vm_wait3<2 * ITB>();
                    }
                    // every wave is past the MFMAs of step s-1, i.e. has the fragments of every earlier step in
                    // registers: the ring slot of step s-1 and (at tap 0) the other span slot may be overwritten
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if constexpr ((VT_SPAN3_ABLATE & 2) == 0 && T < 7) {
                        if (T < np_w) {
                            const int pc = wave + NW * T;
                            set_m0(a_m0);
                            if (pc >= pc_lo && pc < pc_hi) {
                                glds_s<0>(a_vo, a_src);
                            } else {
                                long pix = prow_t + pc * 16 + (lane >> 2);
                                pix = pix < 0 ? 0 : (pix >= (long)p.M ? (long)p.M - 1 : pix);
                                glds_v((unsigned long)xg + (unsigned long)(pix * ldx2 + (lastc ? 0 : (ic + 1) * 64) + cjA * 16));
                            }
                            a_src += a_stride;
                            a_m0 += NW * 1024;
                        }
                    }
                    if constexpr ((VT_SPAN3_ABLATE & 2) == 0) {
                        if (gleft >= 3) {
                            constexpr int T3 = (T + 3) % 9;
                            const char* sb = (T < 6 ? wb_cur : wb_nxt) + (long)(T3 * cin2);
                            set_m0(b_base + (unsigned)((((bcur + 3) & 3) << 13) + wave * ITB * 1024));
                            glds_s<0>(b_voff[0], sb);
                            // the immediate moves the LDS AND the global address: take it back out of the base
                            if constexpr (ITB == 2) glds_s<1024>(b_voff[1], sb - 1024);
                        }
                    }
                    {
                        const char* A = sAb + (a_rd + a_off[T]);
                        const char* Bt = sBb + ((bcur << 13) + b_lane);
                        uint4 af[FM], bf[4];
                        if constexpr ((VT_SPAN3_ABLATE & 4) != 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) bf[j] = make_uint4(lane + j, T, (unsigned)(unsigned long)Bt, 0x3f803f80u);
#pragma unroll
                            for (int i = 0; i < FM; ++i) {
                                const char* src = ((fmask[i] >> T) & 1u) ? A : sZb - i * 1024;
                                af[i] = make_uint4(0x3f803f80u, lane * 3 + i, T, (unsigned)(unsigned long)src);
                            }
                        } else {
                            bf[0] = *(const uint4*)(Bt);
                            bf[1] = *(const uint4*)(Bt + 256);
                            bf[2] = *(const uint4*)(Bt + 2048);
                            bf[3] = *(const uint4*)(Bt + 2304);
#pragma unroll
                            for (int i = 0; i < FM; ++i) {
                                // (sZb - i*1024) + i*1024 == the zero block: the constant stays in the offset field
                                const char* src = ((fmask[i] >> T) & 1u) ? A : sZb - i * 1024;
                                af[i] = *(const uint4*)(src + i * 1024);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < FM; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if constexpr ((VT_SPAN3_ABLATE & 1) != 0)
                                    asm volatile("" ::"v"(af[i].x), "v"(af[i].y), "v"(af[i].z), "v"(af[i].w), "v"(bf[j].x), "v"(bf[j].y), "v"(bf[j].z), "v"(bf[j].w));
                                else
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf16x8, bf[j]), __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
                            }
                    }
                    --gleft;
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

            // ---- epilogue: two 16-byte stores per row fragment, straight from the accumulators ----------
            ArgsPtr Q = fresh_args();
            const bool affine = Q->p.flags & VT_CONV_AFFINE, relu = Q->p.flags & VT_CONV_RELU;
            const bool stats = Q->p.flags & VT_CONV_STATS, has_res = (Q->p.flags & VT_CONV_RESIDUAL) != 0;
            const int Cout_ = Q->p.Cout, M_ = Q->p.M, ldy_ = Q->p.ldy, ldr_ = Q->p.ldr;
            const bool dense_ = Q->p.dense_out;
            bf16_t* __restrict__ yg = (bf16_t*)Q->p.y;
            const bf16_t* __restrict__ rg = (const bf16_t*)Q->p.res;
            const float* scale_ = Q->p.scale;
            const float* shift_ = Q->p.shift;
            float* stats_ = Q->p.stats;
            float s1[16], s2[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) s1[e] = 0.f, s2[e] = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int tr = wrow + i * 16;  // row inside the tile
                const bool row_ok = tr < rows_tile && m0_cur + tr < (long)M_;
                const long po = dense_ ? m0_cur + tr : (long)sPo[par * L::BMX + tr];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = ch0 + h * 32;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = acc[i][2 * h + (e >> 2)][e & 3];
                        if (affine) {
                            const int ne = min(n + e, Cout_ - 1);
                            t = fmaf(t, scale_ ? scale_[ne] : 1.f, shift_[ne]);
                        }
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
                                s1[h * 8 + e] += r8[e];
                                s2[h * 8 + e] = fmaf(r8[e], r8[e], s2[h * 8 + e]);
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
                        if constexpr ((VT_SPAN3_ABLATE & 8) == 0) *(uint4*)(yg + (po * ldy_ + n)) = out;
                        else asm volatile("" ::"v"(out.x), "v"(out.y), "v"(out.z), "v"(out.w));
                    }
                }
            }
            if (stats) {
                // sum over the 16 pixel lanes (same q4): butterfly, then lane c16 == e keeps channel e
                const int rep = (int)((m0_cur / UNIT) % kStatReplicas);
                float u = 0.f, v = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float x1 = s1[e], x2 = s2[e];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        x1 += __shfl_xor(x1, o, 64);
                        x2 += __shfl_xor(x2, o, 64);
                    }
                    u = c16 == e ? x1 : u;
                    v = c16 == e ? x2 : v;
                }
                const int n = ch0 + (c16 >> 3) * 32 + (c16 & 7);
                if (n < Cout_) {
                    vt_stat_add(stats_, ((long)rep * 2 + 0) * Cout_ + n, u);
                    vt_stat_add(stats_, ((long)rep * 2 + 1) * Cout_ + n, v);
                }
            }
        };
        switch (fm) {
            case 4: run(I_<4>{}); break;
            case 5: run(I_<5>{}); break;
            case 6: run(I_<6>{}); break;
            default: run(I_<7>{}); break;
        }
        f_cur = f_nxt;
        m0_cur = m0_nxt;
    }
    set_m0(m0_keep);
#undef VT_ISSUE_A_PIECE
#undef VT_ISSUE_B3
#undef VT_ROW_TABLES
#undef VT_TILE_U0
#undef VT_TILE_F
}

template <int WM>
int launch3(S3Args& a, int wgs_per_cu, hipStream_t st) {
    using L = L3<WM>;
    IgemmArgs& p = a.p;
    constexpr int UNIT = 16 * WM;
    p.tiles_n = (p.Cout + 127) / 128;
    const int g8 = 32 * wgs_per_cu;  // workgroups per XCD (32 CUs each)
    if (g8 % p.tiles_n != 0) return -1;
    a.rslots = g8 / p.tiles_n;
    a.units = (p.M + UNIT - 1) / UNIT;
    a.upx = (a.units + 7) / 8;
    a.aslot_rows = ((16 * WM * kFMX + a.halo + 15) / 16) * 16;
    if ((a.aslot_rows / 16 + 2 * WM - 1) / (2 * WM) > 6) return -1;  // span pieces per wave: one per tap 0..5
    const int smem = L::bytes(a.aslot_rows);
    if (smem * wgs_per_cu > 160 * 1024) return -1;
    auto kern = span3_kernel<WM>;
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(span3)");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("span3_kernel<bf16,WM=%d,%dwg/cu>", WM, wgs_per_cu);
    hipLaunchKernelGGL(kern, dim3(8 * g8), dim3(128 * WM), smem, st, a);
    VT_CHECK_LAUNCH("vt_conv_igemm(span3)");
    return VT_OK;
}

}  // namespace

// returns -1 when this kernel does not apply (the caller then tries vt_span_dispatch)
int vt_span3_dispatch(IgemmArgs& a0, int dtype, void* stream) {
    const int enabled = getenv("VT_SPAN3") ? atoi(getenv("VT_SPAN3")) : 0;  // TODO static once settled
    if (!enabled || dtype != VT_BF16) return -1;
    if (a0.sh != 1 || a0.sw != 1 || a0.Ho != a0.Hi || a0.Wo != a0.Wi) return -1;
    if (a0.Cin % 32 != 0 || a0.ntaps != 9 || a0.Cout < 64) return -1;
    if ((long)a0.M + 2L * a0.Wi * VT_MAX_TAPS > 0x7fffffffL) return -1;
    if ((long)a0.B * a0.oH * a0.oW > 0x7fffffffL) return -1;
    if ((unsigned long)a0.M * a0.ldx * 2 >= 0xffff0000ul) return -1;
    int dmin = 1 << 30, dmax = -(1 << 30);
    for (int t = 0; t < a0.ntaps; ++t) {
        const int d = (a0.h0 + a0.dh[t]) * a0.Wi + (a0.w0 + a0.dw[t]);
        dmin = d < dmin ? d : dmin;
        dmax = d > dmax ? d : dmax;
    }
    // MFMA-bound layers only: enough rows to give every CU at least 4 units
    const int wm_env = getenv("VT_SPAN3_WM") ? atoi(getenv("VT_SPAN3_WM")) : 4;
    static const int fast_dma = getenv("VT_SPAN_FAST_DMA") ? atoi(getenv("VT_SPAN_FAST_DMA")) : 1;
    S3Args a;
    a.p = a0;
    a.dmin = dmin;
    a.halo = dmax - dmin;
    a.fast_dma = fast_dma;
    const long tiles_n = (a0.Cout + 127) / 128;
    if (wm_env == 4) {
        if ((long)a0.M * tiles_n < 256L * 64 * 4) return -1;
        return launch3<4>(a, 1, (hipStream_t)stream);
    }
    if ((long)a0.M * tiles_n < 512L * 32 * 4) return -1;
    return launch3<2>(a, 2, (hipStream_t)stream);
}
