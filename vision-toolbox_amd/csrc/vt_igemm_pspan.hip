// vt_igemm_pspan.hip -- persistent input-span convolution with the WHOLE filter resident in LDS (bf16), one 12-wave
// workgroup per CU: eight compute waves + four LDS-DMA loader waves.  For the short-K, HBM-bound convolutions of the
// first stages: the 3x3 ConvNormAct units with <= 128 output channels (reference components.py:26-35 inside
// backbones/darknet.py:23-24,35,43 and vovnet.py:41-44,84-88), their stride-1 data gradients, the depth-to-space data
// gradient of the stride-2 layers (VT_CONV_D2S) and -- through the space-to-depth view of vt_igemm_span.hip -- the 3x3
// stride-2 convolution that opens a stage.
//
// Why (round 4; measured on 32 -> 32 3x3 @112x112 and 32 -> 64 3x3 stride 2 @224x224, batch 256): on these layers a
// one-tile workgroup of vt_igemm_span.hip lives ~12 us of which 4.7 are the latency of its first loads, ~2 its
// epilogue, and the nine K-steps in between cost a barrier + a counted wait each for 4-16 MFMAs per wave.  Neither more
// slices in flight (209 -> 253 us) nor fewer staged bytes (the space-to-depth view: 445 -> 618 us) helps: what is
// missing is overlap of one tile's latency with another tile's work, and steps that need no synchronisation.  Here
//   * the filter (K x BN <= 72 KB) is staged ONCE per workgroup and stays: a K-step is fragment reads + MFMAs, nothing
//     else -- no barrier, no wait, no DMA issue in a compute wave (vt_igemm_span6.hip's rule);
//   * a workgroup is PERSISTENT over a contiguous range of 256- (or 128-) row tiles of its XCD; the input span of the
//     next channel chunk / next tile is in flight while the current one is multiplied and stored: one workgroup barrier
//     per channel chunk, behind which the loaders issue the span set that is 1 (2) chunks ahead;
//   * swapped MFMA operands as in span6: a lane ends with 8 consecutive output channels of one pixel and stores 16-byte
//     segments straight from the accumulators; the BatchNorm statistics stay in registers over all tiles of the
//     workgroup and leave once;
//   * padding by per-row tap masks (a masked fragment reads a zero block), so spans are plain runs of pixels and the
//     loaders' addresses are a per-tile constant plus the chunk's offset.
// The summation order over K is (chunk, tap) as in vt_igemm_span.hip; for stride 2 the taps of a chunk are visited plane
// by plane.  Results agree with the other conv kernels to the last bf16 rounding of the output, not bit for bit.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

// ablations exist only in -DVT_PSPAN_DIAG builds (tools/pspan_ablate.sh): results are wrong by construction, only the time is read
#ifdef VT_PSPAN_DIAG
#define VT_PDBG(bit) (a.debug & (bit))
#else
#define VT_PDBG(bit) false
#endif
constexpr int kMaxSteps = 16;  // K-steps per channel chunk (taps), at most
constexpr int kMaxP = 8;       // span pieces (16 rows x 64 B) per loader wave and plane, at most


struct PsArgs {
    IgemmArgs p;
    int s2;             // 0: stride-1 grid; 1: stride 2 through the space-to-depth view (four parity planes)
    int nplanes;        // span planes per channel chunk: 1, or 4 (s2)
    int nchunks;        // Cin / 32
    int ntaps;          // K-steps per channel chunk
    int nbuf;           // span-set buffers (2..4): nbuf - 1 sets in flight
    int set_bytes;      // bytes of one span set (all planes of one chunk)
    int tiles_m;        // BM-row tiles
    int tpx;            // tiles per XCD
    int dmin;           // stride 1: first span row = pixel m0 + dmin
    int plane_slot[4];  // byte offset of plane q's span inside a set
    int plane_np[4];    // its length in 16-row pieces
    int plane_off[4];   // s2: byte offset of the plane's pixel inside the 2 x 2 block, (a*W + c) * ldx * 2
    int delta[4];       // s2: byte distance the plane's span starts before the tile's first block
    int wrap;           // s2: extra byte distance when one plane column back leaves the row
    int maxoff;         // s2: byte offset of the block under the last plane position
    int off_filter, off_sets, off_zero;  // LDS layout (bytes)
    int debug;  // diagnostic builds: 1 no row tables after the first tile, 2 no stores, 4 no MFMA steps, 8 no span DMA after the prologue, 16 no tile_offsets
    short t_drow[kMaxSteps];             // step t of a chunk: row offset of its fragments inside the plane's span,
    int8_t t_q[kMaxSteps], t_rs[kMaxSteps];  // its plane and its filter tap
};

__device__ __forceinline__ int swz4(int g) { return (0x1320 >> ((g & 3) * 4)) & 3; }  // filter-slice image
__device__ __forceinline__ int swzA(int g) { return (g & 1) << 1; }                   // span image (span6's)

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long)hi << 32) | lo);
}
// LDS-DMA, 16 B per lane: LDS address = M0 + lane*16, global address = sbase + voff.  (s_nop 4: the scalar base may have
// just been written by a VALU instruction; a VMEM instruction reading such an SGPR needs 5 wait states, which hipcc does
// not insert in front of an asm statement.)
__device__ __forceinline__ void glds_s(unsigned voff, const void* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1"
                 ::"v"(voff), "s"(uniform_ptr(sbase)), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                 : "memory");
}
// the same with operands the caller has already made wave-uniform (one readfirstlane per plane, not three per piece)
__device__ __forceinline__ void glds_u(unsigned voff, const void* sbase_uniform, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1"
                 ::"v"(voff), "s"(sbase_uniform), "s"(lds_addr_uniform)
                 : "memory");
}
__device__ __forceinline__ unsigned get_m0() {
    unsigned v;
    asm volatile("s_mov_b32 %0, m0" : "=s"(v)::"memory");
    return v;
}
__device__ __forceinline__ void set_m0(unsigned v) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(__builtin_amdgcn_readfirstlane(v)) : "memory");
}
template <int N>
__device__ __forceinline__ void vmw() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void vm_wait_dyn(int n) {  // wave-uniform count; waits a little early above 32
    if (n <= 0) { vmw<0>(); return; }
#define VT_W8(b)                                 \
    switch (n - (b)) {                           \
        case 0: vmw<(b) + 0>(); break;           \
        case 1: vmw<(b) + 1>(); break;           \
        case 2: vmw<(b) + 2>(); break;           \
        case 3: vmw<(b) + 3>(); break;           \
        case 4: vmw<(b) + 4>(); break;           \
        case 5: vmw<(b) + 5>(); break;           \
        case 6: vmw<(b) + 6>(); break;           \
        default: vmw<(b) + 7>(); break;          \
    }
    if (n < 8) { VT_W8(0) }
    else if (n < 16) { VT_W8(8) }
    else if (n < 24) { VT_W8(16) }
    else if (n < 32) { VT_W8(24) }
    else vmw<32>();
#undef VT_W8
}
__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// n / d and n % d for 0 <= n < 2^31 by multiply-high with magic = min(ceil(2^32 / d), 2^32 - 1) and one correction each way
__device__ __forceinline__ int div_magic(int n, int d, unsigned magic, int& rem) {
    int q = (int)__umulhi((unsigned)n, magic);
    int r = n - q * d;
    if (r < 0) r += d, --q;
    if (r >= d) r -= d, ++q;
    rem = r;
    return q;
}
__host__ __device__ inline unsigned magic_of(int d) {
    const unsigned long long m = (0x100000000ull + (unsigned)d - 1) / (unsigned)d;
    return m > 0xffffffffull ? 0xffffffffu : (unsigned)m;
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float x) {  // sum over the 16 lanes of a DPP row, in every lane of it
    x = dpp_add<0x128>(x);
    x = dpp_add<0x124>(x);
    x = dpp_add<0x122>(x);
    return dpp_add<0x121>(x);
}

// LDS map (bytes): [tap table 256][row tables 2 x (mask u32[BM] | output pixel i32[BM])][filter nsteps x BN x 64]
//                  [span sets nbuf x set_bytes][zero block 64]
template <int BM>
struct PL {
    static constexpr int kTab = 256;
    static constexpr int kFilter = kTab + 2 * BM * 8;
};

// 12 waves: 0-7 compute (WM x WN, a wave owns 16*FM rows x 32 output channels), 8-11 loaders
// MODE: epilogue, 0 plain (+ residual), 1 BatchNorm statistics, 2 affine (+ ReLU, + residual)
// NT: K-steps per channel chunk known at compile time (9: 3x3; 4: the 2x2-tap data gradients), 0: a run-time loop.  With
// NT > 0 a chunk is straight-line code -- per-lane fragment offsets of every tap held in registers for the whole kernel,
// fragment reads of later steps in flight under the MFMAs of earlier ones; the run-time loop pays two dependent LDS
// round trips per step (tap table, then fragments) and measured 3.9 us per 256-row tile on 32 -> 32 3x3 @112x112 for
// ~0.5 us of MFMAs.
template <int BN, int BM, int MODE, int NT>
__global__ void __launch_bounds__(768, 3) pspan_kernel(const PsArgs a) {
    constexpr int WN = BN / 32, WM = 8 / WN, FM = BM / (16 * WM);
    static_assert(BN == 32 || BN == 64 || BN == 128, "filter tile width");
    static_assert(FM >= 1 && WM * 16 * FM == BM, "tile height");
    const IgemmArgs& p = a.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* sTap = (int4*)smem;
    unsigned* sTbl = (unsigned*)(smem + PL<BM>::kTab);  // [2][2][BM]: masks, output pixels
    const char* sF = smem + a.off_filter;
    const char* sS = smem + a.off_sets;
    const char* sZ = smem + a.off_zero;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's tiles: a contiguous range of its XCD's share ---------------------------------------------
    const int bid = blockIdx.x, xcd = bid & 7, l = bid >> 3;  // 32 workgroups per XCD
    const int tx0 = xcd * a.tpx, tx1 = min(a.tiles_m, tx0 + a.tpx);
    const int nx = max(0, tx1 - tx0);
    const int t0 = __builtin_amdgcn_readfirstlane(tx0 + (int)((unsigned)l * (unsigned)nx / 32u));
    const int t1 = __builtin_amdgcn_readfirstlane(tx0 + (int)((unsigned)(l + 1) * (unsigned)nx / 32u));
    const int ntile = t1 - t0;
    if (ntile <= 0) return;
    const int nchunks = a.nchunks, ntaps = a.ntaps, nbuf = a.nbuf;
    const int G = ntile * nchunks;  // span sets (= barriers after the first) of this workgroup

    if (tid < ntaps) sTap[tid] = make_int4(a.t_drow[tid], a.t_rs[tid], a.plane_slot[a.t_q[tid]], 0);
    if (tid >= 64 && tid < 68) ((unsigned*)sZ)[tid - 64] = 0u;

    if (wave >= 8) {
        // =========================== loader waves ==================================================================
        const int lj = wave - 8;
        const char* xg = (const char*)p.x;
        const char* wg = (const char*)p.w;
        const unsigned f_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + a.off_filter);
        const unsigned s_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + a.off_sets);
        const unsigned m0_keep = get_m0();
        const int ldx2 = p.ldx * 2;
        const int cjA = (lane & 3) ^ swzA(lane >> 4);
        const int M = p.M;

        // ---- the filter, once: piece id -> (step s, 16-row group q); row n = 16 q + (lane >> 2), K offset of step s ----
        {
            const int ppf = BN / 16;  // pieces per slice
            const int npieces = nchunks * ntaps * ppf;
            int inflight = 0;
            for (int id = lj; id < npieces; id += 4) {
                const int s = id / ppf, q = id - s * ppf;
                const int ic = s / ntaps, it = s - ic * ntaps;
                const int n = min(16 * q + (lane >> 2), p.Cout - 1);  // (rows past Cout: their outputs are never stored)
                const int cj = (lane & 3) ^ swz4(2 * q + (lane >> 5));
                const unsigned voff = (unsigned)((n * p.ldw + cj * 8) * 2);
                const char* sb = wg + ((long)a.t_rs[it] * p.Cin + (long)ic * 32) * 2;
                glds_s(voff, sb, f_base + (unsigned)(id * 1024));
                if (++inflight == 24) {  // (the counter is 6 bits wide)
                    vmw<8>();
                    inflight = 8;
                }
            }
        }

        // ---- span sets ----------------------------------------------------------------------------------------------
        // Per tile: the byte offset in x of this lane's row of piece lj + 4 P (plane 0 / the tile's first block), plus its
        // chunk position.  stride 1: pixel m0 + dmin + r clamped into the tensor (a clamped row is only ever read through
        // a masked tap).  stride 2: block (2i, 2j) under plane position m0 + r; bit P of `wrapm`: first block of its row.
        // Per tile: the byte offset in x of this lane's row of piece lj + 4 P (plane 0 / the tile's first block), plus its
        // chunk position.  stride 1: pixel m0 + dmin + r clamped into the tensor (a clamped row is only ever read through
        // a masked tap).  stride 2: block (2i, 2j) under plane position m0 + r; bit P of `wrapm`: first block of its row.
        // Tiles of a workgroup are consecutive, so after the first one every quantity advances by additions only
        // (26 us of a 139 us launch went into eight 64-bit multiplies + clamps / divisions per tile and loader).
        int poff[kMaxP];    // what issue_set uses
        int praw[kMaxP];    // stride 1: unclamped byte offset; stride 2: 4 * v0 * ldx2 + chunk position
        int pj[kMaxP];      // stride 2: column j0 of plane position v0
        unsigned wrapm = 0;
        const unsigned wo_magic = magic_of(p.Wo);
        const int lo_b = cjA * 16, hi_b = (int)((long)(M - 1) * ldx2) + cjA * 16;
        const int step_b = a.s2 ? 4 * BM * ldx2 : BM * ldx2;
        const int jstep = a.s2 ? BM % p.Wo : 0;
        auto tile_offsets_first = [&](int tile) {
            const long m0 = (long)tile * BM;
#pragma unroll
            for (int P = 0; P < kMaxP; ++P) {
                const long r = 16 * (lj + 4 * P) + (lane >> 2);
                if (a.s2) {
                    const long v0 = m0 + r;
                    int j0;
                    (void)div_magic((int)v0, p.Wo, wo_magic, j0);
                    praw[P] = (int)(4 * v0 * (long)ldx2) + cjA * 16;
                    pj[P] = j0;
                } else {
                    praw[P] = (int)((m0 + a.dmin + r) * (long)ldx2) + cjA * 16;
                    pj[P] = 0;
                }
            }
        };
        auto tile_offsets_next = [&]() {
#pragma unroll
            for (int P = 0; P < kMaxP; ++P) {
                praw[P] += step_b;
                if (a.s2) {
                    pj[P] += jstep;
                    pj[P] -= pj[P] >= p.Wo ? p.Wo : 0;
                }
            }
        };
        auto tile_offsets_use = [&]() {
            wrapm = 0;
#pragma unroll
            for (int P = 0; P < kMaxP; ++P) {
                if (a.s2) {
                    poff[P] = praw[P] - 2 * pj[P] * ldx2;
                    wrapm |= (pj[P] == 0 ? 1u : 0u) << P;
                } else {
                    poff[P] = min(max(praw[P], lo_b), hi_b);
                }
            }
        };
        int off_tile = -1;
        // plane parameters in scalar registers (a kernel-argument array indexed inside the per-tile loop is a scalar load
        // with its latency in front of every DMA instruction that depends on it)
        int pl_np[4], pl_dq[4], pl_slot[4];
        long pl_off[4];
        int set_pieces = 0;  // DMA instructions of this wave per set
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool on = q < a.nplanes;
            pl_np[q] = on ? a.plane_np[q] : 0;
            pl_dq[q] = on ? a.delta[q] : 0;
            pl_slot[q] = on ? a.plane_slot[q] : 0;
            pl_off[q] = on ? (long)a.plane_off[q] : 0l;
            set_pieces += on ? (pl_np[q] - lj + 3) / 4 : 0;
        }
        const int s2 = a.s2, wrap = a.wrap, maxoff48 = a.maxoff + 48, set_bytes = a.set_bytes;
        auto issue_set = [&](int gcs) {  // set gcs = (tile t0 + gcs / nchunks, chunk gcs % nchunks)
            const int tl = gcs / nchunks, ic = gcs - tl * nchunks;
            if (tl != off_tile) {  // (sets are issued in order: the next tile, if not the same one)
                if (off_tile < 0) tile_offsets_first(t0 + tl);
                else tile_offsets_next();
                tile_offsets_use();
                off_tile = tl;
            }
            if (VT_PDBG(8) && gcs >= nbuf - 1) return;
            const unsigned buf = s_base + (unsigned)((gcs % nbuf) * set_bytes);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= a.nplanes) break;
                const void* sb = uniform_ptr(xg + (pl_off[q] + (long)ic * 64));
                const int np = pl_np[q], dq = pl_dq[q];
                const bool back1 = s2 && (q == 0 || q == 2);
                const unsigned lds0 = __builtin_amdgcn_readfirstlane(buf + (unsigned)pl_slot[q] + (unsigned)(lj * 1024));
#pragma unroll
                for (int P = 0; P < kMaxP; ++P) {
                    if (lj + 4 * P < np) {
                        int off = poff[P];
                        if (s2) {
                            off -= dq + ((back1 && ((wrapm >> P) & 1u)) ? wrap : 0);
                            off = min(max(off, 0), maxoff48);
                        }
                        glds_u((unsigned)off, sb, lds0 + (unsigned)(P * 4096));
                    }
                }
            }
        };
        // row tables of tile `tl` (local index) into half tl & 1: which taps stay inside the image, where the row goes.
        // Quotients by multiply-high with one correction (exact for m < 2^31 / d); with NT > 0 the taps' offsets sit in
        // scalar registers and the loop over them is straight-line code.
        int th[NT > 0 ? NT : 1], tw[NT > 0 ? NT : 1];
        if constexpr (NT > 0) {
#pragma unroll
            for (int T = 0; T < NT; ++T) th[T] = p.h0 + p.dh[T], tw[T] = p.w0 + p.dw[T];
        }
        const int gw = p.Wo, ghw = p.Ho * p.Wo;
        const unsigned gw_magic = magic_of(gw), ghw_magic = magic_of(ghw);
        auto row_tables = [&](int tl) {
            constexpr int QR = BM / 4;
            unsigned* tm_ = sTbl + (tl & 1) * 2 * BM;
            for (int rr = lane; rr < QR; rr += 64) {
                const int r = lj * QR + rr;
                const long m = (long)(t0 + tl) * BM + r;
                unsigned bits = 0;
                int po = 0;
                if (m < M) {
                    int rem, oj;
                    const int b = div_magic((int)m, ghw, ghw_magic, rem);
                    const int oi = div_magic(rem, gw, gw_magic, oj);
                    const int ih0 = oi * p.sh, iw0 = oj * p.sw;
                    if constexpr (NT > 0) {
#pragma unroll
                        for (int T = 0; T < NT; ++T)  // bit = filter tap index (the taps of a launch are T = 0 .. NT-1)
                            if ((unsigned)(ih0 + th[T]) < (unsigned)p.Hi && (unsigned)(iw0 + tw[T]) < (unsigned)p.Wi) bits |= 1u << T;
                    } else {
                        for (int t = 0; t < p.ntaps; ++t) {
                            const int ih = ih0 + p.h0 + p.dh[t], iw = iw0 + p.w0 + p.dw[t];
                            if ((unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi) bits |= 1u << t;
                        }
                    }
                    po = (b * p.oH + (oi * p.oHs + p.oh0)) * p.oW + (oj * p.oWs + p.ow0);
                }
                tm_[r] = bits;
                tm_[BM + r] = (unsigned)po;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };

        // prologue: sets 0 .. nbuf-2, the first tile's tables; everything landed before barrier 0
        for (int g = 0; g < nbuf - 1 && g < G; ++g) issue_set(g);
        row_tables(0);
        vmw<0>();
        wg_barrier();  // barrier 0
        for (int gc = 0; gc < G; ++gc) {
            // behind barrier gc: chunk gc-1 has been left by every compute wave, its buffer takes set gc + nbuf - 1
            const int gn = gc + nbuf - 1;
            if (gn < G) issue_set(gn);
            // the next tile's tables, while this tile's last chunk is multiplied (the half they go to was last read by
            // the epilogue of the tile before this one)
            const int tl = gc / nchunks;
            if (gc - tl * nchunks == nchunks - 1 && tl + 1 < ntile && !VT_PDBG(1)) row_tables(tl + 1);
            // set gc+1 (everything but this wave's share of the younger sets) has landed before barrier gc+1
            {
                const int younger = min(nbuf - 2, max(0, G - 2 - gc));  // sets gc+2 .. issued so far
                if (younger <= 0) vmw<0>();
                else vm_wait_dyn(younger * set_pieces);
            }
            wg_barrier();  // barrier gc + 1
        }
        vmw<0>();
        set_m0(m0_keep);
        if (MODE == 1) wg_barrier();  // (the statistics fold below: every wave of the workgroup takes its barrier)
        return;
    }

    // =============================== compute waves ================================================================
    const int wm = wave / WN, wn = wave % WN;
    const int q4 = lane >> 4, c16 = lane & 15;
    const int wrow = wm * 16 * FM + c16;  // this lane's row inside the tile, fragment 0
    // this lane's output channels: wn*32 + q4*8 + e, e = 0..7 (filter fragment j = e >> 2)
    const int ch0 = wn * 32 + q4 * 8;
    // filter fragment j of this lane: MFMA row r = c16 -> slice row wn*32 + (r>>2)*8 + j*4 + (r&3); (row >> 3) & 3 = r >> 2
    const int nb0 = wn * 32 + (c16 >> 2) * 8 + (c16 & 3);
    const int b_lane = (nb0 * 4 + (q4 ^ swz4(c16 >> 2))) * 16;  // byte offset inside a slice; j adds 256

    const bool affine = MODE == 2, stats = MODE == 1;
    const bool relu = MODE == 2 && (p.flags & VT_CONV_RELU);
    const bool has_res = MODE != 1 && (p.flags & VT_CONV_RESIDUAL) != 0;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = 0.f, s2[e] = 0.f;
    float sc[8], sf[8];
    if (affine) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ne = min(ch0 + e, p.Cout - 1);
            sc[e] = p.scale ? p.scale[ne] : 1.f;
            sf[e] = p.shift[ne];
        }
    }
    bf16_t* __restrict__ yg = (bf16_t*)p.y;
    const bf16_t* __restrict__ rg = (const bf16_t*)p.res;
    const long ycol = vt_out_col(p, ch0, p.ldy), rcol = vt_out_col(p, ch0, p.ldr);
    const bool col_ok = ch0 < p.Cout;

    // NT > 0: byte offset (inside a span set) of this lane's fragment-0 row for every step of a chunk, and the steps' taps
    int aoff[NT > 0 ? NT : 1], rsv[NT > 0 ? NT : 1];
    unsigned full_mask = 0;
    if constexpr (NT > 0) {
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const int srow = wrow + a.t_drow[T];
            aoff[T] = a.plane_slot[a.t_q[T]] + (srow * 4 + (q4 ^ swzA(srow >> 2))) * 16;
            rsv[T] = a.t_rs[T];
            full_mask |= 1u << a.t_rs[T];
        }
    }
    wg_barrier();  // barrier 0: filter, set 0 and the first tables are in LDS
    int gc = 0;
    for (int tl = 0; tl < ntile; ++tl) {
        if (tl > 0) wg_barrier();  // barrier gc: the tile's first set has landed, its row tables are written
        const unsigned* tm_ = sTbl + (tl & 1) * 2 * BM;
        unsigned fmask[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) fmask[i] = tm_[wrow + i * 16];
        f32x4 acc[FM][2];
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

        // the residual rows of the tile (the accumulate of a data gradient, DarknetBlock's shortcut in inference): requested
        // before the steps, used after them (clamped addresses: nothing orders a load behind a store)
        const long m0 = (long)(t0 + tl) * BM;
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t rres[FM];
        // (asm: as a C++ load the compiler orders it behind the previous tile's stores -- y and the residual may be the
        //  same buffer -- with s_waitcnt vmcnt(0) at the head of every tile: 1.3 us of store latency per tile, 32 -> 32 3x3
        //  @112x112 150 -> 180 us.  A lane's residual element IS the one it stores to later in this tile and nothing else
        //  in the launch touches it, so no order is needed; the wait in front of the first use is explicit.)
        if (MODE != 1 && has_res) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int tr = wrow + i * 16;
                const long po = (int)tm_[BM + tr];
                const bool ok = m0 + tr < p.M && col_ok;
                const bf16_t* src = rg + (ok ? po * p.ldr + rcol : 0l);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rres[i]) : "v"(src) : "memory");
            }
        }
        // (interior tiles: every tap of every row inside the image -- no selects in the steps)
        bool all_in = true;
#pragma unroll
        for (int i = 0; i < FM; ++i) all_in = all_in && fmask[i] == full_mask;
        const bool interior = NT > 0 && __all(all_in);

        for (int ic = 0; ic < nchunks; ++ic, ++gc) {
            if (ic > 0) wg_barrier();  // barrier gc: set gc has landed
            const char* setb = sS + (gc % nbuf) * a.set_bytes;
            const char* fb = sF + (long)(ic * ntaps) * (BN * 64) + b_lane;
            if (VT_PDBG(4)) continue;
            if constexpr (NT > 0) {
                auto chunk = [&](auto maskedc) {
                    constexpr bool MASKED = decltype(maskedc)::value;
                    // D steps of fragments in flight in a ring of D + 1 register sets: every read of step T + D is issued
                    // before the MFMAs of step T (left alone the compiler reads one fragment, waits for it and multiplies --
                    // an LDS round trip per two MFMAs: 0.73 us per 9-step tile instead of 0.15)
                    // (FM = 4: one step ahead -- two more sets of 24 registers beside 32 accumulators, the residual rows and
                    //  the per-tap offsets spill at the 168 registers three waves per SIMD leave)
                    constexpr int D0 = 3;
                    constexpr int D = D0 < NT ? D0 : (NT - 1 > 0 ? NT - 1 : 1);
                    if constexpr (FM > 2) {
                        // (FM = 4: 8 MFMAs per step cover most of a read's latency, and a ring of fragment sets beside 32
                        //  accumulators, the residual rows / statistics and the per-tap offsets spills at the 168 registers
                        //  three waves per SIMD leave: the compiler's own order, no ring)
#pragma unroll
                        for (int T = 0; T < NT; ++T) {
                            const char* Bt = fb + T * (BN * 64);
                            const uint4 bf0 = *(const uint4*)(Bt);
                            const uint4 bf1 = *(const uint4*)(Bt + 256);
                            const char* A = setb + aoff[T];
                            uint4 pf[FM];
#pragma unroll
                            for (int i = 0; i < FM; ++i) {
                                const char* src = (!MASKED || ((fmask[i] >> rsv[T]) & 1u)) ? A + i * 1024 : sZ;
                                pf[i] = *(const uint4*)src;
                            }
#pragma unroll
                            for (int i = 0; i < FM; ++i) {
                                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf0),
                                                                                    __builtin_bit_cast(bf16x8, pf[i]), acc[i][0], 0, 0, 0);
                                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf1),
                                                                                    __builtin_bit_cast(bf16x8, pf[i]), acc[i][1], 0, 0, 0);
                            }
                        }
                        return;
                    }
                    uint4 pfr[D + 1][FM], b0r[D + 1], b1r[D + 1];
                    auto load_step = [&](auto Tc) {
                        constexpr int T = decltype(Tc)::value;
                        constexpr int S = T % (D + 1);
                        const char* Bt = fb + T * (BN * 64);
                        b0r[S] = *(const uint4*)(Bt);
                        b1r[S] = *(const uint4*)(Bt + 256);
                        const char* A = setb + aoff[T];
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            const char* src = (!MASKED || ((fmask[i] >> rsv[T]) & 1u)) ? A + i * 1024 : sZ;
                            pfr[S][i] = *(const uint4*)src;
                        }
                    };
                    auto mma_step = [&](auto Tc) {
                        constexpr int S = decltype(Tc)::value % (D + 1);
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b0r[S]),
                                                                                __builtin_bit_cast(bf16x8, pfr[S][i]), acc[i][0], 0, 0, 0);
                            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b1r[S]),
                                                                                __builtin_bit_cast(bf16x8, pfr[S][i]), acc[i][1], 0, 0, 0);
                        }
                    };
                    auto step = [&](auto Tc) {
                        constexpr int T = decltype(Tc)::value;
                        if constexpr (T + D < NT) load_step(std::integral_constant<int, (T + D < NT ? T + D : 0)>{});
                        __builtin_amdgcn_sched_barrier(0);
#if VT_MFMA_SETPRIO
                        __builtin_amdgcn_s_setprio(1);
#endif
                        mma_step(Tc);
#if VT_MFMA_SETPRIO
                        __builtin_amdgcn_s_setprio(0);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    auto pre = [&](auto Tc) {
                        constexpr int T = decltype(Tc)::value;
                        if constexpr (T < D && T < NT) load_step(std::integral_constant<int, (T < NT ? T : 0)>{});
                    };
                    pre(std::integral_constant<int, 0>{});
                    pre(std::integral_constant<int, 1>{});
                    pre(std::integral_constant<int, 2>{});
                    auto steps = [&](auto... Ts) { (step(Ts), ...); };
                    if constexpr (NT == 9)
                        steps(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{},
                              std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{},
                              std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{}, std::integral_constant<int, 8>{});
                    else
                        steps(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{},
                              std::integral_constant<int, 3>{});
                };
                if (interior) chunk(std::false_type{});
                else chunk(std::true_type{});
            } else
            for (int it = 0; it < ntaps; ++it) {
                const int4 tp = sTap[it];
                const int d = __builtin_amdgcn_readfirstlane(tp.x);
                const int rs = __builtin_amdgcn_readfirstlane(tp.y);
                const int so = __builtin_amdgcn_readfirstlane(tp.z);
                const int srow = wrow + d;
                const char* A = setb + so + (srow * 4 + (q4 ^ swzA(srow >> 2))) * 16;
                const char* Bt = fb + it * (BN * 64);
                const uint4 bf0 = *(const uint4*)(Bt);
                const uint4 bf1 = *(const uint4*)(Bt + 256);
                uint4 pf[FM];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const char* src = ((fmask[i] >> rs) & 1u) ? A + i * 1024 : sZ;
                    pf[i] = *(const uint4*)src;
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf0),
                                                                        __builtin_bit_cast(bf16x8, pf[i]), acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf1),
                                                                        __builtin_bit_cast(bf16x8, pf[i]), acc[i][1], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: one 16-byte store per row fragment, straight from the accumulators --------------------------------
        if (MODE != 1 && has_res) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(rres[i]));  // (defined from here on: no use moves above the wait)
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int tr = wrow + i * 16;
            const long po = (int)tm_[BM + tr];
            const bool row_ok = m0 + tr < p.M;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = acc[i][e >> 2][e & 3];
                if (affine) t = fmaf(t, sc[e], sf[e]);
                if (relu) t = fmaxf(t, 0.f);
                v[e] = t;
            }
            uint4 out = VecIO<bf16_t>::pack(v);
            if (row_ok && col_ok) {
                if (stats) {
                    float r8[8];
                    VecIO<bf16_t>::unpack(out, r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s1[e] += r8[e];
                        s2[e] = fmaf(r8[e], r8[e], s2[e]);
                    }
                }
                if (MODE != 1 && has_res) {
                    float fv[8], fr[8];
                    VecIO<bf16_t>::unpack(out, fv);
                    VecIO<bf16_t>::unpack(make_uint4(rres[i][0], rres[i][1], rres[i][2], rres[i][3]), fr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) fv[e] += fr[e];
                    out = VecIO<bf16_t>::pack(fv);
                }
                if (!VT_PDBG(2)) *(uint4*)(yg + (po * p.ldy + ycol)) = out;
            }
        }
    }
    wg_barrier();  // barrier G: pairs with the loaders' last one

    if (stats) {
        // sums over the 16 pixel lanes (one DPP row), then over the WM row waves in LDS in a fixed order (the span sets
        // are dead: every DMA has landed and every compute wave is past its last read), one fixed-point atomic per column
        float u = 0.f, v = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x1 = row_sum16(s1[e]), x2 = row_sum16(s2[e]);
            u = c16 == e ? x1 : u;
            v = c16 == e ? x2 : v;
        }
        float* sFold = (float*)(smem + a.off_sets);  // [WM][2][BN]
        if (c16 < 8) {
            sFold[(wm * 2 + 0) * BN + ch0 + c16] = u;
            sFold[(wm * 2 + 1) * BN + ch0 + c16] = v;
        }
    }
    if (stats) {
        wg_barrier();  // (with the loaders: see their exit)
        const float* sFold = (const float*)(smem + a.off_sets);
        const int rep = (int)((unsigned)blockIdx.x % (unsigned)kStatReplicas);
        for (int i = tid; i < 2 * BN; i += 512) {  // (tid < 512: the compute waves)
            const int which = i / BN, col = i - which * BN;
            if (col < p.Cout) {
                float acc_ = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) acc_ += sFold[(w * 2 + which) * BN + col];
                vt_stat_add(p.stats, ((long)rep * 2 + which) * p.Cout + col, acc_);
            }
        }
    }
}

template <int BN, int BM>
int ps_launch(PsArgs& a, hipStream_t st, bool dry) {
    IgemmArgs& p = a.p;
    const int nsteps = a.nchunks * a.ntaps;
    a.off_filter = PL<BM>::kFilter;
    a.off_sets = a.off_filter + nsteps * BN * 64;
    // span planes: stride 1: one span of BM + (dmax - dmin) rows; stride 2: the four parity planes
    int off = 0;
    for (int q = 0; q < a.nplanes; ++q) {
        if ((a.plane_np[q] + 3) / 4 > kMaxP) return -1;
        a.plane_slot[q] = off;
        off += a.plane_np[q] * 1024;
    }
    a.set_bytes = off;
    const int budget = 160 * 1024 - 64 - a.off_sets;
    if (budget < 2 * a.set_bytes) return -1;
    // span sets in flight = nbuf - 1: a tile's period cannot be shorter than (issue -> landed) / (sets in flight), and
    // that latency is 2-4 us under load (measured, 32 -> 32 3x3 @112x112: one set in flight 193 us per launch)
    a.nbuf = budget / a.set_bytes;
    const int cap = (4);
    a.nbuf = a.nbuf > cap ? cap : a.nbuf;
    if ((a.nbuf - 2) * ((a.set_bytes / 1024 + 3) / 4 + a.nplanes) > 30) a.nbuf = 2 + 30 / ((a.set_bytes / 1024 + 3) / 4 + a.nplanes);
#ifdef VT_PSPAN_DIAG
    a.debug = VT_KNOB("VT_PSPAN_ABL", 0);
    if ((a.debug & 32) && a.nbuf > 2) a.nbuf = 2;  // (measurement: span sets in flight)
    if ((a.debug & 64) && a.nbuf > 3) a.nbuf = 3;
#endif
    a.off_zero = a.off_sets + a.nbuf * a.set_bytes;
    const int smem = a.off_zero + 64;
    if (a.nbuf * a.set_bytes < 8 * 2 * BN * 4) return -1;  // (the statistics fold reuses the span area)
    a.tiles_m = (p.M + BM - 1) / BM;
    a.tpx = (a.tiles_m + 7) / 8;
    const int mode = (p.flags & VT_CONV_STATS) ? 1 : ((p.flags & VT_CONV_AFFINE) ? 2 : 0);
    if (mode == 1 && (p.flags & (VT_CONV_AFFINE | VT_CONV_RELU | VT_CONV_RESIDUAL))) return -1;
    if (mode == 0 && (p.flags & VT_CONV_RELU)) return -1;
    void (*kern)(const PsArgs) = nullptr;
#define VT_PS_PICK(NTv)                                                                               \
    kern = mode == 1 ? pspan_kernel<BN, BM, 1, NTv> : (mode == 2 ? pspan_kernel<BN, BM, 2, NTv> : pspan_kernel<BN, BM, 0, NTv>)
    static_assert(true, "NT is 9 or 4 or 0: the straight-line chunk lists its steps");
    if (a.ntaps == 9) VT_PS_PICK(9);
    else if (a.ntaps == 4) VT_PS_PICK(4);
    else VT_PS_PICK(0);
#undef VT_PS_PICK
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(pspan)");
        if (rc != VT_OK) return rc;
    }
    if (dry) return VT_OK;
    vt_note_kernel("pspan_kernel<bf16,%d,%d,8+4 waves,%s,nbuf%d>", BM, BN, a.s2 ? "s2d" : "s1", a.nbuf);
    hipLaunchKernelGGL(kern, dim3(256), dim3(768), smem, st, a);
    VT_CHECK_LAUNCH("vt_conv_igemm(pspan)");
    return VT_OK;
}

template <int BM>
int ps_launch_bn(PsArgs& a, hipStream_t st) {
    const int Cout = a.p.Cout;
    if (Cout > 64) {
        // (128 filter columns: 128-row tiles -- with 256 rows a wave holds 64 accumulators + 40 fragment registers per
        //  step and the straight-line chunk spills)
        if constexpr (BM > 128) return -1;
        else return ps_launch<128, BM>(a, st, false);
    }
    if (Cout > 32) return ps_launch<64, BM>(a, st, false);
    return ps_launch<32, BM>(a, st, false);
}

}  // namespace

// returns -1 when this kernel does not apply (the caller then tries the other conv kernels)
int vt_pspan_dispatch(IgemmArgs& a0, int dtype, void* stream) {
    // VT_PSPAN: 0 off, 1 (default) the layers it measured faster on, 2 wherever it applies (tests)
    const int enabled = VT_KNOB("VT_PSPAN", 1);
    if (!enabled || dtype != VT_BF16) return -1;
    if (vt_device_cus() != 256) return -1;  // (the grid and the tile ranges are built for 8 XCDs x 32 CUs)
    if (a0.flags & VT_CONV_NOSTORE) return -1;
    if (a0.Cout > 128 || a0.Cin % 32 != 0 || a0.ntaps > kMaxSteps || a0.ntaps < 1) return -1;
    const bool d2s = (a0.flags & VT_CONV_D2S) != 0;
    if (d2s && (a0.Cout % 32 != 0)) return -1;  // (a lane's 8 channels stay inside one of the four column blocks)
    if ((long)a0.B * a0.Hi * a0.Wi * a0.ldx * 2 >= 0x7fff0000L) return -1;  // (per-lane byte offsets are ints)
    if ((long)a0.B * a0.oH * a0.oW > 0x7fffffffL) return -1;
    if ((long)a0.Cout * a0.ldw * 2 >= 0x7fff0000L) return -1;
    if ((a0.flags & VT_CONV_RESIDUAL) && a0.res) {
        // The residual rows are fetched with asm loads that nothing orders against the tile's later stores to y: sound
        // only while a lane's residual element IS the element it stores later (res == y with the same pixel stride) or the
        // two tensors do not overlap at all.  A shifted or re-strided alias would race silently: the ordered kernels take it.
        const long opix = (long)a0.B * a0.oH * a0.oW;
        const char* y0 = (const char*)a0.y;
        // (channels of one OUTPUT pixel: a depth-to-space launch computes 4 pixels' worth of columns per grid position)
        const long pixc = (a0.flags & VT_CONV_D2S) ? a0.Cout / 4 : a0.Cout;
        const char* y1 = y0 + ((opix - 1) * a0.ldy + pixc) * 2;
        const char* r0 = (const char*)a0.res;
        const char* r1 = r0 + ((opix - 1) * a0.ldr + pixc) * 2;
        const bool overlap = r0 < y1 && y0 < r1;
        if (overlap && !(r0 == y0 && a0.ldr == a0.ldy)) {
            // two channel slices of ONE wider buffer (same pixel stride, disjoint channel ranges) never share an element
            const long cb = pixc * 2, pitch = (long)a0.ldy * 2;
            long off = (r0 - y0) % pitch;
            if (off < 0) off += pitch;
            const bool slices = a0.ldr == a0.ldy && off >= cb && off + cb <= pitch;
            if (!slices) return -1;
        }
    }
    PsArgs a;
    memset(&a, 0, sizeof(a));
    a.p = a0;
    a.nchunks = a0.Cin / 32;
    a.ntaps = a0.ntaps;
    const int nsteps = a.nchunks * a.ntaps;
    const int bn = a0.Cout > 64 ? 128 : (a0.Cout > 32 ? 64 : 32);
    if (nsteps * bn * 64 > 96 * 1024) return -1;  // the resident filter (the span sets must still fit twice: ps_launch)
    const bool s1 = a0.sh == 1 && a0.sw == 1 && a0.Ho == a0.Hi && a0.Wo == a0.Wi;
    const bool s2 = a0.sh == 2 && a0.sw == 2 && a0.ntaps == 9 && a0.h0 == -1 && a0.w0 == -1 && !(a0.Hi & 1) && !(a0.Wi & 1) &&
                    a0.Ho * 2 == a0.Hi && a0.Wo * 2 == a0.Wi && !d2s;
    if (!s1 && !s2) return -1;
    // enough tiles to keep every workgroup busy for a few of them (else the one-tile kernels, which spread better)
    if (enabled < 2 && (long)a0.M < 256L * 256 * 4) return -1;
    // (1x1: one K-step per channel chunk, i.e. a workgroup barrier per 2-16 MFMAs -- the pointwise kernels' territory)
    if (enabled < 2 && a0.ntaps < 4) return -1;
    hipStream_t st = (hipStream_t)stream;
    for (int bm = 256; bm >= 128; bm -= 128) {
        if (s2) {
            for (int t = 0; t < 9; ++t)
                if (a0.dh[t] != t / 3 || a0.dw[t] != t % 3) return -1;
            const int Wo = a0.Wo, W = a0.Wi;
            const int dmins[4] = {-Wo - 1, -Wo, -1, 0};
            const int pa[4] = {1, 1, 0, 0}, pc[4] = {1, 0, 1, 0};
            const long px = (long)a0.ldx * 2;
            a.s2 = 1, a.nplanes = 4;
            for (int q = 0; q < 4; ++q) {
                a.plane_np[q] = (bm - dmins[q] + 15) / 16;
                a.plane_off[q] = (int)((pa[q] * (long)W + pc[q]) * px);
            }
            a.delta[0] = (int)((2L * W + 2) * px), a.delta[1] = (int)(2L * W * px), a.delta[2] = (int)(2 * px), a.delta[3] = 0;
            a.wrap = (int)((long)W * px);
            a.maxoff = (int)((4L * (a0.M - 1) - 2L * (Wo - 1)) * px);
            const int order[9][2] = {{0, 0}, {0, 2}, {2, 0}, {2, 2}, {0, 1}, {2, 1}, {1, 0}, {1, 2}, {1, 1}};
            for (int t = 0; t < 9; ++t) {
                const int r = order[t][0], s_ = order[t][1];
                const int q = ((r & 1) ? 2 : 0) + ((s_ & 1) ? 1 : 0);
                const int d = (r == 0 ? -Wo : 0) + (s_ == 0 ? -1 : 0);
                a.t_q[t] = (int8_t)q, a.t_rs[t] = (int8_t)(3 * r + s_), a.t_drow[t] = (short)(d - dmins[q]);
            }
        } else {
            int dmin = 1 << 30, dmax = -(1 << 30);
            for (int t = 0; t < a0.ntaps; ++t) {
                const int d = (a0.h0 + a0.dh[t]) * a0.Wi + (a0.w0 + a0.dw[t]);
                dmin = d < dmin ? d : dmin;
                dmax = d > dmax ? d : dmax;
            }
            if (dmax - dmin > 30000) return -1;
            a.s2 = 0, a.nplanes = 1, a.dmin = dmin;
            a.plane_np[0] = (bm + dmax - dmin + 15) / 16;
            for (int t = 0; t < a0.ntaps; ++t) {
                const int d = (a0.h0 + a0.dh[t]) * a0.Wi + (a0.w0 + a0.dw[t]);
                a.t_q[t] = 0, a.t_rs[t] = (int8_t)t, a.t_drow[t] = (short)(d - dmin);
            }
        }
        const int rc = bm == 256 ? ps_launch_bn<256>(a, st) : ps_launch_bn<128>(a, st);
        if (rc != -1) return rc;
    }
    return -1;
}
