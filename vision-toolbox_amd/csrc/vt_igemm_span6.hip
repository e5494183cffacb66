// vt_igemm_span6.hip -- persistent input-span convolution (bf16), one 12-wave workgroup per CU: two compute groups
// that run half a step apart plus four LDS-DMA loader waves.  For the MFMA-bound stride-1 3x3 layers: every
// ConvNormAct 3x3 stride-1 forward conv of the Darknet / CSPDarknet / VoVNet stages (reference
// components.py:26-35, darknet.py:23-24, vovnet.py:41-44) and their stride-1 data gradients.
//
// Why this shape (measured on 128->128 3x3 @28x28, tools/exp_span.sh and the VT_SPAN5_ABL ablations):
//   * in vt_igemm_span.hip a step costs the SUM of scalar bookkeeping + barrier (0.23 us), LDS-DMA issue (0.31 us:
//     the per-CU global->LDS path at its limit, every wave blocked on it) and the MFMAs (0.40 us): the waves of the
//     two workgroups on a CU fall into phase and nothing overlaps.
//   * waves that compute must not issue vector memory at all (vt_igemm_span5.hip: 0.51 us per step with ONE
//     compute wave per SIMD), but a 5-wave workgroup at 168 registers is admitted only once per CU.
// So: ONE workgroup per CU, three waves per SIMD --
//   * compute group 0 (waves 0-3) and group 1 (waves 4-7), each 2 x 2 waves over its own (32*FM) x 128 tile stream.
//     Time is cut into TICKS, one workgroup barrier each.  Group 0 reads the fragments of step s in tick 2s and
//     issues its MFMAs in tick 2s+1; group 1 does the same one tick later.  On every SIMD one wave feeds the
//     matrix pipe while the other reads LDS, by construction (cf. the guide's 8-phase GEMM template).
//   * both groups walk the same (chunk, tap) sequence, so they share ONE filter-slice ring: a slice is DMA'd once
//     per 2 x (32*FM) rows -- half the global->LDS bytes per FLOP of the two-workgroup kernel.
//   * loader waves 8-11 issue every LDS-DMA (filter slices, both groups' input spans), three slices and the next
//     chunks' spans in flight, retired by exact counted vmcnt waits before the even ticks; they also build the
//     next tiles' row tables.  A loader wave moves ~25 GB/s, hence four.
//   * PERSISTENT: a workgroup owns a contiguous range of 32-row units of its XCD's share of the flat pixel index,
//     half for each group, cut into tiles of 4..kFMX units (no partial last round); the loaders run ahead across
//     tile boundaries, so the next tile's first span and slices land while the accumulators are stored.
//   * swapped MFMA operands (filter rows = MFMA rows, pixels = MFMA columns): a lane holds 2 x 8 CONSECUTIVE
//     output channels of one pixel and stores them straight from the accumulators (no LDS staging).
// LDS images, swizzles and the summation order are those of vt_igemm_span.hip: outputs are bit-identical to it.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "vt_common.h"
#include "vt_igemm_args.h"

#ifndef VT_SPAN6_FMX
#define VT_SPAN6_FMX 7
#endif
#ifndef VT_SPAN6_SETPRIO
#define VT_SPAN6_SETPRIO 1  // bit 0: s_setprio(1) around the MFMA tick; bit 1: the loader waves at priority 2 (measured: nothing)
#endif

namespace {

// diagnostics (ablation bits, stamps, timed barriers) exist only in -DVT_SPAN6_DIAG builds (tools/build_diag.sh);
// the shipped kernel has none of these branches
#ifdef VT_SPAN6_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif
#define VT_DBG(bit) (kDiag && (a.debug & (bit)))

constexpr int kFMX = VT_SPAN6_FMX;  // row fragments (16 rows) per compute wave, at most
constexpr int kNSB = 4;             // filter-slice ring slots (3 slices in flight)
constexpr int kBSlot = 128 * 64;    // bytes per filter slice: 128 filter rows x 32 channels
constexpr int kBMX = 32 * kFMX;     // rows of the tallest tile of one group


// dev diagnostics (VT_SPAN6_ABL bit 16): wall-clock stamps (100 MHz) per workgroup; never read by the kernel itself
__device__ unsigned long long vt_span6_stamps[512 * 16];
__device__ unsigned long long vt_span6_pstamps[512 * 16];  // prologue detail (loader 0 / compute wave 0), see VT_S6_PSTAMP
#define VT_S6_STAMP(k)                                                                                        \
    do {                                                                                                      \
        if (VT_DBG(16) && lane == 0 && blockIdx.x < 512) vt_span6_stamps[blockIdx.x * 16 + (k)] = wall_clock64(); \
    } while (0)

#define VT_S6_PSTAMP(k)                                                                                        \
    do {                                                                                                      \
        if (VT_DBG(16) && lane == 0 && blockIdx.x < 512) vt_span6_pstamps[blockIdx.x * 16 + (k)] = wall_clock64(); \
    } while (0)

__device__ __attribute__((aligned(16))) unsigned int vt_span6_zero16[4];  // source of every padding row

struct S6Args {
    IgemmArgs p;
    int dmin, halo;  // span row of tap t = (eh*W + ew) - dmin, in [0, halo]
    int units;       // ceil(Mp / 32)
    int upx;         // units per XCD
    int rslots;      // row slots per XCD (workgroups per XCD / tiles_n)
    int npc;         // span pieces (16 rows x 64 B) per chunk: ceil((32*fmx + halo) / 16)
    int fmx;         // tallest tile of this launch in 32-row units (<= kFMX): lowered on wide maps, whose halo would
                     // otherwise push the span past the 28 pieces a chunk slot holds
    int ppt;         // pieces issued per tap at taps 0..5: ceil(npc / 6)
    int Hp, Wp, Mp;  // the image H x W and the number of positions B*H*W (round 5: rows are the pixels themselves; the
                     // names date from the padded (H+1) x (W+1) enumeration, see the kernel's compute-wave comment)
    unsigned hp_magic, wp_magic;  // ceil(2^32 / Hp), ceil(2^32 / Wp): quotients by multiply-high (+ one correction)
    int s64b, s64i, s64j;         // 64 positions = s64b images + s64i rows + s64j columns of the padded image (tile_bases)
    int dtap[9];     // span row of every tap
    int tsel[9];     // tap t reads row i + er - 1, column j + ec - 1: er | ec << 2  (er, ec in 0..2)
    int debug;       // dev ablations (VT_SPAN6_ABL): 1 no DMA in the loop, 2 no MFMA / reads, 4 no vmcnt wait, 16 stamps
};

__device__ __forceinline__ int swz4(int g) { return (0x1320 >> ((g & 3) * 4)) & 3; }  // filter-slice image
// Span image: chunk ^= 2 * ((row >> 2) & 1).  A tap shifts the fragment's 16 rows by an arbitrary offset; this is the
// swizzle (found by enumeration over all 4-entry tables) under which the four 16-lane groups of a ds_read_b128 hit 16
// distinct 16-byte slots for EVERY row offset -- the 0x1320 table is conflict free only for offsets that are
// multiples of 4 (2-way on half of the taps).
__device__ __forceinline__ int swzA(int g) { return (g & 1) << 1; }

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long)hi << 32) | lo);
}
// LDS-DMA, 16 B per lane: LDS address = M0 + lane*16, global address = sbase + voff (or the per-lane address)
// (s_nop 4: the scalar base may have just been written by a VALU instruction -- v_readfirstlane here, or a
//  v_readlane reloading a spilled SGPR -- and a VMEM instruction reading such an SGPR needs 5 wait states, which
//  hipcc does not insert in front of an asm statement.  Without it the load can use a stale base: a memory fault.)
__device__ __forceinline__ void glds_s(unsigned voff, const void* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(uniform_ptr(sbase)) : "memory");
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
// counted wait with a run-time, wave-uniform count (vmcnt takes an immediate)
__device__ __forceinline__ void vm_wait_dyn(int n) {
#define VT_W4(b)                                  \
    switch (n - (b)) {                            \
        case 0: vmw<(b) + 0>(); break;            \
        case 1: vmw<(b) + 1>(); break;            \
        case 2: vmw<(b) + 2>(); break;            \
        default: vmw<(b) + 3>(); break;           \
    }
    if (n <= 0) { vmw<0>(); return; }
    if (n >= 28) { vmw<28>(); return; }  // (a loader never has more than ~12 in flight: larger counts wait a little early)
    if (n < 16) {
        if (n < 8) { if (n < 4) { VT_W4(0) } else { VT_W4(4) } }
        else { if (n < 12) { VT_W4(8) } else { VT_W4(12) } }
    } else {
        if (n < 24) { if (n < 20) { VT_W4(16) } else { VT_W4(20) } }
        else { VT_W4(24) }
    }
#undef VT_W4
}

// LDS map (bytes): [row output pixel 2 groups x 2 x kBMX x 4][16 zero bytes: the fragment of a tap outside the image]
//                  [filter ring kNSB x 8 KiB][group 0: span slot 0, slot 1][group 1: span slot 0, slot 1]
struct L6 {
    static constexpr int kPo = 0;
    static constexpr int kZ = kPo + 4 * kBMX * 4;
    static constexpr int kB = (kZ + 16 + 1023) / 1024 * 1024;
    // (KSPLIT: a ring slot holds two slices, one per compute group)
    __host__ __device__ static constexpr int a_off(bool ksplit) { return kB + kNSB * kBSlot * (ksplit ? 2 : 1); }
    __host__ __device__ static constexpr int bytes(int npc, bool ksplit) { return a_off(ksplit) + 4 * npc * 1024; }
};

// sum of a value over the 16 lanes of its DPP row (lanes 16k .. 16k+15), returned in every lane of the row
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float x) {
    x = dpp_add<0x128>(x);  // row_ror:8
    x = dpp_add<0x124>(x);  // row_ror:4
    x = dpp_add<0x122>(x);  // row_ror:2
    return dpp_add<0x121>(x);  // row_ror:1
}

template <int T>
using I_ = std::integral_constant<int, T>;

typedef const __attribute__((address_space(4))) S6Args* ArgsPtr;
__device__ __forceinline__ ArgsPtr fresh_args() {
    ArgsPtr q = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);  // nothing migrates across a tick boundary
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// diagnostics: a barrier that adds the cycles this wave spent waiting in it to `w` (debug bit 16 only)
#define VT_TBAR(w)                                         \
    do {                                                   \
        if (VT_DBG(16)) {                                  \
            const unsigned long long c0_ = clock64();      \
            wg_barrier();                                  \
            (w) += clock64() - c0_;                        \
        } else {                                           \
            wg_barrier();                                  \
        }                                                  \
    } while (0)

// 12 waves: 0-3 compute group 0, 4-7 compute group 1, 8-11 loaders; three per SIMD = at most 168 registers
// MASKED: rows are the pixels themselves and a tap that leaves the image reads a zero block (else: padded positions)
// KSPLIT (layers whose whole row share fits ONE group's tile -- 512 -> 512 @7x7 at batch 256, the 14 x 14 layers at batch
//   128): both groups work on the SAME rows, group 0 on the first half of the input channels and group 1 on the second;
//   a ring slot carries both groups' filter slices; at the end group 1 hands its accumulators over through LDS and group 0
//   adds them and runs the epilogue.  Half as many K-steps per launch at twice the MFMAs per tick, and every filter
//   slice staged once per 224 rows instead of once per 112.  (One tile per workgroup by construction: the hand-over re-uses
//   the ring and the span slots, which a loader running ahead into a next tile would still be filling.)
template <int MODE, bool MASKED, bool KSPLIT>  // epilogue: 0 plain (+ residual), 1 BatchNorm statistics, 2 affine (+ ReLU, + residual), 3 plain + BatchNorm-backward sums
__global__ void __launch_bounds__(768, 3) span6_kernel(const S6Args a) {
    const IgemmArgs& p = a.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sPo = (int*)(smem + L6::kPo);
    const char* sBb = smem + L6::kB;
    const char* sAb = smem + L6::a_off(KSPLIT);
    constexpr int kBS = KSPLIT ? 2 * kBSlot : kBSlot;  // bytes per ring slot
    const int aslot_bytes = a.npc * 1024;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's share: a contiguous range of 32-row units of one XCD, one filter column tile ----
    const int bid = blockIdx.x, xcd = bid & 7, l = bid >> 3;
    if (l >= a.rslots * p.tiles_n) return;  // (tiles_n does not divide 32)
    const int tn = l % p.tiles_n, rs = l / p.tiles_n;
    const int ux0 = xcd * a.upx, ux1 = min(a.units, ux0 + a.upx);
    const int nx = max(0, ux1 - ux0);
    const int ua = __builtin_amdgcn_readfirstlane(ux0 + (int)((unsigned)rs * (unsigned)nx / (unsigned)a.rslots));
    const int ub = __builtin_amdgcn_readfirstlane(ux0 + (int)((unsigned)(rs + 1) * (unsigned)nx / (unsigned)a.rslots));
    const int nun = ub - ua;
    if (nun <= 0) return;
    // group 0 takes the first ceil(nun/2) units, group 1 the rest; both cut their range into the SAME number of
    // tiles (the two groups run one schedule), heights within a group differing by at most one unit
    const int nun0 = KSPLIT ? nun : (nun + 1) >> 1;
    const int ntile = __builtin_amdgcn_readfirstlane((nun0 + a.fmx - 1) / a.fmx);
    const int nchunks = __builtin_amdgcn_readfirstlane(KSPLIT ? p.Cin / 64 : p.Cin / 32);  // per compute group
    const int koff = KSPLIT ? nchunks * 64 : 0;  // KSPLIT: byte offset of group 1's first channel chunk
    const int nsteps = nchunks * 9;
    // tile k of group g: units [gu0 + k*tb + min(k, te), + tb + (k < te)), gu0 = ua (g = 0) / ua + nun0 (g = 1)
#define VT_G_NUN(g) ((g) && !KSPLIT ? nun - nun0 : nun0)
#define VT_G_U0(g) ((g) && !KSPLIT ? ua + nun0 : ua)
#define VT_TILE_U0(g, k) (VT_G_U0(g) + (k) * (VT_G_NUN(g) / ntile) + min((k), VT_G_NUN(g) % ntile))
#define VT_TILE_F(g, k) (VT_G_NUN(g) / ntile + ((k) < VT_G_NUN(g) % ntile ? 1 : 0))

    // Who issues the FIRST chunk's span pieces: the compute waves (padded coordinates) or the loaders (pixel rows: their
    // piece sources cost no division, and a compute wave still has its tap masks to build before the first tick -- the
    // compute-wave form measured 1.3 us (256 -> 256 @14x14) to 6 us (K-split 512 -> 512 @7x7) SLOWER there).
    constexpr bool kCwPro = !MASKED;
    // ---- address arithmetic shared by the loader waves and, in the prologue only, the compute waves ----------------
    const int lj = wave & 3;  // loader index (waves 8..11) / the quarter of the prologue's work a compute wave takes
    const char* xg = (const char*)p.x;
    const unsigned a_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + L6::a_off(KSPLIT));
    const unsigned m0_keep = get_m0();
    const long ldx2 = (long)p.ldx * 2;
    const int Mp = a.Mp, Wp = a.Wp, Hp = a.Hp, W_ = p.Wi, H_ = p.Hi;
    const unsigned wp_magic = a.wp_magic, hp_magic = a.hp_magic;
    const unsigned long zero_src = (unsigned long)(const void*)vt_span6_zero16;

    // Pixels are enumerated in PADDED coordinates: every image is (H+1) x (W+1) positions whose extra row and
    // column are zero pixels, so a tap is a constant offset in the flat padded index (the zero column right of
    // row i is the one left of row i+1, the zero row below an image the one above the next) and the compute
    // waves need no masks at all.  The span is built from the unpadded tensor: LDS row r of a piece is
    // padded position mp; its source is pixel (b, i, j) or the zero page.
    // span piece = 16 rows x 64 B: lane owns row (lane>>2), source chunk (lane&3)^swzA(lane>>4)
    const int cjA = (lane & 3) ^ swzA(lane >> 4);
    // (b, i, j) of padded position mp (0 <= mp < Mp); false for a padding position
    auto unpad = [&](int mp, int& b_, int& i_, int& j_) -> bool {
        int q = (int)__umulhi((unsigned)mp, wp_magic);
        int j = mp - q * Wp;
        if (j < 0) j += Wp, --q;
        int b = (int)__umulhi((unsigned)q, hp_magic);
        int i = q - b * Hp;
        if (i < 0) i += Hp, --b;
        b_ = b, i_ = i, j_ = j;
        return j < W_ && i < H_;
    };
    // The padded -> pixel mapping of a wave's pieces (piece T of group g = span rows 16*(lj + 4T) ..) depends on
    // the tile only, not on the channel chunk: it is computed once per tile (behind taps 6 / 7 of the previous
    // tile's first chunk) and a piece in the step loop costs one add.  ab[T]: global source (channel chunk 0) of this
    // lane's 16 bytes of piece T's row; vm bit T: a real pixel (the chunk's byte offset applies), else the zero page.
    // Round 6: ONE division per tile -- piece T's row is position mp0 + 64 T, so (b, i, j) advance by the host-computed
    // digits of 64 in the (Hp, Wp) number system with at most one carry each (was: seven divisions, ~45 vector
    // instructions per piece of a prologue that the first tick waits for).
    auto tile_bases = [&](int m0t, unsigned long (&ab)[7], unsigned& vm) {
        vm = 0;
        const int mp0 = m0t + a.dmin + lj * 16 + (lane >> 2);
        if constexpr (MASKED) {  // rows are the pixels themselves: no division
#pragma unroll
            for (int T = 0; T < 7; ++T) {
                const int mp = mp0 + 64 * T;
                const bool ok = (unsigned)mp < (unsigned)Mp;
                ab[T] = ok ? (unsigned long)xg + (unsigned long)((long)mp * ldx2 + cjA * 16) : zero_src;
                vm |= (ok ? 1u : 0u) << T;
            }
        } else {
            ArgsPtr Q = fresh_args();
            const int sb = Q->s64b, si = Q->s64i, sj = Q->s64j, B_ = Q->p.B;
            int b, i, j;
            // (mp0 >= -Wp - 1: one image further on the division sees a non-negative position)
            unpad(mp0 + Hp * Wp, b, i, j);
            --b;
#pragma unroll
            for (int T = 0; T < 7; ++T) {
                const bool ok = b >= 0 && b < B_ && i < H_ && j < W_;  // (b < B: mp < Mp)
                const long pix = ((long)b * H_ + i) * W_ + j;
                ab[T] = ok ? (unsigned long)xg + (unsigned long)(pix * ldx2 + cjA * 16) : zero_src;
                vm |= (ok ? 1u : 0u) << T;
                j += sj;
                if (j >= Wp) j -= Wp, ++i;
                i += si;
                if (i >= Hp) i -= Hp, ++b;
                b += sb;
            }
        }
    };
    // this wave's quarter of the table of group g's tile that starts at padded position m0t (table half par):
    // where each row is stored, -1 for a padding position
    auto row_tables = [&](int g, int par, int m0t) {
        ArgsPtr Q = fresh_args();
        constexpr int QR = (kBMX + 3) / 4;
        const int r = lj * QR + lane;
        if (lane < QR && r < kBMX) {
            const int mp = m0t + r;
            int po = -1;
            if (mp < Mp) {
                int b, i, j;
                if (unpad(mp, b, i, j))
                    po = (b * Q->p.oH + (i * Q->p.oHs + Q->p.oh0)) * Q->p.oW + (j * Q->p.oWs + Q->p.ow0);
            }
            sPo[(g * 2 + par) * kBMX + r] = po;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // written before this wave's next barrier
    };

    if (wave >= 8) {
        // =========================== loader waves ==================================================
        VT_S6_STAMP(0);
#if VT_SPAN6_SETPRIO & 2
        __builtin_amdgcn_s_setprio(2);  // (experiment: the loader wave above both compute waves of its SIMD)
#endif
        const char* wg = (const char*)p.w;
        const unsigned b_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + L6::kB);
        const int cin2 = p.Cin * 2;
        unsigned long ab_cur[2][7], ab_nxt[2][7];
        unsigned vm_cur[2] = {0, 0}, vm_nxt[2] = {0, 0};  // bit T: piece T's row of this lane is a real pixel
        // filter slice = 8 pieces of 16 rows, this loader's are q = 2*lj, 2*lj+1; row n = 16q + (lane>>2); the
        // fragment reads address row n with chunk position kq ^ swz4(n>>3), so the source chunk is
        // (lane&3) ^ swz4(2q + (lane>>5)).  Rows past Cout (N tail) are clamped: their outputs are never stored.
        unsigned b_voff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 2 * lj + i;
            const int n = min(tn * 128 + 16 * q + (lane >> 2), p.Cout - 1);
            const int cj = (lane & 3) ^ swz4(2 * q + (lane >> 5));
            b_voff[i] = (unsigned)(((long)n * p.ldw + cj * 8) * 2);
        }

        auto issue_slice = [&](int slot, int ic, int T) {
            const char* sb = wg + (long)ic * 64 + (long)T * cin2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                set_m0(b_base + (unsigned)(slot * kBS + (2 * lj + i) * 1024));
                glds_s(b_voff[i], sb);
                if constexpr (KSPLIT) {
                    set_m0(b_base + (unsigned)(slot * kBS + kBSlot + (2 * lj + i) * 1024));
                    glds_s(b_voff[i], sb + koff);
                }
            }
        };

        // ---- prologue: slices 0..2 and the piece sources of both groups' first tiles -----------------------------
        // Round 6: the FIRST chunk's span pieces and the first tiles' row tables come from the COMPUTE waves, which
        // have nothing to do before the first tick (compute wave w of group g takes the pieces this loader takes for
        // group g in the loop, below): the first tick waited for ~1250 loader instructions in a row -- 6.0 us before
        // the last prologue LDS-DMA was even issued (profiles/r04_span6_phases.json) -- and now for the longer of a
        // loader's share (the slices) and a compute wave's (one group's piece sources, seven LDS-DMAs, a quarter of a row
        // table).  The piece sources, which this loader needs for the chunks that follow, come from the compute waves too:
        // compute wave (g, lj) parks them in the four pieces this loader owns of group g's SECOND span slot (64 lanes x
        // 64 B; the slot is first written by this loader's own LDS-DMAs of step 0, behind the reads below), because the
        // same ~400 instructions took a loader wave 4 us and a compute wave 1 (stamps, NOTEBOOK R6).  The first wait of
        // the loop is a full one.
        const long S = (long)ntile * nsteps;  // steps of each group
        int m0c[2] = {VT_TILE_U0(0, 0) * 32, VT_TILE_U0(1, 0) * 32};
        int sic = 0, sT = 0;         // (chunk, tap) of the next slice to issue; slices repeat per tile
        long sg = 0;                 // its step
        auto next_slice = [&]() {
            if (sg < S) {
                issue_slice((int)(sg & 3), sic, sT);
                ++sg;
                if (++sT == 9) {
                    sT = 0;
                    if (++sic == nchunks) sic = 0;
                }
            }
        };
        next_slice();
        next_slice();
        next_slice();
        VT_S6_PSTAMP(0);
        if constexpr (!kCwPro) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                tile_bases(m0c[g], ab_cur[g], vm_cur[g]);
#pragma unroll
                for (int T = 0; T < 7; ++T)
                    if (lj + 4 * T < a.npc) {
                        set_m0(a_base + (unsigned)((g * 2) * aslot_bytes + (lj + 4 * T) * 1024));
                        glds_v(ab_cur[g][T] + ((g == 1 && ((vm_cur[g] >> T) & 1u)) ? (unsigned long)koff : 0ul));
                    }
            }
            row_tables(0, 0, m0c[0]);
            row_tables(1, 0, m0c[1]);
        }
        if (lj == 0 && lane < 4) ((unsigned*)(smem + L6::kZ))[lane] = 0u;  // the fragment of a tap outside the image
        VT_S6_STAMP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // zero block written before barrier 0

        // Tick t starts with barrier t.  Group 0 reads step s in tick 2s, group 1 in tick 2s+1 (and both re-read the
        // slice during their MFMA ticks 2s+1 / 2s+2), so a slice's ring slot is free from barrier 2s+3 on; group g's
        // span slot of chunk c-1 is free from its first read tick of chunk c.  Per step this wave issues, in order:
        //   even tick: [its piece of group 0's next span]      (taps 0..NTP-1: piece 4T + lj)
        //   odd tick:  [its piece of group 1's next span] [its two pieces of slice s+3]
        // so before the even tick of tap T exactly 4 + 2 * (P(T-1) + P(T-2)) of its instructions are younger than
        // slice s (P(t) = 1 for 0 <= t < NTP): compile-time counts.
        unsigned long long lwait = 0;
        const unsigned long long lc0 = clock64();
        int acur = 0;                    // span slot (both groups) of the chunk being read
        int bnext = 3;                   // ring slot of the next slice to issue
        long sleft = S;                  // steps left, the current one included
        const bool dma = !VT_DBG(1);
        auto chunks = [&](auto NTPc) {
            constexpr int NTP = decltype(NTPc)::value;
            for (int k = 0; k < ntile; ++k) {
                const bool has_next = k + 1 < ntile;
                int m0n[2] = {0, 0};
                if (has_next) m0n[0] = VT_TILE_U0(0, k + 1) * 32, m0n[1] = VT_TILE_U0(1, k + 1) * 32;
                for (int ic = 0; ic < nchunks; ++ic) {
                    const bool lastc = ic + 1 == nchunks;
                    const bool nextc = !lastc || has_next;  // a chunk follows this one (else: this one is re-loaded, unused)
                    const int cb_t = !nextc ? ic * 64 : (lastc ? 0 : (ic + 1) * 64);
                    const bool nx = lastc && nextc;  // the chunk being loaded belongs to the next tile
                    unsigned m0g[2];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        m0g[g] = a_base + (unsigned)((g * 2 + (acur ^ 1)) * aslot_bytes + lj * 1024);
                    const char* wb_cur = wg + (long)ic * 64;
                    const char* wb_nxt = wg + (long)(lastc ? 0 : ic + 1) * 64;
                    const unsigned long cb64 = (unsigned long)cb_t;
                    auto step = [&](auto Tc) {
                        constexpr int T = decltype(Tc)::value;
                        constexpr auto P = [](int t) { return (t >= 0 && t < NTP) ? 1 : 0; };
                        constexpr int kYounger = (KSPLIT ? 8 : 4) + 2 * (P(T - 1) + P(T - 2));
                        // ---- even tick 2s: slice s (and everything older: both groups' spans of its chunk) has landed
                        if (!VT_DBG(4)) {
                            if (sleft > 2 && sleft != S) vmw<kYounger>();
                            else vmw<0>();  // (the first step: the prologue issued the slices before the spans)
                        }
                        VT_TBAR(lwait);
                        if constexpr (T == 0 && kCwPro) {
                            if (sleft == S) {  // the first tick: the first tiles' piece sources, parked by the compute waves
#pragma unroll
                                for (int g = 0; g < 2; ++g) {
                                    const char* st = sAb + (g * 2 + 1) * aslot_bytes + lj * 1024 + lane * 16;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const uint4 v = *(const uint4*)(st + q * 4096);
                                        ab_cur[g][2 * q] = ((unsigned long)v.y << 32) | v.x;
                                        if (q < 3) ab_cur[g][2 * q + 1] = ((unsigned long)v.w << 32) | v.z;
                                        else vm_cur[g] = v.z;
                                    }
                                }
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before this wave's LDS-DMAs land there
                                VT_S6_PSTAMP(1);
                            }
                        }
                        if constexpr (T < NTP) if (dma) {
                            const unsigned long base = nx ? ab_nxt[0][T] : ab_cur[0][T];
                            const unsigned v = ((nx ? vm_nxt[0] : vm_cur[0]) >> T) & 1u;
                            set_m0(m0g[0] + T * 4096);
                            glds_v(base + (v ? cb64 : 0ul));
                        }
                        // the next tiles' row tables, behind taps 6 (group 0) and 7 (group 1) of this tile's first chunk:
                        // the halves they go to were last read by the previous tiles' epilogues, which every compute
                        // wave left before this tile's first ticks
                        if (T == 6 && ic == 0 && has_next) {
                            row_tables(0, (k + 1) & 1, m0n[0]);
                            tile_bases(m0n[0], ab_nxt[0], vm_nxt[0]);
                        }
                        // ---- odd tick 2s+1: group 1 reads step s
                        VT_TBAR(lwait);
                        // (the piece goes first: at tap NTP-1 = 6 it must be older than the slice whose wait retires it)
                        if constexpr (T < NTP) if (dma) {
                            const unsigned long base = nx ? ab_nxt[1][T] : ab_cur[1][T];
                            const unsigned v = ((nx ? vm_nxt[1] : vm_cur[1]) >> T) & 1u;
                            set_m0(m0g[1] + T * 4096);
                            glds_v(base + (v ? cb64 + (unsigned long)koff : 0ul));
                        }
                        // group 1 has left its MFMA tick of step s-1 (whose second pair of filter fragments it read
                        // during that tick): the ring slot of slice s-1 takes slice s+3
                        if (dma && sleft > 3) {
                            constexpr int T3 = (T + 3) % 9;
                            const char* sb = (T < 6 ? wb_cur : wb_nxt) + (long)(T3 * cin2);
                            const unsigned m0b = b_base + (unsigned)(bnext * kBS + 2 * lj * 1024);
                            set_m0(m0b);
                            glds_s(b_voff[0], sb);
                            set_m0(m0b + 1024);
                            glds_s(b_voff[1], sb);
                            if constexpr (KSPLIT) {
                                set_m0(m0b + kBSlot);
                                glds_s(b_voff[0], sb + koff);
                                set_m0(m0b + kBSlot + 1024);
                                glds_s(b_voff[1], sb + koff);
                            }
                        }
                        if (T == 7 && ic == 0 && has_next) {
                            row_tables(1, (k + 1) & 1, m0n[1]);
                            tile_bases(m0n[1], ab_nxt[1], vm_nxt[1]);
                        }
                        bnext = (bnext + 1) & 3;
                        --sleft;
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
                m0c[0] = m0n[0], m0c[1] = m0n[1];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    vm_cur[g] = vm_nxt[g];
#pragma unroll
                    for (int T = 0; T < 7; ++T) ab_cur[g][T] = ab_nxt[g][T];
                }
            }
        };
        switch (a.npc >> 2) {
            case 4: chunks(I_<4>{}); break;
            case 5: chunks(I_<5>{}); break;
            case 6: chunks(I_<6>{}); break;
            default: chunks(I_<7>{}); break;
        }
        wg_barrier();  // tick 2S: group 1's last MFMA tick
        VT_S6_STAMP(2);
        if (VT_DBG(16) && lj == 0 && lane == 0 && blockIdx.x < 512) {
            vt_span6_stamps[blockIdx.x * 16 + 13] = lwait;
            vt_span6_stamps[blockIdx.x * 16 + 14] = clock64() - lc0;
        }
        vmw<0>();
        if constexpr (KSPLIT) {
            wg_barrier();  // every LDS-DMA of this workgroup has landed: group 1 may overwrite the ring and the span slots
            wg_barrier();  // group 1's accumulators are in LDS
        }
        set_m0(m0_keep);
        return;
    }

    // =============================== compute waves ==================================================
    const int grp = wave >> 2;  // 0: reads in even ticks, MFMAs in odd ticks; 1: one tick later
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int q4 = lane >> 4, c16 = lane & 15;
    // this lane's output channels: ch(h, e8) = tn*128 + wn*64 + h*32 + q4*8 + e8, h = 0,1, e8 = 0..7
    const int ch0 = tn * 128 + wn * 64 + q4 * 8;
    // filter fragment j of this lane: MFMA row r = c16 -> slice row n_j = wn*64 + (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3);
    // (n_j >> 3) & 3 = r >> 2 for every j, so the four fragments share one swizzle term and differ by constants
    const int nb0 = wn * 64 + (c16 >> 2) * 8 + (c16 & 3);
    const int b_lane = (nb0 * 4 + (q4 ^ swz4(c16 >> 2))) * 16;  // byte offset inside a slice; j adds {0,256,2048,2304}
    const int g_nun = VT_G_NUN(grp), g_u0 = VT_G_U0(grp);
    const int g_tb = g_nun / ntile, g_te = g_nun % ntile;
    const char* sAg = sAb + grp * 2 * aslot_bytes;  // this group's two span slots
    int bcur = 0, acur = 0;
    unsigned long long cwait = 0, cR = 0, cM = 0;

    if constexpr (kCwPro) {
        // ---- prologue (round 6): this wave's share of its group's first span chunk and first row table -------------
        const int m0g = VT_TILE_U0(grp, 0) * 32;
        unsigned long ab0[7];
        unsigned vm0;
        tile_bases(m0g, ab0, vm0);
        if (wave == 0) VT_S6_PSTAMP(2);
#pragma unroll
        for (int T = 0; T < 7; ++T)
            if (lj + 4 * T < a.npc) {
                set_m0(a_base + (unsigned)((grp * 2) * aslot_bytes + (lj + 4 * T) * 1024));
                glds_v(ab0[T] + ((grp == 1 && ((vm0 >> T) & 1u)) ? (unsigned long)koff : 0ul));
            }
        if (wave == 0) VT_S6_PSTAMP(3);
        {  // the piece sources for loader lj (see the loader's prologue): its four first pieces of this group's second span slot
            char* st = smem + L6::a_off(KSPLIT) + (grp * 2 + 1) * aslot_bytes + lj * 1024 + lane * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long lo = ab0[2 * q], hi = q < 3 ? ab0[2 * q + 1] : (unsigned long)vm0;
                *(uint4*)(st + q * 4096) = make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
            }
        }
        row_tables(grp, 0, m0g);
        vmw<0>();  // landed before this wave's next barrier: group 0 reads the chunk in tick 0, group 1 in tick 1
        set_m0(m0_keep);
        if (wave == 0) VT_S6_PSTAMP(4);
    }
    if (grp == 1) wg_barrier();  // tick 0: group 0 reads its first step

    for (int k = 0; k < ntile; ++k) {
        const int par = k & 1;
        const int f_cur = g_tb + (k < g_te ? 1 : 0);
        const long m0_cur = (long)(g_u0 + k * g_tb + min(k, g_te)) * 32;
        const int fm = max(f_cur, 4);           // row fragments per wave in this tile (4..kFMX)
        const int rows_tile = 32 * f_cur;       // rows this tile owns (stores / statistics)
        const int tbl = (grp * 2 + par) * kBMX;

        auto run = [&](auto FMc) {
            constexpr int FM = decltype(FMc)::value;
            int wrow = wm * 16 * FM + c16;  // this lane's row inside the tile, fragment 0
            asm volatile("" : "+v"(wrow));    // (opaque: nothing derived from it is hoisted out of the tile loop)
            unsigned a_off = 0;               // byte offset (inside this group's span slots) of this lane's fragment-0 row of the coming step
            // MASKED (round 5): the rows of a tile are the PIXELS (flat index (b*H + i)*W + j), not the positions of a padded
            // (H+1) x (W+1) image.  A tap is still a constant row offset inside the span; where it leaves the image (or wraps
            // into the next row) the lane reads a zero block instead: one validity bit per (row fragment, tap), built once
            // per tile -- tmask[i / 3] bit 9 * (i % 3) + t -- and three vector instructions per fragment read.  Those cost
            // more than the padding positions do (+13 % per step at 28 x 28 against 7 % of padding rows), so the dispatcher
            // takes this form only where the shorter row count saves a whole tile round (14 x 14 at batch 256: 15 units of
            // padded positions for one of every 16 workgroups, two tiles, against 12-13 units of pixels, one tile).
            unsigned tmask[(FM + 2) / 3];
            if constexpr (MASKED) {
                ArgsPtr Q = fresh_args();
                const int W_ = Q->Wp, H_ = Q->Hp;
                const unsigned wmag = Q->wp_magic, hmag = Q->hp_magic;
#pragma unroll
                for (int i3 = 0; i3 < (FM + 2) / 3; ++i3) tmask[i3] = 0;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int mp = (int)m0_cur + wrow + 16 * i;
                    int q = (int)__umulhi((unsigned)mp, wmag);
                    int j = mp - q * W_;
                    if (j < 0) j += W_, --q;
                    int b = (int)__umulhi((unsigned)q, hmag);
                    int ii = q - b * H_;
                    if (ii < 0) ii += H_;
                    // bits 0..2: row i-1 / i / i+1 inside the image; bits 3..5: column j-1 / j / j+1
                    const unsigned rc = (ii > 0 ? 1u : 0u) | 2u | (ii < H_ - 1 ? 4u : 0u) | (j > 0 ? 8u : 0u) | 16u | (j < W_ - 1 ? 32u : 0u);
                    unsigned m9 = 0;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int sel = Q->tsel[t];
                        m9 |= ((rc >> (sel & 3)) & (rc >> (3 + (sel >> 2))) & 1u) << t;
                    }
                    tmask[i / 3] |= m9 << (9 * (i % 3));
                }
            }
            const char* const zfrag = smem + L6::kZ;
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
                    // ---- read tick: this step's slice and span are in LDS
                    VT_TBAR(cwait);
                    if (T == 0 && ic == 0) {
                        if (wave == 0 && k < 4) VT_S6_STAMP(4 + 3 * k);
                        const int srow = wrow + fresh_args()->dtap[0];
                        a_off = a_rd + (unsigned)((srow * 4 + (q4 ^ swzA(srow >> 2))) * 16);
                    }
                    if (VT_DBG(2)) {
                        VT_TBAR(cwait);
                        bcur = (bcur + 1) & 3;
                        return;
                    }
                    const unsigned long long tR0 = VT_DBG(16) ? clock64() : 0;
                    const char* A = sAg + a_off;  // computed during the previous MFMA tick (or at the tile's start)
                    const char* Bt = sBb + (bcur * kBS + (KSPLIT ? grp * kBSlot : 0) + b_lane);
                    // three filter fragments are read in the read tick, the fourth during the MFMA tick into the
                    // first one's registers (168 registers per lane): its latency hides behind 2*FM MFMAs
                    uint4 af[FM], bf0, bf1, bf2;
                    bf0 = *(const uint4*)(Bt);
                    bf1 = *(const uint4*)(Bt + 256);
                    bf2 = *(const uint4*)(Bt + 2048);
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        if constexpr (MASKED) {
                            // (opaque: the 63 lane masks of a tile are not to be hoisted into SGPR pairs -- they do not fit
                            //  and came back as ~13 v_readlane per step -- but tested here)
                            const int mk = (int)tmask[i / 3];
                            // two vector instructions per fragment: the validity bit spread over a word (v_bfe_i32), then a
                            // bitwise select between the row's LDS address and the zero block's (v_bfi_b32)
                            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                            typedef __attribute__((address_space(3))) const u32x4_t* lds_u4;
                            unsigned sel;  // (asm: the compiler expands the builtin into two shifts, and hoists a plain compare)
                            asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(mk), "n"(9 * (i % 3) + T));
                            const unsigned ar = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)(A + i * 1024);
                            const unsigned az = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)zfrag;
                            af[i] = __builtin_bit_cast(uint4, *(lds_u4)(unsigned long)((ar & sel) | (az & ~sel)));
                        } else {
                            af[i] = *(const uint4*)(A + i * 1024);  // padded coordinates: no masks, one address register
                        }
                    }
                    // the next step's span row offset: a scalar load that returns during the MFMA tick
                    const int dnext = fresh_args()->dtap[(T + 1) % 9];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers
                    if (VT_DBG(16)) cR += clock64() - tR0;
                    // ---- MFMA tick (the other group reads meanwhile)
                    VT_TBAR(cwait);
                    const unsigned long long tM0 = VT_DBG(16) ? clock64() : 0;
// the MFMA as an asm statement whose accumulator is an in/out operand: result and addend share their registers by
// construction.  (Left to itself the compiler renames the accumulators from one unrolled step to the next -- D != C --
// and then spills them around the steps: 180-224 bytes of scratch per lane at FM = 7.)  Every use of an accumulator
// is at least a whole step (28 MFMAs and two barriers) away from the MFMA that wrote it.
#define VT_MMA_COL(bfrag, j)                                                                             \
    _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                       \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                          \
                     : "+v"(acc[i][j])                                                                  \
                     : "v"(__builtin_bit_cast(bf16x8, bfrag)), "v"(__builtin_bit_cast(bf16x8, af[i])))
#if VT_SPAN6_SETPRIO & 1
                    // the MFMA-issuing wave wins its SIMD's issue arbitration against the wave that reads fragments and the
                    // loader (the guide's T2): 128 -> 128 @28x28 65.7 -> 63.2 us alone, 128 @56x56 260 -> 257, the other shapes
                    // and the step unchanged (N R6.8); the loaders raised instead, or as well: nothing
                    __builtin_amdgcn_s_setprio(1);
#endif
                    VT_MMA_COL(bf0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    bf0 = *(const uint4*)(Bt + 2304);
                    VT_MMA_COL(bf1, 1);
                    VT_MMA_COL(bf2, 2);
                    {
                        // address of the next step's span rows (same tile), under the MFMAs
                        const int srow = wrow + dnext;
                        const unsigned rd = T == 8 ? (unsigned)((acur ^ 1) * aslot_bytes) : a_rd;
                        a_off = rd + (unsigned)((srow * 4 + (q4 ^ swzA(srow >> 2))) * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    VT_MMA_COL(bf0, 3);
#if VT_SPAN6_SETPRIO & 1
                    __builtin_amdgcn_s_setprio(0);
#endif
#undef VT_MMA_COL
                    if (VT_DBG(16)) {
                        asm volatile("s_nop 0" ::"v"(acc[FM - 1][3][0]));  // (the last MFMA's result: its issue has happened)
                        cM += clock64() - tM0;
                    }
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

            if (wave == 0 && k < 4) VT_S6_STAMP(5 + 3 * k);
            if constexpr (KSPLIT) {
                // group 1's half of the K sum goes to group 0 through LDS: wave w of either group holds the same sub-tile
                if (grp == 0) wg_barrier();  // tick 2S: group 1's last MFMA tick
                wg_barrier();                // the loaders have waited for their last LDS-DMA: ring and span slots are free
                float* red = (float*)(smem + L6::kB) + (wave & 3) * (FM * 4 * 256) + lane * 4;
                if (grp == 1) {
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) *(f32x4*)(red + (i * 4 + j) * 256) = acc[i][j];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                wg_barrier();
                if (grp == 1) return;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 o = *(const f32x4*)(red + (i * 4 + j) * 256);
                        acc[i][j] += o;
                        __builtin_amdgcn_sched_barrier(0);  // (one addend in flight at a time: the accumulators fill the file)
                    }
            }
            // ---- epilogue: two 16-byte stores per row fragment, straight from the accumulators ----------
            ArgsPtr Q = fresh_args();
            constexpr bool affine = MODE == 2, stats = MODE == 1;
            // MODE 3 (round 6): a data gradient that also REDUCES the BatchNorm backward of the unit whose output it
            // differentiates -- this launch's result is that unit's d(y); the unit's stored pre-activation z arrives like a
            // residual operand (p.res / ldr: same pixels, same channels), p.scale / p.shift are the unit's BatchNorm
            // coefficients (the ReLU mask is z * scale + shift > 0, as in bn_bwd_reduce_kernel / bn_bwd_apply_kernel),
            // p.aux0 / p.aux1 its mean / invstd, p.stats the unit's backward sums: [0][c] += sum g, [1][c] += invstd *
            // sum g * (z - mean), g = mask * d(y) AS STORED (bf16).  The separate reduction pass (d(y) and z read once more:
            // 19.5 us alone at 128 ch @28x28, ~50 us beside the filter-gradient stream) disappears.  VT_CONV_RELU: the unit
            // has a ReLU (else the mask is all ones).
            constexpr bool bnred = MODE == 3;
            const bool relu = (MODE == 2 || MODE == 3) && (Q->p.flags & VT_CONV_RELU);
            const bool has_res = MODE != 3 && MODE != 1 && (Q->p.flags & VT_CONV_RESIDUAL) != 0;
            const int Cout_ = Q->p.Cout, ldy_ = Q->p.ldy, ldr_ = Q->p.ldr;
            bf16_t* __restrict__ yg = (bf16_t*)Q->p.y;
            const bf16_t* __restrict__ rg = (const bf16_t*)Q->p.res;
            const float* scale_ = Q->p.scale;
            const float* shift_ = Q->p.shift;
            float* stats_ = Q->p.stats;
            const int rep = (int)((m0_cur / 32) % kStatReplicas);
            if constexpr (bnred) {
                // Register-lean form (the accumulators fill the file): the four channel QUADS of a lane (half h, quad q)
                // one after the other -- z as 8-byte loads, the NEXT quad's FM rows in flight while this one is consumed (a
                // quad's loads cost a round trip to HBM: with one buffer the epilogue was four of them in a row, 12 us per
                // tile); scale / shift and the two running sums for four channels; 8-byte stores (the two quads of a half
                // meet again in the L2 line).  sum g * (z - mean) is formed as sum g * z - mean * sum g on the wave's
                // partial sums (<= 16 FM rows: the cancellation costs ~|mean| / std of an f32 rounding), then scaled by
                // invstd and added in fixed point.
                const float* mean_ = Q->p.aux0;
                const float* istd_ = Q->p.aux1;
                uint2 zr[2][FM];
                auto load_quad = [&](int b, uint2 (&dst)[FM]) {
                    int nq = ch0 + (b >> 1) * 32 + (b & 1) * 4;
                    asm volatile("" : "+v"(nq));
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int tr = wrow + i * 16;
                        const long po = sPo[tbl + tr];
                        const bool ok = tr < rows_tile && po >= 0 && nq < Cout_;
                        dst[i] = *(const uint2*)(rg + (ok ? po * ldr_ + nq : 0l));
                    }
                };
                load_quad(0, zr[0]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b + 1 < 4) load_quad(b + 1, zr[(b + 1) & 1]);
                    int nq = ch0 + (b >> 1) * 32 + (b & 1) * 4;
                    asm volatile("" : "+v"(nq));
                    float sc[4], sf[4], s1[4], s2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ne = min(nq + e, Cout_ - 1);
                        sc[e] = scale_[ne], sf[e] = shift_[ne];
                        s1[e] = 0.f, s2[e] = 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int tr = wrow + i * 16;
                        const long po = sPo[tbl + tr];
                        const bool row_ok = tr < rows_tile && po >= 0 && nq < Cout_;
                        const f32x4 av = acc[i][b];
                        uint2 pk;
                        pk.x = VecIO<bf16_t>::pack2(av[0], av[1]);
                        pk.y = VecIO<bf16_t>::pack2(av[2], av[3]);
                        if (row_ok) {
                            const uint2 zz = zr[b & 1][i];
                            const float g4[4] = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u),
                                                 __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u)};
                            const float z4[4] = {__uint_as_float(zz.x << 16), __uint_as_float(zz.x & 0xffff0000u),
                                                 __uint_as_float(zz.y << 16), __uint_as_float(zz.y & 0xffff0000u)};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float gg = (!relu || fmaf(z4[e], sc[e], sf[e]) > 0.f) ? g4[e] : 0.f;
                                s1[e] += gg;
                                s2[e] = fmaf(gg, z4[e], s2[e]);
                            }
                            *(uint2*)(yg + (po * ldy_ + nq)) = pk;
                        }
                    }
                    float u = 0.f, v = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x1 = row_sum16(s1[e]), x2 = row_sum16(s2[e]);
                        u = c16 == e ? x1 : u;
                        v = c16 == e ? x2 : v;
                    }
                    const int nn = nq + c16;
                    if (c16 < 4 && nn < Cout_) {
                        v = (v - mean_[nn] * u) * istd_[nn];
                        vt_stat_add(stats_, ((long)rep * 2 + 0) * Cout_ + nn, u);
                        vt_stat_add(stats_, ((long)rep * 2 + 1) * Cout_ + nn, v);
                    }
                }
            } else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // (opaque: keeps the per-lane 64-bit output / statistics addresses from being hoisted out of the tile
                //  loop, where they would sit in scratch across the whole step loop)
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
                // the residual rows of this half, all in flight before the first one is used: loaded unconditionally
                // (rows / channels outside the tensor read its first 16 bytes) so that nothing orders them behind the
                // stores below -- one dependent load per row fragment cost 90 us of a 160-channel YOLOv5x layer's 400
                // (MODE 0 too: the accumulating data gradients of VoVNet-39's OSA chains -- its step 25.37 -> 25.03 ms)
                constexpr bool kPreRes = MODE != 1;
                uint4 rres[kPreRes ? FM : 1];
                if (kPreRes && has_res) {
#pragma unroll
                    for (int i = 0; i < (kPreRes ? FM : 0); ++i) {
                        const int tr = wrow + i * 16;
                        const long po = sPo[tbl + tr];
                        const bool ok = tr < rows_tile && po >= 0 && n < Cout_;
                        rres[i] = *(const uint4*)(rg + (ok ? po * ldr_ + n : 0l));
                    }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int tr = wrow + i * 16;       // row inside the tile
                    const long po = sPo[tbl + tr];      // output pixel of this padded position, -1 for padding
                    const bool row_ok = tr < rows_tile && po >= 0;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = acc[i][2 * h + (e >> 2)][e & 3];
                        if (affine) t = fmaf(t, sc[e], sf[e]);
                        if (affine && relu) t = fmaxf(t, 0.f);
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
                            const uint4 rr = kPreRes ? rres[kPreRes ? i : 0] : *(const uint4*)(rg + (po * ldr_ + n));
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
                    // sum over the 16 pixel lanes (same q4 = one DPP row): four row rotations on the vector ALU (no LDS
                    // crossbar round trips), then lanes c16 = 0..7 keep channel e = c16
                    float u = 0.f, v = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x1 = row_sum16(s1[e]), x2 = row_sum16(s2[e]);
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
            if (wave == 0 && k < 4) VT_S6_STAMP(6 + 3 * k);
        };
        switch (fm) {
            case 4: run(I_<4>{}); break;
            case 5:
                if constexpr (kFMX > 5) {
                    run(I_<5>{});
                    break;
                }
            case 6:
                if constexpr (kFMX > 6) {
                    run(I_<6>{});
                    break;
                }
            default: run(I_<kFMX>{}); break;
        }
    }
    if (!KSPLIT && grp == 0) wg_barrier();  // tick 2S: group 1's last MFMA tick
    if (VT_DBG(16) && wave == 0 && lane == 0 && blockIdx.x < 512) {
        vt_span6_stamps[blockIdx.x * 16 + 15] = cwait;
        vt_span6_stamps[blockIdx.x * 16 + 3] = cR;
        vt_span6_stamps[blockIdx.x * 16 + 12] = cM;
    }
#undef VT_G_NUN
#undef VT_G_U0
#undef VT_TILE_U0
#undef VT_TILE_F
}

}  // namespace

// returns -1 when this kernel does not apply; `dry`: every check, no launch
static int span6_run(IgemmArgs& a0, int dtype, void* stream, bool dry) {
    // VT_SPAN6=0 disables, =2 forces this kernel wherever it applies (tests); default: the layers it measured faster
    // on than vt_igemm_span.hip (128-wide filter tiles on maps of 20 x 20 and larger, where the padded coordinates
    // cost <= 10 % extra MFMA work)
    const int enabled = VT_KNOB("VT_SPAN6", 1);
    if (!enabled || dtype != VT_BF16) return -1;
    // one 12-wave workgroup per CU on 8 XCDs x 32 CUs with 160 KiB of LDS each: the grid, the row slots and the unit
    // split below are built for exactly that chip; any other device (or a partitioned one) takes the span kernel
    if (vt_device_cus() != 256) return -1;
    if (enabled < 2 && a0.Cout < 128) return -1;
    const bool small_map = a0.Wi < 14 || a0.Hi < 14;  // (without KSPLIT: not below 14 x 14, where it is 6 % faster: 74.2 -> 69.7 us)
    if (a0.sh != 1 || a0.sw != 1 || a0.Ho != a0.Hi || a0.Wo != a0.Wi) return -1;
    // (Cin >= 64: with a single channel chunk the next tile's piece sources would be needed before they are computed)
    if (a0.Cin % 32 != 0 || a0.Cin < 64 || a0.ntaps != 9 || a0.Cout < 64) return -1;
    if ((long)a0.M + 2L * a0.Wi * VT_MAX_TAPS > 0x7fffffffL) return -1;
    if ((long)a0.B * a0.oH * a0.oW > 0x7fffffffL) return -1;
    if ((unsigned long)a0.M * a0.ldx * 2 >= 0xffff0000ul) return -1;
    if ((unsigned long)a0.Cout * a0.ldw * 2 >= 0xffff0000ul) return -1;
    S6Args a;
    a.p = a0;
    IgemmArgs& p = a.p;
    a.debug = kDiag ? VT_KNOB("VT_SPAN6_ABL", 0) : 0;  // (diagnostic builds only)
    // every tap within one pixel of the centre (3x3, padding 1: forward and stride-1 data gradient)
    for (int t = 0; t < 9; ++t) {
        const int eh = a0.h0 + a0.dh[t], ew = a0.w0 + a0.dw[t];
        if (eh < -1 || eh > 1 || ew < -1 || ew > 1) return -1;
        a.tsel[t] = (eh + 1) | ((ew + 1) << 2);
    }
    p.tiles_n = (p.Cout + 127) / 128;
    if (p.tiles_n > 8) return -1;
    const int g8 = 32 - 32 % p.tiles_n;  // working workgroups per XCD (one per CU; 32 % tiles_n CUs per XCD sit out)
    a.rslots = g8 / p.tiles_n;
    int maxnun = 0;  // units of the workgroup with the most rows (set by geometry())
    // Rows: the positions of the padded (H+1) x (W+1) image (pad = 1: no masks anywhere), or the pixels themselves with a
    // per-tap validity bit per row (pad = 0: the MASKED kernel).  Returns the cost of the slowest workgroup -- its tile rounds
    // plus the 32-row units of its larger group: a launch's time follows (rounds x a + units x b) x K-steps with a ~ b,
    // fitted on nine layer shapes of the three models (NOTEBOOK R5.17) -- or 0 when the geometry does not fit.
    auto geometry = [&](int pad, bool ks = false) -> int {
        a.Hp = a0.Hi + pad, a.Wp = a0.Wi + pad;
        if ((long)a0.B * a.Hp * a.Wp > 0x3fffffffL) return 0;
        a.Mp = a0.B * a.Hp * a.Wp;
        for (int t = 0; t < 9; ++t) a.dtap[t] = (a.tsel[t] & 3) * a.Wp + (a.tsel[t] >> 2);
        a.dmin = -a.Wp - 1;
        a.halo = 2 * a.Wp + 2;
        a.units = (a.Mp + 31) / 32;
        a.upx = (a.units + 7) / 8;
        // tile height: the tallest one whose span (rows + halo) fits the 28 pieces of a chunk slot -- 7 units up to 80-pixel
        // maps, 6 up to 112 (VoVNet-39's 64 -> 128 @112x112), 4 up to ~140
        a.fmx = kFMX;
        for (;;) {
            a.npc = ((32 * a.fmx + a.halo + 15) / 16 + 3) / 4 * 4;  // a multiple of 4: every loader issues one piece per tap 0..npc/4-1
            if (a.npc <= 28 || a.fmx == 4) break;
            --a.fmx;
        }
        a.ppt = (a.npc + 5) / 6;
        if (a.npc < 16 || a.npc > 28) return 0;  // 4..7 taps carry one piece per loader and group
        if (L6::bytes(a.npc, ks) > 160 * 1024) return 0;
        int cost = 1;
        maxnun = 0;
        for (int xcd = 0; xcd < 8; ++xcd) {  // (the kernel's own split)
            const int ux0 = xcd * a.upx, ux1 = std::min(a.units, ux0 + a.upx);
            const int nx = std::max(0, ux1 - ux0);
            for (int rs = 0; rs < a.rslots; ++rs) {
                const int nun = (int)((unsigned)(rs + 1) * (unsigned)nx / (unsigned)a.rslots) - (int)((unsigned)rs * (unsigned)nx / (unsigned)a.rslots);
                cost = std::max(cost, ((nun + 1) / 2 + a.fmx - 1) / a.fmx + (nun + 1) / 2);
                maxnun = std::max(maxnun, nun);
            }
        }
        return cost;
    };
    // VT_SPAN6_MASK: 0 never, 1 (default) where the pixel rows save a tile round, 2 wherever they fit (tests)
    const int maskk = VT_KNOB("VT_SPAN6_MASK", 1);
    // KSPLIT (VT_SPAN6_KSPLIT: 0 never, 1 default): the whole row share of EVERY workgroup fits one group's tile -- then both
    // groups take the same rows (pixel rows: fewest units) and half of the input channels each.  At least four chunks per
    // group; the hand-over of group 1's accumulators (4 waves x 28 KiB) re-uses the ring and the span slots.
    bool ksplit = false;
    if (VT_KNOB("VT_SPAN6_KSPLIT", 1) && maskk && a0.Cin % 64 == 0 && a0.Cin >= 256 && (long)p.M * p.tiles_n >= 24576L) {
        if (geometry(0, true) > 0 && maxnun <= a.fmx && kNSB * 2 * kBSlot + 4 * a.npc * 1024 >= 4 * kFMX * 4 * 1024) ksplit = true;
    }
    if (!ksplit) {
        if (enabled < 2 && small_map) return -1;
        // MFMA-bound layers only: enough rows to give every compute group at least 4 units
        if ((long)p.M * p.tiles_n < 512L * 32 * 4) return -1;
    }
    const int r_masked = (maskk && !ksplit) ? geometry(0) : 0;
    const int r_padded = ksplit ? 0 : geometry(1);
    bool masked = false;
    // (a masked step costs ~5 % more than a padded one -- two vector instructions per fragment read -- so the pixel rows are
    //  taken where the model promises 10 % or more: one tile instead of two at 14 x 14 (-25 %), three instead of four at 40 x 40
    //  (-7 %); not at 28 x 28 (16 : 15, measured -2.5 % / 0 %), 20 x 20 (12 : 11, +5 %), 56 x 56 (59 : 56, +1 %), 112 x 112 (+7 %))
    if (ksplit) {
        masked = true;
        geometry(0, true);
    } else if (r_masked > 0 && (r_padded == 0 || maskk >= 2 || 10 * r_padded > 11 * r_masked)) {
        masked = true;
        geometry(0);
    } else if (r_padded == 0) {
        return -1;
    }
    const int smem = L6::bytes(a.npc, ksplit);
    a.s64j = 64 % a.Wp, a.s64i = (64 / a.Wp) % a.Hp, a.s64b = (64 / a.Wp) / a.Hp;
    a.hp_magic = (unsigned)((0x100000000ull + a.Hp - 1) / a.Hp);
    a.wp_magic = (unsigned)((0x100000000ull + a.Wp - 1) / a.Wp);
    const int mode = (p.flags & VT_CONV_BNRED) ? 3 : ((p.flags & VT_CONV_STATS) ? 1 : ((p.flags & VT_CONV_AFFINE) ? 2 : 0));
    if (mode == 1 && (p.flags & (VT_CONV_AFFINE | VT_CONV_RELU | VT_CONV_RESIDUAL))) return -1;
    if (mode == 0 && (p.flags & VT_CONV_RELU)) return -1;
    if (mode == 3 && ((p.flags & (VT_CONV_AFFINE | VT_CONV_STATS | VT_CONV_RESIDUAL | VT_CONV_D2S)) || !p.res || !p.shift || !p.aux0 ||
                      !p.aux1 || !p.stats))
        return -1;
#define VT_S6_PICK(KS, MK)                                                                       \
    (mode == 1 ? span6_kernel<1, MK, KS>                                                         \
               : (mode == 2 ? span6_kernel<2, MK, KS> : (mode == 3 ? span6_kernel<3, MK, KS> : span6_kernel<0, MK, KS>)))
    auto kern = ksplit ? VT_S6_PICK(true, true) : (masked ? VT_S6_PICK(false, true) : VT_S6_PICK(false, false));
#undef VT_S6_PICK
    {
        // (in the dry run too: the only fallible step of a launch, so a caller that splits the columns over two kernels
        //  knows this half cannot fail once the other one has been issued)
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(span6)");
        if (rc != VT_OK) return rc;
    }
    if (dry) return VT_OK;
    vt_note_kernel(ksplit ? "span6_kernel<bf16,2x4+4 waves,FM%d,masked,ksplit>"
                          : (masked ? "span6_kernel<bf16,2x4+4 waves,FM%d,masked>" : "span6_kernel<bf16,2x4+4 waves,FM%d>"), kFMX);
    hipLaunchKernelGGL(kern, dim3(8 * 32), dim3(768), smem, (hipStream_t)stream, a);
    VT_CHECK_LAUNCH("vt_conv_igemm(span6)");
    if (kDiag && (a.debug & 16)) {
        static int calls = 0;
        if (++calls == 12) {  // a warm launch
            (void)hipStreamSynchronize((hipStream_t)stream);
            static unsigned long long h[512 * 16];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(vt_span6_stamps), sizeof(h));
            const int nb = 256;
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < nb; ++b) t0 = h[b * 16] < t0 ? h[b * 16] : t0;
            double avg[16] = {0};
            for (int b = 0; b < nb; ++b)
                for (int k = 0; k < 16; ++k) avg[k] += ((k >= 12 || k == 3) ? (double)h[b * 16 + k] : (double)(h[b * 16 + k] - t0) * 0.01) / nb;
            fprintf(stderr, "[span6 stamps, us from the first workgroup's start, mean over %d WGs] loader start %.1f prologue done %.1f loop done %.1f |"
                            " tile0: first tick %.1f loop end %.1f epilogue end %.1f | tile1: %.1f %.1f %.1f | tile2: %.1f %.1f %.1f\n",
                    nb, avg[0], avg[1], avg[2], avg[4], avg[5], avg[6], avg[7], avg[8], avg[9], avg[10], avg[11], avg[12]);
            std::vector<double> st, en;
            for (int b = 0; b < nb; ++b) st.push_back((h[b * 16] - t0) * 0.01), en.push_back((h[b * 16 + 2] - t0) * 0.01);
            std::sort(st.begin(), st.end());
            std::sort(en.begin(), en.end());
            fprintf(stderr, "[span6 stamps] shader cycles, mean per WG: loader 0 waiting in barriers %.0f of %.0f in its loop; compute wave 0 waiting in barriers %.0f, read-tick work %.0f, MFMA-tick work %.0f\n",
                    avg[13], avg[14], avg[15], avg[3], avg[12]);
            {
                static unsigned long long hp[512 * 16];
                (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(vt_span6_pstamps), sizeof(hp));
                double pa[6] = {0};
                for (int b = 0; b < nb; ++b)
                    for (int k = 0; k < 6; ++k) pa[k] += (double)(hp[b * 16 + k] - t0) * 0.01 / nb;
                fprintf(stderr, "[span6 prologue stamps, us from the first workgroup's start, mean over %d WGs] loader: slices issued %.2f piece sources read back (tick 0) %.2f | "
                                "compute wave 0: piece sources done %.2f pieces issued %.2f landed + row table %.2f\n", nb, pa[0], pa[1], pa[2], pa[3], pa[4]);
            }
            fprintf(stderr, "[span6 stamps] start times (us), sorted, every 32nd WG:");
            for (int b = 0; b < nb; b += 32) fprintf(stderr, " %.1f", st[b]);
            fprintf(stderr, "\n[span6 stamps] loader end times (us), sorted, every 32nd WG:");
            for (int b = 0; b < nb; b += 32) fprintf(stderr, " %.1f", en[b]);
            fprintf(stderr, "\n");
        }
    }
    return VT_OK;
}

// returns -1 when this kernel does not apply (the caller then tries the other span kernels)
int vt_span6_dispatch(IgemmArgs& a0, int dtype, void* stream) {
    // 128 k + 32 output channels (160: Darknet-YOLOv5x, VoVNet-39): the filter tiles are 128 wide, so the last one would
    // be a quarter full and cost as much as a full one.  The first 128 k columns run here, the last 32 as a second launch
    // over the same input.  Not with batch statistics (their buffer is indexed by the launch's own channel count), not for
    // 64 remaining columns (192 @14x14: 60 us whole, 42 + 32 split).  VT_SPAN6_SPLIT=0: never.
    const int rem = a0.Cout % 128;
    const int split = VT_KNOB("VT_SPAN6_SPLIT", 1);
    if (split && a0.Cout > 128 && rem == 32 && dtype == VT_BF16 && !(a0.flags & (VT_CONV_STATS | VT_CONV_D2S | VT_CONV_NOSTORE | VT_CONV_BNRED))) {
        const int head = a0.Cout - rem;
        IgemmArgs a1 = a0;
        a1.Cout = head;
        if (span6_run(a1, dtype, stream, true) == VT_OK) {
            IgemmArgs a2 = a0;
            a2.Cout = rem;
            a2.w = (const char*)a0.w + (long)head * a0.ldw * 2;
            a2.y = (char*)a0.y + (long)head * 2;
            if (a0.res) a2.res = (const char*)a0.res + (long)head * 2;
            if (a0.scale) a2.scale = a0.scale + head;
            if (a0.shift) a2.shift = a0.shift + head;
            // The 32-column tail.  Round 4: on the persistent resident-filter kernel where it applies (>= 262k rows, a filter
            // of <= 96 KB: 160 -> 160 @80x80 x 64 images -- 63 us, the pair 292 us against 364 whole; Darknet-YOLOv5x forward
            // 13.33 -> 13.06 ms), forward launches included.  Else on the input-span kernel's 32-wide tile, data gradients
            // only (inside the YOLOv5x forward that pair measured no faster than the whole launch; VT_SPAN6_SPLIT=2: those too).
            int rc2 = vt_pspan_dispatch(a2, dtype, stream);
            if (rc2 == -1 && (split >= 2 || !(a0.flags & VT_CONV_AFFINE))) rc2 = vt_span_dispatch(a2, dtype, stream);
            if (rc2 == VT_OK) return span6_run(a1, dtype, stream, false);
            if (rc2 != -1) return rc2;
        }
    }
    return span6_run(a0, dtype, stream, false);
}
