// ARCHIVED (round 5, NOTEBOOK R5.14): built, 42 cases bit-identical to span6, 13 % fewer cycles per step pair, the same wall
// time; not in the build.  To try it: add it to csrc/Makefile and call vt_span7_dispatch before vt_span6_dispatch in vt_igemm.hip.
//
// vt_igemm_span7.hip -- vt_igemm_span6.hip with TWO K-steps per tick.  Same workgroup (two compute groups half a PAIR of
// steps apart + four LDS-DMA loader waves, one workgroup per CU, persistent over 32-row units in padded coordinates), same
// LDS images, same summation order, outputs bit-identical to span6 -- but half as many workgroup barriers per MFMA.
//
// Why (round 5): span6's stamps (profiles/r04_span6_phases.json) put a step at 1461 shader cycles of which the two MFMA
// ticks need 2 x 508: each of the two barriers of a step costs ~220 cycles of latency and skew during which the matrix
// pipe of every SIMD is idle, 30 % of the loop.  A tick that carries 56 MFMAs per wave instead of 28 halves that share.
// The registers (168 per lane) do not hold the fragments of two K-steps; they do not have to:
//   * read tick of the pair (s, s+1): the row fragments and three filter fragments of step s, as in span6;
//   * MFMA tick: columns 0-2 of step s; then column 3 row by row, each row's fragment registers re-read with step s+1's
//     row right behind the MFMA that used them last; the filter fragments of s+1 land in the registers of the columns
//     that are done (B'0 behind column 1, B'1 behind column 2, B'2 behind column 3, B'3 behind column 0'); then the four
//     columns of step s+1.  The same 40 fragment registers, 56 MFMAs between two barriers, and every LDS read of step
//     s+1 has at least seven MFMAs (~120 cycles) between its issue and its use.
// Nine taps per 32-channel chunk: pairs straddle chunk boundaries ((8, 0')), so the schedule repeats every TWO chunks
// (18 steps, 9 pairs) and the kernel takes layers with an even number of chunks (Cin % 64 == 0); the filter-slice ring has
// six slots (a pair in use, a pair landed, a pair in flight), which leaves 24 span pieces per chunk slot: maps up to ~60
// pixels wide at the full tile height.  Everything else goes to span6.
//
// Loader schedule (per pair p of steps s = 2p, s+1; tick t starts with barrier t):
//   tick 2p   (group 0 reads step s; group 1: MFMA tick of pair p-1):  group 1's span pieces of pair p-1's taps
//   tick 2p+1 (group 0: MFMA tick of pair p; group 1 reads step s):    group 0's span pieces of pair p's taps,
//                                                                      slices s+4 and s+5 (2 x 2 pieces per loader)
// A group's piece of tap T of chunk c fills the span slot of chunk c+1, i.e. the slot chunk c-1 was read from: group 0
// reads a chunk's last tap no later than tick 2p+1's barrier when that tap is the pair's first step and inside tick
// 2p+1 when it is the second (then the next chunk's tap 0 is in pair p+1), group 1 one tick later -- hence the one-tick lag
// of both issues behind span6's.  Every LDS-DMA has at least three ticks between issue and first use (pieces are issued
// at taps 0..5 only), so before barrier t a loader waits for everything it issued up to tick t-3: compile-time counts.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

constexpr int kFMX = 7;             // row fragments (16 rows) per compute wave, at most
constexpr int kNSB = 6;             // filter-slice ring slots
constexpr int kBSlot = 128 * 64;    // bytes per filter slice: 128 filter rows x 32 channels
constexpr int kBMX = 32 * kFMX;     // rows of the tallest tile of one group
constexpr int kNTPX = 6;            // taps of a chunk that carry a span piece, at most (24 pieces per chunk slot)

__device__ __attribute__((aligned(16))) unsigned int vt_span7_zero16[4];  // source of every padding row

struct S7Args {
    IgemmArgs p;
    int dmin, halo;  // span row of tap t = (eh*W + ew) - dmin, in [0, halo]
    int units;       // ceil(Mp / 32)
    int upx;         // units per XCD
    int rslots;      // row slots per XCD (workgroups per XCD / tiles_n)
    int npc;         // span pieces (16 rows x 64 B) per chunk: a multiple of 4, <= 24
    int fmx;         // tallest tile of this launch in 32-row units (<= kFMX)
    int Hp, Wp, Mp;  // padded image (H+1) x (W+1) and the number of padded positions B*Hp*Wp
    unsigned hp_magic, wp_magic;  // ceil(2^32 / Hp), ceil(2^32 / Wp)
    int dtap[9];     // span row of every tap
};

__device__ __forceinline__ int swz4(int g) { return (0x1320 >> ((g & 3) * 4)) & 3; }  // filter-slice image
__device__ __forceinline__ int swzA(int g) { return (g & 1) << 1; }                   // span image (span6's)

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long)hi << 32) | lo);
}
// LDS-DMA, 16 B per lane: LDS address = M0 + lane*16, global address = sbase + voff (or the per-lane address)
// (s_nop 4: a VMEM instruction that reads an SGPR written by a VALU instruction needs 5 wait states, which hipcc does not
//  insert in front of an asm statement)
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

// LDS map (bytes): [row output pixel 2 groups x 2 x kBMX x 4][filter ring kNSB x 8 KiB]
//                  [group 0: span slot 0, slot 1][group 1: span slot 0, slot 1]
struct L7 {
    static constexpr int kPo = 0;
    static constexpr int kB = kPo + 4 * kBMX * 4;
    static constexpr int kA = kB + kNSB * kBSlot;
    __host__ __device__ static constexpr int bytes(int npc) { return kA + 4 * npc * 1024; }
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

typedef const __attribute__((address_space(4))) S7Args* ArgsPtr;
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

// step u of the 18-step period of two chunks: its tap, and which of the two chunks it belongs to
constexpr int tap_of(int u) { return ((u % 18) + 18) % 18 % 9; }
constexpr int half_of(int u) { return ((u % 18) + 18) % 18 / 9; }

// 12 waves: 0-3 compute group 0, 4-7 compute group 1, 8-11 loaders; three per SIMD = at most 168 registers
template <int MODE>  // epilogue: 0 plain (+ residual), 1 BatchNorm statistics, 2 affine (+ ReLU, + residual)
__global__ void __launch_bounds__(768, 3) span7_kernel(const S7Args a) {
    const IgemmArgs& p = a.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sPo = (int*)(smem + L7::kPo);
    const char* sBb = smem + L7::kB;
    const char* sAb = smem + L7::kA;
    const int aslot_bytes = a.npc * 1024;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's share: a contiguous range of 32-row units of one XCD, one filter column tile (as span6) ----
    const int bid = blockIdx.x, xcd = bid & 7, l = bid >> 3;
    if (l >= a.rslots * p.tiles_n) return;  // (tiles_n does not divide 32)
    const int tn = l % p.tiles_n, rs = l / p.tiles_n;
    const int ux0 = xcd * a.upx, ux1 = min(a.units, ux0 + a.upx);
    const int nx_ = max(0, ux1 - ux0);
    const int ua = __builtin_amdgcn_readfirstlane(ux0 + (int)((unsigned)rs * (unsigned)nx_ / (unsigned)a.rslots));
    const int ub = __builtin_amdgcn_readfirstlane(ux0 + (int)((unsigned)(rs + 1) * (unsigned)nx_ / (unsigned)a.rslots));
    const int nun = ub - ua;
    if (nun <= 0) return;
    const int nun0 = (nun + 1) >> 1;
    const int ntile = __builtin_amdgcn_readfirstlane((nun0 + a.fmx - 1) / a.fmx);
    const int nchunks = __builtin_amdgcn_readfirstlane(p.Cin / 32);  // even
    const int nsteps = nchunks * 9;
    const int npairs = nsteps >> 1;  // per tile
#define VT_G_NUN(g) ((g) ? nun - nun0 : nun0)
#define VT_G_U0(g) ((g) ? ua + nun0 : ua)
#define VT_TILE_U0(g, k) (VT_G_U0(g) + (k) * (VT_G_NUN(g) / ntile) + min((k), VT_G_NUN(g) % ntile))

    if (wave >= 8) {
        // =========================== loader waves ==================================================
        const int lj = wave - 8;  // 0..3
        const char* xg = (const char*)p.x;
        const char* wg = (const char*)p.w;
        const unsigned a_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + L7::kA);
        const unsigned b_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + L7::kB);
        const unsigned m0_keep = get_m0();
        const long ldx2 = (long)p.ldx * 2;
        const int cin2 = p.Cin * 2;
        const int Mp = a.Mp, Wp = a.Wp, Hp = a.Hp, W_ = p.Wi, H_ = p.Hi;
        const unsigned wp_magic = a.wp_magic, hp_magic = a.hp_magic;
        const unsigned long zero_src = (unsigned long)(const void*)vt_span7_zero16;
        // span piece = 16 rows x 64 B: lane owns row (lane>>2), source chunk (lane&3)^swzA(lane>>4)
        const int cjA = (lane & 3) ^ swzA(lane >> 4);
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
        auto span_src = [&](int mp, bool& ok) -> unsigned long {
            const bool inr = (unsigned)mp < (unsigned)Mp;
            int b, i, j;
            ok = unpad(inr ? mp : 0, b, i, j) && inr;
            const long pix = ((long)b * H_ + i) * W_ + j;
            return ok ? (unsigned long)xg + (unsigned long)(pix * ldx2 + cjA * 16) : zero_src;
        };
        // piece T of group g = span rows 16*(lj + 4T) ..: its source per tile (channel chunk 0), and whether it is a pixel
        unsigned long ab_cur[2][kNTPX], ab_nxt[2][kNTPX];
        unsigned vm_cur[2] = {0, 0}, vm_nxt[2] = {0, 0};
        auto tile_bases = [&](int m0t, unsigned long (&ab)[kNTPX], unsigned& vm) {
            vm = 0;
#pragma unroll
            for (int T = 0; T < kNTPX; ++T) {
                bool ok;
                ab[T] = span_src(m0t + a.dmin + (lj + 4 * T) * 16 + (lane >> 2), ok);
                vm |= (ok ? 1u : 0u) << T;
            }
        };
        // filter slice = 8 pieces of 16 rows, this loader's are q = 2*lj, 2*lj+1 (as span6)
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
                set_m0(b_base + (unsigned)(slot * kBSlot + (2 * lj + i) * 1024));
                glds_s(b_voff[i], sb);
            }
        };
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

        // ---- prologue: slices 0..3, both groups' first span chunk, the first tiles' row tables -----------
        const long S = (long)ntile * nsteps;  // steps of each group
        int m0c[2] = {VT_TILE_U0(0, 0) * 32, VT_TILE_U0(1, 0) * 32};
        int sic = 0, sT = 0;  // (chunk, tap) of the next slice to issue; slices repeat per tile
        long sg = 0;          // its step
        int sslot = 0;        // its ring slot
        auto next_slice = [&]() {
            if (sg < S) {
                issue_slice(sslot, sic, sT);
                ++sg;
                sslot = sslot + 1 == kNSB ? 0 : sslot + 1;
                if (++sT == 9) {
                    sT = 0;
                    if (++sic == nchunks) sic = 0;
                }
            }
        };
        next_slice();
        next_slice();
        next_slice();
        next_slice();
        const int ntp = a.npc >> 2;  // taps that carry a piece
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            tile_bases(m0c[g], ab_cur[g], vm_cur[g]);
#pragma unroll
            for (int T = 0; T < kNTPX; ++T)
                if (T < ntp) {
                    set_m0(a_base + (unsigned)((g * 2) * aslot_bytes + (lj + 4 * T) * 1024));
                    glds_v(ab_cur[g][T]);
                }
        }
        row_tables(0, 0, m0c[0]);
        row_tables(1, 0, m0c[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        // group 1's pieces of the previous pair's taps, issued one tick after group 0's
        unsigned long pend_src[2] = {0, 0};
        unsigned pend_m0[2] = {0, 0};
        long pleft = (long)ntile * npairs;  // pairs left, the current one included
        const long ptotal = pleft;

        auto tiles = [&](auto NTPc) {
            constexpr int NTP = decltype(NTPc)::value;
            constexpr auto P = [](int t) { return (((t % 9) + 9) % 9) < NTP ? 1 : 0; };
            for (int k = 0; k < ntile; ++k) {
                const bool has_next = k + 1 < ntile;
                int m0n[2] = {0, 0};
                if (has_next) m0n[0] = VT_TILE_U0(0, k + 1) * 32, m0n[1] = VT_TILE_U0(1, k + 1) * 32;
                for (int icp = 0; icp < nchunks; icp += 2) {
                    const bool lastp = icp + 2 == nchunks;
                    // pieces of the taps of chunk icp fill chunk icp+1's spans (slot 1); those of chunk icp+1 the spans of
                    // chunk icp+2 or of the next tile's chunk 0 (slot 0) -- or re-load chunk icp+1, unused, at the very end
                    const bool nx1 = lastp && has_next;
                    const unsigned long cb0 = (unsigned long)(icp + 1) * 64;
                    const unsigned long cb1 = (unsigned long)(lastp ? (has_next ? 0 : icp + 1) : icp + 2) * 64;
                    auto pair = [&](auto Jc) {
                        constexpr int J = decltype(Jc)::value;
                        constexpr int u0 = 2 * J, u1 = 2 * J + 1;
                        constexpr int T0 = tap_of(u0), T1 = tap_of(u1), H0 = half_of(u0), H1 = half_of(u1);
                        constexpr int Q0 = tap_of(u0 - 2), Q1 = tap_of(u0 - 1);  // the previous pair's taps
                        constexpr int R0 = tap_of(u0 - 4), R1 = tap_of(u0 - 3);  // the one before
                        // in flight at a barrier: what this wave issued in the two ticks before it
                        constexpr int kEven = P(R0) + P(R1) + P(Q0) + P(Q1) + 4;
                        constexpr int kOdd = 2 * (P(Q0) + P(Q1)) + 4;
                        const bool steady = pleft > 3 && pleft + 2 <= ptotal;
                        // ---- even tick 2p: group 0 reads step s
                        if (steady) vmw<kEven>();
                        else vmw<0>();
                        wg_barrier();
                        if constexpr (P(Q0)) {
                            set_m0(pend_m0[0]);
                            glds_v(pend_src[0]);
                        }
                        if constexpr (P(Q1)) {
                            set_m0(pend_m0[1]);
                            glds_v(pend_src[1]);
                        }
                        // the next tiles' row tables and piece sources, behind taps 6 / 7 of this tile's first chunk: the table
                        // halves they go to were last read by the previous tiles' epilogues
                        if (J == 3 && icp == 0 && has_next) {
                            row_tables(0, (k + 1) & 1, m0n[0]);
                            tile_bases(m0n[0], ab_nxt[0], vm_nxt[0]);
                        }
                        // ---- odd tick 2p+1: group 1 reads step s; group 0's MFMA tick
                        if (steady) vmw<kOdd>();
                        else vmw<0>();
                        wg_barrier();
                        auto piece = [&](auto Tc, auto Hc, int which) {
                            constexpr int T = decltype(Tc)::value;
                            constexpr int H = decltype(Hc)::value;
                            if constexpr (T < NTP) {
                                const bool nx = H == 1 && nx1;
                                const unsigned long cb = H == 0 ? cb0 : cb1;
                                const int slot = H == 0 ? 1 : 0;
#pragma unroll
                                for (int g = 0; g < 2; ++g) {
                                    const unsigned long base = nx ? ab_nxt[g][T] : ab_cur[g][T];
                                    const unsigned v = ((nx ? vm_nxt[g] : vm_cur[g]) >> T) & 1u;
                                    const unsigned m0v = a_base + (unsigned)((g * 2 + slot) * aslot_bytes + (lj + 4 * T) * 1024);
                                    const unsigned long src = base + (v ? cb : 0ul);
                                    if (g == 0) {
                                        set_m0(m0v);
                                        glds_v(src);
                                    } else {
                                        pend_m0[which] = m0v;
                                        pend_src[which] = src;
                                    }
                                }
                            }
                        };
                        piece(I_<T0>{}, I_<H0>{}, 0);
                        piece(I_<T1>{}, I_<H1>{}, 1);
                        next_slice();
                        next_slice();
                        if (J == 3 && icp == 0 && has_next) {
                            row_tables(1, (k + 1) & 1, m0n[1]);
                            tile_bases(m0n[1], ab_nxt[1], vm_nxt[1]);
                        }
                        --pleft;
                    };
                    pair(I_<0>{});
                    pair(I_<1>{});
                    pair(I_<2>{});
                    pair(I_<3>{});
                    pair(I_<4>{});
                    pair(I_<5>{});
                    pair(I_<6>{});
                    pair(I_<7>{});
                    pair(I_<8>{});
                }
                m0c[0] = m0n[0], m0c[1] = m0n[1];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    vm_cur[g] = vm_nxt[g];
#pragma unroll
                    for (int T = 0; T < kNTPX; ++T) ab_cur[g][T] = ab_nxt[g][T];
                }
            }
        };
        switch (ntp) {
            case 4: tiles(I_<4>{}); break;
            case 5: tiles(I_<5>{}); break;
            default: tiles(I_<6>{}); break;
        }
        wg_barrier();  // tick 2P: group 1's last MFMA tick
        vmw<0>();
        set_m0(m0_keep);
        return;
    }

    // =============================== compute waves ==================================================
    const int grp = wave >> 2;  // 0: reads in even ticks, MFMAs in odd ticks; 1: one tick later
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int q4 = lane >> 4, c16 = lane & 15;
    // this lane's output channels: ch(h, e8) = tn*128 + wn*64 + h*32 + q4*8 + e8, h = 0,1, e8 = 0..7
    const int ch0 = tn * 128 + wn * 64 + q4 * 8;
    // filter fragment j of this lane: MFMA row r = c16 -> slice row n_j = wn*64 + (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3)
    const int nb0 = wn * 64 + (c16 >> 2) * 8 + (c16 & 3);
    const int b_lane = (nb0 * 4 + (q4 ^ swz4(c16 >> 2))) * 16;  // byte offset inside a slice; j adds {0,256,2048,2304}
    const int g_nun = VT_G_NUN(grp), g_u0 = VT_G_U0(grp);
    const int g_tb = g_nun / ntile, g_te = g_nun % ntile;
    const char* sAg = sAb + grp * 2 * aslot_bytes;  // this group's two span slots
    int bslot = 0;  // ring slot of the coming pair's first slice

    if (grp == 1) wg_barrier();  // tick 0: group 0 reads its first step

    for (int k = 0; k < ntile; ++k) {
        const int par = k & 1;
        const int f_cur = g_tb + (k < g_te ? 1 : 0);
        const long m0_cur = (long)(g_u0 + k * g_tb + min(k, g_te)) * 32;
        const int fm = max(f_cur, 4);       // row fragments per wave in this tile (4..kFMX)
        const int rows_tile = 32 * f_cur;   // rows this tile owns (stores / statistics)
        const int tbl = (grp * 2 + par) * kBMX;

        auto run = [&](auto FMc) {
            constexpr int FM = decltype(FMc)::value;
            int wrow = wm * 16 * FM + c16;  // this lane's row inside the tile, fragment 0
            asm volatile("" : "+v"(wrow));    // (opaque: nothing derived from it is hoisted out of the tile loop)
            auto row_off = [&](int srow) { return (unsigned)((srow * 4 + (q4 ^ swzA(srow >> 2))) * 16); };
            unsigned a_off = row_off(wrow + fresh_args()->dtap[0]);  // byte offset (inside this group's span slots) of this lane's
                                                                     // fragment-0 row of the coming pair's first step
            f32x4 acc[FM][4];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

            for (int icp = 0; icp < nchunks; icp += 2) {
                bslot = __builtin_amdgcn_readfirstlane(bslot);
                auto pair = [&](auto Jc) {
                    constexpr int J = decltype(Jc)::value;
                    constexpr int u1 = 2 * J + 1, u2 = 2 * J + 2;
                    constexpr int T1 = tap_of(u1), H1 = half_of(u1), T2 = tap_of(u2), H2 = half_of(u2);
                    // ---- read tick: step s's slice and span rows are in LDS
                    wg_barrier();
                    const char* A = sAg + a_off;
                    const char* Bt = sBb + ((bslot << 13) + b_lane);
                    const int bs1 = bslot + 1 == kNSB ? 0 : bslot + 1;
                    const char* Bu = sBb + ((bs1 << 13) + b_lane);
                    uint4 af[FM], bf0, bf1, bf2;
                    bf0 = *(const uint4*)(Bt);
                    bf1 = *(const uint4*)(Bt + 256);
                    bf2 = *(const uint4*)(Bt + 2048);
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[i] = *(const uint4*)(A + i * 1024);
                    // the span row offsets of step s+1 and of the next pair's first step: scalar loads, back before the barrier
                    const int d1 = fresh_args()->dtap[T1], d2 = fresh_args()->dtap[T2];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers
                    // ---- MFMA tick (the other group reads meanwhile): 2 x 4 columns
                    wg_barrier();
// the MFMA as an asm statement whose accumulator is an in/out operand (see span6)
#define VT_MMA(i, j, bfrag)                                                                              \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                              \
                 : "+v"(acc[i][j])                                                                      \
                 : "v"(__builtin_bit_cast(bf16x8, bfrag)), "v"(__builtin_bit_cast(bf16x8, af[i])))
#define VT_MMA_COL(bfrag, j) _Pragma("unroll") for (int i = 0; i < FM; ++i) VT_MMA(i, j, bfrag)
                    VT_MMA_COL(bf0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    bf0 = *(const uint4*)(Bt + 2304);  // B3 of step s
                    VT_MMA_COL(bf1, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    bf1 = *(const uint4*)(Bu);  // B'0
                    const char* A1 = sAg + ((unsigned)(H1 * aslot_bytes) + row_off(wrow + d1));
                    VT_MMA_COL(bf2, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    bf2 = *(const uint4*)(Bu + 256);  // B'1
                    // column 3, and behind each row's last MFMA of step s its fragment of step s+1
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        VT_MMA(i, 3, bf0);
                        __builtin_amdgcn_sched_barrier(0);
                        af[i] = *(const uint4*)(A1 + i * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    bf0 = *(const uint4*)(Bu + 2048);  // B'2
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- step s+1
                    VT_MMA_COL(bf1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    bf1 = *(const uint4*)(Bu + 2304);  // B'3
                    VT_MMA_COL(bf2, 1);
                    a_off = (unsigned)(H2 * aslot_bytes) + row_off(wrow + d2);
                    __builtin_amdgcn_sched_barrier(0);
                    VT_MMA_COL(bf0, 2);
                    VT_MMA_COL(bf1, 3);
#undef VT_MMA_COL
#undef VT_MMA
                    bslot = bs1 + 1 == kNSB ? 0 : bs1 + 1;
                };
                pair(I_<0>{});
                pair(I_<1>{});
                pair(I_<2>{});
                pair(I_<3>{});
                pair(I_<4>{});
                pair(I_<5>{});
                pair(I_<6>{});
                pair(I_<7>{});
                pair(I_<8>{});
            }

            // ---- epilogue: two 16-byte stores per row fragment, straight from the accumulators (as span6) ----------
            ArgsPtr Q = fresh_args();
            constexpr bool affine = MODE == 2, stats = MODE == 1;
            const bool relu = MODE == 2 && (Q->p.flags & VT_CONV_RELU);
            const bool has_res = MODE != 1 && (Q->p.flags & VT_CONV_RESIDUAL) != 0;
            const int Cout_ = Q->p.Cout, ldy_ = Q->p.ldy, ldr_ = Q->p.ldr;
            bf16_t* __restrict__ yg = (bf16_t*)Q->p.y;
            const bf16_t* __restrict__ rg = (const bf16_t*)Q->p.res;
            const float* scale_ = Q->p.scale;
            const float* shift_ = Q->p.shift;
            float* stats_ = Q->p.stats;
            const int rep = (int)((m0_cur / 32) % kStatReplicas);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
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
                            const uint4 rr = rres[kPreRes ? i : 0];
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
        };
        switch (fm) {
            case 4: run(I_<4>{}); break;
            case 5: run(I_<5>{}); break;
            case 6: run(I_<6>{}); break;
            default: run(I_<kFMX>{}); break;
        }
    }
    if (grp == 0) wg_barrier();  // tick 2P: group 1's last MFMA tick
#undef VT_G_NUN
#undef VT_G_U0
#undef VT_TILE_U0
}

}  // namespace

// returns -1 when this kernel does not apply (the caller goes on to span6)
int vt_span7_dispatch(IgemmArgs& a0, int dtype, void* stream) {
    // VT_SPAN7=0 disables, =2 forces this kernel wherever it applies (tests); default: where span6 runs by default
    const int enabled = VT_KNOB("VT_SPAN7", 1);
    if (!enabled || dtype != VT_BF16) return -1;
    if (vt_device_cus() != 256) return -1;
    if (enabled < 2 && (a0.Cout < 128 || a0.Wi < 14 || a0.Hi < 14)) return -1;  // (span6's default territory)
    if (a0.sh != 1 || a0.sw != 1 || a0.Ho != a0.Hi || a0.Wo != a0.Wi) return -1;
    // an even number of 32-channel chunks: pairs of steps never straddle a tile (9 taps per chunk)
    if (a0.Cin % 64 != 0 || a0.ntaps != 9 || a0.Cout < 64) return -1;
    if (a0.Cout % 128 == 32 && a0.Cout > 128) return -1;  // (span6 splits those columns over two kernels)
    if ((long)a0.M + 2L * a0.Wi * VT_MAX_TAPS > 0x7fffffffL) return -1;
    if ((long)a0.B * a0.oH * a0.oW > 0x7fffffffL) return -1;
    if ((unsigned long)a0.M * a0.ldx * 2 >= 0xffff0000ul) return -1;
    if ((unsigned long)a0.Cout * a0.ldw * 2 >= 0xffff0000ul) return -1;
    S7Args a;
    a.p = a0;
    IgemmArgs& p = a.p;
    a.Hp = a0.Hi + 1, a.Wp = a0.Wi + 1;
    if ((long)a0.B * a.Hp * a.Wp > 0x3fffffffL) return -1;
    a.Mp = a0.B * a.Hp * a.Wp;
    for (int t = 0; t < 9; ++t) {
        const int eh = a0.h0 + a0.dh[t], ew = a0.w0 + a0.dw[t];
        if (eh < -1 || eh > 1 || ew < -1 || ew > 1) return -1;
        a.dtap[t] = (eh + 1) * a.Wp + (ew + 1);
    }
    a.dmin = -a.Wp - 1;
    a.halo = 2 * a.Wp + 2;
    p.tiles_n = (p.Cout + 127) / 128;
    if (p.tiles_n > 8) return -1;
    const int g8 = 32 - 32 % p.tiles_n;
    if ((long)p.M * p.tiles_n < 512L * 32 * 4) return -1;
    a.rslots = g8 / p.tiles_n;
    a.units = (a.Mp + 31) / 32;
    a.upx = (a.units + 7) / 8;
    // the full tile height or nothing: lower tiles are span6's (their maps are wide: the halo, not the barrier, is their cost)
    a.fmx = kFMX;
    a.npc = ((32 * a.fmx + a.halo + 15) / 16 + 3) / 4 * 4;
    if (a.npc < 16 || a.npc > 4 * kNTPX) return -1;
    const int smem = L7::bytes(a.npc);
    if (smem > 160 * 1024) return -1;
    a.hp_magic = (unsigned)((0x100000000ull + a.Hp - 1) / a.Hp);
    a.wp_magic = (unsigned)((0x100000000ull + a.Wp - 1) / a.Wp);
    const int mode = (p.flags & VT_CONV_STATS) ? 1 : ((p.flags & VT_CONV_AFFINE) ? 2 : 0);
    if (mode == 1 && (p.flags & (VT_CONV_AFFINE | VT_CONV_RELU | VT_CONV_RESIDUAL))) return -1;
    if (mode == 0 && (p.flags & VT_CONV_RELU)) return -1;
    auto kern = mode == 1 ? span7_kernel<1> : (mode == 2 ? span7_kernel<2> : span7_kernel<0>);
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(span7)");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("span7_kernel<bf16,2x4+4 waves,FM%d,pairs>", kFMX);
    hipLaunchKernelGGL(kern, dim3(8 * 32), dim3(768), smem, (hipStream_t)stream, a);
    VT_CHECK_LAUNCH("vt_conv_igemm(span7)");
    return VT_OK;
}
