// ARCHIVED (round 5, NOTEBOOK R5.13): built, 50 parity cases green, no faster than the gather kernel in the step; not in the
// build.  To try it: add it to csrc/Makefile and call vt_gemm6_dispatch behind vt_span6_dispatch in vt_igemm.hip.
//
// vt_gemm6.hip -- 1x1 stride-1 convolution as a GEMM (bf16), one persistent 12-wave workgroup per CU: two compute groups
// that run half a step apart plus four LDS-DMA loader waves -- vt_igemm_span6.hip's schedule without the span.  For the
// big-K pointwise layers that the pointwise kernels (vt_pointwise.hip) do not take: the concat 1x1 of VoVNet's OSA blocks
// (reference vovnet.py:50-63: 768 -> 256 @56x56 ... 1888 -> 1024 @7x7), its data gradient (256 -> 768 ...), and the
// wide 1x1 units of Darknet-YOLOv5x's C3 stages (darknet.py:127-141) in the forward benchmark.
//
// Why (round 5): these layers ran on the round-1 gather kernel (vt_igemm.hip) at 0.20-0.26 of the MFMA peak -- eight
// lock-step waves that stage, wait, multiply in turn -- and a vendor GEMM does no better on the largest of them (768 -> 256
// @56x56 at batch 256: 0.42 ms forward, 0.63 ms data gradient; tools/diag/gemm_ceiling.py), which moves 1.64 GB and is
// HBM-bound at ~0.30 ms.  Here
//   * Y[M x N] = X[M x K] W[N x K]^T with K in steps of 32 channels.  A workgroup works on ITEMS = (pair of row tiles,
//     128-column filter tile); compute group g (waves 4g .. 4g+3, 2 x 2 waves) owns row tile g of the pair, (32 FM) x 128
//     outputs, FM = 4..7 row fragments per wave; both groups share the filter slice of a step.
//   * time is cut into ticks, one workgroup barrier each: group 0 reads the fragments of step s in tick 2s and issues its
//     MFMAs in tick 2s+1, group 1 one tick later -- on every SIMD one wave feeds the matrix pipe while the other reads LDS.
//   * loader waves 8-11 issue every LDS-DMA: per step 2 x 2FM pieces of x (16 rows x 64 B each) and 8 pieces of the
//     filter slice, THREE steps ahead (ring of four slots), retired by counted vmcnt waits before the even ticks.  The
//     stream of steps runs across item boundaries, so the next item's first steps land while the accumulators are stored.
//   * the items of an XCD are ordered column-tile fastest and dealt round-robin to its 32 workgroups: the workgroups that
//     share a row tile run at the same time and the tile's rows come from HBM once and from that XCD's L2 afterwards.
//   * swapped MFMA operands (filter rows = MFMA rows): a lane ends with 2 x 8 consecutive output channels of one pixel
//     and stores 16-byte segments straight from the accumulators; epilogues as in span6 (plain + residual, BatchNorm
//     statistics in fixed point, affine + ReLU + residual).
// LDS images and swizzles are span6's.  The summation order over K is the channel order in steps of 32, as in every conv
// kernel of this library; results agree with vt_igemm.hip to the last bf16 rounding of the output, not bit for bit.
#include <stdlib.h>

#include <type_traits>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

constexpr int kFMX = 7;         // row fragments (16 rows) per compute wave, at most
constexpr int kD = 4;           // ring slots (steps): three steps in flight
constexpr int kBSlot = 128 * 64;  // bytes per filter slice: 128 filter rows x 32 channels

__device__ __attribute__((aligned(16))) unsigned int vt_gemm6_zero16[4];  // source of every row past M

struct G6Args {
    IgemmArgs p;
    int fm;      // row fragments per wave of this launch (4..kFMX): a group's tile has 32*fm rows
    int pairs;   // row-tile pairs: ceil(M / (64*fm))
    int ppx;     // pairs per XCD
    int nsteps;  // Cin / 32
};

__device__ __forceinline__ int swz4(int g) { return (0x1320 >> ((g & 3) * 4)) & 3; }  // filter-slice image
__device__ __forceinline__ int swzA(int g) { return (g & 1) << 1; }                   // row image

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long)hi << 32) | lo);
}
// LDS-DMA, 16 B per lane: LDS address = M0 + lane*16, global address = sbase + voff (or the per-lane address).
// (s_nop 4: a VMEM instruction that reads an SGPR written by a VALU instruction -- the v_readfirstlane above -- needs 5
//  wait states, which hipcc does not insert in front of an asm statement)
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
// counted wait with a run-time, wave-uniform count (vmcnt takes an immediate); at most 2 steps x 10 instructions here
__device__ __forceinline__ void vm_wait_dyn(int n) {
#define VT_W4(b)                                  \
    switch (n - (b)) {                            \
        case 0: vmw<(b) + 0>(); break;            \
        case 1: vmw<(b) + 1>(); break;            \
        case 2: vmw<(b) + 2>(); break;            \
        default: vmw<(b) + 3>(); break;           \
    }
    if (n <= 0) { vmw<0>(); return; }
    if (n >= 24) { vmw<24>(); return; }
    if (n < 16) {
        if (n < 8) { if (n < 4) { VT_W4(0) } else { VT_W4(4) } }
        else { if (n < 12) { VT_W4(8) } else { VT_W4(12) } }
    } else {
        if (n < 20) { VT_W4(16) } else { VT_W4(20) }
    }
#undef VT_W4
}

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

typedef const __attribute__((address_space(4))) G6Args* ArgsPtr;
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

// 12 waves: 0-3 compute group 0, 4-7 compute group 1, 8-11 loaders; three per SIMD = at most 168 registers
template <int MODE>  // epilogue: 0 plain (+ residual), 1 BatchNorm statistics, 2 affine (+ ReLU, + residual)
__global__ void __launch_bounds__(768, 3) gemm6_kernel(const G6Args a) {
    const IgemmArgs& p = a.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS (bytes): [filter ring kD x 8 KiB][group 0: kD x 2*fm KiB of rows][group 1: the same]
    const int xslot = a.fm * 2048;
    const char* sBb = smem;
    const char* sXb = smem + kD * kBSlot;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's items: every 32nd (row-tile pair, column tile) of its XCD, column tile fastest ----
    const int bid = blockIdx.x, xcd = bid & 7, wq = bid >> 3;
    const int tiles_n = p.tiles_n;
    const int pr0 = xcd * a.ppx;
    const int npx = max(0, min(a.pairs, pr0 + a.ppx) - pr0);
    const int items_x = npx * tiles_n;
    if (wq >= items_x) return;
    const int nit = __builtin_amdgcn_readfirstlane((items_x - wq + 31) >> 5);
    const int nsteps = __builtin_amdgcn_readfirstlane(a.nsteps);
    const int G = nit * nsteps;  // steps of this workgroup, all items
    const int tile_rows = 32 * a.fm;

    if (wave >= 8) {
        // =========================== loader waves ==================================================
        const int lj = wave - 8;  // 0..3
        const char* xg = (const char*)p.x;
        const char* wg = (const char*)p.w;
        const unsigned b_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem);
        const unsigned x_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(smem + kD * kBSlot);
        const unsigned m0_keep = get_m0();
        const long ldx2 = (long)p.ldx * 2;
        const int M = p.M;
        const unsigned long zero_src = (unsigned long)(const void*)vt_gemm6_zero16;
        // a row piece = 16 rows x 64 B: lane owns row (lane>>2), source chunk (lane&3) ^ swzA(lane>>4)
        const int cjA = (lane & 3) ^ swzA(lane >> 4);
        // this loader's pieces of a group tile: T = lj, lj+4, lj+8, lj+12 below 2*fm
        const int npl = __builtin_amdgcn_readfirstlane((2 * a.fm - lj + 3) >> 2);
        const int nps = 2 * npl + 2;  // LDS-DMA instructions of this wave per step

        // the item the next issued step belongs to, and its per-lane sources
        int li = 0, ls = 0;  // item index (of this workgroup), step inside it
        unsigned long xa[2][4];
        unsigned b_voff[2];
        const char* wbase = wg;
        auto item_sources = [&](int it) {
            const int idx = wq + 32 * it;
            const int pair = pr0 + idx / tiles_n;
            const int tn = idx - (idx / tiles_n) * tiles_n;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long m = ((long)pair * 2 + g) * tile_rows + (lj + 4 * t) * 16 + (lane >> 2);
                    xa[g][t] = m < M ? (unsigned long)xg + (unsigned long)(m * ldx2 + cjA * 16) : 0ul;
                }
            // filter slice = 8 pieces of 16 rows, this loader's are q = 2*lj, 2*lj+1; row n = 16q + (lane>>2); the fragment
            // reads address row n with chunk position kq ^ swz4(n>>3), so the source chunk is (lane&3) ^ swz4(2q + (lane>>5)).
            // Rows past Cout (N tail) are clamped: their outputs are never stored.
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = 2 * lj + i;
                const int n = min(tn * 128 + 16 * q + (lane >> 2), p.Cout - 1);
                const int cj = (lane & 3) ^ swz4(2 * q + (lane >> 5));
                b_voff[i] = (unsigned)(((long)n * p.ldw + cj * 8) * 2);
            }
        };
        auto issue_x = [&](int slot, int s) {
            const unsigned long cb = (unsigned long)s * 64;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < npl) {
                        set_m0(x_base + (unsigned)((g * kD + slot) * xslot + (lj + 4 * t) * 1024));
                        glds_v(xa[g][t] ? xa[g][t] + cb : zero_src);
                    }
        };
        auto issue_w = [&](int slot, int s) {
            const char* sb = wbase + (long)s * 64;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                set_m0(b_base + (unsigned)(slot * kBSlot + (2 * lj + i) * 1024));
                glds_s(b_voff[i], sb);
            }
        };
        auto advance = [&]() {  // the next step to issue
            if (++ls == nsteps) {
                ls = 0;
                ++li;
                if (li < nit) item_sources(li);
            }
        };
        item_sources(0);
        // ---- prologue: steps 0..2 ----
#pragma unroll
        for (int k = 0; k < kD - 1; ++k)
            if (k < G) {
                issue_x(k, ls);
                issue_w(k, ls);
                advance();
            }
        int slot = (kD - 1) & (kD - 1);  // slot of step gs + 3
        for (int gs = 0; gs < G; ++gs) {
            // step gs complete: at most the two younger steps stay in flight
            vm_wait_dyn(min(kD - 2, G - 1 - gs) * nps);
            wg_barrier();  // tick 2gs: group 0 reads step gs (group 1: MFMAs of gs-1, its last read of filter slot gs-1)
            const bool more = gs + kD - 1 < G;
            // both groups have left the rows of step gs-1 (read in ticks 2gs-2 and 2gs-1): their slot takes step gs+3
            if (more) issue_x(slot, ls);
            wg_barrier();  // tick 2gs+1: group 1 reads step gs
            if (more) {
                issue_w(slot, ls);
                advance();
            }
            slot = (slot + 1) & (kD - 1);
        }
        wg_barrier();  // tick 2G: group 1's last MFMA tick
        vmw<0>();
        set_m0(m0_keep);
        return;
    }

    // =============================== compute waves ==================================================
    const int grp = wave >> 2;  // 0: reads in even ticks, MFMAs in odd ticks; 1: one tick later
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int q4 = lane >> 4, c16 = lane & 15;
    // filter fragment j of this lane: MFMA row r = c16 -> slice row n_j = wn*64 + (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3);
    // (n_j >> 3) & 3 = r >> 2 for every j, so the four fragments share one swizzle term and differ by constants
    const int nb0 = wn * 64 + (c16 >> 2) * 8 + (c16 & 3);
    const int b_lane = (nb0 * 4 + (q4 ^ swz4(c16 >> 2))) * 16;  // byte offset inside a slice; j adds {0,256,2048,2304}
    const char* sXg = sXb + grp * kD * xslot;

    if (grp == 1) wg_barrier();  // tick 0: group 0 reads its first step

    auto run = [&](auto FMc) {
        constexpr int FM = decltype(FMc)::value;
        int wrow = wm * 16 * FM + c16;  // this lane's row inside the tile, fragment 0
        asm volatile("" : "+v"(wrow));
        const unsigned a_lane = (unsigned)((wrow * 4 + (q4 ^ swzA(wrow >> 2))) * 16);
        int slot = 0;
        for (int it = 0; it < nit; ++it) {
            f32x4 acc[FM][4];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < nsteps; ++s) {
                slot = __builtin_amdgcn_readfirstlane(slot);
                // ---- read tick: this step's slice and rows are in LDS
                wg_barrier();
                const char* A = sXg + slot * xslot + a_lane;
                const char* Bt = sBb + ((slot << 13) + b_lane);
                // three filter fragments are read in the read tick, the fourth during the MFMA tick into the first one's
                // registers (168 registers per lane): its latency hides behind 2*FM MFMAs
                uint4 af[FM], bf0, bf1, bf2;
                bf0 = *(const uint4*)(Bt);
                bf1 = *(const uint4*)(Bt + 256);
                bf2 = *(const uint4*)(Bt + 2048);
#pragma unroll
                for (int i = 0; i < FM; ++i) af[i] = *(const uint4*)(A + i * 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers
                // ---- MFMA tick (the other group reads meanwhile)
                wg_barrier();
// the MFMA as an asm statement whose accumulator is an in/out operand: result and addend share their registers by
// construction (left to itself the compiler renames the accumulators from one unrolled step to the next and spills them)
#define VT_MMA_COL(bfrag, j)                                                                             \
    _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                       \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                          \
                     : "+v"(acc[i][j])                                                                  \
                     : "v"(__builtin_bit_cast(bf16x8, bfrag)), "v"(__builtin_bit_cast(bf16x8, af[i])))
                VT_MMA_COL(bf0, 0);
                __builtin_amdgcn_sched_barrier(0);
                bf0 = *(const uint4*)(Bt + 2304);
                VT_MMA_COL(bf1, 1);
                VT_MMA_COL(bf2, 2);
                __builtin_amdgcn_sched_barrier(0);
                VT_MMA_COL(bf0, 3);
#undef VT_MMA_COL
                slot = (slot + 1) & (kD - 1);
            }

            // ---- epilogue: two 16-byte stores per row fragment, straight from the accumulators ----------
            ArgsPtr Q = fresh_args();
            constexpr bool affine = MODE == 2, stats = MODE == 1;
            const bool relu = MODE == 2 && (Q->p.flags & VT_CONV_RELU);
            const bool has_res = MODE != 1 && (Q->p.flags & VT_CONV_RESIDUAL) != 0;
            const int Cout_ = Q->p.Cout, ldy_ = Q->p.ldy, ldr_ = Q->p.ldr, M_ = Q->p.M;
            bf16_t* __restrict__ yg = (bf16_t*)Q->p.y;
            const bf16_t* __restrict__ rg = (const bf16_t*)Q->p.res;
            const float* scale_ = Q->p.scale;
            const float* shift_ = Q->p.shift;
            float* stats_ = Q->p.stats;
            const int idx = wq + 32 * it;
            const int pair = pr0 + idx / tiles_n;
            const int tn = idx - (idx / tiles_n) * tiles_n;
            const long m0 = ((long)pair * 2 + grp) * (32 * FM);
            const int rep = (int)((m0 / 32) % kStatReplicas);
            // this lane's output channels: ch(h, e8) = tn*128 + wn*64 + h*32 + q4*8 + e8, h = 0,1, e8 = 0..7
            const int ch0 = tn * 128 + wn * 64 + q4 * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int n = ch0 + h * 32;
                asm volatile("" : "+v"(n));  // (opaque: per-lane 64-bit addresses are not hoisted out of the item loop)
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
                // the residual rows of this half, all in flight before the first one is used (rows / channels outside the
                // tensor read its first 16 bytes), so that nothing orders them behind the stores below
                constexpr bool kPreRes = MODE != 1;
                uint4 rres[kPreRes ? FM : 1];
                if (kPreRes && has_res) {
#pragma unroll
                    for (int i = 0; i < (kPreRes ? FM : 0); ++i) {
                        const long m = m0 + wrow + i * 16;
                        const bool ok = m < M_ && n < Cout_;
                        rres[i] = *(const uint4*)(rg + (ok ? m * ldr_ + n : 0l));
                    }
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const long m = m0 + wrow + i * 16;
                    const bool row_ok = m < M_;
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
                        *(uint4*)(yg + (m * ldy_ + n)) = out;
                    }
                }
                if (stats) {
                    // sum over the 16 pixel lanes (same q4 = one DPP row), then lanes c16 = 0..7 keep channel e = c16
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
        }
    };
    switch (a.fm) {
        case 4: run(I_<4>{}); break;
        case 5: run(I_<5>{}); break;
        case 6: run(I_<6>{}); break;
        default: run(I_<kFMX>{}); break;
    }
    if (grp == 0) wg_barrier();  // tick 2G: group 1's last MFMA tick
}

}  // namespace

// returns -1 when this kernel does not apply
int vt_gemm6_dispatch(IgemmArgs& a0, int dtype, void* stream) {
    // VT_GEMM6=0 disables, =2 forces this kernel wherever it applies (tests); default: the big-K layers it measured faster
    // on than the kernels behind it in vt_conv_igemm's chain
    const int enabled = VT_KNOB("VT_GEMM6", 1);
    if (!enabled || dtype != VT_BF16) return -1;
    if (vt_device_cus() != 256) return -1;  // (the item map is built for 8 XCDs x 32 CUs with 160 KiB of LDS each)
    if (a0.ntaps != 1 || a0.sh != 1 || a0.sw != 1 || a0.Ho != a0.Hi || a0.Wo != a0.Wi) return -1;
    if (a0.h0 + a0.dh[0] != 0 || a0.w0 + a0.dw[0] != 0 || !a0.dense_out) return -1;
    if (a0.flags & (VT_CONV_D2S | VT_CONV_NOSTORE)) return -1;
    if (a0.Cin % 32 != 0 || a0.Cin < 64 || a0.Cout < 64 || a0.Cout % 8 != 0) return -1;
    if (enabled < 2 && (a0.Cin < 256 || (long)a0.M * a0.Cout < 4L * 1024 * 1024)) return -1;
    if ((unsigned long)a0.Cout * a0.ldw * 2 >= 0xffff0000ul) return -1;
    G6Args a;
    a.p = a0;
    IgemmArgs& p = a.p;
    p.tiles_n = (p.Cout + 127) / 128;
    a.nsteps = p.Cin / 32;
    // tile height: the one with the least (rounds x (rows + fixed part)) -- an item costs its K loop, proportional to fm,
    // plus the accumulator stores and the stall around them (~1.5 fragments' worth)
    int best = kFMX;
    double best_cost = 1e30;
    for (int fm = kFMX; fm >= 4; --fm) {
        const long pairs = ((long)p.M + 64 * fm - 1) / (64 * fm);
        const long ppx = (pairs + 7) / 8;
        const long rounds = (ppx * p.tiles_n + 31) / 32;
        const double cost = (double)rounds * (fm + 1.5);
        if (cost < best_cost - 1e-9) best_cost = cost, best = fm;
    }
    const int fmk = VT_KNOB("VT_GEMM6_FM", 0);
    a.fm = (fmk >= 4 && fmk <= kFMX) ? fmk : best;
    a.pairs = (int)(((long)p.M + 64 * a.fm - 1) / (64 * a.fm));
    a.ppx = (a.pairs + 7) / 8;
    const int smem = kD * kBSlot + 2 * kD * a.fm * 2048;
    const int mode = (p.flags & VT_CONV_STATS) ? 1 : ((p.flags & VT_CONV_AFFINE) ? 2 : 0);
    if (mode == 1 && (p.flags & (VT_CONV_AFFINE | VT_CONV_RELU | VT_CONV_RESIDUAL))) return -1;
    if (mode == 0 && (p.flags & VT_CONV_RELU)) return -1;
    auto kern = mode == 1 ? gemm6_kernel<1> : (mode == 2 ? gemm6_kernel<2> : gemm6_kernel<0>);
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(gemm6)");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("gemm6_kernel<bf16,2x4+4 waves,FM%d>", a.fm);
    hipLaunchKernelGGL(kern, dim3(8 * 32), dim3(768), smem, (hipStream_t)stream, a);
    VT_CHECK_LAUNCH("vt_conv_igemm(gemm6)");
    return VT_OK;
}
