// vt_igemm_span.hip -- implicit-GEMM convolution with the input staged ONCE per channel chunk and re-used by every
// filter tap ("input span").
//
// Stride-1 grids (every 3x3/1x1 stride-1 conv of ConvNormAct, reference components.py:26-35; every stride-1 data
// gradient; the parity classes of the stride-2 data gradients), Cin a multiple of the 64-byte K chunk.  Why a second
// kernel: vt_igemm.hip stages the gathered A rows separately for each tap, so a 3x3 conv pushes every input pixel through
// the global->LDS path 9 times; measured, that path (~9 TB/s chip wide for 64-byte segments), not the MFMA pipe or HBM,
// bounds it -- at ~570 TFLOP/s on the 128..512-channel layers and at 2-4x the HBM time on the 32/64-channel layers at
// 112x112.  Here, with the flat pixel index m = (b*H + i)*W + j, tap t reads input pixel m + d_t, d_t = eh_t*W + ew_t, so
// the BM output pixels of a tile need ONE contiguous span of BM + (dmax - dmin) input pixels for all taps.  Per channel
// chunk (32 bf16 / 16 f32):
//   * the span is DMA'd once into a 2-slot LDS ring  (A: span x 64 B),
//   * per tap only the BN x 64 B filter slice is DMA'd (3-slot ring, 2 in flight),
//   * tap t's MFMA A-fragments are read from the span at row offset d_t - dmin; a fragment row whose tap leaves the
//     image (padding) reads a 16-byte zero block instead: the per-lane LDS address is selected from a per-row tap mask
//     built once per tile, so the loop carries 3 VALU ops per fragment and nothing between ds_read and MFMA.
//
// Stride 2 (round 4: the 3x3 stride-2 padding-1 conv that opens every Darknet / CSPDarknet stage, reference
// backbones/darknet.py:35,43, on even maps) as the same kernel over the SPACE-TO-DEPTH view of the input (template
// flag S2).  The four parity planes P_ac[b][i][j] = x[b][2i+a][2j+c] have the output's grid, and on them the conv is a
// stride-1 conv whose taps are offsets in {-1, 0}^2: plane (1,1) carries 4 taps, (1,0) and (0,1) two each, (0,0) one --
// nine steps per channel chunk, the same MFMA work as the gather kernel, but every input pixel enters LDS ONCE per
// filter-column tile instead of 2.25 times (and 9 times with padding rounding on the small-channel layers).  A plane's
// span is gathered straight from the NHWC tensor (row r of the span = plane position m0 + dmin_p + r = one 64-byte
// segment at pixel (2i+a, 2j+c)); each plane has its own LDS slot, reloaded for the next chunk as soon as its last tap
// has been read (five or more steps before its first use), in the order (1,1), (1,0), (0,1), (0,0).
//
// 4 waves as WM x WN; BN=128: 2x2, wave tile 128x64; BN=64/32: 4x1, wave tile 64xBN.  Same LDS-DMA / counted-vmcnt
// discipline, swizzle, statistics and XCD-aware tile map as vt_igemm.hip.  All A fragments of a wave are 16 rows apart,
// so their swizzle term is identical and one address per tap serves all of them.
// Epilogue: each wave stages its own 16-row slabs through a private LDS window and writes 16-byte row segments; the
// statistics of the row-waves are folded in LDS and leave as one fixed-point atomic per column and moment.
#include <stdlib.h>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

constexpr int kTapBytes = 32 * 16;  // ntaps <= 32 on this path

__device__ __attribute__((aligned(16))) unsigned int vt_span_zero16[4];
// chunk ^= 2 * ((row >> 2) & 1): the one 4-entry swizzle family (found by enumeration) under which a ds_read_b128 of
// 16 consecutive 64-byte rows is conflict free for EVERY starting row -- a tap shifts the fragment rows by an
// arbitrary offset; the table 0x1320 used before is conflict free only for offsets that are multiples of 4.
// VT_SPAN_SWZ_TABLE: 0x2020 (this), 0x1320 (the old table, for A/B timing).
#ifndef VT_SPAN_SWZ_TABLE
#define VT_SPAN_SWZ_TABLE 0x2020
#endif
__device__ __forceinline__ int swz(int row) { return (VT_SPAN_SWZ_TABLE >> (((row >> 2) & 3) * 4)) & 3; }

__device__ __forceinline__ void glds16(unsigned long gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_base)
        : "memory");
}

// same, SGPR base + per-lane 32-bit byte offset: no 64-bit VALU address arithmetic per instruction
__device__ __forceinline__ void glds16s(unsigned voff, const void* sbase, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 2\n\t"  // 5 wait states in all before a VMEM instruction may read a VALU-written (e.g. reloaded) SGPR base
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_base)
        : "memory");
}

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// counted wait with a run-time count (wave-uniform): vmcnt takes an immediate
__device__ __forceinline__ void vm_wait_dyn(int n) {
    switch (n) {
        case 0: vm_wait<0>(); break;
        case 1: vm_wait<1>(); break;
        case 2: vm_wait<2>(); break;
        case 3: vm_wait<3>(); break;
        case 4: vm_wait<4>(); break;
        case 5: vm_wait<5>(); break;
        case 6: vm_wait<6>(); break;
        case 7: vm_wait<7>(); break;
        case 8: vm_wait<8>(); break;
        case 9: vm_wait<9>(); break;
        case 10: vm_wait<10>(); break;
        case 11: vm_wait<11>(); break;
        case 12: vm_wait<12>(); break;
        case 13: vm_wait<13>(); break;
        case 14: vm_wait<14>(); break;
        case 15: vm_wait<15>(); break;
        default: vm_wait<16>(); break;
    }
}

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <typename T>
__device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4& acc);
template <>
__device__ __forceinline__ void mma<bf16_t>(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma<float>(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// LDS map (bytes): [taps 512][row masks BM*4][row output pixel BM*4][filter ring 3 x BROWS*64]
//                  [zero 16 .. 64][span slot 0][span slot 1 (only when Cin spans > 1 chunk)]
//                  (S2: four plane slots of their own lengths instead of the two span slots)
template <int BM, int BN, int PD>
struct SpanLds {
    static constexpr int BROWS = BN < 64 ? 64 : BN;
    static constexpr int kMask = kTapBytes;
    static constexpr int kPo = kMask + BM * 4;
    static constexpr int kB = kPo + BM * 4;
    static constexpr int kZero = kB + (PD + 1) * BROWS * 64;
    static constexpr int kA = kZero + 64;
    // ita DMA instructions per wave and chunk: nw waves x 16 rows x 64 B each
    __host__ __device__ static constexpr int bytes(int ita, int nslots, int nw = 4) { return kA + nslots * ita * nw * 16 * 64; }
};

// Stride-2 launches (space-to-depth view): plane class q = 0..3 is plane (a, c) = (1,1), (1,0), (0,1), (0,0); its span
// starts dmin_q = {-Wo-1, -Wo, -1, 0} plane positions before the tile and is npieces[q] pieces of 16 rows long.
struct SpanS2 {
    int slot_off[4];   // byte offset of class q's slot inside the span area
    int npieces[4];    // 16-row pieces of its span: ceil((BM - dmin_q) / 16)
    int plane_off[4];  // (a*W + c) * ldx * sizeof(T): byte offset of the plane's pixel inside the 2x2 block
    int delta[4];      // -dmin_q as a byte distance in x for a row that does not wrap: {2W+2, 2W, 2, 0} pixels
    int wrap;          // extra byte distance when the step back by one plane column leaves the row (j = 0): W pixels
    int maxoff;        // byte offset in x of the pixel under the last plane position (B-1, Ho-1, Wo-1)
    int8_t t_q[12], t_rs[12];  // step t of a chunk: plane class and filter tap (3*r + s)
    short t_drow[12];          // and the row offset of its fragments inside the class's span
};
constexpr int kS2MaxP = 6;  // span pieces per wave, at most (4 waves: spans of up to 384 rows)

// ita: span DMA instructions per wave per chunk (span = 64*ita rows >= BM + dmax - dmin)
template <typename T, int BM, int BN, int WM, int WN, int PD, bool S2 = false>
__global__ void __launch_bounds__(64 * WM * WN, 2) span_kernel(const IgemmArgs p, const int dmin, const int ita, const SpanS2 g) {
    constexpr int NW = WM * WN;  // waves: 4, or 8 for the 256 x 128 tile (wave tile 64 x 64, 16 waves per CU)
    constexpr int NT = 64 * NW;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int CH = 4 * EPC;  // channels per chunk (64-byte rows)
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    using L = SpanLds<BM, BN, PD>;
    constexpr int BROWS = L::BROWS;
    constexpr int ITB = BROWS / (16 * NW);  // filter DMA instructions per wave per step
    constexpr int BSLOT = BROWS * 4;
    constexpr int NSB = PD + 1;
    static_assert((NW == 4 || NW == 8) && ITB >= 1 && TM % 16 == 0 && TN % 16 == 0, "tile shape");
    static_assert(L::kZero >= FM * 1024, "zero block must sit above the fragment offsets");
    static_assert(!S2 || (sizeof(T) == 2 && PD == 2), "the stride-2 view is a bf16 path");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* sTap = (int4*)smem;  // x: row offset in the span, y: filter tap index, z/w: eh, ew  (S2: z = slot offset in uint4)
    unsigned* sMask = (unsigned*)(smem + L::kMask);
    int* sPo = (int*)(smem + L::kPo);
    uint4* sB = (uint4*)(smem + L::kB);  // [NSB][BSLOT]
    uint4* sZ = (uint4*)(smem + L::kZero);
    uint4* sA = (uint4*)(smem + L::kA);  // [1 or 2][aslot]
    const int aslot = ita * NW * 64;     // uint4 per span slot

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int tn = slot % p.tiles_n;
    const int ml = slot / p.tiles_n;
    const int tm = xcd * p.chunk + ml;
    if (ml >= p.chunk || tm >= p.tiles_m) return;

    const int W = p.Wi, H = p.Hi, HW = H * W;
    const long m0 = (long)tm * BM;
    if (tid < p.ntaps) {
        if constexpr (S2) {
            sTap[tid] = make_int4(g.t_drow[tid], g.t_rs[tid], g.slot_off[g.t_q[tid]] >> 4, 0);
        } else {
            const int eh = p.h0 + p.dh[tid], ew = p.w0 + p.dw[tid];
            sTap[tid] = make_int4(eh * W + ew - dmin, tid, eh, ew);
        }
    }
    if (tid < 4) ((unsigned*)sZ)[tid] = 0u;
    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ wg = (const T*)p.w;
    const unsigned long zero_src = (unsigned long)(const void*)vt_span_zero16;
    const unsigned a_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sA;
    const unsigned b_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sB;

    // ---- DMA geometry ---------------------------------------------------------------------
    // an instruction fills 16 rows x 64 B; lane l owns row 16j + (l>>2), chunk (l&3)^swz(row)
    const int cj = (lane & 3) ^ ((VT_SPAN_SWZ_TABLE >> (((lane >> 4) & 3) * 4)) & 3);
    // span row r = 16*(wave + NW*i) + (lane>>2) holds input pixel m0 + dmin + r (zero page outside)
    const long pix0 = m0 + dmin + 16 * wave + (lane >> 2);
    const unsigned long a_src0 = (unsigned long)(xg + (pix0 * p.ldx + cj * EPC));
    const unsigned long a_istep = 16ul * NW * (unsigned long)p.ldx * sizeof(T);
    unsigned long bbase[ITB];
    bool bvalid[ITB];
#pragma unroll
    for (int i = 0; i < ITB; ++i) {
        const int n = tn * BN + 16 * (wave + NW * i) + (lane >> 2);
        bvalid[i] = n < p.Cout && 16 * (wave + NW * i) < BN;
        bbase[i] = (unsigned long)(wg + ((long)(bvalid[i] ? n : 0) * p.ldw + cj * EPC));
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = p.Cin / CH;
    const int nsteps = nchunks * p.ntaps;
    const int b_lane = (lane & 15) * 4 + ((lane >> 4) ^ swz(lane & 15));

    // Fast DMA addressing (interior tiles: every span row inside [0, M), every filter row < Cout, all
    // byte offsets < 4 GiB): scalar base + constant per-lane 32-bit offset, no VALU per instruction.
    const bool a_fast = !S2 && __all(pix0 >= 0 && pix0 + 16l * NW * (ita - 1) < p.M) && (p.fast_dma & 1) &&
                        (unsigned long)p.M * p.ldx * sizeof(T) < 0xffff0000ul;
    bool bv_all = true;
#pragma unroll
    for (int i = 0; i < ITB; ++i) bv_all = bv_all && bvalid[i];
    const bool b_fast = __all(bv_all) && (p.fast_dma & 1);
    const char* a_sbase = (const char*)xg + (m0 + dmin) * (long)p.ldx * (long)sizeof(T);  // may precede x: unused then
    const unsigned a_voff0 = (unsigned)((16 * wave + (lane >> 2)) * p.ldx + cj * EPC) * (unsigned)sizeof(T);
    const unsigned a_vstep = (unsigned)(16 * NW * p.ldx) * (unsigned)sizeof(T);
    unsigned b_voff[ITB];
#pragma unroll
    for (int i = 0; i < ITB; ++i)
        b_voff[i] = (unsigned)((tn * BN + 16 * (wave + NW * i) + (lane >> 2)) * p.ldw + cj * EPC) * (unsigned)sizeof(T);

    // S2: byte offset in x of the pixel (b, 2i, 2j) under plane position v0 = m0 + r of this lane's row r of piece
    // wave + NW*P, plus its chunk position; bit P of s2_wrap: that position is the first of its row (j = 0), so one column back is the last column
    // of the row above.  A plane class's source is this minus the class's (wave-uniform) distance.
    int s2_off[S2 ? kS2MaxP : 1];
    unsigned s2_wrap = 0;
    if constexpr (S2) {
        const int Wo = p.Wo;
#pragma unroll
        for (int P = 0; P < kS2MaxP; ++P) {
            const long v0 = m0 + 16 * (wave + NW * P) + (lane >> 2);  // (may pass the last position: its class position need not)
            const int q = (int)(v0 / Wo);
            const int j0 = (int)(v0 - (long)q * Wo);
            s2_off[P] = (int)((4 * v0 - 2 * j0) * (long)p.ldx * (long)sizeof(T)) + cj * 16;
            s2_wrap |= (j0 == 0 ? 1u : 0u) << P;
        }
    }

    // span of channel chunk `ic` into slot `sl`
#define VT_ISSUE_A(sl, ic)                                                                   \
    do {                                                                                     \
        const unsigned long cofs = (unsigned long)(ic) * (CH * sizeof(T));                   \
        if (a_fast) {                                                                        \
            const char* sb = a_sbase + cofs;                                                 \
            for (int i = 0; i < ita; ++i)                                                    \
                glds16s(a_voff0 + i * a_vstep, sb, a_base + (unsigned)(((sl)*aslot + (wave + NW * i) * 64) * 16)); \
            break;                                                                           \
        }                                                                                    \
        for (int i = 0; i < ita; ++i) {                                                      \
            const long pix = pix0 + 16l * NW * i;                                            \
            const unsigned long src = (pix >= 0 && pix < p.M) ? a_src0 + i * a_istep + cofs : zero_src; \
            glds16(src, a_base + (unsigned)(((sl)*aslot + (wave + NW * i) * 64) * 16));      \
        }                                                                                    \
    } while (0)
    // S2: span of plane class q, channel chunk ic, into the class's own slot; returns the instructions issued.  A row
    // before the first image (reachable through padding only: its fragments are masked) or past the last position
    // (feeding discarded outputs only) is clamped into the tensor.
    auto issue_plane = [&](int q, int ic) -> int {
        const char* sb = (const char*)xg + ((long)g.plane_off[q] + (long)ic * (CH * (long)sizeof(T)));
        const int dq = g.delta[q], np = g.npieces[q];
        const bool back1 = (q == 0 || q == 2);  // classes whose span starts one plane column back
        const unsigned lds0 = a_base + (unsigned)g.slot_off[q];
        int n = 0;
#pragma unroll
        for (int P = 0; P < kS2MaxP; ++P) {
            if (wave + NW * P < np) {
                int off = s2_off[P] - dq - ((back1 && ((s2_wrap >> P) & 1u)) ? g.wrap : 0);
                off = min(max(off, 0), g.maxoff + 48);  // (+48: the last position keeps its chunk position)
                glds16s((unsigned)off, sb, lds0 + (unsigned)((wave + NW * P) * 1024));
                ++n;
            }
        }
        return n;
    };
    // filter slice of step (chunk ic, tap it): rows n, K offset it*Cin + ic*CH
#define VT_ISSUE_B(bslot, ic, it)                                                          \
    do {                                                                                   \
        const long koff = ((long)(it)*p.Cin + (long)(ic)*CH) * (long)sizeof(T);            \
        if (b_fast) {                                                                      \
            const char* sb = (const char*)wg + koff;                                       \
            _Pragma("unroll") for (int i = 0; i < ITB; ++i)                                \
                glds16s(b_voff[i], sb, b_base + (unsigned)(((bslot)*BSLOT + (wave + NW * i) * 64) * 16)); \
            break;                                                                         \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < ITB; ++i) {                                  \
            const unsigned long ps = bvalid[i] ? bbase[i] + koff : zero_src;               \
            glds16(ps, b_base + (unsigned)(((bslot)*BSLOT + (wave + NW * i) * 64) * 16));  \
        }                                                                                  \
    } while (0)
    // (S2: step `it` of a chunk multiplies filter tap t_rs[it])
#define VT_TAP_OF(it) (S2 ? (int)g.t_rs[it] : (it))

    // prologue: span of chunk 0 (S2: all four planes of chunk 0), filter slices of steps 0 and 1
    if constexpr (S2) {
        for (int q = 0; q < 4; ++q) (void)issue_plane(q, 0);
    } else {
        VT_ISSUE_A(0, 0);
    }
    int ic_n = 0, it_n = 0;  // (chunk, tap) of the next filter slice to issue
#pragma unroll
    for (int s = 0; s < PD; ++s) {
        if (s < nsteps) {
            VT_ISSUE_B(s, ic_n, VT_TAP_OF(it_n));
            if (++it_n == p.ntaps) it_n = 0, ++ic_n;
        }
    }
    // per output row of the tile: which taps stay inside the image, and where the row goes
    for (int r = tid; r < BM; r += NT) {
        const long m = m0 + r;
        unsigned bits = 0;
        int po = 0;
        if (m < p.M) {
            if constexpr (S2) {
                const int HoWo = p.Ho * p.Wo;
                const int b = (int)(m / HoWo);
                const int rem = (int)(m - (long)b * HoWo);
                const int oi = rem / p.Wo, oj = rem - oi * p.Wo;
                for (int t = 0; t < p.ntaps; ++t) {  // bit = filter tap index
                    const int ih = oi * 2 + p.h0 + p.dh[t], iw = oj * 2 + p.w0 + p.dw[t];
                    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) bits |= 1u << t;
                }
                po = p.dense_out ? (int)m : (b * p.oH + (oi * p.oHs + p.oh0)) * p.oW + (oj * p.oWs + p.ow0);
            } else {
                const int b = (int)(m / HW);
                const int rem = (int)(m - (long)b * HW);
                const int oi = rem / W, oj = rem - oi * W;
                for (int t = 0; t < p.ntaps; ++t) {
                    const int eh = p.h0 + p.dh[t], ew = p.w0 + p.dw[t];
                    if ((unsigned)(oi + eh) < (unsigned)H && (unsigned)(oj + ew) < (unsigned)W) bits |= 1u << t;
                }
                po = p.dense_out ? (int)m : (b * p.oH + (oi * p.oHs + p.oh0)) * p.oW + (oj * p.oWs + p.ow0);
            }
        }
        sMask[r] = bits;
        sPo[r] = po;
    }
    __syncthreads();  // tap table, row masks, zero block

    // Static priority split (fast_dma bits 1..2 = mode): two workgroups share a CU, one wave of each per SIMD.  At
    // equal priority the matrix pipe is shared evenly, which locks the two waves IN phase (both in their MFMA
    // block together, both in their scalar / DMA / barrier block together) and the phases add up instead of
    // overlapping.  Raising the wave in the odd hardware slot makes its MFMA block pre-empt the other wave's, so
    // the pair settles half a step apart (guide: 'static priority for the younger half', no per-segment flips).
    if (p.fast_dma & 6) {
        const unsigned hw = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID.WAVE_ID
        if (hw & 1u) {
            if ((p.fast_dma & 6) == 2) __builtin_amdgcn_s_setprio(1);
            else if ((p.fast_dma & 6) == 4) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
    }
    unsigned fmask[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) fmask[i] = sMask[wm * TM + i * 16 + (lane & 15)];

    int ic = 0, it = 0;  // (chunk, tap) of the step being computed
    int bcur = 0, bnxt = PD % NSB;
    int a_age = 0;  // steps since the last span was issued (0: none yet)
    if constexpr (S2) {
        // ---- stride 2: nine steps per chunk over the four plane slots -------------------------------------------------
        // Step it of a chunk reads class t_q[it] (steps 0-3: (1,1); 4,5: (1,0); 6,7: (0,1); 8: (0,0)).  A class's slot is
        // free once its last step has been left by every wave, i.e. behind the barrier of the step after it: the next
        // chunk's (1,1) is issued in step 4, (1,0) in step 6, (0,1) in step 8 and (0,0) in step 0 of the next chunk --
        // each at least five steps before its first reader, and older than that reader's filter slice (issued PD = 2
        // steps ahead), whose counted wait therefore covers it.  Younger than slice s at the top of step s: the slice
        // of step s+1 and whatever plane pieces step s-1 issued in front of it.
        int prev_pieces = 0;
        for (int s = 0; s < nsteps; ++s) {
            const int nb = min(PD - 1, nsteps - 1 - s);
            if (prev_pieces == 0 && nb == 1) vm_wait<ITB>();
            else vm_wait_dyn(nb * ITB + prev_pieces);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            prev_pieces = 0;
            if (it == 0) {
                if (ic > 0) prev_pieces = issue_plane(3, ic);
            } else if (it >= 4 && !(it & 1) && ic + 1 < nchunks) {
                prev_pieces = issue_plane((it - 4) >> 1, ic + 1);
            }
            if (s + PD < nsteps) {
                VT_ISSUE_B(bnxt, ic_n, VT_TAP_OF(it_n));
                if (++it_n == p.ntaps) it_n = 0, ++ic_n;
            }
            {
                const int4 tp = sTap[it];
                const int d = __builtin_amdgcn_readfirstlane(tp.x);
                const int rs = __builtin_amdgcn_readfirstlane(tp.y);
                const int so = __builtin_amdgcn_readfirstlane(tp.z);
                const int srow0 = wm * TM + (lane & 15) + d;
                const uint4* A = sA + so + srow0 * 4 + ((lane >> 4) ^ swz(srow0));
                const uint4* Bt = sB + bcur * BSLOT + wn * TN * 4 + b_lane;
                uint4 af[FM], bf[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[j] = Bt[j * 64];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const uint4* src = ((fmask[i] >> rs) & 1u) ? A : sZ - i * 64;
                    af[i] = src[i * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#if VT_MFMA_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) mma<T>(af[i], bf[j], acc[i][j]);
#if VT_MFMA_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
            if (++it == p.ntaps) it = 0, ++ic;
            bcur = (bcur + 1 == NSB) ? 0 : bcur + 1;
            bnxt = (bnxt + 1 == NSB) ? 0 : bnxt + 1;
        }
    } else
    for (int s = 0; s < nsteps; ++s) {
        // Retire this step's filter slice B(s).  VM operations retire in issue order, so everything
        // older is complete as well.  Younger than B(s): the slices of the next nb steps and, if it
        // was issued fewer than PD steps ago, the next chunk's span A' (issued just before the
        // slice of its step).  A' is needed at the first tap of its chunk; if that is NOW (few taps),
        // only the slices issued after A' may stay in flight.
        const int nb = min(PD - 1, nsteps - 1 - s);
        if (a_age >= 1 && a_age <= PD - 1) {  // once or twice per chunk: a span is in the window
            int allowed = nb * ITB;
            if (it == 0 && ic > 0 && a_age == p.ntaps)
                allowed = min(a_age, nb) * ITB;
            else
                allowed += ita;
            vm_wait_dyn(allowed);
        } else if (nb >= 2) {  // the common steps: two or three compile-time counts, no decision tree
            vm_wait<2 * ITB>();
        } else if (nb == 1) {
            vm_wait<ITB>();
        } else {
            vm_wait<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // issue: next chunk's span at the first tap of a chunk, then the slice of step s+PD
        a_age = a_age ? a_age + 1 : 0;
        if (it == 0 && ic + 1 < nchunks) {
            VT_ISSUE_A((ic + 1) & 1, ic + 1);
            a_age = 1;
        }
        if (s + PD < nsteps) {
            VT_ISSUE_B(bnxt, ic_n, it_n);
            if (++it_n == p.ntaps) it_n = 0, ++ic_n;
        }

        {
            const int d = __builtin_amdgcn_readfirstlane(sTap[it].x);
            const int srow0 = wm * TM + (lane & 15) + d;
            const uint4* A = sA + (ic & 1) * aslot + srow0 * 4 + ((lane >> 4) ^ swz(srow0));
            const uint4* Bt = sB + bcur * BSLOT + wn * TN * 4 + b_lane;
            uint4 af[FM], bf[FN];
            // filter fragments first: every MFMA of the first A row needs them
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = Bt[j * 64];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                // (sZ - i*64)[i*64] == sZ[0]: the constant stays in the instruction's offset field
                const uint4* src = ((fmask[i] >> it) & 1u) ? A : sZ - i * 64;
                af[i] = src[i * 64];
            }
            // all fragment reads are issued before the first MFMA (the scheduler otherwise funnels
            // the A fragments through one register quad: read, wait lgkmcnt(0), 4 MFMAs, read, ...)
            __builtin_amdgcn_sched_barrier(0);
#if VT_MFMA_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma<T>(af[i], bf[j], acc[i][j]);
#if VT_MFMA_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        if (++it == p.ntaps) it = 0, ++ic;
        bcur = (bcur + 1 == NSB) ? 0 : bcur + 1;
        bnxt = (bnxt + 1 == NSB) ? 0 : bnxt + 1;
    }
#undef VT_ISSUE_A
#undef VT_ISSUE_B
#undef VT_TAP_OF
    // every wave is done with the rings; they become the staging windows
    __syncthreads();

    // ---- epilogue: per wave, 16-row slabs through a private LDS window --------------------------
    constexpr int PITCH = TN + EPC;          // elements; +16 B keeps the 16-byte reads aligned
    constexpr int CPRW = TN / EPC;           // 16-byte chunks per slab row
    constexpr int RPP = 64 / CPRW;           // slab rows per read pass
    constexpr int NPASS = (16 + RPP - 1) / RPP;
    static_assert(CPRW <= 64, "slab read-out shape");
    static_assert(NW * 16 * PITCH * (int)sizeof(T) <= (PD + 1) * BROWS * 64 + 64 + 2 * 4096,
                  "staging windows exceed the filter ring + the smallest span slot");
    T* sW = (T*)(smem + L::kB) + wave * 16 * PITCH;

    const bool affine = p.flags & VT_CONV_AFFINE;
    const bool relu = p.flags & VT_CONV_RELU;
    const bool stats = p.flags & VT_CONV_STATS;
    const bool has_res = (p.flags & VT_CONV_RESIDUAL) != 0;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    const int q = lane >> 4, c = lane & 15;

    float sc[FN], sf[FN], s1[FN], s2[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = tn * BN + wn * TN + j * 16 + c;
        sc[j] = 1.f, sf[j] = 0.f, s1[j] = 0.f, s2[j] = 0.f;
        if (affine && n < p.Cout) {
            if (p.scale) sc[j] = p.scale[n];
            sf[j] = p.shift[n];
        }
    }
    const int rrow = lane / CPRW, rch = lane % CPRW;
    const int ncol = tn * BN + wn * TN + rch * EPC;
    const long ycol = vt_out_col(p, ncol, p.ldy), rcol = vt_out_col(p, ncol, p.ldr);
    // residual chunks one slab ahead of their use, loaded unconditionally (chunks outside the tensor read its first
    // 16 bytes): nothing orders a load behind the previous slab's stores (the accumulate of the 1x1 data gradients)
    uint4 rcur[NPASS], rnxt[NPASS];
    auto fetch_res = [&](int i, uint4 (&dst)[NPASS]) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int lr = ps * RPP + rrow;
            const int tr = wm * TM + i * 16 + lr;
            const bool ok = lr < 16 && m0 + tr < p.M && ncol < p.Cout;
            dst[ps] = *(const uint4*)(rg + (ok ? (long)sPo[ok ? tr : 0] * p.ldr + rcol : 0l));
        }
    };
    if (has_res) fetch_res(0, rcur);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if (has_res && i + 1 < FM) fetch_res(i + 1, rnxt);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r];
                if (affine) v = fmaf(v, sc[j], sf[j]);
                if (relu) v = fmaxf(v, 0.f);
                const T tv = from_float<T>(v);
                sW[(4 * q + r) * PITCH + j * 16 + c] = tv;
                const float fv = (float)tv;
                s1[j] += fv;
                s2[j] = fmaf(fv, fv, s2[j]);
            }
        }
        lds_fence();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int lr = ps * RPP + rrow;        // row inside the slab
            const int tr = wm * TM + i * 16 + lr;  // row inside the tile
            if (lr < 16) {
                const uint4 raw = *(const uint4*)(sW + lr * PITCH + rch * EPC);
                if (m0 + tr < p.M && ncol < p.Cout) {
                    const long po = sPo[tr];
                    uint4 v = raw;
                    if (has_res) {
                        const uint4 rr = rcur[ps];
                        float fv[EPC], fr[EPC];
                        VecIO<T>::unpack(v, fv);
                        VecIO<T>::unpack(rr, fr);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) fv[e] += fr[e];
                        v = VecIO<T>::pack(fv);
                    }
                    *(uint4*)(yg + (po * p.ldy + ycol)) = v;
                }
            }
        }
        lds_fence();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) rcur[ps] = rnxt[ps];
    }
    if (stats) {  // (uniform per launch: every wave takes the barriers)
        // The WM row-waves of a column fold their partials in LDS in a fixed order and ONE fixed-point atomic leaves
        // the workgroup per (column, moment): the 64-bit atomics queue ~100 ns per address, and the 256 x 64 tiles of
        // the large maps (12.5 k tiles of 4 row-waves at 64 -> 64 @112x112) spent 14 % of their time there.
        static_assert(WM * 2 * BN * 4 <= (PD + 1) * BROWS * 64, "statistics fold fits the filter ring");
        float* sFold = (float*)(smem + L::kB);  // [WM][2][BN], over the staging windows: behind a barrier
        __syncthreads();
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            float a = s1[j], b = s2[j];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            if (q == 0) {
                sFold[(wm * 2 + 0) * BN + wn * TN + j * 16 + c] = a;
                sFold[(wm * 2 + 1) * BN + wn * TN + j * 16 + c] = b;
            }
        }
        __syncthreads();
        const int rep = tm % kStatReplicas;
        for (int i = threadIdx.x; i < 2 * BN; i += NT) {
            const int which = i / BN, col = i % BN;
            const int n = tn * BN + col;
            if (n < p.Cout) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) acc += sFold[(w * 2 + which) * BN + col];
                vt_stat_add(p.stats, ((long)rep * 2 + which) * p.Cout + n, acc);
            }
        }
    }
}

template <typename L>
bool smem_ok(int ita, int nw, int nchunks) { return L::bytes(ita, nchunks > 1 ? 2 : 1, nw) <= 160 * 1024; }

template <typename T, int BM, int BN, int WM, int WN>
int launch_span(IgemmArgs& a, int dmin, int span, hipStream_t st) {
    // two filter slices in flight.  A third measured 3 % slower on the MFMA-bound layers once the issue stream was cleaned
    // up (703 vs 727 TFLOP/s on 128 channels @28x28); EVERY slice of a 9-step tile in flight from the prologue on (PD = 8,
    // round 4, for the short-K HBM-bound layers) measured slower too -- 32 -> 32 3x3 @112x112: 209 -> 253 us, 32 -> 64:
    // 262 -> 299 us: the ring's 36 KB cost a workgroup per CU, and what bounds those layers is the first-access latency
    // of a tile (prologue ~4.7 us of a ~12 us workgroup life), which only more resident workgroups or a persistent,
    // prefetching kernel hide.
    constexpr int PD = 2;
    using L = SpanLds<BM, BN, PD>;
    constexpr int NW = WM * WN;
    const int ita = (span + 16 * NW - 1) / (16 * NW);
    if (ita < 1 || ita * NW > 48 || smem_ok<L>(ita, NW, a.Cin / (64 / (int)sizeof(T))) == false) return -1;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.chunk = (a.tiles_m + 7) / 8;
    const int nchunks = a.Cin / (64 / (int)sizeof(T));
    const int smem = L::bytes(ita, nchunks > 1 ? 2 : 1, NW);
    const long blocks = (long)8 * a.chunk * a.tiles_n;
    auto kern = span_kernel<T, BM, BN, WM, WN, PD, false>;
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(span)");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("span_kernel<%s,%d,%d,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", BM, BN, WM, WN, PD);
    SpanS2 g;
    memset(&g, 0, sizeof(g));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), smem, st, a, dmin, ita, g);
    VT_CHECK_LAUNCH("vt_conv_igemm(span)");
    return VT_OK;
}

template <typename T, int BM>
int launch_span_bn(IgemmArgs& a, int dmin, int span, hipStream_t st) {
    if constexpr (BM == 224) {
        // 7 x 32 rows: 2 x 2 waves of 112 x BN/2 for every width
        if (a.Cout > 64) return launch_span<T, BM, 128, 2, 2>(a, dmin, span, st);
        if (a.Cout > 32) return launch_span<T, BM, 64, 2, 2>(a, dmin, span, st);
        return launch_span<T, BM, 32, 2, 2>(a, dmin, span, st);
    } else {
        // 256 x 128 tile: 4 waves of 128 x 64 (8 waves of 64 x 64 measured equal)
        if (a.Cout > 64) return launch_span<T, BM, 128, 2, 2>(a, dmin, span, st);
        if (a.Cout > 32) return launch_span<T, BM, 64, 4, 1>(a, dmin, span, st);
        return launch_span<T, BM, 32, 4, 1>(a, dmin, span, st);
    }
}

// ---- stride 2 (space-to-depth view) ---------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
int launch_s2(IgemmArgs& a, hipStream_t st, bool dry) {
    constexpr int PD = 2, NW = WM * WN;
    using L = SpanLds<BM, BN, PD>;
    static_assert(NW == 4, "piece distribution: four waves");
    SpanS2 g;
    memset(&g, 0, sizeof(g));
    const int Wo = a.Wo, W = a.Wi;
    const int dmins[4] = {-Wo - 1, -Wo, -1, 0};
    const int pa[4] = {1, 1, 0, 0}, pc[4] = {1, 0, 1, 0};
    const long px = (long)a.ldx * 2;  // bytes per pixel step
    int off = 0;
    for (int q = 0; q < 4; ++q) {
        g.npieces[q] = (BM - dmins[q] + 15) / 16;
        if ((g.npieces[q] + NW - 1) / NW > kS2MaxP) return -1;
        g.slot_off[q] = off;
        off += g.npieces[q] * 1024;
        g.plane_off[q] = (int)((pa[q] * (long)W + pc[q]) * px);
    }
    g.delta[0] = (int)((2L * W + 2) * px), g.delta[1] = (int)(2L * W * px), g.delta[2] = (int)(2 * px), g.delta[3] = 0;
    g.wrap = (int)((long)W * px);
    g.maxoff = (int)((4L * (a.M - 1) - 2L * (Wo - 1)) * px);
    const int smem = L::kA + off;
    if (smem > 160 * 1024) return -1;
    // steps of a chunk: class (1,1) taps (0,0) (0,2) (2,0) (2,2), class (1,0) taps (0,1) (2,1), class (0,1) taps (1,0) (1,2),
    // class (0,0) tap (1,1); tap (r, s) sits (r == 0 ? -1 : 0) plane rows and (s == 0 ? -1 : 0) plane columns from the output
    const int order[9][2] = {{0, 0}, {0, 2}, {2, 0}, {2, 2}, {0, 1}, {2, 1}, {1, 0}, {1, 2}, {1, 1}};
    for (int t = 0; t < 9; ++t) {
        const int r = order[t][0], s_ = order[t][1];
        const int q = ((r & 1) ? 2 : 0) + ((s_ & 1) ? 1 : 0);  // a = !(r & 1), c = !(s & 1)
        const int d = (r == 0 ? -Wo : 0) + (s_ == 0 ? -1 : 0);
        g.t_q[t] = (int8_t)q, g.t_rs[t] = (int8_t)(3 * r + s_), g.t_drow[t] = (short)(d - dmins[q]);
    }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.chunk = (a.tiles_m + 7) / 8;
    const long blocks = (long)8 * a.chunk * a.tiles_n;
    auto kern = span_kernel<bf16_t, BM, BN, WM, WN, PD, true>;
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(span, stride 2)");
        if (rc != VT_OK) return rc;
    }
    if (dry) return VT_OK;
    vt_note_kernel("span_kernel<bf16,%d,%d,%d,%d,%d,s2d>", BM, BN, WM, WN, PD);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), smem, st, a, 0, 0, g);
    VT_CHECK_LAUNCH("vt_conv_igemm(span, stride 2)");
    return VT_OK;
}

// 3x3, stride 2, padding 1 on even maps, bf16, Cin a multiple of 32, the taps in raster order
int span_s2_dispatch(IgemmArgs& a, int dtype, hipStream_t st) {
    if (dtype != VT_BF16 || a.sh != 2 || a.sw != 2 || a.ntaps != 9 || a.h0 != -1 || a.w0 != -1) return -1;
    if (a.flags & (VT_CONV_D2S | VT_CONV_NOSTORE)) return -1;
    if ((a.Hi & 1) || (a.Wi & 1) || a.Ho * 2 != a.Hi || a.Wo * 2 != a.Wi || a.Cin % 32 != 0) return -1;
    for (int t = 0; t < 9; ++t)
        if (a.dh[t] != t / 3 || a.dw[t] != t % 3) return -1;
    if ((long)a.B * a.Hi * a.Wi * a.ldx * 2 >= 0x7fff0000L) return -1;  // (per-lane byte offsets are ints)
    if ((long)a.B * a.oH * a.oW > 0x7fffffffL) return -1;
    // VT_SPAN_S2: 0 (default) off, 2 wherever it applies (tests, measurements).  Measured, round 4, batch 256 (us, this kernel
    // against the gather kernel): 32 -> 64 @224 618 vs 445, 64 -> 128 @112 304 vs 222, 128 -> 256 @56 249 vs 173,
    // 256 -> 512 @28 225 vs 151, 512 -> 1024 @14 235 vs 145.  It stages each input pixel once (48 KB per 128-row tile
    // instead of 108), but a one-tile workgroup here waits for four plane spans before its first step and keeps 62 KB of
    // LDS (two workgroups per CU against the gather kernel's four): on these layers the life of a workgroup is first-access
    // latency, not staged bytes.
    const int mode = VT_KNOB("VT_SPAN_S2", 0);
    if (mode < 2) return -1;
    if (a.Cout > 64) return launch_s2<128, 128, 2, 2>(a, st, false);
    if (a.Cout > 32) return launch_s2<128, 64, 4, 1>(a, st, false);
    return launch_s2<128, 32, 4, 1>(a, st, false);
}

}  // namespace

// returns -1 when the span kernel does not apply (the caller then uses the general kernel)
int vt_span_dispatch(IgemmArgs& a, int dtype, void* stream) {
    const int enabled = VT_KNOB("VT_IGEMM_SPAN", 1);
    if (!enabled) return -1;
    const int fast_dma = (1);
    a.fast_dma = fast_dma & 1;
    const int ch = 4 * vt_epc(dtype);
    if (a.sh == 2 && a.sw == 2) return span_s2_dispatch(a, dtype, (hipStream_t)stream);
    if (a.sh != 1 || a.sw != 1 || a.Ho != a.Hi || a.Wo != a.Wi) return -1;
    if (a.Cin % ch != 0 || a.ntaps > 32) return -1;
    if ((long)a.M + 2L * a.Wi * VT_MAX_TAPS > 0x7fffffffL) return -1;
    if ((long)a.B * a.oH * a.oW > 0x7fffffffL) return -1;
    int dmin = 1 << 30, dmax = -(1 << 30);
    for (int t = 0; t < a.ntaps; ++t) {
        const int d = (a.h0 + a.dh[t]) * a.Wi + (a.w0 + a.dw[t]);
        dmin = d < dmin ? d : dmin;
        dmax = d > dmax ? d : dmax;
    }
    hipStream_t st = (hipStream_t)stream;
    // Plain GEMMs with a long K or many filter columns (1x1 convs with >= 320 channels on either side: the OSA
    // aggregation convs of VoVNet-39, the 14x14 stage of CSPDarknet-53): the gather kernel's 8-wave 256 x 128 tile stages
    // the rows once per 128 columns for 16 waves per CU and measured 7-20 % faster there (768 -> 256 @56x56, B=256:
    // 632 -> 570 us; 1472 -> 768 @14x14: 171 -> 136 us; 256 -> 256 @28x28: equal).
    if (dtype == VT_BF16 && a.ntaps == 1 && (a.Cin >= 320 || a.Cout >= 320) && (VT_KNOB("VT_IGEMM_W8", 3) & 2) &&
        (long)((a.M + 255) / 256) * ((a.Cout + 127) / 128) >= 256 && enabled < 2)
        return -1;
    if (dtype == VT_BF16) {
        // 256-row tiles; maps too small to give every CU a tile (7x7 at batch 256) run the
        // 128-row variant for <= 64 output channels and fall back to the general kernel's
        // 128x128 tiles above that (measured faster there: 81 vs 97 us on 512->512 3x3 @7x7).
        // VT_IGEMM_SPAN=2 forces this kernel wherever it applies, =3 also forces 256-row tiles (tests).
        const long tiles_n = (a.Cout + 127) / 128;
        const long tiles256 = (long)((a.M + 255) / 256) * tiles_n;
        if (tiles256 >= 384 || enabled >= 3) {
            // Two workgroups fit a CU (512 slots).  With batch 256 the pixel counts are 2^k * 49, so
            // 256-row tiles often end in a thin last round (784 tiles = 1.53 rounds); 224-row
            // tiles divide those counts exactly (896 tiles = 1.75 rounds of 7/8 the length).
            // Pick the height with the smaller rounds x height.
            const long t224 = (long)((a.M + 223) / 224) * tiles_n;
            const long c256 = (tiles256 + 511) / 512 * 256, c224 = (t224 + 511) / 512 * 224;
            // (only the 128-wide tiles gain: the narrow ones are bound by their epilogue/HBM traffic)
            const bool use224 = c224 < c256 && a.Cout > 64;
            const int rc = use224 ? launch_span_bn<bf16_t, 224>(a, dmin, 224 + dmax - dmin, st)
                                  : launch_span_bn<bf16_t, 256>(a, dmin, 256 + dmax - dmin, st);
            if (rc != -1) return rc;
        }
        if (a.Cout > 64 && enabled < 2) return -1;
        return launch_span_bn<bf16_t, 128>(a, dmin, 128 + dmax - dmin, st);
    }
    // f32 parity mode: 128-row tiles
    return launch_span_bn<float, 128>(a, dmin, 128 + dmax - dmin, st);
}
