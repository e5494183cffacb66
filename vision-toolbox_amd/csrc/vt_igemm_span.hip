// vt_igemm_span.hip -- implicit-GEMM convolution for "stride-1 grid" convs with the input
// staged ONCE per channel chunk and re-used by every filter tap.
//
// Applies when the gather steps the input by 1 and the iterated grid equals the input grid
// (every 3x3/1x1 stride-1 conv of ConvNormAct, reference components.py:26-35; every
// stride-1 data gradient; the parity classes of the stride-2 data gradients) and Cin is a
// multiple of the 64-byte K chunk.  Why a second kernel: vt_igemm.hip stages the gathered
// A rows separately for each tap, so a 3x3 conv pushes every input pixel through the
// global->LDS path 9 times; measured, that path (~9 TB/s chip wide for 64-byte segments),
// not the MFMA pipe or HBM, bounds it -- at ~570 TFLOP/s on the 128..512-channel layers and
// at 2-4x the HBM time on the 32/64-channel layers at 112x112.
// Here, with the flat pixel index m = (b*H + i)*W + j, tap t reads input pixel m + d_t,
// d_t = eh_t*W + ew_t, so the BM output pixels of a tile need ONE contiguous span of
// BM + (dmax - dmin) input pixels for all taps.  Per channel chunk (32 bf16 / 16 f32):
//   * the span is DMA'd once into a 2-slot LDS ring  (A: span x 64 B),
//   * per tap only the BN x 64 B filter slice is DMA'd (3-slot ring, 2 in flight),
//   * tap t's MFMA A-fragments are read from the span at row offset d_t - dmin; a fragment
//     row whose tap leaves the image (padding) reads a 16-byte zero block instead: the
//     per-lane LDS address is selected from a per-row tap mask built once per tile, so the
//     loop carries 3 VALU ops per fragment and nothing between ds_read and MFMA.
//
// 4 waves as WM x WN; BN=128: 2x2, wave tile 128x64; BN=64/32: 4x1, wave tile 64xBN.
// Same LDS-DMA / counted-vmcnt discipline, swizzle, statistics and XCD-aware tile map as
// vt_igemm.hip.  All A fragments of a wave are 16 rows apart, so their swizzle term is
// identical and one address per tap serves all of them.
// Epilogue: each wave stages its own 16-row slabs through a private LDS window and writes
// 16-byte row segments; no block barrier after the main loop's last one, statistics go
// straight to the global replicas from each wave.
#include <stdlib.h>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

constexpr int kTapBytes = 32 * 16;  // ntaps <= 32 on this path

__device__ __attribute__((aligned(16))) unsigned int vt_span_zero16[4];
#ifdef VT_SPAN_STAMPS  // diagnostic build only: per-workgroup phase clocks (never in the shipped library)
__device__ unsigned long long vt_span_stamps[8192 * 4];
__device__ unsigned long long vt_span_loop[8192 * 4];
#define VT_STAMP(k)                                                                  \
    do {                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 8192) vt_span_stamps[blockIdx.x * 4 + (k)] = wall_clock64(); \
    } while (0)
#define VT_LOOP_CLK(var) unsigned long long var = clock64()
#define VT_LOOP_ACC(k, a, b) loop_acc[k] += (b) - (a)
#else
#define VT_STAMP(k) do { } while (0)
#define VT_LOOP_CLK(var) do { } while (0)
#define VT_LOOP_ACC(k, a, b) do { } while (0)
#endif

// diagnostic builds only (tools/build_diag.sh): -DVT_SPAN_ABLATE=<bits>  1: no MFMA, 2: no LDS-DMA inside the
// main loop, 4: no fragment reads.  Results are wrong by construction; only the time is read.
#ifndef VT_SPAN_ABLATE
#define VT_SPAN_ABLATE 0
#endif

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

// ita: span DMA instructions per wave per chunk (span = 64*ita rows >= BM + dmax - dmin)
template <typename T, int BM, int BN, int WM, int WN, int PD, bool PP = false, bool DB = false>
__global__ void __launch_bounds__(64 * WM * WN, (BM >= 512 ? 1 : 2)) span_kernel(const IgemmArgs p, const int dmin, const int ita) {
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

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* sTap = (int4*)smem;  // x: row offset in the span, y: filter tap index, z/w: eh, ew
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

    VT_STAMP(0);
    const int W = p.Wi, H = p.Hi, HW = H * W;
    const long m0 = (long)tm * BM;
    if (tid < p.ntaps) {
        const int eh = p.h0 + p.dh[tid], ew = p.w0 + p.dw[tid];
        sTap[tid] = make_int4(eh * W + ew - dmin, tid, eh, ew);
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
    const bool a_fast = __all(pix0 >= 0 && pix0 + 16l * NW * (ita - 1) < p.M) && (p.fast_dma & 1) &&
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

    // prologue: span of chunk 0, filter slices of steps 0 and 1
    VT_ISSUE_A(0, 0);
    int ic_n = 0, it_n = 0;  // (chunk, tap) of the next filter slice to issue
#pragma unroll
    for (int s = 0; s < PD + (DB ? 1 : 0); ++s) {
        if (s < nsteps) {
            VT_ISSUE_B(s, ic_n, it_n);
            if (++it_n == p.ntaps) it_n = 0, ++ic_n;
        }
    }
    // per output row of the tile: which taps stay inside the image, and where the row goes
    for (int r = tid; r < BM; r += NT) {
        const long m = m0 + r;
        unsigned bits = 0;
        int po = 0;
        if (m < p.M) {
            const int b = (int)(m / HW);
            const int rem = (int)(m - (long)b * HW);
            const int oi = rem / W, oj = rem - oi * W;
            for (int t = 0; t < p.ntaps; ++t) {
                const int eh = p.h0 + p.dh[t], ew = p.w0 + p.dw[t];
                if ((unsigned)(oi + eh) < (unsigned)H && (unsigned)(oj + ew) < (unsigned)W) bits |= 1u << t;
            }
            po = p.dense_out ? (int)m : (b * p.oH + (oi * p.oHs + p.oh0)) * p.oW + (oj * p.oWs + p.ow0);
        }
        sMask[r] = bits;
        sPo[r] = po;
    }
    if constexpr (PP) vm_wait_dyn((min(PD, nsteps) - 1) * ITB);  // span 0 and slice 0 landed before the first LOAD tick
    __syncthreads();  // tap table, row masks, zero block

    VT_STAMP(1);
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
#ifdef VT_SPAN_STAMPS
    unsigned long long loop_acc[4] = {0, 0, 0, 0};
#endif
    if constexpr (PP) {
        // ---- ping-pong schedule (8 waves, one workgroup per CU) -------------------------------------
        // The two row halves of the tile (waves 0-3 / 4-7; one wave of each per SIMD) run half a
        // step apart: in every "tick" (= one workgroup barrier) one half issues its LDS-DMA and reads
        // its 12 fragments while the other half issues its 32 MFMAs, so the MFMA pipe of a SIMD
        // always has one wave feeding it and the other wave's staging hides behind it.
        //   group 0: LOAD(s) at tick 2s,   MFMA(s) at tick 2s+1
        //   group 1: idle at tick 0, LOAD(s) at tick 2s+1, MFMA(s) at tick 2s+2
        // B(s+1) is first read at tick 2s+2 (group 0), so every wave retires its part of it before
        // the barrier that ends tick 2s+1: group 0 after its MFMAs, group 1 after its loads.  Both
        // groups execute 2*nsteps barriers.  Requires ntaps >= PD (a span is never needed within
        // PD steps of its issue).
        static_assert(NW == 8 && WN == 2, "ping-pong needs two 4-wave row halves");
        const int grp = wm >> 1;
        if (grp == 1) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        int a_since = 1 << 20;  // LOAD ticks since this wave issued a span
        for (int s = 0; s < nsteps; ++s) {
            // ---- LOAD(s)
            ++a_since;
            if (it == 0 && ic + 1 < nchunks) {
                VT_ISSUE_A((ic + 1) & 1, ic + 1);
                a_since = 0;
            }
            if (s + PD < nsteps) {
                VT_ISSUE_B(bnxt, ic_n, it_n);
                if (++it_n == p.ntaps) it_n = 0, ++ic_n;
            }
            const int d = __builtin_amdgcn_readfirstlane(sTap[it].x);
            const int srow0 = wm * TM + (lane & 15) + d;
            const uint4* A = sA + (ic & 1) * aslot + srow0 * 4 + ((lane >> 4) ^ swz(srow0));
            const uint4* Bt = sB + bcur * BSLOT + wn * TN * 4 + b_lane;
            uint4 af[FM], bf[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = Bt[j * 64];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const uint4* src = ((fmask[i] >> it) & 1u) ? A : sZ - i * 64;
                af[i] = src[i * 64];
            }
            // younger than B(s+1): the slices issued in the last PD-1 LOAD ticks, plus a span issued
            // in one of them
            const int nb = max(0, min(PD - 1, nsteps - 2 - s));
            const int allowed = nb * ITB + ((a_since <= PD - 2) ? ita : 0);
            if (grp == 1) vm_wait_dyn(s + 1 < nsteps ? allowed : 0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // ---- MFMA(s)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma<T>(af[i], bf[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            if (grp == 0) vm_wait_dyn(s + 1 < nsteps ? allowed : 0);
            if (!(grp == 1 && s + 1 == nsteps)) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (++it == p.ntaps) it = 0, ++ic;
            bcur = (bcur + 1 == NSB) ? 0 : bcur + 1;
            bnxt = (bnxt + 1 == NSB) ? 0 : bnxt + 1;
        }
    } else if constexpr (DB) {
        // ---- register double buffering: the fragments of step s+1 are read while step s's MFMAs run --------
        // Invariant at the top of step s: filter slices <= s+2 have been issued, slice s+1 (and with it everything
        // older, in particular the span of its chunk) must have landed before the barrier, because every wave
        // reads step s+1's fragments during step s.  Slice s+3 is issued after the barrier into the ring slot of
        // slice s, whose fragments are in registers (read during step s-1; the lgkmcnt wait before the barrier
        // retires this wave's reads, the barrier everybody else's).  Ring: PD + 1 = 3 slots.  Requires
        // ntaps >= 3: a span issued at the first tap of a chunk is older than the slice issued with it, which is
        // waited for two steps later, before the span's first reader.
        static_assert(PD == 2, "double-buffered fragments: three ring slots");
        uint4 af0[FM], bf0[FN], af1[FM], bf1[FN];
        auto read_frags = [&](uint4 (&af)[FM], uint4 (&bf)[FN], int ic_r, int it_r, int bslot) {
            const int d = __builtin_amdgcn_readfirstlane(sTap[it_r].x);
            const int srow0 = wm * TM + (lane & 15) + d;
            const uint4* A = sA + (ic_r & 1) * aslot + srow0 * 4 + ((lane >> 4) ^ swz(srow0));
            const uint4* Bt = sB + bslot * BSLOT + wn * TN * 4 + b_lane;
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = Bt[j * 64];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const uint4* src = ((fmask[i] >> it_r) & 1u) ? A : sZ - i * 64;
                af[i] = src[i * 64];
            }
        };
        bool a_prev = false;  // a span was issued in the previous step
        int icr = 0, itr = 0, br = 0;  // (chunk, tap, ring slot) of the step whose fragments are read next
        read_frags(af0, bf0, 0, 0, 0);
        if (++itr == p.ntaps) itr = 0, ++icr;
        br = 1;
        auto step = [&](int s, uint4 (&caf)[FM], uint4 (&cbf)[FN], uint4 (&naf)[FM], uint4 (&nbf)[FN]) {
            const bool more = s + 1 < nsteps;
            if (more) {
                if (a_prev) vm_wait_dyn((s + 2 < nsteps ? ITB : 0) + ita);
                else if (s + 2 < nsteps) vm_wait<ITB>();
                else vm_wait<0>();
            }
            lds_fence();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            a_prev = false;
            if (it == 0 && ic + 1 < nchunks) {
                VT_ISSUE_A((ic + 1) & 1, ic + 1);
                a_prev = true;
            }
            if (s + 3 < nsteps) {
                VT_ISSUE_B(bcur, ic_n, it_n);  // slot of slice s == slot of slice s+3
                if (++it_n == p.ntaps) it_n = 0, ++ic_n;
            }
            // (unconditional: the last step re-reads its own, still valid, fragments -- a branch here makes the
            //  compiler wait for these reads before the MFMAs below)
            read_frags(naf, nbf, more ? icr : ic, more ? itr : it, more ? br : bcur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma<T>(caf[i], cbf[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (++it == p.ntaps) it = 0, ++ic;
            if (++itr == p.ntaps) itr = 0, ++icr;
            bcur = (bcur + 1 == NSB) ? 0 : bcur + 1;
            br = (br + 1 == NSB) ? 0 : br + 1;
        };
        for (int s = 0; s < nsteps; s += 2) {
            step(s, af0, bf0, af1, bf1);
            if (s + 1 < nsteps) step(s + 1, af1, bf1, af0, bf0);
        }
    } else
    for (int s = 0; s < nsteps; ++s) {
        VT_LOOP_CLK(c0);
        // Retire this step's filter slice B(s).  VM operations retire in issue order, so everything
        // older is complete as well.  Younger than B(s): the slices of the next nb steps and, if it
        // was issued fewer than PD steps ago, the next chunk's span A' (issued just before the
        // slice of its step).  A' is needed at the first tap of its chunk; if that is NOW (few taps),
        // only the slices issued after A' may stay in flight.
        const int nb = min(PD - 1, nsteps - 1 - s);
        if constexpr ((VT_SPAN_ABLATE & 2) != 0) {
            if (s == 0) vm_wait<0>();
        } else
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
        VT_LOOP_CLK(c1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        VT_LOOP_CLK(c2);
        // issue: next chunk's span at the first tap of a chunk, then the slice of step s+PD
        a_age = a_age ? a_age + 1 : 0;
        if constexpr ((VT_SPAN_ABLATE & 2) == 0) {
        if (it == 0 && ic + 1 < nchunks) {
            VT_ISSUE_A((ic + 1) & 1, ic + 1);
            a_age = 1;
        }
        if (s + PD < nsteps) {
            VT_ISSUE_B(bnxt, ic_n, it_n);
            if (++it_n == p.ntaps) it_n = 0, ++ic_n;
        }
        }

        VT_LOOP_CLK(c3);
        {
            const int d = __builtin_amdgcn_readfirstlane(sTap[it].x);
            const int srow0 = wm * TM + (lane & 15) + d;
            const uint4* A = sA + (ic & 1) * aslot + srow0 * 4 + ((lane >> 4) ^ swz(srow0));
            const uint4* Bt = sB + bcur * BSLOT + wn * TN * 4 + b_lane;
            uint4 af[FM], bf[FN];
            // filter fragments first: every MFMA of the first A row needs them
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr ((VT_SPAN_ABLATE & 4) != 0) bf[j] = make_uint4(lane + j, s, lane, 0x3f803f80u);
                else bf[j] = Bt[j * 64];
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                // (sZ - i*64)[i*64] == sZ[0]: the constant stays in the instruction's offset field
                const uint4* src = ((fmask[i] >> it) & 1u) ? A : sZ - i * 64;
                if constexpr ((VT_SPAN_ABLATE & 4) != 0) af[i] = make_uint4(0x3f803f80u, lane * 3 + i, s, (unsigned)(unsigned long)src);
                else af[i] = src[i * 64];
            }
            // all fragment reads are issued before the first MFMA (the scheduler otherwise funnels
            // the A fragments through one register quad: read, wait lgkmcnt(0), 4 MFMAs, read, ...)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr ((VT_SPAN_ABLATE & 1) != 0)
                        asm volatile("" ::"v"(af[i].x), "v"(af[i].y), "v"(af[i].z), "v"(af[i].w), "v"(bf[j].x), "v"(bf[j].y), "v"(bf[j].z), "v"(bf[j].w));
                    else mma<T>(af[i], bf[j], acc[i][j]);
                }
        }
#ifdef VT_SPAN_STAMPS
        asm volatile("s_nop 0" ::: "memory");
        unsigned long long c4 = clock64();
        VT_LOOP_ACC(0, c0, c1);
        VT_LOOP_ACC(1, c1, c2);
        VT_LOOP_ACC(2, c2, c3);
        VT_LOOP_ACC(3, c3, c4);
#endif
        if (++it == p.ntaps) it = 0, ++ic;
        bcur = (bcur + 1 == NSB) ? 0 : bcur + 1;
        bnxt = (bnxt + 1 == NSB) ? 0 : bnxt + 1;
    }
#undef VT_ISSUE_A
#undef VT_ISSUE_B
    // every wave is done with the rings; they become the staging windows (ping-pong: nobody reads
    // the rings after the last barrier inside the loop)
    if constexpr (!PP) __syncthreads();
    VT_STAMP(2);
#ifdef VT_SPAN_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 8192)
        for (int k = 0; k < 4; ++k) vt_span_loop[blockIdx.x * 4 + k] = loop_acc[k];
#endif

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
    VT_STAMP(3);
}

template <typename L>
bool smem_ok(int ita, int nw, int nchunks) { return L::bytes(ita, nchunks > 1 ? 2 : 1, nw) <= 160 * 1024; }

template <typename T, int BM, int BN, int WM, int WN, int PD, bool PP = false, bool DB = false>
int launch_span_pd(IgemmArgs& a, int dmin, int span, hipStream_t st) {
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
    auto kern = span_kernel<T, BM, BN, WM, WN, PD, PP, DB>;
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 160 * 1024, "vt_conv_igemm(span)");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("span_kernel<%s,%d,%d,%d,%d,%d%s>", sizeof(T) == 2 ? "bf16" : "f32", BM, BN, WM, WN, PD,
                   PP ? ",pingpong" : (DB ? ",dbuf" : ""));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), smem, st, a, dmin, ita);
    VT_CHECK_LAUNCH("vt_conv_igemm(span)");
#ifdef VT_SPAN_STAMPS
    {
        static int calls = 0;
        if (++calls == 20) {  // a warm launch
            (void)hipStreamSynchronize(st);
            static unsigned long long h[8192 * 4];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(vt_span_stamps), sizeof(h));
            const long nb = blocks < 8192 ? blocks : 8192;
            unsigned long long t0 = ~0ull, t1 = 0;
            double ph[3] = {0, 0, 0};
            for (long b = 0; b < nb; ++b) {
                if (h[b * 4] < t0) t0 = h[b * 4];
                if (h[b * 4 + 3] > t1) t1 = h[b * 4 + 3];
                for (int k = 0; k < 3; ++k) ph[k] += (double)(h[b * 4 + k + 1] - h[b * 4 + k]);
            }
            fprintf(stderr, "[span stamps] blocks %ld span %.2f us | avg per WG: prologue %.2f us, loop %.2f us, epilogue %.2f us\n",
                    nb, (t1 - t0) * 0.01, ph[0] / nb * 0.01, ph[1] / nb * 0.01, ph[2] / nb * 0.01);
            static unsigned long long hl[8192 * 4];
            (void)hipMemcpyFromSymbol(hl, HIP_SYMBOL(vt_span_loop), sizeof(hl));
            double la[4] = {0, 0, 0, 0};
            for (long b = 0; b < nb; ++b)
                for (int k = 0; k < 4; ++k) la[k] += (double)hl[b * 4 + k];
            fprintf(stderr, "[span stamps] loop cycles per WG (wave 0, shader clock): vmwait %.0f barrier %.0f issue %.0f reads+mfma %.0f\n",
                    la[0] / nb, la[1] / nb, la[2] / nb, la[3] / nb);
            // start-time histogram: how many rounds
            int late = 0;
            for (long b = 0; b < nb; ++b) late += (h[b * 4] - t0) * 0.01 > 5.0;
            fprintf(stderr, "[span stamps] workgroups starting > 5 us after the first: %d\n", late);
        }
    }
#endif
    return VT_OK;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_span(IgemmArgs& a, int dmin, int span, hipStream_t st) {
    // two filter slices in flight; a third (VT_SPAN_PD=3, where two workgroups still fit a CU) measured
    // 3 % slower once the issue stream was cleaned up (703 vs 727 TFLOP/s on 128ch @28x28)
    const int pd_env = VT_KNOB("VT_SPAN_PD", 0);
    constexpr int NW = WM * WN;
    const int ita = (span + 16 * NW - 1) / (16 * NW);
    const int nchunks = a.Cin / (64 / (int)sizeof(T));
    const bool fits3 = 2 * SpanLds<BM, BN, 3>::bytes(ita, nchunks > 1 ? 2 : 1, NW) <= 160 * 1024;
    const bool pd3 = pd_env == 3 && fits3 && nchunks * a.ntaps >= 6;
    if (pd3) return launch_span_pd<T, BM, BN, WM, WN, 3>(a, dmin, span, st);
    if constexpr (sizeof(T) == 2 && BN == 128 && WM == 2 && WN == 2 && (BM == 224 || BM == 256)) {
        // the MFMA-bound 3x3 layers: fragments of step s+1 are read under the MFMAs of step s
        const int db_env = VT_KNOB("VT_SPAN_DB", 0);
        if (db_env && a.ntaps >= 3 && nchunks * a.ntaps >= 4)
            return launch_span_pd<T, BM, BN, WM, WN, 2, false, true>(a, dmin, span, st);
    }
    return launch_span_pd<T, BM, BN, WM, WN, 2>(a, dmin, span, st);
}

template <typename T, int BM>
int launch_span_bn(IgemmArgs& a, int dmin, int span, hipStream_t st) {
    if constexpr (BM == 512) {
        // 8 waves of 128 x 64: the filter slices (the larger part of the staged bytes) are fetched
        // once per 512 pixels instead of once per 256.  Narrow layers: 4 waves of 128 x BN -- their
        // K loop is 1-4 steps long, so a workgroup is mostly prologue + epilogue and what matters
        // is the bytes it keeps in flight.
        if (a.Cout > 64) {
            const int pp = VT_KNOB("VT_SPAN_PP", 1);
            if (pp && a.ntaps >= 3) return launch_span_pd<T, BM, 128, 4, 2, 3, true>(a, dmin, span, st);
            return launch_span<T, BM, 128, 4, 2>(a, dmin, span, st);
        }
        if (a.Cout > 32) return launch_span<T, BM, 64, 4, 1>(a, dmin, span, st);
        return launch_span<T, BM, 32, 4, 1>(a, dmin, span, st);
    } else if constexpr (BM == 224) {
        // 7 x 32 rows: 2 x 2 waves of 112 x BN/2 for every width
        if (a.Cout > 64) return launch_span<T, BM, 128, 2, 2>(a, dmin, span, st);
        if (a.Cout > 32) return launch_span<T, BM, 64, 2, 2>(a, dmin, span, st);
        return launch_span<T, BM, 32, 2, 2>(a, dmin, span, st);
    } else {
        if (a.Cout > 64) {
            // 256 x 128 tile: 4 waves of 128 x 64 (VT_SPAN_WAVES=8: 8 waves of 64 x 64; measured equal)
            const int waves = VT_KNOB("VT_SPAN_WAVES", 4);
            if constexpr (BM == 256 && sizeof(T) == 2) {
                if (waves == 8) return launch_span<T, BM, 128, 4, 2>(a, dmin, span, st);
            }
            return launch_span<T, BM, 128, 2, 2>(a, dmin, span, st);
        }
        if (a.Cout > 32) return launch_span<T, BM, 64, 4, 1>(a, dmin, span, st);
        return launch_span<T, BM, 32, 4, 1>(a, dmin, span, st);
    }
}

}  // namespace

// returns -1 when the span kernel does not apply (the caller then uses the general kernel)
int vt_span_dispatch(IgemmArgs& a, int dtype, void* stream) {
    const int enabled = VT_KNOB("VT_IGEMM_SPAN", 1);
    if (!enabled) return -1;
    const int fast_dma = VT_KNOB("VT_SPAN_FAST_DMA", 1);
    const int prio = VT_KNOB("VT_SPAN_PRIO", 0);
    a.fast_dma = (fast_dma & 1) | ((prio & 3) << 1);
    const int ch = 4 * vt_epc(dtype);
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
    // 632 -> 570 us; 1472 -> 768 @14x14: 171 -> 136 us; 256 -> 256 @28x28: equal).  VT_SPAN_GEMM=1: keep them here
    const int gemm_minc = VT_KNOB("VT_SPAN_GEMM_MINC", 320);
    if (dtype == VT_BF16 && a.ntaps == 1 && (a.Cin >= gemm_minc || a.Cout >= gemm_minc) && !VT_KNOB("VT_SPAN_GEMM", 0) &&
        (VT_KNOB("VT_IGEMM_W8", 3) & 2) && (long)((a.M + 255) / 256) * ((a.Cout + 127) / 128) >= 256 && enabled < 2)
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
            // Pick the height with the smaller rounds x height (VT_SPAN_BM=256/224 overrides).
            const int bm_env = VT_KNOB("VT_SPAN_BM", 0);
            const long t224 = (long)((a.M + 223) / 224) * tiles_n;
            const long c256 = (tiles256 + 511) / 512 * 256, c224 = (t224 + 511) / 512 * 224;
            // (only the 128-wide tiles gain: the narrow ones are bound by their epilogue/HBM traffic)
            const bool use224 = bm_env ? bm_env == 224 : (c224 < c256 && a.Cout > 64);
            if (bm_env == 512) {
                const int rc = launch_span_bn<bf16_t, 512>(a, dmin, 512 + dmax - dmin, st);
                if (rc != -1) return rc;
            }
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
