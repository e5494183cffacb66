// vt_igemm_span.hip -- implicit-GEMM convolution for "stride-1 grid" convs with the input
// staged ONCE per channel chunk and re-used by every filter tap.
//
// Applies when the gather steps the input by 1 and the iterated grid equals the input grid
// (every 3x3/1x1 stride-1 conv of ConvNormAct, reference components.py:26-35; every
// stride-1 data gradient; the parity classes of the stride-2 data gradients) and Cin is a
// multiple of the 64-byte K chunk.  Why a second kernel: vt_igemm.hip stages the gathered
// A rows separately for each tap, so a 3x3 conv pushes every input pixel through the
// global->LDS path 9 times and the filter tile once per 128 pixels; measured, that path
// (~9 TB/s chip wide for 64-byte segments) and not the MFMA pipe bounds it at ~570 TFLOP/s.
// Here, with the flat pixel index m = (b*H + i)*W + j, tap t reads input pixel m + d_t,
// d_t = eh_t*W + ew_t, so the BM output pixels of a tile need ONE contiguous span of
// BM + (dmax - dmin) input pixels for all taps.  Per channel chunk (32 bf16 / 16 f32):
//   * the span is DMA'd once into a 2-slot LDS ring  (A: SPAN x 64 B),
//   * per tap only the 128 x 64 B filter slice is DMA'd (3-slot ring, 2 in flight),
//   * tap t's MFMA A-fragments are read from the span at row offset d_t - dmin;
//     taps that leave the image (padding) are masked per fragment row with a precomputed
//     per-lane bit mask (the span itself is loaded unconditionally).
// With a 256 x 128 tile that is ~10 KB staged per 2.1 MFLOP (205 flop/B) instead of 16 KB
// per 1.05 MFLOP (64 flop/B).
//
// 4 waves (2 x 2), wave tile 128 x (BN/2): 8 x FN accumulator tiles of 16x16, 12 ds_read_b128
// per 32 MFMAs.  Same LDS-DMA / counted-vmcnt discipline, swizzle, epilogue, statistics and
// XCD-aware tile map as vt_igemm.hip.  All A fragments of a wave are 16 rows apart, so their
// swizzle term is identical and one address per tap serves all eight.
#include <stdlib.h>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

constexpr int kTapBytes = VT_MAX_TAPS * 16;
constexpr int kStatBytes = 2 * 128 * 4;
constexpr int kHdrBytes = kTapBytes + kStatBytes;

__device__ __attribute__((aligned(16))) unsigned int vt_span_zero16[4];

__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

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

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

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

// ITA: span DMA instructions per wave per chunk (span = 64*ITA rows >= BM + dmax - dmin)
template <typename T, int BM, int BN, int ITA, int PD, int ABL = 0>
__global__ void __launch_bounds__(256, 2) span_kernel(const IgemmArgs p, const int dmin) {
    constexpr int NT = 256, WM = 2, WN = 2;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int CH = 4 * EPC;  // channels per chunk (64-byte rows)
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    constexpr int SPAN = 64 * ITA;        // rows of the A span image
    constexpr int ASLOT = SPAN * 4;       // uint4 slots per A ring slot
    constexpr int ITB = BN / 64;          // filter DMA instructions per wave per step
    constexpr int BSLOT = BN * 4;
    constexpr int NSA = 2, NSB = PD + 1;  // PD filter slices in flight (PD == 3 requires ntaps >= 3)
    static_assert(BM % 32 == 0 && BN % 64 == 0 && BN <= 128, "tile shape");
    static_assert(SPAN >= BM, "span shorter than the tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* sTap = (int4*)smem;  // x: row offset in the span, y: filter tap index, z/w unused
    float* sStat = (float*)(smem + kTapBytes);
    uint4* sA = (uint4*)(smem + kHdrBytes);   // [NSA][ASLOT]
    uint4* sB = sA + NSA * ASLOT;             // [NSB][BSLOT]
    T* sOut = (T*)(smem + kHdrBytes);         // [BM][BN] after the loop

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

    const int W = p.Wi, H = p.Hi;
#pragma unroll
    for (int t = 0; t < VT_MAX_TAPS; ++t) {
        if (t < p.ntaps && tid == t) {
            const int eh = p.h0 + p.dh[t], ew = p.w0 + p.dw[t];
            sTap[t] = make_int4(eh * W + ew - dmin, t, eh, ew);
        }
    }
    for (int i = tid; i < 2 * BN; i += NT) sStat[i] = 0.f;
    __syncthreads();

    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ wg = (const T*)p.w;
    const unsigned long zero_src = (unsigned long)(const void*)vt_span_zero16;
    const unsigned a_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sA;
    const unsigned b_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sB;

    // ---- DMA geometry ---------------------------------------------------------------------
    // an instruction fills 16 rows x 64 B; lane l owns row 16j + (l>>2), chunk (l&3)^swz(row)
    const int cj = (lane & 3) ^ ((0x1320 >> (((lane >> 4) & 3) * 4)) & 3);
    const long m0 = (long)tm * BM;
    unsigned long aptr[ITA];   // running source of this lane's span rows (or the zero page)
    unsigned astep[ITA];
#pragma unroll
    for (int i = 0; i < ITA; ++i) {
        const int row = 16 * (wave + 4 * i) + (lane >> 2);
        const long pix = m0 + dmin + row;
        const bool v = pix >= 0 && pix < p.M;
        aptr[i] = v ? (unsigned long)(xg + (pix * p.ldx + cj * EPC)) : zero_src;
        astep[i] = v ? CH * (unsigned)sizeof(T) : 0u;
    }
    unsigned long bbase[ITB];
    bool bvalid[ITB];
#pragma unroll
    for (int i = 0; i < ITB; ++i) {
        const int n = tn * BN + 16 * (wave + 4 * i) + (lane >> 2);
        bvalid[i] = n < p.Cout;
        bbase[i] = (unsigned long)(wg + ((long)(bvalid[i] ? n : 0) * p.ldw + cj * EPC));
    }

    // ---- per-lane tap validity of the fragment rows -------------------------------------------
    unsigned fmask[FM];  // ntaps <= 32 on this path
    {
        const int HW = H * W;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const long m = m0 + wm * TM + i * 16 + (lane & 15);
            unsigned bits = 0;
            if (m < p.M) {
                const int rem = (int)(m % HW);
                const int oi = rem / W, oj = rem - oi * W;
                for (int t = 0; t < p.ntaps; ++t) {
                    const int4 te = sTap[t];
                    if ((unsigned)(oi + te.z) < (unsigned)H && (unsigned)(oj + te.w) < (unsigned)W) bits |= 1u << t;
                }
            }
            fmask[i] = bits;
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = p.Cin / CH;
    const int nsteps = nchunks * p.ntaps;
    const int b_lane = (lane & 15) * 4 + ((lane >> 4) ^ swz(lane & 15));

#define VT_ISSUE_A(aslot)                                                                  \
    do {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < ITA; ++i) {                                  \
            glds16(aptr[i], a_base + (unsigned)(((aslot)*ASLOT + (wave + 4 * i) * 64) * 16)); \
            aptr[i] += astep[i];                                                           \
        }                                                                                  \
    } while (0)
    // filter slice of step (chunk ic, tap it): rows n, K offset it*Cin + ic*CH
#define VT_ISSUE_B(bslot, ic, it)                                                          \
    do {                                                                                   \
        const long koff = ((long)(it)*p.Cin + (long)(ic)*CH) * (long)sizeof(T);            \
        _Pragma("unroll") for (int i = 0; i < ITB; ++i) {                                  \
            const unsigned long ps = bvalid[i] ? bbase[i] + koff : zero_src;               \
            glds16(ps, b_base + (unsigned)(((bslot)*BSLOT + (wave + 4 * i) * 64) * 16));   \
        }                                                                                  \
    } while (0)

    // prologue: span of chunk 0, filter slices of steps 0 and 1
    VT_ISSUE_A(0);
    int ic_n = 0, it_n = 0;  // (chunk, tap) of the next filter slice to issue
#pragma unroll
    for (int s = 0; s < PD; ++s) {
        if (s < nsteps) {
            VT_ISSUE_B(s, ic_n, it_n);
            if (++it_n == p.ntaps) it_n = 0, ++ic_n;
        }
    }

    int ic = 0, it = 0;      // (chunk, tap) of the step being computed
    int bcur = 0, bnxt = PD % NSB;
    int a_age = 0;  // steps since a span was issued (0: none inside the prefetch window)
    for (int s = 0; s < nsteps; ++s) {
        // Retire this step's filter slice.  VM operations retire in issue order, so everything
        // older -- in particular the span this chunk reads -- is complete as well.  Younger:
        // the slices of the next yb steps and, if one was issued in the last PD-1 steps, the next
        // chunk's span (with ntaps == 1 that span is needed NOW, so only the slice issued after
        // it may stay in flight).
        const int yb = min(PD - 1, nsteps - 1 - s);
        const bool ya = a_age >= 1 && a_age <= PD - 1 && p.ntaps > 1;
        if (yb == 0)
            vm_wait<0>();
        else if (yb == 1)
            ya ? vm_wait<ITB + ITA>() : vm_wait<ITB>();
        else
            ya ? vm_wait<2 * ITB + ITA>() : vm_wait<2 * ITB>();
        if constexpr (ABL != 1) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // issue: next chunk's span at the first tap of a chunk, then the slice of step s+PD
        a_age = a_age ? a_age + 1 : 0;
        if (ABL != 3 && it == 0 && ic + 1 < nchunks) {
            VT_ISSUE_A((ic + 1) & 1);
            a_age = 1;
        }
        if (ABL != 3 && s + PD < nsteps) {
            VT_ISSUE_B(bnxt, ic_n, it_n);
            if (++it_n == p.ntaps) it_n = 0, ++ic_n;
        }
        // with a single tap the span issued now is needed by the very next step: it must not be
        // left in flight behind that step's slice -> handled by vm_wait<ITB> above (ntaps == 1)

        {
            const int d = __builtin_amdgcn_readfirstlane(sTap[it].x);
            const int srow0 = wm * TM + (lane & 15) + d;
            const uint4* A = sA + (ic & 1) * ASLOT + srow0 * 4 + ((lane >> 4) ^ swz(srow0));
            const uint4* Bt = sB + bcur * BSLOT + wn * TN * 4 + b_lane;
            uint4 af[FM], bf[FN];
            if (ABL != 2 || s == 0) {
#pragma unroll
                for (int i = 0; i < FM; ++i) af[i] = A[i * 64];
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[j] = Bt[j * 64];
            } else {
#pragma unroll
                for (int i = 0; i < FM; ++i) af[i] = make_uint4(s + i, lane, 0x3f803f80, 0x3f803f80);
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[j] = make_uint4(s, lane + j, 0x3f803f80, 0x3f803f80);
            }
#pragma unroll
            for (int i = 0; i < (ABL == 4 ? 0 : FM); ++i) {
                const unsigned keep = 0u - ((fmask[i] >> it) & 1u);  // all ones / zero
                af[i].x &= keep;
                af[i].y &= keep;
                af[i].z &= keep;
                af[i].w &= keep;
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (ABL == 5) {
                        if (j == 0) acc[i][0][0] += __uint_as_float(af[i].x ^ bf[0].x);  // keep operands live, no MFMA
                    } else {
                        mma<T>(af[i], bf[j], acc[i][j]);
                    }
                }
        }
        if (++it == p.ntaps) it = 0, ++ic;
        bcur = (bcur + 1 == NSB) ? 0 : bcur + 1;
        bnxt = (bnxt + 1 == NSB) ? 0 : bnxt + 1;
    }
#undef VT_ISSUE_A
#undef VT_ISSUE_B
    __syncthreads();

    // ---- epilogue (as vt_igemm.hip) -----------------------------------------------------------
    const bool affine = p.flags & VT_CONV_AFFINE;
    const bool relu = p.flags & VT_CONV_RELU;
    const bool stats = p.flags & VT_CONV_STATS;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int col = wn * TN + j * 16 + (lane & 15);
        const int n = tn * BN + col;
        float sc = 1.f, sf = 0.f;
        if (affine && n < p.Cout) {
            if (p.scale) sc = p.scale[n];
            sf = p.shift[n];
        }
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r];
                if (affine) v = fmaf(v, sc, sf);
                if (relu) v = fmaxf(v, 0.f);
                const T tv = from_float<T>(v);
                const int row = wm * TM + i * 16 + (lane >> 4) * 4 + r;
                sOut[row * BN + col] = tv;
                const float fv = (float)tv;
                s += fv;
                ss += fv * fv;
            }
        }
        if (stats) {
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (lane < 16) {
                atomicAdd(&sStat[col], s);
                atomicAdd(&sStat[BN + col], ss);
            }
        }
    }
    __syncthreads();

    if (stats) {
        for (int i = tid; i < 2 * BN; i += NT) {
            const int which = i / BN, col = i % BN;
            const int n = tn * BN + col;
            if (n < p.Cout) {
                const int rep = tm % VT_STAT_REPLICAS;
                atomicAdd(&p.stats[((long)rep * 2 + which) * p.Cout + n], sStat[i]);
            }
        }
    }

    constexpr int CPR = BN / EPC;
    const uint4* sOut4 = (const uint4*)sOut;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    const bool has_res = (p.flags & VT_CONV_RESIDUAL) != 0;
    const int HoWo = p.Ho * p.Wo;
#pragma unroll 2
    for (int idx = tid; idx < BM * CPR; idx += NT) {
        const int row = idx / CPR, ch = idx % CPR;
        const long m = m0 + row;
        const int n = tn * BN + ch * EPC;
        if (m < p.M && n < p.Cout) {
            long po = m;
            if (!p.dense_out) {
                const int b = (int)(m / HoWo);
                const int rem = (int)(m - (long)b * HoWo);
                const int oi = rem / p.Wo;
                const int oj = rem - oi * p.Wo;
                po = ((long)b * p.oH + (oi * p.oHs + p.oh0)) * p.oW + (oj * p.oWs + p.ow0);
            }
            uint4 v = sOut4[idx];
            if (has_res) {
                const uint4 r = *(const uint4*)(rg + (po * p.ldr + n));
                float fv[EPC], fr[EPC];
                VecIO<T>::unpack(v, fv);
                VecIO<T>::unpack(r, fr);
#pragma unroll
                for (int e = 0; e < EPC; ++e) fv[e] += fr[e];
                v = VecIO<T>::pack(fv);
            }
            *(uint4*)(yg + (po * p.ldy + n)) = v;
        }
    }
}

template <typename T, int BM, int BN, int ITA, int PD, int ABL = 0>
int launch_span(IgemmArgs& a, int dmin, hipStream_t st) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.chunk = (a.tiles_m + 7) / 8;
    constexpr int ring = 2 * 64 * ITA * 64 + (PD + 1) * BN * 64;
    constexpr int outb = BM * BN * (int)sizeof(T);
    constexpr int smem = kHdrBytes + (ring > outb ? ring : outb);
    static_assert(smem <= 160 * 1024, "exceeds the LDS of a CU");
    const long blocks = (long)8 * a.chunk * a.tiles_n;
    auto kern = span_kernel<T, BM, BN, ITA, PD, ABL>;
    if (smem > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) {
                vt_set_error("vt_conv_igemm(span): cannot raise dynamic LDS to %d: %s", smem, hipGetErrorString(e));
                return VT_ERR_HIP;
            }
            raised = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, st, a, dmin);
    VT_CHECK_LAUNCH("vt_conv_igemm(span)");
    return VT_OK;
}

template <typename T, int BM, int BN>
int launch_span_ita(IgemmArgs& a, int dmin, int ita, hipStream_t st) {
    static const int pd_env = getenv("VT_SPAN_PD") ? atoi(getenv("VT_SPAN_PD")) : 3;
    static const int abl = getenv("VT_SPAN_ABLATE") ? atoi(getenv("VT_SPAN_ABLATE")) : 0;  // timing experiments only
    if (abl && ita == 5 && sizeof(T) == 2) {
        switch (abl) {
            case 1: return launch_span<T, BM, BN, 5, 2, 1>(a, dmin, st);
            case 2: return launch_span<T, BM, BN, 5, 2, 2>(a, dmin, st);
            case 3: return launch_span<T, BM, BN, 5, 2, 3>(a, dmin, st);
            case 4: return launch_span<T, BM, BN, 5, 2, 4>(a, dmin, st);
            case 5: return launch_span<T, BM, BN, 5, 2, 5>(a, dmin, st);
        }
    }
    if (a.ntaps >= 3 && pd_env == 3) {
        switch (ita) {
            case 5: return launch_span<T, BM, BN, 5, 3>(a, dmin, st);
            case 6: return launch_span<T, BM, BN, 6, 3>(a, dmin, st);
            case 7: return launch_span<T, BM, BN, 7, 3>(a, dmin, st);
            case 8: return launch_span<T, BM, BN, 8, 3>(a, dmin, st);
            default: return -1;
        }
    }
    switch (ita) {
        case 5: return launch_span<T, BM, BN, 5, 2>(a, dmin, st);
        case 6: return launch_span<T, BM, BN, 6, 2>(a, dmin, st);
        case 7: return launch_span<T, BM, BN, 7, 2>(a, dmin, st);
        case 8: return launch_span<T, BM, BN, 8, 2>(a, dmin, st);
        default: return -1;
    }
}

}  // namespace

// returns -1 when the span kernel does not apply (the caller then uses the general kernel)
int vt_span_dispatch(IgemmArgs& a, int dtype, void* stream) {
    static const int enabled = getenv("VT_IGEMM_SPAN") ? atoi(getenv("VT_IGEMM_SPAN")) : 0;
    if (!enabled) return -1;
    const int ch = 4 * vt_epc(dtype);
    if (a.sh != 1 || a.sw != 1 || a.Ho != a.Hi || a.Wo != a.Wi) return -1;
    if (a.Cin % ch != 0 || a.Cout <= 32 || a.ntaps > 32) return -1;
    if ((long)a.M + 2L * a.Wi * VT_MAX_TAPS > 0x7fffffffL) return -1;
    int dmin = 1 << 30, dmax = -(1 << 30);
    for (int t = 0; t < a.ntaps; ++t) {
        const int d = (a.h0 + a.dh[t]) * a.Wi + (a.w0 + a.dw[t]);
        dmin = d < dmin ? d : dmin;
        dmax = d > dmax ? d : dmax;
    }
    constexpr int BM = 256;
    const int span = BM + (dmax - dmin);
    int ita = (span + 63) / 64;
    if (ita < 5) ita = 5;
    if (ita > 8) return -1;  // wide images with many taps: the span would not fit; general kernel
    // small problems: the 256-row tile leaves CUs idle
    if (enabled < 2 && (long)((a.M + BM - 1) / BM) * ((a.Cout + 127) / 128) < 96) return -1;  // VT_IGEMM_SPAN=2 forces
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VT_BF16) {
        if (a.Cout > 64) return launch_span_ita<bf16_t, BM, 128>(a, dmin, ita, st);
        return launch_span_ita<bf16_t, BM, 64>(a, dmin, ita, st);
    }
    // f32 parity mode: 128-row tiles (the f32 output tile must fit the same LDS)
    const int span32 = 128 + (dmax - dmin);
    int ita32 = (span32 + 63) / 64;
    if (ita32 < 5) ita32 = 5;
    if (ita32 > 8) return -1;
    if (a.Cout > 64) return launch_span_ita<float, 128, 128>(a, dmin, ita32, st);
    return launch_span_ita<float, 128, 64>(a, dmin, ita32, st);
}
