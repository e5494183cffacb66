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
template <int BM, int BN>
struct SpanLds {
    static constexpr int BROWS = BN < 64 ? 64 : BN;
    static constexpr int kMask = kTapBytes;
    static constexpr int kPo = kMask + BM * 4;
    static constexpr int kB = kPo + BM * 4;
    static constexpr int kZero = kB + 3 * BROWS * 64;
    static constexpr int kA = kZero + 64;
    __host__ __device__ static constexpr int bytes(int ita, int nslots) { return kA + nslots * ita * 64 * 64; }
};

// ita: span DMA instructions per wave per chunk (span = 64*ita rows >= BM + dmax - dmin)
template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256, 2) span_kernel(const IgemmArgs p, const int dmin, const int ita) {
    constexpr int NT = 256;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int CH = 4 * EPC;  // channels per chunk (64-byte rows)
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    using L = SpanLds<BM, BN>;
    constexpr int BROWS = L::BROWS;
    constexpr int ITB = BROWS / 64;  // filter DMA instructions per wave per step
    constexpr int BSLOT = BROWS * 4;
    constexpr int PD = 2, NSB = PD + 1;
    static_assert(WM * WN == 4 && TM % 16 == 0 && TN % 16 == 0, "tile shape");
    static_assert(L::kZero >= FM * 1024, "zero block must sit above the fragment offsets");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* sTap = (int4*)smem;  // x: row offset in the span, y: filter tap index, z/w: eh, ew
    unsigned* sMask = (unsigned*)(smem + L::kMask);
    int* sPo = (int*)(smem + L::kPo);
    uint4* sB = (uint4*)(smem + L::kB);  // [NSB][BSLOT]
    uint4* sZ = (uint4*)(smem + L::kZero);
    uint4* sA = (uint4*)(smem + L::kA);  // [1 or 2][aslot]
    const int aslot = ita * 256;         // uint4 per span slot

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
    const int cj = (lane & 3) ^ ((0x1320 >> (((lane >> 4) & 3) * 4)) & 3);
    // span row r = 16*(wave + 4i) + (lane>>2) holds input pixel m0 + dmin + r (zero page outside)
    const long pix0 = m0 + dmin + 16 * wave + (lane >> 2);
    const unsigned long a_src0 = (unsigned long)(xg + (pix0 * p.ldx + cj * EPC));
    const unsigned long a_istep = 64ul * (unsigned long)p.ldx * sizeof(T);
    unsigned long bbase[ITB];
    bool bvalid[ITB];
#pragma unroll
    for (int i = 0; i < ITB; ++i) {
        const int n = tn * BN + 16 * (wave + 4 * i) + (lane >> 2);
        bvalid[i] = n < p.Cout && 16 * (wave + 4 * i) < BN;
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

    // span of channel chunk `ic` into slot `sl`
#define VT_ISSUE_A(sl, ic)                                                                   \
    do {                                                                                     \
        const unsigned long cofs = (unsigned long)(ic) * (CH * sizeof(T));                   \
        for (int i = 0; i < ita; ++i) {                                                      \
            const long pix = pix0 + 64l * i;                                                 \
            const unsigned long src = (pix >= 0 && pix < p.M) ? a_src0 + i * a_istep + cofs : zero_src; \
            glds16(src, a_base + (unsigned)(((sl)*aslot + (wave + 4 * i) * 64) * 16));       \
        }                                                                                    \
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
    VT_ISSUE_A(0, 0);
    int ic_n = 0, it_n = 0;  // (chunk, tap) of the next filter slice to issue
#pragma unroll
    for (int s = 0; s < PD; ++s) {
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
    __syncthreads();  // tap table, row masks, zero block

    unsigned fmask[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) fmask[i] = sMask[wm * TM + i * 16 + (lane & 15)];

    int ic = 0, it = 0;  // (chunk, tap) of the step being computed
    int bcur = 0, bnxt = PD % NSB;
    bool a_young = false;  // a span was issued in the previous step
    for (int s = 0; s < nsteps; ++s) {
        // Retire this step's filter slice.  VM operations retire in issue order, so everything
        // older -- in particular the span this chunk reads -- is complete as well.  Younger:
        // the slice of the next step and, if one was issued in the previous step, the next
        // chunk's span (with ntaps == 1 that span is needed NOW, so only the slice issued after
        // it may stay in flight).
        if (s + 1 >= nsteps) {
            vm_wait<0>();
        } else if (a_young && p.ntaps > 1) {
            switch (ita) {
                case 2: vm_wait<ITB + 2>(); break;
                case 3: vm_wait<ITB + 3>(); break;
                case 4: vm_wait<ITB + 4>(); break;
                case 5: vm_wait<ITB + 5>(); break;
                case 6: vm_wait<ITB + 6>(); break;
                case 7: vm_wait<ITB + 7>(); break;
                default: vm_wait<ITB + 8>(); break;
            }
        } else {
            vm_wait<ITB>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // issue: next chunk's span at the first tap of a chunk, then the slice of step s+PD
        a_young = false;
        if (it == 0 && ic + 1 < nchunks) {
            VT_ISSUE_A((ic + 1) & 1, ic + 1);
            a_young = true;
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
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                // (sZ - i*64)[i*64] == sZ[0]: the constant stays in the instruction's offset field
                const uint4* src = ((fmask[i] >> it) & 1u) ? A : sZ - i * 64;
                af[i] = src[i * 64];
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = Bt[j * 64];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma<T>(af[i], bf[j], acc[i][j]);
        }
        if (++it == p.ntaps) it = 0, ++ic;
        bcur = (bcur + 1 == NSB) ? 0 : bcur + 1;
        bnxt = (bnxt + 1 == NSB) ? 0 : bnxt + 1;
    }
#undef VT_ISSUE_A
#undef VT_ISSUE_B
    __syncthreads();  // every wave is done with the rings; they become the staging windows

    // ---- epilogue: per wave, 16-row slabs through a private LDS window --------------------------
    constexpr int PITCH = TN + EPC;          // elements; +16 B keeps the 16-byte reads aligned
    constexpr int CPRW = TN / EPC;           // 16-byte chunks per slab row
    constexpr int RPP = 64 / CPRW;           // slab rows per read pass
    constexpr int NPASS = (16 + RPP - 1) / RPP;
    static_assert(CPRW <= 64, "slab read-out shape");
    static_assert(4 * 16 * PITCH * (int)sizeof(T) <= 3 * BROWS * 64 + 64 + 2 * 4096,
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
#pragma unroll
    for (int i = 0; i < FM; ++i) {
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
                        const uint4 rr = *(const uint4*)(rg + (po * p.ldr + ncol));
                        float fv[EPC], fr[EPC];
                        VecIO<T>::unpack(v, fv);
                        VecIO<T>::unpack(rr, fr);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) fv[e] += fr[e];
                        v = VecIO<T>::pack(fv);
                    }
                    *(uint4*)(yg + (po * p.ldy + ncol)) = v;
                }
            }
        }
        lds_fence();
    }
    if (stats) {
        const int rep = tm % VT_STAT_REPLICAS;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            float a = s1[j], b = s2[j];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            const int n = tn * BN + wn * TN + j * 16 + c;
            if (q == 0 && n < p.Cout) {
                atomicAdd(&p.stats[((long)rep * 2 + 0) * p.Cout + n], a);
                atomicAdd(&p.stats[((long)rep * 2 + 1) * p.Cout + n], b);
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_span(IgemmArgs& a, int dmin, int span, hipStream_t st) {
    using L = SpanLds<BM, BN>;
    const int ita = (span + 63) / 64;
    if (ita < 2 || ita > 8) return -1;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.chunk = (a.tiles_m + 7) / 8;
    const int nchunks = a.Cin / (64 / (int)sizeof(T));
    const int smem = L::bytes(ita, nchunks > 1 ? 2 : 1);
    const long blocks = (long)8 * a.chunk * a.tiles_n;
    auto kern = span_kernel<T, BM, BN, WM, WN>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           L::bytes(8, 2));
        if (e != hipSuccess) {
            vt_set_error("vt_conv_igemm(span): cannot raise dynamic LDS to %d: %s", L::bytes(8, 2),
                         hipGetErrorString(e));
            return VT_ERR_HIP;
        }
        raised = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, st, a, dmin, ita);
    VT_CHECK_LAUNCH("vt_conv_igemm(span)");
    return VT_OK;
}

template <typename T, int BM>
int launch_span_bn(IgemmArgs& a, int dmin, int span, hipStream_t st) {
    if (a.Cout > 64) return launch_span<T, BM, 128, 2, 2>(a, dmin, span, st);
    if (a.Cout > 32) return launch_span<T, BM, 64, 4, 1>(a, dmin, span, st);
    return launch_span<T, BM, 32, 4, 1>(a, dmin, span, st);
}

}  // namespace

// returns -1 when the span kernel does not apply (the caller then uses the general kernel)
int vt_span_dispatch(IgemmArgs& a, int dtype, void* stream) {
    static const int enabled = getenv("VT_IGEMM_SPAN") ? atoi(getenv("VT_IGEMM_SPAN")) : 1;
    if (!enabled) return -1;
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
    if (dtype == VT_BF16) {
        // 256-row tiles; maps too small to give every CU a tile (7x7 at batch 256) run the
        // 128-row variant for <= 64 output channels and fall back to the general kernel's
        // 128x128 tiles above that (measured faster there: 81 vs 97 us on 512->512 3x3 @7x7).
        // VT_IGEMM_SPAN=2 forces this kernel wherever it applies, =3 also forces 256-row tiles (tests).
        const long tiles256 = (long)((a.M + 255) / 256) * ((a.Cout + 127) / 128);
        if (tiles256 >= 384 || enabled >= 3) {
            const int rc = launch_span_bn<bf16_t, 256>(a, dmin, 256 + dmax - dmin, st);
            if (rc != -1) return rc;
        }
        if (a.Cout > 64 && enabled < 2) return -1;
        return launch_span_bn<bf16_t, 128>(a, dmin, 128 + dmax - dmin, st);
    }
    // f32 parity mode: 128-row tiles
    return launch_span_bn<float, 128>(a, dmin, 128 + dmax - dmin, st);
}
