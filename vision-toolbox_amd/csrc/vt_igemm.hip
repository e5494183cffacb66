// vt_igemm.hip -- implicit-GEMM convolution on MFMA for gfx950 (MI355X).
//
// One kernel serves the forward conv of ConvNormAct (reference
// vision_toolbox/components.py:26-35), its data gradient, and the per-parity
// data gradient of the stride-2 convs (darknet.py:34,44).  GEMM view:
//
//     M = B*Ho*Wo output pixels, N = Cout, K = ntaps*Cin
//     A[m][k=(t,c)] = in(b, i*sh+h0+dh[t], j*sw+w0+dw[t], c)   gathered, 0 outside
//     B[k][n]       = w[n][t][c]                               ("B^T": K contiguous)
//
// Tiling: BM x BN output tile per 256-thread workgroup (4 waves, 64-wide), K in
// steps of 64 bytes per row (32 bf16 / 16 f32).
//
// Operand staging is an LDS-DMA ring: every wave-instruction is one
// global_load_lds_dwordx4 that moves 64 x 16 B (one 16-byte chunk = 8 bf16
// channels of one NHWC pixel / one filter row per lane) straight from global
// memory into LDS, no VGPR round trip.  NS = PD+1 stage slots are kept; the loads
// of K-step ks+PD are issued right after the single barrier of K-step ks, and a
// COUNTED s_waitcnt vmcnt(IT*(PD-1)) retires exactly the stage about to be read,
// so PD stages stay in flight across barriers (the loop is latency-bound
// otherwise: measured 470 TFLOP/s with a one-deep register-staged prefetch).
// Zero padding, M/N/K tails: the lane's SOURCE address is redirected to a
// 16-byte zero page, so every lane of every wave issues every load (the LDS
// destination of an LDS-DMA is lane-linear and cannot be predicated).
// The LDS image is XOR-swizzled through the source side: LDS slot q of a stage is
// written linearly, and the lane that owns slot q fetches chunk (q&3)^swz(q>>2)
// of row q>>2, which makes the ds_read_b128 of the MFMA fragments conflict free.
//
// The loads are issued from inline asm on purpose: hipcc would otherwise treat
// each LDS-DMA as a pending LDS write and drain vmcnt(0) before the first ds_read
// of every K-step.  Consequently this kernel counts its own VM operations: there
// is no compiler-visible global load between the first DMA and the final
// vmcnt(0) (kernel arguments arrive through the scalar cache).
//
// MFMA: v_mfma_f32_16x16x32_bf16 (bf16) or 4 x v_mfma_f32_16x16x4_f32 (exact
// f32 parity mode).  Both use the same fragment addressing: lane l reads chunk
// (l>>4) of row (l&15); for f32 the four MFMAs consume elements 0..3 of the
// chunk, which only permutes the order of the K summation.
//
// Epilogue: accumulators -> [affine] -> [relu] -> LDS tile in the output dtype
// -> coalesced 16-byte row stores [+ residual].  With VT_CONV_STATS the
// per-channel sum / sum of squares of the ROUNDED outputs are reduced
// wave -> LDS -> one global atomic per channel per workgroup, spread over
// VT_STAT_REPLICAS replicas.
//
// Workgroup -> tile map is XCD aware: blocks b and b+8 share an XCD (and its
// L2), so each XCD walks a contiguous range of M tiles, N tiles fastest: the
// N tiles of one M tile and the 3x3 halos of neighbouring M tiles hit L2.
#include <stdlib.h>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

constexpr int kTapBytes = VT_MAX_TAPS * 16;  // int4 per tap
// statistics scratch, per row-wave (sum, sumsq): WM * 2 * BN floats, at least 2 KiB
constexpr int stat_bytes(int WM, int BN) { return WM * 2 * BN * 4 > 2048 ? WM * 2 * BN * 4 : 2048; }

__device__ __attribute__((aligned(16))) unsigned int vt_zero16[4];  // source of every padded chunk

// chunk swizzle of a 64-byte LDS row: conflict free for ds_read_b128 issued as
// (row = l&15, chunk = l>>4).
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

// one LDS-DMA wave-instruction: lane l copies 16 B from its own global address to
// LDS byte address lds_base + 16*l.  M0 carries the LDS base and is restored.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
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

template <int BM, int BN, int PD, int NW, int WM>
struct Geom {
    static constexpr int NI_A = BM / 16;                // DMA instructions covering the A tile
    static constexpr int NI = (BM + BN) / 16;           // ... the whole stage
    static constexpr int IT = (NI + NW - 1) / NW;       // per wave (NW waves), padded with dummies
    static constexpr int SS = IT * NW * 64;             // uint4 slots per stage incl. dummy area
    static constexpr int HDR = kTapBytes + stat_bytes(WM, BN);
    static constexpr int NS = PD + 1;                   // stage slots
    static constexpr int STAGE_BYTES = NS * SS * 16;
};

template <typename T, int BM, int BN, int WM, int WN, int PD, bool UK>
__global__ void __launch_bounds__(64 * WM * WN) igemm_kernel(const IgemmArgs p) {
    constexpr int NW = WM * WN;  // waves: 4, or 8 for the 256-row tiles of 128 / 160 filter columns
    using G = Geom<BM, BN, PD, NW, WM>;
    constexpr int NT = 64 * NW;
    constexpr int kHdrBytes = G::HDR;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 4 * EPC;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    constexpr int IT = G::IT, SS = G::SS, NS = G::NS;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be 16-granular");
    static_assert(BN <= 160, "filter tile width");
    static_assert(BM % 16 == 0 && BN % 16 == 0, "tile rows come in groups of 16 per DMA instruction");
    static_assert(PD >= 1 && PD <= 3, "prefetch distance");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* sTap = (int4*)smem;
    float* sStat = (float*)(smem + kTapBytes);
    uint4* sStage = (uint4*)(smem + kHdrBytes);  // [NS][SS]: A rows, then B rows, then dummy slots
    T* sOut = (T*)(smem + kHdrBytes);            // [BM][BN], aliases the ring after the K loop

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile map
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int tn = slot % p.tiles_n;
    const int ml = slot / p.tiles_n;
    const int tm = xcd * p.chunk + ml;
    if (ml >= p.chunk || tm >= p.tiles_m) return;

    // static indices keep the by-value tap table in SGPRs (a lane-indexed read
    // of a kernel argument would copy the struct to scratch)
#pragma unroll
    for (int t = 0; t < VT_MAX_TAPS; ++t) {
        if (t < p.ntaps && tid == t) {
            const int dh = p.dh[t], dw = p.dw[t];
            sTap[t] = make_int4(dh, dw, (dh * p.Wi + dw) * p.ldx, 0);
        }
    }
    if (tid < 2 * BN) sStat[tid] = 0.f;
    __syncthreads();

    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ wg = (const T*)p.w;
    const unsigned long zero_src = (unsigned long)(const void*)vt_zero16;
    const unsigned ring_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sStage;

    // ---- per-lane DMA geometry (fixed for the whole K loop) --------------------
    // instruction j = wave + NW*i of a stage fills slots [64j, 64j+64); lane l owns slot
    // q = 64j + l = row (q>>2), position (q&3), and fetches chunk (q&3) ^ swz(row).
    // (row>>2)&3 == (l>>4)&3 for every j, so the chunk is the same for all of a lane's loads.
    const int cj = (lane & 3) ^ ((0x1320 >> (((lane >> 4) & 3) * 4)) & 3);
    int hb[IT], wb[IT];
    long off0[IT];
    bool valid[IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int j = wave + NW * i;
        hb[i] = wb[i] = 0;
        off0[i] = 0;
        valid[i] = false;
        if (j < G::NI_A) {
            const int row = j * 16 + (lane >> 2);
            const int m = tm * BM + row;
            valid[i] = m < p.M;
            const int mm = valid[i] ? m : 0;
            const int b = mm / HoWo;
            const int rem = mm - b * HoWo;
            const int oi = rem / p.Wo;
            const int oj = rem - oi * p.Wo;
            hb[i] = oi * p.sh + p.h0;
            wb[i] = oj * p.sw + p.w0;
            off0[i] = ((long)(b * p.Hi + hb[i]) * p.Wi + wb[i]) * p.ldx;
        } else if (j < G::NI) {
            const int n = tn * BN + (j - G::NI_A) * 16 + (lane >> 2);
            valid[i] = n < p.Cout;
            off0[i] = (long)(valid[i] ? n : 0) * p.ldw;
        }
    }

    int kk = cj * EPC;  // flattened K index of this lane's chunk
    int tap = kk / p.Cin;
    int c = kk - tap * p.Cin;

    // UK ("uniform K"): Cin is a multiple of the K-step, so the tap and the channel offset of
    // a K-step are the same for the whole workgroup and live in SGPRs.  Everything that
    // depends on the lane is then fixed for the whole loop and precomputed: the A rows' base
    // addresses and a per-row bit mask of the taps that fall inside the image, and running
    // filter-row pointers.  A DMA then costs ~7 VALU instead of ~50 -- with three waves per
    // SIMD the address arithmetic, not the MFMA pipe, was what the loop saturated.
    unsigned long abase[IT], amask[IT], bptr[IT];
    unsigned bstep[IT];
    int tap_u = 0, c0_u = 0;
    if constexpr (UK) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int j = wave + NW * i;
            abase[i] = (unsigned long)(xg + (off0[i] + cj * EPC));
            amask[i] = 0;
            bptr[i] = zero_src;
            bstep[i] = 0;
            if (j < G::NI_A) {
                if (valid[i]) {
                    for (int t = 0; t < p.ntaps; ++t) {
                        const int4 te = sTap[t];
                        if ((unsigned)(hb[i] + te.x) < (unsigned)p.Hi && (unsigned)(wb[i] + te.y) < (unsigned)p.Wi)
                            amask[i] |= 1ul << t;
                    }
                }
            } else if (j < G::NI && valid[i]) {
                bptr[i] = (unsigned long)(wg + (off0[i] + cj * EPC));
                bstep[i] = BK * (unsigned)sizeof(T);
            }
        }
    }

    // issue the IT DMA instructions of one K-step into ring slot `st`, then advance K
#define VT_ISSUE_STAGE(st)                                                                      \
    do {                                                                                        \
        if constexpr (UK) {                                                                     \
            const long koff = ((long)__builtin_amdgcn_readfirstlane(sTap[tap_u].z) + c0_u) * (long)sizeof(T); \
            _Pragma("unroll") for (int i = 0; i < IT; ++i) {                                    \
                const int j = wave + NW * i;                                                     \
                const bool isA = (NW * i + NW - 1 < G::NI_A) ? true : (NW * i >= G::NI_A ? false : j < G::NI_A); \
                unsigned long ps;                                                               \
                if (isA) {                                                                      \
                    ps = ((amask[i] >> tap_u) & 1ul) ? abase[i] + koff : zero_src;              \
                } else {                                                                        \
                    ps = bptr[i];                                                               \
                    bptr[i] += bstep[i];                                                        \
                }                                                                               \
                glds16((const void*)ps, ring_base + (unsigned)(((st)*SS + j * 64) * 16));       \
            }                                                                                   \
            c0_u += BK;                                                                         \
            if (c0_u >= p.Cin) {                                                                \
                c0_u = 0;                                                                       \
                ++tap_u;                                                                        \
            }                                                                                   \
            break;                                                                              \
        }                                                                                       \
        const bool kval = tap < p.ntaps;                                                        \
        const int4 te = sTap[kval ? tap : 0];                                                   \
        _Pragma("unroll") for (int i = 0; i < IT; ++i) {                                        \
            const int j = wave + NW * i;                                                         \
            /* A / B / dummy is known at compile time unless the boundary cuts a group of 4 */  \
            const bool isA = (NW * i + NW - 1 < G::NI_A) ? true : (NW * i >= G::NI_A ? false : j < G::NI_A); \
            const bool isB = !isA && ((NW * i + NW - 1 < G::NI) ? true : (NW * i >= G::NI ? false : j < G::NI)); \
            const bool va = kval && valid[i] && (unsigned)(hb[i] + te.x) < (unsigned)p.Hi &&    \
                            (unsigned)(wb[i] + te.y) < (unsigned)p.Wi;                          \
            const unsigned long pa = (unsigned long)(xg + (off0[i] + te.z + c));                \
            const unsigned long pb = (unsigned long)(wg + (off0[i] + kk));                      \
            const unsigned long ps = isA ? (va ? pa : zero_src) : ((isB && kval && valid[i]) ? pb : zero_src); \
            glds16((const void*)ps, ring_base + (unsigned)(((st)*SS + j * 64) * 16));           \
        }                                                                                       \
        kk += BK;                                                                               \
        c += BK;                                                                                \
        while (c >= p.Cin) {                                                                    \
            c -= p.Cin;                                                                         \
            ++tap;                                                                              \
        }                                                                                       \
    } while (0)

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lane_off = (lane & 15) * 4 + ((lane >> 4) ^ swz(lane & 15));
    const int nk = (p.Ktot + BK - 1) / BK;

    // prologue: PD stages in flight
#pragma unroll
    for (int s = 0; s < PD; ++s)
        if (s < nk) VT_ISSUE_STAGE(s);

    int cur = 0;       // ring slot of K-step ks
    int nxt = PD % NS; // ring slot of K-step ks + PD
    for (int ks = 0; ks < nk; ++ks) {
        // retire exactly the stage about to be read; younger stages stay in flight
        const int younger = min(PD - 1, nk - 1 - ks);
        if (younger >= 2)
            vm_wait<2 * IT>();
        else if (younger == 1)
            vm_wait<IT>();
        else
            vm_wait<0>();
        __builtin_amdgcn_s_barrier();  // every wave's DMA of this stage landed; slot `nxt` is free
        asm volatile("" ::: "memory");  // the raw barrier does not order memory ops for the compiler
        if (ks + PD < nk) VT_ISSUE_STAGE(nxt);
        {
            const uint4* A = sStage + cur * SS + wm * TM * 4 + lane_off;
            const uint4* Bt = sStage + cur * SS + BM * 4 + wn * TN * 4 + lane_off;
            uint4 af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = A[i * 64];
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = Bt[j * 64];
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
        cur = (cur + 1 == NS) ? 0 : cur + 1;
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
#undef VT_ISSUE_STAGE
    __syncthreads();  // all waves done reading the ring (no DMA is outstanding: last wait was vmcnt(0))

    // ---- epilogue ------------------------------------------------------------
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
            if (lane < 16) {  // parked per row-wave, folded in a FIXED order below (LDS float atomics would not be)
                sStat[(wm * 2 + 0) * BN + col] = s;
                sStat[(wm * 2 + 1) * BN + col] = ss;
            }
        }
    }
    __syncthreads();

    if (stats && tid < 2 * BN) {
        const int which = tid / BN, col = tid % BN;
        const int n = tn * BN + col;
        if (n < p.Cout) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) acc += sStat[(w * 2 + which) * BN + col];
            vt_stat_add(p.stats, ((long)(tm % kStatReplicas) * 2 + which) * p.Cout + n, acc);
        }
    }

    constexpr int CPR = BN / EPC;  // 16-byte chunks per tile row
    const uint4* sOut4 = (const uint4*)sOut;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    const bool has_res = (p.flags & VT_CONV_RESIDUAL) != 0;
    auto out_pixel = [&](int m) -> long {
        if (p.dense_out) return m;
        const int b = m / HoWo;
        const int rem = m - b * HoWo;
        const int oi = rem / p.Wo;
        const int oj = rem - oi * p.Wo;
        return ((long)b * p.oH + (oi * p.oHs + p.oh0)) * p.oW + (oj * p.oWs + p.ow0);
    };
    if (has_res) {
        // residual rows in batches of four, every load of a batch in flight before its first use: loaded
        // unconditionally (chunks outside the tensor read its first 16 bytes), so nothing orders a load behind the
        // previous chunk's store -- a dependent load per chunk cost 70 us of an 80-channel YOLOv5x layer's 450
        constexpr int ITER = (BM * CPR + NT - 1) / NT, RB = 4;
#pragma unroll
        for (int b0 = 0; b0 < ITER; b0 += RB) {
            uint4 rr[RB];
            long oy[RB];
            bool ok[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int idx = tid + (b0 + u) * NT;
                const int row = idx / CPR, ch = idx % CPR;
                const int m = tm * BM + row;
                const int n = tn * BN + ch * EPC;
                ok[u] = b0 + u < ITER && idx < BM * CPR && m < p.M && n < p.Cout;
                const long po = ok[u] ? out_pixel(m) : 0;
                oy[u] = po * p.ldy + vt_out_col(p, n, p.ldy);
                rr[u] = *(const uint4*)(rg + (ok[u] ? po * p.ldr + vt_out_col(p, n, p.ldr) : 0l));
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                if (!ok[u]) continue;
                float fv[EPC], fr[EPC];
                VecIO<T>::unpack(sOut4[tid + (b0 + u) * NT], fv);
                VecIO<T>::unpack(rr[u], fr);
#pragma unroll
                for (int e = 0; e < EPC; ++e) fv[e] += fr[e];
                *(uint4*)(yg + oy[u]) = VecIO<T>::pack(fv);
            }
        }
        return;
    }
#pragma unroll 2
    for (int idx = tid; idx < BM * CPR; idx += NT) {
        const int row = idx / CPR, ch = idx % CPR;
        const int m = tm * BM + row;
        const int n = tn * BN + ch * EPC;
        if (m < p.M && n < p.Cout) *(uint4*)(yg + (out_pixel(m) * p.ldy + vt_out_col(p, n, p.ldy))) = sOut4[idx];
    }
}

template <typename T, int BM, int BN, int WM, int WN, int PD>
int launch(IgemmArgs& a, hipStream_t st) {
    using G = Geom<BM, BN, PD, WM * WN, WM>;
    constexpr int kHdrBytes = G::HDR;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.chunk = (a.tiles_m + 7) / 8;
    constexpr int outb = BM * BN * (int)sizeof(T);
    constexpr int smem = kHdrBytes + (G::STAGE_BYTES > outb ? G::STAGE_BYTES : outb);
    static_assert(smem <= 160 * 1024, "tile exceeds the 160 KiB LDS of a gfx950 CU");
    const long blocks = (long)8 * a.chunk * a.tiles_n;
    if (blocks > 0x7fffffffL) {
        vt_set_error("vt_conv_igemm: grid too large (%ld blocks)", blocks);
        return VT_ERR_UNSUPPORTED;
    }
    // uniform-K fast path when a K-step never straddles two taps
    constexpr int BKe = 4 * (16 / (int)sizeof(T));
    auto kern = (a.Cin % BKe == 0) ? igemm_kernel<T, BM, BN, WM, WN, PD, true>
                                   : igemm_kernel<T, BM, BN, WM, WN, PD, false>;
    if (smem > 64 * 1024) {
        const int rc = vt_raise_dynamic_lds((const void*)kern, smem, "vt_conv_igemm");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("igemm_kernel<%s,%d,%d,%d,%d,%d,uk%d>", sizeof(T) == 2 ? "bf16" : "f32", BM, BN, WM, WN, PD,
                   (int)(a.Cin % BKe == 0));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WM * WN), smem, st, a);
    VT_CHECK_LAUNCH("vt_conv_igemm");
    return VT_OK;
}

}  // namespace

// the kernel argument block of a validated descriptor
static void fill_args(IgemmArgs& a, const vt_conv_desc* d, const void* x, const void* w, void* y, const float* scale,
                      const float* shift, const void* residual, float* stats) {
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.w = w;
    a.y = y;
    a.scale = scale;
    a.shift = shift;
    a.res = residual;
    a.stats = stats;
    a.B = d->B, a.Hi = d->Hi, a.Wi = d->Wi, a.Cin = d->Cin, a.ldx = d->ldx;
    a.Ho = d->Ho, a.Wo = d->Wo, a.sh = d->sh, a.sw = d->sw, a.h0 = d->h0, a.w0 = d->w0;
    a.Cout = d->Cout, a.ldy = d->ldy, a.oH = d->oH, a.oW = d->oW;
    a.oHs = d->oHs, a.oWs = d->oWs, a.oh0 = d->oh0, a.ow0 = d->ow0;
    a.ldw = d->ldw, a.ldr = d->ldr, a.flags = d->flags, a.ntaps = d->ntaps;
    a.M = d->B * d->Ho * d->Wo;
    a.Ktot = d->ntaps * d->Cin;
    a.dense_out = (d->oHs == 1 && d->oWs == 1 && d->oh0 == 0 && d->ow0 == 0 && d->oH == d->Ho && d->oW == d->Wo);
    memcpy(a.dh, d->dh, VT_MAX_TAPS);
    memcpy(a.dw, d->dw, VT_MAX_TAPS);

}

extern "C" int vt_conv_igemm(const vt_conv_desc* d, const void* x, const void* w, void* y,
                             const float* scale, const float* shift, const void* residual,
                             float* stats, void* stream) {
    VT_REQUIRE(d && x && w && (y || (d->flags & VT_CONV_NOSTORE)), VT_ERR_INVALID, "vt_conv_igemm: null argument");
    VT_REQUIRE(!(d->flags & VT_CONV_NOSTORE) || (d->flags & VT_CONV_STATS), VT_ERR_INVALID,
               "vt_conv_igemm: NOSTORE is the statistics-only pass");
    VT_REQUIRE(d->dtype == VT_F32 || d->dtype == VT_BF16, VT_ERR_UNSUPPORTED,
               "vt_conv_igemm: dtype %d", d->dtype);
    VT_REQUIRE(!(d->flags & VT_CONV_BNRED), VT_ERR_INVALID, "vt_conv_igemm: unknown flag bits 0x%x", d->flags);
    const int epc = vt_epc(d->dtype);
    VT_REQUIRE(d->ntaps >= 1 && d->ntaps <= VT_MAX_TAPS, VT_ERR_UNSUPPORTED,
               "vt_conv_igemm: ntaps %d outside [1,%d]", d->ntaps, VT_MAX_TAPS);
    VT_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && d->Cin > 0 && d->Cout > 0,
               VT_ERR_INVALID, "vt_conv_igemm: non-positive extent");
    VT_REQUIRE(d->Cin % epc == 0 && d->Cout % epc == 0 && d->ldx % epc == 0 && d->ldy % epc == 0 &&
                   d->ldw % epc == 0,
               VT_ERR_UNSUPPORTED,
               "vt_conv_igemm: Cin=%d Cout=%d ldx=%d ldy=%d ldw=%d must be multiples of %d", d->Cin,
               d->Cout, d->ldx, d->ldy, d->ldw, epc);
    const bool d2s = (d->flags & VT_CONV_D2S) != 0;
    if (d2s) {
        VT_REQUIRE(d->Cout % (4 * epc) == 0 && d->oHs == 2 && d->oWs == 2 && d->oh0 == 0 && d->ow0 == 0 &&
                       d->oH == 2 * d->Ho && d->oW == 2 * d->Wo &&
                       !(d->flags & (VT_CONV_STATS | VT_CONV_AFFINE | VT_CONV_RELU)),
                   VT_ERR_UNSUPPORTED, "vt_conv_igemm: D2S needs Cout = 4*C', a 2x2-strided output and a raw epilogue");
    }
    VT_REQUIRE(d->ldx >= d->Cin && d->ldy >= (d2s ? d->Cout / 4 : d->Cout) && d->ldw >= d->ntaps * d->Cin, VT_ERR_INVALID,
               "vt_conv_igemm: stride smaller than extent");
    VT_REQUIRE(vt_aligned16(x) && vt_aligned16(w) && vt_aligned16(y), VT_ERR_INVALID,
               "vt_conv_igemm: x/w/y must be 16-byte aligned");
    VT_REQUIRE(d->oHs >= 1 && d->oWs >= 1 && d->oh0 >= 0 && d->ow0 >= 0 &&
                   (d->Ho - 1) * d->oHs + d->oh0 < d->oH && (d->Wo - 1) * d->oWs + d->ow0 < d->oW,
               VT_ERR_INVALID, "vt_conv_igemm: output placement outside the %dx%d tensor", d->oH, d->oW);
    const long in_elems = (long)d->B * d->Hi * d->Wi * d->ldx;
    const long out_elems = (long)d->B * d->oH * d->oW * d->ldy;
    VT_REQUIRE(in_elems < 0x7fffffffL && out_elems < 0x7fffffffL && (long)d->B * d->Ho * d->Wo < 0x7fffffffL,
               VT_ERR_UNSUPPORTED, "vt_conv_igemm: tensor exceeds 2^31 elements");
    if (d->flags & VT_CONV_RESIDUAL) {
        VT_REQUIRE(residual && vt_aligned16(residual) && d->ldr % epc == 0 && d->ldr >= (d2s ? d->Cout / 4 : d->Cout),
                   VT_ERR_INVALID, "vt_conv_igemm: bad residual");
    }
    if (d->flags & VT_CONV_AFFINE) VT_REQUIRE(shift, VT_ERR_INVALID, "vt_conv_igemm: AFFINE needs shift");
    if (d->flags & VT_CONV_STATS) {
        VT_REQUIRE(stats, VT_ERR_INVALID, "vt_conv_igemm: STATS needs a stats buffer");
        VT_REQUIRE(!(d->flags & (VT_CONV_AFFINE | VT_CONV_RELU | VT_CONV_RESIDUAL)), VT_ERR_UNSUPPORTED,
                   "vt_conv_igemm: STATS is only defined on the raw conv output");
    }

    IgemmArgs a;
    fill_args(a, d, x, w, y, scale, shift, residual, stats);

    hipStream_t st = (hipStream_t)stream;
    if (!d2s) {  // (these kernels write dense rows only)
        const int rc = vt_stem_dispatch(a, d->dtype, stream);  // RGB stem
        if (rc >= 0) return rc;
        const int rc6 = vt_stem6_dispatch(a, d->dtype, stream);  // RGB stem of the YOLOv5 Darknets (6x6 stride 2)
        if (rc6 >= 0) return rc6;
    }
    VT_REQUIRE(!(d->flags & VT_CONV_NOSTORE), VT_ERR_UNSUPPORTED, "vt_conv_igemm: NOSTORE outside the RGB stem kernel");
    if (!d2s) {  // (these kernels write dense rows only)
        const int rc = vt_span6_dispatch(a, d->dtype, stream);  // MFMA-bound 3x3 stride-1 layers: two-group + loader kernel
        if (rc >= 0) return rc;
    }
    {
        const int rc = vt_pspan_dispatch(a, d->dtype, stream);  // short-K, HBM-bound convs: persistent resident-filter kernel
        if (rc >= 0) return rc;
    }
    {
        const int rc = vt_span_dispatch(a, d->dtype, stream);  // stride-1-grid convs: input-span kernel
        if (rc >= 0) return rc;
    }
    if (d->dtype == VT_BF16) {
        // 96- and 64-row tiles (to dodge the "2.04 rounds" wave quantisation of 128-row tiles at
        // batch 256) measured 8-15 % SLOWER on the CSPDarknet-53 layers: the extra filter
        // staging per FLOP costs more than the idle tail saves.  One tile height.
        // 80-wide filter tiles (5 fragments per wave, four row waves) where 128-wide ones would be 3/8 empty: the 80-
        // and 160-channel layers of Darknet-YOLOv5x (80 -> 80 3x3 @160x160: 559 -> 420 us, the 6x6 stem 1316 -> 834 us;
        // a 128-row tile of the same width measured 15-40 % slower).  VT_IGEMM_BN80=0: off
        // VT_IGEMM_W8: 8-wave workgroups on 256-row tiles (bit 0: 160 columns for Cout = 160; bit 1: 128 columns for
        // every Cout > 64) -- the gathered rows are staged once per 160 / 128 filter columns instead of once per 80 / per
        // 128 rows of half the height
        const int w8 = VT_KNOB("VT_IGEMM_W8", 3);
        if (VT_KNOB("VT_IGEMM_BN80", 1) && d->Cout % 80 == 0 && d->Cout <= 160) {
            if ((w8 & 1) && d->Cout == 160) return launch<bf16_t, 256, 160, 4, 2, 2>(a, st);
            return launch<bf16_t, 256, 80, 4, 1, 2>(a, st);
        }
        // (maps too small to give every CU a 256-row tile keep the 128-row one: 512 -> 512 @7x7 at batch 256)
        if (d->Cout > 64 && (w8 & 2) && ((long)(a.M + 255) / 256) * ((d->Cout + 127) / 128) >= ((w8 & 4) ? 0 : 256))
            return launch<bf16_t, 256, 128, 4, 2, 2>(a, st);
        if (d->Cout > 64) return launch<bf16_t, 128, 128, 2, 2, 2>(a, st);
        if (d->Cout > 32) return launch<bf16_t, 128, 64, 2, 2, 2>(a, st);
        return launch<bf16_t, 256, 32, 4, 1, 2>(a, st);
    }
    if (d->Cout > 32) return launch<float, 128, 64, 2, 2, 2>(a, st);
    return launch<float, 128, 32, 4, 1, 2>(a, st);
}

// A data gradient that also reduces the BatchNorm backward of the unit whose output it differentiates (round 6): see
// include/vt_amd.h.  Fused where the two-group persistent kernel takes the launch (its epilogue holds d(y) in registers
// and reads the unit's z like a residual operand); everywhere else the same two launches as before.
extern "C" int vt_conv_dgrad_bnred(const vt_conv_desc* d, const void* dz, const void* w, void* dy, const void* z, int32_t ldz,
                                   const float* scale, const float* shift, const float* mean, const float* invstd,
                                   int32_t relu, float* sums, void* stream) {
    VT_REQUIRE(d && dz && w && dy && z && scale && shift && mean && invstd && sums, VT_ERR_INVALID,
               "vt_conv_dgrad_bnred: null argument");
    VT_REQUIRE(relu == 0 || relu == 1, VT_ERR_UNSUPPORTED, "vt_conv_dgrad_bnred: activation code %d (none / ReLU only)", relu);
    VT_REQUIRE(d->flags == 0, VT_ERR_INVALID, "vt_conv_dgrad_bnred: the launch has a plain epilogue (flags 0x%x)", d->flags);
    const long M = (long)d->B * d->oH * d->oW;
    const bool whole = d->oHs == 1 && d->oWs == 1 && d->oh0 == 0 && d->ow0 == 0 && d->oH == d->Ho && d->oW == d->Wo;
    VT_REQUIRE(whole, VT_ERR_INVALID, "vt_conv_dgrad_bnred: the launch must produce every pixel of d(y)");
    const int epc = vt_epc(d->dtype);
    VT_REQUIRE(ldz % epc == 0 && ldz >= d->Cout && vt_aligned16(z), VT_ERR_INVALID, "vt_conv_dgrad_bnred: bad z");
    if (d->dtype == VT_BF16 && VT_KNOB("VT_DGRAD_BNRED", 1) && d->ntaps >= 1 && d->ntaps <= VT_MAX_TAPS && d->Cin % epc == 0 &&
        d->Cout % epc == 0 && d->ldx % epc == 0 && d->ldy % epc == 0 && d->ldw % epc == 0 && d->ldx >= d->Cin &&
        d->ldy >= d->Cout && d->ldw >= d->ntaps * d->Cin && vt_aligned16(dz) && vt_aligned16(w) && vt_aligned16(dy) &&
        (long)d->B * d->Hi * d->Wi * d->ldx < 0x7fffffffL && M * d->ldy < 0x7fffffffL && M * ldz < 0x7fffffffL) {
        IgemmArgs a;
        fill_args(a, d, dz, w, dy, scale, shift, z, sums);
        a.ldr = ldz;
        a.aux0 = mean, a.aux1 = invstd;
        a.flags = VT_CONV_BNRED | (relu ? VT_CONV_RELU : 0);
        const int rc = vt_span6_dispatch(a, d->dtype, stream);
        if (rc >= 0) return rc;
    }
    const int rc = vt_conv_igemm(d, dz, w, dy, nullptr, nullptr, nullptr, nullptr, stream);
    if (rc != VT_OK) return rc;
    return vt_bn_act_bwd_reduce(dy, d->ldy, z, ldz, scale, shift, mean, invstd, M, d->Cout, relu, d->dtype, sums, stream);
}
