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
// steps of 64 bytes per row (32 bf16 / 16 f32).  Both operands are staged
// global -> VGPR -> LDS in 16-byte chunks (one NHWC pixel contributes 16 B of
// consecutive channels, so HBM/L2 reads are whole 64-byte row segments), LDS is
// double buffered with ONE barrier per K-step and the next step's global loads
// are issued before the current step's MFMAs.  The LDS image is XOR-swizzled so
// that both the ds_write_b128 of staging and the ds_read_b128 of the MFMA
// fragments are bank-conflict free (see swz()).
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
#include "vt_common.h"

namespace {

struct IgemmArgs {
    const void* x;
    const void* w;
    void* y;
    const float* scale;
    const float* shift;
    const void* res;
    float* stats;
    int B, Hi, Wi, Cin, ldx, Ho, Wo, sh, sw, h0, w0, Cout, ldy, oH, oW, oHs, oWs, oh0, ow0;
    int ldw, ldr, flags, ntaps;
    int M, Ktot, tiles_m, tiles_n, chunk, dense_out;
    int8_t dh[VT_MAX_TAPS];
    int8_t dw[VT_MAX_TAPS];
};

constexpr int kTapBytes = VT_MAX_TAPS * 16;  // int4 per tap
constexpr int kStatBytes = 2 * 128 * 4;      // sum, sumsq for BN <= 128
constexpr int kHdrBytes = kTapBytes + kStatBytes;

// chunk swizzle of a 64-byte LDS row: conflict free for ds_read_b128 issued as
// (row = l&15, chunk = l>>4) and for ds_write_b128 issued as (row = t>>2, chunk = t&3).
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

template <typename T>
__device__ __forceinline__ uint4 ldg_pred(const T* ptr, bool valid) {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (valid) r = *(const uint4*)ptr;
    return r;
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

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64) igemm_kernel(const IgemmArgs p) {
    constexpr int NT = WM * WN * 64;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 4 * EPC;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    constexpr int A_IT = (BM * 4 + NT - 1) / NT;
    constexpr int B_IT = (BN * 4 + NT - 1) / NT;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be 16-granular");
    static_assert(BN <= 128, "stat scratch sized for BN <= 128");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* sTap = (int4*)smem;
    float* sStat = (float*)(smem + kTapBytes);
    uint4* sA = (uint4*)(smem + kHdrBytes);  // [2][BM*4]
    uint4* sB = sA + 2 * BM * 4;             // [2][BN*4]
    T* sOut = (T*)(smem + kHdrBytes);        // [BM][BN], aliases staging after the K loop

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
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

    // ---- per-thread staging geometry (fixed for the whole K loop) ----------
    const int cj = tid & 3;
    int hb[A_IT], wb[A_IT];
    long boff[A_IT];
    bool rvalid[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = (tid >> 2) + i * (NT / 4);
        const int m = tm * BM + row;
        rvalid[i] = (row < BM) && (m < p.M);
        const int mm = rvalid[i] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int oi = rem / p.Wo;
        const int oj = rem - oi * p.Wo;
        hb[i] = oi * p.sh + p.h0;
        wb[i] = oj * p.sw + p.w0;
        boff[i] = ((long)(b * p.Hi + hb[i]) * p.Wi + wb[i]) * p.ldx;
    }
    long woff[B_IT];
    bool nvalid[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = (tid >> 2) + i * (NT / 4);
        const int n = tn * BN + row;
        nvalid[i] = (row < BN) && (n < p.Cout);
        woff[i] = (long)(nvalid[i] ? n : 0) * p.ldw;
    }

    int kk = cj * EPC;  // flattened K index of this thread's chunk
    int tap = kk / p.Cin;
    int c = kk - tap * p.Cin;

    uint4 ra[A_IT], rb[B_IT];

    // (macros, not lambdas: by-reference captures of ra/rb would pin them in scratch)
#define VT_GLOAD()                                                                              \
    do {                                                                                        \
        const bool kval = tap < p.ntaps;                                                        \
        const int4 te = sTap[kval ? tap : 0];                                                   \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                      \
            const bool v = kval && rvalid[i] && (unsigned)(hb[i] + te.x) < (unsigned)p.Hi &&    \
                           (unsigned)(wb[i] + te.y) < (unsigned)p.Wi;                           \
            ra[i] = ldg_pred(xg + (boff[i] + te.z + c), v);                                     \
        }                                                                                       \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                        \
            rb[i] = ldg_pred(wg + (woff[i] + kk), kval && nvalid[i]);                           \
        kk += BK;                                                                               \
        c += BK;                                                                                \
        while (c >= p.Cin) {                                                                    \
            c -= p.Cin;                                                                         \
            ++tap;                                                                              \
        }                                                                                       \
    } while (0)
#define VT_LDS_STORE(buf)                                                                       \
    do {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                      \
            const int row = (tid >> 2) + i * (NT / 4);                                          \
            if (row < BM) sA[(buf)*BM * 4 + row * 4 + (cj ^ swz(row))] = ra[i];                 \
        }                                                                                       \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                      \
            const int row = (tid >> 2) + i * (NT / 4);                                          \
            if (row < BN) sB[(buf)*BN * 4 + row * 4 + (cj ^ swz(row))] = rb[i];                 \
        }                                                                                       \
    } while (0)

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lane_off = (lane & 15) * 4 + ((lane >> 4) ^ swz(lane & 15));
    const int nk = (p.Ktot + BK - 1) / BK;

    VT_GLOAD();
    VT_LDS_STORE(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < nk;
        if (more) VT_GLOAD();
        {
            const uint4* A = sA + cur * BM * 4 + wm * TM * 4 + lane_off;
            const uint4* Bt = sB + cur * BN * 4 + wn * TN * 4 + lane_off;
            uint4 af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = A[i * 64];
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = Bt[j * 64];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma<T>(af[i], bf[j], acc[i][j]);
        }
        if (more) VT_LDS_STORE(cur ^ 1);
        __syncthreads();
    }

#undef VT_GLOAD
#undef VT_LDS_STORE
    // ---- epilogue ------------------------------------------------------------
    // (the trailing barrier of the loop guarantees every wave is done with staging)
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

    if (stats && tid < 2 * BN) {
        const int which = tid / BN, col = tid % BN;
        const int n = tn * BN + col;
        if (n < p.Cout) {
            const int rep = tm % VT_STAT_REPLICAS;
            atomicAdd(&p.stats[((long)rep * 2 + which) * p.Cout + n], sStat[tid]);
        }
    }

    constexpr int CPR = BN / EPC;  // 16-byte chunks per tile row
    const uint4* sOut4 = (const uint4*)sOut;
    T* __restrict__ yg = (T*)p.y;
    const T* __restrict__ rg = (const T*)p.res;
    const bool has_res = (p.flags & VT_CONV_RESIDUAL) != 0;
#pragma unroll 2
    for (int idx = tid; idx < BM * CPR; idx += NT) {
        const int row = idx / CPR, ch = idx % CPR;
        const int m = tm * BM + row;
        const int n = tn * BN + ch * EPC;
        if (m < p.M && n < p.Cout) {
            long po = m;
            if (!p.dense_out) {
                const int b = m / HoWo;
                const int rem = m - b * HoWo;
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

template <typename T, int BM, int BN, int WM, int WN>
int launch(IgemmArgs& a, hipStream_t st) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.chunk = (a.tiles_m + 7) / 8;
    constexpr int stage = 2 * (BM + BN) * 64;
    constexpr int outb = BM * BN * (int)sizeof(T);
    constexpr int smem = kHdrBytes + (stage > outb ? stage : outb);
    static_assert(smem <= 64 * 1024, "tile needs more than the default 64 KiB of LDS");
    const long blocks = (long)8 * a.chunk * a.tiles_n;
    if (blocks > 0x7fffffffL) {
        vt_set_error("vt_conv_igemm: grid too large (%ld blocks)", blocks);
        return VT_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN>), dim3((unsigned)blocks), dim3(WM * WN * 64), smem,
                       st, a);
    VT_CHECK_LAUNCH("vt_conv_igemm");
    return VT_OK;
}

}  // namespace

extern "C" int vt_conv_igemm(const vt_conv_desc* d, const void* x, const void* w, void* y,
                             const float* scale, const float* shift, const void* residual,
                             float* stats, void* stream) {
    VT_REQUIRE(d && x && w && y, VT_ERR_INVALID, "vt_conv_igemm: null argument");
    VT_REQUIRE(d->dtype == VT_F32 || d->dtype == VT_BF16, VT_ERR_UNSUPPORTED,
               "vt_conv_igemm: dtype %d", d->dtype);
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
    VT_REQUIRE(d->ldx >= d->Cin && d->ldy >= d->Cout && d->ldw >= d->ntaps * d->Cin, VT_ERR_INVALID,
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
        VT_REQUIRE(residual && vt_aligned16(residual) && d->ldr % epc == 0 && d->ldr >= d->Cout,
                   VT_ERR_INVALID, "vt_conv_igemm: bad residual");
    }
    if (d->flags & VT_CONV_AFFINE) VT_REQUIRE(shift, VT_ERR_INVALID, "vt_conv_igemm: AFFINE needs shift");
    if (d->flags & VT_CONV_STATS) {
        VT_REQUIRE(stats, VT_ERR_INVALID, "vt_conv_igemm: STATS needs a stats buffer");
        VT_REQUIRE(!(d->flags & (VT_CONV_AFFINE | VT_CONV_RELU | VT_CONV_RESIDUAL)), VT_ERR_UNSUPPORTED,
                   "vt_conv_igemm: STATS is only defined on the raw conv output");
    }

    IgemmArgs a;
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

    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == VT_BF16) {
        if (d->Cout > 64) return launch<bf16_t, 128, 128, 2, 2>(a, st);
        if (d->Cout > 32) return launch<bf16_t, 128, 64, 2, 2>(a, st);
        return launch<bf16_t, 256, 32, 4, 1>(a, st);
    }
    if (d->Cout > 32) return launch<float, 128, 64, 2, 2>(a, st);
    return launch<float, 128, 32, 4, 1>(a, st);
}
