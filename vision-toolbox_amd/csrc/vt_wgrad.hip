// vt_wgrad.hip -- convolution filter gradient on MFMA for gfx950 (MI355X).
//
//   dw[n][k=(t,c)] += sum_{m in pixels} dz[m][n] * xg[m][k]
//   xg[m][(t,c)]   = x(b, i*sh+h0+dh[t], j*sw+w0+dw[t], c)  (the forward gather, 0 outside)
//
// This is the autograd backward of the nn.Conv2d inside ConvNormAct
// (reference vision_toolbox/components.py:26-35) with respect to its weight.
//
// GEMM view: rows = Cout, cols = ntaps*Cin, contraction = pixels.  NHWC keeps
// channels contiguous, so BOTH operands arrive "K-strided"; they are staged
// pixel-major in LDS exactly as they sit in HBM (whole 256/512-byte rows,
// fully coalesced) and the MFMA fragments are formed by the hardware transpose
// read ds_read_b64_tr_b16 (bf16) or by plain ds_read_b32 (f32).  Rows are
// padded by 16 elements so both read patterns are bank-conflict free.
//
// The pixel range is split over blockIdx.y; partial tiles are combined with
// f32 global atomics straight into the weight's .grad storage (which is
// [Cout][taps][Cin], the channels_last image of the OIHW gradient).
#include "vt_common.h"

namespace {

struct WgradArgs {
    const void* x;
    const void* dz;
    float* dw;
    int B, Hi, Wi, Cin, ldx, Ho, Wo, sh, sw, h0, w0, Cout, ldy, ntaps;
    int M, Ktot, ldgw, tiles_n, tiles_k, chunk;
    int8_t dh[VT_MAX_TAPS];
    int8_t dwv[VT_MAX_TAPS];
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <typename T>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs p) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PK = 4 * EPC;         // pixels per step: 32 bf16 / 16 f32
    constexpr int LDT = 128 + 16;       // padded LDS row, elements
    constexpr int CPRW = 128 / EPC;     // 16-byte chunks per row
    constexpr int RPP = 256 / CPRW;     // rows staged per pass
    constexpr int NPASS = PK / RPP;     // == 2
    constexpr int TILE = PK * LDT;      // elements per tile
    static_assert(NPASS * RPP == PK, "staging shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int2* sTap = (int2*)smem;                          // 36 * 8 B
    T* sT = (T*)(smem + VT_MAX_TAPS * 8);              // [2 buf][2 (dz,x)][TILE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x % p.tiles_n, tile_k = blockIdx.x / p.tiles_n;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int m_begin = blockIdx.y * p.chunk;
    const int m_end = min(p.M, m_begin + p.chunk);
    if (m_begin >= m_end) return;

    if (tid < p.ntaps) sTap[tid] = make_int2(p.dh[tid], p.dwv[tid]);
    __syncthreads();

    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ zg = (const T*)p.dz;

    // ---- staging geometry ----------------------------------------------------
    const int col = tid % CPRW;
    const int nn = n0 + col * EPC;
    const bool nvalid = nn < p.Cout;
    const int kc = k0 + col * EPC;
    const bool kvalid = kc < p.Ktot;
    const int tap = kvalid ? kc / p.Cin : 0;
    const int cc = kc - tap * p.Cin;
    const int2 dd = sTap[tap];

    int mrow[NPASS], pb[NPASS], pi[NPASS], pj[NPASS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int m = m_begin + tid / CPRW + i * RPP;
        mrow[i] = m;
        const int mm = min(m, p.M - 1);
        pb[i] = mm / HoWo;
        const int rem = mm - pb[i] * HoWo;
        pi[i] = rem / p.Wo;
        pj[i] = rem - pi[i] * p.Wo;
    }

    uint4 rz[NPASS], rx[NPASS];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const bool mv = mrow[i] < m_end;
            rz[i] = zero4;
            rx[i] = zero4;
            if (mv && nvalid) rz[i] = *(const uint4*)(zg + ((long)mrow[i] * p.ldy + nn));
            const int hi = pi[i] * p.sh + p.h0 + dd.x;
            const int wi = pj[i] * p.sw + p.w0 + dd.y;
            if (mv && kvalid && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi)
                rx[i] = *(const uint4*)(xg + (((long)(pb[i] * p.Hi + hi) * p.Wi + wi) * p.ldx + cc));
            // advance this row by PK pixels
            mrow[i] += PK;
            pj[i] += PK;
            while (pj[i] >= p.Wo) {
                pj[i] -= p.Wo;
                pi[i] += 1;
            }
            while (pi[i] >= p.Ho) {
                pi[i] -= p.Ho;
                pb[i] += 1;
            }
        }
    };
    auto lds_store = [&](int buf) {
        T* tz = sT + (buf * 2 + 0) * TILE;
        T* tx = sT + (buf * 2 + 1) * TILE;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int r = tid / CPRW + i * RPP;
            *(uint4*)(tz + r * LDT + col * EPC) = rz[i];
            *(uint4*)(tx + r * LDT + col * EPC) = rx[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, u = lane & 15;
    const int nsteps = (m_end - m_begin + PK - 1) / PK;

    gload();
    lds_store(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        const bool more = s + 1 < nsteps;
        if (more) gload();
        const T* tz = sT + (cur * 2 + 0) * TILE + wm * 64;
        const T* tx = sT + (cur * 2 + 1) * TILE + wn * 64;
        if constexpr (sizeof(T) == 2) {
            // lane 4q+pp of a 16-lane group addresses row q, columns 4pp..4pp+3 of a
            // 4 x 16 block; it receives column u for the block's 4 rows (pixels).
            const int q = u >> 2, pp = u & 3;
            const int ro = (4 * g + q) * LDT + 4 * pp;
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tz + ro + i * 16));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tz + ro + 16 * LDT + i * 16));
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tx + ro + j * 16));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tx + ro + 16 * LDT + j * 16));
                bf[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const int ro = (4 * ss + g) * LDT + u;
                float af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = (float)tz[ro + i * 16];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = (float)tx[ro + j * 16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) lds_store(cur ^ 1);
        __syncthreads();
    }

    // ---- combine: f32 atomics into dw[n][k] -----------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wn * 64 + j * 16 + u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wm * 64 + i * 16 + 4 * g + r;
                if (n < p.Cout && k < p.Ktot) atomicAdd(p.dw + ((long)n * p.ldgw + k), acc[i][j][r]);
            }
        }
    }
}

}  // namespace

extern "C" int vt_conv_wgrad(const vt_conv_desc* d, const void* x, const void* dz, float* dw,
                             int32_t ldgw, void* stream) {
    VT_REQUIRE(d && x && dz && dw, VT_ERR_INVALID, "vt_conv_wgrad: null argument");
    VT_REQUIRE(d->dtype == VT_F32 || d->dtype == VT_BF16, VT_ERR_UNSUPPORTED, "vt_conv_wgrad: dtype %d",
               d->dtype);
    const int epc = vt_epc(d->dtype);
    VT_REQUIRE(d->ntaps >= 1 && d->ntaps <= VT_MAX_TAPS, VT_ERR_UNSUPPORTED, "vt_conv_wgrad: ntaps %d",
               d->ntaps);
    VT_REQUIRE(d->Cin % epc == 0 && d->Cout % epc == 0 && d->ldx % epc == 0 && d->ldy % epc == 0,
               VT_ERR_UNSUPPORTED, "vt_conv_wgrad: Cin=%d Cout=%d ldx=%d ldy=%d must be multiples of %d",
               d->Cin, d->Cout, d->ldx, d->ldy, epc);
    VT_REQUIRE(d->oHs == 1 && d->oWs == 1 && d->oh0 == 0 && d->ow0 == 0 && d->oH == d->Ho && d->oW == d->Wo,
               VT_ERR_UNSUPPORTED, "vt_conv_wgrad: dz must be dense over the output grid");
    VT_REQUIRE(ldgw >= d->ntaps * d->Cin, VT_ERR_INVALID, "vt_conv_wgrad: ldgw %d < K %d", ldgw,
               d->ntaps * d->Cin);
    VT_REQUIRE(vt_aligned16(x) && vt_aligned16(dz), VT_ERR_INVALID, "vt_conv_wgrad: x/dz must be 16-byte aligned");
    const long in_elems = (long)d->B * d->Hi * d->Wi * d->ldx;
    const long M = (long)d->B * d->Ho * d->Wo;
    VT_REQUIRE(in_elems < 0x7fffffffL && M * d->ldy < 0x7fffffffL, VT_ERR_UNSUPPORTED,
               "vt_conv_wgrad: tensor exceeds 2^31 elements");

    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x, a.dz = dz, a.dw = dw;
    a.B = d->B, a.Hi = d->Hi, a.Wi = d->Wi, a.Cin = d->Cin, a.ldx = d->ldx;
    a.Ho = d->Ho, a.Wo = d->Wo, a.sh = d->sh, a.sw = d->sw, a.h0 = d->h0, a.w0 = d->w0;
    a.Cout = d->Cout, a.ldy = d->ldy, a.ntaps = d->ntaps;
    a.M = (int)M;
    a.Ktot = d->ntaps * d->Cin;
    a.ldgw = ldgw;
    a.tiles_n = (d->Cout + 127) / 128;
    a.tiles_k = (a.Ktot + 127) / 128;
    memcpy(a.dh, d->dh, VT_MAX_TAPS);
    memcpy(a.dwv, d->dw, VT_MAX_TAPS);

    const int pk = 4 * epc;
    const long tiles = (long)a.tiles_n * a.tiles_k;
    long split = (1024 + tiles - 1) / tiles;           // aim for ~4 workgroups per CU
    const long max_split = (M + 8L * pk - 1) / (8L * pk);  // at least 8 steps per workgroup
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    long chunk = (M + split - 1) / split;
    chunk = (chunk + pk - 1) / pk * pk;
    split = (M + chunk - 1) / chunk;
    a.chunk = (int)chunk;

    const int smem = VT_MAX_TAPS * 8 + 2 * 2 * pk * (128 + 16) * vt_elem_size(d->dtype);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)tiles, (unsigned)split);
    if (d->dtype == VT_BF16)
        hipLaunchKernelGGL(wgrad_kernel<bf16_t>, grid, dim3(256), smem, st, a);
    else
        hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), smem, st, a);
    VT_CHECK_LAUNCH("vt_conv_wgrad");
    return VT_OK;
}
