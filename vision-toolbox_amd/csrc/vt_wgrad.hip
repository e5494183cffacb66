// vt_wgrad.hip -- convolution filter gradient on MFMA for gfx950 (MI355X).
//
//   dw[n][k=(t,c)] += sum_{m in pixels} dz[m][n] * xg[m][k]
//   xg[m][(t,c)]   = x(b, i*sh+h0+dh[t], j*sw+w0+dw[t], c)  (the forward gather, 0 outside)
//
// This is the autograd backward of the nn.Conv2d inside ConvNormAct
// (reference vision_toolbox/components.py:26-35) with respect to its weight.
//
// GEMM view: rows = Cout, cols = ntaps*Cin, contraction = pixels.  NHWC keeps
// channels contiguous, so BOTH operands arrive "K-strided" for MFMA; they are
// staged pixel-major in LDS exactly as they sit in HBM -- whole 256-byte (bf16) /
// 512-byte (f32) channel rows, i.e. full cache lines -- by an LDS-DMA ring
// (global_load_lds_dwordx4, no VGPR round trip, PD stages in flight across the
// single barrier per step, counted s_waitcnt vmcnt; same scheme as vt_igemm.hip),
// and the MFMA fragments are formed by the hardware transposing read
// ds_read_b64_tr_b16 (bf16) or by plain ds_read_b32 (f32).
// An LDS-DMA writes 1 KiB contiguously, so rows cannot be padded; bank conflicts are
// removed by XOR-swizzling the 16-byte chunk index with the row number through the
// SOURCE address: chunk ^= 2*(row&7) for bf16 (the 8 rows a half-wave's transposing
// read touches land on 8 distinct 32-byte bank groups), chunk ^= 4*(row&1) for f32.
// Out-of-image taps / tails read from a 16-byte zero page.
//
// The pixel range is split over blockIdx.y; partial tiles are combined with
// f32 global atomics straight into the weight's .grad storage (which is
// [Cout][taps][Cin], the channels_last image of the OIHW gradient).
#include <stdlib.h>

#include "vt_common.h"

namespace {

constexpr int kWgMaxGroup = 8;

struct WgradArgs {
    const void* x;
    const void* dz;
    float* dw;
    int B, Hi, Wi, Cin, ldx, Ho, Wo, sh, sw, h0, w0, Cout, ldy, ntaps;
    int M, Ktot, ldgw, tiles_n, tiles_k, chunk, ablate;
    int split, xcds;   // pixel splits; 8 = XCD-blocked item order (vt_xcd_item), 1 = identity
    float* slab;       // partial tiles go to slab[blockIdx.y * slab_stride + ...] with plain stores (no atomics)
    long slab_stride;  // elements per pixel split: Cout * ldgw
    int8_t dh[VT_MAX_TAPS];
    int8_t dwv[VT_MAX_TAPS];
    // grouped launch (round 5): nlayers > 1 same-shape layers, layer l = items [l * tiles * split, (l+1) * tiles * split);
    // x / dz / dw above are layer 0's
    int nlayers;
    const void* xs[kWgMaxGroup];
    const void* dzs[kWgMaxGroup];
    float* dws[kWgMaxGroup];
};

__device__ __attribute__((aligned(16))) unsigned int vt_wg_zero16[4];

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

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

constexpr int kTapHdr = VT_MAX_TAPS * 8;

// G groups of 4 waves share one output tile: group q consumes pixel steps q, q+G, q+2G, ...
// through its own LDS-DMA ring, and the G partial tiles are summed in LDS before ONE set of
// global atomics leaves the workgroup -- the atomic traffic of a launch is
// (#workgroups x 64 KiB), so more waves per workgroup means fewer atomic bytes per FLOP.
// kUnit: 1x1 stride-1 unpadded conv -- the gathered x row of output pixel m IS input pixel m, so the
// per-lane (image, row, column) tracking (two divergent `while` loops and ~60 VALU per step, against
// 16 MFMAs) disappears from the loop.
template <typename T, int G, int kPD, bool kUnit = false>
__global__ void __launch_bounds__(256 * G) wgrad_kernel(const WgradArgs p) {
    constexpr int kNS = kPD + 1;      // ring slots per group
    constexpr int EPC = 16 / sizeof(T);
    constexpr int PK = 4 * EPC;       // pixels per step: 32 bf16 / 16 f32
    constexpr int ROW = 128;          // channels per tile row (256 B bf16 / 512 B f32)
    constexpr int CPRW = ROW / EPC;   // 16-byte chunks per row: 16 / 32
    constexpr int RPI = 64 / CPRW;    // rows per DMA instruction: 4 / 2
    constexpr int TILE = PK * ROW;    // elements per operand tile (8 KiB)
    constexpr int STAGE = 2 * TILE;   // dz tile then x tile
    constexpr int IT = 4;             // DMA instructions per wave per stage (2 dz + 2 x)
    static_assert(PK / RPI == 8, "8 DMA instructions per operand tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int2* sTap = (int2*)smem;
    T* sRing = (T*)(smem + kTapHdr);  // [kNS][STAGE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int gwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = gwave >> 2;       // wave group (pixel-step phase)
    const int wave = gwave & 3;       // wave inside the group
    const int wm = wave >> 1, wn = wave & 1;
    // item = (pixel split, tile), tiles fastest: the workgroups of one XCD share a few pixel splits' rows of dz and x
    const unsigned item0 = vt_xcd_item(blockIdx.x, gridDim.x, p.xcds);
    const int ntile = p.tiles_n * p.tiles_k;
    const unsigned per_layer = (unsigned)(ntile * p.split);
    if (item0 >= per_layer * (unsigned)p.nlayers) return;
    const int layer = __builtin_amdgcn_readfirstlane((int)(item0 / per_layer));
    const unsigned item = item0 - (unsigned)layer * per_layer;
    const int bsplit = (int)(item / (unsigned)ntile), btile = (int)(item - (unsigned)bsplit * ntile);
    const int tile_n = btile % p.tiles_n, tile_k = btile / p.tiles_n;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int m_begin = bsplit * p.chunk;
    const int m_end = min(p.M, m_begin + p.chunk);
    if (m_begin >= m_end) return;

#pragma unroll
    for (int t = 0; t < VT_MAX_TAPS; ++t)
        if (t < p.ntaps && tid == t) sTap[t] = make_int2(p.dh[t], p.dwv[t]);
    __syncthreads();

    const T* __restrict__ xg = (const T*)(p.nlayers > 1 ? p.xs[layer] : p.x);
    const T* __restrict__ zg = (const T*)(p.nlayers > 1 ? p.dzs[layer] : p.dz);
    float* __restrict__ dwg = p.nlayers > 1 ? p.dws[layer] : p.dw;
    const unsigned long zero_src = (unsigned long)(const void*)vt_wg_zero16;
    const unsigned ring_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sRing +
                               (unsigned)(grp * kNS * STAGE * (int)sizeof(T));

    // ---- per-lane DMA geometry ------------------------------------------------------
    // operand tile: instruction j' (0..7) fills 16-byte slots [64j', 64j'+64) = rows
    // RPI*j' .. ; lane l owns row RPI*j' + l/CPRW, position l%CPRW, and fetches the chunk
    // (l%CPRW) ^ swz(row).  Wave w issues j' = w and w+4 for dz, then the same for x, so the
    // row parity that enters swz() is the same for all of a lane's loads.
    const int rloc = lane / CPRW;  // row inside the instruction
    int swz;
    if constexpr (sizeof(T) == 2)
        swz = 2 * ((RPI * wave + rloc) & 7);  // (4*(w+4) + r)&7 == (4w + r)&7
    else
        swz = 4 * (rloc & 1);                 // (2*j' + r)&1 == r&1
    const int col_e = ((lane % CPRW) ^ swz) * EPC;  // element column this lane fetches

    const int nn = n0 + col_e;
    const bool nvalid = nn < p.Cout;
    const int kc = k0 + col_e;
    const bool kvalid = kc < p.Ktot;
    const int tap = kvalid ? kc / p.Cin : 0;
    const int cc = kc - tap * p.Cin;
    const int2 dd = sTap[tap];

    // Per-lane stream state of the two DMA instructions a wave issues per operand and stage: the output pixel mrow, its
    // element offset in dz, and (gather form) the scaled coordinates hs = row * sh, ws = column * sw of the pixel with
    // the element offset of input pixel (b, hs, ws).  Everything advances by ADDITIONS: round 2 advanced (b, row, column)
    // by multiply-high quotients and rebuilt both addresses with 64-bit multiplies at every issue -- 23 quarter-rate
    // integer multiplies among ~80 vector instructions per wave and step against 16 MFMAs, and the loop was bound by
    // vector issue, not by the transposing reads (rocprofv3 PMC at 256 -> 256 @14x14: SQ_INSTS_VALU 24.8 M for 3.6 M
    // MFMAs, LDS array 16 % busy, no bank conflicts).
    constexpr int PKG = PK * G;
    const int WoS = p.Wo * p.sw, HoS = p.Ho * p.sh;
    const int qW = PKG / p.Wo, rW = PKG - qW * p.Wo;  // PKG pixels = qW rows + rW columns
    const int qH = qW / p.Ho, rH = qW - qH * p.Ho;    // qW rows   = qH images + rH rows
    const int adv_w = rW * p.sw, adv_h = rH * p.sh;
    const int d0 = ((qH * p.Hi + rH * p.sh) * p.Wi + rW * p.sw) * p.ldx;  // offset deltas: PKG pixels ahead
    const int d1 = (p.sh * p.Wi - WoS) * p.ldx;                           //   a column wrap
    const int d2 = (p.Hi - HoS) * p.Wi * p.ldx;                           //   a row wrap (next image)
    const int tap_h = p.h0 + dd.x, tap_w = p.w0 + dd.y;
    const int tap_off = (tap_h * p.Wi + tap_w) * p.ldx + cc;              // this lane's tap and channel
    const int zadv = PKG * p.ldy, xadv_unit = PKG * p.ldx;
    int mrow[2], hs[2], ws[2], xoff[2], zoff[2];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m_begin + grp * PK + RPI * (wave + 4 * i) + rloc;
        mrow[i] = m;
        zoff[i] = m * p.ldy + nn;  // (elements of dz < 2^31: vt_conv_wgrad)
        if constexpr (kUnit) {
            xoff[i] = m * p.ldx + cc;
            hs[i] = ws[i] = 0;
        } else {
            const int mm = min(m, p.M - 1);
            const int b_ = mm / HoWo;
            const int rem = mm - b_ * HoWo;
            const int i_ = rem / p.Wo, j_ = rem - i_ * p.Wo;
            hs[i] = i_ * p.sh, ws[i] = j_ * p.sw;
            xoff[i] = ((b_ * p.Hi + hs[i]) * p.Wi + ws[i]) * p.ldx;
        }
    }

#define VT_WG_ISSUE(st)                                                                           \
    do {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                           \
            const int jj = wave + 4 * i;                                                          \
            const bool mv = mrow[i] < m_end;                                                      \
            const unsigned long pz = (unsigned long)(zg + (unsigned)zoff[i]);                     \
            glds16((mv && nvalid) ? pz : zero_src, ring_base + (unsigned)((((st)*STAGE) * (int)sizeof(T)) + jj * 1024)); \
            zoff[i] += zadv;                                                                      \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                           \
            const int jj = wave + 4 * i;                                                          \
            if constexpr (kUnit) {                                                                \
                const bool xv = mrow[i] < m_end && kvalid;                                        \
                const unsigned long px = (unsigned long)(xg + (unsigned)xoff[i]);                 \
                glds16(xv ? px : zero_src,                                                        \
                       ring_base + (unsigned)((((st)*STAGE + TILE) * (int)sizeof(T)) + jj * 1024)); \
                mrow[i] += PKG;                                                                   \
                xoff[i] += xadv_unit;                                                             \
            } else {                                                                              \
                const bool xv = mrow[i] < m_end && kvalid && (unsigned)(hs[i] + tap_h) < (unsigned)p.Hi && \
                                (unsigned)(ws[i] + tap_w) < (unsigned)p.Wi;                       \
                const unsigned long px = (unsigned long)(xg + (unsigned)(xoff[i] + tap_off));     \
                glds16(xv ? px : zero_src,                                                        \
                       ring_base + (unsigned)((((st)*STAGE + TILE) * (int)sizeof(T)) + jj * 1024)); \
                mrow[i] += PKG;                                                                   \
                /* PKG pixels ahead: at most one column wrap, then at most one row wrap (rH + 1 <= Ho) */ \
                ws[i] += adv_w;                                                                   \
                const bool c1 = ws[i] >= WoS;                                                     \
                ws[i] -= c1 ? WoS : 0;                                                            \
                hs[i] += adv_h + (c1 ? p.sh : 0);                                                 \
                const bool c2 = hs[i] >= HoS;                                                     \
                hs[i] -= c2 ? HoS : 0;                                                            \
                xoff[i] += d0 + (c1 ? d1 : 0) + (c2 ? d2 : 0);                                    \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, u = lane & 15;
    // every group runs the same number of iterations (the barrier is workgroup wide); steps
    // past the end of the range stage zeros
    const int nsteps = ((m_end - m_begin + PK - 1) / PK + G - 1) / G;

#pragma unroll
    for (int s = 0; s < kPD; ++s)
        if (s < nsteps) VT_WG_ISSUE(s);

    int cur = 0, nxt = kPD % kNS;
    for (int s = 0; s < nsteps; ++s) {
        const int younger = min(kPD - 1, nsteps - 1 - s);
        if (younger >= 1)
            vm_wait<IT>();
        else
            vm_wait<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + kPD < nsteps) VT_WG_ISSUE(nxt);

        const T* tz = sRing + (grp * kNS + cur) * STAGE;
        const T* tx = tz + TILE;
        if constexpr (sizeof(T) == 2) {
            // lane 4q+pp of a 16-lane group addresses row q, columns 4pp..4pp+3 of a 4 x 16
            // block and receives column u for the block's 4 rows (pixels):
            // fragment element j<4 <-> pixel 4g+j, j>=4 <-> pixel 16+4g+(j-4) for BOTH operands.
            const int q = u >> 2, pp = u & 3;
            const int row = 4 * g + q;
            const int sw2 = 2 * (row & 7);  // == 2*((16+row)&7)
            const int sub = 4 * (pp & 1);   // element offset inside the 16-byte chunk
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = ((wm * 64 + i * 16) >> 3) + (pp >> 1);
                const T* a0 = tz + row * ROW + ((ch ^ sw2) << 3) + sub;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * ROW));
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = ((wn * 64 + j * 16) >> 3) + (pp >> 1);
                const T* b0 = tx + row * ROW + ((ch ^ sw2) << 3) + sub;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)b0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b0 + 16 * ROW));
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
                const int row = 4 * ss + g;
                const int sw4 = 4 * (row & 1);
                float af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = ((wm * 64 + i * 16 + u) >> 2) ^ sw4;
                    af[i] = (float)tz[row * ROW + (ch << 2) + (u & 3)];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = ((wn * 64 + j * 16 + u) >> 2) ^ sw4;
                    bf[j] = (float)tx[row * ROW + (ch << 2) + (u & 3)];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        cur = (cur + 1 == kNS) ? 0 : cur + 1;
        nxt = (nxt + 1 == kNS) ? 0 : nxt + 1;
    }
#undef VT_WG_ISSUE

    // ---- combine -------------------------------------------------------------------
    // partial tiles of the G groups are summed into an LDS image [128 n][128 k] (f32, the
    // ring is dead by now), then every wave issues row-contiguous 256-byte atomics.
    __syncthreads();
    float* sAcc = (float*)(smem + kTapHdr);
    constexpr int HALVES = (G == 1) ? 2 : 1;  // a 4-wave workgroup only owns 48 KiB: two passes of 64 rows
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
            if (grp == q && (HALVES == 1 || wm == h)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int nrow = (HALVES == 1 ? wm * 64 : 0) + i * 16 + 4 * g + r;
                            const int kcol = (wn * 64 + j * 16 + u) ^ (g << 4);  // spread the 4 rows of a wave-instr
                            float* dst = sAcc + nrow * 128 + kcol;
                            if (q == 0)
                                *dst = acc[i][j][r];
                            else
                                *dst += acc[i][j][r];
                        }
            }
            __syncthreads();
        }
        constexpr int ROWS = 128 / HALVES;
        // the pixel splits of a tile finish together and add to the same addresses: each starts at its own rows, so that
        // the adds of different splits do not queue up on one address at a time
        constexpr int NCHK = ROWS * 128 / (256 * G);
        const int rot = (int)(((unsigned)bsplit * 5u) % (unsigned)NCHK);
        for (int jc = 0; jc < NCHK; ++jc) {
            int chk = jc + rot;
            chk -= chk >= NCHK ? NCHK : 0;
            const int idx = chk * (256 * G) + tid;
            const int nrow = idx >> 7, kcol = idx & 127;
            const int n = n0 + h * ROWS + nrow, k = k0 + kcol;
            const int gq = ((nrow & 15) >> 2);  // the row's g at write time
            const float v = sAcc[nrow * 128 + (kcol ^ (gq << 4))];
            if (n < p.Cout && k < p.Ktot) {
                if (p.slab)
                    p.slab[(long)bsplit * p.slab_stride + (long)n * p.ldgw + k] = v;
                else
                    atomicAdd(dwg + ((long)n * p.ldgw + k), v);
            }
        }
        if (HALVES > 1) __syncthreads();
    }
}

}  // namespace

// vt_wgrad_span.hip: stride-1 3x3 bf16 layers; -1 when it does not apply
int vt_wgrad_span_dispatch(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw,
                           float* scratch, int64_t scratch_bytes, void* stream);
// vt_wgrad_span.hip: stride-2 3x3 bf16 layers on even maps as two stride-1 launches over row-parity views
int vt_wgrad_span_s2_dispatch(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw,
                              void* stream);

// vt_wgrad6.hip: stride-1 3x3 bf16 layers wider than 32 channels as a CU-owning kernel (one or several same-shape layers
// per launch); -1 when it does not apply
int vt_wgrad6_dispatch(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw, void* stream);
int vt_wgrad6_group(const vt_conv_desc* d, int G, const void* const* x, const void* const* dz, float* const* dw,
                    int32_t ldgw, void* stream);

// dw[n][k] += sum over the splits of slab[s][n][k], in split order: the ordered second stage of the filter gradient
__global__ void __launch_bounds__(256) wgrad_reduce_slabs_kernel(const float* __restrict__ slab, long stride, int split,
                                                                 float* __restrict__ dw, int Cout, int Ktot, int ldgw) {
    const long total = (long)Cout * (Ktot / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / (Ktot / 4), k = (i - n * (Ktot / 4)) * 4;
        const long off = n * ldgw + k;
        float4 acc = *(const float4*)(slab + off);
        for (int s = 1; s < split; ++s) {
            const float4 v = *(const float4*)(slab + (long)s * stride + off);
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
        float4 o = *(float4*)(dw + off);
        o.x += acc.x, o.y += acc.y, o.z += acc.z, o.w += acc.w;
        *(float4*)(dw + off) = o;
    }
}

int vt_wgrad_reduce_slabs(const float* slab, long stride, int split, float* dw, int Cout, int Ktot, int ldgw,
                          hipStream_t st) {
    const long total = (long)Cout * (Ktot / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slab, stride, split, dw, Cout,
                       Ktot, ldgw);
    VT_CHECK_LAUNCH("vt_conv_wgrad(reduce)");
    return VT_OK;
}

static int wgrad_impl(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw, float* scratch,
                      int64_t scratch_bytes, void* stream);

extern "C" int vt_conv_wgrad(const vt_conv_desc* d, const void* x, const void* dz, float* dw,
                             int32_t ldgw, void* stream) {
    return wgrad_impl(d, x, dz, dw, ldgw, nullptr, 0, stream);
}

// Two-stage filter gradient: every (tile, pixel split) workgroup STORES its partial tile into its own slab of `scratch`
// and a second kernel adds the slabs to dw in split order -- no atomics (they run at ~1 TB/s; a 40 MB scratch that is
// reused by every layer stays in the memory-side cache) and the result does not depend on the order of the workgroups.
// A scratch too small for a layer's usual split gets fewer, longer splits (never atomics across splits); the row-parity
// stride-2 path (atomic, column-mapped) is not taken in this mode.
extern "C" int vt_conv_wgrad_slabs(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw,
                                   void* scratch, int64_t scratch_bytes, void* stream) {
    return wgrad_impl(d, x, dz, dw, ldgw, (float*)scratch, scratch_bytes, stream);
}

// n filter gradients of ONE descriptor (same-shape layers: the 3x3 convs of a stage's DarknetBlocks, an OSA chain) in as
// few launches as the kernels allow: the CU-owning kernel takes up to 8 layers per launch and pays its prologue and its
// f32 atomic flush once per launch; shapes it does not cover run layer by layer.
static int wgrad_general(const vt_conv_desc* d, int nl, const void* const* xs, const void* const* dzs, float* const* dws,
                         int32_t ldgw, float* scratch, int64_t scratch_bytes, bool slabs, void* stream);

extern "C" int vt_conv_wgrad_group(const vt_conv_desc* d, int32_t n, const void* const* x, const void* const* dz,
                                   float* const* dw, int32_t ldgw, void* stream) {
    VT_REQUIRE(d && x && dz && dw && n >= 1, VT_ERR_INVALID, "vt_conv_wgrad_group: bad argument");
    for (int i = 0; i < n; ++i)
        VT_REQUIRE(x[i] && dz[i] && dw[i], VT_ERR_INVALID, "vt_conv_wgrad_group: null operand %d", i);
    // (clamped to what one launch's argument block carries: kWgMaxGroup layers in the general kernel, 8 in wgrad6)
    int maxg = VT_KNOB("VT_WGRAD6_GROUP", 8);
    maxg = maxg < 1 ? 1 : (maxg > kWgMaxGroup ? kWgMaxGroup : maxg);
    int i = 0;
    while (i < n) {
        int g = n - i < maxg ? n - i : maxg;
        if (g < 1) g = 1;
        int rc = -1;
        if (g > 1) {
            // (the argument checks of the single-layer entry, which the grouped kernel shares)
            const long in_elems = (long)d->B * d->Hi * d->Wi * d->ldx;
            const long M = (long)d->B * d->Ho * d->Wo;
            if (d->dtype == VT_BF16 && ldgw >= d->ntaps * d->Cin && in_elems < 0x7fffffffL && M * d->ldy < 0x7fffffffL &&
                d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->oHs == 1 && d->oWs == 1 &&
                d->oh0 == 0 && d->ow0 == 0 && d->oH == d->Ho && d->oW == d->Wo)
                rc = vt_wgrad6_group(d, g, x + i, dz + i, dw + i, ldgw, stream);
            // 1x1 stride-1 layers (the DarknetBlock / CSP 1x1 units of a stage): one launch of the general kernel over the
            // group -- a launch of one such layer is mostly its f32 atomic flush (256 workgroups x 64 KiB) and its ramp
            // (VT_WGRAD_GROUP_1X1=1: off by default -- the engine does not hold these layers back (NOTEBOOK R5.18), and
            //  without the knob every launch list runs exactly the launches it ran before this path existed)
            if (rc == -1 && VT_KNOB("VT_WGRAD_GROUP_1X1", 0) && d->dtype == VT_BF16 && d->ntaps == 1 && d->sh == 1 && d->sw == 1 &&
                d->Ho == d->Hi && d->Wo == d->Wi && d->h0 + d->dh[0] == 0 && d->w0 + d->dw[0] == 0 && ldgw >= d->Cin &&
                in_elems < 0x7fffffffL && M * d->ldy < 0x7fffffffL && d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->ldx % 8 == 0 &&
                d->ldy % 8 == 0 && d->oHs == 1 && d->oWs == 1 && d->oh0 == 0 && d->ow0 == 0 && d->oH == d->Ho && d->oW == d->Wo) {
                bool aligned = true;
                for (int k = 0; k < g; ++k) aligned = aligned && vt_aligned16(x[i + k]) && vt_aligned16(dz[i + k]);
                if (aligned) rc = wgrad_general(d, g, x + i, dz + i, dw + i, ldgw, nullptr, 0, false, stream);
            }
        }
        if (rc == -1) {  // one by one (argument checks, every kernel family)
            g = 1;
            rc = wgrad_impl(d, x[i], dz[i], dw[i], ldgw, nullptr, 0, stream);
        }
        if (rc != VT_OK) return rc;
        i += g;
    }
    return VT_OK;
}

static int wgrad_impl(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw, float* scratch,
                      int64_t scratch_bytes, void* stream) {
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
    const bool slabs = scratch && (d->ntaps * d->Cin) % 4 == 0 && ldgw % 4 == 0;
    if (!scratch) {  // (atomic flush: not in the two-stage / deterministic mode)
        const int rc = vt_wgrad6_dispatch(d, x, dz, dw, ldgw, stream);
        if (rc >= 0) return rc;
    }
    {
        const int rc = vt_wgrad_span_dispatch(d, x, dz, dw, ldgw, slabs ? scratch : nullptr, scratch_bytes, stream);
        if (rc >= 0) return rc;
    }
    if (!slabs) {  // (the row-parity stride-2 launches scatter through a column map: atomic flush only)
        const int rc = vt_wgrad_span_s2_dispatch(d, x, dz, dw, ldgw, stream);
        if (rc >= 0) return rc;
    }

    return wgrad_general(d, 1, &x, &dz, &dw, ldgw, scratch, scratch_bytes, slabs, stream);
}

// the general kernel on nl >= 1 same-shape layers (nl > 1: one launch, the pixel splits -- and with them the f32 atomic
// flush, #workgroups x 64 KiB at ~1.3 TB/s -- sized for the whole group; never with slabs)
static int wgrad_general(const vt_conv_desc* d, int nl, const void* const* xs, const void* const* dzs, float* const* dws,
                         int32_t ldgw, float* scratch, int64_t scratch_bytes, bool slabs, void* stream) {
    const int epc = vt_epc(d->dtype);
    const long M = (long)d->B * d->Ho * d->Wo;
    const void* x = xs[0];
    const void* dz = dzs[0];
    float* dw = dws[0];
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x, a.dz = dz, a.dw = dw;
    a.nlayers = nl;
    for (int i = 0; i < nl && i < kWgMaxGroup; ++i) a.xs[i] = xs[i], a.dzs[i] = dzs[i], a.dws[i] = dws[i];
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

    // Split of the pixel range: aim for `target` workgroups (about 2 per CU), but keep >= 8
    // steps per wave group; atomic traffic is (#workgroups x 64 KiB) whatever the layer.
    const int target = (512);
    const int max_split_env = (4096);
    const int variant = (0);
    const int G = variant == 1 ? 1 : (variant == 2 ? 4 : 2);
    const int pk = 4 * epc;
    const long tiles = (long)a.tiles_n * a.tiles_k;
    // 1x1 layers on the smaller maps are short: their cost is the atomic flush (#workgroups x 64 KiB
    // at ~1.3 TB/s), so they get one workgroup per CU instead of two (measured 38 -> 32 us)
    const int tgt = (d->ntaps == 1 && M <= 262144 && target == 512) ? 256 : target;
    long split = tgt / (tiles * nl);
    // steps per wave group and workgroup, at least (8 until round 3; 24 measured 12.29 vs 12.41 ms at batch 128, same at 256)
    const int min_steps = (24);
    const long max_split = (M + (long)min_steps * pk * G - 1) / ((long)min_steps * pk * G);
    if (split > max_split) split = max_split;
    if (split > max_split_env) split = max_split_env;
    const long slab_stride = (long)d->Cout * ldgw;
    // two-stage mode never falls back to atomics: a scratch too small for the chosen split gets fewer, longer splits
    if (slabs && split * slab_stride * 4 > scratch_bytes) split = scratch_bytes / (slab_stride * 4);
    if (split < 1) split = 1;
    long chunk = (M + split - 1) / split;
    chunk = (chunk + (long)pk * G - 1) / ((long)pk * G) * ((long)pk * G);
    split = (M + chunk - 1) / chunk;
    a.chunk = (int)chunk;
    a.ablate = 0;
    const bool use_slabs = slabs && split > 1 && split * slab_stride * 4 <= scratch_bytes;  // (split 1: one writer per element)
    a.slab = use_slabs ? scratch : nullptr;
    a.slab_stride = slab_stride;

    hipStream_t st = (hipStream_t)stream;
    a.split = (int)split;
    a.xcds = (8);
    dim3 grid(vt_xcd_grid(tiles * split * nl));
    const int stage = 2 * pk * 128 * vt_elem_size(d->dtype);  // 16 KiB
    const bool unit = d->ntaps == 1 && d->sh == 1 && d->sw == 1 && d->h0 + d->dh[0] == 0 && d->w0 + d->dw[0] == 0 &&
                      d->Ho == d->Hi && d->Wo == d->Wi;
#define VT_WG_LAUNCH(TT, GG, PDD)                                                                      \
    do {                                                                                               \
        int smem = kTapHdr + GG * (PDD + 1) * stage;                                                   \
        if (GG > 1 && smem < kTapHdr + 65536) smem = kTapHdr + 65536; /* the f32 reduction image */   \
        auto kern = unit ? wgrad_kernel<TT, GG, PDD, true> : wgrad_kernel<TT, GG, PDD, false>;         \
        if (smem > 64 * 1024) {                                                                        \
            const int rc_ = vt_raise_dynamic_lds((const void*)kern, smem, "vt_conv_wgrad");            \
            if (rc_ != VT_OK) return rc_;                                                              \
        }                                                                                              \
        vt_note_kernel("wgrad_kernel<%s,%d,%d,%s>", sizeof(TT) == 2 ? "bf16" : "f32", GG, PDD, unit ? "unit" : "gather"); \
        hipLaunchKernelGGL(kern, grid, dim3(256 * GG), smem, st, a);                                   \
    } while (0)
    if (d->dtype == VT_BF16) {
        if (G == 1)
            VT_WG_LAUNCH(bf16_t, 1, 2);
        else if (G == 4)
            VT_WG_LAUNCH(bf16_t, 4, 1);
        else
            VT_WG_LAUNCH(bf16_t, 2, 1);
    } else {
        if (G == 1)
            VT_WG_LAUNCH(float, 1, 2);
        else if (G == 4)
            VT_WG_LAUNCH(float, 4, 1);
        else
            VT_WG_LAUNCH(float, 2, 1);
    }
#undef VT_WG_LAUNCH
    VT_CHECK_LAUNCH("vt_conv_wgrad");
    if (use_slabs) return vt_wgrad_reduce_slabs(scratch, slab_stride, (int)split, dw, d->Cout, a.Ktot, ldgw, st);
    return VT_OK;
}
