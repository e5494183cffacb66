// vt_dwconv.hip -- depthwise convolution (nn.Conv2d(C, C, k, groups=C) inside a ConvNormAct: reference components.py:26-35
// with `groups = in_channels`), forward, data gradient and filter gradient (round 6).
//
// Off the Darknet / VoVNet path (no model of SURVEY section 8 uses it); built so that every `groups` the reference's
// ConvNormAct constructor accepts runs on the GPU: the grouped path of the engine (one unit per group over channel slices)
// needs 16-byte channel slices, i.e. >= 8 (bf16) / 4 (f32) channels per group -- depthwise has one.
//
// A depthwise convolution has no reduction over channels: k*k multiply-adds per output element, HBM-bound streaming work,
// no matrix pipe.  One thread mapping for all three kernels (the "RowMap" of vt_elementwise.hip): a thread owns ONE 16-byte
// channel chunk and walks pixels, so the k*k filter taps of its 4 / 8 channels are read through the scalar / L1 cache and
// consecutive lanes touch consecutive 16-byte chunks of a pixel row.
//   forward:  z(b, i, j, c) = sum_t x(b, i*s - pad + r_t*d, j*s - pad + c_t*d, c) * w[c][t]  (+ per-channel sum z, sum z^2 of
//             the STORED values, fixed point, for training-mode BatchNorm: the statistics contract of vt_conv_igemm)
//   dgrad:    dx(b, h, w, c) (+)= sum_t [ (h + pad - r_t*d) and (w + pad - c_t*d) divisible by s and inside the output ]
//             dz(b, (h + pad - r_t*d)/s, (w + pad - c_t*d)/s, c) * w[c][t]
//   wgrad:    dw[c][t] += sum_pixels dz(b, i, j, c) * x(b, i*s - pad + r_t*d, j*s - pad + c_t*d, c)   (f32 atomics)
// The filter is the f32 master [C][k*k] (a torch [C, 1, k, k] weight); bf16 launches round it as the conv kernels' bf16
// mirror does.
#include <stdlib.h>

#include "vt_common.h"

namespace {

constexpr int kT = 256;

#define VT_DISPATCH_T(dtype, NAME, ...)                               \
    do {                                                              \
        if ((dtype) == VT_BF16) {                                     \
            typedef bf16_t T;                                         \
            __VA_ARGS__;                                              \
        } else {                                                      \
            typedef float T;                                          \
            __VA_ARGS__;                                              \
        }                                                             \
    } while (0)
#define VT_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != VT_OK) return rc__; \
    } while (0)

struct DwArgs {
    const void* x;
    const float* w;
    void* y;
    const void* res;
    float* stats;
    float* dw;
    long M;  // rows of the tensor the launch iterates (outputs: fwd / wgrad; inputs: dgrad)
    int B, Hi, Wi, Ho, Wo, C, k, s, pad, dil;
    int ldx, ldy, ldr, accumulate;
    int CPR, CT, RT, iters;
};

template <typename T>
__device__ __forceinline__ void load_w(const DwArgs& a, int c0, int t, float (&wv)[VecIO<T>::EPC]) {
    const int kk = a.k * a.k;
#pragma unroll
    for (int e = 0; e < VecIO<T>::EPC; ++e) wv[e] = VecIO<T>::round(a.w[(long)(c0 + e) * kk + t]);
}

// forward: x [B][Hi][Wi] -> y [B][Ho][Wo]; STATS: sum / sum of squares of the stored outputs
template <typename T, bool STATS>
__global__ void __launch_bounds__(kT) dwconv_fwd_kernel(const DwArgs a) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / a.CT, tc = t % a.CT;
    const bool active = r < a.RT;
    const T* x = (const T*)a.x;
    T* y = (T*)a.y;
    const long row0 = (long)blockIdx.x * a.RT * a.iters + r;
    for (int col = tc; col < a.CPR; col += a.CT) {
        const int c0 = col * EPC;
        float s1[EPC], s2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
        if (active)
            for (int it = 0; it < a.iters; ++it) {
                const long row = row0 + (long)it * a.RT;
                if (row >= a.M) break;
                const long b = row / ((long)a.Ho * a.Wo);
                const int rem = (int)(row - b * a.Ho * a.Wo);
                const int i = rem / a.Wo, j = rem - i * a.Wo;
                float acc[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
                for (int kr = 0; kr < a.k; ++kr) {
                    const int h = i * a.s - a.pad + kr * a.dil;
                    if (h < 0 || h >= a.Hi) continue;
                    for (int kc = 0; kc < a.k; ++kc) {
                        const int w_ = j * a.s - a.pad + kc * a.dil;
                        if (w_ < 0 || w_ >= a.Wi) continue;
                        float xv[EPC], wv[EPC];
                        VecIO<T>::unpack(*(const uint4*)(x + ((b * a.Hi + h) * a.Wi + w_) * a.ldx + c0), xv);
                        load_w<T>(a, c0, kr * a.k + kc, wv);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[e] = fmaf(xv[e], wv[e], acc[e]);
                    }
                }
                const uint4 out = VecIO<T>::pack(acc);
                *(uint4*)(y + row * a.ldy + c0) = out;
                if (STATS) {
                    float rv[EPC];
                    VecIO<T>::unpack(out, rv);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) s1[e] += rv[e], s2[e] = fmaf(rv[e], rv[e], s2[e]);
                }
            }
        if (STATS) {
            // the row lanes of a wave that share this channel chunk fold in registers where the chunks per row are a power
            // of two below 64 (one writer per wave and chunk), then fixed-point atomics: order-free, run-to-run identical
            const bool inwave = a.CT < 64 && (a.CT & (a.CT - 1)) == 0 && a.CPR <= a.CT;
            bool writer = active;
            if (inwave) {
                for (int off = a.CT; off < 64; off <<= 1) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        s1[e] += __shfl_xor(s1[e], off, 64);
                        s2[e] += __shfl_xor(s2[e], off, 64);
                    }
                }
                writer = (t & 63) < a.CT;
            }
            if (writer) {
                const int rep = blockIdx.x % kStatReplicas;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    vt_stat_add(a.stats, ((long)rep * 2 + 0) * a.C + c0 + e, s1[e]);
                    vt_stat_add(a.stats, ((long)rep * 2 + 1) * a.C + c0 + e, s2[e]);
                }
            }
        }
    }
}

// data gradient: iterates the INPUT pixels; a.x = dz [B][Ho][Wo], a.y = dx [B][Hi][Wi], a.res: += (may alias dx)
template <typename T>
__global__ void __launch_bounds__(kT) dwconv_dgrad_kernel(const DwArgs a) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / a.CT, tc = t % a.CT;
    if (r >= a.RT) return;
    const T* dz = (const T*)a.x;
    const T* res = (const T*)a.res;
    T* dx = (T*)a.y;
    const long row0 = (long)blockIdx.x * a.RT * a.iters + r;
    for (int col = tc; col < a.CPR; col += a.CT) {
        const int c0 = col * EPC;
        for (int it = 0; it < a.iters; ++it) {
            const long row = row0 + (long)it * a.RT;
            if (row >= a.M) break;
            const long b = row / ((long)a.Hi * a.Wi);
            const int rem = (int)(row - b * a.Hi * a.Wi);
            const int h = rem / a.Wi, w_ = rem - h * a.Wi;
            float acc[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
            for (int kr = 0; kr < a.k; ++kr) {
                const int hn = h + a.pad - kr * a.dil;
                if (hn < 0 || hn % a.s) continue;
                const int i = hn / a.s;
                if (i >= a.Ho) continue;
                for (int kc = 0; kc < a.k; ++kc) {
                    const int wn = w_ + a.pad - kc * a.dil;
                    if (wn < 0 || wn % a.s) continue;
                    const int j = wn / a.s;
                    if (j >= a.Wo) continue;
                    float gv[EPC], wv[EPC];
                    VecIO<T>::unpack(*(const uint4*)(dz + ((b * a.Ho + i) * a.Wo + j) * a.ldx + c0), gv);
                    load_w<T>(a, c0, kr * a.k + kc, wv);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[e] = fmaf(gv[e], wv[e], acc[e]);
                }
            }
            if (res) {
                // (the sum is rounded to the storage type first, as a separate data-gradient launch + add would)
                float rv[EPC], sv[EPC];
                VecIO<T>::unpack(VecIO<T>::pack(acc), sv);
                VecIO<T>::unpack(*(const uint4*)(res + row * a.ldr + c0), rv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] = sv[e] + rv[e];
            }
            *(uint4*)(dx + row * a.ldy + c0) = VecIO<T>::pack(acc);
        }
    }
}

// filter gradient: iterates the OUTPUT pixels; a.x = x, a.y = dz (read), a.dw f32 [C][k*k] +=; TAPS taps per pass from t0
template <typename T, int TAPS>
__global__ void __launch_bounds__(kT) dwconv_wgrad_kernel(const DwArgs a, int t0) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / a.CT, tc = t % a.CT;
    const bool active = r < a.RT;
    const T* x = (const T*)a.x;
    const T* dz = (const T*)a.y;
    const int kk = a.k * a.k;
    const long row0 = (long)blockIdx.x * a.RT * a.iters + r;
    for (int col = tc; col < a.CPR; col += a.CT) {
        const int c0 = col * EPC;
        float acc[TAPS][EPC];
#pragma unroll
        for (int q = 0; q < TAPS; ++q)
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[q][e] = 0.f;
        if (active)
            for (int it = 0; it < a.iters; ++it) {
                const long row = row0 + (long)it * a.RT;
                if (row >= a.M) break;
                const long b = row / ((long)a.Ho * a.Wo);
                const int rem = (int)(row - b * a.Ho * a.Wo);
                const int i = rem / a.Wo, j = rem - i * a.Wo;
                float gv[EPC];
                VecIO<T>::unpack(*(const uint4*)(dz + row * a.ldy + c0), gv);
#pragma unroll
                for (int q = 0; q < TAPS; ++q) {
                    const int tap = t0 + q;
                    if (tap >= kk) continue;
                    const int h = i * a.s - a.pad + (tap / a.k) * a.dil, w_ = j * a.s - a.pad + (tap % a.k) * a.dil;
                    if (h < 0 || h >= a.Hi || w_ < 0 || w_ >= a.Wi) continue;
                    float xv[EPC];
                    VecIO<T>::unpack(*(const uint4*)(x + ((b * a.Hi + h) * a.Wi + w_) * a.ldx + c0), xv);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[q][e] = fmaf(gv[e], xv[e], acc[q][e]);
                }
            }
        const bool inwave = a.CT < 64 && (a.CT & (a.CT - 1)) == 0 && a.CPR <= a.CT;
        bool writer = active;
        if (inwave) {
            for (int off = a.CT; off < 64; off <<= 1) {
#pragma unroll
                for (int q = 0; q < TAPS; ++q)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[q][e] += __shfl_xor(acc[q][e], off, 64);
            }
            writer = (t & 63) < a.CT;
        }
        if (writer) {
#pragma unroll
            for (int q = 0; q < TAPS; ++q) {
                if (t0 + q >= kk) continue;
#pragma unroll
                for (int e = 0; e < EPC; ++e) atomicAdd(a.dw + (long)(c0 + e) * kk + t0 + q, acc[q][e]);
            }
        }
    }
}

int fill(DwArgs& a, const char* who, int dtype, int B, int Hi, int Wi, int C, int k, int s, int pad, int dil, long rows) {
    VT_REQUIRE(dtype == VT_F32 || dtype == VT_BF16, VT_ERR_UNSUPPORTED, "%s: dtype %d", who, dtype);
    VT_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C > 0 && k >= 1 && k <= 7 && s >= 1 && s <= 4 && pad >= 0 && dil >= 1,
               VT_ERR_INVALID, "%s: bad geometry", who);
    const int epc = vt_epc(dtype);
    VT_REQUIRE(C % epc == 0, VT_ERR_UNSUPPORTED, "%s: C=%d must be a multiple of %d", who, C, epc);
    a.B = B, a.Hi = Hi, a.Wi = Wi, a.C = C, a.k = k, a.s = s, a.pad = pad, a.dil = dil;
    a.Ho = (Hi + 2 * pad - dil * (k - 1) - 1) / s + 1;
    a.Wo = (Wi + 2 * pad - dil * (k - 1) - 1) / s + 1;
    VT_REQUIRE(a.Ho > 0 && a.Wo > 0, VT_ERR_INVALID, "%s: empty output", who);
    a.CPR = C / epc;
    a.CT = a.CPR < kT ? a.CPR : kT;
    a.RT = kT / a.CT;
    a.M = rows < 0 ? (long)B * a.Ho * a.Wo : rows;
    long it = (a.M + (long)a.RT * 2048 - 1) / ((long)a.RT * 2048);
    a.iters = (int)(it < 1 ? 1 : (it > 64 ? 64 : it));
    return VT_OK;
}
unsigned blocks(const DwArgs& a) {
    const long per = (long)a.RT * a.iters;
    return (unsigned)((a.M + per - 1) / per);
}

}  // namespace

extern "C" {

int vt_dwconv_fwd(const void* x, int32_t ldx, const float* w, void* y, int32_t ldy, float* stats, int32_t B, int32_t Hi,
                  int32_t Wi, int32_t C, int32_t k, int32_t s, int32_t pad, int32_t dil, int32_t dtype, void* stream) {
    VT_REQUIRE(x && w && y, VT_ERR_INVALID, "vt_dwconv_fwd: null argument");
    DwArgs a;
    memset(&a, 0, sizeof(a));
    VT_TRY(fill(a, "vt_dwconv_fwd", dtype, B, Hi, Wi, C, k, s, pad, dil, -1));
    VT_REQUIRE(ldx >= C && ldy >= C && ldx % vt_epc(dtype) == 0 && ldy % vt_epc(dtype) == 0 && vt_aligned16(x) && vt_aligned16(y),
               VT_ERR_INVALID, "vt_dwconv_fwd: bad strides / alignment");
    a.x = x, a.w = w, a.y = y, a.stats = stats, a.ldx = ldx, a.ldy = ldy;
    if (stats) {
        VT_DISPATCH_T(dtype, "vt_dwconv_fwd",
                      hipLaunchKernelGGL((dwconv_fwd_kernel<T, true>), dim3(blocks(a)), dim3(kT), 0, (hipStream_t)stream, a));
    } else {
        VT_DISPATCH_T(dtype, "vt_dwconv_fwd",
                      hipLaunchKernelGGL((dwconv_fwd_kernel<T, false>), dim3(blocks(a)), dim3(kT), 0, (hipStream_t)stream, a));
    }
    VT_CHECK_LAUNCH("vt_dwconv_fwd");
    return VT_OK;
}

int vt_dwconv_dgrad(const void* dz, int32_t lddz, const float* w, void* dx, int32_t lddx, const void* residual, int32_t ldr,
                    int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t k, int32_t s, int32_t pad, int32_t dil,
                    int32_t dtype, void* stream) {
    VT_REQUIRE(dz && w && dx, VT_ERR_INVALID, "vt_dwconv_dgrad: null argument");
    DwArgs a;
    memset(&a, 0, sizeof(a));
    VT_TRY(fill(a, "vt_dwconv_dgrad", dtype, B, Hi, Wi, C, k, s, pad, dil, (long)B * Hi * Wi));
    const int epc = vt_epc(dtype);
    VT_REQUIRE(lddz >= C && lddx >= C && lddz % epc == 0 && lddx % epc == 0 && vt_aligned16(dz) && vt_aligned16(dx) &&
                   (!residual || (ldr >= C && ldr % epc == 0 && vt_aligned16(residual))),
               VT_ERR_INVALID, "vt_dwconv_dgrad: bad strides / alignment");
    a.x = dz, a.w = w, a.y = dx, a.res = residual, a.ldx = lddz, a.ldy = lddx, a.ldr = ldr;
    VT_DISPATCH_T(dtype, "vt_dwconv_dgrad",
                  hipLaunchKernelGGL(dwconv_dgrad_kernel<T>, dim3(blocks(a)), dim3(kT), 0, (hipStream_t)stream, a));
    VT_CHECK_LAUNCH("vt_dwconv_dgrad");
    return VT_OK;
}

int vt_dwconv_wgrad(const void* x, int32_t ldx, const void* dz, int32_t lddz, float* dw, int32_t B, int32_t Hi, int32_t Wi,
                    int32_t C, int32_t k, int32_t s, int32_t pad, int32_t dil, int32_t dtype, void* stream) {
    VT_REQUIRE(x && dz && dw, VT_ERR_INVALID, "vt_dwconv_wgrad: null argument");
    DwArgs a;
    memset(&a, 0, sizeof(a));
    VT_TRY(fill(a, "vt_dwconv_wgrad", dtype, B, Hi, Wi, C, k, s, pad, dil, -1));
    const int epc = vt_epc(dtype);
    VT_REQUIRE(ldx >= C && lddz >= C && ldx % epc == 0 && lddz % epc == 0 && vt_aligned16(x) && vt_aligned16(dz), VT_ERR_INVALID,
               "vt_dwconv_wgrad: bad strides / alignment");
    a.x = x, a.y = const_cast<void*>(dz), a.dw = dw, a.ldx = ldx, a.ldy = lddz;
    // nine taps per pass (their accumulators stay in registers); larger filters take ceil(k*k / 9) passes over the data
    for (int t0 = 0; t0 < k * k; t0 += 9) {
        VT_DISPATCH_T(dtype, "vt_dwconv_wgrad",
                      hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 9>), dim3(blocks(a)), dim3(kT), 0, (hipStream_t)stream, a, t0));
        VT_CHECK_LAUNCH("vt_dwconv_wgrad");
    }
    return VT_OK;
}

}  // extern "C"
