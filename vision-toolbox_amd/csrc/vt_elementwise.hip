// vt_elementwise.hip -- the HBM-bound kernels around the MFMA convolutions:
// BatchNorm statistics / normalise / backward, ReLU, residual add, pooling,
// ESE gate, classifier loss, SGD, and the layout / precision plumbing.
//
// All activation kernels use one thread mapping ("RowMap"): a thread owns ONE
// 16-byte channel chunk column and walks rows (pixels), so per-channel
// coefficients live in registers and consecutive lanes touch consecutive
// 16-byte chunks of a pixel row (full-line coalesced NHWC accesses, G13).
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "vt_common.h"
#include "vt_bn_fin.h"

namespace {

#ifndef VT_EW_THREADS
#define VT_EW_THREADS 256
#endif
constexpr int kThreads = VT_EW_THREADS;
constexpr int kUnroll = 4;  // rows in flight per thread in the streaming kernels

// thread -> (channel chunk column, row lane) for an [M][C] matrix of 16-byte chunks
struct RowMap {
    int CPR;  // chunks per row
    int CT;   // threads along the row
    int RT;   // rows per block pass
    int iters;
    int rev;  // walk the row blocks from the last to the first (see vt_bn_order)
    __host__ static RowMap make(int C, int epc, long M, int target_blocks = 4096) {
        RowMap r;
        r.CPR = C / epc;
        r.CT = r.CPR < kThreads ? r.CPR : kThreads;
        r.RT = kThreads / r.CT;
        long it = (M + (long)r.RT * target_blocks - 1) / ((long)r.RT * target_blocks);
        if (it < 1) it = 1;
        if (it > 64) it = 64;
        r.iters = (int)it;
        r.rev = 0;
        return r;
    }
    __host__ unsigned blocks(long M) const {
        const long rows_per_block = (long)RT * iters;
        return (unsigned)((M + rows_per_block - 1) / rows_per_block);
    }
};

// LDS sums over the row lanes of a block as 64-bit fixed point (2^-32): integer atomics, order-free (a float LDS
// atomic made the pooled features, and with them logits and loss, differ in the last bit from run to run)
__device__ __forceinline__ void lds_fix_add(unsigned long long* s, int i, float v) {
    atomicAdd(&s[i], (unsigned long long)__float2ll_rn(v * 4294967296.0f));
}
__device__ __forceinline__ float lds_fix_get(const unsigned long long* s, int i) {
    return (float)((double)(long long)s[i] * (1.0 / 4294967296.0));
}

template <typename T>
__device__ __forceinline__ uint4 ld16(const T* p) { return *(const uint4*)p; }
template <typename T>
__device__ __forceinline__ void st16(T* p, const uint4& v) { *(uint4*)p = v; }

// ---------------------------------------------------------------------------------
// BatchNorm finalize (training): stats -> mean / invstd / scale / shift + running stats
// ---------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const VtFinFwd f) {
    // (the arithmetic: vt_fin_fwd_channel, vt_bn_fin.h -- shared with the passes that finalize for themselves)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= f.C) return;
    // exact integer sums of the fixed-point replicas (vt_common.h), loaded as independent pairs
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    float rm = 0.f, rv = 0.f;
    if (f.running_mean) rm = f.running_mean[c], rv = f.running_var[c];
    const double s = vt_stat_sum(f.stats, c, 2L * f.C), ss = vt_stat_sum(f.stats, (long)f.C + c, 2L * f.C);
    float sc, sf;
    vt_fin_fwd_channel(f, c, s, ss, g, b, rm, rv, sc, sf);
}

// replicas 1.. of int64[R][n] are added into replica 0 and zeroed (n = 2 * C * 2 limbs)
__global__ void stat_fold_kernel(long long* __restrict__ q, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long acc = q[i];
#pragma unroll 4
    for (int r = 1; r < kStatReplicas; ++r) {
        acc += q[(long)r * n + i];
        q[(long)r * n + i] = 0;
    }
    q[i] = acc;
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift, float* mean, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float istd = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * istd;
    scale[c] = sc;
    shift[c] = b - rm[c] * sc;
    if (mean) mean[c] = rm[c];
    if (invstd) invstd[c] = istd;
}

struct BnEvalBatch {
    int n;
    vt_bn_eval_item it[VT_PACK_BATCH];
};
// one workgroup per (BatchNorm, 256 channels)
__global__ void __launch_bounds__(256) bn_eval_coeffs_batch_kernel(const BnEvalBatch b, int blocks_per_item) {
    const vt_bn_eval_item& it = b.it[blockIdx.x / blocks_per_item];
    const int c = (blockIdx.x % blocks_per_item) * 256 + threadIdx.x;
    if (c >= it.C) return;
    const float istd = 1.f / sqrtf(it.running_var[c] + it.eps);
    const float g = it.gamma ? it.gamma[c] : 1.f, bb = it.beta ? it.beta[c] : 0.f;
    const float sc = g * istd;
    it.scale[c] = sc;
    it.shift[c] = bb - it.running_mean[c] * sc;
    if (it.mean) it.mean[c] = it.running_mean[c];
    if (it.invstd) it.invstd[c] = istd;
}

// ---------------------------------------------------------------------------------
// Activation codes of the `relu` argument of vt_bn_act_apply / _bwd_reduce / _bwd_apply (components.py:37-44): 0 none, 1 ReLU,
// 2 LeakyReLU(0.2), 3 SiLU ("swish"), 4 GELU (exact, erf).  Codes 0 / 1 run the kernels' original instantiations (the hot
// path); codes >= 2 a GEN instantiation of the same kernels, so that the Darknet / VoVNet launches keep their code.
__device__ __forceinline__ float act_fwd(int code, float u) {
    switch (code) {
        case 1: return fmaxf(u, 0.f);
        case 2: return u > 0.f ? u : 0.2f * u;
        case 3: return u / (1.f + expf(-u));
        case 4: return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
        default: return u;
    }
}
// d act / d u at pre-activation u
__device__ __forceinline__ float act_grad(int code, float u) {
    switch (code) {
        case 1: return u > 0.f ? 1.f : 0.f;
        case 2: return u > 0.f ? 1.f : 0.2f;
        case 3: {
            const float s = 1.f / (1.f + expf(-u));
            return s * (1.f + u * (1.f - s));
        }
        case 4: return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
        default: return 1.f;
    }
}

// y = [relu](z*scale + shift) [+ residual]
// ---------------------------------------------------------------------------------
template <typename T, bool kRes, bool GEN = false>
__global__ void __launch_bounds__(kThreads)
bn_act_apply_kernel(const T* __restrict__ z, int ldz, const float* __restrict__ scale,
                    const float* __restrict__ shift, const T* __restrict__ res, int ldr, T* __restrict__ y,
                    int ldy, long M, RowMap rm, int relu) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)(rm.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        float sc[EPC], sf[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) sc[e] = 1.f, sf[e] = 0.f;
        if (scale) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) sc[e] = scale[col * EPC + e];
        }
        if (shift) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) sf[e] = shift[col * EPC + e];
        }
        const T* pz = z + row0 * ldz + col * EPC;
        const T* pr = kRes ? res + row0 * ldr + col * EPC : nullptr;
        T* py = y + row0 * ldy + col * EPC;
        const long sz = (long)rm.RT * ldz, sr = (long)rm.RT * ldr, sy = (long)rm.RT * ldy;
        // kUnroll rows per trip: all loads are issued before the first use (bytes in flight)
        for (int it = 0; it < rm.iters; it += kUnroll) {
            uint4 vz[kUnroll], vr[kUnroll];
            bool ok[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                ok[u] = (it + u < rm.iters) && (row0 + (long)(it + u) * rm.RT < M);
                vz[u] = vr[u] = make_uint4(0, 0, 0, 0);
                if (ok[u]) {
                    vz[u] = ld16(pz + (it + u) * sz);
                    if (kRes) vr[u] = ld16(pr + (it + u) * sr);
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                float v[EPC];
                VecIO<T>::unpack(vz[u], v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    v[e] = fmaf(v[e], sc[e], sf[e]);
                    if constexpr (GEN) v[e] = act_fwd(relu, v[e]);
                    else v[e] = relu ? fmaxf(v[e], 0.f) : v[e];
                }
                if (kRes) {
                    float rr[EPC];
                    VecIO<T>::unpack(vr[u], rr);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] += rr[e];
                }
                if (ok[u]) st16(py + (it + u) * sy, VecIO<T>::pack(v));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// BatchNorm apply fused with the MaxPool2d(3, 2, 1) that reads its output (VoVNet: every stage opens with the pool of
// the previous stage's last unit, reference backbones/vovnet.py:94), round 4.
//   forward:  one thread per pooled pixel and 16-byte channel chunk normalises the nine pre-activations of its window
//             (clamped loads, the rounding of the unfused path), writes the pooled maximum + arg-max tap AND the unit's
//             output y for the four pixels (2ho, 2wo) .. (2ho+1, 2wo+1) it owns -- every pixel has exactly one owner.
//             The separate pool pass (read y: 822 MB behind VoVNet-39's stem at batch 256) disappears.
//   backward: the unit's gradient dy is never formed: the BatchNorm-backward reduce / apply passes gather it from the
//             pooled gradient and the arg-max taps (at most four windows contain a pixel) -- a quarter of the bytes, and
//             the pool's backward pass (write dy) disappears.  Only when nothing else contributes to dy.
// ---------------------------------------------------------------------------------
// BatchNorm backward of such a unit from the POOLED gradient.  One thread per pooled position (ho, wo) and 16-byte
// channel chunk handles the four pixels (2ho + a, 2wo + b) it owns: the windows that contain them are (ho + da, wo + db),
// da <= a, db <= b -- four pooled gradients + arg-max words serve four pixels (a per-pixel gather loads four per pixel and
// measured SLOWER than the separate pool backward: bn_bwd_reduce 1.69 -> 2.19 ms, bn_bwd_apply 2.43 -> 3.09 ms on
// VoVNet-39).  d(y) of a pixel = sum of the windows whose arg-max tap it is, rounded where the unfused pool backward
// stores it.  APPLY: dz = a*g - b*z + d; else the reduction sum g, sum g*(z - mean) (folded as in bn_bwd_reduce_kernel).
template <typename T, bool APPLY>
__global__ void __launch_bounds__(kThreads)
bn_bwd_pool_kernel(const T* __restrict__ dp, int lddp, const uint8_t* __restrict__ amax, const T* __restrict__ z, int ldz,
                   const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ c2,
                   const float* __restrict__ invstd, T* __restrict__ dz, int lddz, int H, int W, int Ho, int Wo, int C, long Mo,
                   RowMap rm, int relu, float* __restrict__ sums) {
    constexpr int EPC = VecIO<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) float sred[];
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    const int tc = t % rm.CT;
    const int Wc = rm.CT * EPC;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    const int rep = blockIdx.x % kStatReplicas;
    for (int cbase = 0; cbase < rm.CPR; cbase += rm.CT) {
        const int col = cbase + tc;
        const bool active = (r < rm.RT) && (col < rm.CPR);
        float s1[EPC], s2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
        if (active) {
            float sc[EPC], sf[EPC], ca[EPC], cb[EPC], cd[EPC];  // REDUCE: cb = mean
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int c = col * EPC + e;
                sc[e] = scale[c], sf[e] = shift[c];
                if (APPLY) ca[e] = c2[c], cb[e] = c2[C + c], cd[e] = c2[2 * C + c];
                else ca[e] = 0.f, cb[e] = c2[c], cd[e] = 0.f;
            }
            // (wo, ho, b) of the thread's first pooled row once (64-bit divisions), then advanced by additions: a step is
            // rm.RT rows = qstep full rows of Wo + rstep columns (the per-iteration divisions and the twelve 64-bit
            // multiply chains of the load addresses were a quarter of this loop's vector issue time: 47 v_mul_lo_u32 +
            // 32 v_mad_u64_u32 per iteration)
            int wo = (int)(row0 % Wo);
            long t2_ = row0 / Wo;
            int ho = (int)(t2_ % Ho);
            long b = t2_ / Ho;
            const int qstep = rm.RT / Wo, rstep = rm.RT - qstep * Wo;
            for (int it = 0; it < rm.iters; ++it) {
                const long row = row0 + (long)it * rm.RT;
                if (row >= Mo) break;
                // the four windows (clamped coordinates: every load unconditional) and the four owned pixels: one base
                // offset each, the neighbours by clamped deltas
                uint4 gw[4], vz[4];
                unsigned aw[4][EPC / 4];
                const int dwo = wo + 1 < Wo ? 1 : 0, dho = ho + 1 < Ho ? Wo : 0;
                const int dpw = 2 * wo + 1 < W ? 1 : 0, dph = 2 * ho + 1 < H ? W : 0;
                const T* dp0 = dp + row * lddp + col * EPC;
                const uint8_t* am0 = amax + row * C + col * EPC;
                const T* z0 = z + ((b * H + 2 * ho) * W + 2 * wo) * ldz + col * EPC;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int od = ((k >> 1) ? dho : 0) + ((k & 1) ? dwo : 0);
                    gw[k] = ld16(dp0 + (long)od * lddp);
                    const unsigned* ap = (const unsigned*)(am0 + (long)od * C);
#pragma unroll
                    for (int q = 0; q < EPC / 4; ++q) aw[k][q] = ap[q];
                    const int pd = ((k >> 1) ? dph : 0) + ((k & 1) ? dpw : 0);
                    vz[k] = ld16(z0 + (long)pd * ldz);
                }
                float gf[4][EPC];
#pragma unroll
                for (int k = 0; k < 4; ++k) VecIO<T>::unpack(gw[k], gf[k]);
#pragma unroll
                for (int pk = 0; pk < 4; ++pk) {  // pixel (a, b) = (pk >> 1, pk & 1)
                    const int a = pk >> 1, bb = pk & 1;
                    const int h = 2 * ho + a, w = 2 * wo + bb;
                    if (h >= H || w >= W) continue;
                    float acc[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {  // window (ho + da, wo + db)
                        const int da = k >> 1, db = k & 1;
                        if (da > a || db > bb) continue;
                        if (ho + da >= Ho || wo + db >= Wo) continue;
                        const unsigned tap = (unsigned)((a - 2 * da + 1) * 3 + (bb - 2 * db + 1));
#pragma unroll
                        for (int e = 0; e < EPC; ++e)
                            if (((aw[k][e >> 2] >> (8 * (e & 3))) & 0xffu) == tap) acc[e] += gf[k][e];
                    }
                    float g[EPC], zz[EPC];
                    VecIO<T>::unpack(VecIO<T>::pack(acc), g);  // (the rounding of the stored d(y))
                    VecIO<T>::unpack(vz[pk], zz);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float gg = (!relu || fmaf(zz[e], sc[e], sf[e]) > 0.f) ? g[e] : 0.f;
                        if (APPLY) {
                            g[e] = fmaf(ca[e], gg, fmaf(-cb[e], zz[e], cd[e]));
                        } else {
                            s1[e] += gg;
                            s2[e] = fmaf(gg, zz[e] - cb[e], s2[e]);
                        }
                    }
                    if (APPLY) st16(dz + ((b * H + h) * W + w) * lddz + col * EPC, VecIO<T>::pack(g));
                }
                wo += rstep, ho += qstep;
                if (wo >= Wo) wo -= Wo, ++ho;
                while (ho >= Ho) ho -= Ho, ++b;
            }
        }
        if constexpr (!APPLY) {
            // fold of the row lanes: as bn_bwd_reduce_kernel (in-wave shuffles where CT is a power of two below 64)
            const bool inwave = rm.CT < 64 && (rm.CT & (rm.CT - 1)) == 0;
            int rows_l = rm.RT, r_l = r;
            if (inwave) {
                for (int off = rm.CT; off < 64; off <<= 1) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        s1[e] += __shfl_xor(s1[e], off, 64);
                        s2[e] += __shfl_xor(s2[e], off, 64);
                    }
                }
                rows_l = kThreads / 64;
                r_l = (t & 63) < rm.CT ? (t >> 6) : -1;
            } else if (r >= rm.RT) {
                r_l = -1;
            }
            if (r_l >= 0) {
                float4* d1 = (float4*)(sred + ((long)(r_l * 2 + 0) * Wc + tc * EPC));
                float4* d2 = (float4*)(sred + ((long)(r_l * 2 + 1) * Wc + tc * EPC));
#pragma unroll
                for (int q = 0; q < EPC / 4; ++q) {
                    d1[q] = make_float4(s1[4 * q], s1[4 * q + 1], s1[4 * q + 2], s1[4 * q + 3]);
                    d2[q] = make_float4(s2[4 * q], s2[4 * q + 1], s2[4 * q + 2], s2[4 * q + 3]);
                }
            }
            __syncthreads();
            for (int i = t; i < 2 * Wc; i += kThreads) {
                const int which = i / Wc, lc = i % Wc;
                const int c = cbase * EPC + lc;
                if (c < C) {
                    float acc = 0.f;
                    for (int rr = 0; rr < rows_l; ++rr) acc += sred[(long)(rr * 2 + which) * Wc + lc];
                    if (which) acc *= invstd[c];
                    vt_stat_add(sums, ((long)rep * 2 + which) * C + c, acc);
                }
            }
            __syncthreads();
        }
    }
}

template <typename T, bool kRes>
__global__ void __launch_bounds__(kThreads)
bn_act_apply_pool_kernel(const T* __restrict__ z, int ldz, const float* __restrict__ scale, const float* __restrict__ shift,
                         const T* __restrict__ res, int ldr, T* __restrict__ y, int ldy, T* __restrict__ pooled, int ldp,
                         uint8_t* __restrict__ amax, int H, int W, int Ho, int Wo, int C, long Mo, RowMap rm, int relu) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        float sc[EPC], sf[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) sc[e] = scale[col * EPC + e], sf[e] = shift[col * EPC + e];
        int wo = (int)(row0 % Wo);
        long t2_ = row0 / Wo;
        int ho = (int)(t2_ % Ho);
        long b = t2_ / Ho;
        const int qstep = rm.RT / Wo, rstep = rm.RT - qstep * Wo;
        for (int it = 0; it < rm.iters; ++it) {
            const long row = row0 + (long)it * rm.RT;
            if (row >= Mo) break;
            uint4 rz[9], rr[kRes ? 9 : 1];
            // the window's nine pixels from ONE base offset (its centre, always inside the map) and clamped row / column
            // deltas: additions instead of nine 64-bit multiply chains
            const long pc = (b * H + 2 * ho) * W + 2 * wo;
            const int dh_[3] = {ho > 0 ? -W : 0, 0, 2 * ho + 1 < H ? W : 0};
            const int dw_[3] = {wo > 0 ? -1 : 0, 0, 2 * wo + 1 < W ? 1 : 0};
            const T* zc = z + pc * ldz + col * EPC;
            const T* rc = kRes ? res + pc * ldr + col * EPC : nullptr;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int dpx = dh_[k / 3] + dw_[k % 3];
                rz[k] = ld16(zc + (long)dpx * ldz);
                if (kRes) rr[k] = ld16(rc + (long)dpx * ldr);
            }
            float best[EPC];
            int bi[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) best[e] = -INFINITY, bi[e] = 0;
            bool first = true;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int h = ho * 2 - 1 + k / 3, w = wo * 2 - 1 + k % 3;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                    float v[EPC];
                    VecIO<T>::unpack(rz[k], v);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        v[e] = fmaf(v[e], sc[e], sf[e]);
                        v[e] = relu ? fmaxf(v[e], 0.f) : v[e];
                    }
                    if (kRes) {
                        float q[EPC];
                        VecIO<T>::unpack(rr[kRes ? k : 0], q);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[e] += q[e];
                    }
                    const uint4 yv = VecIO<T>::pack(v);  // what the unfused normalise pass stores ...
                    if (k / 3 >= 1 && k % 3 >= 1) st16(y + ((b * H + h) * W + w) * ldy + col * EPC, yv);  // (an owned pixel)
                    VecIO<T>::unpack(yv, v);             // ... and the pool reads
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        // ATen: first max in (kh, kw) scan order wins; NaN propagates
                        if (first || v[e] > best[e] || v[e] != v[e]) {
                            best[e] = v[e];
                            bi[e] = k;
                        }
                    }
                    first = false;
                }
            }
            st16(pooled + row * ldp + col * EPC, VecIO<T>::pack(best));
            unsigned* ap = (unsigned*)(amax + row * C + col * EPC);
#pragma unroll
            for (int q = 0; q < EPC / 4; ++q)
                ap[q] = (unsigned)bi[4 * q] | ((unsigned)bi[4 * q + 1] << 8) | ((unsigned)bi[4 * q + 2] << 16) | ((unsigned)bi[4 * q + 3] << 24);
            wo += rstep, ho += qstep;
            if (wo >= Wo) wo -= Wo, ++ho;
            while (ho >= Ho) ho -= Ho, ++b;
        }
    }
}

// ---------------------------------------------------------------------------------
// Finalize INSIDE the consuming launch (round 6).  The 134 single-workgroup finalize launches of a CSPDarknet-53 step sit on
// the critical path between a producer of statistics and the streaming pass that needs the coefficients: skipping them
// (diagnostic build) shortens the 19.8 ms step by 1.17 ms, 9 - 13 us each.  Any hand-off INSIDE a launch -- first workgroups
// finalize and publish, the others poll (the first form of these kernels: +2.3 ms per step); the producer's last workgroup
// finalizes behind a ticket (NOTEBOOK R6.10: +-0, removed) -- costs the dependent memory-side round trips it is made of, as much as
// the launch boundary it replaces.  Here nothing is handed over: the sums are complete when the streaming launch starts, so
// EVERY workgroup finalizes the (at most 128) channels of its own channel group for itself -- a thread per (channel, sum),
// its 16 replicas x 2 limbs in one round of plain loads (64 KB per workgroup, L2 hits after the first workgroup of an XCD),
// the arithmetic of bn_finalize_kernel / bn_bwd_finalize_kernel bit for bit (vt_fin_fwd_channel / vt_fin_bwd_channel) --
// and keeps the coefficients in LDS; the workgroups of row block 0 also store them (and the running statistics, d(gamma),
// d(beta)) for the later passes.  The grid is cut for ~1536 workgroups instead of 4096, so that the redundant reads stay
// small beside the tensor traffic (step: 1024 -> 19.50, 1536 -> 19.47, 3072 -> 19.59, 4096 -> 19.81 ms).
// ---------------------------------------------------------------------------------
constexpr int kFinCg = kThreads / 2;  // channels per channel group: a thread pair per channel

// the channel group of C channels: the largest divisor <= 128 that is a multiple of the 16-byte chunk (0: none)
__host__ inline int fin_group(int C, int epc) {
    for (int g = kFinCg / epc * epc; g >= epc; g -= epc)
        if (C % g == 0) return (g >= 32 || g == C) ? g : 0;
    return 0;
}

// y = [relu](z*scale + shift) [+ residual]; grid (row blocks, channel groups of Cg channels)
template <typename T, bool kRes>
__global__ void __launch_bounds__(kThreads)
bn_fin_apply_kernel(const VtFinFwd f, const T* __restrict__ z, int ldz, const T* __restrict__ res, int ldr, T* __restrict__ y,
                    int ldy, long M, RowMap rm, int relu, int Cg) {
    constexpr int EPC = VecIO<T>::EPC;
    __shared__ float s_sc[kFinCg], s_sf[kFinCg];
    const int t = threadIdx.x;
    const int cg0 = blockIdx.y * Cg;
    vt_pair_sums<false>(f.stats, f.C, cg0, cg0 + Cg, [&](int c) { return vt_fin_fwd_pre(f, c); },
                        [&](int c, double s, double ss, const VtFinFwdPre& p) {
                            float sc, sf;
                            vt_fin_fwd_channel(f, c, s, ss, p.g, p.b, p.rm, p.rv, sc, sf, blockIdx.x == 0);
                            s_sc[c - cg0] = sc, s_sf[c - cg0] = sf;
                        });
    __syncthreads();
    z += cg0, y += cg0;
    if (kRes) res += cg0;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)(rm.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        float sc[EPC], sf[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) sc[e] = s_sc[col * EPC + e], sf[e] = s_sf[col * EPC + e];
        const T* pz = z + row0 * ldz + col * EPC;
        const T* pr = kRes ? res + row0 * ldr + col * EPC : nullptr;
        T* py = y + row0 * ldy + col * EPC;
        const long sz = (long)rm.RT * ldz, sr = (long)rm.RT * ldr, sy = (long)rm.RT * ldy;
        for (int it = 0; it < rm.iters; it += kUnroll) {
            uint4 vz[kUnroll], vr[kUnroll];
            bool ok[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                ok[u] = (it + u < rm.iters) && (row0 + (long)(it + u) * rm.RT < M);
                vz[u] = vr[u] = make_uint4(0, 0, 0, 0);
                if (ok[u]) {
                    vz[u] = ld16(pz + (it + u) * sz);
                    if (kRes) vr[u] = ld16(pr + (it + u) * sr);
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                float v[EPC];
                VecIO<T>::unpack(vz[u], v);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    v[e] = fmaf(v[e], sc[e], sf[e]);
                    v[e] = relu ? fmaxf(v[e], 0.f) : v[e];
                }
                if (kRes) {
                    float rr[EPC];
                    VecIO<T>::unpack(vr[u], rr);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] += rr[e];
                }
                if (ok[u]) st16(py + (it + u) * sy, VecIO<T>::pack(v));
            }
        }
    }
}

// dz = a*g - b*z + d; grid (row blocks, channel groups of Cg channels)
template <typename T>
__global__ void __launch_bounds__(kThreads)
bn_bwd_fin_apply_kernel(const VtFinBwd f, const T* __restrict__ dy, int lddy, const T* __restrict__ z, int ldz,
                        const float* __restrict__ scale, const float* __restrict__ shift, T* __restrict__ dz, int lddz, long M,
                        RowMap rm, int relu, int Cg) {
    constexpr int EPC = VecIO<T>::EPC;
    __shared__ float s_a[kFinCg], s_b[kFinCg], s_d[kFinCg];
    const int t = threadIdx.x;
    const int cg0 = blockIdx.y * Cg;
    vt_pair_sums<false>(f.sums, f.C, cg0, cg0 + Cg, [&](int c) { return vt_fin_bwd_pre(f, c); },
                        [&](int c, double s1, double s2, const VtFinBwdPre& p) {
                            float b, d;
                            vt_fin_bwd_channel(f, c, s1, s2, p.a, p.mu, p.istd, p.dg, p.db, b, d, blockIdx.x == 0);
                            s_a[c - cg0] = p.a, s_b[c - cg0] = b, s_d[c - cg0] = d;
                        });
    __syncthreads();
    dy += cg0, z += cg0, dz += cg0, scale += cg0, shift += cg0;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)(rm.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        float sc[EPC], sf[EPC], ca[EPC], cb[EPC], cd[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = col * EPC + e;
            sc[e] = scale[c];
            sf[e] = shift[c];
            ca[e] = s_a[c], cb[e] = s_b[c], cd[e] = s_d[c];
        }
        for (int it = 0; it < rm.iters; it += kUnroll) {
            uint4 vg[kUnroll], vz[kUnroll];
            bool ok[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const long row = row0 + (long)(it + u) * rm.RT;
                ok[u] = (it + u < rm.iters) && row < M;
                vg[u] = vz[u] = make_uint4(0, 0, 0, 0);
                if (ok[u]) {
                    vg[u] = ld16(dy + row * lddy + col * EPC);
                    vz[u] = ld16(z + row * ldz + col * EPC);
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (!ok[u]) continue;
                const long row = row0 + (long)(it + u) * rm.RT;
                float g[EPC], zz[EPC];
                VecIO<T>::unpack(vg[u], g);
                VecIO<T>::unpack(vz[u], zz);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float gg = (!relu || fmaf(zz[e], sc[e], sf[e]) > 0.f) ? g[e] : 0.f;
                    g[e] = fmaf(ca[e], gg, fmaf(-cb[e], zz[e], cd[e]));
                }
                st16(dz + row * lddz + col * EPC, VecIO<T>::pack(g));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// BN backward, pass 1: per-channel sum(g) and sum(g*xhat), g = dy * [z*scale+shift > 0]
// ---------------------------------------------------------------------------------
template <typename T, bool GEN = false>
__global__ void __launch_bounds__(kThreads)
bn_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ z, int ldz,
                     const float* __restrict__ scale, const float* __restrict__ shift,
                     const float* __restrict__ mean, const float* __restrict__ invstd, long M, int C,
                     RowMap rm, int relu_flags, float* __restrict__ sums, int Ctot) {
    const int relu = relu_flags & 1;  // bit 1: dev switch, LDS staging of every row lane (the pre-round-2 fold)
    const int act = relu_flags >> 4;  // (GEN: the activation code)
    constexpr int EPC = VecIO<T>::EPC;
    // [RT][2][CT*EPC] partial sums: each thread parks its 2*EPC partials, then the first
    // 2*CT*EPC threads fold the RT row lanes (no LDS atomics: with RT rows per column they
    // serialised RT-way and cost as much as the streaming itself on the 28x28..7x7 maps).
    extern __shared__ __attribute__((aligned(16))) float sred[];
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    const int tc = t % rm.CT;
    const int W = rm.CT * EPC;  // channels covered per pass
    const long row0 = (long)(rm.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rm.RT * rm.iters + r;
    const int rep = blockIdx.x % kStatReplicas;
    // blockIdx.y: channel group (C channels each of Ctot; round 4).  On the small maps a block's 2 * C fixed-point atomics
    // were half of the launch (C = 512 @7x7: 242 blocks x 1024 atomics, 14.6 us of which ~7 in that tail; twice the blocks:
    // 21.6 us); with C / 64 channel groups and as many fewer row splits a block adds 128.
    const int cg0 = blockIdx.y * rm.CPR;  // first 16-byte chunk of this group
    dy += (long)cg0 * EPC, z += (long)cg0 * EPC;
    scale += cg0 * EPC, shift += cg0 * EPC, mean += cg0 * EPC, invstd += cg0 * EPC;
    for (int cbase = 0; cbase < rm.CPR; cbase += rm.CT) {
        const int col = cbase + tc;
        const bool active = (r < rm.RT) && (col < rm.CPR);
        float s1[EPC], s2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
        if (active) {
            float sc[EPC], sf[EPC], mu[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                sc[e] = scale[col * EPC + e];
                sf[e] = shift[col * EPC + e];
                mu[e] = mean[col * EPC + e];
            }
            const T* pg = dy + row0 * lddy + col * EPC;
            const T* pz = z + row0 * ldz + col * EPC;
            const long sg = (long)rm.RT * lddy, sz = (long)rm.RT * ldz;
            for (int it = 0; it < rm.iters; it += kUnroll) {
                uint4 vg[kUnroll], vz[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const long row = row0 + (long)(it + u) * rm.RT;
                    vg[u] = vz[u] = make_uint4(0, 0, 0, 0);  // zero dy contributes nothing
                    if ((it + u < rm.iters) && row < M) {
                        vg[u] = ld16(pg + (it + u) * sg);
                        vz[u] = ld16(pz + (it + u) * sz);
                    }
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    float g[EPC], zz[EPC];
                    VecIO<T>::unpack(vg[u], g);
                    VecIO<T>::unpack(vz[u], zz);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        float gg;
                        if constexpr (GEN) gg = g[e] * act_grad(act, fmaf(zz[e], sc[e], sf[e]));
                        else gg = (!relu || fmaf(zz[e], sc[e], sf[e]) > 0.f) ? g[e] : 0.f;
                        s1[e] += gg;
                        s2[e] = fmaf(gg, zz[e] - mu[e], s2[e]);  // invstd applied once, below
                    }
                }
            }
        }
        // Rows of the fold: with CT a power of two below 64 a wave holds 64 / CT row lanes of every column it
        // touches; those are folded in registers first (xor shuffles), so the LDS staging shrinks from RT to
        // kThreads / 64 rows -- 4 KiB instead of 16 KiB at C = 128.  (This kernel runs beside the filter gradients
        // of the side stream, whose workgroups hold most of a CU's LDS: at 16 KiB per block its occupancy, not HBM,
        // set its in-step time.)
        const bool inwave = rm.CT < 64 && (rm.CT & (rm.CT - 1)) == 0 && !(relu_flags & 2);
        int rows_l = rm.RT, r_l = r;
        if (inwave) {
            for (int off = rm.CT; off < 64; off <<= 1) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    s1[e] += __shfl_xor(s1[e], off, 64);
                    s2[e] += __shfl_xor(s2[e], off, 64);
                }
            }
            rows_l = kThreads / 64;
            r_l = (t & 63) < rm.CT ? (t >> 6) : -1;  // one writer per (wave, column)
        } else if (r >= rm.RT) {
            r_l = -1;
        }
        if (r_l >= 0) {
            float4* d1 = (float4*)(sred + ((long)(r_l * 2 + 0) * W + tc * EPC));
            float4* d2 = (float4*)(sred + ((long)(r_l * 2 + 1) * W + tc * EPC));
#pragma unroll
            for (int q = 0; q < EPC / 4; ++q) {
                d1[q] = make_float4(s1[4 * q], s1[4 * q + 1], s1[4 * q + 2], s1[4 * q + 3]);
                d2[q] = make_float4(s2[4 * q], s2[4 * q + 1], s2[4 * q + 2], s2[4 * q + 3]);
            }
        }
        __syncthreads();
        for (int i = t; i < 2 * W; i += kThreads) {
            const int which = i / W, lc = i % W;
            const int c = cbase * EPC + lc;
            if (c < C) {
                float acc = 0.f;
                for (int rr = 0; rr < rows_l; ++rr) acc += sred[(long)(rr * 2 + which) * W + lc];
                if (which) acc *= invstd[c];
                vt_stat_add(sums, ((long)rep * 2 + which) * Ctot + cg0 * EPC + c, acc);
            }
        }
        __syncthreads();
    }
}

__global__ void bn_bwd_finalize_kernel(const VtFinBwd f) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= f.C) return;
    const float a = f.scale[c], mu = f.mean[c], istd = f.invstd[c];
    float dg = 0.f, db = 0.f;
    if (f.dgamma) dg = f.dgamma[c];
    if (f.dbeta) db = f.dbeta[c];
    const double s1 = vt_stat_sum(f.sums, c, 2L * f.C), s2 = vt_stat_sum(f.sums, (long)f.C + c, 2L * f.C);
    float b, d;
    vt_fin_bwd_channel(f, c, s1, s2, a, mu, istd, dg, db, b, d);
}

// dz = a*g - b*z + d
template <typename T, bool GEN = false>
__global__ void __launch_bounds__(kThreads)
bn_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ z, int ldz,
                    const float* __restrict__ scale, const float* __restrict__ shift,
                    const float* __restrict__ coef, T* __restrict__ dz, int lddz, long M, int C, RowMap rm,
                    int relu) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)(rm.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        float sc[EPC], sf[EPC], ca[EPC], cb[EPC], cd[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = col * EPC + e;
            if constexpr (GEN) {  // (no coefficients: dz = dy * act'(z), the unit without a BatchNorm)
                sc[e] = scale ? scale[c] : 1.f;
                sf[e] = shift ? shift[c] : 0.f;
                ca[e] = coef ? coef[c] : 1.f;
                cb[e] = coef ? coef[C + c] : 0.f;
                cd[e] = coef ? coef[2 * C + c] : 0.f;
            } else {
                sc[e] = scale[c];
                sf[e] = shift[c];
                ca[e] = coef[c];
                cb[e] = coef[C + c];
                cd[e] = coef[2 * C + c];
            }
        }
        for (int it = 0; it < rm.iters; it += kUnroll) {
            uint4 vg[kUnroll], vz[kUnroll];
            bool ok[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const long row = row0 + (long)(it + u) * rm.RT;
                ok[u] = (it + u < rm.iters) && row < M;
                vg[u] = vz[u] = make_uint4(0, 0, 0, 0);
                if (ok[u]) {
                    vg[u] = ld16(dy + row * lddy + col * EPC);
                    vz[u] = ld16(z + row * ldz + col * EPC);
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (!ok[u]) continue;
                const long row = row0 + (long)(it + u) * rm.RT;
                float g[EPC], zz[EPC];
                VecIO<T>::unpack(vg[u], g);
                VecIO<T>::unpack(vz[u], zz);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    float gg;
                    if constexpr (GEN) gg = g[e] * act_grad(relu, fmaf(zz[e], sc[e], sf[e]));
                    else gg = (!relu || fmaf(zz[e], sc[e], sf[e]) > 0.f) ? g[e] : 0.f;
                    g[e] = fmaf(ca[e], gg, fmaf(-cb[e], zz[e], cd[e]));
                }
                st16(dz + row * lddz + col * EPC, VecIO<T>::pack(g));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// MaxPool2d(3, 2, 1)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads)
maxpool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, uint8_t* __restrict__ amax,
                   int H, int W, int Ho, int Wo, int C, long Mo, RowMap rm) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        for (int it = 0; it < rm.iters; ++it) {
            const long row = row0 + (long)it * rm.RT;
            if (row >= Mo) break;
            const int wo = (int)(row % Wo);
            const long t2 = row / Wo;
            const int ho = (int)(t2 % Ho);
            const long b = t2 / Ho;
            // all nine taps in flight before the first comparison: loaded unconditionally from clamped coordinates
            // (a dependent, predicated load per tap made the pass 3.5x its HBM time)
            uint4 raw[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int h = min(max(ho * 2 - 1 + k / 3, 0), H - 1), w = min(max(wo * 2 - 1 + k % 3, 0), W - 1);
                raw[k] = ld16(x + ((b * H + h) * W + w) * ldx + col * EPC);
            }
            float best[EPC];
            int bi[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                best[e] = -INFINITY;
                bi[e] = 0;
            }
            bool first = true;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int h = ho * 2 - 1 + k / 3, w = wo * 2 - 1 + k % 3;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                    float v[EPC];
                    VecIO<T>::unpack(raw[k], v);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        // ATen: first max in (kh, kw) scan order wins; NaN propagates
                        if (first || v[e] > best[e] || v[e] != v[e]) {
                            best[e] = v[e];
                            bi[e] = k;
                        }
                    }
                    first = false;
                }
            }
            st16(y + row * ldy + col * EPC, VecIO<T>::pack(best));
            // the EPC tap indices of the chunk as one 4- / 8-byte store
            unsigned* ap = (unsigned*)(amax + row * C + col * EPC);
#pragma unroll
            for (int q = 0; q < EPC / 4; ++q)
                ap[q] = (unsigned)bi[4 * q] | ((unsigned)bi[4 * q + 1] << 8) | ((unsigned)bi[4 * q + 2] << 16) | ((unsigned)bi[4 * q + 3] << 24);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
maxpool_bwd_kernel(const T* __restrict__ dy, int lddy, const uint8_t* __restrict__ amax, T* __restrict__ dx,
                   int lddx, int H, int W, int Ho, int Wo, int C, long Mi, RowMap rm, int accumulate) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        for (int it = 0; it < rm.iters; ++it) {
            const long row = row0 + (long)it * rm.RT;
            if (row >= Mi) break;
            const int w = (int)(row % W);
            const long t2 = row / W;
            const int h = (int)(t2 % H);
            const long b = t2 / H;
            float acc[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
            if (accumulate) VecIO<T>::unpack(ld16(dx + row * lddx + col * EPC), acc);
            // windows (ho, wo) with ho*2-1 <= h <= ho*2+1: one for an even coordinate, two for an odd one.  The loads
            // of all of them (gradient chunk + tap indices) are issued before the first use.
            const int ho_lo = h >> 1, wo_lo = w >> 1;
            uint4 gr[4];
            unsigned ar[4][EPC / 4];
            bool need[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ho = ho_lo + (k >> 1), wo = wo_lo + (k & 1);
                need[k] = (k >> 1) <= (h & 1) && (k & 1) <= (w & 1) && ho < Ho && wo < Wo;
                gr[k] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int q = 0; q < EPC / 4; ++q) ar[k][q] = 0xffffffffu;
                if (need[k]) {
                    const long orow = (b * Ho + ho) * Wo + wo;
                    gr[k] = ld16(dy + orow * lddy + col * EPC);
                    const unsigned* ap = (const unsigned*)(amax + orow * C + col * EPC);
#pragma unroll
                    for (int q = 0; q < EPC / 4; ++q) ar[k][q] = ap[q];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ho = ho_lo + (k >> 1), wo = wo_lo + (k & 1);
                const unsigned tap = (unsigned)((h - (ho * 2 - 1)) * 3 + (w - (wo * 2 - 1)));
                float g[EPC];
                VecIO<T>::unpack(gr[k], g);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (need[k] && ((ar[k][e >> 2] >> (8 * (e & 3))) & 0xffu) == tap) acc[e] += g[e];
            }
            st16(dx + row * lddx + col * EPC, VecIO<T>::pack(acc));
        }
    }
}

// ---------------------------------------------------------------------------------
// global average pool: y[b][c] = mean_hw x[b][hw][c]; one block per (b, column group)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads)
avgpool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int HW, int C, int CT) {
    constexpr int EPC = VecIO<T>::EPC;
    extern __shared__ unsigned long long sfix[];  // [CT*EPC] fixed point
    unsigned long long* sred = sfix;
    const int t = threadIdx.x;
    const int RT = kThreads / CT;
    const int b = blockIdx.x;
    const int col = blockIdx.y * CT + t % CT;
    const int r = t / CT;
    const int CPR = C / EPC;
    for (int i = t; i < CT * EPC; i += kThreads) sred[i] = 0ull;
    __syncthreads();
    if (r < RT && col < CPR) {
        float s[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[e] = 0.f;
        for (int hw = r; hw < HW; hw += RT) {
            float v[EPC];
            VecIO<T>::unpack(ld16(x + ((long)b * HW + hw) * ldx + col * EPC), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) lds_fix_add(sred, (t % CT) * EPC + e, s[e]);
    }
    __syncthreads();
    if (r == 0 && col < CPR) {
        float v[EPC];
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = lds_fix_get(sred, (t % CT) * EPC + e) * inv;
        st16(y + (long)b * ldy + col * EPC, VecIO<T>::pack(v));
    }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
avgpool_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int HW, long M, RowMap rm,
                   int accumulate) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    const float inv = 1.f / (float)HW;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        for (int it = 0; it < rm.iters; ++it) {
            const long row = row0 + (long)it * rm.RT;
            if (row >= M) break;
            const long b = row / HW;
            float g[EPC], a[EPC];
            VecIO<T>::unpack(ld16(dy + b * lddy + col * EPC), g);
#pragma unroll
            for (int e = 0; e < EPC; ++e) a[e] = 0.f;
            if (accumulate) VecIO<T>::unpack(ld16(dx + row * lddx + col * EPC), a);
#pragma unroll
            for (int e = 0; e < EPC; ++e) a[e] += g[e] * inv;
            st16(dx + row * lddx + col * EPC, VecIO<T>::pack(a));
        }
    }
}

// ---------------------------------------------------------------------------------
// nearest x2 / x0.5 resampling fused with the sum that follows it (necks)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads)
resample_fwd_kernel(const T* __restrict__ src, int lds, const T* __restrict__ other, int ldo, T* __restrict__ dst,
                    int ldd, int Hd, int Wd, long M, RowMap rm, int mode) {
    // mode 0 / 1: nearest x2 / x0.5; mode 2 / 3: bilinear x2 / x0.5 (nn.Upsample(mode="bilinear"), align_corners False: a
    // destination index d reads source coordinate (d + 0.5) / scale - 0.5 -- x2: taps (k - 1, k) with weights (0.25, 0.75)
    // for d = 2k, (k, k + 1) with (0.75, 0.25) for d = 2k + 1, indices clamped to the map; x0.5: the mean of 2 x 2 pixels)
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    const bool down = mode & 1;
    const int Hs = down ? 2 * Hd : Hd / 2, Ws = down ? 2 * Wd : Wd / 2;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        for (int it = 0; it < rm.iters; ++it) {
            const long row = row0 + (long)it * rm.RT;
            if (row >= M) break;
            const long b = row / ((long)Hd * Wd);
            const int rem = (int)(row - b * Hd * Wd);
            const int i = rem / Wd, j = rem - i * Wd;
            float v[EPC];
            if (mode < 2) {
                const int si = down ? 2 * i : i >> 1, sj = down ? 2 * j : j >> 1;
                const long srow = (b * Hs + si) * Ws + sj;
                VecIO<T>::unpack(ld16(src + srow * lds + col * EPC), v);
            } else {
                int i0, i1, j0, j1;
                float wi0, wj0;  // weight of tap 0 (tap 1: 1 - that)
                if (down) {
                    i0 = 2 * i, i1 = 2 * i + 1, j0 = 2 * j, j1 = 2 * j + 1, wi0 = 0.5f, wj0 = 0.5f;
                } else {
                    const int ki = i >> 1, kj = j >> 1;
                    i0 = (i & 1) ? ki : max(ki - 1, 0), i1 = (i & 1) ? min(ki + 1, Hs - 1) : ki, wi0 = (i & 1) ? 0.75f : 0.25f;
                    j0 = (j & 1) ? kj : max(kj - 1, 0), j1 = (j & 1) ? min(kj + 1, Ws - 1) : kj, wj0 = (j & 1) ? 0.75f : 0.25f;
                }
                float a[EPC], c[EPC], d[EPC], e4[EPC];
                VecIO<T>::unpack(ld16(src + ((b * Hs + i0) * Ws + j0) * lds + col * EPC), a);
                VecIO<T>::unpack(ld16(src + ((b * Hs + i0) * Ws + j1) * lds + col * EPC), c);
                VecIO<T>::unpack(ld16(src + ((b * Hs + i1) * Ws + j0) * lds + col * EPC), d);
                VecIO<T>::unpack(ld16(src + ((b * Hs + i1) * Ws + j1) * lds + col * EPC), e4);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    v[e] = wi0 * (wj0 * a[e] + (1.f - wj0) * c[e]) + (1.f - wi0) * (wj0 * d[e] + (1.f - wj0) * e4[e]);
            }
            if (other) {
                float o[EPC];
                VecIO<T>::unpack(ld16(other + row * ldo + col * EPC), o);
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] += o[e];
            }
            st16(dst + row * ldd + col * EPC, VecIO<T>::pack(v));
        }
    }
}

// one thread row = one pixel of dsrc
template <typename T>
__global__ void __launch_bounds__(kThreads)
resample_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dsrc, int lds, int Hd, int Wd, long Ms,
                    RowMap rm, int mode, int accumulate) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    const bool down = mode & 1;
    const int Hs = down ? 2 * Hd : Hd / 2, Ws = down ? 2 * Wd : Wd / 2;
    // bilinear x2: the weight source index s carries in destination index d (the two taps of d, clamped, that land on s)
    auto wup = [](int d, int s, int n) -> float {
        const int k = d >> 1;
        const int t0 = (d & 1) ? k : max(k - 1, 0), t1 = (d & 1) ? min(k + 1, n - 1) : k;
        const float w0 = (d & 1) ? 0.75f : 0.25f;
        return (t0 == s ? w0 : 0.f) + (t1 == s ? 1.f - w0 : 0.f);
    };
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        for (int it = 0; it < rm.iters; ++it) {
            const long row = row0 + (long)it * rm.RT;
            if (row >= Ms) break;
            const long b = row / ((long)Hs * Ws);
            const int rem = (int)(row - b * Hs * Ws);
            const int i = rem / Ws, j = rem - i * Ws;
            float a[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) a[e] = 0.f;
            if (accumulate) VecIO<T>::unpack(ld16(dsrc + row * lds + col * EPC), a);
            if (mode == 0) {
#pragma unroll
                for (int di = 0; di < 2; ++di)
#pragma unroll
                    for (int dj = 0; dj < 2; ++dj) {
                        float g[EPC];
                        const long drow = (b * Hd + 2 * i + di) * Wd + 2 * j + dj;
                        VecIO<T>::unpack(ld16(dy + drow * lddy + col * EPC), g);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) a[e] += g[e];
                    }
            } else if (mode == 1) {
                if (!(i & 1) && !(j & 1)) {
                    float g[EPC];
                    const long drow = (b * Hd + (i >> 1)) * Wd + (j >> 1);
                    VecIO<T>::unpack(ld16(dy + drow * lddy + col * EPC), g);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) a[e] += g[e];
                }
            } else if (mode == 2) {  // destination rows 2i-1 .. 2i+2 (columns alike) may read this source pixel
                for (int di = max(2 * i - 1, 0); di <= min(2 * i + 2, Hd - 1); ++di) {
                    const float wi = wup(di, i, Hs);
                    if (wi == 0.f) continue;
                    for (int dj = max(2 * j - 1, 0); dj <= min(2 * j + 2, Wd - 1); ++dj) {
                        const float w = wi * wup(dj, j, Ws);
                        if (w == 0.f) continue;
                        float g[EPC];
                        VecIO<T>::unpack(ld16(dy + ((b * Hd + di) * Wd + dj) * lddy + col * EPC), g);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) a[e] = fmaf(w, g[e], a[e]);
                    }
                }
            } else if ((i >> 1) < Hd && (j >> 1) < Wd) {  // x0.5 bilinear: a quarter of the destination pixel's gradient
                float g[EPC];
                VecIO<T>::unpack(ld16(dy + ((b * Hd + (i >> 1)) * Wd + (j >> 1)) * lddy + col * EPC), g);
#pragma unroll
                for (int e = 0; e < EPC; ++e) a[e] = fmaf(0.25f, g[e], a[e]);
            }
            st16(dsrc + row * lds + col * EPC, VecIO<T>::pack(a));
        }
    }
}

// ---------------------------------------------------------------------------------
// ESE gate: y = x * hardsigmoid(s[b][c]) [+ residual]
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float hsig(float v) { return fminf(fmaxf(v * (1.f / 6.f) + 0.5f, 0.f), 1.f); }

template <typename T>
__global__ void __launch_bounds__(kThreads)
ese_fwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ s, int lds, const T* __restrict__ res,
               int ldr, T* __restrict__ y, int ldy, int HW, long M, RowMap rm) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        for (int it = 0; it < rm.iters; ++it) {
            const long row = row0 + (long)it * rm.RT;
            if (row >= M) break;
            const long b = row / HW;
            float v[EPC], g[EPC];
            VecIO<T>::unpack(ld16(x + row * ldx + col * EPC), v);
            VecIO<T>::unpack(ld16(s + b * lds + col * EPC), g);
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] *= hsig(g[e]);
            if (res) {
                float rr[EPC];
                VecIO<T>::unpack(ld16(res + row * ldr + col * EPC), rr);
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] += rr[e];
            }
            st16(y + row * ldy + col * EPC, VecIO<T>::pack(v));
        }
    }
}

// one block per (b, column group): dx (=|+=) dy*hsig(s); ds[b][c] = hsig'(s) * sum_hw dy*x
template <typename T>
__global__ void __launch_bounds__(kThreads)
ese_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx, const T* __restrict__ s,
               int lds, T* __restrict__ dx, int lddx, float* __restrict__ ds, int HW, int C, int CT,
               int accumulate) {
    constexpr int EPC = VecIO<T>::EPC;
    extern __shared__ unsigned long long sfix[];  // [CT*EPC] fixed point
    unsigned long long* sred = sfix;
    const int t = threadIdx.x;
    const int RT = kThreads / CT;
    const int b = blockIdx.x;
    const int col = blockIdx.y * CT + t % CT;
    const int r = t / CT;
    const int CPR = C / EPC;
    for (int i = t; i < CT * EPC; i += kThreads) sred[i] = 0ull;
    __syncthreads();
    float sv[EPC];
    if (r < RT && col < CPR) {
        VecIO<T>::unpack(ld16(s + (long)b * lds + col * EPC), sv);
        float acc[EPC], gate[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            acc[e] = 0.f;
            gate[e] = hsig(sv[e]);
        }
        for (int hw = r; hw < HW; hw += RT) {
            const long row = (long)b * HW + hw;
            float g[EPC], xv[EPC], o[EPC];
            VecIO<T>::unpack(ld16(dy + row * lddy + col * EPC), g);
            VecIO<T>::unpack(ld16(x + row * ldx + col * EPC), xv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = 0.f;
            if (accumulate) VecIO<T>::unpack(ld16(dx + row * lddx + col * EPC), o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                acc[e] += g[e] * xv[e];
                o[e] += g[e] * gate[e];
            }
            st16(dx + row * lddx + col * EPC, VecIO<T>::pack(o));
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) lds_fix_add(sred, (t % CT) * EPC + e, acc[e]);
    }
    __syncthreads();
    if (r == 0 && col < CPR) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float d = (sv[e] > -3.f && sv[e] < 3.f) ? (1.f / 6.f) : 0.f;
            ds[(long)b * C + col * EPC + e] = lds_fix_get(sred, (t % CT) * EPC + e) * d;
        }
    }
}

// ---------------------------------------------------------------------------------
// column sums (bias gradients)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads)
colsum_kernel(const T* __restrict__ a, int lda, long M, int C, RowMap rm, float* __restrict__ out, int fixed) {
    constexpr int EPC = VecIO<T>::EPC;
    const int t = threadIdx.x;
    const int r = t / rm.CT;
    if (r >= rm.RT) return;
    const long row0 = (long)blockIdx.x * rm.RT * rm.iters + r;
    for (int col = t % rm.CT; col < rm.CPR; col += rm.CT) {
        float s[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[e] = 0.f;
        for (int it = 0; it < rm.iters; ++it) {
            const long row = row0 + (long)it * rm.RT;
            if (row >= M) break;
            float v[EPC];
            VecIO<T>::unpack(ld16(a + row * lda + col * EPC), v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            if (fixed)
                vt_stat_add(out, col * EPC + e, s[e]);
            else
                atomicAdd(&out[col * EPC + e], s[e]);
        }
    }
}

// ---------------------------------------------------------------------------------
// softmax cross entropy with label smoothing, mean reduction, fused gradient
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return (float)(*p); }

// Validation (classifier.py:97-109): cross entropy WITHOUT label smoothing and top-1 hits of one batch.
// out[0] += sum over rows of the row loss, out[1] += rows whose arg-max (first maximum, as torch.argmax) is the label,
// out[2] += rows.  f32 adds of at most a few hundred terms per call; the caller zeroes `out` (or keeps accumulating over the
// batches of an epoch) and, data parallel, all-reduces the three sums (the `sync_dist=True` of classifier.py:104).
template <typename T>
__global__ void __launch_bounds__(kThreads)
xent_eval_kernel(const T* __restrict__ logits, int ldl, const int64_t* __restrict__ labels, float* __restrict__ out, int B,
                 int N) {
    __shared__ float sv[kThreads / 64];
    __shared__ int si[kThreads / 64];
    __shared__ float sbc;
    const int b = blockIdx.x, t = threadIdx.x;
    const T* lp = logits + (long)b * ldl;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int i = t; i < N; i += kThreads) {
        const float v = ldf(lp + i);
        if (v > mx || (v == mx && i < am)) mx = v, am = i;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(am, o, 64);
        if (ov > mx || (ov == mx && oi < am)) mx = ov, am = oi;
    }
    if ((t & 63) == 0) sv[t >> 6] = mx, si[t >> 6] = am;
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < kThreads / 64; ++i)
            if (sv[i] > mx || (sv[i] == mx && si[i] < am)) mx = sv[i], am = si[i];
        sbc = mx;
        si[0] = am;
    }
    __syncthreads();
    mx = sbc;
    am = si[0];
    float se = 0.f;
    for (int i = t; i < N; i += kThreads) se += expf(ldf(lp + i) - mx);
    se = wave_sum(se);
    __syncthreads();
    if ((t & 63) == 0) sv[t >> 6] = se;
    __syncthreads();
    if (t == 0) {
        float tse = 0.f;
        for (int i = 0; i < kThreads / 64; ++i) tse += sv[i];
        const int64_t yb = labels[b];
        const bool bad = yb < 0 || yb >= (int64_t)N;  // (torch raises; here the loss sum becomes NaN)
        const float l = bad ? __builtin_nanf("") : mx + logf(tse) - ldf(lp + yb);
        atomicAdd(out + 0, l);
        if (!bad && (int64_t)am == yb) atomicAdd(out + 1, 1.0f);
        atomicAdd(out + 2, 1.0f);
    }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
xent_kernel(const T* __restrict__ logits, int ldl, const int64_t* __restrict__ labels, float eps,
            float grad_scale, float* __restrict__ loss, T* __restrict__ dlogits, int lddl, int B, int N,
            const float* __restrict__ mix) {
    __shared__ float sred[kThreads / 64];
    __shared__ float sbc;
    const int b = blockIdx.x, t = threadIdx.x;
    const T* lp = logits + (long)b * ldl;
    float mx = -INFINITY;
    for (int i = t; i < N; i += kThreads) mx = fmaxf(mx, ldf(lp + i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) sred[t >> 6] = mx;
    __syncthreads();
    if (t == 0) {
        float m = sred[0];
        for (int i = 1; i < kThreads / 64; ++i) m = fmaxf(m, sred[i]);
        sbc = m;
    }
    __syncthreads();
    mx = sbc;
    __syncthreads();
    float se = 0.f, sz = 0.f;
    for (int i = t; i < N; i += kThreads) {
        const float v = ldf(lp + i);
        se += expf(v - mx);
        sz += v;
    }
    se = wave_sum(se);
    sz = wave_sum(sz);
    __shared__ float sred2[kThreads / 64];
    if ((t & 63) == 0) {
        sred[t >> 6] = se;
        sred2[t >> 6] = sz;
    }
    __syncthreads();
    float tse = 0.f, tsz = 0.f;
    for (int i = 0; i < kThreads / 64; ++i) {
        tse += sred[i];
        tsz += sred2[i];
    }
    const float lse = mx + logf(tse);
    const int64_t yb = labels[b];
    // MixUp / CutMix: the target is lam * onehot(own label) + (1 - lam) * onehot(label of sample b-1)
    const bool mixed = mix && mix[0] != 0.f;
    const float lam = mixed ? mix[1] : 1.f;
    const int64_t yp = mixed ? labels[(b + B - 1) % B] : yb;
    // a label outside [0, N) (torch raises there; a kernel cannot): no out-of-bounds read, and the loss becomes NaN
    // so that the bad batch is noticed instead of silently training on garbage
    const bool bad = yb < 0 || yb >= (int64_t)N || yp < 0 || yp >= (int64_t)N;
    if (t == 0) {
        const float zy = bad ? __builtin_nanf("") : lam * ldf(lp + yb) + (1.f - lam) * ldf(lp + yp);
        const float l = lse - (1.f - eps) * zy - (eps / (float)N) * tsz;
        atomicAdd(loss, l / (float)B);
    }
    if (dlogits) {
        T* dp = dlogits + (long)b * lddl;
        const float q0 = eps / (float)N;
        for (int i = t; i < N; i += kThreads) {
            const float pr = expf(ldf(lp + i) - lse);
            float q = q0;
            if ((int64_t)i == yb) q += (1.f - eps) * lam;
            if ((int64_t)i == yp) q += (1.f - eps) * (1.f - lam);
            dp[i] = from_float<T>((pr - q) * grad_scale);
        }
    }
}

// ---------------------------------------------------------------------------------
// SGD with momentum over a flat f32 buffer (+ optional low-precision mirror)
// ---------------------------------------------------------------------------------
template <typename MT>
__global__ void __launch_bounds__(kThreads)
sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, MT* __restrict__ mirror,
           long n, float lr, float mu, float wd, float gs, const float* __restrict__ lr_dev) {
    if (lr_dev) lr = lr_dev[0];
    const long stride = (long)gridDim.x * kThreads * 4;
    for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 pv = *(float4*)(p + i);
            const float4 gv = *(const float4*)(g + i);
            float4 mv = *(float4*)(m + i);
            float* pp = (float*)&pv;
            const float* gp = (const float*)&gv;
            float* mp = (float*)&mv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = fmaf(wd, pp[e], gp[e] * gs);
                mp[e] = fmaf(mu, mp[e], gg);
                pp[e] = fmaf(-lr, mp[e], pp[e]);
            }
            *(float4*)(p + i) = pv;
            *(float4*)(m + i) = mv;
            if (mirror) {
#pragma unroll
                for (int e = 0; e < 4; ++e) mirror[i + e] = from_float<MT>(pp[e]);
            }
        } else {
            for (long j = i; j < n; ++j) {
                const float gg = fmaf(wd, p[j], g[j] * gs);
                m[j] = fmaf(mu, m[j], gg);
                p[j] = fmaf(-lr, m[j], p[j]);
                if (mirror) mirror[j] = from_float<MT>(p[j]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// layout / precision plumbing
// ---------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void __launch_bounds__(kThreads)
copy2d_kernel(const S* __restrict__ src, long lds, D* __restrict__ dst, long ldd, long rows, int cols,
              int accumulate) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        float v = (float)src[r * lds + c];
        if (accumulate) v += (float)dst[r * ldd + c];
        dst[r * ldd + c] = from_float<D>(v);
    }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int C, long HW, int Cpad, long total) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const long b = i / HW, hw = i - b * HW;
        if constexpr (sizeof(T) == 2) {
            // the image batch of a step (3 -> 8 channels, 12.8 M pixels at batch 256): one 16-byte store per pixel
            // instead of eight 2-byte ones (175 -> 70 us)
            if (Cpad == 8) {
                unsigned short h[8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    h[c] = c < C ? __builtin_bit_cast(unsigned short, from_float<T>(x[(b * C + c) * HW + hw])) : (unsigned short)0;
                uint4 v;
                v.x = h[0] | ((unsigned)h[1] << 16), v.y = h[2] | ((unsigned)h[3] << 16);
                v.z = h[4] | ((unsigned)h[5] << 16), v.w = h[6] | ((unsigned)h[7] << 16);
                *(uint4*)(y + i * 8) = v;
                continue;
            }
        }
        for (int c = 0; c < Cpad; ++c) {
            const float v = c < C ? x[(b * C + c) * HW + hw] : 0.f;
            y[i * Cpad + c] = from_float<T>(v);
        }
    }
}

// the same with MixUp / CutMix applied on the way (mix: mode, lambda, x1, y1, x2, y2)
template <typename T>
__global__ void __launch_bounds__(kThreads)
mix_nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int W, long HW, int Cpad,
                        long total, const float* __restrict__ mix) {
    const int mode = (int)mix[0];
    const float lam = mix[1];
    const int x1 = (int)mix[2], y1 = (int)mix[3], x2 = (int)mix[4], y2 = (int)mix[5];
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const long b = i / HW, hw = i - b * HW;
        const long bp = (b + B - 1) % B;
        const int h = (int)(hw / W), w = (int)(hw - (long)h * W);
        const bool inbox = mode == 2 && h >= y1 && h < y2 && w >= x1 && w < x2;
        for (int c = 0; c < Cpad; ++c) {
            float v = 0.f;
            if (c < C) {
                const float own = x[(b * C + c) * HW + hw];
                if (mode == 1)
                    v = own * lam + x[(bp * C + c) * HW + hw] * (1.f - lam);  // extras.py:38-39
                else if (inbox)
                    v = x[(bp * C + c) * HW + hw];                            // extras.py:87
                else
                    v = own;
            }
            y[i * Cpad + c] = from_float<T>(v);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
nhwc_to_nchw_kernel(const T* __restrict__ y, int ldy, float* __restrict__ x, int C, long HW, long total) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const long b = i / HW, hw = i - b * HW;
        for (int c = 0; c < C; ++c) x[(b * C + c) * HW + hw] = (float)y[i * ldy + c];
    }
}

struct SelTable {
    int sel[VT_MAX_TAPS];
};

// out[c][i][n] = w[n][sel[i]][c]  (sel[i] < 0: zeros)
template <typename S, typename D>
__global__ void __launch_bounds__(kThreads)
pack_dgrad_kernel(const S* __restrict__ w, int ldw, D* __restrict__ out, SelTable st, int nsel, int Cout,
                  int Cin, long total) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int n = (int)(i % Cout);
        const long t2 = i / Cout;
        const int s = (int)(t2 % nsel);
        const int c = (int)(t2 / nsel);
        const int src = st.sel[s];
        out[i] = from_float<D>(src < 0 ? 0.f : (float)w[(long)n * ldw + (long)src * Cin + c]);
    }
}

// n filters in one launch: workgroup b belongs to the item whose [first, first + blocks) range holds it
struct PackBatchItem {
    const bf16_t* w;
    bf16_t* out;
    int ldw, nsel, Cout, Cin;
    unsigned first, blocks;
    signed char sel[VT_MAX_TAPS];
};
struct PackBatch {
    int n;
    PackBatchItem it[VT_PACK_BATCH];
};
// a workgroup moves one 64 (out channels) x 64 (in channels) tile of one packed tap through LDS: 16-byte loads along c,
// 16-byte stores along n (the one-element-per-thread kernel above reads w with a stride of ldw elements between lanes).
// Cin, Cout and ldw are multiples of 8 and both bases 16-byte aligned (checked by the entry point).
__global__ void __launch_bounds__(kThreads) pack_dgrad_batch_kernel(const PackBatch b) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];  // [c][n]
    int k = 0;
    while (k + 1 < b.n && blockIdx.x >= b.it[k + 1].first) ++k;  // (<= 40 items, kernel arguments: scalar loads)
    const PackBatchItem& it = b.it[k];
    const int lb = (int)(blockIdx.x - it.first);
    const int tiles_n = (it.Cout + 63) / 64;
    const int s = lb % it.nsel, rest = lb / it.nsel;
    const int n0 = (rest % tiles_n) * 64, c0 = (rest / tiles_n) * 64;
    const int src = it.sel[s];
    const unsigned short* w = (const unsigned short*)it.w;
    unsigned short* out = (unsigned short*)it.out;
#pragma unroll
    for (int e = threadIdx.x; e < 512; e += kThreads) {
        const int r = e >> 3, j = e & 7;  // out channel n0 + r, in channels c0 + 8j ..
        uint4 v = make_uint4(0, 0, 0, 0);
        if (src >= 0 && n0 + r < it.Cout && c0 + 8 * j < it.Cin)
            v = *(const uint4*)(w + ((long)(n0 + r) * it.ldw + (long)src * it.Cin + c0 + 8 * j));
        const unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tile[8 * j + 2 * q][r] = (unsigned short)(d[q] & 0xffffu);
            tile[8 * j + 2 * q + 1][r] = (unsigned short)(d[q] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = threadIdx.x; e < 512; e += kThreads) {
        const int c = e >> 3, j = e & 7;  // in channel c0 + c, out channels n0 + 8j ..
        if (c0 + c < it.Cin && n0 + 8 * j < it.Cout)
            *(uint4*)(out + (((long)(c0 + c) * it.nsel + s) * it.Cout + n0 + 8 * j)) = *(const uint4*)&tile[c][8 * j];
    }
}

// Walking order of the three BatchNorm passes over their rows (bit 0: bn_act_apply, 1: bn_bwd_reduce, 2: bn_bwd_apply;
// set = last row block first).  A pass that follows a kernel which just streamed the same tensor front to back finds
// that tensor's TAIL in the memory-side cache (256 MB), not its head: walking backwards would turn those into hits.
// Measured on the full step, every combination: 23.14 .. 23.25 ms against 23.23 .. 23.25 for the default (0) -- no
// effect beyond noise; the knob stays for experiments.
inline int vt_bn_order() {
    const int v = (0);
    return v;
}

__global__ void __launch_bounds__(kThreads)
fixed_to_f32_kernel(const long long* __restrict__ q, float* __restrict__ dst, long n, int accumulate) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const double v = (double)q[2 * i] * 4096.0 + (double)q[2 * i + 1] * (1.0 / 8589934592.0);
        dst[i] = accumulate ? dst[i] + (float)v : (float)v;
    }
}

inline unsigned flat_blocks(long total, int per_thread = 1) {
    long b = (total + (long)kThreads * per_thread - 1) / ((long)kThreads * per_thread);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

#define VT_DISPATCH_T(dtype, NAME, ...)                               \
    do {                                                              \
        if ((dtype) == VT_BF16) {                                     \
            typedef bf16_t T;                                         \
            __VA_ARGS__;                                              \
        } else if ((dtype) == VT_F32) {                               \
            typedef float T;                                          \
            __VA_ARGS__;                                              \
        } else {                                                      \
            vt_set_error("%s: unsupported dtype %d", NAME, (dtype)); \
            return VT_ERR_UNSUPPORTED;                                \
        }                                                             \
    } while (0)

inline int check_mat(const char* name, const void* p, int ld, int C, int dtype) {
    const int epc = vt_epc(dtype);
    if (!p || !vt_aligned16(p)) {
        vt_set_error("%s: null or misaligned pointer", name);
        return VT_ERR_INVALID;
    }
    if (C <= 0 || C % epc || ld % epc || ld < C) {
        vt_set_error("%s: C=%d ld=%d must be positive multiples of %d with ld>=C", name, C, ld, epc);
        return VT_ERR_UNSUPPORTED;
    }
    return VT_OK;
}
#define VT_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != VT_OK) return rc__; \
    } while (0)

}  // namespace

extern "C" {

int vt_bn_finalize(const float* stats, int32_t C, double count, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var,
                   int64_t* num_batches_tracked, float* scale, float* shift, float* mean, float* invstd,
                   void* stream) {
    VT_REQUIRE(stats && scale && shift && mean && invstd && C > 0 && count > 0, VT_ERR_INVALID,
               "vt_bn_finalize: bad argument");
    VT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), VT_ERR_INVALID,
               "vt_bn_finalize: running_mean/var must both be given or both NULL");
    const double unbias = count > 1.0 ? count / (count - 1.0) : 1.0;
    VtFinFwd f{stats, gamma, beta, running_mean, running_var, num_batches_tracked, scale, shift, mean, invstd,
               1.0 / count, unbias, eps, momentum, C};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, f);
    VT_CHECK_LAUNCH("vt_bn_finalize");
    return VT_OK;
}

int vt_stat_fold(float* stats, int32_t C, void* stream) {
    VT_REQUIRE(stats && C > 0, VT_ERR_INVALID, "vt_stat_fold: bad argument");
    const int n = 4 * C;
    hipLaunchKernelGGL(stat_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (long long*)stats, n);
    VT_CHECK_LAUNCH("vt_stat_fold");
    return VT_OK;
}

int vt_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int32_t C, float* scale, float* shift,
                      float* mean, float* invstd, void* stream) {
    VT_REQUIRE(running_mean && running_var && scale && shift && C > 0, VT_ERR_INVALID,
               "vt_bn_eval_coeffs: bad argument");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma,
                       beta, running_mean, running_var, eps, C, scale, shift, mean, invstd);
    VT_CHECK_LAUNCH("vt_bn_eval_coeffs");
    return VT_OK;
}

int vt_bn_eval_coeffs_batch(const vt_bn_eval_item* items, int32_t n, void* stream) {
    VT_REQUIRE(items && n >= 1, VT_ERR_INVALID, "vt_bn_eval_coeffs_batch: bad argument");
    for (int base = 0; base < n; base += VT_PACK_BATCH) {
        BnEvalBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n - base < VT_PACK_BATCH ? n - base : VT_PACK_BATCH;
        int cmax = 0;
        for (int k = 0; k < b.n; ++k) {
            const vt_bn_eval_item& s = items[base + k];
            VT_REQUIRE(s.running_mean && s.running_var && s.scale && s.shift && s.C > 0, VT_ERR_INVALID,
                       "vt_bn_eval_coeffs_batch: bad item %d", base + k);
            b.it[k] = s;
            cmax = s.C > cmax ? s.C : cmax;
        }
        const int bpi = (cmax + 255) / 256;
        hipLaunchKernelGGL(bn_eval_coeffs_batch_kernel, dim3((unsigned)(b.n * bpi)), dim3(256), 0, (hipStream_t)stream, b, bpi);
        VT_CHECK_LAUNCH("vt_bn_eval_coeffs_batch");
    }
    return VT_OK;
}

int vt_bn_act_apply(const void* z, int32_t ldz, const float* scale, const float* shift, const void* residual,
                    int32_t ldr, void* y, int32_t ldy, int64_t M, int32_t C, int32_t relu, int32_t dtype,
                    void* stream) {
    VT_REQUIRE(M > 0, VT_ERR_INVALID, "vt_bn_act_apply: M=%ld", (long)M);
    VT_TRY(check_mat("vt_bn_act_apply(z)", z, ldz, C, dtype));
    VT_TRY(check_mat("vt_bn_act_apply(y)", y, ldy, C, dtype));
    if (residual) VT_TRY(check_mat("vt_bn_act_apply(residual)", residual, ldr, C, dtype));
    RowMap rm = RowMap::make(C, vt_epc(dtype), M);
    rm.rev = (vt_bn_order() >> 0) & 1;
    VT_REQUIRE(relu >= 0 && relu <= 4, VT_ERR_INVALID, "vt_bn_act_apply: activation code %d", relu);
    if (relu >= 2) {  // LeakyReLU(0.2) / SiLU / GELU: the generic instantiation (off the Darknet / VoVNet path)
        if (residual) {
            VT_DISPATCH_T(dtype, "vt_bn_act_apply",
                          hipLaunchKernelGGL((bn_act_apply_kernel<T, true, true>), dim3(rm.blocks(M)), dim3(kThreads), 0,
                                             (hipStream_t)stream, (const T*)z, ldz, scale, shift, (const T*)residual, ldr,
                                             (T*)y, ldy, (long)M, rm, relu));
        } else {
            VT_DISPATCH_T(dtype, "vt_bn_act_apply",
                          hipLaunchKernelGGL((bn_act_apply_kernel<T, false, true>), dim3(rm.blocks(M)), dim3(kThreads), 0,
                                             (hipStream_t)stream, (const T*)z, ldz, scale, shift, (const T*)residual, ldr,
                                             (T*)y, ldy, (long)M, rm, relu));
        }
        VT_CHECK_LAUNCH("vt_bn_act_apply");
        return VT_OK;
    }
    if (residual) {
        VT_DISPATCH_T(dtype, "vt_bn_act_apply",
                      VT_LAUNCH_STOP((bn_act_apply_kernel<T, true>), dim3(rm.blocks(M)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)z, ldz, scale, shift, (const T*)residual,
                                     ldr, (T*)y, ldy, (long)M, rm, relu));
    } else {
        VT_DISPATCH_T(dtype, "vt_bn_act_apply",
                      VT_LAUNCH_STOP((bn_act_apply_kernel<T, false>), dim3(rm.blocks(M)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)z, ldz, scale, shift, (const T*)residual,
                                     ldr, (T*)y, ldy, (long)M, rm, relu));
    }
    VT_CHECK_LAUNCH("vt_bn_act_apply");
    return VT_OK;
}

// vt_bn_finalize + vt_bn_act_apply in one launch (see bn_fin_apply_kernel); activation codes >= 2 and VT_BN_FIN_APPLY=0:
// the two launches.  `ready`: 4 zeroed bytes.
int vt_bn_finalize_apply(const float* stats, int32_t C, double count, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale,
                         float* shift, float* mean, float* invstd, const void* z, int32_t ldz,
                         const void* residual, int32_t ldr, void* y, int32_t ldy, int64_t M, int32_t relu, int32_t dtype,
                         void* stream) {
    VT_REQUIRE(stats && scale && shift && mean && invstd && C > 0 && count > 0 && M > 0, VT_ERR_INVALID,
               "vt_bn_finalize_apply: bad argument");
    VT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), VT_ERR_INVALID,
               "vt_bn_finalize_apply: running_mean/var must both be given or both NULL");
    VT_REQUIRE(relu >= 0 && relu <= 4, VT_ERR_INVALID, "vt_bn_finalize_apply: activation code %d", relu);
    const int Cg = fin_group(C, vt_epc(dtype));
    if (relu >= 2 || !Cg || !VT_KNOB("VT_BN_FIN_APPLY", 1)) {
        const int rc = vt_bn_finalize(stats, C, count, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                                      scale, shift, mean, invstd, stream);
        if (rc != VT_OK) return rc;
        return vt_bn_act_apply(z, ldz, scale, shift, residual, ldr, y, ldy, M, C, relu, dtype, stream);
    }
    VT_TRY(check_mat("vt_bn_finalize_apply(z)", z, ldz, C, dtype));
    VT_TRY(check_mat("vt_bn_finalize_apply(y)", y, ldy, C, dtype));
    if (residual) VT_TRY(check_mat("vt_bn_finalize_apply(residual)", residual, ldr, C, dtype));
    const int groups = C / Cg, wgs = VT_KNOB("VT_BN_FIN_APPLY_WGS", 1536);
    RowMap rm = RowMap::make(Cg, vt_epc(dtype), M, wgs / groups > 0 ? wgs / groups : 1);
    rm.rev = (vt_bn_order() >> 0) & 1;
    const VtFinFwd f{stats, gamma, beta, running_mean, running_var, num_batches_tracked, scale, shift, mean, invstd,
                     1.0 / count, count > 1.0 ? count / (count - 1.0) : 1.0, eps, momentum, C};
    const dim3 grid(rm.blocks(M), groups);
    if (residual) {
        VT_DISPATCH_T(dtype, "vt_bn_finalize_apply",
                      VT_LAUNCH_STOP((bn_fin_apply_kernel<T, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, f,
                                     (const T*)z, ldz, (const T*)residual, ldr, (T*)y, ldy, (long)M, rm, relu, Cg));
    } else {
        VT_DISPATCH_T(dtype, "vt_bn_finalize_apply",
                      VT_LAUNCH_STOP((bn_fin_apply_kernel<T, false>), grid, dim3(kThreads), 0, (hipStream_t)stream, f,
                                     (const T*)z, ldz, (const T*)residual, ldr, (T*)y, ldy, (long)M, rm, relu, Cg));
    }
    VT_CHECK_LAUNCH("vt_bn_finalize_apply");
    return VT_OK;
}

// vt_bn_bwd_finalize + vt_bn_act_bwd_apply in one launch (see bn_bwd_fin_apply_kernel)
int vt_bn_bwd_finalize_apply(const float* sums, int32_t C, double count, double pscale, const float* scale,
                             const float* shift, const float* mean, const float* invstd, int32_t train, float* dgamma,
                             float* dbeta, float* coef, const void* dy, int32_t lddy, const void* z, int32_t ldz,
                             void* dz, int32_t lddz, int64_t M, int32_t relu, int32_t dtype, void* stream) {
    VT_REQUIRE(sums && scale && shift && mean && invstd && coef && C > 0 && count > 0 && M > 0, VT_ERR_INVALID,
               "vt_bn_bwd_finalize_apply: bad argument");
    VT_REQUIRE(relu >= 0 && relu <= 4, VT_ERR_INVALID, "vt_bn_bwd_finalize_apply: activation code %d", relu);
    const int Cg = fin_group(C, vt_epc(dtype));
    if (relu >= 2 || !Cg || !VT_KNOB("VT_BN_FIN_APPLY", 1)) {
        const int rc = vt_bn_bwd_finalize(sums, C, count, pscale, scale, mean, invstd, train, dgamma, dbeta, coef, stream);
        if (rc != VT_OK) return rc;
        return vt_bn_act_bwd_apply(dy, lddy, z, ldz, scale, shift, coef, dz, lddz, M, C, relu, dtype, stream);
    }
    VT_TRY(check_mat("vt_bn_bwd_finalize_apply(dy)", dy, lddy, C, dtype));
    VT_TRY(check_mat("vt_bn_bwd_finalize_apply(z)", z, ldz, C, dtype));
    VT_TRY(check_mat("vt_bn_bwd_finalize_apply(dz)", dz, lddz, C, dtype));
    const int groups = C / Cg, wgs = VT_KNOB("VT_BN_FIN_APPLY_WGS", 1536);
    RowMap rm = RowMap::make(Cg, vt_epc(dtype), M, wgs / groups > 0 ? wgs / groups : 1);
    rm.rev = (vt_bn_order() >> 2) & 1;
    const VtFinBwd f{sums, scale, mean, invstd, dgamma, dbeta, coef, 1.0 / count, pscale, C, train};
    VT_DISPATCH_T(dtype, "vt_bn_bwd_finalize_apply",
                  VT_LAUNCH_STOP(bn_bwd_fin_apply_kernel<T>, dim3(rm.blocks(M), groups), dim3(kThreads), 0, (hipStream_t)stream, f,
                                 (const T*)dy, lddy, (const T*)z, ldz, scale, shift, (T*)dz, lddz, (long)M, rm, relu, Cg));
    VT_CHECK_LAUNCH("vt_bn_bwd_finalize_apply");
    return VT_OK;
}

// (the two launches above wait for nothing since their second form: always 0; kept for vt_bn_bwd_fused_timeouts' sum)
unsigned vt_fin_timeouts_host() { return 0; }

int vt_bn_act_bwd_reduce(const void* dy, int32_t lddy, const void* z, int32_t ldz, const float* scale,
                         const float* shift, const float* mean, const float* invstd, int64_t M, int32_t C,
                         int32_t relu, int32_t dtype, float* sums, void* stream) {
    VT_REQUIRE(M > 0 && scale && shift && mean && invstd && sums, VT_ERR_INVALID,
               "vt_bn_act_bwd_reduce: bad argument");
    VT_TRY(check_mat("vt_bn_act_bwd_reduce(dy)", dy, lddy, C, dtype));
    VT_TRY(check_mat("vt_bn_act_bwd_reduce(z)", z, ldz, C, dtype));
    const int epc = vt_epc(dtype);
    // every block ends with 2*C 64-bit atomics into one of the statistics replicas: ~100 ns each when they queue on the same address,
    // so fewer, longer blocks win over grid-filling ones (measured per step: 1024 -> 23.53, 512 -> 23.36, 256 -> 23.27 ms)
    // channel groups of 64 on the small maps with many channels (see the kernel): the row splits shrink by the same factor
    const int cgroups = (M <= 65536 && C >= 256 && C % 64 == 0) ? C / 64 : 1;
    const int Cg = C / cgroups;
    // (round 6, beside the CU-owning filter-gradient kernels of the side stream: 128 -> 19.74, 256 -> 19.49, 512 / 1024 -> 19.45,
    //  2048 -> 19.55 ms per step)
    const int blocks_knob = VT_KNOB("VT_BN_RED_BLOCKS", 1024);
    const int target = blocks_knob / cgroups > 0 ? blocks_knob / cgroups : 1;
    RowMap rm = RowMap::make(Cg, epc, M, target);
    rm.rev = (vt_bn_order() >> 1) & 1;
    const int inwave_env = (1);
    const bool inwave = inwave_env && rm.CT < 64 && (rm.CT & (rm.CT - 1)) == 0;  // (as in the kernel)
    const int smem = (inwave ? kThreads / 64 : rm.RT) * 2 * rm.CT * epc * (int)sizeof(float);
    VT_REQUIRE(relu >= 0 && relu <= 4, VT_ERR_INVALID, "vt_bn_act_bwd_reduce: activation code %d", relu);
    if (relu >= 2) {
        VT_DISPATCH_T(dtype, "vt_bn_act_bwd_reduce",
                      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true>), dim3(rm.blocks(M), cgroups), dim3(kThreads), smem,
                                         (hipStream_t)stream, (const T*)dy, lddy, (const T*)z, ldz, scale, shift, mean,
                                         invstd, (long)M, Cg, rm, (relu << 4) | (inwave_env ? 0 : 2), sums, C));
        VT_CHECK_LAUNCH("vt_bn_act_bwd_reduce");
        return VT_OK;
    }
    VT_DISPATCH_T(dtype, "vt_bn_act_bwd_reduce",
                  hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(rm.blocks(M), cgroups), dim3(kThreads), smem,
                                     (hipStream_t)stream, (const T*)dy, lddy, (const T*)z, ldz, scale, shift,
                                     mean, invstd, (long)M, Cg, rm, (relu ? 1 : 0) | (inwave_env ? 0 : 2), sums, C));
    VT_CHECK_LAUNCH("vt_bn_act_bwd_reduce");
    return VT_OK;
}

int vt_bn_bwd_finalize(const float* sums, int32_t C, double count, double pscale, const float* scale, const float* mean,
                       const float* invstd, int32_t train, float* dgamma, float* dbeta, float* coef,
                       void* stream) {
    VT_REQUIRE(sums && scale && mean && invstd && coef && C > 0 && count > 0, VT_ERR_INVALID,
               "vt_bn_bwd_finalize: bad argument");
    VtFinBwd f{sums, scale, mean, invstd, dgamma, dbeta, coef, 1.0 / count, pscale, C, train};
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, f);
    VT_CHECK_LAUNCH("vt_bn_bwd_finalize");
    return VT_OK;
}

int vt_bn_act_bwd_apply(const void* dy, int32_t lddy, const void* z, int32_t ldz, const float* scale,
                        const float* shift, const float* coef, void* dz, int32_t lddz, int64_t M, int32_t C,
                        int32_t relu, int32_t dtype, void* stream) {
    // (scale, shift, coef all NULL: dz = dy * act'(z) -- a ConvNormAct with norm="none", components.py:36)
    const bool act_only = !scale && !shift && !coef;
    VT_REQUIRE(M > 0 && (act_only || (scale && shift && coef)), VT_ERR_INVALID, "vt_bn_act_bwd_apply: bad argument");
    VT_TRY(check_mat("vt_bn_act_bwd_apply(dy)", dy, lddy, C, dtype));
    VT_TRY(check_mat("vt_bn_act_bwd_apply(z)", z, ldz, C, dtype));
    VT_TRY(check_mat("vt_bn_act_bwd_apply(dz)", dz, lddz, C, dtype));
    RowMap rm = RowMap::make(C, vt_epc(dtype), M);
    rm.rev = (vt_bn_order() >> 2) & 1;
    VT_REQUIRE(relu >= 0 && relu <= 4, VT_ERR_INVALID, "vt_bn_act_bwd_apply: activation code %d", relu);
    if (relu >= 2 || act_only) {
        VT_DISPATCH_T(dtype, "vt_bn_act_bwd_apply",
                      hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(rm.blocks(M)), dim3(kThreads), 0,
                                         (hipStream_t)stream, (const T*)dy, lddy, (const T*)z, ldz, scale, shift, coef,
                                         (T*)dz, lddz, (long)M, C, rm, relu));
        VT_CHECK_LAUNCH("vt_bn_act_bwd_apply");
        return VT_OK;
    }
    VT_DISPATCH_T(dtype, "vt_bn_act_bwd_apply",
                  VT_LAUNCH_STOP(bn_bwd_apply_kernel<T>, dim3(rm.blocks(M)), dim3(kThreads), 0,
                                 (hipStream_t)stream, (const T*)dy, lddy, (const T*)z, ldz, scale, shift,
                                 coef, (T*)dz, lddz, (long)M, C, rm, relu));
    VT_CHECK_LAUNCH("vt_bn_act_bwd_apply");
    return VT_OK;
}

static inline int pool_out(int n) { return (n + 2 - 3) / 2 + 1; }

int vt_bn_act_apply_pool(const void* z, int32_t ldz, const float* scale, const float* shift, const void* residual,
                         int32_t ldr, void* y, int32_t ldy, void* pooled, int32_t ldp, uint8_t* argmax, int32_t B,
                         int32_t H, int32_t W, int32_t C, int32_t relu, int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && H > 0 && W > 0 && argmax && scale && shift, VT_ERR_INVALID, "vt_bn_act_apply_pool: bad argument");
    VT_TRY(check_mat("vt_bn_act_apply_pool(z)", z, ldz, C, dtype));
    VT_TRY(check_mat("vt_bn_act_apply_pool(y)", y, ldy, C, dtype));
    VT_TRY(check_mat("vt_bn_act_apply_pool(pooled)", pooled, ldp, C, dtype));
    if (residual) VT_TRY(check_mat("vt_bn_act_apply_pool(residual)", residual, ldr, C, dtype));
    const int Ho = pool_out(H), Wo = pool_out(W);
    const long Mo = (long)B * Ho * Wo;
    RowMap rm = RowMap::make(C, vt_epc(dtype), Mo);
    if (residual) {
        VT_DISPATCH_T(dtype, "vt_bn_act_apply_pool",
                      hipLaunchKernelGGL((bn_act_apply_pool_kernel<T, true>), dim3(rm.blocks(Mo)), dim3(kThreads), 0,
                                         (hipStream_t)stream, (const T*)z, ldz, scale, shift, (const T*)residual, ldr, (T*)y,
                                         ldy, (T*)pooled, ldp, argmax, H, W, Ho, Wo, C, Mo, rm, relu));
    } else {
        VT_DISPATCH_T(dtype, "vt_bn_act_apply_pool",
                      hipLaunchKernelGGL((bn_act_apply_pool_kernel<T, false>), dim3(rm.blocks(Mo)), dim3(kThreads), 0,
                                         (hipStream_t)stream, (const T*)z, ldz, scale, shift, (const T*)residual, ldr, (T*)y,
                                         ldy, (T*)pooled, ldp, argmax, H, W, Ho, Wo, C, Mo, rm, relu));
    }
    VT_CHECK_LAUNCH("vt_bn_act_apply_pool");
    return VT_OK;
}

int vt_bn_act_bwd_reduce_pool(const void* dp, int32_t lddp, const uint8_t* argmax, const void* z, int32_t ldz,
                              const float* scale, const float* shift, const float* mean, const float* invstd, int32_t B,
                              int32_t H, int32_t W, int32_t C, int32_t relu, int32_t dtype, float* sums, void* stream) {
    VT_REQUIRE(B > 0 && H > 0 && W > 0 && argmax && scale && shift && mean && invstd && sums, VT_ERR_INVALID,
               "vt_bn_act_bwd_reduce_pool: bad argument");
    VT_TRY(check_mat("vt_bn_act_bwd_reduce_pool(dp)", dp, lddp, C, dtype));
    VT_TRY(check_mat("vt_bn_act_bwd_reduce_pool(z)", z, ldz, C, dtype));
    const int Ho = pool_out(H), Wo = pool_out(W);
    const long Mo = (long)B * Ho * Wo;
    const int epc = vt_epc(dtype);
    RowMap rm = RowMap::make(C, epc, Mo, 256);
    const bool inwave = rm.CT < 64 && (rm.CT & (rm.CT - 1)) == 0;  // (as in the kernel)
    const int smem = (inwave ? kThreads / 64 : rm.RT) * 2 * rm.CT * epc * (int)sizeof(float);
    VT_DISPATCH_T(dtype, "vt_bn_act_bwd_reduce_pool",
                  hipLaunchKernelGGL((bn_bwd_pool_kernel<T, false>), dim3(rm.blocks(Mo)), dim3(kThreads), smem,
                                     (hipStream_t)stream, (const T*)dp, lddp, argmax, (const T*)z, ldz, scale, shift, mean, invstd,
                                     (T*)nullptr, 0, H, W, Ho, Wo, C, Mo, rm, relu, sums));
    VT_CHECK_LAUNCH("vt_bn_act_bwd_reduce_pool");
    return VT_OK;
}

int vt_bn_act_bwd_apply_pool(const void* dp, int32_t lddp, const uint8_t* argmax, const void* z, int32_t ldz,
                             const float* scale, const float* shift, const float* coef, void* dz, int32_t lddz, int32_t B,
                             int32_t H, int32_t W, int32_t C, int32_t relu, int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && H > 0 && W > 0 && argmax && scale && shift && coef, VT_ERR_INVALID,
               "vt_bn_act_bwd_apply_pool: bad argument");
    VT_TRY(check_mat("vt_bn_act_bwd_apply_pool(dp)", dp, lddp, C, dtype));
    VT_TRY(check_mat("vt_bn_act_bwd_apply_pool(z)", z, ldz, C, dtype));
    VT_TRY(check_mat("vt_bn_act_bwd_apply_pool(dz)", dz, lddz, C, dtype));
    const int Ho = pool_out(H), Wo = pool_out(W);
    const long Mo = (long)B * Ho * Wo;
    RowMap rm = RowMap::make(C, vt_epc(dtype), Mo);
    VT_DISPATCH_T(dtype, "vt_bn_act_bwd_apply_pool",
                  VT_LAUNCH_STOP((bn_bwd_pool_kernel<T, true>), dim3(rm.blocks(Mo)), dim3(kThreads), 0, (hipStream_t)stream,
                                 (const T*)dp, lddp, argmax, (const T*)z, ldz, scale, shift, coef, (const float*)nullptr, (T*)dz,
                                 lddz, H, W, Ho, Wo, C, Mo, rm, relu, (float*)nullptr));
    VT_CHECK_LAUNCH("vt_bn_act_bwd_apply_pool");
    return VT_OK;
}

int vt_maxpool3x3s2_fwd(const void* x, int32_t ldx, void* y, int32_t ldy, uint8_t* argmax, int32_t B,
                        int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && H > 0 && W > 0 && argmax, VT_ERR_INVALID, "vt_maxpool3x3s2_fwd: bad argument");
    VT_TRY(check_mat("vt_maxpool3x3s2_fwd(x)", x, ldx, C, dtype));
    VT_TRY(check_mat("vt_maxpool3x3s2_fwd(y)", y, ldy, C, dtype));
    const int Ho = pool_out(H), Wo = pool_out(W);
    const long Mo = (long)B * Ho * Wo;
    const RowMap rm = RowMap::make(C, vt_epc(dtype), Mo);
    VT_DISPATCH_T(dtype, "vt_maxpool3x3s2_fwd",
                  hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(rm.blocks(Mo)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, argmax, H, W, Ho, Wo,
                                     C, Mo, rm));
    VT_CHECK_LAUNCH("vt_maxpool3x3s2_fwd");
    return VT_OK;
}

int vt_maxpool3x3s2_bwd(const void* dy, int32_t lddy, const uint8_t* argmax, void* dx, int32_t lddx,
                        int32_t B, int32_t H, int32_t W, int32_t C, int32_t accumulate, int32_t dtype,
                        void* stream) {
    VT_REQUIRE(B > 0 && H > 0 && W > 0 && argmax, VT_ERR_INVALID, "vt_maxpool3x3s2_bwd: bad argument");
    VT_TRY(check_mat("vt_maxpool3x3s2_bwd(dy)", dy, lddy, C, dtype));
    VT_TRY(check_mat("vt_maxpool3x3s2_bwd(dx)", dx, lddx, C, dtype));
    const int Ho = pool_out(H), Wo = pool_out(W);
    const long Mi = (long)B * H * W;
    const RowMap rm = RowMap::make(C, vt_epc(dtype), Mi);
    VT_DISPATCH_T(dtype, "vt_maxpool3x3s2_bwd",
                  hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(rm.blocks(Mi)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)dy, lddy, argmax, (T*)dx, lddx, H, W, Ho,
                                     Wo, C, Mi, rm, accumulate));
    VT_CHECK_LAUNCH("vt_maxpool3x3s2_bwd");
    return VT_OK;
}

static inline int pool_ct(int C, int epc) {
    int cpr = C / epc;
    int ct = 1;
    while (ct < cpr && ct < 64) ct <<= 1;
    return ct;
}

int vt_global_avgpool_fwd(const void* x, int32_t ldx, void* y, int32_t ldy, int32_t B, int32_t HW, int32_t C,
                          int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && HW > 0, VT_ERR_INVALID, "vt_global_avgpool_fwd: bad argument");
    VT_TRY(check_mat("vt_global_avgpool_fwd(x)", x, ldx, C, dtype));
    VT_TRY(check_mat("vt_global_avgpool_fwd(y)", y, ldy, C, dtype));
    const int epc = vt_epc(dtype);
    const int CT = pool_ct(C, epc);
    const dim3 grid(B, (C / epc + CT - 1) / CT);
    VT_DISPATCH_T(dtype, "vt_global_avgpool_fwd",
                  hipLaunchKernelGGL(avgpool_fwd_kernel<T>, grid, dim3(kThreads), CT * epc * sizeof(unsigned long long),
                                     (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, HW, C, CT));
    VT_CHECK_LAUNCH("vt_global_avgpool_fwd");
    return VT_OK;
}

int vt_global_avgpool_bwd(const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t B, int32_t HW,
                          int32_t C, int32_t accumulate, int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && HW > 0, VT_ERR_INVALID, "vt_global_avgpool_bwd: bad argument");
    VT_TRY(check_mat("vt_global_avgpool_bwd(dy)", dy, lddy, C, dtype));
    VT_TRY(check_mat("vt_global_avgpool_bwd(dx)", dx, lddx, C, dtype));
    const long M = (long)B * HW;
    const RowMap rm = RowMap::make(C, vt_epc(dtype), M);
    VT_DISPATCH_T(dtype, "vt_global_avgpool_bwd",
                  hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(rm.blocks(M)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)dy, lddy, (T*)dx, lddx, HW, M, rm,
                                     accumulate));
    VT_CHECK_LAUNCH("vt_global_avgpool_bwd");
    return VT_OK;
}

int vt_resample2x_add_fwd(const void* src, int32_t lds, const void* other, int32_t ldo, void* dst, int32_t ldd,
                          int32_t B, int32_t Hd, int32_t Wd, int32_t C, int32_t mode, int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && Hd > 0 && Wd > 0 && mode >= 0 && mode <= 3, VT_ERR_INVALID,
               "vt_resample2x_add_fwd: bad argument");
    VT_REQUIRE((mode & 1) || (Hd % 2 == 0 && Wd % 2 == 0), VT_ERR_UNSUPPORTED,
               "vt_resample2x_add_fwd: x2 upsampling needs an even destination (%dx%d)", Hd, Wd);
    VT_TRY(check_mat("vt_resample2x_add_fwd(src)", src, lds, C, dtype));
    VT_TRY(check_mat("vt_resample2x_add_fwd(dst)", dst, ldd, C, dtype));
    if (other) VT_TRY(check_mat("vt_resample2x_add_fwd(other)", other, ldo, C, dtype));
    const long M = (long)B * Hd * Wd;
    const RowMap rm = RowMap::make(C, vt_epc(dtype), M);
    VT_DISPATCH_T(dtype, "vt_resample2x_add_fwd",
                  hipLaunchKernelGGL(resample_fwd_kernel<T>, dim3(rm.blocks(M)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)src, lds, (const T*)other, ldo, (T*)dst, ldd, Hd,
                                     Wd, M, rm, mode));
    VT_CHECK_LAUNCH("vt_resample2x_add_fwd");
    return VT_OK;
}

int vt_resample2x_bwd(const void* dy, int32_t lddy, void* dsrc, int32_t lds, int32_t B, int32_t Hd, int32_t Wd,
                      int32_t C, int32_t mode, int32_t accumulate, int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && Hd > 0 && Wd > 0 && mode >= 0 && mode <= 3, VT_ERR_INVALID,
               "vt_resample2x_bwd: bad argument");
    VT_REQUIRE((mode & 1) || (Hd % 2 == 0 && Wd % 2 == 0), VT_ERR_UNSUPPORTED,
               "vt_resample2x_bwd: x2 upsampling needs an even destination (%dx%d)", Hd, Wd);
    VT_TRY(check_mat("vt_resample2x_bwd(dy)", dy, lddy, C, dtype));
    VT_TRY(check_mat("vt_resample2x_bwd(dsrc)", dsrc, lds, C, dtype));
    const int Hs = (mode & 1) ? 2 * Hd : Hd / 2, Ws = (mode & 1) ? 2 * Wd : Wd / 2;
    const long Ms = (long)B * Hs * Ws;
    const RowMap rm = RowMap::make(C, vt_epc(dtype), Ms);
    VT_DISPATCH_T(dtype, "vt_resample2x_bwd",
                  hipLaunchKernelGGL(resample_bwd_kernel<T>, dim3(rm.blocks(Ms)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)dy, lddy, (T*)dsrc, lds, Hd, Wd, Ms, rm, mode,
                                     accumulate));
    VT_CHECK_LAUNCH("vt_resample2x_bwd");
    return VT_OK;
}

int vt_ese_gate_fwd(const void* x, int32_t ldx, const void* s, int32_t lds, const void* residual, int32_t ldr,
                    void* y, int32_t ldy, int32_t B, int32_t HW, int32_t C, int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && HW > 0, VT_ERR_INVALID, "vt_ese_gate_fwd: bad argument");
    VT_TRY(check_mat("vt_ese_gate_fwd(x)", x, ldx, C, dtype));
    VT_TRY(check_mat("vt_ese_gate_fwd(s)", s, lds, C, dtype));
    VT_TRY(check_mat("vt_ese_gate_fwd(y)", y, ldy, C, dtype));
    if (residual) VT_TRY(check_mat("vt_ese_gate_fwd(residual)", residual, ldr, C, dtype));
    const long M = (long)B * HW;
    const RowMap rm = RowMap::make(C, vt_epc(dtype), M);
    VT_DISPATCH_T(dtype, "vt_ese_gate_fwd",
                  hipLaunchKernelGGL(ese_fwd_kernel<T>, dim3(rm.blocks(M)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)x, ldx, (const T*)s, lds,
                                     (const T*)residual, ldr, (T*)y, ldy, HW, M, rm));
    VT_CHECK_LAUNCH("vt_ese_gate_fwd");
    return VT_OK;
}

int vt_ese_gate_bwd(const void* dy, int32_t lddy, const void* x, int32_t ldx, const void* s, int32_t lds,
                    void* dx, int32_t lddx, float* ds, int32_t B, int32_t HW, int32_t C, int32_t accumulate,
                    int32_t dtype, void* stream) {
    VT_REQUIRE(B > 0 && HW > 0 && ds, VT_ERR_INVALID, "vt_ese_gate_bwd: bad argument");
    VT_TRY(check_mat("vt_ese_gate_bwd(dy)", dy, lddy, C, dtype));
    VT_TRY(check_mat("vt_ese_gate_bwd(x)", x, ldx, C, dtype));
    VT_TRY(check_mat("vt_ese_gate_bwd(s)", s, lds, C, dtype));
    VT_TRY(check_mat("vt_ese_gate_bwd(dx)", dx, lddx, C, dtype));
    const int epc = vt_epc(dtype);
    const int CT = pool_ct(C, epc);
    const dim3 grid(B, (C / epc + CT - 1) / CT);
    VT_DISPATCH_T(dtype, "vt_ese_gate_bwd",
                  hipLaunchKernelGGL(ese_bwd_kernel<T>, grid, dim3(kThreads), CT * epc * sizeof(unsigned long long),
                                     (hipStream_t)stream, (const T*)dy, lddy, (const T*)x, ldx, (const T*)s,
                                     lds, (T*)dx, lddx, ds, HW, C, CT, accumulate));
    VT_CHECK_LAUNCH("vt_ese_gate_bwd");
    return VT_OK;
}

static int colsum_impl(const void* a, int32_t lda, int64_t M, int32_t C, int32_t dtype, float* out, int fixed,
                       void* stream) {
    VT_REQUIRE(M > 0 && out, VT_ERR_INVALID, "vt_colsum: bad argument");
    VT_TRY(check_mat("vt_colsum(a)", a, lda, C, dtype));
    const RowMap rm = RowMap::make(C, vt_epc(dtype), M, 256);
    VT_DISPATCH_T(dtype, "vt_colsum",
                  hipLaunchKernelGGL(colsum_kernel<T>, dim3(rm.blocks(M)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)a, lda, (long)M, C, rm, out, fixed));
    VT_CHECK_LAUNCH("vt_colsum");
    return VT_OK;
}

int vt_colsum(const void* a, int32_t lda, int64_t M, int32_t C, int32_t dtype, float* out, void* stream) {
    return colsum_impl(a, lda, M, C, dtype, out, 0, stream);
}

int vt_colsum_fixed(const void* a, int32_t lda, int64_t M, int32_t C, int32_t dtype, void* q, void* stream) {
    return colsum_impl(a, lda, M, C, dtype, (float*)q, 1, stream);
}

int vt_fixed_to_f32(const void* q, float* dst, int64_t n, int32_t accumulate, void* stream) {
    VT_REQUIRE(q && dst && n > 0, VT_ERR_INVALID, "vt_fixed_to_f32: bad argument");
    hipLaunchKernelGGL(fixed_to_f32_kernel, dim3(flat_blocks(n)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const long long*)q, dst, (long)n, accumulate);
    VT_CHECK_LAUNCH("vt_fixed_to_f32");
    return VT_OK;
}

int vt_softmax_xent(const void* logits, int32_t ldl, const int64_t* labels, float label_smoothing,
                    float grad_scale, float* loss, void* dlogits, int32_t lddl, int32_t B, int32_t N,
                    int32_t dtype, void* stream) {
    VT_REQUIRE(logits && labels && loss && B > 0 && N > 0 && ldl >= N, VT_ERR_INVALID,
               "vt_softmax_xent: bad argument");
    VT_REQUIRE(!dlogits || lddl >= N, VT_ERR_INVALID, "vt_softmax_xent: lddl < N");
    VT_DISPATCH_T(dtype, "vt_softmax_xent",
                  hipLaunchKernelGGL(xent_kernel<T>, dim3(B), dim3(kThreads), 0, (hipStream_t)stream,
                                     (const T*)logits, ldl, labels, label_smoothing, grad_scale, loss,
                                     (T*)dlogits, lddl, B, N, (const float*)nullptr));
    VT_CHECK_LAUNCH("vt_softmax_xent");
    return VT_OK;
}

int vt_softmax_xent_eval(const void* logits, int32_t ldl, const int64_t* labels, float* out3, int32_t B, int32_t N,
                         int32_t dtype, void* stream) {
    VT_REQUIRE(logits && labels && out3 && B > 0 && N > 0 && ldl >= N, VT_ERR_INVALID, "vt_softmax_xent_eval: bad argument");
    VT_DISPATCH_T(dtype, "vt_softmax_xent_eval",
                  hipLaunchKernelGGL(xent_eval_kernel<T>, dim3(B), dim3(kThreads), 0, (hipStream_t)stream,
                                     (const T*)logits, ldl, labels, out3, B, N));
    VT_CHECK_LAUNCH("vt_softmax_xent_eval");
    return VT_OK;
}

int vt_softmax_xent_mix(const void* logits, int32_t ldl, const int64_t* labels, float label_smoothing,
                        float grad_scale, float* loss, void* dlogits, int32_t lddl, int32_t B, int32_t N,
                        int32_t dtype, const float* mix, void* stream) {
    VT_REQUIRE(logits && labels && loss && mix && B > 0 && N > 0 && ldl >= N, VT_ERR_INVALID,
               "vt_softmax_xent_mix: bad argument");
    VT_REQUIRE(!dlogits || lddl >= N, VT_ERR_INVALID, "vt_softmax_xent_mix: lddl < N");
    VT_DISPATCH_T(dtype, "vt_softmax_xent_mix",
                  hipLaunchKernelGGL(xent_kernel<T>, dim3(B), dim3(kThreads), 0, (hipStream_t)stream,
                                     (const T*)logits, ldl, labels, label_smoothing, grad_scale, loss,
                                     (T*)dlogits, lddl, B, N, mix));
    VT_CHECK_LAUNCH("vt_softmax_xent_mix");
    return VT_OK;
}

int vt_sgd_momentum(float* p, const float* g, float* m, void* mirror, int32_t mirror_dtype, int64_t n,
                    float lr, float momentum, float weight_decay, float grad_scale, const float* lr_dev,
                    void* stream) {
    VT_REQUIRE(p && g && m && n > 0, VT_ERR_INVALID, "vt_sgd_momentum: bad argument");
    VT_REQUIRE(vt_aligned16(p) && vt_aligned16(g) && vt_aligned16(m), VT_ERR_INVALID,
               "vt_sgd_momentum: p/g/m must be 16-byte aligned");
    const unsigned blocks = flat_blocks(n, 4);
    if (mirror && mirror_dtype == VT_BF16)
        hipLaunchKernelGGL(sgd_kernel<bf16_t>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, p, g, m,
                           (bf16_t*)mirror, (long)n, lr, momentum, weight_decay, grad_scale, lr_dev);
    else if (mirror && mirror_dtype != VT_F32) {
        vt_set_error("vt_sgd_momentum: mirror dtype %d", mirror_dtype);
        return VT_ERR_UNSUPPORTED;
    } else
        hipLaunchKernelGGL(sgd_kernel<float>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, p, g, m,
                           (float*)mirror, (long)n, lr, momentum, weight_decay, grad_scale, lr_dev);
    VT_CHECK_LAUNCH("vt_sgd_momentum");
    return VT_OK;
}

int vt_copy2d(const void* src, int32_t src_dtype, int64_t lds, void* dst, int32_t dst_dtype, int64_t ldd,
              int64_t rows, int32_t cols, int32_t accumulate, void* stream) {
    VT_REQUIRE(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols, VT_ERR_INVALID,
               "vt_copy2d: bad argument");
    const unsigned blocks = flat_blocks(rows * cols);
    hipStream_t st = (hipStream_t)stream;
#define VT_CP(S, D)                                                                                         \
    hipLaunchKernelGGL((copy2d_kernel<S, D>), dim3(blocks), dim3(kThreads), 0, st, (const S*)src, (long)lds, \
                       (D*)dst, (long)ldd, (long)rows, cols, accumulate)
    if (src_dtype == VT_F32 && dst_dtype == VT_F32)
        VT_CP(float, float);
    else if (src_dtype == VT_F32 && dst_dtype == VT_BF16)
        VT_CP(float, bf16_t);
    else if (src_dtype == VT_BF16 && dst_dtype == VT_F32)
        VT_CP(bf16_t, float);
    else if (src_dtype == VT_BF16 && dst_dtype == VT_BF16)
        VT_CP(bf16_t, bf16_t);
    else {
        vt_set_error("vt_copy2d: dtypes %d -> %d", src_dtype, dst_dtype);
        return VT_ERR_UNSUPPORTED;
    }
#undef VT_CP
    VT_CHECK_LAUNCH("vt_copy2d");
    return VT_OK;
}

int vt_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                    int32_t dtype, void* stream) {
    VT_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, VT_ERR_INVALID,
               "vt_nchw_to_nhwc: bad argument");
    const long total = (long)B * H * W;
    VT_DISPATCH_T(dtype, "vt_nchw_to_nhwc",
                  hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(flat_blocks(total)), dim3(kThreads), 0,
                                     (hipStream_t)stream, x, (T*)y, C, (long)H * W, Cpad, total));
    VT_CHECK_LAUNCH("vt_nchw_to_nhwc");
    return VT_OK;
}

int vt_mix_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                        int32_t dtype, const float* mix, void* stream) {
    VT_REQUIRE(x && y && mix && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, VT_ERR_INVALID,
               "vt_mix_nchw_to_nhwc: bad argument");
    const long total = (long)B * H * W;
    VT_DISPATCH_T(dtype, "vt_mix_nchw_to_nhwc",
                  hipLaunchKernelGGL(mix_nchw_to_nhwc_kernel<T>, dim3(flat_blocks(total)), dim3(kThreads), 0,
                                     (hipStream_t)stream, x, (T*)y, B, C, W, (long)H * W, Cpad, total, mix));
    VT_CHECK_LAUNCH("vt_mix_nchw_to_nhwc");
    return VT_OK;
}

int vt_nhwc_to_nchw(const void* y, int32_t ldy, float* x, int32_t B, int32_t C, int32_t H, int32_t W,
                    int32_t dtype, void* stream) {
    VT_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && ldy >= C, VT_ERR_INVALID,
               "vt_nhwc_to_nchw: bad argument");
    const long total = (long)B * H * W;
    VT_DISPATCH_T(dtype, "vt_nhwc_to_nchw",
                  hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(flat_blocks(total)), dim3(kThreads), 0,
                                     (hipStream_t)stream, (const T*)y, ldy, x, C, (long)H * W, total));
    VT_CHECK_LAUNCH("vt_nhwc_to_nchw");
    return VT_OK;
}

int vt_pack_dgrad_filter(const void* w, int32_t src_dtype, int32_t ldw, void* out, int32_t dst_dtype,
                         const int32_t* sel_host, int32_t nsel, int32_t Cout, int32_t ntaps, int32_t Cin,
                         void* stream) {
    VT_REQUIRE(w && out && sel_host && nsel >= 1 && nsel <= VT_MAX_TAPS && Cout > 0 && Cin > 0 &&
                   ntaps >= 1 && ntaps <= VT_MAX_TAPS && ldw >= ntaps * Cin,
               VT_ERR_INVALID, "vt_pack_dgrad_filter: bad argument");
    SelTable st;
    for (int i = 0; i < nsel; ++i) {
        VT_REQUIRE(sel_host[i] >= -1 && sel_host[i] < ntaps, VT_ERR_INVALID,
                   "vt_pack_dgrad_filter: sel[%d]=%d outside [-1,%d)", i, sel_host[i], ntaps);
        st.sel[i] = sel_host[i];
    }
    const long total = (long)Cin * nsel * Cout;
    const unsigned blocks = flat_blocks(total);
    hipStream_t s = (hipStream_t)stream;
#define VT_PK(S, D)                                                                                          \
    hipLaunchKernelGGL((pack_dgrad_kernel<S, D>), dim3(blocks), dim3(kThreads), 0, s, (const S*)w, ldw, (D*)out, \
                       st, nsel, Cout, Cin, total)
    if (src_dtype == VT_F32 && dst_dtype == VT_F32)
        VT_PK(float, float);
    else if (src_dtype == VT_F32 && dst_dtype == VT_BF16)
        VT_PK(float, bf16_t);
    else if (src_dtype == VT_BF16 && dst_dtype == VT_BF16)
        VT_PK(bf16_t, bf16_t);
    else {
        vt_set_error("vt_pack_dgrad_filter: dtypes %d -> %d", src_dtype, dst_dtype);
        return VT_ERR_UNSUPPORTED;
    }
#undef VT_PK
    VT_CHECK_LAUNCH("vt_pack_dgrad_filter");
    return VT_OK;
}

int vt_pack_dgrad_filter_batch(const vt_pack_item* items, int32_t n, void* stream) {
    VT_REQUIRE(items && n >= 1, VT_ERR_INVALID, "vt_pack_dgrad_filter_batch: bad argument");
    for (int base = 0; base < n; base += VT_PACK_BATCH) {
        PackBatch b;
        memset(&b, 0, sizeof(b));
        b.n = n - base < VT_PACK_BATCH ? n - base : VT_PACK_BATCH;
        unsigned first = 0;
        for (int k = 0; k < b.n; ++k) {
            const vt_pack_item& s = items[base + k];
            VT_REQUIRE(s.w && s.out && s.nsel >= 1 && s.nsel <= VT_MAX_TAPS && s.Cout > 0 && s.Cin > 0 && s.ntaps >= 1 &&
                           s.ntaps <= VT_MAX_TAPS && s.ldw >= s.ntaps * s.Cin,
                       VT_ERR_INVALID, "vt_pack_dgrad_filter_batch: bad item %d", base + k);
            VT_REQUIRE(s.Cout % 8 == 0 && s.Cin % 8 == 0 && s.ldw % 8 == 0 && vt_aligned16(s.w) && vt_aligned16(s.out),
                       VT_ERR_UNSUPPORTED, "vt_pack_dgrad_filter_batch: item %d: Cout, Cin, ldw must be multiples of 8 and "
                       "both images 16-byte aligned", base + k);
            PackBatchItem& d = b.it[k];
            d.w = (const bf16_t*)s.w, d.out = (bf16_t*)s.out;
            d.ldw = s.ldw, d.nsel = s.nsel, d.Cout = s.Cout, d.Cin = s.Cin;
            for (int i = 0; i < s.nsel; ++i) {
                VT_REQUIRE(s.sel[i] >= -1 && s.sel[i] < s.ntaps, VT_ERR_INVALID,
                           "vt_pack_dgrad_filter_batch: item %d sel[%d]=%d outside [-1,%d)", base + k, i, s.sel[i], s.ntaps);
                d.sel[i] = s.sel[i];
            }
            const long blocks = (long)s.nsel * ((s.Cout + 63) / 64) * ((s.Cin + 63) / 64);  // one 64 x 64 tile of one tap each
            d.first = first, d.blocks = (unsigned)blocks;
            first += (unsigned)blocks;
        }
        hipLaunchKernelGGL(pack_dgrad_batch_kernel, dim3(first), dim3(kThreads), 0, (hipStream_t)stream, b);
        VT_CHECK_LAUNCH("vt_pack_dgrad_filter_batch");
    }
    return VT_OK;
}

}  // extern "C"
