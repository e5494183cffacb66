// vt_pointwise.hip -- the 1x1 ConvNormAct unit as four streaming kernels that never materialise the
// pre-activation z nor its gradient dz.
//
// What they replace: a 1x1 `ConvNormAct` (reference vision_toolbox/components.py:26-44 = nn.Conv2d(k=1, bias=False)
// -> nn.BatchNorm2d -> nn.ReLU) in training mode, and its autograd backward.  A 1x1 conv with <= 128 channels at
// batch 256 is HBM-bound by a factor 4-25 (AI 16-128 flop/B against a chip balance of ~400), and the
// decomposition conv -> z, (z -> y), (dy, z -> sums), (dy, z -> dz), (x, dz -> dW), (dz -> dx) moves 14 tensors of the
// unit's size.  z = W x is cheap to RECOMPUTE from the unit's input, so here
//
//   forward    pw<STATS>  : read x                      -> batch statistics of z (nothing written)
//              pw<APPLY>  : read x (+ residual)         -> y = relu(z * scale + shift) (+ residual)
//   backward   pw<REDUCE> : read dy, x                  -> sum g, sum g * xhat           (g = dy * [y > 0])
//              pw<BWD>    : read dy, x (+ addend)       -> dx = W^T dz (+ addend),  dW += dz^T x
//                                                          with dz = a*g - b*z + d formed in registers
//
// i.e. 8-9 tensor passes instead of 14, and 4 launches instead of 6.  Every kernel recomputes z with the SAME
// instruction sequence (same MFMA, same k order, rounded to bf16 exactly where the unfused path stores it), so the
// four agree bit for bit on z, and the numerics are those of the unfused bf16 path.
//
// Two "output groups" share one launch: CSPDarknetStage's conv1 and conv2 (reference backbones/darknet.py:46-47,53)
// read the same tensor, so they run as ONE GEMM with N = C whose channel halves have their own weights, BatchNorm
// parameters, statistics and destinations.
//
// Mapping (wave64, v_mfma_f32_16x16x32_bf16; lane = 16*q + pl):
//   * a wave owns units of 32 pixels (two 16-pixel tiles) and never talks to another wave inside the loop: no
//     barrier, no LDS traffic except the (read-only) weights and coefficients.
//   * z^T[n][p] = sum_c W[n][c] x[p][c]:  A = W rows (from LDS / registers), B = x: lane (pl, q) loads the 16 bytes
//     x[p0 + pl][32 s + 8 q ..] straight from global memory -- a 16-pixel tile of a 32-channel tensor is one
//     contiguous KiB per instruction.  The A rows of filter tile 2u+h are the channels 32u + 8(i>>2) + 4h + (i&3), so
//     the lane ends with the EIGHT consecutive channels 32u + 8q .. +7 of its pixel in two accumulators: dy, the
//     residual and y are 16-byte accesses in exactly that layout.
//   * dz (bf16) in that layout IS the B operand of the data gradient (k = n in natural order):
//     dx^T[c][p] = sum_n W[n][c] dz[p][n], A = W^T fragments formed by ds_read_b64_tr_b16 from the same LDS image of W
//     (the row permutation that gives 8 consecutive channels per lane costs nothing: a transposing read takes its four
//     4-column pieces from four independent addresses).
//   * dW[n][c] = sum_p dz[p][n] x[p][c] needs both operands pixel-major; the two transposes run on the (otherwise
//     idle) matrix pipe: T = src * E with E a 0/1 selector fragment is exact and leaves lane (pl, q) with pixels
//     4q..4q+3 of channel pl.  Two tiles give the 8 k-slots {4q+r} u {16+4q+r} -- the same permutation for dz and
//     x, so the product is unchanged.  dW lives in accumulators for the whole kernel (N*K <= 4096) and leaves through
//     an LDS fold + one f32 atomic pass per workgroup.
//   * BatchNorm sums: per-lane partials over the wave's pixels, folded across the 16 pixel lanes by shuffles, across
//     the waves in LDS in a FIXED order, then one fixed-point integer atomic per channel and moment (vt_common.h):
//     bit-identical from run to run like every other statistic of the library.
#include <stdlib.h>
#include <string.h>

#include <hip/hip_ext.h>

#include "vt_common.h"
#include "vt_bn_fin.h"

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

enum { PW_STATS = 0, PW_APPLY = 1, PW_REDUCE = 2, PW_BWD = 3 };

// units prefetched ahead of the one being computed: as many as the registers take without costing a wave per SIMD
constexpr int pw_depth(int N, int K, int mode) {
#ifdef VT_PW_DEPTH
    return VT_PW_DEPTH;
#else
    return 1;  // (deeper rings measured equal or slower: they cost a wave per SIMD)
#endif
}

struct PwArgs {
    const bf16_t* x;
    int ldx;
    long M;
    int N0;  // channels of output group 0 (== N: a single group)
    const bf16_t* w[2];
    int ldw[2];
    bf16_t* y[2];          // APPLY: outputs
    const bf16_t* dy[2];   // REDUCE / BWD
    int ldy[2];            // pixel stride of y / dy
    const bf16_t* res[2];  // APPLY: residual operands (optional)
    int ldr[2];
    float* stats[2];        // STATS: statistics buffers; REDUCE: sums buffers (per group, C = that group's channels)
    const float* coef;      // [4][N]: scale | shift | mean | invstd (groups contiguous)
    const float* bcoef[2];  // BWD: [3][C_g] = a | b | d of vt_bn_bwd_finalize
    bf16_t* dx;             // BWD: data gradient [M][K]
    int lddx;
    const bf16_t* add;  // BWD: folded addend of dx (optional; may alias dx)
    int ldadd;
    float* dw[2];  // BWD: filter gradients [C_g][K] (f32, accumulated); null: not wanted
    int lddw[2];
    bf16_t* dz[2];  // BWD, shapes whose dW does not fit the accumulators: dz is written for vt_conv_wgrad (optional)
    int lddz[2];
    int relu;
    int nunits;  // ceil(M / 16)
    // apply pass with channel counts that are no multiples of 32 (YOLOv5x's 80-channel stage in inference, round 4): the
    // kernel is the next multiple of 32 wide and these are the real widths -- filter rows / columns, coefficients and
    // outputs beyond them do not exist (zero rows / columns of the LDS image, masked stores), and an x or residual load
    // beyond them reads the pixel's first 16 bytes instead (times a zero filter column / never stored)
    int Nr, Kr;
    // vt_pw_fwd_apply_finalize / vt_pw_bwd_apply_finalize: the BatchNorm finalize step of each group runs in the prologue of
    // EVERY workgroup (it stages the coefficients in LDS anyway), from the complete sums of the pass before -- see
    // vt_bn_finalize_apply in vt_elementwise.hip; workgroup 0 stores the results.  N <= 128 (a thread pair per channel).
    int fin;
    VtFinFwd ffin[2];  // APPLY
    VtFinBwd bfin[2];  // BWD
};

__device__ __forceinline__ uint4 ldg16(const bf16_t* p) { return *(const uint4*)p; }
__device__ __forceinline__ bf16x8 as_frag(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float bfround(float v) { return (float)(bf16_t)v; }

template <int N, int K>
struct PwGeom {
    static constexpr int NT = N / 16, NU = N / 32, KS = K / 32, CT = K / 16, KV = K / 32;
    static constexpr int PITCH = 2 * K + 16;          // bytes per row of the LDS image of W (padded: bank spread)
    static constexpr int W_BYTES = N * PITCH;
    static constexpr int COEF_OFF = W_BYTES;          // float [5][N]
    static constexpr int RED_OFF = COEF_OFF + 5 * N * 4;
    static constexpr bool kFull = N * K <= 4096;      // dW accumulated in registers inside the backward kernel
#ifndef VT_PW_REGW
#define VT_PW_REGW 2048
#endif
    static constexpr bool kRegW = N * K <= VT_PW_REGW;  // W fragments held in registers (else re-read from LDS per tile)
    // scratch after the coefficients: statistics fold float [4 waves][2][N]; dW fold float [N][K]
    static constexpr int RED_BYTES = (kFull ? N * K * 4 : 0) > 4 * 2 * N * 4 ? N * K * 4 : 4 * 2 * N * 4;
    static constexpr int SMEM = RED_OFF + RED_BYTES;
};

// one workgroup = 4 independent waves; see the header for the lane mapping
// EX: the optional operand is present (APPLY: a residual for EVERY group; BWD: the addend of dx) -- compile time, so
// that the loads of a unit are unconditional and the compiler can count them (s_waitcnt vmcnt(N) instead of 0)
template <int N, int K, int MODE, bool EX>
__global__ void __launch_bounds__(256) pw_kernel(const PwArgs a) {
    using G = PwGeom<N, K>;
    constexpr int NT = G::NT, NU = G::NU, KS = G::KS, CT = G::CT, KV = G::KV, PITCH = G::PITCH;
    constexpr bool kDW = (MODE == PW_BWD) && G::kFull;
    constexpr int kDepth = pw_depth(N, K, MODE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    float* sCoef = (float*)(smem + G::COEF_OFF);
    float* sRed = (float*)(smem + G::RED_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, q = lane >> 4;
    const int N0 = a.N0;

    // ---- stage W (both groups) and the per-channel coefficients ---------------------------------------------
    for (int idx = tid; idx < N * (K / 8); idx += 256) {
        const int n = idx / (K / 8), ch = idx - n * (K / 8);
        const int g = n >= N0;
        const bf16_t* src = a.w[g] + (long)(n - (g ? N0 : 0)) * a.ldw[g] + ch * 8;
        const bool real = n < a.Nr && ch * 8 < a.Kr;
        *(uint4*)(sW + n * PITCH + ch * 16) = real ? ldg16(src) : make_uint4(0, 0, 0, 0);
    }
    bool fin_done = false;
    if constexpr ((MODE == PW_APPLY || MODE == PW_BWD) && N <= 128) {
        if (a.fin) {
            // thread pair (w, n): the two sums of channel n, 16 replicas x 2 limbs in one round of loads
            fin_done = true;
            const int w = tid & 1, n = tid >> 1;
            const bool on = n < a.Nr;
            const int g = (on && n >= N0) ? 1 : 0, lc = on ? n - (g ? N0 : 0) : 0;
            if constexpr (MODE == PW_APPLY) {
                const VtFinFwd& f = a.ffin[g];
                const VtFinFwdPre pre = vt_fin_fwd_pre(f, lc);
                const double v = vt_replica_sum<false>(f.stats, (long)w * f.C + lc, 2L * f.C);
                const double o = __shfl_xor(v, 1, 64);
                if (on && w == 0) {
                    float sc, sf;
                    vt_fin_fwd_channel(f, lc, v, o, pre.g, pre.b, pre.rm, pre.rv, sc, sf, blockIdx.x == 0);
                    sCoef[n] = sc, sCoef[N + n] = sf;
                }
            } else {
                const VtFinBwd& f = a.bfin[g];
                const VtFinBwdPre pre = vt_fin_bwd_pre(f, lc);
                const float sh = on ? a.coef[N + n] : 0.f;
                const double v = vt_replica_sum<false>(f.sums, (long)w * f.C + lc, 2L * f.C);
                const double o = __shfl_xor(v, 1, 64);
                if (on && w == 0) {
                    float b, d;
                    vt_fin_bwd_channel(f, lc, v, o, pre.a, pre.mu, pre.istd, pre.dg, pre.db, b, d, blockIdx.x == 0);
                    sCoef[n] = pre.a, sCoef[N + n] = sh, sCoef[2 * N + n] = pre.a, sCoef[3 * N + n] = b, sCoef[4 * N + n] = d;
                }
            }
            for (int i = tid; i < 5 * N; i += 256) {  // rows and channels the pairs above do not own
                const int which = i / N, n2 = i - which * N;
                if (n2 >= a.Nr || which >= (MODE == PW_APPLY ? 2 : 5)) sCoef[i] = 0.f;
            }
        }
    }
    if (MODE != PW_STATS && !fin_done) {
        for (int i = tid; i < 5 * N; i += 256) {
            const int which = i / N, n = i - which * N;
            float v = 0.f;
            if (MODE == PW_BWD) {
                // scale | shift | a | b | d
                if (which < 2)
                    v = a.coef[which * N + n];
                else {
                    const int g = n >= N0, Cg = g ? N - N0 : N0;
                    v = a.bcoef[g][(which - 2) * Cg + (n - (g ? N0 : 0))];
                }
            } else if (which < 4 && n < a.Nr) {
                v = a.coef[which * a.Nr + n];  // scale | shift | mean | invstd
            }
            sCoef[i] = v;
        }
    }
    if (kDW) {
        for (int i = tid; i < N * K; i += 256) sRed[i] = 0.f;
    }
    __syncthreads();

    // ---- fragment addressing ------------------------------------------------------------------------------------
    // z: A rows of filter tile nt = 2u + h: channel 32u + 8(pl>>2) + 4h + (pl&3), k = 32s + 8q .. +7
    const unsigned za_base = (unsigned)((8 * (pl >> 2) + (pl & 3)) * PITCH + 16 * q);
    // dx: A = W^T: rows n = 32u + 8q + (pl>>2) (+4), columns 32v + 8(pl&3) + 4h .. +3
    const unsigned ta_base = (unsigned)((8 * q + (pl >> 2)) * PITCH + 16 * (pl & 3));
    auto wz_frag = [&](int nt, int s) -> bf16x8 {
        const uint4 v = *(const uint4*)(sW + za_base + (nt >> 1) * 32 * PITCH + (nt & 1) * 4 * PITCH + s * 64);
        return as_frag(v);
    };
    auto wt_frag = [&](int ct, int u) -> bf16x8 {
        const char* p = sW + ta_base + u * 32 * PITCH + (ct >> 1) * 64 + (ct & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * PITCH));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    bf16x8 wz[G::kRegW ? NT : 1][G::kRegW ? KS : 1];
    bf16x8 wt[(G::kRegW && MODE == PW_BWD) ? CT : 1][(G::kRegW && MODE == PW_BWD) ? NU : 1];
    if constexpr (G::kRegW) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < KS; ++s) wz[nt][s] = wz_frag(nt, s);
        if constexpr (MODE == PW_BWD) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int u = 0; u < NU; ++u) wt[ct][u] = wt_frag(ct, u);
        }
    }
    // selector fragments of the transposes: E_tt[k = 8q + j][col = pl] = (8q + j == 16 tt + pl)
    bf16x8 sel[2];
    if constexpr (kDW) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            s16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (8 * q + j == 16 * tt + pl) ? (short)0x3F80 : (short)0;
            sel[tt] = __builtin_bit_cast(bf16x8, v);
        }
    }

    // ---- accumulators that live for the whole kernel ----------------------------------------------------------
    float s1[NU][8], s2[NU][8];  // STATS: sum z, sum z^2;  REDUCE: sum g, sum g * (z - mean)
    if constexpr (MODE == PW_STATS || MODE == PW_REDUCE) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) s1[u][e] = s2[u][e] = 0.f;
    }
    f32x4 dwacc[kDW ? NT : 1][kDW ? CT : 1];
    if constexpr (kDW) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dwacc[nt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- per-unit registers: this unit (16 pixels) and the prefetched next one ----------------------------------
    struct Regs {
        uint4 x[KS];
        uint4 dy[(MODE >= PW_REDUCE) ? NU : 1];
        uint4 ex[(MODE == PW_APPLY) ? NU : (MODE == PW_BWD ? KV : 1)];
    };
    auto group_of = [&](int u) -> int { return 32 * u >= N0; };

    // (nothing here may READ a loaded register: a select on `ok` right behind the load made the compiler wait for the
    // prefetch it had just issued -- vmcnt(0) at the head of every unit.  A pixel past the end loads the last pixel's
    // rows; its contributions are masked where they are used.)
    auto load_unit = [&](Regs& r, int unit) {
        const long p = (long)unit * 16 + pl;
        const long pc = p < a.M ? p : a.M - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) r.x[s] = ldg16(a.x + pc * a.ldx + (32 * s + 8 * q < a.Kr ? 32 * s + 8 * q : 0));
        if constexpr (MODE >= PW_REDUCE) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int g = group_of(u);
                r.dy[u] = ldg16(a.dy[g] + pc * a.ldy[g] + (32 * u - (g ? N0 : 0)) + 8 * q);
            }
        }
        if constexpr (MODE == PW_APPLY) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int g = group_of(u);
                if constexpr (EX)
                    r.ex[u] = ldg16(a.res[g] + pc * a.ldr[g] + (32 * u + 8 * q < a.Nr ? (32 * u - (g ? N0 : 0)) + 8 * q : 0));
            }
        }
        if constexpr (MODE == PW_BWD) {
            if constexpr (EX) {
#pragma unroll
                for (int v = 0; v < KV; ++v) r.ex[v] = ldg16(a.add + pc * a.ldadd + 32 * v + 8 * q);
            }
        }
    };

    auto compute_unit = [&](const Regs& r, int unit) {
        // (keeps the loop-invariant LDS reads -- coefficients, W fragments -- inside the loop: hoisted, they would
        // occupy up to 5 N / 4 + N K / 128 registers per lane for the whole kernel)
        asm volatile("" ::: "memory");
        const long p = (long)unit * 16 + pl;
        const bool ok = p < a.M;
        // ---- z^T = W x ----
        f32x4 zacc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            zacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 A = G::kRegW ? wz[G::kRegW ? nt : 0][G::kRegW ? s : 0] : wz_frag(nt, s);
                zacc[nt] = mma(A, as_frag(r.x[s]), zacc[nt]);
            }
        }
        bf16x8 dzt[(MODE == PW_BWD) ? NU : 1];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = bfround(zacc[2 * u + (e >> 2)][e & 3]);
            const int cb = 32 * u + 8 * q;
            if constexpr (MODE == PW_STATS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float zz = ok ? z[e] : 0.f;
                    s1[u][e] += zz;
                    s2[u][e] = fmaf(zz, zz, s2[u][e]);
                }
            } else {
                float sc[8], sf[8];
                *(f32x4*)&sc[0] = *(const f32x4*)(sCoef + cb);
                *(f32x4*)&sc[4] = *(const f32x4*)(sCoef + cb + 4);
                *(f32x4*)&sf[0] = *(const f32x4*)(sCoef + N + cb);
                *(f32x4*)&sf[4] = *(const f32x4*)(sCoef + N + cb + 4);
                if constexpr (MODE == PW_APPLY) {
                    const int g = group_of(u);
                    float v[8], rr[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) rr[e] = 0.f;
                    if constexpr (EX) VecIO<bf16_t>::unpack(r.ex[u], rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = fmaf(z[e], sc[e], sf[e]);
                        v[e] = a.relu ? fmaxf(v[e], 0.f) : v[e];
                        v[e] += rr[e];
                    }
                    if (ok && 32 * u + 8 * q < a.Nr)
                        *(uint4*)(a.y[g] + p * a.ldy[g] + (32 * u - (g ? N0 : 0)) + 8 * q) = VecIO<bf16_t>::pack(v);
                } else {
                    float gy[8];
                    VecIO<bf16_t>::unpack(r.dy[u], gy);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gy[e] = (ok && (!a.relu || fmaf(z[e], sc[e], sf[e]) > 0.f)) ? gy[e] : 0.f;
                    if constexpr (MODE == PW_REDUCE) {
                        float mu[8];
                        *(f32x4*)&mu[0] = *(const f32x4*)(sCoef + 2 * N + cb);
                        *(f32x4*)&mu[4] = *(const f32x4*)(sCoef + 2 * N + cb + 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            s1[u][e] += gy[e];
                            s2[u][e] = fmaf(gy[e], z[e] - mu[e], s2[u][e]);
                        }
                    } else {
                        float ca[8], cbb[8], cd[8], dz[8];
                        *(f32x4*)&ca[0] = *(const f32x4*)(sCoef + 2 * N + cb);
                        *(f32x4*)&ca[4] = *(const f32x4*)(sCoef + 2 * N + cb + 4);
                        *(f32x4*)&cbb[0] = *(const f32x4*)(sCoef + 3 * N + cb);
                        *(f32x4*)&cbb[4] = *(const f32x4*)(sCoef + 3 * N + cb + 4);
                        *(f32x4*)&cd[0] = *(const f32x4*)(sCoef + 4 * N + cb);
                        *(f32x4*)&cd[4] = *(const f32x4*)(sCoef + 4 * N + cb + 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            dz[e] = fmaf(ca[e], gy[e], fmaf(-cbb[e], z[e], cd[e]));
                            dz[e] = ok ? dz[e] : 0.f;  // (a pixel past the end must not reach dW)
                        }
                        const uint4 dzp = VecIO<bf16_t>::pack(dz);
                        dzt[u] = as_frag(dzp);
                        if constexpr (!kDW) {
                            const int g = group_of(u);
                            if (a.dz[g] && ok) *(uint4*)(a.dz[g] + p * a.lddz[g] + (32 * u - (g ? N0 : 0)) + 8 * q) = dzp;
                        }
                    }
                }
            }
        }
        if constexpr (MODE == PW_BWD) {
            // ---- dx^T = W^T dz (+ addend) ----
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                f32x4 xa[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    xa[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const bf16x8 A = G::kRegW ? wt[G::kRegW ? 2 * v + h : 0][G::kRegW ? u : 0] : wt_frag(2 * v + h, u);
                        xa[h] = mma(A, dzt[u], xa[h]);
                    }
                }
                float o[8], ad[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ad[e] = 0.f;
                if constexpr (EX) VecIO<bf16_t>::unpack(r.ex[v], ad);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = xa[e >> 2][e & 3] + ad[e];
                if (ok) *(uint4*)(a.dx + p * a.lddx + 32 * v + 8 * q) = VecIO<bf16_t>::pack(o);
            }
        }
        if constexpr (kDW) {
            // ---- dW += dz^T x over the tile's 16 pixels.  Both operands are transposed on the matrix pipe: src * E
            // leaves lane (pl, q) with pixels 4q .. 4q+3 of channel 16 tt + pl (exact: E is 0/1), which is the
            // operand layout of v_mfma_f32_16x16x16_bf16 (k = 4q + j). ----
            s16x4 dzT[NT], xT[CT];
            const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 t0 = mma(dzt[nt >> 1], sel[nt & 1], z4);
                const uint32_t lo = VecIO<bf16_t>::pack2(t0[0], t0[1]), hi = VecIO<bf16_t>::pack2(t0[2], t0[3]);
                dzT[nt] = __builtin_bit_cast(s16x4, make_uint2(lo, hi));
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const f32x4 t0 = mma(as_frag(r.x[ct >> 1]), sel[ct & 1], z4);
                const uint32_t lo = VecIO<bf16_t>::pack2(t0[0], t0[1]), hi = VecIO<bf16_t>::pack2(t0[2], t0[3]);
                xT[ct] = __builtin_bit_cast(s16x4, make_uint2(lo, hi));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    dwacc[nt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dzT[nt], xT[ct], dwacc[nt][ct], 0, 0, 0);
        }
    };

    // ---- the wave's units, kDepth of them prefetched ahead ------------------------------------------------------------
    // Every load of the steady-state loop is unconditional (a prefetch past the wave's last unit re-loads that unit), so
    // the compiler waits with a counted vmcnt for exactly the unit it is about to use.
    const int stride = (int)gridDim.x * 4;
    const int first = (int)blockIdx.x * 4 + wave;
    const int T = first < a.nunits ? (a.nunits - first + stride - 1) / stride : 0;  // units of this wave
    if (T > 0) {
        auto unit_of = [&](int i) { return first + (i < T ? i : T - 1) * stride; };
        Regs ring[kDepth + 1];
#pragma unroll
        for (int j = 0; j < kDepth; ++j) load_unit(ring[j], unit_of(j));
        int i = 0;
        for (; i + kDepth + 1 <= T; i += kDepth + 1) {
#pragma unroll
            for (int j = 0; j <= kDepth; ++j) {
                load_unit(ring[(j + kDepth) % (kDepth + 1)], unit_of(i + j + kDepth));
                compute_unit(ring[j], first + (i + j) * stride);
            }
        }
#pragma unroll
        for (int j = 0; j < kDepth; ++j)  // the last T - i < kDepth + 1 units are already in ring[0 ..]
            if (i + j < T) compute_unit(ring[j], first + (i + j) * stride);
    }

    // ---- per-channel sums: lanes -> wave (shuffles) -> workgroup (LDS, fixed order) -> fixed-point atomics -------
    if constexpr (MODE == PW_STATS || MODE == PW_REDUCE) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    s1[u][e] += __shfl_xor(s1[u][e], off, 64);
                    s2[u][e] += __shfl_xor(s2[u][e], off, 64);
                }
            }
        if (pl == 0) {
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sRed[(wave * 2 + 0) * N + 32 * u + 8 * q + e] = s1[u][e];
                    sRed[(wave * 2 + 1) * N + 32 * u + 8 * q + e] = s2[u][e];
                }
        }
        __syncthreads();
        const int rep = (int)(blockIdx.x % kStatReplicas);
        for (int i = tid; i < 2 * N; i += 256) {
            const int which = i / N, n = i - which * N;
            float acc = sRed[(0 * 2 + which) * N + n];
            acc += sRed[(1 * 2 + which) * N + n];
            acc += sRed[(2 * 2 + which) * N + n];
            acc += sRed[(3 * 2 + which) * N + n];
            if (MODE == PW_REDUCE && which) acc *= sCoef[3 * N + n];  // invstd
            const int g = n >= N0, Cg = g ? N - N0 : N0, c = n - (g ? N0 : 0);
            vt_stat_add(a.stats[g], ((long)rep * 2 + which) * Cg + c, acc);
        }
    }
    if constexpr (kDW) {
        // lane (pl, q) holds dW[16 nt + 4q + r][16 ct + pl]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&sRed[(16 * nt + 4 * q + r) * K + 16 * ct + pl], dwacc[nt][ct][r]);
        __syncthreads();
        // every workgroup adds its image to the SAME N x K addresses and all of them finish together: each starts at its
        // own 256-element chunk, so that the adds do not queue up address by address
        constexpr int NCH = N * K / 256;
        const int rot = (int)((blockIdx.x * 7u) % (unsigned)NCH);
        for (int j = 0; j < NCH; ++j) {
            int ch = j + rot;
            ch -= ch >= NCH ? NCH : 0;
            const int i = ch * 256 + tid;
            const int n = i / K, c = i - n * K;
            const int g = n >= N0;
            float* dst = a.dw[g];
            if (dst) atomicAdd(dst + (long)(n - (g ? N0 : 0)) * a.lddw[g] + c, sRed[i]);
        }
    }
}

template <int N, int K, int MODE, bool EX>
int launch_pw(const PwArgs& a, hipStream_t st, const char* who) {
    using G = PwGeom<N, K>;
    auto kern = pw_kernel<N, K, MODE, EX>;
    constexpr int smem = G::SMEM;
    static_assert(smem <= 160 * 1024, "weights exceed the LDS of a CU");
    if (smem > 64 * 1024) {
        const int rc = vt_raise_dynamic_lds((const void*)kern, smem, who);
        if (rc != VT_OK) return rc;
    }
    // every workgroup stages its own copy of W (up to 33 KiB): small tensors get fewer, longer workgroups
    // (the backward kernel that carries dW flushes N x K floats per workgroup through LDS and global atomics: one workgroup
    // per resident slot -- 2 per CU at its register count -- measured best: 64 x 64 at 0.8 M pixels 170 -> 117 us)
    // (round 3, in the step: 768 for every pass 21.49 vs 512 | 1024 21.55 ms at batch 256; at batch 128, where the largest
    //  launch has 100 k units, 512 measured 12.53 vs 12.65 ms: whole multiples of the CU count, three per CU from 160 k units)
    const int knob = (0);
    const int target = knob > 0 ? knob : (a.nunits >= 160000 ? 768 : 512);
    const int per_wave = (8);
    long blocks = ((long)a.nunits + 4 * per_wave - 1) / (4 * per_wave);
    if (blocks > target) blocks = target;
    if (blocks < 1) blocks = 1;
    vt_note_kernel("pw_kernel<%d,%d,%s>", N, K, MODE == PW_STATS ? "stats" : MODE == PW_APPLY ? "apply" : MODE == PW_REDUCE ? "reduce" : "bwd");
    VT_LAUNCH_STOP(kern, dim3((unsigned)blocks), dim3(256), smem, st, a);
    VT_CHECK_LAUNCH(who);
    return VT_OK;
}

template <int MODE>
int dispatch_pw(int N, int K, const PwArgs& a, hipStream_t st, const char* who) {
    const bool ex = MODE == PW_APPLY ? a.res[0] != nullptr : (MODE == PW_BWD ? a.add != nullptr : false);
#define VT_PW_CASE(n, k)                                                           \
    if (N == n && K == k) {                                                        \
        if constexpr (MODE == PW_APPLY || MODE == PW_BWD) {                        \
            if (ex) return launch_pw<n, k, MODE, true>(a, st, who);                \
        }                                                                          \
        return launch_pw<n, k, MODE, false>(a, st, who);                           \
    }
    VT_PW_CASE(32, 32)
    VT_PW_CASE(64, 64)
    VT_PW_CASE(32, 64)
    VT_PW_CASE(128, 128)
    VT_PW_CASE(64, 128)
    VT_PW_CASE(160, 160)
    VT_PW_CASE(160, 320)
#undef VT_PW_CASE
    if constexpr (MODE == PW_APPLY) {
        // the 80-channel shapes: one 32-channel unit more than fits, real widths in a.Nr / a.Kr
        if (N == 80 && K == 80) return ex ? launch_pw<96, 96, MODE, true>(a, st, who) : launch_pw<96, 96, MODE, false>(a, st, who);
        if (N == 80 && K == 160) return ex ? launch_pw<96, 160, MODE, true>(a, st, who) : launch_pw<96, 160, MODE, false>(a, st, who);
    }
    vt_set_error("%s: no pointwise kernel for %d -> %d channels", who, K, N);
    return VT_ERR_UNSUPPORTED;
}

bool shape_ok(int N, int K) {
    return (N == 32 && K == 32) || (N == 64 && K == 64) || (N == 32 && K == 64) || (N == 128 && K == 128) ||
           (N == 64 && K == 128) || (N == 160 && K == 160) || (N == 160 && K == 320);
}

bool ragged_apply_ok(int N, int K, int ngroups) { return ngroups == 1 && N == 80 && (K == 80 || K == 160); }

int fill_common(PwArgs& a, const vt_pw_desc* d, const char* who, bool apply = false) {
    VT_REQUIRE(d, VT_ERR_INVALID, "%s: null descriptor", who);
    VT_REQUIRE(d->dtype == VT_BF16, VT_ERR_UNSUPPORTED, "%s: bf16 only", who);
    VT_REQUIRE(d->M > 0 && d->M < 0x7fffffffL * 16, VT_ERR_INVALID, "%s: bad M", who);
    VT_REQUIRE(d->ngroups == 1 || d->ngroups == 2, VT_ERR_INVALID, "%s: ngroups %d", who, d->ngroups);
    const int N = d->C[0] + (d->ngroups == 2 ? d->C[1] : 0);
    const bool ragged = apply && ragged_apply_ok(N, d->K, d->ngroups);
    VT_REQUIRE(ragged || shape_ok(N, d->K), VT_ERR_UNSUPPORTED, "%s: no pointwise kernel for %d -> %d channels", who, d->K, N);
    VT_REQUIRE(ragged || (d->C[0] > 0 && d->C[0] % 32 == 0 && (d->ngroups == 1 || (d->C[1] > 0 && d->C[1] % 32 == 0))),
               VT_ERR_UNSUPPORTED, "%s: group widths must be multiples of 32", who);
    VT_REQUIRE(d->x && vt_aligned16(d->x) && d->ldx % 8 == 0 && d->ldx >= d->K, VT_ERR_INVALID, "%s: bad x", who);
    VT_REQUIRE(d->M * (int64_t)d->ldx < 0x7fffffff00LL, VT_ERR_UNSUPPORTED, "%s: tensor too large", who);
    memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)d->x;
    a.ldx = d->ldx;
    a.M = d->M;
    a.N0 = d->C[0];
    a.relu = d->relu;
    a.nunits = (int)((d->M + 15) / 16);
    a.Nr = N, a.Kr = d->K;
    for (int g = 0; g < d->ngroups; ++g) {
        VT_REQUIRE(d->w[g] && vt_aligned16(d->w[g]) && d->ldw[g] % 8 == 0 && d->ldw[g] >= d->K, VT_ERR_INVALID,
                   "%s: bad weights of group %d", who, g);
        a.w[g] = (const bf16_t*)d->w[g];
        a.ldw[g] = d->ldw[g];
    }
    return VT_OK;
}

// a [M][C_g] bf16 operand of group g
int check_rows(const char* who, const char* what, int g, const void* p, int ld, int C, bool optional) {
    if (!p && optional) return VT_OK;
    VT_REQUIRE(p && vt_aligned16(p) && ld % 8 == 0 && ld >= C, VT_ERR_INVALID, "%s: bad %s of group %d", who, what, g);
    return VT_OK;
}

}  // namespace

extern "C" {

int vt_pw_supported(int32_t dtype, int32_t K, int32_t C0, int32_t C1) {
    if (dtype != VT_BF16 || C0 <= 0 || C1 < 0 || C0 % 32 != 0 || C1 % 32 != 0) return 0;
    const int N = C0 + C1;
    if (!shape_ok(N, K)) return 0;
    return N * K <= 4096 ? 2 : 1;
}

int vt_pw_apply_supported(int32_t dtype, int32_t K, int32_t C0) {
    return dtype == VT_BF16 && (shape_ok(C0, K) || ragged_apply_ok(C0, K, 1));
}

int vt_pw_fwd_stats(const vt_pw_desc* d, float* const* stats, void* stream) {
    PwArgs a;
    int rc = fill_common(a, d, "vt_pw_fwd_stats");
    if (rc != VT_OK) return rc;
    for (int g = 0; g < d->ngroups; ++g) {
        VT_REQUIRE(stats && stats[g], VT_ERR_INVALID, "vt_pw_fwd_stats: null statistics buffer");
        a.stats[g] = stats[g];
    }
    return dispatch_pw<PW_STATS>(a.N0 + (d->ngroups == 2 ? d->C[1] : 0), d->K, a, (hipStream_t)stream, "vt_pw_fwd_stats");
}

static int pw_fwd_apply_impl(const vt_pw_desc* d, const float* coef, void* const* y, const int32_t* ldy, const void* const* res,
                             const int32_t* ldr, const vt_bn_fin_fwd* fin, void* stream) {
    PwArgs a;
    int rc = fill_common(a, d, "vt_pw_fwd_apply", true);
    if (rc != VT_OK) return rc;
    VT_REQUIRE(coef && y && ldy, VT_ERR_INVALID, "vt_pw_fwd_apply: null argument");
    a.coef = coef;
    if (fin) {  // (checked by vt_pw_fwd_apply_finalize)
        const int Ntot = d->C[0] + (d->ngroups == 2 ? d->C[1] : 0);
        a.fin = 1;
        for (int g = 0; g < d->ngroups; ++g) {
            const int off = g ? d->C[0] : 0;
            float* c = (float*)coef;
            a.ffin[g] = VtFinFwd{fin[g].stats, fin[g].gamma, fin[g].beta, fin[g].running_mean, fin[g].running_var,
                                 fin[g].num_batches_tracked, c + off, c + Ntot + off, c + 2 * Ntot + off, c + 3 * Ntot + off,
                                 1.0 / fin[g].count, fin[g].count > 1.0 ? fin[g].count / (fin[g].count - 1.0) : 1.0, fin[g].eps,
                                 fin[g].momentum, d->C[g]};
        }
    }
    const bool any_res = res && (res[0] || (d->ngroups == 2 && res[1]));
    for (int g = 0; g < d->ngroups; ++g) {
        VT_REQUIRE(!any_res || res[g], VT_ERR_UNSUPPORTED, "vt_pw_fwd_apply: a residual for every group or for none");
        if ((rc = check_rows("vt_pw_fwd_apply", "y", g, y[g], ldy[g], d->C[g], false)) != VT_OK) return rc;
        a.y[g] = (bf16_t*)y[g];
        a.ldy[g] = ldy[g];
        if (res && res[g]) {
            VT_REQUIRE(ldr, VT_ERR_INVALID, "vt_pw_fwd_apply: residual without a stride");
            if ((rc = check_rows("vt_pw_fwd_apply", "residual", g, res[g], ldr[g], d->C[g], false)) != VT_OK) return rc;
            a.res[g] = (const bf16_t*)res[g];
            a.ldr[g] = ldr[g];
        }
    }
    return dispatch_pw<PW_APPLY>(a.N0 + (d->ngroups == 2 ? d->C[1] : 0), d->K, a, (hipStream_t)stream, "vt_pw_fwd_apply");
}

int vt_pw_fwd_apply(const vt_pw_desc* d, const float* coef, void* const* y, const int32_t* ldy, const void* const* res,
                    const int32_t* ldr, void* stream) {
    return pw_fwd_apply_impl(d, coef, y, ldy, res, ldr, nullptr, stream);
}

int vt_pw_fwd_apply_finalize(const vt_pw_desc* d, const vt_bn_fin_fwd* fin, float* coef, void* const* y, const int32_t* ldy,
                             const void* const* res, const int32_t* ldr, void* stream) {
    VT_REQUIRE(d && fin && coef && (d->ngroups == 1 || d->ngroups == 2), VT_ERR_INVALID, "vt_pw_fwd_apply_finalize: null argument");
    const int N = d->C[0] + (d->ngroups == 2 ? d->C[1] : 0);
    for (int g = 0; g < d->ngroups; ++g) {
        VT_REQUIRE(fin[g].stats && fin[g].count > 0, VT_ERR_INVALID, "vt_pw_fwd_apply_finalize: bad statistics of group %d", g);
        VT_REQUIRE((fin[g].running_mean == nullptr) == (fin[g].running_var == nullptr), VT_ERR_INVALID,
                   "vt_pw_fwd_apply_finalize: running_mean/var must both be given or both NULL");
    }
    if (N > 128 || !VT_KNOB("VT_BN_FIN_APPLY", 1)) {
        for (int g = 0; g < d->ngroups; ++g) {
            const int off = g ? d->C[0] : 0;
            const int rc = vt_bn_finalize(fin[g].stats, d->C[g], fin[g].count, fin[g].gamma, fin[g].beta, fin[g].eps, fin[g].momentum,
                                          fin[g].running_mean, fin[g].running_var, fin[g].num_batches_tracked, coef + off,
                                          coef + N + off, coef + 2 * N + off, coef + 3 * N + off, stream);
            if (rc != VT_OK) return rc;
        }
        return pw_fwd_apply_impl(d, coef, y, ldy, res, ldr, nullptr, stream);
    }
    return pw_fwd_apply_impl(d, coef, y, ldy, res, ldr, fin, stream);
}

int vt_pw_bwd_reduce(const vt_pw_desc* d, const float* coef, const void* const* dy, const int32_t* lddy,
                     float* const* sums, void* stream) {
    PwArgs a;
    int rc = fill_common(a, d, "vt_pw_bwd_reduce");
    if (rc != VT_OK) return rc;
    VT_REQUIRE(coef && dy && lddy && sums, VT_ERR_INVALID, "vt_pw_bwd_reduce: null argument");
    a.coef = coef;
    for (int g = 0; g < d->ngroups; ++g) {
        if ((rc = check_rows("vt_pw_bwd_reduce", "dy", g, dy[g], lddy[g], d->C[g], false)) != VT_OK) return rc;
        VT_REQUIRE(sums[g], VT_ERR_INVALID, "vt_pw_bwd_reduce: null sums buffer");
        a.dy[g] = (const bf16_t*)dy[g];
        a.ldy[g] = lddy[g];
        a.stats[g] = sums[g];
    }
    return dispatch_pw<PW_REDUCE>(a.N0 + (d->ngroups == 2 ? d->C[1] : 0), d->K, a, (hipStream_t)stream, "vt_pw_bwd_reduce");
}

static int pw_bwd_apply_impl(const vt_pw_desc* d, const float* coef, const void* const* dy, const int32_t* lddy,
                             const float* const* bcoef, void* dx, int32_t lddx, const void* addend, int32_t ldadd,
                             float* const* dw, const int32_t* lddw, void* const* dz, const int32_t* lddz,
                             const vt_bn_fin_bwd* fin, void* stream) {
    PwArgs a;
    int rc = fill_common(a, d, "vt_pw_bwd_apply");
    if (rc != VT_OK) return rc;
    VT_REQUIRE(coef && dy && lddy && bcoef, VT_ERR_INVALID, "vt_pw_bwd_apply: null argument");
    const int N = a.N0 + (d->ngroups == 2 ? d->C[1] : 0);
    if (fin) {  // (checked by vt_pw_bwd_apply_finalize)
        a.fin = 1;
        for (int g = 0; g < d->ngroups; ++g) {
            const int off = g ? d->C[0] : 0;
            VT_REQUIRE(bcoef[g], VT_ERR_INVALID, "vt_pw_bwd_apply_finalize: null coefficients");
            a.bfin[g] = VtFinBwd{fin[g].sums, coef + off, coef + 2 * N + off, coef + 3 * N + off, fin[g].dgamma, fin[g].dbeta,
                                 (float*)bcoef[g], 1.0 / fin[g].count, fin[g].pscale, d->C[g], fin[g].train};
        }
    }
    const bool full = (int64_t)N * d->K <= 4096;
    a.coef = coef;
    VT_REQUIRE(dx && vt_aligned16(dx) && lddx % 8 == 0 && lddx >= d->K, VT_ERR_INVALID, "vt_pw_bwd_apply: bad dx");
    a.dx = (bf16_t*)dx;
    a.lddx = lddx;
    if (addend) {
        VT_REQUIRE(vt_aligned16(addend) && ldadd % 8 == 0 && ldadd >= d->K, VT_ERR_INVALID, "vt_pw_bwd_apply: bad addend");
        a.add = (const bf16_t*)addend;
        a.ldadd = ldadd;
    }
    for (int g = 0; g < d->ngroups; ++g) {
        if ((rc = check_rows("vt_pw_bwd_apply", "dy", g, dy[g], lddy[g], d->C[g], false)) != VT_OK) return rc;
        VT_REQUIRE(bcoef[g], VT_ERR_INVALID, "vt_pw_bwd_apply: null coefficients");
        a.dy[g] = (const bf16_t*)dy[g];
        a.ldy[g] = lddy[g];
        a.bcoef[g] = bcoef[g];
        if (dw && dw[g]) {
            VT_REQUIRE(full, VT_ERR_UNSUPPORTED,
                       "vt_pw_bwd_apply: %d x %d filter gradients do not fit the accumulators (vt_pw_supported() == 1): "
                       "pass dz and use vt_conv_wgrad", N, d->K);
            VT_REQUIRE(lddw && lddw[g] >= d->K, VT_ERR_INVALID, "vt_pw_bwd_apply: bad dw stride");
            a.dw[g] = dw[g];
            a.lddw[g] = lddw[g];
        }
        if (dz && dz[g]) {
            VT_REQUIRE(!full, VT_ERR_INVALID, "vt_pw_bwd_apply: dz is only produced for shapes with vt_pw_supported() == 1");
            VT_REQUIRE(lddz, VT_ERR_INVALID, "vt_pw_bwd_apply: dz without a stride");
            if ((rc = check_rows("vt_pw_bwd_apply", "dz", g, dz[g], lddz[g], d->C[g], false)) != VT_OK) return rc;
            a.dz[g] = (bf16_t*)dz[g];
            a.lddz[g] = lddz[g];
        }
    }
    return dispatch_pw<PW_BWD>(N, d->K, a, (hipStream_t)stream, "vt_pw_bwd_apply");
}

int vt_pw_bwd_apply(const vt_pw_desc* d, const float* coef, const void* const* dy, const int32_t* lddy,
                    const float* const* bcoef, void* dx, int32_t lddx, const void* addend, int32_t ldadd,
                    float* const* dw, const int32_t* lddw, void* const* dz, const int32_t* lddz, void* stream) {
    return pw_bwd_apply_impl(d, coef, dy, lddy, bcoef, dx, lddx, addend, ldadd, dw, lddw, dz, lddz, nullptr, stream);
}

int vt_pw_bwd_apply_finalize(const vt_pw_desc* d, const float* coef, const void* const* dy, const int32_t* lddy,
                             const vt_bn_fin_bwd* fin, float* const* bcoef, void* dx, int32_t lddx, const void* addend,
                             int32_t ldadd, float* const* dw, const int32_t* lddw, void* const* dz, const int32_t* lddz,
                             void* stream) {
    VT_REQUIRE(d && fin && coef && bcoef && (d->ngroups == 1 || d->ngroups == 2), VT_ERR_INVALID,
               "vt_pw_bwd_apply_finalize: null argument");
    const int N = d->C[0] + (d->ngroups == 2 ? d->C[1] : 0);
    for (int g = 0; g < d->ngroups; ++g)
        VT_REQUIRE(fin[g].sums && fin[g].count > 0 && bcoef[g], VT_ERR_INVALID, "vt_pw_bwd_apply_finalize: bad sums of group %d", g);
    if (N > 128 || !VT_KNOB("VT_BN_FIN_APPLY", 1)) {
        for (int g = 0; g < d->ngroups; ++g) {
            const int off = g ? d->C[0] : 0;
            const int rc = vt_bn_bwd_finalize(fin[g].sums, d->C[g], fin[g].count, fin[g].pscale, coef + off, coef + 2 * N + off,
                                              coef + 3 * N + off, fin[g].train, fin[g].dgamma, fin[g].dbeta, bcoef[g], stream);
            if (rc != VT_OK) return rc;
        }
        return pw_bwd_apply_impl(d, coef, dy, lddy, bcoef, dx, lddx, addend, ldadd, dw, lddw, dz, lddz, nullptr, stream);
    }
    return pw_bwd_apply_impl(d, coef, dy, lddy, bcoef, dx, lddx, addend, ldadd, dw, lddw, dz, lddz, fin, stream);
}

}  // extern "C"
