// vt_bn_fin.h -- the BatchNorm finalize arithmetic, shared by the stand-alone finalize kernels and the streaming passes
// that finalize for themselves (round 6).
//
// A training step of CSPDarknet-53 held 134 single-purpose finalize launches (vt_bn_finalize after the convolution that
// accumulates the batch statistics, vt_bn_bwd_finalize after the backward reduction).  Their work is nothing, but each sat
// between a producer that must drain and a consumer that cannot start: skipping them (diagnostic build, results invalid)
// shortens the 19.8 ms step by 1.17 ms, 9 - 13 us per launch.  What replaced them (vt_bn_finalize_apply,
// vt_bn_bwd_finalize_apply, vt_pw_fwd_apply_finalize, vt_pw_bwd_apply_finalize): the sums are complete when the consuming
// pass starts, so EVERY workgroup of that pass finalizes the channels it needs for itself -- nothing is handed over inside a
// launch.  The two forms with a hand-off measured no gain: the first workgroups finalize and publish while the others poll
// (+2.3 ms per step, NOTEBOOK R6.6); the producer's last workgroup finalizes behind a ticket (+-0.00 ms, and its code at the
// end of the MFMA kernels cost 0.09 ms even unused: NOTEBOOK R6.10, removed) -- a chain of dependent memory-side round trips
// costs inside a launch what it costs at a launch boundary.
//
//   * the arithmetic is vt_fin_fwd_channel / vt_fin_bwd_channel: the sums are exact integers (vt_common.h), so every
//     workgroup computes bit-identical coefficients, equal to the stand-alone kernel's;
//   * what a finalize costs is dependent round trips, not bytes: a thread per (channel, sum) loads its 16 replicas x 2 limbs
//     in ONE round (vt_replica_sum), the plain operands of the arithmetic beside them (vt_pair_sums evaluates `g` first).
#pragma once
#include "vt_common.h"

struct VtFinFwd {  // the arguments of vt_bn_finalize
    const float* stats;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    int64_t* nbt;
    float* scale;
    float* shift;
    float* mean;
    float* invstd;
    double inv_count, unbias;
    float eps, momentum;
    int C;
};
struct VtFinBwd {  // the arguments of vt_bn_bwd_finalize
    const float* sums;
    const float* scale;
    const float* mean;
    const float* invstd;
    float* dgamma;
    float* dbeta;
    float* coef;
    double inv_count, pscale;
    int C, train;
};

#ifdef __HIPCC__
// ---- the finalize arithmetic of one channel (bn_finalize_kernel / bn_bwd_finalize_kernel call these too) ----------------
// One single-block launch per BatchNorm, or a prologue of every workgroup of a pass: what it costs is latency, not work.  The divisions by `count` are
// multiplications by a host-computed reciprocal, and 1/sqrt runs in f32 on the double-precision variance (correctly
// rounded sqrt and division: within one ulp of the double evaluation) -- fp64 division and sqrt are long software
// sequences on this hardware.
// g, b, rm, rv: gamma (or 1), beta (or 0), the running statistics (if any) of channel c, loaded by the caller (beside the
// sums, not behind them).  sc, sf: the normalisation's scale and shift; write = false computes them and stores nothing.
__device__ __forceinline__ void vt_fin_fwd_channel(const VtFinFwd& f, int c, double s, double ss, float g, float b, float rm,
                                                   float rv, float& sc, float& sf, bool write = true) {
    const double mu = s * f.inv_count;
    double var = ss * f.inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float istd = 1.0f / sqrtf((float)(var + (double)f.eps));
    sc = g * istd;
    sf = b - (float)mu * sc;
    if (!write) return;
    f.scale[c] = sc;
    f.shift[c] = sf;
    f.mean[c] = (float)mu;
    f.invstd[c] = istd;
    if (f.running_mean) {
        f.running_mean[c] = (1.f - f.momentum) * rm + f.momentum * (float)mu;
        f.running_var[c] = (1.f - f.momentum) * rv + f.momentum * (float)(var * f.unbias);
    }
    if (c == 0 && f.nbt) f.nbt[0] += 1;
}
// a, mu, istd: scale, mean, invstd of channel c; dg, db: d(gamma), d(beta) so far.  b, d: with a the coefficients of
// dz = a*g - b*z + d; write = false computes them and stores nothing.
__device__ __forceinline__ void vt_fin_bwd_channel(const VtFinBwd& f, int c, double s1, double s2, float a, float mu, float istd,
                                                   float dg, float db, float& b, float& d, bool write = true) {
    b = 0.f, d = 0.f;
    if (f.train) {
        const double c1 = s1 * f.inv_count, c2 = s2 * f.inv_count;
        const double bb = (double)a * c2 * (double)istd;
        b = (float)bb;
        d = (float)(bb * (double)mu - (double)a * c1);
    }
    if (!write) return;
    if (f.dgamma) f.dgamma[c] = dg + (float)(s2 * f.pscale);
    if (f.dbeta) f.dbeta[c] = db + (float)(s1 * f.pscale);
    f.coef[c] = a;
    f.coef[f.C + c] = b;
    f.coef[2 * f.C + c] = d;
}

// Exact integer sum over the 16 replicas of statistics entry `entry` (w * C + c: sum (w = 0) or second sum (w = 1) of channel
// c), `stride` = 2 * C entries per replica, as vt_stat_sum returns it; the 16 x 2 limbs are loaded in ONE round (32
// eight-byte loads in flight).  kCoherent: device-scope loads, for sums produced by other workgroups of the SAME launch
// (no caller left: the ticket experiment); else plain loads of an earlier launch's sums.
template <bool kCoherent>
__device__ __forceinline__ double vt_replica_sum(const float* stats, long entry, long stride) {
    const unsigned long long* q = (const unsigned long long*)stats;
    unsigned long long h[kStatReplicas], l[kStatReplicas];
#pragma unroll
    for (int r = 0; r < kStatReplicas; ++r) {
        const long idx = entry + (long)r * stride;
        if (kCoherent) {
            h[r] = __hip_atomic_load(q + 2 * idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l[r] = __hip_atomic_load(q + 2 * idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            h[r] = q[2 * idx], l[r] = q[2 * idx + 1];
        }
    }
    long long hi = 0, lo = 0;
#pragma unroll
    for (int r = 0; r < kStatReplicas; ++r) hi += (long long)h[r], lo += (long long)l[r];
    if (hi >= (kStatPoison >> 2) || hi <= -(kStatPoison >> 2)) return __longlong_as_double(0x7ff8000000000000LL);  // poisoned
    return (double)hi * 4096.0 + (double)lo * (1.0 / 8589934592.0);
}

// Thread t owns entry (w = t & 1, c = c_begin + t / 2); the lane pair swaps its sums and f(c, s0, s1, pre) runs in the w = 0
// lane.  pre = g(c) is evaluated BEFORE the sums are waited for: the plain operand loads of the finalize arithmetic travel
// beside the replica loads instead of behind them.  At most blockDim.x / 2 channels per call.
template <bool kCoherent, typename G, typename F>
__device__ __forceinline__ void vt_pair_sums(const float* stats, int C, int c_begin, int c_end, G&& g, F&& f) {
    const int t = threadIdx.x, w = t & 1, c = c_begin + (t >> 1);
    const bool on = c < c_end;  // (lane pairs take the branch together)
    const int cc = on ? c : c_begin;
    auto pre = g(cc);
    const double v = vt_replica_sum<kCoherent>(stats, (long)w * C + cc, 2L * C);
    const double other = __shfl_xor(v, 1, 64);
    if (on && w == 0) f(c, v, other, pre);
}

// the plain operands of the finalize arithmetic of channel c
struct VtFinFwdPre { float g, b, rm, rv; };
struct VtFinBwdPre { float a, mu, istd, dg, db; };
__device__ __forceinline__ VtFinFwdPre vt_fin_fwd_pre(const VtFinFwd& f, int c) {
    VtFinFwdPre p{f.gamma ? f.gamma[c] : 1.f, f.beta ? f.beta[c] : 0.f, 0.f, 0.f};
    if (f.running_mean) p.rm = f.running_mean[c], p.rv = f.running_var[c];
    return p;
}
__device__ __forceinline__ VtFinBwdPre vt_fin_bwd_pre(const VtFinBwd& f, int c) {
    VtFinBwdPre p{f.scale[c], f.mean[c], f.invstd[c], 0.f, 0.f};
    if (f.dgamma) p.dg = f.dgamma[c];
    if (f.dbeta) p.db = f.dbeta[c];
    return p;
}
#endif
