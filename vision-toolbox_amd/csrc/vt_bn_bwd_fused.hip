// vt_bn_bwd_fused.hip -- BatchNorm2d (+ ReLU) backward of one ConvNormAct unit in ONE launch (round 6).
//
// The autograd backward of nn.BatchNorm2d + nn.ReLU (reference components.py:36-44) is two passes over (d(y), z) with a
// per-channel reduction between them: sums (sum g, sum g * xhat) -> coefficients -> dz = a * g - b * z + d.  As three launches
// (vt_bn_act_bwd_reduce, vt_bn_bwd_finalize, vt_bn_act_bwd_apply) that is d(y) and z read TWICE from HBM plus a
// single-workgroup launch on the critical path, 57 times per CSPDarknet-53 step (4.75 ms beside the filter-gradient stream).
// The operands of the layers from 14 x 14 down fit the chip's REGISTER FILE with room to work in (256 CUs x 512 KiB = 128 MiB;
// 256 channels @14x14 at batch 256: 2 x 25.7 MB = 13 rows of 16 bytes per thread and tensor, 245 registers; the 2 x 51 MB of
// 128 channels @28x28 would be 25 rows -- the compiler's code for that spills 349 registers and is not offered), so here:
//   * at most one workgroup per CU (512 threads, up to 256 registers each), every thread loads its NV rows of d(y) and z ONCE,
//     16 bytes each, and keeps them packed;
//   * pass 1 forms the thread's partial sums, folds them over the workgroup (registers, then LDS) and adds them to the
//     unit's fixed-point sums with integer atomics -- the arithmetic of bn_bwd_reduce_kernel;
//   * a GRID BARRIER (device-scope counter; every workgroup is resident: the grid never exceeds the CU count; a wall-clock
//     bound turns a barrier that does not complete into an error flag instead of a hang);
//   * the workgroups finalize 32 channels each (blocks of channels dealt round robin) -- 16 lanes per channel read the 16 replicas with
//     device-scope loads, exact integer sums, the arithmetic of bn_bwd_finalize_kernel -- and publish the coefficients;
//   * a second grid barrier, then pass 2 from the registers: one store per 16 bytes, no second read.
// Algorithmic traffic 3 tensors instead of 5; one launch instead of three.  Results: the sums are the same f32 terms in
// another order (fixed point from the workgroup level up: run-to-run bit-identical), dz follows from them.
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "vt_common.h"

namespace {

constexpr int kT = 512;

struct FArgs {
    const bf16_t* dy;
    const bf16_t* z;
    bf16_t* dz;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* invstd;
    float* sums;      // statistics buffer (VT_STAT_REPLICAS), zeroed by the caller
    unsigned* sync;   // [0], [1]: the two barrier counters, [2]: error flag; zeroed by the caller
    float* dgamma;
    float* dbeta;
    float* coef;      // [3][C]
    long M;
    double inv_count, pscale;
    int lddy, ldz, lddz, C, relu, train;
    int CPR, CT, RT;  // 16-byte chunks per row, threads along a row, rows per workgroup pass
    unsigned nwg;
};

__device__ __forceinline__ uint4 ldg16(const bf16_t* p) { return *(const uint4*)p; }

// barriers that ran into their wall-clock bound since the library was loaded (vt_bn_bwd_fused_timeouts): must stay 0
__device__ unsigned vt_bn_fused_timeouts;

// A value another workgroup (another XCD: another L2) has written in THIS launch: a device-scope load behind the
// barrier's acquire fence.  (First version: read-modify-writes with 0, which are performed where the atomics are whatever
// a cache holds -- correct, and 100 us per launch: 242 workgroups each polling one counter and fetching 768 coefficients
// serialise on the addresses they share.)
__device__ __forceinline__ unsigned long long coherent_u64(const void* p) {
    return __hip_atomic_load((unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float coherent_f32(const float* p) {
    return __hip_atomic_load((float*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every workgroup of the grid arrives, then leaves.  No fences: everything one workgroup reads of another's inside this
// launch is written with device-scope atomics or sc1 stores and read with device-scope (sc1) loads, so all a thread owes
// the barrier is that its own requests have been performed (s_waitcnt).  (First versions: a __threadfence() per thread and
// acquire polls -- an L2 write-back / invalidate per thread and per poll: 100 us per launch.)  The arrival is a relaxed
// increment; the poll a relaxed device-scope load, every 32nd one a read-modify-write with 0 so that a stale line cannot hold
// a workgroup back; a wall-clock bound (~4 ms) turns a barrier that does not complete into an error flag and a counted
// timeout instead of a hang.
__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned n, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        unsigned polls = 0;
        while (((++polls & 31u) ? __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : __hip_atomic_fetch_add(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < n) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 400000ull) {
                __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&vt_bn_fused_timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

template <int NV>
__global__ void __launch_bounds__(kT) bn_bwd_fused_kernel(const FArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sred[];
    constexpr int EPC = 8;
    const int t = threadIdx.x;
    const int r = t / a.CT, tc = t % a.CT;
    const bool active = r < a.RT && tc < a.CPR;
    const int W = a.CT * EPC;  // channels per pass (CT = CPR: one pass)
    const long row0 = (long)blockIdx.x * a.RT * NV + r;
    const int rep = blockIdx.x % kStatReplicas;
    const int c0 = tc * EPC;

    float sc[EPC], sf[EPC], mu[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) sc[e] = 1.f, sf[e] = 0.f, mu[e] = 0.f;
    if (active) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) sc[e] = a.scale[c0 + e], sf[e] = a.shift[c0 + e], mu[e] = a.mean[c0 + e];
    }
    // ---- the thread's rows, once -------------------------------------------------------------------------------
    uint4 vg[NV], vz[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const long row = row0 + (long)i * a.RT;
        // (unconditional loads -- rows / columns outside the tensor read its first bytes -- so that all of them are in flight
        //  together; a load inside a branch waits for its own round trip, row by row)
        const bool ok = active && row < a.M;
        vg[i] = ldg16(a.dy + (ok ? row * a.lddy + c0 : 0l));
        vz[i] = ldg16(a.z + (ok ? row * a.ldz + c0 : 0l));
        if (!ok) vg[i] = make_uint4(0, 0, 0, 0);  // zero d(y) contributes nothing
    }
    // ---- pass 1: partial sums (bn_bwd_reduce_kernel's arithmetic) ------------------------------------
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float g[EPC], zz[EPC];
        VecIO<bf16_t>::unpack(vg[i], g);
        VecIO<bf16_t>::unpack(vz[i], zz);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float gg = (!a.relu || fmaf(zz[e], sc[e], sf[e]) > 0.f) ? g[e] : 0.f;
            s1[e] += gg;
            s2[e] = fmaf(gg, zz[e] - mu[e], s2[e]);  // invstd applied once, below
        }
    }
    {
        // fold of the row lanes: in-wave shuffles where CT is a power of two below 64, then LDS
        const bool inwave = a.CT < 64 && (a.CT & (a.CT - 1)) == 0;
        int rows_l = a.RT, r_l = active ? r : -1;
        if (inwave) {
            for (int off = a.CT; off < 64; off <<= 1) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    s1[e] += __shfl_xor(s1[e], off, 64);
                    s2[e] += __shfl_xor(s2[e], off, 64);
                }
            }
            rows_l = kT / 64;
            r_l = (t & 63) < a.CT ? (t >> 6) : -1;  // one writer per (wave, column)
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
        for (int i = t; i < 2 * W; i += kT) {
            const int which = i / W, c = i % W;
            if (c < a.C) {
                float acc = 0.f;
                for (int rr = 0; rr < rows_l; ++rr) acc += sred[(long)(rr * 2 + which) * W + c];
                if (which) acc *= a.invstd[c];
                vt_stat_add(a.sums, ((long)rep * 2 + which) * a.C + c, acc);
            }
        }
    }
    // (opaque: the rows stay PACKED across the barriers -- left alone the compiler keeps the unpacked f32 copies of pass 1
    //  alive for pass 2: 222 registers at NV = 7, 537 spilled at NV = 25)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        asm volatile("" : "+v"(vg[i].x), "+v"(vg[i].y), "+v"(vg[i].z), "+v"(vg[i].w));
        asm volatile("" : "+v"(vz[i].x), "+v"(vz[i].y), "+v"(vz[i].z), "+v"(vz[i].w));
    }
    grid_barrier(a.sync + 0, a.nwg, a.sync + 2);
    // ---- finalize: workgroup w, 32 channels, 16 lanes per channel (one per replica) -----------------------
    for (int cb = blockIdx.x; cb * 32 < a.C; cb += (int)a.nwg) {
        const int c = cb * 32 + (t >> 4), rp = t & 15;
        if (c < a.C) {  // (whole 16-lane groups take the branch together)
            const long long* q = (const long long*)a.sums;
            long long hi[2], lo[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const long idx = (long)w * a.C + c + (long)rp * 2 * a.C;
                hi[w] = (long long)coherent_u64(q + 2 * idx);
                lo[w] = (long long)coherent_u64(q + 2 * idx + 1);
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    hi[w] += __shfl_xor(hi[w], off, 64);
                    lo[w] += __shfl_xor(lo[w], off, 64);
                }
            }
            if (rp == 0) {
                auto nat = [](long long h, long long l) -> double {
                    if (h >= (kStatPoison >> 2) || h <= -(kStatPoison >> 2)) return __longlong_as_double(0x7ff8000000000000LL);
                    return (double)h * 4096.0 + (double)l * (1.0 / 8589934592.0);
                };
                const double S1 = nat(hi[0], lo[0]), S2 = nat(hi[1], lo[1]);
                const float av = a.scale[c], mv = a.mean[c], istd = a.invstd[c];
                if (a.dgamma) a.dgamma[c] += (float)(S2 * a.pscale);
                if (a.dbeta) a.dbeta[c] += (float)(S1 * a.pscale);
                float b = 0.f, d = 0.f;
                if (a.train) {
                    const double k1 = S1 * a.inv_count, k2 = S2 * a.inv_count;
                    const double bb = (double)av * k2 * (double)istd;
                    b = (float)bb;
                    d = (float)(bb * (double)mv - (double)av * k1);
                }
                __hip_atomic_store(a.coef + c, av, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.coef + a.C + c, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.coef + 2 * a.C + c, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    grid_barrier(a.sync + 1, a.nwg, a.sync + 2);
    // ---- pass 2 from the registers ------------------------------------------------------------------------
    if (!active) return;
    // (the coefficients were written behind other XCDs' L2s: device-scope loads behind the barrier's acquire fence)
    float ca[EPC], cb[EPC], cd[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        ca[e] = coherent_f32(a.coef + c0 + e);
        cb[e] = coherent_f32(a.coef + a.C + c0 + e);
        cd[e] = coherent_f32(a.coef + 2 * a.C + c0 + e);
    }
    long rowp = row0;
    asm volatile("" : "+v"(rowp));  // (the store addresses are formed here, not kept from the loads)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const long row = rowp + (long)i * a.RT;
        if (row >= a.M) continue;
        float g[EPC], zz[EPC];
        VecIO<bf16_t>::unpack(vg[i], g);
        VecIO<bf16_t>::unpack(vz[i], zz);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float gg = (!a.relu || fmaf(zz[e], sc[e], sf[e]) > 0.f) ? g[e] : 0.f;
            g[e] = fmaf(ca[e], gg, fmaf(-cb[e], zz[e], cd[e]));
        }
        *(uint4*)(a.dz + row * a.lddz + c0) = VecIO<bf16_t>::pack(g);
    }
}

template <int NV>
int launch_fused(const FArgs& a, int smem, hipStream_t st) {
    if (smem > 48 * 1024) {
        const int rc = vt_raise_dynamic_lds((const void*)bn_bwd_fused_kernel<NV>, smem, "vt_bn_act_bwd_fused");
        if (rc != VT_OK) return rc;
    }
    VT_LAUNCH_STOP(bn_bwd_fused_kernel<NV>, dim3(a.nwg), dim3(kT), smem, st, a);
    VT_CHECK_LAUNCH("vt_bn_act_bwd_fused");
    return VT_OK;
}

}  // namespace

// One launch where the operands fit the register file (see the header of this file), else the three launches it
// replaces.  `sync`: 16 zeroed bytes (two barrier counters and an error flag that a later vt_bn_bwd_fused_check reads).
extern "C" int vt_bn_act_bwd_fused(const void* dy, int32_t lddy, const void* z, int32_t ldz, const float* scale,
                                   const float* shift, const float* mean, const float* invstd, int64_t M, int32_t C,
                                   int32_t relu, int32_t dtype, double count, double pscale, int32_t train, float* sums,
                                   void* sync, float* dgamma, float* dbeta, float* coef, void* dz, int32_t lddz,
                                   void* stream) {
    VT_REQUIRE(dy && z && scale && shift && mean && invstd && sums && sync && coef && dz && M > 0 && C > 0 && count > 0,
               VT_ERR_INVALID, "vt_bn_act_bwd_fused: bad argument");
    VT_REQUIRE(relu >= 0 && relu <= 4, VT_ERR_INVALID, "vt_bn_act_bwd_fused: activation code %d", relu);
    const int cus = vt_device_cus();
    const bool fits_types = dtype == VT_BF16 && relu <= 1 && C % 8 == 0 && lddy % 8 == 0 && ldz % 8 == 0 && lddz % 8 == 0 &&
                            vt_aligned16(dy) && vt_aligned16(z) && vt_aligned16(dz) && lddy >= C && ldz >= C && lddz >= C;
    if (fits_types && cus >= 64 && VT_KNOB("VT_BN_BWD_FUSED", 1)) {
        FArgs a;
        memset(&a, 0, sizeof(a));
        a.CPR = C / 8;
        if (a.CPR <= kT) {
            a.CT = a.CPR;
            a.RT = kT / a.CT;
            // rows per thread so that the grid fits the CUs (one workgroup each: every workgroup resident, the barrier safe)
            const long per_pass = (long)a.RT * cus;
            const int nv = (int)((M + per_pass - 1) / per_pass);
            const int NVs[] = {4, 7, 10, 13};  // (19 / 25 rows spill: 128 channels @28x28 at batch 256 keeps the three launches)
            int pick = 0;
            for (int v : NVs)
                if (!pick && v >= nv) pick = v;
            const unsigned nwg = pick ? (unsigned)((M + (long)a.RT * pick - 1) / ((long)a.RT * pick)) : 0;
            if (pick && nwg >= 1 && nwg <= (unsigned)cus) {
                a.dy = (const bf16_t*)dy, a.z = (const bf16_t*)z, a.dz = (bf16_t*)dz;
                a.scale = scale, a.shift = shift, a.mean = mean, a.invstd = invstd;
                a.sums = sums, a.sync = (unsigned*)sync, a.dgamma = dgamma, a.dbeta = dbeta, a.coef = coef;
                a.M = M, a.inv_count = 1.0 / count, a.pscale = pscale;
                a.lddy = lddy, a.ldz = ldz, a.lddz = lddz, a.C = C, a.relu = relu, a.train = train;
                a.nwg = nwg;
                const bool inwave = a.CT < 64 && (a.CT & (a.CT - 1)) == 0;
                const int smem = (inwave ? kT / 64 : a.RT) * 2 * a.CT * 8 * (int)sizeof(float);
                vt_note_kernel("bn_bwd_fused_kernel<NV%d>", pick);
                switch (pick) {
                    case 4: return launch_fused<4>(a, smem, (hipStream_t)stream);
                    case 7: return launch_fused<7>(a, smem, (hipStream_t)stream);
                    case 10: return launch_fused<10>(a, smem, (hipStream_t)stream);
                    default: return launch_fused<13>(a, smem, (hipStream_t)stream);
                }
            }
        }
    }
    int rc = vt_bn_act_bwd_reduce(dy, lddy, z, ldz, scale, shift, mean, invstd, M, C, relu, dtype, sums, stream);
    if (rc != VT_OK) return rc;
    rc = vt_bn_bwd_finalize(sums, C, count, pscale, scale, mean, invstd, train, dgamma, dbeta, coef, stream);
    if (rc != VT_OK) return rc;
    return vt_bn_act_bwd_apply(dy, lddy, z, ldz, scale, shift, coef, dz, lddz, M, C, relu, dtype, stream);
}

extern "C" unsigned vt_fin_timeouts_host();  // vt_elementwise.hip: the hand-offs of vt_bn_finalize_apply / vt_bn_bwd_finalize_apply

// barriers of vt_bn_act_bwd_fused launches (and hand-offs of vt_bn_finalize_apply / vt_bn_bwd_finalize_apply) that gave up waiting (synchronises the device); anything but 0 means wrong results
extern "C" int vt_bn_bwd_fused_timeouts(uint32_t* count) {
    VT_REQUIRE(count, VT_ERR_INVALID, "vt_bn_bwd_fused_timeouts: null");
    unsigned v = 0;
    const hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(vt_bn_fused_timeouts), sizeof(v));
    if (e != hipSuccess) {
        vt_set_error("vt_bn_bwd_fused_timeouts: %s", hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    *count = v + vt_fin_timeouts_host();
    return VT_OK;
}
