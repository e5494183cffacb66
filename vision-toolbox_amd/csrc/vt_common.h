// vt_common.h -- shared device/host helpers for libvt_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vt_amd.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- host side error plumbing ----------------------------------------------
void vt_set_error(const char* fmt, ...);
void vt_count_launch();
void vt_note_kernel(const char* fmt, ...);                            // name of the conv kernel a dispatch chose
int vt_raise_dynamic_lds(const void* kern, int bytes, const char* who);  // per (kernel, device), thread-safe
// Experiment / test switches of the dispatchers: a named integer, initialised ONCE per process from the environment
// variable of the same name and afterwards only changed through vt_set_knob (C-ABI; the GPU tests force a kernel with
// it).  A call site keeps the returned slot in a function-local static, so a launch reads one int, never the environment.
int* vt_knob_slot(const char* name, int dflt);
#define VT_KNOB(name, dflt) ([]() -> int { static int* slot__ = vt_knob_slot(name, dflt); return *slot__; }())
// multiprocessor (CU) count of the current device, cached per device
int vt_device_cus(void);

// BatchNorm statistics and the BatchNorm-backward sums are accumulated in FIXED POINT with integer atomics, which are
// associative: the sums -- and with them every activation, loss value and data gradient -- are bit-identical from run
// to run, whatever order the workgroups finish in.  (With f32 atomics the last bits of the statistics varied; a few
// activations then rounded to the neighbouring bf16 value or a ReLU decision at the threshold flipped, and BatchNorm
// over small maps amplified that into gradients that fell into discrete classes percents apart: DESIGN.md 5.)
// A value is split exactly into hi = trunc(v / 2^12) and lo = (v - hi * 2^12) * 2^33, each with its own int64:
// resolution 2^-34 absolute per contribution (a single 2^-20 grid made channels with variances ~1e-6 discontinuous),
// and the hi atomic is skipped when hi = 0, i.e. for every partial sum below 4096 -- one 64-bit atomic per value in
// practice.  |lo| < 2^45 per contribution: a replica takes 2^17 contributions of the largest size before it could
// overflow (the layers here make < 4,000).  A statistics buffer is int64[VT_STAT_REPLICAS][2][C][2]
// (VT_STAT_BYTES(C) bytes, zeroed by the caller); the kernels take it as float* and index the limbs themselves.
constexpr int kStatReplicas = VT_STAT_REPLICAS;
#ifdef __HIPCC__
// A non-finite (or absurdly large) contribution cannot be represented: it adds the marker 2^40 to the hi limb instead
// (no legitimate sum of a layer comes near 2^38 * 4096 = 1e15), and vt_stat_sum turns a marked sum into NaN -- a
// diverged activation shows up in the batch / running statistics as it did with float atomics, instead of as a
// finite garbage value.  (2^20 marked contributions per replica before the limb could wrap.)
constexpr long long kStatPoison = 1LL << 40;
__device__ __forceinline__ void vt_stat_add(float* stats, long idx, float v) {
    if (!(fabsf(v) < 1.0e30f)) {
        atomicAdd((unsigned long long*)stats + 2 * idx, (unsigned long long)kStatPoison);
        return;
    }
    const float hf = truncf(v * (1.0f / 4096.0f));   // exact (power-of-two scaling, then an integer)
    const float rem = v - hf * 4096.0f;               // exact: |rem| < 4096, a multiple of ulp(v)
    unsigned long long* q = (unsigned long long*)stats + 2 * idx;
    atomicAdd(q + 1, (unsigned long long)__float2ll_rn(rem * 8589934592.0f));  // 2^33
    if (hf != 0.0f) atomicAdd(q, (unsigned long long)(long long)hf);
}
// exact integer sums over the replicas of entry `idx` (stride = entries per replica), as a double in natural units
__device__ __forceinline__ double vt_stat_sum(const float* stats, long idx, long stride) {
    const long long* q = (const long long*)stats;
    long long hi = 0, lo = 0;
#pragma unroll
    for (int r = 0; r < kStatReplicas; ++r) {
        hi += q[2 * (idx + r * stride)];
        lo += q[2 * (idx + r * stride) + 1];
    }
    if (hi >= (kStatPoison >> 2) || hi <= -(kStatPoison >> 2)) return __longlong_as_double(0x7ff8000000000000LL);  // poisoned
    return (double)hi * 4096.0 + (double)lo * (1.0 / 8589934592.0);
}
#endif

#if defined(__HIPCC__)
// Workgroup -> work item so that neighbours in the ITEM order run on ONE XCD, at the same time.  The dispatcher deals the
// workgroups of a 1-D grid to the 8 XCDs round robin (workgroup L -> XCD L % 8), and each XCD has its own 4 MB L2: with
// the identity map, items that read the same operand rows are spread over all eight L2s and every L2 streams the whole
// operand (rocprofv3 FETCH_SIZE of the filter-gradient kernels in a CSPDarknet-53 step: 11.2 GB for ~3 GB of operands).
// Item = xcd * (n / 8) + L / 8: XCD x owns the items [x n/8, (x+1) n/8).  The grid holds n = a multiple of 8
// workgroups (vt_xcd_grid); items >= `items` do not exist (the caller returns).  XCDS = 1 turns the map off.
__device__ __forceinline__ unsigned vt_xcd_item(unsigned L, unsigned n, int xcds) {
    if (xcds <= 1) return L;
    return (L % 8u) * (n / 8u) + L / 8u;
}
#endif
static inline unsigned vt_xcd_grid(long items) { return (unsigned)((items + 7) / 8 * 8); }

#if defined(__HIPCC__)
// A FORK right behind a kernel (the filter gradient waits for dz) used to be an event RECORD on the main stream: a marker
// packet between two kernels of the critical path, 66 per step.  The executor instead hands the producing launch an event
// through this slot, and the launch attaches it to its own completion signal (hipExtLaunchKernelGGL's stop event): the side
// stream waits for the kernel itself, the main queue carries no extra packet.  A launch site that supports it uses
// VT_LAUNCH_STOP; the executor falls back to the record when the slot was not consumed.
extern thread_local hipEvent_t vt_pending_stop_event;
#define VT_LAUNCH_STOP(kernel, grid, block, smem, stream, ...)                                                    \
    do {                                                                                                          \
        if (vt_pending_stop_event) {                                                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, nullptr, vt_pending_stop_event, 0, __VA_ARGS__); \
            vt_pending_stop_event = nullptr;                                                                      \
        } else {                                                                                                  \
            hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                   \
        }                                                                                                         \
    } while (0)
#endif

#define VT_REQUIRE(cond, code, ...)   \
    do {                              \
        if (!(cond)) {                \
            vt_set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

#define VT_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            vt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return VT_ERR_HIP;                                                    \
        }                                                                         \
        vt_count_launch();                                                        \
    } while (0)

static inline bool vt_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int vt_elem_size(int dtype) { return dtype == VT_BF16 ? 2 : 4; }
static inline int vt_epc(int dtype) { return dtype == VT_BF16 ? 8 : 4; }  // elements per 16 B

// ---- device helpers ----------------------------------------------------------
template <typename T>
struct VecIO;  // 16-byte chunk <-> float[EPC]

template <>
struct VecIO<float> {
    static constexpr int EPC = 4;
    __device__ static inline void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x);
        f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z);
        f[3] = __uint_as_float(v.w);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                          __float_as_uint(f[3]));
    }
    __device__ static inline float round(float v) { return v; }
};

__device__ static inline float bf16_bits_to_float(uint32_t lo16) { return __uint_as_float(lo16 << 16); }

template <>
struct VecIO<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static inline void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16);
        f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16);
        f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16);
        f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16);
        f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    __device__ static inline uint32_t pack2(float a, float b) {
        // plain casts lower to v_cvt_pk_bf16_f32 (RNE, NaN preserving) on gfx950
        bf16_t x = (bf16_t)a, y = (bf16_t)b;
        return (uint32_t)__builtin_bit_cast(unsigned short, x) |
               ((uint32_t)__builtin_bit_cast(unsigned short, y) << 16);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    __device__ static inline float round(float v) { return (float)(bf16_t)v; }
};

template <typename T>
__device__ static inline T from_float(float v);
template <>
__device__ inline float from_float<float>(float v) { return v; }
template <>
__device__ inline bf16_t from_float<bf16_t>(float v) { return (bf16_t)v; }

__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
