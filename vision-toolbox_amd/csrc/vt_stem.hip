// vt_stem.hip -- the 3x3 stride-1 stem convolution of the Darknet family (bf16):
// 3 input channels (padded to one 16-byte pixel) -> 32/64 output channels at full resolution
// (reference darknet.py:75, `ConvNormAct(3, 32, 3, 1)`; 224x224 x batch 256 = 12.8 M pixels).
//
// The layer is HBM-bound (205 MB in, 822 MB out, 59 GFLOP) and K = 9 taps x 8 channels = 72
// does not fill the 64-byte K rows of the general kernel, which spent 0.77 ms on it (76 TFLOP/s,
// 4x its HBM time).  Here the K dimension is re-cut along the memory layout instead: one filter
// ROW (3 taps) of one output pixel is 4 consecutive input pixels x 8 channels = 64 contiguous
// bytes (the 4th pixel pairs with a zero block), so
//   * the tile's input span (256 + 2W + 2 pixels x 16 B, flat pixel index as in vt_igemm_span.hip)
//     is DMA'd into LDS once, linearly, no swizzle: lane (row r, k-quarter q) of an A fragment
//     reads the 16 bytes of span pixel r + e*W + q, consecutive lanes -> consecutive 16-byte
//     slots, conflict free;
//   * K = 3 steps of 32 (one per filter row e), 24 MFMAs per wave and tile;
//   * padding (and the 4th pixel) = per-lane select of a 16-byte zero block from a per-row mask.
// Epilogue as vt_igemm_span.hip (per-wave slabs, BN statistics or affine+ReLU).
#include <stdlib.h>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int vt_stem_zero16[4];

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

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kBM = 256;
constexpr int kMaskOff = 0;             // 256 x 4 B
constexpr int kZeroOff = 1024;          // 16 B (64 reserved)
constexpr int kWOff = 1088;             // BN x 144 B filter image [n][tap][8]

template <int BN>
struct StemLds {
    static constexpr int kStage = kWOff + BN * 144;                 // 4 waves x 16 rows x (BN+8) x 2 B
    static constexpr int kSpan = kStage + 4 * 16 * (BN + 8) * 2;
    __host__ __device__ static constexpr int bytes(int span_instr) { return kSpan + span_instr * 1024; }
};

template <int BN>
__global__ void __launch_bounds__(256) stem_kernel(const IgemmArgs p, const int span_instr) {
    using bf = bf16_t;
    constexpr int FM = 4, FN = BN / 16;
    using L = StemLds<BN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* sMask = (unsigned*)(smem + kMaskOff);
    const uint4* sZ = (const uint4*)(smem + kZeroOff);
    char* sW = smem + kWOff;
    char* sSpan = smem + L::kSpan;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * kBM;
    const int W = p.Wi, H = p.Hi, HW = H * W;

    // ---- stage: input span by LDS-DMA, filter image and row masks by ordinary loads --------------
    const unsigned span_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sSpan;
    const unsigned long zero_src = (unsigned long)(const void*)vt_stem_zero16;
    const bf* __restrict__ xg = (const bf*)p.x;
    for (int i = wave; i < span_instr; i += 4) {
        const long pix = m0 - W - 1 + 64l * i + lane;
        const unsigned long src = (pix >= 0 && pix < p.M) ? (unsigned long)(xg + pix * 8) : zero_src;
        glds16(src, span_base + (unsigned)i * 1024u);
    }
    if (tid < 4) ((unsigned*)(smem + kZeroOff))[tid] = 0u;
    {
        const uint4* wsrc = (const uint4*)p.w;  // [Cout][9][8] bf16 = 9 chunks per output channel
        for (int c = tid; c < BN * 9; c += 256) {
            const int n = c / 9;
            ((uint4*)sW)[c] = n < p.Cout ? wsrc[c] : make_uint4(0, 0, 0, 0);
        }
    }
    {
        const int r = tid;  // 256 threads = 256 tile rows
        const long m = m0 + r;
        unsigned bits = 0;
        if (m < p.M) {
            const int rem = (int)(m % HW);
            const int oi = rem / W, oj = rem - oi * W;
#pragma unroll
            for (int e = 0; e < 3; ++e)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if ((unsigned)(oi + e - 1) < (unsigned)H && (unsigned)(oj + q - 1) < (unsigned)W) bits |= 1u << (e * 4 + q);
        }
        sMask[r] = bits;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- 3 K-steps: one filter row each ---------------------------------------------------------
    const int q = lane >> 4, u = lane & 15;
    unsigned mk[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) mk[i] = sMask[wave * 64 + i * 16 + u] >> q;  // bit 4e = (row e, this lane's pixel)
    const char* abase = sSpan + (wave * 64 + u + q) * 16;
    // filter fragment of rows n = 16j + u, filter row e: chunk (e*3 + q) of the row's 9; q == 3 -> zeros
    const char* bbase = sW + u * 144 + q * 16;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int e = 0; e < 3; ++e) {
        uint4 af[FM], bfr[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const uint4* src = q < 3 ? (const uint4*)(bbase + j * 16 * 144 + e * 48) : sZ;
            bfr[j] = *src;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const uint4* src = ((mk[i] >> (4 * e)) & 1u) ? (const uint4*)(abase + (i * 16 + e * W) * 16) : sZ;
            af[i] = *src;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                    __builtin_bit_cast(bf16x8, bfr[j]), acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: per wave, 16-row slabs through a private LDS window ----------------------------
    constexpr int PITCH = BN + 8;
    constexpr int CPRW = BN / 8;       // 16-byte chunks per row
    constexpr int RPP = 64 / CPRW;     // rows per read pass (16 for BN=32, 8 for BN=64)
    constexpr int NPASS = (16 + RPP - 1) / RPP;
    bf* sWin = (bf*)(smem + L::kStage) + wave * 16 * PITCH;
    const bool affine = p.flags & VT_CONV_AFFINE;
    const bool relu = p.flags & VT_CONV_RELU;
    const bool stats = p.flags & VT_CONV_STATS;
    const bool store = !(p.flags & VT_CONV_NOSTORE);  // statistics-only pass: sums of the f32 accumulator, nothing kept
    bf* __restrict__ yg = (bf*)p.y;
    float sc[FN], sf[FN], s1[FN], s2[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = j * 16 + u;
        sc[j] = 1.f, sf[j] = 0.f, s1[j] = 0.f, s2[j] = 0.f;
        if (affine && n < p.Cout) {
            if (p.scale) sc[j] = p.scale[n];
            sf[j] = p.shift[n];
        }
    }
    const int rrow = lane / CPRW, rch = lane % CPRW;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r];
                if (affine) v = fmaf(v, sc[j], sf[j]);
                if (relu) v = fmaxf(v, 0.f);
                const bf tv = from_float<bf>(v);
                if (store) sWin[(4 * q + r) * PITCH + j * 16 + u] = tv;
                const float fv = store ? (float)tv : v;
                s1[j] += fv;
                s2[j] = fmaf(fv, fv, s2[j]);
            }
        }
        if (!store) continue;
        lds_fence();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int lr = ps * RPP + rrow;
            if (lr < 16) {
                const uint4 v = *(const uint4*)(sWin + lr * PITCH + rch * 8);
                const long m = m0 + wave * 64 + i * 16 + lr;
                if (m < p.M && rch * 8 < p.Cout) *(uint4*)(yg + (m * p.ldy + rch * 8)) = v;
            }
        }
        lds_fence();
    }
    if (stats) {
        const int rep = (int)(blockIdx.x % kStatReplicas);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            float a = s1[j], b = s2[j];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            const int n = j * 16 + u;
            if (q == 0 && n < p.Cout) {
                vt_stat_add(p.stats, ((long)rep * 2 + 0) * p.Cout + n, a);
                vt_stat_add(p.stats, ((long)rep * 2 + 1) * p.Cout + n, b);
            }
        }
    }
}

template <int BN>
int launch_stem(const IgemmArgs& a, hipStream_t st) {
    const int span_px = kBM + 2 * a.Wi + 2;
    const int span_instr = (span_px + 63) / 64;
    const int smem = StemLds<BN>::bytes(span_instr);
    if (smem > 64 * 1024) return -1;
    const long blocks = ((long)a.M + kBM - 1) / kBM;
    vt_note_kernel("stem_kernel<%d>", BN);
    hipLaunchKernelGGL(stem_kernel<BN>, dim3((unsigned)blocks), dim3(256), smem, st, a, span_instr);
    VT_CHECK_LAUNCH("vt_conv_igemm(stem)");
    return VT_OK;
}


// ---- round 3: T tiles of 256 rows per workgroup, transposed product ----------------------------------------------
// The kernel above is bound by its epilogue, not by memory: 242 us for the statistics-only pass against 33 us of HBM
// time (rocprofv3, batch 256) -- ~500 VALU instructions per wave and tile (a 2-byte LDS store, a convert and the
// statistics per accumulator element; masks by integer division), 64 int64 atomics per wave and tile.  Here
//   * the product is formed TRANSPOSED (A = filter rows, B = pixels), with the filter rows of MFMA block j permuted
//     to channel 8*(row>>2) + 4j + (row&3): lane (q, u) then holds, for pixel u, the EIGHT consecutive channels
//     8q .. 8q+7 in acc[i][0] | acc[i][1] -- one packed 16-byte store per pixel block straight to y, no LDS window;
//   * the statistics stay in 16 per-lane registers over all T tiles and are reduced across lanes / waves once per
//     workgroup: 64 atomics per 256 T rows instead of per 64;
//   * one span of T*256 + 2W + 2 pixels is staged per workgroup (T = 4: 1.4 x the tensor instead of 2.75 x), the
//     padding masks advance from tile to tile without divisions, and a masked fragment selects its LDS ADDRESS
//     (one v_cndmask) instead of its 16 bytes.
template <int T>
struct StemTLds {
    static constexpr int kMask = 0;                    // T*256 x 4 B
    static constexpr int kZero = T * 1024;             // 16 B (64 reserved)
    static constexpr int kW = kZero + 64;              // 32 x 144 B filter image [n][tap][8]
    static constexpr int kRed = kW + 32 * 144;         // [4 waves][32 channels][2] floats
    static constexpr int kSpanOff = kRed + 4 * 32 * 2 * 4;
    __host__ __device__ static constexpr int bytes(int span_instr) { return kSpanOff + span_instr * 1024; }
};

template <int T, bool STATS>  // STATS: the statistics pass (no affine / ReLU: vt_conv_igemm), else the epilogue pass
__global__ void __launch_bounds__(256, 4) stem_t_kernel(const IgemmArgs p, const int span_instr) {
    using bf = bf16_t;
    using L = StemTLds<T>;
    constexpr int FM = 4, FN = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* sMask = (unsigned*)(smem + L::kMask);
    char* sW = smem + L::kW;
    float* sRed = (float*)(smem + L::kRed);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * (T * kBM);
    const int W = p.Wi, H = p.Hi;

    // ---- stage: input span by LDS-DMA, filter image and padding masks by ordinary stores ------------------------
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned span_base = lds0 + L::kSpanOff;
    const unsigned long zero_src = (unsigned long)(const void*)vt_stem_zero16;
    const bf* __restrict__ xg = (const bf*)p.x;
    for (int i = wave; i < span_instr; i += 4) {
        const long pix = m0 - W - 1 + 64l * i + lane;
        const unsigned long src = (pix >= 0 && pix < p.M) ? (unsigned long)(xg + pix * 8) : zero_src;
        glds16(src, span_base + (unsigned)i * 1024u);
    }
    if (tid < 4) ((unsigned*)(smem + L::kZero))[tid] = 0u;
    {
        const uint4* wsrc = (const uint4*)p.w;  // [Cout][9][8] bf16 = 9 chunks per output channel
        for (int c = tid; c < 32 * 9; c += 256) {
            const int n = c / 9;
            ((uint4*)sW)[c] = n < p.Cout ? wsrc[c] : make_uint4(0, 0, 0, 0);
        }
    }
    {
        // rows tid, tid + 256, ...: (row, column) of the first by division, the rest by advancing 256 positions
        const long m = m0 + tid;
        const int rem = (int)((unsigned)m % (unsigned)(H * W));  // (M < 2^31: vt_stem_dispatch)
        int oi = rem / W, oj = rem - oi * W;
        const int d256 = kBM / W, r256 = kBM - d256 * W;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            unsigned bits = 0;
            if (m + (long)t * kBM < p.M) {
#pragma unroll
                for (int e = 0; e < 3; ++e)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if ((unsigned)(oi + e - 1) < (unsigned)H && (unsigned)(oj + q - 1) < (unsigned)W) bits |= 1u << (e * 4 + q);
            }
            sMask[t * kBM + tid] = bits;
            oj += r256, oi += d256;
            if (oj >= W) oj -= W, ++oi;
            while (oi >= H) oi -= H;
        }
    }

    const int q = lane >> 4, u = lane & 15;
    const bool affine = !STATS && (p.flags & VT_CONV_AFFINE);
    const bool relu = !STATS && (p.flags & VT_CONV_RELU);
    constexpr bool stats = STATS;
    const bool store = !(p.flags & VT_CONV_NOSTORE);
    // this lane's eight channels 8q + 4j + r
    float sc[FN][4], sf[FN][4], s1[FN][4], s2[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 8 * q + 4 * j + r;
            sc[j][r] = 1.f, sf[j][r] = 0.f, s1[j][r] = 0.f, s2[j][r] = 0.f;
            if (affine && n < p.Cout) {
                if (p.scale) sc[j][r] = p.scale[n];
                sf[j][r] = p.shift[n];
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // filter fragments (A): row u of block j = channel 8*(u>>2) + 4j + (u&3); k chunk q = tap (e, q), q == 3 -> zeros
    uint4 wf[3][FN];
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = 8 * (u >> 2) + 4 * j + (u & 3);
            wf[e][j] = q < 3 ? *(const uint4*)(sW + n * 144 + (e * 3 + q) * 16) : make_uint4(0, 0, 0, 0);
        }
    bf* __restrict__ yg = (bf*)p.y;

#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const long mt = m0 + (long)t * kBM + wave * 64;
        if (mt >= p.M) break;  // (uniform per wave; no barrier below)
        unsigned mk[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) mk[i] = sMask[t * kBM + wave * 64 + i * 16 + u] >> q;  // bit 4e = (row e, this lane's tap)
        const unsigned abase = L::kSpanOff + (unsigned)(t * kBM + wave * 64 + u + q) * 16u;
        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            uint4 xf[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                unsigned off = ((mk[i] >> (4 * e)) & 1u) ? abase + (unsigned)(i * 16 + e * W) * 16u : (unsigned)L::kZero;
                asm volatile("" : "+v"(off));  // select the address, not the data
                xf[i] = *(const uint4*)(smem + off);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[e][j]),
                                                                        __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
        }
        // epilogue: pixel mt + 16i + u, channels 8q .. 8q+7
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            unsigned pk[4];
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v0 = acc[i][j][2 * h], v1 = acc[i][j][2 * h + 1];
                    if (affine) v0 = fmaf(v0, sc[j][2 * h], sf[j][2 * h]), v1 = fmaf(v1, sc[j][2 * h + 1], sf[j][2 * h + 1]);
                    if (relu) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
                    const bf b0 = from_float<bf>(v0), b1 = from_float<bf>(v1);
                    const unsigned w2 = (unsigned)__builtin_bit_cast(unsigned short, b0) |
                                        ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
                    pk[2 * j + h] = w2;
                    if constexpr (STATS) {
                        // statistics of what is stored; of the f32 accumulator when nothing is (NOSTORE: the second
                        // pass normalises the accumulator, not a rounded copy)
                        const float f0 = store ? __builtin_bit_cast(float, w2 << 16) : v0;
                        const float f1 = store ? __builtin_bit_cast(float, w2 & 0xffff0000u) : v1;
                        s1[j][2 * h] += f0, s1[j][2 * h + 1] += f1;
                        s2[j][2 * h] = fmaf(f0, f0, s2[j][2 * h]), s2[j][2 * h + 1] = fmaf(f1, f1, s2[j][2 * h + 1]);
                    }
                }
            const long m = mt + i * 16 + u;
            if (store && m < p.M && 8 * q < p.Cout) *(uint4*)(yg + (m * p.ldy + 8 * q)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
    if (stats) {
        // rows past M contributed exact zeros (their masks are empty and STATS excludes AFFINE)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s1[j][r], b = s2[j][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o, 64), b += __shfl_xor(b, o, 64);
                if (u == 0) {
                    const int n = 8 * q + 4 * j + r;
                    sRed[(wave * 32 + n) * 2] = a, sRed[(wave * 32 + n) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < 64) {
            const int n = tid >> 1, k = tid & 1;
            const float v = (sRed[(0 * 32 + n) * 2 + k] + sRed[(1 * 32 + n) * 2 + k]) +
                            (sRed[(2 * 32 + n) * 2 + k] + sRed[(3 * 32 + n) * 2 + k]);
            const int rep = (int)(blockIdx.x % kStatReplicas);
            if (n < p.Cout) vt_stat_add(p.stats, ((long)rep * 2 + k) * p.Cout + n, v);
        }
    }
}

template <int T>
int launch_stem_t(const IgemmArgs& a, hipStream_t st) {
    const int span_px = T * kBM + 2 * a.Wi + 2;
    const int span_instr = (span_px + 63) / 64;
    const int smem = StemTLds<T>::bytes(span_instr);
    if (smem > 64 * 1024) return -1;
    const long blocks = ((long)a.M + T * kBM - 1) / (T * kBM);
    vt_note_kernel("stem_t_kernel<%d,%d>", T, (a.flags & VT_CONV_STATS) ? 1 : 0);
    if (a.flags & VT_CONV_STATS)
        hipLaunchKernelGGL((stem_t_kernel<T, true>), dim3((unsigned)blocks), dim3(256), smem, st, a, span_instr);
    else
        hipLaunchKernelGGL((stem_t_kernel<T, false>), dim3((unsigned)blocks), dim3(256), smem, st, a, span_instr);
    VT_CHECK_LAUNCH("vt_conv_igemm(stem)");
    return VT_OK;
}

}  // namespace

// returns -1 when this kernel does not apply (the caller then uses the general kernels)
int vt_stem_dispatch(IgemmArgs& a, int dtype, void* stream) {
    const int enabled = (1);
    if (!enabled || dtype != VT_BF16) return -1;
    if (a.Cin != 8 || a.ldx != 8 || a.ntaps != 9 || a.ldw != 72 || a.Cout > 64 || a.Cout % 8) return -1;
    if (a.sh != 1 || a.sw != 1 || a.Ho != a.Hi || a.Wo != a.Wi || a.h0 != -1 || a.w0 != -1) return -1;
    if (!a.dense_out || (a.flags & VT_CONV_RESIDUAL) || a.ldy % 8) return -1;
    for (int t = 0; t < 9; ++t)
        if (a.dh[t] != t / 3 || a.dw[t] != t % 3) return -1;
    if ((long)a.M + 2L * a.Wi + 4 > 0x7fffffffL) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (a.Cout <= 32) {  // the Darknet stems: T tiles per workgroup (falls through when the span does not fit the LDS)
        const int tiles = VT_KNOB("VT_STEM_TILES", 4);
        int rc = -1;
        if (tiles >= 4) rc = launch_stem_t<4>(a, st);
        if (rc < 0 && tiles >= 2) rc = launch_stem_t<2>(a, st);
        if (rc >= 0) return rc;
    }
    return a.Cout > 32 ? launch_stem<64>(a, st) : launch_stem<32>(a, st);
}
