// vt_stem6.hip -- the 6x6 stride-2 stem convolution of the YOLOv5 Darknets (bf16, inference epilogue):
// 3 input channels (padded to one 16-byte pixel) -> 80 output channels at half resolution
// (reference darknet.py:144-146, `DarknetYOLOv5`: stem = ConvNormAct(3, stem_channels, 6, 2) with
// padding (6 - 1) // 2 = 2; Darknet-YOLOv5x: 80 channels, 640x640 x batch 64 -> 6.55 M output pixels).
//
// The layer is HBM-bound (420 MB in, 1.05 GB out, 113 GFLOP of real work).  The gather kernel
// (vt_igemm.hip) stages the 36 taps of every output pixel separately: 256 rows x 9 K-steps x 64 B plus the
// 80 x 288 filter per workgroup = 4.8 GB through the global->LDS path, which bounds it at 0.83-0.87 ms
// (that path moves ~7 TB/s chip-wide; measured on the 80-channel layers of the same model).  Here
//   * a workgroup owns a 4 x 64 block of output pixels (one row per wave) and DMAs the 12 x 132 input
//     pixels under it into LDS ONCE (25 KB; even and odd columns apart, so that the 16 lanes of a
//     fragment -- 16 consecutive output pixels, input stride 2 -- read 16 consecutive 16-byte slots);
//   * workgroups are persistent (two per CU) and keep the whole filter in LDS (9 K-steps x 80 rows x 64 B, chunk
//     positions swizzled by the row: conflict-free 16-row fragment reads);
//   * K = 9 steps of 4 taps x 8 channels in the tensor's own tap order, 20 MFMAs per wave and step;
//     the FILTER is the MFMA's row operand, so a lane ends up with 4 consecutive channels of one pixel
//     per accumulator, and the filter rows are permuted at staging time such that its five accumulators
//     are channels [8q, 8q+8), [32+8q, 32+8q+8), [64+4q, 64+4q+4): 16 + 16 + 8 byte stores straight from
//     the registers, no LDS staging of the output;
//   * the next block's input is requested before the epilogue of the current one.
// Same K order and the same per-step MFMA as the gather kernel, which remains the path for every other
// epilogue (statistics, residual) and shape.
#include <stdlib.h>

#include "vt_common.h"
#include "vt_igemm_args.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int vt_stem6_zero16[4];

__device__ __forceinline__ void glds16(unsigned long gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base))
        : "memory");
}

constexpr int kTH = 4, kTW = 64;                       // output block: one row per wave x 64 columns
constexpr int kPR = 2 * kTH + 4, kPC = 2 * kTW + 4;    // 12 x 132 input pixels, stored [row][column parity][66]
constexpr int kHalf = kPC / 2;
constexpr int kPatchSlots = kPR * kPC;
constexpr int kPatchInstr = (kPatchSlots + 63) / 64;   // 25 LDS-DMA instructions of 64 x 16 B
constexpr int kN = 80, kTaps = 36, kSteps = kTaps / 4;
// filter image in LDS: [K-step s][row R = 16 j + u][4 chunks of 16 B], the chunk position XOR-swizzled by the row as in
// vt_igemm.hip (conflict-free 16-row fragment reads; rows of 592 B with a pad slot measured 50 % bank conflicts)
constexpr int kWSlots = kSteps * kN * 4;
constexpr int kWInstr = (kWSlots + 63) / 64;           // 45
constexpr int kWBytes = kWInstr * 1024, kPatchBytes = kPatchInstr * 1024;
constexpr int kSmem = kWBytes + kPatchBytes;           // 73,728 B: two workgroups per CU

// channel held by accumulator j (0..4), row u = 4q + r of the MFMA result
__host__ __device__ constexpr int stem6_channel(int j, int u) {
    return j < 4 ? (j >> 1) * 32 + 8 * (u >> 2) + (j & 1) * 4 + (u & 3) : 64 + u;
}

__global__ void __launch_bounds__(256, 2) stem6_kernel(const IgemmArgs p, const int tiles_h, const int tiles_w,
                                                       const int ntiles, const int per_xcd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    const unsigned w_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned patch_base = w_base + kWBytes;
    const unsigned long zero_src = (unsigned long)(const void*)vt_stem6_zero16;
    const bf16_t* __restrict__ xg = (const bf16_t*)p.x;
    const bf16_t* __restrict__ wg = (const bf16_t*)p.w;
    const int H = p.Hi, W = p.Wi;

    // ---- the filter, once per workgroup: LDS row R = 16 j + u holds channel stem6_channel(j, u) ----------
    for (int k = wave; k < kWInstr; k += 4) {
        const int slot = k * 64 + lane;
        const int row = slot >> 2, s_ = row / kN, R = row - s_ * kN;
        const int c = (slot & 3) ^ ((0x1320 >> (((R >> 2) & 3) * 4)) & 3);
        const bool ok = slot < kWSlots;
        const int ch = stem6_channel(R >> 4, R & 15);
        glds16(ok ? (unsigned long)(wg + ((long)ch * p.ldw + (4 * s_ + c) * 8)) : zero_src, w_base + (unsigned)k * 1024u);
    }

    // XCD-blocked block order: workgroups b, b + 8, ... share an XCD (and its L2) and walk one contiguous eighth of
    // the blocks, so the 4 halo rows / columns two neighbouring blocks share are fetched from HBM once
    const int xcd = blockIdx.x & 7, stride = gridDim.x >> 3;
    int lt = blockIdx.x >> 3;
    auto tile_of = [&](int l) {
        const int t = xcd * per_xcd + l;
        return (l < per_xcd && t < ntiles) ? t : -1;
    };
    auto issue_patch = [&](int tile) {
        const int tw = tile % tiles_w, rest = tile / tiles_w;
        const int th = rest % tiles_h, b = rest / tiles_h;
        const int row0 = 2 * kTH * th - 2, col0 = 2 * kTW * tw - 2;
        for (int k = wave; k < kPatchInstr; k += 4) {
            const int slot = k * 64 + lane;
            const int pr = slot / kPC, rem = slot - pr * kPC;
            const int par = rem >= kHalf ? 1 : 0;
            const int pc = 2 * (rem - par * kHalf) + par;
            const int ih = row0 + pr, iw = col0 + pc;
            const bool ok = slot < kPatchSlots && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            glds16(ok ? (unsigned long)(xg + (((long)b * H + ih) * W + iw) * 8) : zero_src, patch_base + (unsigned)k * 1024u);
        }
    };

    // ---- per-lane constants ---------------------------------------------------------------------------------
    // pixel fragment i of this wave's row: output column 16 i + c16, tap 4 s + g of step s
    int toff[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        const int t = 4 * s + g, kh = t / 6, kw = t - 6 * kh;
        toff[s] = (kh * kPC + (kw & 1) * kHalf + (kw >> 1)) * 16;
    }
    const char* pix = smem + kWBytes + ((2 * wave) * kPC + c16) * 16;
    const char* wrow = smem + (c16 * 4 + (g ^ ((0x1320 >> (((c16 >> 2) & 3) * 4)) & 3))) * 16;  // + (80 s + 16 j) rows of 64 B
    // epilogue coefficients of the 20 channels this lane stores (rows 4 g + r of the five accumulators)
    const bool affine = (p.flags & VT_CONV_AFFINE) != 0, relu = (p.flags & VT_CONV_RELU) != 0;
    float sc[20], sf[20];
#pragma unroll
    for (int e = 0; e < 20; ++e) {
        const int ch = stem6_channel(e >> 2, 4 * g + (e & 3));
        sc[e] = (affine && p.scale) ? p.scale[ch] : 1.f;
        sf[e] = affine ? p.shift[ch] : 0.f;
    }
    bf16_t* __restrict__ yg = (bf16_t*)p.y;

    int cur = tile_of(lt);
    if (cur >= 0) issue_patch(cur);
    while (cur >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // this block's input (and, the first time, the filter) is in LDS

        f32x4 acc[4][5];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            uint4 wf[5], xf[4];
#pragma unroll
            for (int j = 0; j < 5; ++j) wf[j] = *(const uint4*)(wrow + (s * kN + j * 16) * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *(const uint4*)(pix + toff[s] + i * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]),
                                                                        __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // every wave is done with the input block: the next one may land
        lt += stride;
        const int nxt = tile_of(lt);
        if (nxt >= 0) issue_patch(nxt);

        // ---- epilogue: straight from the accumulators ------------------------------------------------------
        const int tw = cur % tiles_w, rest = cur / tiles_w;
        const int th = rest % tiles_h, b = rest / tiles_h;
        const int oh = kTH * th + wave;
        if (oh < p.Ho) {
            bf16_t* yrow = yg + ((long)b * p.Ho + oh) * p.Wo * (long)p.ldy;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ow = kTW * tw + 16 * i + c16;
                if (ow < p.Wo) {
                    float v[20];
#pragma unroll
                    for (int e = 0; e < 20; ++e) {
                        float t = fmaf(acc[i][e >> 2][e & 3], sc[e], sf[e]);
                        v[e] = relu ? fmaxf(t, 0.f) : t;
                    }
                    bf16_t* yp = yrow + (long)ow * p.ldy;
                    *(uint4*)(yp + 8 * g) = VecIO<bf16_t>::pack(v);
                    *(uint4*)(yp + 32 + 8 * g) = VecIO<bf16_t>::pack(v + 8);
                    *(uint2*)(yp + 64 + 4 * g) = make_uint2(VecIO<bf16_t>::pack2(v[16], v[17]), VecIO<bf16_t>::pack2(v[18], v[19]));
                }
            }
        }
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a workgroup without blocks still has the filter in flight)
}

}  // namespace

// returns -1 when this kernel does not apply (the caller then uses the general kernels)
int vt_stem6_dispatch(IgemmArgs& a, int dtype, void* stream) {
    const int enabled = VT_KNOB("VT_STEM6_KERNEL", 1);
    if (!enabled || dtype != VT_BF16) return -1;
    if (a.Cin != 8 || a.ldx != 8 || a.ntaps != kTaps || a.ldw != kTaps * 8 || a.Cout != kN) return -1;
    if (a.sh != 2 || a.sw != 2 || a.h0 != -2 || a.w0 != -2 || (a.Hi & 1) || (a.Wi & 1) || a.Ho != a.Hi / 2 || a.Wo != a.Wi / 2)
        return -1;
    if (!a.dense_out || (a.flags & ~(VT_CONV_AFFINE | VT_CONV_RELU)) || a.ldy % 8 || a.ldy < kN) return -1;
    for (int t = 0; t < kTaps; ++t)
        if (a.dh[t] != t / 6 || a.dw[t] != t % 6) return -1;
    const int tiles_h = (a.Ho + kTH - 1) / kTH, tiles_w = (a.Wo + kTW - 1) / kTW;
    const long ntiles = (long)a.B * tiles_h * tiles_w;
    if (ntiles > 0x3fffffffL) return -1;
    const int per_xcd = (int)((ntiles + 7) / 8);
    const int cus = vt_device_cus();
    int wgs_per_xcd = cus > 0 ? (2 * cus + 7) / 8 : 64;  // two workgroups per CU
    if (wgs_per_xcd > per_xcd) wgs_per_xcd = per_xcd;
    {
        const int rc = vt_raise_dynamic_lds((const void*)stem6_kernel, kSmem, "vt_conv_igemm(stem6)");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("stem6_kernel<80>");
    hipLaunchKernelGGL(stem6_kernel, dim3((unsigned)(8 * wgs_per_xcd)), dim3(256), kSmem, (hipStream_t)stream, a, tiles_h, tiles_w,
                       (int)ntiles, per_xcd);
    VT_CHECK_LAUNCH("vt_conv_igemm(stem6)");
    return VT_OK;
}
