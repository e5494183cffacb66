// vt_igemm_args.h -- kernel argument block shared by the implicit-GEMM conv kernels.
#pragma once
#include "vt_common.h"

// internal epilogue flag (never in a caller's vt_conv_desc.flags): the launch is vt_conv_dgrad_bnred's fused form
#define VT_CONV_BNRED 0x1000

#ifndef VT_MFMA_SETPRIO
#define VT_MFMA_SETPRIO 0  // 1: s_setprio(1) around the MFMA cluster of a main-loop step (vt_igemm / vt_igemm_span / vt_igemm_pspan)
#endif

struct IgemmArgs {
    const void* x;
    const void* w;
    void* y;
    const float* scale;
    const float* shift;
    const void* res;
    float* stats;
    int B, Hi, Wi, Cin, ldx, Ho, Wo, sh, sw, h0, w0, Cout, ldy, oH, oW, oHs, oWs, oh0, ow0;
    int ldw, ldr, flags, ntaps;
    int M, Ktot, tiles_m, tiles_n, chunk, dense_out;
    int fast_dma;  // span kernel: scalar-base LDS-DMA addressing on interior tiles
    int8_t dh[VT_MAX_TAPS];
    int8_t dw[VT_MAX_TAPS];
    // VT_CONV_BNRED (vt_conv_dgrad_bnred): mean / invstd of the unit whose d(y) this launch produces
    const float* aux0;
    const float* aux1;
};

// element offset of output column n relative to the row's base pixel: n itself, or under VT_CONV_D2S the pixel
// (a, b) = ((n / C') >> 1, (n / C') & 1) of the 2 x 2 block and channel n % C' (C' = Cout / 4)
__device__ __forceinline__ long vt_out_col(const IgemmArgs& p, int n, int ld) {
    if (!(p.flags & VT_CONV_D2S)) return n;
    const int cq = p.Cout >> 2;
    const int blk = n / cq;
    return ((long)(blk >> 1) * p.oW + (blk & 1)) * ld + (n - blk * cq);
}

// vt_igemm_span.hip: input-span kernel for stride-1-grid convs; returns -1 when it does not
// apply to `a` (the caller then launches the general kernel), else a VT_* status.
int vt_span_dispatch(IgemmArgs& a, int dtype, void* stream);

// vt_igemm_span6.hip: persistent span kernel, one 12-wave workgroup per CU (two compute groups half a step apart +
// four loader waves); -1 when it does not apply.
int vt_span6_dispatch(IgemmArgs& a, int dtype, void* stream);

// vt_igemm_pspan.hip: persistent span kernel with the whole filter resident in LDS (eight compute + four loader waves
// per CU) for the short-K, HBM-bound convs with <= 128 output channels, stride 1 or 3x3 stride 2; -1 when it does not apply.
int vt_pspan_dispatch(IgemmArgs& a, int dtype, void* stream);

// vt_stem.hip: 3x3 stride-1 convolution over 8-channel (padded RGB) pixels; -1 when it does not apply.
int vt_stem_dispatch(IgemmArgs& a, int dtype, void* stream);

// vt_stem6.hip: 6x6 stride-2 convolution over 8-channel (padded RGB) pixels to 80 channels, inference epilogue
// (the YOLOv5x stem); -1 when it does not apply.
int vt_stem6_dispatch(IgemmArgs& a, int dtype, void* stream);
