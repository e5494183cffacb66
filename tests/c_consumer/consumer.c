/* A consumer of libvt_amd.so written in C: no Python, no PyTorch -- only include/vt_amd.h, the HIP runtime for device
 * memory and a stream, and (as the checker) the plain-C oracle oracle/ref_ops.c.
 *
 * One training-mode ConvNormAct unit of the reference (vision_toolbox/components.py:26-44: Conv2d(bias=False) ->
 * BatchNorm2d -> ReLU) in f32: vt_conv_igemm with the statistics epilogue -> vt_bn_finalize -> vt_bn_act_apply, on
 * 3 x 24 x 12 x 10 -> 40 channels, 3x3 stride 1, against vt_ref_conv2d_fwd + vt_ref_bn_relu_fwd.  Layouts at the
 * boundary: activations NHWC, filter [Cout][kh*kw][Cin]; the oracle is NCHW / OIHW, converted on the host here.
 * Prints C_CONSUMER_OK and returns 0 when outputs, batch statistics and running statistics agree.
 *
 *   gcc -std=gnu99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_consumer/consumer.c \
 *       vision-toolbox_amd/csrc/libvt_amd.so oracle/libvt_ref.so -L/opt/rocm/lib -lamdhip64 -lm -o consumer
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif
void vt_ref_conv2d_fwd(const float* x, const float* wt, float* y, int B, int Cin, int H, int W, int Cout, int k, int stride,
                       int pad);
void vt_ref_bn_relu_fwd(const float* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float* y, int B, int C, int HW, float eps, float momentum, int training, int relu);
#ifdef __cplusplus
}
#endif

#define CHECK_HIP(e)                                                                  \
    do {                                                                              \
        hipError_t e__ = (e);                                                         \
        if (e__ != hipSuccess) {                                                      \
            fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e__), __LINE__); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)
#define CHECK_VT(e)                                                                   \
    do {                                                                              \
        int rc__ = (e);                                                               \
        if (rc__ != VT_OK) {                                                          \
            fprintf(stderr, "libvt_amd error %d at line %d: %s\n", rc__, __LINE__, vt_last_error()); \
            return 3;                                                                 \
        }                                                                             \
    } while (0)

static float frand(unsigned* s) {  /* deterministic, in (-1, 1) */
    *s = *s * 1664525u + 1013904223u;
    return ((*s >> 8) & 0xffff) / 32768.0f - 1.0f;
}

static double rel_l2(const float* a, const float* b, size_t n) {
    double num = 0, den = 0;
    for (size_t i = 0; i < n; ++i) num += ((double)a[i] - b[i]) * ((double)a[i] - b[i]), den += (double)b[i] * b[i];
    return sqrt(num / (den + 1e-30));
}

int main(void) {
    const int B = 3, Cin = 24, H = 12, W = 10, Cout = 40, k = 3, s = 1, pad = 1;
    const int Ho = (H + 2 * pad - k) / s + 1, Wo = (W + 2 * pad - k) / s + 1;
    const size_t nx = (size_t)B * Cin * H * W, nw = (size_t)Cout * Cin * k * k, ny = (size_t)B * Cout * Ho * Wo;
    const float eps = 1e-5f, momentum = 0.1f;
    unsigned seed = 12345u;

    float *x_nchw = (float*)malloc(nx * 4), *w_oihw = (float*)malloc(nw * 4), *gamma = (float*)malloc(Cout * 4),
          *beta = (float*)malloc(Cout * 4);
    for (size_t i = 0; i < nx; ++i) x_nchw[i] = frand(&seed);
    for (size_t i = 0; i < nw; ++i) w_oihw[i] = frand(&seed) * 0.1f;
    for (int c = 0; c < Cout; ++c) gamma[c] = 1.0f + 0.2f * frand(&seed), beta[c] = 0.2f * frand(&seed);

    /* ---- the oracle (NCHW / OIHW) ---- */
    float *z_ref = (float*)malloc(ny * 4), *y_ref = (float*)malloc(ny * 4);
    float *rm_ref = (float*)calloc(Cout, 4), *rv_ref = (float*)malloc(Cout * 4);
    for (int c = 0; c < Cout; ++c) rv_ref[c] = 1.0f;
    vt_ref_conv2d_fwd(x_nchw, w_oihw, z_ref, B, Cin, H, W, Cout, k, s, pad);
    vt_ref_bn_relu_fwd(z_ref, gamma, beta, rm_ref, rv_ref, y_ref, B, Cout, Ho * Wo, eps, momentum, 1, 1);

    /* ---- the library's layouts: NHWC activations, [Cout][tap][Cin] filter ---- */
    float *x_nhwc = (float*)malloc(nx * 4), *w_krsc = (float*)malloc(nw * 4);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < Cin; ++c)
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) x_nhwc[(((size_t)b * H + i) * W + j) * Cin + c] = x_nchw[(((size_t)b * Cin + c) * H + i) * W + j];
    for (int n = 0; n < Cout; ++n)
        for (int c = 0; c < Cin; ++c)
            for (int t = 0; t < k * k; ++t) w_krsc[((size_t)n * k * k + t) * Cin + c] = w_oihw[((size_t)n * Cin + c) * k * k + t];

    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    float *dx, *dw, *dz, *dy, *dstats, *dgamma, *dbeta, *drm, *drv, *dcoef;
    long long* dnbt;
    CHECK_HIP(hipMalloc((void**)&dx, nx * 4));
    CHECK_HIP(hipMalloc((void**)&dw, nw * 4));
    CHECK_HIP(hipMalloc((void**)&dz, ny * 4));
    CHECK_HIP(hipMalloc((void**)&dy, ny * 4));
    CHECK_HIP(hipMalloc((void**)&dstats, VT_STAT_BYTES(Cout)));
    CHECK_HIP(hipMalloc((void**)&dgamma, Cout * 4));
    CHECK_HIP(hipMalloc((void**)&dbeta, Cout * 4));
    CHECK_HIP(hipMalloc((void**)&drm, Cout * 4));
    CHECK_HIP(hipMalloc((void**)&drv, Cout * 4));
    CHECK_HIP(hipMalloc((void**)&dcoef, 4 * Cout * 4)); /* scale | shift | mean | invstd */
    CHECK_HIP(hipMalloc((void**)&dnbt, 16));
    float* ones = (float*)malloc(Cout * 4);
    for (int c = 0; c < Cout; ++c) ones[c] = 1.0f;
    CHECK_HIP(hipMemcpy(dx, x_nhwc, nx * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dw, w_krsc, nw * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dgamma, gamma, Cout * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dbeta, beta, Cout * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(drm, 0, Cout * 4));
    CHECK_HIP(hipMemcpy(drv, ones, Cout * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(dnbt, 0, 16));
    CHECK_VT(vt_memset(dstats, 0, (uint64_t)VT_STAT_BYTES(Cout), st));

    vt_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = VT_F32;
    d.B = B, d.Hi = H, d.Wi = W, d.Cin = Cin, d.ldx = Cin;
    d.Ho = Ho, d.Wo = Wo, d.sh = s, d.sw = s, d.h0 = -pad, d.w0 = -pad;
    d.Cout = Cout, d.ldy = Cout, d.oH = Ho, d.oW = Wo, d.oHs = 1, d.oWs = 1;
    d.ldw = k * k * Cin, d.ldr = Cout, d.flags = VT_CONV_STATS, d.ntaps = k * k;
    for (int t = 0; t < k * k; ++t) d.dh[t] = (int8_t)(t / k), d.dw[t] = (int8_t)(t % k);

    const uint64_t launches0 = vt_launch_count();
    CHECK_VT(vt_conv_igemm(&d, dx, dw, dz, NULL, NULL, NULL, dstats, st));
    CHECK_VT(vt_bn_finalize(dstats, Cout, (double)B * Ho * Wo, dgamma, dbeta, eps, momentum, drm, drv, (int64_t*)dnbt, dcoef,
                            dcoef + Cout, dcoef + 2 * Cout, dcoef + 3 * Cout, st));
    CHECK_VT(vt_bn_act_apply(dz, Cout, dcoef, dcoef + Cout, NULL, 0, dy, Cout, (int64_t)B * Ho * Wo, Cout, 1, VT_F32, st));
    CHECK_HIP(hipStreamSynchronize(st));
    if (vt_launch_count() <= launches0) {
        fprintf(stderr, "no kernel was launched\n");
        return 4;
    }

    float *y_nhwc = (float*)malloc(ny * 4), *y_got = (float*)malloc(ny * 4), *rm = (float*)malloc(Cout * 4),
          *rv = (float*)malloc(Cout * 4), *coef = (float*)malloc(4 * Cout * 4);
    long long nbt[2];
    CHECK_HIP(hipMemcpy(y_nhwc, dy, ny * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(rm, drm, Cout * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(rv, drv, Cout * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(coef, dcoef, 4 * Cout * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(nbt, dnbt, 16, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < Cout; ++c)
            for (int i = 0; i < Ho; ++i)
                for (int j = 0; j < Wo; ++j)
                    y_got[(((size_t)b * Cout + c) * Ho + i) * Wo + j] = y_nhwc[(((size_t)b * Ho + i) * Wo + j) * Cout + c];

    /* (running_mean started at 0 with momentum 0.1: the batch mean is running_mean / 0.1) */
    float* mean_ref = (float*)malloc(Cout * 4);
    for (int c = 0; c < Cout; ++c) mean_ref[c] = rm_ref[c] / momentum;
    const double ey = rel_l2(y_got, y_ref, ny), em = rel_l2(coef + 2 * Cout, mean_ref, Cout), erm = rel_l2(rm, rm_ref, Cout),
                 erv = rel_l2(rv, rv_ref, Cout);
    printf("library version %d, kernel %s\n", vt_version(), vt_last_kernel_name());
    printf("rel L2: y %.2e  batch mean %.2e  running mean %.2e  running var %.2e  num_batches_tracked %lld\n", ey, em, erm, erv,
           nbt[0]);
    /* f32 kernels against an f32 / double oracle: 2e-4 (the tolerance of the unit tests, tests/test_modules_gpu.py) */
    if (!(ey < 2e-4 && em < 2e-4 && erm < 2e-4 && erv < 2e-4 && nbt[0] == 1)) {
        fprintf(stderr, "MISMATCH\n");
        return 1;
    }
    printf("C_CONSUMER_OK\n");
    return 0;
}
