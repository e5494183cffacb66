/*
 * TEST INFRASTRUCTURE -- plain-C restatement (double accumulation) of the arithmetic
 * behind the hot path, independent of torch: direct convolution, training/eval
 * BatchNorm + ReLU, MaxPool2d(3,2,1), and their backward passes.
 *
 * The reference delegates this arithmetic to torch (nn.Conv2d / nn.BatchNorm2d /
 * nn.ReLU at vision_toolbox/components.py:26-44, nn.MaxPool2d at
 * backbones/vovnet.py:94); this file restates the published definitions
 * (torch.nn docs: cross-correlation with zero padding; batch norm with biased batch
 * variance for normalisation and unbiased variance for the running estimate,
 * momentum update; max pooling with -inf padding, first maximum wins) so that the
 * torch-based oracle (oracle/torch_ref.py) and the HIP kernels can both be checked
 * against something that shares no code with either.  PARITY PINNED through
 * tests/test_oracle_c.py (vs torch CPU, which is itself pinned to the reference's
 * golden vectors in tests/test_oracle.py).
 *
 * Layouts: activations NCHW, filters OIHW, all float32 in memory.
 * Only tests/, smoke() and bench.py's cpu_baseline may use this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define X(b, c, h, w) x[(((size_t)(b) * Cin + (c)) * H + (h)) * W + (w)]
#define Wt(o, c, r, t) wt[(((size_t)(o) * Cin + (c)) * k + (r)) * k + (t)]

static int out_dim(int n, int k, int s, int pad) { return (n + 2 * pad - k) / s + 1; }

void vt_ref_conv2d_fwd(const float* x, const float* wt, float* y, int B, int Cin, int H, int W, int Cout, int k,
                       int s, int pad) {
    const int Ho = out_dim(H, k, s, pad), Wo = out_dim(W, k, s, pad);
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < Cout; ++o)
            for (int i = 0; i < Ho; ++i)
                for (int j = 0; j < Wo; ++j) {
                    double acc = 0.0;
                    for (int c = 0; c < Cin; ++c)
                        for (int r = 0; r < k; ++r)
                            for (int t = 0; t < k; ++t) {
                                const int h = i * s - pad + r, w = j * s - pad + t;
                                if (h >= 0 && h < H && w >= 0 && w < W) acc += (double)X(b, c, h, w) * Wt(o, c, r, t);
                            }
                    y[(((size_t)b * Cout + o) * Ho + i) * Wo + j] = (float)acc;
                }
}

/* dx and dw of the convolution above; both are overwritten */
void vt_ref_conv2d_bwd(const float* x, const float* wt, const float* dy, float* dx, float* dw, int B, int Cin,
                       int H, int W, int Cout, int k, int s, int pad) {
    const int Ho = out_dim(H, k, s, pad), Wo = out_dim(W, k, s, pad);
    const size_t nx = (size_t)B * Cin * H * W, nw = (size_t)Cout * Cin * k * k;
    double* ax = (double*)calloc(nx, sizeof(double));
    double* aw = (double*)calloc(nw, sizeof(double));
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < Cout; ++o)
            for (int i = 0; i < Ho; ++i)
                for (int j = 0; j < Wo; ++j) {
                    const double g = dy[(((size_t)b * Cout + o) * Ho + i) * Wo + j];
                    for (int c = 0; c < Cin; ++c)
                        for (int r = 0; r < k; ++r)
                            for (int t = 0; t < k; ++t) {
                                const int h = i * s - pad + r, w = j * s - pad + t;
                                if (h >= 0 && h < H && w >= 0 && w < W) {
                                    ax[(((size_t)b * Cin + c) * H + h) * W + w] += g * Wt(o, c, r, t);
                                    aw[(((size_t)o * Cin + c) * k + r) * k + t] += g * X(b, c, h, w);
                                }
                            }
                }
    for (size_t i = 0; i < nx; ++i) dx[i] = (float)ax[i];
    for (size_t i = 0; i < nw; ++i) dw[i] = (float)aw[i];
    free(ax);
    free(aw);
}

/* y = [relu](gamma * (z - mean) / sqrt(var + eps) + beta); training: batch statistics (biased
 * var) and running update with the unbiased var; eval: running statistics. */
void vt_ref_bn_relu_fwd(const float* z, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float* y, int B, int C, int HW, float eps, float momentum,
                        int training, int relu) {
    const double n = (double)B * HW;
    for (int c = 0; c < C; ++c) {
        double mean, var;
        if (training) {
            double s = 0.0, ss = 0.0;
            for (int b = 0; b < B; ++b)
                for (int p = 0; p < HW; ++p) s += z[((size_t)b * C + c) * HW + p];
            mean = s / n;
            for (int b = 0; b < B; ++b)
                for (int p = 0; p < HW; ++p) {
                    const double d = z[((size_t)b * C + c) * HW + p] - mean;
                    ss += d * d;
                }
            var = ss / n;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (n > 1 ? ss / (n - 1) : var));
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const double istd = 1.0 / sqrt(var + eps);
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) {
                const size_t i = ((size_t)b * C + c) * HW + p;
                double v = gamma[c] * (z[i] - mean) * istd + beta[c];
                if (relu && v < 0.0) v = 0.0;
                y[i] = (float)v;
            }
    }
}

/* backward of the training-mode forward above w.r.t. z, gamma, beta */
void vt_ref_bn_relu_bwd(const float* z, const float* gamma, const float* beta, const float* dy, float* dz,
                        float* dgamma, float* dbeta, int B, int C, int HW, float eps, int relu) {
    const double n = (double)B * HW;
    for (int c = 0; c < C; ++c) {
        double s = 0.0, ss = 0.0;
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) s += z[((size_t)b * C + c) * HW + p];
        const double mean = s / n;
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) {
                const double d = z[((size_t)b * C + c) * HW + p] - mean;
                ss += d * d;
            }
        const double istd = 1.0 / sqrt(ss / n + eps);
        double sg = 0.0, sgx = 0.0;
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) {
                const size_t i = ((size_t)b * C + c) * HW + p;
                const double xh = (z[i] - mean) * istd;
                const double g = (!relu || gamma[c] * xh + beta[c] > 0.0) ? dy[i] : 0.0;
                sg += g;
                sgx += g * xh;
            }
        dgamma[c] = (float)sgx;
        dbeta[c] = (float)sg;
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) {
                const size_t i = ((size_t)b * C + c) * HW + p;
                const double xh = (z[i] - mean) * istd;
                const double g = (!relu || gamma[c] * xh + beta[c] > 0.0) ? dy[i] : 0.0;
                dz[i] = (float)(gamma[c] * istd * (g - sg / n - xh * sgx / n));
            }
    }
}

/* MaxPool2d(kernel 3, stride 2, padding 1); argmax is the flat h*W+w index of the first maximum */
void vt_ref_maxpool3x3s2_fwd(const float* x, float* y, int32_t* argmax, int B, int C, int H, int W) {
    const int Ho = out_dim(H, 3, 2, 1), Wo = out_dim(W, 3, 2, 1);
    for (int bc = 0; bc < B * C; ++bc)
        for (int i = 0; i < Ho; ++i)
            for (int j = 0; j < Wo; ++j) {
                float best = -INFINITY;
                int bi = -1;
                for (int r = 0; r < 3; ++r)
                    for (int t = 0; t < 3; ++t) {
                        const int h = i * 2 - 1 + r, w = j * 2 - 1 + t;
                        if (h < 0 || h >= H || w < 0 || w >= W) continue;
                        const float v = x[((size_t)bc * H + h) * W + w];
                        if (bi < 0 || v > best || v != v) {
                            best = v;
                            bi = h * W + w;
                        }
                    }
                y[((size_t)bc * Ho + i) * Wo + j] = best;
                argmax[((size_t)bc * Ho + i) * Wo + j] = bi;
            }
}

void vt_ref_maxpool3x3s2_bwd(const float* dy, const int32_t* argmax, float* dx, int B, int C, int H, int W) {
    const int Ho = out_dim(H, 3, 2, 1), Wo = out_dim(W, 3, 2, 1);
    memset(dx, 0, (size_t)B * C * H * W * sizeof(float));
    for (int bc = 0; bc < B * C; ++bc)
        for (int o = 0; o < Ho * Wo; ++o) dx[(size_t)bc * H * W + argmax[(size_t)bc * Ho * Wo + o]] += dy[(size_t)bc * Ho * Wo + o];
}
