/*
 * vt_amd.h -- C-ABI of libvt_amd.so: the MI355X (gfx950) hot path of the
 * Darknet / CSPDarknet / VoVNet backbones.
 *
 * The reference (gau-nernst/vision-toolbox) has NO native layer: every entry
 * point below replaces a torch.nn / ATen call made from the reference's Python.
 * The call site it replaces is cited per function as `file:line` relative to
 * the reference checkout.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching
 *     allocator in our host code); the library allocates nothing persistent.
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it,
 *     no entry point synchronises.
 *   - activations are NHWC ("channels_last"): element (b,h,w,c) of a tensor
 *     with pixel stride `ld` lives at ((b*H + h)*W + w)*ld + c.  `ld >= C`
 *     lets a tensor be a channel slice of a wider concat buffer.
 *   - dtype: VT_F32 (exact-f32 MFMA, parity mode) or VT_BF16 (bf16 storage,
 *     f32 accumulate).  Channel counts / strides / offsets must be multiples
 *     of 16 bytes' worth of elements (4 for f32, 8 for bf16).
 *   - conv weights are [Cout][taps][Cin] ("KRSC"), i.e. the memory image of a
 *     torch OIHW weight held in channels_last format.
 *   - return value: VT_OK or a VT_ERR_* code; vt_last_error() gives the text.
 *     Nothing aborts.  Stateless and re-entrant apart from that thread-local
 *     error string.
 */
#ifndef VT_AMD_H
#define VT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VT_OK 0
#define VT_ERR_INVALID 1     /* bad argument: null pointer, misalignment, size  */
#define VT_ERR_UNSUPPORTED 2 /* shape / dtype this library has no kernel for    */
#define VT_ERR_HIP 3         /* HIP runtime reported an error                   */

#define VT_F32 0
#define VT_BF16 1
#define VT_I64 2 /* collectives only (vt_allreduce_bucket): the fixed-point BatchNorm sums */

#define VT_MAX_TAPS 36 /* 6x6 stem of DarknetYOLOv5 (darknet.py:109) */

/* epilogue flags of vt_conv_igemm */
#define VT_CONV_RELU 1     /* y = max(y, 0) after scale/shift                   */
#define VT_CONV_STATS 2    /* accumulate per-channel sum / sum-of-squares       */
#define VT_CONV_RESIDUAL 4 /* y += residual (after relu)                        */
#define VT_CONV_AFFINE 8   /* y = y*scale[c] + shift[c]; scale==NULL means 1    */
#define VT_CONV_NOSTORE 32 /* with VT_CONV_STATS: only the statistics -- of the f32 accumulator, not of a rounded copy --,
                            * y is not written (y may be NULL); RGB stem only */
#define VT_CONV_D2S 16     /* depth-to-space 2x2: the Cout = 4*C' columns of grid pixel (i, j) are the output pixels
                            * (2i+a, 2j+b) x C' channels, column = (2a+b)*C' + c; needs oHs = oWs = 2, oh0 = ow0 = 0,
                            * oH = 2*Ho, oW = 2*Wo, no STATS / AFFINE / RELU.  One launch then forms the whole data
                            * gradient of a 3x3 stride-2 convolution (the four parity classes as column blocks of a
                            * 2x2-tap filter image, zero where a class has no tap) and reads dz once. */

#define VT_STAT_REPLICAS 16
/* A statistics buffer (`stats` of vt_conv_igemm, `sums` of the BatchNorm-backward reductions) is
 * int64[VT_STAT_REPLICAS][2][C][2], VT_STAT_BYTES(C) bytes, zeroed by the caller and passed as float*: fixed point,
 * value = hi*2^12 + lo/2^33, accumulated with integer atomics.  Integer addition is associative, so the statistics --
 * and with them every activation, loss and data gradient -- are bit-identical from run to run;
 * vt_bn_finalize / vt_bn_bwd_finalize sum the replicas exactly.  SyncBatchNorm all-reduces the buffer as int64. */
#define VT_STAT_BYTES(C) ((int64_t)VT_STAT_REPLICAS * 2 * (C) * 16)

/*
 * Geometry of one implicit-GEMM convolution launch.  One descriptor covers the
 * forward conv, the data gradient (run on dz with a re-packed filter) and the
 * per-parity-class data gradient of stride-2 convs.
 *
 *   out(b, i*oHs+oh0, j*oWs+ow0, n) =
 *       sum_{t<ntaps} sum_{c<Cin} in(b, i*sh+h0+dh[t], j*sw+w0+dw[t], c) * w[n][t][c]
 *   for i<Ho, j<Wo; taps that fall outside [0,Hi)x[0,Wi) read zero.
 */
typedef struct vt_conv_desc {
    int32_t dtype;
    int32_t B, Hi, Wi, Cin, ldx; /* gather source                              */
    int32_t Ho, Wo;              /* output grid iterated (GEMM M = B*Ho*Wo)    */
    int32_t sh, sw, h0, w0;      /* input step / origin                        */
    int32_t Cout, ldy;           /* GEMM N and output pixel stride             */
    int32_t oH, oW;              /* extent of the tensor written               */
    int32_t oHs, oWs, oh0, ow0;  /* placement of grid point (i,j) in it        */
    int32_t ldw;                 /* weight row stride, elements (>= ntaps*Cin) */
    int32_t ldr;                 /* residual pixel stride                      */
    int32_t flags;               /* VT_CONV_*                                  */
    int32_t ntaps;
    int8_t dh[VT_MAX_TAPS];
    int8_t dw[VT_MAX_TAPS];
} vt_conv_desc;

/* ---- library ------------------------------------------------------------ */
int vt_version(void);
const char* vt_last_error(void);
/* name (template instantiation) of the convolution / filter-gradient kernel the calling thread's most
 * recent vt_conv_igemm / vt_conv_wgrad dispatch chose; bench.py labels its roofline entry with it. */
const char* vt_last_kernel_name(void);
/* number of kernel launches issued by this process through the library; the
 * GPU tests assert it advances, i.e. that the HIP path is what ran. */
uint64_t vt_launch_count(void);
/* Experiment / test switch of the kernel dispatchers (the names INTEGRATION.md lists, e.g. "VT_SPAN6"): each is read
 * from the environment variable of the same name ONCE per process; this call sets it afterwards (the GPU tests force a
 * kernel with it).  No switch changes results beyond summation order. */
int vt_set_knob(const char* name, int32_t value);
int vt_memset(void* ptr, int value, uint64_t bytes, void* stream);

/* ---- convolution --------------------------------------------------------
 * Replaces nn.Conv2d inside ConvNormAct (components.py:26-35) and, with the
 * AFFINE/RELU/RESIDUAL epilogue, the eval-mode BatchNorm2d + ReLU that follow
 * it (components.py:36-44) and DarknetBlock's add (darknet.py:28).  With
 * VT_CONV_STATS it also produces the batch statistics BatchNorm2d needs in
 * training mode.  The same entry point computes conv data gradients.
 * `stats` is a statistics buffer of VT_STAT_BYTES(Cout) bytes (see VT_STAT_REPLICAS), zeroed by the caller. */
int vt_conv_igemm(const vt_conv_desc* d, const void* x, const void* w, void* y,
                  const float* scale, const float* shift, const void* residual,
                  float* stats, void* stream);

/* A data gradient that also REDUCES the BatchNorm backward of the unit whose output it differentiates (round 6).
 * `d` describes a convolution run on dz with the re-packed filter (as vt_conv_igemm is called for a data gradient: flags
 * 0, every pixel of the output tensor produced: oHs = oWs = 1, oH = Ho, oW = Wo); its result dy = d(y) is the gradient of
 * the output y = act(z * scale + shift) of a ConvNormAct unit (components.py:26-44) that has no other consumer.  Besides
 * storing dy the call adds that unit's backward sums -- exactly what vt_bn_act_bwd_reduce(dy, z, ...) adds:
 *   sums[0][c] += sum g,  sums[1][c] += invstd[c] * sum g * (z - mean[c]),   g = dy (as stored) * [z * scale + shift > 0]
 * (`relu` 0: no mask) -- so that the separate reduction pass, which reads dy and z once more, disappears: where the
 * two-group persistent 3x3 kernel takes the launch the sums come out of its epilogue (dy in registers, z read like a
 * residual operand); everywhere else the call is the two launches it replaces.  `sums` is a statistics buffer
 * (VT_STAT_REPLICAS), fixed point: the result does not depend on the order in which workgroups finish.
 * Replaces the autograd backward of nn.Conv2d (components.py:26-35) with respect to its input + the reductions of
 * the autograd backward of nn.BatchNorm2d / nn.ReLU (components.py:36-44) of the unit in front of it. */
int vt_conv_dgrad_bnred(const vt_conv_desc* d, const void* dz, const void* w, void* dy, const void* z, int32_t ldz,
                        const float* scale, const float* shift, const float* mean, const float* invstd,
                        int32_t relu, float* sums, void* stream);

/* Depthwise convolution (round 6): nn.Conv2d(C, C, k, stride s, padding pad, dilation dil, groups=C, bias=False) inside a
 * ConvNormAct (components.py:26-35 with `groups = in_channels`) -- forward, data gradient, filter gradient.  x / dx are
 * [B][Hi][Wi][C], z / dz [B][Ho][Wo][C] with Ho = (Hi + 2 pad - dil (k-1) - 1) / s + 1; `w` / `dw` are the f32 [C][k*k]
 * image of the torch [C, 1, k, k] weight (bf16 launches round the filter as the convolution kernels' bf16 mirror does).
 *   vt_dwconv_fwd:   z = conv(x, w); `stats` non-NULL: also the per-channel sum / sum of squares of the STORED z (a
 *                    statistics buffer, VT_STAT_REPLICAS: the contract of vt_conv_igemm's VT_CONV_STATS)
 *   vt_dwconv_dgrad: dx = conv_transpose(dz, w) (+ residual, which may alias dx)
 *   vt_dwconv_wgrad: dw[c][t] += sum_pixels dz * x_shifted  (f32 atomics)
 * C a multiple of 8 (bf16) / 4 (f32), k <= 7.  Off the Darknet / VoVNet path; streaming kernels, no matrix pipe. */
int vt_dwconv_fwd(const void* x, int32_t ldx, const float* w, void* z, int32_t ldz, float* stats, int32_t B, int32_t Hi,
                  int32_t Wi, int32_t C, int32_t k, int32_t s, int32_t pad, int32_t dil, int32_t dtype, void* stream);
int vt_dwconv_dgrad(const void* dz, int32_t lddz, const float* w, void* dx, int32_t lddx, const void* residual, int32_t ldr,
                    int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t k, int32_t s, int32_t pad, int32_t dil,
                    int32_t dtype, void* stream);
int vt_dwconv_wgrad(const void* x, int32_t ldx, const void* dz, int32_t lddz, float* dw, int32_t B, int32_t Hi, int32_t Wi,
                    int32_t C, int32_t k, int32_t s, int32_t pad, int32_t dil, int32_t dtype, void* stream);

/* Filter gradient: dw[n][t][c] += sum_pixels dz(pix,n) * x_gathered(pix,t,c),
 * fp32 accumulation straight into the (channels_last) .grad of the weight.
 * `d` is the forward descriptor (ldy = pixel stride of dz).  Replaces the
 * autograd backward of nn.Conv2d (components.py:26-35). */
int vt_conv_wgrad(const vt_conv_desc* d, const void* x, const void* dz, float* dw,
                  int32_t ldgw, void* stream);

/* The same filter gradient in two stages: every (tile, pixel split) workgroup stores its partial tile into its own slab of
 * `scratch` (plain stores) and a second kernel adds the slabs to dw in split order -- no atomics, and a result that does
 * not depend on the order in which workgroups finish.  One scratch of ~40 MB serves every layer (launches on one stream
 * reuse it, so it stays in the memory-side cache); a scratch too small for a layer's usual split gets fewer, longer splits,
 * never atomics across splits.  (Deterministic mode.) */
int vt_conv_wgrad_slabs(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw,
                        void* scratch, int64_t scratch_bytes, void* stream);

/* n filter gradients of ONE descriptor -- same-shape layers, e.g. the 3x3 convs of a stage's DarknetBlocks
 * (darknet.py:20-28) or of an OSA chain (vovnet.py:41-44) -- in as few launches as the kernels allow: the CU-owning
 * stride-1 3x3 kernel takes up to 8 layers per launch and pays its prologue and its f32 atomic flush once per launch
 * (a filter gradient has no consumer before the optimiser, so a caller may hold the layers of a stage back until the
 * last one's dz exists).  Shapes that kernel does not cover run one by one, exactly as vt_conv_wgrad -- except, with
 * the knob VT_WGRAD_GROUP_1X1=1, 1x1 stride-1 layers, which then share launches of the general kernel (up to 8 layers,
 * one atomic flush; off by default).  x / dz / dw are HOST arrays of n device pointers. */
int vt_conv_wgrad_group(const vt_conv_desc* d, int32_t n, const void* const* x, const void* const* dz,
                        float* const* dw, int32_t ldgw, void* stream);

/* dst[i] (+)= hi[i]*2^12 + lo[i]/2^33 for a fixed-point buffer q = int64[n][2] (vt_colsum_fixed). */
int vt_fixed_to_f32(const void* q, float* dst, int64_t n, int32_t accumulate, void* stream);

/* Re-pack a [Cout][ntaps][Cin] filter (f32 master or dtype mirror) into the
 * [Cin][nsel][Cout] image the data-gradient launch reads; sel[i] is the source
 * tap of packed tap i (-1: a zero tap, for the column blocks of a VT_CONV_D2S filter image). */
int vt_pack_dgrad_filter(const void* w, int32_t src_dtype, int32_t ldw, void* out,
                         int32_t dst_dtype, const int32_t* sel_host, int32_t nsel,
                         int32_t Cout, int32_t ntaps, int32_t Cin, void* stream);
/* The same for n filters in ceil(n / VT_PACK_BATCH) launches (bf16 -> bf16 only): a train step re-packs every filter
 * once, 73 launches of ~6 us for CSPDarknet-53, which cost 0.3 ms of the step through the launches alone.  The
 * executor (vt_run_ops*) gathers consecutive VT_OP_PACK_DGRAD ops of one stream into this call by itself. */
#define VT_PACK_BATCH 40
typedef struct vt_pack_item {
    const void* w;
    void* out;
    int32_t ldw, nsel, Cout, ntaps, Cin;
    int8_t sel[VT_MAX_TAPS];
} vt_pack_item;
int vt_pack_dgrad_filter_batch(const vt_pack_item* items, int32_t n, void* stream);

/* ---- BatchNorm2d + ReLU (components.py:36-44) --------------------------- */
/* training: stats -> batch mean/var, running-stat update (momentum, unbiased
 * var), scale = gamma*invstd, shift = beta - mean*scale. */
int vt_bn_finalize(const float* stats, int32_t C, double count, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, int64_t* num_batches_tracked, float* scale,
                   float* shift, float* mean, float* invstd, void* stream);
/* Fold the VT_STAT_REPLICAS replicas of a statistics / sums buffer into replica 0 (exact integer adds) and zero the
 * others: SyncBatchNorm then all-reduces 32*C bytes per layer instead of the whole buffer. */
int vt_stat_fold(float* stats, int32_t C, void* stream);
/* eval: coefficients from the running statistics. */
int vt_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int32_t C, float* scale,
                      float* shift, float* mean, float* invstd, void* stream);
/* n of them in ceil(n / VT_PACK_BATCH) launches (the executor gathers consecutive VT_OP_BN_EVAL_COEFFS ops itself) */
typedef struct vt_bn_eval_item {
    const float *gamma, *beta, *running_mean, *running_var;
    float *scale, *shift, *mean, *invstd;
    int32_t C;
    float eps;
} vt_bn_eval_item;
int vt_bn_eval_coeffs_batch(const vt_bn_eval_item* items, int32_t n, void* stream);
/* y = act(z*scale + shift) [+ residual].  `relu` here and in vt_bn_act_bwd_reduce / _bwd_apply is an ACTIVATION CODE
 * (ConvNormAct's `act`, components.py:37-44): 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 SiLU ("swish"), 4 GELU (exact).  Codes 0 / 1
 * are the Darknet / VoVNet path; 2-4 run generic instantiations of the same kernels (round 5). */
int vt_bn_act_apply(const void* z, int32_t ldz, const float* scale, const float* shift,
                    const void* residual, int32_t ldr, void* y, int32_t ldy, int64_t M,
                    int32_t C, int32_t relu, int32_t dtype, void* stream);
/* sums (statistics buffer, see VT_STAT_REPLICAS): [0][c] += sum g, [1][c] += sum g*xhat, g = dy*[z*scale+shift>0] */
int vt_bn_act_bwd_reduce(const void* dy, int32_t lddy, const void* z, int32_t ldz,
                         const float* scale, const float* shift, const float* mean,
                         const float* invstd, int64_t M, int32_t C, int32_t relu,
                         int32_t dtype, float* sums, void* stream);
/* dgamma += pscale * sum g*xhat; dbeta += pscale * sum g; coef[3][C] for vt_bn_act_bwd_apply.
 * train!=0: full batch-norm gradient; train==0: statistics are constants.
 * SyncBatchNorm (configs/base.yaml:22): `sums` then hold the all-reduced sums and `count` the
 * global sample count; pscale = 1/world keeps the parameter gradients rank-local in size, so
 * the gradient all-reduce(mean) that follows yields the same values as torch's SyncBatchNorm. */
int vt_bn_bwd_finalize(const float* sums, int32_t C, double count, double pscale,
                       const float* scale, const float* mean, const float* invstd, int32_t train,
                       float* dgamma, float* dbeta, float* coef, void* stream);
/* dz = coef0[c]*g - coef1[c]*z + coef2[c], g = dy * act'(z*scale + shift).
 * scale, shift and coef ALL NULL: dz = dy * act'(z) -- the activation of a ConvNormAct built with norm="none"
 * (reference components.py:36: nn.Identity between the biased conv and the activation). */
int vt_bn_act_bwd_apply(const void* dy, int32_t lddy, const void* z, int32_t ldz,
                        const float* scale, const float* shift, const float* coef,
                        void* dz, int32_t lddz, int64_t M, int32_t C, int32_t relu,
                        int32_t dtype, void* stream);

/* The three calls above in ONE launch where the operands fit the register file (round 6, vt_bn_bwd_fused.hip): every
 * thread keeps its rows of dy and z packed in registers between the reduction and the apply pass, two device-scope grid
 * barriers (at most one workgroup per CU: all resident) separate reduce | finalize | apply -- dy and z are read once instead
 * of twice and two launches disappear.  bf16, activation code 0 / 1, C <= 4096, up to 13 x 16 bytes per thread and tensor
 * (256 channels @14x14 and smaller at batch 256); any other call runs vt_bn_act_bwd_reduce, vt_bn_bwd_finalize and
 * vt_bn_act_bwd_apply, with identical arguments and results that differ by the order of the f32 partial sums only.
 * `sums`: a zeroed statistics buffer; `sync`: 16 zeroed bytes (two barrier counters, one error flag).  A barrier that does not
 * complete within ~4 ms of wall clock gives up (no hang) and is counted: vt_bn_bwd_fused_timeouts must read 0.
 * Replaces the autograd backward of nn.BatchNorm2d + nn.ReLU (components.py:36-44). */
int vt_bn_act_bwd_fused(const void* dy, int32_t lddy, const void* z, int32_t ldz, const float* scale, const float* shift,
                        const float* mean, const float* invstd, int64_t M, int32_t C, int32_t relu, int32_t dtype,
                        double count, double pscale, int32_t train, float* sums, void* sync, float* dgamma, float* dbeta,
                        float* coef, void* dz, int32_t lddz, void* stream);
int vt_bn_bwd_fused_timeouts(uint32_t* count);
/* vt_bn_finalize + vt_bn_act_apply, and vt_bn_bwd_finalize + vt_bn_act_bwd_apply, as ONE launch each (round 6).  The sums
 * are complete when the streaming launch starts, so nothing is handed over inside it: EVERY workgroup finalizes the (at most
 * 128) channels of its own channel group for itself -- a thread per (channel, sum), the 16 replicas x 2 limbs in one round
 * of loads, the same arithmetic bit for bit -- and keeps the coefficients in LDS; the workgroups of row block 0 also store
 * them, the running statistics and d(gamma) / d(beta).  The single-workgroup finalize launch between a producer of
 * statistics and the pass that needs the coefficients (9 - 13 us of the step each, 134 per CSPDarknet-53 step) disappears;
 * the grid is cut for ~1536 workgroups (knob VT_BN_FIN_APPLY_WGS) so that the redundant reads stay small.  Same arguments,
 * same results (bit-identical) as the two calls each replaces.  Those two calls run for activation codes >= 2, for channel
 * counts without a divisor in [32, 128] that is a multiple of the 16-byte chunk (other than C itself), and with the knob
 * VT_BN_FIN_APPLY = 0.  (The first form of these launches -- the first workgroups finalize and publish, all others poll a
 * counter -- measured 2.3 ms slower per step, NOTEBOOK R6.6.) */
int vt_bn_finalize_apply(const float* stats, int32_t C, double count, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale,
                         float* shift, float* mean, float* invstd, const void* z, int32_t ldz,
                         const void* residual, int32_t ldr, void* y, int32_t ldy, int64_t M, int32_t relu, int32_t dtype,
                         void* stream);
int vt_bn_bwd_finalize_apply(const float* sums, int32_t C, double count, double pscale, const float* scale,
                             const float* shift, const float* mean, const float* invstd, int32_t train, float* dgamma,
                             float* dbeta, float* coef, const void* dy, int32_t lddy, const void* z, int32_t ldz,
                             void* dz, int32_t lddz, int64_t M, int32_t relu, int32_t dtype, void* stream);

/* Backward of the stem unit Conv3x3(3 -> C, s1, pad 1) -> BatchNorm2d -> ReLU in one streaming pass
 * (darknet.py:75 `ConvNormAct(3, 32, 3, 1)`; autograd backward of components.py:26-44 with respect to the
 * conv weight and the BatchNorm parameters -- the unit's input is the image, no data gradient exists).
 * vt_stem_bn_bwd_reduce reads x [B*H*W][8] (3 real channels), dy and z ONCE and accumulates
 *   sums  (a statistics buffer, see VT_STAT_REPLICAS) as vt_bn_act_bwd_reduce, and
 *   gzx   vt_stem_bn_bwd_scratch_bytes(C) bytes: the correlations of g, z and 1 with the tap-shifted x, as f32
 *         (fixed = 0: f32 atomics, the first quarter of the buffer) or as fixed-point int64 pairs (fixed = 1: integer
 *         atomics, order-free -- the deterministic mode)
 * (both zeroed by the caller); after vt_bn_bwd_finalize, vt_stem_bn_bwd_combine (same `fixed`) adds
 *   dw[n][t][c] += coef0[n]*G - coef1[n]*Z + coef2[n]*X,  dw float[C][9][cin]
 * i.e. the filter gradient of dz = coef0*g - coef1*z + coef2 without forming dz.  bf16, C = 32 (the stems of the reference's models). */
int64_t vt_stem_bn_bwd_scratch_bytes(int32_t C);
int vt_stem_bn_bwd_reduce(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* x,
                          const void* dy, int32_t lddy, const void* z, int32_t ldz, const float* scale,
                          const float* shift, const float* mean, const float* invstd, int32_t relu,
                          float* sums, float* gzx, int32_t fixed, void* stream);
int vt_stem_bn_bwd_combine(int32_t C, int32_t cin, const float* gzx, const float* coef, float* dw,
                           int32_t fixed, void* stream);
/* `fixed` bit 1 of vt_stem_bn_bwd_reduce: `z` holds the unit's OUTPUT y = relu(z*scale + shift) (the pre-activation was
 * never stored: forward = a VT_CONV_STATS|VT_CONV_NOSTORE pass + a pass with the affine/ReLU epilogue).  The mask is
 * y > 0; the reduction then leaves `sums` entry 1 (sum g * xhat) ZERO, and vt_stem_bn_bwd_s2 fills it from the
 * correlations before vt_bn_bwd_finalize: z is linear in the 27 im2col patch values, so sum g*z = sum_i w[n][i] G[n][i]
 * exactly -- nothing is recovered from the rounded y.  The 32 z rows of gzx hold the correlations of the patch values
 * with the tap-shifted x instead; vt_stem_bn_bwd_combine_y then forms Z = W P with `w` the bf16 filter image [C][9][8]
 * the forward conv read, and adds dW like vt_stem_bn_bwd_combine.
 * Order: reduce(fixed | 2) -> s2 -> vt_bn_bwd_finalize -> combine_y. */
int vt_stem_bn_bwd_s2(int32_t C, const float* gzx, const void* w, const float* mean, const float* invstd, float* sums,
                      int32_t fixed, void* stream);
int vt_stem_bn_bwd_combine_y(int32_t C, int32_t cin, const float* gzx, const float* coef, const void* w, float* dw,
                             int32_t fixed, void* stream);

/* ---- pointwise (1x1) ConvNormAct unit without materialised pre-activations (vt_pointwise.hip) ---------------
 * A 1x1 `ConvNormAct` (components.py:26-44: nn.Conv2d(k=1, bias=False) -> nn.BatchNorm2d -> nn.ReLU) and its autograd
 * backward as four streaming passes that RECOMPUTE z = W x from the unit's input instead of storing z and dz:
 *   vt_pw_fwd_stats   x            -> batch statistics of z                       (then vt_bn_finalize)
 *   vt_pw_fwd_apply   x (+ res)    -> y = relu(z*scale + shift) (+ residual)
 *   vt_pw_bwd_reduce  dy, x        -> sums of g and g*xhat, g = dy*[y>0]          (then vt_bn_bwd_finalize)
 *   vt_pw_bwd_apply   dy, x (+add) -> dx = W^T dz (+ addend), dW += dz^T x, dz = coef0*g - coef1*z + coef2 in registers
 * z is rounded to bf16 where the unfused path stores it, so values agree with vt_conv_igemm + vt_bn_* to summation
 * order.  Up to two OUTPUT GROUPS share a launch -- CSPDarknetStage.conv1 / conv2 (darknet.py:46-47,53) read the same
 * tensor: one GEMM with N = C[0] + C[1] whose halves have their own filters, BatchNorm coefficients, statistics and
 * destinations.  bf16; K and every C[g] multiples of 32; vt_pw_supported() says whether a shape has a kernel:
 *   0 no;  2 yes;  1 yes, but the filter gradient does not fit the kernel's accumulators: vt_pw_bwd_apply then WRITES dz
 *   (pass `dz`) and the caller runs vt_conv_wgrad on it, `dw` must be NULL.
 * `coef` is float[4][N] = scale | shift | mean | invstd over all N = C[0] + C[1] channels (group 1 after group 0);
 * statistics / sums buffers and the vt_bn_bwd_finalize coefficients `bcoef` (float[3][C[g]]) are per group;
 * residual operands of vt_pw_fwd_apply: one for every group or none.
 * Arrays of per-group values have `ngroups` entries. */
typedef struct vt_pw_desc {
    int32_t dtype; /* VT_BF16 */
    int32_t K;     /* input channels */
    int32_t ngroups;
    int32_t relu;
    int64_t M;     /* pixels (B*H*W) */
    const void* x; /* [M][K], pixel stride ldx */
    int32_t ldx;
    int32_t C[2];       /* output channels per group */
    const void* w[2];   /* filters [C[g]][K], row stride ldw[g] */
    int32_t ldw[2];
} vt_pw_desc;
int vt_pw_supported(int32_t dtype, int32_t K, int32_t C0, int32_t C1);
/* the apply pass ALONE (inference: folded running statistics), one output group: additionally the 80-channel shapes of
 * YOLOv5x's first stage (80 -> 80, 160 -> 80), which are no multiples of 32 (darknet.py:124-133 at width 1.25) */
int vt_pw_apply_supported(int32_t dtype, int32_t K, int32_t C0);
int vt_pw_fwd_stats(const vt_pw_desc* d, float* const* stats, void* stream);
int vt_pw_fwd_apply(const vt_pw_desc* d, const float* coef, void* const* y, const int32_t* ldy,
                    const void* const* res, const int32_t* ldr, void* stream);
int vt_pw_bwd_reduce(const vt_pw_desc* d, const float* coef, const void* const* dy, const int32_t* lddy,
                     float* const* sums, void* stream);
int vt_pw_bwd_apply(const vt_pw_desc* d, const float* coef, const void* const* dy, const int32_t* lddy,
                    const float* const* bcoef, void* dx, int32_t lddx, const void* addend, int32_t ldadd,
                    float* const* dw, const int32_t* lddw, void* const* dz, const int32_t* lddz, void* stream);
/* The two apply passes with the BatchNorm finalize step of every group INSIDE them (round 6; as vt_bn_finalize_apply): a
 * workgroup stages the coefficients in LDS anyway, so every workgroup computes them itself from the complete sums of the pass
 * before, and workgroup 0 stores what the later passes read.  Same values bit for bit as
 *   vt_bn_finalize(fin[g].stats, C[g], fin[g].count, gamma, beta, eps, momentum, running_mean, running_var,
 *                  num_batches_tracked, coef + off_g, coef + N + off_g, coef + 2N + off_g, coef + 3N + off_g) per group, then
 *   vt_pw_fwd_apply(d, coef, ...)                                              (off_g = 0 | C[0], N = C[0] + C[1]);
 *   vt_bn_bwd_finalize(fin[g].sums, C[g], fin[g].count, fin[g].pscale, coef + off_g, coef + 2N + off_g, coef + 3N + off_g,
 *                      fin[g].train, fin[g].dgamma, fin[g].dbeta, bcoef[g]) per group, then vt_pw_bwd_apply(d, coef, ...).
 * Those calls run where N > 128 or with the knob VT_BN_FIN_APPLY = 0.  `fin` has ngroups entries. */
typedef struct vt_bn_fin_fwd {
    const float* stats; /* statistics buffer of the group (vt_pw_fwd_stats) */
    double count;
    const float *gamma, *beta; /* NULL: 1 / 0 */
    float eps, momentum;
    float *running_mean, *running_var; /* both or neither */
    int64_t* num_batches_tracked;      /* optional */
} vt_bn_fin_fwd;
typedef struct vt_bn_fin_bwd {
    const float* sums; /* sums buffer of the group (vt_pw_bwd_reduce) */
    double count, pscale;
    int32_t train;
    float *dgamma, *dbeta; /* optional, accumulated */
} vt_bn_fin_bwd;
int vt_pw_fwd_apply_finalize(const vt_pw_desc* d, const vt_bn_fin_fwd* fin, float* coef, void* const* y, const int32_t* ldy,
                             const void* const* res, const int32_t* ldr, void* stream);
int vt_pw_bwd_apply_finalize(const vt_pw_desc* d, const float* coef, const void* const* dy, const int32_t* lddy,
                             const vt_bn_fin_bwd* fin, float* const* bcoef, void* dx, int32_t lddx, const void* addend,
                             int32_t ldadd, float* const* dw, const int32_t* lddw, void* const* dz, const int32_t* lddz,
                             void* stream);

/* ---- pooling ------------------------------------------------------------ */
/* The normalise pass fused with the MaxPool2d(3, 2, 1) that reads its output (VoVNet: `stage.max_pool` on the previous
 * stage's last ConvNormAct, vovnet.py:94 after components.py:36-44): writes y, the pooled map and the arg-max taps in one
 * pass over z; and the unit's BatchNorm-backward passes reading the POOLED gradient through the arg-max taps instead of
 * a materialised d(y) (only when the pool is y's sole consumer).  Same values as the separate calls. */
int vt_bn_act_apply_pool(const void* z, int32_t ldz, const float* scale, const float* shift, const void* residual,
                         int32_t ldr, void* y, int32_t ldy, void* pooled, int32_t ldp, uint8_t* argmax, int32_t B,
                         int32_t H, int32_t W, int32_t C, int32_t relu, int32_t dtype, void* stream);
int vt_bn_act_bwd_reduce_pool(const void* dp, int32_t lddp, const uint8_t* argmax, const void* z, int32_t ldz,
                              const float* scale, const float* shift, const float* mean, const float* invstd, int32_t B,
                              int32_t H, int32_t W, int32_t C, int32_t relu, int32_t dtype, float* sums, void* stream);
int vt_bn_act_bwd_apply_pool(const void* dp, int32_t lddp, const uint8_t* argmax, const void* z, int32_t ldz,
                             const float* scale, const float* shift, const float* coef, void* dz, int32_t lddz, int32_t B,
                             int32_t H, int32_t W, int32_t C, int32_t relu, int32_t dtype, void* stream);

/* nn.MaxPool2d(3, 2, 1) at the head of every VoVNet stage (vovnet.py:94). */
int vt_maxpool3x3s2_fwd(const void* x, int32_t ldx, void* y, int32_t ldy, uint8_t* argmax,
                        int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                        void* stream);
int vt_maxpool3x3s2_bwd(const void* dy, int32_t lddy, const uint8_t* argmax, void* dx,
                        int32_t lddx, int32_t B, int32_t H, int32_t W, int32_t C,
                        int32_t accumulate, int32_t dtype, void* stream);
/* nn.AdaptiveAvgPool2d((1,1)) of the classifier head (classifier.py:61) and of
 * ESEBlock (vovnet.py:23). y is [B][C] with row stride ldy. */
int vt_global_avgpool_fwd(const void* x, int32_t ldx, void* y, int32_t ldy, int32_t B,
                          int32_t HW, int32_t C, int32_t dtype, void* stream);
int vt_global_avgpool_bwd(const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t B,
                          int32_t HW, int32_t C, int32_t accumulate, int32_t dtype,
                          void* stream);

/* ---- nearest-neighbour resampling of the necks (necks.py:66, 70-81) -------- */
/* The `nn.Upsample(scale_factor=2.0 | 0.5, mode="nearest")` of FPN / PAN fused with the `sum`
 * that follows it (`aggregate_sum`, necks.py:19-23):
 *   mode 0 (top-down):  dst[b][i][j] = src[b][i/2][j/2] (+ other[b][i][j]);  src is [B][Hd/2][Wd/2][C]
 *   mode 1 (bottom-up): dst[b][i][j] = src[b][2i][2j]   (+ other[b][i][j]);  src is [B][2Hd][2Wd][C]
 * dst / other are [B][Hd][Wd][C]; `other` may be NULL.
 * Round 6: `mode="bilinear"` (align_corners False, what nn.Upsample computes):
 *   mode 2 (x2):   destination index d reads source coordinate d/2 - 1/4: taps (k-1, k) with weights (1/4, 3/4) for d = 2k,
 *                  (k, k+1) with (3/4, 1/4) for d = 2k+1, indices clamped to the map, rows and columns alike
 *   mode 3 (x0.5): dst[b][i][j] = mean of src[b][2i..2i+1][2j..2j+1] */
int vt_resample2x_add_fwd(const void* src, int32_t lds, const void* other, int32_t ldo, void* dst,
                          int32_t ldd, int32_t B, int32_t Hd, int32_t Wd, int32_t C, int32_t mode,
                          int32_t dtype, void* stream);
/* gradient w.r.t. src of the above (the `other` branch is the identity):
 *   mode 0: dsrc[b][i][j] (+)= sum of the 2x2 block dy[b][2i..2i+1][2j..2j+1]
 *   mode 1: dsrc[b][i][j] (+)= (i, j both even) ? dy[b][i/2][j/2] : 0
 *   mode 2 / 3: the transposes of the bilinear maps above (mode 3: a quarter of dy[b][i/2][j/2])
 * dy is [B][Hd][Wd][C]. */
int vt_resample2x_bwd(const void* dy, int32_t lddy, void* dsrc, int32_t lds, int32_t B, int32_t Hd,
                      int32_t Wd, int32_t C, int32_t mode, int32_t accumulate, int32_t dtype,
                      void* stream);

/* ---- ESEBlock gate (vovnet.py:20-28) ------------------------------------ */
/* y = x * hardsigmoid(s[b][c]) [+ residual]; s is [B][C] (the biased 1x1 conv
 * of the pooled map, computed with vt_conv_igemm). */
int vt_ese_gate_fwd(const void* x, int32_t ldx, const void* s, int32_t lds,
                    const void* residual, int32_t ldr, void* y, int32_t ldy, int32_t B,
                    int32_t HW, int32_t C, int32_t dtype, void* stream);
/* dx (=|+=) dy*hsig(s);  ds[b][c] = hsig'(s) * sum_hw dy*x  (f32 [B][C]) */
int vt_ese_gate_bwd(const void* dy, int32_t lddy, const void* x, int32_t ldx, const void* s,
                    int32_t lds, void* dx, int32_t lddx, float* ds, int32_t B, int32_t HW,
                    int32_t C, int32_t accumulate, int32_t dtype, void* stream);

/* ---- classifier head / loss (classifier.py:58-64, 92) ------------------- */
/* per-column sums of a [M][C] matrix into out[C] (+=): bias gradients. */
int vt_colsum(const void* a, int32_t lda, int64_t M, int32_t C, int32_t dtype, float* out,
              void* stream);
/* the same sums into a zeroed fixed-point buffer q = int64[C][2] (deterministic mode; vt_fixed_to_f32 folds it in) */
int vt_colsum_fixed(const void* a, int32_t lda, int64_t M, int32_t C, int32_t dtype, void* q, void* stream);
/* F.cross_entropy(logits, labels, label_smoothing) with mean reduction and its
 * gradient times grad_scale; loss_sum[0] += sum_b loss_b / B. */
int vt_softmax_xent(const void* logits, int32_t ldl, const int64_t* labels,
                    float label_smoothing, float grad_scale, float* loss, void* dlogits,
                    int32_t lddl, int32_t B, int32_t N, int32_t dtype, void* stream);

/* MixUp / CutMix of the training step (classifier.py:86-87, extras.py:14-109), on device.
 * `mix` is a device float[8] the host refreshes per step (so captured graphs stay valid):
 *   mix[0] mode: 0 none, 1 mixup, 2 cutmix;  mix[1] lambda (for cutmix: 1 - box area / image area,
 *   extras.py:88);  mix[2..5] = x1, y1, x2, y2 of the cutmix box (extras.py:81-84).
 * The partner of sample b is sample b-1 (mod B): `batch.roll(1, 0)`, extras.py:34,69.
 * vt_softmax_xent_mix: loss / gradient against the soft target
 *   lambda * onehot(labels[b]) + (1 - lambda) * onehot(labels[b-1]), then label smoothing --
 *   what F.cross_entropy(logits, mixed_onehot, label_smoothing) computes (classifier.py:92).
 * vt_mix_nchw_to_nhwc: vt_nchw_to_nhwc of the mixed images (mixup: lambda*x[b] + (1-lambda)*x[b-1];
 *   cutmix: x[b-1] inside the box, x[b] outside). */
int vt_softmax_xent_mix(const void* logits, int32_t ldl, const int64_t* labels,
                        float label_smoothing, float grad_scale, float* loss, void* dlogits,
                        int32_t lddl, int32_t B, int32_t N, int32_t dtype, const float* mix,
                        void* stream);
/* Validation step (classifier.py:97-109): out3[0] += sum of the rows' cross entropy WITHOUT label smoothing, out3[1] +=
 * rows whose arg-max (first maximum, as torch.argmax) equals the label, out3[2] += rows.  The caller zeroes out3 or keeps
 * accumulating over an epoch's batches; data parallel it all-reduces the three sums (`sync_dist=True`, classifier.py:104). */
int vt_softmax_xent_eval(const void* logits, int32_t ldl, const int64_t* labels, float* out3, int32_t B, int32_t N,
                         int32_t dtype, void* stream);
int vt_mix_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W,
                        int32_t Cpad, int32_t dtype, const float* mix, void* stream);

/* ---- optimiser (classifier.py:161-169: torch.optim.SGD, momentum) -------- */
/* g' = g*grad_scale + wd*p; m = mu*m + g'; p -= lr*m; mirror = cast(p).
 * lr_dev (optional, device float[1]) overrides `lr`, so a captured graph can
 * follow the warm-up/cosine schedule (classifier.py:171-190) without re-capture. */
int vt_sgd_momentum(float* p, const float* g, float* m, void* mirror, int32_t mirror_dtype,
                    int64_t n, float lr, float momentum, float weight_decay,
                    float grad_scale, const float* lr_dev, void* stream);

/* ---- layout / precision plumbing ---------------------------------------- */
/* rows x cols block copy with dtype conversion: dst (=|+=) src */
int vt_copy2d(const void* src, int32_t src_dtype, int64_t lds, void* dst, int32_t dst_dtype,
              int64_t ldd, int64_t rows, int32_t cols, int32_t accumulate, void* stream);
/* images: NCHW f32 -> NHWC dtype with channels zero-padded to Cpad
 * (the `images.to(memory_format=channels_last)` of classifier.py:88-89). */
int vt_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W,
                    int32_t Cpad, int32_t dtype, void* stream);
/* dx[B][C][H][W] f32 (=) dy NHWC[..., :C] */
int vt_nhwc_to_nchw(const void* y, int32_t ldy, float* x, int32_t B, int32_t C, int32_t H,
                    int32_t W, int32_t dtype, void* stream);

/* ---- native executor -----------------------------------------------------
 * A step of the backbone is a static list of the calls above.  The host builds
 * the list once per (model, input shape) and replays it with one call, or as a
 * captured hipGraph.  Pointers are (base, byte offset) pairs so one list serves
 * any arena placement. */
#define VT_OP_MAX_PTR 24
#define VT_OP_MAX_INT 110
#define VT_OP_MAX_FLT 8
#define VT_MAX_BASES 16

enum vt_op_kind {
    VT_OP_MEMSET = 1,
    VT_OP_CONV_IGEMM,
    VT_OP_CONV_WGRAD,
    VT_OP_PACK_DGRAD,
    VT_OP_BN_FINALIZE,
    VT_OP_BN_EVAL_COEFFS,
    VT_OP_BN_ACT_APPLY,
    VT_OP_BN_BWD_REDUCE,
    VT_OP_BN_BWD_FINALIZE,
    VT_OP_BN_BWD_APPLY,
    VT_OP_MAXPOOL_FWD,
    VT_OP_MAXPOOL_BWD,
    VT_OP_AVGPOOL_FWD,
    VT_OP_AVGPOOL_BWD,
    VT_OP_ESE_FWD,
    VT_OP_ESE_BWD,
    VT_OP_COLSUM,
    VT_OP_XENT,
    VT_OP_SGD,
    VT_OP_COPY2D,
    VT_OP_NCHW_TO_NHWC,
    VT_OP_NHWC_TO_NCHW,
    VT_OP_FORK, /* side stream waits for everything enqueued on the main stream so far */
    VT_OP_JOIN, /* main stream waits for everything enqueued on the side stream so far */
    VT_OP_RESAMPLE_FWD,
    VT_OP_RESAMPLE_BWD,
    VT_OP_FORK_MARK, /* remember the main stream's position (an event record), nothing waits yet */
    VT_OP_FORK_WAIT, /* side stream waits for the last FORK_MARK: FORK split in two, so that the host can enqueue main-stream
                        work between the mark and the side-stream ops that depend on it */
    VT_OP_STEM_BWD_REDUCE,  /* vt_stem_bn_bwd_reduce */
    VT_OP_STEM_BWD_COMBINE, /* vt_stem_bn_bwd_combine */
    VT_OP_FIXED_TO_F32,     /* vt_fixed_to_f32 */
    VT_OP_PW_STATS,         /* vt_pw_fwd_stats */
    VT_OP_PW_APPLY,         /* vt_pw_fwd_apply */
    VT_OP_PW_REDUCE,        /* vt_pw_bwd_reduce */
    VT_OP_PW_BWD,           /* vt_pw_bwd_apply */
    VT_OP_STEM_BWD_S2,      /* vt_stem_bn_bwd_s2 */
    VT_OP_ALLREDUCE,        /* vt_allreduce_bucket (a gradient bucket, in place) */
    VT_OP_STAT_SYNC,        /* vt_stat_sync (one BatchNorm layer's sums over all ranks) */
    VT_OP_XENT_EVAL,        /* vt_softmax_xent_eval (validation: loss sum, top-1 hits, rows) */
    VT_OP_CONV_DGRAD_BNRED, /* vt_conv_dgrad_bnred (a data gradient + the BatchNorm-backward sums of the producing unit) */
    VT_OP_BN_BWD_FUSED,     /* vt_bn_act_bwd_fused (BatchNorm backward of a unit: reduce, finalize, apply in one launch) */
    VT_OP_BN_FIN_APPLY,     /* vt_bn_finalize_apply */
    VT_OP_BN_BWD_FIN_APPLY, /* vt_bn_bwd_finalize_apply */
    VT_OP_DWCONV_FWD,       /* vt_dwconv_fwd */
    VT_OP_DWCONV_DGRAD,     /* vt_dwconv_dgrad */
    VT_OP_DWCONV_WGRAD,     /* vt_dwconv_wgrad */
    VT_OP_PW_APPLY_FIN,      /* vt_pw_fwd_apply_finalize */
    VT_OP_PW_BWD_FIN,        /* vt_pw_bwd_apply_finalize */
    VT_OP_KIND_END
};

/* OR-ed into vt_op.kind: enqueue this op on the side stream.  Filter gradients depend only
 * on a layer's input and dz, nothing downstream waits for them before the optimiser, so they
 * run beside the (HBM-bound) rest of backward instead of in line with it. */
#define VT_OP_SIDE_STREAM 0x10000

typedef struct vt_ptr {
    int32_t base; /* index into the bases[] passed at run time; -1 = NULL */
    int32_t pad;
    int64_t offset; /* bytes */
} vt_ptr;

typedef struct vt_op {
    int32_t kind;
    int32_t tag; /* host-defined id (layer index), echoed in error messages */
    vt_ptr ptr[VT_OP_MAX_PTR];
    int32_t i[VT_OP_MAX_INT];
    double f[VT_OP_MAX_FLT];
} vt_op;

/* run ops[0..n) in order on `stream`; argument order per kind is documented in
 * vt_runtime.hip next to each case. */
int vt_run_ops(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases, void* stream);
/* same, with a side stream for ops flagged VT_OP_SIDE_STREAM (FORK / JOIN order the two).
 * side == NULL or side == stream runs everything in line. */
int vt_run_ops_streams(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases, void* stream,
                       void* side);
/* same, for a list that is run in SEGMENTS (the data-parallel trainer issues a gradient bucket's collective between
 * two segments, reference configs/base.yaml:17-19 = DDP's per-bucket hooks): with VT_RUN_LEAVE_SIDE_OPEN the call
 * does not order `stream` behind the side stream on return -- the caller joins them itself (a later segment's JOIN
 * op, or vt_stream_wait) -- so cutting a list does not serialise the filter-gradient stream. */
#define VT_RUN_LEAVE_SIDE_OPEN 1
int vt_run_ops_streams_ex(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases, void* stream,
                          void* side, int32_t flags);
/* order `waiter` behind everything enqueued on `signaller` so far (a pooled event; no host synchronisation) */
int vt_stream_wait(void* waiter, void* signaller);

/* hipGraph capture of an op list (side-stream ops become parallel graph branches): capture
 * once, replay with one launch. */
int vt_graph_create(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases,
                    void** graph_out);
int vt_graph_launch(void* graph, void* stream);
int vt_graph_destroy(void* graph);

/* ---- data-parallel collectives (RCCL over xGMI, bound at run time) ------------------------------------------------
 * The reference leaves these to DistributedDataParallel / SyncBatchNorm (`strategy: ddp`, `sync_batchnorm: true`,
 * configs/base.yaml:17-22).  Here they are stream-ordered entry points, so a launch list carries them as ops
 * (VT_OP_ALLREDUCE, VT_OP_STAT_SYNC) between the kernels that produce and consume their operands.  One communicator per
 * process (= per GPU): rank 0 draws an id, the host side hands it to every rank (any out-of-band channel), every rank
 * calls vt_comm_init with the device current.  The RCCL image already mapped into the process is used when there is
 * one (PyTorch's), else librccl.so.1 is loaded. */
#define VT_COMM_ID_BYTES 128
int vt_comm_unique_id(void* id128);
int vt_comm_init(const void* id128, int32_t rank, int32_t world);
/* Optional second communicator over the same ranks (a second id, drawn and carried like the first), used by
 * vt_stat_sync only: RCCL serialises the operations of ONE communicator in issue order across streams, so the per-layer
 * statistics exchanges (main stream) would otherwise queue behind the bucket all-reduces (filter-gradient stream). */
int vt_comm_init_stat(const void* id128);
int vt_comm_has_stat(void);
int vt_comm_world(void); /* ranks of this process's communicator, 0 without one */
int vt_comm_destroy(void);
/* in-place sum all-reduce of `count` elements (VT_F32 / VT_BF16 gradients, VT_I64 fixed-point sums) on `stream` */
int vt_allreduce_bucket(void* buf, int64_t count, int32_t dtype, void* stream);
/* SyncBatchNorm exchange of one layer: vt_stat_fold, then the int64 all-reduce of the folded 32*C bytes */
int vt_stat_sync(float* stats, int32_t C, void* stream);

/* ---- timing helpers for bench.py (HIP events on the launch stream) ------- */
int vt_event_create(void** ev);
int vt_event_record(void* ev, void* stream);
int vt_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
int vt_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* VT_AMD_H */
