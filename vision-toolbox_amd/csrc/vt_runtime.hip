// vt_runtime.hip -- library state (error text, launch counter), the native
// op-list executor and hipGraph capture/replay, and HIP-event timing helpers.
//
// The executor is what replaces the reference's per-op Python dispatch: the
// reference walks nn.Sequential in Python and dispatches 3 ATen ops per
// ConvNormAct (vision_toolbox/components.py:26-44, backbones/darknet.py:83-87).
// Here the host builds the launch list once per (model, input shape) and a
// step is ONE call into vt_run_ops / vt_graph_launch.
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "vt_common.h"

static thread_local char g_err[512] = "";
static std::atomic<uint64_t> g_launches{0};

void vt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void vt_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

static thread_local char g_kernel[128] = "";
void vt_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

// Function attributes are per device and per kernel: remember what has been raised for each
// (kernel, device) pair; safe from any thread.
int vt_raise_dynamic_lds(const void* kern, int bytes, const char* who) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> raised;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    int& have = raised[std::make_pair(kern, dev)];
    if (have >= bytes) return VT_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        vt_set_error("%s: cannot raise dynamic LDS to %d: %s", who, bytes, hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    have = bytes;
    return VT_OK;
}

// ---- dispatcher switches -------------------------------------------------------------------------------
namespace {
struct Knob {
    char name[48];
    int value;
};
constexpr int kMaxKnobs = 96;
Knob g_knobs[kMaxKnobs];  // slots never move: call sites keep pointers into this table
int g_nknobs = 0;
std::mutex g_knob_mu;
int* knob_find_or_add(const char* name, int dflt, bool from_env) {
    std::lock_guard<std::mutex> lock(g_knob_mu);
    for (int i = 0; i < g_nknobs; ++i)
        if (strcmp(g_knobs[i].name, name) == 0) return &g_knobs[i].value;
    if (g_nknobs == kMaxKnobs) return nullptr;
    Knob& k = g_knobs[g_nknobs++];
    snprintf(k.name, sizeof(k.name), "%s", name);
    const char* e = from_env ? getenv(name) : nullptr;
    k.value = e ? atoi(e) : dflt;
    return &k.value;
}
}  // namespace

int* vt_knob_slot(const char* name, int dflt) {
    static int overflow = 0;
    int* s = knob_find_or_add(name, dflt, true);
    if (!s) {
        overflow = dflt;
        return &overflow;
    }
    return s;
}

int vt_device_cus(void) {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = -1;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

thread_local hipEvent_t vt_pending_stop_event = nullptr;

namespace {

inline void* rp(const vt_op& op, int k, void* const* bases, int nbases, bool* bad) {
    const vt_ptr& p = op.ptr[k];
    if (p.base < 0) return nullptr;
    if (p.base >= nbases || bases[p.base] == nullptr) {
        *bad = true;
        return nullptr;
    }
    return (char*)bases[p.base] + p.offset;
}

#ifdef VT_DIAG_SKIP_FIN
// the first VT_DIAG_SKIP_FIN_AFTER finalize ops run (the coefficient rows hold real values), the later ones are skipped
static bool vt_diag_skip_fin(int which) {  // which: 1 forward, 2 backward finalize; VT_DIAG_SKIP_FIN_WHICH: mask of the kinds skipped
    static long calls = 0;
    static const long after = getenv("VT_DIAG_SKIP_FIN_AFTER") ? atol(getenv("VT_DIAG_SKIP_FIN_AFTER")) : (1L << 60);
    static const int mask = getenv("VT_DIAG_SKIP_FIN_WHICH") ? atoi(getenv("VT_DIAG_SKIP_FIN_WHICH")) : 3;
    return ++calls > after && (mask & which);
}
// any op kinds (VT_DIAG_SKIP_KINDS = comma-separated kind numbers) skipped after the first VT_DIAG_SKIP_AFTER_OPS ops of the
// process: what a kernel family costs INSIDE the step (tools/runs/r6_skipkinds.sh; the skipped ops' outputs keep the values of the
// warm-up steps -- same batch, nearly the same weights)
static bool vt_diag_skip_kind(int kind) {
    static long calls = 0;
    static const long after = getenv("VT_DIAG_SKIP_AFTER_OPS") ? atol(getenv("VT_DIAG_SKIP_AFTER_OPS")) : (1L << 60);
    static bool skip[256];
    static const bool init = []() {
        const char* e = getenv("VT_DIAG_SKIP_KINDS");
        while (e && *e) {
            const int k = atoi(e);
            if (k > 0 && k < 256) skip[k] = true;
            e = strchr(e, ',');
            if (e) ++e;
        }
        return true;
    }();
    (void)init;
    return ++calls > after && kind > 0 && kind < 256 && skip[kind];
}
#endif
int run_one(const vt_op& op, void* const* bases, int nbases, void* st) {
    bool bad = false;
    void* P[VT_OP_MAX_PTR];
    for (int k = 0; k < VT_OP_MAX_PTR; ++k) P[k] = rp(op, k, bases, nbases, &bad);
    if (bad) {
        vt_set_error("vt_run_ops: op kind %d (tag %d) references an unbound base", op.kind, op.tag);
        return VT_ERR_INVALID;
    }
    const int32_t* I = op.i;
    const double* F = op.f;
    switch (op.kind) {
        case VT_OP_MEMSET:  // ptr: dst | i: value | f: bytes
            return vt_memset(P[0], I[0], (uint64_t)F[0], st);
        case VT_OP_CONV_IGEMM: {  // ptr: x w y scale shift residual stats | i: vt_conv_desc image
            vt_conv_desc d;
            memcpy(&d, I, sizeof(d));
            return vt_conv_igemm(&d, P[0], P[1], P[2], (const float*)P[3], (const float*)P[4], P[5],
                                 (float*)P[6], st);
        }
        case VT_OP_CONV_DGRAD_BNRED: {  // ptr: dz w dy z scale shift mean invstd sums | i: vt_conv_desc image, then ldz, relu
            vt_conv_desc d;
            memcpy(&d, I, sizeof(d));
            return vt_conv_dgrad_bnred(&d, P[0], P[1], P[2], P[3], I[sizeof(d) / 4], (const float*)P[4], (const float*)P[5],
                                       (const float*)P[6], (const float*)P[7], I[sizeof(d) / 4 + 1], (float*)P[8], st);
        }
        case VT_OP_CONV_WGRAD: {  // ptr: x dz dw [scratch] | i: vt_conv_desc image, then ldgw, scratch MiB
            vt_conv_desc d;
            memcpy(&d, I, sizeof(d));
            if (P[3])
                return vt_conv_wgrad_slabs(&d, P[0], P[1], (float*)P[2], I[sizeof(d) / 4], P[3],
                                           (int64_t)I[sizeof(d) / 4 + 1] << 20, st);
            return vt_conv_wgrad(&d, P[0], P[1], (float*)P[2], I[sizeof(d) / 4], st);
        }
        case VT_OP_PACK_DGRAD:  // ptr: w out | i: src_dtype ldw dst_dtype nsel Cout ntaps Cin _ sel[36]
            return vt_pack_dgrad_filter(P[0], I[0], I[1], P[1], I[2], I + 8, I[3], I[4], I[5], I[6], st);
        case VT_OP_BN_FINALIZE:  // ptr: stats gamma beta rm rv nbt scale shift mean invstd | i: C | f: count eps momentum
#ifdef VT_DIAG_SKIP_FIN  // diagnostic builds (tools/runs/r6_skipfin.sh): what the 134 finalize launches cost inside the step
            if (vt_diag_skip_fin(1)) return VT_OK;
#endif
            return vt_bn_finalize((const float*)P[0], I[0], F[0], (const float*)P[1], (const float*)P[2],
                                  (float)F[1], (float)F[2], (float*)P[3], (float*)P[4], (int64_t*)P[5],
                                  (float*)P[6], (float*)P[7], (float*)P[8], (float*)P[9], st);
        case VT_OP_BN_EVAL_COEFFS:  // ptr: gamma beta rm rv scale shift mean invstd | i: C | f: eps
            return vt_bn_eval_coeffs((const float*)P[0], (const float*)P[1], (const float*)P[2],
                                     (const float*)P[3], (float)F[0], I[0], (float*)P[4], (float*)P[5],
                                     (float*)P[6], (float*)P[7], st);
        case VT_OP_BN_ACT_APPLY:  // ptr: z scale shift residual y [pooled argmax] | i: ldz ldr ldy C relu dtype [ldp B H W] | f: M
            if (P[5])  // fused with the max-pool that reads y
                return vt_bn_act_apply_pool(P[0], I[0], (const float*)P[1], (const float*)P[2], P[3], I[1], P[4], I[2], P[5], I[6],
                                            (uint8_t*)P[6], I[7], I[8], I[9], I[3], I[4], I[5], st);
            return vt_bn_act_apply(P[0], I[0], (const float*)P[1], (const float*)P[2], P[3], I[1], P[4], I[2],
                                   (int64_t)F[0], I[3], I[4], I[5], st);
        case VT_OP_BN_BWD_REDUCE:  // ptr: dy z scale shift mean invstd sums [argmax] | i: lddy ldz C relu dtype [B H W] | f: M
            if (P[7])  // dy = the gradient of the POOLED output, gathered through the arg-max taps
                return vt_bn_act_bwd_reduce_pool(P[0], I[0], (const uint8_t*)P[7], P[1], I[1], (const float*)P[2], (const float*)P[3],
                                                 (const float*)P[4], (const float*)P[5], I[5], I[6], I[7], I[2], I[3], I[4],
                                                 (float*)P[6], st);
            return vt_bn_act_bwd_reduce(P[0], I[0], P[1], I[1], (const float*)P[2], (const float*)P[3],
                                        (const float*)P[4], (const float*)P[5], (int64_t)F[0], I[2], I[3],
                                        I[4], (float*)P[6], st);
        case VT_OP_BN_BWD_FINALIZE:  // ptr: sums scale mean invstd dgamma dbeta coef | i: C train | f: count pscale(0 = 1)
#ifdef VT_DIAG_SKIP_FIN
            if (vt_diag_skip_fin(2)) return VT_OK;
#endif
            return vt_bn_bwd_finalize((const float*)P[0], I[0], F[0], F[1] == 0.0 ? 1.0 : F[1], (const float*)P[1], (const float*)P[2],
                                      (const float*)P[3], I[1], (float*)P[4], (float*)P[5], (float*)P[6], st);
        // depthwise convolution: i: ld0 ld1 [ldr] B Hi Wi C k s pad dil dtype
        case VT_OP_DWCONV_FWD:  // ptr: x w z stats | i: ldx ldz _ B Hi Wi C k s pad dil dtype
            return vt_dwconv_fwd(P[0], I[0], (const float*)P[1], P[2], I[1], (float*)P[3], I[3], I[4], I[5], I[6], I[7], I[8], I[9],
                                 I[10], I[11], st);
        case VT_OP_DWCONV_DGRAD:  // ptr: dz w dx residual | i: lddz lddx ldr B Hi Wi C k s pad dil dtype
            return vt_dwconv_dgrad(P[0], I[0], (const float*)P[1], P[2], I[1], P[3], I[2], I[3], I[4], I[5], I[6], I[7], I[8], I[9],
                                   I[10], I[11], st);
        case VT_OP_DWCONV_WGRAD:  // ptr: x dz dw | i: ldx lddz _ B Hi Wi C k s pad dil dtype
            return vt_dwconv_wgrad(P[0], I[0], P[1], I[1], (float*)P[2], I[3], I[4], I[5], I[6], I[7], I[8], I[9], I[10], I[11], st);
        case VT_OP_BN_FIN_APPLY:  // ptr: stats gamma beta rm rv nbt scale shift mean invstd z residual y | i: C ldz ldr ldy relu dtype | f: count eps momentum M
            return vt_bn_finalize_apply((const float*)P[0], I[0], F[0], (const float*)P[1], (const float*)P[2], (float)F[1], (float)F[2],
                                        (float*)P[3], (float*)P[4], (int64_t*)P[5], (float*)P[6], (float*)P[7], (float*)P[8],
                                        (float*)P[9], P[10], I[1], P[11], I[2], P[12], I[3], (int64_t)F[3], I[4], I[5], st);
        case VT_OP_BN_BWD_FIN_APPLY:  // ptr: sums scale shift mean invstd dgamma dbeta coef dy z dz | i: C train lddy ldz lddz relu dtype | f: count pscale(0 = 1) M
            return vt_bn_bwd_finalize_apply((const float*)P[0], I[0], F[0], F[1] == 0.0 ? 1.0 : F[1], (const float*)P[1],
                                            (const float*)P[2], (const float*)P[3], (const float*)P[4], I[1], (float*)P[5],
                                            (float*)P[6], (float*)P[7], P[8], I[2], P[9], I[3], P[10], I[4], (int64_t)F[2],
                                            I[5], I[6], st);
        case VT_OP_BN_BWD_FUSED:  // ptr: dy z scale shift mean invstd sums sync dgamma dbeta coef dz | i: lddy ldz lddz C relu dtype train | f: M count pscale(0 = 1)
            return vt_bn_act_bwd_fused(P[0], I[0], P[1], I[1], (const float*)P[2], (const float*)P[3], (const float*)P[4],
                                       (const float*)P[5], (int64_t)F[0], I[3], I[4], I[5], F[1], F[2] == 0.0 ? 1.0 : F[2], I[6],
                                       (float*)P[6], P[7], (float*)P[8], (float*)P[9], (float*)P[10], P[11], I[2], st);
        case VT_OP_BN_BWD_APPLY:  // ptr: dy z scale shift coef dz [argmax] | i: lddy ldz lddz C relu dtype [B H W] | f: M
            if (P[6])
                return vt_bn_act_bwd_apply_pool(P[0], I[0], (const uint8_t*)P[6], P[1], I[1], (const float*)P[2], (const float*)P[3],
                                                (const float*)P[4], P[5], I[2], I[6], I[7], I[8], I[3], I[4], I[5], st);
            return vt_bn_act_bwd_apply(P[0], I[0], P[1], I[1], (const float*)P[2], (const float*)P[3],
                                       (const float*)P[4], P[5], I[2], (int64_t)F[0], I[3], I[4], I[5], st);
        case VT_OP_STEM_BWD_REDUCE:  // ptr: x dy z scale shift mean invstd sums gzx | i: dtype B H W C lddy ldz relu fixed
            return vt_stem_bn_bwd_reduce(I[0], I[1], I[2], I[3], I[4], P[0], P[1], I[5], P[2], I[6], (const float*)P[3],
                                         (const float*)P[4], (const float*)P[5], (const float*)P[6], I[7], (float*)P[7],
                                         (float*)P[8], I[8], st);
        case VT_OP_STEM_BWD_S2:  // ptr: gzx w mean invstd sums | i: C fixed
            return vt_stem_bn_bwd_s2(I[0], (const float*)P[0], P[1], (const float*)P[2], (const float*)P[3], (float*)P[4], I[1],
                                     st);
        case VT_OP_ALLREDUCE:  // ptr: buf | i: dtype | f: count
            return vt_allreduce_bucket(P[0], (int64_t)F[0], I[0], st);
        case VT_OP_STAT_SYNC:  // ptr: stats | i: C
            return vt_stat_sync((float*)P[0], I[0], st);
        case VT_OP_STEM_BWD_COMBINE:  // ptr: gzx coef dw [w: the reduction read y, not z] | i: C cin fixed
            if (P[3])
                return vt_stem_bn_bwd_combine_y(I[0], I[1], (const float*)P[0], (const float*)P[1], P[3], (float*)P[2], I[2],
                                                st);
            return vt_stem_bn_bwd_combine(I[0], I[1], (const float*)P[0], (const float*)P[1], (float*)P[2], I[2], st);
        case VT_OP_MAXPOOL_FWD:  // ptr: x y argmax | i: ldx ldy B H W C dtype
            return vt_maxpool3x3s2_fwd(P[0], I[0], P[1], I[1], (uint8_t*)P[2], I[2], I[3], I[4], I[5], I[6], st);
        case VT_OP_MAXPOOL_BWD:  // ptr: dy argmax dx | i: lddy lddx B H W C accumulate dtype
            return vt_maxpool3x3s2_bwd(P[0], I[0], (const uint8_t*)P[1], P[2], I[1], I[2], I[3], I[4], I[5],
                                       I[6], I[7], st);
        case VT_OP_RESAMPLE_FWD:  // ptr: src other dst | i: lds ldo ldd B Hd Wd C mode dtype
            return vt_resample2x_add_fwd(P[0], I[0], P[1], I[1], P[2], I[2], I[3], I[4], I[5], I[6], I[7], I[8], st);
        case VT_OP_RESAMPLE_BWD:  // ptr: dy dsrc | i: lddy lds B Hd Wd C mode accumulate dtype
            return vt_resample2x_bwd(P[0], I[0], P[1], I[1], I[2], I[3], I[4], I[5], I[6], I[7], I[8], st);
        case VT_OP_AVGPOOL_FWD:  // ptr: x y | i: ldx ldy B HW C dtype
            return vt_global_avgpool_fwd(P[0], I[0], P[1], I[1], I[2], I[3], I[4], I[5], st);
        case VT_OP_AVGPOOL_BWD:  // ptr: dy dx | i: lddy lddx B HW C accumulate dtype
            return vt_global_avgpool_bwd(P[0], I[0], P[1], I[1], I[2], I[3], I[4], I[5], I[6], st);
        case VT_OP_ESE_FWD:  // ptr: x s residual y | i: ldx lds ldr ldy B HW C dtype
            return vt_ese_gate_fwd(P[0], I[0], P[1], I[1], P[2], I[2], P[3], I[3], I[4], I[5], I[6], I[7], st);
        case VT_OP_ESE_BWD:  // ptr: dy x s dx ds | i: lddy ldx lds lddx B HW C accumulate dtype
            return vt_ese_gate_bwd(P[0], I[0], P[1], I[1], P[2], I[2], P[3], I[3], (float*)P[4], I[4], I[5],
                                   I[6], I[7], I[8], st);
        case VT_OP_COLSUM:  // ptr: a out | i: lda C dtype fixed | f: M
            return I[3] ? vt_colsum_fixed(P[0], I[0], (int64_t)F[0], I[1], I[2], P[1], st)
                        : vt_colsum(P[0], I[0], (int64_t)F[0], I[1], I[2], (float*)P[1], st);
        case VT_OP_FIXED_TO_F32:  // ptr: q dst | i: accumulate | f: n
            return vt_fixed_to_f32(P[0], (float*)P[1], (int64_t)F[0], I[0], st);
        case VT_OP_XENT_EVAL:  // ptr: logits labels out3 | i: ldl B N dtype
            return vt_softmax_xent_eval(P[0], I[0], (const int64_t*)P[1], (float*)P[2], I[1], I[2], I[3], st);
        case VT_OP_XENT:  // ptr: logits labels loss dlogits [mix] | i: ldl lddl B N dtype | f: eps grad_scale
            if (P[4])
                return vt_softmax_xent_mix(P[0], I[0], (const int64_t*)P[1], (float)F[0], (float)F[1], (float*)P[2],
                                           P[3], I[1], I[2], I[3], I[4], (const float*)P[4], st);
            return vt_softmax_xent(P[0], I[0], (const int64_t*)P[1], (float)F[0], (float)F[1], (float*)P[2],
                                   P[3], I[1], I[2], I[3], I[4], st);
        case VT_OP_SGD:  // ptr: p g m mirror lr_dev | i: mirror_dtype | f: n lr momentum wd grad_scale
            return vt_sgd_momentum((float*)P[0], (const float*)P[1], (float*)P[2], P[3], I[0], (int64_t)F[0],
                                   (float)F[1], (float)F[2], (float)F[3], (float)F[4], (const float*)P[4], st);
        case VT_OP_COPY2D:  // ptr: src dst | i: src_dtype dst_dtype cols accumulate | f: lds ldd rows
            return vt_copy2d(P[0], I[0], (int64_t)F[0], P[1], I[1], (int64_t)F[1], (int64_t)F[2], I[2], I[3], st);
        case VT_OP_NCHW_TO_NHWC:  // ptr: x y [mix] | i: B C H W Cpad dtype
            if (P[2])
                return vt_mix_nchw_to_nhwc((const float*)P[0], P[1], I[0], I[1], I[2], I[3], I[4], I[5],
                                           (const float*)P[2], st);
            return vt_nchw_to_nhwc((const float*)P[0], P[1], I[0], I[1], I[2], I[3], I[4], I[5], st);
        case VT_OP_NHWC_TO_NCHW:  // ptr: y x | i: ldy B C H W dtype
            return vt_nhwc_to_nchw(P[0], I[0], (float*)P[1], I[1], I[2], I[3], I[4], I[5], st);
        // pointwise units: i: K ngroups relu C0 C1 ldx ldw0 ldw1 + per-kind strides | f: M | ptr 0..2: x w0 w1
        case VT_OP_PW_STATS:    // ptr: x w0 w1 stats0 stats1
        case VT_OP_PW_APPLY:    // ptr: x w0 w1 coef y0 y1 res0 res1 | i[8..11]: ldy0 ldy1 ldr0 ldr1
        case VT_OP_PW_REDUCE:   // ptr: x w0 w1 coef dy0 dy1 sums0 sums1 | i[8..9]: lddy0 lddy1
        case VT_OP_PW_BWD:      // ptr: x w0 w1 coef dy0 dy1 bcoef0 bcoef1 dx addend dw0 dw1 dz0 dz1
                                // i[8..]: lddy0 lddy1 lddx ldadd lddw0 lddw1 lddz0 lddz1
        // ... with the BatchNorm finalize step of every group inside the pass:
        case VT_OP_PW_APPLY_FIN:  // as PW_APPLY + ptr[8..]: (stats gamma beta rm rv nbt) x 2 | f: M count eps0 momentum0 eps1 momentum1
        case VT_OP_PW_BWD_FIN: {  // as PW_BWD + ptr[14..]: (sums dgamma dbeta) x 2 | i[16]: train | f: M count pscale(0 = 1)
            vt_pw_desc d;
            memset(&d, 0, sizeof(d));
            d.dtype = VT_BF16;
            d.K = I[0], d.ngroups = I[1], d.relu = I[2], d.C[0] = I[3], d.C[1] = I[4];
            d.M = (int64_t)F[0];
            d.x = P[0], d.ldx = I[5];
            d.w[0] = P[1], d.w[1] = P[2], d.ldw[0] = I[6], d.ldw[1] = I[7];
            if (op.kind == VT_OP_PW_STATS) {
                float* st2[2] = {(float*)P[3], (float*)P[4]};
                return vt_pw_fwd_stats(&d, st2, st);
            }
            if (op.kind == VT_OP_PW_APPLY) {
                void* y2[2] = {P[4], P[5]};
                const void* r2[2] = {P[6], P[7]};
                const int32_t ldy2[2] = {I[8], I[9]}, ldr2[2] = {I[10], I[11]};
                return vt_pw_fwd_apply(&d, (const float*)P[3], y2, ldy2, r2, ldr2, st);
            }
            if (op.kind == VT_OP_PW_APPLY_FIN) {
                void* y2[2] = {P[4], P[5]};
                const void* r2[2] = {P[6], P[7]};
                const int32_t ldy2[2] = {I[8], I[9]}, ldr2[2] = {I[10], I[11]};
                vt_bn_fin_fwd fin[2];
                for (int g = 0; g < 2; ++g)
                    fin[g] = vt_bn_fin_fwd{(const float*)P[8 + 6 * g], F[1], (const float*)P[9 + 6 * g], (const float*)P[10 + 6 * g],
                                           (float)F[2 + 2 * g], (float)F[3 + 2 * g], (float*)P[11 + 6 * g], (float*)P[12 + 6 * g],
                                           (int64_t*)P[13 + 6 * g]};
                return vt_pw_fwd_apply_finalize(&d, fin, (float*)P[3], y2, ldy2, r2, ldr2, st);
            }
            const void* dy2[2] = {P[4], P[5]};
            const int32_t lddy2[2] = {I[8], I[9]};
            if (op.kind == VT_OP_PW_REDUCE) {
                float* s2[2] = {(float*)P[6], (float*)P[7]};
                return vt_pw_bwd_reduce(&d, (const float*)P[3], dy2, lddy2, s2, st);
            }
            const float* bc2[2] = {(const float*)P[6], (const float*)P[7]};
            float* dw2[2] = {(float*)P[10], (float*)P[11]};
            void* dz2[2] = {P[12], P[13]};
            const int32_t lddw2[2] = {I[12], I[13]}, lddz2[2] = {I[14], I[15]};
            if (op.kind == VT_OP_PW_BWD_FIN) {
                vt_bn_fin_bwd fin[2];
                for (int g = 0; g < 2; ++g)
                    fin[g] = vt_bn_fin_bwd{(const float*)P[14 + 3 * g], F[1], F[2] == 0.0 ? 1.0 : F[2], I[16], (float*)P[15 + 3 * g],
                                           (float*)P[16 + 3 * g]};
                return vt_pw_bwd_apply_finalize(&d, (const float*)P[3], dy2, lddy2, fin, (float* const*)bc2, P[8], I[10], P[9], I[11],
                                                dw2, lddw2, dz2, lddz2, st);
            }
            return vt_pw_bwd_apply(&d, (const float*)P[3], dy2, lddy2, bc2, P[8], I[10], P[9], I[11], dw2, lddw2, dz2, lddz2, st);
        }
        default:
            vt_set_error("vt_run_ops: unknown op kind %d (tag %d)", op.kind, op.tag);
            return VT_ERR_INVALID;
    }
}

struct Graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

}  // namespace

extern "C" {

int vt_version(void) { return 104; }  // 104: round 6 (finalize inside the passes, vt_op carries 24 pointers; 103: vt_conv_dgrad_bnred, vt_debug_hog left the library)
int vt_set_knob(const char* name, int32_t value) {
    VT_REQUIRE(name && strlen(name) < 48, VT_ERR_INVALID, "vt_set_knob: bad name");
    // (a knob set before its first use overrides the environment: the slot exists from here on)
    int* s = knob_find_or_add(name, value, false);
    VT_REQUIRE(s, VT_ERR_INVALID, "vt_set_knob: table full");
    *s = value;
    return VT_OK;
}
const char* vt_last_error(void) { return g_err; }
const char* vt_last_kernel_name(void) { return g_kernel; }
uint64_t vt_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

int vt_memset(void* ptr, int value, uint64_t bytes, void* stream) {
    VT_REQUIRE(ptr || bytes == 0, VT_ERR_INVALID, "vt_memset: null pointer");
    if (bytes == 0) return VT_OK;
    hipError_t e = hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream);
    if (e != hipSuccess) {
        vt_set_error("vt_memset: %s", hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    vt_count_launch();
    return VT_OK;
}

int vt_run_ops(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases, void* stream) {
    return vt_run_ops_streams(ops, n, bases, nbases, stream, nullptr);
}

// Events: eager execution takes them from a per-thread ring (a wait captures the record it was issued against, so an
// event may be re-recorded while an earlier wait on it is still pending; 140 create/destroy pairs per step were
// measurable); stream capture gets fresh events, released by the caller once the capture has ended (`bag`).
static hipEvent_t take_event(std::vector<hipEvent_t>* bag, hipError_t* err) {
    hipEvent_t ev = nullptr;
    hipError_t e = hipSuccess;
    if (bag) {
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) bag->push_back(ev);
    } else {
        constexpr int kRing = 64;
        static thread_local hipEvent_t ring[kRing] = {};
        static thread_local int ring_dev[kRing];
        static thread_local unsigned next = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned k = next++ % kRing;
        if (ring[k] != nullptr && ring_dev[k] != dev) {
            (void)hipEventDestroy(ring[k]);
            ring[k] = nullptr;
        }
        if (ring[k] == nullptr) {
            e = hipEventCreateWithFlags(&ring[k], hipEventDisableTiming);
            ring_dev[k] = dev;
        }
        ev = ring[k];
    }
    *err = e;
    return ev;
}

// order stream `waiter` behind everything enqueued on `signaller` so far
static int stream_wait(hipStream_t waiter, hipStream_t signaller, std::vector<hipEvent_t>* bag) {
    hipError_t e;
    hipEvent_t ev = take_event(bag, &e);
    if (e == hipSuccess) e = hipEventRecord(ev, signaller);
    if (e == hipSuccess) e = hipStreamWaitEvent(waiter, ev, 0);
    if (e != hipSuccess) {
        vt_set_error("stream fork/join: %s", hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    return VT_OK;
}

static int run_ops_impl(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases, void* stream,
                        void* side, std::vector<hipEvent_t>* bag, int32_t flags) {
    VT_REQUIRE(ops && n >= 0 && nbases >= 0 && nbases <= VT_MAX_BASES && (bases || nbases == 0),
               VT_ERR_INVALID, "vt_run_ops: bad argument");
    const bool two = side != nullptr && side != stream;
    bool side_dirty = false;  // side stream has work the main stream has not been ordered behind
    // last FORK_MARK: per thread in eager execution, so that a list run in segments may be cut between a mark and its
    // wait; per call while capturing (the events of a capture die with it)
    static thread_local hipEvent_t eager_mark = nullptr;
    hipEvent_t capture_mark = nullptr;
    hipEvent_t& mark = bag ? capture_mark : eager_mark;
    const int stop_events = (1);
    hipEvent_t fork_ev = nullptr;  // completion event of the kernel right before the next FORK (see vt_common.h)
    // Capture only: the side stream belongs to the capture from its first wait on the capturing stream.  A list that is
    // captured in SEGMENTS (the data-parallel trainer cuts the backward list behind every op that completes a gradient
    // bucket) may begin with side-stream ops whose FORK sits in the previous segment: eagerly the in-order side stream
    // carries that dependency across the cut; a capture would launch those ops on a stream that is not capturing -- they
    // would run once, at capture time, and be missing from the graph.  So the first side-stream op of a capture that no
    // FORK precedes forks by itself (behind the main stream's position, which is behind the previous segment's graph).
    bool side_captured = false;
    auto capture_fork = [&]() -> int {
        if (!bag || !two || side_captured) return VT_OK;
        side_captured = true;
        return stream_wait((hipStream_t)side, (hipStream_t)stream, bag);
    };
    for (int i = 0; i < n; ++i) {
        vt_op op = ops[i];
        const bool on_side = (op.kind & VT_OP_SIDE_STREAM) != 0;
        op.kind &= ~VT_OP_SIDE_STREAM;
        int rc = VT_OK;
#ifdef VT_DIAG_SKIP_FIN  // (diagnostic builds: a kernel family left out of the step, batched / grouped launches included)
        if (op.kind != VT_OP_FORK && op.kind != VT_OP_FORK_MARK && op.kind != VT_OP_FORK_WAIT && op.kind != VT_OP_JOIN &&
            vt_diag_skip_kind(op.kind))
            continue;
#endif
        if (on_side && op.kind != VT_OP_FORK && op.kind != VT_OP_FORK_MARK && op.kind != VT_OP_FORK_WAIT && op.kind != VT_OP_JOIN) {
            rc = capture_fork();
            if (rc != VT_OK) return rc;
        }
        if (op.kind == VT_OP_FORK) {
            if (bag && two) side_captured = true;
            if (two && fork_ev) {
                if (hipStreamWaitEvent((hipStream_t)side, fork_ev, 0) != hipSuccess) {
                    vt_set_error("fork: waiting for the producer's completion event failed");
                    rc = VT_ERR_HIP;
                }
            } else if (two) {
                rc = stream_wait((hipStream_t)side, (hipStream_t)stream, bag);
            }
            fork_ev = nullptr;
        } else if (stop_events && two && !bag && !on_side && i + 1 < n && ops[i + 1].kind == VT_OP_FORK &&
                   (op.kind == VT_OP_BN_BWD_APPLY || op.kind == VT_OP_BN_ACT_APPLY || op.kind == VT_OP_PW_BWD ||
                    op.kind == VT_OP_BN_BWD_FUSED || op.kind == VT_OP_BN_FIN_APPLY || op.kind == VT_OP_BN_BWD_FIN_APPLY ||
                    op.kind == VT_OP_PW_BWD_FIN)) {
            // (the event must belong to the LAST kernel of the op: the only launch of these ops that takes the pending event)
            hipError_t e = hipSuccess;
            hipEvent_t ev = take_event(nullptr, &e);
            if (e == hipSuccess) vt_pending_stop_event = ev;
            rc = run_one(op, bases, nbases, stream);
            if (e == hipSuccess && vt_pending_stop_event == nullptr) fork_ev = ev;  // consumed: attached to the launch
            vt_pending_stop_event = nullptr;
        } else if (op.kind == VT_OP_FORK_MARK) {
            if (two) {
                // (eager: an event of its own per thread and device, never a slot of the shared ring, which 64 later
                // stream waits could re-record before this mark's FORK_WAIT)
                hipError_t e = hipSuccess;
                if (bag) {
                    mark = take_event(bag, &e);
                } else {
                    static thread_local int mark_dev = -1;
                    int dev = 0;
                    (void)hipGetDevice(&dev);
                    if (eager_mark != nullptr && mark_dev != dev) {
                        (void)hipEventDestroy(eager_mark);
                        eager_mark = nullptr;
                    }
                    if (eager_mark == nullptr) {
                        e = hipEventCreateWithFlags(&eager_mark, hipEventDisableTiming);
                        mark_dev = dev;
                    }
                }
                if (e == hipSuccess) e = hipEventRecord(mark, (hipStream_t)stream);
                if (e != hipSuccess) {
                    vt_set_error("fork mark: %s", hipGetErrorString(e));
                    rc = VT_ERR_HIP;
                }
            }
        } else if (op.kind == VT_OP_FORK_WAIT) {
            if (two && bag && mark == nullptr) {
                rc = capture_fork();  // (the mark sits in a previous segment's capture: wait for the main stream's position)
            } else if (two) {
                if (bag) side_captured = true;
                if (mark == nullptr) {
                    vt_set_error("fork wait without a preceding mark");
                    rc = VT_ERR_INVALID;
                } else if (hipStreamWaitEvent((hipStream_t)side, mark, 0) != hipSuccess) {
                    vt_set_error("fork wait failed");
                    rc = VT_ERR_HIP;
                }
            }
        } else if (op.kind == VT_OP_JOIN) {
            // (unconditional: an earlier SEGMENT of the same list may have left side work open, VT_RUN_LEAVE_SIDE_OPEN;
            //  capture: a side stream that has not joined this capture has nothing of it to wait for -- every captured
            //  segment joins its own side work before it ends)
            if (two && (!bag || side_captured)) rc = stream_wait((hipStream_t)stream, (hipStream_t)side, bag);
            side_dirty = false;
        } else if (op.kind == VT_OP_BN_EVAL_COEFFS && i + 1 < n &&
                   (ops[i + 1].kind & ~VT_OP_SIDE_STREAM) == VT_OP_BN_EVAL_COEFFS) {
            // a run of eval-mode coefficient ops (Program puts them all at the head of the list): batched launches
            std::vector<vt_bn_eval_item> items;
            int j = i;
            bool bad = false;
            for (; j < n; ++j) {
                const vt_op& o = ops[j];
                if ((o.kind & ~VT_OP_SIDE_STREAM) != VT_OP_BN_EVAL_COEFFS || ((o.kind & VT_OP_SIDE_STREAM) != 0) != on_side) break;
                vt_bn_eval_item it;  // ptr: gamma beta rm rv scale shift mean invstd | i: C | f: eps
                it.gamma = (const float*)rp(o, 0, bases, nbases, &bad), it.beta = (const float*)rp(o, 1, bases, nbases, &bad);
                it.running_mean = (const float*)rp(o, 2, bases, nbases, &bad);
                it.running_var = (const float*)rp(o, 3, bases, nbases, &bad);
                it.scale = (float*)rp(o, 4, bases, nbases, &bad), it.shift = (float*)rp(o, 5, bases, nbases, &bad);
                it.mean = (float*)rp(o, 6, bases, nbases, &bad), it.invstd = (float*)rp(o, 7, bases, nbases, &bad);
                it.C = o.i[0], it.eps = (float)o.f[0];
                items.push_back(it);
            }
            if (bad) {
                vt_set_error("vt_run_ops: a BatchNorm coefficient op references an unbound base");
                rc = VT_ERR_INVALID;
            } else {
                rc = vt_bn_eval_coeffs_batch(items.data(), (int)items.size(), (two && on_side) ? side : stream);
            }
            if (two && on_side) side_dirty = true;
            i = j - 1;
        } else if (op.kind == VT_OP_PACK_DGRAD && op.i[0] == VT_BF16 && op.i[2] == VT_BF16 && op.i[1] % 8 == 0 &&
                   op.i[4] % 8 == 0 && op.i[6] % 8 == 0 && op.i[3] >= 1 && op.i[3] <= VT_MAX_TAPS) {
            // (the tap-count check is the gather loop's own: an op that fails it must fall through to run_one, which
            //  reports it, instead of leaving `items` empty and this op to be visited again)
            // consecutive bf16 filter re-packs of one stream leave as ONE batched launch per VT_PACK_BATCH of them
            std::vector<vt_pack_item> items;
            int j = i;
            bool bad = false;
            for (; j < n; ++j) {
                const vt_op& o = ops[j];
                if ((o.kind & ~VT_OP_SIDE_STREAM) != VT_OP_PACK_DGRAD || ((o.kind & VT_OP_SIDE_STREAM) != 0) != on_side ||
                    o.i[0] != VT_BF16 || o.i[2] != VT_BF16 || o.i[3] < 1 || o.i[3] > VT_MAX_TAPS || o.i[1] % 8 || o.i[4] % 8 ||
                    o.i[6] % 8)
                    break;
                vt_pack_item it;  // i: src_dtype ldw dst_dtype nsel Cout ntaps Cin _ sel[36]
                memset(&it, 0, sizeof(it));
                it.w = rp(o, 0, bases, nbases, &bad), it.out = rp(o, 1, bases, nbases, &bad);
                it.ldw = o.i[1], it.nsel = o.i[3], it.Cout = o.i[4], it.ntaps = o.i[5], it.Cin = o.i[6];
                for (int t = 0; t < it.nsel; ++t) it.sel[t] = (int8_t)o.i[8 + t];
                items.push_back(it);
            }
            if (bad) {
                vt_set_error("vt_run_ops: a filter re-pack references an unbound base");
                rc = VT_ERR_INVALID;
            } else {
                rc = vt_pack_dgrad_filter_batch(items.data(), (int)items.size(), (two && on_side) ? side : stream);
            }
            if (two && on_side) side_dirty = true;
            i = j - 1;
        } else if (op.kind == VT_OP_CONV_WGRAD && op.ptr[3].base < 0 && i + 1 < n &&
                   (ops[i + 1].kind & ~VT_OP_SIDE_STREAM) == VT_OP_CONV_WGRAD) {
            // consecutive filter gradients of ONE descriptor on one stream (the engine holds the same-shape layers of a
            // stage back and releases them together): vt_conv_wgrad_group, which shares a launch among up to 8 of them
            constexpr int DI = (int)(sizeof(vt_conv_desc) / 4);
            std::vector<const void*> xs, dzs;
            std::vector<float*> dws;
            int j = i;
            bool bad = false;
            for (; j < n; ++j) {
                const vt_op& o = ops[j];
                if ((o.kind & ~VT_OP_SIDE_STREAM) != VT_OP_CONV_WGRAD || ((o.kind & VT_OP_SIDE_STREAM) != 0) != on_side ||
                    o.ptr[3].base >= 0 || memcmp(o.i, op.i, (DI + 1) * sizeof(int32_t)) != 0)
                    break;
                xs.push_back(rp(o, 0, bases, nbases, &bad));
                dzs.push_back(rp(o, 1, bases, nbases, &bad));
                dws.push_back((float*)rp(o, 2, bases, nbases, &bad));
            }
            if (bad) {
                vt_set_error("vt_run_ops: a filter-gradient op references an unbound base");
                rc = VT_ERR_INVALID;
            } else {
                vt_conv_desc d;
                memcpy(&d, op.i, sizeof(d));
                rc = vt_conv_wgrad_group(&d, (int)xs.size(), xs.data(), dzs.data(), dws.data(), op.i[DI],
                                         (two && on_side) ? side : stream);
            }
            if (two && on_side) side_dirty = true;
            i = j - 1;
        } else {
            rc = run_one(op, bases, nbases, (two && on_side) ? side : stream);
            if (two && on_side) side_dirty = true;
        }
        if (rc != VT_OK) {
            char msg[400];
            snprintf(msg, sizeof(msg), "%s", g_err);
            vt_set_error("op %d (kind %d, tag %d): %s", i, op.kind, op.tag, msg);
            return rc;
        }
    }
    // never return with unjoined side work (the caller only knows about `stream`) unless it asked to keep it open
    if (two && side_dirty && !(flags & VT_RUN_LEAVE_SIDE_OPEN)) return stream_wait((hipStream_t)stream, (hipStream_t)side, bag);
    return VT_OK;
}

int vt_run_ops_streams(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases, void* stream,
                       void* side) {
    return run_ops_impl(ops, n, bases, nbases, stream, side, nullptr, 0);
}

int vt_run_ops_streams_ex(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases, void* stream,
                          void* side, int32_t flags) {
    return run_ops_impl(ops, n, bases, nbases, stream, side, nullptr, flags);
}

int vt_stream_wait(void* waiter, void* signaller) {
    VT_REQUIRE(waiter != signaller, VT_ERR_INVALID, "vt_stream_wait: a stream cannot wait for itself");
    return stream_wait((hipStream_t)waiter, (hipStream_t)signaller, nullptr);
}

int vt_graph_create(const vt_op* ops, int32_t n, void* const* bases, int32_t nbases, void** graph_out) {
    VT_REQUIRE(ops && n > 0 && graph_out, VT_ERR_INVALID, "vt_graph_create: bad argument");
    hipStream_t cs;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) {
        vt_set_error("vt_graph_create: stream: %s", hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    // a second stream joins the capture at the first FORK; its ops become parallel branches
    bool wants_side = false;
    for (int i = 0; i < n; ++i) wants_side |= (ops[i].kind & VT_OP_SIDE_STREAM) != 0;
    hipStream_t ss = nullptr;
    if (wants_side && hipStreamCreateWithFlags(&ss, hipStreamNonBlocking) != hipSuccess) ss = nullptr;
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(cs);
        if (ss) (void)hipStreamDestroy(ss);
        vt_set_error("vt_graph_create: begin capture: %s", hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    std::vector<hipEvent_t> bag;
    const int rc = run_ops_impl(ops, n, bases, nbases, cs, ss, &bag, 0);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(cs, &g);
    for (hipEvent_t ev : bag) (void)hipEventDestroy(ev);  // only after the capture has ended
    (void)hipStreamDestroy(cs);
    if (ss) (void)hipStreamDestroy(ss);
    if (rc != VT_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) {
        vt_set_error("vt_graph_create: end capture: %s", hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        vt_set_error("vt_graph_create: instantiate: %s", hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    Graph* G = new Graph{g, ex};
    *graph_out = G;
    return VT_OK;
}

int vt_graph_launch(void* graph, void* stream) {
    VT_REQUIRE(graph, VT_ERR_INVALID, "vt_graph_launch: null graph");
    hipError_t e = hipGraphLaunch(((Graph*)graph)->exec, (hipStream_t)stream);
    if (e != hipSuccess) {
        vt_set_error("vt_graph_launch: %s", hipGetErrorString(e));
        return VT_ERR_HIP;
    }
    vt_count_launch();
    return VT_OK;
}

int vt_graph_destroy(void* graph) {
    if (!graph) return VT_OK;
    Graph* G = (Graph*)graph;
    (void)hipGraphExecDestroy(G->exec);
    (void)hipGraphDestroy(G->graph);
    delete G;
    return VT_OK;
}

int vt_event_create(void** ev) {
    VT_REQUIRE(ev, VT_ERR_INVALID, "vt_event_create: null");
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) {
        vt_set_error("vt_event_create: %s", hipGetErrorString(rc));
        return VT_ERR_HIP;
    }
    *ev = (void*)e;
    return VT_OK;
}
int vt_event_record(void* ev, void* stream) {
    hipError_t rc = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    if (rc != hipSuccess) {
        vt_set_error("vt_event_record: %s", hipGetErrorString(rc));
        return VT_ERR_HIP;
    }
    return VT_OK;
}
int vt_event_elapsed_ms(void* start, void* stop, float* ms) {
    VT_REQUIRE(ms, VT_ERR_INVALID, "vt_event_elapsed_ms: null");
    hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
    if (rc == hipSuccess) rc = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (rc != hipSuccess) {
        vt_set_error("vt_event_elapsed_ms: %s", hipGetErrorString(rc));
        return VT_ERR_HIP;
    }
    return VT_OK;
}
int vt_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
    return VT_OK;
}

}  // extern "C"
