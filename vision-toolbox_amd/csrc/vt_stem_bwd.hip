// vt_stem_bwd.hip -- backward of the stem unit Conv3x3(3->C, stride 1) -> BatchNorm -> ReLU in ONE
// streaming pass (bf16).  Reference: the autograd backward of ConvNormAct (components.py:26-44) as
// the first layer of the Darknet family (darknet.py:75): 224x224 x batch 256 = 12.8 M pixels,
// 822 MB each for z, dy.
//
// The unit's input is the image, so nothing consumes d(z) except the filter gradient
//     dW[n][t][c] = sum_p dz[p][n] * x[p + t][c],      dz = a_n*g - b_n*z + d_n   (g = dy * [y > 0])
// whose per-channel coefficients (a, b, d) come out of the BatchNorm-backward reduction over the
// same tensors.  dz is linear in (g, z, 1), so the pass that reduces sum(g), sum(g*xhat) can also
// accumulate the three correlations
//     G[n][t][c] = sum_p g[p][n] x[p+t][c],   Z[n][t][c] = sum_p z[p][n] x[p+t][c],   X[t][c] = sum_p x[p+t][c]
// and a 900-thread kernel finishes  dW = a*G - b*Z + d*X  once (a, b, d) exist.  dy, z and x are read
// ONCE (1.85 GB) and d(z) is never formed: this replaces bn_bwd_reduce (1.64 GB) + bn_bwd_apply
// (2.47 GB) + the stem's filter-gradient launch (1.03 GB), the serial tail of every backward pass.
//
//   * positions are enumerated in padded coordinates ((H+1) x (W+1) per image, as vt_wgrad_span.hip): a tap is a
//     constant offset, padded positions carry g = z = 0 and a zero `one`;
//   * per step of 64 positions the 256 threads load dy and z (16 B each), form g, add their per-channel partial
//     sums and park g | z | one-flag as bf16 rows in LDS; x rows (16 B = 8 channels, 3 real) go to a ring;
//   * K = positions: both MFMA operands are position-major in LDS and are formed by ds_read_b64_tr_b16.
//     A = [g | z | one] (2C/16 + 1 fragments), B = filter row e, pixel pair h: 16 columns = 2 pixels x 8 channels
//     of the 4-pixel group starting at tap (e, 0) (the 4th pixel is computed and dropped: 96 columns for 72);
//   * waves 0/1 take positions 0..31 of the step, waves 2/3 positions 32..63; even waves filter-row fragments 0..2,
//     odd waves 3..5.  HBM-bound: 9.2 KB per step against 15-27 MFMAs per wave.
// Partial results leave as f32 atomics into `gzx` [8 replicas][2C+16][96] and `sums` [32 replicas][2][C].
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "vt_common.h"

namespace {

struct SbArgs {
    const bf16_t* x;    // [B*H*W][8]
    const bf16_t* dy;   // ld lddy
    const bf16_t* z;    // ld ldz
    const float *scale, *shift, *mean, *invstd;
    float* sums;        // fixed-point [kStatReplicas][2][C] (vt_common.h)
    float* gzx;         // [kGzxReplicas][2C+16][96] f32, or the same entries as fixed-point int64 pairs (fixed)
    int fixed;
    int B, H, W, C, lddy, ldz, relu;
    int PW, PH, S;      // padded pitch / rows / positions per image
    int NP, chunk;      // total positions, positions per workgroup (multiple of 64)
    int halo, rx;       // ring look-ahead (multiple of 64), ring rows (power of two)
    unsigned magic_pw, magic_ph;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct Pos {  // a padded position, decomposed
    int b, i, j;
    __device__ __forceinline__ void init(long P, int S, int PW) {
        long bb = P / S;
        long rem = P - bb * S;
        if (rem < 0) rem += S, --bb;
        b = (int)bb;
        i = (int)(rem / PW);
        j = (int)(rem - (long)i * PW);
    }
    __device__ __forceinline__ void advance64(int PH, int PW, unsigned magic_pw, unsigned magic_ph) {
        j += 64;
        const int qw = (int)__umulhi((unsigned)j, magic_pw);
        j -= qw * PW;
        i += qw;
        const int qh = (int)__umulhi((unsigned)i, magic_ph);
        i -= qh * PH;
        b += qh;
    }
    __device__ __forceinline__ bool real(int B, int H, int W) const {
        return (unsigned)b < (unsigned)B && i < H && j < W;
    }
    __device__ __forceinline__ long pixel(int H, int W) const { return ((long)b * H + i) * W + j; }
};

__device__ __forceinline__ uint4 ldg16(const void* p) { return *(const uint4*)p; }

// accumulate in place (the builtin lets the compiler rename the accumulator across the two unrolled steps)
__device__ __forceinline__ void mma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

constexpr int kGzxReplicas = 8;
constexpr int kCols = 96;  // 3 filter rows x 4 pixels x 8 channels

// FA = C / 16 (2 or 4)
// FROM_Y (round 3): the unit's OUTPUT y = relu(z*scale + shift) is read instead of the pre-activation z, which the forward
// pass then never stores (the stem conv runs twice: a statistics-only pass and a pass with the normalise + ReLU epilogue).
//   * mask: y > 0 (the same predicate: y = relu(z*scale + shift));
//   * z is LINEAR in the 27 input values under the 3x3 window (the patch), so neither of the two sums that contain z needs
//     z itself:  sum g * z = sum_i W[n][i] * G[n][i]  with G = sum g * x(p+t), which the pass accumulates anyway
//     (vt_stem_bn_bwd_s2: it writes sum g * xhat into `sums` before vt_bn_bwd_finalize -- nothing is recovered from the
//     rounded y, and a BatchNorm weight of exactly 0 is as exact as any other), and  Z = sum z * x(p+t) = W P  with
//     P = sum patch(p) * x(p+t) (vt_stem_bn_bwd_combine_y): the 32-column block that held z holds the patch (27 values +
//     5 zeros), gathered from the x ring: same fragments, same MFMA count.
template <int FA, bool FROM_Y>
__global__ void __launch_bounds__(256, 2) stem_bwd_kernel(const SbArgs p) {
    static_assert(!FROM_Y || FA == 2, "the patch block is 32 columns wide");
    constexpr int C = 16 * FA;
    constexpr int CPR = C / 8;            // 16-byte chunks per dy / z row
    constexpr int NIT = CPR / 4;          // (position, chunk) items per thread and step
    constexpr int PITCH = 2 * C + 32;     // bytes per tile row: C channels + a 16-channel block for the `one` flag
    constexpr int TILE = 64 * PITCH;
    constexpr int NA = 2 * FA + 1;        // A fragments: g, z, one
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sG = smem;                 // [2][64][PITCH]
    char* sZ = smem + 2 * TILE;      // [2][64][PITCH]  (channels C.. = flag block)
    float* sCoef = (float*)(smem + 4 * TILE);  // [4][C]: scale | shift | mean | 1 / scale
    char* sX = smem + 4 * TILE + 4 * C * 4;    // [rx][16 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long Pbeg = (long)blockIdx.x * p.chunk;
    const long Pend = min((long)p.NP, Pbeg + p.chunk);
    if (Pbeg >= Pend) return;
    const int nsteps = (int)((Pend - Pbeg + 63) / 64);
    const unsigned xmask = (unsigned)p.rx - 1u;

    // ---- zero the tiles once (the unused / flag blocks stay zero) ----------------------------------
    for (int i = tid; i < 4 * TILE / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
    if (tid < 3 * C) sCoef[tid] = (tid < C ? p.scale : tid < 2 * C ? p.shift - C : p.mean - 2 * C)[tid];

    // ---- ring prologue: rows [Pbeg - halo, Pbeg + halo) ---------------------------------------------
    {
        const int nch = 2 * p.halo / 64;
        for (int c = wave; c < nch; c += 4) {
            const long P = Pbeg - p.halo + 64l * c + lane;
            Pos q;
            q.init(P, p.S, p.PW);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q.real(p.B, p.H, p.W)) v = ldg16(p.x + q.pixel(p.H, p.W) * 8);
            *(uint4*)(sX + (((unsigned)(int)P & xmask) << 4)) = v;
        }
    }

    // ---- per-thread streams -------------------------------------------------------------------------
    const int chunk = tid % CPR;
    // FROM_Y: column 8*chunk + e of the patch block is input channel i % 3 under tap i / 3 (i = 8*chunk + e < 27)
    int pat_row[8], pat_byte[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = 8 * chunk + e, t = i / 3, c = i - 3 * t;
        pat_row[e] = i < 27 ? (t / 3 - 1) * p.PW + (t % 3 - 1) : 0;
        pat_byte[e] = i < 27 ? 2 * c : -1;
    }
    int ipos[NIT];
    Pos ps[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        ipos[k] = (tid + 256 * k) / CPR;
        ps[k].init(Pbeg + ipos[k], p.S, p.PW);
    }
    Pos px;  // wave 0: the ring's front row of this lane
    px.init(Pbeg + p.halo + lane, p.S, p.PW);

    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;

    constexpr int NPF = 4;  // steps of loads in flight: 2 workgroups x 4 x 9.2 KB per CU against ~2.5 us of loaded HBM latency
    uint4 vg[NPF][NIT], vz[NPF][NIT], vx[NPF];
    bool ok[NPF][NIT];
    auto issue = [&](auto slot_c, int s) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const bool v = s < nsteps && (Pbeg + 64l * s + ipos[k] < Pend) && ps[k].real(p.B, p.H, p.W);
            ok[slot][k] = v;
            vg[slot][k] = vz[slot][k] = make_uint4(0, 0, 0, 0);
            if (v) {
                const long pix = ps[k].pixel(p.H, p.W);
                vg[slot][k] = ldg16(p.dy + pix * p.lddy + chunk * 8);
                vz[slot][k] = ldg16(p.z + pix * p.ldz + chunk * 8);
            }
            ps[k].advance64(p.PH, p.PW, p.magic_pw, p.magic_ph);
        }
        if (wave == 0) {
            vx[slot] = make_uint4(0, 0, 0, 0);
            if (s < nsteps && px.real(p.B, p.H, p.W)) vx[slot] = ldg16(p.x + px.pixel(p.H, p.W) * 8);
            px.advance64(p.PH, p.PW, p.magic_pw, p.magic_ph);
        }
    };

    // ---- fragment addressing (ds_read_b64_tr_b16: lane 4q+pp of a 16-lane group addresses row q, columns
    // 4pp..4pp+3 of a 4 x 16 block and receives column u of the block's four rows; element e<4 <-> position
    // 4g+e, e>=4 <-> 16+4g+(e-4), for both operands) ------------------------------------------------------
    const int g = lane >> 4, u = lane & 15, q = u >> 2, pp = u & 3;
    const int kb = wave >> 1, half = wave & 1;
    const int rowlo = 32 * kb + 4 * g + q;
    const unsigned a_off = (unsigned)(rowlo * PITCH + 8 * pp);
    // filter-row fragment f = 3*half + ff: e = f / 2, h = f % 2; first pixel of the lane's 8 bytes:
    //   position + (e-1)*PW - 1 + 2h + (pp >> 1)
    int b_rel[3];
#pragma unroll
    for (int ff = 0; ff < 3; ++ff) {
        const int f = 3 * half + ff, e = f >> 1, h = f & 1;
        b_rel[ff] = rowlo + (e - 1) * p.PW - 1 + 2 * h + (pp >> 1);
    }
    const unsigned b_sub = 8u * (pp & 1);

    f32x4 acc[NA][3];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int ff = 0; ff < 3; ++ff) acc[a][ff] = f32x4{0.f, 0.f, 0.f, 0.f};

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    issue(S0{}, 0);
    issue(S1{}, 1);
    issue(S2{}, 2);
    issue(S3{}, 3);
    __syncthreads();  // tiles zeroed, ring prologue visible

    auto step = [&](auto slot_c, int s) {
        constexpr int slot = decltype(slot_c)::value;
        // ---- elementwise: g, partial sums, park the rows --------------------------------------------
        char* tg_ = sG + (s & 1) * TILE;
        char* tz_ = sZ + (s & 1) * TILE;
        float sc[8], sf[8], mu[8];  // re-read every step: registers that the MFMA phase gets back
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
            const f32x4 a4 = *(const volatile f32x4*)(sCoef + chunk * 8 + 4 * h4);
            const f32x4 b4 = *(const volatile f32x4*)(sCoef + C + chunk * 8 + 4 * h4);
            const f32x4 c4 = *(const volatile f32x4*)(sCoef + 2 * C + chunk * 8 + 4 * h4);
            sc[4 * h4] = a4[0], sc[4 * h4 + 1] = a4[1], sc[4 * h4 + 2] = a4[2], sc[4 * h4 + 3] = a4[3];
            sf[4 * h4] = b4[0], sf[4 * h4 + 1] = b4[1], sf[4 * h4 + 2] = b4[2], sf[4 * h4 + 3] = b4[3];
            mu[4 * h4] = c4[0], mu[4 * h4 + 1] = c4[1], mu[4 * h4 + 2] = c4[2], mu[4 * h4 + 3] = c4[3];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            float fg[8], fz[8];
            VecIO<bf16_t>::unpack(vg[slot][k], fg);
            VecIO<bf16_t>::unpack(vz[slot][k], fz);
            unsigned keep[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bool on;
                float zv;
                if constexpr (FROM_Y) {
                    on = !p.relu || fz[e] > 0.f;          // fz holds y
                    zv = mu[e];                            // (sum g * xhat comes from G: vt_stem_bn_bwd_s2)
                } else {
                    on = !p.relu || fmaf(fz[e], sc[e], sf[e]) > 0.f;
                    zv = fz[e];
                }
                const float gg = on ? fg[e] : 0.f;
                keep[e] = on ? 0xffffu : 0u;
                s1[e] += gg;
                s2[e] = fmaf(gg, zv - mu[e], s2[e]);
            }
            uint4 m;
            m.x = keep[0] | (keep[1] << 16), m.y = keep[2] | (keep[3] << 16);
            m.z = keep[4] | (keep[5] << 16), m.w = keep[6] | (keep[7] << 16);
            uint4 go = vg[slot][k];
            go.x &= m.x, go.y &= m.y, go.z &= m.z, go.w &= m.w;
            const int row = ipos[k];
            *(uint4*)(tg_ + row * PITCH + chunk * 16) = go;
            if constexpr (FROM_Y) {
                // the patch of this position from the x ring (rows written at least one step ago: halo >= PW + 3 + 64)
                unsigned short pv[8];
                const int P = (int)(Pbeg + 64l * s) + row;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    pv[e] = 0;
                    if (pat_byte[e] >= 0 && ok[slot][k])
                        pv[e] = *(const unsigned short*)(sX + ((((unsigned)(P + pat_row[e])) & xmask) << 4) + pat_byte[e]);
                }
                uint4 pk;
                pk.x = pv[0] | ((unsigned)pv[1] << 16), pk.y = pv[2] | ((unsigned)pv[3] << 16);
                pk.z = pv[4] | ((unsigned)pv[5] << 16), pk.w = pv[6] | ((unsigned)pv[7] << 16);
                *(uint4*)(tz_ + row * PITCH + chunk * 16) = pk;
            } else {
                *(uint4*)(tz_ + row * PITCH + chunk * 16) = vz[slot][k];
            }
            if (chunk == 0) *(uint4*)(tz_ + row * PITCH + 2 * C) = make_uint4(ok[slot][k] ? 0x3f80u : 0u, 0, 0, 0);
        }
        if (wave == 0) {
            const long P = Pbeg + p.halo + 64l * s + lane;
            *(uint4*)(sX + (((unsigned)(int)P & xmask) << 4)) = vx[slot];
        }
        issue(slot_c, s + NPF);
        __syncthreads();

        // ---- MFMA: [g | z | one]^T x [filter-row fragments] over this wave's 32 positions ------------------
        bf16x8 bfr[3];
        const int pbase = (int)(Pbeg + 64l * s);
#pragma unroll
        for (int ff = 0; ff < 3; ++ff) {
            const unsigned lo_o = ((((unsigned)(pbase + b_rel[ff])) & xmask) << 4) + b_sub;
            const unsigned hi_o = ((((unsigned)(pbase + b_rel[ff] + 16)) & xmask) << 4) + b_sub;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sX + lo_o));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sX + hi_o));
            bfr[ff] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const char* t = a < FA ? tg_ + 32 * a : tz_ + 32 * (a - FA);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t + a_off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t + a_off + 16 * PITCH));
            const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int ff = 0; ff < 3; ++ff)
                mma(acc[a][ff], af, bfr[ff]);
        }
    };

    for (int s = 0; s < nsteps; s += NPF) {
        step(S0{}, s);
        if (s + 1 < nsteps) step(S1{}, s + 1);
        if (s + 2 < nsteps) step(S2{}, s + 2);
        if (s + 3 < nsteps) step(S3{}, s + 3);
    }

    // ---- results --------------------------------------------------------------------------------------------
    const int rep = (int)(blockIdx.x % kGzxReplicas);
    float* out = p.gzx + (long)rep * (2 * C + 16) * kCols;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int ff = 0; ff < 3; ++ff)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * a + 4 * g + r;
                if (a < 2 * FA || row == 2 * C)  // of the flag block only its first row carries anything
                {
                    const long e = (long)row * kCols + 16 * (3 * half + ff) + u;
                    if (p.fixed)  // integer atomics: the correlations do not depend on the order of the workgroups
                        vt_stat_add(p.gzx, (long)rep * (2 * C + 16) * kCols + e, acc[a][ff][r]);
                    else
                        atomicAdd(out + e, acc[a][ff][r]);
                }
            }
    // per-channel sums: fold the lanes that share a chunk, one writer per (wave, chunk)
    for (int off = CPR; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[e] += __shfl_xor(s1[e], off, 64);
            s2[e] += __shfl_xor(s2[e], off, 64);
        }
    }
    if (lane < CPR) {
        const int srep = (int)(blockIdx.x % kStatReplicas);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = chunk * 8 + e;
            vt_stat_add(p.sums, ((long)srep * 2 + 0) * C + c, s1[e]);
            if constexpr (!FROM_Y) vt_stat_add(p.sums, ((long)srep * 2 + 1) * C + c, s2[e] * p.invstd[c]);
        }
    }
}

// dW[n][t][c] += a_n*G - b_n*Z + d_n*X  (coef = [a | b | d] of bn_bwd_finalize), c < cin of the 8 staged channels
__global__ void stem_bwd_combine_kernel(const float* __restrict__ gzx, const float* __restrict__ coef, int C, int cin,
                                        float* __restrict__ dw, int fixed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * 9 * cin) return;
    const int c = idx % cin, t = (idx / cin) % 9, n = idx / (9 * cin);
    const int col = (t / 3) * 32 + (t % 3) * 8 + c;
    const int rows = 2 * C + 16;
    double G = 0.0, Z = 0.0, X = 0.0;
    if (fixed) {
        const long long* q = (const long long*)gzx;
        long long h[3] = {0, 0, 0}, l[3] = {0, 0, 0};
        for (int r = 0; r < kGzxReplicas; ++r) {
            const long base = (long)r * rows * kCols;
            const long e[3] = {base + (long)n * kCols + col, base + (long)(C + n) * kCols + col,
                               base + (long)(2 * C) * kCols + col};
            for (int k = 0; k < 3; ++k) h[k] += q[2 * e[k]], l[k] += q[2 * e[k] + 1];
        }
        G = (double)h[0] * 4096.0 + (double)l[0] * (1.0 / 8589934592.0);
        Z = (double)h[1] * 4096.0 + (double)l[1] * (1.0 / 8589934592.0);
        X = (double)h[2] * 4096.0 + (double)l[2] * (1.0 / 8589934592.0);
    } else
    for (int r = 0; r < kGzxReplicas; ++r) {
        const float* o = gzx + (long)r * rows * kCols;
        G += (double)o[(long)n * kCols + col];
        Z += (double)o[(long)(C + n) * kCols + col];
        X += (double)o[(long)(2 * C) * kCols + col];
    }
    const double v = (double)coef[n] * G - (double)coef[C + n] * Z + (double)coef[2 * C + n] * X;
    dw[idx] += (float)v;
}

// the same for the FROM_Y reduction: rows [C, C + 27) of gzx hold P[i][col] = sum patch_i * x(p+t) and
// Z[n][col] = sum_i w[n][i] P[i][col] with w the bf16 filter image [C][9][8] the forward conv read (z is linear in the patch)
__global__ void stem_bwd_combine_y_kernel(const float* __restrict__ gzx, const float* __restrict__ coef,
                                          const bf16_t* __restrict__ w, int C, int cin, float* __restrict__ dw, int fixed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * 9 * cin) return;
    const int c = idx % cin, t = (idx / cin) % 9, n = idx / (9 * cin);
    const int col = (t / 3) * 32 + (t % 3) * 8 + c;
    const int rows = 2 * C + 16;
    auto entry = [&](int row) -> double {
        double v = 0.0;
        if (fixed) {
            const long long* q = (const long long*)gzx;
            long long h = 0, l = 0;
            for (int r = 0; r < kGzxReplicas; ++r) {
                const long e = (long)r * rows * kCols + (long)row * kCols + col;
                h += q[2 * e], l += q[2 * e + 1];
            }
            v = (double)h * 4096.0 + (double)l * (1.0 / 8589934592.0);
        } else {
            for (int r = 0; r < kGzxReplicas; ++r) v += (double)gzx[(long)r * rows * kCols + (long)row * kCols + col];
        }
        return v;
    };
    const double G = entry(n), X = entry(2 * C);
    double Z = 0.0;
    for (int i = 0; i < 27; ++i) Z += (double)(float)w[(long)n * 72 + (i / 3) * 8 + (i % 3)] * entry(C + i);
    const double v = (double)coef[n] * G - (double)coef[C + n] * Z + (double)coef[2 * C + n] * X;
    dw[idx] += (float)v;
}

// FROM_Y: sums[1][n] = invstd_n * (sum_i w[n][i] G[n][i] - mean_n * sum g)   (= sum g * xhat, see the kernel's header)
__global__ void stem_bwd_s2_kernel(const float* __restrict__ gzx, const bf16_t* __restrict__ w, const float* __restrict__ mean,
                                   const float* __restrict__ invstd, float* __restrict__ sums, int C, int fixed) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= C) return;
    const int rows = 2 * C + 16;
    double dot = 0.0;
    for (int i = 0; i < 27; ++i) {
        const int t = i / 3, c = i - 3 * t;
        const int col = (t / 3) * 32 + (t % 3) * 8 + c;
        double v = 0.0;
        if (fixed) {
            const long long* q = (const long long*)gzx;
            long long h = 0, l = 0;
            for (int r = 0; r < kGzxReplicas; ++r) {
                const long e = (long)r * rows * kCols + (long)n * kCols + col;
                h += q[2 * e], l += q[2 * e + 1];
            }
            v = (double)h * 4096.0 + (double)l * (1.0 / 8589934592.0);
        } else {
            for (int r = 0; r < kGzxReplicas; ++r) v += (double)gzx[(long)r * rows * kCols + (long)n * kCols + col];
        }
        dot += (double)(float)w[(long)n * 72 + t * 8 + c] * v;
    }
    const double s1 = vt_stat_sum(sums, n, 2L * C);
    const double s2 = (double)invstd[n] * (dot - (double)mean[n] * s1);
    // (the entry is still zero: the reduction left sum g * xhat to this kernel.  Two limbs as vt_stat_add, from a double.)
    long long* q = (long long*)sums + 2 * ((long)C + n);
    if (!(fabs(s2) < 1.0e30)) {  // non-finite (a poisoned sum g, a diverged G): poison this entry too (vt_common.h)
        q[0] = kStatPoison, q[1] = 0;
        return;
    }
    const double hf = trunc(s2 * (1.0 / 4096.0));
    const double rem = s2 - hf * 4096.0;
    q[0] = (long long)hf;
    q[1] = __double2ll_rn(rem * 8589934592.0);
}

}  // namespace

extern "C" {

int64_t vt_stem_bn_bwd_scratch_bytes(int32_t C) { return (int64_t)kGzxReplicas * (2 * C + 16) * kCols * 16; }

int vt_stem_bn_bwd_reduce(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* x, const void* dy,
                          int32_t lddy, const void* z, int32_t ldz, const float* scale, const float* shift,
                          const float* mean, const float* invstd, int32_t relu, float* sums, float* gzx, int32_t fixed,
                          void* stream) {
    VT_REQUIRE(dtype == VT_BF16 && C == 32 && B > 0 && H > 0 && W > 0 && lddy % 8 == 0 && ldz % 8 == 0 &&
                   lddy >= C && ldz >= C,
               VT_ERR_UNSUPPORTED, "vt_stem_bn_bwd_reduce: bf16, 32 channels, 16-byte aligned rows");
    VT_REQUIRE(x && dy && z && scale && shift && mean && invstd && sums && gzx, VT_ERR_INVALID,
               "vt_stem_bn_bwd_reduce: null pointer");
    SbArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)x, a.dy = (const bf16_t*)dy, a.z = (const bf16_t*)z;
    a.scale = scale, a.shift = shift, a.mean = mean, a.invstd = invstd, a.sums = sums, a.gzx = gzx;
    a.B = B, a.H = H, a.W = W, a.C = C, a.lddy = lddy, a.ldz = ldz, a.relu = relu;
    const bool from_y = (fixed & 2) != 0;  // bit 1 of `fixed`: `z` is the unit's output y (see the kernel's header)
    a.fixed = (fixed & 1) ? 1 : 0;
    a.PW = W + 1, a.PH = H + 1, a.S = a.PW * a.PH;
    const long NP = (long)B * a.S;
    VT_REQUIRE(NP <= 0x7fff0000L, VT_ERR_UNSUPPORTED, "vt_stem_bn_bwd_reduce: more than 2^31 positions");
    a.NP = (int)NP;
    a.magic_pw = (unsigned)((0x100000000ull + a.PW - 1) / a.PW);
    a.magic_ph = (unsigned)((0x100000000ull + a.PH - 1) / a.PH);
    a.halo = (a.PW + 3 + 63) / 64 * 64;  // rows [P - PW - 1, P + 63 + PW + 3] of a step must be in the ring
    if (from_y) a.halo += 64;            // ... one step EARLIER: the patch gather of a step reads the ring before its barrier
    int rx = 256;
    while (rx < 2 * a.halo + 192) rx *= 2;
    a.rx = rx;
    const int pitch = 2 * C + 32;
    const int smem = 4 * 64 * pitch + 4 * C * 4 + rx * 16;
    VT_REQUIRE(smem <= 64 * 1024, VT_ERR_UNSUPPORTED, "vt_stem_bn_bwd_reduce: image too wide for the LDS ring");
    // one workgroup per resident slot (2 per CU at this register count); a chunk must dwarf the ring warm-up (2*halo rows).
    // (1024 until round 3: 512 measured 21.34 vs 21.42 ms per step, 256 21.81, 768 21.51)
    const int target = (512);
    long chunk = (NP + target - 1) / target;
    const long min_chunk = 16l * a.halo;
    if (chunk < min_chunk) chunk = min_chunk;
    chunk = (chunk + 63) / 64 * 64;
    a.chunk = (int)chunk;
    const long blocks = (NP + chunk - 1) / chunk;
    hipStream_t st = (hipStream_t)stream;
    vt_note_kernel("stem_bwd_kernel<%d%s>", C / 16, from_y ? ",y" : "");
    if (from_y)
        hipLaunchKernelGGL((stem_bwd_kernel<2, true>), dim3((unsigned)blocks), dim3(256), smem, st, a);
    else
        hipLaunchKernelGGL((stem_bwd_kernel<2, false>), dim3((unsigned)blocks), dim3(256), smem, st, a);
    VT_CHECK_LAUNCH("vt_stem_bn_bwd_reduce");
    return VT_OK;
}

int vt_stem_bn_bwd_combine(int32_t C, int32_t cin, const float* gzx, const float* coef, float* dw, int32_t fixed,
                           void* stream) {
    VT_REQUIRE(C == 32 && cin >= 1 && cin <= 8 && gzx && coef && dw, VT_ERR_INVALID,
               "vt_stem_bn_bwd_combine: bad argument");
    const int n = C * 9 * cin;
    hipLaunchKernelGGL(stem_bwd_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gzx, coef, C,
                       cin, dw, fixed ? 1 : 0);
    VT_CHECK_LAUNCH("vt_stem_bn_bwd_combine");
    return VT_OK;
}

int vt_stem_bn_bwd_s2(int32_t C, const float* gzx, const void* w, const float* mean, const float* invstd, float* sums,
                      int32_t fixed, void* stream) {
    VT_REQUIRE(C == 32 && gzx && w && mean && invstd && sums, VT_ERR_INVALID, "vt_stem_bn_bwd_s2: bad argument");
    hipLaunchKernelGGL(stem_bwd_s2_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gzx, (const bf16_t*)w, mean, invstd,
                       sums, C, (fixed & 1) ? 1 : 0);
    VT_CHECK_LAUNCH("vt_stem_bn_bwd_s2");
    return VT_OK;
}

int vt_stem_bn_bwd_combine_y(int32_t C, int32_t cin, const float* gzx, const float* coef, const void* w, float* dw,
                             int32_t fixed, void* stream) {
    VT_REQUIRE(C == 32 && cin >= 1 && cin <= 8 && gzx && coef && w && dw, VT_ERR_INVALID,
               "vt_stem_bn_bwd_combine_y: bad argument");
    const int n = C * 9 * cin;
    hipLaunchKernelGGL(stem_bwd_combine_y_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gzx, coef,
                       (const bf16_t*)w, C, cin, dw, (fixed & 1) ? 1 : 0);
    VT_CHECK_LAUNCH("vt_stem_bn_bwd_combine_y");
    return VT_OK;
}

}  // extern "C"
