// vt_wgrad6.hip -- filter gradient of the stride-1 3x3 convolutions (bf16) as a CU-owning, wave-specialised kernel:
// one 12-wave workgroup per CU, two compute groups half a step apart plus four LDS-DMA loader waves (the schedule of
// vt_igemm_span6.hip), and a GROUPED launch: up to kW6MaxGroup layers of one shape share a launch, so that the
// per-workgroup fixed costs (prologue, and above all the f32 atomic flush of its 64 x 64 x 9 accumulator tile, which
// runs at ~1.3 TB/s chip wide whatever else happens) are paid once per group instead of once per layer.
//
//   dw[n][t][c] += sum_pixels dz[pix][n] * x[pix + tap_t][c]      (0 outside the image)
//
// (autograd backward of the nn.Conv2d inside ConvNormAct, reference components.py:26-35, w.r.t. its weight.)
//
// What vt_wgrad_span.hip measured (NOTEBOOK R4.12): a 32-position step costs operands 34 + MFMAs 28 + transposing
// reads 23 + flush 24 + skeleton 18 of 123 us, ADDITIVE -- eight lock-step waves that each issue their own LDS-DMA,
// track their own stream position (~85 vector instructions per 20 MFMAs) and meet at one barrier per step.  Here:
//   * compute waves issue NO vector memory and track nothing: per 64-position step a wave reads its fragments
//     (ds_read_b64_tr_b16, both operands are pixel-major) in one TICK and issues 8 x taps MFMAs in the next, while
//     the other group's wave of the same SIMD does the opposite (one workgroup barrier per tick);
//   * the second K half of the x fragments is read during the MFMA tick into the registers the first half frees;
//   * loader waves 8..11 issue every LDS-DMA (dz slot ring, x position ring with the tap halo), PD steps ahead,
//     retired by counted vmcnt waits; the (image, row, column) tracking of the padded position streams lives there;
//   * pixels are enumerated in PADDED coordinates (vt_wgrad_span.hip): a tap is a constant row offset in the ring;
//   * the x ring is 512 rows x 128 B = 64 KiB at LDS address 0, so a ring offset wraps by a 16-bit mask.
// Products and their summation order inside a (layer, tile, pixel split) are those of vt_wgrad_span.hip's wide tile;
// the split boundaries differ (multiples of 64 positions), so results agree to f32 summation order.
#include <stdlib.h>

#include <type_traits>

#include "vt_common.h"

#ifndef VT_W6_SETPRIO
#define VT_W6_SETPRIO 1  // 1: s_setprio(1) around a compute wave's MFMA tick (measured: see NOTEBOOK R6.8)
#endif

// ablations exist only in -DVT_W6_ABL=<bits> builds (results wrong by construction, only the time is read):
//   1 no LDS-DMA inside the loop, 2 no MFMAs, 4 no fragment reads, 8 no flush
#ifndef VT_W6_ABL
#define VT_W6_ABL 0
#endif
#define VT_W6DBG(bit) ((VT_W6_ABL & (bit)) != 0)
// -DVT_W6_DIAG: shader-clock accounting per wave role (work between barriers vs waiting in them); never in the shipped build
#ifdef VT_W6_DIAG
#define VT_W6_T(...) __VA_ARGS__
#else
#define VT_W6_T(...)
#endif

namespace {

constexpr int kW6MaxGroup = 8;
constexpr int kStep = 64;                // positions per step
constexpr int kDzSlot = kStep * 128;     // bytes: 64 positions x 64 channels
constexpr int kRingRows = 512;           // x ring: 64 KiB at LDS address 0
constexpr unsigned kRingMask = kRingRows * 128 - 1;

struct W6Args {
    const bf16_t* x[kW6MaxGroup];
    const bf16_t* dz[kW6MaxGroup];
    float* dw[kW6MaxGroup];
    int G;                   // layers in this launch (same shape)
    int B, H, W, Cin, ldx, Cout, ldy, ldgw;
    int PH, PW, S, NP;       // padded rows / pitch / positions per image / total positions
    int dmin, NH;            // smallest tap offset, halo chunks (64 rows each)
    int tiles_n, tiles_c, chunk, split;
    int ntaps, tgn;          // taps (9) and how many of them compute group 0 owns (5)
    short o[9];              // d_t - dmin
    long rowx;               // elements between image rows of x
};

__device__ __attribute__((aligned(16))) unsigned int vt_w6_zero16[4];
#ifdef VT_W6_DIAG
__device__ unsigned long long vt_w6_diag[256 * 16];  // per workgroup: [0..3] loader 0: barrier wait, loop, vmcnt wait, steps | [4..6] g0 wave 0: wait, read tick, mfma tick | [8..10] g1 wave 4
#endif

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ void w6_set_m0(unsigned v) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(__builtin_amdgcn_readfirstlane(v)) : "memory");
}
__device__ __forceinline__ unsigned w6_get_m0() {
    unsigned v;
    asm volatile("s_mov_b32 %0, m0" : "=s"(v)::"memory");
    return v;
}
__device__ __forceinline__ void w6_glds(unsigned long gsrc) {
    asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(gsrc) : "memory");
}
template <int N>
__device__ __forceinline__ void w6_vmw() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void w6_barrier() {
    __builtin_amdgcn_sched_barrier(0);  // nothing migrates across a tick boundary
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef VT_W6_DIAG
#define VT_W6_BAR(wacc)                              \
    do {                                             \
        const unsigned long long c0_ = clock64();    \
        w6_barrier();                                \
        (wacc) += clock64() - c0_;                   \
    } while (0)
#else
#define VT_W6_BAR(wacc) w6_barrier()
#endif

struct W6Pos {  // a padded position, decomposed, and the element offset of its pixel
    int b, i, j, off;
    __device__ __forceinline__ void init(long P, int S, int PW, int H, long row, int ld) {
        long bb = P / S;
        long rem = P - bb * S;
        if (rem < 0) rem += S, --bb;
        b = (int)bb;
        i = (int)(rem / PW);
        j = (int)(rem - (long)i * PW);
        off = (int)(((long)b * H + i) * row + (long)j * ld);  // (mod 2^32 while b < 0; exact for every real pixel)
    }
};
struct W6PosStep {  // advance by kStep positions with additions only
    int q, r;        // kStep = q * PW + r
    int c0, c1, c2;  // offset deltas: kStep positions ahead | a column wrap | a row wrap (next image)
    __device__ __forceinline__ void init(int PH, int PW, int H, long row, int ld) {
        q = kStep / PW, r = kStep - q * PW;
        c0 = (int)(r * (long)ld + q * row);
        c1 = (int)(row - (long)PW * ld);
        c2 = (int)((long)(H - PH) * row);
    }
    __device__ __forceinline__ void advance(W6Pos& p, int PH, int PW) const {
        p.j += r, p.i += q, p.off += c0;
        if (p.j >= PW) p.j -= PW, ++p.i, p.off += c1;
        if (p.i >= PH) p.i -= PH, ++p.b, p.off += c2;
        if (p.i >= PH) p.i -= PH, ++p.b, p.off += c2;  // (q + 1 < 2 PH: checked by the launcher)
    }
};

template <int T>
using I_ = std::integral_constant<int, T>;

// 12 waves: 0-3 compute group 0 (taps 0..tgn-1), 4-7 compute group 1 (taps tgn..8), 8-11 loaders; three per SIMD
template <int PD>
__global__ void __launch_bounds__(768, 3) wgrad6_kernel(const W6Args p) {
    constexpr int NS = PD + 2;  // dz slots: a slot is read until two ticks after its step's first
    constexpr int TG = 5;       // taps of group 0 at most
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                      // [512 rows][128 B] at LDS address 0
    char* sDz = smem + kRingRows * 128;   // [NS][64 rows][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's item: (layer, pixel split, tile), tiles fastest -------------------------------------
    const unsigned item = vt_xcd_item(blockIdx.x, gridDim.x, 8);
    const int ntile = p.tiles_n * p.tiles_c;
    const int per_layer = ntile * p.split;
    if (item >= (unsigned)(per_layer * p.G)) return;
    const int layer = __builtin_amdgcn_readfirstlane((int)(item / (unsigned)per_layer));
    const int rem_i = (int)(item - (unsigned)layer * per_layer);
    const int bsplit = rem_i / ntile, btile = rem_i - bsplit * ntile;
    const int tile_n = btile % p.tiles_n, tile_c = btile / p.tiles_n;
    const int n0 = tile_n * 64, c0 = tile_c * 64;
    const long Pbeg = (long)bsplit * p.chunk;
    const long Pend = min((long)p.NP, Pbeg + p.chunk);
    if (Pbeg >= Pend) return;
    const int nsteps = __builtin_amdgcn_readfirstlane((int)((Pend - Pbeg + kStep - 1) / kStep));

    if (wave >= 8) {
        // =========================== loader waves ==================================================
        // a DMA instruction fills 8 rows x 128 B: lane l -> row l>>3, chunk slot l&7, and fetches source chunk
        // (l&7) ^ 2*((row>>1)&3).  Loader lj owns rows 16 lj .. 16 lj + 15 of every 64-row block: two instructions
        // (h = 0, 1) per operand and step.
        const int lj = wave - 8;
        const unsigned m0_keep = w6_get_m0();
        const int r8 = lane >> 3;
        const int srcc = (lane & 7) ^ (2 * ((r8 >> 1) & 3));
        const bool z_col_ok = n0 + srcc * 8 < p.Cout;
        const bool x_col_ok = c0 + srcc * 8 < p.Cin;
        const bf16_t* __restrict__ zb = p.dz[layer] + (n0 + srcc * 8);
        const bf16_t* __restrict__ xb = p.x[layer] + (c0 + srcc * 8);
        const unsigned long zero_src = (unsigned long)(const void*)vt_w6_zero16;
        const unsigned dz_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sDz;
        const int B_ = p.B, H_ = p.H, W_ = p.W, PH_ = p.PH, PW_ = p.PW;
        W6PosStep sz, sx;
        sz.init(PH_, PW_, H_, (long)W_ * p.ldy, p.ldy);
        sx.init(PH_, PW_, H_, p.rowx, p.ldx);
        W6Pos pz[2], px[2];
        int left[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long P0 = Pbeg + 16 * lj + 8 * h + r8;
            pz[h].init(P0, p.S, PW_, H_, (long)W_ * p.ldy, p.ldy);
            px[h].init(P0 + p.dmin, p.S, PW_, H_, p.rowx, p.ldx);
            left[h] = (int)(Pend - P0);  // dz rows at or past the split's end are zero (x rows there meet dz = 0)
        }
        const unsigned rowoff = (unsigned)(16 * lj) * 128u;
        auto issue_z = [&](int h, int slot) {
            const bool ok = z_col_ok && left[h] > 0 && (unsigned)pz[h].b < (unsigned)B_ && pz[h].i < H_ && pz[h].j < W_;
            w6_set_m0(dz_base + (unsigned)(slot * kDzSlot) + rowoff + (unsigned)h * 1024u);
            w6_glds(ok ? (unsigned long)(zb + (unsigned)pz[h].off) : zero_src);
            left[h] -= kStep;
            sz.advance(pz[h], PH_, PW_);
        };
        auto issue_x = [&](int h, int xc) {
            const bool ok = x_col_ok && (unsigned)px[h].b < (unsigned)B_ && px[h].i < H_ && px[h].j < W_;
            w6_set_m0((((unsigned)xc * 8192u) & kRingMask) + rowoff + (unsigned)h * 1024u);
            w6_glds(ok ? (unsigned long)(xb + (unsigned)px[h].off) : zero_src);
            sx.advance(px[h], PH_, PW_);
        };
        // ---- prologue: the halo chunks of x, then PD steps of both operands ----------------------------------
        int xc = 0;  // next x chunk (64 ring rows) to issue
        for (int hc = 0; hc < p.NH; ++hc) {
            issue_x(0, xc);
            issue_x(1, xc);
            ++xc;
        }
#pragma unroll
        for (int k = 0; k < PD; ++k)
            if (k < nsteps) {
                issue_z(0, k % NS);
                issue_x(0, xc);
                issue_z(1, k % NS);
                issue_x(1, xc);
                ++xc;
            }
        // Tick t starts with barrier t.  Group 0 reads step s in tick 2s (and the second K half of its x fragments in
        // tick 2s+1), group 1 one tick later: step s is in LDS before barrier 2s and its dz slot / oldest ring chunk are
        // free from barrier 2s+3 on.  In ticks 2s and 2s+1 this wave issues its four pieces of step s + PD.
        // (Round 5 also built, parity green, and measured NO faster in same-box A/B runs: the position tracking on the scalar
        //  unit; one tracker wave per operand with lane = row that hands the row addresses to the other loaders through an
        //  LDS table; loaders specialised per operand with the pieces dealt 3 / 5; that table pipelined one iteration ahead.
        //  Stamps (-DVT_W6_DIAG): every wave spends 30-45 % of a step waiting in the two tick barriers -- the ticks end with
        //  whichever of the twelve waves is slowest that time -- and a loader's LDS-DMA issue costs it 100-180 cycles a piece
        //  whatever surrounds it.  NOTEBOOK R5.3.)
        int zs = PD % NS;  // dz slot of the step being issued
        VT_W6_T(unsigned long long lwait = 0, lvm = 0; const unsigned long long lt0 = clock64();)
        for (int s = 0; s < nsteps; ++s) {
            VT_W6_T(const unsigned long long v0_ = clock64();)
            if (s + PD - 1 < nsteps) w6_vmw<4 * (PD - 1)>();  // the PD - 1 younger steps may stay in flight
            else w6_vmw<0>();
            VT_W6_T(lvm += clock64() - v0_;)
            VT_W6_BAR(lwait);
            const bool more = s + PD < nsteps && !VT_W6DBG(1);
            if (more) {
                issue_z(0, zs);
                issue_x(0, xc);
            }
            VT_W6_BAR(lwait);
            if (more) {
                issue_z(1, zs);
                issue_x(1, xc);
                ++xc;
            }
            zs = (zs + 1 == NS) ? 0 : zs + 1;
        }
        w6_barrier();  // tick 2 nsteps: group 1's last MFMA tick
        w6_vmw<0>();
        w6_set_m0(m0_keep);
#ifdef VT_W6_DIAG
        if (lj == 0 && lane == 0 && blockIdx.x < 256) {
            vt_w6_diag[blockIdx.x * 16 + 0] = lwait, vt_w6_diag[blockIdx.x * 16 + 1] = clock64() - lt0;
            vt_w6_diag[blockIdx.x * 16 + 2] = lvm, vt_w6_diag[blockIdx.x * 16 + 3] = (unsigned long long)nsteps;
        }
#endif
    } else {
        // =============================== compute waves ==================================================
        const int grp = wave >> 2, w4 = wave & 3;
        const int ntl = grp ? p.ntaps - p.tgn : p.tgn;  // taps this wave owns
        // ds_read_b64_tr_b16: lane 4q+pp of a 16-lane group addresses row q, columns 4pp..4pp+3 of a 4 x 16 block and
        // receives column u for the block's 4 rows: fragment element e<4 <-> position 4g+e, e>=4 <-> 16+4g+(e-4).
        const int g = lane >> 4, u = lane & 15, q = u >> 2, pp = u & 3;
        const int rowlo = 4 * g + q;
        unsigned a_off[4];  // dz slot-relative byte offsets: all 64 output channels (4 fragments)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = 2 * i + (pp >> 1);
            a_off[i] = (unsigned)(rowlo * 128 + ((ch ^ (2 * ((rowlo >> 1) & 3))) << 4) + 8 * (pp & 1));
        }
        unsigned bo[TG];  // ring byte offsets of this lane's first read of the coming step, per owned tap
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            const int t = min(grp * p.tgn + tt, p.ntaps - 1);
            const int row = rowlo + p.o[t];
            const int ch = 2 * w4 + (pp >> 1);  // this wave's 16 input channels
            bo[tt] = (unsigned)(row * 128 + ((ch ^ (2 * ((row >> 1) & 3))) << 4) + 8 * (pp & 1));
        }
        f32x4 acc[TG][4];
#pragma unroll
        for (int t = 0; t < TG; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (grp == 1) w6_barrier();  // tick 0: group 0 reads its first step

        auto run = [&](auto ntl_c) {
            constexpr int NTL = decltype(ntl_c)::value;
            constexpr int NB = NTL > 0 ? NTL : 1;
            int zs = 0;
            VT_W6_T(unsigned long long cwait = 0, cR = 0, cM = 0;)
            for (int s = 0; s < nsteps; ++s) {
                // ---- read tick: the step's dz slot and ring rows are in LDS -------------------------------------
                VT_W6_BAR(cwait);
                VT_W6_T(const unsigned long long r0_ = clock64();)
                const char* dzs = sDz + zs * kDzSlot;
                s16x4 a0l[4], a0h[4], a1l[4], a1h[4], b0l[NB], b0h[NB];
                if (!VT_W6DBG(4)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a0l[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dzs + a_off[i]));
                        a0h[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dzs + a_off[i] + 2048));
                    }
#pragma unroll
                    for (int tt = 0; tt < NTL; ++tt) {
                        b0l[tt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sX + bo[tt]));
                        b0h[tt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sX + ((bo[tt] + 2048u) & kRingMask)));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a1l[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dzs + a_off[i] + 4096));
                        a1h[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dzs + a_off[i] + 6144));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a0l[i] = a0h[i] = a1l[i] = a1h[i] = s16x4{(short)s, 1, 2, 3};
#pragma unroll
                    for (int tt = 0; tt < NB; ++tt) b0l[tt] = b0h[tt] = s16x4{(short)s, 3, 2, 1};
                }
                // (no wait here: the compiler's counted lgkmcnt waits sit in front of the MFMAs that use each fragment)
                __builtin_amdgcn_sched_barrier(0);
                VT_W6_T(cR += clock64() - r0_;)
                // ---- MFMA tick (the other group reads meanwhile) ---------------------------------------------
                VT_W6_BAR(cwait);
                VT_W6_T(const unsigned long long m0_ = clock64();)
#if VT_W6_SETPRIO
                __builtin_amdgcn_s_setprio(1);  // (as span6: the MFMA-issuing wave wins its SIMD's issue arbitration)
#endif
                bf16x8 af[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0l[i], a0h[i], 0, 1, 2, 3, 4, 5, 6, 7));
                s16x4 b1l[NB], b1h[NB];
#pragma unroll
                for (int tt = 0; tt < NTL; ++tt) {
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0l[tt], b0h[tt], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (VT_W6DBG(2)) {
                            asm volatile("" ::"v"(af[i]), "v"(bf));
                            continue;
                        }
                        acc[tt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[tt][i], 0, 0, 0);
                    }
                    // the second K half of this tap's x fragment, behind the MFMAs that freed the first
                    if (!VT_W6DBG(4)) {
                        b1l[tt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sX + ((bo[tt] + 4096u) & kRingMask)));
                        b1h[tt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sX + ((bo[tt] + 6144u) & kRingMask)));
                    } else {
                        b1l[tt] = b1h[tt] = s16x4{(short)s, 5, 2, 1};
                    }
                    bo[tt] = (bo[tt] + 8192u) & kRingMask;  // the next step's rows
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a1l[i], a1h[i], 0, 1, 2, 3, 4, 5, 6, 7));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < NTL; ++tt) {
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b1l[tt], b1h[tt], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (VT_W6DBG(2)) {
                            asm volatile("" ::"v"(af[i]), "v"(bf));
                            continue;
                        }
                        acc[tt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[tt][i], 0, 0, 0);
                    }
                }
                VT_W6_T(asm volatile("s_nop 0" ::"v"(acc[NB - 1][3][0])); cM += clock64() - m0_;)
#if VT_W6_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                zs = (zs + 1 == NS) ? 0 : zs + 1;
            }
#ifdef VT_W6_DIAG
            if (w4 == 0 && lane == 0 && blockIdx.x < 256) {
                const int o_ = blockIdx.x * 16 + 4 + 4 * grp;
                vt_w6_diag[o_] = cwait, vt_w6_diag[o_ + 1] = cR, vt_w6_diag[o_ + 2] = cM;
            }
#endif
        };
        switch (ntl) {
            case 5: run(I_<5>{}); break;
            case 4: run(I_<4>{}); break;
            default: run(I_<0>{}); break;  // (never launched: the launcher requires 9 taps, 5 | 4)
        }
        if (grp == 0) w6_barrier();  // tick 2 nsteps: group 1's last MFMA tick

        // ---- combine, part 1 happens below with all twelve waves ------------------------------------------------
        if (VT_W6DBG(8)) {
            float keep = 0.f;
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) keep += acc[t][i][0] + acc[t][i][1] + acc[t][i][2] + acc[t][i][3];
            if (keep == 12345.678f) p.dw[layer][0] = keep;
        } else {
            // tap pair (tt, tgn + tt) per round: each group stages its 64 x 64 tile in its own LDS image, then all 768
            // threads add whole rows with f32 atomics
            constexpr int PITCH = 64 + 4;
            constexpr int IMG = 64 * PITCH;
            float* sAcc = (float*)smem;
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                __syncthreads();
                if (tt < ntl) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            sAcc[grp * IMG + (16 * i + 4 * g + r) * PITCH + 16 * w4 + u] = acc[tt][i][r];
                }
                __syncthreads();
                if (tt >= p.tgn) continue;
                const int nimg = 1 + (p.tgn + tt < p.ntaps ? 1 : 0);
                float* __restrict__ dwl = p.dw[layer];
                for (int idx = tid; idx < nimg * 64 * 64; idx += 768) {
                    const int img = idx >> 12, e = idx & 4095;
                    const int n = e >> 6, c = e & 63;
                    const int t = img * p.tgn + tt;
                    if (n0 + n < p.Cout && c0 + c < p.Cin)
                        atomicAdd(dwl + ((long)(n0 + n) * p.ldgw + (long)t * p.Cin + c0 + c), sAcc[img * IMG + n * PITCH + c]);
                }
            }
            return;
        }
        return;
    }
    // loader waves take part in the flush (their barriers and a share of the atomics)
    if (!VT_W6DBG(8)) {
        constexpr int PITCH = 64 + 4;
        constexpr int IMG = 64 * PITCH;
        const float* sAcc = (const float*)smem;
#pragma unroll
        for (int tt = 0; tt < 5; ++tt) {
            __syncthreads();
            __syncthreads();
            if (tt >= p.tgn) continue;
            const int nimg = 1 + (p.tgn + tt < p.ntaps ? 1 : 0);
            float* __restrict__ dwl = p.dw[layer];
            for (int idx = tid; idx < nimg * 64 * 64; idx += 768) {
                const int img = idx >> 12, e = idx & 4095;
                const int n = e >> 6, c = e & 63;
                const int t = img * p.tgn + tt;
                if (n0 + n < p.Cout && c0 + c < p.Cin)
                    atomicAdd(dwl + ((long)(n0 + n) * p.ldgw + (long)t * p.Cin + c0 + c), sAcc[img * IMG + n * PITCH + c]);
            }
        }
    }
}

template <int PD>
int launch_w6(const W6Args& a, hipStream_t st) {
    constexpr int NS = PD + 2;
    const int smem = kRingRows * 128 + NS * kDzSlot;
    auto kern = wgrad6_kernel<PD>;
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, smem, "vt_conv_wgrad(wgrad6)");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("wgrad6_kernel<PD%d,G%d,split%d>", PD, a.G, a.split);
    const long items = (long)a.tiles_n * a.tiles_c * a.split * a.G;
    hipLaunchKernelGGL(kern, dim3(vt_xcd_grid(items)), dim3(768), smem, st, a);
    VT_CHECK_LAUNCH("vt_conv_wgrad(wgrad6)");
#ifdef VT_W6_DIAG
    {
        static int calls = 0;
        if (++calls % 8 == 0 && calls <= 64) {  // warm launches
            (void)hipStreamSynchronize(st);
            static unsigned long long h[256 * 16];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(vt_w6_diag), sizeof(h));
            double m_[12] = {0};
            const int nb = items < 256 ? (int)items : 256;
            for (int b = 0; b < nb; ++b)
                for (int k = 0; k < 12; ++k) m_[k] += (double)h[b * 16 + k] / nb;
            const double ns = m_[3] > 0 ? m_[3] : 1;
            fprintf(stderr, "[w6 diag G%d split%d] per step (shader cycles, mean over %d WGs, %.0f steps): loader0 loop %.0f = barrier wait %.0f + vmcnt wait %.0f + issue %.0f | "
                            "g0 wave: barrier wait %.0f, read tick %.0f, mfma tick %.0f | g1 wave: barrier wait %.0f, read tick %.0f, mfma tick %.0f\n",
                    a.G, a.split, nb, ns, m_[1] / ns, m_[0] / ns, m_[2] / ns, (m_[1] - m_[0] - m_[2]) / ns, m_[4] / ns, m_[5] / ns, m_[6] / ns, m_[8] / ns, m_[9] / ns, m_[10] / ns);
        }
    }
#endif
    return VT_OK;
}

}  // namespace

// G layers of ONE shape (descriptor d) in one launch.  -1 when this kernel does not apply (nothing was launched).
int vt_wgrad6_group(const vt_conv_desc* d, int G, const void* const* x, const void* const* dz, float* const* dw,
                    int32_t ldgw, void* stream) {
    const int enabled = VT_KNOB("VT_WGRAD6", 1);
    if (!enabled || G < 1 || G > kW6MaxGroup) return -1;
    if (vt_device_cus() != 256) return -1;  // one 12-wave workgroup per CU: sized for this chip
    if (d->dtype != VT_BF16 || d->ntaps != 9 || d->sh != 1 || d->sw != 1 || d->Ho != d->Hi || d->Wo != d->Wi) return -1;
    if (d->Cin <= 32 || d->Cout <= 32) return -1;  // (narrow layers: the 32-wide tiles of vt_wgrad_span.hip)
    int eh[9], ew[9];
    for (int t = 0; t < 9; ++t) {
        eh[t] = d->h0 + d->dh[t], ew[t] = d->w0 + d->dw[t];
        if (eh[t] < -1 || eh[t] > 1 || ew[t] < -1 || ew[t] > 1) return -1;
    }
    W6Args a;
    memset(&a, 0, sizeof(a));
    a.G = G;
    for (int g = 0; g < G; ++g) {
        a.x[g] = (const bf16_t*)x[g], a.dz[g] = (const bf16_t*)dz[g], a.dw[g] = dw[g];
        if (!vt_aligned16(x[g]) || !vt_aligned16(dz[g]) || !dw[g]) return -1;
    }
    a.B = d->B, a.H = d->Hi, a.W = d->Wi, a.Cin = d->Cin, a.ldx = d->ldx, a.Cout = d->Cout, a.ldy = d->ldy;
    a.ldgw = ldgw;
    a.rowx = (long)d->Wi * d->ldx;
    a.PH = d->Hi + 1, a.PW = d->Wi + 1;
    a.S = a.PH * a.PW;
    const long NP = (long)a.B * a.S;
    if (NP > 0x7ffffff0L || a.PW < 3 || a.PH < 3) return -1;
    if (kStep / a.PW + 1 >= 2 * a.PH) return -1;  // (the position advance wraps at most two rows of images)
    a.NP = (int)NP;
    a.ntaps = 9, a.tgn = 5;
    int dmin = 1 << 30, dmax = -(1 << 30), off[9];
    for (int t = 0; t < 9; ++t) {
        off[t] = eh[t] * a.PW + ew[t];
        dmin = off[t] < dmin ? off[t] : dmin;
        dmax = off[t] > dmax ? off[t] : dmax;
    }
    a.dmin = dmin;
    for (int t = 0; t < 9; ++t) a.o[t] = (short)(off[t] - dmin);
    a.NH = (dmax - dmin + kStep - 1) / kStep;
    // the ring holds chunks s .. s + 1 + PD + NH: PD + NH <= 6 (see the loader's tick accounting)
    const int PD = a.NH <= 3 ? 3 : 2;
    if (PD + a.NH > 6) return -1;
    a.tiles_n = (a.Cout + 63) / 64;
    a.tiles_c = (a.Cin + 63) / 64;
    // pixel split: one workgroup per CU and launch; with G layers in the launch every layer gets 256 / (G tiles) splits
    const long tiles = (long)a.tiles_n * a.tiles_c;
    long split = 256 / (tiles * G);
    const int min_steps = 24;  // (64-position steps; the flush is amortised over at least this many)
    const long max_split = (NP + (long)kStep * min_steps - 1) / ((long)kStep * min_steps);
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    long chunk = (NP + split - 1) / split;
    chunk = (chunk + kStep - 1) / kStep * kStep;
    split = (NP + chunk - 1) / chunk;
    a.chunk = (int)chunk, a.split = (int)split;
    return PD == 3 ? launch_w6<3>(a, (hipStream_t)stream) : launch_w6<2>(a, (hipStream_t)stream);
}

// single layer: -1 when this kernel does not apply (the caller then uses vt_wgrad_span.hip / the general kernel)
int vt_wgrad6_dispatch(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw, void* stream) {
    return vt_wgrad6_group(d, 1, &x, &dz, &dw, ldgw, stream);
}
