// vt_wgrad_span.hip -- filter gradient of the stride-1 3x3 convolutions (bf16) with both
// operands staged ONCE per pixel block for all nine taps.
//
//   dw[n][t][c] += sum_pixels dz[pix][n] * x[pix + tap_t][c]      (0 outside the image)
//
// (autograd backward of the nn.Conv2d inside ConvNormAct, reference components.py:26-35,
// w.r.t. its weight.)  vt_wgrad.hip gives every (128 out x 128 (tap,in)) tile its own
// workgroups, so a 3x3 layer pushes dz AND the tap-shifted x through the global->LDS path
// nine times (64 FLOP per staged byte: bound by that path at ~460 TFLOP/s), and layers with
// <= 64 channels fill a quarter of the tile.  Here a workgroup owns (32|64 out) x (32|64 in)
// x ALL 9 taps (9 accumulator tiles per wave) and walks the pixels once:
//   * Pixels are enumerated in PADDED coordinates: each image is (H+ph) x (W+pw) positions,
//     the extra row(s)/column(s) are zero pixels (their LDS rows come from a zero page).  A
//     tap is then a constant offset d_t = eh*(W+pw) + ew in that flat index -- the zero
//     column to the right of row i is also the zero column to the left of row i+1, the zero
//     row below image b also the one above image b+1 -- and no masks are needed anywhere.
//     Cost: (H+ph)(W+pw)/(HW) more MFMA work (1.07 at 28x28, 1.15 at 14x14).
//   * Per step of 32 positions one LDS-DMA instruction per wave brings 32 dz rows and one
//     brings the 32 NEW x rows into a power-of-two ring that always holds positions
//     [P + dmin, P + 32 + dmax + prefetch): 8 KB staged per 2.4 MFLOP (295 FLOP/B).
//   * Fragments are formed by ds_read_b64_tr_b16 (both operands are pixel-major = K-strided);
//     tap t reads the ring at row offset d_t - dmin.  Rows are 128 B (64 channels); the
//     16-byte chunk index is XOR-ed with 2*((row>>1)&3) through the DMA source address, which
//     makes any 8 consecutive rows hit 8 distinct 32-byte bank groups, so the shifted reads
//     are conflict free for every tap.
// The pixel range is split over blockIdx.y; partial tiles are staged through LDS into whole
// rows and added with row-contiguous f32 atomics, as in vt_wgrad.hip.
#include <stdlib.h>

#include <type_traits>

#include "vt_common.h"

#ifndef VT_WS_PD
#define VT_WS_PD 2
#endif
// ablations exist only in -DVT_WS_ABL=<bits> builds (tools/ws_ablate.sh; compile-time, so that the measured kernel keeps
// the shipped one's registers and schedule): results are wrong by construction, only the time is read
//   1 no LDS-DMA inside the loop, 2 no MFMAs, 4 no fragment reads, 8 no flush, 16 no barrier
#ifndef VT_WS_ABL
#define VT_WS_ABL 0
#endif
#define VT_WSDBG(bit) ((VT_WS_ABL & (bit)) != 0)

namespace {

struct WsArgs {
    const bf16_t* x;
    const bf16_t* dz;
    float* dw;
    int B, H, W, Cin, ldx, Cout, ldy, ldgw;
    int PH, PW, S, NP;       // padded rows / pitch / positions per image / total positions
    int dmin, NH, RX;        // smallest tap offset, halo chunks, ring rows (power of two)
    int tiles_n, tiles_c, chunk, ablate;
    short o[9];              // d_t - dmin
    int ntaps, tgn;          // taps (<= 9) and how many of them the first tap group owns (<= 5)
    int split, xcds;         // pixel splits; 8 = XCD-blocked item order (vt_xcd_item), 1 = identity
    long rowx;               // elements between image rows of x (W*ldx for a dense tensor; larger for a row-parity view)
    float* slab;             // partial tiles go to slab[blockIdx.y * slab_stride + ...] with plain stores (no atomics)
    long slab_stride;        // elements per pixel split: Cout * ldgw
    int cblk, cin_dst;       // flush map (cblk > 0): input channel c -> destination tap map[t][c / cblk], channel c % cblk
    signed char map[9][4];   // (-1: the column is dropped)
};

__device__ __attribute__((aligned(16))) unsigned int vt_ws_zero16[4];

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ void glds16(unsigned long gsrc, unsigned lds_base_v) {
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_base_v);  // (wave-uniform by construction)
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

template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Pos {  // a padded position, decomposed, and the element offset of its pixel; advanced 32 positions per step
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
// The advance is additions only: round 2 recomputed (b, i, j) by multiply-high quotients and the offset by 64-bit
// multiplies every step -- 18 quarter-rate integer multiplies among ~100 vector instructions per wave and step, against
// 20 MFMAs: the loop was bound by vector issue (rocprofv3 PMC, 128 -> 128 @28x28: SQ_INSTS_VALU 22.4 M, SQ_INSTS_MFMA
// 3.9 M, LDS array 16 % busy, no bank conflicts), not by the transposing reads.
struct PosStep {
    int q32, r32;    // 32 = q32 * PW + r32
    int c0, c1, c2;  // offset deltas: 32 positions ahead | a column wrap | a row wrap (next image)
    __device__ __forceinline__ void init(int PH, int PW, int H, long row, int ld) {
        q32 = 32 / PW, r32 = 32 - q32 * PW;
        c0 = (int)(r32 * (long)ld + q32 * row);
        c1 = (int)(row - (long)PW * ld);
        c2 = (int)((long)(H - PH) * row);
    }
    __device__ __forceinline__ void advance(Pos& p, int PH, int PW) const {
        p.j += r32, p.i += q32, p.off += c0;
        if (p.j >= PW) p.j -= PW, ++p.i, p.off += c1;
        if (p.i >= PH) p.i -= PH, ++p.b, p.off += c2;
        if (p.i >= PH) p.i -= PH, ++p.b, p.off += c2;  // (q32 + 1 < 2 PH: launch_ws_taps)
    }
};

constexpr int kDzSlot = 32 * 128;  // bytes: 32 positions x 64 channels

// 8 waves: two tap groups (taps 0..4 and 5..8) x a 2 x 2 wave grid; FI x FJ 16x16 accumulator
// tiles per wave and tap: tile = 32FI out x 32FJ in x 9 taps per workgroup.  ONE workgroup per CU
// and launch: the atomic bytes of a launch are (#workgroups x tile bytes), and global float
// atomics run at ~1.3 TB/s chip wide whatever else happens, so the tile is spread over as many
// waves as the CU holds instead of giving each 4-wave group its own copy.
// The four waves of tap group 0 bring in dz, those of group 1 the new x rows: one LDS-DMA
// instruction and one position counter per wave and step.
// WIDE (64 x 64 tiles only): a wave owns ALL 64 output channels x 16 input channels (4 x 1 fragments) instead of
// 32 x 32 (2 x 2).  Same accumulators, but 8 + 2*taps transposing reads per step instead of 4 + 4*taps: the 2 x 2
// grid is LDS-read bound (8 waves x 24 reads x 4 cycles = 768 LDS cycles per step against 640 MFMA cycles per SIMD).
template <int FI, int FJ, int PD, bool WIDE = false>
__global__ void __launch_bounds__(512, 1) wgrad_span_kernel(const WsArgs p) {
    static_assert(!WIDE || (FI == 2 && FJ == 2), "WIDE is the 64 x 64 tile");
    constexpr int NS = PD + 1;
    constexpr int NI = 32 * FI, NC = 32 * FJ;
    constexpr int WI = WIDE ? 2 * FI : FI, WJ = WIDE ? FJ / 2 : FJ;  // fragments per wave
    constexpr int TG = 5;  // taps per group (the second group owns 4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                   // [RX rows][128 B]: at LDS address 0, so a masked ring offset IS the address
    char* sDz = smem + p.RX * 128;     // [NS][32 rows][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wave8 >> 2, wave = wave8 & 3;
    const int wn = wave >> 1, wc = wave & 1;
    const int rbase = WIDE ? 0 : wn * 16 * FI;        // first output channel of this wave inside the tile
    const int cbase = WIDE ? wave * 16 : wc * 16 * FJ;  // first input channel
    const int ntl = tg ? p.ntaps - p.tgn : p.tgn;  // taps this wave owns
    const unsigned item = vt_xcd_item(blockIdx.x, gridDim.x, p.xcds);  // (pixel split, tile), tiles fastest
    const int ntile = p.tiles_n * p.tiles_c;
    if (item >= (unsigned)(ntile * p.split)) return;
    const int bsplit = (int)(item / (unsigned)ntile), btile = (int)(item - (unsigned)bsplit * ntile);
    const int tile_n = btile % p.tiles_n, tile_c = btile / p.tiles_n;
    const int n0 = tile_n * NI, c0 = tile_c * NC;
    const long Pbeg = (long)bsplit * p.chunk;
    const long Pend = min((long)p.NP, Pbeg + p.chunk);
    if (Pbeg >= Pend) return;
    const int nsteps = (int)((Pend - Pbeg + 31) / 32);

    const unsigned long zero_src = (unsigned long)(const void*)vt_ws_zero16;
    const unsigned dz_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sDz;
    const unsigned x_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sX;
    const unsigned xmask = (unsigned)p.RX * 128u - 1u;

    // ---- DMA geometry: an instruction fills 8 rows x 128 B; lane l -> row l>>3, chunk slot l&7,
    // and fetches source chunk (l&7) ^ 2*((row>>1)&3).  Wave w of a group owns rows 8w..8w+7 of
    // each 32-row block (32-row blocks start at multiples of 8 in the ring).
    const int r8 = lane >> 3;
    const int srcc = (lane & 7) ^ (2 * ((r8 >> 1) & 3));  // (8w + r8)>>1 & 3 == (r8>>1)&3
    const bool col_ok = tg ? (srcc * 8 < NC && c0 + srcc * 8 < p.Cin) : (srcc * 8 < NI && n0 + srcc * 8 < p.Cout);
    const bf16_t* __restrict__ src_base = tg ? p.x + (c0 + srcc * 8) : p.dz + (n0 + srcc * 8);
    const int src_ld = tg ? p.ldx : p.ldy;
    const long src_row = tg ? p.rowx : (long)p.W * p.ldy;
    Pos ps;
    PosStep pst;
    pst.init(p.PH, p.PW, p.H, src_row, src_ld);
    const long Ps0 = Pbeg + (tg ? p.dmin : 0) + 8 * wave + r8;  // this lane's stream position
    ps.init(Ps0, p.S, p.PW, p.H, src_row, src_ld);
    int left = tg ? 0x7fffffff : (int)(Pend - Ps0);  // dz: positions until the split's end (x: never runs out)
    int nissued = 0;  // DMA instructions this wave has issued (x: ring chunk index)

    // dz rows beyond the split's end must be zero; x rows may be anything there (times dz = 0)
#define VT_WS_ISSUE(dst)                                                                          \
    do {                                                                                          \
        const bool ok = col_ok && left > 0 && (unsigned)ps.b < (unsigned)p.B && ps.i < p.H && ps.j < p.W; \
        glds16(ok ? (unsigned long)(src_base + (unsigned)ps.off) : zero_src, (dst) + (unsigned)wave * 1024u); \
        ++nissued;                                                                                \
        if (!tg) left -= 32;                                                                      \
        pst.advance(ps, p.PH, p.PW);                                                              \
    } while (0)
#define VT_WS_ISSUE_STEP(slot)                                                                    \
    do {                                                                                          \
        if (tg)                                                                                   \
            VT_WS_ISSUE(x_base + (((unsigned)nissued * 4096u) & xmask));                          \
        else                                                                                      \
            VT_WS_ISSUE(dz_base + (unsigned)((slot)*kDzSlot));                                    \
    } while (0)

    // ---- fragment addressing -----------------------------------------------------------------
    // ds_read_b64_tr_b16: lane 4q+pp of a 16-lane group addresses row q, columns 4pp..4pp+3 of a
    // 4 x 16 block and receives column u for the block's 4 rows: fragment element e<4 <-> position
    // 4g+e, e>=4 <-> 16+4g+(e-4), for BOTH operands.
    const int g = lane >> 4, u = lane & 15, q = u >> 2, pp = u & 3;
    const int rowlo = 4 * g + q;
    unsigned a_off[WI];  // dz slot-relative byte offsets
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int ch = ((rbase + 16 * i) >> 3) + (pp >> 1);
        a_off[i] = (unsigned)(rowlo * 128 + ((ch ^ (2 * ((rowlo >> 1) & 3))) << 4) + 8 * (pp & 1));
    }
    unsigned b_off[TG];  // ring-relative byte offsets of the j = 0 fragment at step 0, per owned tap
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
        const int t = min(tg * p.tgn + tt, p.ntaps - 1);
        const int row = rowlo + p.o[t];
        const int ch = (cbase >> 3) + (pp >> 1);
        b_off[tt] = (unsigned)(row * 128 + ((ch ^ (2 * ((row >> 1) & 3))) << 4) + 8 * (pp & 1));
    }

    f32x4 acc[TG][WI][WJ];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < WI; ++i)
#pragma unroll
            for (int j = 0; j < WJ; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: halo chunks (x waves), then PD steps ahead -----------------------------------
    if (tg)
        for (int h = 0; h < p.NH; ++h) VT_WS_ISSUE(x_base + (((unsigned)nissued * 4096u) & xmask));
#pragma unroll
    for (int s = 0; s < PD; ++s)
        if (s < nsteps) VT_WS_ISSUE_STEP(s);

    int cur = 0, nxt = PD % NS;
    // one step: NTL = the taps this wave owns, a compile-time count (the guarded `if (tt < ntl)` of round 2 kept every
    // tap's reads behind the previous tap's MFMAs); YOUNGER = DMA steps that may stay in flight across the wait
    auto step = [&](auto ntl_c, auto younger_c, int s) {
        constexpr int NTL = decltype(ntl_c)::value;
        // dz(s) and x chunk s+NH must have landed; the younger steps (1 instruction per wave each) may stay in flight
        vm_wait<decltype(younger_c)::value>();
        if (!VT_WSDBG(16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        // all of the step's fragment reads first (18 in flight), the DMA issue's address arithmetic under their latency,
        // then the MFMAs behind counted waits
        const char* dzs = sDz + cur * kDzSlot;
        s16x4 alo[WI], ahi[WI];
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            if (VT_WSDBG(4)) {
                alo[i] = ahi[i] = s16x4{(short)s, 1, 2, 3};
                continue;
            }
            alo[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dzs + a_off[i]));
            ahi[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dzs + a_off[i] + 16 * 128));
        }
        const unsigned sb = (unsigned)s * 4096u;
        s16x4 blo[NTL > 0 ? NTL : 1][WJ], bhi[NTL > 0 ? NTL : 1][WJ];
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt) {
            const unsigned lo_o = (sb + b_off[tt]) & xmask;
            const unsigned hi_o = (lo_o + 16u * 128u) & xmask;
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                if (VT_WSDBG(4)) {
                    blo[tt][j] = bhi[tt][j] = s16x4{(short)lo_o, 1, 2, 3};
                    continue;
                }
                // the next 16 input channels sit one 32-byte group over: chunk index ^ 2
                blo[tt][j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sX + (lo_o ^ (32u * j))));
                bhi[tt][j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sX + (hi_o ^ (32u * j))));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + PD < nsteps && !VT_WSDBG(1)) VT_WS_ISSUE_STEP(nxt);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 af[WI];
#pragma unroll
        for (int i = 0; i < WI; ++i)
            af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt) {
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo[tt][j], bhi[tt][j], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int i = 0; i < WI; ++i) {
                    if (VT_WSDBG(2)) {
                        asm volatile("" ::"v"(af[i]), "v"(bf));
                        continue;
                    }
                    acc[tt][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[tt][i][j], 0, 0, 0);
                }
            }
        }
        cur = (cur + 1 == NS) ? 0 : cur + 1;
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    };
    auto run = [&](auto ntl_c) {
        using Y = std::integral_constant<int, PD - 1>;
        using Z = std::integral_constant<int, 0>;
        int s = 0;
        for (; s + PD - 1 < nsteps; ++s) step(ntl_c, Y{}, s);  // PD - 1 younger steps exist
        for (; s < nsteps; ++s) step(ntl_c, Z{}, s);            // the last PD - 1 steps: wait for everything
    };
    switch (ntl) {  // 9 taps: 5 | 4; the stride-2 views: 2 | 2 and 1 | 1 (launch_ws_taps)
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        default: run(std::integral_constant<int, 0>{}); break;  // (a group without taps still brings in its operand)
    }
#undef VT_WS_ISSUE
#undef VT_WS_ISSUE_STEP

    // ---- combine: tap pair (tt, TG+tt) per round; each group stages its NI x NC tile in its own
    // LDS image, then all 512 threads add whole rows with f32 atomics ------------------------------
    constexpr int PITCH = NC + 4;
    constexpr int IMG = NI * PITCH;
    float* sAcc = (float*)smem;
    if (VT_WSDBG(8)) {
        float keep = 0.f;
#pragma unroll
        for (int t = 0; t < TG; ++t)
#pragma unroll
            for (int i = 0; i < WI; ++i)
#pragma unroll
                for (int j = 0; j < WJ; ++j) keep += acc[t][i][j][0] + acc[t][i][j][1] + acc[t][i][j][2] + acc[t][i][j][3];
        if (keep == 12345.678f) p.dw[0] = keep;
        return;
    }
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
        __syncthreads();
        if (tt < ntl) {
#pragma unroll
            for (int i = 0; i < WI; ++i)
#pragma unroll
                for (int j = 0; j < WJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sAcc[tg * IMG + (rbase + 16 * i + 4 * g + r) * PITCH + cbase + 16 * j + u] = acc[tt][i][j][r];
        }
        __syncthreads();
        const int nimg = (tt < p.tgn ? 1 : 0) + (p.tgn + tt < p.ntaps ? 1 : 0);  // (tt >= tgn: nothing; group 1 never owns more)
        if (tt >= p.tgn) continue;
        for (int idx = tid; idx < nimg * NI * NC; idx += 512) {
            const int img = idx / (NI * NC), e = idx % (NI * NC);
            const int n = e / NC, c = e % NC;
            const int t = img * p.tgn + tt;
            if (n0 + n < p.Cout && c0 + c < p.Cin) {
                long col = (long)t * p.Cin + c0 + c;
                if (p.cblk) {
                    const int blk = (c0 + c) / p.cblk;
                    const int td = p.map[t][blk];
                    if (td < 0) continue;
                    col = (long)td * p.cin_dst + (c0 + c - blk * p.cblk);
                }
                if (p.slab)
                    p.slab[(long)bsplit * p.slab_stride + (long)(n0 + n) * p.ldgw + col] = sAcc[img * IMG + n * PITCH + c];
                else
                    atomicAdd(p.dw + ((long)(n0 + n) * p.ldgw + col), sAcc[img * IMG + n * PITCH + c]);
            }
        }
    }
}

// Steps of LDS-DMA in flight per wave (-DVT_WS_PD=2..6).  Measured 2 .. 6 on 128->128 @28/56, 64->64 @56, 32->32 @112 and
// the stride-2 views: no difference (0.124 / 0.337 / 0.126 / 0.235 ms at every depth), i.e. the 0.76 us step (a third
// of the MFMA rate) is not memory latency.  Also measured and dropped: a two-tick schedule (tap group 0 reads its fragments
// while group 1 multiplies the previous step from registers, then the reverse; two barriers per step): 0.180 instead of
// 0.125 ms at 128->128 @28x28, 0.526 instead of 0.334 at @56x56 -- the hardware already overlaps one wave's LDS reads
// with the other's MFMAs, and the second barrier costs ~0.45 us per step.
constexpr int kWsPD = VT_WS_PD;
static_assert(kWsPD >= 2 && kWsPD <= 6, "vm_wait switch covers PD - 1 <= 5");

template <int FI, int FJ, bool WIDE = false>
int launch_ws(WsArgs& a, long split, hipStream_t st) {
    a.split = (int)split;
    a.xcds = (8);
    constexpr int PD = kWsPD;
    const int rings = (PD + 1) * kDzSlot + a.RX * 128;
    const int image = 2 * 32 * FI * (32 * FJ + 4) * 4;
    const int smem = rings > image ? rings : image;
    auto kern = wgrad_span_kernel<FI, FJ, PD, WIDE>;
    {
        const int rc = vt_raise_dynamic_lds((const void*)kern, 96 * 1024, "vt_conv_wgrad(span)");
        if (rc != VT_OK) return rc;
    }
    vt_note_kernel("wgrad_span_kernel<%d,%d,%d%s>", FI, FJ, PD, WIDE ? ",wide" : "");
    hipLaunchKernelGGL(kern, dim3(vt_xcd_grid((long)a.tiles_n * a.tiles_c * split)), dim3(512), smem, st, a);
    VT_CHECK_LAUNCH("vt_conv_wgrad(span)");
    return VT_OK;
}

// ring geometry, pixel split and launch for a filled-in WsArgs (x, dz, dw, B, H, W, Cin, ldx, Cout, ldy, ldgw, rowx, PH, PW
// and the flush map are set); eh/ew = tap offsets in the position grid.  -1 when the ring would not fit.
}  // namespace

int vt_wgrad_reduce_slabs(const float* slab, long stride, int split, float* dw, int Cout, int Ktot, int ldgw,
                          hipStream_t st);  // vt_wgrad.hip

namespace {

int launch_ws_taps(WsArgs& a, int ntaps, const int* eh, const int* ew, hipStream_t st, float* scratch = nullptr,
                   long scratch_bytes = 0) {
    a.S = a.PH * a.PW;
    const long NP = (long)a.B * a.S;
    if (NP > 0x7ffffff0L || a.PW < 2 || a.PH < 2) return -1;
    if (32 / a.PW + 1 >= 2 * a.PH) return -1;  // (the position advance of the kernel wraps at most two rows of images)
    a.NP = (int)NP;
    a.ntaps = ntaps, a.tgn = (ntaps + 1) / 2;
    int dmin = 1 << 30, dmax = -(1 << 30), off[9];
    for (int t = 0; t < ntaps; ++t) {
        off[t] = eh[t] * a.PW + ew[t];
        dmin = off[t] < dmin ? off[t] : dmin;
        dmax = off[t] > dmax ? off[t] : dmax;
    }
    a.dmin = dmin;
    for (int t = 0; t < 9; ++t) a.o[t] = (short)(off[t < ntaps ? t : ntaps - 1] - dmin);
    constexpr int PD = kWsPD;
    a.NH = (31 + dmax - dmin) / 32;
    int rx = 64;
    while (rx < 32 * (a.NH + PD + 1)) rx *= 2;
    if (rx > 512) return -1;  // wide maps: the ring would not leave room for two workgroups per CU
    a.RX = rx;
    const int FI = a.Cout > 32 ? 2 : 1, FJ = a.Cin > 32 ? 2 : 1;
    a.tiles_n = (a.Cout + 32 * FI - 1) / (32 * FI);
    a.tiles_c = (a.Cin + 32 * FJ - 1) / (32 * FJ);
    // pixel split: one 8-wave workgroup per CU, at least 16 steps each (the halo warm-up is NH chunks)
    const int target = (256);
    const long tiles = (long)a.tiles_n * a.tiles_c;
    long split = target / tiles;
    // (>= 144 steps per workgroup (96 .. 200 measured alike): at batch 128 the 256-workgroup target cut the 28x28 layers into 52-step pieces, and half
    //  as many workgroups of twice the length measured 12.54 vs 12.65 ms per step; batch 256 is unchanged by this bound)
    const int min_steps = (144);
    const long max_split = (NP + 32L * min_steps - 1) / (32L * min_steps);
    if (split > max_split) split = max_split;
    if (scratch && !a.cblk && split * (long)a.Cout * a.ldgw * 4 > scratch_bytes)
        split = scratch_bytes / ((long)a.Cout * a.ldgw * 4);  // two-stage mode: fewer, longer splits rather than atomics
    if (split < 1) split = 1;
    long chunk = (NP + split - 1) / split;
    chunk = (chunk + 31) / 32 * 32;
    split = (NP + chunk - 1) / chunk;
    a.chunk = (int)chunk;
    a.ablate = 0;
    a.slab_stride = (long)a.Cout * a.ldgw;
    const bool use_slabs = scratch && !a.cblk && split > 1 && split * a.slab_stride * 4 <= scratch_bytes &&
                           (ntaps * a.Cin) % 4 == 0 && a.ldgw % 4 == 0;
    a.slab = use_slabs ? scratch : nullptr;
    int rc;
    const int wide = (1);
    if (FI == 2 && FJ == 2)
        rc = wide ? launch_ws<2, 2, true>(a, split, st) : launch_ws<2, 2>(a, split, st);
    else if (FI == 2)
        rc = launch_ws<2, 1>(a, split, st);
    else if (FJ == 2)
        rc = launch_ws<1, 2>(a, split, st);
    else
        rc = launch_ws<1, 1>(a, split, st);
    if (rc == VT_OK && use_slabs)
        rc = vt_wgrad_reduce_slabs(scratch, a.slab_stride, (int)split, a.dw, a.Cout, ntaps * a.Cin, a.ldgw, st);
    return rc;
}

}  // namespace

// returns -1 when this kernel does not apply (the caller then uses the general kernel)
int vt_wgrad_span_dispatch(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw,
                           float* scratch, int64_t scratch_bytes, void* stream) {
    const int enabled = (1);
    if (!enabled) return -1;
    if (d->dtype != VT_BF16 || d->ntaps != 9 || d->sh != 1 || d->sw != 1 || d->Ho != d->Hi || d->Wo != d->Wi)
        return -1;
    // 14x14 and 7x7 maps with wide layers: the padded enumeration costs 15-30 % extra MFMA work, and in round 2 the
    // general kernel, whose operands stay L2 resident there, was faster ALONE (97 vs 115 us at 256ch 14x14) and those
    // layers went to it.  Inside the step the all-taps kernel wins since round 3 (one 8-wave workgroup per CU next to the
    // main stream's kernels: 21.46 vs 21.55 ms, alternating runs): VT_WGRAD_SPAN_MINW = 20 restores the old rule.
    const int minw_wide = (0);
    if (enabled < 2 && d->Wi < minw_wide && d->Cin > 64 && d->Cout > 64) return -1;
    int ph = 0, pw = 0, eh[9], ew[9];
    for (int t = 0; t < 9; ++t) {
        eh[t] = d->h0 + d->dh[t], ew[t] = d->w0 + d->dw[t];
        ph = abs(eh[t]) > ph ? abs(eh[t]) : ph;
        pw = abs(ew[t]) > pw ? abs(ew[t]) : pw;
    }
    WsArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)x, a.dz = (const bf16_t*)dz, a.dw = dw;
    a.B = d->B, a.H = d->Hi, a.W = d->Wi, a.Cin = d->Cin, a.ldx = d->ldx, a.Cout = d->Cout, a.ldy = d->ldy;
    a.ldgw = ldgw;
    a.rowx = (long)d->Wi * d->ldx;
    a.PH = d->Hi + ph, a.PW = d->Wi + pw;
    return launch_ws_taps(a, 9, eh, ew, (hipStream_t)stream, scratch, (long)scratch_bytes);
}

// Filter gradient of a 3x3 STRIDE-2 convolution (the first conv of every Darknet / CSPDarknet stage,
// darknet.py:33 / :43) on the all-taps kernel.  With an even map the input is, without moving a byte,
//   X_even[b][i][j][(b', c)] = x[b][2i  ][2j + b'][c]      X_odd[b][i][j][(b', c)] = x[b][2i+1][2j + b'][c]
// (pixel pairs = 2*Cin contiguous channels; rows two image rows apart), and output pixel (i, j) reads
//   filter row 1 from X_even(i, j-1 .. j),   rows 0 / 2 from X_odd(i-1, j-1 .. j) / X_odd(i, j-1 .. j):
// two stride-1 launches (2 and 4 taps over 2*Cin channels) whose columns are scattered to the 9 x Cin
// filter taps by the flush map; of the (tap, pixel-half) pairs those that fall on pixel 2j-2 are dropped
// (25 % of the MFMA work).  The general kernel gives each 128-column slice of the 9*Cin columns its own
// workgroups and 128 output channels per tile: at 32 -> 64 channels it fetched 2.6 GB for 1.2 GB of operands.
int vt_wgrad_span_s2_dispatch(const vt_conv_desc* d, const void* x, const void* dz, float* dw, int32_t ldgw,
                              void* stream) {
    // Measured (CSPDarknet-53 / VoVNet-39 steps): per 32-position step the all-taps kernel costs ~0.55 us whatever the
    // tap count, so two launches of 2 and 4 taps only match the general kernel's time where that one is at its worst
    // (32 -> 64 @224->112: 0.49 vs 0.47 ms alone, but 1.6 instead of 2.6 GB fetched: -0.1 ms per step beside the
    // HBM-bound BatchNorm passes); at 64 -> 128 @112->56 it is 0.44 vs 0.29 ms, on the 8-channel VoVNet stem slower too.
    // (round 3: off by default -- with the XCD-blocked work order the general kernel no longer re-fetches its operands
    //  2.7x on these layers: 32 -> 64 @224->112 0.41 ms either way alone, the step 21.49 vs 21.55 ms)
    const int minw = VT_KNOB("VT_WGRAD_S2_MINW", 0);
    const int minc = VT_KNOB("VT_WGRAD_S2_MINC", 32);
    if (d->Cin < minc) return -1;
    if (minw <= 0 || d->dtype != VT_BF16 || d->ntaps != 9 || d->sh != 2 || d->sw != 2 || d->h0 != -1 || d->w0 != -1)
        return -1;
    if (d->Hi % 2 || d->Wi % 2 || d->Ho != d->Hi / 2 || d->Wo != d->Wi / 2 || d->Wo < minw) return -1;
    if (d->ldx != d->Cin || d->Cin % 8 || d->Cin > 127) return -1;  // pixel pairs must be contiguous; cblk fits the map
    for (int t = 0; t < 9; ++t)
        if (d->dh[t] != t / 3 || d->dw[t] != t % 3) return -1;
    WsArgs a;
    memset(&a, 0, sizeof(a));
    a.dz = (const bf16_t*)dz, a.dw = dw;
    a.B = d->B, a.H = d->Ho, a.W = d->Wo, a.Cin = 2 * d->Cin, a.ldx = 2 * d->ldx, a.Cout = d->Cout, a.ldy = d->ldy;
    a.ldgw = ldgw;
    a.rowx = 2L * d->Wi * d->ldx;
    a.cblk = d->Cin, a.cin_dst = d->Cin;
    memset(a.map, -1, sizeof(a.map));
    // odd rows: filter rows 0 (grid row i-1) and 2 (grid row i)
    {
        WsArgs o = a;
        o.x = (const bf16_t*)x + (long)d->Wi * d->ldx;
        o.PH = o.H + 1, o.PW = o.W + 1;
        o.map[0][1] = 0, o.map[1][0] = 1, o.map[1][1] = 2;
        o.map[2][1] = 6, o.map[3][0] = 7, o.map[3][1] = 8;
        const int eh[4] = {-1, -1, 0, 0}, ew[4] = {-1, 0, -1, 0};
        const int rc = launch_ws_taps(o, 4, eh, ew, (hipStream_t)stream);
        if (rc != VT_OK) return rc;  // (-1 before anything was launched: the caller falls back; this launch has the larger ring)
    }
    // even rows: filter row 1.  tap 0 = (0, -1): pixel half 1 is column q = 0; tap 1 = (0, 0): halves 0 / 1 are q = 1 / 2
    {
        WsArgs e = a;
        e.x = (const bf16_t*)x;
        e.PH = e.H, e.PW = e.W + 1;
        e.map[0][1] = 3, e.map[1][0] = 4, e.map[1][1] = 5;
        const int eh[2] = {0, 0}, ew[2] = {-1, 0};
        const int rc = launch_ws_taps(e, 2, eh, ew, (hipStream_t)stream);
        if (rc != VT_OK) return rc == -1 ? VT_ERR_UNSUPPORTED : rc;  // the odd rows are already in dw
    }
    return VT_OK;
}
