// mfma_peak.hip -- what the MFMA pipe of THIS device sustains, without any memory traffic.
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
//
// Every wave keeps 8 A and 4 B fragments in registers (random bf16 or zeros) and issues
// v_mfma_f32_16x16x32_bf16 into 32 accumulator tiles, back to back, for a fixed count.  Reports
// TFLOP/s and the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz) for 1 and 2
// waves per SIMD.  Context for DESIGN.md 4.2: the conv kernels' roofline fraction is quoted
// against the 2.5 PFLOP/s spec peak; this prints what the chip holds under an MFMA-only load.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void __launch_bounds__(256) mfma_loop(const uint4* __restrict__ src, float* __restrict__ sink, int iters,
                                                 unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    uint4 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[(threadIdx.x * 12 + i) & 4095];
    for (int j = 0; j < 4; ++j) b[j] = src[(threadIdx.x * 12 + 8 + j) & 4095];
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                    __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;  // keep the accumulators live
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
    (void)lane;
}

int main() {
    const int iters = 4000;
    uint4* src;
    float* sink;
    unsigned long long* clk;
    hipMalloc(&src, 4096 * 16);
    hipMalloc(&sink, 64);
    hipMalloc(&clk, 4096 * 16);
    std::vector<unsigned short> h(4096 * 8);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (auto& v : h) {
            // bf16 of a uniform [-1,1) value (mode 0) or zero (mode 1)
            float f = mode ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
            unsigned u;
            memcpy(&u, &f, 4);
            v = (unsigned short)(u >> 16);
        }
        hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 2; ++wps) {
            const int blocks = 256 * wps;  // 4 waves per block: 1 or 2 waves per SIMD
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, src, sink, iters, clk);
            hipEventRecord(e0, 0);
            const int reps = 10;
            for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, src, sink, iters, clk);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> c(blocks * 2);
            hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
            double ghz = 0;
            for (int b = 0; b < blocks; ++b) ghz += (double)c[b * 2] / (double)c[b * 2 + 1] * 0.1;
            ghz /= blocks;
            const double flops = (double)reps * blocks * 4 * iters * 32 * (16.0 * 16 * 32 * 2);
            printf("%s operands, %d wave(s)/SIMD: %.1f TFLOP/s, in-kernel clock %.2f GHz, %.3f ms/launch\n",
                   mode ? "zero  " : "random", wps, flops / (ms * 1e-3) / 1e12, ghz, ms / reps);
        }
    }
    return 0;
}
