// barrier_cost.hip -- what one workgroup barrier costs on gfx950, by workgroup size and by what the waves do
// between barriers (nothing / a few SALU ops / an LDS read).   hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ void k(int iters, unsigned long long* out, int* sink) {
    __shared__ int sh[1024];
    sh[threadIdx.x] = threadIdx.x;
    __syncthreads();
    int acc = 0;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE == 1) acc += sh[(threadIdx.x + i) & 1023];
        if (MODE == 2) {
            asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
        }
    }
    const unsigned long long c1 = clock64();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = t1 - t0;
        out[blockIdx.x * 2 + 1] = c1 - c0;
    }
    if (acc == 0x7fffffff) *sink = acc;
}

int main() {
    unsigned long long* d;
    int* sink;
    hipMalloc(&d, 1024 * 16);
    hipMalloc(&sink, 4);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int threads : {64, 256, 320, 512, 768, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, iters, d, sink);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, iters, d, sink);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, iters, d, sink);
                hipDeviceSynchronize();
            }
            unsigned long long h[512];
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double ns = 0, cyc = 0;
            for (int b = 0; b < 256; ++b) ns += h[2 * b] * 10.0 / iters / 256, cyc += (double)h[2 * b + 1] / iters / 256;
            printf("mode %d (%s) threads %4d: %.1f ns = %.0f shader cycles per barrier iteration\n", mode,
                   mode == 0 ? "bare" : (mode == 1 ? "lds read" : "8 s_nop"), threads, ns, cyc);
        }
    return 0;
}
