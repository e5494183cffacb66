// Microbenchmark (GPU box): LDS cycles per wave-instruction of ds_read_b64_tr_b16 against ds_read_b64 / ds_read_b128,
// on the conflict-free image of vt_wgrad_span.hip (128-byte rows, chunk ^= 2*((row>>1)&3)).  One workgroup per CU,
// W waves (1, 4, 8, 16), every wave issues 32 reads back to back per iteration, results folded so nothing is dead.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/tr_read.hip -o build/tr_read && build/tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int KIND>  // 0: ds_read_b64_tr_b16, 1: ds_read_b64, 2: ds_read_b128
__global__ void k(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int g = lane >> 4, u = lane & 15, q = u >> 2, pp = u & 3;
    const int rowlo = 4 * g + q;
    unsigned off;
    if (KIND == 2)
        off = (unsigned)((lane & 15) * 128 + (((lane >> 4) ^ (2 * (((lane & 15) >> 1) & 3))) << 4));  // 16 rows x 16 B, swizzled
    else
        off = (unsigned)(rowlo * 128 + (((pp >> 1) ^ (2 * ((rowlo >> 1) & 3))) << 4) + 8 * (pp & 1));
    off += (wave & 3) * 4096;
    unsigned long long acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const char* p = smem + ((off + r * 2048 + (it & 7) * 4096) & 0xffff);  // (varies with `it`: not loop invariant)
            if (KIND == 0) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
                acc += (unsigned short)v[0] + ((unsigned long long)(unsigned short)v[3] << 20);
            } else if (KIND == 1) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                u32x2 v;
                v = *(const u32x2*)p;
                acc += v[0] + ((unsigned long long)v[1] << 20);
            } else {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                u32x4 v;
                v = *(const u32x4*)p;
                acc += v[0] + ((unsigned long long)v[3] << 20);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 32 + wave] = t1 - t0;
    if (acc == 0x1234567) out[0] = acc;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 256 * 32 * 8);
    unsigned long long h[256 * 32];
    const int iters = 2000;
    const char* names[3] = {"ds_read_b64_tr_b16", "ds_read_b64", "ds_read_b128"};
    for (int kind = 0; kind < 3; ++kind)
        for (int waves : {1, 4, 8, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 65536, 0, d, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 65536, 0, d, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 65536, 0, d, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double mx = 0;
            for (int b = 0; b < 256; ++b)
                for (int w = 0; w < waves; ++w) mx = h[b * 32 + w] > mx ? h[b * 32 + w] : mx;
            // s_memtime ticks at the shader clock: cycles of the slowest wave / reads issued on the CU
            printf("%-20s %2d waves/CU: %.2f cycles per wave-instruction per CU (%.1f per wave)\n", names[kind], waves,
                   mx / (double)(iters * 32) / waves, mx / (double)(iters * 32));
        }
    return 0;
}
