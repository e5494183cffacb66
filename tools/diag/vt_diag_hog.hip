// vt_diag_hog.hip -- DIAGNOSTICS ONLY, not part of libvt_amd.so or of include/vt_amd.h (VERDICT r05 #7d).
// Built on demand by tools/rccl_hog.py into tools/diag/libvt_diag_hog.so:
//     hipcc -O2 -fPIC -shared --offload-arch=gfx950 tools/diag/vt_diag_hog.hip -o tools/diag/libvt_diag_hog.so
//
// vt_diag_hog: `wgs` workgroups of `threads` threads that hold `lds_bytes` of LDS each and do nothing until `microseconds`
// of wall clock have passed -- the CU footprint of a collective library's channel kernels (RCCL: one workgroup per channel,
// 256-512 threads, <= 64 KiB of LDS) beside the train step's own kernels.  Computes nothing; every wave leaves by the
// clock alone (no flag, no dependence on another workgroup), so the grid always drains.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__global__ void hog_kernel(unsigned long long ticks, int lds_bytes) {
    extern __shared__ char hog_smem[];
    if (threadIdx.x * 4 < (unsigned)lds_bytes) ((volatile unsigned*)hog_smem)[threadIdx.x] = threadIdx.x;  // (the allocation is real)
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

// returns 0, or a hipError_t value
extern "C" int vt_diag_hog(int32_t wgs, int32_t threads, int32_t lds_bytes, double microseconds, void* stream) {
    if (wgs < 1 || wgs > 256 || threads < 64 || threads > 1024 || threads % 64 || lds_bytes < 0 || lds_bytes > 160 * 1024 ||
        microseconds < 0 || microseconds > 1e6)
        return (int)hipErrorInvalidValue;
    if (lds_bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(threads), lds_bytes, (hipStream_t)stream,
                       (unsigned long long)(microseconds * 100.0), lds_bytes);
    return (int)hipGetLastError();
}
