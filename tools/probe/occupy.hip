// A stand-in for a collective's kernel on a one-GPU box: `wgs` workgroups of 256 threads that hold their compute units for
// `ticks` ticks of the 100 MHz wall clock and then exit (every wave leaves the loop after at most `ticks`).
#include <hip/hip_runtime.h>
__global__ void occupy_kernel(long long ticks, int* sink) {
    extern __shared__ int occupy_lds[];      // like a collective kernel it needs LDS: it cannot share a CU with a GEMM workgroup
    occupy_lds[threadIdx.x] = 0;
    const long long t0 = wall_clock64();
    int spins = 0;
    while (wall_clock64() - t0 < ticks && spins < (1 << 26)) { __builtin_amdgcn_s_sleep(32); ++spins; }
    if (sink && spins < 0) *sink = spins;
}
extern "C" int occupy(int wgs, long long ticks, void* stream) {
    occupy_kernel<<<wgs, 256, 32768, (hipStream_t)stream>>>(ticks, nullptr);
    return (int)hipGetLastError();
}
