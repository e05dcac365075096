// A stand-in for a collective's kernel: `wgs` workgroups that each HOLD a CU's LDS (64 KB) and 256 threads for `cycles` clock
// ticks.  Used by tools/occupy_probe.py to see what a persistent one-workgroup-per-CU GEMM does when some CUs are taken.
// build: hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/micro/occupy.hip -o tools/micro/libocc.so
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(256) occupy_kernel(long long cycles, int* sink) {
    __shared__ int hold[16384];                       // 64 KB: no 152 KB GEMM workgroup fits beside it
    hold[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (hold[(threadIdx.x * 7) & 16383] == -1) sink[0] = 1;
}
extern "C" int occ_launch(int wgs, long long cycles, int* sink, void* stream) {
    occupy_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(cycles, sink);
    return (int)hipGetLastError();
}
