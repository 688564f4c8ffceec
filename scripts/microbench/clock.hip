// Effective shader clock under a full-chip VALU load: s_memtime ticks vs wall time (hipEvents), and ticks per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t iters, uint32_t* sink) {
    uint32_t a = threadIdx.x, b = blockIdx.x | 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 256 + threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    uint64_t* d; uint32_t* s; (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 256 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {1, 256}) {
        const uint32_t iters = 20000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100u, s);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, s);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        double instr = (double)iters * 66;   // 64 mads + loop overhead ~2
        printf("blocks=%3d: %.3f ms, %llu ticks -> %.1f MHz tick rate, %.2f ticks/instr, %.2f ns/instr\n", blocks, ms,
               (unsigned long long)h, h / (ms * 1e3), h / instr, ms * 1e6 / instr);
    }
    return 0;
}
