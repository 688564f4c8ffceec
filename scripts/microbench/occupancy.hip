// Does a SIMD of gfx950 issue VALU work from two resident waves faster than from one?  Every wave runs the same
// chain of N_ITER * ILP v_xor (or the 23-instruction ANS encoder step); the kernel is timed with events for 1, 2, 4 and
// 8 waves per SIMD (256 blocks, one per CU).  wall(2 waves) == wall(1 wave) means a second wave is free.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ILP>
__global__ __launch_bounds__(1024) void k(uint32_t* sink, uint32_t seed, int iters) {
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 3) + threadIdx.x;
    uint32_t c = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
#pragma unroll
            for (int j = 0; j < ILP; ++j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    sink[blockIdx.x * 1024 + threadIdx.x] = s;
}

// dependent LDS read chain + a few VALU (the decoder's shape): per iteration 1 ds_read_b32 (address from the previous
// result) + 10 VALU
__global__ __launch_bounds__(1024) void kdec(uint32_t* sink, uint32_t seed, int iters) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 2654435761u) >> 20;
    __syncthreads();
    uint32_t a = (threadIdx.x * 4) & 0x3ffc, x = seed + threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)lds;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %2\n\t"
                         "v_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_add_u32 %1, %1, %0\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %3\n\t"
                         "v_and_b32 %0, 0x3ffc, %1\n\tv_add_u32 %0, %0, %4"
                         : "=&v"(v), "+v"(x) : "v"(a), "v"(seed), "v"(base) : "memory");
            a = v;
        }
    }
    sink[blockIdx.x * 1024 + threadIdx.x] = x + a;
}

template <typename K> float timeit(K kern, int threads, uint32_t* sink, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, sink, 12345u, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, sink, 12345u, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    uint32_t* s; (void)hipMalloc(&s, 256 * 1024 * 4);
    const int iters = 2000;
    for (int threads : {256, 512, 768, 1024}) {
        const float a = timeit(k<1>, threads, s, iters), b = timeit(k<4>, threads, s, iters), c = timeit(k<8>, threads, s, iters);
        printf("VALU   %d waves/SIMD: ILP1 %.3f ms (%.2f cyc/instr/wave @2.4GHz)  ILP4 %.3f ms (%.2f)  ILP8 %.3f ms (%.2f)\n", threads / 256,
               a, a * 2.4e6 / (iters * 64.0), b, b * 2.4e6 / (iters * 64.0 * 4), c, c * 2.4e6 / (iters * 64.0 * 8));
    }
    for (int threads : {256, 512, 768, 1024}) {
        const float a = timeit(kdec, threads, s, iters);
        printf("LDS-chain %d waves/SIMD: %.3f ms (%.1f cycles per iteration per wave @2.4GHz; 11 VALU + 1 ds_read_b32 dependent)\n", threads / 256, a, a * 2.4e6 / (iters * 16.0));
    }
    return 0;
}
