// LDS round-trip latency for the decoder's access pattern: per iteration a dependent address -> K random reads of
// WIDTH bytes (+ optionally one conflict-free b32 read) -> 5 dependent VALU ops -> next address.  4 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 512
#define REPS 6

template <int WIDTH, int K, bool RING, int TABLE_WORDS>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed, uint32_t* sink) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < TABLE_WORDS + 2048 + 64; i += 256) lds[i] = (i * 2654435761u) ^ (i >> 3);
    __syncthreads();
    const uint32_t mask = (TABLE_WORDS * 4 / WIDTH) - 1;
    uint32_t a = seed * (threadIdx.x + 1), b = seed + 77 * threadIdx.x, acc = 0;
    uint64_t t0 = 0;
    for (int rep = 0; rep < REPS; ++rep) {
        if (rep == 1) t0 = __builtin_readcyclecounter();
#pragma unroll 4
        for (int i = 0; i < N_ITER; ++i) {
            uint32_t r = 0;
            if (RING) r = lds[TABLE_WORDS + ((i & 31) * 64 + (threadIdx.x & 63))];
            uint32_t v0 = 0, v1 = 0;
            if constexpr (WIDTH == 4) {
                v0 = lds[a & mask];
                if (K > 1) v1 = lds[b & mask];
            } else {
                const uint2 e0 = reinterpret_cast<const uint2*>(lds)[a & mask];
                v0 = e0.x ^ e0.y;
                if (K > 1) { const uint2 e1 = reinterpret_cast<const uint2*>(lds)[b & mask]; v1 = e1.x + e1.y; }
            }
            // 5 dependent ops
            uint32_t x = v0 + v1 + r;
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(seed));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(acc));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(seed), "v"(b));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(a));
            b = a * 3 + 1; a = x; acc += x;
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    sink[threadIdx.x + blockIdx.x * 256] = a + acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int WIDTH, int K, bool RING, int TABLE_WORDS> void run(const char* name, int blocks) {
    uint64_t* d; uint32_t* s; (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 4 * 256 * 1024);
    size_t lds = (TABLE_WORDS + 2048 + 64) * 4;
    (void)hipFuncSetAttribute((const void*)k<WIDTH, K, RING, TABLE_WORDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<WIDTH, K, RING, TABLE_WORDS>), dim3(blocks), dim3(256), lds, 0, d, 12345u, s);
    (void)hipDeviceSynchronize();
    uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s blocks=%4d : %7.1f cycles/iter\n", name, blocks, (double)h / ((REPS - 1) * (double)N_ITER));
    (void)hipFree(d); (void)hipFree(s);
}

int main() {
    run<4, 1, false, 4096>("1 x b32 random (16 KiB table)", 1);
    run<4, 1, false, 4096>("1 x b32 random (16 KiB table)", 256);
    run<8, 1, false, 8192>("1 x b64 random (32 KiB table)", 1);
    run<8, 1, false, 8192>("1 x b64 random (32 KiB table)", 256);
    run<8, 2, false, 8192>("2 x b64 random", 256);
    run<8, 2, true, 8192>("2 x b64 random + ring b32", 256);
    run<4, 2, true, 4096>("2 x b32 random + ring b32", 256);
    run<4, 2, false, 4096>("2 x b32 random", 256);
    return 0;
}
