// Round 4: what does a per-lane word ring cost the LDS?  One wave per SIMD (the C2 shape), per iteration: ONE ds_write_b32 (or
// ds_read_b32) into the ring layout [slot][lane] (row = 256 bytes, lane l always in bank l % 32: conflict-free by the book)
// with the ROW chosen per lane by a pattern, NV filler VALU instructions, and one conflict-free ds_read_b32 whose result the
// next iteration depends on (it completes in order BEHIND the ring access, as the encoder's table reads do).
// Patterns: how many distinct rows the 64 lanes of one instruction touch, and how they are grouped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 2048

template <int PATTERN, int NV, bool WRITE>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16384))) uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 4096 + 256; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(lds + wave * 4096) + 4 * lane;
    const uint32_t tab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(lds + 4 * 4096) + 4 * lane;
    uint32_t group;
    switch (PATTERN) {
        case 2: group = lane >> 1; break;      // pairs of lanes share a row
        case 3: group = lane >> 2; break;
        case 4: group = lane >> 4; break;
        case 5: group = lane >> 5; break;
        default: group = lane;
    }
    uint32_t x = seed * (2 * group + 1) + 12345u, acc = 0, dep = 0;
    const uint32_t t0 = (uint32_t)__builtin_readcyclecounter();
    for (int i = 0; i < N_ITER; ++i) {
        x = x * 1664525u + 1013904223u + dep;
        uint32_t row;
        if (PATTERN == 0) row = i & 63;                                   // the whole wave in one row
        else if (PATTERN == 6) row = (x >> 24) & 7;                       // random, 8 rows
        else if (PATTERN == 7) row = ((i >> 2) + ((x >> 24) & 3)) & 63;   // 4 neighbouring rows (what a per-tile restart gives)
        else if (PATTERN == 8) row = ((i >> 2) + ((x >> 24) & 1)) & 63;   // 2 neighbouring rows
        else if (PATTERN == 9) row = ((i >> 2) + ((x >> 24) % 12)) & 63;  // 12 neighbouring rows
        else row = (x >> 24) & 63;                                        // random over 64 rows (per lane / per group)
        const uint32_t addr = ring + (row << 8);
        uint32_t v = acc;
        if (WRITE) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
        else asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        uint32_t r;
        asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(tab) : "memory");
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(acc) : "v"(seed), "v"(x));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dep = (r & 1) + (WRITE ? 0 : (v & 1));
    }
    const uint32_t t1 = (uint32_t)__builtin_readcyclecounter();
    sink[threadIdx.x + blockIdx.x * 256] = acc + x;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int PATTERN, int NV, bool WRITE> void run(const char* name) {
    uint64_t* d; uint32_t* s; (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 4 * 256 * 1024);
    const size_t lds = (4 * 4096 + 256) * 4;
    (void)hipFuncSetAttribute((const void*)k<PATTERN, NV, WRITE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<PATTERN, NV, WRITE>), dim3(256), dim3(256), lds, 0, d, 12345u, s);
    (void)hipDeviceSynchronize();
    uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-6s NV=%2d %-52s: %7.1f cycles/iter\n", WRITE ? "write" : "read", NV, name, (double)h / (double)N_ITER);
    (void)hipFree(d); (void)hipFree(s);
}

template <int NV, bool WRITE> void all() {
    run<0, NV, WRITE>("one row per instruction");
    run<8, NV, WRITE>("2 neighbouring rows");
    run<7, NV, WRITE>("4 neighbouring rows");
    run<9, NV, WRITE>("12 neighbouring rows");
    run<6, NV, WRITE>("random over 8 rows");
    run<1, NV, WRITE>("random over 64 rows, every lane its own");
    run<2, NV, WRITE>("random over 64 rows, lane pairs together");
    run<3, NV, WRITE>("random over 64 rows, lane quads together");
    run<4, NV, WRITE>("random over 64 rows, 16 lanes together");
    run<5, NV, WRITE>("random over 64 rows, 32 lanes together");
}

int main() {
    all<0, true>(); all<8, true>(); all<20, true>();
    all<0, false>(); all<20, false>();
    return 0;
}
