// LDS throughput on gfx950 for the access shapes of the per-stream-table coder: per-lane random addresses,
// 4 waves per CU, 16 independent reads per wait.  Reports cycles per wave-instruction (per CU: 4 waves share the LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* sink, int iters, uint32_t seed) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)lds;
    uint32_t h = seed + threadIdx.x * 0x9E3779B9u + blockIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            h = h * 1664525u + 1013904223u;
            uint32_t off = (h >> 8) & 0xfff0;                   // 16-byte aligned offset in 64 KiB
            if (KIND == 1) off |= 2;                            // b32 at 2 (mod 4)
            if (KIND == 2) off |= 2;                            // u16
            if (KIND == 3) off |= 8;                            // b64 8-aligned
            if (KIND == 5) off |= 4;                            // b128 at 4 (mod 16)
            if (KIND == 6) off |= 0;                            // write b32
            const uint32_t a = base + off;
            if (KIND == 0 || KIND == 1) { uint32_t v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); acc ^= v; }
            else if (KIND == 2) { uint32_t v; asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(a)); acc ^= v; }
            else if (KIND == 3) { uint64_t v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); acc ^= (uint32_t)v; }
            else if (KIND == 4 || KIND == 5) { __uint128_t v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); acc ^= (uint32_t)v; }
            else if (KIND == 6) { asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(h)); }
            else if (KIND == 8) { __uint128_t v; asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v) : "v"(a | 8)); acc ^= (uint32_t)v; }
            else if (KIND == 9) { __uint128_t v; uint64_t w; asm volatile("ds_read2_b64 %0, %2 offset1:1\n\tds_read_b64 %1, %2 offset:16" : "=&v"(v), "=&v"(w) : "v"(a | 8)); acc ^= (uint32_t)v ^ (uint32_t)w; }
            else if (KIND == 10) { __uint128_t v, w; asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(v), "=&v"(w) : "v"(a)); acc ^= (uint32_t)v ^ (uint32_t)w; }
            else if (KIND == 7) { uint32_t v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(a | (h & 3))); acc ^= v; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND> void run(const char* name) {
    uint32_t* s; (void)hipMalloc(&s, 256 * 256 * 4);
    const int iters = 2000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 65536, 0, s, iters, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 65536, 0, s, iters, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per CU: 4 waves * iters * 16 instructions
    printf("%-34s %.3f ms  -> %.1f cycles per wave-instruction per CU (2.4 GHz), %.1f per wave\n", name, ms, ms * 2.4e6 / (4.0 * iters * 16), ms * 2.4e6 / (iters * 16.0));
    (void)hipFree(s);
}

int main() {
    run<0>("ds_read_b32 aligned, random");
    run<1>("ds_read_b32 2-byte aligned, random");
    run<2>("ds_read_u16, random");
    run<7>("ds_read_u8, random");
    run<3>("ds_read_b64 8-byte aligned, random");
    run<4>("ds_read_b128 16-byte aligned, random");
    run<5>("ds_read_b128 4-byte aligned, random");
    run<6>("ds_write_b32, random");
    run<8>("ds_read2_b64 adjacent, 8-byte aligned (16 B at 8 mod 16)");
    run<9>("ds_read2_b64 + ds_read_b64: 24 B at 8 mod 16 (two instructions)");
    run<10>("2 x ds_read_b128: 32 B 16-byte aligned (two instructions)");
    return 0;
}
