// Single-wave issue behaviour on gfx950: dependent vs independent VALU chains, half-masked waves,
// one vs two waves per SIMD.  All timing with s_memtime on wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITER 512
#define REPS 8

// ILP independent xor/add chains per lane
template <int ILP, int KIND>
__global__ __launch_bounds__(1024) void k(uint64_t* out, uint32_t seed, uint32_t* sink, int active_lanes) {
    const int lane = threadIdx.x & 63;
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 3) + threadIdx.x;
    uint32_t c = seed | 1, d = seed ^ 0x5bd1e995;
    uint64_t t0 = 0;
    if (lane < active_lanes) {
        for (int rep = 0; rep < REPS; ++rep) {
            if (rep == 1) t0 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
                for (int j = 0; j < ILP; ++j) {
                    if constexpr (KIND == 0) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[j]) : "v"(c)); }
                    else if constexpr (KIND == 1) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[j]) : "v"(c)); }
                    else if constexpr (KIND == 2) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d)); }
                    else if constexpr (KIND == 3) { asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[j]) : "v"(c)); }
                }
            }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <int ILP, int KIND> void run(const char* name, int threads, int active) {
    uint64_t* d; uint32_t* s; (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 4096);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<ILP, KIND>), dim3(1), dim3(threads), 0, 0, d, 12345u, s, active);
    (void)hipDeviceSynchronize();
    uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-14s ILP=%d threads=%4d active_lanes=%2d : %6.2f cycles per instruction (wave 0)\n", name, ILP, threads, active,
           (double)h / ((REPS - 1) * (double)N_ITER * ILP));
    (void)hipFree(d); (void)hipFree(s);
}

int main() {
    run<1, 0>("v_xor", 256, 64); run<2, 0>("v_xor", 256, 64); run<4, 0>("v_xor", 256, 64); run<8, 0>("v_xor", 256, 64);
    run<1, 0>("v_xor", 256, 32); run<2, 0>("v_xor", 256, 32); run<4, 0>("v_xor", 256, 32);
    run<1, 0>("v_xor", 512, 64); run<2, 0>("v_xor", 512, 64); run<1, 0>("v_xor", 1024, 64); run<4, 0>("v_xor", 1024, 64);
    run<1, 0>("v_xor", 512, 32); run<1, 0>("v_xor", 1024, 32);
    run<1, 1>("v_mul_hi_u32", 256, 64); run<2, 1>("v_mul_hi_u32", 256, 64); run<4, 1>("v_mul_hi_u32", 256, 64); run<1, 1>("v_mul_hi_u32", 256, 32);
    run<1, 2>("v_mad_u32_u24", 256, 64); run<2, 2>("v_mad_u32_u24", 256, 64);
    run<1, 3>("v_lshl_add", 256, 64); run<4, 3>("v_lshl_add", 256, 64);
    return 0;
}
