// VALU throughput per SIMD by instruction class and number of resident waves (gfx950): what bounds the ANS encoder step.
// Every wave runs ILP = 8 independent chains of one instruction; 256 blocks (one per CU) of 256 * W threads.
// Output: cycles per wave-instruction per SIMD at 2.4 GHz = (kernel time * clock) / (instructions per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 64
template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t* sink, uint32_t seed, int iters) {
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed * (i + 3) + threadIdx.x;
    uint32_t c = seed | 1, d = seed ^ 0x5bd1e995;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
                else if constexpr (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
                else if constexpr (KIND == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
                else if constexpr (KIND == 3) asm volatile("v_mad_u64_u32 v[%0:%1], s[20:21], %2, %3, v[%0:%1]" :: "n"(100), "n"(101), "v"(c), "v"(d) : "v100", "v101", "s20", "s21");
                else if constexpr (KIND == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
                else if constexpr (KIND == 5) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[j]) : "v"(c));
                else if constexpr (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(c) : );
                else if constexpr (KIND == 7) asm volatile("v_add_co_u32 %0, s[20:21], %0, %1" : "+v"(v[j]) : "v"(c) : "s20", "s21");
                else if constexpr (KIND == 8) asm volatile("v_alignbit_b32 %0, %0, %1, 12" : "+v"(v[j]) : "v"(c));
                else if constexpr (KIND == 9) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
                else if constexpr (KIND == 10) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(v[j]) : "v"(c));
                else if constexpr (KIND == 11) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[j]) : "v"(c));
                else if constexpr (KIND == 12) asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(v[j]) : "v"(c));
                else if constexpr (KIND == 13) asm volatile("v_cmp_ge_u32 vcc, %0, %1" :: "v"(v[j]), "v"(c) : "vcc");
                else if constexpr (KIND == 14) asm volatile("v_lshlrev_b64 v[%0:%1], 12, v[%0:%1]" :: "n"(100), "n"(101) : "v100", "v101");
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    sink[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int KIND> void run(const char* name) {
    uint32_t* s; (void)hipMalloc(&s, 256 * 1024 * 4);
    const int iters = 500;
    printf("%-22s", name);
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, s, 12345u, iters);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, s, 12345u, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_wave = ms * 2.4e6 / (iters * (double)ITER * 8);
        printf("  %d wave(s)/SIMD: %5.2f cyc/instr/wave = %5.2f per SIMD", threads / 256, per_wave, per_wave / (threads / 256));
    }
    printf("\n");
    (void)hipFree(s);
}

int main() {
    run<0>("v_xor_b32"); run<1>("v_mul_lo_u32"); run<2>("v_mul_hi_u32"); run<3>("v_mad_u64_u32"); run<4>("v_mad_u32_u24");
    run<5>("v_mul_u32_u24"); run<6>("v_cndmask_b32 (vcc)"); run<7>("v_add_co_u32"); run<8>("v_alignbit_b32"); run<9>("v_and_or_b32");
    run<10>("v_add_u32_sdwa"); run<11>("v_lshl_add_u32"); run<12>("v_mul_u32_u24_sdwa"); run<13>("v_cmp_ge_u32"); run<14>("v_lshlrev_b64");
    return 0;
}
