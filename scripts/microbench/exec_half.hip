// Does gfx950 skip the 16-lane passes of a VALU instruction whose lanes are all masked off?  A wave64 instruction takes 4
// passes of 16 lanes; if passes with EXEC == 0 were skipped, a wave with only its low 32 (or 16) lanes active would issue
// at twice (four times) the rate and two half-populated waves per SIMD could stand in for one full one.
// One wave per SIMD; ticks per VALU instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define M(r) "v_mad_u32_u24 v" #r ", v" #r ", v116, v117\n\t"
#define X(r) "v_xor_b32 v" #r ", v" #r ", v116\n\t"
#define R4(A) A A A A
#define R16(A) R4(R4(A))

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t iters, unsigned long long mask) {
    uint64_t t0 = __builtin_readcyclecounter();
    asm volatile("s_mov_b64 s[26:27], exec\n\ts_mov_b64 exec, %[m]\n\tv_mov_b32 v116, 7\n\tv_mov_b32 v117, 3\n\ts_mov_b32 s23, %[n]\n\t1:\n\t"
                 :: [n] "s"(iters), [m] "s"(mask) : "v116", "v117", "s23", "s26", "s27");
    if constexpr (KIND == 0) asm volatile(R16(X(100) X(101) X(102) X(103) X(104) X(105) X(106) X(107)) ::: "memory");
    if constexpr (KIND == 1) asm volatile(R16(M(100) M(101) M(102) M(103) M(104) M(105) M(106) M(107)) ::: "memory");
    asm volatile("s_sub_u32 s23, s23, 1\n\ts_cmp_lg_u32 s23, 0\n\ts_cbranch_scc1 1b\n\ts_mov_b64 exec, s[26:27]"
                 ::: "s23", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND> void run(const char* name, unsigned long long mask, int threads) {
    uint64_t* d; (void)hipMalloc(&d, 8);
    const uint32_t iters = 200; uint64_t h;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, iters, mask);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-14s exec %016llx, %d waves per SIMD: %5.2f ticks per VALU instruction and wave\n", name, mask, threads / 256, (double)h / (iters * 128.0));
    (void)hipFree(d);
}

int main() {
    for (int threads : {256}) {
        run<0>("v_xor", ~0ull, threads); run<0>("v_xor", 0xffffffffull, threads); run<0>("v_xor", 0xffffull, threads);
        run<1>("v_mad_u32_u24", ~0ull, threads); run<1>("v_mad_u32_u24", 0xffffffffull, threads); run<1>("v_mad_u32_u24", 0xffffull, threads);
    }
    return 0;
}
