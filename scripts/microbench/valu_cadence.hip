// Issue cadence of a lone wave's VALU stream on gfx950, and what interleaved scalar instructions do to it.
// 256 workgroups x 256 threads (one wave per SIMD).  Ticks per VALU instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define X(r) "v_xor_b32 v" #r ", v" #r ", v116\n\t"
#define M(r) "v_mad_u32_u24 v" #r ", v" #r ", v116, v117\n\t"
#define N "s_nop 0\n\t"
#define SO "s_or_b32 s24, s24, s25\n\t"
#define R4(A) A A A A
#define R16(A) R4(R4(A))

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t iters) {
    uint64_t t0 = __builtin_readcyclecounter();
    asm volatile("v_mov_b32 v116, 7\n\tv_mov_b32 v117, 3\n\ts_mov_b32 s23, %[n]\n\t1:\n\t" :: [n] "s"(iters) : "v116", "v117", "s23");
    if constexpr (KIND == 0) asm volatile(R16(X(100) X(101) X(102) X(103) X(104) X(105) X(106) X(107)) ::: "memory");
    if constexpr (KIND == 1) asm volatile(R16(X(100) N X(101) N X(102) N X(103) N X(104) N X(105) N X(106) N X(107) N) ::: "memory");
    if constexpr (KIND == 2) asm volatile(R16(X(100) X(101) N X(102) X(103) N X(104) X(105) N X(106) X(107) N) ::: "memory");
    if constexpr (KIND == 3) asm volatile(R16(X(100) X(101) X(102) X(103) N X(104) X(105) X(106) X(107) N) ::: "memory");
    if constexpr (KIND == 4) asm volatile(R16(X(100) SO X(101) SO X(102) SO X(103) SO X(104) SO X(105) SO X(106) SO X(107) SO) ::: "memory", "s24");
    if constexpr (KIND == 5) asm volatile(R16(M(100) M(101) M(102) M(103) M(104) M(105) M(106) M(107)) ::: "memory");
    if constexpr (KIND == 6) asm volatile(R16(M(100) N M(101) N M(102) N M(103) N M(104) N M(105) N M(106) N M(107) N) ::: "memory");
    if constexpr (KIND == 7) asm volatile(R16(M(100) SO M(101) SO M(102) SO M(103) SO M(104) SO M(105) SO M(106) SO M(107) SO) ::: "memory", "s24");
    if constexpr (KIND == 8) asm volatile(R16(X(100) X(100) X(100) X(100) X(100) X(100) X(100) X(100)) ::: "memory");             // dependent
    if constexpr (KIND == 9) asm volatile(R16(X(100) N X(100) N X(100) N X(100) N X(100) N X(100) N X(100) N X(100) N) ::: "memory"); // dependent + nop
    if constexpr (KIND == 10) asm volatile(R16(X(100) X(101) X(100) X(101) X(100) X(101) X(100) X(101)) ::: "memory");            // ILP 2
    asm volatile("s_sub_u32 s23, s23, 1\n\ts_cmp_lg_u32 s23, 0\n\ts_cbranch_scc1 1b" ::: "s23", "memory", "v100","v101","v102","v103","v104","v105","v106","v107");
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND> void run(const char* name) {
    uint64_t* d; (void)hipMalloc(&d, 8);
    const uint32_t iters = 200; uint64_t h;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-52s %5.2f ticks per VALU instruction\n", name, (double)h / (iters * 128.0));
    (void)hipFree(d);
}

int main() {
    run<0>("v_xor, 8 independent chains"); run<1>("  + s_nop after every one"); run<2>("  + s_nop after every second"); run<3>("  + s_nop after every fourth");
    run<4>("  + s_or_b32 after every one");
    run<5>("v_mad_u32_u24, 8 independent chains"); run<6>("  + s_nop after every one"); run<7>("  + s_or_b32 after every one");
    run<8>("v_xor, ONE dependent chain"); run<9>("  + s_nop after every one"); run<10>("v_xor, two chains");
    return 0;
}
