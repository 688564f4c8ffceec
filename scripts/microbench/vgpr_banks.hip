// VGPR bank conflicts on gfx950 (lone wave per SIMD): ticks per instruction for source operands in the same / different
// banks (register index mod 4), operands not forwarded from the previous instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(A) A A A A
#define R16(A) R4(R4(A))
template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t iters) {
    uint64_t t0 = __builtin_readcyclecounter();
    asm volatile("v_mov_b32 v116, 7\n\tv_mov_b32 v117, 3\n\tv_mov_b32 v118, 5\n\tv_mov_b32 v119, 9\n\ts_mov_b32 s24, 11\n\ts_mov_b32 s23, %[n]\n\t1:\n\t" :: [n] "s"(iters) : "v116", "v117", "v118", "v119", "s23", "s24");
#define X2(d, a, b) "v_xor_b32 v" #d ", v" #a ", v" #b "\n\t"
#define XS(d, a) "v_xor_b32 v" #d ", s24, v" #a "\n\t"
#define M3(d, a, b, c) "v_mad_u32_u24 v" #d ", v" #a ", v" #b ", v" #c "\n\t"
#define AO(d, a, b, c) "v_and_or_b32 v" #d ", v" #a ", v" #b ", v" #c "\n\t"
    // destination registers rotate over 8 so that no source was written by the previous instruction
    if constexpr (KIND == 0) asm volatile(R16(X2(100,100,116) X2(104,104,116) X2(108,108,116) X2(112,112,116) X2(100,100,116) X2(104,104,116) X2(108,108,116) X2(112,112,116)) ::: "memory"); // same bank, 4 chains
    if constexpr (KIND == 1) asm volatile(R16(X2(101,101,116) X2(105,105,116) X2(109,109,116) X2(113,113,116) X2(101,101,116) X2(105,105,116) X2(109,109,116) X2(113,113,116)) ::: "memory"); // different banks
    if constexpr (KIND == 2) asm volatile(R16(XS(100,100) XS(104,104) XS(108,108) XS(112,112) XS(100,100) XS(104,104) XS(108,108) XS(112,112)) ::: "memory");                                 // sgpr + vgpr
    if constexpr (KIND == 3) asm volatile(R16(M3(100,100,116,104) M3(104,104,116,108) M3(108,108,116,112) M3(112,112,116,100) M3(100,100,116,104) M3(104,104,116,108) M3(108,108,116,112) M3(112,112,116,100)) ::: "memory"); // 3 sources, one bank
    if constexpr (KIND == 4) asm volatile(R16(M3(100,100,117,106) M3(104,104,117,110) M3(108,108,117,114) M3(112,112,117,102) M3(100,100,117,106) M3(104,104,117,110) M3(108,108,117,114) M3(112,112,117,102)) ::: "memory"); // 3 sources, 3 banks
    if constexpr (KIND == 5) asm volatile(R16(M3(100,100,116,105) M3(104,104,116,109) M3(108,108,116,113) M3(112,112,116,101) M3(100,100,116,105) M3(104,104,116,109) M3(108,108,116,113) M3(112,112,116,101)) ::: "memory"); // src0/src1 same bank
    if constexpr (KIND == 6) asm volatile(R16(M3(100,100,117,104) M3(104,104,117,108) M3(108,108,117,112) M3(112,112,117,100) M3(100,100,117,104) M3(104,104,117,108) M3(108,108,117,112) M3(112,112,117,100)) ::: "memory"); // src0/src2 same bank
    if constexpr (KIND == 7) asm volatile(R16(M3(100,101,116,104) M3(104,105,116,108) M3(108,109,116,112) M3(112,113,116,100) M3(100,101,116,104) M3(104,105,116,108) M3(108,109,116,112) M3(112,113,116,100)) ::: "memory"); // src1/src2 same bank
    asm volatile("s_sub_u32 s23, s23, 1\n\ts_cmp_lg_u32 s23, 0\n\ts_cbranch_scc1 1b" ::: "s23", "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115");
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int KIND> void run(const char* name) {
    uint64_t* d; (void)hipMalloc(&d, 8);
    const uint32_t iters = 200; uint64_t h;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s %5.2f ticks per instruction\n", name, (double)h / (iters * 128.0));
    (void)hipFree(d);
}
int main() {
    run<0>("v_xor: both sources in bank 0"); run<1>("v_xor: sources in banks 1 and 0"); run<2>("v_xor: one SGPR source");
    run<3>("v_mad_u32_u24: all three sources in bank 0"); run<4>("v_mad_u32_u24: three different banks");
    run<5>("v_mad_u32_u24: src0, src1 same bank"); run<6>("v_mad_u32_u24: src0, src2 same bank"); run<7>("v_mad_u32_u24: src1, src2 same bank");
    return 0;
}
