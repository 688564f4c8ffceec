// Issue cost (lone wave per SIMD, gfx950) of the float instructions a range decoder's quotient estimate would use.
// Each line: 8 independent chains of one instruction, ticks per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(A) A A A A
#define R16(A) R4(R4(A))
#define D(i) "v[" #i ":" #i "+1]"

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t iters) {
    uint64_t t0 = __builtin_readcyclecounter();
    asm volatile("v_mov_b32 v116, 0x40400000\n\tv_mov_b32 v117, 3\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0x40080000\n\ts_mov_b32 s23, %[n]\n\t1:\n\t" :: [n] "s"(iters) : "v116", "v117", "v118", "v119", "s23");
#define EIGHT(OP, a, b) R16(OP(100) OP(102) OP(104) OP(106) OP(108) OP(110) OP(112) OP(114))
#define CVT_F32_U32(r) "v_cvt_f32_u32 v" #r ", v117\n\t"
#define CVT_U32_F32(r) "v_cvt_u32_f32 v" #r ", v116\n\t"
#define RCP_F32(r) "v_rcp_f32 v" #r ", v116\n\t"
#define FMA_F32(r) "v_fma_f32 v" #r ", v116, v116, v116\n\t"
#define MUL_F32(r) "v_mul_f32 v" #r ", v116, v116\n\t"
#define CVT_F64_U32(r) "v_cvt_f64_u32 v[" #r ":" #r "+1], v117\n\t"
#define CVT_U32_F64(r) "v_cvt_u32_f64 v" #r ", v[118:119]\n\t"
#define RCP_F64(r) "v_rcp_f64 v[" #r ":" #r "+1], v[118:119]\n\t"
#define FMA_F64(r) "v_fma_f64 v[" #r ":" #r "+1], v[118:119], v[118:119], v[118:119]\n\t"
#define MUL_F64(r) "v_mul_f64 v[" #r ":" #r "+1], v[118:119], v[118:119]\n\t"
#define CMP_U64(r) "v_cmp_ge_u64 s[24:25], v[118:119], v[116:117]\n\t"
#define SUB_CO(r) "v_sub_co_u32 v" #r ", s[24:25], v116, v117\n\t"
#define FFBH(r) "v_ffbh_u32 v" #r ", v117\n\t"
#define LDEXP(r) "v_ldexp_f32 v" #r ", v116, v117\n\t"
    if constexpr (KIND == 0) asm volatile(EIGHT(CVT_F32_U32,,) ::: "memory");
    if constexpr (KIND == 1) asm volatile(EIGHT(CVT_U32_F32,,) ::: "memory");
    if constexpr (KIND == 2) asm volatile(EIGHT(RCP_F32,,) ::: "memory");
    if constexpr (KIND == 3) asm volatile(EIGHT(FMA_F32,,) ::: "memory");
    if constexpr (KIND == 4) asm volatile(EIGHT(MUL_F32,,) ::: "memory");
    if constexpr (KIND == 5) asm volatile(EIGHT(CVT_F64_U32,,) ::: "memory");
    if constexpr (KIND == 6) asm volatile(EIGHT(CVT_U32_F64,,) ::: "memory");
    if constexpr (KIND == 7) asm volatile(EIGHT(RCP_F64,,) ::: "memory");
    if constexpr (KIND == 8) asm volatile(EIGHT(FMA_F64,,) ::: "memory");
    if constexpr (KIND == 9) asm volatile(EIGHT(MUL_F64,,) ::: "memory");
    if constexpr (KIND == 10) asm volatile(EIGHT(CMP_U64,,) ::: "memory", "s24", "s25");
    if constexpr (KIND == 11) asm volatile(EIGHT(SUB_CO,,) ::: "memory", "s24", "s25");
    if constexpr (KIND == 12) asm volatile(EIGHT(FFBH,,) ::: "memory");
    if constexpr (KIND == 13) asm volatile(EIGHT(LDEXP,,) ::: "memory");
    asm volatile("s_sub_u32 s23, s23, 1\n\ts_cmp_lg_u32 s23, 0\n\ts_cbranch_scc1 1b" ::: "s23", "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115");
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int KIND> void run(const char* name) {
    uint64_t* d; (void)hipMalloc(&d, 8);
    const uint32_t iters = 200; uint64_t h;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-16s %5.2f ticks per instruction\n", name, (double)h / (iters * 128.0));
    (void)hipFree(d);
}
int main() {
    run<0>("v_cvt_f32_u32"); run<1>("v_cvt_u32_f32"); run<2>("v_rcp_f32"); run<3>("v_fma_f32"); run<4>("v_mul_f32");
    run<5>("v_cvt_f64_u32"); run<6>("v_cvt_u32_f64"); run<7>("v_rcp_f64"); run<8>("v_fma_f64"); run<9>("v_mul_f64");
    run<10>("v_cmp_ge_u64"); run<11>("v_sub_co_u32"); run<12>("v_ffbh_u32"); run<13>("v_ldexp_f32");
    return 0;
}
