// Cost of the 24-instruction encoder step in isolation (one wave per SIMD, no memory traffic): ticks per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define STEP(WITH_DS) \
    "v_lshlrev_b32 v225, s20, v181\n\t" \
    "v_sub_u32 v226, s21, v181\n\t" \
    "v_lshlrev_b32 v228, 8, %[wr]\n\t" \
    "v_cmp_ge_u32 vcc, %[hi], v225\n\t" \
    "v_and_or_b32 v228, v228, s22, v199\n\t" \
    "v_add_u32 v227, v180, v226\n\t" \
    "v_cndmask_b32_e64 v212, %[lo], %[hi], vcc\n\t" \
    "v_cndmask_b32_e64 v213, %[hi], 0, vcc\n\t" \
    WITH_DS \
    "v_addc_co_u32 %[wr], vcc, 0, %[wr], vcc\n\t" \
    "v_mul_hi_u32 v214, v212, v182\n\t" \
    "v_mad_u64_u32 v[216:217], vcc, v213, v182, v[214:215]\n\t" \
    "v_mad_u64_u32 v[218:219], vcc, v212, v183, v[216:217]\n\t" \
    "v_mov_b32 v220, v219\n\t" \
    "v_addc_co_u32 v221, vcc, 0, v215, vcc\n\t" \
    "v_mad_u64_u32 v[222:223], vcc, v213, v183, v[220:221]\n\t" \
    "v_mul_lo_u32 v224, v222, v181\n\t" \
    "v_sub_u32 v224, v212, v224\n\t" \
    "v_cmp_ge_u32 vcc, v224, v181\n\t" \
    "v_mad_u64_u32 v[216:217], s[24:25], v222, v226, v[212:213]\n\t" \
    "v_mad_u32_u24 v217, v223, v226, v217\n\t" \
    "v_cndmask_b32 v224, v180, v227, vcc\n\t" \
    "v_add_co_u32 %[lo], vcc, v216, v224\n\t" \
    "v_addc_co_u32 %[hi], vcc, 0, v217, vcc\n\t"
#define S4(D) STEP(D) STEP(D) STEP(D) STEP(D)
#define S32(D) S4(D) S4(D) S4(D) S4(D) S4(D) S4(D) S4(D) S4(D)

template <int KIND> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t* sink, uint32_t iters) {
    __shared__ uint32_t lds[4096];
    lds[threadIdx.x] = threadIdx.x; __syncthreads();
    uint32_t lo = threadIdx.x * 2654435761u, hi = 77 + threadIdx.x, wr = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    asm volatile(
        "v_mov_b32 v215, 0\n\tv_mov_b32 v180, 100\n\tv_mov_b32 v181, 37\n\tv_mov_b32 v182, 0x5bd1e995\n\tv_mov_b32 v183, 0x6eb3e45\n\t"
        "v_lshlrev_b32 v199, 2, %[tid]\n\ts_mov_b32 s20, 20\n\ts_mov_b32 s21, 4096\n\ts_mov_b32 s22, 0x3f00\n\ts_mov_b32 s23, %[n]\n\t"
        "1:\n\t"
        : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr) : [tid] "v"(threadIdx.x & 63), [n] "s"(iters)
        : "v180","v181","v182","v183","v199","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","s20","s21","s22","s23","s24","s25","vcc","memory");
    if constexpr (KIND == 0) asm volatile(S32("") "s_sub_u32 s23, s23, 1\n\ts_cmp_lg_u32 s23, 0\n\ts_cbranch_scc1 1b" : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr) :: "vcc", "memory");
    else asm volatile(S32("ds_write_b32 v228, %[lo]\n\t") "s_sub_u32 s23, s23, 1\n\ts_cmp_lg_u32 s23, 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt lgkmcnt(0)" : [lo] "+v"(lo), [hi] "+v"(hi), [wr] "+v"(wr) :: "vcc", "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 256 + threadIdx.x] = lo ^ hi ^ wr ^ lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int KIND> void run(const char* name, int blocks) {
    uint64_t* d; uint32_t* s; (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 256 * 256 * 4);
    const uint32_t iters = 200;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, d, s, iters);
    (void)hipDeviceSynchronize();
    uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-22s blocks=%3d: %.1f ticks per step (%d instructions)\n", name, blocks, (double)h / (iters * 32.0), KIND ? 24 : 23);
}
int main() { run<0>("step without ds_write", 1); run<1>("step with ds_write", 1); run<0>("step without ds_write", 256); run<1>("step with ds_write", 256); return 0; }
