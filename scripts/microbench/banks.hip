// Does a lone wave's 8-cycle "dependent" cost come from the RAW dependency or from VGPR operand banks?
// Explicit registers, straight-line code, s_memtime on wave 0 (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 256
#define REPS 6
#define CLOBS "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27"

template <int KIND> __device__ __forceinline__ void body() {
    if constexpr (KIND == 0) asm volatile("v_xor_b32 v10, v10, v21" ::: CLOBS);                 // dep, banks 2/1
    if constexpr (KIND == 1) asm volatile("v_xor_b32 v10, v10, v22" ::: CLOBS);                 // dep, banks 2/2
    if constexpr (KIND == 2) asm volatile("v_xor_b32 v9, v9, v24\n\tv_xor_b32 v10, v10, v24\n\tv_xor_b32 v11, v11, v24" ::: CLOBS);  // 3 indep, no clash
    if constexpr (KIND == 3) asm volatile("v_xor_b32 v9, v9, v24\n\tv_xor_b32 v10, v10, v24" ::: CLOBS);                          // dist 2
    if constexpr (KIND == 4) asm volatile("v_xor_b32 v9, v9, v25\n\tv_xor_b32 v10, v10, v25" ::: CLOBS);                          // dist 2, one clash
    if constexpr (KIND == 5) asm volatile("v_xor_b32 v9, v10, v24\n\tv_xor_b32 v10, v9, v24" ::: CLOBS);                          // dep, dst != src
    if constexpr (KIND == 6) asm volatile("v_xor_b32 v9, v10, v24\n\tv_xor_b32 v13, v9, v24\n\tv_xor_b32 v10, v13, v24" ::: CLOBS); // dep ring of 3
    if constexpr (KIND == 7) asm volatile("v_mad_u64_u32 v[10:11], s[20:21], v24, v25, v[10:11]" ::: CLOBS, "s20", "s21");        // dep mad64
    if constexpr (KIND == 8) asm volatile("v_mad_u64_u32 v[10:11], s[20:21], v24, v25, v[10:11]\n\tv_mad_u64_u32 v[12:13], s[20:21], v24, v25, v[12:13]\n\tv_mad_u64_u32 v[14:15], s[20:21], v24, v25, v[14:15]" ::: CLOBS, "s20", "s21");
    if constexpr (KIND == 9) asm volatile("v_add_co_u32 v10, vcc, v10, v21\n\tv_addc_co_u32 v11, vcc, 0, v11, vcc" ::: CLOBS, "vcc");
    if constexpr (KIND == 10) asm volatile("v_cmp_ge_u32 vcc, v10, v21\n\tv_cndmask_b32 v10, v10, v23, vcc" ::: CLOBS, "vcc");
    if constexpr (KIND == 11) asm volatile("v_cmp_ge_u32 vcc, v10, v21\n\tv_xor_b32 v12, v12, v21\n\tv_cndmask_b32 v10, v10, v23, vcc" ::: CLOBS, "vcc");
    if constexpr (KIND == 12) asm volatile("v_cmp_ge_u32 vcc, v10, v21\n\tv_xor_b32 v12, v12, v21\n\tv_xor_b32 v13, v13, v22\n\tv_cndmask_b32 v10, v10, v23, vcc" ::: CLOBS, "vcc");
    if constexpr (KIND == 13) asm volatile("v_mul_hi_u32 v10, v10, v21" ::: CLOBS);
    if constexpr (KIND == 14) asm volatile("v_mul_hi_u32 v9, v9, v24\n\tv_mul_hi_u32 v10, v10, v24\n\tv_mul_hi_u32 v11, v11, v24" ::: CLOBS);
    if constexpr (KIND == 15) asm volatile("v_xor_b32 v9, v9, v24\n\ts_nop 0\n\tv_xor_b32 v10, v10, v24\n\ts_nop 0" ::: CLOBS);
    // 8 instructions per body so that the s_nop the compiler puts between asm statements is amortised
    if constexpr (KIND == 16) asm volatile("v_xor_b32 v9, v10, v24\n\tv_xor_b32 v9, v11, v24\n\tv_xor_b32 v9, v10, v24\n\tv_xor_b32 v9, v11, v24\n\tv_xor_b32 v9, v10, v24\n\tv_xor_b32 v9, v11, v24\n\tv_xor_b32 v9, v10, v24\n\tv_xor_b32 v9, v11, v24" ::: CLOBS);  // WAW only
    if constexpr (KIND == 17) asm volatile("v_xor_b32 v9, v10, v24\n\tv_xor_b32 v12, v9, v24\n\tv_xor_b32 v13, v12, v24\n\tv_xor_b32 v14, v13, v24\n\tv_xor_b32 v15, v14, v24\n\tv_xor_b32 v16, v15, v24\n\tv_xor_b32 v17, v16, v24\n\tv_xor_b32 v10, v17, v24" ::: CLOBS);  // RAW chain, distinct dst
    if constexpr (KIND == 18) asm volatile("v_xor_b32 v9, v9, v24\n\tv_xor_b32 v9, v9, v24\n\tv_xor_b32 v9, v9, v24\n\tv_xor_b32 v9, v9, v24\n\tv_xor_b32 v9, v9, v24\n\tv_xor_b32 v9, v9, v24\n\tv_xor_b32 v9, v9, v24\n\tv_xor_b32 v9, v9, v24" ::: CLOBS);  // RAW + same dst
    if constexpr (KIND == 19) asm volatile("v_mad_u64_u32 v[10:11], s[20:21], v24, v25, v[12:13]\n\tv_mad_u32_u24 v11, v24, v25, v11\n\tv_mad_u64_u32 v[14:15], s[20:21], v24, v25, v[10:11]\n\tv_mad_u32_u24 v15, v24, v25, v15\n\tv_mad_u64_u32 v[10:11], s[20:21], v24, v25, v[14:15]\n\tv_mad_u32_u24 v11, v24, v25, v11\n\tv_mad_u64_u32 v[14:15], s[20:21], v24, v25, v[10:11]\n\tv_mad_u32_u24 v15, v24, v25, v15" ::: CLOBS, "s20", "s21");  // mad64 then mad24 into its high half
    if constexpr (KIND == 20) asm volatile("v_mad_u64_u32 v[10:11], s[20:21], v24, v25, v[12:13]\n\tv_mad_u32_u24 v16, v24, v25, v11\n\tv_mad_u64_u32 v[14:15], s[20:21], v24, v25, v[10:11]\n\tv_mad_u32_u24 v17, v24, v25, v15\n\tv_mad_u64_u32 v[10:11], s[20:21], v24, v25, v[14:15]\n\tv_mad_u32_u24 v16, v24, v25, v11\n\tv_mad_u64_u32 v[14:15], s[20:21], v24, v25, v[10:11]\n\tv_mad_u32_u24 v17, v24, v25, v15" ::: CLOBS, "s20", "s21");  // same, separate dst
    if constexpr (KIND == 21) asm volatile("v_xor_b32 v9, v10, v24\n\tv_xor_b32 v12, v9, v9\n\tv_xor_b32 v13, v12, v12\n\tv_xor_b32 v14, v13, v13\n\tv_xor_b32 v15, v14, v14\n\tv_xor_b32 v16, v15, v15\n\tv_xor_b32 v17, v16, v16\n\tv_xor_b32 v10, v17, v17" ::: CLOBS);  // RAW chain, both sources
}
static const char* kNames[] = {"dep xor (banks differ)", "dep xor (same bank srcs)", "3 indep xor", "2 chains", "2 chains one clash",
    "dep ping-pong", "dep ring of 3", "dep mad_u64_u32", "3 indep mad_u64_u32", "add_co+addc", "cmp+cndmask (vcc)",
    "cmp+1 filler+cndmask", "cmp+2 filler+cndmask", "dep mul_hi", "3 indep mul_hi", "2 chains + s_nop each", "8x WAW only",
    "8x RAW chain distinct dst", "8x RAW same dst", "4x mad64+mad24 hi (same dst)", "4x mad64+mad24 (other dst)", "8x RAW chain both srcs"};
static const int kInstr[] = {1, 1, 3, 2, 2, 2, 3, 1, 3, 2, 2, 3, 4, 1, 3, 4, 8, 8, 8, 8, 8, 8};

template <int KIND> __global__ __launch_bounds__(256) void k(uint64_t* out) {
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    asm volatile("v_mov_b32 v26, 0" ::: CLOBS);
    uint64_t t0 = 0;
    for (int rep = 0; rep < REPS; ++rep) {
        if (rep == 1) t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < N_ITER; ++i) body<KIND>();
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0 + (lds[5] == 77);
}
template <int KIND> void run() {
    uint64_t* d; (void)hipMalloc(&d, 8);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<KIND>), dim3(1), dim3(256), 0, 0, d);
    (void)hipDeviceSynchronize();
    uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); (void)hipFree(d);
    double per_body = (double)h / ((REPS - 1) * (double)N_ITER);
    printf("%-28s %6.2f ticks per body, %5.2f per instruction (%d instr)\n", kNames[KIND], per_body, per_body / kInstr[KIND], kInstr[KIND]);
}
int main() { setvbuf(stdout, NULL, _IONBF, 0);
    run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); run<5>(); run<6>(); run<7>(); run<8>(); run<9>(); run<10>(); run<11>(); run<12>();
    run<13>(); run<14>(); run<15>(); run<16>(); run<17>(); run<18>(); run<19>(); run<20>(); run<21>();
    return 0;
}
