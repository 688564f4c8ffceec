// Per-instruction issue cost (ILP=4 independent chains) and dependent latency (ILP=1) for one wave per SIMD on
// gfx950, for the integer/f64 instructions an rANS step can be built from.  Timing with s_memtime on wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITER 256
#define REPS 6

// (no "vcc" clobber: the compiler would put an s_nop between consecutive asm statements, and for a lone wave
// every instruction, scalar ones included, costs an issue slot.  Ops that write vcc use s[20:21] instead.)
#define OP32(name, text)                                                                                             \
    struct name { static constexpr const char* label = #name;                                                        \
        __device__ static void op(uint32_t& a, uint32_t& b, uint32_t c, uint32_t d) { asm volatile(text : "+v"(a), "+v"(b) : "v"(c), "v"(d)  CLOB); } };
#define OP64(name, text)                                                                                             \
    struct name { static constexpr const char* label = #name;                                                        \
        __device__ static void op(uint64_t& a, uint64_t& b, uint32_t c, uint32_t d) { asm volatile(text : "+v"(a), "+v"(b) : "v"(c), "v"(d)  CLOB); } };

#undef CLOB
#define CLOB 
OP32(xor32, "v_xor_b32 %0, %0, %2")
#undef CLOB
#define CLOB 
OP32(mul_lo_u32, "v_mul_lo_u32 %0, %0, %2")
#undef CLOB
#define CLOB 
OP32(mul_hi_u32, "v_mul_hi_u32 %0, %0, %2")
#undef CLOB
#define CLOB 
OP32(mul_u32_u24, "v_mul_u32_u24 %0, %0, %2")
#undef CLOB
#define CLOB 
OP32(mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %2")
#undef CLOB
#define CLOB 
OP32(mad_u32_u24, "v_mad_u32_u24 %0, %0, %2, %3")
#undef CLOB
#define CLOB : "s20", "s21"
OP32(add_co, "v_add_co_u32 %0, s[20:21], %0, %2")
#undef CLOB
#define CLOB : "s20", "s21"
OP32(addc_pair, "v_add_co_u32 %0, s[20:21], %0, %2\n\tv_addc_co_u32 %1, s[20:21], 0, %1, s[20:21]")
#undef CLOB
#define CLOB 
OP32(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
#undef CLOB
#define CLOB 
OP32(fma_f32, "v_fma_f32 %0, %0, %2, %3")
#undef CLOB
#define CLOB 
OP32(rcp_f32, "v_rcp_f32 %0, %0")
#undef CLOB
#define CLOB : "s20", "s21"
OP32(cndmask, "v_cndmask_b32_e64 %0, %0, %2, s[20:21]")
#undef CLOB
#define CLOB : "s20", "s21"
OP32(cmp_cnd, "v_cmp_ge_u32 s[20:21], %0, %2\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %3, s[20:21]")
#undef CLOB
#define CLOB 
OP32(alignbit, "v_alignbit_b32 %0, %0, %2, 12")
#undef CLOB
#define CLOB 
OP32(bfe, "v_bfe_u32 %0, %0, 3, 12")
#undef CLOB
#define CLOB 
OP32(and_or, "v_and_or_b32 %0, %0, %2, %3")
#undef CLOB
#define CLOB 
OP32(add3, "v_add3_u32 %0, %0, %2, %3")
#undef CLOB
#define CLOB 
OP32(perm, "v_perm_b32 %0, %0, %2, %3")
#undef CLOB
#define CLOB 
OP32(sad, "v_sad_u32 %0, %0, %2, %3")
#undef CLOB
#define CLOB : "s20", "s21"
OP64(mad_u64_u32, "v_mad_u64_u32 %0, s[20:21], %2, %3, %0")
#undef CLOB
#define CLOB 
OP64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1")
#undef CLOB
#define CLOB 
OP64(lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
#undef CLOB
#define CLOB 
OP64(lshrrev_b64, "v_lshrrev_b64 %0, 3, %0")
#undef CLOB
#define CLOB 
OP64(fma_f64, "v_fma_f64 %0, %0, %1, %1")
#undef CLOB
#define CLOB 
OP64(mul_f64, "v_mul_f64 %0, %0, %1")
#undef CLOB
#define CLOB 
OP64(add_f64, "v_add_f64 %0, %0, %1")
#undef CLOB
#define CLOB 
OP64(cvt_f64_u32, "v_cvt_f64_u32 %0, %2")
#undef CLOB
#define CLOB 
OP64(ldexp_f64, "v_ldexp_f64 %0, %0, %2")
#undef CLOB
#define CLOB 
OP64(rcp_f64, "v_rcp_f64 %0, %0")
#undef CLOB
#define CLOB 
OP64(floor_f64, "v_floor_f64 %0, %0")
#undef CLOB
#define CLOB 
OP64(cvt_u32_f64, "v_cvt_u32_f64 %2, %0")  // writes a dummy input reg, fine for timing

template <class OP, class T, int ILP> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed, T* sink) {
    T a[ILP], b[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) { a[i] = (T)seed * (i + 3) + threadIdx.x; b[i] = (T)seed * (i + 7) + 1; }
    uint32_t c = seed | 1, d = seed ^ 0x5bd1e995;
    uint64_t t0 = 0;
    for (int rep = 0; rep < REPS; ++rep) {
        if (rep == 1) t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
            for (int j = 0; j < ILP; ++j) OP::op(a[j], b[j], c, d);
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    T s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i] + b[i];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <class OP, class T, int ILP> double run1() {
    uint64_t* d; T* s; (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 4096);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<OP, T, ILP>), dim3(1), dim3(256), 0, 0, d, 12345u, s);
    (void)hipDeviceSynchronize();
    uint64_t h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(s);
    return (double)h / ((REPS - 1) * (double)N_ITER * ILP);
}
template <class OP, class T> void run() {
    printf("%-16s dependent %6.2f   4 chains %6.2f   cycles(s_memtime ticks)/instr\n", OP::label, run1<OP, T, 1>(), run1<OP, T, 4>());
}

int main() {
    run<xor32, uint32_t>(); run<mul_lo_u32, uint32_t>(); run<mul_hi_u32, uint32_t>(); run<mul_u32_u24, uint32_t>();
    run<mul_hi_u32_u24, uint32_t>(); run<mad_u32_u24, uint32_t>(); run<add_co, uint32_t>(); run<addc_pair, uint32_t>();
    run<cvt_f32_u32, uint32_t>(); run<fma_f32, uint32_t>(); run<rcp_f32, uint32_t>(); run<cndmask, uint32_t>(); run<cmp_cnd, uint32_t>();
    run<alignbit, uint32_t>(); run<bfe, uint32_t>(); run<and_or, uint32_t>(); run<add3, uint32_t>(); run<perm, uint32_t>(); run<sad, uint32_t>();
    run<mad_u64_u32, uint64_t>(); run<lshl_add_u64, uint64_t>(); run<lshlrev_b64, uint64_t>(); run<lshrrev_b64, uint64_t>();
    run<fma_f64, uint64_t>(); run<mul_f64, uint64_t>(); run<add_f64, uint64_t>(); run<cvt_f64_u32, uint64_t>(); run<ldexp_f64, uint64_t>();
    run<rcp_f64, uint64_t>(); run<floor_f64, uint64_t>(); run<cvt_u32_f64, uint64_t>();
    return 0;
}
