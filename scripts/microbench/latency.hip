// Dependent-chain latency microbenchmark for the integer ops on the coder's critical path (gfx950).
// One wave per SIMD (256 threads/block, 1 block), s_memtime around an unrolled dependent chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define N_ITER 256
#define REPS 16

template <int OP>
__global__ __launch_bounds__(256) void lat_kernel(uint64_t* out, uint32_t seed, uint32_t* sink) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (i * 2654435761u) & 4095u;
    __syncthreads();
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed | 1;
    uint64_t w = ((uint64_t)a << 32) | b;
    uint64_t t0 = 0, t1 = 0;
    for (int rep = 0; rep < REPS; ++rep) {
        if (rep == 1) t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < N_ITER; ++i) {
            if constexpr (OP == 0) a = a + b;                                    // v_add_u32
            else if constexpr (OP == 1) a = __umulhi(a, c);                       // v_mul_hi_u32
            else if constexpr (OP == 2) a = a * c;                                // v_mul_lo_u32
            else if constexpr (OP == 3) w = (uint64_t)(uint32_t)w * c + w;        // v_mad_u64_u32
            else if constexpr (OP == 4) a = __umul24(a, c) + 1;    // v_mad_u32_u24 / mul_u24
            else if constexpr (OP == 5) a = lds[a & 4095];                         // ds_read_b32 dependent
            else if constexpr (OP == 6) w = (w >> 7) + c;                          // v_lshrrev_b64 + add
            else if constexpr (OP == 7) a = (a >= b) ? a - b : a + c;              // cmp + cndmask
            else if constexpr (OP == 8) w = w + (uint64_t)c;                       // 64-bit add
            else if constexpr (OP == 9) { double d = __longlong_as_double(w | 0x3ff0000000000000ull); d = d * 1.0000001 + 0.5; w = __double_as_longlong(d); } // dp fma
            else if constexpr (OP == 10) a = __builtin_amdgcn_alignbit(a, b, 12) ^ c; // alignbit + xor
            else if constexpr (OP == 11) { a = (a & 4095) ; a = lds[a] + (a << 2); }    // and + lds + shift-add
        }
    }
    t1 = __builtin_readcyclecounter();
    sink[threadIdx.x] = a + (uint32_t)w + (uint32_t)(w >> 32);
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <int OP> void run(const char* name, int ops_per_iter) {
    uint64_t* d; uint32_t* s; hipMalloc(&d, 8); hipMalloc(&s, 1024);
    hipLaunchKernelGGL(lat_kernel<OP>, dim3(1), dim3(256), 0, 0, d, 12345u, s);
    hipLaunchKernelGGL(lat_kernel<OP>, dim3(1), dim3(256), 0, 0, d, 12345u, s);
    hipDeviceSynchronize();
    uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8.2f cycles/iter (%d dependent op(s))\n", name, (double)h / ((REPS - 1) * (double)N_ITER), ops_per_iter);
    hipFree(d); hipFree(s);
}

int main() {
    run<0>("v_add_u32", 1); run<1>("v_mul_hi_u32", 1); run<2>("v_mul_lo_u32", 1); run<3>("v_mad_u64_u32 (+dep)", 1);
    run<4>("mul_u24+add", 2); run<5>("and+ds_read_b32", 2); run<6>("lshr64+add64", 2); run<7>("cmp+sub+add+cndmask", 3);
    run<8>("add_u64", 1); run<9>("or+fma_f64", 2); run<10>("alignbit+xor", 2); run<11>("and+ds_read+lshl_add", 3);
    return 0;
}
