// What ONE memory instruction costs a lone wave per SIMD in issue time (gfx950): a loop of 16 independent VALU
// instructions + one memory instruction, against the same loop without it.  256 workgroups x 256 threads (one wave per
// SIMD, every CU busy); the buffers are small (L2 / TCP resident) so that bandwidth does not enter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define VALU16 \
    "v_xor_b32 v100, v100, v116\n\tv_xor_b32 v101, v101, v116\n\tv_xor_b32 v102, v102, v116\n\tv_xor_b32 v103, v103, v116\n\t" \
    "v_xor_b32 v104, v104, v116\n\tv_xor_b32 v105, v105, v116\n\tv_xor_b32 v106, v106, v116\n\tv_xor_b32 v107, v107, v116\n\t" \
    "v_xor_b32 v108, v108, v116\n\tv_xor_b32 v109, v109, v116\n\tv_xor_b32 v110, v110, v116\n\tv_xor_b32 v111, v111, v116\n\t" \
    "v_xor_b32 v112, v112, v116\n\tv_xor_b32 v113, v113, v116\n\tv_xor_b32 v114, v114, v116\n\tv_xor_b32 v115, v115, v116\n\t"
#define R8(X) X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t* buf, uint32_t iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
    lds[threadIdx.x] = threadIdx.x; __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // global: row (lane >> 3) of a 6 KiB-stride matrix, 16 bytes per lane (the encoder's tile fetch shape); per-wave region
    const uint32_t goff = ((blockIdx.x * 4 + wave) * 64 + (lane >> 3)) * 6144 + (lane & 7) * 16;
    const uint32_t laddr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)lds + wave * 4096 + lane * 16;
    uint64_t t0 = __builtin_readcyclecounter();
    asm volatile("v_mov_b32 v116, 7\n\ts_mov_b32 s23, %[n]\n\t1:\n\t" :: [n] "s"(iters) : "v116", "s23");
    if constexpr (KIND == 0) asm volatile(R8(VALU16) ::: "memory");
    if constexpr (KIND == 1) asm volatile(R8(VALU16 "global_load_dwordx4 v[120:123], %[o], %[b]\n\t") :: [o] "v"(goff), [b] "s"(buf) : "memory", "v120", "v121", "v122", "v123");
    if constexpr (KIND == 2) asm volatile(R8(VALU16 "global_store_dwordx4 %[o], v[100:103], %[b]\n\t") :: [o] "v"(goff), [b] "s"(buf) : "memory");
    if constexpr (KIND == 3) asm volatile(R8(VALU16 "ds_write_b128 %[a], v[100:103]\n\t") :: [a] "v"(laddr) : "memory");
    if constexpr (KIND == 4) asm volatile(R8(VALU16 "ds_read_b128 v[120:123], %[a]\n\t") :: [a] "v"(laddr) : "memory", "v120", "v121", "v122", "v123");
    if constexpr (KIND == 5) asm volatile(R8(VALU16 "ds_write_b32 %[a], v100\n\t") :: [a] "v"(laddr) : "memory");
    if constexpr (KIND == 6) asm volatile(R8(VALU16 "ds_read_b64 v[120:121], %[a]\n\t") :: [a] "v"(laddr) : "memory", "v120", "v121");
    if constexpr (KIND == 7) asm volatile(R8(VALU16 "global_load_dword v120, %[o], %[b]\n\t") :: [o] "v"(goff), [b] "s"(buf) : "memory", "v120");
    if constexpr (KIND == 8) asm volatile(R8(VALU16 "s_or_b64 s[24:25], s[24:25], s[26:27]\n\t") ::: "memory", "s24", "s25");
    if constexpr (KIND == 9) asm volatile(R8(VALU16 "s_waitcnt lgkmcnt(0)\n\t") ::: "memory");
    asm volatile("s_sub_u32 s23, s23, 1\n\ts_cmp_lg_u32 s23, 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "s23", "memory",
                 "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115");
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND> double run(const char* name, double base) {
    uint64_t* d; uint32_t* b; (void)hipMalloc(&d, 8); (void)hipMalloc(&b, (size_t)1024 * 64 * 6144 + 65536);
    const uint32_t iters = 500; uint64_t h;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, b, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / (iters * 8.0);
    printf("%-26s %6.1f ticks per 16 VALU + 1%s\n", name, per, base > 0 ? "" : " (nothing)");
    if (base > 0) printf("%-26s   -> the memory instruction costs the wave %.1f ticks\n", "", per - base);
    (void)hipFree(d); (void)hipFree(b);
    return per;
}

int main() {
    const double base = run<0>("16 x v_xor", 0);
    run<1>("global_load_dwordx4", base); run<7>("global_load_dword", base); run<2>("global_store_dwordx4", base);
    run<3>("ds_write_b128", base); run<4>("ds_read_b128", base); run<5>("ds_write_b32", base); run<6>("ds_read_b64", base);
    run<8>("s_or_b64", base); run<9>("s_waitcnt lgkmcnt(0)", base);
    return 0;
}
