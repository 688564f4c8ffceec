// What a 16-byte-per-lane vector memory instruction costs a lone wave per SIMD (gfx950) as a function of how many
// different 64-byte segments its 64 lanes touch: the word slabs of the coders are per stream (one lane = one stream =
// one line per instruction) -- is it worth letting four lanes serve one stream's 64-byte group?
// Loop: 80 independent VALU instructions + one memory instruction; 256 workgroups x 256 threads; per-wave regions are
// walked so that lines are not reused by the next instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define V8 "v_xor_b32 v100, v100, v116\n\tv_xor_b32 v101, v101, v116\n\tv_xor_b32 v102, v102, v116\n\tv_xor_b32 v103, v103, v116\n\t" \
           "v_xor_b32 v104, v104, v116\n\tv_xor_b32 v105, v105, v116\n\tv_xor_b32 v106, v106, v116\n\tv_xor_b32 v107, v107, v116\n\t"
#define V80 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8

template <int KIND>   // 0: nothing; 1/2: store with 64 / 16 lines; 3/4: load with 64 / 16 lines
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t* buf, uint32_t iters) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t stride = 6144;                                    // bytes between two streams' slabs
    uint32_t off;
    if (KIND == 1 || KIND == 3) off = lane * stride;                 // one stream per lane: 64 lines
    else off = (lane >> 2) * stride + (lane & 3) * 16;               // four lanes per stream: 16 lines of 64 B
    const uint64_t bv = (uint64_t)reinterpret_cast<uintptr_t>(reinterpret_cast<unsigned char*>(buf) + wave * 64 * stride);
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(bv >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)bv);
    uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t o = off + (it % 96) * 64;                     // walk along the slabs
        if constexpr (KIND == 0) asm volatile("v_mov_b32 v116, 7\n\t" V80 ::: "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116");
        if constexpr (KIND == 1 || KIND == 2)
            asm volatile("v_mov_b32 v116, 7\n\t" V80 "global_store_dwordx4 %[o], v[100:103], %[b]\n\t" :: [o] "v"(o), [b] "s"(base)
                         : "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116");
        if constexpr (KIND == 3 || KIND == 4)
            asm volatile("v_mov_b32 v116, 7\n\t" V80 "global_load_dwordx4 v[120:123], %[o], %[b]\n\t" :: [o] "v"(o), [b] "s"(base)
                         : "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116","v120","v121","v122","v123");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int KIND> double run(const char* name, double base, uint32_t* buf) {
    uint64_t* d; (void)hipMalloc(&d, 8);
    const uint32_t iters = 2000; uint64_t h;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, buf, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / iters;
    printf("%-44s %7.1f ticks per 80 VALU + 1", name, per);
    if (base > 0) printf("   -> the memory instruction costs the wave %.1f ticks", per - base);
    printf("\n");
    (void)hipFree(d);
    return per;
}
// The coders' real shape: per 320 VALU instructions the wave moves ONE 64-byte group per stream (4 instructions): chunk k of
// every stream (A: 64 lines per instruction, what round 1 / early round 2 do) or whole groups of 16 streams (B).
template <int KIND>   // 5/6: stores A/B, 7/8: loads A/B, 9: no memory instruction
__global__ __launch_bounds__(256) void g(uint64_t* out, uint32_t* buf, uint32_t iters) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t stride = 6144;
    const uint64_t bv = (uint64_t)reinterpret_cast<uintptr_t>(reinterpret_cast<unsigned char*>(buf) + wave * 64 * stride);
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(bv >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)bv);
    uint32_t o[4];
    uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        for (int q = 0; q < 4; ++q)
            o[q] = ((KIND & 1) ? lane * stride + q * 16 : ((lane >> 2) + 16 * q) * stride + (lane & 3) * 16) + (it % 96) * 64;
        if constexpr (KIND == 9) asm volatile("v_mov_b32 v116, 7\n\t" V80 V80 V80 V80 ::: "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116");
        if constexpr (KIND == 5 || KIND == 6)
            asm volatile("v_mov_b32 v116, 7\n\t" V80 "global_store_dwordx4 %[o0], v[100:103], %[b]\n\t" V80 "global_store_dwordx4 %[o1], v[100:103], %[b]\n\t"
                         V80 "global_store_dwordx4 %[o2], v[100:103], %[b]\n\t" V80 "global_store_dwordx4 %[o3], v[100:103], %[b]\n\t"
                         :: [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [b] "s"(base)
                         : "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116");
        if constexpr (KIND == 7 || KIND == 8)
            asm volatile("v_mov_b32 v116, 7\n\t" V80 "global_load_dwordx4 v[120:123], %[o0], %[b]\n\t" V80 "global_load_dwordx4 v[124:127], %[o1], %[b]\n\t"
                         V80 "global_load_dwordx4 v[128:131], %[o2], %[b]\n\t" V80 "global_load_dwordx4 v[132:135], %[o3], %[b]\n\t"
                         :: [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [b] "s"(base)
                         : "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116","v120","v121","v122","v123","v124","v125","v126","v127",
                           "v128","v129","v130","v131","v132","v133","v134","v135");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int KIND> double rung(const char* name, double base, uint32_t* buf) {
    uint64_t* d; (void)hipMalloc(&d, 8);
    const uint32_t iters = 2000; uint64_t h;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(g<KIND>, dim3(256), dim3(256), 0, 0, d, buf, iters);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / iters;
    printf("%-52s %7.1f ticks per 320 VALU + 4", name, per);
    if (base > 0) printf("   -> %.1f ticks per memory instruction", (per - base) / 4);
    printf("\n");
    (void)hipFree(d);
    return per;
}

int main() {
    uint32_t* buf; (void)hipMalloc(&buf, (size_t)1024 * 64 * 6144 + (1 << 20)); (void)hipMemset(buf, 1, (size_t)1024 * 64 * 6144);
    const double base = run<0>("80 x v_xor", 0, buf);
    run<1>("store, one stream per lane (64 lines)", base, buf); run<2>("store, four lanes per stream (16 lines)", base, buf);
    run<3>("load, one stream per lane (64 lines)", base, buf); run<4>("load, four lanes per stream (16 lines)", base, buf);
    const double b2 = rung<9>("320 x v_xor", 0, buf);
    rung<5>("64-byte groups: stores, chunk k of every stream", b2, buf); rung<6>("64-byte groups: stores, four lanes per stream", b2, buf);
    rung<7>("64-byte groups: loads, chunk k of every stream", b2, buf); rung<8>("64-byte groups: loads, four lanes per stream", b2, buf);
    return 0;
}
