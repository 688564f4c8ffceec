// What a global_load_dwordx4 that MISSES to HBM costs a lone wave per SIMD in issue time (gfx950), for the shapes of the
// coder's tile fetch.  256 x 4 waves stream through a 1 GiB matrix of 65536 rows x 16 KiB exactly like the encoder (a
// tile = 64 rows x 128 B per wave, 8 loads of 8 rows x 128 B), with 80 independent VALU instructions per load, against
// the same loop without the loads.
//   KIND 1: 8 rows x 128 B per instruction (the encoder's shape)      KIND 2: the same without the nt hint
//   KIND 3: 1 row x 1 KiB per instruction (64 lanes along one row)    KIND 4: 2 rows x 512 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define V8 "v_xor_b32 v100, v100, v116\n\tv_xor_b32 v101, v101, v116\n\tv_xor_b32 v102, v102, v116\n\tv_xor_b32 v103, v103, v116\n\t" \
           "v_xor_b32 v104, v104, v116\n\tv_xor_b32 v105, v105, v116\n\tv_xor_b32 v106, v106, v116\n\tv_xor_b32 v107, v107, v116\n\t"
#define V80 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8

template <int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, const uint32_t* buf, uint32_t tiles, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t N = 4096;    // symbols per row
    uint32_t goff[8];
    for (int q = 0; q < 8; ++q) {
        if (KIND == 1 || KIND == 2) goff[q] = (uint32_t)((((lane >> 3) + 8 * q) * N + 4 * (lane & 7)) * 4);           // 8 rows x 128 B
        if (KIND == 3) goff[q] = (uint32_t)((q * N + 4 * lane) * 4);                                                    // row q, 1 KiB (advance 8 rows per tile)
        if (KIND == 4) goff[q] = (uint32_t)((((lane >> 5) + 2 * q) * N + 4 * (lane & 31)) * 4);                         // 2 rows x 512 B
        if (KIND == 0) goff[q] = 0;
    }
    uint64_t base = (uint64_t)(uintptr_t)(buf + wave * 64 * N);
    uint32_t step = (KIND == 3) ? 8 * N * 4 : (KIND == 4 ? 16 * N * 4 : 128);       // bytes the base advances per tile
    // KIND 3 / 4 walk rows instead of columns: a "tile" is then 8 rows x 1 KiB (resp. 16 rows x 512 B); wrap inside the wave's 1 MiB
    uint32_t acc = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t t = 0; t < tiles; ++t) {
        const uint64_t b = base + (uint64_t)((t * (uint64_t)step) % (64 * N * 4));
        uint32_t bl = __builtin_amdgcn_readfirstlane((uint32_t)b), bh = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        uint64_t sb = ((uint64_t)bh << 32) | bl;
        if constexpr (KIND == 0) {
            asm volatile("v_mov_b32 v116, 7\n\t" V80 V80 V80 V80 V80 V80 V80 V80 ::: "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116");
        } else if constexpr (KIND == 2) {
            asm volatile("v_mov_b32 v116, 7\n\t"
                V80 "global_load_dwordx4 v[120:123], %[o0], %[b]\n\t" V80 "global_load_dwordx4 v[124:127], %[o1], %[b]\n\t"
                V80 "global_load_dwordx4 v[128:131], %[o2], %[b]\n\t" V80 "global_load_dwordx4 v[132:135], %[o3], %[b]\n\t"
                V80 "global_load_dwordx4 v[136:139], %[o4], %[b]\n\t" V80 "global_load_dwordx4 v[140:143], %[o5], %[b]\n\t"
                V80 "global_load_dwordx4 v[144:147], %[o6], %[b]\n\t" V80 "global_load_dwordx4 v[148:151], %[o7], %[b]\n\t"
                :: [b] "s"(sb), [o0] "v"(goff[0]), [o1] "v"(goff[1]), [o2] "v"(goff[2]), [o3] "v"(goff[3]), [o4] "v"(goff[4]), [o5] "v"(goff[5]), [o6] "v"(goff[6]), [o7] "v"(goff[7])
                : "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151");
        } else {
            asm volatile("v_mov_b32 v116, 7\n\t"
                V80 "global_load_dwordx4 v[120:123], %[o0], %[b] nt\n\t" V80 "global_load_dwordx4 v[124:127], %[o1], %[b] nt\n\t"
                V80 "global_load_dwordx4 v[128:131], %[o2], %[b] nt\n\t" V80 "global_load_dwordx4 v[132:135], %[o3], %[b] nt\n\t"
                V80 "global_load_dwordx4 v[136:139], %[o4], %[b] nt\n\t" V80 "global_load_dwordx4 v[140:143], %[o5], %[b] nt\n\t"
                V80 "global_load_dwordx4 v[144:147], %[o6], %[b] nt\n\t" V80 "global_load_dwordx4 v[148:151], %[o7], %[b] nt\n\t"
                :: [b] "s"(sb), [o0] "v"(goff[0]), [o1] "v"(goff[1]), [o2] "v"(goff[2]), [o3] "v"(goff[3]), [o4] "v"(goff[4]), [o5] "v"(goff[5]), [o6] "v"(goff[6]), [o7] "v"(goff[7])
                : "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v116","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND> double run(const char* name, double base, const uint32_t* buf) {
    uint64_t* d; uint32_t* s; (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 256 * 256 * 4);
    const uint32_t tiles = 128; uint64_t h;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, buf, tiles, s);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, buf, tiles, s);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / tiles;
    printf("%-34s %7.1f ticks per tile (1280 VALU + 8 loads), kernel %.3f ms", name, per, ms);
    if (base > 0) printf("   -> %.1f ticks per load", (per - base) / 8);
    printf("\n");
    return per;
}

int main() {
    uint32_t* buf; (void)hipMalloc(&buf, (size_t)1 << 30); (void)hipMemset(buf, 1, (size_t)1 << 30);
    const double base = run<0>("no loads", 0, buf);
    run<1>("8 rows x 128 B, nt", base, buf); run<2>("8 rows x 128 B", base, buf); run<3>("1 row x 1 KiB, nt", base, buf); run<4>("2 rows x 512 B, nt", base, buf);
    return 0;
}
