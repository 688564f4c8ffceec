// gfx950 questions behind the per-stream-table coder:
//  (1) do DS reads at addresses that are only 2-byte (b32) / 2- or 4-byte (b64, read2_b64, b128) aligned return the
//      right bytes, and what do they cost for a lone wave?
//  (2) do SDWA operand selects give the expected values on VOP2 / VOPC (v_mul_u32_u24, v_cmp_ge_u32, v_cndmask_b32,
//      v_add_u32) straight behind the VALU instruction that produced the operand (no wait states inserted by hand)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(64) void k_unaligned(uint32_t* out, uint64_t* ticks) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)(i * 7 + 1);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint16_t*)lds;
    // lane l reads at element offset 37 * l + (l & 3): every alignment class mod 8 elements occurs
    const uint32_t el = 37u * threadIdx.x + (threadIdx.x & 3u);
    const uint32_t addr = base + 2u * el;
    uint32_t r32, r64[2], r2x64[4], r128[4];
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r32) : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)r64) : "v"(addr) : "memory");
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(__uint128_t*)r128) : "v"(addr) : "memory");
    const uint32_t addr8 = addr & ~7u;
    asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(__uint128_t*)r2x64) : "v"(addr8) : "memory");
    uint32_t* o = out + threadIdx.x * 16;
    o[0] = el; o[1] = r32; o[2] = r64[0]; o[3] = r64[1];
    for (int i = 0; i < 4; ++i) { o[4 + i] = r128[i]; o[8 + i] = r2x64[i]; }

    // timing: dependent chains of 256 reads, the address of each derived from the previous result (& 0 == no change)
    uint32_t a = addr;
    uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0, %0\n\tv_add_u32 %1, %1, %0" : "=&v"(v), "+v"(a) :: "memory");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t b = addr8;
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        asm volatile("ds_read2_b64 v[100:103], %0 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 v100, 0, v100\n\tv_add_u32 %0, %0, v100" : "+v"(b) :: "memory", "v100", "v101", "v102", "v103");
    }
    uint64_t t2 = __builtin_readcyclecounter();
    uint32_t c = addr & ~3u;
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0, %0\n\tv_add_u32 %1, %1, %0" : "=&v"(v), "+v"(c) :: "memory");
    }
    uint64_t t3 = __builtin_readcyclecounter();
    uint32_t d = addr;
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
        uint32_t v;
        asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0, %0\n\tv_add_u32 %1, %1, %0" : "=&v"(v), "+v"(d) :: "memory");
    }
    uint64_t t4 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = t2 - t1; ticks[2] = t3 - t2; ticks[3] = t4 - t3; }
    o[12] = a + b + c + d;
}

__global__ __launch_bounds__(64) void k_sdwa(uint32_t* out) {
    const uint32_t l = threadIdx.x;
    uint32_t x = 0x12340000u * (l + 1) + 0x0101u * l + 7u;   // arbitrary
    uint32_t e = ((100u + l) << 16) | (3000u + l);              // W1 = 100+l, W0 = 3000+l
    uint32_t q = 0xfedcba98u - 77777u * l;
    uint32_t r_mul, r_add, r_cnd0, r_cnd1, r_cmp;
    uint64_t sd;
    asm volatile(
        "v_add_u32 %[x], 1, %[x]\n\t"                                    // producer straight in front of the SDWA readers
        "v_mul_u32_u24_sdwa %[rmul], %[q], %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_add_u32_sdwa %[radd], %[x], %[e] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_cmp_ge_u32_sdwa vcc, %[x], %[e] src0_sel:WORD_1 src1_sel:WORD_0\n\t"
        "v_cndmask_b32_sdwa %[rc0], %[e], %[e], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_cndmask_b32_e64 %[rcmp], 0, 1, vcc\n\t"
        "v_cmp_ge_u32_sdwa %[sd], %[e], %[x] src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %[rc1], 0, 1, %[sd]\n\t"
        : [x] "+v"(x), [rmul] "=&v"(r_mul), [radd] "=&v"(r_add), [rc0] "=&v"(r_cnd0), [rc1] "=&v"(r_cnd1), [rcmp] "=&v"(r_cmp), [sd] "=&s"(sd)
        : [e] "v"(e), [q] "v"(q) : "vcc");
    uint32_t* o = out + l * 8;
    o[0] = x; o[1] = e; o[2] = q; o[3] = r_mul; o[4] = r_add; o[5] = r_cnd0; o[6] = r_cmp; o[7] = r_cnd1;
}

int main() {
    uint32_t* d; uint64_t* t;
    (void)hipMalloc(&d, 64 * 16 * 4); (void)hipMalloc(&t, 64);
    hipLaunchKernelGGL(k_unaligned, dim3(1), dim3(64), 0, 0, d, t);
    hipError_t e = hipDeviceSynchronize();
    printf("unaligned kernel: %s\n", hipGetErrorString(e));
    std::vector<uint32_t> h(64 * 16); uint64_t ht[4];
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(ht, t, 32, hipMemcpyDeviceToHost);
    auto ref16 = [](uint32_t i) { return (uint32_t)(uint16_t)(i * 7 + 1); };
    int bad32 = 0, bad64 = 0, bad128 = 0, bad2x = 0, n_al[3] = {0, 0, 0};
    for (int l = 0; l < 64; ++l) {
        const uint32_t* o = &h[l * 16];
        const uint32_t el = o[0];
        auto w = [&](uint32_t i) { return ref16(i) | (ref16(i + 1) << 16); };
        if (o[1] != w(el)) ++bad32;
        if (o[2] != w(el) || o[3] != w(el + 2)) ++bad64;
        for (int i = 0; i < 4; ++i) if (o[4 + i] != w(el + 2 * i)) { ++bad128; break; }
        const uint32_t e8 = el & ~3u;
        for (int i = 0; i < 4; ++i) if (o[8 + i] != w(e8 + 2 * i)) { ++bad2x; break; }
        if (l < 4) printf("lane %d el %u (byte %% 16 = %u): b32 %08x want %08x | b64 %08x %08x | b128 %08x.. want %08x\n", l, el, (2 * el) % 16, o[1], w(el), o[2], o[3], o[4], w(el));
        (void)n_al;
    }
    printf("mismatching lanes: ds_read_b32 %d, ds_read_b64 %d, ds_read_b128 %d (2-byte aligned addresses); ds_read2_b64 (8-byte aligned) %d\n", bad32, bad64, bad128, bad2x);
    printf("dependent-chain ticks per read: b32 unaligned %.1f, read2_b64 aligned %.1f, b32 aligned %.1f, u16 %.1f\n", ht[0] / 256.0, ht[1] / 256.0, ht[2] / 256.0, ht[3] / 256.0);

    hipLaunchKernelGGL(k_sdwa, dim3(1), dim3(64), 0, 0, d);
    e = hipDeviceSynchronize();
    printf("sdwa kernel: %s\n", hipGetErrorString(e));
    (void)hipMemcpy(h.data(), d, 64 * 8 * 4, hipMemcpyDeviceToHost);
    int bmul = 0, badd = 0, bc0 = 0, bcmp = 0, bc1 = 0;
    for (int l = 0; l < 64; ++l) {
        const uint32_t* o = &h[l * 8];
        const uint32_t x = o[0], ee = o[1], q = o[2];
        const uint32_t w1 = ee >> 16, w0 = ee & 0xffff, xw0 = x & 0xffff, xw1 = x >> 16;
        if (o[3] != (uint32_t)((uint64_t)(q & 0xffffff) * w1)) ++bmul;
        if (o[4] != xw0 + w1) ++badd;
        const bool ge = xw1 >= w0;
        if (o[6] != (ge ? 1u : 0u)) ++bcmp;
        if (o[5] != (ge ? w1 : w0)) ++bc0;
        if (o[7] != ((w0 >= xw1) ? 1u : 0u)) ++bc1;
    }
    printf("sdwa mismatching lanes: mul_u32_u24 %d, add_u32 %d, cmp->vcc %d, cndmask_sdwa %d, cmp->sgpr %d\n", bmul, badd, bcmp, bc0, bc1);
    return 0;
}
