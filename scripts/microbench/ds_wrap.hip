// Does the LDS address adder wrap?  ds_read_b32 with a NEGATIVE 32-bit VGPR address plus a positive immediate offset, and the SDWA
// form  v_lshlrev_b32_sdwa ea, four, sext(byte k of a packed dword)  that would produce such addresses (an int8 symbol << 4
// against a table centred at LDS address 2048).  Prints what comes back for symbols -128 .. 127.   (round 5, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(64) void k(uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* t = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < 4096; i += 64) t[i] = 0x1000000u + (uint32_t)i;      // word i at byte address 4 i
    __syncthreads();
    const int lane = threadIdx.x;
    uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    for (int rep = 0; rep < 4; ++rep) {
        const int8_t s = (int8_t)(lane * 4 + rep - 128);
        const uint32_t packed = ((uint32_t)(uint8_t)s) << (8 * rep) | (0x5a5a5a5au & ~(0xffu << (8 * rep)));
        uint32_t ea = 0, v = 0, four = 4;
        if (rep == 0) asm volatile("v_lshlrev_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(ea) : "v"(four), "v"(packed));
        if (rep == 1) asm volatile("v_lshlrev_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ea) : "v"(four), "v"(packed));
        if (rep == 2) asm volatile("v_lshlrev_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(ea) : "v"(four), "v"(packed));
        if (rep == 3) asm volatile("v_lshlrev_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ea) : "v"(four), "v"(packed));
        ea += base;
        asm volatile("ds_read_b32 %0, %1 offset:2048\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ea) : "memory");
        out[2 * (lane * 4 + rep)] = ea;
        out[2 * (lane * 4 + rep) + 1] = v;
    }
}

int main() {
    uint32_t* d; hipMalloc(&d, 2 * 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d);
    uint32_t h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) {
        const int s = i - 128;
        const uint32_t want = 0x1000000u + (uint32_t)((2048 + 16 * s) / 4);
        if (h[2 * i + 1] != want) { if (bad < 8) printf("symbol %d: ea %08x read %08x want %08x\n", s, h[2 * i], h[2 * i + 1], want); ++bad; }
    }
    printf("ds_read with negative address + offset: %d of 256 wrong%s\n", bad, bad ? "" : "  (the LDS address adder wraps: sext SDWA addresses are usable)");
    return 0;
}
