// Do the D16 forms of the LDS byte loads keep the other half of their destination on this chip?  ds_read_u8_d16 writes bits 0 .. 15
// (the byte zero-extended), ds_read_u8_d16_hi bits 16 .. 31; with SRAM-ECC the hardware may zero the other half instead of keeping it
// (LLVM: d16PreservesUnusedBits).  If they keep it, two decoded symbols share a register without a VALU instruction.   (round 5, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(64) void k(uint32_t* out) {
    __shared__ unsigned char t[256];
    for (int i = threadIdx.x; i < 256; i += 64) t[i] = (unsigned char)(i ^ 0x5a);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)t;
    uint32_t a = base + threadIdx.x, b = base + 64 + threadIdx.x, v = 0xdeadbeefu, w = 0xdeadbeefu;
    asm volatile("ds_read_u8_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_read_u8_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(a), "v"(b) : "memory");
    asm volatile("ds_read_u8_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tds_read_u8_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(w) : "v"(a), "v"(b) : "memory");
    out[2 * threadIdx.x] = v;
    out[2 * threadIdx.x + 1] = w;
}

int main() {
    uint32_t* d; (void)hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int keep = 0, zero = 0, other = 0;
    for (int l = 0; l < 64; ++l) {
        const uint32_t lo = (uint32_t)(l ^ 0x5a), hi = (uint32_t)((64 + l) ^ 0x5a);
        for (int j = 0; j < 2; ++j) {
            const uint32_t got = h[2 * l + j];
            if (got == (lo | hi << 16)) ++keep;
            else if (got == (j == 0 ? hi << 16 : lo)) ++zero;
            else { if (other < 4) printf("lane %d order %d: %08x\n", l, j, got); ++other; }
        }
    }
    printf("d16 byte loads: %d of 128 keep the other half, %d zero it, %d something else\n", keep, zero, other);
    return 0;
}
