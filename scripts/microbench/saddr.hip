// Semantics check of the `global_* vdata, voffset, saddr` form used by the decode main loop on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(uint32_t* out, uint32_t n_iter) {
    uint32_t off = threadIdx.x * 16;
    uint32_t garbage = 0xdeadbeef;
    uint64_t base = (uint64_t)out;
    asm volatile(
        "v_mov_b32 v100, %[off]\n\t"
        "v_mov_b32 v101, %[garbage]\n\t"      // the register after the offset: must be ignored
        "v_mov_b32 v104, 1\n\tv_mov_b32 v105, 2\n\tv_mov_b32 v106, 3\n\tv_mov_b32 v107, 4\n\t"
        "s_mov_b64 s[80:81], %[base]\n\t"
        "s_mov_b32 s82, %[n]\n\t"
        "1:\n\t"
        "global_store_dwordx4 v100, v[104:107], s[80:81] nt\n\t"
        "s_add_u32 s80, s80, 0x400\n\t"
        "s_addc_u32 s81, s81, 0\n\t"
        "v_add_u32 v104, 1, v104\n\t"
        "s_sub_u32 s82, s82, 1\n\t"
        "s_cmp_lg_u32 s82, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt vmcnt(0)"
        :: [off] "v"(off), [garbage] "v"(garbage), [base] "s"(base), [n] "s"(n_iter)
        : "v100", "v101", "v104", "v105", "v106", "v107", "s80", "s81", "s82", "memory");
}
int main() {
    const uint32_t n_iter = 1000; uint32_t* d; size_t bytes = (size_t)n_iter * 1024;
    (void)hipMalloc(&d, bytes); (void)hipMemset(d, 0, bytes);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n_iter);
    hipError_t e = hipDeviceSynchronize(); printf("sync: %s\n", hipGetErrorString(e));
    std::vector<uint32_t> h(bytes / 4); (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (uint32_t it = 0; it < n_iter; ++it) for (int t = 0; t < 64; ++t) {
        const uint32_t* p = &h[(size_t)it * 256 + t * 4];
        if (p[0] != 1 + it || p[1] != 2 || p[2] != 3 || p[3] != 4) ++bad;
    }
    printf("saddr store loop: %zu bad chunks of %u\n", bad, n_iter * 64);
    return 0;
}
