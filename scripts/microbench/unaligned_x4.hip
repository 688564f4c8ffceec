// Do 16-byte global loads / stores work at addresses that are only 4-byte aligned on this gfx950 / ROCm setup?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int* in, int* out, int shift, int n4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    v4i v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(in + shift + 4 * i) : "memory");
    v += 1;
    asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(out + shift + 4 * i), "v"(v) : "memory");
}
int main() {
    const int n4 = 1 << 20, n = 4 * n4 + 8;
    std::vector<int> h(n), r(n);
    for (int i = 0; i < n; ++i) h[i] = i * 7 + 3;
    int *a, *b; (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
    for (int shift = 0; shift < 4; ++shift) {
        (void)hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemset(b, 0, n * 4);
        hipLaunchKernelGGL(k, dim3(n4 / 256), dim3(256), 0, 0, a, b, shift, n4);
        hipError_t e = hipDeviceSynchronize();
        (void)hipMemcpy(r.data(), b, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < 4 * n4; ++i) bad += r[shift + i] != h[shift + i] + 1;
        printf("shift %d words: %s, %ld wrong values\n", shift, hipGetErrorString(e), bad);
    }
    return 0;
}
