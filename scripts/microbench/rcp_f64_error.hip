// Largest relative error of gfx950's v_rcp_f64 against the correctly rounded 1 / x, over many x (the range decoder's
// quotient estimate relies on a bound).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__global__ void k(double* maxerr, uint64_t seed) {
    uint64_t h = seed + (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    double worst = 0;
    for (int i = 0; i < 4096; ++i) {
        h ^= h << 13; h ^= h >> 7; h ^= h << 17;
        // x in [2^20, 2^56): mantissa random, exponent random
        const int e = 20 + (int)((h >> 58) % 36);
        const double x = ldexp(1.0 + (double)(h & 0xFFFFFFFFFFFFFull) / 4503599627370496.0, e);
        const double r = __builtin_amdgcn_rcp(x);
        const double exact = 1.0 / x;
        const double rel = fabs(r - exact) / exact;
        worst = rel > worst ? rel : worst;
    }
    atomicMax(reinterpret_cast<unsigned long long*>(maxerr), (unsigned long long)__double_as_longlong(worst));
}
int main() {
    double* d; (void)hipMalloc(&d, 8); (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 12345ull);
    double h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("max relative error of v_rcp_f64 over 2^30 samples: %.3e = 2^%.2f\n", h, log2(h));
    return 0;
}
