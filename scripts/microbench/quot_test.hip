// Checks the range decoder's quotient estimate (scripts/gen_range_decode_loop.py quotient_lookup) against x / scale.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint64_t* xs, const uint64_t* rs, uint32_t* qs, uint32_t P, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x0 = (uint32_t)xs[i], x1 = (uint32_t)(xs[i] >> 32), r0 = (uint32_t)rs[i], r1 = (uint32_t)(rs[i] >> 32), q;
    asm volatile(
        "v_mov_b32 v144, 0\n\tv_mov_b32 v145, 0x41f00000\n\tv_mov_b32 v146, 0\n\tv_mov_b32 v147, 0x3e100000\n\t"
        "v_alignbit_b32 v124, %[r1], %[r0], %[P]\n\t"
        "v_lshrrev_b32 v125, %[P], %[r1]\n\t"
        "v_cvt_f64_u32 v[132:133], v125\n\t"
        "v_cvt_f64_u32 v[134:135], v124\n\t"
        "v_cvt_f64_u32 v[136:137], %[x1]\n\t"
        "v_cvt_f64_u32 v[138:139], %[x0]\n\t"
        "v_fma_f64 v[132:133], v[132:133], v[144:145], v[134:135]\n\t"
        "v_rcp_f64 v[140:141], v[132:133]\n\t"
        "v_fma_f64 v[136:137], v[136:137], v[144:145], v[138:139]\n\t"
        "v_fma_f64 v[148:149], -v[132:133], v[140:141], 1.0\n\t"
        "v_fma_f64 v[140:141], v[140:141], v[148:149], v[140:141]\n\t"
        "v_fma_f64 v[142:143], v[136:137], v[140:141], v[146:147]\n\t"
        "v_cvt_u32_f64 %[q], v[142:143]\n\t"
        : [q] "=v"(q) : [x0] "v"(x0), [x1] "v"(x1), [r0] "v"(r0), [r1] "v"(r1), [P] "s"(P)
        : "v124","v125","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149");
    qs[i] = q;
}
int main() {
    const int n = 1 << 20; const uint32_t P = 12;
    uint64_t *hx = new uint64_t[n], *hr = new uint64_t[n]; uint32_t* hq = new uint32_t[n];
    uint64_t h = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        h ^= h << 13; h ^= h >> 7; h ^= h << 17; uint64_t r = h | (1ull << 32);
        if (i % 3 == 0) r >>= (h % 30);
        if (r < (1ull << 32)) r |= 1ull << 32;
        h ^= h << 13; h ^= h >> 7; h ^= h << 17;
        uint64_t scale = r >> P;
        uint64_t x = (i % 5 == 0) ? scale * (h % 4096) : (uint64_t)((unsigned __int128)h * r >> 64);   // exact multiples and random x < r
        hx[i] = x; hr[i] = r;
    }
    uint64_t *dx, *dr; uint32_t* dq;
    (void)hipMalloc(&dx, 8 * n); (void)hipMalloc(&dr, 8 * n); (void)hipMalloc(&dq, 4 * n);
    (void)hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice); (void)hipMemcpy(dr, hr, 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, dq, P, n);
    (void)hipMemcpy(hq, dq, 4 * n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const uint64_t q = hx[i] / (hr[i] >> P);
        if (hq[i] != q) { if (bad < 10) printf("x=%016llx r=%016llx q=%llu got %u\n", (unsigned long long)hx[i], (unsigned long long)hr[i], (unsigned long long)q, hq[i]); ++bad; }
    }
    printf("%d of %d quotients differ\n", bad, n);
    return 0;
}
