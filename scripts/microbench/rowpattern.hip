// HBM bandwidth of the coder's access pattern: 65536 rows of 16 KiB; every wave owns 64 rows and walks them in
// steps of SEG bytes per row (SEG/16 lanes per row, 1024/SEG rows per instruction).  Read-only, write-only and copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kRows = 65536;

template <int SEG, int MODE, int kRowBytes>   // MODE 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void k(const char* __restrict__ in, char* __restrict__ out, int* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    constexpr int LPR = SEG / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;          // rows per instruction
    constexpr int NI = 64 / RPI;           // instructions per visit of all 64 rows
    const size_t row0 = wave * 64 + lane / LPR;
    const size_t off0 = row0 * kRowBytes + (lane % LPR) * 16;
    v4i acc = {0, 0, 0, 0};
    for (int t = 0; t < kRowBytes / SEG; ++t) {
        v4i v[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const size_t off = off0 + (size_t)i * RPI * kRowBytes + (size_t)t * SEG;
            if (MODE != 1) v[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(in + off));
            else v[i] = acc + i;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const size_t off = off0 + (size_t)i * RPI * kRowBytes + (size_t)t * SEG;
            if (MODE != 0) __builtin_nontemporal_store(v[i], reinterpret_cast<v4i*>(out + off));
            else acc += v[i];
        }
    }
    if (MODE == 0 && acc.x == 0x12345678) sink[0] = acc.y;
}
template <int SEG, int MODE, int kRowBytes = 16384> void run(const char* in, char* out, int* sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<SEG, MODE, kRowBytes>), dim3(kRows / 256), dim3(256), 0, 0, in, out, sink);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<SEG, MODE, kRowBytes>), dim3(kRows / 256), dim3(256), 0, 0, in, out, sink);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double gb = (double)kRows * kRowBytes * (MODE == 2 ? 2 : 1) / 1e9;
    printf("rows of %5d B, segment %4d B  %-5s: %.3f ms  %.2f TB/s\n", kRowBytes, SEG, MODE == 0 ? "read" : MODE == 1 ? "write" : "copy", ms, gb / ms);
}
int main() {
    constexpr size_t kMax = (size_t)kRows * 65536;
    char *a, *b; int* s; (void)hipMalloc(&a, kMax); (void)hipMalloc(&b, kMax); (void)hipMalloc(&s, 64);
    (void)hipMemset(a, 1, kMax);
    run<128, 0>(a, b, s); run<256, 0>(a, b, s); run<512, 0>(a, b, s); run<1024, 0>(a, b, s);
    run<128, 1>(a, b, s); run<256, 1>(a, b, s); run<512, 1>(a, b, s); run<1024, 1>(a, b, s);
    run<128, 2>(a, b, s); run<256, 2>(a, b, s); run<512, 2>(a, b, s); run<1024, 2>(a, b, s);
    // the same pattern over longer rows (2 GiB, 4 GiB)
    run<128, 0, 32768>(a, b, s); run<128, 1, 32768>(a, b, s); run<128, 0, 65536>(a, b, s); run<128, 1, 65536>(a, b, s);
    return 0;
}
