// The decoder's memory traffic without the decoder: 1024 waves, every lane one stream; per 32-symbol tile a wave writes 64
// row segments of 128 B (the symbol tile: eight non-temporal 16-byte stores per lane, 16-KiB rows) and requests 1.4
// 16-byte chunks of compressed words per lane on average, walking each stream's 693 words from the end.  Two layouts of
// the words:
//   slab         stream s at s * 6208 bytes (what the coder uses: every chunk request touches 64 lines in 64 DRAM pages)
//   interleaved  group of 16 words k of the wave's 64 streams in one contiguous 4 KiB: (k >> 2) * 4096 + lane * 64 + (k & 3) * 16
// each with the words hot (the kernel repeated) and cold (1 GiB written in between).  DESIGN.md 3.8 / 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kStreams = 65536, kTiles = 128, kChunks = 174;      // 174 chunks = 696 words per stream
constexpr size_t kStride = 6208;                                   // bytes per slab (cst_ans_max_words(4096) * 4)
constexpr size_t kRowBytes = 16384;

template <bool INTERLEAVED>
__global__ __launch_bounds__(256) void k(const char* __restrict__ words, char* __restrict__ out, int* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s = wave * 64 + lane;
    const char* base = INTERLEAVED ? words + wave * 64 * kStride + (size_t)lane * 64 : words + s * kStride;
    char* orow = out + (wave * 64 + (lane >> 3)) * kRowBytes + (lane & 7) * 16;
    v4i acc = {0, 0, 0, 0};
    int chunk = kChunks - 1;
    for (int t = 0; t < kTiles; ++t) {
        const int n = (t % 5) < 2 ? 2 : 1;                         // 1.4 chunks per tile on average (lane-uniform here)
        for (int j = 0; j < n && chunk >= 0; ++j, --chunk) {
            const size_t off = INTERLEAVED ? (size_t)(chunk >> 2) * 4096 + (size_t)(chunk & 3) * 16 : (size_t)chunk * 16;
            acc += *reinterpret_cast<const v4i*>(base + off);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_nontemporal_store(acc + q, reinterpret_cast<v4i*>(orow + (size_t)q * 8 * kRowBytes + (size_t)t * 128));
    }
    if (acc.x == 0x12345678) sink[0] = acc.y;
}

template <bool INTERLEAVED> void run(const char* name, const char* words, char* out, char* flush, int* sink, bool cold) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float total = 0;
    const int reps = 8;
    hipLaunchKernelGGL((k<INTERLEAVED>), dim3(kStreams / 256), dim3(256), 0, 0, words, out, sink);
    for (int i = 0; i < reps; ++i) {
        if (cold) (void)hipMemsetAsync(flush, i, (size_t)1 << 30, 0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<INTERLEAVED>), dim3(kStreams / 256), dim3(256), 0, 0, words, out, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); total += ms;
    }
    printf("%-12s %-5s %.3f ms per launch (1 GiB of tile stores + %.0f MB of word chunks)\n", name, cold ? "cold" : "hot", total / reps,
           (double)kStreams * kChunks * 16 / 1e6);
}

int main() {
    char *words, *out, *flush; int* sink;
    (void)hipMalloc(&words, (size_t)kStreams * kStride); (void)hipMalloc(&out, (size_t)kStreams * kRowBytes);
    (void)hipMalloc(&flush, (size_t)1 << 30); (void)hipMalloc(&sink, 64);
    (void)hipMemset(words, 1, (size_t)kStreams * kStride);
    for (int cold = 0; cold < 2; ++cold) {
        run<false>("slab", words, out, flush, sink, cold);
        run<true>("interleaved", words, out, flush, sink, cold);
    }
    return 0;
}
