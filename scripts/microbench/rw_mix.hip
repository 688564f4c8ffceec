// Reads scattered into a stream of writes: the decoder's memory traffic without a decoder.
// 65 536 "streams", one wave per 64 of them, one workgroup of four waves per CU -- the headline launch.  Per "tile" a wave
//   * stores the 128-byte piece of each of its 64 rows of a 1-GiB int32 matrix (8 x global_store_dwordx4 nt: 8 rows per
//     instruction), the symbol output of 32 decoded symbols per stream, and
//   * requests up to three 16-byte chunks per lane from that lane's slab of a words buffer (lanes draw whether they need one:
//     on average 1.4 chunks per lane and tile, what 32 symbols of 5.4 bits consume), walking the slab downwards,
//   * then sleeps for the rest of the 2 us a tile of decode steps takes (s_sleep: no issue slots, no memory).
// Nothing ever waits for a load (their sum is stored once, at the end).  Variants: stores only, loads only, both; word loads
// plain or `nt`; the words cache-resident (a warm-up pass reads what will be read) or flushed by a 1-GiB fill; slab strides.
// usage: rw_mix [stride_words ...]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kStreams = 65536, kN = 4096, kTiles = kN / 32;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define LOAD16(p) do { if (NT_LOADS) asm volatile("global_load_dwordx4 v[200:203], %0, off nt" :: "v"(p) : "memory", "v200", "v201", "v202", "v203"); \
                         else asm volatile("global_load_dwordx4 v[200:203], %0, off" :: "v"(p) : "memory", "v200", "v201", "v202", "v203"); } while (0)

// MODE 1: lane quads move 64-byte groups (lane 4 j + i: chunk i of stream 16 p + j, pass p = 0 .. 3), a stream asks for one with
//         probability 0.35 per tile;  MODE 2: lane octets move 128-byte lines (8 passes, probability 0.175);
// MODE 6: lane PAIRS move 32-byte half groups (2 passes, probability 0.7);
// MODE 3: per-lane chunks as in mode 0, but four tiles' worth every fourth tile;  MODE 4: mode 0 with plain (write-back) stores;
// MODE 5: mode 0, and the wave waits for its loads before it stores (reads and writes of a wave never overlap)
template <int MODE, bool STORES, bool NT_LOADS, int SLEEP>
__global__ __launch_bounds__(256) void kx(int* __restrict__ sym, const uint32_t* __restrict__ words, size_t stride, int* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int* row = sym + (wave * 64 + (lane >> 3)) * (size_t)kN + 4 * (lane & 7);
    constexpr int G = MODE == 1 ? 4 : (MODE == 2 ? 8 : (MODE == 6 ? 2 : 1));            // lanes per stream and pass
    constexpr int PASSES = G;
    uint32_t pos[PASSES], rng[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const size_t s = wave * 64 + (size_t)p * (64 / G) + lane / G;
        pos[p] = MODE == 1 ? 784 : (MODE == 2 ? 800 : 776);          // whole groups / lines (776 = 97 pairs of chunks)
        rng[p] = (uint32_t)s * 2654435761u + 12345u;
    }
    for (int t = 0; t < kTiles; ++t) {
        if (MODE == 1 || MODE == 2 || MODE == 6) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const size_t s = wave * 64 + (size_t)p * (64 / G) + lane / G;
                rng[p] = rng[p] * 1664525u + 1013904223u;
                const bool need = (rng[p] >> 16) % 1000u < (MODE == 1 ? 350u : (MODE == 2 ? 175u : 700u)) && pos[p] >= 4u * G;
                if (need) {
                    pos[p] -= 4 * G;
                    const v4i* q = reinterpret_cast<const v4i*>(words + s * stride + pos[p] + 4 * (lane % G));
                    LOAD16(q);
                }
            }
        } else if (MODE != 3 || (t & 3) == 0) {
            const size_t s = wave * 64 + lane;
#pragma unroll
            for (int c = 0; c < (MODE == 3 ? 12 : 3); ++c) {
                rng[0] = rng[0] * 1664525u + 1013904223u;
                const bool need = (rng[0] >> 16) % 100u < 47u && pos[0] >= 4;
                if (need) {
                    pos[0] -= 4;
                    const v4i* q = reinterpret_cast<const v4i*>(words + s * stride + pos[0]);
                    LOAD16(q);
                }
            }
        }
        if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (STORES) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                v4i v = {t, q, lane, (int)wave};
                v4i* dst = reinterpret_cast<v4i*>(row + (size_t)(8 * q) * kN + 32 * t);
                if (MODE == 4) *dst = v; else __builtin_nontemporal_store(v, dst);
            }
        }
        if (SLEEP) __builtin_amdgcn_s_sleep(64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool STORES, bool LOADS, bool NT_LOADS, int SLEEP>
__global__ __launch_bounds__(256) void k(int* __restrict__ sym, const uint32_t* __restrict__ words, size_t stride, int* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t s = wave * 64 + lane;
    // stores: lane -> row (lane >> 3) + 8 k, 16-byte column (lane & 7) of the tile's 128-byte piece
    int* row = sym + (wave * 64 + (lane >> 3)) * (size_t)kN + 4 * (lane & 7);
    const uint32_t* slab = words + s * stride;
    uint32_t pos = 776 & ~3u;                             // words of this stream still unread (5.4 bits x 4096 / 32, + state), chunk aligned
    uint32_t rng = (uint32_t)s * 2654435761u + 12345u;
    v4i acc = {0, 0, 0, 0};
    for (int t = 0; t < kTiles; ++t) {
        if (LOADS) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rng = rng * 1664525u + 1013904223u;
                const bool need = (rng >> 16) % 100u < 47u && pos >= 4;        // 3 x 0.47 = 1.4 chunks per tile
                if (need) {
                    pos -= 4;
                    // (inline asm into registers the compiler never allocates in a kernel this small: it sees no load, so it
                    // never waits for one, and a late return cannot land in a live value)
                    const v4i* p = reinterpret_cast<const v4i*>(slab + pos);
                    if (NT_LOADS) asm volatile("global_load_dwordx4 v[200:203], %0, off nt" :: "v"(p) : "memory", "v200", "v201", "v202", "v203");
                    else asm volatile("global_load_dwordx4 v[200:203], %0, off" :: "v"(p) : "memory", "v200", "v201", "v202", "v203");
                }
            }
        }
        if (STORES) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                v4i v = {t, q, lane, (int)wave};
                __builtin_nontemporal_store(v, reinterpret_cast<v4i*>(row + (size_t)(8 * q) * kN + 32 * t));
            }
        }
        if (SLEEP) {
#pragma unroll
            for (int i = 0; i < SLEEP; ++i) __builtin_amdgcn_s_sleep(64);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) sink[0] = 1;
}

__global__ void warm(const uint32_t* __restrict__ words, size_t stride, int* sink) {     // reads what the streams will read
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0;
    for (uint32_t i = 0; i < 776; i += 4) a += words[s * stride + i];
    if (a == 0x12345678u) sink[0] = 2;
}

template <bool ST, bool LD, bool NT, int SL>
float run(int* sym, const uint32_t* words, size_t stride, int* sink, char* flush, bool cold) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        if (cold) CHECK(hipMemsetAsync(flush, rep, (size_t)1 << 30, 0));
        else hipLaunchKernelGGL(warm, dim3(kStreams / 256), dim3(256), 0, 0, words, stride, sink);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<ST, LD, NT, SL>), dim3(kStreams / 256), dim3(256), 0, 0, sym, words, stride, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

// The ENCODER's traffic: every tile a wave requests the 128-byte pieces of its 64 symbol rows (8 x global_load_dwordx4 nt, never
// waited for) and stores complete groups of compressed words to the streams' slabs, walking them upwards:
// EMODE 0 no stores; 1: 64-byte groups by lane quads (4 passes, a stream has one ready with probability 0.35 per tile);
// 2: 128-byte lines by lane octets (8 passes, probability 0.175); 3: 16-byte chunks, every lane its own stream (3 x 0.47)
template <int EMODE, bool LOADS, int SLEEP>
__global__ __launch_bounds__(256) void ke(const int* __restrict__ sym, uint32_t* __restrict__ words, size_t stride, int* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int* row = sym + (wave * 64 + (lane >> 3)) * (size_t)kN + 4 * (lane & 7);
    constexpr int G = EMODE == 1 ? 4 : (EMODE == 2 ? 8 : 1);
    constexpr int PASSES = G;
    uint32_t pos[PASSES], rng[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const size_t s = wave * 64 + (size_t)p * (64 / G) + lane / G;
        pos[p] = 0; rng[p] = (uint32_t)s * 2654435761u + 12345u;
    }
    for (int t = 0; t < kTiles; ++t) {
        if (LOADS) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int* src = row + (size_t)(8 * q) * kN + 32 * (kTiles - 1 - t);
                asm volatile("global_load_dwordx4 v[200:203], %0, off nt" :: "v"(src) : "memory", "v200", "v201", "v202", "v203");
            }
        }
        if (EMODE) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const size_t s = wave * 64 + (size_t)p * (64 / G) + lane / G;
#pragma unroll
                for (int c = 0; c < (EMODE == 3 ? 3 : 1); ++c) {
                    rng[p] = rng[p] * 1664525u + 1013904223u;
                    const uint32_t thr = EMODE == 1 ? 350u : (EMODE == 2 ? 175u : 470u);
                    if ((rng[p] >> 16) % 1000u < thr && pos[p] + 4u * G <= 800u) {
                        v4i v = {t, p, lane, (int)wave};
                        *reinterpret_cast<v4i*>(words + s * stride + pos[p] + 4 * (lane % G)) = v;
                        pos[p] += 4 * G;
                    }
                }
            }
        }
        if (SLEEP) __builtin_amdgcn_s_sleep(64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EMODE, bool LOADS, int SL>
float rune(const int* sym, uint32_t* words, size_t stride, int* sink, char* flush) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((ke<EMODE, LOADS, SL>), dim3(kStreams / 256), dim3(256), 0, 0, sym, words, stride, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

template <int MODE, bool NT, int SL>
float runx(int* sym, const uint32_t* words, size_t stride, int* sink, char* flush, bool cold) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        if (cold) CHECK(hipMemsetAsync(flush, rep, (size_t)1 << 30, 0));
        else hipLaunchKernelGGL(warm, dim3(kStreams / 256), dim3(256), 0, 0, words, stride, sink);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((kx<MODE, true, NT, SL>), dim3(kStreams / 256), dim3(256), 0, 0, sym, words, stride, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    std::vector<size_t> strides;
    for (int i = 1; i < argc; ++i) strides.push_back((size_t)atoll(argv[i]));
    if (strides.empty()) strides = {1552, 1648, 2048};
    int *sym, *sink; char* flush; uint32_t* words;
    CHECK(hipMalloc(&sym, (size_t)kStreams * kN * 4)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&flush, (size_t)1 << 30));
    CHECK(hipMalloc(&words, (size_t)kStreams * 2048 * 4 + 65536));
    CHECK(hipMemset(words, 1, (size_t)kStreams * 2048 * 4));
    constexpr int SL = 1;            // one s_sleep 64 per tile: ~4100 clocks, the pace of a tile of decode steps (the "pace only" line)
    printf("per launch, best of 5, ms      (a tile's pace without memory: see 'pace only')\n");
    printf("pace only (no loads, no stores): %.3f\n", run<false, false, false, SL>(sym, words, 1552, sink, flush, false));
    for (size_t st : strides) {
        if (st > 2048) continue;
        for (int cold = 0; cold < 2; ++cold) {
            const float w = run<true, false, false, SL>(sym, words, st, sink, flush, cold);
            const float r = run<false, true, false, SL>(sym, words, st, sink, flush, cold);
            const float b = run<true, true, false, SL>(sym, words, st, sink, flush, cold);
            const float bn = run<true, true, true, SL>(sym, words, st, sink, flush, cold);
            const float w0 = run<true, false, false, 0>(sym, words, st, sink, flush, cold);
            const float b0 = run<true, true, false, 0>(sym, words, st, sink, flush, cold);
            printf("stride %4zu words, words %s: stores only %.3f  loads only %.3f  both %.3f  both, nt loads %.3f   |  unpaced: stores only %.3f  both %.3f\n",
                   st, cold ? "flushed " : "in cache", w, r, b, bn, w0, b0);
            printf("      stores + loads another way:  per-lane chunks again %.3f  lane quads (64 B) %.3f  lane octets (128 B) %.3f  four tiles' chunks every fourth tile %.3f"
                   "  plain stores %.3f  loads waited for before the stores %.3f  lane pairs (32 B) %.3f\n",
                   runx<0, false, SL>(sym, words, st, sink, flush, cold), runx<1, false, SL>(sym, words, st, sink, flush, cold),
                   runx<2, false, SL>(sym, words, st, sink, flush, cold), runx<3, false, SL>(sym, words, st, sink, flush, cold),
                   runx<4, false, SL>(sym, words, st, sink, flush, cold), runx<5, false, SL>(sym, words, st, sink, flush, cold),
                   runx<6, false, SL>(sym, words, st, sink, flush, cold));
        }
    }
    printf("\nthe ENCODER's traffic (1 GiB of symbol rows read 128 bytes at a time, nt; word groups stored to the slabs), back-to-back launches:\n");
    for (size_t st : strides) {
        if (st > 2048) continue;
        printf("stride %4zu words: loads only %.3f | stores only: 16-byte chunks %.3f  64-byte quads %.3f  128-byte octets %.3f | loads + 16-byte chunks %.3f  + 64-byte quads %.3f  + 128-byte octets %.3f"
               "   | unpaced: loads only %.3f  + quads %.3f  + octets %.3f\n", st,
               rune<0, true, SL>(sym, words, st, sink, flush),
               rune<3, false, SL>(sym, words, st, sink, flush), rune<1, false, SL>(sym, words, st, sink, flush), rune<2, false, SL>(sym, words, st, sink, flush),
               rune<3, true, SL>(sym, words, st, sink, flush), rune<1, true, SL>(sym, words, st, sink, flush), rune<2, true, SL>(sym, words, st, sink, flush),
               rune<0, true, 0>(sym, words, st, sink, flush), rune<1, true, 0>(sym, words, st, sink, flush), rune<2, true, 0>(sym, words, st, sink, flush));
    }
    return 0;
}
