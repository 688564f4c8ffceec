/* The drop-in boundary from plain C: no Python, no torch -- the HIP runtime for device memory and the library's C ABI.
 * Encodes 4096 streams x 1024 symbols with a shared 12-bit quantized Gaussian (BASELINE config C2 in small), decodes them
 * again and compares; then the same batch as an int8 matrix through two jump points per stream (round 5).  Build (tests/test_gpu_c_example.py does exactly this):
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_roundtrip.c \
 *       -L constriction_amd/lib -lconstriction_amd -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o c_abi_roundtrip
 * What a Rust wrapper of stream::stack::AnsCoder would do through its FFI is these same calls. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "constriction_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_CST(x) do { cst_status s_ = (x); if (s_ != CST_OK) { fprintf(stderr, "%s: status %d (%s)\n", #x, (int)s_, cst_last_hip_error()); return 3; } } while (0)

int main(void) {
    const size_t n_streams = 4096, n_per = 1024, n = n_streams * n_per;
    const cst_coder_config cfg = {32, 64, 12};
    cst_model *model = NULL;
    CHECK_CST(cst_model_create_gaussian(12, -50, 50, 3.2, 9.6, NULL, &model));

    /* symbols: a crude triangular draw around the mean, inside the support */
    int32_t *h_sym = (int32_t *)malloc(n * sizeof(int32_t));
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int a = (int)(x & 31), b = (int)((x >> 8) & 31);
        int v = 3 + (a - b);
        h_sym[i] = v < -50 ? -50 : v > 50 ? 50 : v;
    }

    const size_t stride = cst_ans_max_words(n_per, cfg);
    int32_t *d_sym, *d_dec, *d_status;
    uint32_t *d_words, *d_n_words;
    CHECK_HIP(hipMalloc((void **)&d_sym, n * 4));
    CHECK_HIP(hipMalloc((void **)&d_dec, n * 4));
    CHECK_HIP(hipMalloc((void **)&d_words, n_streams * stride * 4));
    CHECK_HIP(hipMalloc((void **)&d_n_words, n_streams * 4));
    CHECK_HIP(hipMalloc((void **)&d_status, n_streams * 4));
    CHECK_HIP(hipMemcpy(d_sym, h_sym, n * 4, hipMemcpyHostToDevice));

    CHECK_CST(cst_ans_encode_batch(model, cfg, d_sym, n_streams, n_per, CST_LAYOUT_STREAM_MAJOR, d_words, stride, d_n_words, NULL,
                                   d_status, 0, NULL));
    CHECK_CST(cst_ans_decode_batch(model, cfg, d_words, NULL, stride, n_streams * stride, d_n_words, d_dec, n_streams, n_per, CST_LAYOUT_STREAM_MAJOR,
                                   NULL, NULL, d_status, 0, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    int32_t *h_dec = (int32_t *)malloc(n * 4);
    uint32_t *h_n = (uint32_t *)malloc(n_streams * 4);
    CHECK_HIP(hipMemcpy(h_dec, d_dec, n * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_n, d_n_words, n_streams * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, words = 0;
    for (size_t i = 0; i < n; ++i) bad += h_dec[i] != h_sym[i];
    for (size_t s = 0; s < n_streams; ++s) words += h_n[s];
    printf("c_abi_roundtrip: %zu streams x %zu symbols -> %zu words (%.2f bits/symbol), %zu mismatches\n", n_streams, n_per, words,
           32.0 * (double)words / (double)n, bad);

    /* The same batch as an int8 matrix (the reference's coders are generic over the symbol type) through two jump points per stream
     * -- AnsCoder::pos() noted by the encoder, AnsCoder::seek() + 512 symbols per chunk on the decoder's side (stack.rs:1107-1139):
     * rows of whole 128-symbol lines are read and written by the coder loops themselves, the words are those of the call above. */
    const size_t interval = n_per / 2, n_chunks = 2;
    int8_t *h_sym8 = (int8_t *)malloc(n), *h_dec8 = (int8_t *)malloc(n);
    for (size_t i = 0; i < n; ++i) h_sym8[i] = (int8_t)h_sym[i];
    int8_t *d_sym8, *d_dec8;
    uint32_t *d_words8, *d_n_words8, *d_pos;
    uint64_t *d_state;
    int32_t *d_cstatus;
    void *d_scratch;
    const size_t scratch_bytes = cst_ckpt_sym_scratch_bytes(n_streams, n_per, interval, 1);
    CHECK_HIP(hipMalloc((void **)&d_sym8, n));
    CHECK_HIP(hipMalloc((void **)&d_dec8, n));
    CHECK_HIP(hipMalloc((void **)&d_words8, n_streams * stride * 4));
    CHECK_HIP(hipMalloc((void **)&d_n_words8, n_streams * 4));
    CHECK_HIP(hipMalloc((void **)&d_pos, n_streams * n_chunks * 4));
    CHECK_HIP(hipMalloc((void **)&d_state, n_streams * n_chunks * 8));
    CHECK_HIP(hipMalloc((void **)&d_cstatus, n_streams * n_chunks * 4));
    CHECK_HIP(hipMalloc(&d_scratch, scratch_bytes));
    CHECK_HIP(hipMemcpy(d_sym8, h_sym8, n, hipMemcpyHostToDevice));
    CHECK_CST(cst_ans_encode_batch_ckpt_sym(model, cfg, d_sym8, 1, n_streams, n_per, CST_LAYOUT_STREAM_MAJOR, d_words8, stride, d_n_words8, interval,
                                            d_pos, d_state, d_status, d_scratch, NULL));
    printf("  int8 + jump points: encoder %s", cst_last_kernel_name());
    CHECK_CST(cst_ans_decode_batch_ckpt_sym(model, cfg, d_words8, NULL, stride, n_streams * stride, interval, d_pos, d_state, d_dec8, 1, n_streams,
                                            n_per, d_scratch, d_cstatus, NULL));
    printf(", decoder %s\n", cst_last_kernel_name());
    CHECK_HIP(hipDeviceSynchronize());
    uint32_t *h_w = (uint32_t *)malloc(n_streams * stride * 4), *h_w8 = (uint32_t *)malloc(n_streams * stride * 4);
    uint32_t *h_n8 = (uint32_t *)malloc(n_streams * 4);
    CHECK_HIP(hipMemcpy(h_dec8, d_dec8, n, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_w, d_words, n_streams * stride * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_w8, d_words8, n_streams * stride * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_n8, d_n_words8, n_streams * 4, hipMemcpyDeviceToHost));
    size_t bad8 = 0;
    for (size_t i = 0; i < n; ++i) bad8 += h_dec8[i] != h_sym8[i];
    for (size_t s = 0; s < n_streams; ++s) {
        bad8 += h_n8[s] != h_n[s];
        for (size_t w = 0; w < h_n[s] && w < stride; ++w) bad8 += h_w8[s * stride + w] != h_w[s * stride + w];
    }
    printf("  int8 + jump points: %zu mismatches (symbols, counts and words against the int32 call)\n", bad8);
    bad += bad8;

    CHECK_CST(cst_model_destroy(model));
    hipFree(d_sym); hipFree(d_dec); hipFree(d_words); hipFree(d_n_words); hipFree(d_status);
    hipFree(d_sym8); hipFree(d_dec8); hipFree(d_words8); hipFree(d_n_words8); hipFree(d_pos); hipFree(d_state); hipFree(d_cstatus); hipFree(d_scratch);
    free(h_sym); free(h_dec); free(h_n); free(h_sym8); free(h_dec8); free(h_w); free(h_w8); free(h_n8);
    return bad == 0 ? 0 : 1;
}
