/*
 * constriction_amd.h -- C ABI of the MI355X-native batched entropy-coding backend.
 *
 * This is the drop-in boundary for constriction's stream-coder hot path.
 * The reference has NO FFI surface for this path (only PyO3 bindings); each entry point below
 * names the reference interface it replaces (paths relative to the reference checkout).  A Rust
 * `extern "C"` block / ctypes stub binding exactly these symbols is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/torch types.  `stream` is a hipStream_t passed as void*
 *    (NULL = the default stream).  All `d_*` pointers are DEVICE pointers (HBM), all `h_*`
 *    pointers are host pointers.  The library never frees caller memory.
 *  - all batched calls are asynchronous on `stream`; results are valid after the caller
 *    synchronises that stream.  Functions are re-entrant; no coder call writes global state (the only process-wide data are
 *    the debug switches below: read once when the library is loaded).
 *  - every function returns a cst_status (0 = ok, negative = call-level error).  Per-stream
 *    outcomes go to the caller's `d_status` array (cst_stream_status), mirroring the
 *    reference's per-coder Result values (src/lib.rs:313-316, 376-385).
 *  - coder presets are given as (word_bits, state_bits, precision) = (W, S, P):
 *      (32,64,24) DefaultAnsCoder + the Python API        src/stream/stack.rs:139
 *      (32,64,12) benches/lookup.rs:32-34, BASELINE config C2/C3
 *      (16,32,12) SmallAnsCoder                            src/stream/stack.rs:153
 *    Supported: W=32,S=64,1<=P<=24 and W=16,S=32,1<=P<=16 (hand-scheduled kernels, every entry point), and -- through
 *    cst_ans_encode_batch / cst_ans_decode_batch with table models -- the rest of the reference's type grid
 *    (src/stream/stack.rs:1293-1356): W=32,S=64,24<P<=32; W=16,S=64,P<=16; W=8 with S=64, 32 or 16, P<=8.  Compressed
 *    words are always stored one per uint32_t slot (narrower words occupy the low bits).
 */
#ifndef CONSTRICTION_AMD_H
#define CONSTRICTION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CST_ABI_VERSION 5

typedef enum cst_status {
    CST_OK = 0,
    CST_ERR_INVALID_ARGUMENT = -1, /* bad sizes / NULL pointers / unsupported (W,S,P) */
    CST_ERR_HIP = -2,              /* a HIP runtime call failed (see cst_last_hip_error) */
    CST_ERR_NO_DEVICE = -3,        /* no gfx950 device visible: the product path has NO CPU fallback */
    CST_ERR_MODEL = -4,            /* model cannot be built (support too large, sigma<=0, zero probability) */
    CST_ERR_OUT_OF_MEMORY = -5
} cst_status;

/* per-stream result codes written to d_status[stream] */
typedef enum cst_stream_status {
    CST_STREAM_OK = 0,
    CST_STREAM_IMPOSSIBLE_SYMBOL = 1, /* DefaultEncoderFrontendError::ImpossibleSymbol, src/lib.rs:376-385
                                         (Python: KeyError, src/pybindings/stream/mod.rs:82-89) */
    CST_STREAM_CAPACITY = 2,          /* output slab too small (reference: backend WriteError) */
    CST_STREAM_INVALID_DATA = 3,      /* trailing zero word (src/stream/stack.rs:299-318) or
                                         range-decoder InvalidData (src/stream/queue.rs:989-993) */
    CST_STREAM_OUT_OF_DATA = 4        /* chain coder: DecoderFrontendError::OutOfCompressedData /
                                         EncoderFrontendError::OutOfRemainders (src/stream/chain.rs:854-890) */
} cst_stream_status;

/* memory layout of the int32 symbol matrix */
typedef enum cst_layout {
    CST_LAYOUT_STREAM_MAJOR = 0, /* symbols[stream][t]  (BASELINE config C2) */
    CST_LAYOUT_SYMBOL_MAJOR = 1  /* symbols[t][stream]  (lane-coalesced without LDS staging) */
} cst_layout;

typedef struct cst_coder_config {
    int32_t word_bits;  /* W */
    int32_t state_bits; /* S */
    int32_t precision;  /* P */
} cst_coder_config;

/* flags for the batched coder calls */
#define CST_FLAG_NONE 0u
/* encode: do not append the final state words (leave the state in d_state);
 * decode: do not read the initial state from the end of each stream's words (take it from d_state).
 * Used by the single-coder drop-in object to continue an existing coder
 * (AnsCoder::encode_symbols_reverse on a non-empty coder, src/stream/stack.rs:784-849). */
#define CST_FLAG_RAW_STATE 1u
/* decode (a hint; results do not depend on it): the compressed words are NOT expected in the GPU's caches -- they arrived by
 * DMA from the host or a peer, or were written long ago -- as opposed to words an encode call has just left there.  The
 * (32,64), P <= 12 decoder then reads them as whole 64-byte segments moved by lane quads (cst_ans_dq.hip): 11 - 17 % faster on
 * words that come from HBM, 7 % slower on cache-resident ones at a well-chosen slab stride.  Other kernels
 * ignore it. */
#define CST_FLAG_COLD_WORDS 2u
/* ABI 4, the (16,32) preset only (SmallAnsCoder, src/stream/stack.rs:153: the reference holds its words in a Vec<u16>): the
 * compressed words are PACKED, two per uint32 slot -- d_words is an array of uint16_t (little endian, exactly the bytes of the
 * reference's Vec<u16>), and stride_words, d_n_words, d_offsets, words_capacity all count 16-bit words.  Without the flag every
 * word of this preset occupies a uint32 slot (low half), which doubles the word bytes the kernels move.  cst_ans_encode_batch
 * and cst_ans_decode_batch take it for shared-table models, stream-major symbols, 8 <= precision <= 12 (anything else:
 * CST_ERR_INVALID_ARGUMENT); the slabs pack with cst_compact_words16.  The fast path wants 64-byte aligned slabs
 * (stride_words a multiple of 32: cst_ans_max_words rounds to that). */
#define CST_FLAG_PACKED_W16 4u

/* ------------------------------------------------------------------------------------------
 * library / device
 * ---------------------------------------------------------------------------------------- */

/* ABI version of the loaded library (== CST_ABI_VERSION of the header it was built from). */
int32_t cst_abi_version(void);

/* Number of visible gfx950 devices, or a negative cst_status. */
int32_t cst_device_count(void);

/* Text of the most recent HIP error seen by the calling thread ("" if none). */
const char *cst_last_hip_error(void);

/* ABI 4, diagnostics: the kernel family the calling thread's last batched coder call (cst_ans_* / cst_range_* _batch and their
 * _ckpt forms) launched -- "ans_encode_pc_kernel", "ans_decode_dq_kernel", "range_decode_sub_kernel", ... ; "" before any.
 * The dispatcher picks by shape, alignment and flags; a profile or a benchmark that labels its numbers asks here. */
const char *cst_last_kernel_name(void);

/* Upper bound on the words one stream can produce, min(n, ceil(n*P/W)) + S/W, rounded up to a whole number of
 * 64-byte units so that slabs laid out at this stride from a 64-byte aligned base are all 64-byte aligned (the
 * encoder then writes whole aligned 64-byte groups).  Any other stride remains legal.
 * (At most one word per symbol: src/stream/stack.rs:1035-1040; final state: stack.rs:891-895.) */
size_t cst_ans_max_words(size_t n_symbols, cst_coder_config cfg);

/* Same bound for the range coder: one word per symbol (queue.rs:671-702) + seal words (queue.rs:498-522), rounded
 * up to 64-byte units in the same way (the hand-scheduled encoder needs 64-byte aligned slabs). */
size_t cst_range_max_words(size_t n_symbols, cst_coder_config cfg);

/* ------------------------------------------------------------------------------------------
 * entropy models (device-resident cumulative-frequency tables)
 *
 * A cst_model is the device image of one of the reference's entropy models restricted to a
 * contiguous i32 support [min_symbol, min_symbol + n_symbols):
 *   encode side  = EncoderModel::left_cumulative_and_probability as a table
 *                  (ContiguousCategoricalEntropyModel, src/stream/model/categorical/contiguous.rs:673-700)
 *   decode side  = DecoderModel::quantile_function as ContiguousLookupDecoderModel
 *                  (src/stream/model/categorical/lookup_contiguous.rs:169-187, 564-605)
 * cdf[0] = 0 < cdf[1] < ... < cdf[n] = 2^P.
 * ---------------------------------------------------------------------------------------- */

typedef struct cst_model cst_model;

/* One table shared by all streams, from a host cdf[n_symbols+1] (any tabulated model: the
 * "fast" categorical tables of src/stream/model/categorical.rs:16-54, a LeakyQuantizer table, ...). */
cst_status cst_model_create_table(int32_t precision, int32_t min_symbol, int32_t n_symbols,
                                  const uint32_t *h_cdf, cst_model **out);

/* One shared table = LeakyQuantizer<f64,i32,u{prob_bits},P>(min..=max) x Gaussian(mean,std),
 * i.e. constriction.stream.model.QuantizedGaussian(min, max, mean, std)
 * (src/stream/model/quantize.rs:284-308, 525-568; src/pybindings/stream/model.rs:649-660).
 * The table is computed ON DEVICE in bit-exact f64.  prob_bits = 32 for W=32, 16 for W=16. */
cst_status cst_model_create_gaussian(int32_t precision, int32_t min_symbol, int32_t max_symbol,
                                     double mean, double std, void *stream, cst_model **out);

/* One table PER STREAM (BASELINE config C3): stream s uses Gaussian(d_means[s], d_stds[s]).
 * d_means/d_stds are device arrays of n_streams doubles.  Per-stream tables are coded from LDS: precision <= 16 and
 * a support of at most 1023 symbols (larger ones make the coding calls return CST_ERR_INVALID_ARGUMENT; shared
 * tables have no such limit). */
cst_status cst_model_create_gaussian_per_stream(int32_t precision, int32_t min_symbol, int32_t max_symbol,
                                                const double *d_means, const double *d_stds,
                                                size_t n_streams, void *stream, cst_model **out);

/* A tabulated model over an ARBITRARY alphabet of distinct i32 symbols (NonContiguousCategoricalEncoderModel /
 * NonContiguousLookupDecoderModel, src/stream/model/categorical/{non_contiguous,lookup_noncontiguous}.rs:429-470, 602-646):
 * h_symbols[i] is the symbol with left cumulative h_cdf[i].  The model proper works on the indices 0..n-1; the two
 * kernels below translate (in place if wanted): encode = cst_symbols_to_indices + cst_ans_encode_batch (a symbol that is
 * not in the alphabet becomes index n, which the coder reports as CST_STREAM_IMPOSSIBLE_SYMBOL), decode =
 * cst_ans_decode_batch + cst_indices_to_symbols. */
cst_status cst_model_create_table_noncontiguous(int32_t precision, int32_t n_symbols, const int32_t *h_symbols,
                                               const uint32_t *h_cdf, cst_model **out);
cst_status cst_symbols_to_indices(const cst_model *model, const int32_t *d_symbols, size_t count, int32_t *d_indices,
                                  void *stream);
cst_status cst_indices_to_symbols(const cst_model *model, const int32_t *d_indices, size_t count, int32_t *d_symbols,
                                  void *stream);

cst_status cst_model_destroy(cst_model *model);

/* Introspection (tests, and get_cdf for host-side tooling). */
int32_t cst_model_precision(const cst_model *model);
int32_t cst_model_min_symbol(const cst_model *model);
int32_t cst_model_n_symbols(const cst_model *model);
size_t cst_model_n_tables(const cst_model *model); /* 1 = shared, else n_streams */
/* Copies table `index`'s cdf[n_symbols+1] to host (synchronises `stream`). */
cst_status cst_model_get_cdf(const cst_model *model, size_t index, uint32_t *h_cdf, void *stream);
/* Copies the cdfs of tables [first, first + count) -- (n_symbols + 1) entries each, back to back -- into DEVICE memory,
 * asynchronously on `stream` (e.g. to draw test symbols from per-stream models without a host round trip). */
cst_status cst_model_copy_cdfs(const cst_model *model, size_t first, size_t count, uint32_t *d_cdfs, void *stream);

/* ------------------------------------------------------------------------------------------
 * model families other than the quantized Gaussian
 * ---------------------------------------------------------------------------------------- */

typedef enum cst_family {
    CST_FAMILY_LAPLACE = 1, /* constriction.stream.model.QuantizedLaplace(min, max, mean, scale), src/pybindings/stream/model.rs:736-800 */
    CST_FAMILY_CAUCHY = 2,  /* QuantizedCauchy(min, max, loc, scale), model.rs:836-900 */
    CST_FAMILY_BINOMIAL = 3 /* Binomial(n, p) over 0..=n, model.rs:903-966 */
} cst_family;

/* LeakyQuantizer<f64,i32,u32,P>(min..=max) x family(a[row], b[row]) tabulated on the device, one row of
 * (max - min + 2) left cumulatives per parameter pair (src/stream/model/quantize.rs:284-308, 525-568 over the
 * `probability` crate's Laplace / Cauchy / Binomial CDFs, evaluated with the libm-crate algorithms in bit-exact f64).
 * Laplace: a = mean, b = scale.  Cauchy: a = loc, b = scale.  Binomial: a = p, d_b unused (may be NULL), min_symbol
 * must be 0 and max_symbol = n; with d_n_per_row != NULL row r is the model over 0..=d_n_per_row[r] (<= max_symbol)
 * and entries past its own 2^P repeat 2^P (the family form `Binomial()` with per-symbol n).  d_bad (optional, one
 * int32 per row) is set to 1 where a row is not strictly increasing, i.e. where the reference panics
 * (quantize.rs:560-566).  The rows feed cst_model_create_table / the *_rows_batch / *_cp_batch entry points. */
cst_status cst_family_cdf_rows(int32_t family, int32_t precision, int32_t min_symbol, int32_t max_symbol,
                               const double *d_a, const double *d_b, const int32_t *d_n_per_row, size_t n_rows,
                               uint32_t *d_rows, int32_t *d_bad, void *stream);

/* Categorical(probabilities, perfect=True): `perfectly_quantized_probabilities` + cumulation
 * (src/stream/model/categorical.rs:56-177, contiguous.rs:301-313) for Probability = u32.  HOST function (a sequential
 * greedy search); h_probs are f64 (f32 inputs widened by the caller, as `F: Into<f64>` does); writes h_cdf[n + 1].
 * CST_ERR_MODEL where the reference returns Err (n < 2, negative / non-normalisable probabilities). */
cst_status cst_categorical_perfect_cdf(const double *h_probs, size_t n, int32_t precision, uint32_t *h_cdf);

/* test hooks: the elementary functions behind the two calls above as the device evaluates them
 * (which: 0 log, 1 log1p, 2 atan, 3 lgamma for x > 0, 4 exp), and the host's log1p */
cst_status cst_debug_family_fn(int32_t which, const double *d_x, double *d_out, size_t n, void *stream);
double cst_debug_host_log1p(double x);

/* ------------------------------------------------------------------------------------------
 * batched ANS coder: one independent AnsCoder<W,S> per stream
 * ---------------------------------------------------------------------------------------- */

/* Replaces, for every stream s in [0, n_streams):
 *     let mut coder = AnsCoder::new();                             src/stream/stack.rs:249
 *     coder.encode_iid_symbols_reverse(symbols[s], &model)?;       src/stream/stack.rs:835-849
 *     out[s] = coder.into_compressed();                            src/stream/stack.rs:891-895
 * (Python: AnsCoder().encode_reverse(symbols, model); get_compressed(),
 *  src/pybindings/stream/stack.rs:529-591, 411-429.)
 *
 * d_symbols   int32 [n_streams][n_per_stream] (or transposed, see layout)
 * d_words     uint32 slabs: stream s writes d_words[s*stride_words ...], in emission order
 * d_n_words   out: words written per stream
 * d_state     uint64 [n_streams] in/out, only with CST_FLAG_RAW_STATE (else may be NULL)
 * d_status    out: cst_stream_status per stream.  On IMPOSSIBLE_SYMBOL / CAPACITY the stream's
 *             n_words is 0 and its slab content is unspecified.
 */
cst_status cst_ans_encode_batch(const cst_model *model, cst_coder_config cfg, const int32_t *d_symbols,
                                size_t n_streams, size_t n_per_stream, cst_layout layout,
                                uint32_t *d_words, size_t stride_words, uint32_t *d_n_words,
                                uint64_t *d_state, int32_t *d_status, uint32_t flags, void *stream);

/* Replaces, for every stream s:
 *     let mut coder = AnsCoder::from_compressed(words[s])?;        src/stream/stack.rs:299-318, 440-462
 *     symbols[s] = coder.decode_iid_symbols(n_per_stream, &model)  src/stream/mod.rs:1016-1031, stack.rs:1070-1100
 * (Python: AnsCoder(compressed).decode(model, n), src/pybindings/stream/stack.rs:217-241, 688-752.)
 *
 * The words of stream s are d_words[off(s) .. off(s) + d_n_words[s]) with
 *     off(s) = d_offsets ? d_offsets[s] : s * stride_words
 * so both the slab layout written by cst_ans_encode_batch and the packed layout written by
 * cst_compact_words decode without a copy.  Decoding past the end of a stream is legal and
 * deterministic, exactly as in the reference (stack.rs:1062-1065).
 * Memory safety on corrupt metadata (the reference's decoder pops from a Vec and cannot leave it, src/backends.rs:495-507):
 * `words_capacity` = the number of uint32 slots behind d_words.  A stream whose slice [off(s), off(s) + d_n_words[s])
 * leaves the buffer -- or, in slab form, whose d_n_words[s] exceeds stride_words (checked always) -- is decoded as an
 * EMPTY stream and reports CST_STREAM_INVALID_DATA; nothing outside the buffer is read.  words_capacity = 0 means
 * "unknown": the caller vouches for the packed offsets as in ABI 2.  (The kernels read whole aligned 16-byte chunks:
 * up to 12 bytes before the first and after the last word of a stream are touched, never interpreted; with
 * CST_FLAG_COLD_WORDS whole aligned 64-byte groups: up to 60 bytes either side, inside the allocation that holds d_words
 * -- hipMalloc aligns and pads allocations to 256 bytes -- and that decoder is only taken when the span of the words is known,
 * i.e. never for packed offsets with words_capacity = 0.  A capacity that is the true size of a hipMalloc'ed buffer
 * satisfies both.)  Every decode entry point below takes the same argument.
 * The model must have been created on the current device (CST_ERR_INVALID_ARGUMENT otherwise).
 * With CST_FLAG_RAW_STATE the initial state comes from d_state and the remaining state and word
 * count are written back to d_state / d_n_words_out (d_n_words_out may alias nothing; NULL = discard).
 */
cst_status cst_ans_decode_batch(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                const uint64_t *d_offsets, size_t stride_words, size_t words_capacity, const uint32_t *d_n_words,
                                int32_t *d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                uint64_t *d_state, uint32_t *d_n_words_out, int32_t *d_status,
                                uint32_t flags, void *stream);

/* ABI 4: NARROW symbol matrices.  The reference's coders are generic over the symbol type (`Symbol: PrimInt + ...`,
 * src/stream/model/quantize.rs:229-255); the kernels here code int32 matrices.  What a narrow matrix saves is the LINK: a batch
 * that comes from and returns to host memory is bound by PCIe, and int8 symbols are a quarter of its bytes.  symbol_bytes = 1, 2
 * (signed two's complement) or 4; the narrow types are widened / narrowed on the device next to the coder call through
 * d_scratch (cst_symbols_scratch_bytes(...) bytes, 0 for symbol_bytes = 4; contents irrelevant).  Words, counts and status are
 * those of the int32 calls on the widened values; an encoder symbol outside the model's support is an impossible symbol as ever,
 * a decoder whose model's support does not fit the type returns CST_ERR_INVALID_ARGUMENT.  The two conversions are exported on
 * their own for the other coders (range, per-symbol, checkpointed calls take int32).
 * NARROW MATRICES INSIDE THE LOOPS (round 5): symbol_bytes = 1 or 2 with the default preset (32,64), 8 <= P <= 24, a shared table,
 * stream-major rows that are whole 128-BYTE lines (128 int8 / 64 int16 symbols) of a 128-byte aligned matrix, any number of streams
 * (encode: 64-byte aligned slabs with stride_words % 16 == 0, at most 256 (int8) / 1024 (int16) symbols; decode: a known span of
 * the words) are coded by kernels that read / write the narrow matrix themselves -- no conversion, d_scratch is not touched and
 * may be NULL; the same words, counts, status (cst_last_kernel_name: "ans_encode_pc_n8_kernel" / "ans_decode_n8_kernel" /
 * "ans_decode_small_n8_kernel" and their n16 forms; 12 < P <= 24: "ans_encode_pc_n8_kernel<wide>" / "ans_decode_b16_n8_kernel").
 * Every other shape takes the conversion. */
cst_status cst_symbols_widen(const void *d_in, int32_t symbol_bytes, size_t n, int32_t *d_out, void *stream);
cst_status cst_symbols_narrow(const int32_t *d_in, size_t n, void *d_out, int32_t symbol_bytes, void *stream);
size_t cst_symbols_scratch_bytes(size_t n_streams, size_t n_per_stream, int32_t symbol_bytes);
cst_status cst_ans_encode_batch_sym(const cst_model *model, cst_coder_config cfg, const void *d_symbols, int32_t symbol_bytes,
                                    size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                    size_t stride_words, uint32_t *d_n_words, uint64_t *d_state, int32_t *d_status,
                                    uint32_t flags, void *d_scratch, void *stream);
cst_status cst_ans_decode_batch_sym(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                    const uint64_t *d_offsets, size_t stride_words, size_t words_capacity,
                                    const uint32_t *d_n_words, void *d_symbols, int32_t symbol_bytes, size_t n_streams,
                                    size_t n_per_stream, cst_layout layout, uint64_t *d_state, uint32_t *d_n_words_out,
                                    int32_t *d_status, uint32_t flags, void *d_scratch, void *stream);

/* Streams of DIFFERENT lengths -- thousands of small coders with a shared model in one launch: the reference's "compressed
 * index" pattern (tests/issue52.rs:27-60, 63-80: one DefaultAnsCoder per document, `encode_symbol` per character last to
 * first, `into_compressed`; `from_compressed` + `decode_symbol` per document), which costs one device round trip per document
 * through the single-coder binding.
 *     symbols of stream s = d_symbols[d_sym_offsets[s] .. d_sym_offsets[s + 1])      (d_sym_offsets: uint64 [n_streams + 1])
 *     slab of stream s    = d_words[d_word_offsets[s] .. d_word_offsets[s + 1])      (uint64 [n_streams + 1]; a slab of
 *                           cst_ans_max_words(length of s, cfg) words always suffices), or, with d_word_offsets = NULL,
 *                           d_words[s * stride_words .. + stride_words)
 * Every stream's words, count and status are those of cst_ans_encode_batch / the reference coder for that stream alone; the
 * decoder takes the same offsets (only d_word_offsets[s] and d_n_words[s] are read) or a packed layout, with the bounds
 * check of cst_ans_decode_batch.  Shared-table models, stream-major symbols, any preset.  A wave of 64 consecutive streams
 * runs as long as its longest one. */
cst_status cst_ans_encode_ragged(const cst_model *model, cst_coder_config cfg, const int32_t *d_symbols,
                                 const uint64_t *d_sym_offsets, size_t n_streams, uint32_t *d_words,
                                 const uint64_t *d_word_offsets, size_t stride_words, uint32_t *d_n_words,
                                 int32_t *d_status, void *stream);
cst_status cst_ans_decode_ragged(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                 const uint64_t *d_word_offsets, size_t stride_words, size_t words_capacity,
                                 const uint32_t *d_n_words, int32_t *d_symbols, const uint64_t *d_sym_offsets,
                                 size_t n_streams, int32_t *d_status, void *stream);
/* The reference's index stores no lengths: a document ends where its terminator symbol is decoded
 * (tests/issue52.rs:63-80, `core::iter::from_fn(|| { let id = coder.decode_symbol(..); alphabet.get(id) })`).  This is
 * the first pass of that: every stream is decoded until `eof_symbol` appears, nothing is stored but the number of
 * symbols decoded, terminator included (d_lengths, uint64 [n_streams]); a stream without a terminator among its first
 * `max_symbols` symbols reports CST_STREAM_CAPACITY and max_symbols.  An exclusive prefix sum of d_lengths is the
 * d_sym_offsets of cst_ans_decode_ragged. */
cst_status cst_ans_count_until(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                               const uint64_t *d_word_offsets, size_t stride_words, size_t words_capacity,
                               const uint32_t *d_n_words, size_t n_streams, int32_t eof_symbol, size_t max_symbols,
                               uint64_t *d_lengths, int32_t *d_status, void *stream);

/* ABI 5: jump points for ragged batches.  A launch of many small coders lasts as long as its LONGEST document's chain (100 000
 * documents of 20 .. 2000 symbols decode in 0.61 ms however few they are: 2000 dependent steps).  The reference's own remedy is the jump
 * table of Pos / Seek (src/stream/stack.rs:1107-1139): the encoder notes AnsCoder::pos() -- (words in the bulk, coder state) -- in front
 * of every chunk of `jump_interval` symbols of every stream (a multiple of 8) on its way, the words are those of cst_ans_encode_ragged,
 * and the decoder runs every chunk as a coder of its own (AnsCoder::seek + at most jump_interval symbols): the longest chain is
 * jump_interval steps.  Chunk j of stream s is entry d_chunk_offsets[s] + j of d_jump_pos / d_jump_state, with
 * d_chunk_offsets[n_streams + 1] the exclusive prefix sum of ceil(length / jump_interval) (n_chunks_total = its last entry).  The
 * decoder writes one status per STREAM (the worst of its chunks'; a table that does not describe its stream, or a jump point with more
 * words than the stream has, reports CST_STREAM_INVALID_DATA).  d_scratch: cst_ragged_jump_scratch_bytes(n_chunks_total) bytes. */
size_t cst_ragged_jump_scratch_bytes(size_t n_chunks_total);
cst_status cst_ans_encode_ragged_jump(const cst_model *model, cst_coder_config cfg, const int32_t *d_symbols,
                                      const uint64_t *d_sym_offsets, size_t n_streams, const uint32_t *d_order, uint32_t *d_words,
                                      const uint64_t *d_word_offsets, size_t stride_words, uint32_t *d_n_words,
                                      size_t jump_interval, const uint64_t *d_chunk_offsets, uint32_t *d_jump_pos,
                                      uint64_t *d_jump_state, int32_t *d_status, void *stream);
cst_status cst_ans_decode_ragged_jump(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                      const uint64_t *d_word_offsets, size_t stride_words, size_t words_capacity,
                                      const uint32_t *d_n_words, int32_t *d_symbols, const uint64_t *d_sym_offsets,
                                      size_t n_streams, size_t jump_interval, const uint64_t *d_chunk_offsets,
                                      size_t n_chunks_total, const uint32_t *d_jump_pos, const uint64_t *d_jump_state,
                                      void *d_scratch, int32_t *d_status, void *stream);

/* The same three calls with a SCHEDULE: lane slot i of the launch codes stream d_order[i] (uint32 [n_streams], a permutation of
 * 0 .. n_streams - 1; NULL = the identity, i.e. the calls above).  A wave of 64 slots runs as long as its longest stream, so a
 * batch whose lengths differ by orders of magnitude should put streams of similar length side by side, longest first:
 * d_order = the stream indices sorted by length (or, for the decoders, by d_n_words) in descending order.  Results (words,
 * counts, symbols, status, all indexed by STREAM as above) do not depend on the order; an entry that is not a stream index
 * leaves its slot idle, a stream that no entry names is not coded.  1 000 000 documents of 20 .. 2000 symbols: encode
 * 2.0 -> 1.3 ms, decode 4.9 -> 1.4 ms (scripts/bench_ragged_big.py). */
cst_status cst_ans_encode_ragged_ordered(const cst_model *model, cst_coder_config cfg, const int32_t *d_symbols,
                                         const uint64_t *d_sym_offsets, size_t n_streams, const uint32_t *d_order,
                                         uint32_t *d_words, const uint64_t *d_word_offsets, size_t stride_words,
                                         uint32_t *d_n_words, int32_t *d_status, void *stream);
cst_status cst_ans_decode_ragged_ordered(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                         const uint64_t *d_word_offsets, size_t stride_words, size_t words_capacity,
                                         const uint32_t *d_n_words, int32_t *d_symbols, const uint64_t *d_sym_offsets,
                                         size_t n_streams, const uint32_t *d_order, int32_t *d_status, void *stream);
cst_status cst_ans_count_until_ordered(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                       const uint64_t *d_word_offsets, size_t stride_words, size_t words_capacity,
                                       const uint32_t *d_n_words, size_t n_streams, const uint32_t *d_order,
                                       int32_t eof_symbol, size_t max_symbols, uint64_t *d_lengths, int32_t *d_status,
                                       void *stream);

/* Checkpointed streams -- the reference's Pos / Seek jump tables (src/stream/stack.rs:1107-1139; test :1456-1548) for the
 * batched coder.  The encoder notes, in front of every chunk of `ckpt_interval` symbols, what `AnsCoder::pos()` returns
 * there: d_ckpt_pos[s][j] = words in the bulk, d_ckpt_state[s][j] = coder state once symbols [j * interval, n) are
 * encoded (n_chunks = ceil(n_per_stream / interval) entries per stream).  The compressed words are exactly those of
 * cst_ans_encode_batch.  The decoder then treats every (stream, chunk) as an independent coder --
 * `AnsCoder::seek(pos, state)` + interval decoded symbols -- so that ONE long stream (BASELINE config C1) spreads over
 * n_chunks lanes: it is the ordinary batched decode of n_streams * n_chunks virtual streams (stream-major symbols,
 * shared-table models, n_per_stream a multiple of the interval; d_status has n_streams * n_chunks entries;
 * d_scratch: cst_ckpt_scratch_bytes(...) bytes, contents irrelevant).
 * ABI 4: jump points as SUB-LANES of a batch.  Many streams gain from them too: a decoder of 65 536 streams runs one wave per
 * SIMD and waits for its table lookups; with k = n_per_stream / interval jump points per stream the same words decode on k
 * lanes per stream, two waves per SIMD.  Models with one table per stream (config C3; stream-major, the compact-row shapes of
 * cst_model_create_gaussian_per_stream) are taken by both calls: the encoder notes the jump points at the speed of
 * cst_ans_encode_batch when the chunks are whole 32-symbol tiles, the decoder runs k = 2, 4, 8 or 16 lanes per stream that
 * share the stream's table in on-chip memory.  Any other k decodes the streams whole: the jump points are side information,
 * the symbols and the per-chunk status (the stream's status, repeated) are the same. */
cst_status cst_ans_encode_batch_ckpt(const cst_model *model, cst_coder_config cfg, const int32_t *d_symbols,
                                     size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                     size_t stride_words, uint32_t *d_n_words, size_t ckpt_interval,
                                     uint32_t *d_ckpt_pos, uint64_t *d_ckpt_state, int32_t *d_status, void *stream);
size_t cst_ckpt_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval);
/* ABI 5: the per-chunk status of a *_decode_batch_ckpt call as one status per stream (the worst of its chunks: what the plain
 * decoder of the whole stream reports in d_status[s]).  d_chunk_status[n_streams][n_chunks] -> d_stream_status[n_streams]. */
cst_status cst_ckpt_status_per_stream(const int32_t *d_chunk_status, size_t n_streams, size_t n_chunks,
                                      int32_t *d_stream_status, void *stream);
cst_status cst_ans_decode_batch_ckpt(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                     const uint64_t *d_offsets, size_t stride_words, size_t words_capacity, size_t ckpt_interval,
                                     const uint32_t *d_ckpt_pos, const uint64_t *d_ckpt_state, int32_t *d_symbols,
                                     size_t n_streams, size_t n_per_stream, void *d_scratch, int32_t *d_status,
                                     void *stream);

/* Round 5 (additions to ABI 4): jump points at the speed of the plain encoder, and for NARROW symbol matrices.
 * cst_ans_encode_batch_ckpt on a shared-table model, preset (32,64), 8 <= P <= 12, whole workgroups of 256 stream-major rows that
 * are whole aligned tiles, chunks of whole 32-symbol tiles that divide the rows: the producer / consumer encoder notes the jump
 * points on its way ("ans_encode_pc_kernel<ckpt>").  The _sym forms take symbol_bytes = 1, 2 or 4 as cst_ans_*_batch_sym do: an int8
 * matrix of whole 128-symbol lines is read by the encoder loops themselves ("ans_encode_pc_n8_kernel<ckpt>") and, chunk by chunk
 * (ckpt_interval a multiple of 128), written by the decoder loops -- k x n_streams virtual streams, two waves per SIMD from
 * 65 537 of them on ("ans_decode_small_n8_kernel"): 65 536 x 4096 at k = 2 decode in 0.17 ms against 0.24 ms whole.  Every other
 * shape converts next to the int32 calls.  d_scratch: cst_ckpt_sym_scratch_bytes(...) bytes for both calls (the encoder touches
 * it only when it converts; NULL is accepted where it does not). */
cst_status cst_ans_encode_batch_ckpt_sym(const cst_model *model, cst_coder_config cfg, const void *d_symbols, int32_t symbol_bytes,
                                         size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                         size_t stride_words, uint32_t *d_n_words, size_t ckpt_interval, uint32_t *d_ckpt_pos,
                                         uint64_t *d_ckpt_state, int32_t *d_status, void *d_scratch, void *stream);
size_t cst_ckpt_sym_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval, int32_t symbol_bytes);
cst_status cst_ans_decode_batch_ckpt_sym(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                         const uint64_t *d_offsets, size_t stride_words, size_t words_capacity,
                                         size_t ckpt_interval, const uint32_t *d_ckpt_pos, const uint64_t *d_ckpt_state,
                                         void *d_symbols, int32_t symbol_bytes, size_t n_streams, size_t n_per_stream,
                                         void *d_scratch, int32_t *d_status, void *stream);

/* The same for the reference's flagship call, every symbol its own (mean, std) (cst_ans_encode_gaussian_batch): the fused encoder
 * notes the jump points (batches of at least 16 384 streams; ckpt_interval a multiple of 16 that divides n_per_stream), and the
 * decoder runs every (stream, chunk) pair as a coder of its own -- the parameter matrices have the symbols' shape, so a chunk's
 * models are a row of the [n_streams * n_chunks][interval] view of d_means / d_stds.  Stream-major.  With two jump points per
 * stream a 65 536-stream batch decodes with two resident waves per SIMD (3.4 -> 2.6 ms at 4096 symbols per stream).
 * d_scratch: cst_ckpt_scratch_bytes(...). */
cst_status cst_ans_encode_gaussian_batch_ckpt(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol,
                                              const int32_t *d_symbols, const double *d_means, const double *d_stds,
                                              size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                              size_t stride_words, uint32_t *d_n_words, size_t ckpt_interval,
                                              uint32_t *d_ckpt_pos, uint64_t *d_ckpt_state, int32_t *d_status, void *stream);
cst_status cst_ans_decode_gaussian_batch_ckpt(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol,
                                              const uint32_t *d_words, const uint64_t *d_offsets, size_t stride_words,
                                              size_t words_capacity, size_t ckpt_interval, const uint32_t *d_ckpt_pos,
                                              const uint64_t *d_ckpt_state, const double *d_means, const double *d_stds,
                                              int32_t *d_symbols, size_t n_streams, size_t n_per_stream, void *d_scratch,
                                              int32_t *d_status, void *stream);

/* The same for the range coder: RangeEncoder::pos() / RangeDecoder::seek (src/stream/queue.rs:172-196, 900-926; test
 * :1333-1396).  A jump point is (d_ckpt_pos[s][j] = words emitted so far INCLUDING held-back ones, (d_ckpt_lower[s][j],
 * d_ckpt_range[s][j]) = RangeCoderState) in front of chunk j; seeking continues reading at word `pos`, re-reads `point` from
 * there and takes the state.  The words are exactly those of cst_range_encode_batch.  Shared-table models, any preset and
 * layout on the encoder side (the hand-scheduled (32,64) kernel for stream-major batches notes the jump points on its way);
 * the decoder wants stream-major symbols and n_per_stream a multiple of the interval, takes the whole streams' counts
 * (d_n_words: a lane reads past its chunk, never past its stream) and writes n_streams * n_chunks status entries; a jump point
 * beyond its stream's words reports CST_STREAM_INVALID_DATA for that chunk.  d_scratch: cst_range_ckpt_scratch_bytes(...). */
cst_status cst_range_encode_batch_ckpt(const cst_model *model, cst_coder_config cfg, const int32_t *d_symbols,
                                       size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                       size_t stride_words, uint32_t *d_n_words, size_t ckpt_interval,
                                       uint32_t *d_ckpt_pos, uint64_t *d_ckpt_lower, uint64_t *d_ckpt_range,
                                       int32_t *d_status, void *stream);
size_t cst_range_ckpt_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval);
cst_status cst_range_decode_batch_ckpt(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                       const uint64_t *d_offsets, size_t stride_words, size_t words_capacity,
                                       const uint32_t *d_n_words, size_t ckpt_interval, const uint32_t *d_ckpt_pos,
                                       const uint64_t *d_ckpt_lower, const uint64_t *d_ckpt_range, int32_t *d_symbols,
                                       size_t n_streams, size_t n_per_stream, void *d_scratch, int32_t *d_status,
                                       void *stream);

/* ABI 5: which jump points should a batch carry?  Jump points never change a stream's words; what they buy is lanes -- k jump
 * points per stream let the same words decode on k lanes, and every decoder of this library except the int32 shared-table one
 * (P <= 12) is latency-bound at one wave of streams per SIMD.  The library's own answer for a batch about to be
 * ENCODED with the given arguments (the same meaning as in cst_ans_encode_batch_sym / cst_range_encode_batch; symbol_bytes = 1, 2
 * or 4; d_symbols / d_words / stride_words are looked at for alignment only):
 *   returns the ckpt_interval to pass to cst_ans_encode_batch_ckpt[_sym] / cst_range_encode_batch_ckpt and to the matching
 *   *_decode_batch_ckpt call, or 0 = carry none (the plain decoder is as fast, or the shape is not one the checkpointing encoders
 *   take at the plain encoders' speed).
 * coder: CST_CODER_ANS or CST_CODER_RANGE.  cst_jump_points_auto_gaussian: the same for cst_ans_encode_gaussian_batch_ckpt (every
 * symbol its own mean and std).  The Python layer (constriction_amd.batched, jump_points="auto", the default) and the Rust
 * wrappers ask here; CST_AUTO_JUMP=0 in the environment makes both answer 0. */
#define CST_CODER_ANS 0
#define CST_CODER_RANGE 1
size_t cst_jump_points_auto(const cst_model *model, cst_coder_config cfg, int32_t coder, int32_t symbol_bytes,
                            const void *d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout,
                            const void *d_words, size_t stride_words);
size_t cst_jump_points_auto_gaussian(cst_coder_config cfg, int32_t coder, size_t n_streams, size_t n_per_stream,
                                     cst_layout layout);

/* Debug switches (ABI 5).  The dispatcher's A/B switches are environment variables read ONCE, when the library is loaded --
 * no coder call reads the environment.  Each selects among kernels that produce the same words and symbols (the parity suite
 * runs under every one of them, scripts/alt_paths.sh):
 *   CST_NO_PC_ENCODER=1      never the producer / consumer encoders          CST_NO_PC_WIDE=1   12 < P <= 24 on ans_encode_wide_kernel
 *   CST_PC_COMBINED=1        their helper waves load AND store                CST_NO_N8=1        narrow matrices through the conversion kernels
 *   CST_SMALL_KERNELS=0|enc|dec   never / only the encoder / only the decoder of the small-footprint kernels
 *   CST_DQ_DECODER=1         the lane-quad decoder without CST_FLAG_COLD_WORDS CST_PT_SUB_WAVES=8 sub-lane decoder: never sixteen waves
 *   CST_SUB_ORDER=0          range sub-lane decoder: chunks side by side      CST_LANE_GEO=big|small   per-symbol lane decoder geometry
 *   CST_FUSED_MIN_STREAMS=n  from how many streams the fused per-symbol encoder runs
 *   CST_AUTO_JUMP=0          cst_jump_points_auto* answer 0                  CST_RAGGED_GROUP=8|16|32   ragged encoder: symbols per memory point
 * (CST_RCCL_LIB=<path>, read at the first collective call, names the RCCL library to open.)
 * cst_debug_reload_knobs re-reads them: for tests that drive several paths inside one process; not thread-safe against
 * concurrent coder calls. */
void cst_debug_reload_knobs(void);

/* Exclusive prefix sum of d_n_words into d_offsets[n_streams+1] and gather of the slabs into one packed buffer (the
 * concatenation of every stream's `into_compressed()` result) -- ONE kernel (single-pass scan with decoupled
 * look-back, fused with the copy), fully asynchronous on `stream`: no host synchronisation, no allocation.
 *   d_offsets        out: offsets[s] = first word of stream s in the packed buffer; offsets[n_streams] = total words
 *   d_packed         may be NULL to compute the offsets only
 *   packed_capacity  words available at d_packed (an upper bound such as n_streams * stride_words always suffices);
 *                    a stream that would end beyond it is not copied -- the caller sees offsets[n_streams] > capacity
 *   d_scratch        cst_compact_scratch_bytes(n_streams) bytes of device memory, contents irrelevant on entry
 * (The reference has no counterpart: its coders each own a Vec<u32>; this is the container layout of
 *  src/pybindings/stream/stack.rs:149-166 -- little-endian u32 words, one offset per message.) */
size_t cst_compact_scratch_bytes(size_t n_streams);
cst_status cst_compact_words(const uint32_t *d_words, size_t stride_words, const uint32_t *d_n_words,
                             size_t n_streams, uint64_t *d_offsets, uint32_t *d_packed,
                             size_t packed_capacity, void *d_scratch, void *stream);
/* The same for slabs of PACKED 16-bit words (CST_FLAG_PACKED_W16): stride, counts, offsets and capacity in 16-bit words; the
 * packed buffer is the concatenation of every stream's Vec<u16>.  Two kernels (the scan of cst_compact_words + a halfword
 * gather), same scratch. */
cst_status cst_compact_words16(const uint16_t *d_words16, size_t stride_words, const uint32_t *d_n_words,
                               size_t n_streams, uint64_t *d_offsets, uint16_t *d_packed16,
                               size_t packed_capacity, void *d_scratch, void *stream);

/* cst_ans_encode_batch with CST_FLAG_PACKED_W16 + a jump point in front of every ckpt_interval symbols (ABI 5, round 6):
 * `AnsCoder::pos()` (src/stream/stack.rs:1107-1139) of the (16,32) preset whose words lie as the reference holds them, a
 * Vec<u16> -- d_ckpt_pos[s * n_chunks + j] = 16-bit words in the bulk in front of chunk j, d_ckpt_state[...] = the coder state
 * there (its low 32 bits; n_chunks = n_per_stream / ckpt_interval).  The packed encoder notes them on its way at the plain call's
 * speed; the words are the plain call's.  Stream-major, shared table, 8 <= P <= 12, chunks of whole 32-symbol tiles that divide
 * the rows -- other shapes: CST_ERR_INVALID_ARGUMENT (note the points with cst_ans_encode_batch_ckpt into a scratch slab
 * instead).  Decode from the points: cst_ans_decode_batch on the n_streams * n_chunks chunks as streams of their own --
 * d_offsets[v] = s * stride_words (16-bit words), d_n_words[v] = pos, d_state = A COPY of the states (CST_FLAG_RAW_STATE makes
 * that array in AND out), flags CST_FLAG_RAW_STATE | CST_FLAG_PACKED_W16; symbols as the matrix [n_streams * n_chunks][interval]. */
cst_status cst_ans_encode_batch_ckpt_packed16(const cst_model *model, cst_coder_config cfg, const int32_t *d_symbols,
                                              size_t n_streams, size_t n_per_stream, uint16_t *d_words16,
                                              size_t stride_words, uint32_t *d_n_words, size_t ckpt_interval,
                                              uint32_t *d_ckpt_pos, uint64_t *d_ckpt_state, int32_t *d_status,
                                              void *stream);

/* Every stream's words in REVERSED order: out[s][i] = in[s][n_words[s] - 1 - i].  The reference reads an ANS stream from
 * its END (a stack); `AnsCoder::from_reversed_compressed` (src/stream/stack.rs:734-748) and `Cursor::into_reversed`
 * (src/backends.rs:1424-1448; docs :774-803) are its coders over words stored last-written-first, the order in which a decoder
 * consumes them -- what a file or socket that is decoded while it arrives holds.  This entry point converts a batch between the
 * two orders (it is its own inverse); the decoders of this library take the reference's default order.
 *   d_offsets_in / d_offsets_out   packed layouts (the n_streams + 1 offsets of cst_compact_words: offsets[s] = first word of
 *                                  stream s), or NULL for slabs `stride` words apart
 * In place (same buffer, same layout on both sides) is allowed.  One asynchronous kernel, a wave per stream.  A stream whose
 * count exceeds its slab (stride) or its slice of a packed buffer (offsets[s + 1] - offsets[s]) is left untouched -- counts and
 * offsets are caller data, and the decoders would report such a stream as CST_STREAM_INVALID_DATA. */
cst_status cst_words_reverse(const uint32_t *d_words_in, const uint64_t *d_offsets_in, size_t stride_in,
                             const uint32_t *d_n_words, size_t n_streams, uint32_t *d_words_out,
                             const uint64_t *d_offsets_out, size_t stride_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * multi-GPU: gather of the packed compressed words of every rank to one root over RCCL / xGMI (BASELINE config C5).
 * One process per GPU; streams shard in contiguous blocks and no collective touches the coding path; this is the only
 * exchange step.  The library opens librccl at first use (no link-time dependency); `comm` is an ncclComm_t -- the
 * caller's own, or one made with the two helpers below from an id that rank 0 creates and the caller's launcher
 * distributes (128 bytes).  (No reference counterpart: its coders are single-threaded host objects.)
 * ---------------------------------------------------------------------------------------- */
cst_status cst_rccl_get_unique_id(void *h_id /* 128 bytes out */);
cst_status cst_rccl_comm_init(const void *h_id, int32_t n_ranks, int32_t rank, void **out_comm); /* on the current device */
cst_status cst_rccl_comm_destroy(void *comm);

/* Step 1 (every rank, asynchronous on `stream`): d_sizes[2 * n_ranks] (device, uint64) receives (n_streams, total_words)
 * of every rank, in rank order; d_offsets is the rank's own offsets[n_streams_local + 1] from cst_compact_words. */
cst_status cst_gather_sizes_rccl(void *comm, int32_t n_ranks, int32_t rank, const uint64_t *d_offsets,
                                 size_t n_streams_local, uint64_t *d_sizes, void *stream);

/* Step 2 (every rank): h_sizes = host copy of d_sizes.  Root: d_all_packed (capacity >= sum of words) receives the packed
 * words of all ranks in rank order, d_all_offsets[sum of streams + 1] their global offsets; other ranks pass NULL for
 * both.  Grouped point-to-point transfers straight into their final positions, asynchronous on `stream` on every
 * rank.  A failing transfer never leaves an RCCL group open: the group is closed first, then the error is returned. */
cst_status cst_gather_rccl(void *comm, int32_t n_ranks, int32_t rank, int32_t root, const uint32_t *d_packed,
                           const uint64_t *d_offsets, const uint64_t *h_sizes, uint32_t *d_all_packed,
                           uint64_t *d_all_offsets, void *stream);

/* The inverse (decoding on the ranks what one rank holds): the root passes the packed words of all ranks'
 * streams in rank order and their global offsets[sum of streams + 1] (NULL on the other ranks); every rank receives its
 * own words in d_packed (capacity >= h_sizes[2 * rank + 1]) and d_offsets[n_streams_local + 1] rebased to start at 0 --
 * what the `d_offsets` form of cst_ans_decode_batch / cst_range_decode_batch takes.  h_sizes as above, on every rank. */
cst_status cst_scatter_rccl(void *comm, int32_t n_ranks, int32_t rank, int32_t root, const uint32_t *d_all_packed,
                            const uint64_t *d_all_offsets, const uint64_t *h_sizes, uint32_t *d_packed,
                            uint64_t *d_offsets, void *stream);

/* ------------------------------------------------------------------------------------------
 * per-symbol quantized Gaussians: the reference's flagship Python call
 *     coder.encode_reverse(symbols, QuantizedGaussian(min, max), means, stds)
 *     coder.decode(QuantizedGaussian(min, max), means, stds)
 * (src/pybindings/stream/stack.rs:567-588, 733-751; src/pybindings/stream/model/internals.rs:188-249)
 * d_means / d_stds have the same shape and layout as d_symbols (f64; f32 callers widen first,
 * src/pybindings/mod.rs:211-216).  std <= 0 or a non-finite parameter yields
 * CST_STREAM_IMPOSSIBLE_SYMBOL for that stream (the reference panics: pybindings/stream/model.rs:654-657).
 * ---------------------------------------------------------------------------------------- */
cst_status cst_ans_encode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol,
                                         const int32_t *d_symbols, const double *d_means, const double *d_stds,
                                         size_t n_streams, size_t n_per_stream, cst_layout layout,
                                         uint32_t *d_words, size_t stride_words, uint32_t *d_n_words,
                                         uint64_t *d_state, int32_t *d_status, uint32_t flags, void *stream);

cst_status cst_ans_decode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol,
                                         const uint32_t *d_words, const uint64_t *d_offsets, size_t stride_words, size_t words_capacity,
                                         const uint32_t *d_n_words, const double *d_means, const double *d_stds,
                                         int32_t *d_symbols, size_t n_streams, size_t n_per_stream,
                                         cst_layout layout, uint64_t *d_state, uint32_t *d_n_words_out,
                                         int32_t *d_status, uint32_t flags, void *stream);

/* Per-symbol models given explicitly (any model family with per-symbol parameters, e.g.
 * Categorical(perfect=False) with a probability matrix, src/pybindings/stream/model/internals.rs:188-249):
 *   encode: d_left / d_prob hold EncoderModel::left_cumulative_and_probability of every symbol
 *           (same shape/layout as the symbol matrix; prob == 0 marks an impossible symbol);
 *   decode: d_cdf_rows holds one cdf[n_symbols+1] row per coded symbol, row index = element index of the
 *           symbol matrix (s*n_per_stream + t, or t*n_streams + s for CST_LAYOUT_SYMBOL_MAJOR). */
cst_status cst_ans_encode_cp_batch(cst_coder_config cfg, const uint32_t *d_left, const uint32_t *d_prob,
                                   size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                   size_t stride_words, uint32_t *d_n_words, uint64_t *d_state, int32_t *d_status,
                                   uint32_t flags, void *stream);

cst_status cst_ans_decode_rows_batch(cst_coder_config cfg, const uint32_t *d_words, const uint64_t *d_offsets,
                                     size_t stride_words, size_t words_capacity, const uint32_t *d_n_words, const uint32_t *d_cdf_rows,
                                     int32_t n_symbols, int32_t min_symbol, int32_t *d_symbols, size_t n_streams,
                                     size_t n_per_stream, cst_layout layout, uint64_t *d_state,
                                     uint32_t *d_n_words_out, int32_t *d_status, uint32_t flags, void *stream);

/* ------------------------------------------------------------------------------------------
 * batched range coder (BASELINE config C4): one RangeEncoder / RangeDecoder per stream
 *   encode: RangeEncoder::encode_iid_symbols + get_compressed   src/stream/queue.rs:612-705, 458-523
 *   decode: RangeDecoder::from_compressed + decode_iid_symbols  src/stream/queue.rs:847-868, 968-1033
 * Symbols are coded in forward order (a queue).  Same slab/offset conventions as the ANS calls.
 * ---------------------------------------------------------------------------------------- */
/* Raw coder state of one range coder, for continuing a coder across calls (the single-coder drop-in):
 * RangeCoderState{lower, range} + EncoderSituation (queue.rs:60-71, 126-142) on the encoder side,
 * RangeCoderState + point + words consumed on the decoder side. */
typedef struct cst_range_state {
    uint64_t lower;
    uint64_t range;
    uint64_t point;        /* decoder only */
    uint32_t inverted_n;   /* encoder only: 0 = EncoderSituation::Normal */
    uint32_t inverted_first;
    uint64_t position;     /* decoder only: words consumed so far */
} cst_range_state;

/* d_rstate (may be NULL unless CST_FLAG_RAW_STATE) is in/out with CST_FLAG_RAW_STATE: the encoder then starts
 * from it and appends no seal words, the decoder does not re-read `point` and continues at ->position. */
cst_status cst_range_encode_batch(const cst_model *model, cst_coder_config cfg, const int32_t *d_symbols,
                                  size_t n_streams, size_t n_per_stream, cst_layout layout,
                                  uint32_t *d_words, size_t stride_words, uint32_t *d_n_words,
                                  cst_range_state *d_rstate, int32_t *d_status, uint32_t flags, void *stream);

cst_status cst_range_decode_batch(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                  const uint64_t *d_offsets, size_t stride_words, size_t words_capacity, const uint32_t *d_n_words,
                                  int32_t *d_symbols, size_t n_streams, size_t n_per_stream, cst_layout layout,
                                  cst_range_state *d_rstate, int32_t *d_status, uint32_t flags, void *stream);

/* ABI 5: NARROW symbol matrices for the range coder -- the reference's RangeEncoder / RangeDecoder are generic over the symbol type
 * (src/stream/queue.rs:612, 968; src/stream/model/quantize.rs:229-255).  The four calls below are cst_range_{en,de}code_batch[_ckpt] with
 * the matrix given as symbol_bytes = 1 (int8), 2 (int16) or 4 (int32: the plain call); words, counts, status and jump points are
 * those of the int32 call on the widened values.  An int8 matrix of stream-major rows of whole 32-symbol tiles at (32,64),
 * 8 <= P <= 24, is read by the hand-scheduled encoder itself ("range_encode_n8_kernel" / "range_encode_ckpt_n8_kernel": a quarter of the
 * symbol bytes, one more instruction per symbol) and written by the sub-lane decoder itself ("range_decode_n8_kernel" /
 * "range_decode_sub_n8_kernel": its byte tiles hold the symbols); every other shape converts next to the int32 call
 * (cst_symbols_widen / cst_symbols_narrow).  The decoders want a support that fits the type.
 * d_scratch: cst_range_sym_scratch_bytes(n_streams, n_per_stream, ckpt_interval or 0, symbol_bytes) bytes, contents irrelevant. */
size_t cst_range_sym_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval, int32_t symbol_bytes);
cst_status cst_range_encode_batch_sym(const cst_model *model, cst_coder_config cfg, const void *d_symbols, int32_t symbol_bytes,
                                      size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                      size_t stride_words, uint32_t *d_n_words, cst_range_state *d_rstate, int32_t *d_status,
                                      uint32_t flags, void *d_scratch, void *stream);
cst_status cst_range_decode_batch_sym(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                      const uint64_t *d_offsets, size_t stride_words, size_t words_capacity,
                                      const uint32_t *d_n_words, void *d_symbols, int32_t symbol_bytes, size_t n_streams,
                                      size_t n_per_stream, cst_layout layout, cst_range_state *d_rstate, int32_t *d_status,
                                      uint32_t flags, void *d_scratch, void *stream);
cst_status cst_range_encode_batch_ckpt_sym(const cst_model *model, cst_coder_config cfg, const void *d_symbols, int32_t symbol_bytes,
                                           size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                           size_t stride_words, uint32_t *d_n_words, size_t ckpt_interval, uint32_t *d_ckpt_pos,
                                           uint64_t *d_ckpt_lower, uint64_t *d_ckpt_range, int32_t *d_status, void *d_scratch,
                                           void *stream);
cst_status cst_range_decode_batch_ckpt_sym(const cst_model *model, cst_coder_config cfg, const uint32_t *d_words,
                                           const uint64_t *d_offsets, size_t stride_words, size_t words_capacity,
                                           const uint32_t *d_n_words, size_t ckpt_interval, const uint32_t *d_ckpt_pos,
                                           const uint64_t *d_ckpt_lower, const uint64_t *d_ckpt_range, void *d_symbols,
                                           int32_t symbol_bytes, size_t n_streams, size_t n_per_stream, void *d_scratch,
                                           int32_t *d_status, void *stream);


/* Per-symbol-model variants of the range coder (same argument meaning as the cst_ans_* twins). */
cst_status cst_range_encode_cp_batch(cst_coder_config cfg, const uint32_t *d_left, const uint32_t *d_prob,
                                     size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                     size_t stride_words, uint32_t *d_n_words, cst_range_state *d_rstate,
                                     int32_t *d_status, uint32_t flags, void *stream);

cst_status cst_range_encode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol,
                                           const int32_t *d_symbols, const double *d_means, const double *d_stds,
                                           size_t n_streams, size_t n_per_stream, cst_layout layout,
                                           uint32_t *d_words, size_t stride_words, uint32_t *d_n_words,
                                           cst_range_state *d_rstate, int32_t *d_status, uint32_t flags, void *stream);

cst_status cst_range_decode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol,
                                           const uint32_t *d_words, const uint64_t *d_offsets, size_t stride_words, size_t words_capacity,
                                           const uint32_t *d_n_words, const double *d_means, const double *d_stds,
                                           int32_t *d_symbols, size_t n_streams, size_t n_per_stream,
                                           cst_layout layout, cst_range_state *d_rstate, int32_t *d_status,
                                           uint32_t flags, void *stream);

/* ABI 5: jump points for the per-symbol Gaussian calls of the RANGE coder (RangeEncoder::pos / RangeDecoder::seek, src/stream/queue.rs:172-196,
 * 900-926), as cst_ans_{encode,decode}_gaussian_batch_ckpt are for ANS: the fused encoder notes (words emitted including held-back ones,
 * lower, range) in front of every chunk of ckpt_interval symbols (a multiple of 16 that divides n_per_stream; stream-major), the decoder
 * runs every (stream, chunk) pair as a decoder of its own -- chunk j of stream s codes row s * n_chunks + j of the three matrices viewed
 * as [n_streams * n_chunks][interval] -- and writes n_streams * n_chunks status entries.  The words are those of
 * cst_range_encode_gaussian_batch.  d_scratch: cst_range_gaussian_ckpt_scratch_bytes(...). */
cst_status cst_range_encode_gaussian_batch_ckpt(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol,
                                                const int32_t *d_symbols, const double *d_means, const double *d_stds,
                                                size_t n_streams, size_t n_per_stream, cst_layout layout, uint32_t *d_words,
                                                size_t stride_words, uint32_t *d_n_words, size_t ckpt_interval,
                                                uint32_t *d_ckpt_pos, uint64_t *d_ckpt_lower, uint64_t *d_ckpt_range,
                                                int32_t *d_status, void *stream);
size_t cst_range_gaussian_ckpt_scratch_bytes(size_t n_streams, size_t n_per_stream, size_t ckpt_interval);
cst_status cst_range_decode_gaussian_batch_ckpt(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol,
                                                const uint32_t *d_words, const uint64_t *d_offsets, size_t stride_words,
                                                size_t words_capacity, const uint32_t *d_n_words, size_t ckpt_interval,
                                                const uint32_t *d_ckpt_pos, const uint64_t *d_ckpt_lower,
                                                const uint64_t *d_ckpt_range, const double *d_means, const double *d_stds,
                                                int32_t *d_symbols, size_t n_streams, size_t n_per_stream, void *d_scratch,
                                                int32_t *d_status, void *stream);

cst_status cst_range_decode_rows_batch(cst_coder_config cfg, const uint32_t *d_words, const uint64_t *d_offsets,
                                       size_t stride_words, size_t words_capacity, const uint32_t *d_n_words, const uint32_t *d_cdf_rows,
                                       int32_t n_symbols, int32_t min_symbol, int32_t *d_symbols, size_t n_streams,
                                       size_t n_per_stream, cst_layout layout, cst_range_state *d_rstate,
                                       int32_t *d_status, uint32_t flags, void *stream);

/* ------------------------------------------------------------------------------------------
 * ChainCoder (src/stream/chain.rs:231-246; DefaultChainCoder = <u32, u64, 24>, SmallChainCoder = <u16, u32, 12>)
 *
 * A chain coder holds two word stacks, `compressed` and `remainders`, and two heads (chain.rs:248-258):
 *   compressed_head   the bits of the last compressed word that are not consumed yet, below a leading 1 bit
 *   remainders_head   2^(S-W-P) <= head < 2^(S-P) between calls
 * Decoding (decode_symbol, chain.rs:1044-1122) POPS P bits per symbol from `compressed` -- whatever the model -- and
 * PUSHES what the symbol did not use onto `remainders`; encoding (encode_symbol, chain.rs:1140-1209; symbols last to
 * first) pops from `remainders` and pushes P bits per symbol onto `compressed`.  The constructors and terminators
 * (from_binary / from_compressed / from_remainders, into_remainders / into_compressed / into_binary: a handful of
 * words each) are host code in the binding (constriction_amd/stream/chain.py); these entry points are the symbol loops.
 *
 *   d_pop_words / d_pop_offsets / pop_stride / d_n_pop   the stack that is popped, laid out like d_words of the ANS
 *             decoder (stream s: d_pop_words[off(s) .. off(s) + d_n_pop[s]), consumed from the END); d_n_pop is updated;
 *             d_pop_words must be a valid device pointer (at least one word) even if every d_n_pop[s] is 0
 *   d_push_words / push_stride / d_n_push                the words pushed, in push order, stream s at s * push_stride;
 *             at most one word per symbol; d_n_push[s] = their number
 *   d_heads   in / out, one per stream
 * d_status: CST_STREAM_OUT_OF_DATA if the popped stack ran empty (the reference returns an error; the stream stops
 * there), IMPOSSIBLE_SYMBOL / CAPACITY as for the other coders.
 * ---------------------------------------------------------------------------------------- */
typedef struct cst_chain_heads {
    uint64_t remainders_head;
    uint32_t compressed_head;
    uint32_t reserved;
} cst_chain_heads;

cst_status cst_chain_encode_cp_batch(cst_coder_config cfg, const uint32_t *d_left, const uint32_t *d_prob, size_t n_streams,
                                     size_t n_per_stream, cst_layout layout, const uint32_t *d_pop_words,
                                     const uint64_t *d_pop_offsets, size_t pop_stride, uint32_t *d_n_pop, uint32_t *d_push_words,
                                     size_t push_stride, uint32_t *d_n_push, cst_chain_heads *d_heads, int32_t *d_status,
                                     void *stream);
cst_status cst_chain_encode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const int32_t *d_symbols,
                                           const double *d_means, const double *d_stds, size_t n_streams, size_t n_per_stream,
                                           cst_layout layout, const uint32_t *d_pop_words, const uint64_t *d_pop_offsets,
                                           size_t pop_stride, uint32_t *d_n_pop, uint32_t *d_push_words, size_t push_stride,
                                           uint32_t *d_n_push, cst_chain_heads *d_heads, int32_t *d_status, void *stream);
cst_status cst_chain_decode_gaussian_batch(cst_coder_config cfg, int32_t min_symbol, int32_t max_symbol, const uint32_t *d_pop_words,
                                           const uint64_t *d_pop_offsets, size_t pop_stride, uint32_t *d_n_pop, const double *d_means,
                                           const double *d_stds, int32_t *d_symbols, size_t n_streams, size_t n_per_stream,
                                           cst_layout layout, uint32_t *d_push_words, size_t push_stride, uint32_t *d_n_push,
                                           cst_chain_heads *d_heads, int32_t *d_status, void *stream);
/* explicit cdf rows [n_symbols + 1] per coded symbol; row_stride = n_symbols + 1, or 0 for ONE row shared by every
 * symbol (a concrete model: decode_iid_symbols) */
cst_status cst_chain_decode_rows_batch(cst_coder_config cfg, const uint32_t *d_pop_words, const uint64_t *d_pop_offsets,
                                       size_t pop_stride, uint32_t *d_n_pop, const uint32_t *d_cdf_rows, size_t row_stride,
                                       int32_t n_symbols, int32_t min_symbol, int32_t *d_symbols, size_t n_streams,
                                       size_t n_per_stream, cst_layout layout, uint32_t *d_push_words, size_t push_stride,
                                       uint32_t *d_n_push, cst_chain_heads *d_heads, int32_t *d_status, void *stream);

/* The per-symbol entry points above take their scratch (16 B per symbol for encoding; 1 KiB per symbol of cdf rows, at
 * most 64 MiB at a time, for decoding fewer than 64 streams) from a stream-ordered memory pool that the LIBRARY owns (one
 * per device, created at first use; the device's default pool is never touched) and that keeps freed memory
 * (re-allocating 4 GiB per call cost more than coding them).  This hands it back: synchronises the device and trims the
 * library's pool. */
cst_status cst_release_scratch(void);

/* ------------------------------------------------------------------------------------------
 * bit-exact f64 special functions on device (test hooks for the model kernels)
 * out[i] = erf(x[i]) resp. Gaussian cdf, evaluated by the same device code the table kernels use.
 * ---------------------------------------------------------------------------------------- */
cst_status cst_debug_erf(const double *d_x, double *d_out, size_t n, void *stream);
/* the same erf as the per-symbol kernels evaluate it (one Horner recurrence over per-lane coefficients from LDS) */
cst_status cst_debug_erf_tab(const double *d_x, double *d_out, size_t n, void *stream);
/* The per-symbol kernels evaluate the Gaussian left cumulative through a FAST erf and fall back to the bit-exact one
 * wherever free_weight * cdf lies so close to an integer that the difference could change the truncation (cst_math.hpp);
 * the integer is the reference's in every case.  Hooks: which = 0: out[i] = fast erf(x[i]); 1: |fast - exact|. */
cst_status cst_debug_erf_fast(int32_t which, const double *d_x, double *d_out, size_t n, void *stream);
/* left cumulative of symbol index d_index[i] (0 .. n_symbols) under Gaussian(d_means[i], d_stds[i]) both ways:
 * d_counts[0] += results that differ (must stay 0), d_counts[1] += exact fallbacks taken (caller zeroes d_counts). */
cst_status cst_debug_gaussian_left_quick(int32_t precision, int32_t min_symbol, int32_t max_symbol, const int32_t *d_index,
                                         const double *d_means, const double *d_stds, size_t n, uint64_t *d_counts,
                                         void *stream);
cst_status cst_debug_gaussian_lcp(int32_t precision, int32_t prob_bits, int32_t min_symbol, int32_t max_symbol,
                                  const int32_t *d_symbols, const double *d_means, const double *d_stds,
                                  uint32_t *d_left, uint32_t *d_prob, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CONSTRICTION_AMD_H */
