// cst_common.hpp -- shared host/device declarations of the MI355X entropy-coding backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/constriction_amd.h"

namespace cst {

constexpr int kWave = 64;        // gfx950 wavefront
constexpr int kBlock = 256;      // 4 waves = one per SIMD of a CU
constexpr int kTileSyms = 32;    // symbols per stream staged per LDS tile (128 B rows)
constexpr int kTileStride = 36;  // words per LDS tile row: 16-B aligned and conflict-free for b128

// Encoder table entry (one per symbol of the support): left cumulative, probability and the
// 64-bit reciprocal m = floor(2^64 / p) (p >= 2; m = 2^64-1 for p = 1) used for the exact
// state / p of src/stream/stack.rs:1042-1043.  The high word of m is floor(2^32 / p), the
// reciprocal the 32-bit-state preset uses.
struct __attribute__((aligned(16))) EncEntry {
    uint32_t c;
    uint32_t p;
    uint32_t m_lo;
    uint32_t m_hi;
};

// Decoder lookup for quantile q (lookup_contiguous.rs:564-605 collapsed into table reads), P <= 16:
//   cp[q]  = c | p << 16   (u32: left cumulative and probability of the bin that holds q, both < 2^16; one random
//                           32-bit LDS read -- 64-bit random reads cost ~35 cycles more on gfx950)
//   idx[q] = symbol index  (u16, needed only for the output, off the coder's critical path)
__host__ __device__ inline uint32_t pack_cp(uint32_t c, uint32_t p) { return (c & 0xffffu) | (p << 16); }

enum DecMode : int {
    kDecLutCP = 1,   // cp[2^P] + idx[2^P] in LDS (P <= 14) or global
    kDecBucket = 2,  // cdf[n+1] + bucket index, linear scan (any P)
};

// kernel-internal launch flag (never part of the ABI): the workgroup's LDS holds a second symbol tile per wave
constexpr uint32_t CST_KFLAG_TWO_TILES = 0x80000000u;

struct DecLut {      // pointers into LDS or global memory
    const uint32_t* cp;     // c | p << 16 per quantile
    const uint16_t* idx;    // symbol index per quantile ...
    const int32_t* sym;     // ... or, if non-null, the decoded symbol itself (idx + min_symbol)
    int32_t min_symbol;
    // bucket mode, tables in LDS: ONE 16-byte entry per bucket of quantiles,
    //   { cdf[i0] | i0 << idx_shift, cdf[i0 + 1], cdf[i0 + 2], cdf[i0 + 3] }   (i0 = the symbol that holds the bucket's first
    // quantile, cumulatives beyond the table read as 2^P): a single LDS round trip resolves up to three symbols per bucket.
    // The index shares the first word with its cumulative: 8 bits above 24 (alphabets of <= 256 symbols, P <= 24) or 10 bits
    // above 22 (<= 1024 symbols, P <= 22).
    const uint4* b16;
    int32_t idx_shift;
    // Second-level tables (0 = none): a bucket in which MORE than three symbols begin -- the far tails of a distribution: for
    // the BASELINE Gaussian at P = 24, 22 symbols share bucket 0 -- would send its lanes on a walk over the cdf table, one LDS
    // round trip per symbol (28 % of the P = 24 decoder's time, all of it in two buckets).  Such a bucket owns SLOT
    // (bucket mod kSubTables) of the area behind the bucket entries (`sub`): 2^sub_bits entries of the same form for the
    // bucket's 2^sub_bits equal parts, read with the low five bits of the bucket number and the next sub_bits bits of the
    // quantile by the lanes whose quantile lies at or above their entry's fourth cumulative -- one more round trip, then the
    // selects again; what still overflows there walks.  The bucket's own entry stays a real one (lanes below its fourth
    // cumulative never leave the main path).  Two crowded buckets with the same slot: the lower one has it.  An entry read from
    // a slot is taken if it is a real entry (second word above the first cumulative; free slots hold zeros) whose first
    // cumulative does not lie above the quantile: the slot's owner passes, a LOWER owner passes too (its symbols lie below q,
    // the walk from there is longer but ends at the same symbol), a higher owner does not (the lane walks from its own entry).
    const uint4* sub;
    int32_t sub_bits;
};
constexpr int kSubTables = 32;                  // second-level tables per workgroup image
constexpr int kSubBitsMax = 4;                  // entries per table = 2^min(kSubBitsMax, P - bucket_bits)
constexpr int kSubTableBytes = 16 << kSubBitsMax;
constexpr int kSubAreaBytes = kSubTables * kSubTableBytes + 16 + 4 * kSubTables;     // tables, their count, their buckets
__host__ __device__ inline int bucket16_index_shift(int n_symbols) { return n_symbols <= 256 ? 24 : 22; }
__host__ __device__ inline bool bucket16_usable(int n_symbols, int P) {
    return (n_symbols <= 256 && P <= 24) || (n_symbols <= 1024 && P <= 22);
}

// Per-stream tables in their compact form (cst_ans_pt.hip).  A quantized distribution over a support much wider than
// its scale (the learned-compression case: support -127..127, std 0.5..16) consists mostly of runs of unit
// probabilities: the leaky quantizer gives every symbol at least 1/2^P (quantize.rs:525-568).
//   encoder row (16-bit cumulatives c[a..b+1], indexed by symbol):
//       a = last index of the leading run of unit probabilities (0 if p[0] > 1), b = first index of the trailing run
//       (n-1 if p[n-1] > 1); a symbol i < a has (c, p) = (c[a] - (a - i), 1), a symbol i > b has (c[b] + (i - b), 1).
//   decoder row (32-bit entries sorted by c, searched by quantile): one entry per symbol,
//       c << 20 | (p - 1) << 8 | index,  except that every maximal run of >= 2 unit probabilities is ONE entry
//       c << 20 | 0xfff << 8 | first index  (quantile q of the run is symbol index + (q - c), with (c, p) = (q, 1));
//       kPtRowPad sentinels 0xffffffff behind the row (more up to a whole quad of entries: rows start on 16 bytes).  P <= 12, n <= 256.
// Rows of different lengths lie back to back, blocks of kBlock streams are contiguous.
struct PtMeta {
    uint32_t enc_off;   // first 16-bit entry of the encoder row, relative to its block
    uint32_t dec_off;   // first 32-bit entry of the decoder row, relative to its block
    uint16_t a;         // first symbol index held by the encoder row
    uint16_t m;         // symbols held by the encoder row (it has m + 1 entries)
    uint16_t m_dec;     // entries of the decoder row (sentinels not counted)
    uint16_t pad;
};
constexpr int kPtBucketBits = 7, kPtBuckets = 1 << kPtBucketBits;
constexpr int kPtRowPad = 7;          // sentinel entries behind every decoder row: the decoder reads EIGHT entries at a time from the
                                      // 16-byte aligned quad that holds its first candidate (two ds_read_b128: scripts/gen_pt_decode_loop.py);
                                      // rows start on 16 bytes (their length is rounded up to whole quads)
constexpr uint32_t kPtRunMark = 0xfffu;

} // namespace cst

// The opaque model handle of the C ABI.
struct cst_model {
    int32_t precision = 0;
    int32_t min_symbol = 0;
    int32_t n_symbols = 0;
    size_t n_tables = 1;     // number of tables
    bool per_stream = false; // true: table s belongs to stream s (config C3); false: one table shared by all streams
    int device = 0;
    int cu_count = 256;      // compute units of that device (a batch of more than cu_count * 256 streams has more than one wave per SIMD)
    uint32_t* d_cdf = nullptr;        // [n_tables][n_symbols + 1]
    // shared-table artefacts (n_tables == 1)
    cst::EncEntry* d_enc = nullptr;   // [n_symbols]
    uint32_t* d_dec_cp = nullptr;     // [2^P] or null (P <= 16)
    uint16_t* d_dec_idx = nullptr;    // [2^P] or null
    uint16_t* d_bucket = nullptr;     // [2^bucket_bits + 1]
    int32_t bucket_bits = 0;
    // per-stream artefacts (n_tables > 1): 16-bit cdf rows (row length = cdf16_stride, a power of two >= n+1;
    // values are taken modulo 2^16, so 2^16 is stored as 0) and the reciprocal table floor(2^64 / p), p < 2^P
    uint16_t* d_cdf16 = nullptr;      // [n_tables][cdf16_stride]
    int32_t cdf16_stride = 0;
    uint64_t* d_recip = nullptr;      // [2^P]
    // non-contiguous alphabets (symbol remapping, lookup_noncontiguous.rs / non_contiguous.rs): symbol of index i, and the
    // symbols sorted with the index each one stands for
    int32_t* d_symbol_of_index = nullptr;   // [n_symbols] or null (contiguous model)
    int32_t* d_sorted_symbols = nullptr;    // [n_symbols]
    int32_t* d_sorted_index = nullptr;      // [n_symbols]
    // per-stream TRIMMED PACKED rows (cst_ans_pt.hip; built when P <= 12 and n <= 256): see cst::PtMeta
    bool pt_ok = false;
    cst::PtMeta* d_pt_meta = nullptr;     // [n_tables]
    uint16_t* d_pt_enc = nullptr;         // encoder rows of all blocks, back to back
    uint32_t* d_pt_dec = nullptr;         // decoder rows
    uint8_t* d_pt_l1 = nullptr;           // [n_tables][kPtBuckets] quantile bucket -> first candidate (position in the decoder row)
    uint32_t* d_pt_block_base = nullptr;  // [2][n_blocks + 1] first encoder / decoder row entry of every block of kBlock tables
    uint32_t pt_max_enc = 0, pt_max_dec = 0;   // largest block (entries): sizes the LDS image of a workgroup
    uint32_t pt_max_dec64 = 0;                 // largest group of 64 consecutive tables' decoder rows (entries): the sub-lane decoder's workgroups
};

namespace cst {

// The dispatcher's A/B switches (include/constriction_amd.h, "Debug switches"): CST_* environment variables, read ONCE when the
// library is loaded into this struct -- no coder call reads the environment.  cst_debug_reload_knobs() re-reads them (the parity
// tests that drive an alternate kernel path inside one process).  They select among kernels that produce the SAME words and symbols.
struct Knobs {
    bool no_pc_encoder = false;       // CST_NO_PC_ENCODER: never the producer / consumer encoders (cst_ans_pc.hip)
    bool no_n8 = false;               // CST_NO_N8: int8 / int16 matrices always through the conversion kernels
    bool no_pc_wide = false;          // CST_NO_PC_WIDE: 12 < P <= 24 on ans_encode_wide_kernel
    bool pc_combined = false;         // CST_PC_COMBINED: every helper wave of the pc encoder loads AND stores
    bool dq_decoder = false;          // CST_DQ_DECODER: the lane-quad decoder without CST_FLAG_COLD_WORDS
    bool small_encoders = true;       // CST_SMALL_KERNELS=0|enc|dec: never / only the encoder / only the decoder of the small-footprint kernels
    bool small_decoders = true;
    bool pt_sub_8_waves = false;      // CST_PT_SUB_WAVES=8: the per-stream-table sub-lane decoder never takes sixteen waves
    bool sub_order_flat = false;      // CST_SUB_ORDER=0: range sub-lane decoder, the chunks of a group side by side
    int lane_geo = 0;                 // CST_LANE_GEO=big|small: 1 / 2 forces a geometry of the per-symbol lane decoder (0: by shape)
    size_t fused_min_streams = 16384; // CST_FUSED_MIN_STREAMS: from how many streams the fused per-symbol encoder runs
    int auto_jump = 1;                // CST_AUTO_JUMP=0: cst_jump_points_auto answers 0 (the plain decoders everywhere)
    int ragged_group = 16;            // CST_RAGGED_GROUP=8|16|32: symbols per memory point of the ragged encoder (cst_ans_ragged.hip)
};
const Knobs& knobs();

// thread-local record of the last HIP failure (cst_last_hip_error)
void set_hip_error(hipError_t e, const char* what);
// thread-local record of which kernel family the last coder call of this thread launched (cst_last_kernel_name); returns rc
cst_status note_kernel(const char* name, cst_status rc);
// int8 matrices inside the loops (cst_api.hip / cst_ans_n8.hip): false = not their shape, the caller converts instead
bool ans_encode_n8_try(const cst_model* model, cst_coder_config cfg, const void* d_symbols8, int32_t symbol_bytes, size_t n_streams, size_t n_per_stream,
                       cst_layout layout, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words, uint64_t* d_state, int32_t* d_status,
                       uint32_t flags, void* stream, cst_status* rc);
bool ans_decode_n8_try(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words,
                       size_t words_capacity, const uint32_t* d_n_words, void* d_symbols8, int32_t symbol_bytes, size_t n_streams, size_t n_per_stream,
                       cst_layout layout, uint64_t* d_state, uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags, void* stream,
                       cst_status* rc);

// jump tables are caller data: a point that claims more words than its stream's first point (the whole bulk), or than a slab
// holds (stride_words != 0), flags its chunk CST_STREAM_INVALID_DATA behind the decode (cst_ans_ckpt.hip)
cst_status flag_bad_jump_points(const uint32_t* d_ckpt_pos, size_t n_streams, size_t n_chunks, size_t stride_words, int32_t* d_status, hipStream_t hs);

#define CST_HIP_TRY(expr)                                      \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) {                                \
            ::cst::set_hip_error(_e, #expr);                   \
            return CST_ERR_HIP;                                \
        }                                                      \
    } while (0)

// per-stream-table coder launches (cst_ans_ps.hip)
cst_status ans_encode_per_stream(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams,
                                 size_t n_per_stream, cst_layout layout, uint32_t* d_words, size_t stride_words,
                                 uint32_t* d_n_words, uint64_t* d_state, int32_t* d_status, uint32_t flags, hipStream_t hs);
cst_status ans_decode_per_stream(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                                 size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols, size_t n_streams,
                                 size_t n_per_stream, cst_layout layout, uint64_t* d_state, uint32_t* d_n_words_out,
                                 int32_t* d_status, uint32_t flags, hipStream_t hs);

// cst_ans_ragged.hip: streams of different lengths (arguments checked by the C entry points in cst_api.hip)
cst_status ans_encode_ragged(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, const uint64_t* d_sym_offsets,
                             size_t n_streams, uint32_t* d_words, const uint64_t* d_word_offsets, size_t stride_words,
                             uint32_t* d_n_words, int32_t* d_status, const uint32_t* d_order, hipStream_t hs);
cst_status ans_decode_ragged(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                             size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols,
                             const uint64_t* d_sym_offsets, size_t n_streams, int32_t* d_status, const uint32_t* d_order, hipStream_t hs);
cst_status ans_count_until(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                           size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, size_t n_streams, int32_t eof_symbol,
                           size_t max_symbols, uint64_t* d_lengths, int32_t* d_status, const uint32_t* d_order, hipStream_t hs);

// ... with jump points in front of every `interval` symbols of every stream (round 6): chunk j of stream s = entry d_chunk_offsets[s] + j
cst_status ans_encode_ragged_jump(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, const uint64_t* d_sym_offsets,
                                  size_t n_streams, uint32_t* d_words, const uint64_t* d_word_offsets, size_t stride_words, uint32_t* d_n_words,
                                  int32_t* d_status, const uint32_t* d_order, uint32_t interval, const uint64_t* d_chunk_offsets,
                                  uint32_t* d_jump_pos, uint64_t* d_jump_state, hipStream_t hs);
cst_status ans_decode_ragged_jump(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_word_offsets,
                                  size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols,
                                  const uint64_t* d_sym_offsets, size_t n_streams, uint32_t interval, const uint64_t* d_chunk_offsets,
                                  size_t n_chunks_total, const uint32_t* d_jump_pos, const uint64_t* d_jump_state, void* d_scratch,
                                  int32_t* d_status, hipStream_t hs);

// trimmed-packed-row coder launches (cst_ans_pt.hip); return CST_ERR_INVALID_ARGUMENT if the shape is not theirs
bool pt_usable(const cst_model* model, cst_coder_config cfg, cst_layout layout, size_t n_per_stream);
cst_status ans_encode_pt(const cst_model* model, cst_coder_config cfg, const int32_t* d_symbols, size_t n_streams,
                         size_t n_per_stream, uint32_t* d_words, size_t stride_words, uint32_t* d_n_words, uint64_t* d_state,
                         int32_t* d_status, uint32_t flags, hipStream_t hs);
cst_status ans_decode_pt(const cst_model* model, cst_coder_config cfg, const uint32_t* d_words, const uint64_t* d_offsets,
                         size_t stride_words, size_t words_capacity, const uint32_t* d_n_words, int32_t* d_symbols, size_t n_streams,
                         size_t n_per_stream, uint64_t* d_state, uint32_t* d_n_words_out, int32_t* d_status, uint32_t flags,
                         hipStream_t hs);

// ... with jump points (Pos / Seek): the checkpointing encoder and the sub-lane decoder (k lanes per stream, two waves per SIMD)
cst_status ans_encode_pt_ckpt(const cst_model* model, const int32_t* d_symbols, size_t n_streams, size_t n_per_stream, uint32_t* d_words,
                              size_t stride_words, uint32_t* d_n_words, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state,
                              int32_t* d_status, hipStream_t hs);
bool pt_sub_usable(const cst_model* model, cst_coder_config cfg, size_t n_streams, size_t n_per_stream, size_t interval);
// (symbol_bytes = 1: d_symbols is an int8_t matrix in disguise and pt_sub_n8_usable has said yes)
cst_status ans_decode_pt_sub(const cst_model* model, const uint32_t* d_words, const uint64_t* d_offsets, size_t stride_words,
                             size_t words_capacity, size_t interval, const uint32_t* d_ckpt_pos, const uint64_t* d_ckpt_state,
                             int32_t* d_symbols, size_t n_streams, size_t n_per_stream, int32_t* d_status, hipStream_t hs, int symbol_bytes = 4);
// int8 symbol matrices inside the per-stream-table loops (round 6)
bool pt_n8_encode_usable(const cst_model* model, cst_coder_config cfg, cst_layout layout, const void* d_symbols, size_t n_streams, size_t n_per_stream,
                         size_t interval);
cst_status ans_encode_pt_ckpt_n8(const cst_model* model, const void* d_symbols8, size_t n_streams, size_t n_per_stream, uint32_t* d_words,
                                 size_t stride_words, uint32_t* d_n_words, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state,
                                 int32_t* d_status, hipStream_t hs);
bool pt_sub_n8_usable(const cst_model* model, cst_coder_config cfg, size_t n_streams, size_t n_per_stream, size_t interval, const void* d_symbols);

// (16,32) with packed words (CST_FLAG_PACKED_W16, cst_ans_w16pk.hip): two 16-bit words per 32-bit slot, every count in 16-bit words
struct AnsEncodeArgs;
struct AnsDecodeArgs;
bool w16pk_usable(const cst_model* model, cst_coder_config cfg, cst_layout layout);
cst_status ans_encode_w16pk(const AnsEncodeArgs& a, hipStream_t hs);
cst_status ans_decode_w16pk(const AnsDecodeArgs& a, hipStream_t hs);
bool w16pk_ckpt_usable(const cst_model* model, cst_coder_config cfg, cst_layout layout, size_t n_streams, size_t n_per_stream, size_t interval);
cst_status ans_encode_w16pk_ckpt(const AnsEncodeArgs& a, size_t interval, uint32_t* d_ckpt_pos, uint64_t* d_ckpt_state, hipStream_t hs);

// Where stream s's compressed words lie in the caller's buffer -- CHECKED.  The reference's decoder cannot read out of
// bounds (its backend is a Vec: src/backends.rs:495-507); here the counts and offsets are caller data, so a slice
// [off, off + n) that leaves the buffer of `capacity` words (0 = capacity unknown, the caller vouches) or, in slab form,
// its own slab of `stride` words (stride 0: every stream at offset 0, unchecked) is replaced by the empty slice at offset 0 and the stream reports
// CST_STREAM_INVALID_DATA (decoding an empty stream is well defined: stack.rs:1070-1100 with state 0).
struct WordSlice { uint64_t off; uint32_t n; bool bad; };
__device__ __forceinline__ WordSlice word_slice(const uint64_t* offsets, size_t stride, const uint32_t* n_words, size_t s,
                                                uint64_t capacity) {
    WordSlice w;
    w.off = offsets ? offsets[s] : (uint64_t)s * stride;
    w.n = n_words[s];
    w.bad = (!offsets && stride != 0 && w.n > stride) || (capacity != 0 && (w.off > capacity || w.n > capacity - w.off));
    if (w.bad) { w.off = 0; w.n = 0; }
    return w;
}

// ... with the count given (a jump point: the words in front of it)
__device__ __forceinline__ WordSlice word_slice_n(const uint64_t* offsets, size_t stride, uint32_t n, size_t s, uint64_t capacity) {
    WordSlice w;
    w.off = offsets ? offsets[s] : (uint64_t)s * stride;
    w.n = n;
    w.bad = (!offsets && stride != 0 && w.n > stride) || (capacity != 0 && (w.off > capacity || w.n > capacity - w.off));
    if (w.bad) { w.off = 0; w.n = 0; }
    return w;
}

inline bool config_supported(cst_coder_config c) {
    if (c.word_bits == 32 && c.state_bits == 64) return c.precision >= 1 && c.precision <= 24;
    if (c.word_bits == 16 && c.state_bits == 32) return c.precision >= 1 && c.precision <= 16;
    return false;
}

} // namespace cst
