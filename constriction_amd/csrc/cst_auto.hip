// cst_auto.hip -- the library's own choices: the A/B switches (read once) and the jump points a batch should carry.
//
// Jump points (the reference's Pos / Seek, src/stream/stack.rs:1107-1139, src/stream/queue.rs:172-196, 900-926) are side
// information: the words of a stream are the same with or without them.  What they buy on this chip is LANES: a decoder's
// per-symbol recurrence is a chain of dependent lookups, and a batch of 65 536 streams is one wave per SIMD -- every decoder but the
// int32 shared-table one (whose helper-free loop already sits at its issue floor) waits for its own lookups most of the time.  With
// k jump points per stream the same words decode on k lanes each: two or more waves per SIMD, the waits overlap.  Round 5 measured
// that (65 536 x 4096, decode ms plain -> with k):  one table per stream 0.718 -> 0.404 (k = 8);  range coder 0.390 -> 0.334 (P = 12),
// 0.527 -> 0.432 (P = 24);  int8 matrices 0.243 -> 0.168;  P = 24 entries 0.387 -> 0.335;  per-symbol Gaussians 3.39 -> 2.41 --
// and nothing for int32 symbols with one 12-bit table (0.267 -> 0.267).  The checkpointing encoders note the points on their way at
// the plain encoders' speed, 12 bytes per point (20 for the range coder).
//
// cst_jump_points_auto answers: how many symbols apart should the jump points of THIS batch lie (0 = carry none)?
#include <cstdlib>
#include <cstring>

#include "cst_ans_kernels.hpp"
#include "cst_range_kernels.hpp"

namespace cst {

static Knobs read_knobs() {
    Knobs k;
    auto env = [](const char* name) -> const char* { const char* v = getenv(name); return (v && *v) ? v : nullptr; };   // the ONE place
    k.no_pc_encoder = env("CST_NO_PC_ENCODER") != nullptr;
    k.no_n8 = env("CST_NO_N8") != nullptr;
    k.no_pc_wide = env("CST_NO_PC_WIDE") != nullptr;
    k.pc_combined = env("CST_PC_COMBINED") != nullptr;
    k.dq_decoder = env("CST_DQ_DECODER") != nullptr;
    if (const char* e = env("CST_SMALL_KERNELS")) {
        if (e[0] == '0') k.small_encoders = k.small_decoders = false;
        else if (e[0] == 'e') k.small_decoders = false;
        else if (e[0] == 'd') k.small_encoders = false;
    }
    if (const char* e = env("CST_PT_SUB_WAVES")) k.pt_sub_8_waves = e[0] == '8';
    if (const char* e = env("CST_SUB_ORDER")) k.sub_order_flat = e[0] == '0';
    if (const char* e = env("CST_LANE_GEO")) k.lane_geo = e[0] == 's' ? 2 : e[0] == 'b' ? 1 : 0;
    if (const char* e = env("CST_FUSED_MIN_STREAMS")) k.fused_min_streams = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = env("CST_AUTO_JUMP")) k.auto_jump = e[0] == '0' ? 0 : 1;
    if (const char* e = env("CST_RAGGED_GROUP")) k.ragged_group = e[0] == '8' ? 8 : e[0] == '3' ? 32 : 16;
    return k;
}

static Knobs g_knobs = read_knobs();          // when the library is loaded
const Knobs& knobs() { return g_knobs; }

static size_t pow2_at_least(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }

// k lanes per stream -> the interval, or 0: k is halved until the chunks are whole multiples of `granule` symbols of at least
// `min_interval`
static size_t interval_for(size_t n_per_stream, size_t k, size_t granule, size_t min_interval) {
    for (; k >= 2; k >>= 1) {
        if (n_per_stream % k != 0) continue;
        const size_t iv = n_per_stream / k;
        if (iv % granule == 0 && iv >= min_interval) return iv;
    }
    return 0;
}

constexpr size_t kMinInterval = 256;          // shorter chunks: the jump table and the decoders' start-up begin to show
constexpr size_t kMaxLanesPerStream = 64;

} // namespace cst

using namespace cst;

extern "C" {

void cst_debug_reload_knobs(void) { g_knobs = read_knobs(); }

size_t cst_jump_points_auto(const cst_model* model, cst_coder_config cfg, int32_t coder, int32_t symbol_bytes, const void* d_symbols,
                            size_t n_streams, size_t n_per_stream, cst_layout layout, const void* d_words, size_t stride_words) {
    if (!knobs().auto_jump || !model || n_streams == 0 || n_per_stream < 2 * kMinInterval) return 0;
    if (layout != CST_LAYOUT_STREAM_MAJOR || cfg.word_bits != 32 || cfg.state_bits != 64 || cfg.precision != model->precision) return 0;
    if (symbol_bytes != 1 && symbol_bytes != 2 && symbol_bytes != 4) return 0;
    if (model->d_symbol_of_index) return 0;                                     // (non-contiguous alphabets: the callers map to indices first)
    const size_t slots = (size_t)model->cu_count * kBlock;                       // lanes of one wave per SIMD
    // lanes per stream that fill the chip: the smallest power of two with more than one wave per SIMD ...
    size_t fill = n_streams > slots ? 1 : pow2_at_least((slots + n_streams) / n_streams);
    if (n_streams * fill <= slots) fill <<= 1;

    if (coder == CST_CODER_RANGE) {
        if (model->per_stream) return 0;
        RangeEncodeArgs e{};
        e.symbols = reinterpret_cast<const int32_t*>(symbol_bytes == 4 ? d_symbols : nullptr);      // (narrow matrices are widened into a fresh, aligned buffer)
        e.n_streams = n_streams; e.n_per_stream = n_per_stream; e.n_symbols = model->n_symbols; e.min_symbol = model->min_symbol;
        e.precision = model->precision; e.words = reinterpret_cast<uint32_t*>(const_cast<void*>(d_words)); e.stride_words = stride_words;
        if (!range_encode_ckpt_fast_usable(e, layout)) return 0;
        const size_t k = fill < kMaxLanesPerStream ? fill : kMaxLanesPerStream;
        const size_t iv = interval_for(n_per_stream, k, kTileSyms, kMinInterval);
        if (!iv) return 0;
        RangeDecodeArgs d{};
        d.n_streams = n_streams; d.n_per_stream = n_per_stream; d.dec_cp = model->d_dec_cp; d.dec_idx = model->d_dec_idx; d.cdf = model->d_cdf;
        d.bucket = model->d_bucket; d.bucket_bits = model->bucket_bits; d.n_symbols = model->n_symbols; d.precision = model->precision;
        d.interval = iv; d.n_chunks = n_per_stream / iv;
        return range_decode_sub_usable(d) ? iv : 0;
    }
    if (coder != CST_CODER_ANS) return 0;

    if (model->per_stream) {
        // one table per stream: the sub-lane decoder shares a stream's table among its lanes -- eight of them where the tables fit
        // (int8 matrices are read / written by the same kernels' int8 forms: round 6; int16 would convert -- no points for it)
        if ((symbol_bytes != 4 && symbol_bytes != 1) || model->n_tables != n_streams || !pt_usable(model, cfg, layout, n_per_stream)) return 0;
        if (symbol_bytes == 1 && (model->min_symbol < -128 || model->min_symbol + model->n_symbols - 1 > 127)) return 0;
        size_t k = fill < 8 ? 8 : fill;
        if (k > 16) k = 16;
        for (; k >= 2; k >>= 1) {
            const size_t iv = interval_for(n_per_stream, k, kTileSyms, kMinInterval);
            if (iv && n_per_stream / iv == k && pt_sub_usable(model, cfg, n_streams, n_per_stream, iv)) return iv;
        }
        return 0;
    }

    AnsEncodeArgs e{};
    e.symbols = reinterpret_cast<const int32_t*>(d_symbols); e.n_streams = n_streams; e.n_per_stream = n_per_stream; e.enc = model->d_enc;
    e.n_symbols = model->n_symbols; e.min_symbol = model->min_symbol; e.precision = model->precision;
    e.words = reinterpret_cast<uint32_t*>(const_cast<void*>(d_words)); e.stride_words = stride_words;
    // int32 symbols with one table of at most 12 bits: the plain decoder is at its issue floor at one wave per SIMD -- jump points only
    // to FILL the chip (fewer streams than lanes)
    size_t k = fill;
    if (symbol_bytes == 4 && model->precision <= 12) k = n_streams >= slots ? 1 : pow2_at_least((slots + n_streams - 1) / n_streams);
    if (k > kMaxLanesPerStream) k = kMaxLanesPerStream;
    if (k < 2) return 0;
    const size_t granule = symbol_bytes == 4 ? (size_t)kTileSyms : 128;          // narrow chunks: whole 128-byte lines of an int8 row
    const size_t iv = interval_for(n_per_stream, k, granule, kMinInterval);
    if (!iv) return 0;
    const bool fast = symbol_bytes == 4 ? pc_encode_ckpt_usable(e, cfg, layout, iv)
                    : symbol_bytes == 1 ? pc_n8_encode_ckpt_usable(e, cfg, layout, iv) : pc_n16_encode_ckpt_usable(e, cfg, layout, iv);
    if (!fast) return 0;                                                          // (the one-lane-per-stream checkpointing encoder is 5x slower)
    // the decoders of the virtual streams: P <= 12 needs the 2^P-entry tables, wider models the bucket entries
    if (model->precision <= 12) return (model->d_dec_cp && model->d_dec_idx && model->n_symbols <= 256) ? iv : 0;
    return (bucket16_usable(model->n_symbols, model->precision) && model->d_bucket && model->d_cdf) ? iv : 0;
}

size_t cst_jump_points_auto_gaussian(cst_coder_config cfg, int32_t coder, size_t n_streams, size_t n_per_stream, cst_layout layout) {
    if (!knobs().auto_jump || (coder != CST_CODER_ANS && coder != CST_CODER_RANGE) || layout != CST_LAYOUT_STREAM_MAJOR) return 0;
    if (n_streams < knobs().fused_min_streams || n_streams < 16384 || n_per_stream < 2 * kMinInterval) return 0;     // (the fused encoder notes the points)
    if (!config_supported(cfg)) return 0;
    int cus = 256;
    { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount; }
    const size_t slots = (size_t)cus * kBlock;
    if (n_streams > slots) return 0;                                              // already the small geometry, two waves per SIMD
    size_t k = pow2_at_least((slots + n_streams) / n_streams);
    if (n_streams * k <= slots) k <<= 1;
    return interval_for(n_per_stream, k, 16, kMinInterval);
}

} // extern "C"
