// cst_model.hip -- entropy-model tables: device construction (bit-exact f64) and host-side derivation
// of the encoder / decoder images the coder kernels consume.
#include <algorithm>
#include <vector>
#include <new>

#include "cst_common.hpp"
#include "cst_math.hpp"

namespace cst {

// cdf[table][i] = left cumulative of symbol index i (i = n gives 2^P); one thread per entry.
// This is the tabulation of LeakilyQuantizedDistribution::left_cumulative_and_probability
// (src/stream/model/quantize.rs:525-568) -- NOT of symbol_table() (SURVEY.md hazard 1).
__global__ void gaussian_cdf_kernel(int P, int32_t lo, int32_t n, const double* __restrict__ means,
                                    const double* __restrict__ stds, double mean0, double std0, size_t n_tables,
                                    uint32_t* __restrict__ cdf) {
    __shared__ double2 erf_tab[kErfTabEntries];
    erf_tab_fill(erf_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)n + 1;
    if (gid >= n_tables * per) return;
    const size_t tbl = gid / per;
    const int32_t i = (int32_t)(gid - tbl * per);
    const double mu = means ? means[tbl] : mean0;
    const double sd = stds ? stds[tbl] : std0;
    cdf[gid] = leaky_gaussian_left_quick(i, lo, n, P, 32, mu, sd, erf_tab);
}

// 16-bit per-stream cdf rows for the LDS-resident per-stream models (values modulo 2^16).  Also validates
// every table: status[tbl] = 1 if some probability is zero (quantize.rs:562-565 panics) or a parameter is invalid.
__global__ void cdf_to_u16_kernel(const uint32_t* __restrict__ cdf, size_t n_tables, int32_t n, int32_t stride16,
                                  uint16_t* __restrict__ out, const double* __restrict__ means,
                                  const double* __restrict__ stds, int32_t* __restrict__ bad) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_tables * (size_t)stride16) return;
    const size_t tbl = gid / stride16;
    const int32_t i = (int32_t)(gid - tbl * stride16);
    const uint32_t* row = cdf + tbl * ((size_t)n + 1);
    out[gid] = (i <= n) ? (uint16_t)row[i] : (uint16_t)0xffff;
    if (i < n && row[i + 1] <= row[i]) atomicOr(&bad[tbl], 1);
    if (i == 0) {
        const double m = means[tbl], s = stds[tbl];
        if (!(s > 0.0 && s <= 1.7976931348623157e308 && m == m && m <= 1.7976931348623157e308 && m >= -1.7976931348623157e308))
            atomicOr(&bad[tbl], 1);
    }
}

// ---- compact per-stream rows (cst::PtMeta, cst_ans_pt.hip) ----

// pass 1, one thread per table: the unit runs at both ends -> (a, m); entries of the run-length decoder row -> m_dec
__global__ void pt_measure_kernel(const uint32_t* __restrict__ cdf, size_t n_tables, int32_t n, PtMeta* __restrict__ meta) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_tables) return;
    const uint32_t* row = cdf + s * ((size_t)n + 1);
    int r = 0;                                   // leading run: p[0..r) == 1
    while (r < n && row[r + 1] - row[r] == 1u) ++r;
    int t = 0;                                   // trailing run: p[n-t..n) == 1
    while (t < n && row[n - t] - row[n - t - 1] == 1u) ++t;
    int a = r > 0 ? r - 1 : 0;
    const int b = t > 0 ? n - t : n - 1;
    if (a > b) a = b;                            // the runs overlap (every probability is 1): one entry is enough
    int cnt = 0;
    for (int i = 0; i < n;) {
        int j = i + 1;
        if (row[i + 1] - row[i] == 1u) while (j < n && row[j + 1] - row[j] == 1u) ++j;
        cnt += (j - i >= 2) ? 1 : (j - i);
        i = j;
    }
    PtMeta mt{};
    mt.a = (uint16_t)a; mt.m = (uint16_t)(b - a + 1); mt.m_dec = (uint16_t)cnt;
    meta[s] = mt;
}

// pass 2, one workgroup per block of kBlock tables: row offsets inside the block, block sizes
__global__ __launch_bounds__(256) void pt_offsets_kernel(PtMeta* __restrict__ meta, size_t n_tables, uint32_t* __restrict__ enc_size,
                                                          uint32_t* __restrict__ dec_size) {
    __shared__ uint32_t se[256], sd[256];
    const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t le = s < n_tables ? (uint32_t)meta[s].m + 1u : 0u;
    const uint32_t ld = s < n_tables ? ((uint32_t)meta[s].m_dec + kPtRowPad + 3u) & ~3u : 0u;   // rows start on 16 bytes
    se[threadIdx.x] = le; sd[threadIdx.x] = ld;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t ve = threadIdx.x >= (unsigned)d ? se[threadIdx.x - d] : 0u, vd = threadIdx.x >= (unsigned)d ? sd[threadIdx.x - d] : 0u;
        __syncthreads();
        se[threadIdx.x] += ve; sd[threadIdx.x] += vd;
        __syncthreads();
    }
    if (s < n_tables) { meta[s].enc_off = se[threadIdx.x] - le; meta[s].dec_off = sd[threadIdx.x] - ld; }
    if (threadIdx.x == 255) { enc_size[blockIdx.x] = se[255]; dec_size[blockIdx.x] = sd[255]; }
}

// decoder-row entries of every aligned group of 64 tables (the sub-lane decoder stages 64, 128 or 256 streams per workgroup): the largest
__global__ void pt_group_max_kernel(const PtMeta* __restrict__ meta, size_t n_tables, const uint32_t* __restrict__ dec_size, uint32_t* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t s0 = g * 64;
    if (s0 >= n_tables) return;
    const size_t s1 = s0 + 64 < n_tables ? s0 + 64 : n_tables;
    // rows lie back to back inside a block of kBlock tables (dec_off is relative to it); a group never straddles two blocks
    const uint32_t end = (s1 % kBlock != 0 && s1 < n_tables) ? meta[s1].dec_off : dec_size[s0 / kBlock];
    atomicMax(out, end - meta[s0].dec_off);
}

// pass 3, one thread per table: the rows and the quantile bucket index of the decoder
__global__ void pt_fill_kernel(const uint32_t* __restrict__ cdf, size_t n_tables, int32_t n, int P, const PtMeta* __restrict__ meta,
                               const uint32_t* __restrict__ enc_base, const uint32_t* __restrict__ dec_base, uint16_t* __restrict__ enc,
                               uint32_t* __restrict__ dec, uint8_t* __restrict__ l1) {
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_tables) return;
    const uint32_t* row = cdf + s * ((size_t)n + 1);
    const PtMeta mt = meta[s];
    uint16_t* e = enc + (size_t)enc_base[s / kBlock] + mt.enc_off;
    for (uint32_t j = 0; j <= mt.m; ++j) e[j] = (uint16_t)row[mt.a + j];
    // decoder row and, alongside, the bucket index: bucket k = quantiles [k * w, (k + 1) * w) -> a QUARTER of the position of
    // the entry that holds k * w (the decoder reads eight entries from a 16-byte aligned address with two ds_read_b128:
    // misaligned LDS reads are 5x slower, and two aligned 16-byte reads cost 23 LDS cycles where ds_read2_b64 + ds_read_b64
    // of six entries cost 37 -- scripts/microbench/lds_tput.hip)
    uint32_t* d = dec + (size_t)dec_base[s / kBlock] + mt.dec_off;
    uint8_t* b = l1 + s * kPtBuckets;
    const int shift = P - kPtBucketBits;
    uint32_t pos = 0, k = 0;
    for (int i = 0; i < n;) {
        int j = i + 1;
        const uint32_t c = row[i], p = row[i + 1] - c;
        if (p == 1u) while (j < n && row[j + 1] - row[j] == 1u) ++j;
        if (j - i >= 2) {
            d[pos] = (c << 20) | (kPtRunMark << 8) | (uint32_t)i;
        } else {
            j = i + 1;
            d[pos] = (c << 20) | ((p - 1u) << 8) | (uint32_t)i;
        }
        const uint32_t c_end = row[j];                        // quantiles [c, c_end) belong to this entry
        while (k < (uint32_t)kPtBuckets && (k << shift) < c_end) b[k++] = (uint8_t)(pos >> 2);
        ++pos;
        i = j;
    }
    for (uint32_t j = pos; j < ((pos + kPtRowPad + 3u) & ~3u); ++j) d[j] = 0xffffffffu;   // kPtRowPad sentinels or up to three more: whole quads
}

// Builds the compact image of a per-stream model (8 <= P <= 12, n <= 256).  Failure to allocate only leaves pt_ok
// false: the coder then uses the full-row kernels of cst_ans_ps.hip.
static void build_pt_image(cst_model* m, hipStream_t hs) {
    const int n = m->n_symbols, P = m->precision;
    if (P > 12 || P < 8 || n > 256 || m->n_tables > 0x7fffffffull / 520) return;
    const size_t nt = m->n_tables, n_blocks = (nt + kBlock - 1) / kBlock;
    uint32_t* d_size = nullptr;
    std::vector<uint32_t> h_size(2 * n_blocks), h_base(2 * (n_blocks + 1), 0);
    bool ok = hipMalloc(&m->d_pt_meta, sizeof(PtMeta) * nt) == hipSuccess && hipMalloc(&d_size, 8 * n_blocks) == hipSuccess &&
              hipMalloc(&m->d_pt_block_base, 8 * (n_blocks + 1)) == hipSuccess && hipMalloc(&m->d_pt_l1, nt * kPtBuckets) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(pt_measure_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, hs, (const uint32_t*)m->d_cdf, nt, n, m->d_pt_meta);
        hipLaunchKernelGGL(pt_offsets_kernel, dim3((unsigned)n_blocks), dim3(256), 0, hs, m->d_pt_meta, nt, d_size, d_size + n_blocks);
        ok = hipMemcpyAsync(h_size.data(), d_size, 8 * n_blocks, hipMemcpyDeviceToHost, hs) == hipSuccess &&
             hipStreamSynchronize(hs) == hipSuccess;
    }
    if (ok) {
        uint32_t* d_max = nullptr;
        ok = hipMalloc(&d_max, 4) == hipSuccess && hipMemsetAsync(d_max, 0, 4, hs) == hipSuccess;
        if (ok) {
            const size_t groups = (nt + 63) / 64;
            hipLaunchKernelGGL(pt_group_max_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, hs, (const PtMeta*)m->d_pt_meta, nt,
                               (const uint32_t*)(d_size + n_blocks), d_max);
            ok = hipMemcpyAsync(&m->pt_max_dec64, d_max, 4, hipMemcpyDeviceToHost, hs) == hipSuccess && hipStreamSynchronize(hs) == hipSuccess;
        }
        if (d_max) (void)hipFree(d_max);
    }
    uint32_t* eb = h_base.data(), *db = h_base.data() + n_blocks + 1;
    if (ok) {
        for (size_t k = 0; k < n_blocks; ++k) {
            eb[k + 1] = eb[k] + ((h_size[k] + 1u) & ~1u);   // blocks of 16-bit rows start on a word
            db[k + 1] = db[k] + h_size[n_blocks + k];
            if (h_size[k] > m->pt_max_enc) m->pt_max_enc = h_size[k];
            if (h_size[n_blocks + k] > m->pt_max_dec) m->pt_max_dec = h_size[n_blocks + k];
        }
        ok = hipMalloc(&m->d_pt_enc, 2 * ((size_t)eb[n_blocks] + 8)) == hipSuccess && hipMalloc(&m->d_pt_dec, 4 * ((size_t)db[n_blocks] + 4)) == hipSuccess &&
             hipMemcpyAsync(m->d_pt_block_base, h_base.data(), 8 * (n_blocks + 1), hipMemcpyHostToDevice, hs) == hipSuccess;
    }
    if (ok) {
        hipLaunchKernelGGL(pt_fill_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, hs, (const uint32_t*)m->d_cdf, nt, n, P,
                           (const PtMeta*)m->d_pt_meta, (const uint32_t*)m->d_pt_block_base, (const uint32_t*)(m->d_pt_block_base + n_blocks + 1),
                           m->d_pt_enc, m->d_pt_dec, m->d_pt_l1);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(hs) == hipSuccess;   // h_base must outlive the copy
    }
    if (d_size) (void)hipFree(d_size);
    m->pt_ok = ok;
}

// ---- non-contiguous alphabets: symbol <-> index ----
__global__ void symbols_to_indices_kernel(const int32_t* __restrict__ sorted, const int32_t* __restrict__ sorted_index, int32_t n,
                                          const int32_t* __restrict__ in, int32_t* __restrict__ out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int32_t v = in[i];
    int32_t lo = 0, hi = n;                        // first position with sorted[pos] >= v
    while (lo < hi) {
        const int32_t mid = lo + (hi - lo) / 2;
        if (sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    out[i] = (lo < n && sorted[lo] == v) ? sorted_index[lo] : n;     // n = no such symbol: the coder reports it as impossible
}

__global__ void indices_to_symbols_kernel(const int32_t* __restrict__ symbol_of_index, int32_t n, const int32_t* __restrict__ in,
                                          int32_t* __restrict__ out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t k = (uint32_t)in[i];
    out[i] = k < (uint32_t)n ? symbol_of_index[k] : 0;
}

__global__ void debug_erf_kernel(const double* __restrict__ x, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = erf_exact(x[i]);
}

__global__ void debug_erf_tab_kernel(const double* __restrict__ x, double* __restrict__ out, size_t n) {
    __shared__ double2 tab[kErfTabEntries];
    erf_tab_fill(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = erf_exact_tab(x[i], tab);
}

// which: 0 erf_fast_poly(x) -> out; 1 |erf_fast_poly - erf_exact_tab| -> out
__global__ void debug_erf_fast_kernel(int which, const double* __restrict__ x, double* __restrict__ out, size_t n) {
    __shared__ double2 tab[kErfTabEntries];
    erf_tab_fill(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double f = erf_fast_poly(x[i], tab);
    out[i] = which == 0 ? f : fabs(f - erf_exact_tab(x[i], tab));
}

// the quick (fast erf + exact fallback) and the plain exact left cumulatives side by side: counts[0] += mismatches,
// counts[1] += exact fallbacks taken
__global__ void debug_left_quick_kernel(int P, int32_t lo, int32_t n_sym, const int32_t* __restrict__ idx, const double* __restrict__ means,
                                        const double* __restrict__ stds, size_t n, unsigned long long* __restrict__ counts) {
    __shared__ double2 tab[kErfTabEntries];
    erf_tab_fill(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t fallbacks = 0;
    const uint32_t quick = leaky_gaussian_left_quick(idx[i], lo, n_sym, P, 32, means[i], stds[i], tab, &fallbacks);
    const uint32_t exact = leaky_gaussian_left<true>(idx[i], lo, n_sym, P, 32, means[i], stds[i], tab);
    if (quick != exact) atomicAdd(&counts[0], 1ull);
    if (fallbacks) atomicAdd(&counts[1], (unsigned long long)fallbacks);
}

__global__ void debug_lcp_kernel(int P, int prob_bits, int32_t lo, int32_t hi, const int32_t* __restrict__ sym,
                                 const double* __restrict__ means, const double* __restrict__ stds,
                                 uint32_t* __restrict__ left, uint32_t* __restrict__ prob, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t l = 0, p = 0;
    if (!leaky_gaussian_lcp(sym[i], lo, hi, P, prob_bits, means[i], stds[i], l, p)) { l = 0xffffffffu; p = 0; }
    left[i] = l; prob[i] = p;
}

static bool cdf_valid(const uint32_t* cdf, int n, int P) {
    if (cdf[0] != 0u) return false;
    if (cdf[n] != (uint32_t)((uint64_t)1 << P)) return false;         // (2^32 wraps to 0: Probability = u32 at PRECISION = 32)
    for (int i = 0; i + 1 < n; ++i)
        if (cdf[i + 1] <= cdf[i]) return false;
    return P == 32 ? true : cdf[n] > cdf[n - 1];
}

// Derive the encoder entries, decoder lookup and bucket index from a validated host cdf and
// upload them.  (lookup_contiguous.rs:611-636 builds the same quantile -> index map.)
static cst_status upload_shared_tables(cst_model* m, const uint32_t* cdf) {
    const int n = m->n_symbols, P = m->precision;
    std::vector<EncEntry> enc((size_t)n);
    for (int i = 0; i < n; ++i) {
        const uint32_t p = cdf[i + 1] - cdf[i];
        const unsigned __int128 one = (unsigned __int128)1 << 64;
        const uint64_t mm = p == 1 ? ~0ull : (uint64_t)(one / p);
        enc[i] = EncEntry{cdf[i], p, (uint32_t)mm, (uint32_t)(mm >> 32)};
    }
    CST_HIP_TRY(hipMalloc(&m->d_enc, sizeof(EncEntry) * (size_t)n));
    CST_HIP_TRY(hipMemcpy(m->d_enc, enc.data(), sizeof(EncEntry) * (size_t)n, hipMemcpyHostToDevice));

    if (P > 24) return CST_OK;     // 24 < P <= 32 is coded by cst_ans_generic.hip from the cumulatives alone
    if (P <= 16) {
        const size_t total = (size_t)1 << P;
        std::vector<uint32_t> cp(total);
        std::vector<uint16_t> ix(total);
        int i = 0;
        for (size_t q = 0; q < total; ++q) {
            while (cdf[i + 1] <= q) ++i;
            cp[q] = pack_cp(cdf[i], cdf[i + 1] - cdf[i]);
            ix[q] = (uint16_t)i;
        }
        CST_HIP_TRY(hipMalloc(&m->d_dec_cp, 4 * total));
        CST_HIP_TRY(hipMemcpy(m->d_dec_cp, cp.data(), 4 * total, hipMemcpyHostToDevice));
        CST_HIP_TRY(hipMalloc(&m->d_dec_idx, 2 * total));
        CST_HIP_TRY(hipMemcpy(m->d_dec_idx, ix.data(), 2 * total, hipMemcpyHostToDevice));
    }
    // bucket index for the generic (any P) decoder
    m->bucket_bits = P < 11 ? P : 11;
    {
        const size_t nb = (size_t)1 << m->bucket_bits;
        const int shift = P - m->bucket_bits;
        std::vector<uint16_t> b(nb);
        int i = 0;
        for (size_t k = 0; k < nb; ++k) {
            const uint64_t q = (uint64_t)k << shift;
            while (cdf[i + 1] <= q) ++i;
            b[k] = (uint16_t)(i > 0xffff ? 0xffff : i);
        }
        CST_HIP_TRY(hipMalloc(&m->d_bucket, 2 * nb));
        CST_HIP_TRY(hipMemcpy(m->d_bucket, b.data(), 2 * nb, hipMemcpyHostToDevice));
    }
    return CST_OK;
}

static cst_status check_model_args(int32_t P, int64_t n_symbols) {
    if (P < 1 || P > 24) return CST_ERR_INVALID_ARGUMENT;
    // LeakyQuantizer::new panics on < 2 symbols or more than 2^P (quantize.rs:292-301)
    if (n_symbols < 2 || n_symbols > ((int64_t)1 << P) || n_symbols > 65536) return CST_ERR_MODEL;
    return CST_OK;
}

} // namespace cst

using namespace cst;

extern "C" {

cst_status cst_model_create_table(int32_t precision, int32_t min_symbol, int32_t n_symbols, const uint32_t* h_cdf,
                                  cst_model** out) {
    if (!out || !h_cdf) return CST_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    // (a table may have up to 32 bits of precision -- the u32 / PRECISION = 32 corner of the reference's grid, stack.rs:1293-1296;
    //  the model families below stop at 24, the Python API's precision)
    if (precision > 24 && precision <= 32) { if (n_symbols < 2 || n_symbols > 65536) return CST_ERR_MODEL; }
    else if (cst_status st = check_model_args(precision, n_symbols)) return st;
    if ((int64_t)min_symbol + n_symbols - 1 > INT32_MAX) return CST_ERR_INVALID_ARGUMENT;
    if (!cdf_valid(h_cdf, n_symbols, precision)) return CST_ERR_MODEL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return CST_ERR_NO_DEVICE;
    cst_model* m = new (std::nothrow) cst_model();
    if (!m) return CST_ERR_OUT_OF_MEMORY;
    m->precision = precision; m->min_symbol = min_symbol; m->n_symbols = n_symbols; m->n_tables = 1;
    hipGetDevice(&m->device); (void)hipDeviceGetAttribute(&m->cu_count, hipDeviceAttributeMultiprocessorCount, m->device);
    const size_t bytes = 4 * ((size_t)n_symbols + 1);
    hipError_t e = hipMalloc(&m->d_cdf, bytes);
    if (e == hipSuccess) e = hipMemcpy(m->d_cdf, h_cdf, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_hip_error(e, "model upload"); cst_model_destroy(m); return CST_ERR_HIP; }
    if (cst_status st = upload_shared_tables(m, h_cdf)) { cst_model_destroy(m); return st; }
    *out = m;
    return CST_OK;
}

cst_status cst_model_create_gaussian(int32_t precision, int32_t min_symbol, int32_t max_symbol, double mean, double std,
                                     void* stream, cst_model** out) {
    if (!out) return CST_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (max_symbol <= min_symbol) return CST_ERR_MODEL; // degenerate support, quantize.rs:292-294
    const int64_t n64 = (int64_t)max_symbol - min_symbol + 1;
    if (cst_status st = check_model_args(precision, n64)) return st;
    // `assert!(std > 0.0)` in the reference's constructor (pybindings/stream/model.rs:654-657)
    if (!(std > 0.0) || !(mean == mean) || std > 1.7976931348623157e308 || mean > 1.7976931348623157e308 ||
        mean < -1.7976931348623157e308)
        return CST_ERR_MODEL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return CST_ERR_NO_DEVICE;
    const int n = (int)n64;
    hipStream_t hs = (hipStream_t)stream;
    cst_model* m = new (std::nothrow) cst_model();
    if (!m) return CST_ERR_OUT_OF_MEMORY;
    m->precision = precision; m->min_symbol = min_symbol; m->n_symbols = n; m->n_tables = 1;
    hipGetDevice(&m->device); (void)hipDeviceGetAttribute(&m->cu_count, hipDeviceAttributeMultiprocessorCount, m->device);
    std::vector<uint32_t> h((size_t)n + 1);
    hipError_t e = hipMalloc(&m->d_cdf, 4 * ((size_t)n + 1));
    if (e == hipSuccess) {
        const int threads = 256, blocks = (n + 1 + threads - 1) / threads;
        hipLaunchKernelGGL(gaussian_cdf_kernel, dim3(blocks), dim3(threads), 0, hs, precision, min_symbol, n,
                           (const double*)nullptr, (const double*)nullptr, mean, std, (size_t)1, m->d_cdf);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), m->d_cdf, 4 * ((size_t)n + 1), hipMemcpyDeviceToHost, hs);
    if (e == hipSuccess) e = hipStreamSynchronize(hs);
    if (e != hipSuccess) { set_hip_error(e, "gaussian table"); cst_model_destroy(m); return CST_ERR_HIP; }
    // "Invalid underlying continuous probability distribution" panic in the reference (quantize.rs:562-565)
    if (!cdf_valid(h.data(), n, precision)) { cst_model_destroy(m); return CST_ERR_MODEL; }
    if (cst_status st = upload_shared_tables(m, h.data())) { cst_model_destroy(m); return st; }
    *out = m;
    return CST_OK;
}

cst_status cst_model_create_gaussian_per_stream(int32_t precision, int32_t min_symbol, int32_t max_symbol,
                                                const double* d_means, const double* d_stds, size_t n_streams,
                                                void* stream, cst_model** out) {
    if (!out || !d_means || !d_stds || n_streams == 0) return CST_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (max_symbol <= min_symbol) return CST_ERR_MODEL;
    const int64_t n64 = (int64_t)max_symbol - min_symbol + 1;
    if (cst_status st = check_model_args(precision, n64)) return st;
    if (precision > 16) return CST_ERR_INVALID_ARGUMENT; // per-stream tables are kept as 16-bit rows
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return CST_ERR_NO_DEVICE;
    const int n = (int)n64;
    hipStream_t hs = (hipStream_t)stream;
    cst_model* m = new (std::nothrow) cst_model();
    if (!m) return CST_ERR_OUT_OF_MEMORY;
    m->precision = precision; m->min_symbol = min_symbol; m->n_symbols = n; m->n_tables = n_streams; m->per_stream = true;
    hipGetDevice(&m->device); (void)hipDeviceGetAttribute(&m->cu_count, hipDeviceAttributeMultiprocessorCount, m->device);
    m->cdf16_stride = 8;
    while (m->cdf16_stride < n + 1) m->cdf16_stride <<= 1;   // power of two (rows are rotated per lane in LDS)
    const size_t total = n_streams * ((size_t)n + 1);
    hipError_t e = hipMalloc(&m->d_cdf, 4 * total);
    if (e == hipSuccess) e = hipMalloc(&m->d_cdf16, 2 * n_streams * (size_t)m->cdf16_stride);
    if (e == hipSuccess) {
        const int threads = 256;
        const size_t blocks = (total + threads - 1) / threads;
        hipLaunchKernelGGL(gaussian_cdf_kernel, dim3((unsigned)blocks), dim3(threads), 0, hs, precision, min_symbol, n,
                           d_means, d_stds, 0.0, 1.0, n_streams, m->d_cdf);
        const size_t total16 = n_streams * (size_t)m->cdf16_stride;
        int32_t* d_bad = nullptr;
        e = hipMallocAsync((void**)&d_bad, 4 * n_streams, hs);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 4 * n_streams, hs);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(cdf_to_u16_kernel, dim3((unsigned)((total16 + threads - 1) / threads)), dim3(threads), 0, hs,
                               (const uint32_t*)m->d_cdf, n_streams, n, m->cdf16_stride, m->d_cdf16, d_means, d_stds, d_bad);
            e = hipGetLastError();
        }
        // any invalid table makes the whole model invalid (the reference panics when such a model is constructed)
        if (e == hipSuccess) {
            std::vector<int32_t> h_bad(n_streams);
            e = hipMemcpyAsync(h_bad.data(), d_bad, 4 * n_streams, hipMemcpyDeviceToHost, hs);
            if (e == hipSuccess) e = hipStreamSynchronize(hs);
            if (e == hipSuccess) for (size_t i = 0; i < n_streams; ++i) if (h_bad[i]) { (void)hipFree(d_bad); cst_model_destroy(m); return CST_ERR_MODEL; }
        }
        if (d_bad) (void)hipFreeAsync(d_bad, hs);
    }
    if (e == hipSuccess) {   // reciprocal table floor(2^64 / p) for p in [1, 2^P)
        const size_t np = (size_t)1 << precision;
        std::vector<uint64_t> rec(np, 0);
        rec[1] = ~0ull;
        for (size_t p = 2; p < np; ++p) rec[p] = (uint64_t)((((unsigned __int128)1) << 64) / p);
        e = hipMalloc(&m->d_recip, 8 * np);
        if (e == hipSuccess) e = hipMemcpy(m->d_recip, rec.data(), 8 * np, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { set_hip_error(e, "per-stream gaussian tables"); cst_model_destroy(m); return CST_ERR_HIP; }
    build_pt_image(m, hs);
    *out = m;
    return CST_OK;
}

cst_status cst_model_create_table_noncontiguous(int32_t precision, int32_t n_symbols, const int32_t* h_symbols, const uint32_t* h_cdf,
                                               cst_model** out) {
    if (!out || !h_symbols || !h_cdf) return CST_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (n_symbols < 2) return CST_ERR_MODEL;
    std::vector<int32_t> order((size_t)n_symbols);
    for (int32_t i = 0; i < n_symbols; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return h_symbols[a] < h_symbols[b]; });
    std::vector<int32_t> sorted((size_t)n_symbols);
    for (int32_t i = 0; i < n_symbols; ++i) {
        sorted[(size_t)i] = h_symbols[order[(size_t)i]];
        if (i > 0 && sorted[(size_t)i] == sorted[(size_t)i - 1]) return CST_ERR_MODEL;     // symbols must be distinct (non_contiguous.rs: HashMap insert)
    }
    cst_model* m = nullptr;
    if (cst_status st = cst_model_create_table(precision, 0, n_symbols, h_cdf, &m)) return st;   // the model proper works on indices 0..n-1
    const size_t bytes = 4 * (size_t)n_symbols;
    hipError_t e = hipMalloc(&m->d_symbol_of_index, bytes);
    if (e == hipSuccess) e = hipMalloc(&m->d_sorted_symbols, bytes);
    if (e == hipSuccess) e = hipMalloc(&m->d_sorted_index, bytes);
    if (e == hipSuccess) e = hipMemcpy(m->d_symbol_of_index, h_symbols, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_sorted_symbols, sorted.data(), bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_sorted_index, order.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_hip_error(e, "non-contiguous alphabet"); cst_model_destroy(m); return CST_ERR_HIP; }
    *out = m;
    return CST_OK;
}

cst_status cst_symbols_to_indices(const cst_model* m, const int32_t* d_symbols, size_t count, int32_t* d_indices, void* stream) {
    if (!m || !m->d_sorted_symbols || (count > 0 && (!d_symbols || !d_indices))) return CST_ERR_INVALID_ARGUMENT;
    if (count == 0) return CST_OK;
    hipLaunchKernelGGL(symbols_to_indices_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)m->d_sorted_symbols, (const int32_t*)m->d_sorted_index, m->n_symbols, d_symbols, d_indices, count);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status cst_indices_to_symbols(const cst_model* m, const int32_t* d_indices, size_t count, int32_t* d_symbols, void* stream) {
    if (!m || !m->d_symbol_of_index || (count > 0 && (!d_symbols || !d_indices))) return CST_ERR_INVALID_ARGUMENT;
    if (count == 0) return CST_OK;
    hipLaunchKernelGGL(indices_to_symbols_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const int32_t*)m->d_symbol_of_index, m->n_symbols, d_indices, d_symbols, count);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status cst_model_destroy(cst_model* m) {
    if (!m) return CST_OK;
    hipFree(m->d_cdf); hipFree(m->d_enc); hipFree(m->d_dec_cp); hipFree(m->d_dec_idx); hipFree(m->d_bucket);
    hipFree(m->d_cdf16); hipFree(m->d_recip);
    hipFree(m->d_symbol_of_index); hipFree(m->d_sorted_symbols); hipFree(m->d_sorted_index);
    hipFree(m->d_pt_meta); hipFree(m->d_pt_enc); hipFree(m->d_pt_dec); hipFree(m->d_pt_l1); hipFree(m->d_pt_block_base);
    delete m;
    return CST_OK;
}

int32_t cst_model_precision(const cst_model* m) { return m ? m->precision : 0; }
int32_t cst_model_min_symbol(const cst_model* m) { return m ? m->min_symbol : 0; }
int32_t cst_model_n_symbols(const cst_model* m) { return m ? m->n_symbols : 0; }
size_t cst_model_n_tables(const cst_model* m) { return m ? m->n_tables : 0; }

cst_status cst_model_get_cdf(const cst_model* m, size_t index, uint32_t* h_cdf, void* stream) {
    if (!m || !h_cdf || index >= m->n_tables) return CST_ERR_INVALID_ARGUMENT;
    const size_t per = (size_t)m->n_symbols + 1;
    CST_HIP_TRY(hipMemcpyAsync(h_cdf, m->d_cdf + index * per, 4 * per, hipMemcpyDeviceToHost, (hipStream_t)stream));
    CST_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return CST_OK;
}

cst_status cst_model_copy_cdfs(const cst_model* m, size_t first, size_t count, uint32_t* d_cdfs, void* stream) {
    if (!m || first > m->n_tables || count > m->n_tables - first) return CST_ERR_INVALID_ARGUMENT;
    if (count == 0) return CST_OK;
    if (!d_cdfs) return CST_ERR_INVALID_ARGUMENT;
    const size_t per = (size_t)m->n_symbols + 1;
    CST_HIP_TRY(hipMemcpyAsync(d_cdfs, m->d_cdf + first * per, 4 * per * count, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CST_OK;
}

cst_status cst_debug_erf_tab(const double* d_x, double* d_out, size_t n, void* stream) {
    if (!d_x || !d_out) return CST_ERR_INVALID_ARGUMENT;
    if (n == 0) return CST_OK;
    hipLaunchKernelGGL(debug_erf_tab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, d_out, n);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status cst_debug_erf(const double* d_x, double* d_out, size_t n, void* stream) {
    if (!d_x || !d_out) return CST_ERR_INVALID_ARGUMENT;
    if (n == 0) return CST_OK;
    hipLaunchKernelGGL(debug_erf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, d_out,
                       n);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status cst_debug_erf_fast(int32_t which, const double* d_x, double* d_out, size_t n, void* stream) {
    if (!d_x || !d_out || which < 0 || which > 1) return CST_ERR_INVALID_ARGUMENT;
    if (n == 0) return CST_OK;
    hipLaunchKernelGGL(debug_erf_fast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, which, d_x, d_out, n);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status cst_debug_gaussian_left_quick(int32_t precision, int32_t min_symbol, int32_t max_symbol, const int32_t* d_index,
                                         const double* d_means, const double* d_stds, size_t n, uint64_t* d_counts, void* stream) {
    if (!d_index || !d_means || !d_stds || !d_counts || precision < 1 || precision > kQuickMaxPrecision || max_symbol <= min_symbol) return CST_ERR_INVALID_ARGUMENT;
    if (n == 0) return CST_OK;
    hipLaunchKernelGGL(debug_left_quick_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, precision, min_symbol,
                       (int32_t)((int64_t)max_symbol - min_symbol + 1), d_index, d_means, d_stds, n, reinterpret_cast<unsigned long long*>(d_counts));
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

cst_status cst_debug_gaussian_lcp(int32_t precision, int32_t prob_bits, int32_t min_symbol, int32_t max_symbol,
                                  const int32_t* d_symbols, const double* d_means, const double* d_stds,
                                  uint32_t* d_left, uint32_t* d_prob, size_t n, void* stream) {
    if (!d_symbols || !d_means || !d_stds || !d_left || !d_prob) return CST_ERR_INVALID_ARGUMENT;
    if (precision < 1 || precision > prob_bits || (prob_bits != 16 && prob_bits != 32)) return CST_ERR_INVALID_ARGUMENT;
    if (n == 0) return CST_OK;
    hipLaunchKernelGGL(debug_lcp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, precision,
                       prob_bits, min_symbol, max_symbol, d_symbols, d_means, d_stds, d_left, d_prob, n);
    CST_HIP_TRY(hipGetLastError());
    return CST_OK;
}

} // extern "C"
