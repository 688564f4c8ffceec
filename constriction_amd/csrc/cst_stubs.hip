// temporary: entry points not implemented yet
#include "cst_common.hpp"
extern "C" {
cst_status cst_ans_encode_gaussian_batch(cst_coder_config, int32_t, int32_t, const int32_t*, const double*, const double*,
                                         size_t, size_t, cst_layout, uint32_t*, size_t, uint32_t*, uint64_t*, int32_t*,
                                         uint32_t, void*) { return CST_ERR_INVALID_ARGUMENT; }
cst_status cst_ans_decode_gaussian_batch(cst_coder_config, int32_t, int32_t, const uint32_t*, const uint64_t*, size_t,
                                         const uint32_t*, const double*, const double*, int32_t*, size_t, size_t,
                                         cst_layout, uint64_t*, uint32_t*, int32_t*, uint32_t, void*) { return CST_ERR_INVALID_ARGUMENT; }
}
