// gemm_v2.h -- pipelined bf16 MFMA GEMM (placeholder until the kernel lands).
#pragma once
#include "common.h"
template <typename T>
static inline bool gemm_v2_eligible(int64_t, int64_t, int64_t, int64_t, int64_t) { return false; }
template <typename T, bool DUAL, class Epi>
static int launch_gemm_v2(hipStream_t, const T*, const T*, int64_t, const T*, const T*, int64_t, int, int, int, const Epi&) {
    vbnn_set_error("gemm_v2 not built");
    return VBNN_ERR_UNSUPPORTED;
}
