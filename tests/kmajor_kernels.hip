// Instantiates every kernel that reads K-major operand fragments with ds_read_b64_tr_b16 from inline asm, for
// tests/test_kernel_hazards.py (compiled with hipcc -S, never run): accGradParameters (both operands K-major, fused and
// split launches), updateGradInput (A K-major) and the pipelined kernel's K-major form.
#include "common.h"
#include "epilogues.h"
#include "gemm_v1.h"
#include "gemm_v2.h"
#include "gemm_v3.h"
void vbnn_set_error(const char*, ...) {}
int vbnn_cu_count() { return 256; }
int dw(vbnn_ctx* c, const bf16_t* a, const EpiDw& e) { return launch_gemm_v3<bf16_t, true, true, true, EpiDw>(c, a, a, 4096, a, a, 4096, 4096, 4096, 4096, e); }
int dw1(vbnn_ctx* c, const bf16_t* a, const EpiDw& e) { return launch_gemm_v3<bf16_t, false, true, true, EpiDw>(c, a, a, 4096, a, a, 4096, 4096, 4096, 4096, e); }
int dws(vbnn_ctx* c, const bf16_t* a, const EpiDw& e) { return launch_gemm_v3_split<bf16_t, EpiDw>(c, a, a, 1024, a, a, 4096, 785, 4096, 4096, e); }
int dx(vbnn_ctx* c, const bf16_t* a, const EpiDx<bf16_t>& e) { return launch_gemm_v3<bf16_t, true, true, false, EpiDx<bf16_t>>(c, a, a, 4096, a, a, 4096, 4096, 4096, 4096, e); }
int v2(vbnn_ctx* c, const bf16_t* a, const EpiDw& e) { return launch_gemm_v2<bf16_t, false, EpiDw>(c, a, a, 4096, a, a, 4096, 4096, 4096, 4096, e, true); }
