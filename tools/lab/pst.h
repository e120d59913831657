// tools/lab/pst.h -- LAB ONLY: cycle stamps of gemm_nt_v3's alternating K step (M cluster / C cluster), force-included into a
// separate build of the library:
//   cd vbnn_amd/csrc && make LIBDIR=../lib/pst EXTRA='-include ../../tools/lab/pst.h -DPST_READER=vbnn_lab_pst_$(*F)'
//   VBNN_HIP_LIB=.../lib/pst/libvbnn_hip.so python tools/pst_read.py
// The shipped build defines none of this (gemm_v3.h's V3_PST* hooks are empty there) and exports no such symbol.
// s_memtime answers on lgkmcnt, so a stamp is ISSUED where it is wanted and read behind a wait the loop has anyway (the M
// cluster's lgkmcnt(0)) or one the lab adds where no LDS read is in flight (the end of the C cluster).
#pragma once
#include <hip/hip_runtime.h>
static __device__ unsigned long long g_pst[8 * 8];
#define V3_PST_DECL unsigned long long pst_t0 = 0, pst_t1 = 0, pst_t2 = 0, pst_t3 = 0, pst_t4 = 0, pst_a0 = 0, pst_a1 = 0, pst_a2 = 0, pst_a3 = 0, pst_n = 0;
#define V3_PST(k) asm volatile("s_memtime %0" : "=s"(pst_t##k)::"memory")
#define V3_PST_ACC() do { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pst_t0), "+s"(pst_t1), "+s"(pst_t2), "+s"(pst_t3)::"memory"); pst_a0 += pst_t1 - pst_t0; pst_a1 += pst_t2 - pst_t1; \
                          pst_a2 += pst_t3 - pst_t2; ++pst_n; } while (0)
#define V3_PST_PASS_BEGIN() V3_PST(4)
#define V3_PST_PASS_END() do { V3_PST(3); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pst_t3), "+s"(pst_t4)::"memory"); pst_a3 += pst_t3 - pst_t4; } while (0)
#define V3_PST_FLUSH(wave_, lane_) do { if (blockIdx.x == 8 && (lane_) == 0) { unsigned long long* o_ = g_pst + (wave_) * 8; \
        o_[0] = pst_a0; o_[1] = pst_a1; o_[2] = pst_a2; o_[3] = pst_a3; o_[4] = pst_n; } } while (0)
#ifdef PST_READER
extern "C" __attribute__((visibility("default"))) int PST_READER(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pst), sizeof(g_pst));
}
#endif
