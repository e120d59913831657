// tools/lab/pst.h -- LAB ONLY: cycle stamps of gemm_nt_v3's alternating K step (M cluster / C cluster), force-included into a
// separate build of the library through the kernel's stamp hook V3_ST(k):
//   cd vbnn_amd/csrc && make LIBDIR=../lib/pst EXTRA='-include ../../tools/lab/pst.h -DPST_READER=vbnn_lab_pst_$(*F)'
//   VBNN_HIP_LIB=$PWD/vbnn_amd/lib/pst/libvbnn_hip.so python tools/pst_read.py
// The shipped build defines none of this (V3_ST is empty there) and exports no such symbol.
// s_memtime answers on lgkmcnt, so a stamp is ISSUED where it is wanted and READ behind a wait the loop has anyway: the lgkmcnt(0) that
// closes the next M cluster. (A stamp in front of that wait makes the wait wait for the stamp -- ~200 cycles under this load -- and one
// read at the end of the C cluster puts that latency into every slot: both were tried, both bend the picture.)
//   V3_ST(10) M cluster starts   (11) C cluster starts: behind the M cluster's closing wait + barrier   (12) the C cluster's MFMAs are issued
//   V3_ST(13) a pass starts      (14) a pass ends (results written by workgroup 8)
#pragma once
#include <hip/hip_runtime.h>
static __device__ unsigned long long g_pst[8 * 8];           // [wave]: cycles M start -> C start, C, pass, phases counted
static __device__ unsigned long long g_pst_raw[8 * 16];      // one K step's stamps as they are: [wave][phase][M start, C start, C end]
#define V3_ST_DECL unsigned long long pst_t10 = 0, pst_t11 = 0, pst_t12 = 0, pst_t13 = 0, pst_t14 = 0, pst_p10 = 0, pst_p11 = 0, pst_p12 = 0, \
    pst_am = 0, pst_ac = 0, pst_ap = 0, pst_n = 0, pst_r0 = 0, pst_r1 = 0, pst_r2 = 0, pst_r3 = 0, pst_r4 = 0, pst_r5 = 0;
#define V3_ST_S(k) asm volatile("s_memtime %0" : "=s"(pst_t##k)::"memory")
#define V3_ST_0() V3_ST_DECL
#define V3_ST_1() do { } while (0)
#define V3_ST_4() do { } while (0)
#define V3_ST_5() do { } while (0)
#define V3_ST_6() do { } while (0)
#define V3_ST_10() V3_ST_S(10)
// behind the M cluster's lgkmcnt(0): everything stamped before it is readable -- this phase's M start and the phase BEFORE's C start / end
#define V3_ST_11() do { asm volatile("" : "+s"(pst_t10), "+s"(pst_t11), "+s"(pst_t12)); \
        if (pst_n) { pst_am += pst_t11 - pst_p10; pst_ac += pst_t12 - pst_t11; \
            if (pst_n == 2 * 100 - 1) { pst_r0 = pst_p10; pst_r1 = pst_t11; pst_r2 = pst_t12; } \
            if (pst_n == 2 * 100) { pst_r3 = pst_p10; pst_r4 = pst_t11; pst_r5 = pst_t12; } } \
        pst_p10 = pst_t10; ++pst_n; V3_ST_S(11); } while (0)
#define V3_ST_12() V3_ST_S(12)
#define V3_ST_13() V3_ST_S(13)
#define V3_ST_14() do { V3_ST_S(14); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pst_t13), "+s"(pst_t14)::"memory"); pst_ap += pst_t14 - pst_t13; \
        if (blockIdx.x == 8 && lane == 0) { unsigned long long* o_ = g_pst + wave * 8; o_[0] = pst_am; o_[1] = pst_ac; o_[2] = pst_ap; o_[3] = pst_n; \
            unsigned long long* r_ = g_pst_raw + wave * 16; r_[0] = pst_r0; r_[1] = pst_r1; r_[2] = pst_r2; r_[3] = pst_r3; r_[4] = pst_r4; r_[5] = pst_r5; } } while (0)
#define V3_ST(k) V3_ST_##k()
#ifdef PST_READER
extern "C" __attribute__((visibility("default"))) int PST_READER(unsigned long long* out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pst), sizeof(g_pst)) != hipSuccess) return 1;
    return (int)hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(g_pst_raw), sizeof(g_pst_raw));
}
#endif
