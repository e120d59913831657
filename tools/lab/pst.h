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
static __device__ unsigned long long g_pst_raw[8 * 16];
static __device__ unsigned long long g_pst_m[8 * 8];        // phase 0's M cluster, cycles from its start: after read group / piece      // one K step's stamps as they are: [wave][phase][t0, t1, t5, t6, t2, t3]
#define V3_PST_DECL unsigned long long pst_t0 = 0, pst_t1 = 0, pst_t2 = 0, pst_t3 = 0, pst_t4 = 0, pst_t5 = 0, pst_t6 = 0, pst_a4 = 0, pst_a5 = 0, pst_r0 = 0, pst_r1 = 0, pst_r2 = 0, pst_r3 = 0, pst_r4 = 0, pst_r5 = 0, pst_r6 = 0, pst_r7 = 0, pst_r8 = 0, pst_r9 = 0, pst_r10 = 0, pst_r11 = 0, pst_r12 = 0, pst_r13 = 0, pst_r14 = 0, pst_r15 = 0, pst_r16 = 0, pst_r17 = 0, pst_r18 = 0, pst_r19 = 0, pst_step = 0, pst_m0 = 0, pst_m1 = 0, pst_m2 = 0, pst_m3 = 0, pst_m4 = 0, pst_m5 = 0, pst_m6 = 0, pst_m7 = 0, pst_t0p = 0, pst_t1p = 0, pst_a0 = 0, pst_a1 = 0, pst_a2 = 0, pst_a3 = 0, pst_n = 0;
#define V3_PST_S(k) asm volatile("s_memtime %0" : "=s"(pst_t##k)::"memory")
// (no stamp 1 -- "reads and pieces issued" -- in front of the M cluster's lgkmcnt(0): that wait would then wait for the stamp's own answer,
// ~200 cycles under this load, and every M cluster would look, and be, that much longer. What is stamped is what needs no wait of its own.)
#define V3_PST_1() do { } while (0)
#define V3_PST_0() V3_PST_S(0)
#define V3_PST_2() V3_PST_S(2)
#define V3_PST_3() V3_PST_S(3)
#define V3_PST_4() V3_PST_S(4)
#define V3_PST_5() V3_PST_S(5)
#define V3_PST_6() V3_PST_S(6)
#define V3_PST(k) V3_PST_##k()
// inside phase 0's M cluster (piece placement 2): after each read group and after each piece -- issued only, read one phase later
#define V3_PSTM(i) asm volatile("s_memtime %0" : "=s"(pst_m##i)::"memory")
// Nothing is waited for at the end of the C cluster (a wait there for the stamp's answer would put the scalar cache's latency into
// every slot): the stamps of a phase are added up behind the lgkmcnt(0) of the NEXT phase's M cluster.
#define V3_PST_ACC() do { } while (0)
// the M cluster's closing wait, split: LDS reads back (stamp 5), counted pieces landed (stamp 6), barrier
#define V3_PST_WAIT_BARRIER(N) do { \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pst_t0), "+s"(pst_t1), "+s"(pst_t2), "+s"(pst_t3), "+s"(pst_t5), "+s"(pst_t6)::"memory"); \
        if (pst_n) { pst_a1 += pst_t2 - pst_t1p; pst_a2 += pst_t3 - pst_t2; pst_a4 += pst_t5 - pst_t1p; pst_a5 += pst_t6 - pst_t5; \
            if (pst_n == 2 * 100 - 1) { pst_r0 = pst_t0p; pst_r1 = pst_t1p; pst_r2 = pst_t5; pst_r3 = pst_t6; pst_r4 = pst_t2; pst_r5 = pst_t3; } \
            if (pst_n == 2 * 100) { pst_r6 = pst_t0p; pst_r7 = pst_t1p; pst_r8 = pst_t5; pst_r9 = pst_t6; pst_r10 = pst_t2; pst_r11 = pst_t3; } } \
        pst_t1 = pst_t0; pst_t0p = pst_t0; pst_t1p = pst_t1; ++pst_n; \
        if (pst_n == 2 * 100 - 1) { asm volatile("" : "+s"(pst_m0), "+s"(pst_m1), "+s"(pst_m2), "+s"(pst_m3), "+s"(pst_m4), "+s"(pst_m5), "+s"(pst_m6), "+s"(pst_m7)); \
            pst_r12 = pst_m0 - pst_t0; pst_r13 = pst_m1 - pst_t0; pst_r14 = pst_m2 - pst_t0; pst_r15 = pst_m3 - pst_t0; pst_r16 = pst_m4 - pst_t0; \
            pst_r17 = pst_m5 - pst_t0; pst_r18 = pst_m6 - pst_t0; pst_r19 = pst_m7 - pst_t0; } \
        V3_PST(5); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); V3_PST(6); asm volatile("s_barrier" ::: "memory"); } while (0)
#define V3_PST_PASS_BEGIN() V3_PST(4)
#define V3_PST_PASS_END() do { V3_PST(3); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pst_t3), "+s"(pst_t4)::"memory"); pst_a3 += pst_t3 - pst_t4; } while (0)
#define V3_PST_FLUSH(wave_, lane_) do { if (blockIdx.x == 8 && (lane_) == 0) { unsigned long long* o_ = g_pst + (wave_) * 8; \
        o_[0] = pst_a0; o_[1] = pst_a1; o_[2] = pst_a2; o_[3] = pst_a3; o_[4] = pst_n; o_[5] = pst_a4; o_[6] = pst_a5; unsigned long long* r_ = g_pst_raw + (wave_) * 16; \
        r_[0] = pst_r0; r_[1] = pst_r1; r_[2] = pst_r2; r_[3] = pst_r3; r_[4] = pst_r4; r_[5] = pst_r5; r_[6] = pst_r6; r_[7] = pst_r7; r_[8] = pst_r8; \
        r_[9] = pst_r9; r_[10] = pst_r10; r_[11] = pst_r11; unsigned long long* m_ = g_pst_m + (wave_) * 8; \
        m_[0] = pst_r12; m_[1] = pst_r13; m_[2] = pst_r14; m_[3] = pst_r15; m_[4] = pst_r16; m_[5] = pst_r17; m_[6] = pst_r18; m_[7] = pst_r19; } } while (0)
#ifdef PST_READER
extern "C" __attribute__((visibility("default"))) int PST_READER(unsigned long long* out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pst), sizeof(g_pst)) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(g_pst_raw), sizeof(g_pst_raw)) != hipSuccess) return 2;
    return (int)hipMemcpyFromSymbol(out + 192, HIP_SYMBOL(g_pst_m), sizeof(g_pst_m));
}
#endif
