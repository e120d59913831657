// LAB: what an LDS fragment read costs the wave that issues it, one reading wave per SIMD (the M cluster of gemm_nt_v3's alternating K
// step), with the SIMD partner idle / issuing MFMAs / reading too.   hipcc -O3 --offload-arch=gfx950 tools/lab/lds_rate.hip -o gpurun_out/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// KIND 0: ds_read_b64_tr_b16 (32 per round, the K-major image of gemm_v3.h: 64 k-rows x 256 B, chunk ^= 2 h(k))
// KIND 1: ds_read_b128 (16 per round, the K-contiguous image: 128 rows x 128 B, chunk ^= (row >> 1) & 7)
// PARTNER 0: waves 4-7 exit; 1: waves 4-7 issue MFMAs for as long; 2: waves 4-7 read too
template <int KIND, int PARTNER>
__global__ __launch_bounds__(512) void k_rate(unsigned long long* out, float* sink, int rounds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
    __syncthreads();
    const bool reader = wave < 4 || PARTNER == 2;
    if (!reader && PARTNER == 0) return;
    unsigned long long t0 = 0, t1 = 0;
    if (reader) {
        const int q = lane >> 4, c = lane & 15, trq = c >> 2, trp = c & 3;
        const int thx = (trq | ((q & 1) << 2)) << 1;
        unsigned a[4];
        for (int i = 0; i < 4; ++i) {
            if (KIND == 0) a[i] = (8 * q + trq) * 256 + ((((wave & 1) * 8 + 2 * i + (trp >> 1)) ^ thx) * 16) + (trp & 1) * 8;
            else           a[i] = (((wave & 1) * 64 + 16 * i + c) * 128) + (((q) ^ ((c >> 1) & 7)) * 16);
        }
        bf16x4 r[16];
        bf16x8 w[8];
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < rounds; ++it) {
            if (KIND == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
                                 "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
                                 "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
                                 "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
                                 "ds_read_b64_tr_b16 %0, %8 offset:%14\n\tds_read_b64_tr_b16 %1, %8 offset:%15\n\t"
                                 "ds_read_b64_tr_b16 %2, %9 offset:%14\n\tds_read_b64_tr_b16 %3, %9 offset:%15\n\t"
                                 "ds_read_b64_tr_b16 %4, %10 offset:%14\n\tds_read_b64_tr_b16 %5, %10 offset:%15\n\t"
                                 "ds_read_b64_tr_b16 %6, %11 offset:%14\n\tds_read_b64_tr_b16 %7, %11 offset:%15"
                                 : "=&v"(r[8 * h + 0]), "=&v"(r[8 * h + 1]), "=&v"(r[8 * h + 2]), "=&v"(r[8 * h + 3]), "=&v"(r[8 * h + 4]),
                                   "=&v"(r[8 * h + 5]), "=&v"(r[8 * h + 6]), "=&v"(r[8 * h + 7])
                                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(0), "n"(1024), "n"(8192), "n"(8192 + 1024)
                                 : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[8])::"memory");
            } else {
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
                             "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %9 offset:64\n\tds_read_b128 %6, %10 offset:64\n\tds_read_b128 %7, %11 offset:64\n\t"
                             "ds_read_b128 %0, %8 offset:16384\n\tds_read_b128 %1, %9 offset:16384\n\tds_read_b128 %2, %10 offset:16384\n\tds_read_b128 %3, %11 offset:16384\n\t"
                             "ds_read_b128 %4, %8 offset:16448\n\tds_read_b128 %5, %9 offset:16448\n\tds_read_b128 %6, %10 offset:16448\n\tds_read_b128 %7, %11 offset:16448\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
                             : "memory");
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (blockIdx.x == 3 && lane == 0) out[wave] = t1 - t0;
        if (KIND == 0) { if (sink && (float)r[0][0] == 123.f) sink[0] = (float)r[9][1]; }
        else           { if (sink && (float)w[0][0] == 123.f) sink[0] = (float)w[5][1]; }
    } else {
        // the partner: 32 MFMAs per round (512 matrix-pipe cycles), about what a C cluster issues beside an M cluster
        f32x4 acc[8];
        bf16x8 fa, fb;
        for (int i = 0; i < 8; ++i) fa[i] = fb[i] = (__bf16)1.0f;
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < rounds; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[m & 7], 0, 0, 0);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (blockIdx.x == 3 && lane == 0) out[wave] = t1 - t0;
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (sink && s == 123.f) sink[1] = s;
    }
}


// ---- LDS-DMA pieces (buffer_load_dwordx4 ... lds: 64 lanes x 16 B = 1 KiB per instruction), as gemm_v3.h issues them: M0 = LDS
// destination, one VGPR of byte offsets, an SGPR of step offset. MIX 0: four pieces back to back per round; 1: a piece after every
// four ds_read_b128 (16 reads + 4 pieces per round); 2: 16 reads, then the four pieces. PARTNER as above. Source: a 4 MiB buffer
// that stays in L2; vmcnt is allowed eight pieces behind.
typedef __attribute__((address_space(3))) unsigned char* lptr3_t;
template <int MIX, int PARTNER>
__global__ __launch_bounds__(512) void k_piece(unsigned long long* out, float* sink, int rounds, const unsigned char* src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 131072 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
    __syncthreads();
    const bool worker = wave < 4 || PARTNER == 2;
    if (!worker && PARTNER == 0) return;
    unsigned long long t0 = 0, t1 = 0;
    if (worker) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
        const int voff = ((blockIdx.x & 63) * 8 + wave) * 1024 + lane * 16;
        const int q = lane >> 4, c = lane & 15;
        unsigned a[4];
        for (int i = 0; i < 4; ++i) a[i] = (((wave & 1) * 64 + 16 * i + c) * 128) + (((q) ^ ((c >> 1) & 7)) * 16);
        bf16x8 w[8];
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < rounds; ++it) {
            const int so = (it & 31) * 65536;
            auto piece = [&](int k) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr3_t)(lds + 65536 + ((it & 1) * 4 + k) * 8192 + wave * 1024), 16, voff, so + k * 8192, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto reads4 = [&](auto g_c, bf16x8 (&w)[8], unsigned (&a)[4]) {
                constexpr int G = decltype(g_c)::value;
                asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8"
                             : "=&v"(w[(G & 1) * 4 + 0]), "=&v"(w[(G & 1) * 4 + 1]), "=&v"(w[(G & 1) * 4 + 2]), "=&v"(w[(G & 1) * 4 + 3])
                             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"((G & 1) * 64 + (G >> 1) * 16384)
                             : "memory");
            };
            std::integral_constant<int, 0> g0; std::integral_constant<int, 1> g1; std::integral_constant<int, 2> g2; std::integral_constant<int, 3> g3;
            if (MIX == 0) { piece(0); piece(1); piece(2); piece(3); }
            if (MIX == 1) { reads4(g0, w, a); piece(0); reads4(g1, w, a); piece(1); reads4(g2, w, a); piece(2); reads4(g3, w, a); piece(3); }
            if (MIX == 2) { reads4(g0, w, a); reads4(g1, w, a); reads4(g2, w, a); reads4(g3, w, a); piece(0); piece(1); piece(2); piece(3); }
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : "+v"(w[0]), "+v"(w[4])::"memory");
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (blockIdx.x == 3 && lane == 0) out[wave] = t1 - t0;
        if (sink && (float)w[0][0] == 123.f) sink[0] = (float)w[5][1];
    } else {
        f32x4 acc[8];
        bf16x8 fa, fb;
        for (int i = 0; i < 8; ++i) fa[i] = fb[i] = (__bf16)1.0f;
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < rounds; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[m & 7], 0, 0, 0);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (blockIdx.x == 3 && lane == 0) out[wave] = t1 - t0;
        float sm = 0.f;
        for (int i = 0; i < 8; ++i) sm += acc[i][0];
        if (sink && sm == 123.f) sink[1] = sm;
    }
}

template <int MIX, int PARTNER>
int run_piece(const char* what, unsigned long long* d_out, int rounds, const unsigned char* src) {
    auto kern = k_piece<MIX, PARTNER>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipMemset(d_out, 0, 64));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, d_out, (float*)nullptr, rounds, src);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost));
    printf("%-44s", what);
    for (int w = 0; w < 8; ++w) printf(" %7.1f", (double)h[w] / rounds);
    printf("\n");
    return 0;
}

// ---- the alternating K step in miniature: both halves of the workgroup swap M cluster (LDS fragment reads + LDS-DMA pieces) and C cluster
// (32 MFMAs on 32 of the wave's 128 accumulators) behind barriers, one cluster apart, as gemm_nt_v3's kstep_pp does -- no epilogue, no
// tile mapping, operands streamed from a buffer that is (BIG = 1) far larger than the caches or (0) resident in L2.
//   READS 0 none (fragments stay as they are), 1 ds_read_b128 (16 in phase 0, 8 in phase 1), 2 ds_read_b64_tr_b16 (32 / 16)
//   PIECES 0 none, 1 four per M cluster between the read groups
template <int READS, int PIECES, int BIG>
__global__ __launch_bounds__(512) void k_pp(unsigned long long* out, float* sink, int rounds, const unsigned char* src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 163840 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const int voff = wave * 1024 + lane * 16;
    const int q = lane >> 4, c = lane & 15, trq = c >> 2, trp = c & 3;
    const int thx = (trq | ((q & 1) << 2)) << 1;
    unsigned a[4];
    for (int i = 0; i < 4; ++i) {
        if (READS == 2) a[i] = (8 * q + trq) * 256 + ((((wave & 1) * 8 + 2 * i + (trp >> 1)) ^ thx) * 16) + (trp & 1) * 8;
        else            a[i] = (((wave & 1) * 64 + 16 * i + c) * 128) + (((q) ^ ((c >> 1) & 7)) * 16);
    }
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][4], bf[2][4];
    for (int s2 = 0; s2 < 2; ++s2) for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { af[s2][i][e] = (__bf16)(0.01f * (lane + e)); bf[s2][i][e] = (__bf16)(0.02f * (lane - e)); }
    unsigned long long t0 = 0, t1 = 0;
    int so = BIG ? (blockIdx.x & 63) * (1 << 20) : (blockIdx.x & 127) * (1 << 14);          // BIG: 64 streams (four workgroups of one XCD share each: ~75 % L2 hits), 64 KiB per round
    unsigned slot = 0;
    std::integral_constant<int, 0> o0; std::integral_constant<int, 8192> o1;
    auto rd4 = [&](unsigned base, auto off_c, bf16x8 (&f)[4], unsigned (&a)[4]) {
        constexpr int OFF = decltype(off_c)::value;
        if constexpr (READS == 1) {
            asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8"
                         : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(a[0] + base), "v"(a[1] + base), "v"(a[2] + base), "v"(a[3] + base), "n"(OFF) : "memory");
        } else if constexpr (READS == 2) {
            bf16x4 l[4], h[4];
            asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
                         "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
                         "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
                         "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13"
                         : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3])
                         : "v"(a[0] + base), "v"(a[1] + base), "v"(a[2] + base), "v"(a[3] + base), "n"(OFF), "n"(OFF + 1024) : "memory");
            for (int i = 0; i < 4; ++i) f[i] = __builtin_shufflevector(l[i], h[i], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int rot = wave & 3;
    auto piece_c = [&](int k) {
        if constexpr (PIECES == 2 || PIECES == 3 || PIECES == 9 || PIECES == 10) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr3_t)(lds + 65536 + ((slot + k) % 12) * 8192 + wave * 1024), 16, voff, so + k * 8192, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the C cluster's MFMA number m (0..31) has just been issued: PIECES 2: wave r of the half issues piece k behind MFMA 8 k + 2 r + 1 (the
    // sixteen pieces of a slot 32 cycles apart on the texture path); PIECES 3: every wave behind MFMAs 3, 11, 19, 27
    auto after_mfma = [&](int m) {
        if constexpr (PIECES == 2) { if ((m & 7) == 2 * rot + 1) piece_c(m >> 3); }
        if constexpr (PIECES == 3) { if ((m & 7) == 3) piece_c(m >> 3); }
    };
    auto piece = [&](int k) {
        // PIECES 5 / 6: as 1, with two more scalar (s_nop) / vector (v_mov) instructions per piece in the M cluster: what does an
        // INSTRUCTION cost the wave that issues it beside a streaming partner, whatever it does?
        if constexpr (PIECES == 5) { asm volatile("s_nop 0\n\ts_nop 0" ::: "memory"); }
        if constexpr (PIECES == 6) { int d0, d1; asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(d0), "=v"(d1)); asm volatile("" ::"v"(d0), "v"(d1)); }
        if constexpr (PIECES == 1 || PIECES == 5 || PIECES == 6 || PIECES == 7 || PIECES == 8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr3_t)(lds + 65536 + ((slot + k) % 12) * 8192 + wave * 1024), 16, voff, so + k * 8192, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one C cluster on accumulators I0 .. I0 + 3; the stagger is a compile-time constant per copy (a compare + branch behind every MFMA
    // would cost the wave an issue slot each: 1,020 cycles per slot measured), the copy is chosen once per cluster
    auto cbody = [&](auto i0_c, auto rot_c) {
        constexpr int I0 = decltype(i0_c)::value, ROT = decltype(rot_c)::value;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[I0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][i], bf[s2][j], acc[I0 + i][j], 0, 0, 0);
                    constexpr int dummy = 0; (void)dummy;
                    const int m = s2 * 16 + i * 4 + j;
                    if constexpr (PIECES == 2) { if ((m & 7) == 2 * ROT + 1) { __builtin_amdgcn_sched_barrier(0); piece_c(m >> 3); } }
                    if constexpr (PIECES == 3 || PIECES == 9 || PIECES == 10) { if ((m & 7) == 3) { __builtin_amdgcn_sched_barrier(0); piece_c(m >> 3); } }
                }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto ccluster = [&](auto i0_c) {
        // PIECES 9 / 10: the pieces among the C cluster's MFMAs at the same places for every wave, but wave r of the half starts the
        // cluster 16 r / 32 r cycles late (its partner is in an M cluster then: the s_nops cost what they say)
        if constexpr (PIECES == 9) { for (int i = 0; i < rot; ++i) asm volatile("s_nop 15" ::: "memory"); }
        if constexpr (PIECES == 10) { for (int i = 0; i < rot; ++i) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
        if constexpr (PIECES == 2) {
            if (rot == 0) cbody(i0_c, std::integral_constant<int, 0>());
            else if (rot == 1) cbody(i0_c, std::integral_constant<int, 1>());
            else if (rot == 2) cbody(i0_c, std::integral_constant<int, 2>());
            else cbody(i0_c, std::integral_constant<int, 3>());
        } else cbody(i0_c, std::integral_constant<int, 0>());
    };
    auto rd1 = [&](unsigned addr, auto off_c, bf16x8& f) {
        constexpr int OFF = decltype(off_c)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f) : "v"(addr), "n"(OFF) : "memory");
    };
    auto piece_m = [&](int k) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr3_t)(lds + 65536 + ((slot + k) % 12) * 8192 + wave * 1024), 16, voff, so + k * 8192, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // PIECES 4: one piece behind read number 4 k + rot of the M cluster (phase 0: 16 reads; phase 1: 8 reads, behind read 2 k + (rot >> 1)):
    // at any one read only ONE (phase 1: two) of the half's four waves offers the texture path a piece
    auto m0_rot = [&](unsigned base) {
        int n = 0;
        auto one = [&](unsigned addr, auto off_c, bf16x8& f) { rd1(addr, off_c, f); __builtin_amdgcn_sched_barrier(0); if ((n & 3) == rot) piece_m(n >> 2); ++n; };
        for (int i = 0; i < 4; ++i) one(a[i] + base, o0, bf[0][i]);
        for (int i = 0; i < 4; ++i) one(a[i] + base, o1, bf[1][i]);
        for (int i = 0; i < 4; ++i) one(a[i] + base + 16384, o0, af[0][i]);
        for (int i = 0; i < 4; ++i) one(a[i] + base + 16384, o1, af[1][i]);
    };
    auto m1_rot = [&](unsigned base) {
        int n = 0;
        auto one = [&](unsigned addr, auto off_c, bf16x8& f) { rd1(addr, off_c, f); __builtin_amdgcn_sched_barrier(0); if ((n & 1) == (rot >> 1)) piece_m(n >> 1); ++n; };
        for (int i = 0; i < 4; ++i) one(a[i] + base + 16384, o0, af[0][i]);
        for (int i = 0; i < 4; ++i) one(a[i] + base + 16384, o1, af[1][i]);
    };
    // PIECES 11: NO per-wave code path -- every wave runs the same 16 piece instructions per M cluster, one behind every read, with EXEC
    // all ones on the four whose slot number is its own (slot & 3 == wave & 3) and ZERO on the other twelve: at any one read only one of the
    // half's four waves offers the texture path a real piece. (Inline asm: M0, EXEC and the load in one statement, EXEC restored in it.)
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const unsigned long long sp = (unsigned long long)src;
    const i32x4 rsv = {(int)(unsigned)sp, (int)((unsigned)(sp >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
    const int em0 = __builtin_amdgcn_readfirstlane(rot == 0 ? -1 : 0), em1 = __builtin_amdgcn_readfirstlane(rot == 1 ? -1 : 0);
    const int em2 = __builtin_amdgcn_readfirstlane(rot == 2 ? -1 : 0), em3 = __builtin_amdgcn_readfirstlane(rot == 3 ? -1 : 0);
    auto piece_x = [&](auto j_c) {
        constexpr int J = decltype(j_c)::value;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr3_t)(lds + 65536 + ((slot + (J >> 2)) % 12) * 8192 + wave * 1024));
        const int soff = __builtin_amdgcn_readfirstlane(so + (J >> 2) * 8192);
        const int em = __builtin_amdgcn_readfirstlane((J & 3) == 0 ? em0 : (J & 3) == 1 ? em1 : (J & 3) == 2 ? em2 : em3);
        const int voff_ = voff;                    // (operands of an asm statement inside a generic lambda are not captured by themselves)
        const i32x4 rsv_ = rsv;
        asm volatile("s_mov_b32 m0, %0\n\ts_mov_b32 exec_lo, %1\n\ts_mov_b32 exec_hi, %1\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, -1"
                     ::"s"(dst), "s"(em), "v"(voff_), "s"(rsv_), "s"(soff) : "memory");
    };
#define LAB_IC(n) std::integral_constant<int, n>()
    auto m0_mask = [&](unsigned base) {
        rd1(a[0] + base, o0, bf[0][0]); piece_x(LAB_IC(0)); rd1(a[1] + base, o0, bf[0][1]); piece_x(LAB_IC(1));
        rd1(a[2] + base, o0, bf[0][2]); piece_x(LAB_IC(2)); rd1(a[3] + base, o0, bf[0][3]); piece_x(LAB_IC(3));
        rd1(a[0] + base, o1, bf[1][0]); piece_x(LAB_IC(4)); rd1(a[1] + base, o1, bf[1][1]); piece_x(LAB_IC(5));
        rd1(a[2] + base, o1, bf[1][2]); piece_x(LAB_IC(6)); rd1(a[3] + base, o1, bf[1][3]); piece_x(LAB_IC(7));
        rd1(a[0] + base + 16384, o0, af[0][0]); piece_x(LAB_IC(8)); rd1(a[1] + base + 16384, o0, af[0][1]); piece_x(LAB_IC(9));
        rd1(a[2] + base + 16384, o0, af[0][2]); piece_x(LAB_IC(10)); rd1(a[3] + base + 16384, o0, af[0][3]); piece_x(LAB_IC(11));
        rd1(a[0] + base + 16384, o1, af[1][0]); piece_x(LAB_IC(12)); rd1(a[1] + base + 16384, o1, af[1][1]); piece_x(LAB_IC(13));
        rd1(a[2] + base + 16384, o1, af[1][2]); piece_x(LAB_IC(14)); rd1(a[3] + base + 16384, o1, af[1][3]); piece_x(LAB_IC(15));
    };
    auto m1_mask = [&](unsigned base) {
        rd1(a[0] + base + 16384, o0, af[0][0]); piece_x(LAB_IC(0)); piece_x(LAB_IC(1)); rd1(a[1] + base + 16384, o0, af[0][1]); piece_x(LAB_IC(2)); piece_x(LAB_IC(3));
        rd1(a[2] + base + 16384, o0, af[0][2]); piece_x(LAB_IC(4)); piece_x(LAB_IC(5)); rd1(a[3] + base + 16384, o0, af[0][3]); piece_x(LAB_IC(6)); piece_x(LAB_IC(7));
        rd1(a[0] + base + 16384, o1, af[1][0]); piece_x(LAB_IC(8)); piece_x(LAB_IC(9)); rd1(a[1] + base + 16384, o1, af[1][1]); piece_x(LAB_IC(10)); piece_x(LAB_IC(11));
        rd1(a[2] + base + 16384, o1, af[1][2]); piece_x(LAB_IC(12)); piece_x(LAB_IC(13)); rd1(a[3] + base + 16384, o1, af[1][3]); piece_x(LAB_IC(14)); piece_x(LAB_IC(15));
    };
    auto landed = [&](bf16x8 (&f)[4]) { asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); };
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (wave >= 4) asm volatile("s_barrier" ::: "memory");
    for (int it = 0; it < rounds; ++it) {
        const unsigned base = (it & 1) * 32768;
        // M(t, 0)   (PIECES 7 / 8: wave r of the half starts its M cluster 32 r / 48 r cycles late, so the four waves' pieces reach the
        //            texture path one after the other instead of at the same moment)
        if constexpr (PIECES == 7) { for (int i = 0; i < rot; ++i) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
        if constexpr (PIECES == 8) { for (int i = 0; i < rot; ++i) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
        if constexpr (PIECES == 4) m0_rot(base);
        else if constexpr (PIECES == 11) m0_mask(base);
        else { rd4(base, o0, bf[0], a); piece(0); rd4(base, o1, bf[1], a); piece(1); rd4(base + 16384, o0, af[0], a); piece(2); rd4(base + 16384, o1, af[1], a); piece(3); }
        asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        landed(af[0]); landed(af[1]); landed(bf[0]); landed(bf[1]);
        __builtin_amdgcn_sched_barrier(0);
        ccluster(std::integral_constant<int, 0>());
        asm volatile("s_barrier" ::: "memory");
        slot += 4; so += 32768;
        // M(t, 1)
        if constexpr (PIECES == 7) { for (int i = 0; i < rot; ++i) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
        if constexpr (PIECES == 8) { for (int i = 0; i < rot; ++i) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
        if constexpr (PIECES == 4) m1_rot(base);
        else if constexpr (PIECES == 11) m1_mask(base);
        else { rd4(base + 16384, o0, af[0], a); piece(0); piece(1); rd4(base + 16384, o1, af[1], a); piece(2); piece(3); }
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        landed(af[0]); landed(af[1]);
        __builtin_amdgcn_sched_barrier(0);
        ccluster(std::integral_constant<int, 4>());
        asm volatile("s_barrier" ::: "memory");
        slot += 4; so += 32768;
        if (!BIG) so &= (1 << 21) - 1;
    }
    if (wave < 4) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (blockIdx.x == 3 && lane == 0) out[wave] = t1 - t0;
    float sm = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) sm += acc[i][j][0] + acc[i][j][3];
    if (sink && sm == 123.f) sink[1] = sm;
}

template <int READS, int PIECES, int BIG>
int run_pp(const char* what, unsigned long long* d_out, int rounds, const unsigned char* src) {
    auto kern = k_pp<READS, PIECES, BIG>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CK(hipMemset(d_out, 0, 64));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 163840, 0, d_out, (float*)nullptr, rounds, src);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost));
    printf("%-58s cycles per slot (4 per K step; 512 = the C cluster's MFMAs): %7.1f\n", what, (double)h[0] / rounds / 4);
    return 0;
}

template <int KIND, int PARTNER>
int run(const char* what, unsigned long long* d_out, int rounds) {
    auto kern = k_rate<KIND, PARTNER>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipMemset(d_out, 0, 64));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, d_out, (float*)nullptr, rounds);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost));
    const int per = KIND == 0 ? 32 : 16;
    printf("%-44s", what);
    for (int w = 0; w < 8; ++w) printf(" %7.1f", (double)h[w] / rounds);
    printf("   | reader: %.1f cycles per read (%d reads of %d B per round)\n", (double)h[0] / rounds / per, per, KIND == 0 ? 512 : 1024);
    return 0;
}

int main() {
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, 64));
    const int rounds = 2000;
    printf("cycles per round, waves 0..7 of workgroup 3 (readers: waves 0-3%s)\n", "");
    if (run<0, 0>("tr_b16  partner idle", d_out, rounds)) return 1;
    if (run<0, 1>("tr_b16  partner: 32 MFMAs per round", d_out, rounds)) return 1;
    if (run<0, 2>("tr_b16  partner reads too", d_out, rounds)) return 1;
    if (run<1, 0>("b128    partner idle", d_out, rounds)) return 1;
    if (run<1, 1>("b128    partner: 32 MFMAs per round", d_out, rounds)) return 1;
    if (run<1, 2>("b128    partner reads too", d_out, rounds)) return 1;
    unsigned char* src;
    CK(hipMalloc(&src, 8u << 20)); CK(hipMemset(src, 0x3f, 8u << 20));
    printf("cycles per round: 4 pieces (+ 16 ds_read_b128) per working wave\n");
    if (run_piece<0, 0>("4 pieces               partner idle", d_out, rounds, src)) return 1;
    if (run_piece<0, 1>("4 pieces               partner MFMAs", d_out, rounds, src)) return 1;
    if (run_piece<0, 2>("4 pieces               partner pieces too", d_out, rounds, src)) return 1;
    if (run_piece<1, 0>("4 x (4 reads, piece)   partner idle", d_out, rounds, src)) return 1;
    if (run_piece<1, 1>("4 x (4 reads, piece)   partner MFMAs", d_out, rounds, src)) return 1;
    if (run_piece<2, 0>("16 reads, 4 pieces     partner idle", d_out, rounds, src)) return 1;
    if (run_piece<2, 1>("16 reads, 4 pieces     partner MFMAs", d_out, rounds, src)) return 1;
    unsigned char* big;
    CK(hipMalloc(&big, (size_t)1 << 31)); CK(hipMemset(big, 0x3c, (size_t)1 << 31));
    const int r2 = 240;        // (BIG: 128 streams x 240 rounds x 64 KiB < 16 MiB each... offsets stay below 2^31)
    if (run_pp<0, 0, 0>("alternating K step: MFMAs only", d_out, r2, big)) return 1;
    if (run_pp<1, 0, 0>("  + b128 reads", d_out, r2, big)) return 1;
    if (run_pp<0, 1, 0>("  + pieces (L2-resident source)", d_out, r2, big)) return 1;
    if (run_pp<0, 1, 1>("  + pieces (streamed source)", d_out, r2, big)) return 1;
    if (run_pp<1, 1, 0>("  + b128 reads + pieces (L2-resident)", d_out, r2, big)) return 1;
    if (run_pp<1, 1, 1>("  + b128 reads + pieces (streamed)", d_out, r2, big)) return 1;
    if (run_pp<1, 5, 1>("  + b128 reads + pieces + 8 s_nop per M cluster", d_out, r2, big)) return 1;
    if (run_pp<1, 6, 1>("  + b128 reads + pieces + 8 v_mov per M cluster", d_out, r2, big)) return 1;
    if (run_pp<1, 11, 1>("  + b128 reads, 16 piece slots per wave, EXEC zero on 12", d_out, r2, big)) return 1;
    if (run_pp<1, 9, 1>("  + b128 reads, pieces in C, wave r starts C 16 r late", d_out, r2, big)) return 1;
    if (run_pp<1, 10, 1>("  + b128 reads, pieces in C, wave r starts C 32 r late", d_out, r2, big)) return 1;
    if (run_pp<2, 9, 1>("  + tr reads, pieces in C, wave r starts C 16 r late", d_out, r2, big)) return 1;
    if (run_pp<2, 3, 1>("  + tr reads, pieces in C, all waves at once", d_out, r2, big)) return 1;
    if (run_pp<1, 7, 1>("  + b128 reads + pieces, wave r of a half 32 r cycles late", d_out, r2, big)) return 1;
    if (run_pp<1, 8, 1>("  + b128 reads + pieces, wave r of a half 48 r cycles late", d_out, r2, big)) return 1;
    if (run_pp<2, 7, 1>("  + tr reads + pieces, wave r of a half 32 r cycles late", d_out, r2, big)) return 1;
    if (run_pp<1, 4, 1>("  + b128 reads, pieces in M ROTATED by wave (one per read)", d_out, r2, big)) return 1;
    if (run_pp<1, 3, 1>("  + b128 reads, pieces in C, all waves at once", d_out, r2, big)) return 1;
    if (run_pp<2, 0, 0>("  + tr reads", d_out, r2, big)) return 1;
    if (run_pp<2, 1, 1>("  + tr reads + pieces (streamed)", d_out, r2, big)) return 1;
    return 0;
}
