// v0_lab.hip -- the latency kernel (csrc/gemm_v0.h) beside gemm_v1's 32 x 32 tile at the small MLP's shapes (lab, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/v0_lab.hip -o gpurun_out/v0_lab && gpurun_out/v0_lab
// Random fp32 operands, a trivial epilogue, results of both kernels against float64 on the host, us per launch over a
// stream of dependent launches (the launch floor of tools/launch_floor.hip is in every figure).
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../vbnn_amd/csrc/common.h"
void vbnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int vbnn_cu_count() { return 256; }
#include "../vbnn_amd/csrc/gemm_v1.h"

struct EpiOut {          // out[n][m] = a1 + 2 a2, 16-byte stores
    float* out; int M, N;
    __device__ __forceinline__ void bind_draw() {}
    __host__ __device__ __forceinline__ float* t1_ptr() const { return nullptr; }
    __host__ __device__ __forceinline__ float* t2_ptr() const { return nullptr; }
    __device__ __forceinline__ int m_dim() const { return M; }
    __device__ __forceinline__ int n_dim() const { return N; }
    __device__ __forceinline__ void operator()(int m, int n, f32x4 a1, f32x4 a2) const {
        for (int r = 0; r < 4; ++r) if (m + r < M && n < N) out[(size_t)n * M + m + r] = a1[r] + 2.f * a2[r];
    }
    struct Pre {};
    struct Lane { unsigned o; };
    __host__ __device__ bool fast_ok() const { return M % 4 == 0; }
    __device__ __forceinline__ Lane lane_init(int nl, int ml) const { return Lane{(unsigned)(nl * M + ml)}; }
    __device__ __forceinline__ Pre load_fast(int, int, const Lane&) const { return Pre{}; }
    __device__ __forceinline__ void apply_fast(int um, int un, const Lane& ln, f32x4 a1, f32x4 a2, const Pre&, float (&)[4], float (&)[4]) const {
        *reinterpret_cast<f32x4*>(out + ((size_t)un * M + um) + ln.o) = a1 + 2.f * a2;
    }
};
template <int FM, int FN, bool DUAL, class Epi, bool TA, bool TB, int SQ>
static int launch_gemm_v0_tile(hipStream_t stream, const float* A, const float* A2, int64_t lda, const float* B, const float* B2, int64_t ldb,
                               int M, int N, int K, const Epi& epi, int ones_row) {
    const int gx = (M + 16 * FM - 1) / (16 * FM), gy = (N + 16 * FN - 1) / (16 * FN);
    hipLaunchKernelGGL((gemm_nt_v0<FM, FN, DUAL, Epi, TA, TB, SQ>), dim3(gx * gy), dim3(64 * V0_W), 0, stream, A, A2, lda, B, B2, ldb, M, N, K,
                       ones_row, gx, epi);
    return 0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float* dev(const std::vector<float>& h) {
    float* d; CK(hipMalloc(&d, h.size() * 4 + 64)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return d;
}

// one case: element (row, k) of an operand at X[row * ld + k] (K-contiguous) or X[k * ld + row] (K-major)
template <bool TA, bool TB, int SQ>
static void run_case(const char* name, int M, int N, int K, int ones_row, int reps) {
    const int Ms = M - (ones_row >= 0 ? 1 : 0);                   // stored rows of A
    const int64_t lda = TA ? (Ms + 3) / 4 * 4 : (K + 3) / 4 * 4, ldb = TB ? (N + 3) / 4 * 4 : (K + 3) / 4 * 4;
    std::vector<float> hA((size_t)(TA ? K : Ms) * lda, 0.f), hA2(hA.size(), 0.f), hB((size_t)(TB ? K : N) * ldb, 0.f), hB2(hB.size(), 0.f);
    srand(1234);
    auto rnd = [] { return (float)(rand() % 2001 - 1000) / 1000.f; };
    auto ia = [&](int r, int k) { return TA ? (size_t)k * lda + r : (size_t)r * lda + k; };
    auto ib = [&](int r, int k) { return TB ? (size_t)k * ldb + r : (size_t)r * ldb + k; };
    for (int r = 0; r < Ms; ++r) for (int k = 0; k < K; ++k) { hA[ia(r, k)] = rnd(); hA2[ia(r, k)] = SQ == 2 ? hA[ia(r, k)] * hA[ia(r, k)] : fabsf(rnd()); }
    for (int r = 0; r < N; ++r) for (int k = 0; k < K; ++k) { hB[ib(r, k)] = rnd(); hB2[ib(r, k)] = SQ == 1 ? hB[ib(r, k)] * hB[ib(r, k)] : fabsf(rnd()); }
    float *A = dev(hA), *A2 = dev(hA2), *B = dev(hB), *B2 = dev(hB2), *o0, *o1;
    CK(hipMalloc(&o0, (size_t)M * N * 4 + 64)); CK(hipMalloc(&o1, (size_t)M * N * 4 + 64));
    std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m) {
            double s = 0, a = 0;
            for (int k = 0; k < K; ++k) {
                const double x = m == ones_row ? 1.0 : hA[ia(m, k)], x2 = m == ones_row ? 1.0 : hA2[ia(m, k)];
                s += x * hB[ib(n, k)] + 2.0 * x2 * hB2[ib(n, k)];
                a += fabs(x * hB[ib(n, k)]) + 2.0 * fabs(x2 * hB2[ib(n, k)]);
            }
            ref[(size_t)n * M + m] = s; mag[(size_t)n * M + m] = a;
        }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[5] = {"gemm_nt_v1", "v0 16x16", "v0 16x32", "v0 32x16", "v0 32x32"};
    for (int which = 0; which < 5; ++which) {
        float* out = which ? o1 : o0;
        EpiOut epi{out, M, N};
        const float *pA2 = SQ == 2 ? nullptr : A2, *pB2 = SQ == 1 ? nullptr : B2;
        auto go = [&]() -> int {
            switch (which) {
            case 0: {
                dim3 grid((M + 31) / 32, (N + 31) / 32);
                hipLaunchKernelGGL((gemm_nt_v1<float, true, 1, 2, EpiOut, TA, TB, SQ>), grid, dim3(256), 0, st, A, pA2, lda, B, pB2, ldb, M, N,
                                   (K + 31) / 32 * 32, K, ones_row, epi);
                return 0;
            }
            case 1: return launch_gemm_v0_tile<1, 1, true, EpiOut, TA, TB, SQ>(st, A, pA2, lda, B, pB2, ldb, M, N, K, epi, ones_row);
            case 2: return launch_gemm_v0_tile<1, 2, true, EpiOut, TA, TB, SQ>(st, A, pA2, lda, B, pB2, ldb, M, N, K, epi, ones_row);
            case 3: return launch_gemm_v0_tile<2, 1, true, EpiOut, TA, TB, SQ>(st, A, pA2, lda, B, pB2, ldb, M, N, K, epi, ones_row);
            default: return launch_gemm_v0_tile<2, 2, true, EpiOut, TA, TB, SQ>(st, A, pA2, lda, B, pB2, ldb, M, N, K, epi, ones_row);
            }
        };
        CK(hipMemsetAsync(out, 0xff, (size_t)M * N * 4, st));
        go();
        CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        std::vector<float> h((size_t)M * N);
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t i = 0; i < h.size(); ++i) worst = std::max(worst, fabs(h[i] - ref[i]) / (mag[i] + 1e-30));
        for (int i = 0; i < 20; ++i) go();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) go();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-36s M %4d N %4d K %4d  %-10s %6.2f us per launch   worst |err| / sum|ab| %.2e %s\n", name, M, N, K, names[which],
               ms * 1e3 / reps, worst, worst < 4e-6 ? "ok" : "WRONG");
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    if (argc > 2) {       // K sweep of the forward form
        for (int K : {64, 128, 256, 384, 512, 640, 784, 1024, 1536}) run_case<false, false, 1>("forward, K sweep", 400, 256, K, -1, reps);
        return 0;
    }
    run_case<false, false, 1>("forward 1 (w, w2 | x, x.x)", 400, 256, 784, -1, reps);
    run_case<false, false, 1>("forward 2", 400, 256, 400, -1, reps);
    run_case<true, false, 0>("gradInput 2 (w K-major | g, gv)", 400, 256, 400, -1, reps);
    run_case<true, true, 2>("accGrad 2 (x K-major, ones | g, gv)", 401, 400, 256, 400, reps);
    run_case<true, true, 2>("accGrad 1", 785, 400, 256, 784, reps);
    run_case<false, false, 1>("forward, ragged 100 x 50 x 37", 100, 50, 37, -1, reps);   // K % 4 != 0: the row pitch's padding is zero here
    run_case<true, true, 2>("accGrad, ragged 38 x 100 x 50", 38, 100, 50, 37, reps);
    return 0;
}
