// gemm_lab.hip -- stand-alone playground for the pipelined GEMM main loop (diagnostic build, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DV2_DIAG tools/gemm_lab.hip -o gpurun_out/gemm_lab && gpurun_out/gemm_lab
// Random bf16 operands, trivial epilogue, event timing of both schedules and s_memtime segment shares.
#define V2_DIAG 1
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <vector>
#include <algorithm>
#include "../vbnn_amd/csrc/common.h"
void vbnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#include "../vbnn_amd/csrc/gemm_v2.h"

struct EpiSum {          // keeps both accumulators live with one 16-byte store per four outputs
    typedef bf16_t elem_t;
    float* out; int M, N;
    __device__ __forceinline__ bf16_t* t1_ptr() const { return nullptr; }
    __device__ __forceinline__ bf16_t* t2_ptr() const { return nullptr; }
    __device__ __forceinline__ int64_t t_ld() const { return 0; }
    __device__ __forceinline__ int m_dim() const { return M; }
    __device__ __forceinline__ int n_dim() const { return N; }
    template <bool ST>
    __device__ __forceinline__ void apply(int m, int n, f32x4 a1, f32x4 a2, float (&t1)[4], float (&t2)[4]) const {
        if (m < M && n < N) *reinterpret_cast<f32x4*>(out + (size_t)n * M + m) = a1 + a2;
    }
};
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    const size_t ea = (size_t)M * K, eb = (size_t)N * K;
    std::vector<unsigned short> ha(ea), hb(eb);
    srand(1);
    auto rnd = [] { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd();
    bf16_t *A, *A2, *B, *B2; float* out; unsigned long long* diag;
    CK(hipMalloc(&A, ea * 2)); CK(hipMalloc(&A2, ea * 2)); CK(hipMalloc(&B, eb * 2)); CK(hipMalloc(&B2, eb * 2));
    CK(hipMalloc(&out, (size_t)M * N * 4));
    CK(hipMemcpy(A, ha.data(), ea * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(A2, hb.data(), std::min(ea, eb) * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hb.data(), eb * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B2, ha.data(), std::min(ea, eb) * 2, hipMemcpyHostToDevice));
    const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
    CK(hipMalloc(&diag, (size_t)tiles * 8 * 4 * 8)); CK(hipMemset(diag, 0, (size_t)tiles * 8 * 4 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_v2_diag), &diag, sizeof(diag)));
    EpiSum epi{out, M, N};
    hipStream_t st = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int dual = 0; dual < 2; ++dual)
        for (int sched = 0; sched < 5; sched += 2) {
            g_v2_sched = sched;
            auto launch = [&] {
                if (dual) launch_gemm_v2<bf16_t, true, EpiSum>(st, A, A2, K, B, B2, K, M, N, K, epi);
                else launch_gemm_v2<bf16_t, false, EpiSum>(st, A, nullptr, K, B, nullptr, K, M, N, K, epi);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            std::vector<float> ts;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 10; ++i) launch();
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 10 * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            const double fl = 2.0 * M * N * K * (dual ? 2 : 1);
            printf("M=%d N=%d K=%d %s sched%d: median %.1f us  %.0f TF", M, N, K, dual ? "dual  " : "single", sched, ts[2], fl / ts[2] / 1e6);
            if (true) {
                std::vector<unsigned long long> hd((size_t)tiles * 8 * 4);
                CK(hipMemcpy(hd.data(), diag, hd.size() * 8, hipMemcpyDeviceToHost));
                double s[4] = {0, 0, 0, 0};
                for (size_t w = 0; w < (size_t)tiles * 8; ++w) for (int k = 0; k < 4; ++k) s[k] += (double)hd[w * 4 + k];
                const double tot = s[0] + s[1] + s[2] + s[3], nw = (double)tiles * 8;
                printf("   cycles/wave: wait_dma %.0f (%.0f%%) barrier %.0f (%.0f%%) issue %.0f (%.0f%%) read+mfma %.0f (%.0f%%)",
                       s[0] / nw, 100 * s[0] / tot, s[1] / nw, 100 * s[1] / tot, s[2] / nw, 100 * s[2] / tot, s[3] / nw, 100 * s[3] / tot);
            }
            printf("\n");
        }
    return 0;
}
