// split_lab.hip -- stand-alone timing of gemm_v3's kernels with in-kernel stamps (diagnostic build, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DV3_STAMP tools/split_lab.hip -o gpurun_out/split_lab && gpurun_out/split_lab
// Shapes of the 784 x 4096 gradient (M = 785 on a 1024 pitch, N = 4096, K = 4096), K-major operands, the real EpiDw functor.
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <vector>
#include <algorithm>
#include "lab/common.h"
void vbnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int vbnn_cu_count() { return 256; }
#include "lab/epilogues.h"
#include "lab/gemm_v2.h"
#include "lab/gemm_v3.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int I = argc > 1 ? atoi(argv[1]) : 784, O = argc > 2 ? atoi(argv[2]) : 4096, N = argc > 3 ? atoi(argv[3]) : 4096;
    const int M = I + 1, lda = (M + 255) / 256 * 256;
    std::vector<unsigned short> hx((size_t)N * lda), hg((size_t)N * O);
    srand(1);
    auto rnd = [] { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (int n = 0; n < N; ++n) for (int c = 0; c < lda; ++c) hx[(size_t)n * lda + c] = c < I ? rnd() : (c == I ? 0x3f80 : 0);
    for (auto& v : hg) v = rnd();
    bf16_t *x, *x2, *g, *gv; float *means, *lvars, *gmu, *glv, *gb; double* stats;
    CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&x2, hx.size() * 2)); CK(hipMalloc(&g, hg.size() * 2)); CK(hipMalloc(&gv, hg.size() * 2));
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(x2, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), hg.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(gv, hg.data(), hg.size() * 2, hipMemcpyHostToDevice));
    const size_t W = (size_t)O * I;
    CK(hipMalloc(&means, W * 4)); CK(hipMalloc(&lvars, W * 4)); CK(hipMalloc(&gmu, W * 4)); CK(hipMalloc(&glv, W * 4)); CK(hipMalloc(&gb, O * 4));
    CK(hipMemset(means, 0, W * 4)); CK(hipMemset(lvars, 0, W * 4));
    double hs[4] = {1.0, 0.0, 1e-3, (double)W};
    CK(hipMalloc(&stats, 32)); CK(hipMemcpy(stats, hs, 32, hipMemcpyHostToDevice));
    vbnn_ctx ctx{};
    ctx.stream = 0;
    CK(hipMalloc((void**)&ctx.counters, VBNN_CNT_TOTAL * 4)); CK(hipMemset(ctx.counters, 0, VBNN_CNT_TOTAL * 4));
    EpiDw e{};
    e.lrt = 1; e.scale = 1.f; e.accumulate = 0; e.vec = 1; e.lvars = lvars; e.grad_mu = gmu; e.grad_lv = glv; e.means = means;
    e.mu_s = x; e.var_s = x; e.ld_w = lda; e.stats = stats; e.B = 1e6f; e.S = 1.f; e.kl_scale = 1.f; e.gradBias = gb; e.I = I; e.O = O;
    const int tiles = ((M + 255) / 256) * (O / 256), nblk = tiles * 4;
    unsigned long long* stamp;
    CK(hipMalloc(&stamp, (size_t)nblk * 8 * 8)); CK(hipMemset(stamp, 0, (size_t)nblk * 8 * 8));
#ifdef V3_STAMP
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_v3_stamp), &stamp, sizeof(stamp)));
#endif
    g_v3_split = 1;
    auto launch = [&] {
        int st = launch_gemm_v3_split<bf16_t, EpiDw>(&ctx, x, x2, lda, g, gv, O, M, O, N, e);
        if (st != VBNN_OK) { printf("launch status %d\n", st); exit(1); }
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 10 * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("split launch I=%d O=%d N=%d: median %.1f us (%d workgroups)\n", I, O, N, ts[2], nblk);
    CK(hipMemset(stamp, 0, (size_t)nblk * 8 * 8));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)std::max(nblk, 256) * 8);
    CK(hipMemcpy(h.data(), stamp, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, tend = 0;
    for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, h[b * 8]); for (int k = 0; k < 5; ++k) tend = std::max(tend, h[b * 8 + k]); }
    auto med = [&](int k0, int k1, bool last_only) {
        std::vector<double> v;
        for (int b = 0; b < nblk; ++b) if (h[b * 8 + k1] && h[b * 8 + k0] && (!last_only || h[b * 8 + 3])) v.push_back((h[b * 8 + k1] - h[b * 8 + k0]) * 0.01);
        std::sort(v.begin(), v.end());
        if (v.empty()) { printf(" (none)"); return; }
        printf(" med %.1f max %.1f (n=%zu)", v[v.size() / 2], v.back(), v.size());
    };
    printf("kernel span (first start -> last stamp) %.1f us\n", (tend - t0) * 0.01);
    std::vector<double> starts;
    for (int b = 0; b < nblk; ++b) starts.push_back((h[b * 8] - t0) * 0.01);
    std::sort(starts.begin(), starts.end());
    printf("block start offsets: med %.1f max %.1f us\n", starts[nblk / 2], starts.back());
    printf("main loop  :"); med(0, 1, false); printf("\n");
    printf("store+tick :"); med(1, 2, false); printf("\n");
    printf("load other :"); med(2, 3, true); printf("\n");
    printf("epilogue   :"); med(3, 4, true); printf("\n");

    // ---- the forward launch (EpiFwd: Philox fold between the passes, ReLU / h / h.h / r epilogue) at K = I and K = O
    for (int K : {I, O}) {
        const int Kp = (K + 63) / 64 * 64;
        bf16_t *w, *w2, *xx, *xx2, *hh, *h2, *r; float* bias;
        const size_t ew = (size_t)O * Kp, ex = (size_t)N * Kp, eo = (size_t)N * O;
        CK(hipMalloc(&w, ew * 2)); CK(hipMalloc(&w2, ew * 2)); CK(hipMalloc(&xx, ex * 2)); CK(hipMalloc(&xx2, ex * 2));
        CK(hipMalloc(&hh, eo * 2)); CK(hipMalloc(&h2, eo * 2)); CK(hipMalloc(&r, eo * 2)); CK(hipMalloc(&bias, O * 4));
        std::vector<unsigned short> hw(ew), hxx(ex);
        for (auto& v : hw) v = rnd();
        for (auto& v : hxx) v = rnd() & 0x7fff;                       // the variance pair must be non-negative
        CK(hipMemcpy(w, hw.data(), ew * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(xx, hxx.data(), ex * 2, hipMemcpyHostToDevice));
        for (auto& v : hw) v &= 0x7fff;
        CK(hipMemcpy(w2, hw.data(), ew * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(xx2, hxx.data(), ex * 2, hipMemcpyHostToDevice));
        CK(hipMemset(bias, 0, O * 4));
        EpiFwd<bf16_t> f{};
        f.rpd = 0; f.bias = bias; f.noise = getenv("LAB_FAKE_NOISE") ? 2 : 1; f.seed = 3; f.layer = 0; f.draw = 1; f.row0 = 0;
        f.r_t = r; f.ld_r = O; f.r_vec = 1; f.relu = 1; f.h = hh; f.h2 = h2; f.ld_h = O; f.O = O; f.N = N;
        const int nb = (O / 256) * (N / 256);
        auto fl = [&] {
            int st = launch_gemm_v3<bf16_t, true, false, false, EpiFwd<bf16_t>>(&ctx, w, w2, Kp, xx, xx2, Kp, O, N, K, f);
            if (st != VBNN_OK) { printf("forward launch status %d\n", st); exit(1); }
        };
        for (int i = 0; i < 3; ++i) fl();
        CK(hipDeviceSynchronize());
        ts.clear();
        for (int rr = 0; rr < 5; ++rr) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) fl();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 10 * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("forward O=%d N=%d K=%d%s: median %.1f us\n", O, N, K, f.noise == 2 ? " [fake noise]" : "", ts[2]);
        CK(hipMemset(stamp, 0, (size_t)nblk * 8 * 8));
        fl();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), stamp, (size_t)nb * 8 * 8, hipMemcpyDeviceToHost));
        auto medf = [&](int k0, int k1) {
            std::vector<double> v;
            for (int b = 0; b < nb; ++b) if (h[b * 8 + k1] && h[b * 8 + k0]) v.push_back((h[b * 8 + k1] - h[b * 8 + k0]) * 0.01);
            std::sort(v.begin(), v.end());
            if (v.empty()) { printf(" (none)"); return; }
            printf(" med %.1f max %.1f", v[v.size() / 2], v.back());
        };
        printf("  pass 1   :"); medf(0, 5); printf("\n  fold     :"); medf(5, 6); printf("\n  pass 2   :"); medf(6, 1);
        printf("\n  epilogue :"); medf(1, 4); printf("\n");
    }
    // ---- the 4096-wide layer's two backward launches (K-major operands): accGradParameters and updateGradInput
    {
        const int Q = O;                                               // I = O = N = Q
        const size_t e2 = (size_t)Q * Q;
        bf16_t *a, *a2, *b, *b2, *xq, *rq, *gq, *gvq; float *mq, *lq, *gmq, *glq;
        for (bf16_t** p : {&a, &a2, &b, &b2, &xq, &rq, &gq, &gvq}) CK(hipMalloc(p, e2 * 2));
        for (float** p : {&mq, &lq, &gmq, &glq}) CK(hipMalloc(p, e2 * 4));
        std::vector<unsigned short> hv(e2);
        for (auto& v : hv) v = rnd();
        for (bf16_t* p : {a, a2, b, b2, xq, rq}) CK(hipMemcpy(p, hv.data(), e2 * 2, hipMemcpyHostToDevice));
        CK(hipMemset(mq, 0, e2 * 4)); CK(hipMemset(lq, 0, e2 * 4));
        auto report = [&](const char* what, auto&& fn) {
            for (int i = 0; i < 3; ++i) fn();
            CK(hipDeviceSynchronize());
            ts.clear();
            for (int rr = 0; rr < 5; ++rr) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 10; ++i) fn();
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 10 * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            printf("%s %d^3: median %.1f us\n", what, Q, ts[2]);
            CK(hipMemset(stamp, 0, (size_t)nblk * 8 * 8));
            fn();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), stamp, (size_t)256 * 8 * 8, hipMemcpyDeviceToHost));
            auto medf = [&](int k0, int k1) {
                std::vector<double> v;
                for (int bb = 0; bb < 256; ++bb) if (h[bb * 8 + k1] && h[bb * 8 + k0]) v.push_back((h[bb * 8 + k1] - h[bb * 8 + k0]) * 0.01);
                std::sort(v.begin(), v.end());
                if (v.empty()) { printf(" (none)"); return; }
                printf(" med %.1f max %.1f", v[v.size() / 2], v.back());
            };
            printf("  pass 1   :"); medf(0, 5); printf("\n  fold     :"); medf(5, 6); printf("\n  pass 2   :"); medf(6, 1);
            printf("\n  epilogue :"); medf(1, 4); printf("\n");
        };
        EpiDw d = e;
        d.lvars = lq; d.means = mq; d.mu_s = a; d.var_s = a2; d.ld_w = Q; d.grad_mu = gmq; d.grad_lv = glq; d.gradBias = nullptr; d.I = Q; d.O = Q;
        report("accGradParameters", [&] {
            int st = launch_gemm_v3<bf16_t, true, true, true, EpiDw>(&ctx, a, a2, Q, b, b2, Q, Q, Q, Q, d);
            if (st != VBNN_OK) { printf("dw launch status %d\n", st); exit(1); }
        });
        EpiDx<bf16_t> dx{};
        dx.dual = 1; dx.x = xq; dx.ld_x = Q; dx.relu_mask = 1; dx.r_prev_t = rq; dx.ld_r_prev = Q; dx.r_vec = 1;
        dx.g_prev = gq; dx.gv_prev = gvq; dx.ld_gp = Q; dx.I = Q; dx.N = Q;
        report("updateGradInput", [&] {
            int st = launch_gemm_v3<bf16_t, true, true, false, EpiDx<bf16_t>>(&ctx, a, a2, Q, b, b2, Q, Q, Q, Q, dx);
            if (st != VBNN_OK) { printf("dx launch status %d\n", st); exit(1); }
        });
    }
    return 0;
}