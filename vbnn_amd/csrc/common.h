// common.h -- shared definitions of the gfx950 VBLinear library (device + host glue).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <type_traits>
#include "../../include/vbnn_hip.h"
#include "../../include/vbnn_philox.h"

// ---- the bf16 path's form of the contract's normals (include/vbnn_philox.h): the SAME Philox4x32-10 words for the same
// (seed, stream, layer, draw, row, quad), Box-Muller on them by the hardware's log2 / sqrt / sin / cos (v_log_f32, v_sqrt_f32,
// v_sin_f32 / v_cos_f32 take their angle in revolutions: exactly Box-Muller's 2 pi u2) instead of the bit-exact polynomial
// forms -- ~70 VALU slots per four normals against ~270. The values differ from the contract's by a few 1e-7 absolute
// (tests: max |dz| over 2^22 normals < 2e-6; three orders below the bf16 rounding of everything a bf16 forward does with
// them), so the fp32 path -- held sample for sample against the oracle -- keeps the exact form, and the bf16 path, held
// against rounding-point emulation at bf16 tolerances, takes this one: the draw was 23 us of a 4096 x 4096 forward launch.
#if defined(__HIPCC__)
// Box-Muller on two Philox words, hardware form (the words' meaning is vbnn_box_muller's, include/vbnn_philox.h)
__device__ __forceinline__ void vbnn_box_muller_hw(uint32_t x0, uint32_t x1, float* z0, float* z1) {
    const float u1 = (float)((x0 >> 8) + 1u) * 5.96046448e-8f;          // (0, 1], as vbnn_box_muller
    const float t = (float)(x1 >> 8) * 5.96046448e-8f;                  // the angle in revolutions, [0, 1)
    const float rr = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), ln = ln 2 . log2
    *z0 = rr * __builtin_amdgcn_cosf(t);
    *z1 = rr * __builtin_amdgcn_sinf(t);
}
__device__ __forceinline__ vbnn_f32x4 vbnn_normal4_hw(uint64_t seed, uint32_t stream, uint32_t layer, uint32_t draw, uint32_t row,
                                                      uint32_t quad) {
    const vbnn_u32x4 u = vbnn_philox4x32_10(quad, row, draw, (layer << 8) | stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    vbnn_f32x4 z;
    vbnn_box_muller_hw(u.v[0], u.v[1], &z.v[0], &z.v[1]);
    vbnn_box_muller_hw(u.v[2], u.v[3], &z.v[2], &z.v[3]);
    return z;
}
#endif

// A loop the compiler MUST unroll: `#pragma unroll` is silently dropped when the unrolled body exceeds LLVM's
// pragma-unroll-threshold (16 K instructions), and a rolled loop indexing a register array sends the WHOLE array to
// scratch (seen: gemm_nt_v2<.., EpiDw>'s generic epilogue, 528 B of scratch and 64 stores on every path).
template <int I, int N, class F>
__device__ __forceinline__ void vbnn_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        vbnn_static_for<I + 1, N>(f);
    }
}

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct vbnn_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    double* scratch;       // device scratch for block partial sums (prior / KL reductions)
    size_t scratch_doubles;
    unsigned* counters;    // arrival tickets of the in-launch second stages (vbnn_last_arriver); zero between launches
    int cu_budget = 0;     // > 0: the stream is CU-masked to this many compute units (vbnn_ctx_create_cu_budget)
};
// ticket slots
constexpr int VBNN_CNT_HEAD_FWD = 0, VBNN_CNT_TILES = 16, VBNN_CNT_TILES_MAX = 1008, VBNN_CNT_TOTAL = 1024;   // [16, 1024): one ticket per column tile of the head's in-launch finish

void vbnn_set_error(const char* fmt, ...);
inline int g_head_stream = -1;   // vbnn_debug_set key 10 (VBNN_DEBUG_HEAD_BACKWARD): the head's backward -- -1 by shape, 0 tile form, 1 streaming form whenever the operands allow

// hipFuncSetAttribute (the dynamic-LDS opt-in of the pipelined kernels) is per DEVICE: a per-instantiation flag that
// remembers "done" must remember it per device, or the second GPU of a process launches unconfigured kernels.
constexpr int VBNN_MAX_DEVICES = 64;
struct vbnn_per_device_flag {
    bool done[VBNN_MAX_DEVICES] = {};
    bool& operator[](int device) { return done[(device >= 0 && device < VBNN_MAX_DEVICES) ? device : 0]; }
};
// compute units the shape heuristics plan for: the CU budget of the context whose API call is running on this thread
// (vbnn_cu_scope, a thread-local: entered by the entry points that pick a kernel by shape), else the device's (all GPUs of a
// node are one model; read once from the current device instead of assuming MI355X's 256)
int vbnn_cu_count();
struct vbnn_cu_scope {
    int prev;
    explicit vbnn_cu_scope(const vbnn_ctx* c);
    ~vbnn_cu_scope();
    vbnn_cu_scope(const vbnn_cu_scope&) = delete;
    vbnn_cu_scope& operator=(const vbnn_cu_scope&) = delete;
};

#define VBNN_CHECK_HIP(expr)                                                             \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            vbnn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return VBNN_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define VBNN_REQUIRE(cond, msg)                                                          \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            vbnn_set_error("invalid argument: %s (%s) (%s:%d)", msg, #cond, __FILE__, __LINE__); \
            return VBNN_ERR_INVALID;                                                     \
        }                                                                                \
    } while (0)

#define VBNN_API_BEGIN try {
#define VBNN_API_END                                                                     \
    } catch (const std::exception& ex) {                                                 \
        vbnn_set_error("C++ exception: %s", ex.what());                                  \
        return VBNN_ERR_INVALID;                                                         \
    } catch (...) {                                                                      \
        vbnn_set_error("unknown C++ exception");                                         \
        return VBNN_ERR_INVALID;                                                         \
    }

static inline int vbnn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vbnn_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return VBNN_ERR_HIP;
    }
    return VBNN_OK;
}

// ---- in-launch second stage of a two-stage reduction ----------------------------------------
// vbnn_last_arriver returns true in exactly ONE of the `n` workgroups that share `counter` -- the last to arrive --
// once the partials every one of them stored with vbnn_store_wt before the call are readable by it; that workgroup
// then adds the partials in a fixed order (deterministic, no float atomics, no second launch: a tiny kernel costs
// ~4.5 us of an otherwise busy stream). The hand-off is cdna_hip_programming.md Guideline 16 R1 with a ticket in
// place of the flag: payload stored WRITE-THROUGH (sc1: relaxed agent-scope atomic stores of 4 / 8 bytes), so no
// release fence (an agent-scope release writes back the XCD's whole dirty L2: +40-50 us measured in kernels that
// had just written tens of MB), every storing wave drains vmcnt, barrier, one lane takes a relaxed agent-scope
// ticket; the workgroup whose ticket is n - 1 does ONE agent-scope acquire (drops this CU's L1), drains, barrier,
// then loads. The ticket is left at zero for the next launch (vbnn_ctx_create zeroes the array; launches on a
// context's stream do not overlap), and no partial is read anywhere in the launch before that acquire.
// Measured (r01): worth it for the classifier head's forward (256 workgroups; also removes two memsets per step);
// NOT for the streaming sweeps (prep_layer, column sums, head backward: 1.5-2 k workgroups) -- the per-workgroup
// store drain and the last arriver's serial latency chain at the kernel's tail cost more than the ~4.6 us launch
// of a separate finish kernel, so those keep their second launch.
#ifdef __HIPCC__
template <typename V> __device__ __forceinline__ void vbnn_store_wt(V* p, V v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void vbnn_store_wt2(float* p /* 8-byte aligned */, float a, float b) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename V> __device__ __forceinline__ V vbnn_load_wt(const V* p) {      // sc1 load: L2-served, never this CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool vbnn_last_arriver(unsigned* counter, unsigned n, int* flag_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == n - 1u) ? 1 : 0;
        if (last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // holds the barrier below until the invalidate is done
        }
        *flag_lds = last;
    }
    __syncthreads();
    return *flag_lds != 0;
}
#endif

// ---- element-type helpers ------------------------------------------------------------------
template <typename T> struct Elt;
template <> struct Elt<float> {
    static __device__ __forceinline__ float to(float v) { return v; }
    static __device__ __forceinline__ float from(float v) { return v; }
};
template <> struct Elt<bf16_t> {
    static __device__ __forceinline__ bf16_t to(float v) { return (bf16_t)v; }   // v_cvt_pk_bf16_f32, RNE, NaN-safe
    static __device__ __forceinline__ float from(bf16_t v) { return (float)v; }
};

// store 4 consecutive elements (p 4-element aligned when `vec` is true)
template <typename T>
__device__ __forceinline__ void store4(T* p, float a, float b, float c, float d, int valid, bool vec);
template <>
__device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d, int valid, bool vec) {
    if (vec && valid == 4) {
        *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
    } else {
        if (valid > 0) p[0] = a;
        if (valid > 1) p[1] = b;
        if (valid > 2) p[2] = c;
        if (valid > 3) p[3] = d;
    }
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d, int valid, bool vec) {
    if (vec && valid == 4) {
        *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    } else {
        if (valid > 0) p[0] = (bf16_t)a;
        if (valid > 1) p[1] = (bf16_t)b;
        if (valid > 2) p[2] = (bf16_t)c;
        if (valid > 3) p[3] = (bf16_t)d;
    }
}

// Output store of an epilogue in the SGPR-base form: `base` wave-uniform, `off` the lane's ELEMENT offset.
// (One place to change the stores' cache policy. Tried in r02 as inline asm with `sc1` / `nt`: write-through stores do not
// shorten the step -- the ~5 us between two large launches is not an L2 write-back -- and inline-asm VMEM behind a
// v_readfirstlane'd base needs its own `s_nop 4`: the hazard recogniser does not look inside asm, the `nt` build
// faulted on a stale SGPR base. Plain stores it is.)
template <class V, class T>
__device__ __forceinline__ void vbnn_store_out(T* base, unsigned off, V v) {
#ifdef VBNN_NT_STORES        // A/B (r03): the compiler's own nontemporal store -- 0.808-0.833 against 0.794-0.799 ms per wide step: the consumers of
                             // these outputs (the next launch's operands and epilogue reads) find less of them in the caches. Off.
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(base + off));
#else
    *reinterpret_cast<V*>(base + off) = v;
#endif
}

// The same for the parameter GRADIENTS (accGradParameters' outputs, 160 MB per wide step): nobody reads them inside the step --
// the update sweep or the exchange does -- so they leave with the nontemporal hint and do not displace what the step's next
// launches read (operand shadows, activations). Two A/B rounds on two boxes: 0.8046-0.8061 against 0.8075-0.8153 ms and
// 0.816-0.828 against 0.828-0.831 ms per wide step (the next forward alone 0.1966 -> 0.1895 ms), the training step with the
// update unchanged to slightly better (1.054-1.059 against 1.051-1.069 ms). (Nontemporal stores for EVERY epilogue output,
// vbnn_store_out above, cost 15-30 us: those outputs ARE the next launch's operands.) -DVBNN_NT_GRADS=0: plain stores.
#ifndef VBNN_NT_GRADS
#define VBNN_NT_GRADS 1
#endif
template <class V, class T>
__device__ __forceinline__ void vbnn_store_grad(T* base, unsigned off, V v) {
#if VBNN_NT_GRADS
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(base + off));
#else
    *reinterpret_cast<V*>(base + off) = v;
#endif
}

// ... and for outputs / operands whose next (or last) use lies a whole layer away: the noise factor r of a layer that is
// not the last (stored by its forward, read once by the gradInput epilogue of the layer above, ~half a step later) --
// -DVBNN_NT_R=0: plain.
#ifndef VBNN_NT_R
#define VBNN_NT_R 1
#endif
template <class V, class T>
__device__ __forceinline__ void vbnn_store_stream(T* base, unsigned off, V v, bool stream) {
#if VBNN_NT_R
    if (stream) { __builtin_nontemporal_store(v, reinterpret_cast<V*>(base + off)); return; }
#endif
    *reinterpret_cast<V*>(base + off) = v;
}
template <class V>
__device__ __forceinline__ V vbnn_load_last_use(const V* p) {
#if VBNN_NT_R
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// load 4 consecutive elements as floats (p 4-element aligned when `vec` is true); lanes past `valid` read 0
template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4], int valid, bool vec);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&v)[4], int valid, bool vec) {
    if (vec && valid == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (j < valid) ? p[j] : 0.f;
    }
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4], int valid, bool vec) {
    if (vec && valid == 4) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (j < valid) ? (float)p[j] : 0.f;
    }
}
