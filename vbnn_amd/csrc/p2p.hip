// p2p.hip -- the data-parallel exchange WITHOUT a collective library: a direct reduce-scatter + all-gather over peer-mapped
// gradient arenas (include/vbnn_hip.h, vbnn_p2p_*). SURVEY.md section 5's fallback for the case that RCCL serialises the SUM
// all-reduce of a 160 MB arena onto one ring: on an 8-GPU MI355X node every GPU has seven point-to-point xGMI links, a ring
// is bound by ONE of them (2 (7/8) 160 MB / 153 GB/s = 1.8 ms, twice the compute step), while a direct exchange moves an
// eighth of the bucket over EACH link at once in both phases (0.26 ms by the same arithmetic).
//
//   memory     every rank's arena is ONE hipMalloc allocation owned by this library and exported with hipIpcGetMemHandle; the
//              host hands the 128-byte handles (arena + flag page) round by its own means (as RCCL's unique id) and every rank
//              maps its peers' (hipIpcOpenMemHandle): kernels then read a peer's arena over xGMI like local memory.
//   all-reduce of arena[off, off + n), in place, on the exchange stream (ordered behind the compute stream by an event):
//     barrier   every rank's bucket is complete (each rank signals behind its own accGradParameters)
//     reduce-scatter   rank r sums chunk r of every rank's bucket IN RANK ORDER into its own arena (7 remote reads + 1 local
//                      per element, one link per peer) -- nobody else reads or writes that chunk of rank r's arena
//     barrier   every chunk is reduced
//     all-gather       rank r copies chunk q (q != r) from rank q's arena into its own
//     barrier   nobody still reads this rank's chunk: the arena may be overwritten (the next minibatch)
//   (r05: the third barrier is ONE launch per step, in vbnn_p2p_finish -- or earlier if a region is exchanged twice without a finish
//   in between: four launches per message instead of five)
//   The sum has ONE order (rank 0 + rank 1 + ...), so every rank holds bitwise the same arena afterwards.
//   coherence  data crosses devices only at KERNEL BOUNDARIES (a kernel's stores are written back when it ends, a kernel's
//              loads see them when it starts after the barrier kernel that observed the producer's signal): the arena is
//              ordinary coarse-grained memory at full speed. On this chip that is not an assumption about the runtime's
//              fence scopes but a necessity of the hardware: the eight XCDs' L2s are not coherent with EACH OTHER
//              (MI355X_MICROARCH.md), so consecutive kernels of one device already need every L2 written back to memory at
//              a kernel's end and non-coherent lines dropped at the next kernel's start -- a peer's read over xGMI is
//              served by the owner's memory side, behind that write-back. Only the barrier's flag words are polled while
//              a kernel runs: they live in an uncached (fine-grained) page and are accessed with system-scope atomics.
//   barrier    a one-workgroup kernel: lane p stores this rank's epoch into slot `rank` of peer p's flag page and polls slot p
//              of its own page. The poll is BOUNDED (about a second): a peer that never arrives makes the barrier give up and
//              raise the status word (vbnn_p2p_status) instead of hanging the device.
// The reference has nothing of this (single process, main.lua:142 sets BLAS threads): the exchange step is north_star's.
#include "common.h"

constexpr int P2P_MAX_WORLD = 8;
// A barrier's poll is bounded in WALL time (s_memrealtime: the 100 MHz constant counter), default 20 s, VBNN_P2P_TIMEOUT_S in the
// environment or vbnn_p2p_set_timeout: long enough for ordinary skew between ranks (a first-launch code-object load, a
// rank-0-only evaluation, a stalled data loader), short enough that a dead peer does not hold the device for minutes.
constexpr double P2P_TICKS_PER_S = 1.0e8;
constexpr double P2P_DEFAULT_TIMEOUT_S = 20.0;

struct P2PFlags { unsigned* page[P2P_MAX_WORLD]; };
struct P2PArenas { float* a[P2P_MAX_WORLD]; };

// flag page of a rank (uncached, 4 KiB): words [0, 8) = the epoch each peer has signalled, word 8 = this rank's status (the
// epoch of a barrier that failed, else 0), words [16, 24) = "peer q's exchange is dead" (written by q when ITS barrier fails)
constexpr int P2P_STATUS_WORD = P2P_MAX_WORLD, P2P_DEAD_WORD0 = 16;
// word 24: this rank's TRIGGER count -- written by a one-thread kernel on the COMPUTE stream behind the launch that fills a bucket,
// polled by the first barrier of that bucket's exchange (see p2p_trigger)
constexpr int P2P_TRIGGER_WORD = 24;
constexpr int P2P_DONE_WORD = 25;      // ... and the way back: bumped on the EXCHANGE stream behind the step's exit barrier, polled by a one-wave
                                       // kernel on the compute stream (vbnn_p2p_finish) -- the compute stream resumes within a microsecond of the
                                       // exchange's end instead of a marker packet's ~10 us later

// what a barrier needs, by value in a kernel's arguments
struct P2PSync {
    P2PFlags peers; unsigned* mine; unsigned* status;
    int rank, world; unsigned epoch; unsigned long long timeout_ticks;
    unsigned trig;            // != 0: before it signals, the barrier waits for this rank's own trigger word to reach this count
};

__global__ __launch_bounds__(64) void k_p2p_signal(unsigned* word, unsigned value) {
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (unbounded on purpose: what it waits for sits on the exchange stream behind kernels already enqueued, every one of them bounded)
__global__ __launch_bounds__(64) void k_p2p_wait(const unsigned* word, unsigned value) {
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) __builtin_amdgcn_s_sleep(8);
}

// One wave's worth of barrier: lane p < world signals peer p (when `signal`) and polls slot p of this rank's own page. Called by a
// whole wave; returns (to every lane) whether every peer arrived.
__device__ __forceinline__ bool p2p_wave_barrier(const P2PSync& s, bool signal) {
    const int p = threadIdx.x & 63;
    bool ok = true;
    if (s.trig != 0u) {
        // the bucket is complete on THIS rank once the compute stream's trigger kernel has run (it sits behind the launch that fills
        // the bucket; that launch's end wrote its data back). Bounded like every poll; a rank whose trigger never comes fails its
        // barrier AFTER signalling (below), so its peers do not wait out their own timeouts on it.
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(__hip_atomic_load(s.mine + P2P_TRIGGER_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - s.trig) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > s.timeout_ticks) { ok = false; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    const bool trig_ok = ok;
    if (p < s.world) {
        // (a rank whose exchange is dead still SIGNALS, so that no peer waits out its timeout on it; it does not poll again --
        // the exchange stays dead until the host clears the status on every rank)
        if (signal) __hip_atomic_store(s.peers.page[p] + s.rank, s.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!trig_ok || __hip_atomic_load(s.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
            ok = false;
            if (!trig_ok) {
                __hip_atomic_store(s.status, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (p != s.rank) __hip_atomic_store(s.peers.page[p] + P2P_DEAD_WORD0 + s.rank, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            ok = false;
            for (;;) {
                const unsigned seen = __hip_atomic_load(s.mine + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                // a peer whose own barrier failed says so in THIS rank's page: its data kernels have stopped, so what it holds are
                // not sums -- this rank's exchange is dead too, from this barrier on (the failure reaches every rank within one barrier)
                const unsigned dead = __hip_atomic_load(s.mine + P2P_DEAD_WORD0 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (dead != 0u) break;
                if ((int)(seen - s.epoch) >= 0) { ok = true; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > s.timeout_ticks) break;       // gave up: never hang the device
                __builtin_amdgcn_s_sleep(48);
            }
            if (!ok) {            // raise this rank's status, tell every peer (idempotent: every polling wave of a kernel may do it)
                __hip_atomic_store(s.status, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (p != s.rank) __hip_atomic_store(s.peers.page[p] + P2P_DEAD_WORD0 + s.rank, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    return __ballot(!ok) == 0ull;
}

// the barrier as a launch of its own: the ONE barrier per step that remains (vbnn_p2p_finish: "nobody still reads this rank's
// chunks -- the arena may be overwritten"), and between two exchanges of overlapping regions
__global__ __launch_bounds__(64) void k_p2p_barrier(P2PSync s) { (void)p2p_wave_barrier(s, true); }

// r05, priced and NOT kept: a phase's entry barrier INSIDE its data kernel (wave 0 of every workgroup polling, the other waves at
// s_barrier) to save the barrier launches -- an all-reduce is dependent launches on the exchange stream, 8-10 us each whatever they
// do, and the step's last message has nothing to hide behind. Correct (334 tests) and much SLOWER in the one-GPU stand-in: 1.026 ms
// per step against 0.891, the overlapped GEMMs 120 -> 186 us and 190 -> 237 us. A barrier's acquire / release at SYSTEM scope is
// cache maintenance on the executing XCD's L2 (write back, invalidate): done by one wave of a one-workgroup kernel it is noise;
// done by 256 workgroups' waves beside a GEMM it keeps throwing the GEMM's operand panels out of the L2s. A second form -- the flag
// page polled RELAXED (it is uncached: no cache maintenance) and exactly ONE acquire fence per workgroup once the last signal was
// seen -- costs the GEMMs nothing and buys 3 us of a 0.89 ms step (the exposed tail is the last message's data and the stream
// hand-offs, not the barrier launches: LAB_NOTES.md section 11): not kept either, two launches per message are not worth a second
// synchronisation protocol that no second device has ever run. What WAS kept from that exercise: the exit barrier ("nobody still
// reads this rank's chunk") is ONE launch per step, in vbnn_p2p_finish, not one per message -- four launches per all-reduce instead of five.

// The data kernels do NOTHING once a barrier of this rank has given up: a reduce-scatter on buckets a peer has not
// finished would put wrong sums into the gradient arena IN PLACE, where the next update reads them. With the status raised the
// arena keeps this rank's own (unsummed) gradients, every later phase of the exchange is a no-op, and the host finds the
// status at its next check (P2PExchange.check: before update(), at the end of a bench block, in loss_and_accuracy).
__device__ __forceinline__ bool p2p_dead(const unsigned* status) {
    return __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}

// ---- the data kernels. What they must be ("the exchange kernels are sized to fill the chip, the GEMMs they must overlap
// leave no room for them"): the launches they run beside are one 512-thread workgroup per CU with ALL 160 KiB of LDS and 209-226
// VGPRs at two waves per SIMD -- 2 x 232 of a SIMD's 512 registers -- so what is left on every SIMD is 48 registers, no LDS and
// six wave slots. A data kernel that wants to CO-RESIDE with such a workgroup (instead of waiting for a CU to drain and then
// keeping a GEMM tile off it: a 256-tile launch with one CU missing is two rounds) therefore uses no LDS and at most 48 VGPRs
// (__attribute__((amdgpu_num_vgpr(48))): the compiler would spill before it crossed that line; tools/kernel_regs.py shows 40-44),
// and gets its bandwidth from loads IN FLIGHT rather than from waves: the world is a template parameter, a lane issues the W
// 16-byte loads of an element group before it adds anything (rank order: ((a0 + a1) + a2) + ..., the bits r03's rolled loop gave),
// so ~100 workgroups keep seven links busy where the rolled loop (one dependent load at a time) needed a thousand.
// PACED (the one-GPU STAND-IN only, vbnn_p2p_standin): iteration k of a workgroup does not start before k / iters of the phase's
// wall-time budget has passed (s_memrealtime, s_sleep between looks) -- a wave that waits for a link looks to its CU like a wave
// that sleeps: it holds its slot and registers and issues nothing.
__device__ __forceinline__ void p2p_pace(unsigned long long t0, unsigned long long ticks, int64_t k, int64_t iters) {
    const unsigned long long due = t0 + (unsigned long long)((double)ticks * (double)k / (double)(iters > 0 ? iters : 1));
    while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(16);
}

// out[i] = sum over ranks p = 0 .. W - 1 of arena_p[base + i], i in [0, n): into this rank's own arena
template <int W, bool PACED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48)))
void k_p2p_reduce_scatter(P2PArenas peers, int rank, size_t base, int64_t n, int vec, const unsigned* status, unsigned long long pace_ticks) {
    if (p2p_dead(status)) return;
    float* out = peers.a[rank] + base;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long t0 = 0;
    if constexpr (PACED) t0 = __builtin_amdgcn_s_memrealtime();
    if (vec) {
        const int64_t n4 = n >> 2;
        const int64_t iters = (n4 + stride - 1) / stride;
        int64_t k = 0;
        for (int64_t i = first; i < n4; i += stride, ++k) {
            if constexpr (PACED) p2p_pace(t0, pace_ticks, k, iters);
            // four peers' loads in flight at a time (48 registers is all a wave gets beside two GEMM waves: see above); the
            // adds stay in rank order across the batches
            f32x4 s;
#pragma unroll
            for (int p0 = 0; p0 < W; p0 += 4) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (p0 + u < W) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(peers.a[p0 + u] + base) + i);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (p0 + u < W) { if (p0 + u == 0) s = v[u]; else s += v[u]; }
                __builtin_amdgcn_sched_barrier(0);            // (or the scheduler hoists the next batch's loads above these adds)
            }
            reinterpret_cast<f32x4*>(out)[i] = s;
        }
        for (int64_t i = (n4 << 2) + first; i < n; i += stride) {
            float s = peers.a[0][base + i];
#pragma unroll
            for (int p = 1; p < W; ++p) s += peers.a[p][base + i];
            out[i] = s;
        }
    } else {
        for (int64_t i = first; i < n; i += stride) {
            float s = peers.a[0][base + i];
#pragma unroll
            for (int p = 1; p < W; ++p) s += peers.a[p][base + i];
            out[i] = s;
        }
    }
}

// mine[base_q + i] = arena_q[base_q + i] for every peer chunk q != rank (blockIdx.y = q); four 16-byte loads in flight per lane
template <bool PACED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48)))
void k_p2p_all_gather(P2PArenas peers, int rank, size_t off, int64_t n, int64_t cs, int vec, const unsigned* status, unsigned long long pace_ticks) {
    const int q = blockIdx.y;
    if (q == rank || p2p_dead(status)) return;
    const int64_t c0 = (int64_t)q * cs;
    const int64_t cn = min(cs, n - c0);
    if (cn <= 0) return;
    const float* src = peers.a[q] + off + c0;
    float* dst = peers.a[rank] + off + c0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long t0 = 0;
    if constexpr (PACED) t0 = __builtin_amdgcn_s_memrealtime();
    if (vec) {
        const int64_t n4 = cn >> 2;
        const int64_t iters = (n4 + 4 * stride - 1) / (4 * stride);
        int64_t k = 0;
        for (int64_t i = first; i < n4; i += 4 * stride, ++k) {
            if constexpr (PACED) p2p_pace(t0, pace_ticks, k, iters);
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * stride < n4) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + i + u * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * stride < n4) reinterpret_cast<f32x4*>(dst)[i + u * stride] = v[u];
        }
        for (int64_t i = (n4 << 2) + first; i < cn; i += stride) dst[i] = src[i];
    } else {
        for (int64_t i = first; i < cn; i += stride) dst[i] = src[i];
    }
}

struct vbnn_p2p {
    vbnn_ctx* ctx;
    int rank, world;
    float* arena; size_t arena_floats;
    unsigned* flags;                 // this rank's flag page: slot p is written by peer p
    unsigned* status;                // device word (same uncached page, slot P2P_MAX_WORLD): epoch of a barrier that gave up, else 0
    P2PArenas arenas;                // every rank's arena as THIS process maps it (own pointer at [rank])
    P2PFlags pages;
    bool connected;
    hipStream_t stream;
    hipEvent_t ready, done;
    int64_t pending;
    unsigned epoch;
    unsigned long long timeout_ticks;
    bool have_stream, have_ready, have_done;
    bool flag_trigger;               // buckets are handed to the exchange stream through the trigger word (p2p_trigger) instead of an event
    unsigned trig_count, trig_pending;   // triggers issued so far; the count the NEXT entry barrier must wait for (0: none)
    unsigned done_count;             // finishes handed back through P2P_DONE_WORD
    bool need_final;                 // data kernels were enqueued since the last barrier launch: vbnn_p2p_finish owes the step's ONE barrier
    int n_regions; size_t reg_off[16]; int64_t reg_n[16];    // arena regions exchanged since then (an overlapping one forces that barrier early)
    int rs_blocks, ag_blocks;        // grids of the data kernels (vbnn_p2p_set_grid): workgroups of the reduce-scatter, of the all-gather PER PEER
    int sim_world;                   // > 1 (world == 1 only, vbnn_p2p_standin): the one-GPU stand-in of a sim_world-rank exchange
    double sim_GBps;                 // its pacing: inbound bytes per second a rank's links would deliver (0: unpaced)
};

// Default grids (r05, from the one-GPU stand-in, profiles/r05_overlap_standin.json): one reduce-scatter
// workgroup per CU, 32 all-gather workgroups per peer. Paced to 770 GB/s of inbound link bandwidth the stand-in's step is within
// noise for 128 .. 256 / 16 .. 37 (+0.133 .. 0.138 ms over the step without exchange calls, the overlapped GEMMs within 2 % of
// their undisturbed times); 1024 / 256 costs 30 us more (its waves crowd the half-height launch); below 64 / 8 the kernels cannot
// keep up with the links even from LOCAL memory (32 / 4: +0.38 ms). Remote latency is higher than local, so the larger of the
// equal grids is the default. VBNN_P2P_RS_BLOCKS / VBNN_P2P_AG_BLOCKS override at create, vbnn_p2p_set_grid later.
constexpr int P2P_DEFAULT_RS_BLOCKS = 256, P2P_DEFAULT_AG_BLOCKS = 32;
// (r05, the stand-in at 110 GB/s per link, two rounds on one box: 0.8979 / 0.8950 ms per step with event hand-offs, 0.8858 / 0.8825 with the
// trigger word; VBNN_P2P_FLAG_TRIGGER=0 at create: the events)
constexpr bool P2P_DEFAULT_FLAG_TRIGGER = true;

// the next phase's barrier arguments (every phase -- every data kernel, every barrier launch -- is one epoch)
static P2PSync p2p_next_sync(vbnn_p2p* p) {
    p->epoch += 1;
    P2PSync s;
    s.peers = p->pages; s.mine = p->flags; s.status = p->status; s.rank = p->rank; s.world = p->world; s.epoch = p->epoch;
    s.timeout_ticks = p->timeout_ticks;
    s.trig = p->trig_pending; p->trig_pending = 0;            // (the first barrier behind a trigger waits for it)
    return s;
}
// "the bucket is complete on this rank": everything enqueued on the compute stream so far comes first. An EVENT (record on the
// compute stream, wait on the exchange stream) costs the compute stream a marker packet -- ~8 us of bubble between the two launches it
// separates -- and the exchange stream ~12 us to wake up (kernel trace of the stand-in); the TRIGGER is a one-thread kernel on the
// compute stream that bumps a word of this rank's own flag page, and the exchange's first barrier -- already resident, polling -- goes
// on within a microsecond of it.
static int p2p_trigger(vbnn_p2p* p) {
    if (!p->flag_trigger || (p->world == 1 && p->sim_world <= 1)) {
        VBNN_CHECK_HIP(hipEventRecord(p->ready, p->ctx->stream));
        VBNN_CHECK_HIP(hipStreamWaitEvent(p->stream, p->ready, 0));
        return VBNN_OK;
    }
    p->trig_count += 1;
    hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(64), 0, p->ctx->stream, p->flags + P2P_TRIGGER_WORD, p->trig_count);
    const int st = vbnn_check_launch("k_p2p_signal");
    if (st == VBNN_OK) p->trig_pending = p->trig_count;       // (the entry barrier's wait for it is bounded either way)
    return st;
}
// a phase's ENTRY barrier: a one-workgroup launch in front of its data kernel
static void p2p_entry_barrier(vbnn_p2p* p) {
    hipLaunchKernelGGL(k_p2p_barrier, dim3(1), dim3(64), 0, p->stream, p2p_next_sync(p));
}
// the EXIT barrier as a launch: "every rank has finished every phase enqueued so far" -- nobody still reads this rank's arena
static int p2p_barrier(vbnn_p2p* p) {
    hipLaunchKernelGGL(k_p2p_barrier, dim3(1), dim3(64), 0, p->stream, p2p_next_sync(p));
    p->need_final = false; p->n_regions = 0;
    return vbnn_check_launch("k_p2p_barrier");
}
// a region about to be exchanged: one that overlaps a region exchanged since the last barrier (the same bucket twice in a row: a
// timing loop) needs that barrier first -- a peer may still be gathering from the chunk this rank is about to reduce into
static int p2p_claim_region(vbnn_p2p* p, size_t off, int64_t n) {
    bool clash = p->n_regions >= 16;
    for (int i = 0; i < p->n_regions && !clash; ++i)
        clash = off < p->reg_off[i] + (size_t)p->reg_n[i] && p->reg_off[i] < off + (size_t)n;
    if (clash) { const int st = p2p_barrier(p); if (st != VBNN_OK) return st; }
    p->reg_off[p->n_regions] = off; p->reg_n[p->n_regions] = n; p->n_regions += 1;
    p->need_final = true;
    return VBNN_OK;
}

template <bool PACED>
static void p2p_launch_rs(vbnn_p2p* p, const P2PArenas& t, int rank, int W, size_t base, int64_t n, int vec, unsigned long long pace) {
    p2p_entry_barrier(p);                                         // "every rank's region is complete"
    const int64_t want = ((n + 3) / 4 + 255) / 256;
    const unsigned nb = (unsigned)(want < p->rs_blocks ? (want > 0 ? want : 1) : p->rs_blocks);
#define VBNN_RS(Wc) case Wc: hipLaunchKernelGGL((k_p2p_reduce_scatter<Wc, PACED>), dim3(nb), dim3(256), 0, p->stream, t, rank, base, n, vec, p->status, pace); break;
    switch (W) { VBNN_RS(2) VBNN_RS(3) VBNN_RS(4) VBNN_RS(5) VBNN_RS(6) VBNN_RS(7) VBNN_RS(8) default: break; }
#undef VBNN_RS
}
template <bool PACED>
static void p2p_launch_ag(vbnn_p2p* p, const P2PArenas& t, int rank, int W, size_t off, int64_t n, int64_t cs, int vec, unsigned long long pace) {
    const int64_t want = ((cs + 3) / 4 + 1023) / 1024;          // (four 16-byte loads in flight per lane)
    const unsigned nb = (unsigned)(want < p->ag_blocks ? (want > 0 ? want : 1) : p->ag_blocks);
    p2p_entry_barrier(p);                                         // "every rank's chunk is complete (reduced / updated)"
    hipLaunchKernelGGL((k_p2p_all_gather<PACED>), dim3(nb, W), dim3(256), 0, p->stream, t, rank, off, n, cs, vec, p->status, pace);
}
// the stand-in's pointer tables: "peer q's arena" is THIS arena shifted by whole chunks, so that the bytes this device's memory
// serves and takes per phase are a real rank's -- reduce-scatter: the whole bucket read once (a real rank reads its own chunk and
// its seven peers read theirs from it), an eighth written; all-gather: seven chunks read, seven written
static P2PArenas p2p_standin_table(const vbnn_p2p* p, int64_t cs, bool gather) {
    P2PArenas t;
    for (int q = 0; q < P2P_MAX_WORLD; ++q) t.a[q] = nullptr;
    const int W = p->sim_world;
    // (gather: chunk q is filled from chunk q - 1, which is always a full one -- no read runs past a short last chunk)
    for (int q = 0; q < W; ++q) t.a[q] = gather ? p->arena + (int64_t)(((q + W - 1) % W) - q) * cs : p->arena + (int64_t)q * cs;
    if (gather) t.a[0] = p->arena;                                // (rank 0 = this rank: the destination)
    return t;
}
static unsigned long long p2p_standin_ticks(const vbnn_p2p* p, double bytes_received) {
    return p->sim_GBps > 0.0 ? (unsigned long long)(bytes_received / (p->sim_GBps * 1e9) * P2P_TICKS_PER_S) : 0ull;
}

static void p2p_release(vbnn_p2p* p) {           // everything create / connect may have made, in any state of completion
    for (int q = 0; q < p->world; ++q) {
        if (q == p->rank) continue;
        if (p->arenas.a[q]) (void)hipIpcCloseMemHandle(p->arenas.a[q]);
        if (p->pages.page[q]) (void)hipIpcCloseMemHandle(p->pages.page[q]);
        p->arenas.a[q] = nullptr; p->pages.page[q] = nullptr;
    }
    if (p->have_ready) (void)hipEventDestroy(p->ready);
    if (p->have_done) (void)hipEventDestroy(p->done);
    if (p->have_stream) (void)hipStreamDestroy(p->stream);
    if (p->arena) (void)hipFree(p->arena);
    if (p->flags) (void)hipFree(p->flags);
    delete p;
}

static_assert(VBNN_P2P_HANDLE_BYTES == 2 * sizeof(hipIpcMemHandle_t), "the handle the host passes round: arena + flag page");

extern "C" int vbnn_p2p_create(vbnn_ctx* ctx, int rank, int world, size_t arena_floats, vbnn_p2p** out, void** arena_out, void* handle_out) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && out && arena_out && handle_out, "null argument");
    VBNN_REQUIRE(world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world, "rank / world (at most 8 ranks: one node)");
    VBNN_REQUIRE(arena_floats > 0, "arena size");
    VBNN_CHECK_HIP(hipSetDevice(ctx->device));
    vbnn_p2p* p = new vbnn_p2p();
    p->ctx = ctx; p->rank = rank; p->world = world; p->arena_floats = arena_floats; p->pending = 0; p->epoch = 0; p->connected = world == 1;
    p->arena = nullptr; p->flags = nullptr; p->have_stream = p->have_ready = p->have_done = false;
    p->rs_blocks = P2P_DEFAULT_RS_BLOCKS; p->ag_blocks = P2P_DEFAULT_AG_BLOCKS; p->sim_world = 0; p->sim_GBps = 0.0;
    p->need_final = false; p->n_regions = 0;
    p->flag_trigger = P2P_DEFAULT_FLAG_TRIGGER; p->trig_count = 0; p->trig_pending = 0; p->done_count = 0;
    if (const char* e = getenv("VBNN_P2P_FLAG_TRIGGER")) p->flag_trigger = e[0] == '1';
    if (const char* e = getenv("VBNN_P2P_RS_BLOCKS")) { const int v = atoi(e); if (v > 0 && v <= 4096) p->rs_blocks = v; }
    if (const char* e = getenv("VBNN_P2P_AG_BLOCKS")) { const int v = atoi(e); if (v > 0 && v <= 4096) p->ag_blocks = v; }
    for (int q = 0; q < P2P_MAX_WORLD; ++q) { p->arenas.a[q] = nullptr; p->pages.page[q] = nullptr; }
    {
        double secs = P2P_DEFAULT_TIMEOUT_S;
        if (const char* e = getenv("VBNN_P2P_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) secs = v; }
        p->timeout_ticks = (unsigned long long)(secs * P2P_TICKS_PER_S);
    }
    hipError_t e = hipMalloc((void**)&p->arena, arena_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemset(p->arena, 0, arena_floats * sizeof(float));
    // the flag page is polled by running kernels of OTHER devices: uncached (fine-grained) memory
    if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&p->flags, 4096, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(p->flags, 0, 4096);
    if (e != hipSuccess) {
        vbnn_set_error("p2p arena / flag page: %s", hipGetErrorString(e));
        p2p_release(p);
        return VBNN_ERR_NOMEM;
    }
    p->status = p->flags + P2P_STATUS_WORD;
    p->arenas.a[rank] = p->arena; p->pages.page[rank] = p->flags;
    hipIpcMemHandle_t h[2];
    memset(h, 0, sizeof h);
    if (world > 1) {
        e = hipIpcGetMemHandle(&h[0], p->arena);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&h[1], p->flags);
    }
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e == hipSuccess) { e = hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, greatest); p->have_stream = e == hipSuccess; }
    {
        // The polled hand-offs need the two streams to make progress INDEPENDENTLY: a kernel polling at the head of a hardware queue
        // that also carries the stream it waits for would wait for ever. Streams of different priority never share a queue; a
        // context whose own stream already has the highest priority keeps the event hand-offs.
        int prio = least;
        if (ctx->stream && hipStreamGetPriority(ctx->stream, &prio) == hipSuccess && prio == greatest) p->flag_trigger = false;
        if (greatest == least) p->flag_trigger = false;              // no priorities on this device: no such guarantee
    }
    if (e == hipSuccess) { e = hipEventCreateWithFlags(&p->ready, hipEventDisableTiming); p->have_ready = e == hipSuccess; }
    if (e == hipSuccess) { e = hipEventCreateWithFlags(&p->done, hipEventDisableTiming); p->have_done = e == hipSuccess; }
    if (e != hipSuccess) {
        vbnn_set_error("p2p export / stream: %s", hipGetErrorString(e));
        p2p_release(p);                                           // (the stream and the events made so far as well)
        return VBNN_ERR_HIP;
    }
    memcpy(handle_out, h, sizeof h);
    *arena_out = p->arena;
    *out = p;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_p2p_connect(vbnn_p2p* p, const void* all_handles) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p && (all_handles || p->world == 1), "argument");
    if (p->connected) return VBNN_OK;
    VBNN_CHECK_HIP(hipSetDevice(p->ctx->device));
    const hipIpcMemHandle_t* h = static_cast<const hipIpcMemHandle_t*>(all_handles);
    for (int q = 0; q < p->world; ++q) {
        if (q == p->rank) continue;
        // every mapping is RECORDED as soon as it is open, so that vbnn_p2p_destroy closes whatever a failed connect left
        // (a second connect after a failure skips what is already mapped)
        if (!p->arenas.a[q]) {
            void* a = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&a, h[2 * q], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { vbnn_set_error("hipIpcOpenMemHandle(rank %d's arena): %s", q, hipGetErrorString(e)); return VBNN_ERR_HIP; }
            p->arenas.a[q] = static_cast<float*>(a);
        }
        if (!p->pages.page[q]) {
            void* f = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&f, h[2 * q + 1], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { vbnn_set_error("hipIpcOpenMemHandle(rank %d's flag page): %s", q, hipGetErrorString(e)); return VBNN_ERR_HIP; }
            p->pages.page[q] = static_cast<unsigned*>(f);
        }
    }
    p->connected = true;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_p2p_allreduce(vbnn_p2p* p, size_t offset_floats, int64_t n) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p && n > 0 && offset_floats + (size_t)n <= p->arena_floats, "bucket outside the arena");
    VBNN_REQUIRE(p->connected, "vbnn_p2p_connect first");
    // everything enqueued on the compute stream so far (the accGradParameters launch that fills the bucket) comes first
    { const int tst = p2p_trigger(p); if (tst != VBNN_OK) return tst; }
    p->pending += 1;
    if (p->world == 1 && p->sim_world <= 1) return VBNN_OK;       // the sum over one rank (still ordered: finish waits for the stream)
    const bool sim = p->world == 1;                               // the one-GPU stand-in of a sim_world-rank exchange (vbnn_p2p_standin)
    const int W = sim ? p->sim_world : p->world;
    const int64_t cs = ((n + W - 1) / W + 3) / 4 * 4;             // chunk length: a multiple of 4 floats
    const int vec = (offset_floats % 4 == 0) ? 1 : 0;             // (hipMalloc bases are 256-byte aligned in every process)
    const int64_t c0 = (int64_t)p->rank * cs, cn = n - c0 < cs ? n - c0 : cs;
    int st = p2p_claim_region(p, offset_floats, n);
    if (st != VBNN_OK) return st;
    // four launches: barrier, reduce-scatter, barrier, all-gather; the exit barrier ("nobody still reads this rank's chunk") is the
    // step's one barrier launch, in vbnn_p2p_finish
    if (sim) {
        const int64_t last = n - (int64_t)(W - 1) * cs;           // the stand-in reads every chunk over the length of the shortest
        p2p_launch_rs<true>(p, p2p_standin_table(p, cs, false), 0, W, offset_floats, last > 0 ? (last < cs ? last : cs) : 0, vec, p2p_standin_ticks(p, 4.0 * (double)cn * (W - 1)));
        p2p_launch_ag<true>(p, p2p_standin_table(p, cs, true), 0, W, offset_floats, n, cs, vec, p2p_standin_ticks(p, 4.0 * (double)cs * (W - 1)));
    } else {
        // (a rank whose chunk is empty -- a bucket shorter than the world -- still issues the phase: its barrier is what its peers wait for)
        p2p_launch_rs<false>(p, p->arenas, p->rank, W, offset_floats + (size_t)(cn > 0 ? c0 : 0), cn > 0 ? cn : 0, vec, 0ull);
        p2p_launch_ag<false>(p, p->arenas, p->rank, W, offset_floats, n, cs, vec, 0ull);
    }
    return vbnn_check_launch("vbnn_p2p_allreduce");
    VBNN_API_END
}

// ---- the two phases as calls of their own (the sharded-update exchange): equal chunks, rank r's at offset + r * n_per_rank.
// reduce-scatter: barrier (every rank's region is complete) -> rank r sums chunk r of every rank's arena into its own ->
// barrier (nobody still reads this rank's other chunks: they may be overwritten). all-gather: barrier (every rank's chunk is
// complete) -> copy the peers' chunks -> barrier.
extern "C" int vbnn_p2p_reduce_scatter(vbnn_p2p* p, size_t offset_floats, int64_t n_per_rank) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p && n_per_rank > 0 && offset_floats + (size_t)n_per_rank * (size_t)p->world <= p->arena_floats, "region outside the arena");
    VBNN_REQUIRE(p->connected, "vbnn_p2p_connect first");
    { const int tst = p2p_trigger(p); if (tst != VBNN_OK) return tst; }
    p->pending += 1;
    if (p->world == 1) return VBNN_OK;
    const size_t c0 = offset_floats + (size_t)p->rank * (size_t)n_per_rank;
    const int vec = (c0 % 4 == 0 && offset_floats % 4 == 0 && n_per_rank % 4 == 0) ? 1 : 0;
    const int st = p2p_claim_region(p, offset_floats, n_per_rank * (int64_t)p->world);
    if (st != VBNN_OK) return st;
    p2p_launch_rs<false>(p, p->arenas, p->rank, p->world, c0, n_per_rank, vec, 0ull);      // (entry barrier + kernel; the exit barrier is vbnn_p2p_finish's)
    return vbnn_check_launch("vbnn_p2p_reduce_scatter");
    VBNN_API_END
}

extern "C" int vbnn_p2p_all_gather(vbnn_p2p* p, size_t offset_floats, int64_t n_per_rank) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p && n_per_rank > 0 && offset_floats + (size_t)n_per_rank * (size_t)p->world <= p->arena_floats, "region outside the arena");
    VBNN_REQUIRE(p->connected, "vbnn_p2p_connect first");
    { const int tst = p2p_trigger(p); if (tst != VBNN_OK) return tst; }
    p->pending += 1;
    if (p->world == 1) return VBNN_OK;
    const int vec = (offset_floats % 4 == 0 && n_per_rank % 4 == 0) ? 1 : 0;
    const int st = p2p_claim_region(p, offset_floats, n_per_rank * (int64_t)p->world);
    if (st != VBNN_OK) return st;
    p2p_launch_ag<false>(p, p->arenas, p->rank, p->world, offset_floats, n_per_rank * (int64_t)p->world, n_per_rank, vec, 0ull);
    return vbnn_check_launch("vbnn_p2p_all_gather");
    VBNN_API_END
}

// grids of the data kernels: workgroups of the reduce-scatter, workgroups of the all-gather per peer (0: keep)
extern "C" int vbnn_p2p_set_grid(vbnn_p2p* p, int rs_blocks, int ag_blocks_per_peer) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p && rs_blocks >= 0 && rs_blocks <= 4096 && ag_blocks_per_peer >= 0 && ag_blocks_per_peer <= 4096, "grid");
    if (rs_blocks > 0) p->rs_blocks = rs_blocks;
    if (ag_blocks_per_peer > 0) p->ag_blocks = ag_blocks_per_peer;
    return VBNN_OK;
    VBNN_API_END
}

// LAB, one rank only: from now on vbnn_p2p_allreduce runs what a rank of a `sim_world`-rank exchange would run -- the same
// barriers, the same two data kernels with the same grids and register footprint on the same high-priority stream behind the same
// event -- against "peers" that are this arena itself, shifted by whole chunks (the bytes this device's memory serves and takes
// per phase are a real rank's), each phase PACED to the wall time `inbound_GBps` of link bandwidth would need for the bytes a
// rank receives in it (0: unpaced, i.e. at local-memory speed). The arena afterwards holds nothing meaningful: timing only
// (tools/overlap_standin.py: what the exchange costs the launches it overlaps). sim_world = 0 switches it off.
extern "C" int vbnn_p2p_standin(vbnn_p2p* p, int sim_world, double inbound_GBps) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p && p->world == 1, "the stand-in is a world-of-one lab mode");
    VBNN_REQUIRE(sim_world == 0 || (sim_world >= 2 && sim_world <= P2P_MAX_WORLD), "sim_world: 0 (off) or 2 .. 8");
    VBNN_REQUIRE(inbound_GBps >= 0.0, "inbound_GBps");
    p->sim_world = sim_world; p->sim_GBps = inbound_GBps;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_p2p_finish(vbnn_p2p* p) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p, "null p2p");
    if (p->pending == 0) return VBNN_OK;
    if (p->need_final) {                     // the step's one barrier launch: every rank has finished every phase -- the arena may be overwritten
        const int st = p2p_barrier(p);
        if (st != VBNN_OK) return st;
    }
    if (p->flag_trigger && !(p->world == 1 && p->sim_world <= 1)) {
        p->done_count += 1;
        p->pending = 0;
        hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(64), 0, p->stream, p->flags + P2P_DONE_WORD, p->done_count);
        const int st = vbnn_check_launch("vbnn_p2p_finish");
        if (st != VBNN_OK) return st;       // no poll without its signal: it would hold the context's stream for ever
        hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, p->ctx->stream, p->flags + P2P_DONE_WORD, p->done_count);
        return vbnn_check_launch("vbnn_p2p_finish");
    }
    VBNN_CHECK_HIP(hipEventRecord(p->done, p->stream));
    VBNN_CHECK_HIP(hipStreamWaitEvent(p->ctx->stream, p->done, 0));
    p->pending = 0;
    return VBNN_OK;
    VBNN_API_END
}

// blocks until the exchange stream is idle; *gave_up = the epoch of a barrier whose peers never arrived (0: none)
extern "C" int vbnn_p2p_status(vbnn_p2p* p, int* rank, int* world, unsigned* gave_up) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p, "null p2p");
    if (rank) *rank = p->rank;
    if (world) *world = p->world;
    if (gave_up) {
        VBNN_CHECK_HIP(hipStreamSynchronize(p->stream));
        unsigned v = 0;
        VBNN_CHECK_HIP(hipMemcpy(&v, p->status, sizeof v, hipMemcpyDeviceToHost));
        *gave_up = v;
    }
    return VBNN_OK;
    VBNN_API_END
}

// the bound of a barrier's poll, in seconds (default 20, or VBNN_P2P_TIMEOUT_S at vbnn_p2p_create); takes effect for the barriers
// enqueued after the call
extern "C" int vbnn_p2p_set_timeout(vbnn_p2p* p, double seconds) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p && seconds > 0.0 && seconds < 3600.0, "timeout in (0, 3600) seconds");
    p->timeout_ticks = (unsigned long long)(seconds * P2P_TICKS_PER_S);
    return VBNN_OK;
    VBNN_API_END
}

// after a failure the host has re-synchronised the ranks by its own means (a host barrier AFTER every rank's exchange stream
// has drained, and BEFORE any rank's next exchange) and refilled the arena: every rank calls this, then the exchange may be used again
extern "C" int vbnn_p2p_clear_status(vbnn_p2p* p) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(p, "null p2p");
    VBNN_CHECK_HIP(hipStreamSynchronize(p->stream));
    const unsigned zero[P2P_MAX_WORLD] = {};
    VBNN_CHECK_HIP(hipMemcpy(p->status, zero, sizeof(unsigned), hipMemcpyHostToDevice));
    VBNN_CHECK_HIP(hipMemcpy(p->flags + P2P_DEAD_WORD0, zero, sizeof zero, hipMemcpyHostToDevice));     // the peers' "dead" words too
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_p2p_destroy(vbnn_p2p* p) {
    VBNN_API_BEGIN
    if (!p) return VBNN_OK;
    (void)hipSetDevice(p->ctx->device);
    if (p->have_stream) (void)hipStreamSynchronize(p->stream);
    p2p_release(p);
    return VBNN_OK;
    VBNN_API_END
}
