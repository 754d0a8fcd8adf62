// poa_kernel.hip — gfx950 kernels and launch logic of the batched window POA.
//
// Execution model (MI355X: 256 CUs, 64-lane waves, 160 KiB LDS per CU):
//   * one lane group (the whole wavefront from class 1 on; 16 or 32 lanes in class 0) owns one window from its first sequence
//     to its consensus; the window's state stays in that group's LDS slice (HBM scratch for the two largest
//     classes), the only algorithmic HBM traffic is the packed input (read once) and the consensus (written once);
//   * workgroups are single waves and persistent: each pulls window indices from a per-class queue
//     with one atomic per window (~0.3-1 us, against >= 50 us of work per window), so occupancy is bounded
//     only by LDS bytes per window;
//   * plan kernels bin windows into size classes (poa_classes.hpp) and cost buckets; a window that still
//     overflows its class is re-queued by the kernel to the next class (no host round trip, no CPU fallback);
//   * the three common classes run concurrently on three streams, their mop-up passes behind them, then the
//     rare classes with grids sized from the plan's counts (poa_run).
// MFMA is not used: the work is integer max/+ over irregular <=128-wide rows with a serial graph
// update between sequences (see DESIGN.md for the roofline evidence).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "poa_classes.hpp"
#include "poa_kernel.hpp"
#include "poa_giant.hpp"
#include <cstring>

namespace hypo {

// ------------------------------------------------------------------------------------------------
// plan: size class + cost bucket per window, then a counting sort so that every class queue is ordered
// by decreasing cost.  Groups of one wave dequeue neighbouring indices, i.e. windows of similar size:
// the wave's loops then run for about the same trip counts in all of its groups, and the expensive
// windows start first (short tail).
// ------------------------------------------------------------------------------------------------
struct ClassLimits { int lmax, nmax, dircells, ringcells, seqmax, cpl; };

template <class Cfg> __host__ __device__ constexpr ClassLimits limits_of() {
    return ClassLimits{Cfg::LMAX, Cfg::NMAX, Cfg::DIRCELLS, Cfg::RINGCELLS, Cfg::SEQMAX, Cfg::CPL};
}

constexpr int PLAN_THREADS = 256;
constexpr int PLAN_STAGE = 8192;          // arm lengths staged per workgroup (32 KiB of LDS)
constexpr int PLAN_LANES = 4;             // lanes per window in poa_plan_count_kernel
constexpr int PLAN_WPB = PLAN_THREADS / PLAN_LANES;

__device__ __forceinline__ uint32_t plan_key_from(const HypoWindow& W, uint32_t maxarm, uint32_t changes, bool* trivial, int min_short_class = 0) {
    const uint32_t narm = W.n_internal + W.n_prefix + W.n_suffix;
    // longest sequence the window will align (markers included) and a node estimate: the first sequence's chain plus what the
    // arms that differ from their predecessor are expected to add (`changes`: differing packed bytes between consecutive internal
    // arms of equal length, about two per new node).  The kernel re-queues a window that outgrows its class all the same.
    uint32_t maxlen = (W.n_internal == 0 || W.type != HYPO_WIN_SHORT) ? W.draft_len + 2 : 0;
    if (narm) maxlen = maxarm + 2 > maxlen ? maxarm + 2 : maxlen;
#define HYPO_PLAN_GROW_Q 2            // expected new nodes per differing byte, in quarters
    const uint32_t slack = maxlen / 16 + 3, grow = changes * HYPO_PLAN_GROW_Q / 4 + 3;
    const uint32_t est_nodes = maxlen + (grow > slack ? grow : slack);
    const ClassLimits lim[kNumPoaClasses] = {
#define HYPO_LIM(ID, CFG) limits_of<CFG>(),
        HYPO_FOR_EACH_CLASS(HYPO_LIM)
#undef HYPO_LIM
    };
    int cls = kNumPoaClasses - 1;
    for (int c = 0; c < kNumPoaClasses; ++c) {
        const uint32_t S = (maxlen + 1 + lim[c].cpl - 1) / lim[c].cpl * lim[c].cpl;
        if ((int)maxlen <= lim[c].lmax && (int)est_nodes <= lim[c].nmax && (int)narm + 1 <= lim[c].seqmax &&
            (uint64_t)est_nodes * S <= (uint64_t)lim[c].dircells && lim[c].ringcells / (int)S >= 6) { cls = c; break; }
    }
    if (W.type != HYPO_WIN_SHORT && cls < kFirstLongClass) cls = kFirstLongClass;
    if (W.type == HYPO_WIN_SHORT && cls < min_short_class) cls = min_short_class;      // hypo_gpu_set_option("poa_min_class"): parity sweeps run small windows through the code of the larger classes
    *trivial = W.n_empty > narm || narm < 2;
    // cost ~ rows x sequences; bucket 0 = most expensive of the class
    const uint32_t cost = *trivial ? 0u : maxlen * (narm + 1);
    const uint32_t cmax = (uint32_t)lim[cls].lmax * (uint32_t)(lim[cls].seqmax < 64 ? lim[cls].seqmax : 64);
    uint32_t b = (uint32_t)(((uint64_t)cost * kPlanBuckets) / (cmax ? cmax : 1));
    b = b >= (uint32_t)kPlanBuckets ? kPlanBuckets - 1 : b;
    return (uint32_t)cls * kPlanBuckets + (kPlanBuckets - 1 - b);
}

// PLAN_LANES lanes per window (a window's arms are compared with their predecessors one after the other, two dependent HBM
// round trips each: with one lane per window that chain was the kernel's 78 us on the C2 batch; the lanes of a window take
// every PLAN_LANES-th arm and fold their counts with two shuffles).  The arm lengths of a workgroup's windows are normally one
// contiguous range of arm_len: it is staged through LDS with coalesced loads, then every lane scans its share there.
// The (class, bucket) histogram is accumulated in LDS and flushed with one global atomic per non-empty key.
__global__ void __launch_bounds__(PLAN_THREADS)
poa_plan_count_kernel(PoaParams P, PoaQueues Q, uint32_t n_windows) {
    __shared__ uint32_t lens[PLAN_STAGE];
    __shared__ uint32_t hist[kNumPoaClasses * kPlanBuckets];
    __shared__ uint32_t amin, amax, ntriv;
    const uint32_t w = blockIdx.x * PLAN_WPB + threadIdx.x / PLAN_LANES, sub = threadIdx.x % PLAN_LANES;
    for (int i = threadIdx.x; i < kNumPoaClasses * kPlanBuckets; i += PLAN_THREADS) hist[i] = 0;
    if (threadIdx.x == 0) { amin = 0xffffffffu; amax = 0; ntriv = 0; }
    __syncthreads();
    HypoWindow W{};
    uint32_t narm = 0;
    if (w < n_windows) {
        W = P.windows[w];
        narm = W.n_internal + W.n_prefix + W.n_suffix;
        // a descriptor whose arms run past the arm table is answered HYPO_ST_INVALID by whichever class gets it (Poa::run_window);
        // the plan must not walk its "arms" either (a wrapped count made one lane read 4 G arm lengths: 20 s for one window)
        if ((uint64_t)W.n_internal + W.n_prefix + W.n_suffix + (uint64_t)W.first_arm > P.n_arms) { narm = 0; W.n_internal = W.n_prefix = W.n_suffix = 0; }
        if (narm && sub == 0) { atomicMin(&amin, W.first_arm); atomicMax(&amax, W.first_arm + narm); }
    }
    __syncthreads();
    const uint32_t a0 = amin, a1 = amax;
    const bool staged = a1 > a0 && a1 - a0 <= (uint32_t)PLAN_STAGE;
    if (staged) for (uint32_t i = threadIdx.x; i < a1 - a0; i += PLAN_THREADS) lens[i] = P.arm_len[a0 + i];
    __syncthreads();
    {
        uint32_t maxarm = 0;
        if (staged) { for (uint32_t a = sub; a < narm; a += PLAN_LANES) { const uint32_t l = lens[W.first_arm - a0 + a]; maxarm = l > maxarm ? l : maxarm; } }
        else { for (uint32_t a = sub; a < narm; a += PLAN_LANES) { const uint32_t l = P.arm_len[W.first_arm + a]; maxarm = l > maxarm ? l : maxarm; } }
        // arm diversity: packed bytes (4 bases each) in which an internal arm differs from the arm before it, equal lengths only.
        // A read error changes one byte against the predecessor and one against the successor, at any error rate: about two
        // differing bytes per node the window's graph will grow beyond its first chain.  (13.6 MB of arms on the C2 batch, read
        // once here and once more by the size-class kernels.)
        uint32_t changes = 0;
        if (W.type == HYPO_WIN_SHORT && (uint64_t)W.first_arm + narm <= P.n_arms) {
            auto arm_at = [&](uint32_t a, uint32_t* l_out) -> const uint8_t* {
                const uint32_t l = staged ? lens[W.first_arm - a0 + a] : P.arm_len[W.first_arm + a];
                const uint64_t o = P.arm_off[W.first_arm + a];
                const uint32_t nb = (l + 3) / 4;
                *l_out = l;
                return (o <= P.arms2_bytes && nb <= P.arms2_bytes - o) ? P.arms2 + o : nullptr;
            };
            for (uint32_t a = 1 + sub; a < W.n_internal; a += PLAN_LANES) {          // arm a against arm a - 1
                uint32_t l, pl;
                const uint8_t* p = arm_at(a, &l);
                const uint8_t* q = arm_at(a - 1, &pl);
                const uint32_t nb = (l + 3) / 4;
                if (p && q && l == pl) {                       // eight bytes per (unaligned) load, the rest one by one
                    typedef uint64_t __attribute__((aligned(1))) u64u;
                    uint32_t b = 0;
                    for (; b + 8 <= nb; b += 8) {
                        uint64_t x = *(const u64u*)(p + b) ^ *(const u64u*)(q + b);
                        x |= x >> 4; x |= x >> 2; x |= x >> 1;
                        changes += (uint32_t)__popcll(x & 0x0101010101010101ull);
                    }
                    for (; b < nb; ++b) changes += p[b] != q[b];
                }
            }
        }
        for (int d = 1; d < PLAN_LANES; d <<= 1) {            // (every lane of the workgroup is here: lanes without a window carry zeros)
            changes += (uint32_t)__shfl_xor((int)changes, d, PLAN_LANES);
            const uint32_t om = (uint32_t)__shfl_xor((int)maxarm, d, PLAN_LANES);
            maxarm = om > maxarm ? om : maxarm;
        }
        if (w < n_windows && sub == 0) {
            bool trivial;
            const uint32_t key = plan_key_from(W, maxarm, changes, &trivial, (P.flags >> POA_MIN_CLASS_SHIFT) & 3);
            Q.keys[w] = (uint16_t)key;
            Q.carry[w] = 0;                                   // no spill yet (poa_class_kernel sets it when it re-queues the window)
            // Every window's status starts as "not written" (HYPO_ST_UNWRITTEN) and its length as 0: a window no kernel answered cannot come
            // back looking like HYPO_ST_OK with whatever the result buffers held before (the host mirror treats the sentinel as fatal).
            // Class 3 is polled (poa_class_kernel<.., POLL>): its queue slots read "unpublished" until the scatter kernel or a re-queue
            // fills them.  (Three hipMemsetAsync calls until round 6: a fill kernel of ~6 us each in front of every call.)
            P.out_status[w] = 0xff;
            P.out_len[w] = 0;
            Q.items[(size_t)3 * Q.stride + w] = kQueueUnpublished;
            atomicAdd(&hist[key], 1u);
            if (trivial) atomicAdd(&ntriv, 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kNumPoaClasses * kPlanBuckets; i += PLAN_THREADS)
        if (hist[i]) atomicAdd(&Q.hist[i], hist[i]);
    if (threadIdx.x == 0 && ntriv) atomicAdd((unsigned long long*)&Q.stats->n_trivial, (unsigned long long)ntriv);
}

__global__ void poa_plan_scan_kernel(PoaQueues Q) {       // one wave: lane c scans the kPlanBuckets entries of class c
    const int c = threadIdx.x;
    if (c < kNumPoaClasses) {
        uint32_t acc = 0;
        for (int b = 0; b < kPlanBuckets; ++b) {
            const int k = c * kPlanBuckets + b;
            Q.start[k] = acc;
            acc += Q.hist[k];
        }
        Q.count[c] = acc;
        Q.planned[c] = acc;          // windows the plan put into the class (before any re-queue)
        Q.head2[c] = acc;
    }
    // the plan's counts for the host (poa_run sizes the NEXT call's grids from them; the first call of a context waits for them), written
    // straight into page-locked host memory: a 32-byte hipMemcpyAsync was a copy command of its own on the stream
    if (c < 8) {
        uint32_t v = 0;
        if (c < kNumPoaClasses) v = Q.count[c];
        __hip_atomic_store(Q.host + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Scatter with one global atomic per (workgroup, key): local ranks come from an LDS histogram.
__global__ void __launch_bounds__(PLAN_THREADS)
poa_plan_scatter_kernel(PoaQueues Q, uint32_t n_windows) {
    __shared__ uint32_t cnt[kNumPoaClasses * kPlanBuckets];
    __shared__ uint32_t base[kNumPoaClasses * kPlanBuckets];
    const uint32_t w = blockIdx.x * PLAN_THREADS + threadIdx.x;
    for (int i = threadIdx.x; i < kNumPoaClasses * kPlanBuckets; i += PLAN_THREADS) cnt[i] = 0;
    __syncthreads();
    uint32_t key = 0, local = 0;
    if (w < n_windows) { key = Q.keys[w]; local = atomicAdd(&cnt[key], 1u); }
    __syncthreads();
    for (int i = threadIdx.x; i < kNumPoaClasses * kPlanBuckets; i += PLAN_THREADS)
        if (cnt[i]) base[i] = Q.start[i] + atomicAdd(&Q.cursor[i], cnt[i]);
    __syncthreads();
    if (w < n_windows) Q.items[(size_t)(key / kPlanBuckets) * Q.stride + base[key] + local] = w;
}

// ------------------------------------------------------------------------------------------------
// persistent per-class kernel
// ------------------------------------------------------------------------------------------------
// All arguments travel as one struct: the kernel never names it, it reads the kernel-argument segment through an
// opaque constant-address-space pointer at the (rare) points of use.  Passing PoaParams / PoaQueues as ordinary by-value
// arguments made the compiler keep ~20 64-bit pointers in SGPRs across the whole persistent loop; the row loop of
// Poa::align then spilled its own scalars into VGPR lanes and read them back with v_readlane on every row.
struct PoaKArgs {
    PoaParams P;
    PoaQueues Q;
    int cls;
    char* scratch;
    uint32_t* head;
    const uint32_t* bound;
    char* dirg;             // Cfg::DIRG: direction-code slices, one per resident group
    uint32_t expected;      // POLL: lane groups of the producing classes' launches (Q.done[0 .. cls) reach this when all have exited)
};
typedef const PoaKArgs __attribute__((address_space(4)))* PoaKArgPtr;
__device__ __forceinline__ PoaKArgPtr fresh(PoaKArgPtr p) { asm volatile("" : "+s"(p)); return p; }

// The kernel's side of Poa::fetch_next (poa_core.hpp): where the queue of this launch is.  The pointers are read from the
// kernel-argument segment at the point of use (nothing lives in registers between windows but the two bounds the loop had before).
struct PoaPrefetch {
    static constexpr bool enabled = true;
    // queue slots of this launch (what *bound holds does not change while a launch of an LDS class outside a polling launch runs:
    // first passes end at `planned`, everything else starts when its producers are done) and where the re-queued windows begin
    __device__ __forceinline__ static uint32_t count() { return *ka()->bound; }
    __device__ __forceinline__ static uint32_t planned() { const PoaKArgPtr k = ka(); return k->Q.planned[k->cls]; }
    __device__ __forceinline__ static PoaKArgPtr ka() { return fresh((PoaKArgPtr)__builtin_amdgcn_kernarg_segment_ptr()); }
    __device__ __forceinline__ uint32_t claim() const { return atomicAdd(ka()->head, 1u); }
    __device__ __forceinline__ uint32_t item(uint32_t idx) const { const PoaKArgPtr k = ka(); return k->Q.items[(size_t)k->cls * k->Q.stride + idx]; }
    __device__ __forceinline__ uint32_t carry(uint32_t w) const { return HYPO_LD_DEV(&ka()->Q.carry[w]); }
};
#define HYPO_PREFETCH_NEXT 1

#define HYPO_C4_WAVES 2
// (class 3 — the wide windows, and everything re-queued — is held to four waves per SIMD: it sat at 129 registers with the lazy rank order)
// (class 0's sub-wave geometries sit at the edge of three waves per SIMD, 168 VGPRs: the bound keeps them there)
template <class Cfg> struct PoaMinWaves { static constexpr int value = Cfg::HYBRID ? HYPO_C4_WAVES : (Cfg::DIRG ? 4 : (Cfg::GW < 64 ? 3 : 1)); };
// POLL: the kernel runs NEXT to the classes that feed it and takes re-queued windows as they arrive: it leaves when every lane
// group of every launch of the lower classes has exited (Q.done) and the queue is drained.  Every launch it waits for is
// submitted BEFORE it, so whatever the streams' mapping to hardware queues is, nothing it depends on can be stuck behind it; a
// time limit without progress (kPollLimitTicks) is the belt to those braces, and a regular launch of the class follows anyway.
constexpr uint64_t kPollLimitTicks = 50u * 1000u * 1000u;     // wall_clock64() runs at 100 MHz: 0.5 s
template <class Cfg, bool USE_LDS, bool POLL = false>
__global__ void __launch_bounds__(64, PoaMinWaves<Cfg>::value) poa_class_kernel(PoaKArgs /*read through the kernarg segment*/) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PoaKArgPtr ka = (PoaKArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int GW = Cfg::GW;
    constexpr int GPW = 64 / GW;                       // groups per wave
    const int wl = (int)(threadIdx.x & 63);
    const int grp = wl / GW;
    Grp<GW> g{wl & (GW - 1)};
    char* mem = USE_LDS ? smem + (size_t)grp * PoaLayout<Cfg>::BYTES
                        : fresh(ka)->scratch + ((size_t)blockIdx.x * GPW + grp) * PoaLayout<Cfg>::BYTES;
    char* fast = smem + (size_t)grp * PoaLayout<Cfg>::FAST_BYTES;     // hybrid classes only
    char* dirg = Cfg::DIRG ? fresh(ka)->dirg + ((size_t)blockIdx.x * GPW + grp) * PoaLayout<Cfg>::DIRG_BYTES : nullptr;
    const int cls = fresh(ka)->cls;
    // wave-time this launch takes (what poa_run's wave shares of the NEXT call are made from): -start now, +end at exit, so
    // that nothing lives in a register in between (a polling launch mostly waits: not counted)
    if (!POLL && USE_LDS && wl == 0) atomicAdd((unsigned long long*)(fresh(ka)->Q.work + cls), 0ull - (unsigned long long)wall_clock64());
    if (!POLL && USE_LDS && cls == 0 && blockIdx.x == 0 && wl == 0) fresh(ka)->Q.work[6] = (uint64_t)GW;      // class 0's geometry in the call these times come from
    const uint32_t count = POLL ? 0u : *fresh(ka)->bound;   // queue slots [.., *bound) are final when this launch starts (POLL: the queue grows)
    const uint32_t planned = fresh(ka)->Q.planned[cls];      // slots from here on hold re-queued windows (they may come with a spill)
    const PoaParamRef P{&ka->P};
    // per-wave totals, flushed once at exit.  The LDS classes keep them in 32 bits (a wave of those sees at most a few thousand
    // windows of < 1 M cells; in the sub-wave classes every group-uniform value is a vector register per lane)
    // The LDS classes keep the totals in the group's stat block in LDS (Poa::ACC_*, 32 bits: a wave of those sees at most a few
    // thousand windows of < 1 M cells); in registers they were live across every window — a vector register each in the sub-wave
    // classes.  The HBM-scratch classes (two waves per SIMD anyway) keep 64-bit registers.
    // the LDS classes outside a polling launch fetch a group's next window into its LDS block (PoaPrefetch, Poa::fetch_next)
    constexpr bool PF = HYPO_PREFETCH_NEXT && USE_LDS && !POLL;
    typedef typename std::conditional<PF, Poa<Cfg, PoaPrefetch>, Poa<Cfg>>::type PoaT;
    uint64_t cells = 0, aligns = 0, abytes = 0, n_reused = 0, n_thr = 0, c_scored = 0, c_thr = 0;
    uint32_t n_ok = 0, n_esc = 0, n_fail = 0, n_carried = 0;
    uint32_t carry_in = 0;                                      // Q.carry value of the window in hand
#ifdef HYPO_PHASE_TIMERS
    uint64_t tph[PH_N] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t dbg[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // [17]: cycles in account(), [18], [19]: of those, in the spill-pool allocation / in Poa::spill
    const uint64_t tstart = (uint64_t)clock64();
#endif
    // what happens to a window once Poa::run / step has returned something other than RES_CONTINUE
    auto account = [&](PoaT& poa, uint32_t w, int rc) {
        uint32_t* const stt = poa.stat;
        if constexpr (USE_LDS) {
            if (g.lane == 0) {
                if (rc == RES_OK) {      // reference-equivalent work of FINISHED windows only
                    stt[PoaT::ACC_CELLS] += stt[PoaT::ST_CELLS]; stt[PoaT::ACC_ALIGNS] += stt[PoaT::ST_ALIGNS];
                    stt[PoaT::ACC_REUSED] += stt[PoaT::ST_REUSED]; stt[PoaT::ACC_THR] += stt[PoaT::ST_XHITS];
                }
                stt[PoaT::ACC_CSCORED] += stt[PoaT::ST_CSCORED]; stt[PoaT::ACC_CTHR] += stt[PoaT::ST_CEXACT];     // executed work, finished or not
            }
        } else {
            if (rc == RES_OK) { cells += stt[PoaT::ST_CELLS]; aligns += stt[PoaT::ST_ALIGNS]; n_reused += stt[PoaT::ST_REUSED]; n_thr += stt[PoaT::ST_XHITS]; }
            c_scored += stt[PoaT::ST_CSCORED]; c_thr += stt[PoaT::ST_CEXACT];
        }
#ifdef HYPO_PHASE_TIMERS
        for (int i = 0; i < PH_N; ++i) tph[i] += poa.tphase[i];
        dbg[0] += poa.rows_done; dbg[1] += stt[PoaT::ST_ALIGNS] - stt[PoaT::ST_REUSED]; dbg[2] += stt[PoaT::ST_REUSED]; dbg[3] += poa.topo_runs; dbg[4] += poa.cons_serial; dbg[5] += poa.rows_slow; dbg[6] += poa.exact_tries; dbg[7] += stt[PoaT::ST_XHITS]; dbg[8] += poa.guided_hits; dbg[9] += poa.rows_scored_n; dbg[10] += poa.topo_dfs; dbg[11] += poa.topo_fast; dbg[12] += poa.one_sub_hits; dbg[13] += poa.cols_hits; dbg[14] += poa.topo_inserts; dbg[15] += poa.lazy_updates; dbg[16] += poa.tie_sorts;
#endif
        if (rc == RES_OK) {
            if (g.lane == 0) {                                  // algorithmic bytes, SURVEY.md 8(d)
                if constexpr (PF) {                              // (the window left its arm bytes and its answer's length in the stat block)
                    stt[PoaT::ACC_ABYTES] += stt[PoaT::CUR_STATIC] + stt[PoaT::ST_OLEN] + stt[PoaT::ST_ARMB];
                    stt[PoaT::ACC_NOK] += 1;
                } else {
                    const HypoWindow W = P->windows[w];
                    const uint32_t narm = W.n_internal + W.n_prefix + W.n_suffix;
                    uint32_t a = (uint32_t)((W.draft_len + 1) / 2 + 16 + 8 * (1 + narm) + P->out_len[w]);
                    const uint32_t* alen = P->arm_len;
                    for (uint32_t t = 0; t < narm; ++t) a += (alen[W.first_arm + t] + 3) / 4;
                    if constexpr (USE_LDS) { stt[PoaT::ACC_ABYTES] += a; stt[PoaT::ACC_NOK] += 1; } else abytes += a;
                }
            }
            if constexpr (!USE_LDS) ++n_ok;
        } else if ((rc == RES_OVERFLOW || rc == RES_UNSUPPORTED) && cls + 1 < kNumPoaClasses) {
            // Where to: every SHORT class hands over to class 3 (kRequeueClass), the one SHORT class that runs next to the others and
            // polls its queue — a window re-queued from class 0 used to wait for class 0 to finish, run in class 1's mop-up pass,
            // and, if it outgrew that too, wait again.  Class 3 and later hand over to the next class; a window that projects
            // to more nodes than class 3 holds goes straight to the LONG class.
            // (The projection Poa::run_short leaves in ST_NEED no longer picks the class: an estimate beyond class 3's node table
            // sent windows to the LONG class, 20 x slower per row, that class 3 would have finished; a window that really
            // outgrows class 3 arrives there with its graph all the same.)
            const int to = cls < kRequeueClass ? kRequeueClass : cls + 1;
            // what the window takes along: the graph of the sequences it has been through (Poa::spill) when the step that failed
            // left one, else the spill it came with
            uint32_t cv = 0;
            if (rc == RES_OVERFLOW && stt[PoaT::ST_CKIND] != PoaT::CARRY_NONE) {
                const uint32_t sz16 = poa.spill_size() >> 4;
                // One fetch-add on a 64-bit cursor (it cannot wrap; a request that does not fit is refused and every later one with it: the
                // pool is full then).  Rounds 3-5 claimed with a compare-and-swap loop so that a refused request left the cursor alone:
                // when a fifth of a batch's windows outgrow their class at about the same time, every success invalidated the value
                // thousands of other waves were about to swap in — a quadratic storm of retries on one address, 160 of the 162 ms of such a
                // batch (round 6, profiles/r06_grid_diag.txt: the 8 % cells of the SURVEY 8(d) grid).
                const uint32_t cap16 = fresh(ka)->Q.spill_cap16;
                uint32_t off = 0xffffffffu;
#ifdef HYPO_PHASE_TIMERS
                const uint64_t tc0 = (uint64_t)clock64();
#endif
                if (g.lane == 0) {
                    const unsigned long long got = atomicAdd(fresh(ka)->Q.spill_used, (unsigned long long)sz16);
                    if (got + sz16 <= (unsigned long long)cap16) off = (uint32_t)got;
                }
                off = (uint32_t)g.shfl((int)off, 0);
#ifdef HYPO_PHASE_TIMERS
                const uint64_t tc1 = (uint64_t)clock64();
                dbg[18] += tc1 - tc0;
#endif
                if (off != 0xffffffffu) {
                    poa.spill((uint8_t*)fresh(ka)->Q.spill + (size_t)off * 16);
#ifdef HYPO_PHASE_TIMERS
                    dbg[19] += (uint64_t)clock64() - tc1;
#endif
                    cv = off + 1;
                    if constexpr (USE_LDS) { if (g.lane == 0) stt[PoaT::ACC_NCARRIED] += 1; } else ++n_carried;
                }
            } else if (rc == RES_OVERFLOW && stt[PoaT::ST_CPASS]) cv = carry_in;
            if (g.lane == 0) {
                HYPO_ST_DEV(&fresh(ka)->Q.carry[w], cv);
                HYPO_RELEASE_STORES();                          // the spill and carry[w] (device-coherent stores, grp.hpp) have landed before the queue entry is published
                const uint32_t slot = atomicAdd(&fresh(ka)->Q.count[to], 1u);
                __hip_atomic_store(&fresh(ka)->Q.items[(size_t)to * fresh(ka)->Q.stride + slot], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if constexpr (USE_LDS) { if (g.lane == 0) stt[PoaT::ACC_NESC] += 1; } else ++n_esc;
        } else if (rc == RES_OVERFLOW || rc == RES_UNSUPPORTED) {
            // beyond the last table-driven class: the window goes to size class 6 (poa_giant.hpp), which runs behind this launch and takes
            // its windows from the queue region class 0 no longer needs (classes 0-2 were joined before this class started)
            if (g.lane == 0) {
                const uint32_t slot = atomicAdd(&fresh(ka)->Q.count[kGiantClass], 1u);
                __hip_atomic_store(&fresh(ka)->Q.items[slot], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (g.lane == 0) {
                P->out_len[w] = 0;
                P->out_status[w] = (uint8_t)(rc == RES_UNDEFINED ? HYPO_ST_UNDEFINED : (rc == RES_INVALID ? HYPO_ST_INVALID : HYPO_ST_CAPACITY));
            }
            if constexpr (USE_LDS) { if (g.lane == 0) stt[PoaT::ACC_NFAIL] += 1; } else ++n_fail;
        }
    };
    auto aload = [](const uint32_t* p) -> uint32_t { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto dequeue = [&](uint32_t* w) -> bool {
        uint32_t idx = 0;
        if constexpr (POLL) {
            const uint32_t* const cnt = fresh(ka)->Q.count + cls;
            const uint32_t* const done = fresh(ka)->Q.done;
            const uint32_t expected = fresh(ka)->expected;
            auto producers_done = [&]() -> bool {
                uint32_t d = 0;
                for (int c = 0; c < cls; ++c) d += aload(done + c);
                return d >= expected;
            };
            uint64_t t0 = wall_clock64();
            for (;;) {
                // Once the classes that feed this one are done the queue is final, and this launch leaves when it is empty.  What is left
                // then (the last waves of class 2 publish theirs just before they count themselves done) is taken at once, next to the
                // regular launch (poa_run starts it at that moment): until round 6 a polling wave left as soon as the producers were
                // done, the regular launch started only behind the last wave of this one, and a large window handed to it then ran
                // 5 ms behind an otherwise finished call (the non-i.i.d. batch: 14.2 ms in half of the calls, 11 in the others).
                const bool final_queue = producers_done();
                if (final_queue) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (the count read below is the one the producers left)
                const uint32_t h = aload(fresh(ka)->head);
                const uint32_t c = aload(cnt);
                if (final_queue && h >= c) return false;
                if (h < c) {
                    // Claimed with a compare-and-swap on the value just seen to be below the count: a claimed slot is always one a
                    // producer has already counted (its entry is at most a few instructions away).  An unconditional atomicAdd let
                    // several idle pollers that saw the same new entry claim slots BEHIND the count; one that then timed out left a
                    // slot claimed for good, and the window later published into it was never run.
                    uint32_t got = kQueueUnpublished;
                    if (g.lane == 0 && atomicCAS(fresh(ka)->head, h, h + 1u) == h) got = h;
                    got = (uint32_t)g.shfl((int)got, 0);
                    if (got == kQueueUnpublished) continue;                           // another group took it: look again
                    idx = got;
                    // the slot is this group's now; its entry may still be on its way (the producer bumps the count first).  It is
                    // never abandoned: the producer that counted it is a running wave of a launch submitted before this one.
                    const uint32_t* const slot = fresh(ka)->Q.items + (size_t)cls * fresh(ka)->Q.stride + idx;
                    for (;;) {
                        const uint32_t v = aload(slot);
                        if (v != kQueueUnpublished) { *w = v; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // (carry[w] and the spill were written before the entry and are read with device-coherent loads)
                    break;
                }
                if (wall_clock64() - t0 > kPollLimitTicks) return false;
                __builtin_amdgcn_s_sleep(127);
            }
        } else {
            // a plain look first: the waves of an empty class (most launches of the rare classes) leave without queueing up
            // on one atomic counter
            if (__atomic_load_n(fresh(ka)->head, __ATOMIC_RELAXED) >= count) return false;
            if (g.lane == 0) idx = atomicAdd(fresh(ka)->head, 1u);
            idx = (uint32_t)g.shfl((int)idx, 0);
            if (idx >= count) return false;
            *w = fresh(ka)->Q.items[(size_t)cls * fresh(ka)->Q.stride + idx];
        }
        carry_in = idx >= planned ? HYPO_LD_DEV(&fresh(ka)->Q.carry[*w]) : 0u;
        return true;
    };
    PoaT poa(g, P, mem, fast, dirg);
    // The groups of a wavefront (GPW > 1) take windows in lock step: all dequeue, all run, all write their consensus.
    // (Letting a finished group open its next window while its neighbours are still aligning was measured and is
    // slower: the dequeue + descriptor + arm staging round trips of one group then stall the other three, 4x as often.)
    if constexpr (PF) {
        // A window arrives through Poa::fetch_next: queue slot, descriptor, output range and carry word are in the group's LDS
        // block when it starts, and its statistics need no HBM read when it ends.
        uint32_t* const stt = poa.stat;
        poa.fetch_next();
        for (;;) {
            // (group-uniform values read from LDS: Grp::uniform makes them scalars again in the 64-lane classes)
            const uint32_t w = (uint32_t)g.uniform((int)stt[PoaT::NX_WIDX]);
            if (w == PoaT::NX_NONE) break;
            HypoWindow W;
            { uint32_t* const wd = (uint32_t*)&W; HYPO_UNROLL for (int i = 0; i < 10; ++i) wd[i] = (uint32_t)g.uniform((int)stt[PoaT::NX_W0 + i]); }
            carry_in = (uint32_t)g.uniform((int)stt[PoaT::NX_CARRY]);
            const uint32_t oc = g.lane < 4 ? stt[PoaT::NX_OFF + g.lane] : 0u;
            g.sync();
            if (g.lane < 4) stt[PoaT::CUR_OFF + g.lane] = oc;
            if (g.lane == 4) stt[PoaT::CUR_STATIC] = (uint32_t)((W.draft_len + 1) / 2 + 16 + 8 * (1 + W.n_internal + W.n_prefix + W.n_suffix));
            const int rc = poa.run_window(w, W, carry_in ? (const uint8_t*)fresh(ka)->Q.spill + (size_t)(carry_in - 1) * 16 : nullptr);
#ifdef HYPO_PHASE_TIMERS
            const uint64_t ta = (uint64_t)clock64();
#endif
            account(poa, w, rc);
#ifdef HYPO_PHASE_TIMERS
            const uint64_t tb = (uint64_t)clock64();
#endif
            poa.fetch_next();
#ifdef HYPO_PHASE_TIMERS
            dbg[17] += tb - ta;
#endif
        }
    } else {
        uint32_t w;
        while (dequeue(&w)) account(poa, w, poa.run(w, carry_in ? (const uint8_t*)fresh(ka)->Q.spill + (size_t)(carry_in - 1) * 16 : nullptr));
    }
    if (g.lane == 0) {
        HypoPoaStats* st = fresh(ka)->Q.stats;
        if constexpr (USE_LDS) {
            const uint32_t* const stt = poa.stat;
            cells = stt[PoaT::ACC_CELLS]; aligns = stt[PoaT::ACC_ALIGNS]; abytes = stt[PoaT::ACC_ABYTES]; n_reused = stt[PoaT::ACC_REUSED];
            n_thr = stt[PoaT::ACC_THR]; c_scored = stt[PoaT::ACC_CSCORED]; c_thr = stt[PoaT::ACC_CTHR]; n_ok = stt[PoaT::ACC_NOK];
            n_esc = stt[PoaT::ACC_NESC]; n_fail = stt[PoaT::ACC_NFAIL]; n_carried = stt[PoaT::ACC_NCARRIED];
        }
        atomicAdd((unsigned long long*)&st->n_carried, (unsigned long long)n_carried);
        atomicAdd((unsigned long long*)&st->n_class[cls], (unsigned long long)n_ok);
        atomicAdd((unsigned long long*)&st->n_escalated, (unsigned long long)n_esc);
        atomicAdd((unsigned long long*)&st->n_failed, (unsigned long long)n_fail);
        atomicAdd((unsigned long long*)&st->dp_cells, (unsigned long long)cells);
        atomicAdd((unsigned long long*)&st->n_alignments, (unsigned long long)aligns);
        atomicAdd((unsigned long long*)&st->alg_bytes[cls], (unsigned long long)abytes);
        atomicAdd((unsigned long long*)&st->n_reused, (unsigned long long)n_reused);
        atomicAdd((unsigned long long*)&st->n_threaded, (unsigned long long)n_thr);
        atomicAdd((unsigned long long*)&st->cells_scored, (unsigned long long)c_scored);
        atomicAdd((unsigned long long*)&st->cells_threaded, (unsigned long long)c_thr);
#ifdef HYPO_PHASE_TIMERS
        static_assert(kNumPoaClasses * 32 * 8 <= 2048 - 512 && PH_N + 2 + 20 <= 32, "phase block of the header");
        unsigned long long* ph = (unsigned long long*)((char*)fresh(ka)->Q.count + 512) + (size_t)cls * 32;   // header + 512: [class][32]
        for (int i = 0; i < PH_N; ++i) atomicAdd(&ph[i], (unsigned long long)tph[i]);
        atomicAdd(&ph[PH_N], (unsigned long long)((uint64_t)clock64() - tstart));                // wave lifetime
        atomicAdd(&ph[PH_N + 1], 1ull);                                                           // waves
        for (int i = 0; i < 20; ++i) atomicAdd(&ph[PH_N + 2 + i], (unsigned long long)dbg[i]);      // rows, real alignments, reused, toposorts, serial consensus passes, slow rows, exact tries / hits
#endif
        // this group will push nothing more: what it re-queued is in the queues (a polling kernel of a later class counts these).
        // (every re-queue waited for its own device-coherent stores: no agent-scope fence — an L2 write-back per exiting wave — here either)
        HYPO_RELEASE_STORES();
        atomicAdd(fresh(ka)->Q.done + cls, 1u);
    }
    if (!POLL && USE_LDS && wl == 0) atomicAdd((unsigned long long*)(fresh(ka)->Q.work + cls), (unsigned long long)wall_clock64());
}

// ------------------------------------------------------------------------------------------------
// size class 6: one wave per window, the window's state in a slice of PoaAux::giant_arena (poa_giant.hpp)
// ------------------------------------------------------------------------------------------------
struct GiantKArgs { PoaParams P; PoaQueues Q; char* arena; uint64_t slice_bytes; };
__global__ void __launch_bounds__(64) poa_giant_kernel(GiantKArgs /*read through the kernarg segment*/) {
    typedef const GiantKArgs __attribute__((address_space(4)))* KPtr;
    const KPtr ka = (KPtr)__builtin_amdgcn_kernarg_segment_ptr();
    // This launch is the last of a call and every other has finished when it starts: it leaves the call's final counts and wave-times where
    // the host picks them up for the next call's grids and wave shares (page-locked host memory; two copy commands on the stream until round 6)
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        uint32_t* const host = ka->Q.host;
        __hip_atomic_store(host + 8 + threadIdx.x, ka->Q.count[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((unsigned long long*)(host + 24) + threadIdx.x, (unsigned long long)ka->Q.work[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t pending = ka->Q.count[kGiantClass];
    if (pending == 0) return;                                  // (nearly every call)
    const Grp<64> g{(int)(threadIdx.x & 63)};
    const PoaParamRef P{&ka->P};
    uint64_t cells = 0, aligns = 0, abytes = 0;
    uint32_t n_ok = 0, n_fail = 0;
    char* const slice = ka->arena ? ka->arena + (size_t)blockIdx.x * ka->slice_bytes : nullptr;
    for (;;) {
        uint32_t idx = 0;
        if (g.lane == 0) idx = atomicAdd(ka->Q.head + kGiantClass, 1u);
        idx = (uint32_t)g.shfl((int)idx, 0);
        if (idx >= pending) break;
        const uint32_t w = ka->Q.items[idx];
        Giant<Grp<64>> gi(g, P);
        const int rc = slice ? gi.run(w, slice, ka->slice_bytes) : (int)RES_OVERFLOW;
        if (rc == RES_OK) {
            cells += gi.cells; aligns += gi.aligns; ++n_ok;
            const HypoWindow W = P->windows[w];
            const uint32_t narm = W.n_internal + W.n_prefix + W.n_suffix;
            uint64_t a = (uint64_t)(W.draft_len + 1) / 2 + 16 + 8ull * (1 + narm) + P->out_len[w];
            uint32_t part = 0;
            for (uint32_t t = (uint32_t)g.lane; t < narm; t += 64) part += (P->arm_len[W.first_arm + t] + 3) / 4;
            abytes += a + (uint64_t)(uint32_t)g.reduce_add((int)part);
        } else {
            if (g.lane == 0) {
                P->out_len[w] = 0;
                P->out_status[w] = (uint8_t)(rc == RES_UNDEFINED ? HYPO_ST_UNDEFINED : (rc == RES_INVALID ? HYPO_ST_INVALID : HYPO_ST_CAPACITY));
            }
            ++n_fail;
        }
    }
    if (g.lane == 0) {
        HypoPoaStats* st = ka->Q.stats;
        atomicAdd((unsigned long long*)&st->n_class[kGiantClass], (unsigned long long)n_ok);
        atomicAdd((unsigned long long*)&st->n_failed, (unsigned long long)n_fail);
        atomicAdd((unsigned long long*)&st->dp_cells, (unsigned long long)cells);
        atomicAdd((unsigned long long*)&st->n_alignments, (unsigned long long)aligns);
        atomicAdd((unsigned long long*)&st->cells_scored, (unsigned long long)cells);
        atomicAdd((unsigned long long*)&st->alg_bytes[kGiantClass], (unsigned long long)abytes);
    }
}
static int g_giant_arena_mb = 1024;
void poa_set_giant_arena_mb(int mb) { g_giant_arena_mb = mb < 0 ? 0 : mb; }

// ------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------
// what a launch needs beyond the queues: the HBM scratch of the class (state slices of the HBM-scratch classes, direction-code
// slices of a Cfg::DIRG class) and how many resident groups it is provisioned for
struct ClassScratch { char* base; int groups; };
template <class Cfg, bool USE_LDS, bool POLL = false>
static hipError_t launch_class(const PoaParams& P, const PoaQueues& Q, int cls, uint32_t n_windows,
                               ClassScratch scr, int num_cus, hipStream_t stream,
                               int waves_per_cu_cap = 0, bool mop_up = false, uint32_t* groups_launched = nullptr, uint32_t expected = 0,
                               int first_slice = 0 /* scratch slices [first_slice, scr.groups) are this launch's: two launches of a class side by side */) {
    auto kern = poa_class_kernel<Cfg, USE_LDS, POLL>;
    constexpr int GPW = 64 / Cfg::GW;
    char* const scratch = scr.base + (size_t)first_slice * (Cfg::DIRG ? PoaLayout<Cfg>::DIRG_BYTES : PoaLayout<Cfg>::BYTES);
    const int group_cap = scr.groups - first_slice;
    const size_t lds = USE_LDS ? (size_t)GPW * PoaLayout<Cfg>::BYTES : (Cfg::HYBRID ? (size_t)GPW * PoaLayout<Cfg>::FAST_BYTES : 0);
    hipError_t e;
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    if (const char* cap = getenv("HYPO_POA_WAVES_PER_CU")) {      // tuning/diagnostic knob
        const int c = atoi(cap);
        if (c >= 1 && c < per_cu) per_cu = c;
    }
    if (waves_per_cu_cap >= 1 && waves_per_cu_cap < per_cu) per_cu = waves_per_cu_cap;
    long grid = (long)per_cu * num_cus;
    if ((!USE_LDS || Cfg::DIRG) && grid * GPW > group_cap) grid = group_cap / GPW;
    const long need = ((long)n_windows + GPW - 1) / GPW;      // never more waves than windows
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    // first pass: slots [0, planned) with cursor head[cls]; mop-up pass: slots [planned, count) with cursor head2[cls]
    uint32_t* head = mop_up ? Q.head2 + cls : Q.head + cls;
    const uint32_t* bound = (mop_up || waves_per_cu_cap == 0) ? Q.count + cls : Q.planned + cls;
    PoaKArgs a;
    a.P = P; a.Q = Q; a.cls = cls; a.scratch = Cfg::DIRG ? nullptr : scratch; a.head = head; a.bound = bound;
    a.dirg = Cfg::DIRG ? scratch : nullptr; a.expected = expected;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, stream, a);
    if (groups_launched) *groups_launched += (uint32_t)(grid * GPW);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// arm offsets on the device: a caller whose arms lie back to back in arm order (each on a byte boundary) passes
// arm_off == NULL and saves the upload of 8 bytes per arm (a third of the C2 batch's input); the offsets are an exclusive
// prefix sum of ceil(len / 4), three small kernels into the tail of the workspace.
// ------------------------------------------------------------------------------------------------
constexpr int AO_THREADS = 256, AO_ITEMS = 1024;           // arms per workgroup
__global__ void __launch_bounds__(AO_THREADS)
arm_off_partial(const uint32_t* __restrict__ arm_len, uint64_t n_arms, uint64_t* __restrict__ bsum) {
    __shared__ uint32_t red[AO_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * AO_ITEMS;
    uint32_t s = 0;
    for (int i = threadIdx.x; i < AO_ITEMS; i += AO_THREADS) { const uint64_t a = b0 + i; if (a < n_arms) s += (arm_len[a] + 3) >> 2; }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t t = 0; for (int i = 0; i < AO_THREADS / 64; ++i) t += red[i]; bsum[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(1024) arm_off_blocksums(uint64_t* __restrict__ bsum, uint64_t n_blocks) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t c0 = 0; c0 < n_blocks; c0 += 1024) {
        const uint64_t i = c0 + threadIdx.x;
        const uint64_t v = i < n_blocks ? bsum[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint64_t add = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        const uint64_t incl = sh[threadIdx.x];
        if (i < n_blocks) bsum[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(AO_THREADS)
arm_off_final(const uint32_t* __restrict__ arm_len, uint64_t n_arms, const uint64_t* __restrict__ bsum, uint64_t* __restrict__ arm_off) {
    __shared__ uint32_t wsum[AO_THREADS / 64];
    const uint64_t a0 = (uint64_t)blockIdx.x * AO_ITEMS + (uint64_t)threadIdx.x * 4;     // each lane owns 4 consecutive arms
    uint32_t c[4], mine = 0;
    for (int i = 0; i < 4; ++i) { c[i] = a0 + i < n_arms ? (arm_len[a0 + i] + 3) >> 2 : 0; mine += c[i]; }
    uint32_t inc = mine;
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) woff += wsum[i];
    uint64_t run = bsum[blockIdx.x] + woff + (inc - mine);
    for (int i = 0; i < 4; ++i) { if (a0 + i < n_arms) arm_off[a0 + i] = run; run += c[i]; }
}
static size_t arm_off_region_bytes(uint64_t n_arms) {        // offsets + block sums
    if (!n_arms) return 0;
    const uint64_t nb = (n_arms + AO_ITEMS - 1) / AO_ITEMS;
    return ((size_t)n_arms * 8 + 255) / 256 * 256 + ((size_t)nb * 8 + 255) / 256 * 256;
}

// spill pool of the re-queued windows' graphs (Poa::spill: 1.5 KB for a class-0 window, 12 KB for a full class-3 one): a bump
// allocator, a window that finds it full starts again from its first sequence as before
static size_t poa_spill_bytes(uint32_t n_windows) {
    size_t b = (size_t)n_windows * 1024;         // (256 until round 6: a batch in which a fifth of the windows outgrow their class ran out, and what found no room started over)
    const size_t lo = (size_t)1 << 20, hi = (size_t)1 << 30;
    b = b < lo ? lo : (b > hi ? hi : b);
    return b;
}
static int groups3_for(uint32_t n_windows) { return n_windows < (uint32_t)kMaxGroups3 ? (n_windows < 16u ? 16 : (int)n_windows) : kMaxGroups3; }
static size_t poa_workspace_prefix(uint32_t n_windows) {       // header, class queues, plan keys, carry words, spill pool, class 3's direction codes
    size_t b = kPoaHeaderBytes;
    b += (size_t)kNumPoaClasses * n_windows * sizeof(uint32_t);
    b = (b + 255) / 256 * 256;
    b += ((size_t)n_windows * 2 + 255) / 256 * 256;
    b += ((size_t)n_windows * 4 + 255) / 256 * 256;
    b += poa_spill_bytes(n_windows);
    b += (size_t)groups3_for(n_windows) * PoaLayout<PoaClass3>::DIRG_BYTES;
    return b;
}

// ------------------------------------------------------------------------------------------------
// wave shares of the three concurrent LDS-class kernels from the work of the last call
// ------------------------------------------------------------------------------------------------
// What a wave of a class takes of a CU: LDS bytes (granules of 512) and vector registers (granules of 8, 2 048 per CU).  The
// shares {w0, w1, w2} minimise max_c work[c] / w[c] under both budgets; the LDS budget is a CU's 160 KB plus the 5 % by which the
// measured-best fixed shares {5,5,6} overbook it (167 KB: the kernel submitted last grows into what the first one to run dry
// leaves), less what the polling class-3 waves hold.  Among shares within 3 % of the best the one with most waves wins (latency).
struct WaveFootprint { size_t lds; int vgprs; int max_waves; };
template <class Cfg> static WaveFootprint footprint_of() {
    auto kern = poa_class_kernel<Cfg, true, false>;
    constexpr int GPW = 64 / Cfg::GW;
    WaveFootprint f;
    f.lds = ((size_t)GPW * PoaLayout<Cfg>::BYTES + 511) / 512 * 512;
    hipFuncAttributes at;
    f.vgprs = 128;
    if (hipFuncGetAttributes(&at, (const void*)kern) == hipSuccess && at.numRegs >= 8 && at.numRegs <= 512) f.vgprs = (at.numRegs + 7) / 8 * 8;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, (size_t)GPW * PoaLayout<Cfg>::BYTES) != hipSuccess || per_cu < 1) per_cu = 4;
    f.max_waves = per_cu;
    // (the occupancy query knows the register file: a footprint it contradicts is not trusted)
    if (f.max_waves * f.vgprs > 2048) f.vgprs = 2048 / f.max_waves / 8 * 8;
    return f;
}
static void pick_wave_shares(const uint64_t work[3], bool four_groups, int poll_waves_per_cu, int caps[]) {
    static const WaveFootprint fp0 = footprint_of<PoaClass0>(), fp0w = footprint_of<PoaClass0W>(), fp1 = footprint_of<PoaClass1>(),
                               fp2 = footprint_of<PoaClass2>(), fp3 = footprint_of<PoaClass3>();
    const WaveFootprint fp[3] = {four_groups ? fp0 : fp0w, fp1, fp2};
    const double total = (double)work[0] + (double)work[1] + (double)work[2];
    // nothing measured (first call, or a torn / implausible read: a wave lives < 10 s = 1e9 ticks, a launch has < 1e5 waves)
    if (total <= 0.0) return;
    for (int c = 0; c < 3; ++c) if (work[c] > (uint64_t)1e14) return;
    double lds_budget = 160.0 * 1024.0 * 1.05 - (double)poll_waves_per_cu * (double)fp3.lds;
    double vgpr_budget = 2048.0 - (double)poll_waves_per_cu * (double)fp3.vgprs * 0.5;      // (a polling wave sleeps most of the time, but it holds its registers)
    double best_t = 1e300; int best[3] = {caps[0], caps[1], caps[2]}, best_sum = 0;
    for (int pass = 0; pass < 2; ++pass) {                 // pass 0: the best time; pass 1: most waves within 3 % of it
        for (int w0 = 1; w0 <= fp[0].max_waves; ++w0)
            for (int w1 = 1; w1 <= fp[1].max_waves; ++w1)
                for (int w2 = 1; w2 <= fp[2].max_waves; ++w2) {
                    const double lds = (double)w0 * fp[0].lds + (double)w1 * fp[1].lds + (double)w2 * fp[2].lds;
                    const double vg = (double)w0 * fp[0].vgprs + (double)w1 * fp[1].vgprs + (double)w2 * fp[2].vgprs;
                    if (lds > lds_budget || vg > vgpr_budget) continue;
                    double t = (double)work[0] / w0;
                    if ((double)work[1] / w1 > t) t = (double)work[1] / w1;
                    if ((double)work[2] / w2 > t) t = (double)work[2] / w2;
                    if (pass == 0) { if (t < best_t) best_t = t; }
                    else if (t <= best_t * 1.03 && w0 + w1 + w2 > best_sum) { best_sum = w0 + w1 + w2; best[0] = w0; best[1] = w1; best[2] = w2; }
                }
        if (best_t >= 1e300) return;                       // nothing fits (cannot happen: {1,1,1} does)
    }
    caps[0] = best[0]; caps[1] = best[1]; caps[2] = best[2];
    if (getenv("HYPO_POA_ADAPT_LOG")) fprintf(stderr, "[hypo_gpu] wave shares {%d,%d,%d} from wave-time {%.2f, %.2f, %.2f} ms (x 1 wave), poll %d\n",
                                               caps[0], caps[1], caps[2], work[0] * 1e-5, work[1] * 1e-5, work[2] * 1e-5, poll_waves_per_cu);
}

size_t poa_workspace_bytes(uint32_t n_windows, int long_groups, uint64_t computed_arm_offsets) {
    size_t big = 0;                                         // the HBM-scratch classes run one after the other and share the region
    int g4 = long_groups > 0 ? long_groups : max_global_groups(4, n_windows);
    g4 = g4 < kMinGlobalGroups ? kMinGlobalGroups : (g4 > kMaxGlobalGroups4 ? kMaxGlobalGroups4 : g4);
    int g5 = long_groups == kMinGlobalGroups ? kMinGlobalGroups5 : max_global_groups(5, n_windows);   // the minimum applies to both classes
    g5 = g5 < kMinGlobalGroups5 ? kMinGlobalGroups5 : g5;
    const size_t x4 = (size_t)g4 * PoaLayout<PoaClass4>::BYTES, x5 = (size_t)g5 * PoaLayout<PoaClass5>::BYTES;
    big = x4 > x5 ? x4 : x5;
    return poa_workspace_prefix(n_windows) + big + arm_off_region_bytes(computed_arm_offsets);
}

hipError_t poa_run(const PoaParams& P_in, uint32_t n_windows, void* workspace, size_t workspace_bytes,
                   int num_cus, hipStream_t stream, KernelEvents* prof, PoaAux* A) {
    if (n_windows == 0) return hipSuccess;
    if (!A) return hipErrorInvalidValue;
    PoaParams P = P_in;
    const size_t ao_bytes = (!P.arm_off && P.n_arms) ? arm_off_region_bytes(P.n_arms) : 0;     // arm offsets to be computed here
    if (workspace_bytes < poa_workspace_bytes(n_windows, kMinGlobalGroups, ao_bytes ? P.n_arms : 0)) return hipErrorInvalidValue;
    if (ao_bytes) {                                           // tail of the workspace
        uint64_t* ao = (uint64_t*)((char*)workspace + workspace_bytes - ao_bytes);
        uint64_t* bsum = (uint64_t*)((char*)ao + ((size_t)P.n_arms * 8 + 255) / 256 * 256);
        const uint64_t nb = (P.n_arms + AO_ITEMS - 1) / AO_ITEMS;
        hipLaunchKernelGGL(arm_off_partial, dim3((unsigned)nb), dim3(AO_THREADS), 0, stream, P.arm_len, P.n_arms, bsum);
        hipLaunchKernelGGL(arm_off_blocksums, dim3(1), dim3(1024), 0, stream, bsum, nb);
        hipLaunchKernelGGL(arm_off_final, dim3((unsigned)nb), dim3(AO_THREADS), 0, stream, P.arm_len, P.n_arms, bsum, ao);
        P.arm_off = ao;
    }
    // resident groups of the HBM-scratch classes = what the provided scratch holds
    const size_t scratch_bytes = workspace_bytes - ao_bytes - poa_workspace_prefix(n_windows);
    const size_t fit4 = scratch_bytes / PoaLayout<PoaClass4>::BYTES, fit5 = scratch_bytes / PoaLayout<PoaClass5>::BYTES;
    const int groups4 = (int)(fit4 < (size_t)max_global_groups(4, n_windows) ? fit4 : (size_t)max_global_groups(4, n_windows));
    const int groups5 = (int)(fit5 < (size_t)max_global_groups(5, n_windows) ? fit5 : (size_t)max_global_groups(5, n_windows));
    char* ws = (char*)workspace;
    PoaQueues Q;
    Q.count = (uint32_t*)ws;
    Q.head = (uint32_t*)(ws + 64);
    Q.planned = (uint32_t*)(ws + 7680);
    Q.head2 = (uint32_t*)(ws + 7744);
    Q.stats = (HypoPoaStats*)(ws + 128);
    Q.hist = (uint32_t*)(ws + 2048);
    Q.start = (uint32_t*)(ws + 4096);
    Q.cursor = (uint32_t*)(ws + 6144);
    Q.items = (uint32_t*)(ws + kPoaHeaderBytes);
    Q.stride = n_windows;
    size_t off = kPoaHeaderBytes + (size_t)kNumPoaClasses * n_windows * sizeof(uint32_t);
    off = (off + 255) / 256 * 256;
    Q.keys = (uint16_t*)(ws + off);
    off += ((size_t)n_windows * 2 + 255) / 256 * 256;
    Q.carry = (uint32_t*)(ws + off);
    off += ((size_t)n_windows * 4 + 255) / 256 * 256;
    Q.spill = ws + off;
    Q.spill_cap16 = (uint32_t)(poa_spill_bytes(n_windows) / 16);
    off += poa_spill_bytes(n_windows);
    Q.spill_used = (unsigned long long*)(ws + 7872);
    Q.done = (uint32_t*)(ws + 7808);
    Q.work = (uint64_t*)(ws + 7936);
    const ClassScratch scr3{ws + off, groups3_for(n_windows)};
    off += (size_t)scr3.groups * PoaLayout<PoaClass3>::DIRG_BYTES;
    char* scratch = ws + off;
    const ClassScratch scr4{scratch, groups4}, scr5{scratch, groups5}, scr_lds{nullptr, 0};
    hipError_t e;
    if (!A->planned_host) {
        if ((e = hipHostMalloc((void**)&A->planned_host, 40 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) return e;
        memset(A->planned_host, 0, 40 * sizeof(uint32_t));
        if ((e = hipEventCreateWithFlags(&A->planned_ev, hipEventDisableTiming)) != hipSuccess) return e;
    }
    Q.host = A->planned_host;
    if ((e = hipMemsetAsync(ws, 0, kPoaHeaderBytes, stream)) != hipSuccess) return e;
    // (statuses, lengths and class 3's queue slots are reset by poa_plan_count_kernel)
    int pe = 0;
    if (prof) (void)hipEventRecord(prof->ev[0], stream);
    hipLaunchKernelGGL(poa_plan_count_kernel, dim3((n_windows + PLAN_WPB - 1) / PLAN_WPB), dim3(PLAN_THREADS), 0, stream, P, Q, n_windows);
    hipLaunchKernelGGL(poa_plan_scan_kernel, dim3(1), dim3(64), 0, stream, Q);
    hipLaunchKernelGGL(poa_plan_scatter_kernel, dim3((n_windows + PLAN_THREADS - 1) / PLAN_THREADS), dim3(PLAN_THREADS), 0, stream, Q, n_windows);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (prof) (void)hipEventRecord(prof->ev[1], stream);
    // The plan's class counts decide class 0's geometry and wave share and size the grids
    // of the rare classes (3: oversized, 4: LONG, 5: catch-all): a grid of 2 048 single-wave workgroups costs ~0.1 ms of
    // dispatch even if every wave leaves at once, and these classes are empty in most batches (a few escalated windows still
    // find a small grid waiting).
    // The FIRST call of a context waits for its own plan (nothing to go by yet).  Every later call is queued without a host wait:
    // it sizes its grids and picks class 0's geometry from the plan of the previous call, scaled to this batch's size (the
    // batches of one run look alike; a grid that turns out small only lowers the parallelism of that class, the persistent
    // waves still drain the queue), and leaves its own counts behind for the next one.  HYPO_POA_SYNC_PLAN=1 waits every time.
    uint32_t* const pinned = A->planned_host;          // [0..7] this call's plan (async copy), [8..15] final counts of the last finished call
    const hipEvent_t planned_ev = A->planned_ev;
    const bool have_history = A->history_valid && A->next_kind == A->last_kind;      // (PoaAux::next_kind)
    const bool wait_for_plan = !have_history || getenv("HYPO_POA_SYNC_PLAN") != nullptr;
    uint32_t hist[24];
    if (!wait_for_plan) {
        // what the previous call left in the pinned buffer (complete unless that call is still running: then the one before it)
        for (int c = 0; c < 8; ++c) {
            const uint64_t prev_planned = pinned[c], prev_count = pinned[8 + c];
            hist[c] = (uint32_t)(prev_planned * n_windows / (A->history_windows ? A->history_windows : 1));
            hist[8 + c] = (uint32_t)(prev_count * n_windows / (A->history_windows ? A->history_windows : 1));
            hist[16 + c] = hist[c];
        }
    }
    (void)hipEventRecord(planned_ev, stream);
    if (wait_for_plan) {
        if ((e = hipEventSynchronize(planned_ev)) != hipSuccess) return e;
        for (int c = 0; c < 8; ++c) { hist[c] = pinned[c]; hist[8 + c] = have_history ? pinned[8 + c] : 0u; hist[16 + c] = have_history ? A->last_planned[c] : 0u; }
    }
    const uint32_t* const planned_host = hist;
    // Windows re-queued into a class are only known on the device.  The grids of the mop-up passes and of the rare classes
    // are sized from what the plan put there plus what the LAST finished call saw arrive later (its counters come back
    // asynchronously at the end of every call): batches of one run look alike, and noisier reads re-queue many windows (at 1 %
    // read error 3 % of the windows outgrow their class).  Without history a floor of a few hundred waves applies.
    const uint32_t* const last_count = planned_host + 8;
    const uint32_t* const last_planned = planned_host + 16;
    auto late_arrivals = [&](int cls) -> uint32_t {
        const uint32_t seen = last_count[cls] > last_planned[cls] ? last_count[cls] - last_planned[cls] : 0u;
        // the mop-up passes of classes 1 and 2 are launched on the side streams, where the dispatch of idle waves (they leave
        // after one look at the queue) costs nothing: they get a generous floor.  Class 3 and later start on the caller's stream.
        // (a rare class the last finished call saw nothing of gets 32 waves instead of 256: a launch whose waves all leave at once costs 5 us
        // with 32 of them and 15-17 with 256, twice per call on the caller's stream = 1.3 % of the C2 step; waves are persistent, a surprise
        // is still drained, and the next call sizes for it)
        const uint32_t floor = cls < 3 ? (n_windows / 8 > 256 ? n_windows / 8 : 256) : ((have_history && last_count[cls] == 0) ? 32u : 256u);
        const uint32_t want = seen + seen / 2 + floor;
        return want < n_windows ? want : n_windows;
    };
    auto rare_grid_hint = [&](int cls) -> uint32_t {         // windows to size the grid of a rare class for
        const uint64_t want = (uint64_t)planned_host[cls] + late_arrivals(cls);
        return (uint32_t)(want < n_windows ? want : n_windows);
    };
    // An idle persistent wave exits after one failed dequeue, so grids may be generous.
    //
    // The three LDS classes run CONCURRENTLY on the caller's stream and two auxiliary streams: the small-window
    // kernels are latency bound (most sequences reuse an alignment) while the large-window kernel saturates VALU
    // issue, so sharing the CUs fills issue slots either would leave idle (measured: ~11 % per step).  Each gets a
    // share of a CU's LDS through a waves-per-CU cap.  A window re-queued by a class that ran next to its successor
    // is picked up by a mop-up launch afterwards; the rare classes follow on the caller's stream.
    hipStream_t* const aux = A->aux;
    hipEvent_t* const join_ev = A->join_ev;
    hipEvent_t& fork_ev = A->fork_ev;
    // waves per CU of the three concurrent kernels (they share the CU's 160 KB of LDS, which is what bounds residency: 8 / 8 /
    // 14.5 KB per wave of classes 0 / 1 / 2 since class 1 runs one window per wave; 7.6 / 15.8 / 14.1 KB when the sweep below was made).  Caps whose footprints add up to about one CU's LDS make the split independent of
    // which kernel the dispatcher happens to serve first — with {5,5,5} (187 KB) the last one to arrive got what was left until
    // another finished, and which one that was depended on the stream -> hardware queue mapping of the process (C2 call 3.5 - 4.2 ms
    // for the same code).  Swept on C2 under two mappings (profiles/diag/caps_fit_sweep.sh, ms per call): {4,4,5} 3.48 / 3.48,
    // {3,4,5} 3.64 / 3.55, {4,3,5} 3.60 / 3.71, {4,4,4} 3.64 / 3.94, {5,5,5} 3.61 / 3.78, {5,4,4} 3.58 / 4.13.  Class 0 is set below.
    // After Poa::fetch_next and with class 1 at one window per wave (8 KB per wave): {5,5,6} (profiles/diag/r03_wave_wide_sweep2.sh:
    // C2 2.56 ms, 0.5 % read error 4.76, 1 % 10.2; {4,5,6} 2.56 / 4.72 / 11.1, {4,6,6} 2.65 / 4.96 / 11.3, {4,4,5} 2.94 / - / 9.8-11).
    // Round 5, after the one-substitution shortcut and Poa::topo_insert took a third off class 2's work: {5,5,5} (154.6 KB: every wave
    // resident, the three kernels' times stop swapping places between runs) — profiles/diag/r05_caps_fit.txt, ms per call at
    // 0.2 / 0.5 / 2 / 3 % read error: {5,5,6} 1.61-1.64 / 2.43-2.44 / 15.2-15.7 / 24.7-24.9, {5,5,5} 1.55 x 3 / 2.33 / 13.7-14.1 /
    // 22.5-22.6, {6,4,5} 1.56-1.59, {5,4,6} 1.55-1.69, {4,5,6} 1.61-1.66; 1 % alone prefers {5,5,6} (5.4-5.8 against 5.8-6.6).
    int caps[kNumPoaClasses] = {5, 5, 5, 0, 0, 0};
    if (const char* cs = getenv("HYPO_POA_CAPS")) sscanf(cs, "%d,%d,%d,%d,%d", &caps[0], &caps[1], &caps[2], &caps[3], &caps[4]);
    // wave-time per class of the last finished call (PoaQueues::work; read like the counts above: whatever call finished last)
    uint64_t last_work[3];
    for (int c = 0; c < 3; ++c) last_work[c] = wait_for_plan && !have_history ? 0ull : ((const volatile uint64_t*)(pinned + 24))[c];
    // One kernel after the other instead: when the last call left more than a tenth of its windows to class 3 (read error of
    // several per cent) every kernel is long and fills the chip alone, and fixed LDS shares only leave the share of whichever
    // kernel ends first idle: 5 % read error 47 -> 40 ms, 3 % 30 -> 29 ms; below that the concurrent schedule wins (2 %: 20 against
    // 21.6 ms, C2: 3.7 against 4.2 ms; profiles/diag/r03_seq.sh, r03_backfill.sh).  HYPO_POA_SEQUENTIAL=0|1 forces.
    const char* seq_env = getenv("HYPO_POA_SEQUENTIAL");
    const bool sequential = seq_env ? atoi(seq_env) > 0 : (have_history && (uint64_t)last_count[3] * kSequentialDivisor > n_windows);
    auto rec = [&](int idx, hipStream_t st) { if (prof) (void)hipEventRecord(prof->ev[idx], st); };
    if (sequential) {
#define HYPO_LAUNCH(ID, CFG)                                                                              \
        rec(2 + 2 * ID, stream);                                                                          \
        if ((e = launch_class<CFG, (ID < kFirstGlobalClass)>(P, Q, ID, ID >= 3 ? rare_grid_hint(ID) : n_windows,                   \
                                                             (ID == 3 ? scr3 : (ID == 4 ? scr4 : (ID == 5 ? scr5 : scr_lds))), num_cus, stream)) != hipSuccess) return e; \
        rec(3 + 2 * ID, stream);
        HYPO_FOR_EACH_CLASS(HYPO_LAUNCH)
#undef HYPO_LAUNCH
    } else {
        if (!aux[0]) {
            for (int i = 0; i < 3; ++i) {          // (aux[3] only when a batch needs it, below: a stream that merely exists changes how the others map to hardware queues)
                if ((e = hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking)) != hipSuccess) return e;
                if ((e = hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming)) != hipSuccess) return e;
            }
            if ((e = hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming)) != hipSuccess) return e;
        }
        // The host looks at the plan before it launches anything: a batch made of tiny windows almost only (dense short reads
        // on a large genome) runs class 0 with twice the waves (the default split starves it: 44 -> 55 M windows/s there).
        const uint64_t lds_windows = (uint64_t)planned_host[0] + planned_host[1] + planned_host[2];
        bool four_groups = (uint64_t)planned_host[0] * 100 > lds_windows * 85;
        if (const char* g0 = getenv("HYPO_POA_CLASS0")) four_groups = atoi(g0) == 16;      // 16 | 32: lanes per group (tests)
        if (!getenv("HYPO_POA_CAPS") && four_groups) { caps[0] = 7; caps[1] = 6; caps[2] = 5; }      // (dense shape: {7,6,5} 76.0 M windows/s, {8,5,4} 74.8, {7,4,5} 73.4)
        // ... and from the second call on, batches of tiny windows get their shares from the WORK of the last finished call
        // (pick_wave_shares: wave-time per class as the kernels measured it, shares that let the three end together within the
        // CU's LDS and registers): the mix of classes differs from genome to genome there (dense short reads: 91 % / 8 % / 1 % of
        // the windows -> {7,5,1}, 69.6 -> 76.9-77.5 M windows/s; HiFi-like 56.3 -> 59.3-59.8 M).  Mixed batches keep the fixed
        // shares: on the C2 batch the model's pick lost (0.2 % read error {4,4,7}: 2.57 -> 3.03 ms — class 2 issues VALU work back
        // to back and gains little from a seventh wave, class 0 ends with single long windows, not with a shortage of waves), and
        // from 1 % read error on the chip is full whatever the shares are: nine fixed splits, each run twice, all landed within
        // 10.0-11.1 ms at 1 % and 17.4-19.9 ms at 2 % while the three kernels' own times swapped places from run to run
        // (profiles/diag/r03_adapt_ab.sh, r03_caps_err_sweep.sh + .txt).  HYPO_POA_ADAPT=1 forces the model everywhere, 0 turns it off.
        const char* adapt_env = getenv("HYPO_POA_ADAPT");
        const bool adapt = adapt_env ? atoi(adapt_env) != 0 : four_groups;
        // (the measurement must come from a call that ran class 0 in the geometry this call picks: the calls of a run are
        // queued without waiting for each other, so the last FINISHED call may be a batch of another kind, and a class-0 wave
        // of four groups takes twice the time of one of two)
        const uint64_t measured_gw = ((const volatile uint64_t*)(pinned + 24))[6];
        const bool same_kind = measured_gw == (four_groups ? (uint64_t)PoaClass0::GW : (uint64_t)PoaClass0W::GW);
        if (!getenv("HYPO_POA_CAPS") && adapt && have_history && same_kind) {
            const uint32_t seen3_now = last_count[3] > planned_host[3] ? last_count[3] : planned_host[3];
            const int poll_waves_per_cu = seen3_now == 0 ? 0 : ((seen3_now + seen3_now / 4) > 512u ? 2 : 1);
            pick_wave_shares(last_work, four_groups, poll_waves_per_cu, caps);
        }
        (void)hipEventRecord(fork_ev, stream);
        (void)hipStreamWaitEvent(aux[0], fork_ev, 0);
        (void)hipStreamWaitEvent(aux[1], fork_ev, 0);
        (void)hipStreamWaitEvent(aux[2], fork_ev, 0);
        uint32_t producers = 0;                              // lane groups of every launch of classes 0 - 2 (what class 3's polling pass waits for)
        // (experiment knob, off: a second class-2 launch behind class 0 / class 1 on their streams, to take over the LDS share those leave
        // when they end early.  profiles/r06_backfill.txt: non-i.i.d. batch 10.4 -> 10.1-10.2 ms, C2 1.33 -> 1.38-1.46 ms, 1-2 % read error
        // within noise — the waves that are resident already are what issue is shared between; more of them add little.)
        int backfill[2] = {0, 0};
        if (const char* bf = getenv("HYPO_POA_BACKFILL")) sscanf(bf, "%d,%d", &backfill[0], &backfill[1]);
        auto first2 = [&]() -> hipError_t {
            rec(2 + 2 * 2, stream);
            hipError_t r = launch_class<PoaClass2, true>(P, Q, 2, n_windows, scr_lds, num_cus, stream, caps[2], false, &producers);
            rec(3 + 2 * 2, stream);
            return r;
        };
        auto first0 = [&]() -> hipError_t {
            rec(2 + 2 * 0, aux[0]);
            hipError_t r = four_groups ? launch_class<PoaClass0, true>(P, Q, 0, n_windows, scr_lds, num_cus, aux[0], caps[0], false, &producers)
                                       : launch_class<PoaClass0W, true>(P, Q, 0, n_windows, scr_lds, num_cus, aux[0], caps[0], false, &producers);
            if (r == hipSuccess && backfill[0] > 0) r = launch_class<PoaClass2, true>(P, Q, 2, n_windows, scr_lds, num_cus, aux[0], backfill[0], false, &producers);
            rec(3 + 2 * 0, aux[0]);
            return r;
        };
        auto first1 = [&]() -> hipError_t {
            rec(2 + 2 * 1, aux[1]);
            hipError_t r = launch_class<PoaClass1, true>(P, Q, 1, n_windows, scr_lds, num_cus, aux[1], caps[1], false, &producers);
            if (r == hipSuccess && backfill[1] > 0) r = launch_class<PoaClass2, true>(P, Q, 2, n_windows, scr_lds, num_cus, aux[1], backfill[1], false, &producers);
            rec(3 + 2 * 1, aux[1]);
            return r;
        };
        // Submission order decides who gets LDS first: class 2 and class 0 start with their full share, class 1 takes what is
        // left and grows when class 0 runs dry.  Measured on C2 (ms per call): 201 3.89, 021 4.28, 120 4.38, 210 4.69, 102 4.72,
        // 012 4.76 (HYPO_POA_ORDER, for experiments).
        const char* order = getenv("HYPO_POA_ORDER");
        if (!order || strlen(order) != 3) order = "201";
        for (int i = 0; i < 3; ++i) {
            e = order[i] == '2' ? first2() : (order[i] == '0' ? first0() : first1());
            if (e != hipSuccess) return e;
        }
        // (Classes 1 and 2 receive no re-queued windows: every SHORT class hands over to class 3, see poa_class_kernel.)
        (void)hipEventRecord(join_ev[0], aux[0]);
        (void)hipEventRecord(join_ev[1], aux[1]);
        (void)hipStreamWaitEvent(stream, join_ev[0], 0);
        (void)hipStreamWaitEvent(stream, join_ev[1], 0);
        // then the rare classes.  LONG windows are planned straight into class 4, so its first pass does not have to wait for
        // anybody: it runs on a third stream next to the short-window kernels (a LONG window occupies one wave for ~0.1 s; its
        // latency is the floor of the whole call), and only the few windows escalated into class 4 later wait for the mop-up
        // pass at the end.
        // (only while every LONG window of the batch gets a wave of its own: the kernel is then bound by the latency of its
        // windows and leaves room.  A batch with more LONG windows keeps the whole chip busy for many window lifetimes, and
        // running it next to the short-window kernels was measured slower than after them — C4 mix, 400 000 SHORT + 8 000 LONG
        // windows: 313 ms against 285 ms — once the two really overlapped, which depends on the process's hardware queues)
        const bool long_first_pass = planned_host[4] > 0 && planned_host[4] <= (uint32_t)groups4;
        if (long_first_pass) {
            rec(2 + 2 * 4, aux[2]);
            if ((e = launch_class<PoaClass4, false>(P, Q, 4, planned_host[4], scr4, num_cus, aux[2], 8)) != hipSuccess) return e;
            rec(3 + 2 * 4, aux[2]);
            (void)hipEventRecord(join_ev[2], aux[2]);
        }
        // Class 3 (what outgrows class 2, and the wide SHORT windows) runs NEXT to the classes that feed it: a polling launch on a
        // stream of its own takes the windows the plan put there and every re-queued window as it arrives (with its graph, Poa::
        // spill), instead of starting when classes 0 - 2 are done — at 0.5 % read error ONE window re-queued into class 3 used to
        // run 2.7 ms on its own behind 5.2 ms of first passes.  It is submitted after every launch it waits for (see
        // poa_class_kernel), sized by what the plan and the last call's late arrivals say; HYPO_POA_POLL=0 turns it off.
        // The regular launch behind the join takes what is left (nothing, unless the polling pass was cut short).
        // A polling wave holds its 16 KB of LDS while it waits, and the launches it waits for must always find room: at most two
        // per CU (one unless the last call saw many windows in the class), and no polling launch at all when the history says the class
        // stays empty (HYPO_POA_POLL_WAVES overrides the count).
        const char* poll_env = getenv("HYPO_POA_POLL");
        const bool poll3 = !(poll_env && atoi(poll_env) == 0);
        const uint32_t seen3 = last_count[3];                     // windows class 3 ended up with in the last finished call, scaled
        // Waves of the polling launch: what the plan put into class 3 (known work: one wave per window and a quarter more) + what the last
        // call saw ARRIVE later.  A polling wave holds 16 KB of LDS while it waits, next to first-pass kernels whose shares fill the CU:
        // a wave per expected arrival (rounds 3-5) cost the 1 % read-error point 3 of its 7 ms — 640 waves, two per CU, idling through a
        // call whose 514 late windows are 0.18 s of wave time — so up to 1 024 arrivals get an eighth of that (they are served as they come,
        // a few per wave, and the regular launch behind the join takes what is left), up to 2 048 a quarter (1.25 % / 1.5 %: 7.2 / 7.7 -> 5.7 / 6.7 ms);
        // beyond that the class has real work and keeps a wave per window (2 % read error: 8.2 ms against 10.1-10.5 with 64-256 waves).  profiles/r06_poll_waves.txt; HYPO_POA_POLL_RULE=0: the old sizing.
        const uint32_t planned3 = planned_host[3], arrivals = seen3 > planned3 ? seen3 - planned3 : 0u;
        uint32_t poll_waves;
        const char* rule_env = getenv("HYPO_POA_POLL_RULE");
        if (rule_env && atoi(rule_env) == 0) { poll_waves = (seen3 > planned3 ? seen3 : planned3); poll_waves += poll_waves / 4; }
        else {
            poll_waves = planned3 + planned3 / 4;
            // ... and none at all once more than about 3 % of the batch arrives late: the first-pass kernels then keep the chip busy for as long
            // as the late windows take anyway, and class 3 runs behind the join with every CU to itself (2.5 / 3 / 4 % read error: 11.9 / 15.4 /
            // 20.9 -> 10.8 / 12.9 / 17.0 ms; at 2 % read error, 3.0 % late, polling still wins: 8.4 against 9.0)
            const bool many_late = (uint64_t)arrivals * 32u > (uint64_t)n_windows;
            if (many_late) poll_waves = 0;                       // (what the plan put into class 3 waits for the join with the rest)
            if (arrivals && !many_late) poll_waves += arrivals <= 1024u ? (arrivals / 8 > 8u ? arrivals / 8 : 8u) : (arrivals <= 2048u ? arrivals / 4 : arrivals + arrivals / 4);
        }                                                         // (none: no polling launch; the regular one below still takes what turns up)
        int poll_cap = poll_waves > 512u ? 2 : 1;
        if (getenv("HYPO_POA_CAPS") && caps[3] > 0) poll_cap = caps[3];
        if (const char* pw = getenv("HYPO_POA_POLL_WAVES")) poll_waves = (uint32_t)atoi(pw);
        // its stream: the third side stream, unless the LONG first pass is on it (then a fourth one, created on first use)
        hipStream_t poll_stream = aux[2];
        if (long_first_pass) {
            if (!aux[3]) {
                if ((e = hipStreamCreateWithFlags(&aux[3], hipStreamNonBlocking)) != hipSuccess) return e;
                if ((e = hipEventCreateWithFlags(&join_ev[3], hipEventDisableTiming)) != hipSuccess) return e;
            }
            (void)hipStreamWaitEvent(aux[3], fork_ev, 0);
            poll_stream = aux[3];
        }
        rec(2 + 2 * 3, poll_stream);
        uint32_t poll_groups = 0;
        if (poll3 && poll_waves > 0) {
            if ((e = launch_class<PoaClass3, true, true>(P, Q, 3, poll_waves, scr3, num_cus, poll_stream, poll_cap, false, &poll_groups, producers)) != hipSuccess) return e;
        }
        rec(3 + 2 * 3, poll_stream);
        hipEvent_t& poll_join = long_first_pass ? join_ev[3] : join_ev[2];
        (void)hipEventRecord(poll_join, poll_stream);
        // The regular launch starts when classes 0 - 2 are done (the queue is final then) and does NOT wait for the polling launch's last
        // window: the two drain the queue side by side (one cursor; the polling waves claim with a compare-and-swap below the count, the
        // regular ones with a fetch-add that may run past it), each with direction-code slices of its own.  Behind the polling launch it
        // started up to a millisecond late at 1 % read error (3.5-4.0 ms in most calls, 4.8-5.3 in a third of them).  When the scratch has
        // no room for a second set of slices (tiny batches) it waits as before.
        const bool side_by_side = poll_groups > 0 && scr3.groups - (int)poll_groups >= 64 && !getenv("HYPO_POA_POLL_JOIN_FIRST");
        if (!side_by_side) (void)hipStreamWaitEvent(stream, poll_join, 0);
        if ((e = launch_class<PoaClass3, true>(P, Q, 3, rare_grid_hint(3), scr3, num_cus, stream, 0, false, nullptr, 0, side_by_side ? (int)poll_groups : 0)) != hipSuccess) return e;
        if (side_by_side) (void)hipStreamWaitEvent(stream, poll_join, 0);
        if (long_first_pass) {
            (void)hipStreamWaitEvent(stream, join_ev[2], 0);
            if ((e = launch_class<PoaClass4, false>(P, Q, 4, late_arrivals(4), scr4, num_cus, stream, 8, true)) != hipSuccess) return e;
        } else {
            rec(2 + 2 * 4, stream);
            if ((e = launch_class<PoaClass4, false>(P, Q, 4, rare_grid_hint(4), scr4, num_cus, stream)) != hipSuccess) return e;
            rec(3 + 2 * 4, stream);
        }
        rec(2 + 2 * 5, stream);
        if ((e = launch_class<PoaClass5, false>(P, Q, 5, rare_grid_hint(5), scr5, num_cus, stream)) != hipSuccess) return e;
        rec(3 + 2 * 5, stream);
    }
    // size class 6 behind everything else: what class 5 could not hold (its launch is empty-handed in nearly every call and leaves at once)
    {
        if (!A->giant_arena && g_giant_arena_mb > 0) {
            const size_t want = (size_t)g_giant_arena_mb << 20;
            if (hipMalloc((void**)&A->giant_arena, want) == hipSuccess) A->giant_bytes = want; else { A->giant_arena = nullptr; A->giant_bytes = 0; (void)hipGetLastError(); }
        }
        GiantKArgs ga{P, Q, A->giant_arena, (uint64_t)(A->giant_bytes / kGiantWaves) / 256 * 256};
        hipLaunchKernelGGL(poa_giant_kernel, dim3(kGiantWaves), dim3(64), 0, stream, ga);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    // this call's final and planned counts for the next call's grid sizes (no wait: whoever reads them gets the last finished call)
    // (poa_giant_kernel has written this call's final counts and wave-times to A->planned_host)
    for (int c = 0; c < 8; ++c) A->last_planned[c] = hist[c];
    A->history_valid = true;
    A->history_windows = n_windows;
    A->last_kind = A->next_kind;
    rec(2 + 2 * kNumPoaClasses, stream);
    pe = 3 + 2 * kNumPoaClasses;
    if (prof) prof->n = pe;
    return hipSuccess;
}

void poa_release(PoaAux* a) {
    if (!a) return;
    for (int i = 0; i < 4; ++i) {
        if (a->aux[i]) (void)hipStreamDestroy(a->aux[i]);
        if (a->join_ev[i]) (void)hipEventDestroy(a->join_ev[i]);
    }
    if (a->fork_ev) (void)hipEventDestroy(a->fork_ev);
    if (a->planned_ev) (void)hipEventDestroy(a->planned_ev);
    if (a->planned_host) (void)hipHostFree(a->planned_host);
    if (a->giant_arena) (void)hipFree(a->giant_arena);
    *a = PoaAux();
}

}  // namespace hypo
