// poa_kernel.hpp — host-visible interface of poa_kernel.hip (internal to libhypo_gpu.so).
#pragma once
#include <hip/hip_runtime.h>
#include "poa_core.hpp"

namespace hypo {

constexpr int kFirstGlobalClass = 4;     // classes >= this keep their state in HBM scratch, not LDS
constexpr int kFirstLongClass = 4;       // LONG windows (<= 500 bp, ~1.3 k nodes) start here
constexpr int kRequeueClass = 3;         // where a SHORT window goes that outgrows classes 0 - 2 (the class that polls its queue)
// resident groups of the HBM-scratch classes (one wavefront each; the scratch is provisioned for this many).  The LONG class
// is register-bound at 2 waves per SIMD: 8 per CU x 256 CUs; the last class is a rare safety net.
#define HYPO_C4_GROUPS 2048
constexpr int kMaxGlobalGroups4 = HYPO_C4_GROUPS;
constexpr int kMaxGlobalGroups5 = 32;
inline int max_global_groups(int cls, uint32_t n_windows) {
    const int cap = cls == 4 ? kMaxGlobalGroups4 : kMaxGlobalGroups5;
    const uint32_t lo = cls == 4 ? 16u : 4u;
    const int want = n_windows < lo ? (int)lo : (n_windows > (uint32_t)cap ? cap : (int)n_windows);   // no batch needs more groups than windows
    return want < cap ? want : cap;
}
constexpr int kMinGlobalGroups = 16;     // the smallest scratch poa_run accepts holds this many groups of the LONG class ...
constexpr int kMinGlobalGroups5 = 4;     // ... and this many of the last one (61 MB each)
constexpr size_t kPoaHeaderBytes = 8192; // count[8] | head[8] @64 | HypoPoaStats @128 | phase cycles @512 | hist @2048 | start @4096 | cursor @6144 | planned @7680 | head2 @7744 | done[8] @7808 | spill_used @7872 | work[8] (u64) @7936
// resident groups of class 3 (direction codes in HBM scratch, PoaLayout::DIRG_BYTES each): what 256 CUs hold at 10 waves per CU
constexpr int kMaxGroups3 = 2560;
constexpr uint32_t kSequentialDivisor = 10;           // class kernels one after the other when the last call left more than 1/10 of its windows to class 3
constexpr uint32_t kQueueUnpublished = 0xffffffffu;   // queue slot of a polled class that no producer has filled yet

constexpr int kPlanBuckets = 64;         // cost buckets per class of the plan's counting sort

struct PoaQueues {
    uint32_t* count;        // [classes] windows queued per class
    uint32_t* head;         // [classes] next queue slot to hand out
    uint32_t* planned;      // [classes] count right after the plan (bound of the concurrent first pass)
    uint32_t* head2;        // [classes] cursor of the mop-up pass over slots [planned, count)
    HypoPoaStats* stats;
    uint32_t* hist;         // [classes * kPlanBuckets] plan: windows per (class, cost bucket)
    uint32_t* start;        // exclusive prefix of hist inside each class
    uint32_t* cursor;       // scatter cursors
    uint16_t* keys;         // [n_windows] class * kPlanBuckets + bucket
    uint32_t* items;        // [classes][stride] window indices, each class ordered by decreasing cost
    uint32_t stride;
    // re-queued SHORT windows take their graph along (Poa::spill): carry[w] = 1 + offset of the window's spill in `spill`, in
    // 16-byte units (0 = none: the window starts from its first sequence); spill_used = bump cursor in the same units
    uint32_t* carry;        // [n_windows]
    char* spill;
    uint32_t spill_cap16;   // pool size in 16-byte units
    unsigned long long* spill_used;   // 64-bit: claimed with one fetch-add (poa_class_kernel), never wraps
    uint32_t* done;         // [classes] lane groups of class c's kernels that have exited (what a polling kernel waits for)
    uint64_t* work;         // [classes] summed lifetimes of the waves of class c's (non-polling) launches, 100 MHz ticks
    uint32_t* host;         // PoaAux::planned_host (page-locked host memory the kernels write their counts to: no copy commands on the stream)
};

// optional event recorder: ev[0]/ev[1] around the plan kernels, ev[2+2c]/ev[3+2c] around size-class kernel c
// (recorded on the stream that kernel runs on), ev[2+2*classes] after everything has joined the caller's stream
struct KernelEvents { hipEvent_t ev[16]; int n; };

// Streams, events and the pinned readback buffer poa_run() needs beyond the caller's stream.  Owned by the library context
// (capi.hip): created on first use on the current device, released by poa_release() at shutdown / device change.
struct PoaAux {
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[4] = {nullptr, nullptr, nullptr, nullptr}, planned_ev = nullptr;
    uint32_t* planned_host = nullptr;     // pinned: [0..7] planned counts of the call in flight, [8..15] final counts of the last finished call,
                                          // [24..39] = 8 x u64 wave-time per class of the last finished call (PoaQueues::work)
    uint32_t last_planned[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // planned counts the previous call worked with
    bool history_valid = false;           // a call has been queued on this context before
    uint32_t history_windows = 0;         // its batch size
    // What the caller knows about the batch it hands over: 0 = windows of short reads, 1 = the LONG windows of a -B run.  The history
    // above describes the batch before this one; when the kind changes it says nothing about this one (a LONG batch sized from
    // the plan of the SHORT batch before it got 256 of its 2 048 waves: 1.84 s instead of 0.35 s on the 250 Mbp set of round 4),
    // and the call waits for its own plan as a first call does.
    int next_kind = 0, last_kind = 0;
    // size class 6 (poa_giant.hpp): HBM for the windows nothing else holds, kGiantWaves slices; allocated by the first call
    char* giant_arena = nullptr;
    size_t giant_bytes = 0;
};
constexpr int kGiantWaves = 2;                 // resident windows of class 6 (one wave and one slice of the arena each)
// megabytes of PoaAux::giant_arena of contexts created from now on (default 1024; 0: no class 6, such windows answer HYPO_ST_CAPACITY)
void poa_set_giant_arena_mb(int mb);
void poa_release(PoaAux* a);

// Bytes for a batch of n_windows windows.  long_groups = resident groups of the LONG class the scratch is provisioned for
// (0 = as many as the batch could use, up to kMaxGlobalGroups4); poa_run derives the group counts from the size it is given.
// computed_arm_offsets = n_arms of a batch that comes without arm_off (the offsets are then computed into the workspace), else 0
size_t poa_workspace_bytes(uint32_t n_windows, int long_groups = 0, uint64_t computed_arm_offsets = 0);
hipError_t poa_run(const PoaParams& P, uint32_t n_windows, void* workspace, size_t workspace_bytes,
                   int num_cus, hipStream_t stream, KernelEvents* prof, PoaAux* aux_state);

}  // namespace hypo
