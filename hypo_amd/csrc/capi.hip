// capi.hip — the C-ABI of include/hypo_gpu.h on top of the gfx950 kernels.
// No CPU implementation of the hot path exists in this library: without a HIP device every entry
// point fails with HYPO_E_NODEVICE / HYPO_E_NOTINIT.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/hypo_gpu.h"
#include "poa_kernel.hpp"
#include "arms_kernel.hpp"
#include "support_kernel.hpp"
#include "scan_kernel.hpp"

namespace {

thread_local char tl_err[512] = "";
thread_local HypoPoaStats tl_stats;

int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(tl_err, sizeof(tl_err), fmt, ap); va_end(ap);
    return code;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(HYPO_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

// ---- host -> device copies of the host-pointer entry points ---------------------------------------------------------------
// A caller of the C-ABI hands over ordinary (pageable) memory: the reference's objects know nothing of page-locked buffers, and
// locking a few GB for one upload costs more than the upload (MI355X box, profiles/history/r04_pin_bench.txt: hipHostMalloc 0.18 s per GB
// + 0.12 s per GB to free it; a copy out of page-locked memory 57 GB/s, out of pageable memory 15-25 GB/s).  Large copies out of
// pageable memory are therefore staged here: a few threads copy the next 32 MB into one of two page-locked bounce buffers of the
// context while the DMA engine drains the other.  Memory the caller did page-lock (hypo_gpu_host_alloc) is copied from directly.
class CopyPool {                                   // a handful of threads that memcpy slices side by side (one pool per process)
public:
    void copy(char* d, const char* s, size_t n) {
        constexpr size_t kSlice = (size_t)2 << 20;
        if (n <= kSlice) { memcpy(d, s, n); return; }
        std::unique_lock<std::mutex> lk(mu);
        if (th.empty()) for (int i = 0; i < kWorkers; ++i) th.emplace_back([this] { work(); });
        for (size_t at = 0; at < n; at += kSlice) jobs.push_back(Job{d + at, s + at, n - at < kSlice ? n - at : kSlice});
        left += jobs.size() - next;
        cv.notify_all();
        while (next < jobs.size()) {                 // the caller takes slices too
            const Job j = jobs[next++];
            lk.unlock(); memcpy(j.d, j.s, j.n); lk.lock();
            --left;
        }
        done.wait(lk, [this] { return left == 0; });
        jobs.clear(); next = 0;
    }
    ~CopyPool() { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv.notify_all(); for (auto& t : th) t.join(); }
private:
    static constexpr int kWorkers = 7;          // (+ the caller; 16 threads were no faster: 13-21 GB/s next to the parser threads of the next batch)
    struct Job { char* d; const char* s; size_t n; };
    std::vector<std::thread> th; std::vector<Job> jobs; size_t next = 0, left = 0; bool quit = false;
    std::mutex mu; std::condition_variable cv, done;
    void work() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return quit || next < jobs.size(); });
            if (quit) return;
            const Job j = jobs[next++];
            lk.unlock(); memcpy(j.d, j.s, j.n); lk.lock();
            if (--left == 0) done.notify_all();
        }
    }
};
CopyPool g_copy_pool;
std::mutex g_copy_mu;                              // one staged copy at a time uses the pool

struct Bounce {
    static constexpr size_t kChunk = (size_t)32 << 20;
    void* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; int turn = 0;
    hipError_t ensure() {
        if (buf[0]) return hipSuccess;
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc(&buf[i], kChunk, hipHostMallocNonCoherent);
            if (e != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)) != hipSuccess) return e;
        }
        return hipSuccess;
    }
    void release() { for (int i = 0; i < 2; ++i) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (buf[i]) (void)hipHostFree(buf[i]); buf[i] = nullptr; ev[i] = nullptr; used[i] = false; } }
};

// Device buffers of the host-pointer entry points: grow-only arenas owned by the context (no allocation in the
// steady-state path); released at shutdown.  Slot numbers are local to each entry point.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t alloc(size_t n) {
        n = n ? n : 16;
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 4;                       // head room: the next batch is rarely exactly this size
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, n); if (e != hipSuccess) return e; cap = n; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// HIP-event recorder for the next calls (hypo_gpu_profile_*)
struct ProfCall { int kind = 0; hypo::KernelEvents ke; };      // kind 1 = POA, 2 = scan
struct Prof { std::vector<ProfCall> calls; int used = 0; };

// One context per device handed to hypo_gpu_init ("one context per device inside one process").  Each owns its
// stream, the POA side streams / events / plan read-back buffer, the grow-only arenas of the host-pointer entry points and
// the uploaded solid-kmer set.  A calling thread works on the context it selected with hypo_gpu_use_device (slot 0 by
// default); entry points lock that context only, so threads driving different devices never wait for each other.
struct Ctx {
    bool ready = false; int device = -1; int num_cus = 0; hipStream_t stream = nullptr;
    // Two batches of the host-pointer POA entry points can be in flight (hypo_gpu_poa_batch_begin / _end): each has its own
    // device buffers, workspace, stream and side streams, so the upload of one overlaps the kernels of the other.  Slot 0's
    // stream is `stream` (also used by the scan and by hypo_gpu_poa_batch_sharded).
    struct Slot {
        DevBuf arena[10]; hypo::PoaAux aux; hipStream_t stream = nullptr; bool busy = false; uint32_t n = 0; HypoPoaStats* stats_pinned = nullptr;
    } slots[2];
    DevBuf scan_arena[7];
    // resident window batch of hypo_gpu_arms_build (inputs, work arrays, the batch, consensus slots, POA workspace)
    // resident window batches built by the arm kernels: [0] SHORT windows (hypo_gpu_arms_build), [1] LONG windows (hypo_gpu_arms_build_long)
    struct ArmsSet { DevBuf arena[5]; HypoArmsSummary sum{}; bool ready = false; hypo::ArmsOut out{}; };
    ArmsSet arms[2];
    // the short reads of the contig batch in hand (hypo_gpu_reads_upload): the support kernels vote with them, hypo_gpu_arms_build
    // (reads == NULL) cuts them into arms
    struct ResidentReads {
        DevBuf data, work; bool ready = false;
        uint32_t n = 0; uint64_t n_cig = 0, reads2_bytes = 0; uint32_t max_span = 0; uint64_t sum_span = 0;
        uint64_t total_len = 0;                            // the coordinate space the reads were checked against
        std::vector<uint32_t> ctg_min_rb, ctg_max_re;      // per contig index of read_contig: what its reads span (later calls check their tables against it)
        const uint32_t *rb = nullptr, *re = nullptr, *qae = nullptr, *cigar_off = nullptr, *cigar = nullptr, *read_contig = nullptr, *file_rank = nullptr;
        const uint64_t* seq_off = nullptr; const uint8_t* reads2 = nullptr;
    } rr;
    DevBuf solid_set; uint32_t solid_k = 0;            // hypo_gpu_solid_set_upload
    // hypo_gpu_solid_scan_keep: the marked positions (contig-local) and their k-mers stay on the device, one pair of exact-size
    // buffers per handle (the caller's contig number); hypo_gpu_support_kmers_kept votes against them
    struct KeptScan { void* kids = nullptr; uint32_t* spos = nullptr; uint64_t n = 0, n_bases = 0; uint32_t k = 0; bool used = false; };
    std::vector<KeptScan> kept;
    // released scans are parked, not freed: hipFree waits for the device and costs 0.1-0.3 ms a call — a contig batch of the 3 Gbp
    // run releases 50 scans (100 buffers) inside its vote call.  Freed together when 16 GB are parked, and at shutdown.
    std::vector<void*> kept_parked; size_t kept_parked_bytes = 0;
    int poa_flags = 0;                                 // hypo_gpu_set_option
    Bounce bounce;                                     // page-locked staging of large copies out of pageable memory (h2d below)
    std::vector<HypoWindow> sh_win; std::vector<uint64_t> sh_aoff, sh_off;   // rebased descriptors of this device's share (hypo_gpu_poa_batch_sharded)
    Prof prof;
    std::recursive_mutex mu;                           // recursive: the host-buffer variants call the device variants
};
constexpr int kMaxDevices = HYPO_MAX_DEVICES;
Ctx g_ctxs[kMaxDevices];
int g_nctx = 0;
std::mutex g_init_mu;                                  // init / shutdown only
thread_local int tl_slot = 0;
Ctx& cur() { return g_ctxs[(tl_slot >= 0 && tl_slot < kMaxDevices) ? tl_slot : 0]; }
#define g_ctx (cur())
#define g_prof (cur().prof)
#define HYPO_LOCKED() std::lock_guard<std::recursive_mutex> hypo_lock_(cur().mu)
// the device of the context is made current for the calling thread (hypo_gpu_init did that for its own thread only)
#define HYPO_ON_DEVICE() do { if (g_ctx.ready) HIP_TRY(hipSetDevice(g_ctx.device)); } while (0)

// Host memory -> device memory on `st`.  On return the host range may be reused (as with hipMemcpyAsync out of pageable
// memory); when `src` is page-locked the copy is queued as it is and the caller's usual rule applies (leave it alone until the
// stream has been synchronised — every entry point that takes page-locked memory does that before it returns or says so).
hipError_t h2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
    constexpr size_t kStagedFrom = (size_t)8 << 20;
    if (bytes < kStagedFrom) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, src) == hipSuccess && at.type != hipMemoryTypeUnregistered) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    (void)hipGetLastError();                           // (an ordinary pointer is "invalid" to that query)
    Ctx& c = cur();
    hipError_t e = c.bounce.ensure();
    if (e != hipSuccess) { (void)hipGetLastError(); return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); }
    std::lock_guard<std::mutex> lk(g_copy_mu);
    Bounce& b = c.bounce;
    for (size_t at0 = 0; at0 < bytes; at0 += Bounce::kChunk) {
        const size_t n = bytes - at0 < Bounce::kChunk ? bytes - at0 : Bounce::kChunk;
        const int i = b.turn; b.turn ^= 1;
        if (b.used[i] && (e = hipEventSynchronize(b.ev[i])) != hipSuccess) return e;      // the copy that last read this buffer has left it
        g_copy_pool.copy((char*)b.buf[i], (const char*)src + at0, n);
        if ((e = hipMemcpyAsync((char*)dst + at0, b.buf[i], n, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        if ((e = hipEventRecord(b.ev[i], st)) != hipSuccess) return e;
        b.used[i] = true;
    }
    return hipSuccess;
}

// Device memory -> host memory behind what is queued on `st`; the data is in `dst` on return.  The way back of h2d: the DMA engine
// fills one bounce buffer while the copy threads empty the other into the caller's ordinary memory (160 MB of vote counters per
// 50 Mbp batch of the 3 Gbp run come back this way).  Page-locked `dst`: copied into directly (queued; the caller synchronises).
hipError_t d2h(void* dst, const void* src, size_t bytes, hipStream_t st) {
    constexpr size_t kStagedFrom = (size_t)8 << 20;
    if (bytes < kStagedFrom) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dst) == hipSuccess && at.type != hipMemoryTypeUnregistered) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    (void)hipGetLastError();
    Ctx& c = cur();
    hipError_t e = c.bounce.ensure();
    if (e != hipSuccess) { (void)hipGetLastError(); return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st); }
    std::lock_guard<std::mutex> lk(g_copy_mu);
    Bounce& b = c.bounce;
    size_t prev_at = 0, prev_n = 0; int prev_i = -1;
    for (size_t at0 = 0; at0 < bytes; at0 += Bounce::kChunk) {
        const size_t n = bytes - at0 < Bounce::kChunk ? bytes - at0 : Bounce::kChunk;
        const int i = b.turn; b.turn ^= 1;
        if (b.used[i] && (e = hipEventSynchronize(b.ev[i])) != hipSuccess) return e;      // (an upload that last read this buffer)
        if ((e = hipMemcpyAsync(b.buf[i], (const char*)src + at0, n, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipEventRecord(b.ev[i], st)) != hipSuccess) return e;
        b.used[i] = true;
        if (prev_i >= 0) {                                                                 // empty the other buffer while this one fills
            if ((e = hipEventSynchronize(b.ev[prev_i])) != hipSuccess) return e;
            g_copy_pool.copy((char*)dst + prev_at, (const char*)b.buf[prev_i], prev_n);
        }
        prev_i = i; prev_at = at0; prev_n = n;
    }
    if (prev_i >= 0) {
        if ((e = hipEventSynchronize(b.ev[prev_i])) != hipSuccess) return e;
        g_copy_pool.copy((char*)dst + prev_at, (const char*)b.buf[prev_i], prev_n);
    }
    return hipSuccess;
}

ProfCall* prof_next(int kind) {
    if (g_prof.used >= (int)g_prof.calls.size()) return nullptr;
    ProfCall* c = &g_prof.calls[g_prof.used++];
    c->kind = kind; c->ke.n = 0;
    return c;
}

struct Rccl {
    void* h = nullptr; bool tried = false; bool ok = false;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* comms[kMaxDevices] = {};
    int n_comms = 0; int devs[kMaxDevices] = {};
    bool load() {
        if (tried) return ok;
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return false;
        CommInitAll = (int (*)(void**, int, const int*))dlsym(h, "ncclCommInitAll");
        CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
        GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
        Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclBroadcast");
        GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && GetErrorString;
        return ok;
    }
    void release() { for (int i = 0; i < n_comms; ++i) if (comms[i] && CommDestroy) (void)CommDestroy(comms[i]); n_comms = 0; }
};
Rccl g_rccl;
std::mutex g_shard_mu;                               // one sharded call at a time (it uses every context)
constexpr int kNcclUint8 = 1;                         // ncclUint8 (rccl.h)


int check_scores(const HypoScoreParams* s) {
    if (!s) return fail(HYPO_E_INVALID, "scores == NULL");
    if (s->sr_gap > 0 || s->lr_gap > 0)
        return fail(HYPO_E_INVALID, "gap penalties must be non-positive (spoa alignment_engine.cpp:43-50)");
    return HYPO_OK;
}

hypo::PoaParams make_params(const HypoScoreParams* s, const HypoWindowBatch* in, HypoConsensusBatch* out) {
    hypo::PoaParams P;
    P.windows = in->windows; P.draft4 = in->draft4; P.arm_off = in->arm_off; P.arm_len = in->arm_len; P.arms2 = in->arms2;
    P.out_bases = out->bases; P.out_off = out->off; P.out_len = out->len; P.out_status = out->status;
    P.sr_m = s->sr_match; P.sr_n = s->sr_mismatch; P.sr_g = s->sr_gap;
    P.lr_m = s->lr_match; P.lr_n = s->lr_mismatch; P.lr_g = s->lr_gap;
    P.n_arms = in->n_arms; P.draft4_bytes = in->draft4_bytes; P.arms2_bytes = in->arms2_bytes;
    P.flags = g_ctx.poa_flags;
    return P;
}

struct Carver {                                       // lays arrays out in one device buffer, 256-byte aligned
    size_t at = 0;
    size_t take(size_t bytes) { const size_t o = at; at += (bytes + 255) / 256 * 256; return o; }
};


// First-use costs of a HIP process — loading this library's code objects onto the device, the occupancy queries, the side
// streams and events of the POA call — come to ≈ 20–40 ms, several steady-state POA calls of the C2 batch, and used to land in
// whichever call came first.  hypo_gpu_init pays them once per context: a two-window batch, a 64-base scan and a four-element
// prefix sum on throw-away buffers.  HYPO_NO_WARMUP=1 leaves them to the first calls (profiles/diag/first_call.py measures
// the difference).
void warm_up(Ctx* c) {
    if (hipSetDevice(c->device) != hipSuccess) return;
    // On the HIP null stream, with side streams of its own that are destroyed again: a process gets four hardware queues by
    // default (GPU_MAX_HW_QUEUES), the POA call uses four streams (the caller's + three side streams), and a warm-up on a fifth
    // stream made two of them share a queue for the rest of the run — measured as a shifted balance of the three concurrent class
    // kernels, C2 step 3.70 -> 4.27 ms.  HYPO_WARMUP_STREAM=0 (diagnostic) runs it on the context's stream as before, 2 on a
    // stream that is created and destroyed here.
    const char* ws_env = getenv("HYPO_WARMUP_STREAM");
    const int ws_mode = ws_env ? atoi(ws_env) : 1;
    hipStream_t st = nullptr, own = nullptr;
    if (ws_mode == 0) st = c->stream;
    if (ws_mode == 2 && hipStreamCreateWithFlags(&own, hipStreamNonBlocking) == hipSuccess) st = own;
    const uint32_t nw = 2, na = 4;
    const size_t wsb = hypo::poa_workspace_bytes(nw, hypo::kMinGlobalGroups, 0), swb = hypo::scan_workspace_bytes(64);
    Carver cv;
    const size_t o_win = cv.take(nw * sizeof(HypoWindow)), o_dr = cv.take(64), o_aoff = cv.take(na * 8), o_alen = cv.take(na * 4), o_arms = cv.take(64),
                 o_bases = cv.take(256), o_off = cv.take((nw + 1) * 8), o_len = cv.take(nw * 4), o_st = cv.take(64), o_p4 = cv.take(64), o_bits = cv.take(64),
                 o_words = cv.take(64), o_kids = cv.take(64 * 8), o_rank = cv.take(64), o_n = cv.take(64), o_sws = cv.take(swb), o_scan = cv.take(64 * 8), o_ws = cv.take(wsb);
    char* d = nullptr;
    if (hipMalloc((void**)&d, cv.at) != hipSuccess) return;
    (void)hipMemsetAsync(d, 0, o_ws, st);                  // all-A drafts and arms, an empty 2-mer set
    HypoWindow hw[nw] = {};
    uint64_t aoff[na] = {0, 8, 16, 24}, off[nw + 1] = {0, 64, 128};
    uint32_t alen[na] = {24, 24, 24, 24};
    for (uint32_t w = 0; w < nw; ++w) { hw[w].type = HYPO_WIN_SHORT; hw[w].draft_len = 24; hw[w].draft_off = 16 * w; hw[w].first_arm = 2 * w; hw[w].n_internal = 2; }
    (void)hipMemcpyAsync(d + o_win, hw, sizeof(hw), hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(d + o_aoff, aoff, sizeof(aoff), hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(d + o_alen, alen, sizeof(alen), hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(d + o_off, off, sizeof(off), hipMemcpyHostToDevice, st);
    (void)hypo::scan_run((const uint8_t*)(d + o_p4), 64, 2, (const uint64_t*)(d + o_bits), (uint64_t*)(d + o_words), (uint64_t*)(d + o_kids), 64, (uint64_t*)(d + o_rank),
                         (uint64_t*)(d + o_n), d + o_sws, swb, st, nullptr);
    hypo::PoaParams P;
    P.windows = (const HypoWindow*)(d + o_win); P.draft4 = (const uint8_t*)(d + o_dr); P.arm_off = (const uint64_t*)(d + o_aoff); P.arm_len = (const uint32_t*)(d + o_alen);
    P.arms2 = (const uint8_t*)(d + o_arms); P.out_bases = d + o_bases; P.out_off = (const uint64_t*)(d + o_off); P.out_len = (uint32_t*)(d + o_len); P.out_status = (uint8_t*)(d + o_st);
    P.sr_m = 5; P.sr_n = -4; P.sr_g = -8; P.lr_m = 3; P.lr_n = -5; P.lr_g = -4;
    P.n_arms = na; P.draft4_bytes = 64; P.arms2_bytes = 64; P.flags = 0;
    hypo::PoaAux tmp;
    (void)hypo::poa_run(P, nw, d + o_ws, wsb, c->num_cus, st, nullptr, &tmp);
    (void)hypo::scan32((const uint32_t*)(d + o_alen), na, (uint64_t*)(d + o_scan), (uint64_t*)(d + o_sws), (uint64_t*)(d + o_n), st);
    (void)hipStreamSynchronize(st);
    hypo::poa_release(&tmp);
    if (own) (void)hipStreamDestroy(own);
    (void)hipFree(d);
    (void)hipGetLastError();
}

// Hardware queues.  The library runs up to seven HIP streams (the caller's, three or four side streams of a POA call, a second
// batch in flight, copies) and a ROCm process gets four hardware queues unless GPU_MAX_HW_QUEUES says otherwise; streams that
// share a queue serialise (two batches in flight: 4.58 instead of 4.15 ms per C2 batch).  The runtime reads the variable when it
// initialises, i.e. at the process's first HIP call — the library asks for eight queues when it is LOADED (linked or dlopen'ed),
// which is earlier unless the host has already used HIP by then; a value the host's environment sets is kept.
__attribute__((constructor)) void ask_for_hardware_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0); }

}  // namespace

extern "C" {

int hypo_gpu_abi_version(void) { return HYPO_GPU_ABI_VERSION; }
const char* hypo_gpu_last_error(void) { return tl_err; }
int hypo_gpu_num_cus(void) { return g_ctx.ready ? g_ctx.num_cus : 0; }
int hypo_gpu_num_devices(void) { return g_nctx; }
#ifndef HYPO_BUILD_ID
#define HYPO_BUILD_ID "unknown"
#endif
const char* hypo_gpu_build_id(void) { return HYPO_BUILD_ID; }

static void release_ctx(Ctx& c) {
    if (c.ready) {
        (void)hipSetDevice(c.device);
        (void)hipDeviceSynchronize();
        for (auto& pc : c.prof.calls) for (auto& e : pc.ke.ev) if (e) (void)hipEventDestroy(e);
        c.prof.calls.clear(); c.prof.used = 0;
        for (auto& sl : c.slots) { hypo::poa_release(&sl.aux); for (auto& a : sl.arena) a.release(); sl.busy = false; if (sl.stats_pinned) (void)hipHostFree(sl.stats_pinned); sl.stats_pinned = nullptr; }
        if (c.slots[1].stream) (void)hipStreamDestroy(c.slots[1].stream);
        c.slots[0].stream = c.slots[1].stream = nullptr;
        for (auto& a : c.scan_arena) a.release();
        for (auto& as : c.arms) { for (auto& a : as.arena) a.release(); as.ready = false; }
        c.rr.data.release(); c.rr.work.release(); c.rr.ready = false;
        c.solid_set.release(); c.solid_k = 0;
        c.bounce.release();
        for (auto& ks : c.kept) { if (ks.kids) (void)hipFree(ks.kids); if (ks.spos) (void)hipFree(ks.spos); }
        c.kept.clear();
        for (void* p : c.kept_parked) (void)hipFree(p);
        c.kept_parked.clear(); c.kept_parked_bytes = 0;
        if (c.stream) (void)hipStreamDestroy(c.stream);
    }
    c.ready = false; c.device = -1; c.num_cus = 0; c.stream = nullptr;
}

int hypo_gpu_init(const int* device_ids, int n_devices) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(HYPO_E_NODEVICE, "no HIP device visible");
    const int zero = 0;
    if (!device_ids || n_devices <= 0) { device_ids = &zero; n_devices = 1; }       // (NULL, 0): device 0
    if (n_devices > kMaxDevices) return fail(HYPO_E_INVALID, "%d devices requested, at most %d", n_devices, kMaxDevices);
    for (int i = 0; i < n_devices; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= n) return fail(HYPO_E_INVALID, "device %d out of range (0..%d)", device_ids[i], n - 1);
        // HYPO_ALLOW_DUP_DEVICES=1 (tests on a one-GPU box): several contexts on one device exercise the sharded path
        for (int j = 0; j < i; ++j) if (device_ids[j] == device_ids[i] && !getenv("HYPO_ALLOW_DUP_DEVICES")) return fail(HYPO_E_INVALID, "device %d listed twice", device_ids[i]);
    }
    for (int i = 0; i < g_nctx; ++i) release_ctx(g_ctxs[i]);   // re-initialisation: streams and events belong to the previous devices
    g_nctx = 0;
    for (int i = 0; i < n_devices; ++i) {
        Ctx& c = g_ctxs[i];
        HIP_TRY(hipSetDevice(device_ids[i]));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_ids[i]));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(HYPO_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device_ids[i], prop.gcnArchName);
        HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        c.slots[0].stream = c.stream;                      // (slot 1's stream is created when a second batch is first put in flight)
        c.device = device_ids[i]; c.num_cus = prop.multiProcessorCount; c.ready = true;
        g_nctx = i + 1;
    }
    if (!getenv("HYPO_NO_WARMUP")) for (int i = 0; i < g_nctx; ++i) warm_up(&g_ctxs[i]);
    HIP_TRY(hipSetDevice(g_ctxs[0].device));
    tl_slot = 0;
    return HYPO_OK;
}

int hypo_gpu_set_option(const char* name, int value) {
    if (!name) return fail(HYPO_E_INVALID, "NULL option name");
    if (!strcmp(name, "native_klov")) {                  // all contexts: the mode belongs to the run, not to a device
        for (int i = 0; i < kMaxDevices; ++i) g_ctxs[i].poa_flags = (g_ctxs[i].poa_flags & ~hypo::POA_NATIVE_KLOV) | (value ? hypo::POA_NATIVE_KLOV : 0);
        return HYPO_OK;
    }
    if (!strcmp(name, "poa_min_class")) {                // 0 .. 3: SHORT windows start in at least this size class (parity sweeps: every class's code on small windows)
        if (value < 0 || value > 3) return fail(HYPO_E_INVALID, "poa_min_class %d out of range 0..3", value);
        for (int i = 0; i < kMaxDevices; ++i) g_ctxs[i].poa_flags = (g_ctxs[i].poa_flags & ~(3 << hypo::POA_MIN_CLASS_SHIFT)) | (value << hypo::POA_MIN_CLASS_SHIFT);
        return HYPO_OK;
    }
    if (!strcmp(name, "giant_arena_mb")) {               // HBM of size class 6 per POA context created from now on (poa_giant.hpp); 0 = none
        if (value < 0 || value > (1 << 20)) return fail(HYPO_E_INVALID, "giant_arena_mb %d out of range", value);
        hypo::poa_set_giant_arena_mb(value);
        return HYPO_OK;
    }
    return fail(HYPO_E_INVALID, "unknown option %s", name);
}

int hypo_gpu_use_device(int slot) {
    if (slot < 0 || slot >= g_nctx) return fail(HYPO_E_INVALID, "context %d out of range (hypo_gpu_init created %d)", slot, g_nctx);
    tl_slot = slot;
    HIP_TRY(hipSetDevice(g_ctxs[slot].device));
    return HYPO_OK;
}

int hypo_gpu_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    g_rccl.release();
    for (int i = 0; i < g_nctx; ++i) release_ctx(g_ctxs[i]);
    g_nctx = 0;
    return HYPO_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------
int hypo_gpu_profile_begin(int max_calls) {
    HYPO_LOCKED();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (max_calls < 0 || max_calls > 256) return fail(HYPO_E_INVALID, "max_calls out of range 0..256");
    for (auto& c : g_prof.calls) for (auto& e : c.ke.ev) if (e) (void)hipEventDestroy(e);
    g_prof.calls.assign((size_t)max_calls, ProfCall());
    g_prof.used = 0;
    for (auto& c : g_prof.calls) for (auto& e : c.ke.ev) HIP_TRY(hipEventCreate(&e));
    return HYPO_OK;
}
int hypo_gpu_profile_calls(void) { return g_prof.used; }
int hypo_gpu_profile_read(int call, float* ms, int n) {
    HYPO_LOCKED();
    if (call < 0 || call >= g_prof.used || !ms) return fail(HYPO_E_INVALID, "no such profiled call");
    ProfCall& c = g_prof.calls[(size_t)call];
    if (c.ke.n < 2) return 0;
    HIP_TRY(hipEventSynchronize(c.ke.ev[c.ke.n - 1]));
    int out = 0;
    float t = 0.f;
    if (c.kind == 1) {          // POA: [plan, class 0..4, whole call]; event pairs may sit on different streams
        const int pairs = (c.ke.n - 1) / 2;
        for (int i = 0; i < pairs && out < n; ++i) {
            HIP_TRY(hipEventSynchronize(c.ke.ev[2 * i + 1]));
            HIP_TRY(hipEventElapsedTime(&t, c.ke.ev[2 * i], c.ke.ev[2 * i + 1]));
            ms[out++] = t;
        }
        if (out < n) { HIP_TRY(hipEventElapsedTime(&t, c.ke.ev[0], c.ke.ev[c.ke.n - 1])); ms[out++] = t; }
        return out;
    }
    for (int i = 0; i + 1 < c.ke.n && out < n; ++i) {
        HIP_TRY(hipEventElapsedTime(&t, c.ke.ev[i], c.ke.ev[i + 1]));
        ms[out++] = t;
    }
    return out;
}

// ---- POA -------------------------------------------------------------------------------------------
size_t hypo_gpu_poa_workspace_bytes(uint32_t n_windows, uint32_t n_arms) {
    return hypo::poa_workspace_bytes(n_windows, 0, n_arms);       // room for arm offsets computed on the device (arm_off == NULL)
}

int hypo_gpu_poa_batch_device(const HypoScoreParams* scores, const HypoWindowBatch* in, HypoConsensusBatch* out,
                              void* workspace, size_t workspace_bytes, void* hip_stream) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    int rc = check_scores(scores);
    if (rc) return rc;
    if (!in || !out) return fail(HYPO_E_INVALID, "NULL batch");
    if (in->n_windows == 0) return HYPO_OK;
    if (!in->windows || !in->draft4 || !out->bases || !out->off || !out->len || !out->status ||
        (in->n_arms && (!in->arm_len || !in->arms2)))
        return fail(HYPO_E_INVALID, "NULL buffer in batch");
    if (!workspace || workspace_bytes < hypo::poa_workspace_bytes(in->n_windows, hypo::kMinGlobalGroups, in->arm_off ? 0 : in->n_arms))
        return fail(HYPO_E_WORKSPACE, "workspace %zu < minimum %zu (hypo_gpu_poa_workspace_bytes recommends %zu)", workspace_bytes,
                    hypo::poa_workspace_bytes(in->n_windows, hypo::kMinGlobalGroups, in->arm_off ? 0 : in->n_arms), hypo::poa_workspace_bytes(in->n_windows, 0, in->n_arms));
    hypo::PoaParams P = make_params(scores, in, out);
    hipStream_t st = (hipStream_t)hip_stream;                  // NULL is the HIP null stream itself (what torch calls its default stream)
    ProfCall* pc = prof_next(1);
    HIP_TRY(hypo::poa_run(P, in->n_windows, workspace, workspace_bytes, g_ctx.num_cus, st, pc ? &pc->ke : nullptr, &g_ctx.slots[0].aux));
    return HYPO_OK;
}

// device-side stats of the last device call live at workspace + 128
int hypo_gpu_poa_read_stats(const void* workspace, void* hip_stream, HypoPoaStats* out) {
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    hipStream_t st = (hipStream_t)hip_stream;                  // NULL is the HIP null stream itself (what torch calls its default stream)
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(out, (const char*)workspace + 128, sizeof(HypoPoaStats), hipMemcpyDeviceToHost));
    return HYPO_OK;
}

int hypo_gpu_poa_last_stats(HypoPoaStats* out) {
    if (!out) return fail(HYPO_E_INVALID, "NULL");
    *out = tl_stats;
    return HYPO_OK;
}

int hypo_gpu_poa_slot_layout(const HypoWindowBatch* in, uint64_t* off) {
    if (!in || !off) return fail(HYPO_E_INVALID, "NULL");
    uint64_t acc = 0;
    for (uint32_t w = 0; w < in->n_windows; ++w) {
        const HypoWindow& W = in->windows[w];
        uint64_t longest = W.draft_len;
        const uint64_t narm = (uint64_t)W.n_internal + W.n_prefix + W.n_suffix;
        if (W.first_arm + narm > in->n_arms) return fail(HYPO_E_INVALID, "window %u: arms [%u, %llu) outside the batch's %u arms", w, W.first_arm, (unsigned long long)(W.first_arm + narm), in->n_arms);
        for (uint64_t a = 0; a < narm; ++a) if (in->arm_len[W.first_arm + a] > longest) longest = in->arm_len[W.first_arm + a];
        off[w] = acc;
        acc += (longest + longest / 2 + 24 + 7) / 8 * 8;
    }
    off[in->n_windows] = acc;
    return HYPO_OK;
}

// Queues one batch: upload, kernels, download of the results into the caller's buffers, all on the slot's stream; returns
// without waiting.  `in` and `out` buffers must stay untouched until hypo_gpu_poa_batch_end(ticket).
int hypo_gpu_poa_batch_begin(const HypoScoreParams* scores, const HypoWindowBatch* in, HypoConsensusBatch* out, int* ticket) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    int rc = check_scores(scores);
    if (rc) return rc;
    if (!in || !out || !ticket) return fail(HYPO_E_INVALID, "NULL batch");
    const uint32_t n = in->n_windows, na = in->n_arms;
    if (n && (!in->windows || !in->draft4 || !out->bases || !out->off || !out->len || !out->status || (na && (!in->arm_len || !in->arms2))))
        return fail(HYPO_E_INVALID, "NULL buffer in batch");
    int si = -1;
    for (int i = 0; i < 2; ++i) if (!g_ctx.slots[i].busy) { si = i; break; }
    if (si < 0) return fail(HYPO_E_INVALID, "two batches are already in flight on this context: call hypo_gpu_poa_batch_end first");
    Ctx::Slot& S = g_ctx.slots[si];
    S.n = n;
    *ticket = si;
    S.busy = true;
    if (n == 0) return HYPO_OK;
    const uint64_t out_bytes = out->off[n];
    DevBuf &dW = S.arena[0], &dD = S.arena[1], &dAO = S.arena[2], &dAL = S.arena[3], &dA = S.arena[4],
           &dB = S.arena[5], &dO = S.arena[6], &dL = S.arena[7], &dS = S.arena[8], &dWS = S.arena[9];
    // the host sees the window types: scratch for as many resident LONG groups as there are LONG windows (+ escalations)
    uint32_t n_long = 0;
    for (uint32_t w = 0; w < n; ++w) n_long += in->windows[w].type != HYPO_WIN_SHORT;
    const size_t wsb = hypo::poa_workspace_bytes(n, (int)(n_long + 64 < 2048u ? n_long + 64 : 2048u), in->arm_off ? 0 : na);
    auto undo = [&](int code) { S.busy = false; return code; };
#define HIP_TRY_S(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return undo(fail(HYPO_E_HIP, "%s: %s", #x, hipGetErrorString(e_))); } while (0)
    HIP_TRY_S(dW.alloc((size_t)n * sizeof(HypoWindow))); HIP_TRY_S(dD.alloc(in->draft4_bytes));
    if (in->arm_off) HIP_TRY_S(dAO.alloc((size_t)na * 8));
    HIP_TRY_S(dAL.alloc((size_t)na * 4)); HIP_TRY_S(dA.alloc(in->arms2_bytes));
    HIP_TRY_S(dB.alloc(out_bytes)); HIP_TRY_S(dO.alloc((size_t)(n + 1) * 8)); HIP_TRY_S(dL.alloc((size_t)n * 4));
    HIP_TRY_S(dS.alloc(n)); HIP_TRY_S(dWS.alloc(wsb));
    if (!S.stream) HIP_TRY_S(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    if (!S.stats_pinned) HIP_TRY_S(hipHostMalloc((void**)&S.stats_pinned, sizeof(HypoPoaStats), hipHostMallocDefault));
    hipStream_t st = S.stream;
    HIP_TRY_S(h2d(dW.p, in->windows, (size_t)n * sizeof(HypoWindow), st));
    HIP_TRY_S(h2d(dD.p, in->draft4, in->draft4_bytes, st));
    if (na) {
        if (in->arm_off) HIP_TRY_S(hipMemcpyAsync(dAO.p, in->arm_off, (size_t)na * 8, hipMemcpyHostToDevice, st));
        HIP_TRY_S(hipMemcpyAsync(dAL.p, in->arm_len, (size_t)na * 4, hipMemcpyHostToDevice, st));
        HIP_TRY_S(h2d(dA.p, in->arms2, in->arms2_bytes, st));
    }
    HIP_TRY_S(hipMemcpyAsync(dO.p, out->off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY_S(hipMemsetAsync(dB.p, 0, out_bytes ? out_bytes : 16, st));   // slack bytes of a slot come back as 0
    HypoWindowBatch din = *in;
    din.windows = (const HypoWindow*)dW.p; din.draft4 = (const uint8_t*)dD.p; din.arm_off = in->arm_off ? (const uint64_t*)dAO.p : nullptr;
    din.arm_len = (const uint32_t*)dAL.p; din.arms2 = (const uint8_t*)dA.p;
    HypoConsensusBatch dout;
    dout.bases = (char*)dB.p; dout.off = (const uint64_t*)dO.p; dout.len = (uint32_t*)dL.p; dout.status = (uint8_t*)dS.p;
    hypo::PoaParams P = make_params(scores, &din, &dout);
    // both slots share the side streams, events and plan history of slot 0: the runtime maps streams onto a handful of hardware
    // queues, and every extra stream ends up sharing one with a size-class kernel (measured: classes 0 and 1 serialised)
    HIP_TRY_S(hypo::poa_run(P, n, dWS.p, wsb, g_ctx.num_cus, st, nullptr, &g_ctx.slots[0].aux));
    HIP_TRY_S(hipMemcpyAsync(out->bases, dB.p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY_S(hipMemcpyAsync(out->len, dL.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY_S(hipMemcpyAsync(out->status, dS.p, n, hipMemcpyDeviceToHost, st));
    HIP_TRY_S(hipMemcpyAsync(S.stats_pinned, (char*)dWS.p + 128, sizeof(HypoPoaStats), hipMemcpyDeviceToHost, st));   // (page-locked: a copy into pageable memory would wait for the whole batch here)
#undef HIP_TRY_S
    return HYPO_OK;
}

int hypo_gpu_poa_batch_end(int ticket) {
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (ticket < 0 || ticket > 1 || !g_ctx.slots[ticket].busy) return fail(HYPO_E_INVALID, "no batch in flight under ticket %d", ticket);
    Ctx::Slot& S = g_ctx.slots[ticket];
    HYPO_ON_DEVICE();
    const hipError_t e = S.n ? hipStreamSynchronize(S.stream) : hipSuccess;      // (no context lock while waiting: the other slot stays usable)
    HYPO_LOCKED();
    S.busy = false;
    if (e != hipSuccess) return fail(HYPO_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
    if (S.n && S.stats_pinned) tl_stats = *S.stats_pinned; else memset(&tl_stats, 0, sizeof(tl_stats));
    tl_stats.n_windows = S.n;
    return HYPO_OK;
}

int hypo_gpu_poa_batch(const HypoScoreParams* scores, const HypoWindowBatch* in, HypoConsensusBatch* out) {
    memset(&tl_stats, 0, sizeof(tl_stats));
    int ticket = -1;
    const int rc = hypo_gpu_poa_batch_begin(scores, in, out, &ticket);
    if (rc) return rc;
    return hypo_gpu_poa_batch_end(ticket);
}

// ---- POA over all contexts ----------------------------------------------------------------------------
// Windows are independent (src/Hypo.cpp:238-247 is a parallel loop over them): the batch is cut into one contiguous,
// cost-balanced range per device, every device polishes its range, and the consensus bytes, lengths and status bytes are
// exchanged with ONE grouped RCCL all-gatherv over xGMI (ncclBroadcast per owner inside a group: the ranges differ in
// size) so that every device, device 0 in particular, holds the whole result before it goes back to the host for contig
// re-assembly (src/Contig.cpp:345-366).  RCCL is loaded on first use (a single-device run never touches it).
namespace {
struct Share { uint32_t w0 = 0, w1 = 0; uint64_t a0 = 0, a1 = 0, d0 = 0, d1 = 0, b0 = 0, b1 = 0; int rc = HYPO_OK; char err[256] = ""; HypoPoaStats st; };

// cost of a window as the plan sees it (rows x sequences), LONG windows twice
inline uint64_t window_cost(const HypoWindowBatch* in, uint32_t w) {
    const HypoWindow& W = in->windows[w];
    const uint64_t narm = (uint64_t)W.n_internal + W.n_prefix + W.n_suffix;
    uint64_t s = 0;
    if ((uint64_t)W.first_arm + narm <= in->n_arms) for (uint64_t a = 0; a < narm; ++a) s += in->arm_len[W.first_arm + a] + 1;
    return ((uint64_t)W.draft_len + 2) * (s + 1) * (W.type != HYPO_WIN_SHORT ? 2 : 1) + 64;
}
}  // namespace

int hypo_gpu_poa_batch_sharded(const HypoScoreParams* scores, const HypoWindowBatch* in, HypoConsensusBatch* out) {
    const int nd = g_nctx;
    if (nd <= 0) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (nd == 1 && !getenv("HYPO_MULTI_GATHER")) { const int keep = tl_slot; tl_slot = 0; const int rc = hypo_gpu_poa_batch(scores, in, out); tl_slot = keep; return rc; }
    std::lock_guard<std::mutex> shard_lock(g_shard_mu);
    int rc = check_scores(scores);
    if (rc) return rc;
    if (!in || !out) return fail(HYPO_E_INVALID, "NULL batch");
    memset(&tl_stats, 0, sizeof(tl_stats));
    const uint32_t n = in->n_windows;
    if (n == 0) return HYPO_OK;
    if (!in->windows || !in->draft4 || !out->bases || !out->off || !out->len || !out->status ||
        (in->n_arms && (!in->arm_off || !in->arm_len || !in->arms2)))
        return fail(HYPO_E_INVALID, "NULL buffer in batch");
    // ---- 1. contiguous cost-balanced ranges: per-chunk costs on nd threads, then a short prefix walk ----
    std::vector<uint64_t> cost(n);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nd; ++t) th.emplace_back([&, t]() {
            const uint32_t lo = (uint32_t)((uint64_t)n * t / nd), hi = (uint32_t)((uint64_t)n * (t + 1) / nd);
            for (uint32_t w = lo; w < hi; ++w) cost[w] = window_cost(in, w);
        });
        for (auto& t : th) t.join();
    }
    uint64_t total = 0;
    for (uint32_t w = 0; w < n; ++w) total += cost[w];
    std::vector<Share> sh((size_t)nd);
    {
        uint64_t acc = 0; uint32_t w = 0;
        for (int d = 0; d < nd; ++d) {
            sh[d].w0 = w;
            const uint64_t target = total / nd * (uint64_t)(d + 1) + (d + 1 == nd ? total % nd : 0);
            while (w < n && (d + 1 == nd || acc + cost[w] / 2 < target)) acc += cost[w++];
            sh[d].w1 = w;
        }
    }
    // ---- 2. every device: rebase its descriptors, upload its slices, run ----
    const bool want_rccl = nd > 1 ? !(getenv("HYPO_MULTI_GATHER") && !strcmp(getenv("HYPO_MULTI_GATHER"), "direct"))
                                  : (getenv("HYPO_MULTI_GATHER") && !strcmp(getenv("HYPO_MULTI_GATHER"), "rccl"));
    bool use_rccl = want_rccl && g_rccl.load();
    if (use_rccl) {                                    // communicators for the current device list (created once)
        bool same = g_rccl.n_comms == nd;
        for (int d = 0; same && d < nd; ++d) same = g_rccl.devs[d] == g_ctxs[d].device;
        if (!same) {
            g_rccl.release();
            int devs[kMaxDevices];
            for (int d = 0; d < nd; ++d) devs[d] = g_rccl.devs[d] = g_ctxs[d].device;
            const int r = g_rccl.CommInitAll(g_rccl.comms, nd, devs);
            if (r != 0) { fprintf(stderr, "[hypo_gpu] RCCL unavailable for this device list (%s): results go back per device\n", g_rccl.GetErrorString(r)); use_rccl = false; }
            else g_rccl.n_comms = nd;
        }
    }
    const uint64_t out_bytes = out->off[n];
    auto device_part = [&](int d) {
        Share& S = sh[d];
        Ctx& c = g_ctxs[d];
        std::lock_guard<std::recursive_mutex> lk(c.mu);
        auto bad = [&](int code, const char* what, hipError_t e) { S.rc = code; snprintf(S.err, sizeof(S.err), "device %d: %s: %s", c.device, what, hipGetErrorString(e)); };
        hipError_t e = hipSetDevice(c.device);
        if (e != hipSuccess) return bad(HYPO_E_HIP, "hipSetDevice", e);
        const uint32_t nw = S.w1 - S.w0;
        // result buffers of the WHOLE batch on every device (the gather fills the other devices' ranges)
        DevBuf* const ar = c.slots[0].arena;
        DevBuf &dW = ar[0], &dD = ar[1], &dAO = ar[2], &dAL = ar[3], &dA = ar[4], &dB = ar[5], &dO = ar[6], &dL = ar[7], &dS = ar[8], &dWS = ar[9];
        if ((e = dB.alloc(out_bytes)) != hipSuccess || (e = dL.alloc((size_t)n * 4)) != hipSuccess || (e = dS.alloc(n)) != hipSuccess)
            return bad(HYPO_E_HIP, "hipMalloc", e);
        memset(&S.st, 0, sizeof(S.st));
        if (nw == 0) return;
        // ranges of the shared buffers this share touches
        uint64_t a0 = ~0ull, a1 = 0, d0 = ~0ull, d1 = 0;
        for (uint32_t w = S.w0; w < S.w1; ++w) {
            const HypoWindow& W = in->windows[w];
            const uint64_t narm = (uint64_t)W.n_internal + W.n_prefix + W.n_suffix;
            if (narm && (uint64_t)W.first_arm + narm <= in->n_arms) { a0 = W.first_arm < a0 ? W.first_arm : a0; a1 = W.first_arm + narm > a1 ? W.first_arm + narm : a1; }
            if (W.draft_off <= in->draft4_bytes) {
                const uint64_t de = W.draft_off + ((uint64_t)W.draft_len + 1) / 2;
                d0 = W.draft_off < d0 ? W.draft_off : d0; d1 = (de < in->draft4_bytes ? de : in->draft4_bytes) > d1 ? (de < in->draft4_bytes ? de : in->draft4_bytes) : d1;
            }
        }
        if (a0 == ~0ull) { a0 = 0; a1 = 0; }
        if (d0 == ~0ull) { d0 = 0; d1 = 0; }
        uint64_t b0 = ~0ull, b1 = 0;
        for (uint64_t a = a0; a < a1; ++a) {
            const uint64_t o = in->arm_off[a], e2 = o + ((uint64_t)in->arm_len[a] + 3) / 4;
            if (o > in->arms2_bytes) continue;                 // left as it is: the device answers the window with HYPO_ST_INVALID
            b0 = o < b0 ? o : b0; b1 = (e2 < in->arms2_bytes ? e2 : in->arms2_bytes) > b1 ? (e2 < in->arms2_bytes ? e2 : in->arms2_bytes) : b1;
        }
        if (b0 == ~0ull) { b0 = 0; b1 = 0; }
        S.a0 = a0; S.a1 = a1; S.d0 = d0; S.d1 = d1; S.b0 = b0; S.b1 = b1;
        c.sh_win.assign(in->windows + S.w0, in->windows + S.w1);
        for (auto& W : c.sh_win) {
            // a window without arms does not take part in [a0, a1): its first_arm is rebased to 0 when it was valid in the whole
            // batch (the single-device call answers it with an empty consensus or its draft) and stays out of range otherwise
            if ((uint64_t)W.n_internal + W.n_prefix + W.n_suffix == 0) W.first_arm = (uint64_t)W.first_arm <= in->n_arms ? 0u : 0xffffffffu;
            else W.first_arm -= (uint32_t)(W.first_arm >= a0 ? a0 : 0);
            W.draft_off -= (W.draft_off >= d0 ? d0 : 0);
        }
        c.sh_aoff.resize((size_t)(a1 - a0));
        for (uint64_t a = a0; a < a1; ++a) c.sh_aoff[(size_t)(a - a0)] = in->arm_off[a] >= b0 ? in->arm_off[a] - b0 : in->arm_off[a];
        c.sh_off.resize((size_t)nw + 1);
        for (uint32_t i = 0; i <= nw; ++i) c.sh_off[i] = out->off[S.w0 + i];      // GLOBAL slot offsets: results land where the gather expects them
        uint32_t n_long = 0;
        for (const auto& W : c.sh_win) n_long += W.type != HYPO_WIN_SHORT;
        const size_t wsb = hypo::poa_workspace_bytes(nw, (int)(n_long + 64 < 2048u ? n_long + 64 : 2048u));
        if ((e = dW.alloc((size_t)nw * sizeof(HypoWindow))) != hipSuccess || (e = dD.alloc(d1 - d0)) != hipSuccess ||
            (e = dAO.alloc((a1 - a0) * 8)) != hipSuccess || (e = dAL.alloc((a1 - a0) * 4)) != hipSuccess ||
            (e = dA.alloc(b1 - b0)) != hipSuccess || (e = dO.alloc(((size_t)nw + 1) * 8)) != hipSuccess || (e = dWS.alloc(wsb)) != hipSuccess)
            return bad(HYPO_E_HIP, "hipMalloc", e);
        hipStream_t st = c.stream;
        (void)hipMemcpyAsync(dW.p, c.sh_win.data(), (size_t)nw * sizeof(HypoWindow), hipMemcpyHostToDevice, st);
        if (d1 > d0) (void)hipMemcpyAsync(dD.p, in->draft4 + d0, d1 - d0, hipMemcpyHostToDevice, st);
        if (a1 > a0) {
            (void)hipMemcpyAsync(dAO.p, c.sh_aoff.data(), (a1 - a0) * 8, hipMemcpyHostToDevice, st);
            (void)hipMemcpyAsync(dAL.p, in->arm_len + a0, (a1 - a0) * 4, hipMemcpyHostToDevice, st);
            if (b1 > b0) (void)hipMemcpyAsync(dA.p, in->arms2 + b0, b1 - b0, hipMemcpyHostToDevice, st);
        }
        (void)hipMemcpyAsync(dO.p, c.sh_off.data(), ((size_t)nw + 1) * 8, hipMemcpyHostToDevice, st);
        (void)hipMemsetAsync((char*)dB.p + out->off[S.w0], 0, out->off[S.w1] - out->off[S.w0] ? out->off[S.w1] - out->off[S.w0] : 1, st);
        if ((e = hipGetLastError()) != hipSuccess) return bad(HYPO_E_HIP, "upload", e);
        hypo::PoaParams P;
        P.windows = (const HypoWindow*)dW.p; P.draft4 = (const uint8_t*)dD.p; P.arm_off = (const uint64_t*)dAO.p;
        P.arm_len = (const uint32_t*)dAL.p; P.arms2 = (const uint8_t*)dA.p;
        P.out_bases = (char*)dB.p; P.out_off = (const uint64_t*)dO.p;
        P.out_len = (uint32_t*)dL.p + S.w0; P.out_status = (uint8_t*)dS.p + S.w0;
        P.sr_m = scores->sr_match; P.sr_n = scores->sr_mismatch; P.sr_g = scores->sr_gap;
        P.lr_m = scores->lr_match; P.lr_n = scores->lr_mismatch; P.lr_g = scores->lr_gap;
        P.n_arms = a1 - a0; P.draft4_bytes = d1 - d0; P.arms2_bytes = b1 - b0; P.flags = c.poa_flags;
        if ((e = hypo::poa_run(P, nw, dWS.p, wsb, c.num_cus, st, nullptr, &c.slots[0].aux)) != hipSuccess) return bad(HYPO_E_HIP, "poa_run", e);
        (void)hipMemcpyAsync(&S.st, (char*)dWS.p + 128, sizeof(HypoPoaStats), hipMemcpyDeviceToHost, st);
        if (!use_rccl) {                                   // results of this share straight back to the caller's buffers
            (void)hipMemcpyAsync(out->bases + out->off[S.w0], (char*)dB.p + out->off[S.w0], out->off[S.w1] - out->off[S.w0], hipMemcpyDeviceToHost, st);
            (void)hipMemcpyAsync(out->len + S.w0, (uint32_t*)dL.p + S.w0, (size_t)nw * 4, hipMemcpyDeviceToHost, st);
            (void)hipMemcpyAsync(out->status + S.w0, (uint8_t*)dS.p + S.w0, nw, hipMemcpyDeviceToHost, st);
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return bad(HYPO_E_HIP, "hipStreamSynchronize", e);
        }
    };
    {
        std::vector<std::thread> th;
        for (int d = 0; d < nd; ++d) th.emplace_back(device_part, d);
        for (auto& t : th) t.join();
    }
    for (int d = 0; d < nd; ++d) if (sh[d].rc != HYPO_OK) return fail(sh[d].rc, "%s", sh[d].err);
    if (use_rccl) {
        // ---- 3. all-gatherv: owner r broadcasts its three ranges into the same places of every device's whole-batch buffers ----
        int r = g_rccl.GroupStart();
        for (int root = 0; r == 0 && root < nd; ++root) {
            const Share& S = sh[root];
            if (S.w1 == S.w0) continue;
            for (int d = 0; r == 0 && d < nd; ++d) {
                Ctx& c = g_ctxs[d];
                char* B = (char*)c.slots[0].arena[5].p; uint32_t* L = (uint32_t*)c.slots[0].arena[7].p; uint8_t* St = (uint8_t*)c.slots[0].arena[8].p;
                const size_t nb = out->off[S.w1] - out->off[S.w0], nw = S.w1 - S.w0;
                if (nb) r = g_rccl.Broadcast(B + out->off[S.w0], B + out->off[S.w0], nb, kNcclUint8, root, g_rccl.comms[d], c.stream);
                if (r == 0) r = g_rccl.Broadcast(L + S.w0, L + S.w0, nw * 4, kNcclUint8, root, g_rccl.comms[d], c.stream);
                if (r == 0) r = g_rccl.Broadcast(St + S.w0, St + S.w0, nw, kNcclUint8, root, g_rccl.comms[d], c.stream);
            }
        }
        const int r2 = g_rccl.GroupEnd();
        if (r != 0 || r2 != 0) return fail(HYPO_E_HIP, "RCCL all-gather of the consensus: %s", g_rccl.GetErrorString(r ? r : r2));
        // ---- 4. device 0 holds everything: one copy back for contig re-assembly; the other devices only have to finish ----
        Ctx& c0 = g_ctxs[0];
        HIP_TRY(hipSetDevice(c0.device));
        HIP_TRY(hipMemcpyAsync(out->bases, c0.slots[0].arena[5].p, out_bytes, hipMemcpyDeviceToHost, c0.stream));
        HIP_TRY(hipMemcpyAsync(out->len, c0.slots[0].arena[7].p, (size_t)n * 4, hipMemcpyDeviceToHost, c0.stream));
        HIP_TRY(hipMemcpyAsync(out->status, c0.slots[0].arena[8].p, n, hipMemcpyDeviceToHost, c0.stream));
        for (int d = nd - 1; d >= 0; --d) { HIP_TRY(hipSetDevice(g_ctxs[d].device)); HIP_TRY(hipStreamSynchronize(g_ctxs[d].stream)); }
    }
    HIP_TRY(hipSetDevice(g_ctx.device));
    for (int d = 0; d < nd; ++d) {
        const HypoPoaStats& s = sh[d].st;
        tl_stats.n_trivial += s.n_trivial; tl_stats.n_escalated += s.n_escalated; tl_stats.n_failed += s.n_failed;
        tl_stats.dp_cells += s.dp_cells; tl_stats.n_alignments += s.n_alignments;
        tl_stats.n_reused += s.n_reused; tl_stats.n_threaded += s.n_threaded; tl_stats.cells_scored += s.cells_scored; tl_stats.cells_threaded += s.cells_threaded; tl_stats.n_carried += s.n_carried;
        for (int k = 0; k < 8; ++k) { tl_stats.n_class[k] += s.n_class[k]; tl_stats.alg_bytes[k] += s.alg_bytes[k]; }
    }
    tl_stats.n_windows = n;
    return HYPO_OK;
}

// ---- solid scan ------------------------------------------------------------------------------------
size_t hypo_gpu_solid_scan_workspace_bytes(uint64_t n_bases) { return hypo::scan_workspace_bytes(n_bases); }

int hypo_gpu_solid_scan_device(const uint8_t* packed4, uint64_t n_bases, uint32_t k, const uint64_t* bits,
                               uint64_t* solid_pos_words, uint64_t* kids, uint64_t kids_cap,
                               uint64_t* word_rank, uint64_t* n_solid,
                               void* workspace, size_t workspace_bytes, void* hip_stream) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (k < 2 || k > 31) return fail(HYPO_E_INVALID, "k=%u out of range 2..31", k);
    if ((n_bases && !packed4) || !bits || (n_bases && !solid_pos_words)) return fail(HYPO_E_INVALID, "NULL buffer");
    if (!workspace || workspace_bytes < hypo::scan_workspace_bytes(n_bases))
        return fail(HYPO_E_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, hypo::scan_workspace_bytes(n_bases));
    hipStream_t st = (hipStream_t)hip_stream;                  // NULL is the HIP null stream itself (what torch calls its default stream)
    ProfCall* pc = prof_next(2);
    if (pc) pc->ke.n = 4;
    HIP_TRY(hypo::scan_run(packed4, n_bases, k, bits, solid_pos_words, kids, kids_cap, word_rank, n_solid,
                           workspace, workspace_bytes, st, pc ? pc->ke.ev : nullptr));
    return HYPO_OK;
}

int hypo_gpu_solid_set_upload(const uint64_t* bits, uint32_t k) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (k < 2 || k > 31) return fail(HYPO_E_INVALID, "k=%u out of range 2..31", k);
    if (!bits) return fail(HYPO_E_INVALID, "NULL buffer");
    const uint64_t bit_words = (1ull << (2 * k)) / 64 ? (1ull << (2 * k)) / 64 : 1;
    g_ctx.solid_k = 0;
    HIP_TRY(g_ctx.solid_set.alloc(bit_words * 8));
    HIP_TRY(h2d(g_ctx.solid_set.p, bits, bit_words * 8, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    g_ctx.solid_k = k;
    return HYPO_OK;
}

int hypo_gpu_solid_scan(const uint8_t* packed4, uint64_t n_bases, uint32_t k, const uint64_t* bits,
                        uint64_t* solid_pos_words, uint64_t* kids, uint64_t kids_cap,
                        uint64_t* word_rank, uint64_t* n_solid) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (k < 2 || k > 31) return fail(HYPO_E_INVALID, "k=%u out of range 2..31", k);
    if ((n_bases && !packed4) || (n_bases && !solid_pos_words)) return fail(HYPO_E_INVALID, "NULL buffer");
    if (!bits && g_ctx.solid_k != k) return fail(HYPO_E_INVALID, "bitset_words == NULL but no %u-mer set was uploaded (hypo_gpu_solid_set_upload)", k);
    const uint64_t nw = (n_bases + 63) / 64, nbytes = (n_bases + 1) / 2, bit_words = (1ull << (2 * k)) / 64 ? (1ull << (2 * k)) / 64 : 1;
    DevBuf* const sa = g_ctx.scan_arena;
    DevBuf &dP = sa[0], &dBits = sa[1], &dWords = sa[2], &dKids = sa[3], &dRank = sa[4], &dN = sa[5], &dWS = sa[6];
    const size_t wsb = hypo::scan_workspace_bytes(n_bases);
    HIP_TRY(dP.alloc(nbytes)); if (bits) HIP_TRY(dBits.alloc(bit_words * 8)); HIP_TRY(dWords.alloc(nw * 8));
    HIP_TRY(dKids.alloc(kids_cap * 8)); HIP_TRY(dRank.alloc((nw + 1) * 8)); HIP_TRY(dN.alloc(8)); HIP_TRY(dWS.alloc(wsb));
    hipStream_t st = g_ctx.stream;
    if (nbytes) HIP_TRY(h2d(dP.p, packed4, nbytes, st));
    if (bits) HIP_TRY(h2d(dBits.p, bits, bit_words * 8, st));
    int rc = hypo_gpu_solid_scan_device((const uint8_t*)dP.p, n_bases, k, (const uint64_t*)(bits ? dBits.p : g_ctx.solid_set.p), (uint64_t*)dWords.p,
                                        kids ? (uint64_t*)dKids.p : nullptr, kids ? kids_cap : 0, (uint64_t*)dRank.p,
                                        (uint64_t*)dN.p, dWS.p, wsb, st);
    if (rc) return rc;
    uint64_t ns = 0;
    HIP_TRY(hipMemcpyAsync(&ns, dN.p, 8, hipMemcpyDeviceToHost, st));
    if (nw) HIP_TRY(d2h(solid_pos_words, dWords.p, nw * 8, st));
    if (word_rank) HIP_TRY(d2h(word_rank, dRank.p, (nw + 1) * 8, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (kids && kids_cap) {
        const uint64_t cnt = ns < kids_cap ? ns : kids_cap;
        if (cnt) HIP_TRY(hipMemcpy(kids, dKids.p, cnt * 8, hipMemcpyDeviceToHost));
    }
    if (n_solid) *n_solid = ns;
    return HYPO_OK;
}

// ---- a scan whose k-mer ids and positions stay on the device (round 4) -------------------------------------------------------
// The host of a run needs the mark bits and their rank directory (Contig::_solid_pos); the k-mer ids and the positions are read
// by the support votes only, on the device: 8 bytes per marked position went to pageable host vectors here and came back, 12 per
// position with the positions, for hypo_gpu_support_kmers (2 x 2 GB at 250 Mbp / k = 15, where nearly every position is marked).
int hypo_gpu_solid_scan_keep(uint32_t handle, const uint8_t* packed4, uint64_t n_bases, uint32_t k,
                             uint64_t* solid_pos_words, uint64_t* word_rank, uint64_t* n_solid) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (k < 2 || k > 31) return fail(HYPO_E_INVALID, "k=%u out of range 2..31", k);
    if ((n_bases && !packed4) || (n_bases && !solid_pos_words)) return fail(HYPO_E_INVALID, "NULL buffer");
    if (g_ctx.solid_k != k) return fail(HYPO_E_INVALID, "no %u-mer set was uploaded (hypo_gpu_solid_set_upload)", k);
    if (handle >= (1u << 24) || n_bases >= 0xfffffff0ull) return fail(HYPO_E_INVALID, "handle or contig length out of range");
    const uint64_t nw = (n_bases + 63) / 64, nbytes = (n_bases + 1) / 2, cap = n_bases ? n_bases : 1;
    const bool narrow = k <= 16;
    DevBuf* const sa = g_ctx.scan_arena;
    DevBuf &dP = sa[0], &dWords = sa[2], &dKids = sa[3], &dRank = sa[4], &dN = sa[5], &dWS = sa[6];
    const size_t wsb = hypo::scan_workspace_bytes(n_bases);
    const size_t kid_bytes = narrow ? 4 : 8;
    HIP_TRY(dP.alloc(nbytes)); HIP_TRY(dWords.alloc(nw * 8)); HIP_TRY(dKids.alloc(cap * (kid_bytes + 4) + 256));
    HIP_TRY(dRank.alloc((nw + 1) * 8)); HIP_TRY(dN.alloc(8)); HIP_TRY(dWS.alloc(wsb));
    hipStream_t st = g_ctx.stream;
    char* const kid_arena = (char*)dKids.p;
    uint32_t* const spos_arena = (uint32_t*)(kid_arena + (cap * kid_bytes + 255) / 256 * 256);
    if (nbytes) HIP_TRY(h2d(dP.p, packed4, nbytes, st));
    HIP_TRY(hypo::scan_run((const uint8_t*)dP.p, n_bases, k, (const uint64_t*)g_ctx.solid_set.p, (uint64_t*)dWords.p,
                           narrow ? nullptr : (uint64_t*)kid_arena, cap, (uint64_t*)dRank.p, (uint64_t*)dN.p, dWS.p, wsb, st, nullptr,
                           narrow ? (uint32_t*)kid_arena : nullptr, spos_arena));
    uint64_t ns = 0;
    HIP_TRY(hipMemcpyAsync(&ns, dN.p, 8, hipMemcpyDeviceToHost, st));
    if (nw) HIP_TRY(d2h(solid_pos_words, dWords.p, nw * 8, st));
    if (word_rank) HIP_TRY(d2h(word_rank, dRank.p, (nw + 1) * 8, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (handle >= g_ctx.kept.size()) g_ctx.kept.resize((size_t)handle + 1);
    Ctx::KeptScan& ks = g_ctx.kept[handle];
    if (ks.kids) { (void)hipFree(ks.kids); ks.kids = nullptr; }
    if (ks.spos) { (void)hipFree(ks.spos); ks.spos = nullptr; }
    ks.used = false;
    if (ns) {
        HIP_TRY(hipMalloc(&ks.kids, ns * kid_bytes));
        HIP_TRY(hipMalloc((void**)&ks.spos, ns * 4));
        HIP_TRY(hipMemcpyAsync(ks.kids, kid_arena, ns * kid_bytes, hipMemcpyDeviceToDevice, st));      // (stream order keeps the arena
        HIP_TRY(hipMemcpyAsync(ks.spos, spos_arena, ns * 4, hipMemcpyDeviceToDevice, st));             //  intact until these ran)
    }
    ks.n = ns; ks.n_bases = n_bases; ks.k = k; ks.used = true;
    if (n_solid) *n_solid = ns;
    return HYPO_OK;
}

int hypo_gpu_solid_release(uint32_t handle) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    Ctx& cx = g_ctx;
    auto drop = [&cx](Ctx::KeptScan& ks) {
        const size_t kb = (size_t)ks.n * (ks.k <= 16 ? 4 : 8), sb = (size_t)ks.n * 4;
        if (ks.kids) { cx.kept_parked.push_back(ks.kids); cx.kept_parked_bytes += kb; }
        if (ks.spos) { cx.kept_parked.push_back(ks.spos); cx.kept_parked_bytes += sb; }
        ks = Ctx::KeptScan();
    };
    if (handle == 0xffffffffu) for (auto& ks : cx.kept) drop(ks);
    else {
        if (handle >= cx.kept.size() || !cx.kept[handle].used) return fail(HYPO_E_INVALID, "handle %u holds no scan on this context", handle);
        drop(cx.kept[handle]);
    }
    if (handle == 0xffffffffu || cx.kept_parked_bytes > ((size_t)16 << 30)) {
        HIP_TRY(hipStreamSynchronize(cx.stream));
        for (void* p : cx.kept_parked) (void)hipFree(p);
        cx.kept_parked.clear(); cx.kept_parked_bytes = 0;
    }
    return HYPO_OK;
}

int hypo_gpu_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(HYPO_E_INVALID, "NULL");
    *out = nullptr;
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return HYPO_OK;
}
int hypo_gpu_host_free(void* p) {
    if (!p) return HYPO_OK;
    HIP_TRY(hipHostFree(p));
    return HYPO_OK;
}
int hypo_gpu_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return fail(HYPO_E_INVALID, "NULL / empty range");
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return HYPO_OK;
}
int hypo_gpu_host_unregister(void* p) {
    if (!p) return HYPO_OK;
    HIP_TRY(hipHostUnregister(p));
    return HYPO_OK;
}

// ---- support votes on the device (kernels in support_kernel.hip) -----------------------------------------

int hypo_gpu_reads_upload(const HypoArmsReads* A, const uint32_t* read_contig, uint64_t total_len) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    auto& rr = g_ctx.rr;
    rr.ready = false;
    if (!A || !read_contig) return fail(HYPO_E_INVALID, "NULL argument");
    const uint32_t na = A->n_alignments;
    if (na && (!A->rb || !A->re || !A->qae || !A->seq_off || !A->reads2 || !A->cigar_off || !A->cigar)) return fail(HYPO_E_INVALID, "NULL buffer in reads");
    uint32_t max_span = 0; uint64_t sum_span = 0;
    rr.ctg_min_rb.clear(); rr.ctg_max_re.clear();
    {   // every record is checked before anything is sent (10 M records per 50 Mbp batch: 40 ms on one thread, so eight share them)
        struct Part { uint32_t bad = 0xffffffffu, max_span = 0; uint64_t sum_span = 0; std::vector<uint32_t> lo, hi; };
        const int P = na >= (1u << 18) ? 8 : 1;
        std::vector<Part> part((size_t)P);
        auto bad_record = [&](uint32_t a) -> bool {
            return A->re[a] <= A->rb[a] || A->re[a] > total_len || (a && A->rb[a - 1] > A->rb[a]) || A->seq_off[a] + ((uint64_t)A->qae[a] + 3) / 4 > A->reads2_bytes ||
                   A->cigar_off[a] > A->cigar_off[a + 1] || read_contig[a] >= 0x01000000u;
        };
        auto run = [&](int t) {
            Part& pt = part[(size_t)t];
            const uint32_t a0 = (uint32_t)((uint64_t)na * (uint64_t)t / (uint64_t)P), a1 = (uint32_t)((uint64_t)na * ((uint64_t)t + 1) / (uint64_t)P);
            for (uint32_t a = a0; a < a1; ++a) {
                if (bad_record(a)) { pt.bad = a; return; }
                const uint32_t span = A->re[a] - A->rb[a];
                pt.max_span = span > pt.max_span ? span : pt.max_span;
                pt.sum_span += span;
                const uint32_t rc = read_contig[a];
                if (rc >= pt.lo.size()) { pt.lo.resize((size_t)rc + 1, 0xffffffffu); pt.hi.resize((size_t)rc + 1, 0u); }
                if (A->rb[a] < pt.lo[rc]) pt.lo[rc] = A->rb[a];
                if (A->re[a] > pt.hi[rc]) pt.hi[rc] = A->re[a];
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < P; ++t) th.emplace_back(run, t);
        run(0);
        for (auto& x : th) x.join();
        for (const Part& pt : part) {
            if (pt.bad != 0xffffffffu) {                       // (parts are in record order: this is the first bad record)
                const uint32_t a = pt.bad;
                if (A->re[a] <= A->rb[a] || A->re[a] > total_len) return fail(HYPO_E_INVALID, "alignment %u: span [%u, %u) outside the %llu bases", a, A->rb[a], A->re[a], (unsigned long long)total_len);
                if (a && A->rb[a - 1] > A->rb[a]) return fail(HYPO_E_INVALID, "alignments are not sorted by reference start (alignment %u)", a);
                if (A->seq_off[a] + ((uint64_t)A->qae[a] + 3) / 4 > A->reads2_bytes) return fail(HYPO_E_INVALID, "alignment %u: read outside reads2", a);
                if (A->cigar_off[a] > A->cigar_off[a + 1]) return fail(HYPO_E_INVALID, "alignment %u: cigar_off decreases", a);
                return fail(HYPO_E_INVALID, "alignment %u: contig index %u out of range", a, read_contig[a]);
            }
            max_span = pt.max_span > max_span ? pt.max_span : max_span;
            sum_span += pt.sum_span;
            if (pt.lo.size() > rr.ctg_min_rb.size()) { rr.ctg_min_rb.resize(pt.lo.size(), 0xffffffffu); rr.ctg_max_re.resize(pt.lo.size(), 0u); }
            for (size_t c = 0; c < pt.lo.size(); ++c) {
                if (pt.lo[c] < rr.ctg_min_rb[c]) rr.ctg_min_rb[c] = pt.lo[c];
                if (pt.hi[c] > rr.ctg_max_re[c]) rr.ctg_max_re[c] = pt.hi[c];
            }
        }
    }
    rr.total_len = total_len;
    const uint64_t n_cig = na ? A->cigar_off[na] : 0;
    Carver c;
    const size_t o_rb = c.take((size_t)na * 4), o_re = c.take((size_t)na * 4), o_qae = c.take((size_t)na * 4), o_soff = c.take((size_t)na * 8),
                 o_reads = c.take(A->reads2_bytes), o_coff = c.take((size_t)(na + 1) * 4), o_cig = c.take(n_cig * 4), o_ctg = c.take((size_t)na * 4),
                 o_rank = c.take(A->file_rank ? (size_t)na * 4 : 0);
    HIP_TRY(rr.data.alloc(c.at ? c.at : 256));
    char* d = (char*)rr.data.p;
    hipStream_t st = g_ctx.stream;
#define UP(off, src, bytes) do { if (bytes) HIP_TRY(h2d(d + (off), (src), (bytes), st)); } while (0)
    UP(o_rb, A->rb, (size_t)na * 4); UP(o_re, A->re, (size_t)na * 4); UP(o_qae, A->qae, (size_t)na * 4); UP(o_soff, A->seq_off, (size_t)na * 8);
    UP(o_reads, A->reads2, A->reads2_bytes); if (na) UP(o_coff, A->cigar_off, (size_t)(na + 1) * 4); UP(o_cig, A->cigar, n_cig * 4); UP(o_ctg, read_contig, (size_t)na * 4);
    if (A->file_rank) UP(o_rank, A->file_rank, (size_t)na * 4);
#undef UP
    HIP_TRY(hipStreamSynchronize(st));                          // the caller may release its arrays
    rr.file_rank = A->file_rank ? (const uint32_t*)(d + o_rank) : nullptr;
    rr.n = na; rr.n_cig = n_cig; rr.reads2_bytes = A->reads2_bytes; rr.max_span = max_span; rr.sum_span = sum_span;
    rr.rb = (const uint32_t*)(d + o_rb); rr.re = (const uint32_t*)(d + o_re); rr.qae = (const uint32_t*)(d + o_qae); rr.seq_off = (const uint64_t*)(d + o_soff);
    rr.reads2 = (const uint8_t*)(d + o_reads); rr.cigar_off = (const uint32_t*)(d + o_coff); rr.cigar = (const uint32_t*)(d + o_cig); rr.read_contig = (const uint32_t*)(d + o_ctg);
    rr.ready = true;
    return HYPO_OK;
}

static hypo::SupportReads support_reads_of(const Ctx& c) {
    hypo::SupportReads R;
    R.n_alignments = c.rr.n; R.rb = c.rr.rb; R.re = c.rr.re; R.qae = c.rr.qae; R.seq_off = c.rr.seq_off; R.reads2 = c.rr.reads2; R.read_contig = c.rr.read_contig;
    R.mean_span = c.rr.n ? (uint32_t)(c.rr.sum_span / c.rr.n) : 0u;
    return R;
}

int hypo_gpu_support_kmers(uint32_t k, uint64_t n_solid, const uint32_t* spos, const uint64_t* kids, uint32_t* coverage, uint32_t* support) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (!g_ctx.rr.ready) return fail(HYPO_E_INVALID, "no resident reads: call hypo_gpu_reads_upload first");
    if (k < 2 || k > 31) return fail(HYPO_E_INVALID, "k=%u out of range 2..31", k);
    if (n_solid >= 0xfffffff0ull) return fail(HYPO_E_CAPACITY, "%llu solid k-mers exceed the 32-bit counters of the boundary", (unsigned long long)n_solid);
    if (n_solid && (!spos || !kids || !coverage || !support)) return fail(HYPO_E_INVALID, "NULL buffer");
    if (!n_solid) return HYPO_OK;
    for (uint64_t i = 1; i < n_solid; ++i) if (spos[i - 1] >= spos[i]) return fail(HYPO_E_INVALID, "solid positions are not increasing (entry %llu)", (unsigned long long)i);
    Carver c;
    const size_t o_sp = c.take(n_solid * 4), o_kd = c.take(n_solid * 8), o_cov = c.take(n_solid * 4), o_sup = c.take(n_solid * 4);
    auto& wk = g_ctx.rr.work;
    HIP_TRY(wk.alloc(c.at));
    char* d = (char*)wk.p;
    hipStream_t st = g_ctx.stream;
    HIP_TRY(h2d(d + o_sp, spos, n_solid * 4, st));
    HIP_TRY(h2d(d + o_kd, kids, n_solid * 8, st));
    HIP_TRY(hipMemsetAsync(d + o_cov, 0, c.at - o_cov, st));
    HIP_TRY(hypo::support_kmers(support_reads_of(g_ctx), k, (uint32_t)n_solid, (const uint32_t*)(d + o_sp), (const uint64_t*)(d + o_kd), (uint32_t*)(d + o_cov), (uint32_t*)(d + o_sup), st));
    HIP_TRY(d2h(coverage, d + o_cov, n_solid * 4, st));
    HIP_TRY(d2h(support, d + o_sup, n_solid * 4, st));
    HIP_TRY(hipStreamSynchronize(st));
    return HYPO_OK;
}

int hypo_gpu_support_kmers_kept(uint32_t k, uint32_t n_contigs, const uint32_t* handles, const uint32_t* contig_base,
                                uint32_t* coverage, uint32_t* support, uint64_t* n_solid_total) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (!g_ctx.rr.ready) return fail(HYPO_E_INVALID, "no resident reads: call hypo_gpu_reads_upload first");
    if (k < 2 || k > 31) return fail(HYPO_E_INVALID, "k=%u out of range 2..31", k);
    if (n_contigs && (!handles || !contig_base)) return fail(HYPO_E_INVALID, "NULL argument");
    uint64_t ns = 0;
    for (uint32_t c = 0; c < n_contigs; ++c) {
        if (handles[c] >= g_ctx.kept.size() || !g_ctx.kept[handles[c]].used) return fail(HYPO_E_INVALID, "contig %u: handle %u holds no scan on this context", c, handles[c]);
        const Ctx::KeptScan& ks = g_ctx.kept[handles[c]];
        if (ks.k != k) return fail(HYPO_E_INVALID, "contig %u was scanned with k = %u", c, ks.k);
        if ((uint64_t)contig_base[c] + ks.n_bases > g_ctx.rr.total_len) return fail(HYPO_E_INVALID, "contig %u ends behind the %llu bases of the resident reads", c, (unsigned long long)g_ctx.rr.total_len);
        if (c && (uint64_t)contig_base[c] < (uint64_t)contig_base[c - 1] + g_ctx.kept[handles[c - 1]].n_bases) return fail(HYPO_E_INVALID, "contig %u overlaps the one before it", c);
        ns += ks.n;
    }
    if (n_solid_total) *n_solid_total = ns;
    if (ns >= 0xfffffff0ull) return fail(HYPO_E_CAPACITY, "%llu solid k-mers exceed the 32-bit counters of the boundary", (unsigned long long)ns);
    if (!ns) return HYPO_OK;
    if (!coverage || !support) return fail(HYPO_E_INVALID, "NULL buffer");
    const bool narrow = k <= 16;
    const size_t kid_bytes = narrow ? 4 : 8;
    Carver c;
    const size_t o_sp = c.take(ns * 4), o_kd = c.take(ns * kid_bytes), o_cov = c.take(ns * 4), o_sup = c.take(ns * 4);
    auto& wk = g_ctx.rr.work;
    HIP_TRY(wk.alloc(c.at));
    char* d = (char*)wk.p;
    hipStream_t st = g_ctx.stream;
    uint64_t at = 0;
    for (uint32_t ci = 0; ci < n_contigs; ++ci) {
        const Ctx::KeptScan& ks = g_ctx.kept[handles[ci]];
        if (!ks.n) continue;
        HIP_TRY(hypo::add_base(ks.spos, (uint32_t*)(d + o_sp) + at, ks.n, contig_base[ci], st));
        HIP_TRY(hipMemcpyAsync(d + o_kd + at * kid_bytes, ks.kids, ks.n * kid_bytes, hipMemcpyDeviceToDevice, st));
        at += ks.n;
    }
    HIP_TRY(hipMemsetAsync(d + o_cov, 0, c.at - o_cov, st));
    if (narrow) HIP_TRY(hypo::support_kmers32(support_reads_of(g_ctx), k, (uint32_t)ns, (const uint32_t*)(d + o_sp), (const uint32_t*)(d + o_kd), (uint32_t*)(d + o_cov), (uint32_t*)(d + o_sup), st));
    else HIP_TRY(hypo::support_kmers(support_reads_of(g_ctx), k, (uint32_t)ns, (const uint32_t*)(d + o_sp), (const uint64_t*)(d + o_kd), (uint32_t*)(d + o_cov), (uint32_t*)(d + o_sup), st));
    HIP_TRY(d2h(coverage, d + o_cov, ns * 4, st));
    HIP_TRY(d2h(support, d + o_sup, ns * 4, st));
    HIP_TRY(hipStreamSynchronize(st));
    return HYPO_OK;
}

int hypo_gpu_support_minimizers(const HypoMegaWindows* W, uint32_t* coverage, uint32_t* support) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (!g_ctx.rr.ready) return fail(HYPO_E_INVALID, "no resident reads: call hypo_gpu_reads_upload first");
    if (!W || !W->n_contigs || !W->contig_base || !W->reg_base || !W->win_even || !W->info_base || !W->start || !W->mw_off) return fail(HYPO_E_INVALID, "NULL argument");
    const uint32_t nc = W->n_contigs;
    const uint64_t n_start = W->reg_base[nc], n_ent = W->mw_off[W->n_info];
    if (n_ent && (!W->rel_pos || !W->minimisers || !coverage || !support)) return fail(HYPO_E_INVALID, "NULL buffer");
    if (!n_ent) return HYPO_OK;
    for (uint32_t c = 0; c < nc; ++c) {
        if (W->reg_base[c] > W->reg_base[c + 1]) return fail(HYPO_E_INVALID, "contig %u: reg_base decreases", c);
        for (uint64_t i = W->reg_base[c] + 1; i < W->reg_base[c + 1]; ++i) if (W->start[i - 1] >= W->start[i]) return fail(HYPO_E_INVALID, "contig %u: region starts are not increasing", c);
    }
    for (uint32_t x = 0; x < W->n_info; ++x) if (W->mw_off[x] > W->mw_off[x + 1]) return fail(HYPO_E_INVALID, "mw_off decreases at %u", x);
    // the tables against the resident reads (another caller than DeviceArms may pass tables of other contigs): every read's contig
    // exists, its span lies inside that contig, and the MWMinimiserInfo entries its mega-windows index exist (a read that ends in
    // the last region looks at entry info_base + (borders - 1) / 2, one past the contig's last: mw_off is padded by one on the device)
    {
        const auto& rr = g_ctx.rr;
        if (rr.ctg_min_rb.size() > nc) return fail(HYPO_E_INVALID, "the resident reads name contig %zu, the tables hold %u contigs", rr.ctg_min_rb.size() - 1, nc);
        for (uint32_t c = 0; c < nc; ++c) {
            const uint32_t nb = W->reg_base[c + 1] - W->reg_base[c];
            const bool has_reads = c < rr.ctg_min_rb.size() && rr.ctg_max_re[c] > 0;
            if (!has_reads) continue;
            if (nb < 2) return fail(HYPO_E_INVALID, "contig %u has reads but fewer than two region borders", c);
            const uint64_t len = W->start[W->reg_base[c + 1] - 1];
            if (rr.ctg_min_rb[c] < W->contig_base[c] || (uint64_t)rr.ctg_max_re[c] > (uint64_t)W->contig_base[c] + len)
                return fail(HYPO_E_INVALID, "contig %u: its resident reads span [%u, %u), the tables place it at [%u, %llu)", c, rr.ctg_min_rb[c], rr.ctg_max_re[c],
                            W->contig_base[c], (unsigned long long)((uint64_t)W->contig_base[c] + len));
            if ((uint64_t)W->info_base[c] + (nb - 1) / 2 > (uint64_t)W->n_info)
                return fail(HYPO_E_INVALID, "contig %u: %u region borders need MWMinimiserInfo entries up to %llu, the tables hold %u", c, nb,
                            (unsigned long long)((uint64_t)W->info_base[c] + (nb - 1) / 2), W->n_info);
        }
    }
    Carver c;
    const size_t o_cb = c.take((size_t)nc * 4), o_rbase = c.take((size_t)(nc + 1) * 4), o_even = c.take(nc), o_ib = c.take((size_t)nc * 4), o_start = c.take(n_start * 4),
                 o_off = c.take((size_t)(W->n_info + 2) * 4), o_rel = c.take(n_ent * 4), o_min = c.take(n_ent * 4), o_cov = c.take(n_ent * 4), o_sup = c.take(n_ent * 4);
    auto& wk = g_ctx.rr.work;
    HIP_TRY(wk.alloc(c.at));
    char* d = (char*)wk.p;
    hipStream_t st = g_ctx.stream;
#define UP(off, src, bytes) do { if (bytes) HIP_TRY(h2d(d + (off), (src), (bytes), st)); } while (0)
    UP(o_cb, W->contig_base, (size_t)nc * 4); UP(o_rbase, W->reg_base, (size_t)(nc + 1) * 4); UP(o_even, W->win_even, nc); UP(o_ib, W->info_base, (size_t)nc * 4);
    UP(o_start, W->start, n_start * 4); UP(o_off, W->mw_off, (size_t)(W->n_info + 1) * 4); UP(o_rel, W->rel_pos, n_ent * 4); UP(o_min, W->minimisers, n_ent * 4);
    UP(o_off + (size_t)(W->n_info + 1) * 4, W->mw_off + W->n_info, 4);          // the pad entry: an empty range behind the last info
#undef UP
    HIP_TRY(hipMemsetAsync(d + o_cov, 0, c.at - o_cov, st));
    hypo::MegaWindows M;
    M.contig_base = (const uint32_t*)(d + o_cb); M.reg_base = (const uint32_t*)(d + o_rbase); M.win_even = (const uint8_t*)(d + o_even); M.info_base = (const uint32_t*)(d + o_ib);
    M.start = (const uint32_t*)(d + o_start); M.mw_off = (const uint32_t*)(d + o_off); M.rel_pos = (uint32_t*)(d + o_rel); M.minimisers = (const uint32_t*)(d + o_min);
    HIP_TRY(hypo::support_minimizers(support_reads_of(g_ctx), M, nc, W->n_info, (uint32_t*)(d + o_cov), (uint32_t*)(d + o_sup), st));
    HIP_TRY(d2h(coverage, d + o_cov, n_ent * 4, st));
    HIP_TRY(d2h(support, d + o_sup, n_ent * 4, st));
    HIP_TRY(hipStreamSynchronize(st));
    return HYPO_OK;
}

// ---- arm selection on the device (kernels in arms_kernel.hip) ------------------------------------------

// which = 0: short reads -> SHORT windows; 1: long reads over the pseudo regions of Contig::prepare_long_windows -> LONG windows
static int arms_build_impl(int which, const HypoArmsRegions* R, const HypoArmsReads* A, uint8_t* region_valid, HypoArmsSummary* sum) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    auto& AS = g_ctx.arms[which];
    const bool long_mode = which == 1;
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    if (!R || !region_valid || !sum) return fail(HYPO_E_INVALID, "NULL argument");
    // reads == NULL: the reads hypo_gpu_reads_upload left on the device (already checked there)
    const bool resident = A == nullptr;
    if (resident && (long_mode || !g_ctx.rr.ready)) return fail(HYPO_E_INVALID, "reads == NULL but no resident reads (hypo_gpu_reads_upload)");
    HypoArmsReads Ares{};
    if (resident) { Ares.n_alignments = g_ctx.rr.n; Ares.reads2_bytes = g_ctx.rr.reads2_bytes; A = &Ares; }
    if (!R->n_regions || !R->start || !R->type || (!long_mode && !R->info) || !R->contig4 || (R->n_anchor_kmers && !R->anchor_kmers)) return fail(HYPO_E_INVALID, "NULL buffer in regions");
    if (!resident && A->n_alignments && (!A->rb || !A->re || !A->qae || !A->seq_off || !A->reads2 || !A->cigar_off || !A->cigar)) return fail(HYPO_E_INVALID, "NULL buffer in reads");
    if (R->k < 2 || R->k > 31) return fail(HYPO_E_INVALID, "k=%u out of range 2..31", R->k);
    AS.ready = false;
    const uint32_t nr = R->n_regions, na = A->n_alignments;
    const uint64_t total_len = R->start[nr];
    uint32_t max_span = 0;
    uint64_t sum_span = 0;
    for (uint32_t i = 0; i < nr; ++i) if (R->start[i] >= R->start[i + 1]) return fail(HYPO_E_INVALID, "region %u is empty or the starts are not increasing", i);
    if (resident) {
        if (total_len != g_ctx.rr.total_len)
            return fail(HYPO_E_INVALID, "the regions cover %llu bases, the resident reads were checked against %llu (hypo_gpu_reads_upload)", (unsigned long long)total_len, (unsigned long long)g_ctx.rr.total_len);
        max_span = g_ctx.rr.max_span; sum_span = g_ctx.rr.sum_span;
    }
    for (uint32_t a = 0; a < (resident ? 0u : na); ++a) {
        if (A->re[a] <= A->rb[a] || A->re[a] > total_len) return fail(HYPO_E_INVALID, "alignment %u: span [%u, %u) outside the %llu bases", a, A->rb[a], A->re[a], (unsigned long long)total_len);
        if (a && A->rb[a - 1] > A->rb[a]) return fail(HYPO_E_INVALID, "alignments are not sorted by reference start (alignment %u)", a);
        if (A->seq_off[a] + ((uint64_t)A->qae[a] + 3) / 4 > A->reads2_bytes) return fail(HYPO_E_INVALID, "alignment %u: read outside reads2", a);
        if (A->cigar_off[a] > A->cigar_off[a + 1]) return fail(HYPO_E_INVALID, "alignment %u: cigar_off decreases", a);
        const uint32_t span = A->re[a] - A->rb[a];
        max_span = span > max_span ? span : max_span;
        sum_span += span;
    }
    // arms_window_kernel looks, per window, at every alignment that starts within max_span bases before it: one record with a
    // huge reference span (a 100 kb D / N operation) would make every window walk a 100 kb neighbourhood.  Such input takes the
    // host loops (the caller falls back on any error of this call).
    if (na && max_span > 16384u && (uint64_t)max_span * na > 64ull * sum_span)
        return fail(HYPO_E_CAPACITY, "an alignment spans %u reference bases, more than 64 x the mean span (%llu): arm selection stays on the host",
                    max_span, (unsigned long long)(sum_span / na));
    const uint64_t n_cig = resident ? g_ctx.rr.n_cig : (na ? A->cigar_off[na] : 0);
    hipStream_t st = g_ctx.stream;
    DevBuf &dIn = AS.arena[0], &dWork = AS.arena[1], &dBatch = AS.arena[2];
    // inputs
    Carver ci;
    const size_t o_start = ci.take((size_t)(nr + 1) * 4), o_type = ci.take(nr + 1), o_info = ci.take((size_t)(nr + 1) * 4),
                 o_anchor = ci.take(R->n_anchor_kmers * 8), o_contig = ci.take((total_len + 1) / 2), o_rb = ci.take(resident ? 0 : (size_t)na * 4), o_re = ci.take(resident ? 0 : (size_t)na * 4),
                 o_qae = ci.take(resident ? 0 : (size_t)na * 4), o_soff = ci.take(resident ? 0 : (size_t)na * 8), o_reads = ci.take(resident ? 0 : A->reads2_bytes), o_coff = ci.take(resident ? 0 : (size_t)(na + 1) * 4),
                 o_cig = ci.take(resident ? 0 : n_cig * 4), o_frank = ci.take((resident || !A->file_rank) ? 0 : (size_t)na * 4);
    HIP_TRY(dIn.alloc(ci.at));
    char* in = (char*)dIn.p;
#define UP(off, src, bytes) do { if (bytes) HIP_TRY(h2d(in + (off), (src), (bytes), st)); } while (0)
    UP(o_start, R->start, (size_t)(nr + 1) * 4); UP(o_type, R->type, (size_t)nr + 1); if (R->info) UP(o_info, R->info, (size_t)(nr + 1) * 4);
    UP(o_anchor, R->anchor_kmers, R->n_anchor_kmers * 8); UP(o_contig, R->contig4, (total_len + 1) / 2);
    if (!resident) {
        UP(o_rb, A->rb, (size_t)na * 4); UP(o_re, A->re, (size_t)na * 4); UP(o_qae, A->qae, (size_t)na * 4); UP(o_soff, A->seq_off, (size_t)na * 8);
        UP(o_reads, A->reads2, A->reads2_bytes); if (na) UP(o_coff, A->cigar_off, (size_t)(na + 1) * 4); UP(o_cig, A->cigar, n_cig * 4);
        if (A->file_rank) UP(o_frank, A->file_rank, (size_t)na * 4);
    }
#undef UP
    hypo::ArmsIn I;
    I.n_regions = nr; I.reg_start = (const uint32_t*)(in + o_start); I.reg_type = (const uint8_t*)(in + o_type); I.reg_info = (const uint32_t*)(in + o_info);
    I.anchor_kmers = (const uint64_t*)(in + o_anchor); I.k = R->k; I.contig4 = (const uint8_t*)(in + o_contig);
    I.n_alignments = na; I.rb = (const uint32_t*)(in + o_rb); I.re = (const uint32_t*)(in + o_re); I.qae = (const uint32_t*)(in + o_qae);
    I.seq_off = (const uint64_t*)(in + o_soff); I.reads2 = (const uint8_t*)(in + o_reads); I.cigar_off = (const uint32_t*)(in + o_coff);
    I.cigar = (const uint32_t*)(in + o_cig); I.max_span = max_span; I.long_mode = long_mode ? 1u : 0u;
    I.file_rank = (!resident && A->file_rank) ? (const uint32_t*)(in + o_frank) : nullptr;
    if (resident) {
        const auto& rr = g_ctx.rr;
        I.file_rank = rr.file_rank;
        I.rb = rr.rb; I.re = rr.re; I.qae = rr.qae; I.seq_off = rr.seq_off; I.reads2 = rr.reads2; I.cigar_off = rr.cigar_off; I.cigar = rr.cigar;
    }
    // work arrays that do not depend on the number of touched regions
    Carver cw;
    const size_t scan_n = nr > na ? nr : na;
    const size_t w_bind = cw.take((size_t)na * 4), w_nt = cw.take((size_t)na * 4), w_toff = cw.take((size_t)na * 8), w_bsum = cw.take(hypo::scan32_scratch_bytes(scan_n)),
                 w_tot = cw.take(64), w_flags = cw.take((size_t)nr * 4), w_valid = cw.take((size_t)nr * 4), w_counts = cw.take((size_t)nr * 16),
                 w_arms = cw.take((size_t)nr * 4), w_bytes = cw.take((size_t)nr * 4), w_bint = cw.take((size_t)nr * 4), w_bpre = cw.take((size_t)nr * 4),
                 w_dbytes = cw.take((size_t)nr * 4), w_slot = cw.take((size_t)nr * 4), w_winoff = cw.take((size_t)nr * 8), w_armoff = cw.take((size_t)nr * 8),
                 w_byteoff = cw.take((size_t)nr * 8), w_droff = cw.take((size_t)nr * 8), w_slotoff = cw.take((size_t)nr * 8), w_widx = cw.take((size_t)nr * 4),
                 w_minlen = cw.take(long_mode ? (size_t)nr * 4 : 0), w_minoff = cw.take(long_mode ? (size_t)nr * 8 : 0), w_mincnt = cw.take(long_mode ? (size_t)nr * 4 : 0);
    const size_t fixed_work = cw.at;
    // the candidate arrays follow: a read of span s touches at most s / (shortest region) + 2 regions; sized after phase 1
    HIP_TRY(dWork.alloc(fixed_work));
    char* wk = (char*)dWork.p;
    uint64_t* tot = (uint64_t*)(wk + w_tot);             // [0] touched regions, [1] windows, [2] arms, [3] arm bytes, [4] draft bytes, [5] slot bytes, [6] bad records
    HIP_TRY(hipMemsetAsync(tot, 0, 64, st));
    HIP_TRY(hypo::arms_phase1(I, (uint32_t*)(wk + w_bind), (uint32_t*)(wk + w_nt), (uint32_t*)(tot + 6), st));
    HIP_TRY(hypo::scan32((const uint32_t*)(wk + w_nt), na, (uint64_t*)(wk + w_toff), (uint64_t*)(wk + w_bsum), tot + 0, st));
    hypo::ArmsOut O{};
    if (long_mode) {      // slots for the minimizers of every LONG window's draft (Filter::initialise): one per base at most
        O.reg_min_len = (uint32_t*)(wk + w_minlen); O.reg_min_off = (const uint64_t*)(wk + w_minoff); O.reg_min_cnt = (uint32_t*)(wk + w_mincnt);
        HIP_TRY(hypo::arms_long_minlen(I, O, st));
        HIP_TRY(hypo::scan32(O.reg_min_len, nr, (uint64_t*)(wk + w_minoff), (uint64_t*)(wk + w_bsum), tot + 7, st));
    }
    uint64_t h_tot[8] = {0};
    HIP_TRY(hipMemcpyAsync(h_tot, tot, 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_tot[6]) return fail(HYPO_E_INVALID, "%llu alignment record(s) whose CIGAR disagrees with their span / aligned length", (unsigned long long)h_tot[6]);
    const uint64_t n_touch = h_tot[0];
    // candidates (in a buffer of their own: growing dWork would move the arrays phase 1 filled)
    DevBuf& dCand = AS.arena[3];
    Carver cc;
    const size_t c_bp = cc.take(n_touch * 4), c_cand = cc.take(n_touch * 8), c_dmin = cc.take(long_mode ? h_tot[7] * 4 + 16 : 0);
    HIP_TRY(dCand.alloc(cc.at));
    char* cd = (char*)dCand.p;
    if (long_mode) { O.draft_min = (uint32_t*)(cd + c_dmin); HIP_TRY(hypo::arms_long_draftmin(I, O, st)); }
    O.reg_flags = (uint32_t*)(wk + w_flags); O.reg_valid = (uint32_t*)(wk + w_valid); O.reg_counts = (uint4*)(wk + w_counts); O.reg_arms = (uint32_t*)(wk + w_arms);
    O.reg_bytes = (uint32_t*)(wk + w_bytes); O.reg_bytes_int = (uint32_t*)(wk + w_bint); O.reg_bytes_pre = (uint32_t*)(wk + w_bpre);
    O.reg_draft_bytes = (uint32_t*)(wk + w_dbytes); O.reg_slot = (uint32_t*)(wk + w_slot);
    O.reg_arm_off = (const uint64_t*)(wk + w_armoff); O.reg_byte_off = (const uint64_t*)(wk + w_byteoff); O.reg_draft_off = (const uint64_t*)(wk + w_droff);
    O.reg_slot_off = (const uint64_t*)(wk + w_slotoff); O.win_index = (uint32_t*)(wk + w_widx);
    HIP_TRY(hypo::arms_phase2(I, (const uint32_t*)(wk + w_bind), (const uint32_t*)(wk + w_nt), (const uint64_t*)(wk + w_toff), (uint32_t*)(cd + c_bp), (uint2*)(cd + c_cand), O, st));
    uint64_t* bsum = (uint64_t*)(wk + w_bsum);
    HIP_TRY(hypo::scan32(O.reg_valid, nr, (uint64_t*)(wk + w_winoff), bsum, tot + 1, st));
    HIP_TRY(hypo::scan32(O.reg_arms, nr, (uint64_t*)(wk + w_armoff), bsum, tot + 2, st));
    HIP_TRY(hypo::scan32(O.reg_bytes, nr, (uint64_t*)(wk + w_byteoff), bsum, tot + 3, st));
    HIP_TRY(hypo::scan32(O.reg_draft_bytes, nr, (uint64_t*)(wk + w_droff), bsum, tot + 4, st));
    HIP_TRY(hypo::scan32(O.reg_slot, nr, (uint64_t*)(wk + w_slotoff), bsum, tot + 5, st));
    HIP_TRY(hipMemcpyAsync(h_tot, tot, 64, hipMemcpyDeviceToHost, st));
    std::vector<uint32_t> valid32(nr);
    HIP_TRY(hipMemcpyAsync(valid32.data(), O.reg_valid, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < nr; ++i) region_valid[i] = (uint8_t)valid32[i];
    const uint64_t n_win = h_tot[1], n_arms = h_tot[2], arm_bytes = h_tot[3], draft_bytes = h_tot[4], slot_bytes = h_tot[5];
    if (n_arms > 0xfff00000ull || n_win > 0x7fffffffull)
        return fail(HYPO_E_CAPACITY, "%llu windows / %llu arms exceed the 32-bit counters of the boundary", (unsigned long long)n_win, (unsigned long long)n_arms);
    // the batch
    Carver cb;
    const size_t b_win = cb.take(n_win * sizeof(HypoWindow)), b_wreg = cb.take(n_win * 4), b_alen = cb.take(n_arms * 4), b_aoff = cb.take(n_arms * 8),
                 b_arms2 = cb.take(arm_bytes + 16), b_draft = cb.take(draft_bytes + 16), b_ooff = cb.take((n_win + 1) * 8);
    HIP_TRY(dBatch.alloc(cb.at));
    char* bt = (char*)dBatch.p;
    O.windows = (HypoWindow*)(bt + b_win); O.win_region = (uint32_t*)(bt + b_wreg); O.arm_len = (uint32_t*)(bt + b_alen); O.arm_off = (uint64_t*)(bt + b_aoff);
    O.arms2 = (uint8_t*)(bt + b_arms2); O.draft4 = (uint8_t*)(bt + b_draft); O.out_off = (uint64_t*)(bt + b_ooff);
    if (n_win) {
        HIP_TRY(hypo::arms_phase3(I, (const uint32_t*)(wk + w_bind), (const uint32_t*)(wk + w_nt), (const uint64_t*)(wk + w_toff), (const uint2*)(cd + c_cand), O,
                                  (const uint64_t*)(wk + w_winoff), st));
        HIP_TRY(hipMemcpyAsync(O.out_off + n_win, tot + 5, 8, hipMemcpyDeviceToDevice, st));
    }
    sum->n_windows = (uint32_t)n_win; sum->n_arms = (uint32_t)n_arms; sum->arms2_bytes = arm_bytes; sum->draft4_bytes = draft_bytes; sum->out_bytes = slot_bytes;
    AS.sum = *sum; AS.out = O; AS.ready = true;
    return HYPO_OK;
}
int hypo_gpu_arms_build(const HypoArmsRegions* R, const HypoArmsReads* A, uint8_t* region_valid, HypoArmsSummary* sum) { return arms_build_impl(0, R, A, region_valid, sum); }
int hypo_gpu_arms_build_long(const HypoArmsRegions* R, const HypoArmsReads* A, uint8_t* region_valid, HypoArmsSummary* sum) { return arms_build_impl(1, R, A, region_valid, sum); }

static int arms_download_impl(int which, HypoWindow* windows, uint32_t* win_region, uint32_t* arm_len, uint64_t* arm_off, uint8_t* arms2, uint8_t* draft4) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    auto& AS = g_ctx.arms[which];
    if (!AS.ready) return fail(HYPO_E_INVALID, "no resident batch: call hypo_gpu_arms_build%s first", which ? "_long" : "");
    const HypoArmsSummary& S = AS.sum; const hypo::ArmsOut& O = AS.out;
    hipStream_t st = g_ctx.stream;
    if (windows && S.n_windows) HIP_TRY(hipMemcpyAsync(windows, O.windows, (size_t)S.n_windows * sizeof(HypoWindow), hipMemcpyDeviceToHost, st));
    if (win_region && S.n_windows) HIP_TRY(hipMemcpyAsync(win_region, O.win_region, (size_t)S.n_windows * 4, hipMemcpyDeviceToHost, st));
    if (arm_len && S.n_arms) HIP_TRY(hipMemcpyAsync(arm_len, O.arm_len, (size_t)S.n_arms * 4, hipMemcpyDeviceToHost, st));
    if (arm_off && S.n_arms) HIP_TRY(hipMemcpyAsync(arm_off, O.arm_off, (size_t)S.n_arms * 8, hipMemcpyDeviceToHost, st));
    if (arms2 && S.arms2_bytes) HIP_TRY(d2h(arms2, O.arms2, S.arms2_bytes, st));
    if (draft4 && S.draft4_bytes) HIP_TRY(hipMemcpyAsync(draft4, O.draft4, S.draft4_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return HYPO_OK;
}

int hypo_gpu_arms_download(HypoWindow* windows, uint32_t* win_region, uint32_t* arm_len, uint64_t* arm_off, uint8_t* arms2, uint8_t* draft4) {
    return arms_download_impl(0, windows, win_region, arm_len, arm_off, arms2, draft4);
}
int hypo_gpu_arms_download_long(HypoWindow* windows, uint32_t* win_region, uint32_t* arm_len, uint64_t* arm_off, uint8_t* arms2, uint8_t* draft4) {
    return arms_download_impl(1, windows, win_region, arm_len, arm_off, arms2, draft4);
}

static int arms_poa_impl(int which, const HypoScoreParams* scores, char* bases, uint64_t* off, uint32_t* len, uint8_t* status) {
    HYPO_LOCKED();
    HYPO_ON_DEVICE();
    if (!g_ctx.ready) return fail(HYPO_E_NOTINIT, "hypo_gpu_init was not called");
    int rc = check_scores(scores);
    if (rc) return rc;
    auto& AS = g_ctx.arms[which];
    if (!AS.ready) return fail(HYPO_E_INVALID, "no resident batch: call hypo_gpu_arms_build%s first", which ? "_long" : "");
    const HypoArmsSummary& S = AS.sum; const hypo::ArmsOut& O = AS.out;
    memset(&tl_stats, 0, sizeof(tl_stats));
    const uint32_t n = S.n_windows;
    if (!n) return HYPO_OK;
    if (!bases || !off || !len || !status) return fail(HYPO_E_INVALID, "NULL buffer");
    hipStream_t st = g_ctx.stream;
    DevBuf& dOut = AS.arena[4];
    Carver co;
    // SHORT windows only: the HBM-scratch classes see the odd escalated window, their smallest scratch (16 resident groups) will do;
    // LONG windows: one resident group per window up to what the device holds
    const size_t wsb = hypo::poa_workspace_bytes(n, which ? (int)(n + 64 < 2048u ? n + 64 : 2048u) : hypo::kMinGlobalGroups, 0);
    const size_t o_bases = co.take(S.out_bytes + 16), o_len = co.take((size_t)n * 4), o_st = co.take(n), o_ws = co.take(wsb);
    const bool timing = getenv("HYPO_HOST_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(dOut.alloc(co.at));
    const auto t1 = std::chrono::steady_clock::now();
    char* ob = (char*)dOut.p;
    HIP_TRY(hipMemsetAsync(ob + o_bases, 0, S.out_bytes + 16, st));
    HypoWindowBatch din;
    din.n_windows = n; din.n_arms = S.n_arms; din.windows = O.windows; din.draft4 = O.draft4; din.draft4_bytes = S.draft4_bytes;
    din.arm_off = O.arm_off; din.arm_len = O.arm_len; din.arms2 = O.arms2; din.arms2_bytes = S.arms2_bytes;
    HypoConsensusBatch dout;
    dout.bases = ob + o_bases; dout.off = O.out_off; dout.len = (uint32_t*)(ob + o_len); dout.status = (uint8_t*)(ob + o_st);
    hypo::PoaParams P = make_params(scores, &din, &dout);
    g_ctx.slots[0].aux.next_kind = which;                      // (a LONG batch says nothing about the SHORT one behind it, and the other way round)
    const hipError_t pr = hypo::poa_run(P, n, ob + o_ws, wsb, g_ctx.num_cus, st, nullptr, &g_ctx.slots[0].aux);
    g_ctx.slots[0].aux.next_kind = 0;
    HIP_TRY(pr);
    if (timing) HIP_TRY(hipStreamSynchronize(st));
    const auto t2 = std::chrono::steady_clock::now();
    HIP_TRY(d2h(bases, ob + o_bases, S.out_bytes, st));
    HIP_TRY(hipMemcpyAsync(off, O.out_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(len, ob + o_len, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(status, ob + o_st, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&tl_stats, ob + o_ws + 128, sizeof(HypoPoaStats), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (timing) {
        auto d = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count() * 1e3; };
        fprintf(stderr, "[timing] hypo_gpu_arms_poa: device buffers (%zu MB) %.2f ms, kernels %.2f ms, results to the host %.2f ms\n", co.at >> 20, d(t0, t1), d(t1, t2), d(t2, std::chrono::steady_clock::now()));
    }
    tl_stats.n_windows = n;
    return HYPO_OK;
}
int hypo_gpu_arms_poa(const HypoScoreParams* scores, char* bases, uint64_t* off, uint32_t* len, uint8_t* status) { return arms_poa_impl(0, scores, bases, off, len, status); }
int hypo_gpu_arms_poa_long(const HypoScoreParams* scores, char* bases, uint64_t* off, uint32_t* len, uint8_t* status) { return arms_poa_impl(1, scores, bases, off, len, status); }

}  // extern "C"
