// emu_poa.cpp — TEST-ONLY lockstep emulator for hypo_amd/csrc/poa_core.hpp.
//
// Compiles the kernel's per-window code with HYPO_EMU: every lane of a group is a fiber with its own
// stack, scheduled round-robin; Grp::sync() yields to the next lane, so each collective is a
// rendezvous of all GW lanes (the kernel's control flow is group-uniform by construction — the
// emulator aborts if a lane tries to rendezvous with a lane that already returned).
// The group's memory slice is one heap block of exactly PoaLayout::BYTES, so AddressSanitizer sees
// out-of-slice accesses.  Built by tests/emu/Makefile into tests/_build/libhypo_emu[_asan].so and driven
// by tests/test_poa_emulator.py against the oracle.  Not part of the product.
#define HYPO_EMU 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../hypo_amd/csrc/poa_core.hpp"
#include "../../hypo_amd/csrc/poa_giant.hpp"

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__)
#define EMU_ASAN 1
#endif
#ifdef EMU_ASAN
extern "C" void __asan_unpoison_memory_region(void const volatile*, size_t);
#endif

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;

struct Sched {
    int gw = 0, cur = 0;
    void* sp[64] = {};
    void* main_sp = nullptr;
    bool done[64] = {};
    char* stacks = nullptr;
    void (*body)(int lane, void* arg) = nullptr;
    void* arg = nullptr;
};
thread_local Sched* tl_sched = nullptr;

void yield_cb(void* s_) {
    Sched* s = (Sched*)s_;
    int cur = s->cur, nxt = (cur + 1) % s->gw;
    static thread_local uint64_t n_yield[64];
    n_yield[cur]++;
    if (s->done[nxt]) { fprintf(stderr, "[emu] non-uniform control flow: lane %d waits for finished lane %d (yields so far: lane 0 %llu, lane 1 %llu, lane 62 %llu, lane 63 %llu)\n", cur, nxt,
                                (unsigned long long)n_yield[0], (unsigned long long)n_yield[1], (unsigned long long)n_yield[62], (unsigned long long)n_yield[63]); abort(); }
    s->cur = nxt;
    emu_switch(&s->sp[cur], s->sp[nxt]);
}

void fiber_entry() {
    Sched* s = tl_sched;
    int lane = s->cur;
    s->body(lane, s->arg);
    s->done[lane] = true;
    int nxt = lane + 1;
    void* dummy;
    if (nxt == s->gw) { emu_switch(&dummy, s->main_sp); }
    else {
        if (s->done[nxt]) { fprintf(stderr, "[emu] lane order violated\n"); abort(); }
        s->cur = nxt; emu_switch(&dummy, s->sp[nxt]);
    }
    abort();   // a finished fiber is never resumed
}

void run_group(Sched& s, int gw, void (*body)(int, void*), void* arg) {
    s.gw = gw; s.body = body; s.arg = arg;
    if (!s.stacks) s.stacks = (char*)aligned_alloc(64, STACK_BYTES * 64);
#ifdef EMU_ASAN
    __asan_unpoison_memory_region(s.stacks, STACK_BYTES * 64);
#endif
    for (int l = 0; l < gw; ++l) {
        s.done[l] = false;
        uintptr_t top = ((uintptr_t)(s.stacks + STACK_BYTES * (l + 1))) & ~(uintptr_t)15;
        void** p = (void**)top;
        *--p = nullptr;                 // fake return address of fiber_entry's caller
        *--p = (void*)&fiber_entry;     // `ret` target
        for (int k = 0; k < 6; ++k) *--p = nullptr;   // rbp rbx r12 r13 r14 r15
        s.sp[l] = (void*)p;
    }
    tl_sched = &s;
    s.cur = 0;
    emu_switch(&s.main_sp, s.sp[0]);
    for (int l = 0; l < gw; ++l) if (!s.done[l]) { fprintf(stderr, "[emu] lane %d did not finish\n", l); abort(); }
}

uint64_t g_exact[4] = {0, 0, 0, 0};   // exact-threading attempts / successes / reused alignments / successes along the guide since the last emu_exact_stats

// what a re-queued window takes along (Poa::spill), as the kernel's queue entry would
struct Carry {
    std::vector<uint8_t> blob;      // empty = the window starts from its first sequence
    int need_nodes = 0;
};

template <class Cfg>
struct Job {
    hypo::EmuGroup eg;
    const hypo::PoaParams* P;
    char* mem;
    char* fast;           // second slice of a hybrid class (PoaLayout::FAST_BYTES)
    char* dirg;           // direction codes of a Cfg::DIRG class (PoaLayout::DIRG_BYTES)
    uint32_t w;
    const uint8_t* carry_in;
    uint8_t* carry_out;   // room for the largest spill of this class
    uint32_t carry_len;   // 0 = nothing spilled; ~0u = the incoming spill travels on unchanged
    int need_nodes;
    int rc[64];
    uint64_t cells, aligns;
};

template <class Cfg>
void lane_body(int lane, void* arg) {
    Job<Cfg>* j = (Job<Cfg>*)arg;
    hypo::Grp<Cfg::GW> g{lane, &j->eg};
    hypo::Poa<Cfg> poa(g, hypo::PoaParamRef{j->P}, j->mem, j->fast, j->dirg);
    const int rc = poa.run(j->w, j->carry_in);
    j->rc[lane] = rc;
    typedef hypo::Poa<Cfg> PoaT;
    if (rc == hypo::RES_OVERFLOW && poa.stat_get(PoaT::ST_CKIND) != PoaT::CARRY_NONE) {
        poa.spill(j->carry_out);                    // every lane writes its share
        if (lane == 0) j->carry_len = poa.spill_size();
    } else if (rc == hypo::RES_OVERFLOW && poa.stat_get(PoaT::ST_CPASS) && lane == 0) j->carry_len = ~0u;
    if (lane == 0) {
        j->need_nodes = (int)poa.stat_get(PoaT::ST_NEED); j->cells = poa.stat_get(PoaT::ST_CELLS); j->aligns = poa.stat_get(PoaT::ST_ALIGNS);
        g_exact[0] += poa.exact_tries; g_exact[1] += poa.stat_get(PoaT::ST_XHITS); g_exact[2] += poa.stat_get(PoaT::ST_REUSED); g_exact[3] += poa.guided_hits;
    }
}

// one window through one class; `carry` is consumed and, on RES_OVERFLOW, replaced by what the window takes to the next class
template <class Cfg>
int run_window(Sched& s, const hypo::PoaParams& P, uint32_t w, Carry* carry, uint64_t* cells, uint64_t* aligns) {
    Job<Cfg> job;
    job.P = &P;
    job.eg.gw = Cfg::GW; job.eg.yield = yield_cb; job.eg.sched = &s;
    job.mem = (char*)malloc(hypo::PoaLayout<Cfg>::BYTES);       // exact size: ASan sees overruns
    // LDS / scratch are not initialised on the device: HYPO_EMU_FILL selects the garbage (default 0xA5; tests also use 0x00 / 0xFF)
    const int fill = getenv("HYPO_EMU_FILL") ? (int)strtol(getenv("HYPO_EMU_FILL"), nullptr, 0) : 0xA5;
    memset(job.mem, fill, hypo::PoaLayout<Cfg>::BYTES);
    job.fast = (char*)malloc(hypo::PoaLayout<Cfg>::FAST_BYTES);
    memset(job.fast, fill ^ 0xFF, hypo::PoaLayout<Cfg>::FAST_BYTES);
    job.dirg = (char*)malloc(hypo::PoaLayout<Cfg>::DIRG_BYTES ? hypo::PoaLayout<Cfg>::DIRG_BYTES : 1);
    memset(job.dirg, fill ^ 0x3C, hypo::PoaLayout<Cfg>::DIRG_BYTES ? hypo::PoaLayout<Cfg>::DIRG_BYTES : 1);
    const uint32_t cap = hypo::Poa<Cfg>::spill_bytes(Cfg::NMAX, Cfg::KIN);
    std::vector<uint8_t> out(cap);
    job.carry_in = (carry && !carry->blob.empty()) ? carry->blob.data() : nullptr;
    job.carry_out = out.data(); job.carry_len = 0; job.need_nodes = 0;
    job.w = w; job.cells = job.aligns = 0;
    run_group(s, Cfg::GW, lane_body<Cfg>, &job);
    for (int l = 1; l < Cfg::GW; ++l) if (job.rc[l] != job.rc[0]) { fprintf(stderr, "[emu] lanes disagree on the result of window %u\n", w); abort(); }
    if (carry) {
        if (job.carry_len == ~0u) { /* keeps its blob */ }
        else if (job.carry_len) { if (job.carry_len > cap) abort(); carry->blob.assign(out.begin(), out.begin() + job.carry_len); }
        else carry->blob.clear();
        carry->need_nodes = job.need_nodes;
    }
    *cells += job.cells; *aligns += job.aligns;
    free(job.mem);
    free(job.fast);
    free(job.dirg);
    return job.rc[0];
}

template <class Cfg>
int run_cfg(const hypo::PoaParams& P, uint32_t n_windows, uint8_t* res, uint64_t* cells, uint64_t* aligns) {
    Sched s;
    for (uint32_t w = 0; w < n_windows; ++w) {
        const int rc = run_window<Cfg>(s, P, w, nullptr, cells, aligns);
        res[w] = (uint8_t)rc;
        if (rc != hypo::RES_OK) { P.out_len[w] = 0; P.out_status[w] = 0xFF; }
    }
    free(s.stacks);
    return 0;
}

}  // namespace

// configurations mirrored from hypo_amd/csrc/poa_kernel.hip (keep in sync: see poa_classes.hpp)
#include "../../hypo_amd/csrc/poa_classes.hpp"

extern "C" int emu_poa_batch(const HypoScoreParams* sp, const HypoWindowBatch* in, HypoConsensusBatch* out,
                             int cfg_id, uint8_t* res, uint64_t* cells, uint64_t* aligns) {
    hypo::PoaParams P;
    P.windows = in->windows; P.draft4 = in->draft4; P.arm_off = in->arm_off; P.arm_len = in->arm_len; P.arms2 = in->arms2;
    P.out_bases = out->bases; P.out_off = out->off; P.out_len = out->len; P.out_status = out->status;
    P.sr_m = sp->sr_match; P.sr_n = sp->sr_mismatch; P.sr_g = sp->sr_gap;
    P.lr_m = sp->lr_match; P.lr_n = sp->lr_mismatch; P.lr_g = sp->lr_gap;
    P.n_arms = in->n_arms; P.draft4_bytes = in->draft4_bytes; P.arms2_bytes = in->arms2_bytes;
    P.flags = getenv("HYPO_EMU_NATIVE_KLOV") ? hypo::POA_NATIVE_KLOV : 0;
    *cells = 0; *aligns = 0;
    switch (cfg_id) {
#define HYPO_CLASS_CASE(ID, CFG) case ID: return run_cfg<hypo::CFG>(P, in->n_windows, res, cells, aligns);
        HYPO_FOR_EACH_CLASS(HYPO_CLASS_CASE)
        HYPO_FOR_EACH_ALT_CLASS(HYPO_CLASS_CASE)
#undef HYPO_CLASS_CASE
        default: return -1;
    }
}

// The kernel's re-queue chain on the CPU: every window starts in class `cfg_from`; a window that answers RES_OVERFLOW /
// RES_UNSUPPORTED moves to the class poa_class_kernel would pick (the next one, or the first later SHORT class whose node table
// holds what it projects to need) and takes its spill along.  hops[w] = classes visited, carried[w] = hops that started from a spill.
extern "C" int emu_poa_chain(const HypoScoreParams* sp, const HypoWindowBatch* in, HypoConsensusBatch* out,
                             int cfg_from, uint8_t* res, uint8_t* hops, uint8_t* carried, int use_carry) {
    hypo::PoaParams P;
    P.windows = in->windows; P.draft4 = in->draft4; P.arm_off = in->arm_off; P.arm_len = in->arm_len; P.arms2 = in->arms2;
    P.out_bases = out->bases; P.out_off = out->off; P.out_len = out->len; P.out_status = out->status;
    P.sr_m = sp->sr_match; P.sr_n = sp->sr_mismatch; P.sr_g = sp->sr_gap;
    P.lr_m = sp->lr_match; P.lr_n = sp->lr_mismatch; P.lr_g = sp->lr_gap;
    P.n_arms = in->n_arms; P.draft4_bytes = in->draft4_bytes; P.arms2_bytes = in->arms2_bytes;
    P.flags = getenv("HYPO_EMU_NATIVE_KLOV") ? hypo::POA_NATIVE_KLOV : 0;
    constexpr int nmax[hypo::kNumPoaClasses] = {
#define HYPO_NMAX(ID, CFG) hypo::CFG::NMAX,
        HYPO_FOR_EACH_CLASS(HYPO_NMAX)
#undef HYPO_NMAX
    };
    constexpr int kFirstLong = 4, kRequeue = 3;           // poa_kernel.hpp: kFirstLongClass, kRequeueClass
    Sched s;
    uint64_t cells = 0, aligns = 0;
    for (uint32_t w = 0; w < in->n_windows; ++w) {
        Carry carry;
        int cls = cfg_from, rc = hypo::RES_OVERFLOW;
        hops[w] = 0; carried[w] = 0;
        for (;;) {
            if (!carry.blob.empty()) carried[w] += 1;
            hops[w] += 1;
            switch (cls) {
#define HYPO_CLASS_CASE(ID, CFG) case ID: rc = run_window<hypo::CFG>(s, P, w, &carry, &cells, &aligns); break;
                HYPO_FOR_EACH_CLASS(HYPO_CLASS_CASE)
#undef HYPO_CLASS_CASE
                default: return -1;
            }
            if (!use_carry) carry.blob.clear();
            if ((rc == hypo::RES_OVERFLOW || rc == hypo::RES_UNSUPPORTED) && cls + 1 < hypo::kNumPoaClasses) {
                int to = cls < kRequeue ? kRequeue : cls + 1;
                if (rc == hypo::RES_OVERFLOW && carry.need_nodes > 0) while (to + 1 < hypo::kNumPoaClasses && to < kFirstLong && nmax[to] < carry.need_nodes) ++to;
                cls = to;
                continue;
            }
            break;
        }
        res[w] = (uint8_t)rc;
        if (rc != hypo::RES_OK) { P.out_len[w] = 0; P.out_status[w] = 0xFF; }
    }
    free(s.stacks);
    return 0;
}

extern "C" void emu_exact_stats(uint64_t* out) { for (int i = 0; i < 3; ++i) out[i] = g_exact[i]; for (int i = 0; i < 4; ++i) g_exact[i] = 0; }
extern "C" uint64_t emu_guided_hits() { return g_exact[3]; }      // (read before emu_exact_stats clears it)

extern "C" int emu_class_bytes(int cfg_id) {
    switch (cfg_id) {
#define HYPO_CLASS_CASE(ID, CFG) case ID: return hypo::PoaLayout<hypo::CFG>::BYTES;
        HYPO_FOR_EACH_CLASS(HYPO_CLASS_CASE)
        HYPO_FOR_EACH_ALT_CLASS(HYPO_CLASS_CASE)
#undef HYPO_CLASS_CASE
        default: return -1;
    }
}
#ifdef HYPO_EMU_DBG
extern "C" void emu_dbg_hist(unsigned long* out) { for (int i = 0; i < 64; ++i) { out[i] = hypo::g_dbg_hist[i]; hypo::g_dbg_hist[i] = 0; } }
extern "C" void emu_dbg_reasons(unsigned long* out) { for (int i = 0; i < 16; ++i) { out[i] = hypo::g_dbg_reason[i]; hypo::g_dbg_reason[i] = 0; } }
#endif


// ---- size class 6 (hypo_amd/csrc/poa_giant.hpp): every window of the batch through Giant::run with a slice of `slice_bytes` ------------
namespace {
struct GiantJob {
    hypo::EmuGroup eg;
    const hypo::PoaParams* P;
    char* slice; uint64_t slice_bytes;
    uint32_t w;
    int rc[64];
    uint64_t cells, aligns;
};
void giant_lane_body(int lane, void* arg) {
    GiantJob* j = (GiantJob*)arg;
    hypo::Grp<64> g{lane, &j->eg};
    const hypo::PoaParamRef pr{j->P};
    hypo::Giant<hypo::Grp<64>> gi(g, pr);
    j->rc[lane] = gi.run(j->w, j->slice, j->slice_bytes);
    if (getenv("HYPO_EMU_DEBUG") && (lane < 2 || lane == 63)) fprintf(stderr, "[emu] giant lane %d rc %d\n", lane, j->rc[lane]);
    if (lane == 0) { j->cells = gi.cells; j->aligns = gi.aligns; }
}
}  // namespace

extern "C" int emu_poa_giant(const HypoScoreParams* sp, const HypoWindowBatch* in, HypoConsensusBatch* out, uint64_t slice_bytes,
                             uint8_t* res, uint64_t* cells, uint64_t* aligns) {
    hypo::PoaParams P;
    P.windows = in->windows; P.draft4 = in->draft4; P.arm_off = in->arm_off; P.arm_len = in->arm_len; P.arms2 = in->arms2;
    P.out_bases = out->bases; P.out_off = out->off; P.out_len = out->len; P.out_status = out->status;
    P.sr_m = sp->sr_match; P.sr_n = sp->sr_mismatch; P.sr_g = sp->sr_gap;
    P.lr_m = sp->lr_match; P.lr_n = sp->lr_mismatch; P.lr_g = sp->lr_gap;
    P.n_arms = in->n_arms; P.draft4_bytes = in->draft4_bytes; P.arms2_bytes = in->arms2_bytes;
    P.flags = getenv("HYPO_EMU_NATIVE_KLOV") ? hypo::POA_NATIVE_KLOV : 0;
    *cells = 0; *aligns = 0;
    Sched s;
    const int fill = getenv("HYPO_EMU_FILL") ? (int)strtol(getenv("HYPO_EMU_FILL"), nullptr, 0) : 0xA5;
    for (uint32_t w = 0; w < in->n_windows; ++w) {
        GiantJob job;
        job.P = &P; job.eg.gw = 64; job.eg.yield = yield_cb; job.eg.sched = &s;
        job.slice = (char*)malloc(slice_bytes);                  // exact size: ASan sees overruns
        memset(job.slice, fill, slice_bytes);
        job.slice_bytes = slice_bytes; job.w = w; job.cells = job.aligns = 0;
        out->status[w] = 0xFF; out->len[w] = 0;
        run_group(s, 64, giant_lane_body, &job);
        for (int l = 1; l < 64; ++l) if (job.rc[l] != job.rc[0]) { fprintf(stderr, "[emu] giant: lanes disagree on the result of window %u\n", w); abort(); }
        res[w] = (uint8_t)job.rc[0];
        if (job.rc[0] == hypo::RES_OK) { *cells += job.cells; *aligns += job.aligns; }
        free(job.slice);
    }
    free(s.stacks);
    return 0;
}
