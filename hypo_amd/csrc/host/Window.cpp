// Window.cpp — batched device consensus behind hypo::Window (see Window.hpp).
#include "Window.hpp"
#include <cstring>
#include <cstdio>
#include <cstdlib>

namespace hypo {

ScoreParams Window::_score_params = {5, -4, -8, 3, -5, -4};     // reference defaults, src/main.cpp:100-113

// The reference creates one spoa engine pair per thread here (Window.cpp:31-42); on the device the scores
// are an argument of the batch call and `num_threads` has no meaning.
void Window::prepare_for_poa(const ScoreParams& sp, uint32_t /*num_threads*/) { _score_params = sp; }

// One device call for `windows` (all of them fit the 32-bit counters of the boundary).  `slot_hint`: consensus slot sizes
// to use instead of the library's recommendation (second attempt of windows whose consensus outgrew the first slot).
int Window::consensus_call(const ScoreParams& sp, const std::vector<Window*>& windows, const std::vector<uint32_t>* slot_hint,
                          std::vector<uint8_t>& st, std::vector<uint32_t>& len) {
    // flatten on all threads: sizes, exclusive prefix sums, then every window copies into its own slices
    const int64_t nw = (int64_t)windows.size();
    std::vector<HypoWindow> win((size_t)nw);
    std::vector<uint64_t> d_off((size_t)nw + 1), a_cnt((size_t)nw + 1), a_off((size_t)nw + 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nw; ++i) {
        const Window& w = *windows[(size_t)i];
        uint64_t bytes = 0;
        for (const auto* group : {&w._internal_arms, &w._pre_arms, &w._suf_arms}) for (const auto& a : *group) bytes += a.byte_size();
        d_off[(size_t)i + 1] = w._draft.byte_size();
        a_cnt[(size_t)i + 1] = w._internal_arms.size() + w._pre_arms.size() + w._suf_arms.size();
        a_off[(size_t)i + 1] = bytes;
    }
    d_off[0] = a_cnt[0] = a_off[0] = 0;
    for (int64_t i = 0; i < nw; ++i) { d_off[(size_t)i + 1] += d_off[(size_t)i]; a_cnt[(size_t)i + 1] += a_cnt[(size_t)i]; a_off[(size_t)i + 1] += a_off[(size_t)i]; }
    std::vector<uint8_t> draft4(d_off[(size_t)nw] + 16), arms2(a_off[(size_t)nw] + 16);
    std::vector<uint64_t> arm_off(a_cnt[(size_t)nw]);
    std::vector<uint32_t> arm_len(a_cnt[(size_t)nw]);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nw; ++i) {
        const Window& w = *windows[(size_t)i];
        HypoWindow d{};
        d.type = w._wtype == WindowType::SHORT ? HYPO_WIN_SHORT : HYPO_WIN_LONG;
        d.draft_len = (uint32_t)w._draft.get_seq_size();
        d.draft_off = d_off[(size_t)i];
        std::memcpy(draft4.data() + d_off[(size_t)i], w._draft.data(), w._draft.byte_size());
        d.first_arm = (uint32_t)a_cnt[(size_t)i];
        d.n_internal = (uint32_t)w._internal_arms.size();
        d.n_prefix = (uint32_t)w._pre_arms.size();
        d.n_suffix = (uint32_t)w._suf_arms.size();
        d.n_empty = w._num_empty;
        uint64_t ai = a_cnt[(size_t)i], ao = a_off[(size_t)i];
        for (const auto* group : {&w._internal_arms, &w._pre_arms, &w._suf_arms})      // insertion order, as stored
            for (const auto& a : *group) {
                arm_off[ai] = ao; arm_len[ai] = (uint32_t)a.get_seq_size();
                std::memcpy(arms2.data() + ao, a.data(), a.byte_size());
                ++ai; ao += a.byte_size();
            }
        win[(size_t)i] = d;
    }
    // the arms were laid back to back above: a single device computes their offsets itself (arm_off = NULL saves a third of the upload)
    const bool sharded = hypo_gpu_num_devices() > 1;
    HypoWindowBatch in{(uint32_t)win.size(), (uint32_t)arm_len.size(), win.data(), draft4.data(), draft4.size(),
                       sharded ? arm_off.data() : nullptr, arm_len.data(), arms2.data(), arms2.size()};
    std::vector<uint64_t> off(win.size() + 1);
    int rc = HYPO_OK;
    if (slot_hint) { off[0] = 0; for (size_t i = 0; i < win.size(); ++i) off[i + 1] = off[i] + ((uint64_t)(*slot_hint)[i] + 7) / 8 * 8; }
    else { HypoWindowBatch lay = in; lay.arm_off = arm_off.data(); rc = hypo_gpu_poa_slot_layout(&lay, off.data()); }
    if (rc != HYPO_OK) return rc;
    std::vector<char> bases(off.back() + 1);
    len.assign(win.size(), 0);
    st.assign(win.size(), 0);
    HypoConsensusBatch out{bases.data(), off.data(), len.data(), st.data()};
    // all contexts of hypo_gpu_init share the batch (one context: the plain call)
    rc = sharded ? hypo_gpu_poa_batch_sharded(&sp, &in, &out) : hypo_gpu_poa_batch(&sp, &in, &out);
    if (rc != HYPO_OK) return rc;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nw; ++i)
        if (st[(size_t)i] == HYPO_ST_OK) windows[(size_t)i]->_consensus.assign(bases.data() + off[(size_t)i], len[(size_t)i]);
    return HYPO_OK;
}

int Window::generate_consensus_batch(const std::vector<Window*>& windows) {
    if (windows.empty()) return HYPO_OK;
    // The boundary counts windows and arms in 32 bits (HypoWindow::first_arm, HypoWindowBatch::n_arms): a contig batch
    // with more (the default -p 0 puts a whole genome into one batch) goes over in several calls.
    constexpr uint64_t kMaxArmsPerCall = 0xfff00000ull, kMaxWindowsPerCall = 0x7fffffffull;
    uint64_t n_capacity = 0, n_undefined = 0;
    size_t beg = 0;
    while (beg < windows.size()) {
        size_t end = beg;
        uint64_t arms = 0;
        while (end < windows.size() && end - beg < kMaxWindowsPerCall) {
            const Window& w = *windows[end];
            const uint64_t a = w._internal_arms.size() + w._pre_arms.size() + w._suf_arms.size();
            if (end > beg && arms + a > kMaxArmsPerCall) break;
            if (a > kMaxArmsPerCall) { std::fprintf(stderr, "[Hypo::Window] Error: a window with %llu arms exceeds the device boundary\n", (unsigned long long)a); return HYPO_E_INVALID; }
            arms += a; ++end;
        }
        std::vector<Window*> part(windows.begin() + (std::ptrdiff_t)beg, windows.begin() + (std::ptrdiff_t)end);
        std::vector<uint8_t> st; std::vector<uint32_t> len;
        int rc = consensus_call(_score_params, part, nullptr, st, len);
        if (rc != HYPO_OK) return rc;
        // a consensus longer than its slot (status CONS_OVERFLOW, len = the size it needs): once more with that size
        std::vector<Window*> again; std::vector<uint32_t> need;
        for (size_t i = 0; i < part.size(); ++i) if (st[i] == HYPO_ST_CONS_OVERFLOW) { again.push_back(part[i]); need.push_back(len[i] + 8); }
        if (!again.empty()) {
            std::vector<uint8_t> st2; std::vector<uint32_t> len2;
            rc = consensus_call(_score_params, again, &need, st2, len2);
            if (rc != HYPO_OK) return rc;
            size_t j = 0;
            for (size_t i = 0; i < part.size(); ++i) if (st[i] == HYPO_ST_CONS_OVERFLOW) st[i] = st2[j++];
        }
        // Degraded path (documented in DESIGN.md): a window beyond the largest device size class (more than 16 382
        // sequences, a graph of more than 32 767 nodes or 58 in-edges per node, an arm longer than 1 021 bases), or one
        // whose alignment is undefined in the reference itself, keeps its draft; the run goes on.
        // Anything else is a defect of the flattening / sharding above, not a property of the window (HYPO_ST_INVALID: a descriptor
        // that points outside the batch; a slot overflow that survived the retry): fatal, as every non-OK status was before the
        // degraded path existed.
        for (size_t i = 0; i < part.size(); ++i) {
            if (st[i] == HYPO_ST_OK) continue;
            if (st[i] == HYPO_ST_UNDEFINED) ++n_undefined;
            else if (st[i] == HYPO_ST_CAPACITY) ++n_capacity;
            else {
                std::fprintf(stderr, "[Hypo::Window] Error: window %llu of the batch came back with status %d (%s)\n", (unsigned long long)(beg + i),
                             (int)st[i], st[i] == HYPO_ST_INVALID ? "descriptor outside the batch" : "unexpected");
                return HYPO_E_INVALID;
            }
            part[i]->_consensus = part[i]->_draft.unpack();
        }
        beg = end;
    }
    if (n_capacity || n_undefined)
        std::fprintf(stderr, "[Hypo::Window] Warning: %llu window(s) exceed the device's largest size class and %llu hit an alignment "
                     "the reference leaves undefined: their draft sequence is kept unpolished\n",
                     (unsigned long long)n_capacity, (unsigned long long)n_undefined);
    return HYPO_OK;
}

std::string Window::dump_text() const {
    std::string t = _draft.unpack();
    const char* tag[3] = {"|I:", "|P:", "|S:"};
    int gi = 0;
    for (const auto* group : {&_internal_arms, &_pre_arms, &_suf_arms}) {
        for (const auto& a : *group) { t += tag[gi]; t += a.unpack(); }
        ++gi;
    }
    return t;
}

uint32_t Window::arms_crc32() const {
    uint32_t crc = 0xffffffffu;
    auto feed = [&crc](unsigned char c) { crc ^= c; for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xedb88320u & (0u - (crc & 1u))); };
    bool first = true;
    for (const auto* group : {&_internal_arms, &_pre_arms, &_suf_arms})
        for (const auto& a : *group) {
            if (!first) feed('\n');
            first = false;
            for (char c : a.unpack()) feed((unsigned char)c);
        }
    return ~crc;
}

void Window::generate_consensus(uint32_t /*engine_idx*/) {
    std::vector<Window*> one{this};
    const int rc = generate_consensus_batch(one);
    if (rc != HYPO_OK) {          // the reference's error style: message + exit(1)
        std::fprintf(stderr, "[Hypo::Window] Error: %s\n", hypo_gpu_last_error());
        std::exit(1);
    }
}

}  // namespace hypo
