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

struct WindowFlattener {
    std::vector<HypoWindow> win;
    std::vector<uint8_t> draft4, arms2;
    std::vector<uint64_t> arm_off;
    std::vector<uint32_t> arm_len;
    void add(const Window& w) {
        HypoWindow d{};
        d.type = w._wtype == WindowType::SHORT ? HYPO_WIN_SHORT : HYPO_WIN_LONG;
        d.draft_len = (uint32_t)w._draft.get_seq_size();
        d.draft_off = draft4.size();
        draft4.insert(draft4.end(), w._draft.data(), w._draft.data() + w._draft.byte_size());
        d.first_arm = (uint32_t)arm_len.size();
        d.n_internal = (uint32_t)w._internal_arms.size();
        d.n_prefix = (uint32_t)w._pre_arms.size();
        d.n_suffix = (uint32_t)w._suf_arms.size();
        d.n_empty = w._num_empty;
        for (const auto* group : {&w._internal_arms, &w._pre_arms, &w._suf_arms})
            for (const auto& a : *group) {
                arm_off.push_back(arms2.size());
                arm_len.push_back((uint32_t)a.get_seq_size());
                arms2.insert(arms2.end(), a.data(), a.data() + a.byte_size());
            }
        win.push_back(d);
    }
};

int Window::generate_consensus_batch(const std::vector<Window*>& windows) {
    if (windows.empty()) return HYPO_OK;
    // flatten on all threads: sizes, exclusive prefix sums, then every window copies into its own slices
    WindowFlattener f;
    const int64_t nw = (int64_t)windows.size();
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
    f.win.resize((size_t)nw);
    f.draft4.resize(d_off[(size_t)nw] + 16); f.arms2.resize(a_off[(size_t)nw] + 16);
    f.arm_off.resize(a_cnt[(size_t)nw]); f.arm_len.resize(a_cnt[(size_t)nw]);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nw; ++i) {
        const Window& w = *windows[(size_t)i];
        HypoWindow d{};
        d.type = w._wtype == WindowType::SHORT ? HYPO_WIN_SHORT : HYPO_WIN_LONG;
        d.draft_len = (uint32_t)w._draft.get_seq_size();
        d.draft_off = d_off[(size_t)i];
        std::memcpy(f.draft4.data() + d_off[(size_t)i], w._draft.data(), w._draft.byte_size());
        d.first_arm = (uint32_t)a_cnt[(size_t)i];
        d.n_internal = (uint32_t)w._internal_arms.size();
        d.n_prefix = (uint32_t)w._pre_arms.size();
        d.n_suffix = (uint32_t)w._suf_arms.size();
        d.n_empty = w._num_empty;
        uint64_t ai = a_cnt[(size_t)i], ao = a_off[(size_t)i];
        for (const auto* group : {&w._internal_arms, &w._pre_arms, &w._suf_arms})      // insertion order, as stored
            for (const auto& a : *group) {
                f.arm_off[ai] = ao; f.arm_len[ai] = (uint32_t)a.get_seq_size();
                std::memcpy(f.arms2.data() + ao, a.data(), a.byte_size());
                ++ai; ao += a.byte_size();
            }
        f.win[(size_t)i] = d;
    }
    HypoWindowBatch in{(uint32_t)f.win.size(), (uint32_t)f.arm_len.size(), f.win.data(), f.draft4.data(), f.draft4.size(),
                       f.arm_off.data(), f.arm_len.data(), f.arms2.data(), f.arms2.size()};
    std::vector<uint64_t> off(f.win.size() + 1);
    int rc = hypo_gpu_poa_slot_layout(&in, off.data());
    if (rc != HYPO_OK) return rc;
    std::vector<char> bases(off.back() + 1);
    std::vector<uint32_t> len(f.win.size());
    std::vector<uint8_t> st(f.win.size());
    HypoConsensusBatch out{bases.data(), off.data(), len.data(), st.data()};
    rc = hypo_gpu_poa_batch(&_score_params, &in, &out);
    if (rc != HYPO_OK) return rc;
    for (size_t i = 0; i < windows.size(); ++i)
        if (st[i] != HYPO_ST_OK) {
            std::fprintf(stderr, "[Hypo::Window] Error: window %zu could not be polished on the device (status %u)\n", i, (unsigned)st[i]);
            return HYPO_E_INVALID;
        }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nw; ++i) windows[(size_t)i]->_consensus.assign(bases.data() + off[(size_t)i], len[(size_t)i]);
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
