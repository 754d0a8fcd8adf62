// Window.hpp — host mirror of hypo::Window (reference: include/Window.hpp:41-146, src/Window.cpp).
// Same object surface: construct from a draft slice, add_prefix / add_suffix / add_internal / add_empty,
// counters, clear_pre_suf, prepare_for_poa, generate_consensus, get_consensus.  The consensus itself is
// computed on the MI355X through the C-ABI (include/hypo_gpu.h); generate_consensus_batch() is the batched
// form Hypo::polish() should call once per contig batch, generate_consensus(engine_idx) is
// kept for source compatibility and polishes a single window.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>
#include "../../../include/hypo_gpu.h"
#include "Filter.hpp"
#include "PackedSeq.hpp"

namespace hypo {

using ScoreParams = HypoScoreParams;          // include/globalDefs.hpp:58-66 (same fields, same order)

enum class WindowType : uint8_t { SHORT, LONG };

class Window {
public:
    Window() = default;
    Window(const PackedSeq<4>& ps, size_t left_ind, size_t right_ind, WindowType wt)
        : _wtype(wt), _draft(ps, left_ind, right_ind) {
        if (wt == WindowType::LONG) _filter.initialise(ps.unpack(left_ind, right_ind));
    }
    Window(const Window&) = delete;
    Window& operator=(const Window&) = delete;

    static void prepare_for_poa(const ScoreParams& sp, uint32_t num_threads);
    void generate_consensus(uint32_t engine_idx);
    // polishes every window of the list with ONE device call; returns HYPO_OK or the C-ABI error
    static int generate_consensus_batch(const std::vector<Window*>& windows);
    std::string get_consensus() const { return _consensus; }
    const std::string& consensus_ref() const { return _consensus; }
    size_t get_window_len() const { return _draft.get_seq_size(); }

    // The reference takes const references and copies (Window.hpp:66-101); an rvalue is adopted without a second copy.
    void add_prefix(const PackedSeq<2>& ps) { add_prefix(PackedSeq<2>(ps)); }
    void add_suffix(const PackedSeq<2>& ps) { add_suffix(PackedSeq<2>(ps)); }
    void add_internal(const PackedSeq<2>& ps) { add_internal(PackedSeq<2>(ps)); }
    void add_prefix(PackedSeq<2>&& ps) {
        if (_wtype == WindowType::LONG && !_filter.is_good(ps.unpack())) return;
        ++_num_pre;
        if (ps.get_seq_size() > _longest_pre_len) _longest_pre_len = (uint32_t)ps.get_seq_size();
        _pre_arms.push_back(std::move(ps));
    }
    void add_suffix(PackedSeq<2>&& ps) {
        if (_wtype == WindowType::LONG && !_filter.is_good(ps.unpack())) return;
        ++_num_suf;
        if (ps.get_seq_size() > _longest_suf_len) _longest_suf_len = (uint32_t)ps.get_seq_size();
        _suf_arms.push_back(std::move(ps));
    }
    void add_internal(PackedSeq<2>&& ps) {
        if (_wtype == WindowType::LONG && !_filter.is_good(ps.unpack())) return;
        ++_num_internal;
        _internal_arms.push_back(std::move(ps));
    }
    void add_empty() { ++_num_empty; }

    uint32_t get_num_pre() const { return _num_pre; }
    uint32_t get_num_suf() const { return _num_suf; }
    uint32_t get_num_internal() const { return _num_internal + _num_empty; }      // Window.hpp:107
    uint32_t get_num_total() const { return _num_internal + _num_empty + _num_pre + _num_suf; }
    uint32_t get_maxlen_pre() const { return _longest_pre_len; }
    uint32_t get_maxlen_suf() const { return _longest_suf_len; }
    // diagnostics used by the region dump of Hypo::polish (tests): counts as the reference's inspect file prints them
    std::string dump_counts() const {
        return std::to_string(_num_internal) + "\t" + std::to_string(_num_pre) + "\t" + std::to_string(_num_suf) + "\t" + std::to_string(_num_empty);
    }
    bool is_long() const { return _wtype == WindowType::LONG; }
    std::string dump_text() const;            // draft and arms as text (debugging aid of the region dump: HYPO_REGION_DUMP_ARMS=1)
    uint32_t arms_crc32() const;             // crc32 of the arms (internal, prefix, suffix; insertion order) joined by '\n'
    void clear_pre_suf() {
        _num_pre = 0; _num_suf = 0;
        _pre_arms.clear(); _suf_arms.clear();
        _pre_arms.shrink_to_fit(); _suf_arms.shrink_to_fit();
    }

private:
    friend class DeviceArms;
    static int consensus_call(const ScoreParams& sp, const std::vector<Window*>& windows, const std::vector<uint32_t>* slot_hint,
                              std::vector<uint8_t>& st, std::vector<uint32_t>& len);
    WindowType _wtype = WindowType::SHORT;
    uint32_t _num_internal = 0, _num_pre = 0, _num_suf = 0, _num_empty = 0;
    uint32_t _longest_pre_len = 0, _longest_suf_len = 0;
    PackedSeq<4> _draft;
    std::vector<PackedSeq<2>> _internal_arms, _pre_arms, _suf_arms;
    std::string _consensus;
    Filter _filter;                              // only for long windows
    static ScoreParams _score_params;
};

}  // namespace hypo
