// Contig.hpp — host mirror of the part of hypo::Contig that is on the hot path this round:
// construction from a sequence and find_solid_pos (reference: include/Contig.hpp:62-68,137-140,
// src/Contig.cpp:30-74).  The solid scan runs on the MI355X; the rank/select directory the reference builds
// with sdsl (Contig.cpp:72-73) is the word-rank array the device returns plus two small lookups.
// Segmentation, window construction and output (Contig.cpp:75-711) are "next" rows of SURVEY.md §8.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../../include/hypo_gpu.h"
#include "PackedSeq.hpp"

namespace hypo {

// what suk::SolidKmers exposes to the path (external/suk/include/suk/SolidKmers.hpp:108-140)
struct SolidKmers {
    uint32_t k = 0;
    std::vector<uint64_t> words;              // 4^k bits, bit i at words[i >> 6] bit (i & 63)
    uint32_t get_k() const { return k; }
    bool is_solid(uint64_t kid) const { return (words[kid >> 6] >> (kid & 63)) & 1; }
};

class Contig {
public:
    Contig(uint32_t id, const std::string& name, const std::string& seq)
        : _id(id), _name(name.substr(0, name.find_first_of(" \t"))), _len(seq.size()), _pseq(seq) {}

    // Contig::find_solid_pos; returns HYPO_OK or the C-ABI error code
    int find_solid_pos(const SolidKmers& sk) {
        const uint64_t nw = (_len + 63) / 64;
        _solid_pos.assign(nw, 0); _rank.assign(nw + 1, 0); _kids.assign(_len ? _len : 1, 0);
        uint64_t n = 0;
        const int rc = hypo_gpu_solid_scan(_pseq.data(), _len, sk.get_k(), sk.words.data(), _solid_pos.data(),
                                           _kids.data(), _kids.size(), _rank.data(), &n);
        if (rc != HYPO_OK) return rc;
        _kids.resize(n);
        return HYPO_OK;
    }
    uint64_t get_num_solid() const { return _kids.size(); }
    bool is_solid_pos(uint64_t p) const { return (_solid_pos[p >> 6] >> (p & 63)) & 1; }
    uint64_t kid_at(uint64_t i) const { return _kids[i]; }           // _kmerinfo[i]->kid (Contig.cpp:68)
    // sdsl::rank_support_v semantics: number of marked positions in [0, p)
    uint64_t rank(uint64_t p) const {
        const uint64_t w = p >> 6, b = p & 63;
        return _rank[w] + (b ? (uint64_t)__builtin_popcountll(_solid_pos[w] & ((1ULL << b) - 1)) : 0);
    }
    // sdsl::select_support_mcl semantics: position of the i-th marked bit, i is 1-based
    uint64_t select(uint64_t i) const {
        uint64_t lo = 0, hi = _solid_pos.size();                     // largest word w with _rank[w] < i
        while (lo + 1 < hi) { const uint64_t mid = (lo + hi) / 2; if (_rank[mid] < i) lo = mid; else hi = mid; }
        uint64_t word = _solid_pos[lo], need = i - _rank[lo];
        while (--need) word &= word - 1;
        return lo * 64 + (uint64_t)__builtin_ctzll(word);
    }
    const std::string& get_name() const { return _name; }
    uint64_t get_len() const { return _len; }

private:
    uint32_t _id;
    std::string _name;
    uint64_t _len;
    PackedSeq<4> _pseq;
    std::vector<uint64_t> _solid_pos, _rank, _kids;
};

}  // namespace hypo
