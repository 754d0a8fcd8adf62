// BitVec.hpp — bit vector with rank / select, the only part of sdsl-lite the path uses (reference:
// sdsl::bit_vector, rank_1_type, select_1_type; every use is listed in SURVEY.md §2.2).
// Word layout equals sdsl's payload: bit i lives at words[i >> 6], bit (i & 63).
//   rank(i)   = number of set bits in [0, i)            (sdsl::rank_support_v::operator())
//   select(i) = position of the i-th set bit, i >= 1    (sdsl::select_support_mcl::operator())
#pragma once
#include <cstdint>
#include <vector>

namespace hypo {

class BitVec {
public:
    BitVec() = default;
    explicit BitVec(uint64_t n_bits) : _n(n_bits), _w((n_bits + 63) / 64, 0) {}
    uint64_t size() const { return _n; }
    bool operator[](uint64_t i) const { return (_w[i >> 6] >> (i & 63)) & 1; }
    void set(uint64_t i) { _w[i >> 6] |= 1ULL << (i & 63); }
    uint64_t* data() { return _w.data(); }
    const uint64_t* data() const { return _w.data(); }
    uint64_t n_words() const { return _w.size(); }
    // (re)build the rank directory; call after the last set()
    void init_support() {
        _r.assign(_w.size() + 1, 0);
        for (size_t i = 0; i < _w.size(); ++i) _r[i + 1] = _r[i] + (uint64_t)__builtin_popcountll(_w[i]);
    }
    // adopt a directory computed elsewhere (the device scan returns it)
    void adopt_rank(std::vector<uint64_t>&& r) { _r = std::move(r); }
    uint64_t count() const { return _r.empty() ? 0 : _r.back(); }
    uint64_t rank(uint64_t i) const {
        const uint64_t w = i >> 6, b = i & 63;
        return _r[w] + (b ? (uint64_t)__builtin_popcountll(_w[w] & ((1ULL << b) - 1)) : 0);
    }
    uint64_t select(uint64_t i) const {
        uint64_t lo = 0, hi = _w.size();                 // largest word index with _r[w] < i
        while (lo + 1 < hi) { const uint64_t mid = (lo + hi) / 2; if (_r[mid] < i) lo = mid; else hi = mid; }
        uint64_t word = _w[lo], need = i - _r[lo];
        while (--need) word &= word - 1;
        return lo * 64 + (uint64_t)__builtin_ctzll(word);
    }
    // the positions of all set bits, in order (entry i = select(i + 1)): one pass over the words instead of a binary search per bit
    template <class T> void list_set(std::vector<T>& out) const {
        out.clear();
        if (!_r.empty()) out.reserve(_r.back());
        for (uint64_t wi = 0; wi < _w.size(); ++wi) {
            uint64_t word = _w[wi];
            while (word) { out.push_back((T)(wi * 64 + (uint64_t)__builtin_ctzll(word))); word &= word - 1; }
        }
    }
    void clear() { _n = 0; std::vector<uint64_t>().swap(_w); std::vector<uint64_t>().swap(_r); }

private:
    uint64_t _n = 0;
    std::vector<uint64_t> _w, _r;
};

}  // namespace hypo
