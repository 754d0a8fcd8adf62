// Filter.hpp — host mirror of hypo::Filter + MinimizerDeque (reference: include/Filter.hpp:33-102,
// include/MinimizerDeque.hpp).  LONG windows keep an arm only if it shares >= 1 canonical (k=10, w=10)
// window minimizer with the window's draft per 50 bases.  Same sliding-minimum semantics, including the
// reference's quirks: the rolling k-mers and the processed-k-mer counter are NOT reset by an N, only the
// run length of non-N bases is.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_set>
#include <vector>
#include "PackedSeq.hpp"

namespace hypo {

class Filter {
    static constexpr unsigned K = 10, W = 10, BP_PER_MINIMIZER = 50;
    struct Item { uint64_t kmer; uint32_t pos; };
    // monotone queue on a fixed ring of W + 1 slots (MinimizerDeque.hpp)
    struct Ring {
        Item slot[W + 1]; unsigned head = 0, tail = W, count = 0;
        bool empty() const { return count == 0; }
        Item& front() { return slot[head]; }
        Item& back() { return slot[tail]; }
        void push_back(Item x) { ++count; tail = (tail + 1) % (W + 1); slot[tail] = x; }
        void pop_back() { --count; tail = tail == 0 ? W : tail - 1; }
        void pop_front() { head = (head + 1) % (W + 1); --count; }
    };
    template <class OnWindow>
    static void scan(const std::string& s, OnWindow&& on_window) {
        const uint64_t shift = 2 * (K - 1), mask = (1ULL << (2 * K)) - 1;
        uint64_t fwd = 0, rev = 0;
        Ring q;
        unsigned run = 0, processed = 0;
        for (size_t i = 0; i < s.size(); ++i) {
            const uint8_t c = nt4((unsigned char)s[i]);
            if (c >= 4) { run = 0; continue; }
            ++run;
            fwd = ((fwd << 2) | c) & mask;
            rev = (rev >> 2) | ((uint64_t)(3 ^ c) << shift);
            const uint64_t canon = fwd < rev ? fwd : rev;
            if (run < K) continue;
            while (!q.empty() && q.back().kmer > canon) q.pop_back();
            q.push_back(Item{canon, (uint32_t)i});
            while (q.front().pos + W <= i) q.pop_front();
            if (++processed >= W) on_window(q.front());
        }
    }

public:
    void initialise(const std::string& draft) {
        scan(draft, [this](const Item& m) { _draft_minimizers.insert(m.kmer); });
    }
    bool is_good(const std::string& arm) const {
        std::vector<Item> found;
        scan(arm, [&found](const Item& m) { if (found.empty() || found.back().pos != m.pos) found.push_back(m); });
        uint32_t hits = 0;
        for (const Item& m : found) hits += _draft_minimizers.count(m.kmer) ? 1u : 0u;
        return (size_t)hits * BP_PER_MINIMIZER >= arm.size();
    }

private:
    std::unordered_set<uint64_t> _draft_minimizers;
};

}  // namespace hypo
