// ReadBatch.cpp — see ReadBatch.hpp.
#include "ReadBatch.hpp"
#include <chrono>
#include <omp.h>
#include <algorithm>
#include <cstdlib>
#include "../../../include/hypo_gpu.h"

namespace hypo {

bool ReadStaging::reserve(size_t reads, size_t cig, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    struct Timer { std::chrono::steady_clock::time_point t0; double& acc; ~Timer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } timer{t0, reserve_seconds};
    rb = _b[0].get<uint32_t>(reads + 1); re = _b[1].get<uint32_t>(reads + 1); qae = _b[2].get<uint32_t>(reads + 1); ctg = _b[3].get<uint32_t>(reads + 1);
    cigar_off = _b[4].get<uint32_t>(reads + 2); seq_off = _b[5].get<uint64_t>(reads + 2); file_rank = _b[6].get<uint32_t>(reads + 1);
    cigar = _b[7].get<uint32_t>(cig + 1); reads2 = _b[8].get<uint8_t>(bytes + 16);
    return rb && re && qae && ctg && cigar_off && seq_off && file_rank && cigar && reads2;
}

void ReadBatch::add(const std::shared_ptr<ParsedBlock>& blk, size_t r0, size_t r1) {
    if (r0 >= r1) return;
    for (uint32_t c = 0; c < blk->chunks.size(); ++c) {
        const ReadChunk& ch = blk->chunks[c];
        if (ch.raw.empty() || ch.raw.back() < r0 || ch.raw.front() >= r1) continue;
        const uint32_t k0 = (uint32_t)(std::lower_bound(ch.raw.begin(), ch.raw.end(), (uint32_t)r0) - ch.raw.begin());
        const uint32_t k1 = (uint32_t)(std::lower_bound(ch.raw.begin(), ch.raw.end(), (uint32_t)r1) - ch.raw.begin());
        if (k0 >= k1) continue;
        _slices.push_back(Slice{blk, c, k0, k1});
        // (runs of one contig: a sorted file changes contig a few times per batch)
        uint32_t k = k0;
        while (k < k1) {
            const int32_t id = ch.cid[k];
            uint32_t e = k + 1;
            while (e < k1 && ch.cid[e] == id) ++e;
            if ((size_t)id >= _per_contig.size()) _per_contig.resize((size_t)id + 1, 0);
            _per_contig[(size_t)id] += e - k;
            k = e;
        }
        _n += k1 - k0;
    }
}

uint32_t ReadBatch::max_span(uint32_t cid) const {
    uint32_t best = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(max : best)
    for (int64_t s = 0; s < (int64_t)_slices.size(); ++s) {
        const Slice& sl = _slices[(size_t)s];
        const ReadChunk& ch = sl.blk->chunks[sl.chunk];
        for (uint32_t k = sl.k0; k < sl.k1; ++k)
            if ((uint32_t)ch.cid[k] == cid && ch.re[k] - ch.rb[k] > best) best = ch.re[k] - ch.rb[k];
    }
    return best;
}

void ReadBatch::append(ReadBatch& other) {
    for (auto& s : other._slices) _slices.push_back(std::move(s));
    if (other._per_contig.size() > _per_contig.size()) _per_contig.resize(other._per_contig.size(), 0);
    for (size_t c = 0; c < other._per_contig.size(); ++c) _per_contig[c] += other._per_contig[c];
    _n += other._n;
    other._slices.clear(); std::fill(other._per_contig.begin(), other._per_contig.end(), 0); other._n = 0;
}

void ReadBatch::prepend(uint32_t cid, const std::vector<std::unique_ptr<Alignment>>& objs, size_t at_slice) {
    if (objs.empty()) return;
    auto blk = std::make_shared<ParsedBlock>();
    blk->chunks.resize(1);
    ReadChunk& d = blk->chunks[0];
    d.clear();
    for (const auto& a : objs) {
        d.raw.push_back((uint32_t)d.raw.size()); d.cid.push_back((int32_t)cid); d.rb.push_back(a->_rb); d.re.push_back(a->_re); d.qae.push_back(a->_qae);
        d.seq.insert(d.seq.end(), a->_apseq.data(), a->_apseq.data() + a->_apseq.byte_size());
        d.cig.insert(d.cig.end(), a->_cigar.begin(), a->_cigar.end());
        d.seq_at.push_back((uint32_t)d.seq.size()); d.cig_at.push_back((uint32_t)d.cig.size());
    }
    blk->n = d.raw.size();
    blk->status.assign(blk->n, ParsedBlock::ST_KEPT);
    blk->cid = d.cid;
    ReadBatch front;
    front.reset(_per_contig.size());
    front.add(blk, 0, blk->n);
    const size_t n_new = front._slices.size();
    front.append(*this);
    if (at_slice > 0 && front._slices.size() >= n_new + at_slice)   // [.. carried record | objs | the rest]: the slices that were in front of `at_slice` move back to the front
        std::rotate(front._slices.begin(), front._slices.begin() + n_new, front._slices.begin() + n_new + at_slice);
    _slices.swap(front._slices); _per_contig.swap(front._per_contig); _n = front._n;
}

void ReadBatch::clear(std::vector<std::shared_ptr<ParsedBlock>>* pool, std::mutex* pool_mu) {
    if (pool && pool_mu) {
        std::lock_guard<std::mutex> lk(*pool_mu);
        for (auto& s : _slices)
            if (s.blk && s.blk.use_count() == 1 && pool->size() < 8) pool->push_back(std::move(s.blk)); else s.blk.reset();
    }
    _slices.clear();
    std::fill(_per_contig.begin(), _per_contig.end(), 0);
    _n = 0;
}

bool ReadBatch::flatten(uint32_t c0, uint32_t c1, const std::vector<uint64_t>& base, ReadStaging& out, bool& sorted, const uint32_t* span) const {
    struct Run { uint32_t slice, k0, k1; int32_t cid; uint64_t at, bytes, cigs, byte_at, cig_at; };
    std::vector<Run> runs;
    std::vector<uint64_t> taken(c1 - c0, 0);
    for (uint32_t s = 0; s < _slices.size(); ++s) {
        const Slice& sl = _slices[s];
        const ReadChunk& ch = sl.blk->chunks[sl.chunk];
        uint32_t k = sl.k0;
        while (k < sl.k1) {
            const int32_t id = ch.cid[k];
            uint32_t e = k + 1;
            while (e < sl.k1 && ch.cid[e] == id) ++e;
            if ((uint32_t)id >= c0 && (uint32_t)id < c1) {
                if (!span) { runs.push_back(Run{s, k, e, id, 0, (uint64_t)ch.seq_at[e] - ch.seq_at[k], (uint64_t)ch.cig_at[e] - ch.cig_at[k], 0, 0}); taken[(uint32_t)id - c0] += e - k; }
                else {                                   // stretches of records that overlap the span
                    uint32_t a = k;
                    while (a < e) {
                        while (a < e && !(ch.rb[a] < span[1] && ch.re[a] > span[0])) ++a;
                        uint32_t b = a;
                        while (b < e && ch.rb[b] < span[1] && ch.re[b] > span[0]) ++b;
                        if (b > a) { runs.push_back(Run{s, a, b, id, 0, (uint64_t)ch.seq_at[b] - ch.seq_at[a], (uint64_t)ch.cig_at[b] - ch.cig_at[a], 0, 0}); taken[(uint32_t)id - c0] += b - a; }
                        a = b;
                    }
                }
            }
            k = e;
        }
    }
    // where every run goes: contigs in order, a contig's runs in file order
    std::vector<uint64_t> first(c1 - c0 + 1, 0);
    for (uint32_t c = c0; c < c1; ++c) first[c - c0 + 1] = first[c - c0] + taken[c - c0];
    std::vector<uint64_t> seen(c1 - c0, 0);
    for (Run& r : runs) { r.at = first[(uint32_t)r.cid - c0] + seen[(uint32_t)r.cid - c0]; seen[(uint32_t)r.cid - c0] += r.k1 - r.k0; }
    std::vector<uint32_t> order(runs.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return runs[a].at < runs[b].at; });
    uint64_t nb = 0, ncg = 0;
    for (uint32_t i : order) { runs[i].byte_at = nb; runs[i].cig_at = ncg; nb += runs[i].bytes; ncg += runs[i].cigs; }
    const uint64_t n = first[c1 - c0];
    if (ncg >= 0xfffffff0ull) return false;
    if (!out.reserve(n, ncg, nb)) return false;
    out.n_reads = n; out.n_cigar = ncg; out.n_bytes = nb;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ri = 0; ri < (int64_t)runs.size(); ++ri) {
        const Run& r = runs[(size_t)ri];
        const ReadChunk& ch = _slices[r.slice].blk->chunks[_slices[r.slice].chunk];
        const uint64_t b = base[(uint32_t)r.cid - c0];
        const uint32_t s0 = ch.seq_at[r.k0], g0 = ch.cig_at[r.k0];
        for (uint32_t k = r.k0; k < r.k1; ++k) {
            const uint64_t g = r.at + (k - r.k0);
            out.rb[g] = (uint32_t)(b + ch.rb[k]); out.re[g] = (uint32_t)(b + ch.re[k]); out.qae[g] = ch.qae[k]; out.ctg[g] = (uint32_t)r.cid - c0;
            out.seq_off[g] = r.byte_at + (ch.seq_at[k] - s0);
            out.cigar_off[g] = (uint32_t)(r.cig_at + (ch.cig_at[k] - g0));
        }
        if (r.bytes) std::memcpy(out.reads2 + r.byte_at, ch.seq.data() + s0, r.bytes);
        if (r.cigs) std::memcpy(out.cigar + r.cig_at, ch.cig.data() + g0, r.cigs * 4);
    }
    out.cigar_off[n] = (uint32_t)ncg;
    out.seq_off[n] = nb;
    bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok)
    for (int64_t g = 1; g < (int64_t)n; ++g)
        if (out.ctg[g] == out.ctg[g - 1] && out.rb[g - 1] > out.rb[g]) ok = false;
    sorted = ok;
    out.ranked = false;
    if (!ok) {
        // An unsorted file (the reference takes any order, src/Hypo.cpp:278-329): the device's kernels find a window's reads by
        // binary search over the start positions, so the records are sorted here — stably, by position in the coordinate space —
        // and carry their places in the file along: the arms of a window are laid out in FILE order (arms_window_kernel).
        std::vector<uint32_t> perm(n);
        for (uint64_t g = 0; g < n; ++g) perm[g] = (uint32_t)g;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return out.rb[a] < out.rb[b]; });
        std::vector<uint32_t> t_rb(n), t_re(n), t_qae(n), t_ctg(n), t_coff(n + 1), t_cig(ncg ? ncg : 1);
        std::vector<uint64_t> t_soff(n);
        uint32_t acc = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t g = perm[i];
            t_rb[i] = out.rb[g]; t_re[i] = out.re[g]; t_qae[i] = out.qae[g]; t_ctg[i] = out.ctg[g]; t_soff[i] = out.seq_off[g];
            const uint32_t c0g = out.cigar_off[g], c1g = out.cigar_off[g + 1];
            t_coff[i] = acc;
            std::memcpy(t_cig.data() + acc, out.cigar + c0g, (size_t)(c1g - c0g) * 4);
            acc += c1g - c0g;
        }
        t_coff[n] = acc;
        std::memcpy(out.rb, t_rb.data(), n * 4); std::memcpy(out.re, t_re.data(), n * 4); std::memcpy(out.qae, t_qae.data(), n * 4);
        std::memcpy(out.ctg, t_ctg.data(), n * 4); std::memcpy(out.seq_off, t_soff.data(), n * 8);
        std::memcpy(out.cigar_off, t_coff.data(), (n + 1) * 4); std::memcpy(out.cigar, t_cig.data(), (size_t)acc * 4);
        std::memcpy(out.file_rank, perm.data(), n * 4);
        out.ranked = true;
    }
    return true;
}

void ReadBatch::materialize(uint32_t cid, std::vector<std::unique_ptr<Alignment>>& into) const {
    into.reserve(into.size() + count(cid));
    for (const Slice& sl : _slices) {
        const ReadChunk& ch = sl.blk->chunks[sl.chunk];
        for (uint32_t k = sl.k0; k < sl.k1; ++k) {
            if ((uint32_t)ch.cid[k] != cid) continue;
            into.emplace_back(new Alignment(ch.rb[k], ch.re[k], ch.qae[k], ch.seq.data() + ch.seq_at[k], ch.cig.data() + ch.cig_at[k], ch.cig_at[k + 1] - ch.cig_at[k]));
        }
    }
}

void ReadBatch::carry_beyond(uint32_t first_cid, ReadBatch& into) const {
    bool any = false;
    for (size_t c = first_cid; c < _per_contig.size() && !any; ++c) any = _per_contig[c] != 0;
    if (!any) return;
    auto blk = std::make_shared<ParsedBlock>();
    blk->chunks.resize(1);
    ReadChunk& d = blk->chunks[0];
    d.clear();
    for (const Slice& sl : _slices) {
        const ReadChunk& ch = sl.blk->chunks[sl.chunk];
        for (uint32_t k = sl.k0; k < sl.k1; ++k) {
            if ((uint32_t)ch.cid[k] < first_cid) continue;
            d.raw.push_back((uint32_t)d.raw.size()); d.cid.push_back(ch.cid[k]); d.rb.push_back(ch.rb[k]); d.re.push_back(ch.re[k]); d.qae.push_back(ch.qae[k]);
            d.seq.insert(d.seq.end(), ch.seq.begin() + ch.seq_at[k], ch.seq.begin() + ch.seq_at[k + 1]);
            d.cig.insert(d.cig.end(), ch.cig.begin() + ch.cig_at[k], ch.cig.begin() + ch.cig_at[k + 1]);
            d.seq_at.push_back((uint32_t)d.seq.size()); d.cig_at.push_back((uint32_t)d.cig.size());
        }
    }
    blk->n = d.raw.size();
    blk->status.assign(blk->n, ParsedBlock::ST_KEPT);
    blk->cid = d.cid;
    into.add(blk, 0, blk->n);
}

}  // namespace hypo
