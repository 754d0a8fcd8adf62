// Alignment.cpp — support voting and arm selection of one mapped read (reference: src/Alignment.cpp).
#include "Alignment.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include "Contig.hpp"

namespace hypo {

void Alignment::check_bounds(const Contig& contig, const SamRecord& rec, uint32_t rb, uint32_t re) {
    const uint32_t clen = (uint32_t)contig.get_len();
    if (rb >= clen || re > clen) {                       // Alignment.cpp:31-36 (message to stdout, exit 1)
        std::fprintf(stdout, "[Hypo::Alignment] Error: Alignment File error: Looks like the reference in the alignment file is different from the draft. "
                             "Contig (%s): Read (%s): rb (%u): re (%u): clen (%u)\n", contig.get_name().c_str(), rec.qname.c_str(), rb, re, clen);
        std::exit(1);
    }
}

Alignment::Alignment(Contig& contig, const SamRecord& rec) {
    initialise_pos(rec);
    check_bounds(contig, rec, _rb, _re);
    copy_data(rec);
}

// long read: dropped when the NM-based normalised edit distance exceeds the threshold; the reference divides the
// integers first (edit_dist*100/rlen) and takes ceil of the quotient (Alignment.cpp:51-58)
Alignment::Alignment(Contig& contig, uint64_t norm_edit_th, const SamRecord& rec) {
    initialise_pos(rec);
    check_bounds(contig, rec, _rb, _re);
    const uint32_t rlen = _re - _rb;
    if (rec.has_nm && rlen) {
        const int64_t q = rec.nm * 100 / (int64_t)rlen;
        if ((double)q > (double)norm_edit_th) is_valid = false;
    }
    if (is_valid) copy_data(rec);
}

// Alignment.cpp:514-549: reference span and the query range left after soft clips
void Alignment::initialise_pos(const SamRecord& rec) {
    _rb = rec.pos;
    _qab = 0;
    uint32_t qp = 0, rp = _rb, clip_end = 0;
    bool clip_before = true;
    for (uint32_t c : rec.cigar) {
        const uint32_t op = cigar_op(c), len = cigar_len(c);
        if (clip_before) {
            if (op == CIG_S) _qab += len;
            else if (op != CIG_H) clip_before = false;
        }
        const uint32_t t = cigar_type(op);
        if ((t & 3) == 3) { rp += len; qp += len; }
        else if (t & 2) rp += len;
        else if (t & 1) { if (!clip_before && op == CIG_S) clip_end += len; qp += len; }
    }
    _qae = qp - clip_end;
    _re = rp;
}

void Alignment::span_of(const Contig& contig, const SamRecord& rec, uint32_t& rb, uint32_t& re, uint32_t& qab, uint32_t& qae) {
    rb = rec.pos; qab = 0;
    uint32_t qp = 0, rp = rb, clip_end = 0;
    bool clip_before = true;
    for (uint32_t c : rec.cigar) {
        const uint32_t op = cigar_op(c), len = cigar_len(c);
        if (clip_before) {
            if (op == CIG_S) qab += len;
            else if (op != CIG_H) clip_before = false;
        }
        const uint32_t t = cigar_type(op);
        if ((t & 3) == 3) { rp += len; qp += len; }
        else if (t & 2) rp += len;
        else if (t & 1) { if (!clip_before && op == CIG_S) clip_end += len; qp += len; }
    }
    qae = qp - clip_end;
    re = rp;
    check_bounds(contig, rec, rb, re);
}

// Alignment.cpp:551-571: the aligned part 2-bit packed; a read with a non-ACGT base there is dropped
// (htslib's 4-bit codes: only A, C, G, T in either case map to 1, 2, 4, 8)
void Alignment::copy_data(const SamRecord& rec) {
    const uint32_t qlen = _qae - _qab;
    if ((size_t)_qab + qlen > rec.seq.size()) { is_valid = false; return; }
    if (!_apseq.assign_acgt(rec.seq.data() + _qab, qlen)) { is_valid = false; return; }
    _qae -= _qab;
    _qab = 0;
    _cigar = rec.cigar;
}

// ---- Alignment::update_solidkmers_support (Alignment.cpp:65-132) ------------------------------------------------------
void Alignment::update_solidkmers_support(unsigned k, Contig& contig) {
    const uint64_t first = contig._solid_pos.rank(_rb);
    uint64_t last = contig._solid_pos.rank(_re);
    for (uint64_t i = last; i > first; --i)                    // drop k-mers that do not lie wholly inside the read
        if (contig._solid_pos.select(i) + k <= _re) { last = i; break; }
    if (last <= first) return;
    const uint32_t n = (uint32_t)(last - first);
    // The reference keeps (kid -> index) in an unordered_multimap and walks equal_range(kmer) for every k-mer of the read: with
    // libstdc++ that is reverse insertion order (index descending), and the order matters below (pvs_supp_* state).  Entries
    // outside [left, right] are skipped without effect, and those inside have their contig k-mer within k bases of the read
    // k-mer's offset — so the same visits, in the same order, come from a window over the (position-sorted) solid k-mers of the
    // span that slides along with the read position: no map, no sort, no allocation for reads of ordinary length.
    constexpr uint32_t kStack = 512;
    uint64_t spos_stack[kStack];
    std::vector<uint64_t> spos_heap;
    uint64_t* spos = spos_stack;
    if (n > kStack) { spos_heap.resize(n); spos = spos_heap.data(); }
    for (uint32_t t = 0; t < n; ++t) {
        contig.increment_coverage((uint32_t)(first + t));
        spos[t] = contig._solid_pos.select(first + t + 1);
    }
    const uint64_t* kids = contig._kids.data() + first;
    uint64_t kmer = 0; unsigned kmer_len = 0;
    const uint64_t kmask = (1ULL << (2 * k)) - 1;
    const size_t nq = _apseq.get_seq_size();
    const uint32_t num_cbases = _re - _rb;
    int64_t pvs_supp_kpos = -1; uint32_t pvs_supp_r_bind = 0;
    uint32_t lo = 0, hi = 0;                                  // solid k-mers [lo, hi): offset within k of the read k-mer's
    for (size_t r_ind = 0; r_ind < nq; ++r_ind) {
        kmer = ((kmer << 2) | _apseq.enc_base_at(r_ind)) & kmask;
        if (kmer_len < k) ++kmer_len;
        if (kmer_len != k) continue;
        const uint32_t r_bind = (uint32_t)(r_ind + 1 - k);
        while (hi < n && (int64_t)spos[hi] - (int64_t)_rb <= (int64_t)r_bind + (int64_t)k) ++hi;
        while (lo < hi && (int64_t)spos[lo] - (int64_t)_rb + (int64_t)k < (int64_t)r_bind) ++lo;
        for (uint32_t c = hi; c-- > lo;) {
            if (kids[c] != kmer) continue;
            const uint32_t c_ind = c;
            const int64_t c_dist = (int64_t)spos[c_ind] - (int64_t)_rb;
            const uint32_t left = c_dist > (int64_t)k ? (uint32_t)(c_dist - k) : 0u;
            const uint32_t right = (uint32_t)std::min<int64_t>((int64_t)num_cbases, c_dist + (int64_t)k);
            if (r_bind < left || r_bind > right) continue;
            bool should_update = true;
            if (pvs_supp_kpos > -1 && spos[c_ind] <= (uint64_t)k + (uint64_t)pvs_supp_kpos)      // overlapping/adjacent neighbour:
                if ((uint64_t)(r_bind - pvs_supp_r_bind) != spos[c_ind] - (uint64_t)pvs_supp_kpos) should_update = false;   // offsets must agree
            if (should_update) {
                pvs_supp_kpos = (int64_t)spos[c_ind];
                pvs_supp_r_bind = r_bind;
                contig.increment_support((uint32_t)(first + c_ind));
            }
        }
    }
}

// ---- Alignment::update_minimisers_support (Alignment.cpp:134-220) -----------------------------------------------------
void Alignment::update_minimisers_support(Contig& contig) {
    const uint32_t K = Minimizer_settings.k, W = Minimizer_settings.w;
    const uint64_t first = contig._reg_pos.rank((uint64_t)_rb + 1) - 1;
    const uint64_t last = contig._reg_pos.rank(_re);
    const bool even = contig._is_win_even;
    const int64_t first_w = ((even && first % 2 == 0) || (!even && first % 2 == 1)) ? (int64_t)first : (int64_t)first + 1;
    const int64_t last_w = ((even && last % 2 == 0) || (!even && last % 2 == 1)) ? (int64_t)last : (int64_t)last - 1;
    if (last_w < first_w) return;
    {   // nothing to vote on when none of the mega-windows the read touches kept a minimizer (most are shorter than the window
        // size and have none): the read's own minimizers are then never looked at
        bool any = false;
        for (int64_t i = first_w; i <= last_w && !any; i += 2) any = !contig._minimserinfo[even ? (size_t)(i / 2) : (size_t)((i - 1) / 2)].rel_pos.empty();
        if (!any) return;
    }
    // forward-strand window minimizers of the read (position = start of the k-mer), duplicates by position removed
    const uint32_t mask = (uint32_t)((1ULL << (2 * K)) - 1);
    struct Item { uint32_t kmer, pos; };
    if (W + 1 > kMinimizerRingCap) { std::fprintf(stderr, "[Hypo] Error: minimizer window %u exceeds the queue capacity %u\n", W, kMinimizerRingCap - 1); std::exit(1); }
    Item ring[kMinimizerRingCap]; const uint32_t cap = W + 1; uint32_t head = 0, tail = cap - 1, count = 0;
    uint32_t kmer = 0, run = 0, processed = 0;
    uint32_t last_found = (uint32_t)_apseq.get_seq_size() + 1;
    std::vector<std::pair<uint32_t, uint32_t>> found;            // (minimizer, start): an unordered_multimap in the reference; only membership is used
    found.reserve(_apseq.get_seq_size() / (W > 1 ? W / 2 : 1) + 4);
    for (size_t i = 0; i < _apseq.get_seq_size(); ++i) {
        const uint8_t c = _apseq.enc_base_at(i);
        if (c >= 4) { run = 0; continue; }
        ++run;
        kmer = ((kmer << 2) | c) & mask;
        if (run < K) continue;
        while (count && ring[tail].kmer > kmer) { --count; tail = tail == 0 ? cap - 1 : tail - 1; }
        ++count; tail = (tail + 1) % cap; ring[tail] = Item{kmer, (uint32_t)i};
        while (ring[head].pos + W <= i) { head = (head + 1) % cap; --count; }
        if (++processed >= W) {
            const uint32_t start = ring[head].pos - K + 1;
            if (start != last_found) found.emplace_back(ring[head].kmer, start);
            last_found = start;
        }
    }
    std::sort(found.begin(), found.end());
    const uint16_t num_cbases = (uint16_t)(_re - _rb);          // 16-bit in the reference (Alignment.cpp:188)
    for (int64_t i = first_w; i <= last_w; i += 2) {
        const uint32_t minfoidx = even ? (uint32_t)(i / 2) : (uint32_t)((i - 1) / 2);
        MWMinimiserInfo& mi = contig._minimserinfo[minfoidx];
        const size_t num_min = mi.rel_pos.size();
        uint64_t minimiser_pos = contig._reg_pos.select((uint64_t)i + 1);
        for (uint32_t m = 0; m < num_min; ++m) {
            minimiser_pos += mi.rel_pos[m];
            const uint32_t c_dist = (uint32_t)(minimiser_pos - _rb);
            const uint32_t range_left = c_dist > 2 * K ? c_dist - 2 * K : 0u;
            const uint32_t range_right = std::min<uint16_t>(num_cbases, (uint16_t)(c_dist + 3 * K));
            if (minimiser_pos >= _rb && minimiser_pos < _re) {
                contig.increment_minimser_coverage(minfoidx, m);
                const uint32_t want = mi.minimisers[m];
                for (auto it = std::lower_bound(found.begin(), found.end(), std::make_pair(want, 0u)); it != found.end() && it->first == want; ++it)
                    if (it->second >= range_left && it->second <= range_right) contig.increment_minimser_support(minfoidx, m);
            }
            if (minimiser_pos >= _re) break;
        }
    }
}

// ---- Alignment::find_short_arms (Alignment.cpp:222-259) ---------------------------------------------------------------
void Alignment::find_short_arms(unsigned k, Contig& contig) {
    uint64_t b_ind = contig._reg_pos.rank(_rb);
    if (!contig._reg_pos[_rb]) --b_ind;
    const uint64_t e_ind = contig._reg_pos.rank(_re);
    if (e_ind - b_ind <= 1) return;
    const std::vector<uint32_t> bp = find_bp(contig._reg_pos, contig._reg_type, (uint32_t)b_ind, (uint32_t)e_ind);
    auto is_sr = [&](uint64_t r) { return contig._reg_type[r] == RegionType::SR || contig._reg_type[r] == RegionType::MSR; };
    ArmType at = contig._reg_pos[_rb] ? ArmType::INTERNAL : ArmType::SUFFIX;
    if (!is_sr(b_ind)) prepare_short_arm(k, (uint32_t)b_ind, _qab, bp[0], at, contig);
    uint32_t bi = 0;
    for (uint64_t ind = b_ind + 1; ind < e_ind - 1; ++ind, ++bi) {
        if (is_sr(ind)) continue;
        if (bp[bi + 1] == bp[bi]) _arms.emplace_back((uint32_t)ind);
        else prepare_short_arm(k, (uint32_t)ind, bp[bi], bp[bi + 1], ArmType::INTERNAL, contig);
    }
    at = contig._reg_pos[_re] ? ArmType::INTERNAL : ArmType::PREFIX;
    if (!is_sr(e_ind - 1)) prepare_short_arm(k, (uint32_t)(e_ind - 1), bp[bi], _qae, at, contig);
}

// ---- Alignment::find_long_arms (Alignment.cpp:262-299) ----------------------------------------------------------------
void Alignment::find_long_arms(Contig& contig) {
    uint64_t b_ind = contig._pseudo_reg_pos.rank(_rb);
    if (!contig._pseudo_reg_pos[_rb]) --b_ind;
    const uint64_t e_ind = contig._pseudo_reg_pos.rank(_re);
    if (e_ind - b_ind <= 1) return;
    const std::vector<uint32_t> bp = find_bp(contig._pseudo_reg_pos, contig._pseudo_reg_type, (uint32_t)b_ind, (uint32_t)e_ind);
    auto is_sr = [&](uint64_t r) { return contig._pseudo_reg_type[r] == RegionType::SR; };
    ArmType at = contig._pseudo_reg_pos[_rb] ? ArmType::INTERNAL : ArmType::SUFFIX;
    if (!is_sr(b_ind)) _arms.emplace_back(contig._true_reg_id[b_ind], _apseq, _qab, bp[0], at);
    uint32_t bi = 0;
    for (uint64_t ind = b_ind + 1; ind < e_ind - 1; ++ind, ++bi) {
        if (is_sr(ind)) continue;
        if (bp[bi + 1] == bp[bi]) _arms.emplace_back(contig._true_reg_id[ind]);
        else _arms.emplace_back(contig._true_reg_id[ind], _apseq, bp[bi], bp[bi + 1], ArmType::INTERNAL);
    }
    at = contig._pseudo_reg_pos[_re] ? ArmType::INTERNAL : ArmType::PREFIX;
    if (!is_sr(e_ind - 1)) _arms.emplace_back(contig._true_reg_id[e_ind - 1], _apseq, bp[bi], _qae, at);
}

void Alignment::add_arms(const Contig& contig) {                     // Alignment.cpp:301-318
    for (const Arm& a : _arms) {
        Window* w = contig.window(a.windex);
        if (a.armtype == ArmType::PREFIX) w->add_prefix(a.arm);
        else if (a.armtype == ArmType::SUFFIX) w->add_suffix(a.arm);
        else if (a.armtype == ArmType::INTERNAL) w->add_internal(a.arm);
        else w->add_empty();
    }
    _arms.clear();
}

void Alignment::add_arms(const Contig& contig, uint32_t w0, uint32_t w1) {
    for (Arm& a : _arms) {
        if (a.windex < w0 || a.windex >= w1) continue;
        Window* w = contig.window(a.windex);
        if (a.armtype == ArmType::PREFIX) w->add_prefix(std::move(a.arm));
        else if (a.armtype == ArmType::SUFFIX) w->add_suffix(std::move(a.arm));
        else if (a.armtype == ArmType::INTERNAL) w->add_internal(std::move(a.arm));
        else w->add_empty();
    }
}

// ---- Alignment::find_bp (Alignment.cpp:321-406): query positions where the read crosses region borders -----------------
// An op that ends exactly on a border defers the decision (`corner`): a following M/D emits the current query position, a
// following insertion goes to the right-hand window if the region on the left is an SR/MSR and to the left-hand one otherwise.
std::vector<uint32_t> Alignment::find_bp(const BitVec& reg_pos, const std::vector<RegionType>& reg_type, uint32_t beg_ind, uint32_t end_ind) const {
    std::vector<uint32_t> res;
    uint32_t ref_pos = _rb, cur = beg_ind + 1, query_pos = 0;
    uint32_t next_ref = (uint32_t)reg_pos.select((uint64_t)cur + 1);
    bool corner = false;
    auto advance = [&]() { ++cur; next_ref = (uint32_t)reg_pos.select((uint64_t)cur + 1); };
    for (uint32_t c : _cigar) {
        const uint32_t op = cigar_op(c);
        uint32_t len = cigar_len(c);
        if (op == CIG_S || op == CIG_H) continue;
        const uint32_t t = cigar_type(op);
        if ((t & 3) == 3 || (t & 2)) {
            const bool both = (t & 3) == 3;
            if (corner) { res.push_back(query_pos); corner = false; advance(); }
            while (ref_pos + len >= next_ref && !corner) {
                const uint32_t d = next_ref - ref_pos;
                ref_pos = next_ref;
                if (both) query_pos += d;
                len -= d;
                if (len > 0) { res.push_back(query_pos); advance(); }
                else corner = true;
            }
            if (len > 0) { ref_pos += len; if (both) query_pos += len; }
        } else if (t & 1) {
            if (corner) {
                const RegionType lt = reg_type[cur - 1];
                res.push_back((lt == RegionType::SR || lt == RegionType::MSR) ? query_pos : query_pos + len);
                advance();
                corner = false;
            }
            query_pos += len;
        }
        if (cur == end_ind) break;
    }
    return res;
}

// ---- Alignment::prepare_short_arm (Alignment.cpp:408-511) --------------------------------------------------------------
void Alignment::prepare_short_arm(unsigned k, uint32_t windex, uint32_t qb, uint32_t qe, ArmType armtype, Contig& contig) {
    const uint32_t mk = Minimizer_settings.k;
    const uint64_t curr_pos = contig._reg_pos.select((uint64_t)windex + 1), next_pos = contig._reg_pos.select((uint64_t)windex + 2);
    if (next_pos - curr_pos > (uint64_t)Arms_settings.short_arm_coef * (uint64_t)(qe - qb)) return;
    const RegionType wt = contig._reg_type[windex];
    bool valid = true;
    uint32_t q_beg = qb, q_end = qe;
    size_t hit = 0;
    if ((wt == RegionType::SWS || wt == RegionType::SW || wt == RegionType::SWM) && armtype != ArmType::SUFFIX) {      // SR on the left
        if (q_beg < k) valid = false;
        else {
            const uint64_t anchor = contig._anchor_kmers[(size_t)contig._reg_info[windex - 1] << 1];                  // last k-mer of that SR
            if (!_apseq.check_kmer(anchor, k, q_beg - k)) {
                const uint32_t s = q_beg < 2 * k ? 0 : q_beg - 2 * k, e = q_end < q_beg + k ? q_end : q_beg + k;
                if (_apseq.find_kmer(anchor, k, s, e, false, hit)) q_beg = (uint32_t)hit + k; else valid = false;
            }
        }
    }
    if ((wt == RegionType::SWS || wt == RegionType::WS || wt == RegionType::MWS) && armtype != ArmType::PREFIX) {       // SR on the right
        if (q_end + k > _qae) valid = false;
        else {
            const uint64_t anchor = contig._anchor_kmers[((size_t)contig._reg_info[windex + 1] << 1) - 1];            // first k-mer of that SR
            if (!_apseq.check_kmer(anchor, k, q_end)) {
                const uint32_t s = q_end < q_beg + k ? q_beg : q_end - k, e = std::min(_qae, q_end + 2 * k);
                if (_apseq.find_kmer(anchor, k, s, e, true, hit)) q_end = (uint32_t)hit; else valid = false;
            }
        }
    }
    if ((wt == RegionType::MWM || wt == RegionType::MW || wt == RegionType::MWS) && armtype != ArmType::SUFFIX) {       // minimizer on the left
        if (q_beg < mk) valid = false;
        else {
            const uint32_t mn = contig._reg_info[windex - 1];
            if (!_apseq.check_kmer(mn, mk, q_beg - mk)) {
                const uint32_t s = q_beg < 3 * mk ? 0 : q_beg - 3 * mk, e = q_end < q_beg + 2 * mk ? q_end : q_beg + 2 * mk;
                if (_apseq.find_kmer(mn, mk, s, e, false, hit)) q_beg = (uint32_t)hit + mk; else valid = false;
            }
        }
    }
    if ((wt == RegionType::MWM || wt == RegionType::WM || wt == RegionType::SWM) && armtype != ArmType::PREFIX) {       // minimizer on the right
        if (q_end + mk > _qae) valid = false;
        else {
            const uint32_t mn = contig._reg_info[windex + 1];
            if (!_apseq.check_kmer(mn, mk, q_end)) {
                const uint32_t s = q_end < q_beg + 2 * mk ? q_beg : q_end - 2 * mk, e = std::min(_qae, q_end + 3 * mk);
                if (_apseq.find_kmer(mn, mk, s, e, true, hit)) q_end = (uint32_t)hit; else valid = false;
            }
        }
    }
    if (valid && q_beg < q_end) _arms.emplace_back(windex, _apseq, q_beg, q_end, armtype);
}

}  // namespace hypo
