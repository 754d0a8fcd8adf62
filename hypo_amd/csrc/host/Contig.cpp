// Contig.cpp — strong-region detection, minimizer-anchored window division, arm filling/pruning, long-window
// merging and FASTA re-assembly (reference: src/Contig.cpp).  Written from the behaviour described in
// SURVEY.md Appendix A.4/A.5 and the cited lines; data lives in flat vectors and BitVec instead of sdsl objects and
// per-k-mer heap nodes.
#include "Contig.hpp"
#include <atomic>
#include <omp.h>
#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <ostream>
#include <unordered_map>

namespace hypo {

bool Contig::_no_long_reads = false;

bool SolidKmers::load(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    uint64_t nbits = 0;
    f.read((char*)&nbits, 8);
    if (!f || nbits != (1ULL << (2 * k))) return false;
    f.close();
    // (2 GiB at k = 17: the payload is read in pieces on all threads, each piece into its own — first touched — part of the vector,
    // and counted there; one stream read + a serial popcount were 1.1 s of the 3 Gbp run)
    const size_t nwords = (size_t)((nbits + 63) / 64);
    words.resize(nwords);
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    const size_t piece = (size_t)1 << 20;                  // words per piece (8 MiB)
    const int64_t np = (int64_t)((nwords + piece - 1) / piece);
    uint64_t ones = 0;
    bool ok = true;
#pragma omp parallel for schedule(static) reduction(+ : ones) reduction(&& : ok)
    for (int64_t i = 0; i < np; ++i) {
        const size_t a = (size_t)i * piece, b = std::min(nwords, a + piece);
        size_t got = 0; const size_t want = (b - a) * 8;
        while (got < want) {
            const ssize_t r = ::pread(fd, (char*)(words.data() + a) + got, want - got, (off_t)(8 + a * 8 + got));
            if (r <= 0) { ok = false; break; }
            got += (size_t)r;
        }
        for (size_t w = a; w < b; ++w) ones += (uint64_t)__builtin_popcountll(words[w]);
    }
    ::close(fd);
    if (!ok) return false;
    num_solid = ones / 2;          // both strands are set per canonical k-mer (SolidKmers.cpp:182-186); reported only
    return true;
}
bool SolidKmers::store(const std::string& path) const {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const uint64_t nbits = 1ULL << (2 * k);
    f.write((const char*)&nbits, 8);
    f.write((const char*)words.data(), (std::streamsize)(words.size() * 8));
    return (bool)f;
}

// Contig.cpp:30-36: the name is the first token of the header; _reg_pos has one extra (dummy) bit
Contig::Contig(uint32_t id, const std::string& name, const std::string& seq)
    : _id(id), _name(name.substr(0, name.find_first_of(" \t"))), _len((uint32_t)seq.size()), _pseq(seq),
      _solid_pos(seq.size()), _reg_pos(seq.size() + 1) {}

// ---- Contig::find_solid_pos (src/Contig.cpp:40-74): on the device -------------------------------------------------
// `set_on_device`: the caller has sent sk's bit set to the device with hypo_gpu_solid_set_upload (once per run)
int Contig::find_solid_pos(const SolidKmers& sk, bool set_on_device) {
    const uint64_t nw = ((uint64_t)_len + 63) / 64;
    std::vector<uint64_t> words(nw ? nw : 1), rank(nw + 1);
    _scan_k = sk.get_k();
    // Round 4: the k-mer ids and positions stay on the device for the support votes (hypo_gpu_solid_scan_keep); the host takes the
    // mark bits and their rank directory only.  HYPO_SCAN_KEEP=0, a library without the entry point (the CPU test shim) or several
    // device contexts (a kept scan lives on the context that made it): the ids come back as before.
    if (set_on_device && hypo_gpu_num_devices() == 1 && !(std::getenv("HYPO_SCAN_KEEP") && std::atoi(std::getenv("HYPO_SCAN_KEEP")) == 0)) {
        uint64_t ns = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = hypo_gpu_solid_scan_keep(_id, _pseq.data(), _len, sk.get_k(), words.data(), rank.data(), &ns);
        if (rc == HYPO_OK) {
            _solid_pos = BitVec(_len);
            std::memcpy(_solid_pos.data(), words.data(), nw * 8);
            _solid_pos.adopt_rank(std::move(rank));
            _kids.clear(); _n_solid = ns; _scan_kept = true;
            _kcov.assign(ns, 0); _ksup.assign(ns, 0);
            if (std::getenv("HYPO_HOST_TIMING"))
                std::fprintf(stderr, "[timing] find_solid_pos: hypo_gpu_solid_scan_keep %.3f s (%llu marked positions stay on the device)\n",
                             std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), (unsigned long long)ns);
            return HYPO_OK;
        }
        if (rc != HYPO_E_UNSUPPORTED) return rc;
        rank.assign(nw + 1, 0);
    }
    const uint64_t cap = _len ? _len : 1;                             // at most one k-mer id per position; only n of them are written
    std::unique_ptr<uint64_t[]> kids(new uint64_t[cap]);              // (not zero-filled: 8 bytes per base)
    uint64_t n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = hypo_gpu_solid_scan(_pseq.data(), _len, sk.get_k(), set_on_device ? nullptr : sk.words.data(), words.data(), kids.get(), cap,
                                       rank.data(), &n);
    if (rc != HYPO_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    adopt_solid_scan(words.data(), rank.data(), kids.get(), n);
    if (std::getenv("HYPO_HOST_TIMING"))
        std::fprintf(stderr, "[timing] find_solid_pos: hypo_gpu_solid_scan %.3f s, adoption %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    return HYPO_OK;
}
void Contig::adopt_solid_scan(const uint64_t* words, const uint64_t* rank, const uint64_t* kids, uint64_t n_solid) {
    const uint64_t nw = ((uint64_t)_len + 63) / 64;
    _solid_pos = BitVec(_len);
    for (uint64_t w = 0; w < nw; ++w) _solid_pos.data()[w] = words[w];
    if (rank) _solid_pos.adopt_rank(std::vector<uint64_t>(rank, rank + nw + 1)); else _solid_pos.init_support();
    _kids.assign(kids, kids + n_solid);
    _n_solid = n_solid; _scan_kept = false;
    _kcov.assign(n_solid, 0);
    _ksup.assign(n_solid, 0);
}

void Contig::ensure_kids() {
    if (!_kids.empty() || !_n_solid) return;
    _kids.resize(_n_solid);
#pragma omp parallel for schedule(static, 4096)
    for (int64_t i = 0; i < (int64_t)_n_solid; ++i) _kids[(size_t)i] = kmer_at(_solid_pos.select((uint64_t)i + 1), _scan_k);
}

// Which branches of the segmentation a run reached (printed by Hypo::polish under HYPO_STAGE_COUNTERS=1; the manifests of the round-6
// goldens quote them): [0] solid k-mers accepted with 40-80 % support (src/Contig.cpp:104-109), [1] such k-mers refused because the
// one before was one too, [2] Contig::force_divide calls (src/Contig.cpp:630-711), [3] minimizers dropped because they recur in their
// mega-window, [4] poly-base minimizers dropped (src/Contig.cpp:509-511)
std::atomic<uint64_t> g_stage_counters[5];

// ---- Contig::prepare_for_division (src/Contig.cpp:75-185) -----------------------------------------------------------
void Contig::prepare_for_division(unsigned k) {
    std::vector<uint32_t> sr_pos, sr_len;
    _anchor_kmers.clear();
    _anchor_kmers.push_back(0);                                  // index 0 is a dummy
    uint32_t last_kind = 0, first_kind = 0;
    uint64_t last_sr_pos = 0, first_sr_pos = 0;
    bool in_sr = false, pvs_80 = true;
    const bool have_kids = !_kids.empty();
    (void)first_kind; (void)last_kind;
    uint32_t i = 0;
    auto close_sr = [&] {
        sr_pos.push_back((uint32_t)first_sr_pos); sr_len.push_back((uint32_t)(last_sr_pos - first_sr_pos));
        // (the k-mer ids of a resident scan are on the device: the id of a marked position is the k bases that start there;
        // the last valid k-mer of the SR starts at last_sr_pos - k)
        _anchor_kmers.push_back(have_kids ? _kids[first_kind] : kmer_at(first_sr_pos, k));
        _anchor_kmers.push_back(have_kids ? _kids[last_kind] : kmer_at(last_sr_pos - k, k));
        in_sr = false; pvs_80 = true;
    };
    // The reference walks every position (src/Contig.cpp:96-150); only marked ones change the state, and an SR ends at the first
    // position == last_sr_pos that no valid k-mer has moved on: the marked positions are taken out of the words of the bit vector
    // one by one (a 250 Mbp contig: 45 M of them), an open SR that ends before the next one is closed first.
    const uint64_t* const words = _solid_pos.data();
    const uint64_t n_words = _solid_pos.n_words();
    for (uint64_t wi = 0; wi < n_words; ++wi) {
        uint64_t word = words[wi];
        while (word) {
            const uint32_t pos = (uint32_t)(wi * 64 + (uint64_t)__builtin_ctzll(word));
            word &= word - 1;
            if (pos >= _len) break;
            if (in_sr && last_sr_pos < pos) close_sr();
            bool is_valid = false;
            const uint32_t cov = _kcov[i] & 0xffffu, sup = _ksup[i] & 0xffffu;
            if (cov >= Sr_settings.cov_th) {
                const uint32_t supp_th = (uint32_t)(Sr_settings.supp_frac * cov);
                if (sup >= 2 * supp_th) { is_valid = true; pvs_80 = true; }
                else if (sup >= supp_th) { if (pvs_80) is_valid = true; pvs_80 = false; g_stage_counters[is_valid ? 0 : 1].fetch_add(1, std::memory_order_relaxed); }
            }
            if (is_valid) {
                if (!in_sr) { first_kind = i; first_sr_pos = pos; in_sr = true; }
                last_kind = i;
                last_sr_pos = (uint64_t)pos + k;
            }
            ++i;
            if (in_sr && pos == last_sr_pos) close_sr();
        }
    }
    if (in_sr && last_sr_pos < _len) close_sr();                 // (an SR that ends before the contig does, behind the last marked position)
    if (in_sr) {
        sr_pos.push_back((uint32_t)first_sr_pos); sr_len.push_back((uint32_t)(last_sr_pos - first_sr_pos));
        _anchor_kmers.push_back(have_kids ? _kids[first_kind] : kmer_at(first_sr_pos, k));
        _anchor_kmers.push_back(have_kids ? _kids[last_kind] : kmer_at(last_sr_pos - k, k));
    }
    std::vector<uint64_t>().swap(_kids); std::vector<uint32_t>().swap(_kcov); std::vector<uint32_t>().swap(_ksup);
    _solid_pos.clear();

    _numSR = sr_pos.size();
    int acc = 0;                                                 // std::accumulate(..., 0): int accumulator
    for (uint32_t l : sr_len) acc += (int)l;
    _lenSR = (uint64_t)acc;

    // strong regions and the mega-windows between them
    _is_win_even = !(_numSR > 0 && sr_pos[0] == 0);
    _minimserinfo.clear();
    _minimserinfo.reserve(_numSR + 1);
    _reg_pos.set(0);
    sr_pos.push_back(_len);                                      // dummy SR at the end
    _reg_pos.set(_len);
    // region borders first (bit sets into shared words: serial), then the minimizers of every mega-window on all threads
    // (each mega-window fills its own MWMinimiserInfo; the reference does both in one serial loop)
    const uint32_t first_w = _is_win_even ? 1u : 0u;
    _minimserinfo.resize((size_t)_numSR + first_w);
    for (uint32_t ind = 0; ind < _numSR; ++ind) { _reg_pos.set(sr_pos[ind]); _reg_pos.set(sr_pos[ind] + sr_len[ind]); }
    if (_is_win_even && sr_pos[0] > Window_settings.ideal_swind_size) initialise_minimserinfo(_pseq.unpack(0, sr_pos[0]), 0);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t ii = 0; ii < (int64_t)_numSR; ++ii) {
        const uint32_t ind = (uint32_t)ii;
        const uint32_t mw_start = sr_pos[ind] + sr_len[ind];
        const uint32_t mw_len = sr_pos[ind + 1] - mw_start;
        if (mw_len > Window_settings.ideal_swind_size) initialise_minimserinfo(_pseq.unpack(mw_start, sr_pos[ind + 1]), ind + first_w);
    }
    _reg_pos.init_support();
    _mreg_ready = true;
}

// ---- Contig::initialise_minimserinfo (src/Contig.cpp:455-524) -------------------------------------------------------
// Forward-strand (k=10, w=10) window minimizers of a mega-window; only those occurring once and not poly-base are kept,
// positions relative to the previous kept one.  The rolling k-mer is 32 bits wide and is not reset by an N.
void Contig::initialise_minimserinfo(const std::string& draft_seq, uint32_t minfoind) {
    const uint32_t K = Minimizer_settings.k, W = Minimizer_settings.w;
    const uint32_t mask = (uint32_t)((1ULL << (2 * K)) - 1);
    struct Item { uint32_t kmer, pos; };
    if (W + 1 > kMinimizerRingCap) { std::fprintf(stderr, "[Hypo] Error: minimizer window %u exceeds the queue capacity %u\n", W, kMinimizerRingCap - 1); std::exit(1); }
    Item ring[kMinimizerRingCap]; uint32_t cap = W + 1, head = 0, tail = cap - 1, count = 0;
    uint32_t kmer = 0, run = 0, processed = 0;
    uint32_t last_found_position = (uint32_t)draft_seq.size() + 1;
    std::vector<uint32_t> found, found_pos;
    std::unordered_map<uint32_t, uint8_t> counter;
    for (size_t i = 0; i < draft_seq.size(); ++i) {
        const uint8_t c = nt4((unsigned char)draft_seq[i]);
        if (c >= 4) { run = 0; continue; }
        ++run;
        kmer = ((kmer << 2) | c) & mask;
        if (run < K) continue;
        while (count && ring[tail].kmer > kmer) { --count; tail = tail == 0 ? cap - 1 : tail - 1; }
        ++count; tail = (tail + 1) % cap; ring[tail] = Item{kmer, (uint32_t)i};
        while (ring[head].pos + W <= i) { head = (head + 1) % cap; --count; }
        if (++processed >= W) {
            const uint32_t start = ring[head].pos - K + 1;
            if (start != last_found_position) { found_pos.push_back(start); found.push_back(ring[head].kmer); ++counter[ring[head].kmer]; }
            last_found_position = start;
        }
    }
    MWMinimiserInfo& mi = _minimserinfo[minfoind];
    last_found_position = 0;
    for (size_t i = 0; i < found.size(); ++i) {
        if (counter[found[i]] != 1) { g_stage_counters[3].fetch_add(1, std::memory_order_relaxed); continue; }
        const uint32_t m = found[i];
        if (m == Minimizer_settings.polyA || m == Minimizer_settings.polyC || m == Minimizer_settings.polyG || m == Minimizer_settings.polyT) { g_stage_counters[4].fetch_add(1, std::memory_order_relaxed); continue; }
        mi.minimisers.push_back(m);
        mi.rel_pos.push_back(found_pos[i] - last_found_position);
        last_found_position = found_pos[i];
    }
    mi.support.assign(mi.rel_pos.size(), 0);
    mi.coverage.assign(mi.rel_pos.size(), 0);
}

// ---- Contig::divide_into_regions (src/Contig.cpp:187-245) -----------------------------------------------------------
void Contig::divide_into_regions() {
    uint32_t sr_rank = 1, reg_start = 0, reg_ind = 0;
    // (the borders set so far, out of the words of the bit vector: divide() adds borders inside the region in hand only, which lie
    // behind the walk — the reference tests all _len + 1 positions)
    const uint64_t n_words = _reg_pos.n_words();
    for (uint64_t wi = 0; wi < n_words; ++wi) {
      uint64_t word = _reg_pos.data()[wi];
      if (wi == 0) word &= ~1ULL;                                  // (position 0 opens the first region)
      while (word) {
        const uint64_t i = wi * 64 + (uint64_t)__builtin_ctzll(word);
        word &= word - 1;
        if (i > _len) break;
        const uint32_t reg_end = (uint32_t)i;
        if ((_is_win_even && reg_ind % 2 == 0) || (!_is_win_even && reg_ind % 2 == 1)) {
            const char pvs = reg_ind == 0 ? 'n' : 's';
            const char nxt = i == _len ? 'n' : 's';
            divide(reg_ind, reg_start, reg_end, pvs, nxt);
        } else {
            _reg_info.push_back(sr_rank++);
            _reg_type.push_back(RegionType::SR);
        }
        ++reg_ind;
        reg_start = reg_end;
      }
    }
    _reg_type.push_back(RegionType::SR);                         // dummy
    std::vector<MWMinimiserInfo>().swap(_minimserinfo);
    _mreg_ready = false;
    _reg_pos.init_support();
    // the windows: region i = [border i + 1, border i + 2) in rank terms; the borders are listed once and the Window objects (each
    // with its own copy of the draft stretch) are made on all threads when the contig has the team to itself
    const size_t n_reg = _reg_type.size();
    std::vector<uint32_t> border;
    border.reserve(n_reg + 1);
    for (uint64_t wi = 0; wi < n_words; ++wi) {
        uint64_t word = _reg_pos.data()[wi];
        while (word) { border.push_back((uint32_t)(wi * 64 + (uint64_t)__builtin_ctzll(word))); word &= word - 1; }
    }
    _pwindows.clear();
    _pwindows.resize(n_reg);
#pragma omp parallel for schedule(static)
    for (int64_t ii = 0; ii < (int64_t)n_reg; ++ii) {
        const size_t i = (size_t)ii;
        if (_reg_type[i] == RegionType::SR || _reg_type[i] == RegionType::MSR || i + 1 >= border.size()) continue;
        _pwindows[i].reset(new Window(_pseq, border[i], border[i + 1], WindowType::SHORT));
    }
}

// ---- Contig::divide (src/Contig.cpp:526-628) --------------------------------------------------------------------------
void Contig::divide(uint32_t reg_index, uint32_t beg, uint32_t end, char pvs, char nxt) {
    const uint32_t IDEAL = Window_settings.ideal_swind_size, MK = Minimizer_settings.k, TOO_LARGE = 2 * IDEAL;
    const uint32_t minfoidx = _is_win_even ? reg_index / 2 : (reg_index - 1) / 2;
    const MWMinimiserInfo& mi = _minimserinfo[minfoidx];
    const size_t num_min = mi.rel_pos.size();
    uint32_t minimiser_pos = beg;
    std::vector<uint32_t> supp_pos, supp_min;
    for (size_t m = 0; m < num_min; ++m) {
        minimiser_pos += mi.rel_pos[m];
        const uint32_t cov = mi.coverage[m] & 0xffffu, sup = mi.support[m] & 0xffffu;
        if (cov >= Minimizer_settings.cov_th) {
            const uint32_t supp_th = (uint32_t)(Minimizer_settings.supp_frac * cov);
            if (sup >= supp_th && minimiser_pos + MK < end) { supp_pos.push_back(minimiser_pos); supp_min.push_back(mi.minimisers[m]); }
        }
    }
    uint32_t remaining = end - beg, start = beg;
    std::vector<uint32_t> cut;
    for (uint32_t m = 0; m < supp_pos.size() && remaining > IDEAL; ++m) {
        const bool should_break = (m == supp_pos.size() - 1) ? true : (supp_pos[m + 1] > IDEAL + start);
        if (should_break && supp_pos[m] > start) { cut.push_back(m); start = supp_pos[m] + MK; remaining = end - start; }
    }
    auto plain = [&](RegionType t, uint32_t at) { _reg_pos.set(at); _reg_info.push_back(0); _reg_type.push_back(t); };
    auto msr = [&](uint32_t m) { _reg_pos.set(supp_pos[m]); _reg_info.push_back(supp_min[m]); _reg_type.push_back(RegionType::MSR); };
    const uint32_t nmw = (uint32_t)cut.size();
    if (nmw == 0) {
        if (end > beg + TOO_LARGE) force_divide(beg, end, pvs, nxt);
        else plain(pvs == 's' && nxt == 's' ? RegionType::SWS : pvs == 's' ? RegionType::SW : nxt == 's' ? RegionType::WS : RegionType::OTHER, beg);
        return;
    }
    const uint32_t first_end = supp_pos[cut[0]];
    if (first_end > beg + TOO_LARGE) force_divide(beg, first_end, pvs, 'm');
    else plain(pvs == 's' ? RegionType::SWM : RegionType::WM, beg);
    for (uint32_t c = 1; c < nmw; ++c) {
        const uint32_t pm = cut[c - 1];
        msr(pm);
        const uint32_t ws = supp_pos[pm] + MK, we = supp_pos[cut[c]];
        if (we > TOO_LARGE + ws) force_divide(ws, we, 'm', 'm');
        else plain(RegionType::MWM, ws);
    }
    const uint32_t pm = cut[nmw - 1];
    msr(pm);
    const uint32_t ws = supp_pos[pm] + MK;
    if (end > TOO_LARGE + ws) force_divide(ws, end, 'm', nxt);
    else plain(nxt == 's' ? RegionType::MWS : RegionType::MW, start);      // start == ws here (Contig.cpp:622)
}

// ---- Contig::force_divide (src/Contig.cpp:630-711) --------------------------------------------------------------------
void Contig::force_divide(uint32_t beg, uint32_t end, char pvs, char nxt) {
    g_stage_counters[2].fetch_add(1, std::memory_order_relaxed);
    uint32_t start = beg, remaining = end - start;
    std::vector<uint32_t> cut_pos;
    while (remaining > Window_settings.ideal_swind_size) {
        uint32_t s = start + Window_settings.wind_size_search_th;
        while (s < end) {                                             // no homopolymer across the cut: ..AAB | CDD..
            const uint8_t base = _pseq.enc_base_at(s);
            if (base == _pseq.enc_base_at(s - 1)) s += 1;
            else if (s + 1 < end && base == _pseq.enc_base_at(s + 1)) s += 2;
            else if (s + 2 < end && _pseq.enc_base_at(s + 2) == _pseq.enc_base_at(s + 1)) s += 3;
            else break;
        }
        if (s < end) { cut_pos.push_back(start); start = s + 1; remaining = end - start; }
        else break;
    }
    if (start < end) cut_pos.push_back(start);
    auto add = [&](RegionType t, uint32_t at) { _reg_pos.set(at); _reg_info.push_back(0); _reg_type.push_back(t); };
    const uint32_t nw = (uint32_t)cut_pos.size();
    if (nw == 1) {
        RegionType t = RegionType::OTHER;
        if (pvs == 's' && nxt == 's') t = RegionType::SWS;
        else if (pvs == 's' && nxt == 'm') t = RegionType::SWM;
        else if (pvs == 's' && nxt == 'n') t = RegionType::SW;
        else if (pvs == 'm' && nxt == 's') t = RegionType::MWS;
        else if (pvs == 'm' && nxt == 'm') t = RegionType::MWM;
        else if (pvs == 'm' && nxt == 'n') t = RegionType::MW;
        else if (pvs == 'n' && nxt == 's') t = RegionType::WS;
        // (pvs 'n', nxt 'm') stays OTHER: the reference tests `nxt=='n' && nxt=='m'` (Contig.cpp:684)
        add(t, beg);
        return;
    }
    add(pvs == 's' ? RegionType::SW : pvs == 'm' ? RegionType::MW : RegionType::OTHER, beg);
    for (uint32_t i = 1; i + 1 < nw; ++i) add(RegionType::OTHER, cut_pos[i]);
    add(nxt == 's' ? RegionType::WS : nxt == 'm' ? RegionType::WM : RegionType::OTHER, cut_pos[nw - 1]);
}

// The reference hands every alignment's arms to their windows one alignment after the other (serial inside a contig:
// the order of the arms inside a window is the order of the records and it decides the POA result).  Here every thread
// owns a contiguous range of windows and walks ALL alignments in record order, taking only the arms of its own windows:
// same per-window order, no locks; the (first, last) window index of every alignment is tabulated first so that the
// walk touches an alignment's arm list only when it overlaps the range.
void Contig::add_arms_by_window_range(std::vector<std::unique_ptr<Alignment>>& alignments) {
    const int64_t n = (int64_t)alignments.size();
    const uint32_t nreg = (uint32_t)_pwindows.size();
    std::vector<uint32_t> first((size_t)n), last((size_t)n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        uint32_t f, l;
        if (alignments[(size_t)i]->arm_span(f, l)) { first[(size_t)i] = f; last[(size_t)i] = l; }
        else { first[(size_t)i] = 1; last[(size_t)i] = 0; }           // no arms: overlaps nothing
    }
#pragma omp parallel
    {
        const uint32_t nt = (uint32_t)omp_get_num_threads(), t = (uint32_t)omp_get_thread_num();
        const uint32_t w0 = (uint32_t)((uint64_t)nreg * t / nt), w1 = (uint32_t)((uint64_t)nreg * (t + 1) / nt);
        if (w0 < w1)
            for (int64_t i = 0; i < n; ++i)
                if (first[(size_t)i] < w1 && last[(size_t)i] >= w0 && first[(size_t)i] <= last[(size_t)i]) alignments[(size_t)i]->add_arms(*this, w0, w1);
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) alignments[(size_t)i].reset();
}

// ---- Contig::fill_short_windows (src/Contig.cpp:249-289) --------------------------------------------------------------
void Contig::fill_short_windows(std::vector<std::unique_ptr<Alignment>>& alignments) {
    add_arms_by_window_range(alignments);                              // arm order inside a window = BAM order
    std::vector<uint64_t>().swap(_anchor_kmers);
    std::vector<uint32_t>().swap(_reg_info);
#pragma omp parallel for schedule(static)
    for (int64_t ii = 0; ii < (int64_t)_reg_type.size(); ++ii) {
        const size_t i = (size_t)ii;
        if (_reg_type[i] == RegionType::SR || _reg_type[i] == RegionType::MSR || !_pwindows[i]) continue;
        Window& w = *_pwindows[i];
        bool discarded = false;
        const uint64_t internal = w.get_num_internal();
        if (internal < Arms_settings.min_short_num) {
            const uint32_t win_len = (uint32_t)(_reg_pos.select(i + 2) - _reg_pos.select(i + 1));
            const bool covered = w.get_maxlen_pre() + w.get_maxlen_suf() >= win_len;
            const bool enough = w.get_num_pre() >= Arms_settings.min_short_num && w.get_num_suf() >= Arms_settings.min_short_num;
            if (!(covered && enough)) { _pwindows[i].reset(); discarded = true; }
        }
        if (!discarded) {
            const uint64_t contrib = w.get_num_total();
            const bool c0 = internal > Arms_settings.min_internal_num1;
            const bool c1 = contrib >= Arms_settings.min_contrib && (double)internal >= std::floor(Arms_settings.min_internal_contrib * (double)contrib);
            const RegionType t = _reg_type[i];
            const bool c2 = (t == RegionType::SWS || t == RegionType::SW || t == RegionType::WS || t == RegionType::MWS || t == RegionType::SWM) &&
                            internal >= Arms_settings.min_internal_num2;
            if (c0 || c1 || c2) w.clear_pre_suf();
        }
    }
}

// ---- Contig::prepare_long_windows (src/Contig.cpp:292-343) ------------------------------------------------------------
void Contig::prepare_long_windows() {
    _pseudo_reg_pos = BitVec((uint64_t)_len + 1);
    const size_t num_reg = _reg_type.size();
    _pseudo_reg_type.clear(); _true_reg_id.clear();
    bool pvs_iswin = true;
    uint32_t cur_len = 0;
    // (region i starts at the (i + 1)-th border: the borders are listed once instead of two select() calls per region — 7 M regions
    // on a 250 Mbp contig)
    std::vector<uint32_t> border;
    border.reserve(num_reg + 1);
    for (uint64_t wi = 0; wi < _reg_pos.n_words(); ++wi) {
        uint64_t word = _reg_pos.data()[wi];
        while (word) { border.push_back((uint32_t)(wi * 64 + (uint64_t)__builtin_ctzll(word))); word &= word - 1; }
    }
    border.push_back(_len);                                       // (never read for a well-formed table: the last region is the SR end marker)
    for (uint32_t i = 0; i < num_reg; ++i) {
        const uint32_t pos = border[i];
        if (_reg_type[i] == RegionType::SR || _reg_type[i] == RegionType::MSR || _pwindows[i]) {
            if (pvs_iswin || i == num_reg - 1) {
                _pseudo_reg_pos.set(pos); _pseudo_reg_type.push_back(RegionType::SR); _true_reg_id.push_back(i); cur_len = 0;
            }
            pvs_iswin = false;
        } else {
            const uint32_t winlen = border[i + 1] - pos;
            if (pos == 0 || cur_len + winlen > Window_settings.ideal_lwind_size || !pvs_iswin) {
                _pseudo_reg_pos.set(pos); _pseudo_reg_type.push_back(RegionType::LONG); _true_reg_id.push_back(i);
                _reg_type[i] = RegionType::LONG; cur_len = winlen;
            } else cur_len += winlen;
            pvs_iswin = true;
        }
    }
    _pseudo_reg_pos.init_support();
    for (size_t i = 0; i + 1 < _pseudo_reg_type.size(); ++i)
        if (_pseudo_reg_type[i] == RegionType::LONG)
            _pwindows[_true_reg_id[i]].reset(new Window(_pseq, _pseudo_reg_pos.select(i + 1), _pseudo_reg_pos.select(i + 2), WindowType::LONG));
}

// ---- Contig::fill_long_windows (include/Contig.hpp:91-113) ------------------------------------------------------------
void Contig::fill_long_windows(std::vector<std::unique_ptr<Alignment>>& alignments) {
    add_arms_by_window_range(alignments);
    const size_t num_reg = _reg_type.size() - 1;
    for (size_t i = 0; i < num_reg; ++i)
        if (_reg_type[i] == RegionType::LONG && _pwindows[i]->get_num_internal() > Arms_settings.min_internal_num3) _pwindows[i]->clear_pre_suf();
    _pseudo_reg_pos.clear();
    std::vector<RegionType>().swap(_pseudo_reg_type);
    std::vector<uint32_t>().swap(_true_reg_id);
}

void Contig::dump_votes(std::FILE* f, int which) const {
    const uint32_t head[2] = {(uint32_t)which, _id};
    std::fwrite(head, 4, 2, f);
    if (which == 0) {
        const uint64_t n = _kcov.size();
        std::fwrite(&n, 8, 1, f);
        std::fwrite(_kcov.data(), 4, n, f); std::fwrite(_ksup.data(), 4, n, f);
    } else {
        const uint64_t n = _minimserinfo.size();
        std::fwrite(&n, 8, 1, f);
        for (const MWMinimiserInfo& mi : _minimserinfo) {
            const uint64_t m = mi.coverage.size();
            std::fwrite(&m, 8, 1, f);
            std::fwrite(mi.coverage.data(), 4, m, f); std::fwrite(mi.support.data(), 4, m, f);
        }
    }
}

void Contig::release_after_output() {
    std::vector<std::unique_ptr<Window>>().swap(_pwindows);
    std::vector<RegionType>(1, RegionType::SR).swap(_reg_type);      // (get_num_regions() stays defined: 0)
    std::vector<uint32_t>().swap(_reg_info);
    std::vector<MWMinimiserInfo>().swap(_minimserinfo);
    _reg_pos = BitVec(1);
    _pseq = PackedSeq<4>();
}

// ---- operator<< (src/Contig.cpp:345-366): one-line FASTA record -------------------------------------------------------
std::ostream& operator<<(std::ostream& os, const Contig& ctg) {
    os << ">" << ctg._name << std::endl;
    const size_t num_reg = ctg._reg_type.size() - 1;
    // the record is put together in one string: where every region's text goes is a prefix sum over the regions, the pieces are
    // copied on all threads (the reference streams them one by one; the bytes are the same)
    std::vector<uint64_t> starts, at(num_reg + 1, 0);
    starts.reserve(num_reg + 2);                        // (the borders out of the words of the bit vector: a select() per region was 0.7 of the
    for (uint64_t wi = 0; wi < ctg._reg_pos.n_words(); ++wi) {         // 0.8 s a 250 Mbp contig with 7 M regions took to write)
        uint64_t word = ctg._reg_pos.data()[wi];
        while (word) { starts.push_back(wi * 64 + (uint64_t)__builtin_ctzll(word)); word &= word - 1; }
    }
    if (starts.size() < num_reg + 1) { std::fprintf(stderr, "[Hypo::Contig] Error: region table of contig %s is inconsistent (%zu borders for %zu regions)\n", ctg._name.c_str(), starts.size(), num_reg); std::exit(1); }
    auto kind = [&](size_t i) {                        // 0: draft text, 1: consensus, 2: nothing
        if (ctg._reg_type[i] == RegionType::SR || ctg._reg_type[i] == RegionType::MSR) return 0;
        if (ctg._pwindows[i]) return 1;
        return Contig::_no_long_reads ? 0 : 2;
    };
    for (size_t i = 0; i < num_reg; ++i) {
        const int kd = kind(i);
        at[i + 1] = at[i] + (kd == 0 ? starts[i + 1] - starts[i] : kd == 1 ? ctg._pwindows[i]->consensus_ref().size() : 0);
    }
    std::string text(at[num_reg], 'N');
#pragma omp parallel for schedule(static, 256)
    for (int64_t ii = 0; ii < (int64_t)num_reg; ++ii) {
        const size_t i = (size_t)ii;
        const int kd = kind(i);
        char* dst = &text[at[i]];
        if (kd == 0) { for (uint64_t p = starts[i]; p < starts[i + 1]; ++p) *dst++ = ctg._pseq.base_at(p); }
        else if (kd == 1) { const std::string& c = ctg._pwindows[i]->consensus_ref(); std::memcpy(dst, c.data(), c.size()); }
    }
    os << text;
    os << std::endl;
    return os;
}

}  // namespace hypo
