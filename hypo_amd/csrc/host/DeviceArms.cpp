// DeviceArms.cpp — see DeviceArms.hpp.
#include "DeviceArms.hpp"
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

namespace hypo {

// The short-read alignments of contigs [c0, c1) in one coordinate space (every contig starts on an even position: its PackedSeq<4>
// bytes are copied as they are; an odd-length contig is followed by one filler base), on the device.  Round 4: the records come
// as the flat slices the parser threads wrote (ReadBatch); flatten() lays them out in this context's page-locked staging arrays,
// which hypo_gpu_reads_upload copies at the link's rate — no per-record objects to walk, no fresh pages per batch, nothing to
// release afterwards.
uint32_t DeviceArms::longest_owned_window(const Contig& ctg, bool long_windows) const {
    uint32_t longest = 0;
    if (!long_windows) {
        std::vector<uint32_t> pos;
        ctg._reg_pos.list_set(pos);                              // region starts, then the contig's length
        const size_t nr = (size_t)ctg.get_num_regions();
        for (size_t r = 0; r < nr && r + 1 < pos.size(); ++r) {
            const RegionType t = ctg._reg_type[r];
            if (t == RegionType::SR || t == RegionType::MSR || !owns(pos[r])) continue;
            longest = std::max(longest, pos[r + 1] - pos[r]);
        }
    } else if (!ctg._pseudo_reg_type.empty()) {
        std::vector<uint32_t> pos;
        ctg._pseudo_reg_pos.list_set(pos);
        const size_t np = ctg._pseudo_reg_type.size() - 1;
        for (size_t i = 0; i < np && i + 1 < pos.size(); ++i)
            if (ctg._pseudo_reg_type[i] == RegionType::LONG && owns(pos[i])) longest = std::max(longest, pos[i + 1] - pos[i]);
    }
    return longest;
}

bool DeviceArms::upload_reads(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, const ReadBatch& reads) {
    _reads_resident = false;
    const bool timing = std::getenv("HYPO_HOST_TIMING") != nullptr;
    const auto tu1 = std::chrono::steady_clock::now();
    if (hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    uint64_t total = 0, n_aln = 0;
    std::vector<uint64_t> base(c1 - c0, 0);
    for (uint32_t c = c0; c < c1; ++c) {
        base[c - c0] = total;
        total += contigs[c]->_len + (contigs[c]->_len & 1);
        n_aln += reads.count(c);
    }
    if (n_aln >= 0xfffffff0ull || total >= 0xfffffff0ull) return false;
    bool sorted = true;
    if (_piece && c1 - c0 != 1) return false;
    if (!reads.flatten(c0, c1, base, _stage, sorted, _piece ? _span : nullptr)) {
        std::fprintf(stdout, "[Hypo::Hypo] Info: no page-locked staging memory for %llu alignments: support votes and short arms are computed on the host\n", (unsigned long long)n_aln);
        return false;
    }
    if (!sorted) std::fprintf(stdout, "[Hypo::Hypo] Info: alignments are not sorted by position: sorted on ingest, the arms of a window keep their file order\n");
    HypoArmsReads A;
    A.file_rank = _stage.ranked ? _stage.file_rank : nullptr;
    A.n_alignments = (uint32_t)_stage.n_reads; A.rb = _stage.rb; A.re = _stage.re; A.qae = _stage.qae; A.seq_off = _stage.seq_off;
    A.reads2 = _stage.reads2; A.reads2_bytes = _stage.n_bytes; A.cigar_off = _stage.cigar_off; A.cigar = _stage.cigar;
    const auto tu2 = std::chrono::steady_clock::now();
    const int rc = hypo_gpu_reads_upload(&A, _stage.ctg, total);
    if (timing) {
        auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        std::fprintf(stderr, "[timing] upload_reads: flatten into the staging arrays %.3f s (growing them, so far: %.3f s), hypo_gpu_reads_upload %.3f s (%.0f MB)\n",
                     sec(tu1, tu2), _stage.reserve_seconds, sec(tu2, std::chrono::steady_clock::now()), (_stage.n_bytes + 4.0 * _stage.n_cigar + 28.0 * n_aln) / 1e6);
    }
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: support votes and short arms are computed on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
    _reads_resident = true; _reads_c0 = c0; _reads_c1 = c1;
    return true;
}

// Alignment::update_solidkmers_support for every resident read at once (support_kernel.hip).  Contigs whose scan stayed on the
// device (Contig::scan_kept) vote against that copy: nothing goes over, KmerInfo::coverage / support come back into page-locked
// memory; otherwise the solid k-mers of the contigs go over as positions + k-mers.
bool DeviceArms::support_kmers(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, unsigned k) {
    if (!_reads_resident || c0 != _reads_c0 || c1 != _reads_c1 || hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    uint64_t ns = 0;
    const auto ts0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> kbase(c1 - c0 + 1, 0);
    bool all_kept = !_piece;
    for (uint32_t c = c0; c < c1; ++c) { kbase[c - c0] = ns; ns += contigs[c]->_n_solid; all_kept = all_kept && contigs[c]->_scan_kept; }
    kbase[c1 - c0] = ns;
    if (ns == 0) return true;
    if (ns >= 0xfffffff0ull) return false;
    // results: page-locked, grow-only
    if (ns > _votes_cap) {
        if (_votes) (void)hypo_gpu_host_free(_votes);
        _votes = nullptr; _votes_cap = 0;
        void* p = nullptr;
        if (hypo_gpu_host_alloc((ns + ns / 8 + 1024) * 8, &p) != HYPO_OK || !p) return false;
        _votes = (uint32_t*)p; _votes_cap = ns + ns / 8 + 1024;
    }
    uint32_t* const cov = _votes; uint32_t* const sup = _votes + _votes_cap;
    int rc = HYPO_E_INVALID;
    double t_pos = 0;
    if (all_kept) {
        std::vector<uint32_t> handles(c1 - c0), bases(c1 - c0);
        uint64_t cbase = 0;
        for (uint32_t c = c0; c < c1; ++c) { handles[c - c0] = contigs[c]->_id; bases[c - c0] = (uint32_t)cbase; cbase += contigs[c]->_len + (contigs[c]->_len & 1); }
        uint64_t got = 0;
        rc = hypo_gpu_support_kmers_kept(k, c1 - c0, handles.data(), bases.data(), cov, sup, &got);
        if (rc == HYPO_OK && got != ns) { std::fprintf(stderr, "[Hypo::Hypo] Error: the device kept %llu solid k-mers, the host counts %llu\n", (unsigned long long)got, (unsigned long long)ns); std::exit(1); }
        if (rc == HYPO_OK) for (uint32_t c = c0; c < c1; ++c) { (void)hypo_gpu_solid_release(contigs[c]->_id); contigs[c]->_scan_kept = false; }
    }
    if (rc != HYPO_OK) {
        std::vector<uint32_t> spos(ns);
        std::vector<uint64_t> kids(ns);
        uint64_t cbase = 0;
        for (uint32_t c = c0; c < c1; ++c) {
            Contig& ctg = *contigs[c];
            ctg.ensure_kids();
            const uint64_t b = kbase[c - c0], n = ctg._n_solid;
            std::memcpy(kids.data() + b, ctg._kids.data(), n * 8);
#pragma omp parallel for schedule(static, 4096)
            for (int64_t i = 0; i < (int64_t)n; ++i) spos[b + (uint64_t)i] = (uint32_t)(cbase + ctg._solid_pos.select((uint64_t)i + 1));
            cbase += ctg._len + (ctg._len & 1);
        }
        t_pos = std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
        rc = hypo_gpu_support_kmers(k, ns, spos.data(), kids.data(), cov, sup);
    }
    if (std::getenv("HYPO_HOST_TIMING"))
        std::fprintf(stderr, "[timing] support_kmers: %llu solid k-mers (%s), positions + k-mers on the host %.3f s, whole call %.3f s\n", (unsigned long long)ns,
                     all_kept ? "kept on the device" : "sent over", t_pos, std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count());
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: k-mer support is counted on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t cc = (int64_t)c0; cc < (int64_t)c1; ++cc) {
        const uint32_t c = (uint32_t)cc;
        Contig& ctg = *contigs[c];
        const uint64_t b = kbase[c - c0], n = ctg._n_solid;
        // (piece mode: the counters of the solid k-mers this context owns; the others are another context's)
        const uint64_t i0 = _piece ? ctg._solid_pos.rank(_own0) : 0, i1 = _piece ? ctg._solid_pos.rank(std::min<uint64_t>(_own1, ctg._len)) : n;
        if (i1 > i0) {
            std::memcpy(ctg._kcov.data() + i0, cov + b + i0, (i1 - i0) * 4);
            std::memcpy(ctg._ksup.data() + i0, sup + b + i0, (i1 - i0) * 4);
        }
    }
    std::fprintf(stdout, "[Hypo::Hypo] Info: k-mer support counted on the device: %llu solid k-mers\n", (unsigned long long)ns);
    return true;
}

// Alignment::update_minimisers_support likewise: region borders and the minimizers of the mega-windows go over, MWMinimiserInfo::
// coverage / support come back.
bool DeviceArms::support_minimizers(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1) {
    if (!_reads_resident || c0 != _reads_c0 || c1 != _reads_c1 || hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    const auto tm0 = std::chrono::steady_clock::now();
    const uint32_t nc = c1 - c0;
    // sizes first, then every contig (its borders) and every mega-window (its minimizers) copies into its own slice on all threads
    std::vector<uint32_t> contig_base(nc), reg_base(nc + 1, 0), info_base(nc + 1, 0);
    std::vector<uint8_t> even(nc);
    uint64_t cbase = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const Contig& ctg = *contigs[c];
        contig_base[c - c0] = (uint32_t)cbase;
        even[c - c0] = ctg._is_win_even ? 1 : 0;
        reg_base[c - c0 + 1] = reg_base[c - c0] + (uint32_t)ctg._reg_pos.count();      // set bits: 0, SR starts and ends, the length
        info_base[c - c0 + 1] = info_base[c - c0] + (uint32_t)ctg._minimserinfo.size();
        cbase += ctg._len + (ctg._len & 1);
    }
    const uint64_t n_start = reg_base[nc], n_info = info_base[nc];
    std::vector<MWMinimiserInfo*> infos(n_info);
    uint32_t* const mw_off = _pb[0].get<uint32_t>(n_info + 2);
    if (!mw_off) return false;
    {
        uint64_t x = 0, e = 0;
        for (uint32_t c = c0; c < c1; ++c)
            for (MWMinimiserInfo& mi : contigs[c]->_minimserinfo) { infos[x] = &mi; mw_off[x] = (uint32_t)e; e += mi.rel_pos.size(); ++x; if (e >= 0xfffffff0ull) return false; }
        mw_off[n_info] = (uint32_t)e;
    }
    const uint64_t n_ent = mw_off[n_info];
    if (!n_ent) return true;
    if (n_start >= 0xfffffff0ull) return false;
    uint32_t* const start = _pb[1].get<uint32_t>(n_start);
    uint32_t* const rel_pos = _pb[2].get<uint32_t>(n_ent);
    uint32_t* const minimisers = _pb[3].get<uint32_t>(n_ent);
    uint32_t* const cov = _pb[4].get<uint32_t>(n_ent);
    uint32_t* const sup = _pb[5].get<uint32_t>(n_ent);
    if (!start || !rel_pos || !minimisers || !cov || !sup) return false;
    for (uint32_t c = c0; c < c1; ++c) {
        const Contig& ctg = *contigs[c];
        const uint64_t s0 = reg_base[c - c0], nb = reg_base[c - c0 + 1] - s0;
#pragma omp parallel for schedule(static, 4096)
        for (int64_t i = 0; i < (int64_t)nb; ++i) start[s0 + (uint64_t)i] = (uint32_t)ctg._reg_pos.select((uint64_t)i + 1);
    }
#pragma omp parallel for schedule(static, 256)
    for (int64_t x = 0; x < (int64_t)n_info; ++x) {
        const MWMinimiserInfo& mi = *infos[(size_t)x];
        if (mi.rel_pos.empty()) continue;
        std::memcpy(rel_pos + mw_off[x], mi.rel_pos.data(), mi.rel_pos.size() * 4);
        std::memcpy(minimisers + mw_off[x], mi.minimisers.data(), mi.minimisers.size() * 4);
    }
    const auto tm1 = std::chrono::steady_clock::now();
    HypoMegaWindows W;
    W.n_contigs = nc; W.contig_base = contig_base.data(); W.reg_base = reg_base.data(); W.win_even = even.data(); W.info_base = info_base.data();
    W.start = start; W.n_info = (uint32_t)n_info; W.mw_off = mw_off; W.rel_pos = rel_pos; W.minimisers = minimisers;
    const int rc = hypo_gpu_support_minimizers(&W, cov, sup);
    const auto tm2 = std::chrono::steady_clock::now();
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: minimizer support is counted on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
    if (!_piece) {
#pragma omp parallel for schedule(static, 256)
        for (int64_t x = 0; x < (int64_t)n_info; ++x) {
            MWMinimiserInfo& mi = *infos[(size_t)x];
            const size_t n = mi.rel_pos.size();
            if (!n) continue;
            std::memcpy(mi.coverage.data(), cov + mw_off[x], n * 4);
            std::memcpy(mi.support.data(), sup + mw_off[x], n * 4);
        }
    } else {
        // piece mode (one contig): the counters of the minimizers that lie in the owned range.  Mega-window x of the contig is region
        // 2 x (+ 1 when the contig starts with an SR) of the border table; a minimizer's position is the window's start plus the
        // distances before it (support_kernel.hip: MwCursor)
        const Contig& ctg = *contigs[c0];
#pragma omp parallel for schedule(static, 256)
        for (int64_t x = 0; x < (int64_t)n_info; ++x) {
            MWMinimiserInfo& mi = *infos[(size_t)x];
            const size_t n = mi.rel_pos.size();
            if (!n) continue;
            const uint64_t w = ctg._is_win_even ? 2 * (uint64_t)x : 2 * (uint64_t)x + 1;
            uint64_t pos = start[w];
            for (size_t m = 0; m < n; ++m) {
                pos += mi.rel_pos[m];
                if (pos >= _own0 && pos < _own1) { mi.coverage[m] = cov[mw_off[x] + m]; mi.support[m] = sup[mw_off[x] + m]; }
            }
        }
    }
    if (std::getenv("HYPO_HOST_TIMING")) {
        auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        std::fprintf(stderr, "[timing] support_minimizers: tables %.3f s, hypo_gpu_support_minimizers %.3f s, counters back %.3f s\n", sec(tm0, tm1), sec(tm1, tm2), sec(tm2, std::chrono::steady_clock::now()));
    }
    std::fprintf(stdout, "[Hypo::Hypo] Info: minimizer support counted on the device: %llu minimizers\n", (unsigned long long)n_ent);
    return true;
}

bool DeviceArms::build(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, const ReadBatch& reads, unsigned k) {
    _active = false;
    if (hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    const bool timing = std::getenv("HYPO_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = now();
    // the reads: resident since the support votes, or put there now
    if (!(_reads_resident && c0 == _reads_c0 && c1 == _reads_c1) && !upload_reads(contigs, c0, c1, reads)) return false;
    _reads_resident = false;                                   // (this build consumes them: the next batch uploads its own)
    // the coordinate space and its region tables: sizes per contig first, then every contig fills its own slices (page-locked,
    // reused by every batch) on a thread of its own
    const uint32_t nc = c1 - c0;
    std::vector<uint64_t> cbase(nc + 1, 0), rbase(nc + 1, 0), abase(nc + 1, 0);
    for (uint32_t c = c0; c < c1; ++c) {
        const Contig& ctg = *contigs[c];
        cbase[c - c0 + 1] = cbase[c - c0] + ctg._len + (ctg._len & 1);
        rbase[c - c0 + 1] = rbase[c - c0] + ctg.get_num_regions() + (ctg._len & 1);
        // Contig::_anchor_kmers is [dummy, first k-mer of SR 1, last k-mer of SR 1, first of SR 2, ...] and the SR ranks start at 1:
        // one dummy for the whole coordinate space, the ranks of later contigs shifted by the SRs before them
        abase[c - c0 + 1] = abase[c - c0] + (ctg._anchor_kmers.size() > 1 ? ctg._anchor_kmers.size() - 1 : 0);
    }
    const uint64_t total = cbase[nc], n_reg = rbase[nc], n_anchor = abase[nc] + 1;
    if (total >= 0xfffffff0ull || n_reg >= 0xfffffff0ull || n_reg == 0) return false;
    uint32_t* const start = _pb[6].get<uint32_t>(n_reg + 1);
    uint32_t* const info = _pb[7].get<uint32_t>(n_reg + 1);
    uint8_t* const type = _pb[8].get<uint8_t>(n_reg + 1);
    uint64_t* const anchors = _pb[9].get<uint64_t>(n_anchor);
    uint8_t* const contig4 = _pb[10].get<uint8_t>((total + 1) / 2 + 16);
    if (!start || !info || !type || !anchors || !contig4) return false;
    _reg_window.assign(n_reg, nullptr);
    anchors[0] = 0;
    bool bad = false;
#pragma omp parallel for schedule(dynamic, 1) reduction(|| : bad)
    for (int64_t ci = 0; ci < (int64_t)nc; ++ci) {
        Contig& ctg = *contigs[c0 + (uint32_t)ci];
        const uint32_t nr = (uint32_t)ctg.get_num_regions();
        const uint64_t base = cbase[(size_t)ci];
        uint64_t r = rbase[(size_t)ci];
        const uint32_t sr_before = (uint32_t)(abase[(size_t)ci] / 2);
        std::vector<uint32_t> border;                              // (a 250 Mbp contig has 7 M regions: no select() per region)
        ctg._reg_pos.list_set(border);
        if (border.size() < nr) { bad = true; continue; }
        for (uint32_t i = 0; i < nr; ++i, ++r) {
            start[r] = (uint32_t)(base + border[i]);
            type[r] = (uint8_t)ctg._reg_type[i];
            info[r] = ctg._reg_info[i] + (ctg._reg_type[i] == RegionType::SR ? sr_before : 0u);
            _reg_window[r] = ctg._pwindows[i].get();
            // (never: every non-SR region has a window here — unless another context of a shared contig has already pruned a window
            // it owns: not this context's business)
            if (!_piece && type[r] != (uint8_t)RegionType::SR && type[r] != (uint8_t)RegionType::MSR && !_reg_window[r]) bad = true;
        }
        if (ctg._anchor_kmers.size() > 1) std::memcpy(anchors + 1 + abase[(size_t)ci], ctg._anchor_kmers.data() + 1, (ctg._anchor_kmers.size() - 1) * 8);
        std::memcpy(contig4 + base / 2, ctg._pseq.data(), ctg._pseq.byte_size());
        if (ctg._len & 1) { start[r] = (uint32_t)(base + ctg._len); type[r] = (uint8_t)RegionType::SR; info[r] = 0; ++r; }
    }
    if (bad) return false;
    start[n_reg] = (uint32_t)total; type[n_reg] = (uint8_t)RegionType::SR; info[n_reg] = 0;
    HypoArmsRegions R;
    R.n_regions = (uint32_t)n_reg; R.start = start; R.type = type; R.info = info;
    R.n_anchor_kmers = n_anchor; R.anchor_kmers = anchors; R.k = k; R.contig4 = contig4;
    std::vector<uint8_t> valid(n_reg, 0);
    const auto t1 = now();
    const int rc = hypo_gpu_arms_build(&R, nullptr, valid.data(), &_sum);       // the resident reads
    const auto t2 = now();
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: short arms are computed on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
    // what Contig::fill_short_windows leaves behind (src/Contig.cpp:249-289): pruned windows are gone, the anchors are freed
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ci = 0; ci < (int64_t)nc; ++ci) {
        Contig& ctg = *contigs[c0 + (uint32_t)ci];
        const uint32_t nr = (uint32_t)ctg.get_num_regions();
        uint64_t r = rbase[(size_t)ci];
        for (uint32_t i = 0; i < nr; ++i, ++r) {
            // (piece mode: a region belongs to the context that owns its start; the windows this context built in its halo are
            // another context's to judge and to polish)
            if (_piece && !owns((uint64_t)start[r] - cbase[(size_t)ci])) { _reg_window[r] = nullptr; continue; }
            if (ctg._pwindows[i] && !valid[r]) { ctg._pwindows[i].reset(); _reg_window[r] = nullptr; }
        }
        if (!_piece) {                                     // (piece mode: the other contexts of the contig still read them; Hypo::polish frees them)
            std::vector<uint64_t>().swap(ctg._anchor_kmers);
            std::vector<uint32_t>().swap(ctg._reg_info);
        }
    }
    if (timing) std::fprintf(stderr, "[timing] device arms: flatten %.3f s, hypo_gpu_arms_build %.3f s, prune + release %.3f s\n", secs(t0, t1), secs(t1, t2), secs(t2, now()));
    std::fprintf(stdout, "[Hypo::Hypo] Info: short arms cut on the device: %u windows, %u arms\n", _sum.n_windows, _sum.n_arms);
    _active = true;
    return true;
}

bool DeviceArms::build_long(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, const ReadBatch& reads) {
    _active_long = false; _long_failed = false;
    if (hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    const bool timing = std::getenv("HYPO_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = now();
    // the coordinate space of build(): contigs back to back, each starting on an even position (a 1-base filler region of type SR
    // behind an odd-length contig); its regions are the PSEUDO regions of Contig::prepare_long_windows
    uint64_t total = 0, n_reg = 0, n_aln = 0;
    std::vector<uint64_t> base(c1 - c0, 0);
    if (_piece && c1 - c0 != 1) return false;
    for (uint32_t c = c0; c < c1; ++c) {
        const Contig& ctg = *contigs[c];
        if (ctg._pseudo_reg_type.empty()) return false;
        base[c - c0] = total;
        total += ctg._len + (ctg._len & 1);
        n_reg += ctg._pseudo_reg_type.size() - 1 + (ctg._len & 1);
        n_aln += reads.count(c);
    }
    if (total >= 0xfffffff0ull || n_reg >= 0xfffffff0ull || n_aln >= 0xfffffff0ull || n_reg == 0) return false;
    // the long reads of the contigs, flat (ReadBatch.hpp), laid out contig after contig in the staging arrays of this context (piece
    // mode: the reads that overlap this context's span of the contig)
    bool sorted = true;
    if (!reads.flatten(c0, c1, base, _stage_long, sorted, _piece ? _span : nullptr)) return false;
    // (an unsorted -B file — the reference takes any order, src/Hypo.cpp:278-329 — is sorted by position on ingest and carries its file
    // ranks like an unsorted -b file: the arms of a LONG window are laid out in file order, arms_window_kernel)
    if (!sorted) std::fprintf(stdout, "[Hypo::Hypo] Info: long-read alignments are not sorted by position: sorted on ingest, the arms of a window keep their file order\n");
    std::vector<uint32_t> start(n_reg + 1);
    std::vector<uint8_t> type(n_reg + 1, (uint8_t)RegionType::SR);
    std::vector<uint8_t> contig4((total + 1) / 2, 0);
    _preg_window.assign(n_reg, nullptr);
    uint64_t r = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        Contig& ctg = *contigs[c];
        const uint64_t b0 = base[c - c0];
        const size_t np = ctg._pseudo_reg_type.size() - 1;       // the last pseudo region is the end marker at the contig's length
        for (size_t i = 0; i < np; ++i, ++r) {
            start[r] = (uint32_t)(b0 + ctg._pseudo_reg_pos.select((uint64_t)i + 1));
            const bool lw = ctg._pseudo_reg_type[i] == RegionType::LONG;
            type[r] = (uint8_t)(lw ? RegionType::LONG : RegionType::SR);
            if (lw) {
                _preg_window[r] = ctg._pwindows[ctg._true_reg_id[i]].get();
                if (!_preg_window[r]) return false;
                if (_piece && !owns((uint64_t)start[r] - b0)) _preg_window[r] = nullptr;      // (a LONG window of the halo: another context's)
            }
        }
        std::memcpy(contig4.data() + b0 / 2, ctg._pseq.data(), ctg._pseq.byte_size());
        if (ctg._len & 1) { start[r] = (uint32_t)(b0 + ctg._len); type[r] = (uint8_t)RegionType::SR; ++r; }
    }
    start[r] = (uint32_t)total;
    HypoArmsRegions R;
    R.n_regions = (uint32_t)n_reg; R.start = start.data(); R.type = type.data(); R.info = nullptr;
    R.n_anchor_kmers = 0; R.anchor_kmers = nullptr; R.k = 10; R.contig4 = contig4.data();
    HypoArmsReads A;
    A.file_rank = _stage_long.ranked ? _stage_long.file_rank : nullptr;
    A.n_alignments = (uint32_t)_stage_long.n_reads; A.rb = _stage_long.rb; A.re = _stage_long.re; A.qae = _stage_long.qae; A.seq_off = _stage_long.seq_off;
    A.reads2 = _stage_long.reads2; A.reads2_bytes = _stage_long.n_bytes; A.cigar_off = _stage_long.cigar_off; A.cigar = _stage_long.cigar;
    std::vector<uint8_t> valid(n_reg, 0);
    const auto t1 = now();
    const int rc = hypo_gpu_arms_build_long(&R, &A, valid.data(), &_sum_long);
    if (timing) std::fprintf(stderr, "[timing] device long arms: flatten %.3f s (%.0f MB of bases, %.0f MB of CIGAR), hypo_gpu_arms_build_long %.3f s\n", secs(t0, t1), _stage_long.n_bytes / 1e6, _stage_long.n_cigar * 4 / 1e6, secs(t1, now()));
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) { _long_failed = true; std::fprintf(stdout, "[Hypo::Hypo] Info: long arms are computed on the host (%s)\n", hypo_gpu_last_error()); }
        return false;
    }
    // what Contig::fill_long_windows leaves behind (include/Contig.hpp:91-113): the pseudo tables are gone
    if (!_piece) finish_long(contigs, c0, c1);                 // (piece mode: the other contexts of the contig still read them; Hypo::polish calls it)
    std::fprintf(stdout, "[Hypo::Hypo] Info: long arms cut on the device: %u windows, %u arms\n", _sum_long.n_windows, _sum_long.n_arms);
    _active_long = true;
    return true;
}

void DeviceArms::finish_long(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1) {
    for (uint32_t c = c0; c < c1; ++c) {
        Contig& ctg = *contigs[c];
        ctg._pseudo_reg_pos.clear();
        std::vector<RegionType>().swap(ctg._pseudo_reg_type);
        std::vector<uint32_t>().swap(ctg._true_reg_id);
    }
}
void DeviceArms::finish_short(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1) {
    for (uint32_t c = c0; c < c1; ++c) {
        std::vector<uint64_t>().swap(contigs[c]->_anchor_kmers);
        std::vector<uint32_t>().swap(contigs[c]->_reg_info);
    }
}

void DeviceArms::adopt_arms(const std::vector<uint32_t>& which, const std::vector<HypoWindow>& hw, const std::vector<uint32_t>& win_region, bool lng) {
    const HypoArmsSummary& sum = lng ? _sum_long : _sum;
    const std::vector<Window*>& rw = lng ? _preg_window : _reg_window;
    std::vector<uint32_t> arm_len(sum.n_arms);
    std::vector<uint64_t> arm_off(sum.n_arms);
    std::vector<uint8_t> arms2(sum.arms2_bytes ? sum.arms2_bytes : 1);
    if ((lng ? hypo_gpu_arms_download_long : hypo_gpu_arms_download)(nullptr, nullptr, arm_len.data(), arm_off.data(), arms2.data(), nullptr) != HYPO_OK) {
        std::fprintf(stderr, "[Hypo::Window] Error: %s\n", hypo_gpu_last_error()); std::exit(1);
    }
    for (uint32_t wi : which) {
        Window& w = *rw[win_region[wi]];
        const HypoWindow& d = hw[wi];
        uint32_t a = d.first_arm;
        auto take = [&](uint32_t arm) { return PackedSeq<2>(arms2.data() + arm_off[arm], arm_len[arm]); };
        for (uint32_t i = 0; i < d.n_internal; ++i) w.add_internal(take(a++));
        for (uint32_t i = 0; i < d.n_prefix; ++i) w.add_prefix(take(a++));
        for (uint32_t i = 0; i < d.n_suffix; ++i) w.add_suffix(take(a++));
        w._num_empty = d.n_empty;
    }
}

int DeviceArms::polish(const ScoreParams& sp, bool keep_arms, std::vector<Window*>* retry) {
    _n_pol[0] = 0;
    if (!_active) return HYPO_OK;
    _active = false;
    return polish_impl(false, sp, keep_arms, retry);
}
int DeviceArms::polish_long(const ScoreParams& sp, bool keep_arms, std::vector<Window*>* retry) {
    _n_pol[1] = 0;
    if (!_active_long) return HYPO_OK;
    _active_long = false;
    return polish_impl(true, sp, keep_arms, retry);
}
int DeviceArms::polish_impl(bool lng, const ScoreParams& sp, bool keep_arms, std::vector<Window*>* retry) {
    if (hypo_gpu_use_device(_slot) != HYPO_OK) return HYPO_E_INVALID;
    const HypoArmsSummary& _sum = lng ? _sum_long : this->_sum;
    const std::vector<Window*>& _reg_window = lng ? _preg_window : this->_reg_window;
    const uint32_t n = _sum.n_windows;
    if (!n) return HYPO_OK;
    const auto tp = std::chrono::steady_clock::now();
    // results: page-locked, reused by every batch (fresh pageable vectors cost their page faults and a staged copy per batch)
    PinnedBuf* const pb = lng ? _pbl : _pbr;
    char* const bases = pb[0].get<char>(_sum.out_bytes + 16);
    uint64_t* const off = pb[1].get<uint64_t>((size_t)n + 1);
    uint32_t* const len = pb[2].get<uint32_t>(n);
    uint32_t* const win_region = pb[3].get<uint32_t>(n);
    uint8_t* const st = pb[4].get<uint8_t>(n);
    if (!bases || !off || !len || !win_region || !st) return HYPO_E_HIP;
    std::vector<HypoWindow> hw;                                  // (descriptors: only when arms are adopted)
    const bool timing = std::getenv("HYPO_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = now();
    int rc = (lng ? hypo_gpu_arms_poa_long : hypo_gpu_arms_poa)(&sp, bases, off, len, st);
    if (rc != HYPO_OK) return rc;
    const auto t1 = now();
    rc = (lng ? hypo_gpu_arms_download_long : hypo_gpu_arms_download)(nullptr, win_region, nullptr, nullptr, nullptr, nullptr);
    if (rc != HYPO_OK) return rc;
    if (timing) {
        std::fprintf(stderr, "[timing] device arms: buffers %.3f s, hypo_gpu_arms_poa%s %.3f s, descriptors %.3f s\n", secs(tp, t0), lng ? "_long" : "", secs(t0, t1), secs(t1, now()));
        HypoPoaStats ps;
        if (hypo_gpu_poa_last_stats(&ps) == HYPO_OK)
            std::fprintf(stderr, "[timing]   windows finished per class [%llu %llu %llu %llu %llu %llu], re-queued %llu (with their graph: %llu), alignments %llu (reused %llu, threaded %llu), cells scored %.2f G of %.2f G\n",
                         (unsigned long long)ps.n_class[0], (unsigned long long)ps.n_class[1], (unsigned long long)ps.n_class[2], (unsigned long long)ps.n_class[3],
                         (unsigned long long)ps.n_class[4], (unsigned long long)ps.n_class[5], (unsigned long long)ps.n_escalated, (unsigned long long)ps.n_carried,
                         (unsigned long long)ps.n_alignments, (unsigned long long)ps.n_reused, (unsigned long long)ps.n_threaded, ps.cells_scored / 1e9, ps.dp_cells / 1e9);
        if (const char* dp = lng && std::getenv("HYPO_DUMP_LONG") && *std::getenv("HYPO_DUMP_LONG") ? std::getenv("HYPO_DUMP_LONG") : nullptr) {      // the resident LONG batch as a file (profiles/diag/r04_long_replay.py runs it again)
            std::vector<HypoWindow> dw(n);
            std::vector<uint32_t> al(_sum.n_arms ? _sum.n_arms : 1);
            std::vector<uint64_t> ao(_sum.n_arms ? _sum.n_arms : 1);
            std::vector<uint8_t> a2(_sum.arms2_bytes ? _sum.arms2_bytes : 1), d4(_sum.draft4_bytes ? _sum.draft4_bytes : 1);
            if (hypo_gpu_arms_download_long(dw.data(), nullptr, al.data(), ao.data(), a2.data(), d4.data()) == HYPO_OK) {
                if (FILE* f = std::fopen(dp, "wb")) {
                    const uint64_t hd[4] = {n, _sum.n_arms, _sum.arms2_bytes, _sum.draft4_bytes};
                    std::fwrite(hd, 8, 4, f); std::fwrite(dw.data(), sizeof(HypoWindow), n, f); std::fwrite(al.data(), 4, _sum.n_arms, f);
                    std::fwrite(ao.data(), 8, _sum.n_arms, f); std::fwrite(a2.data(), 1, _sum.arms2_bytes, f); std::fwrite(d4.data(), 1, _sum.draft4_bytes, f);
                    std::fclose(f);
                }
            }
        }
        if (lng) {                                                  // the shapes of the LONG batch (what the rate files of profiles/ are to be compared with)
            std::vector<HypoWindow> dw(n);
            std::vector<uint32_t> al(_sum.n_arms ? _sum.n_arms : 1);
            if (hypo_gpu_arms_download_long(dw.data(), nullptr, al.data(), nullptr, nullptr, nullptr) == HYPO_OK) {
                uint64_t arms = 0, dsum = 0, lsum = 0, over384 = 0, over512 = 0; uint32_t amax = 0, dmax = 0, lmax = 0, wmax_arms = 0;
                for (const HypoWindow& w : dw) {
                    const uint32_t na = w.n_internal + w.n_prefix + w.n_suffix;
                    arms += na; dsum += w.draft_len; dmax = std::max(dmax, w.draft_len); wmax_arms = std::max(wmax_arms, na);
                    for (uint32_t a = 0; a < na; ++a) { const uint32_t l = al[w.first_arm + a]; lsum += l; lmax = std::max(lmax, l); over384 += l > 383; over512 += l > 511; amax = std::max(amax, l); }
                }
                std::fprintf(stderr, "[timing]   LONG batch: %u windows, draft %.0f bases on average (longest %u), %.1f arms per window (most %u), arms %.0f bases on average (longest %u; %llu over 383, %llu over 511)\n",
                             n, (double)dsum / n, dmax, (double)arms / n, wmax_arms, arms ? (double)lsum / arms : 0.0, lmax, (unsigned long long)over384, (unsigned long long)over512);
            }
        }
    }
    // (every status byte starts as HYPO_ST_UNWRITTEN on the device: a window no kernel answered is an internal error, not a retry)
    for (uint32_t i = 0; i < n; ++i)
        if (st[i] == HYPO_ST_UNWRITTEN) {
            std::fprintf(stderr, "[Hypo::Window] Error: resident window %u of %u came back unanswered by the device (status 0xff)\n", i, n);
            return HYPO_E_INVALID;
        }
    std::vector<uint32_t> again, all;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        if (st[(size_t)i] == HYPO_ST_OK && _reg_window[win_region[(size_t)i]]) _reg_window[win_region[(size_t)i]]->_consensus.assign(bases + off[(size_t)i], len[(size_t)i]);
    uint64_t owned = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!_reg_window[win_region[i]]) continue;           // (piece mode: a window of the halo, another context's)
        ++owned;
        if (st[i] != HYPO_ST_OK) again.push_back(i);
        if (keep_arms) all.push_back(i);
    }
    _n_pol[lng ? 1 : 0] = owned;
    if (keep_arms || !again.empty()) {
        hw.resize(n);
        rc = (lng ? hypo_gpu_arms_download_long : hypo_gpu_arms_download)(hw.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc != HYPO_OK) return rc;
    }
    if (keep_arms || !again.empty()) {
        const std::vector<uint32_t> wr(win_region, win_region + n);
        adopt_arms(keep_arms ? all : again, hw, wr, lng);
    }
    if (!again.empty()) {          // a consensus longer than its slot, a window beyond the size classes: the host's retry / degraded path
        std::vector<Window*> ws;
        for (uint32_t i : again) ws.push_back(_reg_window[win_region[i]]);
        if (retry) retry->insert(retry->end(), ws.begin(), ws.end());
        else rc = Window::generate_consensus_batch(ws);
    }
    return rc;
}

}  // namespace hypo
