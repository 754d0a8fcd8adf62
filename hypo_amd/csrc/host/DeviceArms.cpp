// DeviceArms.cpp — see DeviceArms.hpp.
#include "DeviceArms.hpp"
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

namespace hypo {

// The short-read alignments of contigs [c0, c1) as flat arrays in one coordinate space (every contig starts on an even position:
// its PackedSeq<4> bytes are copied as they are; an odd-length contig is followed by one filler base), on the device.
bool DeviceArms::upload_reads(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1,
                              std::vector<std::vector<std::unique_ptr<Alignment>>>& store) {
    _reads_resident = false;
    const bool timing = std::getenv("HYPO_HOST_TIMING") != nullptr;
    const auto tu0 = std::chrono::steady_clock::now();
    wait_released();
    const auto tu1 = std::chrono::steady_clock::now();
    if (hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    uint64_t total = 0, n_aln = 0, n_cig = 0;
    std::vector<uint64_t> aln_base(c1 - c0 + 1, 0);
    for (uint32_t c = c0; c < c1; ++c) {
        total += contigs[c]->_len + (contigs[c]->_len & 1);
        aln_base[c - c0] = n_aln;
        n_aln += store[c].size();
    }
    if (n_aln >= 0xfffffff0ull || total >= 0xfffffff0ull) return false;
    // per alignment: bytes of its read, CIGAR operations (exclusive prefix sums below), and the sort check
    // (plain arrays: nothing here is read before it is written, and zero-filling 28 bytes per record on one thread was 0.15 s of the C3 run)
    std::unique_ptr<uint64_t[]> seq_off(new uint64_t[n_aln + 1]);
    std::unique_ptr<uint32_t[]> cigar_off(new uint32_t[n_aln + 1]);
    bool sorted = true;
    for (uint32_t c = c0; c < c1; ++c) {
        const auto& alns = store[c];
        const uint64_t a0 = aln_base[c - c0];
#pragma omp parallel for schedule(static) reduction(&& : sorted)
        for (int64_t t = 0; t < (int64_t)alns.size(); ++t) {
            const Alignment& a = *alns[(size_t)t];
            seq_off[a0 + (uint64_t)t + 1] = a._apseq.byte_size();
            cigar_off[a0 + (uint64_t)t + 1] = (uint32_t)a._cigar.size();
            if (t && alns[(size_t)t - 1]->_rb > a._rb) sorted = false;
        }
    }
    if (!sorted) { std::fprintf(stdout, "[Hypo::Hypo] Info: alignments are not sorted by position: support votes and short arms are computed on the host\n"); return false; }
    seq_off[0] = 0; cigar_off[0] = 0;
    for (uint64_t g = 0; g < n_aln; ++g) { n_cig += cigar_off[g + 1]; if (n_cig >= 0xfffffff0ull) return false; seq_off[g + 1] += seq_off[g]; cigar_off[g + 1] += cigar_off[g]; }
    const uint64_t read_bytes = seq_off[n_aln];
    std::unique_ptr<uint32_t[]> rb(new uint32_t[n_aln + 1]), re(new uint32_t[n_aln + 1]), qae(new uint32_t[n_aln + 1]), ctg_of(new uint32_t[n_aln + 1]);
    std::unique_ptr<uint32_t[]> cigar(new uint32_t[n_cig ? n_cig : 1]);             // (not zero-filled: every element is written below)
    std::unique_ptr<uint8_t[]> reads2(new uint8_t[read_bytes ? read_bytes : 1]);
    {   // the copies, on all threads
        uint64_t cbase = 0;
        for (uint32_t c = c0; c < c1; ++c) {
            auto& alns = store[c];
            const uint64_t a0 = aln_base[c - c0];
#pragma omp parallel for schedule(static)
            for (int64_t t = 0; t < (int64_t)alns.size(); ++t) {
                const Alignment& a = *alns[(size_t)t];
                const uint64_t g = a0 + (uint64_t)t;
                rb[g] = (uint32_t)(cbase + a._rb); re[g] = (uint32_t)(cbase + a._re); qae[g] = a._qae; ctg_of[g] = c - c0;
                std::memcpy(reads2.get() + seq_off[g], a._apseq.data(), a._apseq.byte_size());
                std::memcpy(cigar.get() + cigar_off[g], a._cigar.data(), a._cigar.size() * 4);
            }
            cbase += contigs[c]->_len + (contigs[c]->_len & 1);
        }
    }
    HypoArmsReads A;
    A.n_alignments = (uint32_t)n_aln; A.rb = rb.get(); A.re = re.get(); A.qae = qae.get(); A.seq_off = seq_off.get();
    A.reads2 = reads2.get(); A.reads2_bytes = read_bytes; A.cigar_off = cigar_off.get(); A.cigar = cigar.get();
    const auto tu2 = std::chrono::steady_clock::now();
    const int rc = hypo_gpu_reads_upload(&A, ctg_of.get(), total);
    if (timing) {
        auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        std::fprintf(stderr, "[timing] upload_reads: wait for the last batch's release %.3f s, flatten %.3f s, hypo_gpu_reads_upload %.3f s (%.0f MB)\n",
                     sec(tu0, tu1), sec(tu1, tu2), sec(tu2, std::chrono::steady_clock::now()), (read_bytes + 4.0 * n_cig + 28.0 * n_aln) / 1e6);
    }
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: support votes and short arms are computed on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
    _reads_resident = true; _reads_c0 = c0; _reads_c1 = c1;
    return true;
}

// Alignment::update_solidkmers_support for every resident read at once (support_kernel.hip): the solid k-mers of the contigs go
// over as positions + k-mers, KmerInfo::coverage / support come back.
bool DeviceArms::support_kmers(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, unsigned k) {
    if (!_reads_resident || c0 != _reads_c0 || c1 != _reads_c1 || hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    uint64_t ns = 0;
    const auto ts0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> kbase(c1 - c0 + 1, 0);
    for (uint32_t c = c0; c < c1; ++c) { kbase[c - c0] = ns; ns += contigs[c]->_kids.size(); }
    kbase[c1 - c0] = ns;
    if (ns == 0) return true;
    if (ns >= 0xfffffff0ull) return false;
    std::vector<uint32_t> spos(ns), cov(ns), sup(ns);
    std::vector<uint64_t> kids(ns);
    uint64_t cbase = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const Contig& ctg = *contigs[c];
        const uint64_t b = kbase[c - c0], n = ctg._kids.size();
        std::memcpy(kids.data() + b, ctg._kids.data(), n * 8);
#pragma omp parallel for schedule(static, 4096)
        for (int64_t i = 0; i < (int64_t)n; ++i) spos[b + (uint64_t)i] = (uint32_t)(cbase + ctg._solid_pos.select((uint64_t)i + 1));
        cbase += ctg._len + (ctg._len & 1);
    }
    const auto ts1 = std::chrono::steady_clock::now();
    const int rc = hypo_gpu_support_kmers(k, ns, spos.data(), kids.data(), cov.data(), sup.data());
    if (std::getenv("HYPO_HOST_TIMING"))
        std::fprintf(stderr, "[timing] support_kmers: positions of %llu solid k-mers %.3f s, hypo_gpu_support_kmers %.3f s\n", (unsigned long long)ns,
                     std::chrono::duration<double>(ts1 - ts0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - ts1).count());
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: k-mer support is counted on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
    for (uint32_t c = c0; c < c1; ++c) {
        Contig& ctg = *contigs[c];
        const uint64_t b = kbase[c - c0], n = ctg._kids.size();
        std::memcpy(ctg._kcov.data(), cov.data() + b, n * 4);
        std::memcpy(ctg._ksup.data(), sup.data() + b, n * 4);
    }
    std::fprintf(stdout, "[Hypo::Hypo] Info: k-mer support counted on the device: %llu solid k-mers\n", (unsigned long long)ns);
    return true;
}

// Alignment::update_minimisers_support likewise: region borders and the minimizers of the mega-windows go over, MWMinimiserInfo::
// coverage / support come back.
bool DeviceArms::support_minimizers(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1) {
    if (!_reads_resident || c0 != _reads_c0 || c1 != _reads_c1 || hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    const uint32_t nc = c1 - c0;
    std::vector<uint32_t> contig_base(nc), reg_base(nc + 1, 0), info_base(nc), start, mw_off(1, 0), rel_pos, minimisers;
    std::vector<uint8_t> even(nc);
    uint64_t cbase = 0, n_info = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const Contig& ctg = *contigs[c];
        contig_base[c - c0] = (uint32_t)cbase;
        even[c - c0] = ctg._is_win_even ? 1 : 0;
        info_base[c - c0] = (uint32_t)n_info;
        const uint64_t nb = ctg._reg_pos.count();               // set bits: 0, SR starts and ends, the length
        const size_t s0 = start.size();
        start.resize(s0 + nb);
#pragma omp parallel for schedule(static, 4096)
        for (int64_t i = 0; i < (int64_t)nb; ++i) start[s0 + (size_t)i] = (uint32_t)ctg._reg_pos.select((uint64_t)i + 1);
        reg_base[c - c0 + 1] = (uint32_t)start.size();
        for (const MWMinimiserInfo& mi : ctg._minimserinfo) {
            rel_pos.insert(rel_pos.end(), mi.rel_pos.begin(), mi.rel_pos.end());
            minimisers.insert(minimisers.end(), mi.minimisers.begin(), mi.minimisers.end());
            mw_off.push_back((uint32_t)rel_pos.size());
        }
        n_info += ctg._minimserinfo.size();
        cbase += ctg._len + (ctg._len & 1);
    }
    if (rel_pos.empty()) return true;
    if (start.size() >= 0xfffffff0ull || rel_pos.size() >= 0xfffffff0ull) return false;
    std::vector<uint32_t> cov(rel_pos.size()), sup(rel_pos.size());
    HypoMegaWindows W;
    W.n_contigs = nc; W.contig_base = contig_base.data(); W.reg_base = reg_base.data(); W.win_even = even.data(); W.info_base = info_base.data();
    W.start = start.data(); W.n_info = (uint32_t)n_info; W.mw_off = mw_off.data(); W.rel_pos = rel_pos.data(); W.minimisers = minimisers.data();
    const int rc = hypo_gpu_support_minimizers(&W, cov.data(), sup.data());
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: minimizer support is counted on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
    size_t x = 0;
    for (uint32_t c = c0; c < c1; ++c)
        for (MWMinimiserInfo& mi : contigs[c]->_minimserinfo) {
            const size_t e0 = mw_off[x], n = mi.rel_pos.size();
            for (size_t m = 0; m < n; ++m) { mi.coverage[m] = cov[e0 + m]; mi.support[m] = sup[e0 + m]; }
            ++x;
        }
    std::fprintf(stdout, "[Hypo::Hypo] Info: minimizer support counted on the device: %llu minimizers\n", (unsigned long long)rel_pos.size());
    return true;
}

bool DeviceArms::build(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1,
                       std::vector<std::vector<std::unique_ptr<Alignment>>>& store, unsigned k) {
    _active = false;
    wait_released();
    if (hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    const bool timing = std::getenv("HYPO_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = now();
    // the reads: resident since the support votes, or put there now
    if (!(_reads_resident && c0 == _reads_c0 && c1 == _reads_c1) && !upload_reads(contigs, c0, c1, store)) return false;
    _reads_resident = false;                                   // (this build consumes them: the next batch uploads its own)
    uint64_t total = 0, n_reg = 0, n_anchor = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const Contig& ctg = *contigs[c];
        total += ctg._len + (ctg._len & 1);
        n_reg += ctg.get_num_regions() + (ctg._len & 1);
        n_anchor += ctg._anchor_kmers.size();
    }
    if (total >= 0xfffffff0ull || n_reg >= 0xfffffff0ull || n_reg == 0) return false;
    std::vector<uint32_t> start(n_reg + 1), info(n_reg + 1, 0);
    std::vector<uint8_t> type(n_reg + 1, (uint8_t)RegionType::SR);
    std::vector<uint64_t> anchors; anchors.reserve(n_anchor);
    std::vector<uint8_t> contig4((total + 1) / 2, 0);
    _reg_window.assign(n_reg, nullptr);
    uint64_t base = 0, r = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        Contig& ctg = *contigs[c];
        const uint32_t nr = (uint32_t)ctg.get_num_regions();
        // Contig::_anchor_kmers is [dummy, first k-mer of SR 1, last k-mer of SR 1, first of SR 2, ...] and the SR ranks start at 1:
        // one dummy for the whole coordinate space, the ranks of later contigs shifted by the SRs before them
        if (anchors.empty()) anchors.push_back(0);
        const uint32_t sr_before = (uint32_t)((anchors.size() - 1) / 2);
        for (uint32_t i = 0; i < nr; ++i, ++r) {
            start[r] = (uint32_t)(base + ctg._reg_pos.select((uint64_t)i + 1));
            type[r] = (uint8_t)ctg._reg_type[i];
            info[r] = ctg._reg_info[i] + (ctg._reg_type[i] == RegionType::SR ? sr_before : 0u);
            _reg_window[r] = ctg._pwindows[i].get();
            if (type[r] != (uint8_t)RegionType::SR && type[r] != (uint8_t)RegionType::MSR && !_reg_window[r]) return false;   // (never: every non-SR region has a window here)
        }
        if (ctg._anchor_kmers.size() > 1) anchors.insert(anchors.end(), ctg._anchor_kmers.begin() + 1, ctg._anchor_kmers.end());
        std::memcpy(contig4.data() + base / 2, ctg._pseq.data(), ctg._pseq.byte_size());
        if (ctg._len & 1) { start[r] = (uint32_t)(base + ctg._len); type[r] = (uint8_t)RegionType::SR; info[r] = 0; ++r; }
        base += ctg._len + (ctg._len & 1);
    }
    start[r] = (uint32_t)total;
    HypoArmsRegions R;
    R.n_regions = (uint32_t)n_reg; R.start = start.data(); R.type = type.data(); R.info = info.data();
    R.n_anchor_kmers = anchors.size(); R.anchor_kmers = anchors.data(); R.k = k; R.contig4 = contig4.data();
    std::vector<uint8_t> valid(n_reg, 0);
    const auto t1 = now();
    const int rc = hypo_gpu_arms_build(&R, nullptr, valid.data(), &_sum);       // the resident reads
    const auto t2 = now();
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: short arms are computed on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
    // what Contig::fill_short_windows leaves behind (src/Contig.cpp:249-289): pruned windows are gone, the anchors are freed
    r = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        Contig& ctg = *contigs[c];
        const uint32_t nr = (uint32_t)ctg.get_num_regions();
        for (uint32_t i = 0; i < nr; ++i, ++r)
            if (ctg._pwindows[i] && !valid[r]) { ctg._pwindows[i].reset(); _reg_window[r] = nullptr; }
        if (ctg._len & 1) ++r;
        std::vector<uint64_t>().swap(ctg._anchor_kmers);
        std::vector<uint32_t>().swap(ctg._reg_info);
        _spent.emplace_back(std::move(store[c]));
        store[c].clear();
    }
    // a million small objects: they are released behind the next phases of the run (joined in the destructor)
    _releaser = std::thread([this] {
        constexpr size_t kHelpers = 24;
        for (auto& alns : _spent) {
            std::vector<std::thread> helpers;
            const size_t n = alns.size(), per = (n + kHelpers - 1) / kHelpers;
            for (size_t h = 0; h < kHelpers; ++h)
                helpers.emplace_back([&alns, h, per, n] { for (size_t t = h * per; t < n && t < (h + 1) * per; ++t) alns[t].reset(); });
            for (auto& th : helpers) th.join();
        }
        std::vector<std::vector<std::unique_ptr<Alignment>>>().swap(_spent);
    });
    if (timing) std::fprintf(stderr, "[timing] device arms: flatten %.3f s, hypo_gpu_arms_build %.3f s, prune + release %.3f s\n", secs(t0, t1), secs(t1, t2), secs(t2, now()));
    std::fprintf(stdout, "[Hypo::Hypo] Info: short arms cut on the device: %u windows, %u arms\n", _sum.n_windows, _sum.n_arms);
    _active = true;
    return true;
}

bool DeviceArms::build_long(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1,
                            std::vector<std::vector<std::unique_ptr<Alignment>>>& store) {
    _active_long = false;
    if (hypo_gpu_use_device(_slot) != HYPO_OK) return false;
    // the coordinate space of build(): contigs back to back, each starting on an even position (a 1-base filler region of type SR
    // behind an odd-length contig); its regions are the PSEUDO regions of Contig::prepare_long_windows
    uint64_t total = 0, n_reg = 0, n_aln = 0;
    std::vector<uint64_t> aln_base(c1 - c0 + 1, 0);
    for (uint32_t c = c0; c < c1; ++c) {
        const Contig& ctg = *contigs[c];
        if (ctg._pseudo_reg_type.empty()) return false;
        total += ctg._len + (ctg._len & 1);
        n_reg += ctg._pseudo_reg_type.size() - 1 + (ctg._len & 1);
        aln_base[c - c0] = n_aln;
        n_aln += store[c].size();
    }
    if (total >= 0xfffffff0ull || n_reg >= 0xfffffff0ull || n_aln >= 0xfffffff0ull || n_reg == 0) return false;
    std::vector<uint64_t> seq_off(n_aln + 1);
    std::vector<uint32_t> cigar_off(n_aln + 1);
    bool sorted = true;
    for (uint32_t c = c0; c < c1; ++c) {
        const auto& alns = store[c];
        const uint64_t a0 = aln_base[c - c0];
        for (size_t t = 0; t < alns.size(); ++t) {
            seq_off[a0 + t + 1] = alns[t]->_apseq.byte_size();
            cigar_off[a0 + t + 1] = (uint32_t)alns[t]->_cigar.size();
            if (t && alns[t - 1]->_rb > alns[t]->_rb) sorted = false;
        }
    }
    if (!sorted) { std::fprintf(stdout, "[Hypo::Hypo] Info: long-read alignments are not sorted by position: long arms are computed on the host\n"); return false; }
    seq_off[0] = 0; cigar_off[0] = 0;
    uint64_t n_cig = 0;
    for (uint64_t g = 0; g < n_aln; ++g) { n_cig += cigar_off[g + 1]; if (n_cig >= 0xfffffff0ull) return false; seq_off[g + 1] += seq_off[g]; cigar_off[g + 1] += cigar_off[g]; }
    const uint64_t read_bytes = seq_off[n_aln];
    std::vector<uint32_t> start(n_reg + 1);
    std::vector<uint8_t> type(n_reg + 1, (uint8_t)RegionType::SR);
    std::vector<uint8_t> contig4((total + 1) / 2, 0);
    std::vector<uint32_t> rb(n_aln), re(n_aln), qae(n_aln);
    std::unique_ptr<uint32_t[]> cigar(new uint32_t[n_cig ? n_cig : 1]);
    std::unique_ptr<uint8_t[]> reads2(new uint8_t[read_bytes ? read_bytes : 1]);
    _preg_window.assign(n_reg, nullptr);
    uint64_t base = 0, r = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        Contig& ctg = *contigs[c];
        const size_t np = ctg._pseudo_reg_type.size() - 1;       // the last pseudo region is the end marker at the contig's length
        for (size_t i = 0; i < np; ++i, ++r) {
            start[r] = (uint32_t)(base + ctg._pseudo_reg_pos.select((uint64_t)i + 1));
            const bool lw = ctg._pseudo_reg_type[i] == RegionType::LONG;
            type[r] = (uint8_t)(lw ? RegionType::LONG : RegionType::SR);
            if (lw) { _preg_window[r] = ctg._pwindows[ctg._true_reg_id[i]].get(); if (!_preg_window[r]) return false; }
        }
        std::memcpy(contig4.data() + base / 2, ctg._pseq.data(), ctg._pseq.byte_size());
        if (ctg._len & 1) { start[r] = (uint32_t)(base + ctg._len); type[r] = (uint8_t)RegionType::SR; ++r; }
        const uint64_t a0 = aln_base[c - c0];
        auto& alns = store[c];
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)alns.size(); ++t) {
            const Alignment& a = *alns[(size_t)t];
            const uint64_t g = a0 + (uint64_t)t;
            rb[g] = (uint32_t)(base + a._rb); re[g] = (uint32_t)(base + a._re); qae[g] = a._qae;
            std::memcpy(reads2.get() + seq_off[g], a._apseq.data(), a._apseq.byte_size());
            std::memcpy(cigar.get() + cigar_off[g], a._cigar.data(), a._cigar.size() * 4);
        }
        base += ctg._len + (ctg._len & 1);
    }
    start[r] = (uint32_t)total;
    HypoArmsRegions R;
    R.n_regions = (uint32_t)n_reg; R.start = start.data(); R.type = type.data(); R.info = nullptr;
    R.n_anchor_kmers = 0; R.anchor_kmers = nullptr; R.k = 10; R.contig4 = contig4.data();
    HypoArmsReads A;
    A.n_alignments = (uint32_t)n_aln; A.rb = rb.data(); A.re = re.data(); A.qae = qae.data(); A.seq_off = seq_off.data();
    A.reads2 = reads2.get(); A.reads2_bytes = read_bytes; A.cigar_off = cigar_off.data(); A.cigar = cigar.get();
    std::vector<uint8_t> valid(n_reg, 0);
    const int rc = hypo_gpu_arms_build_long(&R, &A, valid.data(), &_sum_long);
    if (rc != HYPO_OK) {
        if (rc != HYPO_E_UNSUPPORTED) std::fprintf(stdout, "[Hypo::Hypo] Info: long arms are computed on the host (%s)\n", hypo_gpu_last_error());
        return false;
    }
    // what Contig::fill_long_windows leaves behind (include/Contig.hpp:91-113): the alignments are spent, the pseudo tables gone
    for (uint32_t c = c0; c < c1; ++c) {
        Contig& ctg = *contigs[c];
        store[c].clear();
        ctg._pseudo_reg_pos.clear();
        std::vector<RegionType>().swap(ctg._pseudo_reg_type);
        std::vector<uint32_t>().swap(ctg._true_reg_id);
    }
    std::fprintf(stdout, "[Hypo::Hypo] Info: long arms cut on the device: %u windows, %u arms\n", _sum_long.n_windows, _sum_long.n_arms);
    _active_long = true;
    return true;
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
    if (!_active) return HYPO_OK;
    _active = false;
    return polish_impl(false, sp, keep_arms, retry);
}
int DeviceArms::polish_long(const ScoreParams& sp, bool keep_arms, std::vector<Window*>* retry) {
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
    std::vector<char> bases(_sum.out_bytes ? _sum.out_bytes : 1);
    std::vector<uint64_t> off((size_t)n + 1);
    std::vector<uint32_t> len(n), win_region(n);
    std::vector<uint8_t> st(n);
    std::vector<HypoWindow> hw(n);
    const bool timing = std::getenv("HYPO_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = now();
    int rc = (lng ? hypo_gpu_arms_poa_long : hypo_gpu_arms_poa)(&sp, bases.data(), off.data(), len.data(), st.data());
    if (rc != HYPO_OK) return rc;
    const auto t1 = now();
    rc = (lng ? hypo_gpu_arms_download_long : hypo_gpu_arms_download)(hw.data(), win_region.data(), nullptr, nullptr, nullptr, nullptr);
    if (rc != HYPO_OK) return rc;
    if (timing) std::fprintf(stderr, "[timing] device arms: buffers %.3f s, hypo_gpu_arms_poa%s %.3f s, descriptors %.3f s\n", secs(tp, t0), lng ? "_long" : "", secs(t0, t1), secs(t1, now()));
    // (every status byte starts as HYPO_ST_UNWRITTEN on the device: a window no kernel answered is an internal error, not a retry)
    for (uint32_t i = 0; i < n; ++i)
        if (st[i] == HYPO_ST_UNWRITTEN) {
            std::fprintf(stderr, "[Hypo::Window] Error: resident window %u of %u came back unanswered by the device (status 0xff)\n", i, n);
            return HYPO_E_INVALID;
        }
    std::vector<uint32_t> again, all;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i)
        if (st[(size_t)i] == HYPO_ST_OK) _reg_window[win_region[(size_t)i]]->_consensus.assign(bases.data() + off[(size_t)i], len[(size_t)i]);
    for (uint32_t i = 0; i < n; ++i) { if (st[i] != HYPO_ST_OK) again.push_back(i); if (keep_arms) all.push_back(i); }
    if (keep_arms) adopt_arms(all, hw, win_region, lng);
    else if (!again.empty()) adopt_arms(again, hw, win_region, lng);
    if (!again.empty()) {          // a consensus longer than its slot, a window beyond the size classes: the host's retry / degraded path
        std::vector<Window*> ws;
        for (uint32_t i : again) ws.push_back(_reg_window[win_region[i]]);
        if (retry) retry->insert(retry->end(), ws.begin(), ws.end());
        else rc = Window::generate_consensus_batch(ws);
    }
    return rc;
}

}  // namespace hypo
