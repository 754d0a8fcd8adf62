// Contig.hpp — host mirror of hypo::Contig (reference: include/Contig.hpp:62-218, src/Contig.cpp).
// Same public surface: find_solid_pos, prepare_for_division, divide_into_regions, fill_short_windows,
// prepare_long_windows, fill_long_windows, get_num_regions / num_sr / len_sr, is_valid_window, generate_consensus,
// set_no_long_reads, operator<<.  The solid scan runs on the MI355X (hypo_gpu_solid_scan); the per-window POA is
// batched by Hypo::polish through Window::generate_consensus_batch.  sdsl's bit vectors are BitVec; the per-k-mer
// mutexes of the reference are atomic adds.
#pragma once
#include <cstdio>
#include <cstdint>
#include <iosfwd>
#include <memory>
#include <string>
#include <vector>
#include "../../../include/hypo_gpu.h"
#include "Alignment.hpp"
#include "BitVec.hpp"
#include "PackedSeq.hpp"
#include "Settings.hpp"
#include "Window.hpp"

namespace hypo {

// what suk::SolidKmers exposes to the path (external/suk/include/suk/SolidKmers.hpp:108-140)
struct SolidKmers {
    uint32_t k = 0;
    std::vector<uint64_t> words;              // 4^k bits, bit i at words[i >> 6] bit (i & 63)
    uint64_t num_solid = 0;
    uint32_t get_k() const { return k; }
    bool is_solid(uint64_t kid) const { return (words[kid >> 6] >> (kid & 63)) & 1; }
    // sdsl::store_to_file / load_from_file layout of a bit_vector: uint64 bit count, then the 64-bit words
    bool load(const std::string& path);
    bool store(const std::string& path) const;
};

struct MWMinimiserInfo {                       // include/Contig.hpp:46-52
    std::vector<uint32_t> minimisers, rel_pos;
    std::vector<uint32_t> support, coverage;   // 16-bit counters in the reference: read through & 0xffff
};

class Contig {
public:
    Contig(uint32_t id, const std::string& name, const std::string& seq);

    int find_solid_pos(const SolidKmers& sk, bool set_on_device = false);                        // device scan; HYPO_OK or C-ABI error
    void adopt_solid_scan(const uint64_t* words, const uint64_t* rank, const uint64_t* kids, uint64_t n_solid);
    // test hook (HYPO_DUMP_VOTES): the vote counters as they stand — which = 0: KmerInfo coverage / support of the solid k-mers (before
    // prepare_for_division spends them), 1: MWMinimiserInfo coverage / support (before divide_into_regions does)
    void dump_votes(std::FILE* f, int which) const;
    void prepare_for_division(unsigned k);
    void divide_into_regions();
    void fill_short_windows(std::vector<std::unique_ptr<Alignment>>& alignments);
    void prepare_long_windows();
    void fill_long_windows(std::vector<std::unique_ptr<Alignment>>& alignments);
    // Alignment::add_arms for all alignments, in file order per window, on all threads (each thread owns a window range)
    void add_arms_by_window_range(std::vector<std::unique_ptr<Alignment>>& alignments);

    uint64_t get_num_regions() const { return _reg_type.size() - 1; }
    uint64_t get_num_sr() const { return _numSR; }
    uint64_t get_len_sr() const { return _lenSR; }
    bool is_valid_window(uint32_t ind) const { return _pwindows[ind] != nullptr; }
    void generate_consensus(uint64_t ind, uint32_t th) { _pwindows[ind]->generate_consensus(th); }
    Window* window(uint32_t ind) const { return _pwindows[ind].get(); }
    static void set_no_long_reads() { _no_long_reads = true; }
    // the polished record has been written: windows, region tables and the packed draft go; name and length stay (the record parser
    // checks later alignments against them)
    void release_after_output();
    friend std::ostream& operator<<(std::ostream&, const Contig&);
    friend class Alignment;
    friend class DeviceArms;

    // scan results (Contig::_solid_pos / _kmerinfo[i]->kid)
    uint64_t get_num_solid() const { return _n_solid; }
    bool is_solid_pos(uint64_t p) const { return _solid_pos[p]; }
    uint64_t kid_at(uint64_t i) const { return _kids.empty() ? kmer_at((uint64_t)_solid_pos.select(i + 1), _scan_k) : _kids[i]; }
    // the k-mer that starts at position p (what the scan stored as Contig::_kmerinfo[i]->kid for a marked p: no N among its bases)
    uint64_t kmer_at(uint64_t p, unsigned k) const { uint64_t v = 0; for (unsigned i = 0; i < k; ++i) v = (v << 2) | (_pseq.enc_base_at(p + i) & 3u); return v; }
    // the k-mer ids of the marked positions as a host array (a resident scan, hypo_gpu_solid_scan_keep, leaves them on the device:
    // the host loops of the reference that read them — Alignment::update_solidkmers_support — ask for them here first)
    void ensure_kids();
    bool scan_kept() const { return _scan_kept; }
    uint64_t rank(uint64_t p) const { return _solid_pos.rank(p); }
    uint64_t select(uint64_t i) const { return _solid_pos.select(i); }
    const std::string& get_name() const { return _name; }
    std::string draft_segment(uint32_t beg, uint32_t end) const { return _pseq.unpack(beg, end); }
    uint64_t get_len() const { return _len; }
    // region map for diagnostics / tests: (begin, end, type) of region i
    void region(uint32_t i, uint32_t& beg, uint32_t& end, RegionType& t) const { beg = (uint32_t)_reg_pos.select(i + 1); end = (uint32_t)_reg_pos.select(i + 2); t = _reg_type[i]; }

private:
    uint32_t _id;
    std::string _name;
    uint32_t _len;
    PackedSeq<4> _pseq;
    BitVec _solid_pos;
    std::vector<uint64_t> _kids;                 // empty after a resident scan (ensure_kids() fills it)
    uint64_t _n_solid = 0; unsigned _scan_k = 0; bool _scan_kept = false;
    std::vector<uint32_t> _kcov, _ksup;          // KmerInfo::coverage / support (UINT16 in the reference)
    std::vector<uint64_t> _anchor_kmers;
    BitVec _reg_pos;
    bool _is_win_even = true;
    bool _mreg_ready = false;
    std::vector<MWMinimiserInfo> _minimserinfo;
    std::vector<RegionType> _reg_type;
    std::vector<uint32_t> _reg_info;
    std::vector<std::unique_ptr<Window>> _pwindows;
    uint64_t _numSR = 0, _lenSR = 0;
    BitVec _pseudo_reg_pos;
    std::vector<RegionType> _pseudo_reg_type;
    std::vector<uint32_t> _true_reg_id;
    static bool _no_long_reads;

    void increment_support(uint32_t ind) { __atomic_fetch_add(&_ksup[ind], 1u, __ATOMIC_RELAXED); }
    void increment_coverage(uint32_t ind) { __atomic_fetch_add(&_kcov[ind], 1u, __ATOMIC_RELAXED); }
    void increment_minimser_support(uint32_t mi, uint32_t idx) { __atomic_fetch_add(&_minimserinfo[mi].support[idx], 1u, __ATOMIC_RELAXED); }
    void increment_minimser_coverage(uint32_t mi, uint32_t idx) { __atomic_fetch_add(&_minimserinfo[mi].coverage[idx], 1u, __ATOMIC_RELAXED); }
    void initialise_minimserinfo(const std::string& draft_seq, uint32_t minfoind);
    void divide(uint32_t reg_index, uint32_t beg, uint32_t end, char pvs, char nxt);
    void force_divide(uint32_t beg, uint32_t end, char pvs, char nxt);
};

}  // namespace hypo
