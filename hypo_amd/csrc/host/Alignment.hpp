// Alignment.hpp — host mirror of hypo::Alignment (reference: include/Alignment.hpp:47-93, src/Alignment.cpp).
// One mapped read: its reference span, the 2-bit packed aligned part of the query (soft clips dropped), its CIGAR;
// votes support/coverage for solid k-mers and window minimizers and is cut at region borders into arms.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "PackedSeq.hpp"
#include "SeqIO.hpp"
#include "Settings.hpp"

namespace hypo {

class Contig;
class BitVec;

enum class ArmType : uint8_t { INTERNAL, PREFIX, SUFFIX, EMPTY };

struct Arm {
    uint32_t windex;
    PackedSeq<2> arm;
    ArmType armtype;
    Arm(uint32_t ind, const PackedSeq<2>& ps, uint32_t left, uint32_t right, ArmType at) : windex(ind), arm(ps, left, right), armtype(at) {}
    explicit Arm(uint32_t ind) : windex(ind), armtype(ArmType::EMPTY) {}
};

class Alignment {
public:
    Alignment(Contig& contig, const SamRecord& rec);                          // short read
    Alignment(Contig& contig, uint64_t norm_edit_th, const SamRecord& rec);  // long read (normalised edit distance filter)
    // from the flat record a ReadBatch holds (ReadBatch::materialize): span, aligned query length, packed bases, CIGAR
    Alignment(uint32_t rb, uint32_t re, uint32_t qae, const uint8_t* seq2, const uint32_t* cigar, size_t n_cigar)
        : _rb(rb), _re(re), _qab(0), _qae(qae), _apseq(seq2, qae), _cigar(cigar, cigar + n_cigar) {}
    // What the two constructors above compute before they copy anything (Alignment.cpp:514-549 + the bounds check of :31-36, fatal):
    // reference span and the aligned part [qab, qae) of the query.  Used by the flat parser of Hypo::create_alignments_flat.
    static void span_of(const Contig& contig, const SamRecord& rec, uint32_t& rb, uint32_t& re, uint32_t& qab, uint32_t& qae);
    Alignment(const Alignment&) = delete;
    Alignment& operator=(const Alignment&) = delete;

    bool is_valid = true;

    void update_solidkmers_support(unsigned k, Contig& contig);
    void update_minimisers_support(Contig& contig);
    void find_short_arms(unsigned k, Contig& contig);
    void find_long_arms(Contig& contig);
    void add_arms(const Contig& contig);
    // the arms that belong to windows [w0, w1) only (their bytes move into the windows) (Contig::add_arms_by_window_range)
    void add_arms(const Contig& contig, uint32_t w0, uint32_t w1);
    bool arm_span(uint32_t& first, uint32_t& last) const {          // smallest / largest window index among the arms
        if (_arms.empty()) return false;
        first = last = _arms[0].windex;
        for (const Arm& a : _arms) { first = a.windex < first ? a.windex : first; last = a.windex > last ? a.windex : last; }
        return true;
    }

private:
    friend class DeviceArms;
    friend class ReadBatch;
    uint32_t _rb = 0, _re = 0, _qab = 0, _qae = 0;
    PackedSeq<2> _apseq;
    std::vector<uint32_t> _cigar;
    std::vector<Arm> _arms;

    void initialise_pos(const SamRecord& rec);
    void copy_data(const SamRecord& rec);
    std::vector<uint32_t> find_bp(const BitVec& reg_pos, const std::vector<RegionType>& reg_type, uint32_t beg_ind, uint32_t end_ind) const;
    void prepare_short_arm(unsigned k, uint32_t windex, uint32_t qb, uint32_t qe, ArmType armtype, Contig& contig);
    static void check_bounds(const Contig& contig, const SamRecord& rec, uint32_t rb, uint32_t re);
};

}  // namespace hypo
