// arms_kernel.hpp — device arm selection (arms_kernel.hip): the flat inputs (regions of a contig, its short-read alignments
// in file order) and the per-region / per-window outputs.  All pointers are DEVICE pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/hypo_gpu.h"

namespace hypo {

struct ArmsIn {
    // regions (Contig::_reg_pos / _reg_type / _reg_info / _anchor_kmers, include/Contig.hpp:148-176)
    uint32_t n_regions;
    const uint32_t* reg_start;      // [n_regions + 1]: start of every region, then the contig length
    const uint8_t* reg_type;        // [n_regions + 1] RegionType (the entry of the end marker is never an SR)
    const uint32_t* reg_info;       // [n_regions + 1]: SR -> its rank, MSR -> its minimizer
    const uint64_t* anchor_kmers;   // first / last k-mer of every SR
    uint32_t k;
    const uint8_t* contig4;         // PackedSeq<4> of the contig
    // alignments, sorted by reference start (= file order of a coordinate-sorted BAM)
    uint32_t n_alignments;
    const uint32_t* rb;             // reference span [rb, re)
    const uint32_t* re;
    const uint32_t* qae;            // aligned query length (soft clips dropped)
    const uint64_t* seq_off;        // BYTE offset of the read's PackedSeq<2> in reads2
    const uint8_t* reads2;
    const uint32_t* cigar_off;      // [n_alignments + 1]
    const uint32_t* cigar;          // BAM encoding: len << 4 | op
    const uint32_t* file_rank;      // NULL, or the alignments' positions in the file when that is not the order they are given in
    uint32_t max_span;              // max(re - rb)
    uint32_t long_mode;             // 1: long reads over pseudo regions -> LONG windows (no anchors; Filter::is_good decides what stays)
};

struct ArmsOut {
    // per region
    uint32_t* reg_flags;            // bit 0: a valid window; bit 1: prefix / suffix arms kept
    uint32_t* reg_valid;            // 0 / 1 (scan input; downloaded by the host)
    uint4* reg_counts;              // internal, prefix, suffix, empty (after pruning)
    uint32_t* reg_arms;             // arms kept
    uint32_t* reg_bytes;            // bytes of the kept arms (each arm on a byte boundary)
    uint32_t* reg_bytes_int;
    uint32_t* reg_bytes_pre;
    uint32_t* reg_draft_bytes;
    uint32_t* reg_slot;             // consensus slot bytes
    const uint64_t* reg_arm_off;    // exclusive scans of the above
    const uint64_t* reg_byte_off;
    const uint64_t* reg_draft_off;
    const uint64_t* reg_slot_off;
    uint32_t* win_index;            // region -> window of the batch
    // the batch (include/hypo_gpu.h: HypoWindowBatch) and the consensus slots
    HypoWindow* windows;
    uint32_t* win_region;           // window -> region
    uint32_t* arm_len;
    uint64_t* arm_off;
    uint8_t* arms2;
    uint8_t* draft4;
    uint64_t* out_off;              // [n_windows] (the host appends the total)
    // long mode: the window minimizers of every LONG window's draft (Filter::initialise), sorted, at reg_min_off[w] .. + reg_min_cnt[w]
    uint32_t* reg_min_len;          // per region: slots it needs (its length for a LONG window, else 0); scan input
    const uint64_t* reg_min_off;
    uint32_t* reg_min_cnt;
    uint32_t* draft_min;
};

hipError_t scan32(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* bsum, uint64_t* total, hipStream_t st);
size_t scan32_scratch_bytes(uint64_t n);
hipError_t arms_phase1(const ArmsIn& I, uint32_t* b_ind, uint32_t* ntouch, uint32_t* bad, hipStream_t st);
hipError_t arms_phase2(const ArmsIn& I, const uint32_t* b_ind, const uint32_t* ntouch, const uint64_t* touch_off, uint32_t* bp, uint2* cand,
                       const ArmsOut& O, hipStream_t st);
// long mode, between phase 1's scan and phase 2: per-region minimizer slots (a), then the drafts' minimizers (b)
hipError_t arms_long_minlen(const ArmsIn& I, const ArmsOut& O, hipStream_t st);
hipError_t arms_long_draftmin(const ArmsIn& I, const ArmsOut& O, hipStream_t st);
hipError_t arms_phase3(const ArmsIn& I, const uint32_t* b_ind, const uint32_t* ntouch, const uint64_t* touch_off, const uint2* cand,
                       const ArmsOut& O, const uint64_t* win_off, hipStream_t st);

}  // namespace hypo
