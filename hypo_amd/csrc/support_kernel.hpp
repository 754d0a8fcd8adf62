// support_kernel.hpp — support votes of the short reads on the device (support_kernel.hip).  All pointers are DEVICE pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace hypo {

struct SupportReads {                // the resident short reads of a contig batch (hypo_gpu_reads_upload)
    uint32_t n_alignments;
    const uint32_t* rb;              // reference span [rb, re) in the batch's coordinate space
    const uint32_t* re;
    const uint32_t* qae;             // aligned query length
    const uint64_t* seq_off;         // byte offset of the aligned query (PackedSeq<2>) in reads2
    const uint8_t* reads2;
    const uint32_t* read_contig;     // contig of the batch the read maps to
    uint32_t mean_span;              // mean reference span of the reads (block shapes of the vote kernels)
};

struct MegaWindows {                 // Contig::_reg_pos / _is_win_even / _minimserinfo after prepare_for_division, all contigs of the batch
    const uint32_t* contig_base;     // [n_contigs] start of the contig in the coordinate space
    const uint32_t* reg_base;        // [n_contigs + 1] first entry of the contig in `start`
    const uint8_t* win_even;         // [n_contigs] Contig::_is_win_even
    const uint32_t* info_base;       // [n_contigs] first MWMinimiserInfo of the contig
    const uint32_t* start;           // region borders, contig-local (the set bits of _reg_pos: 0, SR starts and ends, the length)
    const uint32_t* mw_off;          // [n_info + 1] minimizers of every mega-window: entries mw_off[x] .. mw_off[x + 1]
    uint32_t* rel_pos;               // per entry: distance from the previous minimizer (from the window's start for the first); support_minimizers()
                                     // turns it into the contig-local position itself, in place, before the reads are walked
    const uint32_t* minimisers;      // per entry: the k-mer
};

hipError_t support_kmers(const SupportReads& R, uint32_t k, uint32_t n_solid, const uint32_t* spos, const uint64_t* kids, uint32_t* cov, uint32_t* sup, hipStream_t st);
hipError_t support_kmers32(const SupportReads& R, uint32_t k, uint32_t n_solid, const uint32_t* spos, const uint32_t* kids32, uint32_t* cov, uint32_t* sup, hipStream_t st);   // k <= 16
hipError_t add_base(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t base, hipStream_t st);
hipError_t support_minimizers(const SupportReads& R, const MegaWindows& M, uint32_t n_contigs, uint32_t n_info, uint32_t* cov, uint32_t* sup, hipStream_t st);

}  // namespace hypo
