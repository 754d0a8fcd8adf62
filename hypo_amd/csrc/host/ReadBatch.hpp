// ReadBatch.hpp — the short-read alignments of a contig batch as flat arrays (round 4).
//
// The reference keeps one heap object per mapped read (std::unique_ptr<Alignment> in Hypo::_alignment_store, src/Hypo.cpp:314-318)
// and so did rounds 1-3 here: 20 M objects on the 100 Mbp set, each with a PackedSeq and a CIGAR vector of its own, built by the
// parser threads, walked once to be flattened for hypo_gpu_reads_upload, and released again (a helper thread per batch, which the
// next batch had to wait for).  What the device path reads of a record is five numbers, its packed bases and its CIGAR: the parser
// threads now write exactly that, back to back, into per-thread chunks of the block of records in hand (ReadChunk / ParsedBlock);
// a contig batch is a list of slices of such blocks (ReadBatch), and flatten() lays them out — per contig, in file order — in the
// caller's page-locked staging arrays, which go to the device as they are.  Alignment objects are made from the same slices only
// when a host loop of the reference has to run (materialize(): --host-arms, an unsorted file, a device error, the CPU test shim).
#pragma once
#include <sys/mman.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../../include/hypo_gpu.h"
#include "Alignment.hpp"

namespace hypo {

class Contig;

struct ReadChunk {                           // the KEPT records of one stretch of a block, in file order (filled by one thread)
    std::vector<uint32_t> raw;               // index of the record within its block
    std::vector<int32_t> cid;                // contig id
    std::vector<uint32_t> rb, re, qae;       // contig-local reference span, aligned query length (Alignment::_rb / _re / _qae)
    std::vector<uint32_t> seq_at, cig_at;    // where the record's packed bases / CIGAR start in seq / cig (n + 1 entries)
    std::vector<uint8_t> seq;                // PackedSeq<2> bytes of the aligned part, each record on a byte boundary
    std::vector<uint32_t> cig;               // op | len << 4
    size_t n() const { return raw.size(); }
    void clear() { raw.clear(); cid.clear(); rb.clear(); re.clear(); qae.clear(); seq_at.assign(1, 0); cig_at.assign(1, 0); seq.clear(); cig.clear(); }
};

struct ParsedBlock {
    std::vector<ReadChunk> chunks;           // in file order
    std::vector<uint8_t> status;             // per raw record: ST_*
    std::vector<int32_t> cid;                // per raw record (-1: skipped or unknown reference)
    std::string bad_ref_name;                // read name of the first record with an unknown reference
    size_t n = 0;
    enum : uint8_t { ST_KEPT = 0, ST_SKIPPED = 1, ST_BADREF = 2, ST_INVALID = 3 };
};

// A grow-only buffer, page-locked (hypo_gpu_host_alloc) between 4 and 64 MB: copies from / into it run at the link's rate.
// Small ones are plain memory — pinning has a per-call cost (a millisecond or more) that a 5 Mbp run, with two dozen staging arrays,
// would pay for nothing.  LARGE ones are plain memory as well: page-locking costs 0.18 s per GB and 0.12 s per GB to undo (MI355X
// box, profiles/history/r04_pin_bench.txt) — the 3.9 GB of read staging of the 250 Mbp set took 0.84 s to lock for an upload of 0.17 s —
// while the library stages a large copy out of ordinary memory through its own bounce buffers at the same rate (capi.hip, h2d).
// Plain memory it is, too, when the library has no pinned memory to give (the CPU test shim).
struct PinnedBuf {
    void* p = nullptr; size_t cap = 0; bool pinned = false, registered = false; unsigned uses = 0;
    template <class T> T* get(size_t n) {
        const size_t bytes = n * sizeof(T) + 64;
        // a large buffer that is used AGAIN (the second contig batch of a run) is page-locked where it lies: 0.045 s per GB once,
        // and every later upload out of it runs at the link's rate instead of through the bounce buffers (0.8 GB of reads per 50 Mbp
        // batch of the 3 Gbp run: 55 -> 15 ms)
        if (bytes <= cap && !pinned && !registered && cap > ((size_t)64 << 20) && ++uses == 2 && !std::getenv("HYPO_NO_REGISTER"))
            registered = hypo_gpu_host_register(p, cap) == HYPO_OK;
        if (bytes > cap) {
            release();
            const size_t want = bytes + bytes / 4;
            void* q = nullptr;
            if (want >= ((size_t)4 << 20) && want <= ((size_t)64 << 20) && hypo_gpu_host_alloc(want, &q) == HYPO_OK && q) { p = q; pinned = true; }
            else if (want > ((size_t)64 << 20)) {            // (2 MB pages where the system hands them out: a fresh GB is 512 page faults, not 262 144)
                if (::posix_memalign(&q, (size_t)2 << 20, (want + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1)) != 0) q = nullptr;
                if (q) (void)::madvise(q, want, MADV_HUGEPAGE);
                p = q; pinned = false;
            } else { p = std::malloc(want); pinned = false; }
            cap = p ? want : 0;
            uses = 1;
        }
        return (T*)p;
    }
    void release() {
        if (p) { if (registered) (void)hypo_gpu_host_unregister(p); if (pinned) (void)hypo_gpu_host_free(p); else std::free(p); }
        p = nullptr; cap = 0; registered = false; uses = 0;
    }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { release(); }
};

// Page-locked staging arrays of one device context (hypo_gpu_host_alloc; grow-only, reused by every batch): what
// hypo_gpu_reads_upload takes.
struct ReadStaging {
    uint32_t *rb = nullptr, *re = nullptr, *qae = nullptr, *ctg = nullptr, *cigar_off = nullptr, *cigar = nullptr;
    uint64_t* seq_off = nullptr; uint8_t* reads2 = nullptr;
    uint32_t* file_rank = nullptr; bool ranked = false;      // ranked: the records were NOT sorted in the file; they are here, and file_rank says where each one was
    uint64_t n_reads = 0, n_cigar = 0, n_bytes = 0;
    double reserve_seconds = 0;                                      // time spent growing the buffers, all calls
    bool reserve(size_t reads, size_t cig, size_t bytes);          // false: no memory
private:
    PinnedBuf _b[9];
};

class ReadBatch {
public:
    struct Slice { std::shared_ptr<ParsedBlock> blk; uint32_t chunk, k0, k1; };
    void reset(size_t n_contigs) { _slices.clear(); _per_contig.assign(n_contigs, 0); _n = 0; }
    // the kept records among raw records [r0, r1) of blk
    void add(const std::shared_ptr<ParsedBlock>& blk, size_t r0, size_t r1);
    void append(ReadBatch& other);                                  // other's slices behind this one's (other is left empty)
    // Alignment objects of contig cid (in their order) in FRONT of everything the batch holds
    // (at_slice > 0: in front of slice `at_slice` instead — behind the one record the loader carried over from the call before)
    void prepend(uint32_t cid, const std::vector<std::unique_ptr<Alignment>>& objs, size_t at_slice = 0);
    size_t n_slices() const { return _slices.size(); }
    uint64_t size() const { return _n; }
    uint64_t count(uint32_t cid) const { return cid < _per_contig.size() ? _per_contig[cid] : 0; }
    bool empty() const { return _slices.empty(); }
    uint32_t max_span(uint32_t cid) const;                          // longest reference span among the records of contig cid
    // The records of contigs [c0, c1) in `out`: contig after contig (contig c starts at base[c - c0] of the coordinate space), each
    // contig's records in file order.  `sorted` = every contig's records came with non-decreasing rb; when they did not they are
    // sorted by rb here (stably) and out.file_rank holds their places in the file (out.ranked).  false: no memory.
    // span != nullptr (one contig only): the records that overlap [span[0], span[1]) of the contig — a device context that owns a
    // coordinate range of a large contig takes the reads its windows, k-mers and minimizers can see.
    bool flatten(uint32_t c0, uint32_t c1, const std::vector<uint64_t>& base, ReadStaging& out, bool& sorted, const uint32_t* span = nullptr) const;
    // Alignment objects for contig cid (the reference's store entry), appended to `into`
    void materialize(uint32_t cid, std::vector<std::unique_ptr<Alignment>>& into) const;
    // the records of contigs >= first_cid as a batch of their own (copied into a block of its own): what a later batch inherits
    void carry_beyond(uint32_t first_cid, ReadBatch& into) const;
    void clear(std::vector<std::shared_ptr<ParsedBlock>>* pool = nullptr, std::mutex* pool_mu = nullptr);
private:
    std::vector<Slice> _slices;
    std::vector<uint64_t> _per_contig;
    uint64_t _n = 0;
};

}  // namespace hypo
