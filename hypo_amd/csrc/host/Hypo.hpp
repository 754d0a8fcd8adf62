// Hypo.hpp — host mirror of hypo::Hypo (reference: include/Hypo.hpp:37-58, src/Hypo.cpp:37-329): phase sequencing of
// one polishing run.  Same phases and log labels; the two hot-path phases call the MI355X through the C-ABI:
// "Found Solid pos in contigs" (Contig::find_solid_pos -> hypo_gpu_solid_scan) and "POA of windows"
// (all valid windows of a contig batch in ONE Window::generate_consensus_batch call).
#pragma once
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>
#include "Alignment.hpp"
#include "Contig.hpp"
#include "ReadBatch.hpp"
#include "SeqIO.hpp"
#include "Settings.hpp"

namespace hypo {

struct PhaseTimes { std::vector<std::pair<std::string, double>> phases; double overall = 0; };

class Hypo {
public:
    explicit Hypo(const InputFlags& flags);
    void polish();
    const PhaseTimes& times() const { return _times; }
    // test hook: windows of every contig after arm filling, as (beg, end, type, ni, np, ns, ne, crc32 of arms)
    void set_region_dump(const std::string& path) { _region_dump = path; }

private:
    const InputFlags _cFlags;
    std::vector<std::unique_ptr<Contig>> _contigs;
    std::unordered_map<std::string, uint32_t> _cname_to_id;
    using AlignmentStore = std::vector<std::vector<std::unique_ptr<Alignment>>>;
    AlignmentStore _alignment_store;
    uint32_t _contig_batch_size = 0;
    std::unique_ptr<SamReader> _sf_short, _sf_long;
    // per alignment file: the block of raw records being consumed (a contig batch may end in the middle of it), the block a
    // reader thread fetched meanwhile, and whether the file has more
    // One long-lived thread per alignment file that fetches the next block of raw records when asked (BGZF: inflates a run of blocks
    // on its own team of threads, which lives as long as it does: a thread per block had to raise a new team of 64-128 threads 1 300
    // times on the 3 Gbp set).
    struct BlockReader {
        std::thread th; std::mutex mu; std::condition_variable cv;
        SamReader* sf = nullptr; SamReader::RecordBlock* dst = nullptr; size_t max_rec = 0, max_bytes = 0;
        bool pending = false, done = false, quit = false, more = false;
        void start(SamReader* s) {
            sf = s;
            th = std::thread([this] {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    cv.wait(lk, [this] { return pending || quit; });
                    if (quit) return;
                    lk.unlock();
                    const bool m = sf->read_block(*dst, max_rec, max_bytes);
                    lk.lock();
                    more = m; pending = false; done = true;
                    cv.notify_all();
                }
            });
        }
        void request(SamReader::RecordBlock* d, size_t n, size_t b) {
            std::lock_guard<std::mutex> lk(mu);
            dst = d; max_rec = n; max_bytes = b; pending = true; done = false;
            cv.notify_all();
        }
        bool wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return done; }); done = false; return more; }
        ~BlockReader() { { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); } if (th.joinable()) th.join(); }
    };
    struct RecordStream {
        std::unique_ptr<BlockReader> reader;
        SamReader::RecordBlock cur, ahead; size_t pos = 0; bool have_ahead = false, more = true;
        // flat path (short reads): the parsed form of `cur`, how far it has been consumed, and the one record a call consumed for
        // a contig of a later batch (the reference files it in that contig's store entry, src/Hypo.cpp:314-325)
        std::shared_ptr<ParsedBlock> parsed; size_t ppos = 0;
        std::shared_ptr<ParsedBlock> carry_blk; size_t carry_r0 = 0, carry_r1 = 0; int32_t carry_cid = -1;
        int32_t opened_with_cid = -1;      // the batch loaded last began with the record carried over from the call before, of this contig (-1: it did not)
    };
    RecordStream _rs_short, _rs_long;
    PhaseTimes _times;
    std::string _region_dump;
    std::chrono::steady_clock::time_point _t0, _tstart;

    void start() { _t0 = std::chrono::steady_clock::now(); }
    void stop(const char* label);
    // Round 4: the short reads of a contig batch as flat slices (ReadBatch.hpp) instead of one Alignment object per record
    ReadBatch _reads;
    ReadBatch _reads_long;                   // the long reads of the batch in hand (-B), flat as well
    std::vector<std::shared_ptr<ParsedBlock>> _block_pool; std::mutex _pool_mu;
    void create_alignments_flat(uint32_t batch_id, ReadBatch& into, bool is_sr = true);
    // long_reads: the normalised-edit-distance filter of the long-read constructor applies (src/Alignment.cpp:51-58)
    void parse_block(const SamReader& sf, const SamReader::RecordBlock& raw, ParsedBlock& blk, bool long_reads);
    // Alignment objects of contigs [c0, c1) from _reads into _alignment_store (the host loops of the reference read those)
    void materialize_alignments(uint32_t c0, uint32_t c1, std::vector<char>& done);
    uint32_t _mat_base = 0;
};

}  // namespace hypo
