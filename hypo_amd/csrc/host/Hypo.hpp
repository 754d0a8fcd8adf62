// Hypo.hpp — host mirror of hypo::Hypo (reference: include/Hypo.hpp:37-58, src/Hypo.cpp:37-329): phase sequencing of
// one polishing run.  Same phases and log labels; the two hot-path phases call the MI355X through the C-ABI:
// "Found Solid pos in contigs" (Contig::find_solid_pos -> hypo_gpu_solid_scan) and "POA of windows"
// (all valid windows of a contig batch in ONE Window::generate_consensus_batch call).
#pragma once
#include <chrono>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#include "Alignment.hpp"
#include "Contig.hpp"
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
    struct RecordStream { SamReader::RecordBlock cur, ahead; size_t pos = 0; bool have_ahead = false, more = true; };
    RecordStream _rs_short, _rs_long;
    PhaseTimes _times;
    std::string _region_dump;
    std::chrono::steady_clock::time_point _t0, _tstart;

    void start() { _t0 = std::chrono::steady_clock::now(); }
    void stop(const char* label);
    void create_alignments(bool is_sr, uint32_t batch_id, AlignmentStore* into = nullptr);   // into: the helper thread's own store (Hypo::polish)
};

}  // namespace hypo
