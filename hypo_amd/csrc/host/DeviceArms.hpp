// DeviceArms.hpp — host side of arm selection on the device (include/hypo_gpu.h: hypo_gpu_arms_*).
// Flattens the regions of a contig batch and its short-read alignments into the arrays the C-ABI takes, lets the device cut
// the reads into arms and prune the windows (what Alignment::find_short_arms + Contig::fill_short_windows do on the host,
// src/Alignment.cpp:222-259, src/Contig.cpp:249-289), and later hands the consensus of the resident batch back to the Window
// objects.  The windows never see their arms unless somebody asks (region dump, a window that needs the host's retry path).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>
#include "Contig.hpp"
#include "ReadBatch.hpp"

namespace hypo {

class DeviceArms {
public:
    // slot: the device context (index into the device list of hypo_gpu_init) this object's batches live on; with several
    // devices the contigs of a batch are dealt out to one DeviceArms per context (Hypo::polish)
    explicit DeviceArms(int slot = 0) : _slot(slot) {}
    // Piece mode (round 4): this context owns [own0, own1) of ONE contig (a batch with fewer contigs than device contexts: BASELINE
    // config C4 is a single 250 Mbp contig).  It is given the contig's whole tables — the coordinates stay the contig's — but only
    // the reads that overlap [own0 - halo, own1 + halo); what it counts or builds is adopted where it owns the position: a solid
    // k-mer / a minimizer by where it lies, a window by where its region starts.  Every read that can vote for an owned k-mer or
    // give an arm to an owned window lies inside the halo (halo >= longest read span + longest window), and a read's walk over the
    // k-mers / minimizers of its own span sees all of them, so the owned results are what a single context computes.
    void set_piece(uint32_t own0, uint32_t own1, uint32_t halo, uint32_t contig_len) {
        _piece = true; _own0 = own0; _own1 = own1; _halo = halo; _piece_len = contig_len;
        _span[0] = own0 > halo ? own0 - halo : 0; _span[1] = (uint64_t)own1 + halo < contig_len ? own1 + halo : contig_len;
    }
    // The halo is chosen before the contig is divided (the votes that decide the division need the reads first); a read gives an arm to
    // an owned window [ws, we) iff rb < we, so every such read lies inside the span iff the halo is at least the longest owned window.
    // longest_owned_window: that length, over the SHORT windows (long_windows = false: the regions of divide_into_regions that are not
    // strong) or the LONG pseudo-windows of prepare_long_windows — a weak region inside a homopolymer run that force_divide cannot cut
    // (src/Contig.cpp:641-666) has no bound.  widen_halo: true when the halo had to grow (the span is recomputed and the resident short
    // reads are dropped, so that build() uploads the reads of the wider span; build_long() flattens the long reads by the span anyway).
    uint32_t longest_owned_window(const Contig& ctg, bool long_windows) const;
    bool widen_halo(uint32_t need) {
        if (!_piece || need <= _halo) return false;
        set_piece(_own0, _own1, need, _piece_len);
        _reads_resident = false;
        return true;
    }
    uint32_t halo() const { return _halo; }
    void clear_piece() { _piece = false; }
    bool piece() const { return _piece; }
    bool owns(uint64_t pos) const { return !_piece || (pos >= _own0 && pos < _own1); }
    uint64_t polished_windows() const { return _n_pol[0] + _n_pol[1]; }       // owned windows the last polish() + polish_long() answered
    DeviceArms(const DeviceArms&) = delete;
    DeviceArms& operator=(const DeviceArms&) = delete;
    ~DeviceArms() { if (_votes) (void)hypo_gpu_host_free(_votes); }
    // true: the windows of contigs [c0, c1) are pruned, their arms lie on the device and `store` is consumed; false: nothing
    // was changed and the host path must run (several devices, an unsorted alignment file, a batch beyond 32-bit coordinates)
    bool build(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, const ReadBatch& reads, unsigned k);
    bool active() const { return _active; }
    // N1: the short reads of contigs [c0, c1) go to the device once, right after they were loaded; the support votes
    // (Alignment::update_solidkmers_support / update_minimisers_support, src/Alignment.cpp:65-220) are counted there and come back
    // into the contigs' counters, and build() later cuts the same resident copy into arms.  false: nothing changed, host loops.
    bool upload_reads(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, const ReadBatch& reads);
    bool support_kmers(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, unsigned k);
    bool support_minimizers(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1);
    // consensus of every SHORT window of the resident batch; `keep_arms`: also copy the arms into the Window objects.  Windows
    // whose status asks for the host's retry / degraded path (a consensus longer than its slot, a window beyond the size classes)
    // get their arms and are appended to `retry` (the caller runs Window::generate_consensus_batch on them: one call for all
    // contexts); retry == nullptr: done here.  May be called from a thread of its own per context.
    int polish(const ScoreParams& sp, bool keep_arms, std::vector<Window*>* retry = nullptr);
    uint64_t num_windows() const { return _sum.n_windows; }
    // The same for the long reads of the contigs [c0, c1) after Contig::prepare_long_windows: the device cuts them at the pseudo
    // regions' borders (Alignment::find_long_arms), filters the arms (Filter::is_good against each LONG window's draft) and keeps
    // the LONG windows as a second resident batch.  true: `store` is consumed and the pseudo-region tables of the contigs are
    // released (what Contig::fill_long_windows does at its end); false: nothing changed, the host loops must run.
    bool build_long(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1, const ReadBatch& long_reads);
    bool active_long() const { return _active_long; }
    bool long_failed() const { return _long_failed; }            // the last build_long() reached the device and failed there (--require-device)
    // what build() / build_long() leave behind in the contigs once ALL contexts that work on them are done (piece mode defers it)
    static void finish_short(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1);
    static void finish_long(std::vector<std::unique_ptr<Contig>>& contigs, uint32_t c0, uint32_t c1);
    void reset_polished() { _n_pol[0] = _n_pol[1] = 0; }
    void drop() { _active = false; _active_long = false; }      // forget the resident batches (another context of the same contig failed)
    void drop_long() { _active_long = false; }
    int polish_long(const ScoreParams& sp, bool keep_arms, std::vector<Window*>* retry = nullptr);
    uint64_t num_long_windows() const { return _sum_long.n_windows; }

private:
    int _slot = 0;
    bool _piece = false; uint32_t _own0 = 0, _own1 = 0, _span[2] = {0, 0}, _halo = 0, _piece_len = 0;
    uint64_t _n_pol[2] = {0, 0};
    bool _reads_resident = false; uint32_t _reads_c0 = 0, _reads_c1 = 0;
    bool _active = false;
    HypoArmsSummary _sum{};
    ReadStaging _stage_long;                 // ... and of its long reads
    ReadStaging _stage;                      // page-locked staging arrays of this context's reads (grow-only, reused by every batch)
    uint32_t* _votes = nullptr; uint64_t _votes_cap = 0;       // page-locked: coverage [_votes_cap] then support [_votes_cap] of the k-mer votes
    PinnedBuf _pb[12];                       // staging of the minimizer tables and the region tables
    PinnedBuf _pbr[5], _pbl[5];              // POA results of the SHORT / LONG resident batch (polish and polish_long may run side by side)
    std::vector<Window*> _reg_window;        // region of the coordinate space -> its window (nullptr: SR, filler, pruned)
    void adopt_arms(const std::vector<uint32_t>& which, const std::vector<HypoWindow>& hw, const std::vector<uint32_t>& win_region, bool lng);
    int polish_impl(bool lng, const ScoreParams& sp, bool keep_arms, std::vector<Window*>* retry);
    bool _active_long = false;
    bool _long_failed = false;
    HypoArmsSummary _sum_long{};
    std::vector<Window*> _preg_window;       // pseudo region of the coordinate space -> its LONG window (nullptr: pseudo SR, filler)
};

}  // namespace hypo
