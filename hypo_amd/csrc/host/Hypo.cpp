// Hypo.cpp — orchestration of one polishing run (reference: src/Hypo.cpp).
#include "Hypo.hpp"
#include "DeviceArms.hpp"
#include <omp.h>
#include <sys/resource.h>
#include <thread>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include <atomic>
namespace hypo {
extern std::atomic<uint64_t> g_stage_counters[5];      // host/Contig.cpp
namespace {
std::string pending_tmp;                               // <output>.tmp while a run is writing it (Hypo::polish); removed by a run that fails
void remove_pending_output() { if (!pending_tmp.empty()) std::remove(pending_tmp.c_str()); }
}

Hypo::Hypo(const InputFlags& flags) : _cFlags(flags) {
    omp_set_num_threads((int)_cFlags.threads);
    _tstart = std::chrono::steady_clock::now();
}

// slog::Monitor::stop prints wall time and RSS per phase (external/slog/src/Monitor.cpp:31-65); same shape of line
void Hypo::stop(const char* label) {
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - _t0).count();
    struct rusage ru; getrusage(RUSAGE_SELF, &ru);
    std::fprintf(stdout, "RESOURCES (%s): TIME= %g sec; PEAK RSS (so far)= %ldMB.\n", label, s, ru.ru_maxrss / 1024);
    _times.phases.emplace_back(label, s);
}

void Hypo::polish() {
    std::ofstream stagefile(HYPO_STAGEFILE, std::ofstream::out | std::ofstream::app);
    if (_cFlags.intermed && !stagefile.is_open()) {
        std::fprintf(stderr, "[Hypo::Hypo] Error: File open error: Stage File (%s) exists but could not be opened!\n", HYPO_STAGEFILE);
        std::exit(1);
    }
    // ---- solid k-mers: loaded from aux/solid_kmers.bvsd (the KMC-based construction is outside the hot path) ------
    start();
    SolidKmers sk; sk.k = _cFlags.k;
    if (_cFlags.done_stage < 1) {
        std::fprintf(stderr, "[Hypo::SolidKmers] Error: this build does not run KMC/suk; provide %s and run with -i "
                             "(stage file %s with stage 1), as written by the reference or by tests/golden/gen_e2e.py\n", HYPO_SKFILE, HYPO_STAGEFILE);
        std::exit(1);
    }
    if (!sk.load(HYPO_SKFILE)) {
        std::fprintf(stderr, "[Hypo::SolidKmers] Error: File Loading: Could not load the DS for Solid kmers (%s)!\n", HYPO_SKFILE);
        std::exit(1);
    }
    stop("[Hypo:Hypo]: Loaded Solid kmers. ");
    std::fprintf(stdout, "[Hypo::Hypo] Info: Number of (canonical) solid kmers (nonhp) : %lu\n", (unsigned long)sk.num_solid);

    // ---- contigs ------------------------------------------------------------------------------------------------------
    start();
    {
        std::vector<FastaRecord> recs;
        if (!read_fastx(_cFlags.draft_filename, recs)) {
            std::fprintf(stderr, "[Hypo::Hypo] Error: File open error: Draft File (%s) could not be read!\n", _cFlags.draft_filename.c_str());
            std::exit(1);
        }
        // (the records are packed into Contig objects on all threads: 100 x 1 Mbp took 0.7 s one after the other)
        // (a handful of large contigs: one after the other, each packed by all threads — PackedSeq::assign)
        _contigs.resize(recs.size());
        const bool many = recs.size() >= (size_t)std::max(2u, _cFlags.threads / 2);
#pragma omp parallel for schedule(dynamic, 1) if (many)
        for (int64_t i = 0; i < (int64_t)recs.size(); ++i) {
            _contigs[(size_t)i].reset(new Contig((uint32_t)i, recs[(size_t)i].name, recs[(size_t)i].seq));
            std::string().swap(recs[(size_t)i].seq);
        }
        for (size_t i = 0; i < recs.size(); ++i) _cname_to_id[recs[i].name] = (uint32_t)i;
    }
    stop("[Hypo:Hypo]: Loaded Contigs. ");
    _alignment_store.resize(_contigs.size());

    _contig_batch_size = _cFlags.processing_batch_size == 0 ? (uint32_t)_contigs.size() : _cFlags.processing_batch_size;
    uint32_t num_batches = _contig_batch_size ? (uint32_t)_contigs.size() / _contig_batch_size : 0;
    if (_contig_batch_size && _contigs.size() % _contig_batch_size != 0) ++num_batches;
    _sf_short.reset(new SamReader(_cFlags.sr_bam_filename));
    if (!_sf_short->ok()) { std::fprintf(stderr, "[Hypo::Hypo] Error: File open error: %s\n", _cFlags.sr_bam_filename.c_str()); std::exit(1); }
    // (a run of ONE batch has nothing beside the parser but the scans: it keeps whole teams)
    const int side_team = num_batches > 1 ? std::max(1, (int)_cFlags.threads / 2) : std::max(1, (int)_cFlags.threads);
    const int inflate_threads = std::getenv("HYPO_INFLATE_THREADS") ? std::max(1, std::atoi(std::getenv("HYPO_INFLATE_THREADS"))) : side_team;
    _sf_short->set_inflate_threads(inflate_threads);
    // the short reads of the first batch are parsed while the contigs are scanned (the parser needs the contigs' names and lengths only)
    std::thread prefetch, long_release;
    ReadBatch staged;                                      // the next batch's short reads while the helper parses them
    const bool prefetch_on = !(std::getenv("HYPO_PREFETCH") && std::atoi(std::getenv("HYPO_PREFETCH")) == 0);
    // The parser's team and the team that inflates BGZF blocks for it are HALF of -t each: with all three teams (these two and the main
    // thread's phases) at -t the stages only got in each other's way — 500 Mbp at k = 17, -t 64 on the 128-core box: 4.4-4.8 s with
    // 64 / 64, 3.7-4.0 s with 32 / 32, 3.8 s with 16 / 16 or 24 / 24 (profiles/history/r04_thread_split.txt).
    int helper_threads = side_team;
    if (const char* e = std::getenv("HYPO_HELPER_THREADS")) helper_threads = std::max(1, std::atoi(e));        // (experiments)
    if (prefetch_on && num_batches > 0) {
        staged.reset(_contigs.size());
        prefetch = std::thread([this, &staged, helper_threads] { omp_set_num_threads(helper_threads); create_alignments_flat(0, staged); });
    }

    // ---- solid positions: device scan (the C-ABI call is serialised on the context's stream) ------
    // the 4^k-bit set goes to the device once (2 GiB at the default k = 17), not once per contig
    start();
    if (hypo_gpu_solid_set_upload(sk.words.data(), sk.get_k()) != HYPO_OK) { std::fprintf(stderr, "[Hypo::Hypo] Error: %s\n", hypo_gpu_last_error()); std::exit(1); }
    {   // A few contigs side by side: the device part of a scan (copy in, kernel, copy out) runs under the context's lock, one
        // contig at a time; what a thread does around it — fresh pages for 8 bytes per base, the copy into the contig's own
        // vectors — overlaps with the next contig's device part (250 x 1 Mbp at k = 15, nearly every position solid: 1.1 s one
        // after the other)
        const int nt = std::max(1, std::min((int)_cFlags.threads, 4));
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
        for (int64_t i = 0; i < (int64_t)_contigs.size(); ++i) {
            if (_contigs[(size_t)i]->find_solid_pos(sk, true) != HYPO_OK) { std::fprintf(stderr, "[Hypo::Contig] Error: %s\n", hypo_gpu_last_error()); std::exit(1); }
        }
    }
    stop("[Hypo:Hypo]: Found Solid pos in contigs. ");

    std::fprintf(stdout, "[Hypo::Hypo] Info: Number.of contigs: %lu; Number of batches: %u\n", (unsigned long)_contigs.size(), num_batches);
    if (!_cFlags.lr_bam_filename.empty()) {
        _sf_long.reset(new SamReader(_cFlags.lr_bam_filename));
        if (!_sf_long->ok()) { std::fprintf(stderr, "[Hypo::Hypo] Error: File open error: %s\n", _cFlags.lr_bam_filename.c_str()); std::exit(1); }
        _sf_long->set_inflate_threads(inflate_threads);
    }
    // which inflate path this run takes (libdeflate is bound by name at run time, zlib is the fall-back: the two differ by 1.5 x on a
    // 3 Gbp run, so a number quoted from this binary should say which one it was)
    if (_sf_short->bgzf() || (_sf_long && _sf_long->bgzf()))
        std::fprintf(stdout, "[Hypo::Hypo] Info: BGZF blocks are inflated by %s on %d threads\n", BlockInflater::name(), inflate_threads);
    std::ofstream dump;
    if (!_region_dump.empty()) dump.open(_region_dump);

    // one per device context (they outlive the batches: spent alignments are released behind the phases that follow)
    const int n_ctx = std::max(1, hypo_gpu_num_devices());
    std::vector<std::unique_ptr<DeviceArms>> device_arms;
    for (int d = 0; d < n_ctx; ++d) device_arms.emplace_back(new DeviceArms(d));
    _reads.reset(_contigs.size());
    // The polished contigs of a batch are filed by a writer thread while the next batch is processed (the reference writes
    // everything at the end, src/Hypo.cpp:256-268: the same bytes in the same order); what a written contig no longer needs is
    // released there.
    // The records go to <output>.tmp, which takes the output's name only when every contig is in it and the file closed without an error:
    // a run that fails half way (a device error, a bad record three batches in) leaves no truncated file under the name the caller asked
    // for, and an earlier result under that name stays what it was.  A failing run removes its .tmp on the way out.
    pending_tmp = _cFlags.output_filename + ".tmp";
    static bool cleanup_registered = false;
    if (!cleanup_registered) { cleanup_registered = true; std::atexit(remove_pending_output); }
    std::ofstream ofile(pending_tmp);
    if (!ofile.is_open()) {
        std::fprintf(stderr, "[Hypo::Hypo] Error: File open error: Output File (%s) could not be opened!\n", pending_tmp.c_str());
        std::exit(1);
    }
    std::thread writer;
    for (uint32_t batch_id = 0; batch_id < num_batches; ++batch_id) {
        std::fprintf(stdout, "********** [Hypo::Hypo] Info: BATCH-ID: %u\n", batch_id);
        const uint32_t initial_cid = batch_id * _contig_batch_size;
        const uint32_t final_cid = std::min<uint32_t>((uint32_t)_contigs.size(), initial_cid + _contig_batch_size);
        const bool over_contigs = (final_cid - initial_cid) >= _cFlags.threads;
        // The short-read records of the NEXT batch are parsed on a helper thread while this batch is with the device (support
        // votes, arms and POA leave the host's cores idle most of the time); it fills a store of its own,
        // the reader state belongs to create_alignments alone.  HYPO_PREFETCH=0: one batch after the other.
        start();
        // (records of contigs behind the previous batch that it had consumed — _reads holds them — come first, then what the helper
        // or this thread parses now)
        const size_t slices_before = _reads.n_slices();
        if (prefetch.joinable()) {
            prefetch.join();
            _reads.append(staged);
        } else {
            create_alignments_flat(batch_id, _reads);
        }
        // (read before the helper starts on the next batch: did this batch's load open with the record carried over, and of which contig)
        const int32_t opened_with_cid = _rs_short.opened_with_cid;
        stop("[Hypo:Hypo]: Loaded alignments. ");
        if (prefetch_on && batch_id + 1 < num_batches) {
            staged.reset(_contigs.size());
            // (the helper's team: all of -t while the machine has threads to spare, half of it otherwise — the main thread's own
            // parallel phases run next to it)
            prefetch = std::thread([this, batch_id, &staged, helper_threads] { omp_set_num_threads(helper_threads); create_alignments_flat(batch_id + 1, staged); });
        }
        // ... and THIS batch's long reads (-B) are read and parsed while its short-read phases run (the reference loads them behind the
        // short arms, src/Hypo.cpp:225-229; what they are does not depend on anything those phases compute): 2.4 of the 8 s of the
        // 250 Mbp set.  The record the long reader consumed for a later contig during the LAST batch is already in that contig's
        // store entry (below), so the order of the two streams is the reference's.
        std::thread long_prefetch;
        if (prefetch_on && !_cFlags.lr_bam_filename.empty()) {
            if (long_release.joinable()) long_release.join();
            _reads_long.reset(_contigs.size());
            long_prefetch = std::thread([this, batch_id, helper_threads] { omp_set_num_threads(helper_threads); create_alignments_flat(batch_id, _reads_long, false); });
        }
        std::vector<char> materialized(final_cid - initial_cid, 0);        // per contig of the batch: its Alignment objects exist (host loops)
        _mat_base = initial_cid;
        // The first long read of a contig may have been consumed while the previous batch's long reads were loaded; the reference
        // files it in this contig's store entry, where the short-read phases of THIS batch find it in front of the short reads and
        // treat it as one of them (src/Hypo.cpp:314-325, :126-199).  Same here: it moves to the front of the flat batch.
        // ... in front of them but for ONE: when the short-read loader of the batch before had already consumed this contig's first short
        // read, that record was filed first (store entry = [short carry, long carry, this batch's short reads ..]): the long read goes
        // behind the batch's opening record then.  Arm order inside a window is record order, and POA depends on it.
        // (the contig the batch opened with goes FIRST: `slices_before + 1` is an index into the slice list as it stands now, and every
        // other contig's records are put in front of everything afterwards, which shifts indices but not the order inside a contig)
        if (opened_with_cid >= (int32_t)initial_cid && opened_with_cid < (int32_t)final_cid && !_alignment_store[(size_t)opened_with_cid].empty()) {
            _reads.prepend((uint32_t)opened_with_cid, _alignment_store[(size_t)opened_with_cid], slices_before + 1);
            _alignment_store[(size_t)opened_with_cid].clear();
        }
        for (uint32_t c = initial_cid; c < final_cid; ++c)
            if (!_alignment_store[c].empty()) { _reads.prepend(c, _alignment_store[c], 0); _alignment_store[c].clear(); }

        // With several devices the contigs of the batch are dealt out to the contexts in contiguous ranges of about equal
        // numbers of alignments: every context keeps the reads of its contigs, counts their support votes, cuts their arms and
        // polishes its own resident windows; no window travels.  A batch with FEWER contigs than contexts (BASELINE config C4 is one
        // 250 Mbp contig) shares its contigs out instead (round 4): a contig's contexts each own a coordinate range of it and work
        // on the reads around that range (DeviceArms::set_piece) — the reference's loop is over the windows of ONE contig too
        // (src/Hypo.cpp:236-248).
        struct CtxWork { uint32_t c0 = 0, c1 = 0; bool piece = false; uint32_t own0 = 0, own1 = 0; };
        std::vector<CtxWork> work((size_t)n_ctx);
        const uint32_t n_batch_contigs = final_cid - initial_cid;
        work[0].c0 = initial_cid; work[0].c1 = final_cid;
        if ((uint32_t)n_ctx > 1 && !_cFlags.host_arms && n_batch_contigs >= (uint32_t)n_ctx) {
            std::vector<uint32_t> ctx_cut((size_t)n_ctx + 1, final_cid);
            ctx_cut[0] = initial_cid;
            uint64_t total = 0, acc = 0;
            for (uint32_t c = initial_cid; c < final_cid; ++c) total += _reads.count(c) + 1;
            int d = 1;
            for (uint32_t c = initial_cid; c < final_cid && d < n_ctx; ++c) {
                acc += _reads.count(c) + 1;
                // the cut behind contig c belongs to context d when the first d shares are full (every context gets >= 1 contig)
                while (d < n_ctx && acc * (uint64_t)n_ctx >= total * (uint64_t)d && final_cid - (c + 1) >= (uint32_t)(n_ctx - d)) ctx_cut[(size_t)d++] = c + 1;
            }
            for (; d < n_ctx; ++d) ctx_cut[(size_t)d] = std::max(ctx_cut[(size_t)d - 1] + 1, final_cid - (uint32_t)(n_ctx - d));
            for (int x = 0; x < n_ctx; ++x) { work[(size_t)x].c0 = ctx_cut[(size_t)x]; work[(size_t)x].c1 = ctx_cut[(size_t)x + 1]; }
        } else if ((uint32_t)n_ctx > 1 && !_cFlags.host_arms && !std::getenv("HYPO_NO_PIECES") && hypo_gpu_reads_upload(nullptr, nullptr, 0) != HYPO_E_UNSUPPORTED) {
            // (the probe: a library without the resident-read entry points — the CPU test shim — answers HYPO_E_UNSUPPORTED, and the
            // host loops, which know nothing of pieces, take the batch as before)
            // contexts per contig: one each, the rest one at a time to the contig with most alignments per context it has
            std::vector<uint32_t> share(n_batch_contigs, 1);
            for (uint32_t extra = (uint32_t)n_ctx - n_batch_contigs; extra > 0; --extra) {
                uint32_t best = 0; double best_load = -1;
                for (uint32_t i = 0; i < n_batch_contigs; ++i) {
                    const double load = (double)(_reads.count(initial_cid + i) + 1) / share[i];
                    if (load > best_load) { best_load = load; best = i; }
                }
                ++share[best];
            }
            int d = 0;
            for (uint32_t i = 0; i < n_batch_contigs; ++i) {
                const uint32_t c = initial_cid + i, len = (uint32_t)_contigs[c]->get_len();
                for (uint32_t j = 0; j < share[i]; ++j, ++d) {
                    CtxWork& w = work[(size_t)d];
                    w.c0 = c; w.c1 = c + 1; w.piece = share[i] > 1;
                    w.own0 = (uint32_t)((uint64_t)len * j / share[i]); w.own1 = j + 1 == share[i] ? len : (uint32_t)((uint64_t)len * (j + 1) / share[i]);
                }
            }
        }
        for (int d = 0; d < n_ctx; ++d) {
            const CtxWork& w = work[(size_t)d];
            if (w.piece) {
                // halo: the longest read of the contig + the longest window a read at the edge can still reach into
                const uint32_t halo = _reads.max_span(w.c0) + 2048;
                device_arms[(size_t)d]->set_piece(w.own0, w.own1, halo, (uint32_t)_contigs[w.c0]->get_len());
                std::fprintf(stdout, "[Hypo::Hypo] Info: context %d owns [%u, %u) of contig %s (halo %u)\n", d, w.own0, w.own1, _contigs[w.c0]->get_name().c_str(), halo);
            } else device_arms[(size_t)d]->clear_piece();
        }
        (void)hypo_gpu_last_error();
        auto piece_failed = [&](int d, const char* what) {
            std::fprintf(stderr, "[Hypo::Hypo] Error: %s failed on context %d, which shares contig %s with other contexts (%s); run on one device or with --host-arms\n",
                         what, d, _contigs[work[(size_t)d].c0]->get_name().c_str(), hypo_gpu_last_error());
            std::exit(1);
        };
        // N1: the reads go to the device once, now; the support votes are counted there (support_kernel.hip) and the arm kernels
        // use the same copy later.  --host-arms, an unsorted file or a device error: the reference's host loops, per contig range.
        start();
        // --require-device / HYPO_REQUIRE_DEVICE=1: a stage that was meant for the device and is about to run in the host loops ends
        // the run instead (the Info lines of DeviceArms say why it did not run there).  Not with --host-arms / HYPO_HOST_SUPPORT,
        // which ask for the host loops.
        const bool require_device = (_cFlags.require_device || (std::getenv("HYPO_REQUIRE_DEVICE") && std::atoi(std::getenv("HYPO_REQUIRE_DEVICE")) != 0)) &&
                                    !_cFlags.host_arms && !std::getenv("HYPO_HOST_SUPPORT");
        auto host_fallback = [&](const char* what) {
            if (!require_device) return;
            std::fflush(stdout);
            std::fprintf(stderr, "[Hypo::Hypo] Error: --require-device: %s would be computed on the host (last device message: %s)\n", what, hypo_gpu_last_error());
            std::exit(1);
        };
        std::vector<char> votes_dev((size_t)n_ctx, 0);         // per context: its reads are resident
        if (!_cFlags.host_arms && !std::getenv("HYPO_HOST_SUPPORT"))
            for (int d = 0; d < n_ctx; ++d) {
                const uint32_t c0 = work[(size_t)d].c0, c1 = work[(size_t)d].c1;
                if (c0 < c1) votes_dev[(size_t)d] = device_arms[(size_t)d]->upload_reads(_contigs, c0, c1, _reads) ? 1 : 0;
                if (c0 < c1 && work[(size_t)d].piece && !votes_dev[(size_t)d]) piece_failed(d, "the upload of the reads");
            }
        for (int d = 0; d < n_ctx; ++d) {
            const uint32_t c0 = work[(size_t)d].c0, c1 = work[(size_t)d].c1;
            if (c0 >= c1) continue;
            if (votes_dev[(size_t)d] && device_arms[(size_t)d]->support_kmers(_contigs, c0, c1, _cFlags.k)) continue;
            if (work[(size_t)d].piece) piece_failed(d, "the k-mer support votes");
            { uint64_t nr = 0; for (uint32_t c = c0; c < c1; ++c) nr += _reads.count(c); if (nr) host_fallback("the k-mer support votes"); }
            materialize_alignments(c0, c1, materialized);
            for (uint32_t cid = c0; cid < c1; ++cid) {
                _contigs[cid]->ensure_kids();
                auto& alns = _alignment_store[cid];
#pragma omp parallel for
                for (int64_t t = 0; t < (int64_t)alns.size(); ++t) alns[(size_t)t]->update_solidkmers_support(_cFlags.k, *_contigs[cid]);
            }
        }
        hypo_gpu_use_device(0);
        if (const char* vp = std::getenv("HYPO_DUMP_VOTES"))        // (tests: device votes against the host loops of the reference, counter by counter)
            if (std::FILE* vf = std::fopen(vp, batch_id == 0 ? "wb" : "ab")) { for (uint32_t c = initial_cid; c < final_cid; ++c) _contigs[c]->dump_votes(vf, 0); std::fclose(vf); }
        stop("[Hypo:Hypo]: Solid kmers support update. ");

        start();
        {   // many contigs: one contig per thread as in the reference; fewer contigs than threads: the contigs side by side, each
            // with its share of the threads for the minimizers of its mega-windows (a nested team; -p 10 on 64 threads took 45 ms per
            // batch one contig after the other)
            const int nc = (int)(final_cid - initial_cid), T = (int)_cFlags.threads;
            const int outer = std::max(1, std::min(nc, T)), inner = over_contigs ? 1 : std::max(1, T / outer);
            if (inner > 1) omp_set_max_active_levels(2);
#pragma omp parallel for schedule(static, 1) num_threads(outer)
            for (int64_t i = initial_cid; i < (int64_t)final_cid; ++i) {
                omp_set_num_threads(inner);
                _contigs[(size_t)i]->prepare_for_division(_cFlags.k);
            }
            omp_set_max_active_levels(1);
        }
        uint64_t num_sr = 0, len_sr = 0;
        for (uint32_t i = initial_cid; i < final_cid; ++i) { num_sr += _contigs[i]->get_num_sr(); len_sr += _contigs[i]->get_len_sr(); }
        std::fprintf(stdout, "[Hypo::Hypo] Info: Total number of SR: %lu; Total length of SR: %lu\n", (unsigned long)num_sr, (unsigned long)len_sr);
        stop("[Hypo:Hypo]: Finding SR (and preparing for division). ");

        start();
        for (int d = 0; d < n_ctx; ++d) {
            const uint32_t c0 = work[(size_t)d].c0, c1 = work[(size_t)d].c1;
            if (c0 >= c1) continue;
            if (votes_dev[(size_t)d] && device_arms[(size_t)d]->support_minimizers(_contigs, c0, c1)) continue;
            if (work[(size_t)d].piece) piece_failed(d, "the minimizer support votes");
            { uint64_t nr = 0; for (uint32_t c = c0; c < c1; ++c) nr += _reads.count(c); if (nr) host_fallback("the minimizer support votes"); }
            materialize_alignments(c0, c1, materialized);
            for (uint32_t cid = c0; cid < c1; ++cid) {
                auto& alns = _alignment_store[cid];
#pragma omp parallel for
                for (int64_t t = 0; t < (int64_t)alns.size(); ++t) alns[(size_t)t]->update_minimisers_support(*_contigs[cid]);
            }
        }
        hypo_gpu_use_device(0);
        if (const char* vp = std::getenv("HYPO_DUMP_VOTES"))
            if (std::FILE* vf = std::fopen(vp, "ab")) { for (uint32_t c = initial_cid; c < final_cid; ++c) _contigs[c]->dump_votes(vf, 1); std::fclose(vf); }
        stop("[Hypo:Hypo]: Minimisers support update. ");

        start();
        {   // (as above: fewer contigs than threads share the team, each contig builds its Window objects with its share)
            const int nc = (int)(final_cid - initial_cid), T = (int)_cFlags.threads;
            const int outer = std::max(1, std::min(nc, T)), inner = over_contigs ? 1 : std::max(1, T / outer);
            if (inner > 1) omp_set_max_active_levels(2);
#pragma omp parallel for schedule(static, 1) num_threads(outer)
            for (int64_t i = initial_cid; i < (int64_t)final_cid; ++i) {
                omp_set_num_threads(inner);
                _contigs[(size_t)i]->divide_into_regions();
            }
            omp_set_max_active_levels(1);
        }
        stop("[Hypo:Hypo]: Division into windows. ");

        start();
        // The device cuts the reads into arms, prunes the windows and keeps the window batch in its memory (DeviceArms.hpp);
        // --host-arms, several devices or an unsorted alignment file take the host loops of the reference instead.
        std::vector<char> on_dev(final_cid - initial_cid, 0);  // per contig of the batch: its short arms were cut on a device
        std::vector<char> long_dev(final_cid - initial_cid, 0);  // ... and its long arms (LONG windows resident on the device)
        if (!_cFlags.host_arms) {
            for (int d = 0; d < n_ctx; ++d) {
                const uint32_t c0 = work[(size_t)d].c0, c1 = work[(size_t)d].c1;
                if (c0 >= c1) continue;
                if (work[(size_t)d].piece) {
                    // the halo was chosen before the division: a window longer than it (a weak region force_divide could not cut) would lose
                    // the arms of the reads beyond it — the span grows to the longest window this context owns and its reads go over again
                    const uint32_t longest = device_arms[(size_t)d]->longest_owned_window(*_contigs[c0], false);
                    if (device_arms[(size_t)d]->widen_halo(longest + 64))
                        std::fprintf(stdout, "[Hypo::Hypo] Info: context %d owns a window of %u bases: halo widened to %u, its reads are uploaded again\n", d, longest, device_arms[(size_t)d]->halo());
                }
                if (device_arms[(size_t)d]->build(_contigs, c0, c1, _reads, _cFlags.k))
                    for (uint32_t c = c0; c < c1; ++c) on_dev[c - initial_cid] = 1;
                else if (work[(size_t)d].piece) piece_failed(d, "short-arm selection");
            }
            for (int d = 0; d < n_ctx; ++d) if (work[(size_t)d].piece) DeviceArms::finish_short(_contigs, work[(size_t)d].c0, work[(size_t)d].c1);
            hypo_gpu_use_device(0);
        }
        for (uint32_t cid = initial_cid; cid < final_cid; ++cid) {
            if (on_dev[cid - initial_cid]) { _alignment_store[cid].clear(); continue; }       // (objects a host vote loop had asked for)
            if (!_cFlags.host_arms && _reads.count(cid) > 0) host_fallback("short-arm selection");
            materialize_alignments(cid, cid + 1, materialized);
            auto& alns = _alignment_store[cid];
#pragma omp parallel for
            for (int64_t t = 0; t < (int64_t)alns.size(); ++t) alns[(size_t)t]->find_short_arms(_cFlags.k, *_contigs[cid]);
        }
        stop("[Hypo:Hypo]: Short arms computing. ");
        start();
        // few contigs: the parallelism is inside a contig (window ranges); many contigs: one contig per thread as in the reference
#pragma omp parallel for schedule(static, 1) if (over_contigs)
        for (int64_t i = initial_cid; i < (int64_t)final_cid; ++i) {
            if (on_dev[(size_t)i - initial_cid]) continue;
            _contigs[(size_t)i]->fill_short_windows(_alignment_store[(size_t)i]); _alignment_store[(size_t)i].clear();
        }
        {   // the batch's short reads are spent; what it consumed for contigs of later batches stays for them (src/Hypo.cpp:314-325)
            ReadBatch later;
            later.reset(_contigs.size());
            _reads.carry_beyond(final_cid, later);
            _reads.clear(&_block_pool, &_pool_mu);
            _reads.reset(_contigs.size());
            _reads.append(later);
        }
        stop("[Hypo:Hypo]: Short arms filling. ");

        if (!_cFlags.lr_bam_filename.empty()) {
            start();
            // the long reads of the batch, flat like the short ones (ReadBatch.hpp; round 4: 1.2 M objects of 8 kb each took 2.5 s to
            // build on the 250 Mbp set).  The reader stops behind the first kept record of a later contig; the reference files that
            // record in ITS contig's store entry (src/Hypo.cpp:314-325), where that batch's short-read phases find it: it becomes
            // an object there (see the top of the batch loop).
            if (long_prefetch.joinable()) long_prefetch.join();
            else { _reads_long.reset(_contigs.size()); create_alignments_flat(batch_id, _reads_long, false); }
            if (_rs_long.carry_blk) {
                ReadBatch one;
                one.reset(_contigs.size());
                one.add(_rs_long.carry_blk, _rs_long.carry_r0, _rs_long.carry_r1);
                if (_rs_long.carry_cid >= 0) one.materialize((uint32_t)_rs_long.carry_cid, _alignment_store[(size_t)_rs_long.carry_cid]);
                _rs_long.carry_blk.reset();
            }
            stop("[Hypo:Hypo]: Loaded alignments of Long reads. ");
            start();
            const auto tl0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(static, 1)
            for (int64_t i = initial_cid; i < (int64_t)final_cid; ++i) _contigs[(size_t)i]->prepare_long_windows();
            const auto tl1 = std::chrono::steady_clock::now();
            // the long reads are cut into arms, filtered (Filter::is_good) and kept as a second resident batch by the context that
            // holds the contig's short arms; --host-arms, an unsorted file or a device error take the reference's host loops
            std::vector<char> long_on_dev(final_cid - initial_cid, 0);
            if (!_cFlags.host_arms) {
                for (int d = 0; d < n_ctx; ++d) {
                    const uint32_t c0 = work[(size_t)d].c0, c1 = work[(size_t)d].c1;
                    if (c0 >= c1) continue;
                    if (work[(size_t)d].piece) {                     // (as for the short arms: the LONG pseudo-windows exist only now)
                        const uint32_t longest = device_arms[(size_t)d]->longest_owned_window(*_contigs[c0], true);
                        if (device_arms[(size_t)d]->widen_halo(longest + 64))
                            std::fprintf(stdout, "[Hypo::Hypo] Info: context %d owns a LONG window of %u bases: halo widened to %u\n", d, longest, device_arms[(size_t)d]->halo());
                    }
                    if (device_arms[(size_t)d]->build_long(_contigs, c0, c1, _reads_long))
                        for (uint32_t c = c0; c < c1; ++c) long_on_dev[c - initial_cid] |= 1;
                    else {
                        if (device_arms[(size_t)d]->long_failed()) host_fallback("long-arm selection");
                        if (work[(size_t)d].piece) long_on_dev[c0 - initial_cid] |= 2;      // (a shared contig: all of its contexts or none)
                    }
                }
                for (int d = 0; d < n_ctx; ++d) {
                    if (!work[(size_t)d].piece) continue;
                    const uint32_t c = work[(size_t)d].c0;
                    if (long_on_dev[c - initial_cid] & 2) device_arms[(size_t)d]->drop_long();
                }
                for (uint32_t c = initial_cid; c < final_cid; ++c) {
                    char& f = long_on_dev[c - initial_cid];
                    const bool shared = f != 0 && n_batch_contigs < (uint32_t)n_ctx;
                    if (f & 2) f = 0;
                    else if (f == 1 && shared) DeviceArms::finish_long(_contigs, c, c + 1);
                }
                hypo_gpu_use_device(0);
            }
            for (uint32_t cid = initial_cid; cid < final_cid; ++cid) {
                if (long_on_dev[cid - initial_cid]) continue;
                auto& alns = _alignment_store[cid];                     // (the host loops of the reference read objects)
                _reads_long.materialize(cid, alns);
#pragma omp parallel for
                for (int64_t t = 0; t < (int64_t)alns.size(); ++t) alns[(size_t)t]->find_long_arms(*_contigs[cid]);
            }
#pragma omp parallel for schedule(static, 1) if (over_contigs)
            for (int64_t i = initial_cid; i < (int64_t)final_cid; ++i) {
                if (long_on_dev[(size_t)i - initial_cid]) continue;
                _contigs[(size_t)i]->fill_long_windows(_alignment_store[(size_t)i]); _alignment_store[(size_t)i].clear();
            }
            for (uint32_t c = initial_cid; c < final_cid; ++c) long_dev[c - initial_cid] = long_on_dev[c - initial_cid];
            const auto tl2 = std::chrono::steady_clock::now();
            // (7 GB of parsed long reads on the 250 Mbp set: handed back behind the POA, not in front of it)
            if (long_release.joinable()) long_release.join();
            long_release = std::thread([this, spent = std::make_shared<ReadBatch>(std::move(_reads_long))] { spent->clear(&_block_pool, &_pool_mu); });
            _reads_long = ReadBatch();
            if (std::getenv("HYPO_HOST_TIMING")) {
                auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
                std::fprintf(stderr, "[timing] long arms: prepare_long_windows %.3f s, arms %.3f s, long reads released %.3f s\n", sec(tl0, tl1), sec(tl1, tl2), sec(tl2, std::chrono::steady_clock::now()));
            }
            stop("[Hypo:Hypo]: Long arms filling. ");
        } else {
            Contig::set_no_long_reads();
        }

        // ---- POA: every valid window of the contig batch in one device call (reference: per-window OpenMP loop) ----
        start();
        Window::prepare_for_poa(_cFlags.score_params, _cFlags.threads);
        std::vector<Window*> wins;
        for (uint32_t i = initial_cid; i < final_cid; ++i)
            for (uint64_t w = 0; w < _contigs[i]->get_num_regions(); ++w)
                if (_contigs[i]->is_valid_window((uint32_t)w)) {
                    const bool lw = _contigs[i]->window((uint32_t)w)->is_long();
                    if (!(lw ? long_dev[i - initial_cid] : on_dev[i - initial_cid])) wins.push_back(_contigs[i]->window((uint32_t)w));
                }
        uint64_t n_resident = 0;
        {   // the resident batches: every context polishes its own, side by side; what needs the host's retry path joins `wins`
            std::vector<std::vector<Window*>> retry((size_t)n_ctx);
            std::vector<int> prc((size_t)n_ctx, HYPO_OK);
            std::vector<std::string> perr((size_t)n_ctx);
            std::vector<std::thread> th;
            for (int d = 0; d < n_ctx; ++d) {
                DeviceArms& da = *device_arms[(size_t)d];
                da.reset_polished();
                if (!da.active() && !da.active_long()) continue;

                auto job = [&, d] {
                    DeviceArms& me = *device_arms[(size_t)d];
                    prc[(size_t)d] = me.polish(_cFlags.score_params, dump.is_open(), &retry[(size_t)d]);
                    if (prc[(size_t)d] == HYPO_OK) prc[(size_t)d] = me.polish_long(_cFlags.score_params, dump.is_open(), &retry[(size_t)d]);
                    if (prc[(size_t)d] != HYPO_OK) perr[(size_t)d] = hypo_gpu_last_error();
                };
                if (n_ctx == 1) job(); else th.emplace_back(job);
            }
            for (auto& t : th) t.join();
            for (int d = 0; d < n_ctx; ++d) n_resident += device_arms[(size_t)d]->polished_windows();      // (windows a context owns: its halo's are another's)
            hypo_gpu_use_device(0);
            for (int d = 0; d < n_ctx; ++d) {
                if (prc[(size_t)d] != HYPO_OK) { std::fprintf(stderr, "[Hypo::Window] Error: %s\n", perr[(size_t)d].c_str()); std::exit(1); }
                wins.insert(wins.end(), retry[(size_t)d].begin(), retry[(size_t)d].end());
            }
        }
        if (Window::generate_consensus_batch(wins) != HYPO_OK) { std::fprintf(stderr, "[Hypo::Window] Error: %s\n", hypo_gpu_last_error()); std::exit(1); }
        std::fprintf(stdout, "[Hypo::Hypo] Info: polished windows (Batch %u): %lu\n", batch_id, (unsigned long)(wins.size() + n_resident));
        stop("[Hypo:Hypo]: POA of windows. ");

        if (dump.is_open())
            for (uint32_t i = initial_cid; i < final_cid; ++i)
                for (uint64_t w = 0; w < _contigs[i]->get_num_regions(); ++w) {
                    uint32_t b, e; RegionType t;
                    _contigs[i]->region((uint32_t)w, b, e, t);
                    const Window* win = _contigs[i]->window((uint32_t)w);
                    if (!win && t != RegionType::SR && t != RegionType::MSR && !_cFlags.lr_bam_filename.empty()) continue;   // swallowed by a LONG window
                    if (win) e = b + (uint32_t)win->get_window_len();           // a LONG window spans the arm-less regions that follow it
                    dump << _contigs[i]->get_name() << '\t' << b << '\t' << e << '\t' << region_name(t);
                    if (win) dump << '\t' << win->dump_counts() << '\t' << win->arms_crc32() << '\t' << win->get_consensus();
                    if (win && std::getenv("HYPO_REGION_DUMP_ARMS")) dump << '\t' << (win->is_long() ? "L" : "S") << '\t' << win->dump_text();
                    else if (t != RegionType::SR && t != RegionType::MSR) dump << "\t0\t0\t0\t0\t0\t" << _contigs[i]->draft_segment(b, e);   // no arms: draft kept
                    dump << '\n';
                }
        if (writer.joinable()) writer.join();
        writer = std::thread([this, &ofile, initial_cid, final_cid] {
            omp_set_num_threads(std::max(1, std::min((int)_cFlags.threads, 8)));
            for (uint32_t c = initial_cid; c < final_cid; ++c) { ofile << *_contigs[c]; _contigs[c]->release_after_output(); }
        });
    }
    _alignment_store.clear();
    if (long_release.joinable()) long_release.join();

    start();
    if (writer.joinable()) writer.join();
    ofile.close();
    if (!ofile) { std::fprintf(stderr, "[Hypo::Hypo] Error: writing the output file (%s) failed!\n", pending_tmp.c_str()); std::exit(1); }
    if (std::rename(pending_tmp.c_str(), _cFlags.output_filename.c_str()) != 0) {
        std::fprintf(stderr, "[Hypo::Hypo] Error: could not move %s to %s!\n", pending_tmp.c_str(), _cFlags.output_filename.c_str());
        std::exit(1);
    }
    pending_tmp.clear();
    stop("[Hypo:Hypo]: Writing results. ");
    _times.overall = std::chrono::duration<double>(std::chrono::steady_clock::now() - _tstart).count();
    std::fprintf(stdout, "RESOURCES ([Hypo:Hypo]: Overall. ): TIME= %g sec.\n", _times.overall);
    if (std::getenv("HYPO_STAGE_COUNTERS"))
        std::fprintf(stdout, "[Hypo::Hypo] Info: stage counters: solid k-mers accepted with 40-80 %% support %llu, refused after another such k-mer %llu; force_divide calls %llu; "
                             "minimizers dropped as recurring %llu, as poly-base %llu\n", (unsigned long long)g_stage_counters[0].load(), (unsigned long long)g_stage_counters[1].load(),
                     (unsigned long long)g_stage_counters[2].load(), (unsigned long long)g_stage_counters[3].load(), (unsigned long long)g_stage_counters[4].load());
    // (1.5 M windows with their arms and consensus strings: freed contig by contig on all threads, 0.37 s of the C3 run's wall otherwise)
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < (int64_t)_contigs.size(); ++i) _contigs[(size_t)i].reset();
    _contigs.clear();
}

// ---- the flat path of the short reads (ReadBatch.hpp) -------------------------------------------------------------------------
void Hypo::materialize_alignments(uint32_t c0, uint32_t c1, std::vector<char>& done) {
    // `done` belongs to the batch in hand: entry i = contig (_mat_base + i), _mat_base = the batch's first contig
    for (uint32_t c = c0; c < c1; ++c) {
        char& d = done[c - _mat_base];
        if (d) continue;
        d = 1;
        _reads.materialize(c, _alignment_store[c]);
    }
}

void Hypo::parse_block(const SamReader& sf, const SamReader::RecordBlock& raw, ParsedBlock& blk, bool long_reads) {
    const uint32_t mq = _cFlags.map_qual_th;
    const size_t count = raw.n();
    blk.n = count;
    blk.status.assign(count, ParsedBlock::ST_SKIPPED);
    blk.cid.assign(count, -1);
    blk.bad_ref_name.clear();
    // (a block of long reads is a few thousand records of 10+ kb: shares small enough for every thread to get some)
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)omp_get_max_threads(), count / (long_reads ? 8 : 512) + 1));
    blk.chunks.resize((size_t)T);
#pragma omp parallel num_threads(T)
    {
        SamRecord rec;                                   // one per thread: its strings and CIGAR vector are reused
        int32_t tid_seen = -2; int64_t cid_seen = -1;
        for (int c = omp_get_thread_num(); c < T; c += omp_get_num_threads()) {
            ReadChunk& ch = blk.chunks[(size_t)c];
            ch.clear();
            const size_t a = count * (size_t)c / (size_t)T, b = count * ((size_t)c + 1) / (size_t)T;
            const bool bam = sf.is_bam();
            for (size_t i = a; i < b; ++i) {
                SamReader::BamCore bc;
                const bool direct = bam && sf.bam_core(raw.rec(i), raw.len(i), bc);       // BAM: fixed fields in place, bases straight from 4 to 2 bits
                if (direct) {
                    if ((bc.flag & (SAM_FUNMAP | SAM_FSECONDARY | SAM_FQCFAIL | SAM_FDUP)) || bc.mapq < mq) continue;
                    rec.tid = bc.tid; rec.pos = bc.pos; rec.flag = bc.flag; rec.mapq = bc.mapq;
                    rec.cigar.resize(bc.n_cigar);
                    if (bc.n_cigar) std::memcpy(rec.cigar.data(), bc.cigar, 4ull * bc.n_cigar);
                    rec.qname.assign(bc.qname, bc.l_qname);
                    if (long_reads) rec.has_nm = sf.bam_nm(raw.rec(i), raw.len(i), bc, rec.nm);
                } else {
                    sf.parse(raw.rec(i), raw.len(i), rec);
                    if ((rec.flag & (SAM_FUNMAP | SAM_FSECONDARY | SAM_FQCFAIL | SAM_FDUP)) || rec.mapq < mq) continue;
                }
                if (rec.tid != tid_seen) {                 // (one look-up per run of records of a contig)
                    auto it = rec.tid < 0 ? _cname_to_id.end() : _cname_to_id.find(sf.tid2name(rec.tid));
                    tid_seen = rec.tid;
                    cid_seen = it == _cname_to_id.end() ? -1 : (int64_t)it->second;
                }
                if (cid_seen < 0) { blk.status[i] = ParsedBlock::ST_BADREF; continue; }
                blk.cid[i] = (int32_t)cid_seen;
                uint32_t rb, re, qab, qae;
                Alignment::span_of(*_contigs[(size_t)cid_seen], rec, rb, re, qab, qae);
                const uint32_t qlen = qae - qab;
                // a long read is dropped when its NM-based normalised edit distance exceeds the threshold; the reference divides the
                // integers first (edit_dist * 100 / rlen) and compares the quotient (Alignment.cpp:51-58)
                if (long_reads && rec.has_nm && re > rb && (double)(rec.nm * 100 / (int64_t)(re - rb)) > (double)_cFlags.norm_edit_th) { blk.status[i] = ParsedBlock::ST_INVALID; continue; }
                // Alignment.cpp:551-571: the aligned part 2-bit packed; a read with a non-ACGT base there is dropped
                bool ok = (size_t)qab + qlen <= (direct ? (size_t)bc.l_seq : rec.seq.size());
                const size_t at = ch.seq.size();
                if (ok) {
                    ch.seq.resize(at + (qlen + 3) / 4);
                    ok = direct ? pack2_from_bam4(bc.seq4, qab, qlen, ch.seq.data() + at) : pack2_acgt(rec.seq.data() + qab, qlen, ch.seq.data() + at);
                    if (!ok) ch.seq.resize(at);
                }
                if (!ok) { blk.status[i] = ParsedBlock::ST_INVALID; continue; }
                blk.status[i] = ParsedBlock::ST_KEPT;
                ch.raw.push_back((uint32_t)i); ch.cid.push_back((int32_t)cid_seen); ch.rb.push_back(rb); ch.re.push_back(re); ch.qae.push_back(qlen);
                ch.cig.insert(ch.cig.end(), rec.cigar.begin(), rec.cigar.end());
                ch.seq_at.push_back((uint32_t)ch.seq.size()); ch.cig_at.push_back((uint32_t)ch.cig.size());
            }
        }
    }
    for (size_t i = 0; i < count; ++i)
        if (blk.status[i] == ParsedBlock::ST_BADREF) { blk.bad_ref_name = sf.record_name(raw.rec(i), raw.len(i)); break; }
}

// src/Hypo.cpp:278-329 for the short reads: stream the (coordinate-sorted) file, stop when a record of the next batch shows up.
// A reader thread inflates the file and cuts the next block of raw records while this block is parsed on all threads, every
// thread writing the records of its stretch into a chunk of flat arrays; the batch takes the kept records as slices of the block.
void Hypo::create_alignments_flat(uint32_t batch_id, ReadBatch& into, bool is_sr) {
    SamReader& sf = is_sr ? *_sf_short : *_sf_long;
    RecordStream& rs = is_sr ? _rs_short : _rs_long;
    const uint32_t final_cid = batch_id * _contig_batch_size + _contig_batch_size;
    uint64_t num_invalid = 0, num_alns = 0;
    constexpr size_t kBlock = 1 << 19, kBlockBytes = (size_t)128 << 20;     // (a block of records = about one run of inflated BGZF blocks, SeqIO.hpp)
    // (the record a call consumed for a contig behind its batch: a short read opens that contig's batch; a LONG read has been filed
    // as an object in that contig's store entry by Hypo::polish, where the reference's short-read phases find it)
    rs.opened_with_cid = -1;
    if (rs.carry_blk && is_sr) { into.add(rs.carry_blk, rs.carry_r0, rs.carry_r1); rs.opened_with_cid = rs.carry_cid; }
    rs.carry_blk.reset();
    bool stop = false, more_ahead = true;
    double t_wait = 0, t_par = 0, t_col = 0; auto now = []{ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    if (!rs.reader) { rs.reader.reset(new BlockReader()); rs.reader->start(&sf); }
    while (!stop) {
        bool asked = false;
        const double t0 = now();
        if (!rs.parsed || rs.ppos >= rs.parsed->n) {
            if (rs.have_ahead) { std::swap(rs.cur, rs.ahead); rs.have_ahead = false; }
            else if (rs.more) rs.more = sf.read_block(rs.cur, kBlock, kBlockBytes);
            else break;
            if (rs.cur.n() == 0) { if (!rs.more) break; continue; }
            if (rs.more && !rs.have_ahead) { rs.reader->request(&rs.ahead, kBlock, kBlockBytes); asked = true; }
            std::shared_ptr<ParsedBlock> blk;
            {
                std::lock_guard<std::mutex> lk(_pool_mu);
                if (rs.parsed && rs.parsed.use_count() == 1 && _block_pool.size() < 8) _block_pool.push_back(std::move(rs.parsed));
                rs.parsed.reset();
                if (!_block_pool.empty()) { blk = std::move(_block_pool.back()); _block_pool.pop_back(); }
            }
            if (!blk) blk = std::make_shared<ParsedBlock>();
            const double t1 = now(); t_wait += t1 - t0;
            parse_block(sf, rs.cur, *blk, !is_sr);
            t_par += now() - t1;
            rs.parsed = blk; rs.ppos = 0;
        }
        const double t2 = now();
        const ParsedBlock& B = *rs.parsed;
        // where the batch ends: the first record that is not skipped and belongs to a contig of a later batch is consumed as well,
        // as in the reference; a record with an unknown reference before that is fatal
        size_t s_first = B.n;
        for (size_t i = rs.ppos; i < B.n; ++i) {
            const uint8_t st = B.status[i];
            if (st == ParsedBlock::ST_SKIPPED) continue;
            if (st == ParsedBlock::ST_BADREF) {
                std::fprintf(stderr, "[Hypo::Hypo] Error: Alignment File error: Contig-reference of record %s does not exist in the draft!\n", B.bad_ref_name.c_str());
                // (this may be the helper thread, with the main thread inside a device call: leave without running the static
                // destructors under it)
                std::fflush(nullptr);
                remove_pending_output();
                std::_Exit(1);
            }
            if (st == ParsedBlock::ST_KEPT) ++num_alns; else ++num_invalid;
            if ((uint32_t)B.cid[i] >= final_cid) { s_first = i; break; }
        }
        into.add(rs.parsed, rs.ppos, s_first);
        if (s_first < B.n) {
            if (B.status[s_first] == ParsedBlock::ST_KEPT) { rs.carry_blk = rs.parsed; rs.carry_r0 = s_first; rs.carry_r1 = s_first + 1; rs.carry_cid = B.cid[s_first]; }
            rs.ppos = s_first + 1;
            stop = true;
        } else rs.ppos = B.n;
        const double t3 = now();
        t_col += t3 - t2;
        if (asked) { more_ahead = rs.reader->wait(); rs.more = more_ahead; rs.have_ahead = rs.ahead.n() > 0; }
        t_wait += now() - t3;
    }
    if (std::getenv("HYPO_HOST_TIMING")) std::fprintf(stderr, "[timing] create_alignments_flat: waiting for records %.3f s, parse %.3f s, into the batch %.3f s (BGZF blocks: %s)\n", t_wait, t_par, t_col, BlockInflater::name());
    std::fprintf(stdout, "[Hypo::Hypo] Info: Number of alignments (Batch %u): loaded (%lu) invalid (%lu)\n", batch_id,
                 (unsigned long)num_alns, (unsigned long)num_invalid);
}

}  // namespace hypo
