// ref_harness.cpp — thin C entry points around the REAL reference classes (hypo::Window,
// hypo::PackedSeq, spoa::Graph / AlignmentEngine), compiled from the sources where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libhyporef.so.  TEST INFRASTRUCTURE ONLY:
// used to validate oracle/hypo_oracle.c and to generate tests/golden/ (tests/golden/make_golden.py).
// Nothing here restates the algorithm; it only drives the reference's own public interface
// (include/Window.hpp:41-121, external/spoa/include/spoa/*.hpp).
#include <omp.h>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "Window.hpp"
#include "../include/hypo_gpu.h"

using namespace hypo;

static int g_engines = 0;

extern "C" {

// Appends one short + one long engine with these scores (Window::prepare_for_poa, Window.cpp:31-42)
// and returns its engine index.
int hyporef_new_engine(const int8_t sc[6]) {
    ScoreParams sp{sc[0], sc[1], sc[2], sc[3], sc[4], sc[5]};
    Window::prepare_for_poa(sp, 1);
    return g_engines++;
}

// Builds a real Window, adds the arms in the given order (internal, prefix, suffix, empties) and
// returns Window::get_consensus().  kept[i] = 1 if arm i was stored (LONG windows filter arms,
// Window.hpp:66-101).  Returns the consensus length or -1 if out_cap is too small.
int hyporef_window(int engine_idx, int is_long, const char* draft,
                   int ni, const char* const* in, int np, const char* const* pre,
                   int ns, const char* const* suf, int n_empty,
                   char* out, int out_cap, unsigned char* kept) {
    std::string d(draft);
    PackedSeq<4> pd(d);
    Window w(pd, 0, d.size(), is_long ? WindowType::LONG : WindowType::SHORT);
    int k = 0;
    for (int i = 0; i < ni; ++i) {
        auto before = w.get_num_internal();
        w.add_internal(PackedSeq<2>(std::string(in[i])));
        if (kept) kept[k] = (unsigned char)(w.get_num_internal() != before);
        ++k;
    }
    for (int i = 0; i < np; ++i) {
        auto before = w.get_num_pre();
        w.add_prefix(PackedSeq<2>(std::string(pre[i])));
        if (kept) kept[k] = (unsigned char)(w.get_num_pre() != before);
        ++k;
    }
    for (int i = 0; i < ns; ++i) {
        auto before = w.get_num_suf();
        w.add_suffix(PackedSeq<2>(std::string(suf[i])));
        if (kept) kept[k] = (unsigned char)(w.get_num_suf() != before);
        ++k;
    }
    for (int i = 0; i < n_empty; ++i) w.add_empty();
    w.generate_consensus((UINT32)engine_idx);
    std::string c = w.get_consensus();
    if ((int)c.size() > out_cap) return -1;
    std::memcpy(out, c.data(), c.size());
    return (int)c.size();
}

// Sequence-level replay on the reference's spoa: align + add_alignment for every sequence with the
// given mode (1 = kNW, 3 = kLOV, 4 = kROV); reports the last alignment, the final rank order and
// the heaviest-bundle consensus.
int hyporef_replay(int m, int n, int g, int n_seq, const char* const* seqs, const int* modes,
                   int32_t* pairs_out, int pairs_cap, int* n_pairs,
                   int32_t* rank_out, int rank_cap, int* n_nodes,
                   char* cons_out, int cons_cap, int* cons_len) {
    auto engine = spoa::createAlignmentEngine(spoa::AlignmentType::kNW, (int8_t)m, (int8_t)n, (int8_t)g);
    auto graph = spoa::createGraph();
    for (int i = 0; i < n_seq; ++i) {
        engine->changeAlignType(static_cast<spoa::AlignmentType>(modes[i]));
        std::string s(seqs[i]);
        auto aln = engine->align(s, graph);
        if (i == n_seq - 1 && n_pairs) {
            *n_pairs = (int)aln.size();
            for (int k = 0; k < (int)aln.size() && k < pairs_cap; ++k) {
                pairs_out[2 * k] = aln[k].first; pairs_out[2 * k + 1] = aln[k].second;
            }
        }
        graph->add_alignment(aln, s);
    }
    const auto& r = graph->rank_to_node_id();
    if (n_nodes) *n_nodes = (int)r.size();
    for (int k = 0; k < (int)r.size() && k < rank_cap; ++k) rank_out[k] = (int32_t)r[k];
    if (!r.empty()) {
        std::string c = graph->generate_consensus();
        if (cons_len) *cons_len = (int)c.size();
        for (int k = 0; k < (int)c.size() && k < cons_cap; ++k) cons_out[k] = c[k];
    } else if (cons_len) *cons_len = 0;
    return 0;
}

// PackedSeq round trip through the reference (src/PackedSeq.cpp): returns unpack() of a
// PackedSeq<NB> built from text; used to pin oracle_pack/unpack.
int hyporef_pack_roundtrip(int nb, const char* text, char* out, int out_cap) {
    std::string s(text), u;
    if (nb == 2) { PackedSeq<2> p(s); u = p.unpack(); } else { PackedSeq<4> p(s); u = p.unpack(); }
    if ((int)u.size() > out_cap) return -1;
    std::memcpy(out, u.data(), u.size());
    return (int)u.size();
}

// The reference's POA phase (src/Hypo.cpp:236-247) on a flattened batch.  Real hypo::Window objects are built from the packed
// batch first (not timed); then the reference's own loop shape runs: Window::prepare_for_poa(sp, threads) and
// `#pragma omp parallel for schedule(static,1)` over windows calling generate_consensus(omp_get_thread_num()).
// *seconds_out = wall time of that loop.  Used for parity at full batch size and as the CPU baseline of kind "reference".
int hyporef_batch(const int8_t sc[6], const HypoWindowBatch* in, HypoConsensusBatch* out, int n_threads, double* seconds_out) {
    static const char kB4[] = "ACGTN", kB2[] = "ACGT";
    auto text4 = [&](const uint8_t* p, uint32_t n) { std::string t(n, 'N'); for (uint32_t i = 0; i < n; ++i) { unsigned c = (p[i >> 1] >> (4 - 4 * (i & 1))) & 15; t[i] = kB4[c < 4 ? c : 4]; } return t; };
    auto text2 = [&](const uint8_t* p, uint32_t n) { std::string t(n, 'A'); for (uint32_t i = 0; i < n; ++i) t[i] = kB2[(p[i >> 2] >> (6 - 2 * (i & 3))) & 3]; return t; };
    if (n_threads < 1) n_threads = omp_get_max_threads();
    const uint32_t n = in->n_windows;
    std::vector<std::unique_ptr<Window>> ws(n);
    std::vector<uint8_t> filtered(n, 0);
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t w = 0; w < (int64_t)n; ++w) {
        const HypoWindow& W = in->windows[w];
        PackedSeq<4> pd(text4(in->draft4 + W.draft_off, W.draft_len));
        ws[w].reset(new Window(pd, 0, W.draft_len, W.type == HYPO_WIN_SHORT ? WindowType::SHORT : WindowType::LONG));
        uint32_t a = W.first_arm;
        for (uint32_t i = 0; i < W.n_internal; ++i, ++a) ws[w]->add_internal(PackedSeq<2>(text2(in->arms2 + in->arm_off[a], in->arm_len[a])));
        for (uint32_t i = 0; i < W.n_prefix; ++i, ++a) ws[w]->add_prefix(PackedSeq<2>(text2(in->arms2 + in->arm_off[a], in->arm_len[a])));
        for (uint32_t i = 0; i < W.n_suffix; ++i, ++a) ws[w]->add_suffix(PackedSeq<2>(text2(in->arms2 + in->arm_off[a], in->arm_len[a])));
        for (uint32_t i = 0; i < W.n_empty; ++i) ws[w]->add_empty();
        // A LONG window filters its arms as they are added (Filter::is_good in Window::add_*, include/Window.hpp:66-101).  The C-ABI batch
        // holds the arms that PASSED (the arm kernels / the host apply the filter when they cut the arms); a window handed over here
        // with an arm its own filter drops is not the window the batch describes: reported as HYPOREF_ST_FILTERED, not compared.
        if (ws[w]->get_num_total() != W.n_internal + W.n_prefix + W.n_suffix + W.n_empty) filtered[w] = 1;
    }
    ScoreParams sp{sc[0], sc[1], sc[2], sc[3], sc[4], sc[5]};
    const int base = g_engines;
    Window::prepare_for_poa(sp, (UINT32)n_threads);
    g_engines += n_threads;
    std::vector<uint8_t> undefined(n, 0);
    const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(static, 1) num_threads(n_threads)
    for (int64_t w = 0; w < (int64_t)n; ++w) {
        // (a window the reference itself cannot answer — its consensus is shorter than the two markers that Window.hpp:144 cuts off, the
        // iterator range is negative and libstdc++ throws — is reported as HYPO_ST_UNDEFINED; nothing else is caught)
        try { ws[w]->generate_consensus((UINT32)(base + omp_get_thread_num())); }
        catch (const std::length_error&) { undefined[w] = 1; }
    }
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (uint32_t w = 0; w < n; ++w) {
        if (undefined[w]) { out->len[w] = 0; out->status[w] = HYPO_ST_UNDEFINED; continue; }
        if (filtered[w]) { out->len[w] = 0; out->status[w] = 0xF0; continue; }      // HYPOREF_ST_FILTERED
        const std::string c = ws[w]->get_consensus();
        const uint64_t cap = out->off[w + 1] - out->off[w];
        out->len[w] = (uint32_t)c.size();
        out->status[w] = c.size() > cap ? HYPO_ST_CONS_OVERFLOW : HYPO_ST_OK;
        if (c.size() <= cap) std::memcpy(out->bases + out->off[w], c.data(), c.size());
    }
    return 0;
}

}  // extern "C"
