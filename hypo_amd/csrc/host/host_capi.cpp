// host_capi.cpp — C entry points over the C++ host mirror so that tests can drive it like the reference's
// own classes (same call sequence as oracle/ref_harness.cpp uses on the real reference).
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "SeqIO.hpp"
#include "Contig.hpp"
#include "Window.hpp"

using namespace hypo;

extern "C" {

int hypo_host_set_scores(const int8_t sc[6]) {
    ScoreParams sp{sc[0], sc[1], sc[2], sc[3], sc[4], sc[5]};
    Window::prepare_for_poa(sp, 1);
    return 0;
}

// n windows given as flat string arrays; arms of window w are [arm_first[w], arm_first[w+1]) with counts per kind.
// out: consensus strings concatenated, lens per window; kept flags per arm (LONG windows filter arms).
int hypo_host_windows(int n, const int* is_long, const char* const* drafts, const int* ni, const int* np, const int* ns,
                      const int* n_empty, const char* const* arms, char* out, long out_cap, int* out_len,
                      unsigned char* kept, int batched) {
    std::vector<std::unique_ptr<Window>> ws;
    size_t a = 0;
    for (int w = 0; w < n; ++w) {
        std::string d(drafts[w]);
        PackedSeq<4> pd(d);
        ws.emplace_back(new Window(pd, 0, d.size(), is_long[w] ? WindowType::LONG : WindowType::SHORT));
        Window& W = *ws.back();
        for (int i = 0; i < ni[w]; ++i, ++a) { auto b = W.get_num_internal(); W.add_internal(PackedSeq<2>(std::string(arms[a]))); if (kept) kept[a] = W.get_num_internal() != b; }
        for (int i = 0; i < np[w]; ++i, ++a) { auto b = W.get_num_pre(); W.add_prefix(PackedSeq<2>(std::string(arms[a]))); if (kept) kept[a] = W.get_num_pre() != b; }
        for (int i = 0; i < ns[w]; ++i, ++a) { auto b = W.get_num_suf(); W.add_suffix(PackedSeq<2>(std::string(arms[a]))); if (kept) kept[a] = W.get_num_suf() != b; }
        for (int i = 0; i < n_empty[w]; ++i) W.add_empty();
    }
    if (batched) {
        std::vector<Window*> ptrs;
        for (auto& w : ws) ptrs.push_back(w.get());
        const int rc = Window::generate_consensus_batch(ptrs);
        if (rc != HYPO_OK) return rc;
    } else {
        for (auto& w : ws) w->generate_consensus(0);
    }
    long o = 0;
    for (int w = 0; w < n; ++w) {
        const std::string c = ws[w]->get_consensus();
        if (o + (long)c.size() > out_cap) return -100;
        std::memcpy(out + o, c.data(), c.size());
        out_len[w] = (int)c.size();
        o += (long)c.size();
    }
    return 0;
}

// Filter::initialise + is_good without a device (CPU test of the host logic)
int hypo_host_filter(const char* draft, int n_arms, const char* const* arms, unsigned char* good) {
    Filter f;
    f.initialise(std::string(draft));
    for (int i = 0; i < n_arms; ++i) good[i] = f.is_good(std::string(arms[i])) ? 1 : 0;
    return 0;
}

int hypo_host_pack_roundtrip(int nb, const char* text, char* out, int cap) {
    std::string u;
    if (nb == 2) u = PackedSeq<2>(std::string(text)).unpack(); else u = PackedSeq<4>(std::string(text)).unpack();
    if ((int)u.size() > cap) return -1;
    std::memcpy(out, u.data(), u.size());
    return (int)u.size();
}

// Contig::find_solid_pos + rank/select spot checks
int hypo_host_contig_scan(const char* seq, unsigned k, const uint64_t* words, uint64_t n_words,
                          uint64_t* n_solid, uint64_t* kids_out, uint64_t kids_cap,
                          const uint64_t* rank_q, uint64_t* rank_a, int n_rank,
                          const uint64_t* sel_q, uint64_t* sel_a, int n_sel) {
    SolidKmers sk; sk.k = k; sk.words.assign(words, words + n_words);
    Contig c(0, "ctg test", std::string(seq));
    const int rc = c.find_solid_pos(sk);
    if (rc != HYPO_OK) return rc;
    *n_solid = c.get_num_solid();
    for (uint64_t i = 0; i < c.get_num_solid() && i < kids_cap; ++i) kids_out[i] = c.kid_at(i);
    for (int i = 0; i < n_rank; ++i) rank_a[i] = c.rank(rank_q[i]);
    for (int i = 0; i < n_sel; ++i) sel_a[i] = c.select(sel_q[i]);
    return 0;
}

// read_fastx through the mapped reader (line_by_line = 0) or the line-by-line one: "name\tsequence\n" per record into out; the number of
// bytes, -1 when the file is not FASTA / FASTQ, -2 when out is too small
int hypo_host_read_fastx(const char* path, int line_by_line, char* out, int cap) {
    std::vector<hypo::FastaRecord> recs;
    if (!hypo::read_fastx(path, recs, line_by_line == 0)) return -1;
    size_t at = 0;
    for (const auto& r : recs) {
        const size_t need = r.name.size() + 1 + r.seq.size() + 1;
        if (at + need > (size_t)cap) return -2;
        std::memcpy(out + at, r.name.data(), r.name.size()); at += r.name.size(); out[at++] = '\t';
        std::memcpy(out + at, r.seq.data(), r.seq.size()); at += r.seq.size(); out[at++] = '\n';
    }
    return (int)at;
}

}  // extern "C"
