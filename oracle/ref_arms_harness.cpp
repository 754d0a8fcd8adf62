// ref_arms_harness.cpp — C entry point around the REAL short-read stage of the reference between "alignments loaded" and
// "windows filled" (src/Hypo.cpp:126-199): hypo::Contig::find_solid_pos, Alignment::update_solidkmers_support,
// Contig::prepare_for_division, Alignment::update_minimisers_support, Contig::divide_into_regions,
// Alignment::find_short_arms, Contig::fill_short_windows — and the reference's own per-region dump
// (Contig::generate_inspect_file, src/Contig.cpp:368-453, + the Window printer src/Window.cpp:63-84) as the output.
// Compiled from the sources where they lie under /root/reference by oracle/Makefile into oracle/_ref/libhyporef_arms.so.
// TEST INFRASTRUCTURE ONLY: it pins segmentation (A14), the support votes (N1) and short-arm selection (A13 / N2) of this
// repo's host and device code on inputs generated at test time (tests/test_oracle_vs_ref.py), next to the committed dumps of the
// CMake-built reference binary.  Nothing here restates any of those functions.
//
// How it links without the reference's build system and without htslib (whose sources want a configure-generated config.h:
// unbuildable here, see tests/golden/build_reference_binary.sh): src/Alignment.cpp, Contig.cpp, Window.cpp, PackedSeq.cpp,
// suk's SolidKmers.cpp and sdsl's plain lib/*.cpp are compiled as they are with one section per function and hidden
// visibility; src/main.cpp is compiled too because it DEFINES the settings objects the other files read (Sr_settings,
// Minimizer_settings, Window_settings, Arms_settings, src/main.cpp:85-88) — its main() is renamed on the command line
// (-Dmain=...) and, like everything else hyporef_arms() does not reach (the long-read constructor with its bam_aux_get, the
// BAM / FASTA readers, KMC, spoa), dropped by --gc-sections; -z defs proves that nothing undefined is left.  The only htslib
// pieces used are the struct bam1_t and the bam_get_* accessor macros of htslib/sam.h (header only); the record is laid out
// here the way sam.h documents it (qname, cigar, 4-bit sequence, qualities).  Short reads only: the long-read constructor
// needs htslib code.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>
#include <unistd.h>
#include <sys/stat.h>
#include <omp.h>
#include "Contig.hpp"
#include "Alignment.hpp"
#include "Window.hpp"

namespace {
uint8_t nt16(char c) {
    switch (c) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 4;
        case 'T': case 't': return 8;
        default: return 15;
    }
}
}  // namespace

namespace {
// the records of one file as Alignment objects through the SHORT-read constructor (src/Alignment.cpp:39-46; the long-read one,
// :48-63, differs by the NM-based filter only and needs htslib's bam_aux_get: the caller passes long reads that would all pass it)
void make_alignments(hypo::Contig& c, uint32_t n_reads, const uint32_t* pos, const uint32_t* cigar_off, const uint32_t* cigar,
                     const uint64_t* seq_off, const char* seq, std::vector<std::unique_ptr<hypo::Alignment>>& als, uint64_t& invalid) {
    als.reserve(n_reads);
    std::vector<uint8_t> data;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint32_t nc = cigar_off[r + 1] - cigar_off[r];
        const uint32_t lq = (uint32_t)(seq_off[r + 1] - seq_off[r]);
        const char* s = seq + seq_off[r];
        bam1_t b;
        std::memset(&b, 0, sizeof b);
        b.core.pos = pos[r];
        b.core.tid = 0;
        b.core.qual = 60;
        b.core.l_qname = 4;     // "r\0\0\0": sam.h pads the name to a multiple of four
        b.core.l_extranul = 2;
        b.core.n_cigar = nc;
        b.core.l_qseq = (int32_t)lq;
        b.core.mtid = -1;
        b.core.mpos = -1;
        data.assign(4 + 4ull * nc + (lq + 1) / 2 + lq, 0);
        data[0] = 'r';
        std::memcpy(data.data() + 4, cigar + cigar_off[r], 4ull * nc);
        uint8_t* q = data.data() + 4 + 4ull * nc;
        for (uint32_t i = 0; i < lq; ++i) q[i >> 1] |= (uint8_t)(nt16(s[i]) << ((~i & 1) << 2));
        std::memset(q + (lq + 1) / 2, 30, lq);
        b.data = data.data();
        b.l_data = (int)data.size();
        b.m_data = (uint32_t)data.size();
        als.emplace_back(std::make_unique<hypo::Alignment>(c, &b));       // src/Hypo.cpp:309
        if (!als.back()->is_valid) { als.pop_back(); ++invalid; }         // src/Hypo.cpp:314-318
    }
}
}  // namespace

extern "C" __attribute__((visibility("default")))
// The same stage followed by the LONG-read stage of a `-B` run (src/Hypo.cpp:203-229): Contig::prepare_long_windows,
// Alignment::find_long_arms (src/Alignment.cpp:262-299), Contig::fill_long_windows (include/Contig.hpp:91-113) with the real
// Window::add_* and their Filter (include/Window.hpp:66-103, include/Filter.hpp).  Long reads as the short ones (l*): the records
// the reference would keep (flag / mapping-quality filter AND its NM filter, which the caller guarantees nothing fails).
long hyporef_arms_long(const char* contig, uint64_t n, uint32_t k, const char* bvsd_path, uint32_t n_reads, const uint32_t* pos,
                       const uint32_t* cigar_off, const uint32_t* cigar, const uint64_t* seq_off, const char* seq,
                       uint32_t ln_reads, const uint32_t* lpos, const uint32_t* lcigar_off, const uint32_t* lcigar, const uint64_t* lseq_off, const char* lseq,
                       const char* work_dir, uint64_t* n_invalid) {
    auto sk = std::make_unique<suk::SolidKmers>(k);
    if (!sk->load(std::string(bvsd_path))) return -1;
    hypo::Contig c(0, "c", std::string(contig, (size_t)n));
    c.find_solid_pos(sk);
    uint64_t invalid = 0;
    {
        std::vector<std::unique_ptr<hypo::Alignment>> als;
        make_alignments(c, n_reads, pos, cigar_off, cigar, seq_off, seq, als, invalid);
        #pragma omp parallel for
        for (uint64_t t = 0; t < als.size(); ++t) als[t]->update_solidkmers_support(k, c);
        c.prepare_for_division(k);
        #pragma omp parallel for
        for (uint64_t t = 0; t < als.size(); ++t) als[t]->update_minimisers_support(c);
        c.divide_into_regions();
        #pragma omp parallel for
        for (uint64_t t = 0; t < als.size(); ++t) als[t]->find_short_arms(k, c);
        c.fill_short_windows(als);
    }
    {
        std::vector<std::unique_ptr<hypo::Alignment>> lals;
        make_alignments(c, ln_reads, lpos, lcigar_off, lcigar, lseq_off, lseq, lals, invalid);
        c.prepare_long_windows();                                          // src/Hypo.cpp:212
        #pragma omp parallel for
        for (uint64_t t = 0; t < lals.size(); ++t) lals[t]->find_long_arms(c);   // :217
        c.fill_long_windows(lals);                                         // :222
    }
    if (n_invalid) *n_invalid = invalid;
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd) || chdir(work_dir) != 0) return -2;
    mkdir("aux", 0777);
    long rc = (long)c.get_num_regions();
    {
        std::ofstream bed(BEDFILE);
        if (!bed.is_open()) rc = -2;
        else c.generate_inspect_file(bed);
    }
    if (chdir(cwd) != 0) return -2;
    return rc;
}

extern "C" __attribute__((visibility("default")))
// contig: n ASCII bases.  bvsd_path: the solid k-mer bit vector (as hyporef_solid_scan).  Reads: the primary mapped records
// of one contig in file order — pos (0-based leftmost reference position), BAM-encoded CIGAR operations cigar[cigar_off[r] ..
// cigar_off[r+1]) and the ASCII read seq[seq_off[r] .. seq_off[r+1]) (SEQ as in the file, soft clips included).  work_dir:
// an existing directory; the dump is written to <work_dir>/aux/inspect_c.txt (INSPECTFILEPREF is relative) and
// <work_dir>/aux/regions.bed.  Returns the number of regions, -1 when the bit set cannot be loaded, -2 on a file error.
// Not re-entrant (chdir).
long hyporef_arms(const char* contig, uint64_t n, uint32_t k, const char* bvsd_path, uint32_t n_reads, const uint32_t* pos,
                  const uint32_t* cigar_off, const uint32_t* cigar, const uint64_t* seq_off, const char* seq,
                  const char* work_dir, uint64_t* n_invalid) {
    auto sk = std::make_unique<suk::SolidKmers>(k);
    if (!sk->load(std::string(bvsd_path))) return -1;
    hypo::Contig c(0, "c", std::string(contig, (size_t)n));
    c.find_solid_pos(sk);

    std::vector<std::unique_ptr<hypo::Alignment>> als;
    als.reserve(n_reads);
    std::vector<uint8_t> data;
    uint64_t invalid = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint32_t nc = cigar_off[r + 1] - cigar_off[r];
        const uint32_t lq = (uint32_t)(seq_off[r + 1] - seq_off[r]);
        const char* s = seq + seq_off[r];
        bam1_t b;
        std::memset(&b, 0, sizeof b);
        b.core.pos = pos[r];
        b.core.tid = 0;
        b.core.qual = 60;
        b.core.l_qname = 4;     // "r\0\0\0": sam.h pads the name to a multiple of four
        b.core.l_extranul = 2;
        b.core.n_cigar = nc;
        b.core.l_qseq = (int32_t)lq;
        b.core.mtid = -1;
        b.core.mpos = -1;
        data.assign(4 + 4ull * nc + (lq + 1) / 2 + lq, 0);
        data[0] = 'r';
        std::memcpy(data.data() + 4, cigar + cigar_off[r], 4ull * nc);
        uint8_t* q = data.data() + 4 + 4ull * nc;
        for (uint32_t i = 0; i < lq; ++i) q[i >> 1] |= (uint8_t)(nt16(s[i]) << ((~i & 1) << 2));
        std::memset(q + (lq + 1) / 2, 30, lq);
        b.data = data.data();
        b.l_data = (int)data.size();
        b.m_data = (uint32_t)data.size();
        als.emplace_back(std::make_unique<hypo::Alignment>(c, &b));       // src/Hypo.cpp:309
        if (!als.back()->is_valid) { als.pop_back(); ++invalid; }         // src/Hypo.cpp:314-318
    }
    if (n_invalid) *n_invalid = invalid;

    // src/Hypo.cpp:135-199, one contig, in that order
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->update_solidkmers_support(k, c);
    c.prepare_for_division(k);
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->update_minimisers_support(c);
    c.divide_into_regions();
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->find_short_arms(k, c);
    c.fill_short_windows(als);
    hypo::Contig::set_no_long_reads();

    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd) || chdir(work_dir) != 0) return -2;
    mkdir("aux", 0777);
    long rc = (long)c.get_num_regions();
    {
        std::ofstream bed(BEDFILE);
        if (!bed.is_open()) rc = -2;
        else c.generate_inspect_file(bed);
    }
    if (chdir(cwd) != 0) return -2;
    return rc;
}

extern "C" __attribute__((visibility("default")))
// Row A15 in place: the whole short-read polish of ONE contig by the reference's own code — the stage above, then
// Window::prepare_for_poa + Contig::generate_consensus on every valid window (src/Hypo.cpp:237-247: the real Window class and
// spoa), then the REAL `operator<<(std::ostream&, const Contig&)` (src/Contig.cpp:345-366) into out_path: `>name`, the
// concatenation of strong regions, window consensus and untouched draft, one line.  What this repo's `hypo` writes for the same
// records must be these bytes (tests/test_oracle_vs_ref.py, tests/test_gpu_e2e.py).  scores: -m -x -g -M -X -G as the CLI
// stores them (src/main.cpp:100-113).  Returns the number of regions, -1 / -2 as hyporef_arms.
long hyporef_fasta(const char* contig, uint64_t n, const char* name, uint32_t k, const char* bvsd_path, uint32_t n_reads, const uint32_t* pos,
                   const uint32_t* cigar_off, const uint32_t* cigar, const uint64_t* seq_off, const char* seq, const char* out_path, const int8_t* scores) {
    auto sk = std::make_unique<suk::SolidKmers>(k);
    if (!sk->load(std::string(bvsd_path))) return -1;
    hypo::Contig c(0, std::string(name), std::string(contig, (size_t)n));
    c.find_solid_pos(sk);
    uint64_t invalid = 0;
    std::vector<std::unique_ptr<hypo::Alignment>> als;
    make_alignments(c, n_reads, pos, cigar_off, cigar, seq_off, seq, als, invalid);
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->update_solidkmers_support(k, c);
    c.prepare_for_division(k);
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->update_minimisers_support(c);
    c.divide_into_regions();
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->find_short_arms(k, c);
    c.fill_short_windows(als);
    hypo::Contig::set_no_long_reads();                                     // src/Hypo.cpp:230-232 (a run without -B)
    hypo::ScoreParams sp{scores[0], scores[1], scores[2], scores[3], scores[4], scores[5]};      // -m -x -g -M -X -G
    const int threads = omp_get_max_threads();
    hypo::Window::prepare_for_poa(sp, (hypo::UINT32)threads);              // src/Hypo.cpp:237
    const uint64_t num_reg = c.get_num_regions();
    #pragma omp parallel for schedule(static, 1)
    for (uint64_t w = 0; w < num_reg; ++w)
        if (c.is_valid_window(w)) c.generate_consensus(w, omp_get_thread_num());   // :238-247
    std::ofstream ofile(out_path);
    if (!ofile.is_open()) return -2;
    ofile << c;                                                            // :261-263
    return (long)num_reg;
}
