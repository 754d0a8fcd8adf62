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

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: whole FILES through the reference's own code — rows T1 (the 3 Gbp north-star run pinned contig by contig) and N3 (record
// field extraction pinned in place).  An independent minimal decoder (BGZF blocks inflated with plain zlib, BAM records cut out as the
// SAM specification section 4.2 lays them down; SAM text split on tabs) hands every record to the reference as the bam1_t that htslib's
// bam_read1 / sam_parse1 would have built: the core fields copied one by one, the variable part (name, CIGAR, 4-bit bases, qualities,
// tags) VERBATIM from the file behind the NUL-padded name (htslib/sam.h:206-215).  From there on everything is the reference's:
// the flag / mapping-quality filter of src/Hypo.cpp:299-301 (restated in three lines below because create_alignments itself needs
// htslib's reader), Alignment::Alignment(Contig&, bam1_t*) with its own bam_get_* field extraction (src/Alignment.cpp:28-38,513-571),
// the short-read stage, Window::generate_consensus, operator<<(Contig).  Nothing of this repo's readers (hypo_amd/csrc/host/SeqIO.hpp) is
// used, so "this repo's FASTA record == these bytes" pins its BAM / SAM parsing together with everything behind it.
// Short reads only (the long-read constructor calls bam_aux_get, which is htslib code; see the header of this file).
#include <chrono>
#include <map>
#include <sstream>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct RawRec {                        // the fixed part of a BAM record (SAM spec 4.2) + where its variable part lies in ContigRecs::bytes
    int32_t tid, pos; uint8_t l_read_name, mapq; uint16_t bin, flag; uint32_t n_cigar; int32_t l_seq, mtid, mpos, tlen;
    size_t off, len;                   // read_name .. end of record
};
struct ContigRecs { std::vector<RawRec> recs; std::vector<uint8_t> bytes; void clear() { recs.clear(); bytes.clear(); } };

std::string g_dump_dir;                // hyporef_set_dump_dir: where polish_contig leaves the reference's own per-region dump (empty: nowhere)

struct FileTotals { double align = 0, stage = 0, poa = 0, write = 0; uint64_t kept = 0, invalid = 0, regions = 0, windows = 0, contigs = 0, bases = 0; };

// One contig through the reference (src/Hypo.cpp:126-268 for a batch of one contig and no -B file)
std::unique_ptr<hypo::Alignment> make_alignment(hypo::Contig& c, const ContigRecs& R, const RawRec& r, std::vector<uint8_t>& data) {
    bam1_t b;
    std::memset(&b, 0, sizeof b);
    const uint32_t extranul = (4 - (r.l_read_name & 3)) & 3;           // htslib pads the name so that the CIGAR is 32-bit aligned
    b.core.tid = r.tid; b.core.pos = r.pos; b.core.bin = r.bin; b.core.qual = r.mapq; b.core.l_extranul = (uint8_t)extranul;
    b.core.flag = r.flag; b.core.l_qname = (uint16_t)(r.l_read_name + extranul); b.core.n_cigar = r.n_cigar; b.core.l_qseq = r.l_seq;
    b.core.mtid = r.mtid; b.core.mpos = r.mpos; b.core.isize = r.tlen;
    // (+ 16 bytes per CIGAR operation of slack: Alignment::copy_data copies n_cigar * 4 UINT32s from the CIGAR's address, src/Alignment.cpp:565-567)
    data.assign(r.len + extranul + 16ull * r.n_cigar + 16, 0);
    std::memcpy(data.data(), R.bytes.data() + r.off, r.l_read_name);
    std::memcpy(data.data() + r.l_read_name + extranul, R.bytes.data() + r.off + r.l_read_name, r.len - r.l_read_name);
    b.data = data.data();
    b.l_data = (int)(r.len + extranul);
    b.m_data = (uint32_t)data.size();
    return std::make_unique<hypo::Alignment>(c, &b);
}

// L: the long reads of a `-B` file (nullptr: a run without one).  They go through the SHORT-read constructor — the long-read one
// (src/Alignment.cpp:40-63) differs by its NM filter only and calls htslib's bam_aux_get, which this build does not have; the file
// reader below refuses a file with a record that filter would drop (-6), so the two constructors build the same objects.
void polish_contig(const std::unique_ptr<suk::SolidKmers>& sk, uint32_t k, uint32_t cid, const std::string& name, const std::string& seq,
                   const ContigRecs& R, std::ostream& out, FileTotals& T, const ContigRecs* L = nullptr, bool long_run = false) {
    double t0 = now_s();
    hypo::Contig c(cid, name, seq);
    c.find_solid_pos(sk);                                                  // src/Hypo.cpp:100-103
    double t1 = now_s();
    T.stage += t1 - t0;
    std::vector<std::unique_ptr<hypo::Alignment>> als;
    als.reserve(R.recs.size());
    std::vector<uint8_t> data;
    for (const RawRec& r : R.recs) {
        als.emplace_back(make_alignment(c, R, r, data));                   // src/Hypo.cpp:309
        if (!als.back()->is_valid) { als.pop_back(); ++T.invalid; }        // :314-318
    }
    T.kept += als.size();
    double t2 = now_s();
    T.align += t2 - t1;
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->update_solidkmers_support(k, c);      // :135-142
    c.prepare_for_division(k);
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->update_minimisers_support(c);
    c.divide_into_regions();
    #pragma omp parallel for
    for (uint64_t t = 0; t < als.size(); ++t) als[t]->find_short_arms(k, c);
    c.fill_short_windows(als);
    als.clear();                                                           // :196-199 (the alignments of the batch are released before the POA)
    if (long_run) {                                                        // :203-229
        std::vector<std::unique_ptr<hypo::Alignment>> lals;
        if (L) for (const RawRec& r : L->recs) {
            lals.emplace_back(make_alignment(c, *L, r, data));
            if (!lals.back()->is_valid) { lals.pop_back(); ++T.invalid; }
        }
        T.kept += lals.size();
        c.prepare_long_windows();                                          // :212
        #pragma omp parallel for
        for (uint64_t t = 0; t < lals.size(); ++t) lals[t]->find_long_arms(c);   // :217
        c.fill_long_windows(lals);                                         // :222
    }
    double t3 = now_s();
    T.stage += t3 - t2;
    const uint64_t num_reg = c.get_num_regions();
    uint64_t nwin = 0;
    #pragma omp parallel for schedule(static, 1) reduction(+ : nwin)
    for (uint64_t w = 0; w < num_reg; ++w)
        if (c.is_valid_window(w)) { c.generate_consensus(w, omp_get_thread_num()); ++nwin; }   // :238-247
    double t4 = now_s();
    T.poa += t4 - t3;
    out << c;                                                              // :261-263
    T.write += now_s() - t4;
    if (!g_dump_dir.empty()) {
        // the reference's own debug dump of every region — type, arm counts, draft, CONSENSUS, arms (Contig::generate_inspect_file,
        // src/Contig.cpp:368-453, which the reference calls from src/Hypo.cpp:265 when that line is un-commented): <dir>/aux/inspect_<name>.txt
        char cwd[4096];
        if (getcwd(cwd, sizeof cwd) && chdir(g_dump_dir.c_str()) == 0) {
            mkdir("aux", 0777);
            { std::ofstream bed(BEDFILE, std::ios::app); if (bed.is_open()) c.generate_inspect_file(bed); }
            if (chdir(cwd) != 0) std::abort();
        }
    }
    T.regions += num_reg; T.windows += nwin; T.contigs += 1; T.bases += seq.size();
}

// minimal FASTA reader: the name is the header line up to the first blank (kseq's rule), the sequence every following line joined
bool read_fasta(const char* path, std::vector<std::string>& names, std::vector<std::string>& seqs, const std::vector<char>* want) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::string line;
    long idx = -1; bool keep = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            ++idx;
            size_t e = 1; while (e < line.size() && line[e] != ' ' && line[e] != '\t') ++e;
            names.emplace_back(line.substr(1, e - 1)); seqs.emplace_back();
            keep = !want || ((size_t)idx < want->size() && (*want)[(size_t)idx]);
        } else if (idx >= 0 && keep) seqs.back() += line;
    }
    return true;
}

// BGZF: gzip members with a 'BC' extra subfield that holds the member's size (SAM spec 4.1).  The stream is inflated in runs of blocks,
// side by side (plain zlib, raw deflate, CRC-32 and ISIZE checked); take() hands out the next n bytes of the inflated stream.
struct BgzfStream {
    const uint8_t* map = nullptr; size_t size = 0, at = 0;
    std::vector<uint8_t> buf; size_t head = 0, len = 0;      // buf[head, len) = inflated bytes not handed out yet (buf only grows: no zero-fill per run)
    bool bad = false;
    bool open(const char* path) {
        int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0) { ::close(fd); return false; }
        size = (size_t)st.st_size;
        void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return false;
        map = (const uint8_t*)p;
        return true;
    }
    ~BgzfStream() { if (map) munmap((void*)map, size); }
    bool fill() {                                   // appends the next run of blocks behind the unread tail; false at the end of the file
        if (at >= size || bad) return false;
        struct Blk { size_t c0, clen, out, isize; };
        std::vector<Blk> blks;
        size_t total = 0;
        while (at < size && blks.size() < 4096) {
            if (at + 18 > size || map[at] != 0x1f || map[at + 1] != 0x8b || map[at + 2] != 8 || !(map[at + 3] & 4)) { bad = true; return false; }
            const size_t xlen = map[at + 10] | (size_t)map[at + 11] << 8;
            size_t bsize = 0, x = at + 12;
            while (x + 4 <= at + 12 + xlen) {
                const size_t slen = map[x + 2] | (size_t)map[x + 3] << 8;
                if (map[x] == 'B' && map[x + 1] == 'C' && slen == 2) bsize = (map[x + 4] | (size_t)map[x + 5] << 8) + 1;
                x += 4 + slen;
            }
            if (!bsize || at + bsize > size || bsize < 12 + xlen + 8) { bad = true; return false; }
            const size_t isize = map[at + bsize - 4] | (size_t)map[at + bsize - 3] << 8 | (size_t)map[at + bsize - 2] << 16 | (size_t)map[at + bsize - 1] << 24;
            blks.push_back({at + 12 + xlen, bsize - 12 - xlen - 8, total, isize});
            total += isize; at += bsize;
        }
        if (head) { std::memmove(buf.data(), buf.data() + head, len - head); len -= head; head = 0; }
        const size_t base = len;
        if (buf.size() < base + total) buf.resize(base + total);
        len = base + total;
        int failed = 0;
        #pragma omp parallel for schedule(dynamic, 16) reduction(+ : failed)
        for (size_t i = 0; i < blks.size(); ++i) {
            const Blk& b = blks[i];
            if (!b.isize) continue;
            z_stream zs; std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { ++failed; continue; }
            zs.next_in = (Bytef*)(map + b.c0); zs.avail_in = (uInt)b.clen;
            zs.next_out = buf.data() + base + b.out; zs.avail_out = (uInt)b.isize;
            const int rc = inflate(&zs, Z_FINISH);
            const bool ok = rc == Z_STREAM_END && zs.total_out == b.isize;
            inflateEnd(&zs);
            const uint8_t* t = map + b.c0 + b.clen;
            const uint32_t crc = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
            if (!ok || (uint32_t)crc32(crc32(0, nullptr, 0), buf.data() + base + b.out, (uInt)b.isize) != crc) ++failed;
        }
        if (failed) { bad = true; return false; }
        return true;
    }
    const uint8_t* take(size_t n) {
        while (len - head < n) if (!fill()) return nullptr;
        const uint8_t* p = buf.data() + head;
        head += n;
        return p;
    }
};
int32_t le32(const uint8_t* p) { int32_t v; std::memcpy(&v, p, 4); return v; }
uint16_t le16(const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }

struct FileJob {
    std::unique_ptr<suk::SolidKmers> sk;
    uint32_t k = 0, min_mapq = 2;
    std::vector<std::string> names, seqs;
    std::vector<char> want;                         // by contig index of the draft file; empty = all
    std::map<std::string, uint32_t> cname_to_id;   // src/Hypo.cpp:88 (_cname_to_id)
    std::ofstream out;
    FileTotals T;
    ContigRecs cur; long cur_cid = -1;
    std::vector<char> done;
    bool long_run = false;                          // a `-B` file was given
    std::map<uint32_t, ContigRecs> longs;           // its records by contig (read whole before the short reads stream through)
    uint32_t ned_th = 20;                           // -n
    bool picked(uint32_t cid) const { return want.empty() || (cid < want.size() && want[cid]); }
    void flush() {
        if (cur_cid >= 0) {
            auto it = longs.find((uint32_t)cur_cid);
            polish_contig(sk, k, (uint32_t)cur_cid, names[(size_t)cur_cid], seqs[(size_t)cur_cid], cur, out, T, it == longs.end() ? nullptr : &it->second, long_run);
            done[(size_t)cur_cid] = 1;
        }
        cur.clear(); cur_cid = -1;
    }
    // A record of contig `cid` in file order.  The records of a contig are taken to be contiguous in the file (coordinate-sorted or
    // grouped by reference, as every set of this repo is); a contig that comes back after another one was begun is an error (-4).
    int record(uint32_t cid, const RawRec& r, const uint8_t* var) {
        if (r.flag & (0x4 | 0x100 | 0x200 | 0x400)) return 0;             // BAM_FUNMAP | BAM_FSECONDARY | BAM_FQCFAIL | BAM_FDUP, src/Hypo.cpp:299
        if (r.mapq < min_mapq) return 0;                                   // :301
        if (!picked(cid)) return 0;
        if ((long)cid != cur_cid) { if (done[cid]) return -4; flush(); cur_cid = (long)cid; }
        RawRec q = r; q.off = cur.bytes.size();
        cur.bytes.insert(cur.bytes.end(), var, var + r.len);
        cur.recs.push_back(q);
        return 0;
    }
    // picked contigs without a single kept record are polished too (the reference writes every contig of the draft), in draft order at the end
    void finish() {
        flush();
        for (size_t c = 0; c < names.size(); ++c) if (picked((uint32_t)c) && !done[c]) { cur_cid = (long)c; flush(); }
    }
};

bool job_begin(FileJob& J, const char* draft_path, uint32_t k, const char* bvsd_path, uint32_t n_pick, const uint32_t* pick, uint32_t min_mapq,
               const char* out_path, const int8_t* scores) {
    J.k = k; J.min_mapq = min_mapq;
    J.sk = std::make_unique<suk::SolidKmers>(k);
    if (!J.sk->load(std::string(bvsd_path))) return false;
    if (n_pick) { uint32_t mx = 0; for (uint32_t i = 0; i < n_pick; ++i) mx = pick[i] > mx ? pick[i] : mx; J.want.assign((size_t)mx + 1, 0); for (uint32_t i = 0; i < n_pick; ++i) J.want[pick[i]] = 1; }
    if (!read_fasta(draft_path, J.names, J.seqs, n_pick ? &J.want : nullptr)) return false;
    for (size_t c = 0; c < J.names.size(); ++c) J.cname_to_id[J.names[c]] = (uint32_t)c;
    J.done.assign(J.names.size(), 0);
    J.out.open(out_path);
    if (!J.out.is_open()) return false;
    if (!J.long_run) hypo::Contig::set_no_long_reads();                    // src/Hypo.cpp:230-232 (a run without -B; sticky for the process: a `-B` job needs a private copy of the library)
    static bool engines = false;                                           // Window::prepare_for_poa appends engines at every call (src/Window.cpp:31-42): once per process
    if (!engines) {
        hypo::ScoreParams sp{scores[0], scores[1], scores[2], scores[3], scores[4], scores[5]};
        hypo::Window::prepare_for_poa(sp, (hypo::UINT32)omp_get_max_threads());   // :237
        engines = true;
    }
    return true;
}
void job_report(const FileJob& J, double decode_s, double* seconds, uint64_t* counts) {
    if (seconds) { seconds[0] = decode_s; seconds[1] = J.T.align; seconds[2] = J.T.stage; seconds[3] = J.T.poa; seconds[4] = J.T.write; }
    if (counts) { counts[0] = J.T.contigs; counts[1] = J.T.bases; counts[2] = J.T.kept; counts[3] = J.T.invalid; counts[4] = J.T.regions; counts[5] = J.T.windows; }
}
}  // namespace

extern "C" __attribute__((visibility("default")))
// The reference's short-read polish of the contigs `pick[0..n_pick)` (indices into the draft FASTA; n_pick = 0: every contig) with the
// records of a BAM file, written to out_path in draft order of first appearance in the BAM (contigs without records last).  One process
// may call this with ONE score set only (the reference keeps its engines in statics).  seconds[5]: this decoder (not the reference's
// work), Alignment objects, stage (find_solid_pos, votes, division, arms, windows), POA loop, operator<<; counts[6]: contigs, draft bases,
// alignments kept, invalid, regions, valid windows.  Returns the number of contigs written; -1 solid set / draft / output cannot be
// opened, -3 malformed BAM, -4 a contig's records are not contiguous, -5 a reference name that the draft does not have (src/Hypo.cpp:303-306).
long hyporef_fasta_bam(const char* draft_path, const char* bam_path, uint32_t k, const char* bvsd_path, uint32_t n_pick, const uint32_t* pick,
                       uint32_t min_mapq, const char* out_path, const int8_t* scores, double* seconds, uint64_t* counts);
namespace {
// reference span and NM tag of a BAM record's variable part; the long-read filter of src/Alignment.cpp:51-58 would drop the record when
// ceil(NM * 100 / span) > -n (integer division first, as written there)
bool nm_filter_drops(const RawRec& r, const uint8_t* var, uint32_t ned_th) {
    const uint8_t* cg = var + r.l_read_name;
    uint64_t span = 0;
    for (uint32_t i = 0; i < r.n_cigar; ++i) { uint32_t v; std::memcpy(&v, cg + 4ull * i, 4); const uint32_t op = v & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4; }
    const uint8_t* a = cg + 4ull * r.n_cigar + ((size_t)r.l_seq + 1) / 2 + (size_t)r.l_seq;
    const uint8_t* const e = var + r.len;
    while (a + 3 <= e) {
        const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
        a += 3;
        int64_t val = 0; size_t sz = 0; bool is_int = true;
        switch (ty) {
            case 'c': val = (int8_t)a[0]; sz = 1; break;
            case 'C': val = a[0]; sz = 1; break;
            case 's': { int16_t v; std::memcpy(&v, a, 2); val = v; sz = 2; break; }
            case 'S': { uint16_t v; std::memcpy(&v, a, 2); val = v; sz = 2; break; }
            case 'i': { int32_t v; std::memcpy(&v, a, 4); val = v; sz = 4; break; }
            case 'I': { uint32_t v; std::memcpy(&v, a, 4); val = v; sz = 4; break; }
            case 'A': sz = 1; is_int = false; break;
            case 'f': sz = 4; is_int = false; break;
            case 'Z': case 'H': { const uint8_t* z = a; while (z < e && *z) ++z; sz = (size_t)(z - a) + 1; is_int = false; break; }
            case 'B': { const char st = (char)a[0]; uint32_t cnt; std::memcpy(&cnt, a + 1, 4); const size_t es = (st == 'c' || st == 'C') ? 1 : ((st == 's' || st == 'S') ? 2 : 4); sz = 5 + es * cnt; is_int = false; break; }
            default: return false;
        }
        if (t0 == 'N' && t1 == 'M' && is_int) return span && (uint64_t)(val * 100 / (int64_t)span) > ned_th;
        a += sz;
    }
    return false;
}

// every record of a BAM file that passes the flag / mapping-quality filter and belongs to a picked contig -> fn(cid, record, variable part)
template <class Fn> long read_bam(FileJob& J, const char* bam_path, double& decode, Fn fn) {
    BgzfStream S;
    if (!S.open(bam_path)) return -1;
    double t0 = now_s();
    const uint32_t min_mapq = J.min_mapq;
    const uint8_t* p = S.take(8);
    if (!p || std::memcmp(p, "BAM\1", 4) != 0) return -3;
    const int32_t l_text = le32(p + 4);
    if (l_text < 0 || !S.take((size_t)l_text)) return -3;
    p = S.take(4); if (!p) return -3;
    const int32_t n_ref = le32(p);
    std::vector<long> tid_to_cid((size_t)(n_ref > 0 ? n_ref : 0), -1);
    for (int32_t i = 0; i < n_ref; ++i) {
        p = S.take(4); if (!p) return -3;
        const int32_t l_name = le32(p);
        p = S.take((size_t)l_name + 4); if (!p || l_name < 1) return -3;
        auto it = J.cname_to_id.find(std::string((const char*)p, (size_t)l_name - 1));
        if (it != J.cname_to_id.end()) tid_to_cid[(size_t)i] = it->second;
    }
    for (;;) {
        p = S.take(4);
        if (!p) break;
        const int32_t bs = le32(p);
        if (bs < 32) return -3;
        p = S.take((size_t)bs);
        if (!p) return -3;
        RawRec r;
        r.tid = le32(p); r.pos = le32(p + 4); r.l_read_name = p[8]; r.mapq = p[9]; r.bin = le16(p + 10); r.n_cigar = le16(p + 12); r.flag = le16(p + 14);
        r.l_seq = le32(p + 16); r.mtid = le32(p + 20); r.mpos = le32(p + 24); r.tlen = le32(p + 28);
        r.off = 0; r.len = (size_t)bs - 32;
        if ((size_t)r.l_read_name + 4ull * r.n_cigar + ((size_t)r.l_seq + 1) / 2 + (size_t)r.l_seq > r.len) return -3;
        if (r.flag & (0x4 | 0x100 | 0x200 | 0x400)) continue;              // (before the name lookup, as src/Hypo.cpp:299-306 orders them)
        if (r.mapq < min_mapq) continue;
        if (r.tid < 0 || r.tid >= n_ref || tid_to_cid[(size_t)r.tid] < 0) return -5;
        const uint32_t cid = (uint32_t)tid_to_cid[(size_t)r.tid];
        if (!J.picked(cid)) continue;
        decode += now_s() - t0;
        const int rc = fn(cid, r, p + 32);
        if (rc) return rc;
        t0 = now_s();
    }
    if (S.bad) return -3;
    decode += now_s() - t0;
    return 0;
}
}  // namespace

extern "C" __attribute__((visibility("default")))
// dir: an existing directory (NULL or "": off).  Every contig a later hyporef_fasta_* call polishes also leaves the reference's per-region
// dump there (not re-entrant: chdir).
void hyporef_set_dump_dir(const char* dir) { g_dump_dir = dir ? dir : ""; }

extern "C" __attribute__((visibility("default")))
// ... and with the long reads of a `-B` BAM file (long_bam_path; NULL = none): read whole first, then the short reads stream through and
// every contig runs the short-read stage, the long-read stage (prepare_long_windows, find_long_arms, fill_long_windows) and the POA of its
// SHORT and LONG windows.  ned_th = -n.  -6: a long read that the reference's NM filter would drop (this build cannot apply it).
// A process that ever ran a job WITHOUT long reads cannot run one with (Contig::set_no_long_reads is sticky): use a private copy of the library.
long hyporef_fasta_bam2(const char* draft_path, const char* bam_path, const char* long_bam_path, uint32_t k, const char* bvsd_path, uint32_t n_pick, const uint32_t* pick,
                        uint32_t min_mapq, uint32_t ned_th, const char* out_path, const int8_t* scores, double* seconds, uint64_t* counts) {
    FileJob J;
    J.long_run = long_bam_path != nullptr; J.ned_th = ned_th;
    if (!job_begin(J, draft_path, k, bvsd_path, n_pick, pick, min_mapq, out_path, scores)) return -1;
    double decode = 0;
    if (long_bam_path) {
        const long rc = read_bam(J, long_bam_path, decode, [&](uint32_t cid, const RawRec& r, const uint8_t* var) -> int {
            if (nm_filter_drops(r, var, J.ned_th)) return -6;
            ContigRecs& L = J.longs[cid];
            RawRec q = r; q.off = L.bytes.size();
            L.bytes.insert(L.bytes.end(), var, var + r.len);
            L.recs.push_back(q);
            return 0;
        });
        if (rc) return rc;
    }
    const long rc = read_bam(J, bam_path, decode, [&](uint32_t cid, const RawRec& r, const uint8_t* var) -> int { return J.record(cid, r, var); });
    if (rc) return rc;
    J.finish();
    job_report(J, decode, seconds, counts);
    return (long)J.T.contigs;
}

extern "C" __attribute__((visibility("default")))
long hyporef_fasta_bam(const char* draft_path, const char* bam_path, uint32_t k, const char* bvsd_path, uint32_t n_pick, const uint32_t* pick,
                       uint32_t min_mapq, const char* out_path, const int8_t* scores, double* seconds, uint64_t* counts) {
    return hyporef_fasta_bam2(draft_path, bam_path, nullptr, k, bvsd_path, n_pick, pick, min_mapq, 20, out_path, scores, seconds, counts);
}

extern "C" __attribute__((visibility("default")))
// The same from SAM text: every alignment line split on tabs, CIGAR and bases encoded as sam_parse1 stores them (operation codes
// "MIDNSHP=XB", 4-bit bases "=ACMGRSVTWYHKDBN", qualities minus 33 or 0xff for "*"), no tags (the short-read path reads none).
long hyporef_fasta_sam(const char* draft_path, const char* sam_path, uint32_t k, const char* bvsd_path, uint32_t n_pick, const uint32_t* pick,
                       uint32_t min_mapq, const char* out_path, const int8_t* scores, double* seconds, uint64_t* counts) {
    FileJob J;
    if (!job_begin(J, draft_path, k, bvsd_path, n_pick, pick, min_mapq, out_path, scores)) return -1;
    std::ifstream f(sam_path);
    if (!f.is_open()) return -1;
    double decode = 0, t0 = now_s();
    std::string line;
    std::vector<uint8_t> var;
    static const char kOps[] = "MIDNSHP=XB", kNt[] = "=ACMGRSVTWYHKDBN";
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '@') continue;
        if (line.back() == '\r') line.pop_back();
        std::vector<std::string> fld;
        { size_t a = 0; while (fld.size() < 11) { size_t b = line.find('\t', a); if (b == std::string::npos) { fld.emplace_back(line.substr(a)); break; } fld.emplace_back(line.substr(a, b - a)); a = b + 1; } }
        if (fld.size() < 11) return -3;
        RawRec r; std::memset(&r, 0, sizeof r);
        r.flag = (uint16_t)std::strtoul(fld[1].c_str(), nullptr, 10);
        r.mapq = (uint8_t)std::strtoul(fld[4].c_str(), nullptr, 10);
        if (r.flag & (0x4 | 0x100 | 0x200 | 0x400)) continue;
        if (r.mapq < min_mapq) continue;
        auto it = J.cname_to_id.find(fld[2]);
        if (it == J.cname_to_id.end()) return -5;
        const uint32_t cid = it->second;
        if (!J.picked(cid)) continue;
        r.tid = (int32_t)cid; r.pos = (int32_t)std::strtol(fld[3].c_str(), nullptr, 10) - 1; r.mtid = -1; r.mpos = -1;
        var.clear();
        var.insert(var.end(), fld[0].begin(), fld[0].end()); var.push_back(0);
        if (var.size() > 255) return -3;
        r.l_read_name = (uint8_t)var.size();
        uint32_t ncig = 0;
        if (fld[5] != "*") {
            const char* s = fld[5].c_str();
            while (*s) {
                char* e; const unsigned long n = std::strtoul(s, &e, 10);
                const char* o = std::strchr(kOps, *e);
                if (e == s || !*e || !o) return -3;
                const uint32_t v = (uint32_t)(n << 4) | (uint32_t)(o - kOps);
                var.insert(var.end(), (const uint8_t*)&v, (const uint8_t*)&v + 4);
                ++ncig; s = e + 1;
            }
        }
        r.n_cigar = ncig;
        const std::string& sq = fld[9];
        const size_t lq = sq == "*" ? 0 : sq.size();
        r.l_seq = (int32_t)lq;
        const size_t s0 = var.size();
        var.resize(s0 + (lq + 1) / 2 + lq, 0);
        for (size_t i = 0; i < lq; ++i) {
            const char ch = (char)std::toupper((unsigned char)sq[i]);
            const char* o = std::strchr(kNt, ch);
            const uint8_t code = (o && ch) ? (uint8_t)(o - kNt) : 15;
            var[s0 + (i >> 1)] |= (uint8_t)(code << ((~i & 1) << 2));
        }
        if (fld[10] == "*" || fld[10].size() != lq) std::memset(var.data() + s0 + (lq + 1) / 2, 0xff, lq);
        else for (size_t i = 0; i < lq; ++i) var[s0 + (lq + 1) / 2 + i] = (uint8_t)(fld[10][i] - 33);
        r.len = var.size();
        decode += now_s() - t0;
        const int rc = J.record(cid, r, var.data());
        if (rc) return rc;
        t0 = now_s();
    }
    decode += now_s() - t0;
    J.finish();
    job_report(J, decode, seconds, counts);
    return (long)J.T.contigs;
}
