// gen_e2e_fast.cpp — deterministic end-to-end inputs at the size of BASELINE's configs C3 / C4 / T1 (100 x 1 Mbp ... 3 Gbp), fast.
//
// Same model as tests/golden/gen_e2e.py (which is Python and takes 20 s per 5 Mbp): per contig a random truth, a draft with
// 0.4 % substitutions / insertions / deletions each, 30x reads of 150 bp with 0.2 % substitutions whose CIGARs are the
// composition of the truth -> draft edit script, coordinate-sorted; the solid k-mers (canonical k-mers that occur
// exactly once in all truths together, no homopolymer at either end, both strands set: external/suk/src/SolidKmers.cpp:166-189)
// as aux/solid_kmers.bvsd + aux/stage.txt, which `hypo -i` loads instead of running KMC.  NOT byte-compatible with gen_e2e.py:
// its own generator (splitmix64 per contig, integer arithmetic only).  The golden of such a set is the md5 of what the REAL
// reference binary (tests/golden/build_reference_binary.sh) makes of these files, kept in tests/golden/<name>.manifest.json
// together with the checksums this program prints.
//
// Round 4: every contig's draft length comes from a first pass over the edit script alone (the @SQ lines / BAM reference table
// need them before the first record), so the records stream straight to their file — no second copy of a 116 GB body —; contigs
// are generated on all cores (`--threads`), a writer thread files chunk i while chunk i + 1 is made; k up to 17 (atomic saturating
// k-mer counters, the 4^k sweep on all cores); `--bam` writes the same records as BAM (BGZF, deflate level 1, compressed per
// contig on the generating thread) into sr.bam instead of sr.sam; `--fast-hash` replaces the serial FNV-1a over the whole
// files (1 GB/s: two minutes for a 3 Gbp set) by FNV-1a over the per-contig FNV-1a values.  Without the two flags the files
// and the printed checksums are byte for byte what the round-3 program produced (the committed manifests depend on that).
//
// Config C4 (one chr1-sized contig, short reads + noisy long reads, `-B`): `--join` files all pieces as ONE contig "ctg1" (the pieces
// are generated independently, so reads never cross a junction: a coverage dip every <contig_len> bases); `--long <cov> <len>` adds
// long reads (lr.sam / lr.bam: 3 % deletions, 3 % substitutions, 2 % insertions against the truth, CIGAR against the draft by
// composing the two edit scripts, NM tag) and `--gaps <every> <len>` leaves the short reads out of <len> bases every <every> bases,
// which is where the reference builds LONG windows from the long reads (src/Contig.cpp:292-343).
//
// Round 6 (non-i.i.d. genomes; every option off = the files of rounds 3-5 byte for byte, the new draws come from a generator of their own):
//   --repeats <ppm>     this share of the truth's bases lies in low-complexity blocks: tandem repeats (unit 2-12, 30-600 bases), homopolymer
//                       runs (6-40), dispersed copies of an earlier stretch of the contig (200-1500 bases), all with ~1 % divergence — k-mers
//                       there are not unique, so no solid positions: long weak regions, Contig::force_divide (src/Contig.cpp:630-711), minimizers
//                       that recur or are poly-base (src/Contig.cpp:455-524)
//   --diploid <ppm>     a second haplotype: SNPs at this rate, 1-base deletions and insertions at a fifth of it each; every read comes from
//                       one of the two with equal probability (solid k-mers and the draft come from the first): k-mers over a heterozygous
//                       site get about half the support — the 40-80 % branch of SR detection (src/Contig.cpp:96-127)
//   --read-indel <ppm>  deletion and insertion errors in the short reads at this rate each, five times that inside homopolymer runs
//   --mismap <ppm>      this share of the short reads carries the sequence of ANOTHER place of the contig under the CIGAR of this one
//
// usage: gen_e2e_fast <outdir> <seed> <n_contigs> <contig_len> <k> [coverage=30] [read_len=150] [read_sub_ppm=2000]
//                     [--bam] [--fast-hash] [--threads N] [--join] [--long <cov> <len>] [--gaps <every> <len>]
//                     [--repeats <ppm>] [--diploid <ppm>] [--read-indel <ppm>] [--mismap <ppm>] [--homopolymer <pos> <len>]
// build: g++ -O2 -fopenmp -o gen_e2e_fast gen_e2e_fast.cpp -lz      (test infrastructure: tests/ and bench.py's e2e legs only)
#include <zlib.h>
#include <omp.h>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>

namespace {

struct Rng {                                     // splitmix64
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }
    bool ppm(uint32_t p) { return below(1000000u) < p; }
};

const char kA[] = "ACGT";
constexpr uint64_t kFnv0 = 1469598103934665603ull;

uint64_t fnv(const char* p, size_t n, uint64_t h = kFnv0) {
    for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
    return h;
}
uint64_t fnv(const std::string& s, uint64_t h = kFnv0) { return fnv(s.data(), s.size(), h); }

void append_uint(std::string& o, uint64_t v) { char b[24]; int n = snprintf(b, sizeof b, "%llu", (unsigned long long)v); o.append(b, (size_t)n); }

struct Contig {
    std::string truth, draft, recs, lrecs;       // recs / lrecs: SAM text, or BGZF blocks of BAM records (short / long reads)
    uint64_t n_reads = 0, n_long = 0, h_draft = 0, h_recs = 0, h_lrecs = 0;
};
struct Extra {                                   // config C4 options
    bool join = false; uint64_t pos_off = 0;     // join: all pieces are one contig; this piece starts at pos_off of it
    uint32_t long_cov = 0, long_len = 0;         // long reads (0 = none)
    uint32_t gap_every = 0, gap_len = 0;         // short reads leave [x * gap_every + gap_every / 2, ... + gap_len) of the truth alone
    uint32_t rep_ppm = 0, het_ppm = 0, rindel_ppm = 0, mismap_ppm = 0;      // round 6: low-complexity truth, second haplotype, read indels, mis-placed reads
    uint32_t hp_pos = 0, hp_len = 0;             // one poly-A run of hp_len bases at truth position hp_pos of every contig (0 = none)
    bool realistic() const { return rep_ppm || het_ppm || rindel_ppm || mismap_ppm; }
};

Rng contig_rng(uint64_t seed, int idx) {
    // (the start state is itself a splitmix output of (seed, contig): states that differ by a multiple of the generator's
    // increment would give shifted copies of one stream)
    Rng seeder(seed * 0x100000001b3ull + 12345);
    uint64_t s0 = seeder.next() ^ (0xd1b54a32d192ed03ull * (uint64_t)(idx + 1));
    s0 = (s0 ^ (s0 >> 29)) * 0xbf58476d1ce4e5b9ull; s0 ^= s0 >> 32;
    return Rng(s0);
}

// BGZF (SAM spec 4.1): gzip members with a BC extra field, at most 64 KiB of payload each
struct Bgzf {
    std::string* out;
    std::string buf;
    explicit Bgzf(std::string* o) : out(o) { buf.reserve(0xff00); }
    void put(const void* p, size_t n) {
        const char* c = (const char*)p;
        while (n) {
            const size_t take = n < 0xff00 - buf.size() ? n : 0xff00 - buf.size();
            buf.append(c, take); c += take; n -= take;
            if (buf.size() == 0xff00) flush();
        }
    }
    void flush() {
        if (buf.empty()) return;
        unsigned char z[0x10000 + 64];
        z_stream zs; memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef*)buf.data(); zs.avail_in = (uInt)buf.size();
        zs.next_out = z + 18; zs.avail_out = sizeof z - 18 - 8;
        const int rc = deflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END) { fprintf(stderr, "gen_e2e_fast: deflate failed\n"); exit(1); }
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        static const unsigned char hd[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(z, hd, 16);
        const uint32_t bsize = (uint32_t)(clen + 18 + 8 - 1);
        z[16] = (unsigned char)(bsize & 0xff); z[17] = (unsigned char)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), (const Bytef*)buf.data(), (uInt)buf.size()), isz = (uint32_t)buf.size();
        memcpy(z + 18 + clen, &crc, 4); memcpy(z + 18 + clen + 4, &isz, 4);
        out->append((const char*)z, clen + 26);
        buf.clear();
    }
};
const unsigned char kBgzfEof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

int reg2bin(int64_t beg, int64_t end) {            // SAM spec 5.3
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

// the generator of the round-6 options of a contig: its own stream, so that the main one draws what it always drew
Rng extra_rng(uint64_t seed, int idx, uint64_t salt) {
    Rng base = contig_rng(seed ^ 0x6a09e667f3bcc909ull, idx);
    uint64_t s0 = base.next() ^ salt;
    s0 = (s0 ^ (s0 >> 31)) * 0x94d049bb133111ebull; s0 ^= s0 >> 29;
    return Rng(s0);
}

// low-complexity blocks written over a random truth (--repeats): the same for both passes over a contig
template <class Str> void apply_repeats(Str& truth, uint64_t seed, int idx, uint32_t G, uint32_t rep_ppm, uint32_t hp_pos = 0, uint32_t hp_len = 0) {
    // --homopolymer: a run no window border can be put into (Contig::force_divide never cuts inside one, src/Contig.cpp:641-666): one weak
    // region, and one window, at least as long as the run
    for (uint32_t j = 0; j < hp_len && hp_pos + j < G; ++j) truth[hp_pos + j] = 'A';
    if (!rep_ppm) return;
    Rng x = extra_rng(seed, idx, 0x7265706561747321ull);
    const uint32_t mean_block = 160;                              // blocks start with probability rep_ppm / mean_block per base
    uint32_t i = 0;
    while (i < G) {
        if (x.below(1000000u) >= rep_ppm / mean_block + 1) { ++i; continue; }
        const uint32_t kind = x.below(10);
        uint32_t len;
        if (kind < 4) {                                           // tandem repeat
            const uint32_t u = 2 + x.below(11);
            len = 30 + x.below(571);
            char unit[12];
            for (uint32_t j = 0; j < u; ++j) unit[j] = kA[x.below(4)];
            for (uint32_t j = 0; j < len && i + j < G; ++j) truth[i + j] = x.below(100) == 0 ? kA[x.below(4)] : unit[j % u];
        } else if (kind < 7) {                                    // homopolymer run
            len = 6 + x.below(35);
            const char b = kA[x.below(4)];
            for (uint32_t j = 0; j < len && i + j < G; ++j) truth[i + j] = b;
        } else {                                                  // dispersed copy of an earlier stretch
            len = 200 + x.below(1301);
            if (i > len + 1000) {
                const uint32_t from = x.below(i - len);
                for (uint32_t j = 0; j < len && i + j < G; ++j) truth[i + j] = x.below(100) == 0 ? kA[x.below(4)] : truth[from + j];
            }
        }
        i += len;
    }
}

// ops: 0 = M (truth base == draft base), 1 = X (substituted), 2 = D (draft lacks the truth base), 3 = I (draft has an extra base)
// cnt: saturating (at 2) counters of the canonical k-mers of all truths, shared by the generating threads
void make_contig(Contig& c, uint64_t seed, int idx, uint32_t G, uint32_t cov, uint32_t rl, uint32_t sub_ppm, bool bam, int K, uint8_t* cnt, const Extra& ex = Extra()) {
    Rng r = contig_rng(seed, idx);
    c.truth.resize(G);
    for (uint32_t i = 0; i < G; ++i) c.truth[i] = kA[r.below(4)];
    apply_repeats(c.truth, seed, idx, G, ex.rep_ppm, ex.hp_pos, ex.hp_len);
    // second haplotype (--diploid): per truth position 0 = as the first, 1 = another base, 2 = the base is missing, 3 = an extra base behind it
    std::vector<uint8_t> hv; std::vector<char> hb;
    if (ex.het_ppm) {
        Rng hx = extra_rng(seed, idx, 0x6469706c6f696421ull);
        hv.assign(G, 0); hb.assign(G, 0);
        for (uint32_t i = 0; i < G; ++i) {
            const uint32_t q = hx.below(1000000u);
            if (q < ex.het_ppm) { char d; do d = kA[hx.below(4)]; while (d == c.truth[i]); hv[i] = 1; hb[i] = d; }
            else if (q < ex.het_ppm + ex.het_ppm / 5) hv[i] = 2;
            else if (q < ex.het_ppm + 2 * (ex.het_ppm / 5)) { hv[i] = 3; hb[i] = kA[hx.below(4)]; }
        }
    }
    Rng rx = extra_rng(seed, idx, 0x7265616465727221ull);          // per-read draws of the round-6 options
    std::vector<uint8_t> op; std::vector<char> tb, db;       // per op: kind, truth base (0 = none), draft base (0 = none)
    op.reserve(G + G / 100); tb.reserve(G + G / 100); db.reserve(G + G / 100);
    for (uint32_t i = 0; i < G; ++i) {
        const char t = c.truth[i];
        const uint32_t x = r.below(1000000u);
        if (x < 4000) { op.push_back(2); tb.push_back(t); db.push_back(0); }
        else if (x < 8000) { char d; do d = kA[r.below(4)]; while (d == t); op.push_back(1); tb.push_back(t); db.push_back(d); }
        else { op.push_back(0); tb.push_back(t); db.push_back(t); }
        if (r.ppm(4000)) { op.push_back(3); tb.push_back(0); db.push_back(kA[r.below(4)]); }
    }
    const size_t n_ops = op.size();
    std::vector<uint32_t> tpos(G), dprefix(n_ops + 1);
    c.draft.clear();
    { uint32_t t = 0, d = 0; for (size_t i = 0; i < n_ops; ++i) { dprefix[i] = d; if (tb[i]) tpos[t++] = (uint32_t)i; if (db[i]) { c.draft.push_back(db[i]); ++d; } } dprefix[n_ops] = d; }
    // reads in order of their truth start (=> non-decreasing draft start): k reads start at a position with the Poisson(cov / rl)
    // probabilities of k = 0, 1, 2 (more than two at one position: 0.1 % of the positions at 30x / 150 bp, folded into two)
    const double lam = (double)cov / rl;
    double e = 1.0; { double term = 1.0, sum = 1.0; for (int i = 1; i < 30; ++i) { term *= lam / i; sum += term; } e = 1.0 / sum; }   // exp(-lam)
    const uint32_t p0 = (uint32_t)(e * 1e6), p1 = (uint32_t)((e + e * lam) * 1e6);
    char name[32]; const int nl = snprintf(name, sizeof name, "ctg%d", ex.join ? 1 : idx + 1);
    const int ref_id = ex.join ? 0 : idx;
    std::string& o = c.recs;
    o.clear();
    o.reserve((size_t)((double)G * lam * (bam ? (rl / 2 + 48) : (rl + 48))));
    Bgzf bz(&o);
    std::string seq, cig, rec;
    std::vector<uint32_t> cigops;
    uint64_t rid = 0;
    for (uint32_t s = 0; s + rl < G; ++s) {
        const uint32_t x = r.below(1000000u);
        int k = x < p0 ? 0 : (x < p1 ? 1 : 2);
        if (ex.gap_every && k) {                             // (the draws stay what they are: a gap only drops reads)
            const uint32_t g0 = s / ex.gap_every * ex.gap_every + ex.gap_every / 2, g1 = g0 + ex.gap_len;
            if (s + rl > g0 && s < g1) k = -k;
        }
        const bool dropped = k < 0;
        if (dropped) k = -k;
        for (int q = 0; q < k; ++q) {
            size_t i0 = tpos[s], i1 = (size_t)tpos[s + rl - 1] + 1;
            while (op[i0] > 1) ++i0;                       // a read starts and ends on a base both sequences have
            while (op[i1 - 1] > 1) --i1;
            seq.clear(); cig.clear(); cigops.clear();
            char last = 0; uint32_t run = 0, rspan = 0;
            auto emit = [&]() { if (!last) return; if (bam) cigops.push_back((run << 4) | (last == 'M' ? 0u : (last == 'I' ? 1u : 2u))); else { append_uint(cig, run); cig.push_back(last); } };
            auto push = [&](char ch) { if (ch == last) ++run; else { emit(); last = ch; run = 1; } };
            if (!ex.realistic()) {
            for (size_t i = i0; i < i1; ++i) {
                if (op[i] <= 1) { char b = tb[i]; if (r.ppm(sub_ppm)) b = kA[r.below(4)]; seq.push_back(b); push('M'); ++rspan; }
                else if (op[i] == 2) { seq.push_back(tb[i]); push('I'); }
                else { push('D'); ++rspan; }
            }
            } else {
                // round 6: the read follows one of two haplotypes and has indel errors of its own; its CIGAR against the draft composes
                // truth -> draft, haplotype and read errors.  First and last base stay matches (a record starts and ends on M).
                const bool hap_b = ex.het_ppm && rx.below(2) == 1;
                uint32_t ti = s;                              // truth position of the next op that has a truth base
                char prev = 0;
                for (size_t i = i0; i < i1; ++i) {
                    if (op[i] == 3) { push('D'); ++rspan; continue; }       // a base only the draft has
                    const bool edge = i == i0 || i + 1 == i1;
                    char b = tb[i];
                    const uint32_t t = ti++;
                    bool skip = false, extra = false; char xb = 0;
                    if (hap_b && !edge) { if (hv[t] == 1) b = hb[t]; else if (hv[t] == 2) skip = true; else if (hv[t] == 3) { extra = true; xb = hb[t]; } }
                    if (ex.rindel_ppm && !edge) {
                        const uint32_t rate = ex.rindel_ppm * (b == prev ? 5u : 1u);
                        if (rx.below(1000000u) < rate) skip = true;
                        if (rx.below(1000000u) < rate) { extra = true; xb = (b == prev) ? b : kA[rx.below(4)]; }
                    }
                    if (op[i] <= 1 && r.ppm(sub_ppm)) b = kA[r.below(4)];
                    if (skip) { if (op[i] <= 1) { push('D'); ++rspan; } }
                    else { seq.push_back(b); if (op[i] <= 1) { push('M'); ++rspan; } else push('I'); prev = b; }
                    if (extra) { seq.push_back(xb); push('I'); }
                }
                if (ex.mismap_ppm && rx.below(1000000u) < ex.mismap_ppm && G > 4 * rl) {
                    const uint32_t s2 = rx.below(G - (uint32_t)seq.size() - 1);
                    for (size_t j = 0; j < seq.size(); ++j) seq[j] = c.truth[s2 + j];
                }
            }
            emit();
            if (dropped) continue;                           // (every random draw of the read was made: the stream behind it is unchanged)
            if (!bam) {
                o.push_back('r'); append_uint(o, rid++); o.append("\t0\t"); o.append(name, (size_t)nl); o.push_back('\t');
                append_uint(o, ex.pos_off + (uint64_t)dprefix[i0] + 1); o.append("\t60\t"); o += cig; o.append("\t*\t0\t0\t"); o += seq; o.append("\t*\n");
            } else {
                char qn[24]; const int ql = snprintf(qn, sizeof qn, "r%llu", (unsigned long long)rid++) + 1;
                const int32_t pos = (int32_t)(ex.pos_off + dprefix[i0]), lseq = (int32_t)seq.size();
                const uint32_t ncig = (uint32_t)cigops.size();
                const int32_t bs = 32 + ql + 4 * (int32_t)ncig + (lseq + 1) / 2 + lseq;
                rec.resize((size_t)bs + 4);
                char* p = &rec[0];
                auto w32 = [&](int32_t v) { memcpy(p, &v, 4); p += 4; };
                auto w16 = [&](uint16_t v) { memcpy(p, &v, 2); p += 2; };
                w32(bs); w32(ref_id); w32(pos); *p++ = (char)ql; *p++ = 60; w16((uint16_t)reg2bin(pos, pos + (int64_t)rspan)); w16((uint16_t)ncig); w16(0); w32(lseq); w32(-1); w32(-1); w32(0);
                memcpy(p, qn, (size_t)ql); p += ql;
                memcpy(p, cigops.data(), 4 * (size_t)ncig); p += 4 * (size_t)ncig;
                for (int32_t i = 0; i < lseq; i += 2) {
                    auto code = [](char ch) { return ch == 'A' ? 1 : (ch == 'C' ? 2 : (ch == 'G' ? 4 : 8)); };
                    *p++ = (char)((code(seq[(size_t)i]) << 4) | (i + 1 < lseq ? code(seq[(size_t)i + 1]) : 0));
                }
                memset(p, 0xff, (size_t)lseq);
                bz.put(rec.data(), rec.size());
            }
        }
    }
    if (bam) bz.flush();
    c.n_reads = rid;
    c.lrecs.clear(); c.n_long = 0;
    if (ex.long_cov && ex.long_len && G > 2 * ex.long_len) {
        // long reads: a generator of their own (the short reads' stream is what it was without them); a read is the truth over
        // [s, s + len) with 3 % deletions, 3 % substitutions, 2 % insertions; its CIGAR against the DRAFT composes both edit scripts
        Rng lr(r.next() ^ 0x5bd1e9955bd1e995ull);
        Bgzf lbz(&c.lrecs);
        const uint32_t ll = ex.long_len;
        const uint32_t pstart = (uint32_t)((double)ex.long_cov / ll * 1e9);      // starts per 1e9 positions
        uint64_t lid = 0;
        std::vector<std::pair<uint64_t, std::string>> sorted_recs;       // (position, record): trimming a read's leading non-M operations can move its start past its successor's
        std::string lseq_s, lcig, lrec;
        std::vector<uint32_t> lops;
        for (uint32_t s = 0; s + ll < G; ++s) {
            if (lr.below(1000000000u) >= pstart) continue;
            size_t i0 = tpos[s], i1 = (size_t)tpos[s + ll - 1] + 1;
            lseq_s.clear(); lcig.clear(); lops.clear();
            char last = 0; uint32_t run = 0, rspan = 0, nm = 0; size_t first_m = (size_t)-1; uint32_t span_at_first = 0;
            // ops are collected first (kind per step), then trimmed to start and end on an M
            std::vector<char> kinds; std::vector<char> bases;
            kinds.reserve(ll + ll / 8); bases.reserve(ll + ll / 8);
            for (size_t i = i0; i < i1; ++i) {
                const uint32_t e = lr.below(1000u);
                const bool del = e < 30, sub = e >= 30 && e < 60;
                if (op[i] <= 1) {                              // truth and draft both have a base here
                    if (del) { kinds.push_back('D'); bases.push_back(0); }
                    else { char b = tb[i]; if (sub) b = kA[lr.below(4)]; kinds.push_back(b == db[i] ? 'M' : 'X'); bases.push_back(b); }
                } else if (op[i] == 2) {                       // the draft lacks this truth base
                    if (!del) { char b = tb[i]; if (sub) b = kA[lr.below(4)]; kinds.push_back('I'); bases.push_back(b); }
                } else { kinds.push_back('D'); bases.push_back(0); }        // a base only the draft has
                if (op[i] != 3 && lr.below(1000u) < 20) { kinds.push_back('I'); bases.push_back(kA[lr.below(4)]); }
            }
            size_t a = 0, b = kinds.size();
            uint32_t lead_ref = 0;
            while (a < b && kinds[a] != 'M' && kinds[a] != 'X') { if (kinds[a] == 'D') ++lead_ref; ++a; }
            while (b > a && kinds[b - 1] != 'M' && kinds[b - 1] != 'X') --b;
            if (b - a < ll / 2) continue;
            (void)first_m; (void)span_at_first;
            auto emit = [&]() { if (!last) return; if (bam) lops.push_back((run << 4) | (last == 'M' ? 0u : (last == 'I' ? 1u : 2u))); else { append_uint(lcig, run); lcig.push_back(last); } };
            auto push = [&](char ch) { if (ch == last) ++run; else { emit(); last = ch; run = 1; } };
            for (size_t t = a; t < b; ++t) {
                const char kd = kinds[t];
                if (kd == 'M' || kd == 'X') { push('M'); lseq_s.push_back(bases[t]); ++rspan; nm += kd == 'X'; }
                else if (kd == 'I') { push('I'); lseq_s.push_back(bases[t]); ++nm; }
                else { push('D'); ++rspan; ++nm; }
            }
            emit();
            const uint64_t pos0 = ex.pos_off + dprefix[i0] + lead_ref;      // 0-based position of the first M in the (joined) draft
            if (!bam) {
                sorted_recs.emplace_back(pos0, std::string());
                std::string& lo = sorted_recs.back().second;
                lo.push_back('l'); append_uint(lo, lid++); lo.append("\t0\t"); lo.append(name, (size_t)nl); lo.push_back('\t');
                append_uint(lo, pos0 + 1); lo.append("\t60\t"); lo += lcig; lo.append("\t*\t0\t0\t"); lo += lseq_s; lo.append("\t*\tNM:i:"); append_uint(lo, nm); lo.push_back('\n');
            } else {
                char qn[24]; const int ql = snprintf(qn, sizeof qn, "l%llu", (unsigned long long)lid++) + 1;
                const int32_t pos = (int32_t)pos0, lseq = (int32_t)lseq_s.size();
                const uint32_t ncig = (uint32_t)lops.size();
                const int32_t bs = 32 + ql + 4 * (int32_t)ncig + (lseq + 1) / 2 + lseq + 7;      // + NM:I tag
                lrec.resize((size_t)bs + 4);
                char* p = &lrec[0];
                auto w32 = [&](int32_t v) { memcpy(p, &v, 4); p += 4; };
                auto w16 = [&](uint16_t v) { memcpy(p, &v, 2); p += 2; };
                w32(bs); w32(ref_id); w32(pos); *p++ = (char)ql; *p++ = 60; w16((uint16_t)reg2bin(pos, pos + (int64_t)rspan)); w16((uint16_t)ncig); w16(0); w32(lseq); w32(-1); w32(-1); w32(0);
                memcpy(p, qn, (size_t)ql); p += ql;
                memcpy(p, lops.data(), 4 * (size_t)ncig); p += 4 * (size_t)ncig;
                for (int32_t i = 0; i < lseq; i += 2) {
                    auto code = [](char ch) { return ch == 'A' ? 1 : (ch == 'C' ? 2 : (ch == 'G' ? 4 : 8)); };
                    *p++ = (char)((code(lseq_s[(size_t)i]) << 4) | (i + 1 < lseq ? code(lseq_s[(size_t)i + 1]) : 0));
                }
                memset(p, 0xff, (size_t)lseq); p += lseq;
                *p++ = 'N'; *p++ = 'M'; *p++ = 'I'; memcpy(p, &nm, 4);
                sorted_recs.emplace_back(pos0, lrec);
            }
        }
        std::stable_sort(sorted_recs.begin(), sorted_recs.end(), [](const std::pair<uint64_t, std::string>& a, const std::pair<uint64_t, std::string>& b) { return a.first < b.first; });
        for (const auto& pr : sorted_recs) { if (bam) lbz.put(pr.second.data(), pr.second.size()); else c.lrecs += pr.second; }
        if (bam) lbz.flush();
        c.n_long = lid;
    }
    c.h_draft = fnv(c.draft);
    c.h_recs = fnv(c.recs);
    c.h_lrecs = fnv(c.lrecs);
    // canonical k-mers of the truth into the shared counters (saturating at 2: the result does not depend on the order)
    uint64_t fw = 0, rv = 0; const uint64_t mask = (1ull << (2 * K)) - 1;
    for (uint32_t i = 0; i < G; ++i) {
        const uint64_t b = (uint64_t)(strchr(kA, c.truth[i]) - kA);
        fw = ((fw << 2) | b) & mask;
        rv = (rv >> 2) | ((3 - b) << (2 * (K - 1)));
        if (i + 1 >= (uint32_t)K) {
            const uint64_t cn = fw < rv ? fw : rv;
            uint8_t v = __atomic_load_n(&cnt[cn], __ATOMIC_RELAXED);
            while (v < 2 && !__atomic_compare_exchange_n(&cnt[cn], &v, (uint8_t)(v + 1), true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
    }
    std::string().swap(c.truth);
}

// first pass: the draft's length only — the same draws as make_contig's truth and edit script (a substituted base is drawn until
// it differs from the truth base, so the truth bases are kept for the length of this call: 1 byte per base)
uint32_t draft_length_exact(uint64_t seed, int idx, uint32_t G, uint32_t rep_ppm, uint32_t hp_pos, uint32_t hp_len) {
    Rng r = contig_rng(seed, idx);
    std::vector<char> truth(G);
    for (uint32_t i = 0; i < G; ++i) truth[i] = kA[r.below(4)];
    apply_repeats(truth, seed, idx, G, rep_ppm, hp_pos, hp_len);
    uint32_t d = 0;
    for (uint32_t i = 0; i < G; ++i) {
        const char t = truth[i];
        const uint32_t x = r.below(1000000u);
        if (x < 4000) {}
        else if (x < 8000) { char b; do b = kA[r.below(4)]; while (b == t); ++d; }
        else ++d;
        if (r.ppm(4000)) { (void)r.below(4); ++d; }
    }
    return d;
}

}  // namespace

int main(int argc, char** argv) {
    std::vector<const char*> pos;
    bool bam = false, fast_hash = false;
    Extra ex0;
    int threads = omp_get_max_threads();
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--bam")) bam = true;
        else if (!strcmp(argv[i], "--fast-hash")) fast_hash = true;
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--join")) ex0.join = true;
        else if (!strcmp(argv[i], "--long") && i + 2 < argc) { ex0.long_cov = (uint32_t)atoi(argv[++i]); ex0.long_len = (uint32_t)atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--gaps") && i + 2 < argc) { ex0.gap_every = (uint32_t)atoi(argv[++i]); ex0.gap_len = (uint32_t)atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--repeats") && i + 1 < argc) ex0.rep_ppm = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--diploid") && i + 1 < argc) ex0.het_ppm = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--read-indel") && i + 1 < argc) ex0.rindel_ppm = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--mismap") && i + 1 < argc) ex0.mismap_ppm = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--homopolymer") && i + 2 < argc) { ex0.hp_pos = (uint32_t)atoi(argv[++i]); ex0.hp_len = (uint32_t)atoi(argv[++i]); }
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 5) { fprintf(stderr, "usage: gen_e2e_fast <outdir> <seed> <n_contigs> <contig_len> <k> [coverage=30] [read_len=150] [read_sub_ppm=2000] [--bam] [--fast-hash] [--threads N]\n"); return 2; }
    const std::string out = pos[0];
    const uint64_t seed = strtoull(pos[1], nullptr, 10);
    const int nc = atoi(pos[2]);
    const uint32_t G = (uint32_t)strtoul(pos[3], nullptr, 10);
    const int K = atoi(pos[4]);
    const uint32_t cov = pos.size() > 5 ? (uint32_t)atoi(pos[5]) : 30, rl = pos.size() > 6 ? (uint32_t)atoi(pos[6]) : 150, sub = pos.size() > 7 ? (uint32_t)atoi(pos[7]) : 2000;
    if (nc < 1 || G < 4 * rl || K < 5 || K > 17 || threads < 1) { fprintf(stderr, "gen_e2e_fast: bad arguments\n"); return 2; }
    omp_set_num_threads(threads);
    mkdir(out.c_str(), 0777); mkdir((out + "/aux").c_str(), 0777);
    FILE* fd = fopen((out + "/draft.fa").c_str(), "wb");
    FILE* fs = fopen((out + (bam ? "/sr.bam" : "/sr.sam")).c_str(), "wb");
    if (!fd || !fs) { perror("gen_e2e_fast"); return 1; }
    setvbuf(fd, nullptr, _IOFBF, 1 << 22); setvbuf(fs, nullptr, _IOFBF, 1 << 22);
    // ---- first pass: draft lengths (the header names them) ----
    std::vector<uint32_t> dlen((size_t)nc);
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < nc; ++c) dlen[(size_t)c] = draft_length_exact(seed, c, G, ex0.rep_ppm, ex0.hp_pos, ex0.hp_len);
    std::vector<uint64_t> pos_off((size_t)nc + 1, 0);
    for (int c = 0; c < nc; ++c) pos_off[(size_t)c + 1] = pos_off[(size_t)c] + dlen[(size_t)c];
    if (ex0.join && pos_off[(size_t)nc] >= 0x7fffffffull) { fprintf(stderr, "gen_e2e_fast: a joined contig of %llu bases exceeds BAM's 2^31\n", (unsigned long long)pos_off[(size_t)nc]); return 2; }
    FILE* fl = nullptr;
    if (ex0.long_cov) { fl = fopen((out + (bam ? "/lr.bam" : "/lr.sam")).c_str(), "wb"); if (!fl) { perror("gen_e2e_fast"); return 1; } setvbuf(fl, nullptr, _IOFBF, 1 << 22); }
    {   // header(s)
        std::string h = "@HD\tVN:1.6\tSO:coordinate\n";
        const int n_ref = ex0.join ? 1 : nc;
        auto ref_len = [&](int c) -> uint64_t { return ex0.join ? pos_off[(size_t)nc] : dlen[(size_t)c]; };
        for (int c = 0; c < n_ref; ++c) { char q[96]; const int m = snprintf(q, sizeof q, "@SQ\tSN:ctg%d\tLN:%llu\n", c + 1, (unsigned long long)ref_len(c)); h.append(q, (size_t)m); }
        std::string blk;
        if (bam) {
            std::string hb;
            hb.append("BAM\1", 4);
            const int32_t lt = (int32_t)h.size(), nr = n_ref;
            hb.append((const char*)&lt, 4); hb += h; hb.append((const char*)&nr, 4);
            for (int c = 0; c < n_ref; ++c) {
                char nm[32]; const int32_t ln = snprintf(nm, sizeof nm, "ctg%d", c + 1) + 1, lr = (int32_t)ref_len(c);
                hb.append((const char*)&ln, 4); hb.append(nm, (size_t)ln); hb.append((const char*)&lr, 4);
            }
            Bgzf bz(&blk); bz.put(hb.data(), hb.size()); bz.flush();
        }
        const std::string& hd = bam ? blk : h;
        fwrite(hd.data(), 1, hd.size(), fs);
        if (fl) fwrite(hd.data(), 1, hd.size(), fl);
    }
    // k-mer counts over all truths (canonical, saturating at 2)
    const uint64_t nk = 1ull << (2 * K);
    uint8_t* cnt = (uint8_t*)calloc(nk, 1);
    if (!cnt) { fprintf(stderr, "gen_e2e_fast: no memory for %llu k-mer counters\n", (unsigned long long)nk); return 1; }
    uint64_t h_draft = kFnv0, h_sam = kFnv0, h_long = kFnv0, n_reads = 0, n_long = 0, draft_bases = 0;
    if (ex0.join) fputs(">ctg1\n", fd);
    // chunks of `par` contigs: made on all threads, filed in order by a writer thread while the next chunk is made
    const int par = threads < 16 ? 16 : threads;
    std::vector<Contig> bufs[2]; bufs[0].resize((size_t)par); bufs[1].resize((size_t)par);
    std::thread writer;
    auto file_chunk = [&](std::vector<Contig>* cs, int c0, int c1) {
        for (int c = c0; c < c1; ++c) {
            Contig& C = (*cs)[(size_t)(c - c0)];
            if (C.draft.size() != dlen[(size_t)c]) { fprintf(stderr, "gen_e2e_fast: internal error: draft length of contig %d\n", c + 1); exit(1); }
            if (!ex0.join) { char hd[64]; const int n = snprintf(hd, sizeof hd, ">ctg%d\n", c + 1); fwrite(hd, 1, (size_t)n, fd); }
            fwrite(C.draft.data(), 1, C.draft.size(), fd);
            if (!ex0.join) fputc('\n', fd);
            fwrite(C.recs.data(), 1, C.recs.size(), fs);
            if (fl) fwrite(C.lrecs.data(), 1, C.lrecs.size(), fl);
            if (fast_hash) { h_draft = fnv((const char*)&C.h_draft, 8, h_draft); h_sam = fnv((const char*)&C.h_recs, 8, h_sam); h_long = fnv((const char*)&C.h_lrecs, 8, h_long); }
            else { h_draft = fnv(C.draft, h_draft); h_sam = fnv(C.recs, h_sam); h_long = fnv(C.lrecs, h_long); }
            n_reads += C.n_reads; n_long += C.n_long; draft_bases += C.draft.size();
            std::string().swap(C.recs); std::string().swap(C.lrecs); std::string().swap(C.draft);
        }
    };
    int flip = 0;
    for (int c0 = 0; c0 < nc; c0 += par, flip ^= 1) {
        const int c1 = c0 + par < nc ? c0 + par : nc;
        std::vector<Contig>& cs = bufs[flip];
#pragma omp parallel for schedule(dynamic, 1)
        for (int c = c0; c < c1; ++c) { Extra ex = ex0; ex.pos_off = ex0.join ? pos_off[(size_t)c] : 0; make_contig(cs[(size_t)(c - c0)], seed, c, G, cov, rl, sub, bam, K, cnt, ex); }
        if (writer.joinable()) writer.join();
        writer = std::thread(file_chunk, &cs, c0, c1);
    }
    if (writer.joinable()) writer.join();
    if (bam) { fwrite(kBgzfEof, 1, sizeof kBgzfEof, fs); if (fl) fwrite(kBgzfEof, 1, sizeof kBgzfEof, fl); }
    if (ex0.join) fputc('\n', fd);
    fclose(fd); fclose(fs); if (fl) fclose(fl);
    if (getenv("GEN_DEBUG")) { uint64_t hst[3] = {0, 0, 0}; for (uint64_t v = 0; v < nk; ++v) hst[cnt[v]]++; fprintf(stderr, "counts 0/1/2+: %llu %llu %llu\n", (unsigned long long)hst[0], (unsigned long long)hst[1], (unsigned long long)hst[2]); }
    // solid set: count == 1 and no homopolymer at either end; both strands
    std::vector<uint64_t> words(nk / 64, 0);
    uint64_t n_solid = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_solid)
    for (int64_t wv = 0; wv < (int64_t)(nk / 64); ++wv) {
        for (uint64_t v = (uint64_t)wv * 64; v < (uint64_t)wv * 64 + 64; ++v) {
            if (cnt[v] != 1) continue;
            auto base_at = [&](uint64_t x, int p) { return (int)((x >> (2 * (K - 1 - p))) & 3); };      // p 0 = first base
            // v is canonical (the smaller of the two strands' codes, which is how gen_e2e.py's min() of the strings orders them too)
            if (base_at(v, 0) == base_at(v, 1) || base_at(v, K - 1) == base_at(v, K - 2)) continue;
            uint64_t rcv = 0;
            for (int p = 0; p < K; ++p) rcv |= (uint64_t)(3 - base_at(v, p)) << (2 * p);
            __atomic_fetch_or(&words[v >> 6], 1ull << (v & 63), __ATOMIC_RELAXED);
            __atomic_fetch_or(&words[rcv >> 6], 1ull << (rcv & 63), __ATOMIC_RELAXED);
            ++n_solid;
        }
    }
    free(cnt);
    uint64_t h_bv = kFnv0;
    {
        FILE* f = fopen((out + "/aux/solid_kmers.bvsd").c_str(), "wb");
        fwrite(&nk, 8, 1, f); fwrite(words.data(), 8, words.size(), f); fclose(f);
        if (fast_hash) {                                         // FNV-1a over the FNV-1a of 1 MiB pieces
            const size_t piece = (size_t)1 << 17, np = (words.size() + piece - 1) / piece;      // words per piece
            std::vector<uint64_t> hp(np);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < (int64_t)np; ++i) {
                const size_t a = (size_t)i * piece, b = a + piece < words.size() ? a + piece : words.size();
                hp[(size_t)i] = fnv((const char*)(words.data() + a), (b - a) * 8);
            }
            h_bv = fnv((const char*)hp.data(), hp.size() * 8);
        } else h_bv = fnv((const char*)words.data(), words.size() * 8);
        f = fopen((out + "/aux/stage.txt").c_str(), "wb");
        fputs("Stage:SolidKmers [2026-09-28 12:00:00]\t1\n", f); fclose(f);
        f = fopen((out + "/reads.fa").c_str(), "wb");          // named on the command line, not read when -i finds the aux files
        fputs(">unused\nACGT\n", f); fclose(f);
    }
 char extra[384] = "";
    if (ex0.realistic())
        snprintf(extra, sizeof extra, ", \"joined\": %s, \"long_reads\": %llu, \"fnv_long_records\": \"%016llx\", \"gaps\": [%u, %u], \"repeats_ppm\": %u, \"diploid_ppm\": %u, \"read_indel_ppm\": %u, \"mismap_ppm\": %u",
                 ex0.join ? "true" : "false", (unsigned long long)n_long, (unsigned long long)h_long, ex0.gap_every, ex0.gap_len, ex0.rep_ppm, ex0.het_ppm, ex0.rindel_ppm, ex0.mismap_ppm);
    else if (ex0.long_cov || ex0.join || ex0.gap_every)
        snprintf(extra, sizeof extra, ", \"joined\": %s, \"long_reads\": %llu, \"fnv_long_records\": \"%016llx\", \"gaps\": [%u, %u]", ex0.join ? "true" : "false",
                 (unsigned long long)n_long, (unsigned long long)h_long, ex0.gap_every, ex0.gap_len);
    printf("{\"contigs\": %d, \"draft_bases\": %llu, \"reads\": %llu, \"solid_kmers\": %llu, \"fnv_draft\": \"%016llx\", \"%s\": \"%016llx\", \"fnv_bitvector\": \"%016llx\"%s%s}\n",
           nc, (unsigned long long)draft_bases, (unsigned long long)n_reads, (unsigned long long)n_solid,
           (unsigned long long)h_draft, bam ? "fnv_bam_blocks" : "fnv_sam_records", (unsigned long long)h_sam, (unsigned long long)h_bv,
           fast_hash ? ", \"hash\": \"fnv of per-contig fnv\"" : "", extra);
    return 0;
}
