// gen_e2e_fast.cpp — deterministic end-to-end inputs at the size of BASELINE's configs C3 / T1 (100 x 1 Mbp and larger), fast.
//
// Same model as tests/golden/gen_e2e.py (which is Python and takes 20 s per 5 Mbp): per contig a random truth, a draft with
// 0.4 % substitutions / insertions / deletions each, 30x reads of 150 bp with 0.2 % substitutions whose CIGARs are the
// composition of the truth -> draft edit script, as coordinate-sorted SAM text; the solid k-mers (canonical k-mers that occur
// exactly once in all truths together, no homopolymer at either end, both strands set: external/suk/src/SolidKmers.cpp:166-189)
// as aux/solid_kmers.bvsd + aux/stage.txt, which `hypo -i` loads instead of running KMC.  NOT byte-compatible with gen_e2e.py:
// its own generator (splitmix64 per contig, integer arithmetic only), so that a 100-contig set is written in seconds on all
// cores.  The golden of such a set is the md5 of what the REAL reference binary (tests/golden/build_reference_binary.sh) makes
// of these files, kept in tests/golden/<name>.manifest.json together with the checksums this program prints.
//
// usage: gen_e2e_fast <outdir> <seed> <n_contigs> <contig_len> <k> [coverage=30] [read_len=150] [read_sub_ppm=2000]
// build: g++ -O2 -fopenmp -o gen_e2e_fast gen_e2e_fast.cpp        (test infrastructure: tests/ and bench.py's e2e_c3 leg only)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

uint64_t fnv(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
    return h;
}

void append_uint(std::string& o, uint64_t v) { char b[24]; int n = snprintf(b, sizeof b, "%llu", (unsigned long long)v); o.append(b, (size_t)n); }

struct Contig {
    std::string truth, draft, sam;
    uint64_t n_reads = 0;
};

// ops: 0 = M (truth base == draft base), 1 = X (substituted), 2 = D (draft lacks the truth base), 3 = I (draft has an extra base)
void make_contig(Contig& c, uint64_t seed, int idx, uint32_t G, uint32_t cov, uint32_t rl, uint32_t sub_ppm) {
    // (the start state is itself a splitmix output of (seed, contig): states that differ by a multiple of the generator's
    // increment would give shifted copies of one stream)
    Rng seeder(seed * 0x100000001b3ull + 12345);
    uint64_t s0 = seeder.next() ^ (0xd1b54a32d192ed03ull * (uint64_t)(idx + 1));
    s0 = (s0 ^ (s0 >> 29)) * 0xbf58476d1ce4e5b9ull; s0 ^= s0 >> 32;
    Rng r(s0);
    c.truth.resize(G);
    for (uint32_t i = 0; i < G; ++i) c.truth[i] = kA[r.below(4)];
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
    { uint32_t t = 0, d = 0; for (size_t i = 0; i < n_ops; ++i) { dprefix[i] = d; if (tb[i]) tpos[t++] = (uint32_t)i; if (db[i]) { c.draft.push_back(db[i]); ++d; } } dprefix[n_ops] = d; }
    // reads in order of their truth start (=> non-decreasing draft start): k reads start at a position with the Poisson(cov / rl)
    // probabilities of k = 0, 1, 2 (more than two at one position: 0.1 % of the positions at 30x / 150 bp, folded into two)
    const double lam = (double)cov / rl;
    double e = 1.0; { double term = 1.0, sum = 1.0; for (int i = 1; i < 30; ++i) { term *= lam / i; sum += term; } e = 1.0 / sum; }   // exp(-lam)
    const uint32_t p0 = (uint32_t)(e * 1e6), p1 = (uint32_t)((e + e * lam) * 1e6);
    char name[32]; const int nl = snprintf(name, sizeof name, "ctg%d", idx + 1);
    std::string& o = c.sam;
    o.reserve((size_t)((double)G * lam * (rl + 48)));
    std::string seq, cig;
    uint64_t rid = 0;
    for (uint32_t s = 0; s + rl < G; ++s) {
        const uint32_t x = r.below(1000000u);
        const int k = x < p0 ? 0 : (x < p1 ? 1 : 2);
        for (int q = 0; q < k; ++q) {
            size_t i0 = tpos[s], i1 = (size_t)tpos[s + rl - 1] + 1;
            while (op[i0] > 1) ++i0;                       // a read starts and ends on a base both sequences have
            while (op[i1 - 1] > 1) --i1;
            seq.clear(); cig.clear();
            char last = 0; uint32_t run = 0;
            auto push = [&](char ch) { if (ch == last) ++run; else { if (last) { append_uint(cig, run); cig.push_back(last); } last = ch; run = 1; } };
            for (size_t i = i0; i < i1; ++i) {
                if (op[i] <= 1) { char b = tb[i]; if (r.ppm(sub_ppm)) b = kA[r.below(4)]; seq.push_back(b); push('M'); }
                else if (op[i] == 2) { seq.push_back(tb[i]); push('I'); }
                else push('D');
            }
            append_uint(cig, run); cig.push_back(last);
            o.push_back('r'); append_uint(o, rid++); o.append("\t0\t"); o.append(name, (size_t)nl); o.push_back('\t');
            append_uint(o, (uint64_t)dprefix[i0] + 1); o.append("\t60\t"); o += cig; o.append("\t*\t0\t0\t"); o += seq; o.append("\t*\n");
        }
    }
    c.n_reads = rid;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: gen_e2e_fast <outdir> <seed> <n_contigs> <contig_len> <k> [coverage=30] [read_len=150] [read_sub_ppm=2000]\n"); return 2; }
    const std::string out = argv[1];
    const uint64_t seed = strtoull(argv[2], nullptr, 10);
    const int nc = atoi(argv[3]);
    const uint32_t G = (uint32_t)strtoul(argv[4], nullptr, 10);
    const int K = atoi(argv[5]);
    const uint32_t cov = argc > 6 ? (uint32_t)atoi(argv[6]) : 30, rl = argc > 7 ? (uint32_t)atoi(argv[7]) : 150, sub = argc > 8 ? (uint32_t)atoi(argv[8]) : 2000;
    if (nc < 1 || G < 4 * rl || K < 5 || K > 15) { fprintf(stderr, "gen_e2e_fast: bad arguments\n"); return 2; }
    mkdir(out.c_str(), 0777); mkdir((out + "/aux").c_str(), 0777);
    FILE* fd = fopen((out + "/draft.fa").c_str(), "wb");
    FILE* fs = fopen((out + "/sr.sam").c_str(), "wb");
    if (!fd || !fs) { perror("gen_e2e_fast"); return 1; }
    {   // header
        std::string h = "@HD\tVN:1.6\tSO:coordinate\n";
        fwrite(h.data(), 1, h.size(), fs);
    }
    // k-mer counts over all truths (canonical, saturating at 2)
    const uint64_t nk = 1ull << (2 * K);
    std::vector<uint8_t> cnt(nk, 0);
    std::vector<Contig> cs((size_t)nc);
    uint64_t h_draft = 1469598103934665603ull, h_sam = 1469598103934665603ull, n_reads = 0, draft_bases = 0;
    std::vector<std::string> sq((size_t)nc);
    // contigs are generated in chunks of `par` in parallel and written in order; the @SQ lines need every draft length first, so
    // the records are kept per chunk and the header is completed by a first pass that only builds the drafts' lengths — cheaper:
    // generate everything chunk by chunk into a body file and prepend the header at the end.
    FILE* fb = fopen((out + "/sr.body.tmp").c_str(), "wb");
    if (!fb) { perror("gen_e2e_fast"); return 1; }
    const int par = 16;
    for (int c0 = 0; c0 < nc; c0 += par) {
        const int c1 = c0 + par < nc ? c0 + par : nc;
#pragma omp parallel for schedule(dynamic, 1)
        for (int c = c0; c < c1; ++c) make_contig(cs[(size_t)c], seed, c, G, cov, rl, sub);
        for (int c = c0; c < c1; ++c) {
            Contig& C = cs[(size_t)c];
            char hd[64]; const int n = snprintf(hd, sizeof hd, ">ctg%d\n", c + 1);
            fwrite(hd, 1, (size_t)n, fd); fwrite(C.draft.data(), 1, C.draft.size(), fd); fputc('\n', fd);
            h_draft = fnv(C.draft, h_draft);
            char q[96]; const int m = snprintf(q, sizeof q, "@SQ\tSN:ctg%d\tLN:%zu\n", c + 1, C.draft.size());
            sq[(size_t)c].assign(q, (size_t)m);
            fwrite(C.sam.data(), 1, C.sam.size(), fb);
            h_sam = fnv(C.sam, h_sam);
            n_reads += C.n_reads; draft_bases += C.draft.size();
            // canonical k-mers of the truth
            uint64_t fw = 0, rv = 0; const uint64_t mask = nk - 1;
            for (uint32_t i = 0; i < G; ++i) {
                const uint64_t b = (uint64_t)(strchr(kA, C.truth[i]) - kA);
                fw = ((fw << 2) | b) & mask;
                rv = (rv >> 2) | ((3 - b) << (2 * (K - 1)));
                if (i + 1 >= (uint32_t)K) { const uint64_t cn = fw < rv ? fw : rv; if (cnt[cn] < 2) ++cnt[cn]; }
            }
            std::string().swap(C.sam); std::string().swap(C.draft);
        }
    }
    fclose(fb); fclose(fd);
    for (int c = 0; c < nc; ++c) fwrite(sq[(size_t)c].data(), 1, sq[(size_t)c].size(), fs);
    {   // body behind the header
        FILE* fi = fopen((out + "/sr.body.tmp").c_str(), "rb");
        std::vector<char> buf(1 << 24);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), fi)) > 0) fwrite(buf.data(), 1, n, fs);
        fclose(fi); remove((out + "/sr.body.tmp").c_str());
    }
    fclose(fs);
    if (getenv("GEN_DEBUG")) { uint64_t hst[3] = {0, 0, 0}; for (uint64_t v = 0; v < nk; ++v) hst[cnt[v]]++; fprintf(stderr, "counts 0/1/2+: %llu %llu %llu\n", (unsigned long long)hst[0], (unsigned long long)hst[1], (unsigned long long)hst[2]); }
    // solid set: count == 1 and no homopolymer at either end; both strands
    std::vector<uint64_t> words(nk / 64, 0);
    uint64_t n_solid = 0;
    auto base_at = [&](uint64_t v, int pos) { return (int)((v >> (2 * (K - 1 - pos))) & 3); };      // pos 0 = first base
    for (uint64_t v = 0; v < nk; ++v) {
        if (cnt[v] != 1) continue;
        // v is canonical (the smaller of the two strands' codes, which is how gen_e2e.py's min() of the strings orders them too)
        if (base_at(v, 0) == base_at(v, 1) || base_at(v, K - 1) == base_at(v, K - 2)) continue;
        uint64_t rcv = 0;
        for (int p = 0; p < K; ++p) rcv |= (uint64_t)(3 - base_at(v, p)) << (2 * p);
        words[v >> 6] |= 1ull << (v & 63);
        words[rcv >> 6] |= 1ull << (rcv & 63);
        ++n_solid;
    }
    uint64_t h_bv = 1469598103934665603ull;
    {
        FILE* f = fopen((out + "/aux/solid_kmers.bvsd").c_str(), "wb");
        fwrite(&nk, 8, 1, f); fwrite(words.data(), 8, words.size(), f); fclose(f);
        for (uint64_t w : words) for (int b = 0; b < 8; ++b) h_bv = (h_bv ^ ((w >> (8 * b)) & 0xff)) * 1099511628211ull;
        f = fopen((out + "/aux/stage.txt").c_str(), "wb");
        fputs("Stage:SolidKmers [2026-09-28 12:00:00]\t1\n", f); fclose(f);
        f = fopen((out + "/reads.fa").c_str(), "wb");          // named on the command line, not read when -i finds the aux files
        fputs(">unused\nACGT\n", f); fclose(f);
    }
    printf("{\"contigs\": %d, \"draft_bases\": %llu, \"reads\": %llu, \"solid_kmers\": %llu, \"fnv_draft\": \"%016llx\", \"fnv_sam_records\": \"%016llx\", \"fnv_bitvector\": \"%016llx\"}\n",
           nc, (unsigned long long)draft_bases, (unsigned long long)n_reads, (unsigned long long)n_solid,
           (unsigned long long)h_draft, (unsigned long long)h_sam, (unsigned long long)h_bv);
    return 0;
}
