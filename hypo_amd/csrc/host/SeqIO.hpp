// SeqIO.hpp — FASTA/FASTQ and SAM-text readers for the host pipeline (plain or gzip, via zlib).
// The reference reads drafts with klib's kseq (src/Hypo.cpp:82-95) and alignments with htslib
// (src/Hypo.cpp:278-329); only the fields HyPo consumes are parsed here: FLAG, RNAME, POS, MAPQ, CIGAR, SEQ and
// the NM:i tag (src/Alignment.cpp:514-571, :51-58).  BAM (BGZF) input is not decoded by this reader.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace hypo {

class LineReader {                           // lines of a plain or gzip file: block reads through zlib, memchr for the line ends
public:
    explicit LineReader(const std::string& path) : _fp(gzopen(path.c_str(), "r")), _buf(kBuf) { if (_fp) gzbuffer(_fp, 1 << 20); }
    ~LineReader() { if (_fp) gzclose(_fp); }
    LineReader(const LineReader&) = delete;
    LineReader& operator=(const LineReader&) = delete;
    bool ok() const { return _fp != nullptr; }
    bool next(std::string& line) {
        line.clear();
        if (!_fp) return false;
        bool any = false;
        for (;;) {
            if (_pos == _end) {
                if (_eof) return any;
                const int n = gzread(_fp, _buf.data(), (unsigned)kBuf);
                if (n <= 0) { _eof = true; return any; }
                _pos = 0; _end = (size_t)n;
            }
            const char* b = _buf.data() + _pos;
            const char* nl = (const char*)std::memchr(b, '\n', _end - _pos);
            any = true;
            if (nl) {
                line.append(b, (size_t)(nl - b));
                _pos += (size_t)(nl - b) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(b, _end - _pos);
            _pos = _end;
        }
    }
private:
    static constexpr size_t kBuf = 4u << 20;
    gzFile _fp;
    std::vector<char> _buf;
    size_t _pos = 0, _end = 0;
    bool _eof = false;
};

struct FastaRecord { std::string name, seq; };

// name = first token of the header line (kseq: ks->name), multi-line sequences, FASTA or FASTQ
inline bool read_fastx(const std::string& path, std::vector<FastaRecord>& out) {
    LineReader lr(path);
    if (!lr.ok()) return false;
    std::string line;
    bool have = lr.next(line);
    while (have) {
        if (line.empty()) { have = lr.next(line); continue; }
        if (line[0] != '>' && line[0] != '@') return false;
        const bool fq = line[0] == '@';
        FastaRecord r;
        const size_t e = line.find_first_of(" \t");
        r.name = line.substr(1, e == std::string::npos ? std::string::npos : e - 1);
        have = lr.next(line);
        while (have && !(line.size() && (line[0] == '>' || (fq && line[0] == '+')))) { r.seq += line; have = lr.next(line); }
        if (fq && have) {                                   // skip qualities: as many characters as bases
            size_t q = 0;
            have = lr.next(line);
            while (have && q < r.seq.size()) { q += line.size(); have = lr.next(line); }
        }
        out.push_back(std::move(r));
    }
    return true;
}

// BAM flag bits and CIGAR encoding as in the SAM specification / htslib
enum { SAM_FUNMAP = 4, SAM_FSECONDARY = 256, SAM_FQCFAIL = 512, SAM_FDUP = 1024 };
enum { CIG_M = 0, CIG_I = 1, CIG_D = 2, CIG_N = 3, CIG_S = 4, CIG_H = 5, CIG_P = 6, CIG_EQ = 7, CIG_X = 8 };
inline uint32_t cigar_op(uint32_t c) { return c & 0xf; }
inline uint32_t cigar_len(uint32_t c) { return c >> 4; }
inline uint32_t cigar_type(uint32_t op) { return (0x3C1A7u >> (op << 1)) & 3; }   // bit 0 consumes query, bit 1 reference

struct SamRecord {
    std::string qname, seq;
    int32_t tid = -1;
    uint32_t flag = 0, pos = 0, mapq = 0;      // pos 0-based
    std::vector<uint32_t> cigar;               // op | len << 4
    bool has_nm = false;
    int64_t nm = 0;
};

class SamReader {
public:
    explicit SamReader(const std::string& path) : _lr(path) {
        // header: @SQ SN: names define the tids
        while (_lr.next(_pending)) {
            if (_pending.empty() || _pending[0] != '@') { _have_pending = true; break; }
            if (_pending.compare(0, 3, "@SQ") == 0) {
                const size_t p = _pending.find("\tSN:");
                if (p != std::string::npos) {
                    const size_t e = _pending.find('\t', p + 4);
                    _names.push_back(_pending.substr(p + 4, e == std::string::npos ? std::string::npos : e - p - 4));
                    _tid[_names.back()] = (int32_t)_names.size() - 1;
                }
            }
        }
    }
    bool ok() const { return _lr.ok(); }
    const std::string& tid2name(int32_t tid) const { return _names[(size_t)tid]; }
    bool next(SamRecord& r) {
        std::string line;
        if (!next_line(line)) return false;
        parse(line, r);
        return true;
    }
    // next non-empty alignment line (I/O and inflate are serial; parsing is not, see parse())
    bool next_line(std::string& line) {
        if (_have_pending) { line.swap(_pending); _have_pending = false; }
        else if (!_lr.next(line)) return false;
        while (line.empty()) if (!_lr.next(line)) return false;
        return true;
    }
    // up to `max` lines appended to `out`; false at end of file
    bool read_lines(std::vector<std::string>& out, size_t max) {
        std::string line;
        for (size_t i = 0; i < max; ++i) {
            if (!next_line(line)) return false;
            out.emplace_back(std::move(line));
            line.clear();
        }
        return true;
    }
    // one alignment line -> record; const and re-entrant (Hypo::create_alignments parses blocks of lines in parallel)
    void parse(const std::string& line, SamRecord& r) const {
        size_t f[12]; int nf = 0; f[0] = 0;
        for (size_t i = 0; i < line.size() && nf < 11; ++i) if (line[i] == '\t') f[++nf] = i + 1;
        if (nf < 10) { std::fprintf(stderr, "[Hypo::SamReader] Error: malformed SAM record: %s\n", line.substr(0, 60).c_str()); std::exit(1); }
        auto field = [&](int k) { const size_t b = f[k], e = (k < nf ? f[k + 1] - 1 : line.size()); return line.substr(b, e - b); };
        r.qname = field(0);
        r.flag = (uint32_t)std::strtoul(line.c_str() + f[1], nullptr, 10);
        const std::string rname = field(2);
        auto it = _tid.find(rname);
        r.tid = it == _tid.end() ? -1 : it->second;
        r.pos = (uint32_t)(std::strtoul(line.c_str() + f[3], nullptr, 10) - 1);
        r.mapq = (uint32_t)std::strtoul(line.c_str() + f[4], nullptr, 10);
        r.cigar.clear();
        const char* c = line.c_str() + f[5];
        if (*c != '*') {
            while (*c && *c != '\t') {
                char* e;
                const uint32_t len = (uint32_t)std::strtoul(c, &e, 10);
                const char* ops = "MIDNSHP=X";
                const char* q = std::strchr(ops, *e);
                if (!q || !*e) { std::fprintf(stderr, "[Hypo::SamReader] Error: bad CIGAR in %s\n", r.qname.c_str()); std::exit(1); }
                r.cigar.push_back((uint32_t)(q - ops) | (len << 4));
                c = e + 1;
            }
        }
        r.seq = field(9);
        r.has_nm = false;
        if (nf >= 11) {
            const size_t p = line.find("\tNM:i:", f[10] - 1);
            if (p != std::string::npos) { r.has_nm = true; r.nm = std::strtoll(line.c_str() + p + 6, nullptr, 10); }
        }
    }
private:
    LineReader _lr;
    std::vector<std::string> _names;
    std::unordered_map<std::string, int32_t> _tid;
    std::string _pending;
    bool _have_pending = false;
};

}  // namespace hypo
