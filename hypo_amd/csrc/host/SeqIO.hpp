// SeqIO.hpp — FASTA/FASTQ and SAM-text readers for the host pipeline (plain or gzip, via zlib).
// The reference reads drafts with klib's kseq (src/Hypo.cpp:82-95) and alignments with htslib
// (src/Hypo.cpp:278-329); only the fields HyPo consumes are parsed here: FLAG, RNAME, POS, MAPQ, CIGAR, SEQ and
// the NM:i tag (src/Alignment.cpp:514-571, :51-58).  Alignment files may be SAM text (plain or gzip) or BAM: BGZF is a
// series of gzip members, which zlib's gz* layer inflates as one stream, and the BAM records are decoded here.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <omp.h>
#include <zlib.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace hypo {

// Raw-deflate decoder for BGZF blocks.  libdeflate (whole-buffer decoder, word-at-a-time copies, carry-less-multiply CRC-32) inflates
// a 64 KiB block two to three times as fast as zlib 1.2.11's streaming inflate() and checks it ten times as fast; the image ships
// its runtime library without headers, so its four entry points are bound by name at the first use.  Not there (or HYPO_INFLATE=zlib):
// zlib, as before.  Both produce the same bytes or fail the same blocks; the CRC of every block is checked either way.
struct BlockInflater {
    typedef void* (*alloc_fn)();
    typedef int (*inflate_fn)(void*, const void*, size_t, void*, size_t, size_t*);
    typedef uint32_t (*crc_fn)(uint32_t, const void*, size_t);
    typedef void (*free_fn)(void*);
    struct Api { alloc_fn alloc = nullptr; inflate_fn inflate = nullptr; crc_fn crc = nullptr; free_fn free_ = nullptr; };
    static const Api& api() {
        static const Api a = [] {
            Api x;
            const char* e = std::getenv("HYPO_INFLATE");
            if (e && std::strcmp(e, "zlib") == 0) return x;
            void* h = ::dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) return x;
            x.alloc = (alloc_fn)::dlsym(h, "libdeflate_alloc_decompressor");
            x.inflate = (inflate_fn)::dlsym(h, "libdeflate_deflate_decompress");
            x.crc = (crc_fn)::dlsym(h, "libdeflate_crc32");
            x.free_ = (free_fn)::dlsym(h, "libdeflate_free_decompressor");
            if (!x.alloc || !x.inflate || !x.crc || !x.free_) x = Api();
            return x;
        }();
        return a;
    }
    static const char* name() { return api().inflate ? "libdeflate" : "zlib"; }
    void* d = nullptr;
    ~BlockInflater() { if (d) api().free_(d); }
    // n_out bytes out of n_in: 0 = good, 1 = does not inflate (to that size), 2 = CRC mismatch
    int run(const unsigned char* in, size_t n_in, char* out, size_t n_out, uint32_t want_crc) {
        const Api& a = api();
        if (a.inflate) {
            if (!d && !(d = a.alloc())) return 1;
            size_t got = 0;
            if (a.inflate(d, in, n_in, out, n_out, &got) != 0 || got != n_out) return 1;
            return a.crc(0, out, n_out) == want_crc ? 0 : 2;
        }
        z_stream zs; std::memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return 1;
        zs.next_in = (Bytef*)in; zs.avail_in = (uInt)n_in;
        zs.next_out = (Bytef*)out; zs.avail_out = (uInt)n_out;
        const int rc = inflate(&zs, Z_FINISH);
        const bool good = rc == Z_STREAM_END && zs.total_out == n_out;
        inflateEnd(&zs);
        if (!good) return 1;
        return (uint32_t)crc32(crc32(0L, nullptr, 0), (const Bytef*)out, (uInt)n_out) == want_crc ? 0 : 2;
    }
};

class LineReader {                           // lines of a plain or gzip file: block reads through zlib, memchr for the line ends
public:
    // A plain (not gzip) regular file is mapped and consumed in place: no read() copies, no zlib pass-through; everything else
    // goes through zlib's gz* layer in 4 MiB blocks.
    explicit LineReader(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd >= 0) {
            struct stat sb;
            unsigned char magic[2] = {0, 0};
            unsigned char hd[18] = {0};
            const bool regular = ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0;
            if (regular && ::pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
                void* m = ::mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) { _map = (const char*)m; _map_len = (size_t)sb.st_size; (void)::madvise(m, _map_len, MADV_SEQUENTIAL); _src = _map; _end = _map_len; _eof = true; }
            } else if (regular && sb.st_size >= 28 && ::pread(fd, hd, 18, 0) == 18 && hd[0] == 0x1f && hd[1] == 0x8b && hd[2] == 8 && (hd[3] & 4) &&
                       hd[10] == 6 && hd[11] == 0 && hd[12] == 'B' && hd[13] == 'C' && !std::getenv("HYPO_BGZF_SERIAL")) {
                // BGZF (a BAM, or a bgzip-ed SAM): the file is mapped and its blocks are inflated side by side (fill_bgzf)
                void* m = ::mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) { _cmap = (const unsigned char*)m; _cmap_len = (size_t)sb.st_size; (void)::madvise(m, _cmap_len, MADV_SEQUENTIAL); _src = nullptr; }
            }
            ::close(fd);
        }
        if (!_map && !_cmap) { _fp = gzopen(path.c_str(), "r"); if (_fp) { gzbuffer(_fp, 1 << 20); _buf.resize(kBuf); _src = _buf.data(); } }
    }
    ~LineReader() { if (_fp) gzclose(_fp); if (_map) ::munmap((void*)_map, _map_len); if (_cmap) ::munmap((void*)_cmap, _cmap_len); }
    // threads that inflate BGZF blocks side by side (a reader thread calls the read functions; its team is its own)
    void set_inflate_threads(int n) { _inflate_threads = n < 1 ? 1 : n; _bgzf_run = std::max<size_t>((size_t)32 << 20, (size_t)_inflate_threads << 21); }
    LineReader(const LineReader&) = delete;
    LineReader& operator=(const LineReader&) = delete;
    bool ok() const { return _fp != nullptr || _map != nullptr || _cmap != nullptr; }
    bool next(std::string& line) {
        line.clear();
        if (!ok()) return false;
        bool any = false;
        for (;;) {
            if (_pos == _end) {
                if (_eof || !refill()) return any;
            }
            const char* b = _src + _pos;
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
    // next line appended to dst WITHOUT its line end (so that a block of records can live in one buffer); false at end of file
    bool append_line(std::vector<char>& dst) {
        if (!ok()) return false;
        bool any = false;
        for (;;) {
            if (_pos == _end) {
                if (_eof || !refill()) return any;
            }
            const char* b = _src + _pos;
            const char* nl = (const char*)std::memchr(b, '\n', _end - _pos);
            any = true;
            if (nl) {
                dst.insert(dst.end(), b, nl);
                _pos += (size_t)(nl - b) + 1;
                if (!dst.empty() && dst.back() == '\r') dst.pop_back();
                return true;
            }
            dst.insert(dst.end(), b, b + (_end - _pos));
            _pos = _end;
        }
    }
    // exactly n raw bytes of the (inflated) stream; false at end of file
    bool read_bytes(void* dst, size_t n) {
        char* d = (char*)dst;
        while (n) {
            if (_pos == _end) {
                if (_eof || !refill()) return false;
            }
            const size_t take = n < _end - _pos ? n : _end - _pos;
            std::memcpy(d, _src + _pos, take);
            d += take; _pos += take; n -= take;
        }
        return true;
    }
    // a mapped plain file: the unread bytes in place (SamReader cuts records out of them without copying)
    bool mapped() const { return _map != nullptr && _map[_map_len - 1] == '\n'; }   // (in-place parsing wants every record terminated)
    const char* map_data() const { return _map; }                                    // a mapped plain file as it is (nullptr: not mapped)
    size_t map_size() const { return _map_len; }
    const char* unread() const { return _src + _pos; }
    size_t unread_bytes() const { return _end - _pos; }
    void consume(size_t n) {
        _pos += n;
        // a mapped file: pages far behind the read position are handed back (every record in flight — the block being parsed and
        // the one read ahead, 32 MB each at most — lies within the last 256 MB), so that a 4 GB alignment file does not sit in
        // the process's resident set
        constexpr size_t kKeep = (size_t)256 << 20, kStep = (size_t)64 << 20;
        if (_map && _pos > kKeep && _pos - kKeep >= _released + kStep) {
            const size_t upto = (_pos - kKeep) & ~(size_t)4095;
            (void)::madvise((void*)(_map + _released), upto - _released, MADV_DONTNEED);
            _released = upto;
        }
    }
    // first bytes of the stream without consuming them (used once, right after opening)
    bool starts_with(const char* magic, size_t n) {
        if (_pos == _end && !_eof) (void)refill();
        return _end - _pos >= n && std::memcmp(_src + _pos, magic, n) == 0;
    }
    // the unread part of the buffer in hand and a way to take bytes from it without copying (BAM records are cut out in place)
    const char* buffered() const { return _src + _pos; }
    size_t buffered_bytes() const { return _end - _pos; }
    void skip(size_t n) { _pos += n; }
    bool more() { return _pos < _end || (!_eof && refill()); }
    // a mapped BGZF file: more inflated bytes BEHIND the unread ones (which move to the front of the other buffer); false at the
    // end of the file.  Pointers into the buffer that was current before the call stay valid until the call after the next.
    bool bgzf() const { return _cmap != nullptr; }
    bool extend() { return _cmap && !_eof && fill_bgzf(); }
private:
    // the next stretch of the (inflated) stream into _buf; false at the end of the file
    bool refill() {
        if (_cmap) return fill_bgzf();
        if (!_fp) { _eof = true; return false; }
        const int n = gzread(_fp, _buf.data(), (unsigned)kBuf);
        if (n <= 0) { _eof = true; return false; }
        _pos = 0; _end = (size_t)n;
        return true;
    }
    // BGZF (SAM spec 4.1): independent gzip members of at most 64 KiB with their compressed size in a "BC" extra field and the
    // inflated size in the trailer.  A run of blocks (about 32 MB inflated) is located by hopping over the sizes, then inflated side
    // by side, every block straight to its place in _buf; the CRC-32 of every block is checked as htslib does.  gzread inflated
    // the file as ONE stream on one thread: 0.3 GB/s, the bound of every BAM run (a 100 Mbp / 30x set holds 5.6 GB of records).
    bool fill_bgzf() {
        struct Blk { size_t at, data, clen, out; uint32_t isize; };
        std::vector<Blk> blks;
        size_t total = 0;
        while (_cpos + 28 <= _cmap_len && total < _bgzf_run) {
            const unsigned char* h = _cmap + _cpos;
            if (!(h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4))) bgzf_fail("not a BGZF block header");
            const size_t xlen = h[10] | ((size_t)h[11] << 8);
            size_t bsize = 0;
            for (size_t x = 12; x + 4 <= 12 + xlen && _cpos + x + 4 <= _cmap_len;) {      // extra subfields: SI1 SI2 SLEN data
                const size_t slen = h[x + 2] | ((size_t)h[x + 3] << 8);
                if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen && _cpos + x + 6 <= _cmap_len) bsize = (h[x + 4] | ((size_t)h[x + 5] << 8)) + 1;
                x += 4 + slen;
            }
            if (bsize < 12 + xlen + 8 || _cpos + bsize > _cmap_len) bgzf_fail("truncated or malformed BGZF block");
            const unsigned char* t = h + bsize - 4;
            const uint32_t isize = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            if (isize > 65536) bgzf_fail("BGZF block claims more than 64 KiB");
            if (isize) blks.push_back(Blk{_cpos, _cpos + 12 + xlen, bsize - (12 + xlen) - 8, total, isize});
            total += isize;
            _cpos += bsize;
        }
        if (blks.empty()) { _eof = true; return false; }          // (what is unread stays where it is)
        // Two buffers take turns: the records a caller cut out of the previous one in place (SamReader::read_block, BAM) are
        // still being parsed while this one fills; the unread tail of the previous buffer (a record that straddles the two
        // runs of blocks) moves to the front of this one.
        const size_t keep = _end - _pos;
        std::vector<char>& nb = _bz[_bz_cur ^ 1];
        if (nb.size() < keep + total) nb.resize(keep + total);
        if (keep) std::memcpy(nb.data(), _src + _pos, keep);
        char* const dst = nb.data() + keep;
        int bad = 0;
        const int nt = (int)std::min<size_t>((size_t)_inflate_threads, blks.size());
#pragma omp parallel num_threads(nt) reduction(| : bad)
        {
            BlockInflater inf;                                 // (one decoder per thread and run of blocks)
#pragma omp for schedule(dynamic, 4)
            for (int64_t i = 0; i < (int64_t)blks.size(); ++i) {
                const Blk& b = blks[(size_t)i];
                const unsigned char* t = _cmap + b.data + b.clen;
                const uint32_t want = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                bad |= inf.run(_cmap + b.data, b.clen, dst + b.out, b.isize, want);
            }
        }
        if (bad) bgzf_fail(bad & 2 ? "CRC mismatch in a BGZF block" : "a BGZF block does not inflate");
        // compressed pages behind the read position are handed back (see consume())
        constexpr size_t kStep = (size_t)64 << 20;
        if (_cpos >= _creleased + kStep) {
            const size_t upto = _cpos & ~(size_t)4095;
            (void)::madvise((void*)(_cmap + _creleased), upto - _creleased, MADV_DONTNEED);
            _creleased = upto;
        }
        _bz_cur ^= 1;
        _src = nb.data(); _pos = 0; _end = keep + total;
        return true;
    }
    [[noreturn]] static void bgzf_fail(const char* what) { std::fprintf(stderr, "[Hypo::SeqIO] Error: %s\n", what); std::exit(1); }
    static constexpr size_t kBuf = 4u << 20;
    const unsigned char* _cmap = nullptr; size_t _cmap_len = 0, _cpos = 0, _creleased = 0;     // a mapped BGZF file and the next block
    int _inflate_threads = 8;
    size_t _bgzf_run = (size_t)32 << 20;               // inflated bytes per fill_bgzf call: a few blocks per inflating thread
    gzFile _fp = nullptr;
    std::vector<char> _buf;
    std::vector<char> _bz[2]; int _bz_cur = 0;          // BGZF: the inflated run of blocks in hand and the one before it
    const char* _map = nullptr; size_t _map_len = 0, _released = 0;
    const char* _src = nullptr;                        // the bytes being consumed: _buf or the mapping
    size_t _pos = 0, _end = 0;
    bool _eof = false;
};

struct FastaRecord { std::string name, seq; };

// name = first token of the header line (kseq: ks->name), multi-line sequences, FASTA or FASTQ
// A mapped plain FASTA file on all threads: the record starts (a '>' at the start of a line) are found chunk by chunk, then every
// record takes its name and its sequence lines out of the mapping by itself.  The serial reader below copied 3 GB on one thread for
// the 3 Gbp draft (0.7 of the 1.1 s of "Loaded Contigs").  Same records as the line-by-line reader: empty lines are skipped, a CR in
// front of a line feed is dropped, a sequence may be wrapped.  false: not a file this path takes (FASTQ, or no '>' first) — the
// caller reads it line by line.
inline bool read_fasta_mapped(const char* d, size_t n, std::vector<FastaRecord>& out) {
    size_t first = 0;
    while (first < n && (d[first] == '\n' || (d[first] == '\r' && first + 1 < n && d[first + 1] == '\n'))) ++first;
    if (first >= n || d[first] != '>') return false;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)omp_get_max_threads(), n / ((size_t)4 << 20) + 1));
    std::vector<std::vector<size_t>> found((size_t)T);
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; ++t) {
        const size_t a = std::max(first, n * (size_t)t / (size_t)T), b = n * ((size_t)t + 1) / (size_t)T;
        for (size_t p = a; p < b;) {
            const char* q = (const char*)std::memchr(d + p, '>', b - p);
            if (!q) break;
            const size_t at = (size_t)(q - d);
            if (at == first || d[at - 1] == '\n') found[(size_t)t].push_back(at);
            p = at + 1;
        }
    }
    std::vector<size_t> starts;
    for (auto& f : found) starts.insert(starts.end(), f.begin(), f.end());
    const size_t nrec = starts.size();
    starts.push_back(n);
    out.clear();
    out.resize(nrec);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t r = 0; r < (int64_t)nrec; ++r) {
        const size_t p = starts[(size_t)r], e = starts[(size_t)r + 1];
        const char* nl = (const char*)std::memchr(d + p, '\n', e - p);
        size_t he = nl ? (size_t)(nl - d) : e;                   // header: [p + 1, he)
        size_t body = nl ? he + 1 : e;
        if (he > p + 1 && d[he - 1] == '\r') --he;
        size_t ne = p + 1;
        while (ne < he && d[ne] != ' ' && d[ne] != '\t') ++ne;
        FastaRecord& rec = out[(size_t)r];
        rec.name.assign(d + p + 1, ne - (p + 1));
        // sequence lines
        const char* l2 = body < e ? (const char*)std::memchr(d + body, '\n', e - body) : nullptr;
        const size_t le = l2 ? (size_t)(l2 - d) : e;
        bool single = true;                                         // one sequence line (the usual case for assemblies): behind it only empty lines
        for (size_t q = le; q < e && single; ++q) single = d[q] == '\n' || d[q] == '\r';
        if (single) {
            size_t se = le;
            if (se > body && d[se - 1] == '\r') --se;
            rec.seq.assign(d + body, se - body);
        } else {
            rec.seq.reserve(e - body);
            for (size_t q = body; q < e;) {
                const char* x = (const char*)std::memchr(d + q, '\n', e - q);
                size_t qe = x ? (size_t)(x - d) : e;
                const size_t next = x ? qe + 1 : e;
                if (qe > q && d[qe - 1] == '\r') --qe;
                rec.seq.append(d + q, qe - q);
                q = next;
            }
        }
    }
    return true;
}

inline bool read_fastx(const std::string& path, std::vector<FastaRecord>& out, bool allow_mapped = true) {
    LineReader lr(path);
    if (!lr.ok()) return false;
    if (allow_mapped && lr.map_data() && read_fasta_mapped(lr.map_data(), lr.map_size(), out)) return true;
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
        // (a sequence written on one line — the usual case for assemblies — changes hands without a copy)
        while (have && !(line.size() && (line[0] == '>' || (fq && line[0] == '+')))) { if (r.seq.empty()) r.seq.swap(line); else r.seq += line; have = lr.next(line); }
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
    std::string rname_seen;                    // SamReader::parse: the last reference name this object saw and its tid (records of a
    int32_t tid_seen = -1;                     // sorted file repeat it millions of times: no string, no hash per record)
};

class SamReader {
public:
    explicit SamReader(const std::string& path) : _lr(path) {
        if (_lr.ok() && _lr.starts_with("BAM\1", 4)) { _bam = true; read_bam_header(); return; }
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
    void set_inflate_threads(int n) { _lr.set_inflate_threads(n); }
    bool bgzf() const { return _lr.bgzf(); }                   // a BGZF (BAM) file: its blocks are inflated by BlockInflater::name()
    const std::string& tid2name(int32_t tid) const { return _names[(size_t)tid]; }
    // A block of raw records in one buffer: record i = bytes [off[i], off[i + 1] - 1), followed by a NUL.  SAM: the text of
    // one alignment line; BAM: one alignment block without its 4-byte size.  I/O and inflate are serial (one reader thread),
    // parse() is const and re-entrant: Hypo::create_alignments parses the records of a block on all threads.
    // A plain SAM file is mapped (LineReader) and its records stay where they are: `base` points into the mapping and a record
    // ends with its line feed instead of a NUL (parse() never reads past the n bytes it is given).
    struct RecordBlock {
        std::vector<char> buf;
        std::vector<uint64_t> off;
        std::vector<uint32_t> lens;                  // BAM records cut out of the inflated buffer in place: their lengths (else from off)
        const char* base = nullptr;
        size_t n() const { return off.empty() ? 0 : off.size() - 1; }
        const char* rec(size_t i) const { return (base ? base : buf.data()) + off[i]; }
        size_t len(size_t i) const { return lens.empty() ? (size_t)(off[i + 1] - off[i] - 1) : (size_t)lens[i]; }
        void clear() { buf.clear(); off.clear(); lens.clear(); base = nullptr; }
    };
    // fills b with up to max_records records (or ~max_bytes); false once the end of the file has been reached
    bool read_block(RecordBlock& b, size_t max_records, size_t max_bytes = 32u << 20) {
        b.clear();
        b.off.push_back(0);
        if (!_bam && !_have_pending && _lr.mapped()) {                   // records in place: only the line ends are located here
            const char* const p0 = _lr.unread();
            const size_t avail = _lr.unread_bytes();
            size_t at = 0;
            b.base = p0;
            while (b.n() < max_records && at < max_bytes && at < avail) {
                const char* nl = (const char*)std::memchr(p0 + at, '\n', avail - at);
                const size_t stop = nl ? (size_t)(nl - p0) : avail;         // (a last line without a line feed ends at the end of the file)
                // empty line (also one that holds a carriage return only, as the gz / append_line path treats it): leading ones are
                // skipped, one inside ends the block
                if (stop == at || (stop == at + 1 && p0[at] == '\r')) { if (b.off.size() == 1) { at = stop + 1; b.off[0] = at; continue; } break; }
                b.off.push_back(stop + 1);
                at = stop + 1;
            }
            _lr.consume(at < avail ? at : avail);
            return at < avail;
        }
        if (_bam && _lr.bgzf()) {
            // BAM out of a mapped BGZF file: the records stay where the blocks were inflated to (LineReader keeps the buffer alive
            // while the next one fills); only their borders are located here — one hop over block_size per record, no copy (copying
            // 280 bytes per record on this one thread was what bound the 3 Gbp run: 57 of its 75 s).  A block of records ends where the
            // inflated buffer does; the record that straddles the end is completed by the next call.
            if (!_bam_ok) return false;
            for (;;) {
                const char* const p0 = _lr.buffered();
                const size_t avail = _lr.buffered_bytes();
                size_t at = 0;
                b.base = p0;
                while (b.n() < max_records && at + 4 <= avail) {
                    int32_t bs; std::memcpy(&bs, p0 + at, 4);
                    // (the hop is a chain of dependent loads over 100+ MB that has just been written by other cores: 60 ns per
                    // record as it stands — 3 s per 50 M records, the whole BAM reader's rate.  Records of a file are about the
                    // same size, so the size field sixteen records on is where this one's length says, give or take a line)
                    __builtin_prefetch(p0 + at + 16 * (4 + (size_t)bs));
                    __builtin_prefetch(p0 + at + 16 * (4 + (size_t)bs) + 64);
                    if (bs < 32) { std::fprintf(stderr, "[Hypo::SamReader] Error: malformed BAM record\n"); std::exit(1); }
                    if (at + 4 + (size_t)bs > avail) break;
                    b.off.back() = at + 4;
                    b.lens.push_back((uint32_t)bs);
                    b.off.push_back(at + 4 + (size_t)bs);
                    at += 4 + (size_t)bs;
                }
                _lr.skip(at);
                if (b.n() > 0) return true;                          // (more may follow: the next call finds out)
                if (!_lr.extend()) {
                    if (_lr.buffered_bytes()) { std::fprintf(stderr, "[Hypo::SamReader] Error: truncated BAM record\n"); std::exit(1); }
                    return false;
                }
            }
        }
        while (b.n() < max_records && b.buf.size() < max_bytes) {
            const size_t start = b.buf.size();
            if (_bam) {
                int32_t bs = 0;
                if (!_bam_ok || !_lr.read_bytes(&bs, 4) || bs < 32) return false;
                b.buf.resize(start + (size_t)bs);
                if (!_lr.read_bytes(b.buf.data() + start, (size_t)bs)) { std::fprintf(stderr, "[Hypo::SamReader] Error: truncated BAM record\n"); std::exit(1); }
            } else if (_have_pending) {
                b.buf.insert(b.buf.end(), _pending.begin(), _pending.end());
                _have_pending = false;
            } else if (!_lr.append_line(b.buf)) {
                return false;
            }
            if (b.buf.size() == start) continue;                       // empty line
            b.buf.push_back('\0');
            b.off.push_back(b.buf.size());
        }
        return true;
    }
    // one raw record (NUL-terminated, n bytes) -> fields
    void parse(const char* line, size_t n, SamRecord& r) const {
        if (_bam) { parse_bam(line, n, r); return; }
        if (n && line[n - 1] == '\r') --n;                               // (a record cut out of a mapped file keeps its CR)
        size_t f[12]; int nf = 0; f[0] = 0;
        for (size_t i = 0; i < n && nf < 11; ++i) if (line[i] == '\t') f[++nf] = i + 1;
        if (nf < 10) { std::fprintf(stderr, "[Hypo::SamReader] Error: malformed SAM record: %.*s\n", (int)(n < 60 ? n : 60), line); std::exit(1); }
        auto fbeg = [&](int k) { return line + f[k]; };
        auto flen = [&](int k) { return (k < nf ? f[k + 1] - 1 : n) - f[k]; };
        r.qname.assign(fbeg(0), flen(0));
        r.flag = (uint32_t)std::strtoul(fbeg(1), nullptr, 10);
        if (!r.rname_seen.empty() && r.rname_seen.size() == flen(2) && std::memcmp(r.rname_seen.data(), fbeg(2), flen(2)) == 0) r.tid = r.tid_seen;
        else {
            r.rname_seen.assign(fbeg(2), flen(2));
            auto it = _tid.find(r.rname_seen);
            r.tid = r.tid_seen = it == _tid.end() ? -1 : it->second;
        }
        r.pos = (uint32_t)(std::strtoul(fbeg(3), nullptr, 10) - 1);
        r.mapq = (uint32_t)std::strtoul(fbeg(4), nullptr, 10);
        r.cigar.clear();
        const char* c = fbeg(5);
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
        r.seq.assign(fbeg(9), flen(9));
        r.has_nm = false;
        if (nf >= 11) {
            const char* p = (const char*)memmem(line + f[10] - 1, n - (f[10] - 1), "\tNM:i:", 6);
            if (p) { r.has_nm = true; r.nm = std::strtoll(p + 6, nullptr, 10); }
        }
    }
    // BAM only: the fixed fields of a raw record and where its CIGAR and 4-bit bases lie (Hypo::parse_block packs the bases straight
    // into 2 bits: no SamRecord, no ASCII detour).  false: a long-CIGAR placeholder (the real operations are in the CG tag) — the
    // caller takes parse() for that record.
    bool is_bam() const { return _bam; }
    struct BamCore { int32_t tid; uint32_t pos, mapq, flag, n_cigar, l_seq; const char* qname; uint32_t l_qname; const char* cigar; const uint8_t* seq4; };      // l_qname: bytes of the name without its terminator
    bool bam_core(const char* p, size_t n, BamCore& c) const {
        const int32_t ref = le32(p), l_seq = le32(p + 16);
        const unsigned l_name = (unsigned char)p[8];
        c.mapq = (unsigned char)p[9]; c.n_cigar = le16(p + 12); c.flag = le16(p + 14);
        if (l_seq < 0 || 32 + (size_t)l_name + 4ull * c.n_cigar + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > n) {
            std::fprintf(stderr, "[Hypo::SamReader] Error: malformed BAM record\n"); std::exit(1);
        }
        c.tid = (ref >= 0 && (size_t)ref < _names.size()) ? ref : -1;
        c.pos = (uint32_t)le32(p + 4); c.l_seq = (uint32_t)l_seq;
        c.qname = p + 32; c.l_qname = (uint32_t)::strnlen(p + 32, l_name); c.cigar = p + 32 + l_name; c.seq4 = (const uint8_t*)(c.cigar + 4ull * c.n_cigar);
        if (c.n_cigar == 2) {
            const uint32_t c0 = (uint32_t)le32(c.cigar), c1 = (uint32_t)le32(c.cigar + 4);
            if ((c0 & 0xf) == 4 && (c0 >> 4) == c.l_seq && (c1 & 0xf) == 3) return false;
        }
        return true;
    }
    // BAM only: the NM:i tag of a raw record whose fixed part bam_core has read (the long-read filter of Alignment.cpp:51-58 wants
    // it); false when the record has none
    bool bam_nm(const char* p, size_t n, const BamCore& c, int64_t& nm) const {
        size_t o = (size_t)((const char*)c.seq4 - p) + ((size_t)c.l_seq + 1) / 2 + (size_t)c.l_seq;
        while (o + 3 <= n) {
            const char t0 = p[o], t1 = p[o + 1], ty = p[o + 2];
            o += 3;
            int64_t iv = 0; bool is_int = true; size_t adv = 0;
            switch (ty) {
                case 'c': iv = (int8_t)p[o]; adv = 1; break;
                case 'C': iv = (uint8_t)p[o]; adv = 1; break;
                case 's': iv = (int16_t)le16(p + o); adv = 2; break;
                case 'S': iv = le16(p + o); adv = 2; break;
                case 'i': iv = le32(p + o); adv = 4; break;
                case 'I': iv = (uint32_t)le32(p + o); adv = 4; break;
                case 'A': is_int = false; adv = 1; break;
                case 'f': is_int = false; adv = 4; break;
                case 'Z': case 'H': is_int = false; adv = ::strnlen(p + o, n - o) + 1; break;
                case 'B': {
                    is_int = false;
                    if (o + 5 > n) return false;
                    const char st = p[o]; const uint32_t cnt = (uint32_t)le32(p + o + 1);
                    const size_t es = (st == 'c' || st == 'C') ? 1 : ((st == 's' || st == 'S') ? 2 : 4);
                    adv = 5 + es * cnt; break;
                }
                default: return false;                                // unknown type: stop scanning (as parse_bam does)
            }
            if (o + adv > n) return false;
            if (t0 == 'N' && t1 == 'M' && is_int) { nm = iv; return true; }
            o += adv;
        }
        return false;
    }
    // read name of a raw record (for messages)
    std::string record_name(const char* line, size_t n) const {
        if (_bam) return n > 32 ? std::string(line + 32, ::strnlen(line + 32, std::min<size_t>(n - 32, (unsigned char)line[8]))) : std::string("?");
        const char* t = (const char*)std::memchr(line, '\t', n);
        return std::string(line, t ? (size_t)(t - line) : n);
    }
private:
    static int32_t le32(const char* p) { int32_t v; std::memcpy(&v, p, 4); return v; }
    static uint16_t le16(const char* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }
    void read_bam_header() {                          // SAM spec 4.2: magic, l_text, text, n_ref, (l_name, name, l_ref)*
        char magic[4]; int32_t l_text = 0, n_ref = 0;
        _bam_ok = _lr.read_bytes(magic, 4) && _lr.read_bytes(&l_text, 4) && l_text >= 0;
        if (_bam_ok) { std::string text((size_t)l_text, '\0'); _bam_ok = l_text == 0 || _lr.read_bytes(&text[0], (size_t)l_text); }
        _bam_ok = _bam_ok && _lr.read_bytes(&n_ref, 4) && n_ref >= 0;
        for (int32_t i = 0; _bam_ok && i < n_ref; ++i) {
            int32_t l_name = 0, l_ref = 0;
            _bam_ok = _lr.read_bytes(&l_name, 4) && l_name > 0;
            if (!_bam_ok) break;
            std::string name((size_t)l_name, '\0');
            _bam_ok = _lr.read_bytes(&name[0], (size_t)l_name) && _lr.read_bytes(&l_ref, 4);
            name.resize(::strnlen(name.data(), name.size()));
            _names.push_back(name);
            _tid[name] = (int32_t)_names.size() - 1;
        }
        if (!_bam_ok) { std::fprintf(stderr, "[Hypo::SamReader] Error: malformed BAM header\n"); std::exit(1); }
    }
    void parse_bam(const char* p, size_t n, SamRecord& r) const {   // SAM spec 4.2.1; fields after block_size
        const int32_t ref = le32(p), pos = le32(p + 4), l_seq = le32(p + 16);
        const unsigned l_name = (unsigned char)p[8], mapq = (unsigned char)p[9];
        const unsigned n_cig = le16(p + 12), flag = le16(p + 14);
        size_t o = 32;
        if (l_seq < 0 || o + l_name + 4ull * n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > n) {
            std::fprintf(stderr, "[Hypo::SamReader] Error: malformed BAM record\n"); std::exit(1);
        }
        r.qname.assign(p + o, l_name ? l_name - 1 : 0); o += l_name;
        r.flag = flag; r.mapq = mapq;
        r.tid = (ref >= 0 && (size_t)ref < _names.size()) ? ref : -1;
        r.pos = (uint32_t)pos;                                       // 0-based in BAM
        r.cigar.resize(n_cig);
        for (unsigned i = 0; i < n_cig; ++i) { const uint32_t c = (uint32_t)le32(p + o + 4 * i); r.cigar[i] = (c & 0xf) | ((c >> 4) << 4); }
        o += 4ull * n_cig;
        static const char kBase[] = "=ACMGRSVTWYHKDBN";
        r.seq.resize((size_t)l_seq);
        for (int32_t i = 0; i < l_seq; ++i) { const unsigned char v = (unsigned char)p[o + (size_t)i / 2]; r.seq[(size_t)i] = kBase[(i & 1) ? (v & 15) : (v >> 4)]; }
        o += ((size_t)l_seq + 1) / 2 + (size_t)l_seq;                // seq, qual
        r.has_nm = false;
        // Long-CIGAR convention (SAM spec 4.2.2): a record with more than 65535 operations stores `<l_seq>S<ref_len>N` in the
        // CIGAR field and the real operations in the CG:B,I tag; htslib restores them on read (bam_tag2cigar), so the
        // reference sees the real CIGAR.
        const bool cg_placeholder = n_cig == 2 && (r.cigar[0] & 0xf) == 4 && (r.cigar[0] >> 4) == (uint32_t)l_seq && (r.cigar[1] & 0xf) == 3;
        bool cg_found = false;
        while (o + 3 <= n) {                                          // optional fields: tag[2] type value
            const char t0 = p[o], t1 = p[o + 1], ty = p[o + 2];
            o += 3;
            int64_t iv = 0; bool is_int = true; size_t adv = 0;
            switch (ty) {
                case 'c': iv = (int8_t)p[o]; adv = 1; break;
                case 'C': iv = (uint8_t)p[o]; adv = 1; break;
                case 's': iv = (int16_t)le16(p + o); adv = 2; break;
                case 'S': iv = le16(p + o); adv = 2; break;
                case 'i': iv = le32(p + o); adv = 4; break;
                case 'I': iv = (uint32_t)le32(p + o); adv = 4; break;
                case 'A': is_int = false; adv = 1; break;
                case 'f': is_int = false; adv = 4; break;
                case 'Z': case 'H': is_int = false; adv = ::strnlen(p + o, n - o) + 1; break;      // (records are cut in place from the inflated block: no terminator behind them)
                case 'B': {
                    is_int = false;
                    const char st = p[o]; const uint32_t cnt = (uint32_t)le32(p + o + 1);
                    const size_t es = (st == 'c' || st == 'C') ? 1 : ((st == 's' || st == 'S') ? 2 : 4);
                    adv = 5 + es * cnt; break;
                }
                default: adv = n; break;                              // unknown type: stop scanning
            }
            if (o + adv > n) break;
            if (t0 == 'N' && t1 == 'M' && is_int) { r.has_nm = true; r.nm = iv; if (!cg_placeholder || cg_found) return; }
            if (cg_placeholder && t0 == 'C' && t1 == 'G' && ty == 'B' && p[o] == 'I') {
                const uint32_t cnt = (uint32_t)le32(p + o + 1);
                r.cigar.resize(cnt);
                for (uint32_t i = 0; i < cnt; ++i) { const uint32_t c = (uint32_t)le32(p + o + 5 + 4ull * i); r.cigar[i] = (c & 0xf) | ((c >> 4) << 4); }
                cg_found = true;
                if (r.has_nm) return;
            }
            o += adv;
        }
        if (cg_placeholder && !cg_found) {
            std::fprintf(stderr, "[Hypo::SamReader] Error: BAM record %s has a long-CIGAR placeholder but no CG:B,I tag\n", r.qname.c_str());
            std::exit(1);
        }
    }
    bool _bam = false, _bam_ok = true;
    LineReader _lr;
    std::vector<std::string> _names;
    std::unordered_map<std::string, int32_t> _tid;
    std::string _pending;
    bool _have_pending = false;
};

}  // namespace hypo
