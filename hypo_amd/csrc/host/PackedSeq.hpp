// PackedSeq.hpp — host mirror of hypo::PackedSeq<NB> (reference: include/PackedSeq.hpp:80-160,
// src/PackedSeq.cpp:58-262).  Same public surface and the same byte layout (MSB-first, NB bits per base,
// last byte zero padded), written from scratch around one std::vector<uint8_t>; the bytes are what
// HypoWindowBatch.draft4 / arms2 carry to the device unchanged.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace hypo {

// cNt4Table (include/globalDefs.hpp:160-178): A/a 0, C/c 1, G/g 2, T/t/U/u 3, bytes 0..3 themselves, else 4
inline uint8_t nt4(unsigned char c) {
    struct Lut { uint8_t v[256]; Lut() {
        for (int i = 0; i < 256; ++i) v[i] = 4;
        v['A'] = v['a'] = v[0] = 0; v['C'] = v['c'] = v[1] = 1; v['G'] = v['g'] = v[2] = 2; v['T'] = v['t'] = v['U'] = v['u'] = v[3] = 3; } };
    static const Lut lut;                          // (a table, not a switch: a random genome mispredicts every other base of one)
    return lut.v[c];
}

// n characters of A, C, G, T (either case) -> PackedSeq<2> bytes at dst ((n + 3) / 4 of them, the last one zero padded);
// false when another character is among them (dst is then garbage)
inline bool pack2_acgt(const char* s, size_t n, uint8_t* dst) {
    struct Lut { uint8_t v[256]; Lut() { for (int i = 0; i < 256; ++i) v[i] = 0x80; v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; } };
    static const Lut lut;
    const unsigned char* u = (const unsigned char*)s;
    uint8_t* d = dst;
    unsigned bad = 0;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const unsigned a = lut.v[u[i]], b = lut.v[u[i + 1]], c = lut.v[u[i + 2]], e = lut.v[u[i + 3]];
        bad |= a | b | c | e;
        *d++ = (uint8_t)((a << 6) | (b << 4) | (c << 2) | e);
    }
    if (i < n) {
        unsigned byte = 0;
        for (int sh = 6; i < n; ++i, sh -= 2) { const unsigned a = lut.v[u[i]]; bad |= a; byte |= (a & 3u) << sh; }
        *d = (uint8_t)byte;
    }
    return !(bad & 0x80u);
}

// the same from BAM's 4-bit codes (=ACMGRSVTWYHKDBN: A 1, C 2, G 4, T 8; two per byte, first base in the high nibble): bases
// [first, first + n) of seq4 -> PackedSeq<2> bytes at dst; false when a code other than A, C, G, T is among them
inline bool pack2_from_bam4(const uint8_t* seq4, size_t first, size_t n, uint8_t* dst) {
    struct Lut { uint16_t v[256]; Lut() {
        auto nib = [](unsigned c) -> unsigned { return c == 1 ? 0u : c == 2 ? 1u : c == 4 ? 2u : c == 8 ? 3u : 0x100u; };
        for (unsigned b = 0; b < 256; ++b) { const unsigned h = nib(b >> 4), l = nib(b & 15); v[b] = (uint16_t)((((h & 3u) << 2) | (l & 3u)) | ((h | l) & 0x100u)); } } };
    static const Lut lut;
    auto code = [&](size_t i) -> unsigned { const unsigned c = (seq4[i >> 1] >> ((~i & 1) << 2)) & 15u; return c == 1 ? 0u : c == 2 ? 1u : c == 4 ? 2u : c == 8 ? 3u : 0x100u; };
    unsigned bad = 0;
    size_t i = 0;
    uint8_t* d = dst;
    if ((first & 1) == 0) {
        const uint8_t* s = seq4 + (first >> 1);
        for (; i + 4 <= n; i += 4, s += 2) {
            const unsigned a = lut.v[s[0]], b = lut.v[s[1]];
            bad |= a | b;
            *d++ = (uint8_t)(((a & 15u) << 4) | (b & 15u));
        }
    } else {
        for (; i + 4 <= n; i += 4) {
            const unsigned a = code(first + i), b = code(first + i + 1), c = code(first + i + 2), e = code(first + i + 3);
            bad |= a | b | c | e;
            *d++ = (uint8_t)(((a & 3u) << 6) | ((b & 3u) << 4) | ((c & 3u) << 2) | (e & 3u));
        }
    }
    if (i < n) {
        unsigned byte = 0;
        for (int sh = 6; i < n; ++i, sh -= 2) { const unsigned a = code(first + i); bad |= a; byte |= (a & 3u) << sh; }
        *d = (uint8_t)byte;
    }
    return !(bad & 0x100u);
}

template <int NB>
class PackedSeq {
    static_assert(NB == 2 || NB == 4, "2 or 4 bits per base");
    static constexpr int PER_BYTE = 8 / NB;
    static constexpr unsigned MASK = (1u << NB) - 1u;

public:
    PackedSeq() = default;
    explicit PackedSeq(const std::string& text) { assign(text.data(), text.size()); }
    // [left, right) of another packed sequence of the same width (PackedSeq.cpp:155-194)
    PackedSeq(const PackedSeq& ps, size_t left, size_t right) {
        resize(right - left);
        for (size_t i = left; i < right; ++i) put(i - left, ps.enc_base_at(i));
    }
    // 2-bit copy of [left, right) of a 4-bit sequence; non-ACGT is fatal like the reference (PackedSeq.cpp:197-227)
    template <int MB>
    PackedSeq(const PackedSeq<MB>& ps, size_t left, size_t right) {
        static_assert(NB == 2 && MB == 4, "only 4 -> 2 bit conversion exists");
        resize(right - left);
        for (size_t i = left; i < right; ++i) {
            const uint8_t b = ps.enc_base_at(i);
            if (b > 3) { std::fprintf(stderr, "[Hypo::PackedSeq] Error: Wrong base (Can not pack in 2 bits): Base code at %zu in a sequence is not A, C, G, or T !\n", i); std::exit(1); }
            put(i - left, b);
        }
    }

    // adopts n bases already packed in this layout (an arm cut on the device: hypo_gpu_arms_download)
    PackedSeq(const uint8_t* bytes, size_t n) : _data(bytes, bytes + (n + PER_BYTE - 1) / PER_BYTE), _len(n) {}

    // n characters of A, C, G, T (either case) packed four per byte in one pass; false — and nothing kept — when another
    // character is among them (what Alignment::copy_data needs for every record of the alignment file: the per-base put()
    // of assign() and a separate validation pass were 2/3 of the time the C3 run spent loading alignments)
    bool assign_acgt(const char* s, size_t n) {
        static_assert(NB == 2 || NB == 4, "");
        if (NB != 2) return false;
        struct Lut { uint8_t v[256]; Lut() { for (int i = 0; i < 256; ++i) v[i] = 0x80; v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; } };
        static const Lut lut;
        const unsigned char* u = (const unsigned char*)s;
        _data.resize((n + 3) / 4);
        uint8_t* d = _data.data();
        unsigned bad = 0;
        size_t i = 0;
        for (; i + 4 <= n; i += 4) {
            const unsigned a = lut.v[u[i]], b = lut.v[u[i + 1]], c = lut.v[u[i + 2]], e = lut.v[u[i + 3]];
            bad |= a | b | c | e;
            *d++ = (uint8_t)((a << 6) | (b << 4) | (c << 2) | e);
        }
        if (i < n) {
            unsigned byte = 0;
            for (int sh = 6; i < n; ++i, sh -= 2) { const unsigned a = lut.v[u[i]]; bad |= a; byte |= (a & 3u) << sh; }
            *d = (uint8_t)byte;
        }
        if (bad & 0x80u) { _data.clear(); _len = 0; return false; }
        _len = n;
        return true;
    }

    bool is_valid() const { return _valid; }
    size_t get_seq_size() const { return _len; }
    uint8_t enc_base_at(size_t i) const { return (uint8_t)((_data[i / PER_BYTE] >> (8 - NB - NB * (int)(i % PER_BYTE))) & MASK); }
    char base_at(size_t i) const { const uint8_t c = enc_base_at(i); return "ACGTN"[c < 4 ? c : 4]; }
    std::string unpack() const { return unpack(0, _len); }
    std::string unpack(size_t left, size_t right) const {
        std::string s; s.reserve(right - left);
        for (size_t i = left; i < right; ++i) s.push_back(base_at(i));
        return s;
    }
    // k-mer (2 bits per base, first base most significant) ending inside [ind, ind + k): true iff the k bases starting
    // at ind spell target (src/PackedSeq.cpp:264-289)
    bool check_kmer(uint64_t target, unsigned k, size_t ind) const {
        uint64_t kmer = 0; unsigned len = 0;
        const uint64_t mask = (1ULL << (2 * k)) - 1;
        for (size_t i = ind; i < ind + k; ++i) {
            const uint8_t b = enc_base_at(i);
            if (b < 4) { kmer = ((kmer << 2) | b) & mask; if (len < k) ++len; } else { len = 0; kmer = 0; }
            if (len == k && kmer == target) return true;
        }
        return false;
    }
    // first (is_first) or last start position of target among the k-mers lying inside [left, right)
    // (src/PackedSeq.cpp:291-320)
    bool find_kmer(uint64_t target, unsigned k, size_t left, size_t right, bool is_first, size_t& result) const {
        if (left == right) return false;
        uint64_t kmer = 0; unsigned len = 0;
        const uint64_t mask = (1ULL << (2 * k)) - 1;
        bool found = false;
        for (size_t i = left; i < right; ++i) {
            const uint8_t b = enc_base_at(i);
            if (b < 4) { kmer = ((kmer << 2) | b) & mask; if (len < k) ++len; } else { len = 0; kmer = 0; }
            if (len == k && kmer == target) { result = i - k + 1; found = true; if (is_first) break; }
        }
        return found;
    }
    // raw PackedSeq bytes for the C-ABI batch
    const uint8_t* data() const { return _data.data(); }
    size_t byte_size() const { return _data.size(); }

private:
    void resize(size_t n) { _len = n; _data.assign((n + PER_BYTE - 1) / PER_BYTE, 0); }
    void put(size_t i, uint8_t code) { _data[i / PER_BYTE] |= (uint8_t)(code << (8 - NB - NB * (int)(i % PER_BYTE))); }
    void assign(const char* s, size_t n) {
        if (n > 0xffffffffu) { std::fprintf(stderr, "[Hypo::PackedSeq] Error: Length exceed limit: The length of a sequence is %zu which exceeds the limit of %u !\n", n, 0xffffffffu); std::exit(1); }
        resize(n);
        if (NB == 4 && n >= (1u << 20)) {            // a contig: whole bytes on all threads (4-bit packing cannot fail)
            const int64_t nb = (int64_t)(n / 2);
#pragma omp parallel for schedule(static)
            for (int64_t j = 0; j < nb; ++j) _data[(size_t)j] = (uint8_t)((nt4((unsigned char)s[2 * j]) << 4) | nt4((unsigned char)s[2 * j + 1]));
            if (n & 1) _data[n / 2] = (uint8_t)(nt4((unsigned char)s[n - 1]) << 4);
            return;
        }
        for (size_t i = 0; i < n; ++i) {
            const uint8_t b = nt4((unsigned char)s[i]);
            if (NB == 2 && b > 3) {       // PackedSeq.cpp:75-79: marks the sequence invalid and stops packing
                std::fprintf(stderr, "[Hypo::PackedSeq] Error: Wrong base (Can not pack in 2 bits): Base %c in a sequence is not A, C, G, or T !\n", s[i]);
                _valid = false;
                break;
            }
            put(i, b);
        }
    }
    std::vector<uint8_t> _data;
    size_t _len = 0;
    bool _valid = true;
};

}  // namespace hypo
