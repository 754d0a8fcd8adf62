// ref_scan_harness.cpp — C entry point around the REAL hypo::Contig::find_solid_pos (src/Contig.cpp:40-74) and the
// real suk::SolidKmers bit set (external/suk/src/SolidKmers.cpp:47-62), compiled from the sources where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libhyporef_scan.so.  TEST INFRASTRUCTURE ONLY: it pins
// oracle_solid_scan (tests/test_oracle_vs_ref.py) and generates tests/golden/scan_cases.json.gz
// (tests/golden/make_scan_golden.py).  Nothing here restates the scan.
//
// How it links without the reference's build system: src/Contig.cpp and src/PackedSeq.cpp, suk's SolidKmers.cpp and the
// plain sources of sdsl-lite's lib/ that the bit vector and its rank/select supports need (bits, memory_management,
// ram_fs, util, io, sfstream) are compiled as they are with -ffunction-sections and hidden visibility; the link keeps
// only what hyporef_solid_scan() reaches (-Wl,--gc-sections), so the parts of those translation units that would need
// KMC, htslib, spoa or sdsl's cmake-generated structure_tree.cpp (SolidKmers::initialise, the other Contig methods,
// serialize()) are dropped instead of being stubbed.
//
// Private members (Contig::_solid_pos, _kmerinfo, _Rsolid_pos) are read through explicit template instantiation, which
// the language exempts from access checks; the reference headers are included unmodified.
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include "Contig.hpp"

namespace {
template <class Tag, typename Tag::type M> struct Rob { friend typename Tag::type get(Tag) { return M; } };
struct SolidPosTag { typedef sdsl::bit_vector hypo::Contig::*type; friend type get(SolidPosTag); };
struct KmerInfoTag { typedef std::vector<std::unique_ptr<hypo::KmerInfo>> hypo::Contig::*type; friend type get(KmerInfoTag); };
struct RankTag { typedef sdsl::bit_vector::rank_1_type hypo::Contig::*type; friend type get(RankTag); };
template struct Rob<SolidPosTag, &hypo::Contig::_solid_pos>;
template struct Rob<KmerInfoTag, &hypo::Contig::_kmerinfo>;
template struct Rob<RankTag, &hypo::Contig::_Rsolid_pos>;
}  // namespace

extern "C" __attribute__((visibility("default")))
// contig: n ASCII bases (ACGT, anything else is N: PackedSeq.cpp:44-48).  bvsd_path: sdsl bit_vector file of 4^k bits
// (uint64 bit count + little-endian 64-bit words), loaded by the real SolidKmers::load.  words_out: ceil(n/64) words of
// Contig::_solid_pos; kids_out: KmerInfo::kid of every marked position in order (up to kids_cap); rank_out (optional):
// _Rsolid_pos(64*w) for w = 0..ceil(n/64) (rank_out[last] = rank(n)).  Returns 0, or -1 when the bit set cannot be loaded.
int hyporef_solid_scan(const char* contig, uint64_t n, uint32_t k, const char* bvsd_path,
                       uint64_t* words_out, uint64_t* kids_out, uint64_t kids_cap, uint64_t* rank_out, uint64_t* n_solid) {
    auto sk = std::make_unique<suk::SolidKmers>(k);
    if (!sk->load(std::string(bvsd_path))) return -1;
    hypo::Contig c(0, "c", std::string(contig, (size_t)n));
    c.find_solid_pos(sk);
    const sdsl::bit_vector& sp = c.*get(SolidPosTag());
    const auto& ki = c.*get(KmerInfoTag());
    const auto& rk = c.*get(RankTag());
    const uint64_t nw = (n + 63) / 64;
    std::memset(words_out, 0, nw * 8);
    for (uint64_t i = 0; i < n; ++i) if (sp[i]) words_out[i >> 6] |= 1ULL << (i & 63);
    for (uint64_t i = 0; i < ki.size() && i < kids_cap; ++i) kids_out[i] = ki[i]->kid;
    if (rank_out) for (uint64_t w = 0; w <= nw; ++w) rank_out[w] = rk(w * 64 < n ? w * 64 : n);
    *n_solid = ki.size();
    return 0;
}
