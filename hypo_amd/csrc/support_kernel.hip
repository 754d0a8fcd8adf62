// support_kernel.hip — the support votes of the mapped short reads on the device.  Replaces
//   Alignment::update_solidkmers_support   src/Alignment.cpp:65-132    support_kmers_kernel
//   Alignment::update_minimisers_support   src/Alignment.cpp:134-220   support_minimizers_kernel
// (the reference takes a mutex per solid k-mer / per mega-window, include/Contig.hpp:207-214; here the counters are device
// atomics).  Both are per-read walks over integer data with data-dependent trip counts: one lane per read, HBM / latency
// bound, no reuse to stage through LDS; the reads are the flat arrays hypo_gpu_reads_upload left on the device, the same copy
// hypo_gpu_arms_build cuts into arms afterwards.
//
// What must match the reference exactly is which (k-mer, read) pairs vote — the counters decide where the strong regions and
// the window borders are.  The order of the votes of ONE read matters for the k-mer support (the pvs_supp_* state below), the
// order between reads does not (increments commute).
#include <hip/hip_runtime.h>
#include "support_kernel.hpp"

namespace hypo {

namespace {
constexpr int T = 256;
__device__ __forceinline__ uint32_t base2(const uint8_t* p, uint32_t i) { return (p[i >> 2] >> (6 - 2 * (i & 3))) & 3u; }
// number of entries < v / <= v in the sorted array a[0, n)
__device__ __forceinline__ uint32_t count_less(const uint32_t* a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo;
}
__device__ __forceinline__ uint32_t count_less_equal(const uint32_t* a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (a[m] <= v) lo = m + 1; else hi = m; }
    return lo;
}
}  // namespace

// ---- Alignment::update_solidkmers_support ----------------------------------------------------------------------------------
// spos: positions of the solid k-mers of the coordinate space, ascending (Contig::_solid_pos); kids: their k-mers (_kmerinfo[i]->
// kid).  The reference looks every k-mer of the read up in an unordered_multimap of the span's solid k-mers and walks equal_range
// in reverse insertion order; the matches that can vote lie within k bases of the read k-mer's offset, so a window over the
// position-sorted solid k-mers that slides along with the read gives the same visits in the same order (host/Alignment.cpp).
// Round 4 (end): the kernel was the largest of an end-to-end run at k = 17 (a random genome marks 40 % of its positions: 60 solid
// k-mers inside a 150-base read, ~14 of them within k of every read k-mer; 34 ms per 10 M reads, 2 s of the 27 s of the 3 Gbp set),
// and what it waited for was one chain of ~2 000 dependent loads per lane from L2.  The reads of a block are neighbours (the file is
// sorted), so the solid k-mers they can see are one short stretch of the arrays: the block copies that stretch (positions, ids) into
// LDS once and every lane's walk reads LDS; coverage becomes a +1 / -1 pair per read on a difference array in LDS, support an LDS
// counter, and one global atomic per touched k-mer and block goes out at the end (120 global atomics per read before).  The read's
// bases come in 16 at a time.  A block whose stretch does not fit (sparse reads) takes the same walk over global memory.  Which
// (k-mer, read) pairs vote, and in which order within a read, is unchanged.
namespace {
constexpr uint32_t KCAP = 2048;                           // solid k-mers of a block's tile held in LDS
constexpr uint32_t kLongReadSpan = 1000;                  // mean reference span above which a block takes 64 reads instead of 256
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
    typedef uint32_t __attribute__((aligned(1))) u32u;
    return *(const u32u*)p;
}
// A read's walk (src/Alignment.cpp:78-132) as a RESUMABLE state: the block keeps a tile of KCAP solid k-mers in LDS, every lane
// walks its read as far as the tile reaches and pauses; the next tile starts at the first entry an unfinished lane still needs.
// Reads of 150 bases see their whole stretch in one tile (round 4's kernel); reads of 15 kbp (BASELINE config C5: HiFi reads passed
// as the short reads) walk through a few dozen tiles instead of chasing ~30 dependent loads per position through L2 — 1 111 ms of
// the 1 570 ms of device time of the 250 Mbp C5 slice before (profiles/r05_c5_slice_before.txt).  Which (k-mer, read) pairs vote,
// and in which order within a read, is unchanged: the state a lane carries from tile to tile is the serial loop's own.
struct WalkState {
    uint64_t kmer = 0; uint32_t kmer_len = 0;
    int64_t pvs_supp_kpos = -1; uint32_t pvs_supp_r_bind = 0;
    uint32_t lo = 0, hi = 0;                                  // solid k-mers [lo, hi) of the read's [0, n): offset within k of the read k-mer's
    uint32_t r_ind = 0;                                       // next base of the read
    uint32_t word = 0;                                        // 16 bases of the read: byte j of the word holds bases 4j .. 4j + 3, first base in the top bits
};
// sp / kd: entry c of the read's stretch is sp[c] / kd[c] for c in [lo, tile_end) (pointers offset by the caller; entries before
// the tile are never touched: lo only grows and the tile starts at or before it); tile_end: entries [.., tile_end) are in the
// tile, n: entries of the read.  tags: low byte of every k-mer id of the tile, four to a word, entry c of the read = tags entry
// toff + c: the ~14 candidates of a read k-mer are screened four at a time on that byte (no false negatives; a hit is verified
// on the whole id), from the last one down, as the plain loop visits them.  Returns true when the read is finished.
template <typename KidT, typename Vote>
__device__ __forceinline__ bool walk_read(WalkState& S, const uint32_t* sp, const KidT* kd, const uint32_t* tags, int32_t toff, uint32_t n, uint32_t tile_end,
                                          uint32_t rb, uint32_t re, const uint8_t* rd, uint32_t nq, uint32_t k, Vote vote) {
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t num_cbases = re - rb;
    uint64_t kmer = S.kmer; uint32_t kmer_len = S.kmer_len;
    int64_t pvs_supp_kpos = S.pvs_supp_kpos; uint32_t pvs_supp_r_bind = S.pvs_supp_r_bind;
    uint32_t lo = S.lo, hi = S.hi, word = S.word, r_ind = S.r_ind;
    bool done = true;
    for (; r_ind < nq; ++r_ind) {
        if (r_ind + 1 >= k) {
            // the window of this position first: when it reaches the end of the tile with entries of the read left, the position
            // waits for the next tile (nothing of it has happened yet)
            const uint32_t r_bind = r_ind + 1 - k;
            while (hi < tile_end && (int64_t)sp[hi] - (int64_t)rb <= (int64_t)r_bind + (int64_t)k) ++hi;
            if (hi == tile_end && hi < n) { done = false; break; }
        }
        if ((r_ind & 15u) == 0 || r_ind == S.r_ind) word = load_u32_unaligned(rd + ((r_ind >> 4) << 2));
        const uint32_t b = (word >> (8 * ((r_ind >> 2) & 3u) + 6 - 2 * (r_ind & 3u))) & 3u;
        kmer = ((kmer << 2) | b) & kmask;
        if (kmer_len < k) ++kmer_len;
        if (kmer_len != k) continue;
        const uint32_t r_bind = r_ind + 1 - k;
        while (lo < hi && (int64_t)sp[lo] - (int64_t)rb + (int64_t)k < (int64_t)r_bind) ++lo;
        auto candidate = [&](uint32_t c) {
            if ((uint64_t)kd[c] != kmer) return;
            const int64_t c_dist = (int64_t)sp[c] - (int64_t)rb;
            const uint32_t left = c_dist > (int64_t)k ? (uint32_t)(c_dist - k) : 0u;
            const int64_t rr = c_dist + (int64_t)k;
            const uint32_t right = (uint32_t)(rr < (int64_t)num_cbases ? rr : (int64_t)num_cbases);
            if (r_bind < left || r_bind > right) return;
            bool should_update = true;
            if (pvs_supp_kpos > -1 && (uint64_t)sp[c] <= (uint64_t)k + (uint64_t)pvs_supp_kpos)      // overlapping / adjacent neighbour:
                if ((uint64_t)(r_bind - pvs_supp_r_bind) != (uint64_t)sp[c] - (uint64_t)pvs_supp_kpos) should_update = false;   // offsets must agree
            if (should_update) {
                pvs_supp_kpos = (int64_t)sp[c];
                pvs_supp_r_bind = r_bind;
                vote(c);
            }
        };
        if (hi > lo) {
            const uint32_t A = (uint32_t)(toff + (int32_t)lo), B = (uint32_t)(toff + (int32_t)hi);      // entries [A, B) of the tile
            const uint32_t want = ((uint32_t)kmer & 0xffu) * 0x01010101u;
            for (uint32_t wi = (B - 1) >> 2;; --wi) {
                const uint32_t x = tags[wi] ^ want;
                uint32_t m = (x - 0x01010101u) & ~x & 0x80808080u;      // bytes of x that may be zero (every zero byte is among them)
                while (m) {
                    const uint32_t j = (31u - (uint32_t)__clz(m)) >> 3;       // highest flagged byte first
                    m &= ~(0x80u << (8 * j));
                    const uint32_t e = 4 * wi + j;
                    if (e >= A && e < B) candidate((uint32_t)((int32_t)e - toff));
                }
                if (wi == (A >> 2)) break;
            }
        }
    }
    S.kmer = kmer; S.kmer_len = kmer_len; S.pvs_supp_kpos = pvs_supp_kpos; S.pvs_supp_r_bind = pvs_supp_r_bind;
    S.lo = lo; S.hi = hi; S.word = word; S.r_ind = r_ind;
    return done;
}
}  // namespace

// TT lanes = TT reads per block: 256 for reads of a few hundred bases (their stretches overlap almost entirely), 64 for long reads
// (the lanes of a block are then in the same tile most of the time)
template <typename KidT, int TT>
__global__ void __launch_bounds__(TT) support_kmers_kernel(SupportReads R, uint32_t k, uint32_t n_solid, const uint32_t* __restrict__ spos,
                                                           const KidT* __restrict__ kids, uint32_t* __restrict__ cov, uint32_t* __restrict__ sup) {
    // (64 long reads per block walk serially for milliseconds: what hides their LDS latency is other blocks on the CU, so their tile
    // is a quarter of the short reads' — 10.5 KB instead of 42 KB of LDS per block, 15 blocks per CU instead of 3)
    constexpr uint32_t KCAP = TT >= 256 ? hypo::KCAP : hypo::KCAP / 4;
    __shared__ uint32_t s_sp[KCAP];
    __shared__ KidT s_kd[KCAP];
    __shared__ int s_cov[KCAP + 1];                           // difference array of the coverage
    __shared__ uint32_t s_sup[KCAP];
    __shared__ uint32_t s_tag[KCAP / 4];                      // low byte of every id, entry i in byte i % 4 of word i / 4
    __shared__ int s_part[TT];
    __shared__ uint32_t s_lo, s_hi, s_next;
    const uint32_t tid = threadIdx.x;
    const uint32_t a = blockIdx.x * TT + tid;
    bool active = a < R.n_alignments;
    uint32_t rb = 0, re = 0, first = 0, last = 0;
    if (active) {
        rb = R.rb[a]; re = R.re[a];
        first = count_less(spos, n_solid, rb);
        last = count_less(spos, n_solid, re);
        for (uint32_t i = last; i > first; --i)               // drop k-mers that do not lie wholly inside the read (:70-77)
            if (spos[i - 1] + k <= re) { last = i; break; }
        active = last > first;
    }
    if (tid == 0) { s_lo = 0xffffffffu; s_hi = 0u; }
    __syncthreads();
    if (active) { atomicMin(&s_lo, first); atomicMax(&s_hi, last); }
    __syncthreads();
    const uint32_t blo = s_lo, bhi = s_hi;
    if (blo >= bhi) return;                                   // (block-uniform: no read of the block sees a solid k-mer)
    const uint32_t n = active ? last - first : 0u;
    const uint8_t* rd = active ? R.reads2 + R.seq_off[a] : nullptr;
    const uint32_t nq = active ? R.qae[a] : 0u;
    WalkState S;
    bool walking = active;                                    // the read has positions left
    uint32_t g0 = blo;                                        // the tile: entries [g0, g0 + tn) of the tables
    for (;;) {
        const uint32_t tn = bhi - g0 < KCAP ? bhi - g0 : KCAP;
        for (uint32_t i = tid; i < tn; i += TT) { s_sp[i] = spos[g0 + i]; s_kd[i] = kids[g0 + i]; s_sup[i] = 0u; }
        if (tid == 0) s_next = 0xffffffffu;
        __syncthreads();
        for (uint32_t w = tid; w < (tn + 3) / 4; w += TT) {
            uint32_t v = 0;
            for (uint32_t j = 0; j < 4 && 4 * w + j < tn; ++j) v |= ((uint32_t)s_kd[4 * w + j] & 0xffu) << (8 * j);
            s_tag[w] = v;
        }
        __syncthreads();
        if (walking && first + S.lo < g0 + tn) {              // (a read whose first entry lies behind the tile waits)
            const int32_t off = (int32_t)(first - g0);       // entry c of the read = entry off + c of the tile (negative once the read began in an earlier tile)
            const uint32_t tile_end = g0 + tn - first < n ? g0 + tn - first : n;
            const bool done = walk_read<KidT>(S, s_sp + off, s_kd + off, s_tag, off, n, tile_end, rb, re, rd, nq, k, [&](uint32_t c) { atomicAdd(&s_sup[off + (int32_t)c], 1u); });
            if (done) walking = false;
        }
        if (walking) atomicMin(&s_next, first + S.lo);
        __syncthreads();
        // what the tile collected goes out; the coverage of the entries no later tile comes back to: [g0, next tile's start)
        for (uint32_t i = tid; i < tn; i += TT) { const uint32_t v = s_sup[i]; if (v) atomicAdd(&sup[g0 + i], v); }
        uint32_t g1 = s_next == 0xffffffffu ? bhi : s_next;           // (>= g0: cursors only grow; > g0 unless a lane paused without moving, which a tile of KCAP entries excludes)
        // forward-progress guard, as in support_minimizers_kernel: should the invariant above ever be violated (duplicate positions in a
        // batch's coordinate space, a k beyond what KCAP / 4 entries cover) the tile still advances and the launch ends instead of spinning
        if (g1 <= g0) g1 = g0 + 1 < bhi ? g0 + 1 : bhi;
        for (uint32_t c0 = g0; c0 < g1; c0 += KCAP) {
            const uint32_t c1 = g1 - c0 < KCAP ? g1 : c0 + KCAP, cn = c1 - c0;
            __syncthreads();
            for (uint32_t i = tid; i <= cn; i += TT) s_cov[i] = 0;
            __syncthreads();
            if (active) {
                const uint32_t x0 = first < c0 ? c0 : first, x1 = last > c1 ? c1 : last;
                if (x0 < x1) { atomicAdd(&s_cov[x0 - c0], 1); atomicAdd(&s_cov[x1 - c0], -1); }
            }
            __syncthreads();
            // coverage = prefix sums of the difference array: every thread owns a run of consecutive entries
            const uint32_t per = (cn + TT - 1) / TT;
            const uint32_t i0 = tid * per < cn ? tid * per : cn, i1 = i0 + per < cn ? i0 + per : cn;
            int sum = 0;
            for (uint32_t i = i0; i < i1; ++i) sum += s_cov[i];
            s_part[tid] = sum;
            __syncthreads();
            int run = 0;
            for (uint32_t t = 0; t < tid; ++t) run += s_part[t];
            for (uint32_t i = i0; i < i1; ++i) {
                run += s_cov[i];
                if (run) atomicAdd(&cov[c0 + i], (uint32_t)run);
            }
        }
        if (g1 >= bhi) break;
        g0 = g1;
        __syncthreads();
    }
}

// ---- Alignment::update_minimisers_support -----------------------------------------------------------------------------------
// The reference collects the read's window minimizers (forward strand, k = w = 10) in a multimap and then, for every minimizer
// of every mega-window the read touches, looks for that k-mer among them within [c_dist - 2k, c_dist + 3k] of where the contig
// has it.  Both lists are sorted by position, so the device joins them on the fly: the read's minimizers are produced in
// order and matched against the contig minimizers whose range can contain them (two cursors that only move forward) — the
// same (contig minimizer, read minimizer) pairs vote, nothing is stored per read.

// The minimizers of the mega-windows a read touches are ONE run of table entries (the windows first_w, first_w + 2, ... of a contig
// are consecutive MWMinimiserInfo entries, positions ascending), so the cursor over windows is an index into that run; and the reads
// of a block are neighbours, so their runs are one short stretch of the table: like the k-mer kernel above, the block keeps that
// stretch (positions, k-mers) and its counters in LDS.  C3 batch of 2 M reads: 10.1 -> see profiles (the largest kernel of that run).
namespace {
constexpr uint32_t MCAP = 1024;                           // minimizers of a block's tile held in LDS
// One read's votes (src/Alignment.cpp:134-220) as a resumable walk over a tile of the table, like WalkState above: P[e], Q[e] =
// position and k-mer of entry e for e in [tile start, tile_end) (pointers offset by the caller).  The read's own window minimizers
// are produced in order and matched against the contig minimizers whose range [c_dist - 2K, c_dist + 3K] holds them; a position
// is only begun when the tile reaches beyond everything it can look at.
struct MinState {
    uint32_t key[10];
    uint32_t kmer = 0, processed = 0, last_found = 0, lo = 0, i = 0, word = 0;
};
template <typename Vote>
__device__ __forceinline__ bool walk_minimizers(MinState& S, const uint32_t* P, const uint32_t* Q, uint32_t tile_end, uint32_t E1, uint32_t rb, uint32_t re,
                                                const uint8_t* rd, uint32_t nq, Vote vote) {
    constexpr uint32_t K = 10, W = 10;                       // Minimizer_settings (include/globalDefs.hpp:128-139)
    const uint32_t mask = (1u << (2 * K)) - 1u;
    const uint16_t num_cbases = (uint16_t)(re - rb);          // 16-bit in the reference (:188)
    // positions the tile suffices for: an entry is looked at while its c_dist <= start + 2K <= i + 2K, so position i may begin
    // when the tile's last entry lies behind that (or the tile ends the read's entries)
    const uint32_t i_limit = tile_end >= E1 ? nq : (P[tile_end - 1] - rb > 2 * K ? P[tile_end - 1] - rb - 2 * K : 0u);
    uint32_t kmer = S.kmer, processed = S.processed, last_found = S.last_found, lo = S.lo, word = S.word, i = S.i;
    uint32_t key[W];
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) key[j] = S.key[j];
    const uint32_t i0 = i;
    bool done = true;
    for (; i < nq; ++i) {                                     // (a 2-bit read has no N: every position from K - 1 on pushes a k-mer)
        if (i >= i_limit) { done = false; break; }
#pragma unroll
        for (int j = W - 1; j >= 1; --j) key[j] = key[j - 1];
        if ((i & 15u) == 0 || i == i0) word = load_u32_unaligned(rd + ((i >> 4) << 2));
        kmer = ((kmer << 2) | ((word >> (8 * ((i >> 2) & 3u) + 6 - 2 * (i & 3u))) & 3u)) & mask;
        key[0] = i + 1 >= K ? kmer : 0xffffffffu;
        if (i + 1 < K) continue;
        if (++processed < W) continue;
        uint32_t best = key[0], at = 0;
#pragma unroll
        for (uint32_t j = 1; j < W; ++j) if (key[j] <= best) { best = key[j]; at = j; }      // the earliest of equal keys (monotone queue's front)
        const uint32_t start = (i - at) - K + 1;
        if (start == last_found) continue;
        last_found = start;
        // contig minimizers with c_dist + 3K < start can never match again (start only grows)
        while (lo < tile_end && (uint64_t)(P[lo] - rb) + 3 * K < start) ++lo;
        if (lo >= E1) { i = nq; break; }
        for (uint32_t e = lo; e < tile_end; ++e) {
            const uint32_t p = P[e];
            if (p >= re) break;
            const uint32_t c_dist = p - rb;
            if (c_dist > start + 2 * K) break;                // its range starts behind `start`: so do all later ones
            const uint32_t range_left = c_dist > 2 * K ? c_dist - 2 * K : 0u;
            const uint16_t rr16 = (uint16_t)(c_dist + 3 * K);
            const uint32_t range_right = num_cbases < rr16 ? num_cbases : rr16;
            if (Q[e] == best && start >= range_left && start <= range_right) vote(e);
        }
    }
    S.kmer = kmer; S.processed = processed; S.last_found = last_found; S.lo = lo; S.word = word; S.i = i;
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) S.key[j] = key[j];
    return done;
}
}  // namespace

template <int TT>
__global__ void __launch_bounds__(TT) support_minimizers_kernel(SupportReads R, MegaWindows M, uint32_t* __restrict__ cov, uint32_t* __restrict__ sup) {
    constexpr uint32_t MCAP = TT >= 256 ? hypo::MCAP : hypo::MCAP / 4;      // (long reads: small tiles, many blocks per CU; see support_kmers_kernel)
    __shared__ uint32_t s_pos[MCAP], s_min[MCAP], s_sup[MCAP];
    __shared__ int s_cov[MCAP + 1];                           // difference array of the coverage
    __shared__ int s_part[TT];
    __shared__ uint32_t s_lo, s_hi, s_next;
    const uint32_t tid = threadIdx.x;
    const uint32_t a = blockIdx.x * TT + tid;
    bool active = a < R.n_alignments;
    uint32_t rb = 0, re = 0, E0 = 0, E1 = 0, Ec = 0;
    if (active) {
        const uint32_t c = R.read_contig[a];
        const uint32_t base = M.contig_base[c];
        rb = R.rb[a] - base; re = R.re[a] - base;             // contig-local
        const uint32_t* S = M.start + M.reg_base[c];
        const uint32_t nS = M.reg_base[c + 1] - M.reg_base[c];
        const int64_t first = (int64_t)count_less_equal(S, nS, rb) - 1;        // _reg_pos.rank(_rb + 1) - 1
        const int64_t last = (int64_t)count_less(S, nS, re);                   // _reg_pos.rank(_re)
        const bool even = M.win_even[c] != 0;
        const int64_t first_w = ((even && first % 2 == 0) || (!even && first % 2 == 1)) ? first : first + 1;
        const int64_t last_w = ((even && last % 2 == 0) || (!even && last % 2 == 1)) ? last : last - 1;
        active = last_w >= first_w;
        if (active) {
            const uint32_t x0 = M.info_base[c] + (uint32_t)(even ? first_w / 2 : (first_w - 1) / 2);
            const uint32_t x1 = M.info_base[c] + (uint32_t)(even ? last_w / 2 : (last_w - 1) / 2);
            // the first window from its first minimizer at or behind the read's start on (M.rel_pos holds POSITIONS by now,
            // minimizer_positions_kernel: a read deep inside a mega-window of tens of kbp used to add up the distances of every
            // minimizer in front of it); minimizers in front of the read neither count as covered nor can they match
            uint32_t lo = M.mw_off[x0], hi = M.mw_off[x0 + 1];
            while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (M.rel_pos[m] < rb) lo = m + 1; else hi = m; }
            E0 = lo; E1 = M.mw_off[x1 + 1];
            active = E0 < E1;
            // the read covers the mega-window minimizers lying inside its span (:189-203): entries [E0, Ec), positions ascending.
            // (`break` of the reference's inner loop: a minimizer at or behind the read's end ends its window; the windows behind it
            // start behind the read's end too.)
            uint32_t l2 = E0, h2 = E1;
            while (l2 < h2) { const uint32_t m = (l2 + h2) >> 1; if (M.rel_pos[m] < re) l2 = m + 1; else h2 = m; }
            Ec = l2;
            if (Ec == E0) active = false;                     // (nothing covered: the reference returns before it looks at the read)
        }
    }
    if (tid == 0) { s_lo = 0xffffffffu; s_hi = 0u; }
    __syncthreads();
    if (active) { atomicMin(&s_lo, E0); atomicMax(&s_hi, E1); }
    __syncthreads();
    const uint32_t blo = s_lo, bhi = s_hi;
    if (blo >= bhi) return;                                   // (block-uniform)
    const uint8_t* rd = active ? R.reads2 + R.seq_off[a] : nullptr;
    const uint32_t nq = active ? R.qae[a] : 0u;
    MinState S;
#pragma unroll
    for (uint32_t j = 0; j < 10; ++j) S.key[j] = 0xffffffffu;
    S.last_found = nq + 1; S.lo = E0;
    bool walking = active;
    uint32_t g0 = blo;
    for (;;) {
        const uint32_t tn = bhi - g0 < MCAP ? bhi - g0 : MCAP;
        for (uint32_t i = tid; i < tn; i += TT) { s_pos[i] = M.rel_pos[g0 + i]; s_min[i] = M.minimisers[g0 + i]; s_sup[i] = 0u; }
        if (tid == 0) s_next = 0xffffffffu;
        __syncthreads();
        if (walking && S.lo < g0 + tn) {
            const uint32_t tile_end = g0 + tn < E1 ? g0 + tn : E1;
            // (pointers shifted so that absolute entry numbers index the tile; entries before g0 are never touched: S.lo >= g0)
            const bool done = walk_minimizers(S, s_pos - g0, s_min - g0, tile_end, E1, rb, re, rd, nq, [&](uint32_t e) { atomicAdd(&s_sup[e - g0], 1u); });
            if (done) walking = false;
        }
        if (walking) atomicMin(&s_next, S.lo);
        __syncthreads();
        for (uint32_t i = tid; i < tn; i += TT) { const uint32_t v = s_sup[i]; if (v) atomicAdd(&sup[g0 + i], v); }
        uint32_t g1 = s_next == 0xffffffffu ? bhi : s_next;
        if (g1 <= g0) g1 = g0 + 1 < bhi ? g0 + 1 : bhi;       // (cannot happen: a tile of >= 256 entries spans more than the 5K bases a position looks at)
        for (uint32_t c0 = g0; c0 < g1; c0 += MCAP) {
            const uint32_t c1 = g1 - c0 < MCAP ? g1 : c0 + MCAP, cn = c1 - c0;
            __syncthreads();
            for (uint32_t i = tid; i <= cn; i += TT) s_cov[i] = 0;
            __syncthreads();
            if (active) {
                const uint32_t x0 = E0 < c0 ? c0 : E0, x1 = Ec > c1 ? c1 : Ec;
                if (x0 < x1) { atomicAdd(&s_cov[x0 - c0], 1); atomicAdd(&s_cov[x1 - c0], -1); }
            }
            __syncthreads();
            const uint32_t per = (cn + TT - 1) / TT;
            const uint32_t i0 = tid * per < cn ? tid * per : cn, i1 = i0 + per < cn ? i0 + per : cn;
            int sum = 0;
            for (uint32_t i = i0; i < i1; ++i) sum += s_cov[i];
            s_part[tid] = sum;
            __syncthreads();
            int run = 0;
            for (uint32_t t = 0; t < tid; ++t) run += s_part[t];
            for (uint32_t i = i0; i < i1; ++i) {
                run += s_cov[i];
                if (run) atomicAdd(&cov[c0 + i], (uint32_t)run);
            }
        }
        if (g1 >= bhi) break;
        g0 = g1;
        __syncthreads();
    }
}

hipError_t support_kmers(const SupportReads& R, uint32_t k, uint32_t n_solid, const uint32_t* spos, const uint64_t* kids, uint32_t* cov, uint32_t* sup, hipStream_t st) {
    if (!R.n_alignments || !n_solid) return hipSuccess;
    if (R.mean_span > kLongReadSpan) hipLaunchKernelGGL((support_kmers_kernel<uint64_t, 64>), dim3((R.n_alignments + 63) / 64), dim3(64), 0, st, R, k, n_solid, spos, kids, cov, sup);
    else hipLaunchKernelGGL((support_kmers_kernel<uint64_t, T>), dim3((R.n_alignments + T - 1) / T), dim3(T), 0, st, R, k, n_solid, spos, kids, cov, sup);
    return hipGetLastError();
}
hipError_t support_kmers32(const SupportReads& R, uint32_t k, uint32_t n_solid, const uint32_t* spos, const uint32_t* kids32, uint32_t* cov, uint32_t* sup, hipStream_t st) {
    if (!R.n_alignments || !n_solid) return hipSuccess;
    if (R.mean_span > kLongReadSpan) hipLaunchKernelGGL((support_kmers_kernel<uint32_t, 64>), dim3((R.n_alignments + 63) / 64), dim3(64), 0, st, R, k, n_solid, spos, kids32, cov, sup);
    else hipLaunchKernelGGL((support_kmers_kernel<uint32_t, T>), dim3((R.n_alignments + T - 1) / T), dim3(T), 0, st, R, k, n_solid, spos, kids32, cov, sup);
    return hipGetLastError();
}
// out[i] = in[i] + base: the contig-local positions a resident scan kept, moved into the batch's coordinate space
__global__ void __launch_bounds__(T) add_base_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n, uint32_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * T + threadIdx.x;
    if (i < n) out[i] = in[i] + base;
}
hipError_t add_base(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t base, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(add_base_kernel, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, st, in, out, n, base);
    return hipGetLastError();
}
// MWMinimiserInfo::rel_pos -> positions: one lane per mega-window adds its distances up once (instead of every read that lies in it)
__global__ void __launch_bounds__(T) minimizer_positions_kernel(MegaWindows M, uint32_t n_contigs, uint32_t n_info) {
    const uint32_t x = blockIdx.x * T + threadIdx.x;
    if (x >= n_info) return;
    const uint32_t e0 = M.mw_off[x], e1 = M.mw_off[x + 1];
    if (e0 >= e1) return;
    const uint32_t c = count_less_equal(M.info_base, n_contigs, x) - 1;       // the contig whose tables hold entry x
    const uint32_t xl = x - M.info_base[c];
    const uint32_t w = M.win_even[c] ? 2 * xl : 2 * xl + 1;                    // (MwCursor::info_of, the other way round)
    uint32_t pos = M.start[M.reg_base[c] + w];
    for (uint32_t e = e0; e < e1; ++e) { pos += M.rel_pos[e]; M.rel_pos[e] = pos; }
}

hipError_t support_minimizers(const SupportReads& R, const MegaWindows& M, uint32_t n_contigs, uint32_t n_info, uint32_t* cov, uint32_t* sup, hipStream_t st) {
    if (!R.n_alignments || !n_info || !n_contigs) return hipSuccess;
    hipLaunchKernelGGL(minimizer_positions_kernel, dim3((n_info + T - 1) / T), dim3(T), 0, st, M, n_contigs, n_info);
    if (R.mean_span > kLongReadSpan) hipLaunchKernelGGL(support_minimizers_kernel<64>, dim3((R.n_alignments + 63) / 64), dim3(64), 0, st, R, M, cov, sup);
    else hipLaunchKernelGGL(support_minimizers_kernel<T>, dim3((R.n_alignments + T - 1) / T), dim3(T), 0, st, R, M, cov, sup);
    return hipGetLastError();
}

}  // namespace hypo
