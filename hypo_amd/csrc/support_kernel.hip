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
constexpr uint32_t KCAP = 2048;                           // solid k-mers of a block's stretch held in LDS
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
    typedef uint32_t __attribute__((aligned(1))) u32u;
    return *(const u32u*)p;
}
// one read's walk (src/Alignment.cpp:78-132).  sp / kd: the solid k-mers [first, last) of the read's span (LDS or global);
// vote(c): k-mer first + c is supported by the read
// TAGS: `tags` holds the low byte of every k-mer id of the block's stretch, four to a word, entry `toff` being the read's first
// k-mer: the ~14 candidates of a read k-mer are screened four at a time on that byte (no false negatives; a hit is verified on the
// whole id), from the last one down, as the plain loop visits them
template <typename KidT, bool TAGS, typename Vote>
__device__ __forceinline__ void walk_read(const uint32_t* sp, const KidT* kd, const uint32_t* tags, uint32_t toff, uint32_t n, uint32_t rb, uint32_t re, const uint8_t* rd, uint32_t nq, uint32_t k, Vote vote) {
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t num_cbases = re - rb;
    uint64_t kmer = 0; uint32_t kmer_len = 0;
    int64_t pvs_supp_kpos = -1; uint32_t pvs_supp_r_bind = 0;
    uint32_t lo = 0, hi = 0;                                  // solid k-mers [lo, hi): offset within k of the read k-mer's
    uint32_t word = 0;                                        // 16 bases of the read: byte j of the word holds bases 4j .. 4j + 3, first base in the top bits
    for (uint32_t r_ind = 0; r_ind < nq; ++r_ind) {
        if ((r_ind & 15u) == 0) word = load_u32_unaligned(rd + (r_ind >> 2));
        const uint32_t b = (word >> (8 * ((r_ind >> 2) & 3u) + 6 - 2 * (r_ind & 3u))) & 3u;
        kmer = ((kmer << 2) | b) & kmask;
        if (kmer_len < k) ++kmer_len;
        if (kmer_len != k) continue;
        const uint32_t r_bind = r_ind + 1 - k;
        while (hi < n && (int64_t)sp[hi] - (int64_t)rb <= (int64_t)r_bind + (int64_t)k) ++hi;
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
        if (!TAGS) {
            for (uint32_t c = hi; c-- > lo;) candidate(c);
        } else if (hi > lo) {
            const uint32_t A = toff + lo, B = toff + hi;      // entries [A, B) of the block's stretch
            const uint32_t want = ((uint32_t)kmer & 0xffu) * 0x01010101u;
            for (uint32_t wi = (B - 1) >> 2;; --wi) {
                const uint32_t x = tags[wi] ^ want;
                uint32_t m = (x - 0x01010101u) & ~x & 0x80808080u;      // bytes of x that may be zero (every zero byte is among them)
                while (m) {
                    const uint32_t j = (31u - (uint32_t)__clz(m)) >> 3;       // highest flagged byte first
                    m &= ~(0x80u << (8 * j));
                    const uint32_t e = 4 * wi + j;
                    if (e >= A && e < B) candidate(e - toff);
                }
                if (wi == (A >> 2)) break;
            }
        }
    }
}
}  // namespace

template <typename KidT>
__global__ void __launch_bounds__(T) support_kmers_kernel(SupportReads R, uint32_t k, uint32_t n_solid, const uint32_t* __restrict__ spos,
                                                          const KidT* __restrict__ kids, uint32_t* __restrict__ cov, uint32_t* __restrict__ sup) {
    __shared__ uint32_t s_sp[KCAP];
    __shared__ KidT s_kd[KCAP];
    __shared__ int s_cov[KCAP + 1];                           // difference array of the coverage
    __shared__ uint32_t s_sup[KCAP];
    __shared__ uint32_t s_tag[KCAP / 4];                      // low byte of every id, entry i in byte i % 4 of word i / 4
    __shared__ int s_part[T];
    __shared__ uint32_t s_lo, s_hi;
    const uint32_t tid = threadIdx.x;
    const uint32_t a = blockIdx.x * T + tid;
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
    const uint32_t span = bhi - blo;
    const bool in_lds = span <= KCAP;
    if (in_lds) {
        for (uint32_t i = tid; i < span; i += T) { s_sp[i] = spos[blo + i]; s_kd[i] = kids[blo + i]; s_cov[i] = 0; s_sup[i] = 0u; }
        if (tid == 0) s_cov[span] = 0;
        __syncthreads();
        for (uint32_t w = tid; w < (span + 3) / 4; w += T) {
            uint32_t v = 0;
            for (uint32_t j = 0; j < 4 && 4 * w + j < span; ++j) v |= ((uint32_t)s_kd[4 * w + j] & 0xffu) << (8 * j);
            s_tag[w] = v;
        }
        __syncthreads();
    }
    if (active) {
        const uint32_t n = last - first;
        const uint8_t* rd = R.reads2 + R.seq_off[a];
        const uint32_t nq = R.qae[a];
        if (in_lds) {
            const uint32_t off = first - blo;
            atomicAdd(&s_cov[off], 1); atomicAdd(&s_cov[off + n], -1);
            walk_read<KidT, true>(s_sp + off, s_kd + off, s_tag, off, n, rb, re, rd, nq, k, [&](uint32_t c) { atomicAdd(&s_sup[off + c], 1u); });
        } else {
            for (uint32_t t = 0; t < n; ++t) atomicAdd(&cov[first + t], 1u);
            walk_read<KidT, false>(spos + first, kids + first, nullptr, 0u, n, rb, re, rd, nq, k, [&](uint32_t c) { atomicAdd(&sup[first + c], 1u); });
        }
    }
    if (!in_lds) return;
    __syncthreads();
    // coverage = prefix sums of the difference array: every thread owns a run of consecutive entries
    const uint32_t per = (span + T - 1) / T;
    const uint32_t i0 = tid * per, i1 = i0 + per < span ? i0 + per : span;
    int sum = 0;
    for (uint32_t i = i0; i < i1; ++i) sum += s_cov[i];
    s_part[tid] = sum;
    __syncthreads();
    int run = 0;
    for (uint32_t t = 0; t < tid; ++t) run += s_part[t];
    for (uint32_t i = i0; i < i1; ++i) {
        run += s_cov[i];
        if (run) atomicAdd(&cov[blo + i], (uint32_t)run);
        const uint32_t v = s_sup[i];
        if (v) atomicAdd(&sup[blo + i], v);
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
constexpr uint32_t MCAP = 1024;                           // minimizers of a block's stretch held in LDS
// one read's votes (src/Alignment.cpp:134-220) over entries [E0, E1): P[e - pbase] = position, Q[e - pbase] = k-mer of entry e
template <typename Cover, typename Vote>
__device__ __forceinline__ void walk_minimizers(const uint32_t* P, const uint32_t* Q, uint32_t pbase, uint32_t E0, uint32_t E1, uint32_t rb, uint32_t re,
                                                const uint8_t* rd, uint32_t nq, Cover cover, Vote vote) {
    constexpr uint32_t K = 10, W = 10;                       // Minimizer_settings (include/globalDefs.hpp:128-139)
    // the read's minimizers cover the mega-window minimizers lying inside its span (:189-203): coverage first.  (`break` of the
    // reference's inner loop: a minimizer at or behind the read's end ends its window; the windows behind it start behind the read's
    // end too.)  Entries in front of the read were skipped by the caller.
    uint32_t Ec = E0;
    while (Ec < E1 && P[Ec - pbase] < re) ++Ec;
    if (Ec == E0) return;
    cover(E0, Ec);
    // the read's own window minimizers, in order, against the contig minimizers whose range [c_dist - 2K, c_dist + 3K] holds them
    const uint32_t mask = (1u << (2 * K)) - 1u;
    const uint16_t num_cbases = (uint16_t)(re - rb);          // 16-bit in the reference (:188)
    uint32_t key[W];
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) key[j] = 0xffffffffu;
    uint32_t kmer = 0, processed = 0, last_found = nq + 1;
    uint32_t lo = E0;                                         // first contig minimizer that can still match
    uint32_t word = 0;                                        // 16 bases of the read at a time
    for (uint32_t i = 0; i < nq; ++i) {                       // (a 2-bit read has no N: every position from K - 1 on pushes a k-mer)
#pragma unroll
        for (int j = W - 1; j >= 1; --j) key[j] = key[j - 1];
        if ((i & 15u) == 0) word = load_u32_unaligned(rd + (i >> 2));
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
        while (lo < E1 && (uint64_t)(P[lo - pbase] - rb) + 3 * K < start) ++lo;
        if (lo >= E1) break;
        for (uint32_t e = lo; e < E1; ++e) {
            const uint32_t p = P[e - pbase];
            if (p >= re) break;
            const uint32_t c_dist = p - rb;
            if (c_dist > start + 2 * K) break;                // its range starts behind `start`: so do all later ones
            const uint32_t range_left = c_dist > 2 * K ? c_dist - 2 * K : 0u;
            const uint16_t rr16 = (uint16_t)(c_dist + 3 * K);
            const uint32_t range_right = num_cbases < rr16 ? num_cbases : rr16;
            if (Q[e - pbase] == best && start >= range_left && start <= range_right) vote(e);
        }
    }
}
}  // namespace

__global__ void __launch_bounds__(T) support_minimizers_kernel(SupportReads R, MegaWindows M, uint32_t* __restrict__ cov, uint32_t* __restrict__ sup) {
    __shared__ uint32_t s_pos[MCAP], s_min[MCAP], s_sup[MCAP];
    __shared__ int s_cov[MCAP + 1];                           // difference array of the coverage
    __shared__ int s_part[T];
    __shared__ uint32_t s_lo, s_hi;
    const uint32_t tid = threadIdx.x;
    const uint32_t a = blockIdx.x * T + tid;
    bool active = a < R.n_alignments;
    uint32_t rb = 0, re = 0, E0 = 0, E1 = 0;
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
        }
    }
    if (tid == 0) { s_lo = 0xffffffffu; s_hi = 0u; }
    __syncthreads();
    if (active) { atomicMin(&s_lo, E0); atomicMax(&s_hi, E1); }
    __syncthreads();
    const uint32_t blo = s_lo, bhi = s_hi;
    if (blo >= bhi) return;                                   // (block-uniform)
    const uint32_t span = bhi - blo;
    const bool in_lds = span <= MCAP;
    if (in_lds) {
        for (uint32_t i = tid; i < span; i += T) { s_pos[i] = M.rel_pos[blo + i]; s_min[i] = M.minimisers[blo + i]; s_cov[i] = 0; s_sup[i] = 0u; }
        if (tid == 0) s_cov[span] = 0;
        __syncthreads();
    }
    if (active) {
        const uint8_t* rd = R.reads2 + R.seq_off[a];
        const uint32_t nq = R.qae[a];
        if (in_lds)
            walk_minimizers(s_pos, s_min, blo, E0, E1, rb, re, rd, nq,
                            [&](uint32_t e0, uint32_t e1) { atomicAdd(&s_cov[e0 - blo], 1); atomicAdd(&s_cov[e1 - blo], -1); },
                            [&](uint32_t e) { atomicAdd(&s_sup[e - blo], 1u); });
        else
            walk_minimizers(M.rel_pos, M.minimisers, 0u, E0, E1, rb, re, rd, nq,
                            [&](uint32_t e0, uint32_t e1) { for (uint32_t e = e0; e < e1; ++e) atomicAdd(&cov[e], 1u); },
                            [&](uint32_t e) { atomicAdd(&sup[e], 1u); });
    }
    if (!in_lds) return;
    __syncthreads();
    const uint32_t per = (span + T - 1) / T;
    const uint32_t i0 = tid * per < span ? tid * per : span, i1 = i0 + per < span ? i0 + per : span;
    int sum = 0;
    for (uint32_t i = i0; i < i1; ++i) sum += s_cov[i];
    s_part[tid] = sum;
    __syncthreads();
    int run = 0;
    for (uint32_t t = 0; t < tid; ++t) run += s_part[t];
    for (uint32_t i = i0; i < i1; ++i) {
        run += s_cov[i];
        if (run) atomicAdd(&cov[blo + i], (uint32_t)run);
        const uint32_t v = s_sup[i];
        if (v) atomicAdd(&sup[blo + i], v);
    }
}

hipError_t support_kmers(const SupportReads& R, uint32_t k, uint32_t n_solid, const uint32_t* spos, const uint64_t* kids, uint32_t* cov, uint32_t* sup, hipStream_t st) {
    if (!R.n_alignments || !n_solid) return hipSuccess;
    hipLaunchKernelGGL(support_kmers_kernel<uint64_t>, dim3((R.n_alignments + T - 1) / T), dim3(T), 0, st, R, k, n_solid, spos, kids, cov, sup);
    return hipGetLastError();
}
hipError_t support_kmers32(const SupportReads& R, uint32_t k, uint32_t n_solid, const uint32_t* spos, const uint32_t* kids32, uint32_t* cov, uint32_t* sup, hipStream_t st) {
    if (!R.n_alignments || !n_solid) return hipSuccess;
    hipLaunchKernelGGL(support_kmers_kernel<uint32_t>, dim3((R.n_alignments + T - 1) / T), dim3(T), 0, st, R, k, n_solid, spos, kids32, cov, sup);
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
    hipLaunchKernelGGL(support_minimizers_kernel, dim3((R.n_alignments + T - 1) / T), dim3(T), 0, st, R, M, cov, sup);
    return hipGetLastError();
}

}  // namespace hypo
