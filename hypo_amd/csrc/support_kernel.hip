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
template <typename KidT>
__global__ void __launch_bounds__(T) support_kmers_kernel(SupportReads R, uint32_t k, uint32_t n_solid, const uint32_t* __restrict__ spos,
                                                          const KidT* __restrict__ kids, uint32_t* __restrict__ cov, uint32_t* __restrict__ sup) {
    const uint32_t a = blockIdx.x * T + threadIdx.x;
    if (a >= R.n_alignments) return;
    const uint32_t rb = R.rb[a], re = R.re[a];
    const uint32_t first = count_less(spos, n_solid, rb);
    uint32_t last = count_less(spos, n_solid, re);
    for (uint32_t i = last; i > first; --i)                   // drop k-mers that do not lie wholly inside the read (:70-77)
        if (spos[i - 1] + k <= re) { last = i; break; }
    if (last <= first) return;
    const uint32_t n = last - first;
    for (uint32_t t = 0; t < n; ++t) atomicAdd(&cov[first + t], 1u);
    const uint32_t* sp = spos + first;
    const KidT* kd = kids + first;
    const uint8_t* rd = R.reads2 + R.seq_off[a];
    const uint32_t nq = R.qae[a];
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t num_cbases = re - rb;
    uint64_t kmer = 0; uint32_t kmer_len = 0;
    int64_t pvs_supp_kpos = -1; uint32_t pvs_supp_r_bind = 0;
    uint32_t lo = 0, hi = 0;                                  // solid k-mers [lo, hi): offset within k of the read k-mer's
    for (uint32_t r_ind = 0; r_ind < nq; ++r_ind) {
        kmer = ((kmer << 2) | base2(rd, r_ind)) & kmask;
        if (kmer_len < k) ++kmer_len;
        if (kmer_len != k) continue;
        const uint32_t r_bind = r_ind + 1 - k;
        while (hi < n && (int64_t)sp[hi] - (int64_t)rb <= (int64_t)r_bind + (int64_t)k) ++hi;
        while (lo < hi && (int64_t)sp[lo] - (int64_t)rb + (int64_t)k < (int64_t)r_bind) ++lo;
        for (uint32_t c = hi; c-- > lo;) {
            if ((uint64_t)kd[c] != kmer) continue;
            const int64_t c_dist = (int64_t)sp[c] - (int64_t)rb;
            const uint32_t left = c_dist > (int64_t)k ? (uint32_t)(c_dist - k) : 0u;
            const int64_t rr = c_dist + (int64_t)k;
            const uint32_t right = (uint32_t)(rr < (int64_t)num_cbases ? rr : (int64_t)num_cbases);
            if (r_bind < left || r_bind > right) continue;
            bool should_update = true;
            if (pvs_supp_kpos > -1 && (uint64_t)sp[c] <= (uint64_t)k + (uint64_t)pvs_supp_kpos)      // overlapping / adjacent neighbour:
                if ((uint64_t)(r_bind - pvs_supp_r_bind) != (uint64_t)sp[c] - (uint64_t)pvs_supp_kpos) should_update = false;   // offsets must agree
            if (should_update) {
                pvs_supp_kpos = (int64_t)sp[c];
                pvs_supp_r_bind = r_bind;
                atomicAdd(&sup[first + c], 1u);
            }
        }
    }
}

// ---- Alignment::update_minimisers_support -----------------------------------------------------------------------------------
// The reference collects the read's window minimizers (forward strand, k = w = 10) in a multimap and then, for every minimizer
// of every mega-window the read touches, looks for that k-mer among them within [c_dist - 2k, c_dist + 3k] of where the contig
// has it.  Both lists are sorted by position, so the device joins them on the fly: the read's minimizers are produced in
// order and matched against the contig minimizers whose range can contain them (two cursors that only move forward) — the
// same (contig minimizer, read minimizer) pairs vote, nothing is stored per read.
namespace {
struct MwCursor {                // walks the minimizers of the mega-windows i = first_w, first_w + 2, ... <= last_w of one contig
    const MegaWindows* M; const uint32_t* S; uint32_t info_base; bool even; int64_t i, last_w; uint32_t e, e_end; uint32_t pos;
    __device__ uint32_t info_of(int64_t w) const { return info_base + (uint32_t)(even ? w / 2 : (w - 1) / 2); }
    __device__ void open(int64_t w) {                        // positions of window w start at its own start
        i = w;
        if (w > last_w) { e = e_end = 0; return; }
        const uint32_t x = info_of(w);
        e = M->mw_off[x]; e_end = M->mw_off[x + 1]; pos = S[w];
    }
    // next minimizer in position order: its entry index and position (contig-local); false at the end
    __device__ bool next(uint32_t* entry, uint32_t* p) {
        while (i <= last_w) {
            if (e < e_end) { pos += M->rel_pos[e]; *entry = e; *p = pos; ++e; return true; }
            open(i + 2);
        }
        return false;
    }
};
}  // namespace

__global__ void __launch_bounds__(T) support_minimizers_kernel(SupportReads R, MegaWindows M, uint32_t* __restrict__ cov, uint32_t* __restrict__ sup) {
    const uint32_t a = blockIdx.x * T + threadIdx.x;
    if (a >= R.n_alignments) return;
    constexpr uint32_t K = 10, W = 10;                       // Minimizer_settings (include/globalDefs.hpp:128-139)
    const uint32_t c = R.read_contig[a];
    const uint32_t base = M.contig_base[c];
    const uint32_t rb = R.rb[a] - base, re = R.re[a] - base;  // contig-local
    const uint32_t* S = M.start + M.reg_base[c];
    const uint32_t nS = M.reg_base[c + 1] - M.reg_base[c];
    const int64_t first = (int64_t)count_less_equal(S, nS, rb) - 1;        // _reg_pos.rank(_rb + 1) - 1
    const int64_t last = (int64_t)count_less(S, nS, re);                   // _reg_pos.rank(_re)
    const bool even = M.win_even[c] != 0;
    const int64_t first_w = ((even && first % 2 == 0) || (!even && first % 2 == 1)) ? first : first + 1;
    const int64_t last_w = ((even && last % 2 == 0) || (!even && last % 2 == 1)) ? last : last - 1;
    if (last_w < first_w) return;
    // the read's minimizers cover the mega-window minimizers lying inside its span (:189-203): coverage first
    MwCursor cur{&M, S, M.info_base[c], even, 0, last_w, 0, 0, 0};
    cur.open(first_w);
    bool any = false;
    {
        uint32_t e, p;
        MwCursor t = cur;
        // (`break` of the reference's inner loop: a minimizer at or behind the read's end ends its window; the windows behind
        // it start behind the read's end too)
        while (t.next(&e, &p)) { if (p >= re) break; if (p >= rb) { atomicAdd(&cov[e], 1u); any = true; } }
    }
    if (!any) return;
    // the read's own window minimizers, in order, against the contig minimizers whose range [c_dist - 2K, c_dist + 3K] holds them
    const uint8_t* rd = R.reads2 + R.seq_off[a];
    const uint32_t nq = R.qae[a];
    const uint32_t mask = (1u << (2 * K)) - 1u;
    const uint16_t num_cbases = (uint16_t)(re - rb);          // 16-bit in the reference (:188)
    uint32_t key[W];
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) key[j] = 0xffffffffu;
    uint32_t kmer = 0, processed = 0, last_found = nq + 1;
    MwCursor lo = cur;                                        // first contig minimizer that can still match
    uint32_t lo_e = 0, lo_p = 0; bool lo_ok = lo.next(&lo_e, &lo_p);
    for (uint32_t i = 0; i < nq; ++i) {                       // (a 2-bit read has no N: every position from K - 1 on pushes a k-mer)
#pragma unroll
        for (int j = W - 1; j >= 1; --j) key[j] = key[j - 1];
        kmer = ((kmer << 2) | base2(rd, i)) & mask;
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
        while (lo_ok && (lo_p < rb || (uint64_t)(lo_p - rb) + 3 * K < start)) lo_ok = lo.next(&lo_e, &lo_p);
        if (!lo_ok) break;
        MwCursor t = lo; uint32_t e = lo_e, p = lo_p; bool ok = true;
        while (ok && p < re) {
            const uint32_t c_dist = p - rb;
            if (c_dist > start + 2 * K) break;                // its range starts behind `start`: so do all later ones
            const uint32_t range_left = c_dist > 2 * K ? c_dist - 2 * K : 0u;
            const uint16_t rr16 = (uint16_t)(c_dist + 3 * K);
            const uint32_t range_right = num_cbases < rr16 ? num_cbases : rr16;
            if (M.minimisers[e] == best && start >= range_left && start <= range_right) atomicAdd(&sup[e], 1u);
            ok = t.next(&e, &p);
        }
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
hipError_t support_minimizers(const SupportReads& R, const MegaWindows& M, uint32_t* cov, uint32_t* sup, hipStream_t st) {
    if (!R.n_alignments) return hipSuccess;
    hipLaunchKernelGGL(support_minimizers_kernel, dim3((R.n_alignments + T - 1) / T), dim3(T), 0, st, R, M, cov, sup);
    return hipGetLastError();
}

}  // namespace hypo
