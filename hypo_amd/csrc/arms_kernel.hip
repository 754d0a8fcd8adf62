// arms_kernel.hip — arm selection on the device: which segment of which mapped short read becomes which
// kind of arm of which window, the pruning of windows with too few arms, and the packed window batch itself, written straight
// into HBM for the POA kernels.  Replaces, for short reads,
//   Alignment::find_short_arms   src/Alignment.cpp:222-259      arms_walk_kernel (regions a read touches, arm kinds)
//   Alignment::find_bp           src/Alignment.cpp:321-406      arms_walk_kernel (CIGAR walk -> query break points)
//   Alignment::prepare_short_arm src/Alignment.cpp:408-511      arms_walk_kernel (anchor k-mer / minimizer re-anchoring)
//   Alignment::add_arms + Contig::fill_short_windows  src/Alignment.cpp:301-318, src/Contig.cpp:249-289
//                                                               arms_window_kernel<false> (counts, pruning) and <true> (the batch)
// MI355X mapping: integer / byte work, HBM and latency bound; one lane per alignment for the walk (a read touches a handful of
// regions), one lane per region for the gather.  The gather is window-centric on purpose: the arms of a window must appear in
// alignment (file) order, because the POA result depends on it; a window therefore looks up, in order, the alignments whose
// reference span overlaps it (binary search on the sorted start positions) and finds each one's candidate arm by direct
// indexing — no atomics, no sort, deterministic.  Long reads (round 3): the same kernels over the pseudo regions of
// Contig::prepare_long_windows with ArmsIn::long_mode set — find_long_arms' plain cuts, Filter::initialise per LONG window
// (arms_draftmin_kernel) and Filter::is_good per arm (arm_is_good, inside the walk).
#include <hip/hip_runtime.h>
#include "arms_kernel.hpp"

namespace hypo {

namespace {
constexpr int T = 256;
enum : uint8_t { R_SWS, R_SW, R_WS, R_MWM, R_MW, R_WM, R_SWM, R_MWS, R_OTHER, R_LONG, R_SR, R_MSR };   // host/Settings.hpp RegionType
enum : uint32_t { A_NONE = 0, A_INTERNAL = 1, A_PREFIX = 2, A_SUFFIX = 3, A_EMPTY = 4 };
// Arms_settings / Minimizer_settings (include/globalDefs.hpp:110-156)
constexpr uint32_t kMinShortNum = 3, kMinInternal1 = 20, kMinInternal2 = 5, kMinInternal3 = 10, kMinContrib = 10, kShortArmCoef = 10, kMinimizerK = 10;
constexpr double kMinInternalContrib = 0.4;

__device__ __forceinline__ bool is_sr(uint8_t t) { return t == R_SR || t == R_MSR; }
__device__ __forceinline__ uint32_t base2(const uint8_t* p, uint32_t i) { return (p[i >> 2] >> (6 - 2 * (i & 3))) & 3u; }

// the k-mer starting at read position `ind` equals `target` (PackedSeq::check_kmer, src/PackedSeq.cpp:264-289; a 2-bit read has no N)
__device__ bool check_kmer(const uint8_t* rd, uint64_t target, uint32_t k, uint32_t ind) {
    uint64_t kmer = 0;
    for (uint32_t i = 0; i < k; ++i) kmer = (kmer << 2) | base2(rd, ind + i);
    return kmer == target;
}
// first / last start of `target` among the k-mers lying inside [left, right) (PackedSeq::find_kmer, src/PackedSeq.cpp:291-320)
__device__ bool find_kmer(const uint8_t* rd, uint64_t target, uint32_t k, uint32_t left, uint32_t right, bool first, uint32_t* hit) {
    if (left >= right) return false;
    const uint64_t mask = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    uint64_t kmer = 0; uint32_t len = 0; bool found = false;
    for (uint32_t i = left; i < right; ++i) {
        kmer = ((kmer << 2) | base2(rd, i)) & mask;
        if (len < k) ++len;
        if (len == k && kmer == target) { *hit = i - k + 1; found = true; if (first) break; }
    }
    return found;
}

// Alignment::prepare_short_arm: the arm [qb, qe) of `windex`, re-anchored on the k-mer / minimizer of a neighbouring SR / MSR
__device__ uint2 prepare_short_arm(const ArmsIn& I, const uint8_t* rd, uint32_t qae, uint32_t windex, uint32_t qb, uint32_t qe, uint32_t kind) {
    const uint32_t k = I.k, mk = kMinimizerK;
    const uint64_t curr = I.reg_start[windex], next = I.reg_start[windex + 1];
    if (next - curr > (uint64_t)kShortArmCoef * (uint64_t)(qe - qb)) return make_uint2(0, A_NONE);
    const uint8_t wt = I.reg_type[windex];
    bool valid = true;
    uint32_t q_beg = qb, q_end = qe, hit = 0;
    if ((wt == R_SWS || wt == R_SW || wt == R_SWM) && kind != A_SUFFIX) {                       // SR on the left
        if (q_beg < k) valid = false;
        else {
            const uint64_t anchor = I.anchor_kmers[(size_t)I.reg_info[windex - 1] << 1];         // last k-mer of that SR
            if (!check_kmer(rd, anchor, k, q_beg - k)) {
                const uint32_t s = q_beg < 2 * k ? 0 : q_beg - 2 * k, e = q_end < q_beg + k ? q_end : q_beg + k;
                if (find_kmer(rd, anchor, k, s, e, false, &hit)) q_beg = hit + k; else valid = false;
            }
        }
    }
    if ((wt == R_SWS || wt == R_WS || wt == R_MWS) && kind != A_PREFIX) {                        // SR on the right
        if (q_end + k > qae) valid = false;
        else {
            const uint64_t anchor = I.anchor_kmers[((size_t)I.reg_info[windex + 1] << 1) - 1];   // first k-mer of that SR
            if (!check_kmer(rd, anchor, k, q_end)) {
                const uint32_t s = q_end < q_beg + k ? q_beg : q_end - k, e = qae < q_end + 2 * k ? qae : q_end + 2 * k;
                if (find_kmer(rd, anchor, k, s, e, true, &hit)) q_end = hit; else valid = false;
            }
        }
    }
    if ((wt == R_MWM || wt == R_MW || wt == R_MWS) && kind != A_SUFFIX) {                        // minimizer on the left
        if (q_beg < mk) valid = false;
        else {
            const uint64_t mn = I.reg_info[windex - 1];
            if (!check_kmer(rd, mn, mk, q_beg - mk)) {
                const uint32_t s = q_beg < 3 * mk ? 0 : q_beg - 3 * mk, e = q_end < q_beg + 2 * mk ? q_end : q_beg + 2 * mk;
                if (find_kmer(rd, mn, mk, s, e, false, &hit)) q_beg = hit + mk; else valid = false;
            }
        }
    }
    if ((wt == R_MWM || wt == R_WM || wt == R_SWM) && kind != A_PREFIX) {                        // minimizer on the right
        if (q_end + mk > qae) valid = false;
        else {
            const uint64_t mn = I.reg_info[windex + 1];
            if (!check_kmer(rd, mn, mk, q_end)) {
                const uint32_t s = q_end < q_beg + 2 * mk ? q_beg : q_end - 2 * mk, e = qae < q_end + 3 * mk ? qae : q_end + 3 * mk;
                if (find_kmer(rd, mn, mk, s, e, true, &hit)) q_end = hit; else valid = false;
            }
        }
    }
    if (valid && q_beg < q_end) return make_uint2(q_beg, q_end | (kind << 28));
    return make_uint2(0, A_NONE);
}

// ---- Filter (include/Filter.hpp:33-102, include/MinimizerDeque.hpp): canonical (k = 10, w = 10) window minimizers -------------
// The reference keeps a monotone deque: a new k-mer pops larger keys off the back, entries older than w positions leave at the
// front, and the front is the window's minimizer.  Equal keys stay, so the front is the EARLIEST of the smallest keys among the
// k-mers pushed at positions i - w + 1 .. i: a register file of the last w keys (0xffffffff where nothing was pushed: an N, or
// fewer than k bases since one) gives the same (key, position) without a queue.  A k-mer of 10 bases fits 20 bits.
constexpr uint32_t kFilterK = 10, kFilterW = 10, kFilterBpPerMinimizer = 50, kNoKey = 0xffffffffu;
struct MinimizerScan {
    uint32_t key[kFilterW];
    uint32_t fwd = 0, rev = 0, run = 0, processed = 0;
    __device__ MinimizerScan() { for (uint32_t j = 0; j < kFilterW; ++j) key[j] = kNoKey; }
    // next base (c < 4) or an N (c >= 4) at position i; true: the window has a minimizer, (*mkey, *mpos)
    __device__ bool step(uint32_t c, uint32_t i, uint32_t* mkey, uint32_t* mpos) {
#pragma unroll
        for (int j = kFilterW - 1; j >= 1; --j) key[j] = key[j - 1];
        key[0] = kNoKey;
        if (c >= 4) { run = 0; return false; }            // (the rolling k-mers and `processed` are NOT reset: Filter.hpp:48-65)
        ++run;
        fwd = ((fwd << 2) | c) & ((1u << (2 * kFilterK)) - 1u);
        rev = (rev >> 2) | ((3u ^ c) << (2 * (kFilterK - 1)));
        if (run < kFilterK) return false;
        key[0] = fwd < rev ? fwd : rev;
        if (++processed < kFilterW) return false;
        uint32_t best = key[0], at = 0;
#pragma unroll
        for (uint32_t j = 1; j < kFilterW; ++j) if (key[j] <= best) { best = key[j]; at = j; }      // <=: the earliest of equal keys
        *mkey = best; *mpos = i - at;
        return true;
    }
};
// Filter::is_good for the arm read[qb, qe) against the sorted minimizers of a window's draft
__device__ bool arm_is_good(const uint8_t* rd, uint32_t qb, uint32_t qe, const uint32_t* dmin, uint32_t n_dmin) {
    const uint32_t len = qe - qb;
    if (len == 0) return true;                               // 0 * 50 >= 0
    MinimizerScan ms;
    uint32_t hits = 0, last_pos = 0xffffffffu;
    for (uint32_t i = 0; i < len; ++i) {
        uint32_t k, p;
        if (!ms.step(base2(rd, qb + i), i, &k, &p) || p == last_pos) continue;      // found_minimizers: one entry per minimizer position
        last_pos = p;
        uint32_t lo = 0, hi = n_dmin;
        while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (dmin[m] < k) lo = m + 1; else hi = m; }
        hits += (lo < n_dmin && dmin[lo] == k) ? 1u : 0u;
    }
    return (uint64_t)hits * kFilterBpPerMinimizer >= len;
}
}  // namespace

// ---- long mode: Filter::initialise per LONG window ------------------------------------------------------------------------
__global__ void __launch_bounds__(T) arms_minlen_kernel(ArmsIn I, ArmsOut O) {
    const uint32_t w = blockIdx.x * T + threadIdx.x;
    if (w >= I.n_regions) return;
    O.reg_min_len[w] = I.reg_type[w] == R_LONG ? I.reg_start[w + 1] - I.reg_start[w] : 0u;
}
__global__ void __launch_bounds__(T) arms_draftmin_kernel(ArmsIn I, ArmsOut O) {
    const uint32_t w = blockIdx.x * T + threadIdx.x;
    if (w >= I.n_regions) return;
    if (I.reg_type[w] != R_LONG) { O.reg_min_cnt[w] = 0; return; }
    const uint32_t ws = I.reg_start[w], len = I.reg_start[w + 1] - ws;
    uint32_t* out = O.draft_min + O.reg_min_off[w];
    MinimizerScan ms;
    uint32_t n = 0;
    for (uint32_t i = 0; i < len; ++i) {
        const uint32_t p = ws + i;
        uint32_t c = (I.contig4[p >> 1] >> (4 - 4 * (p & 1))) & 15u;      // PackedSeq<4>: A C G T = 0..3, anything else reads as N
        uint32_t k, pos;
        if (ms.step(c, i, &k, &pos) && (n == 0 || out[n - 1] != k)) out[n++] = k;     // (the set takes a key once; neighbours repeat)
    }
    for (uint32_t a = 1; a < n; ++a) {                        // sort (about len / 5 keys)
        const uint32_t v = out[a];
        uint32_t b = a;
        while (b > 0 && out[b - 1] > v) { out[b] = out[b - 1]; --b; }
        out[b] = v;
    }
    uint32_t m = 0;
    for (uint32_t a = 0; a < n; ++a) if (m == 0 || out[m - 1] != out[a]) out[m++] = out[a];
    O.reg_min_cnt[w] = m;
}

// ---- 1. regions an alignment touches (Alignment.cpp:222-227: rank on the region bit vector = binary search on the starts) ----
__global__ void __launch_bounds__(T) arms_span_kernel(ArmsIn I, uint32_t* __restrict__ b_ind, uint32_t* __restrict__ ntouch, uint32_t* __restrict__ bad) {
    const uint32_t a = blockIdx.x * T + threadIdx.x;
    if (a >= I.n_alignments) return;
    const uint32_t rb = I.rb[a], re = I.re[a];
    {   // the record must be consistent (Alignment::initialise_pos derives re and the aligned length from this very CIGAR):
        // the walk below indexes the read with positions it computes from the CIGAR
        uint64_t qsum = 0, rsum = 0;
        for (uint32_t ci = I.cigar_off[a]; ci < I.cigar_off[a + 1]; ++ci) {
            const uint32_t c = I.cigar[ci], op = c & 0xf, len = c >> 4;
            if (op == 4 || op == 5) continue;
            const uint32_t t = (0x3C1A7u >> (op << 1)) & 3u;
            if (t & 1) qsum += len;
            if (t & 2) rsum += len;
        }
        if (qsum != I.qae[a] || rsum != (uint64_t)re - rb || re <= rb || re > I.reg_start[I.n_regions] || (a && I.rb[a - 1] > rb)) {
            atomicAdd(bad, 1u); b_ind[a] = 0; ntouch[a] = 0; return;
        }
    }
    uint32_t lo = 0, hi = I.n_regions + 1;                   // b = (number of starts <= rb) - 1
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (I.reg_start[m] <= rb) lo = m + 1; else hi = m; }
    const uint32_t b = lo ? lo - 1 : 0;
    lo = 0; hi = I.n_regions + 1;                            // e = number of starts < re
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (I.reg_start[m] < re) lo = m + 1; else hi = m; }
    const uint32_t e = lo;
    b_ind[a] = b;
    ntouch[a] = e > b + 1 ? e - b : 0u;                      // a read inside one region contributes nothing (:227)
}

// ---- 2. CIGAR walk + candidate arms, one lane per alignment ----------------------------------------------------------
__global__ void __launch_bounds__(T) arms_walk_kernel(ArmsIn I, const uint32_t* __restrict__ b_ind, const uint32_t* __restrict__ ntouch,
                                                      const uint64_t* __restrict__ touch_off, uint32_t* __restrict__ bp, uint2* __restrict__ cand, ArmsOut O) {
    const uint32_t a = blockIdx.x * T + threadIdx.x;
    if (a >= I.n_alignments) return;
    const uint32_t nt = ntouch[a];
    if (nt == 0) return;
    const uint32_t beg = b_ind[a], end = beg + nt;
    const uint64_t t0 = touch_off[a];
    const uint32_t rb = I.rb[a], re = I.re[a], qae = I.qae[a];
    uint32_t* mybp = bp + t0;                                // nt - 1 break points
    // Alignment::find_bp (Alignment.cpp:321-406).  cigar_type: bit 0 consumes the query, bit 1 the reference (htslib bam_cigar_type)
    {
        uint32_t n = 0, ref_pos = rb, cur = beg + 1, query_pos = 0;
        uint32_t next_ref = I.reg_start[cur];
        bool corner = false;
        const uint32_t c0 = I.cigar_off[a], c1 = I.cigar_off[a + 1];
        for (uint32_t ci = c0; ci < c1 && cur != end; ++ci) {
            const uint32_t c = I.cigar[ci], op = c & 0xf;
            uint32_t len = c >> 4;
            if (op == 4 || op == 5) continue;                // S, H
            const uint32_t t = (0x3C1A7u >> (op << 1)) & 3u;
            if (t & 2) {                                     // M = X (both) or D N (reference only)
                const bool both = (t & 3) == 3;
                if (corner) { if (n < nt - 1) mybp[n] = query_pos; ++n; corner = false; ++cur; next_ref = I.reg_start[cur < I.n_regions + 1 ? cur : I.n_regions]; }
                while (cur != end && ref_pos + len >= next_ref && !corner) {
                    const uint32_t d = next_ref - ref_pos;
                    ref_pos = next_ref;
                    if (both) query_pos += d;
                    len -= d;
                    if (len > 0) { if (n < nt - 1) mybp[n] = query_pos; ++n; ++cur; next_ref = I.reg_start[cur < I.n_regions + 1 ? cur : I.n_regions]; }
                    else corner = true;
                }
                if (len > 0) { ref_pos += len; if (both) query_pos += len; }
            } else if (t & 1) {                              // I (S was skipped): insertion
                if (corner) {
                    const uint8_t lt = I.reg_type[cur - 1];
                    if (n < nt - 1) mybp[n] = is_sr(lt) ? query_pos : query_pos + len;
                    ++n; ++cur; next_ref = I.reg_start[cur < I.n_regions + 1 ? cur : I.n_regions];
                    corner = false;
                }
                query_pos += len;
            }
        }
        for (; n < nt - 1; ++n) mybp[n] = qae;              // (a CIGAR shorter than its span: never on consistent input)
    }
    const uint8_t* rd = I.reads2 + I.seq_off[a];
    uint2* mc = cand + t0;
    if (I.long_mode) {
        // Alignment::find_long_arms (Alignment.cpp:262-299): the read cut at the borders of the pseudo regions, nothing re-anchored;
        // Window::add_* keeps an arm of a LONG window only if Filter::is_good says so (include/Window.hpp:66-101).  A first /
        // last arm of zero length is still an arm (is_good("") holds); between two equal break points the window gets an
        // EMPTY arm, which no filter sees.
        auto long_arm = [&](uint32_t ind, uint32_t qb, uint32_t qe, uint32_t kind) -> uint2 {
            if (I.reg_type[ind] != R_LONG) return make_uint2(0, A_NONE);
            if (!arm_is_good(rd, qb, qe, O.draft_min + O.reg_min_off[ind], O.reg_min_cnt[ind])) return make_uint2(0, A_NONE);
            return make_uint2(qb, qe | (kind << 28));
        };
        mc[0] = long_arm(beg, 0, mybp[0], I.reg_start[beg] == rb ? A_INTERNAL : A_SUFFIX);
        for (uint32_t i = 1; i + 1 < nt; ++i) {
            const uint32_t ind = beg + i;
            if (I.reg_type[ind] != R_LONG) mc[i] = make_uint2(0, A_NONE);
            else if (mybp[i] == mybp[i - 1]) mc[i] = make_uint2(0, A_EMPTY << 28);
            else mc[i] = long_arm(ind, mybp[i - 1], mybp[i], A_INTERNAL);
        }
        mc[nt - 1] = long_arm(end - 1, mybp[nt - 2], qae, I.reg_start[end] == re ? A_INTERNAL : A_PREFIX);
        return;
    }
    // Alignment::find_short_arms (Alignment.cpp:228-258)
    {
        const uint32_t kind = I.reg_start[beg] == rb ? A_INTERNAL : A_SUFFIX;
        mc[0] = is_sr(I.reg_type[beg]) ? make_uint2(0, A_NONE) : prepare_short_arm(I, rd, qae, beg, 0, mybp[0], kind);
    }
    for (uint32_t i = 1; i + 1 < nt; ++i) {
        const uint32_t ind = beg + i;
        if (is_sr(I.reg_type[ind])) { mc[i] = make_uint2(0, A_NONE); continue; }
        if (mybp[i] == mybp[i - 1]) mc[i] = make_uint2(0, A_EMPTY << 28);
        else mc[i] = prepare_short_arm(I, rd, qae, ind, mybp[i - 1], mybp[i], A_INTERNAL);
    }
    {
        const uint32_t kind = I.reg_start[end] == re ? A_INTERNAL : A_PREFIX;
        mc[nt - 1] = is_sr(I.reg_type[end - 1]) ? make_uint2(0, A_NONE) : prepare_short_arm(I, rd, qae, end - 1, mybp[nt - 2], qae, kind);
    }
}

// ---- 3./4. one lane per region: its arms in alignment order; WRITE = false counts and prunes, WRITE = true emits the batch ----
template <bool WRITE>
__global__ void __launch_bounds__(T) arms_window_kernel(ArmsIn I, const uint32_t* __restrict__ b_ind, const uint32_t* __restrict__ ntouch,
                                                        const uint64_t* __restrict__ touch_off, const uint2* __restrict__ cand, ArmsOut O) {
    const uint32_t w = blockIdx.x * T + threadIdx.x;
    if (w >= I.n_regions) return;
    if (is_sr(I.reg_type[w]) || (I.long_mode && I.reg_type[w] != R_LONG)) { if (!WRITE) { O.reg_flags[w] = 0; O.reg_valid[w] = 0; O.reg_arms[w] = 0; O.reg_bytes[w] = 0; O.reg_draft_bytes[w] = 0; O.reg_slot[w] = 0; } return; }
    if (WRITE && !(O.reg_flags[w] & 1)) return;
    const uint32_t ws = I.reg_start[w], we = I.reg_start[w + 1];
    // alignments that can overlap [ws, we): start position in [ws - max_span, we)   (the starts are sorted: file order)
    uint32_t lo = 0, hi = I.n_alignments;
    const uint32_t from = ws > I.max_span ? ws - I.max_span : 0;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (I.rb[m] < from) lo = m + 1; else hi = m; }
    const uint32_t a_lo = lo;
    lo = a_lo; hi = I.n_alignments;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (I.rb[m] < we) lo = m + 1; else hi = m; }
    const uint32_t a_hi = lo;
    uint32_t n_int = 0, n_pre = 0, n_suf = 0, n_empty = 0, max_pre = 0, max_suf = 0, max_int = 0, bytes_int = 0, bytes_ps = 0;
    // write cursors (WRITE): internal arms first, then prefix, then suffix, each group in alignment order (include/hypo_gpu.h)
    uint64_t arm_i = 0, arm_p = 0, arm_s = 0, byte_i = 0, byte_p = 0, byte_s = 0;
    bool keep_ps = true;
    if (WRITE) {
        const uint32_t flags = O.reg_flags[w];
        keep_ps = (flags & 2) != 0;
        const HypoWindow* hw = O.windows + O.win_index[w];
        arm_i = hw->first_arm; arm_p = arm_i + hw->n_internal; arm_s = arm_p + hw->n_prefix;
        byte_i = O.reg_byte_off[w]; byte_p = byte_i + O.reg_bytes_int[w]; byte_s = byte_p + O.reg_bytes_pre[w];
    }
    uint32_t bytes_pre = 0;
    for (uint32_t a = a_lo; a < a_hi; ++a) {
        const uint32_t nt = ntouch[a], b = b_ind[a];
        if (nt == 0 || w < b || w >= b + nt) continue;
        const uint2 c = cand[touch_off[a] + (w - b)];
        const uint32_t kind = c.y >> 28;
        if (kind == A_NONE) continue;
        if (kind == A_EMPTY) { ++n_empty; continue; }
        const uint32_t qb = c.x, qe = c.y & 0x0fffffffu, len = qe - qb, nb = (len + 3) >> 2;
        if (kind == A_INTERNAL) { ++n_int; bytes_int += nb; max_int = len > max_int ? len : max_int; }
        else if (kind == A_PREFIX) { ++n_pre; max_pre = len > max_pre ? len : max_pre; bytes_pre += nb; }
        else { ++n_suf; max_suf = len > max_suf ? len : max_suf; bytes_ps += nb; }
        if (WRITE) {
            if (kind != A_INTERNAL && !keep_ps) continue;
            uint64_t& ai = kind == A_INTERNAL ? arm_i : (kind == A_PREFIX ? arm_p : arm_s);
            uint64_t& bo = kind == A_INTERNAL ? byte_i : (kind == A_PREFIX ? byte_p : byte_s);
            if (I.file_rank) {
                // records of an unsorted file, sorted by position on ingest: the arm's place in its group is the number of the
                // window's arms of that kind that come EARLIER IN THE FILE (a few dozen candidates per window: counted, not sorted)
                const uint32_t my = I.file_rank[a];
                uint64_t before = 0, bytes_before = 0;
                for (uint32_t a2 = a_lo; a2 < a_hi; ++a2) {
                    const uint32_t nt2 = ntouch[a2], b2 = b_ind[a2];
                    if (a2 == a || nt2 == 0 || w < b2 || w >= b2 + nt2 || I.file_rank[a2] >= my) continue;
                    const uint2 c2 = cand[touch_off[a2] + (w - b2)];
                    if ((c2.y >> 28) != kind) continue;
                    ++before; bytes_before += (((c2.y & 0x0fffffffu) - c2.x) + 3) >> 2;
                }
                const HypoWindow* hw = O.windows + O.win_index[w];
                const uint64_t g0 = kind == A_INTERNAL ? hw->first_arm : (kind == A_PREFIX ? (uint64_t)hw->first_arm + hw->n_internal : (uint64_t)hw->first_arm + hw->n_internal + hw->n_prefix);
                const uint64_t y0 = kind == A_INTERNAL ? O.reg_byte_off[w] : (kind == A_PREFIX ? O.reg_byte_off[w] + O.reg_bytes_int[w] : O.reg_byte_off[w] + O.reg_bytes_int[w] + O.reg_bytes_pre[w]);
                ai = g0 + before; bo = y0 + bytes_before;
            }
            O.arm_len[ai] = len;
            O.arm_off[ai] = bo;
            // PackedSeq<2>(ps, left, right): bases qb.. repacked from bit 7 of a fresh byte (src/PackedSeq.cpp:91-139)
            const uint8_t* rd = I.reads2 + I.seq_off[a];
            uint8_t* dst = O.arms2 + bo;
            for (uint32_t j = 0; j < nb; ++j) {
                uint32_t v = 0;
                for (uint32_t t = 0; t < 4; ++t) { const uint32_t q = 4 * j + t; v = (v << 2) | (q < len ? base2(rd, qb + q) : 0u); }
                dst[j] = (uint8_t)v;
            }
            ++ai; bo += nb;
        }
    }
    if (!WRITE) {
        // Contig::fill_short_windows (src/Contig.cpp:264-288): windows with too few arms are dropped, prefix / suffix arms are
        // thrown away where enough internal arms exist.  get_num_internal() counts empty arms too (include/Window.hpp:107).
        const uint32_t internal = n_int + n_empty;
        bool valid = true;
        if (I.long_mode) {
            // Contig::fill_long_windows (include/Contig.hpp:91-113): every LONG window stays; above min_internal_num3 internal arms
            // (empty ones count) the prefix / suffix arms go
            const bool clear_ps = internal > kMinInternal3;
            const uint32_t np = clear_ps ? 0 : n_pre, ns = clear_ps ? 0 : n_suf;
            O.reg_flags[w] = 1u | (clear_ps ? 0u : 2u);
            O.reg_counts[w] = make_uint4(n_int, np, ns, n_empty);
            O.reg_arms[w] = n_int + np + ns;
            O.reg_bytes_int[w] = bytes_int;
            O.reg_bytes_pre[w] = clear_ps ? 0 : bytes_pre;
            O.reg_bytes[w] = bytes_int + (clear_ps ? 0 : bytes_pre + bytes_ps);
            O.reg_draft_bytes[w] = (we - ws + 1) / 2;
            uint32_t longest = we - ws;
            longest = max_int > longest ? max_int : longest;
            if (!clear_ps) { longest = max_pre > longest ? max_pre : longest; longest = max_suf > longest ? max_suf : longest; }
            O.reg_slot[w] = (longest + longest / 2 + 24 + 7) / 8 * 8;
            O.reg_valid[w] = 1u;
            return;
        }
        if (internal < kMinShortNum) {
            const bool covered = max_pre + max_suf >= we - ws;
            const bool enough = n_pre >= kMinShortNum && n_suf >= kMinShortNum;
            valid = covered && enough;
        }
        bool clear_ps = false;
        if (valid) {
            const uint32_t contrib = internal + n_pre + n_suf;
            const bool c0 = internal > kMinInternal1;
            const bool c1 = contrib >= kMinContrib && (double)internal >= floor(kMinInternalContrib * (double)contrib);
            const uint8_t t = I.reg_type[w];
            const bool c2 = (t == R_SWS || t == R_SW || t == R_WS || t == R_MWS || t == R_SWM) && internal >= kMinInternal2;
            clear_ps = c0 || c1 || c2;
        }
        const uint32_t np = clear_ps ? 0 : n_pre, ns = clear_ps ? 0 : n_suf;
        O.reg_flags[w] = (valid ? 1u : 0u) | (clear_ps ? 0u : 2u);
        O.reg_counts[w] = make_uint4(n_int, np, ns, n_empty);
        O.reg_arms[w] = valid ? n_int + np + ns : 0;
        O.reg_bytes_int[w] = bytes_int;
        O.reg_bytes_pre[w] = clear_ps ? 0 : bytes_pre;
        O.reg_bytes[w] = valid ? bytes_int + (clear_ps ? 0 : bytes_pre + bytes_ps) : 0;
        O.reg_draft_bytes[w] = valid ? (we - ws + 1) / 2 : 0;
        uint32_t longest = we - ws;                         // consensus slot of the window (hypo_gpu_poa_slot_layout's rule)
        longest = max_int > longest ? max_int : longest;
        if (!clear_ps) { longest = max_pre > longest ? max_pre : longest; longest = max_suf > longest ? max_suf : longest; }
        O.reg_slot[w] = valid ? (longest + longest / 2 + 24 + 7) / 8 * 8 : 0;
        O.reg_valid[w] = valid ? 1u : 0u;
    }
}

// ---- 5. window descriptors + drafts (PackedSeq<4> slice of the contig, re-aligned to a byte boundary) --------------------
__global__ void __launch_bounds__(T) arms_describe_kernel(ArmsIn I, ArmsOut O) {
    const uint32_t w = blockIdx.x * T + threadIdx.x;
    if (w >= I.n_regions || !(O.reg_flags[w] & 1)) return;
    const uint32_t ws = I.reg_start[w], we = I.reg_start[w + 1], len = we - ws;
    const uint4 c = O.reg_counts[w];
    HypoWindow hw;
    hw.type = I.long_mode ? HYPO_WIN_LONG : HYPO_WIN_SHORT; hw.reserved[0] = hw.reserved[1] = hw.reserved[2] = 0;
    hw.draft_len = len; hw.draft_off = O.reg_draft_off[w];
    hw.first_arm = (uint32_t)O.reg_arm_off[w];
    hw.n_internal = c.x; hw.n_prefix = c.y; hw.n_suffix = c.z; hw.n_empty = c.w; hw.reserved2 = 0;
    O.windows[O.win_index[w]] = hw;
    O.win_region[O.win_index[w]] = w;
    uint8_t* dst = O.draft4 + hw.draft_off;
    const uint8_t* src = I.contig4;
    for (uint32_t j = 0; j < (len + 1) / 2; ++j) {
        const uint32_t p = ws + 2 * j;
        const uint32_t hi = (src[p >> 1] >> (4 - 4 * (p & 1))) & 15u;
        const uint32_t lo = 2 * j + 1 < len ? (src[(p + 1) >> 1] >> (4 - 4 * ((p + 1) & 1))) & 15u : 0u;
        dst[j] = (uint8_t)((hi << 4) | lo);
    }
    O.out_off[O.win_index[w]] = O.reg_slot_off[w];
}

// ---- exclusive prefix sums of per-region u32 values (three small kernels each) ---------------------------------------------
namespace {
constexpr int S_ITEMS = 1024;
__global__ void __launch_bounds__(T) scan32_partial(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ bsum) {
    __shared__ uint32_t red[T / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * S_ITEMS;
    uint32_t s = 0;
    for (int i = threadIdx.x; i < S_ITEMS; i += T) { const uint64_t a = b0 + i; if (a < n) s += in[a]; }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t t = 0; for (int i = 0; i < T / 64; ++i) t += red[i]; bsum[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(1024) scan32_blocksums(uint64_t* __restrict__ bsum, uint64_t n_blocks, uint64_t* __restrict__ total) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t c0 = 0; c0 < n_blocks; c0 += 1024) {
        const uint64_t i = c0 + threadIdx.x;
        const uint64_t v = i < n_blocks ? bsum[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint64_t add = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        const uint64_t incl = sh[threadIdx.x];
        if (i < n_blocks) bsum[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(T) scan32_final(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ bsum, uint64_t* __restrict__ out) {
    __shared__ uint32_t wsum[T / 64];
    const uint64_t a0 = (uint64_t)blockIdx.x * S_ITEMS + (uint64_t)threadIdx.x * 4;
    uint32_t c[4], mine = 0;
    for (int i = 0; i < 4; ++i) { c[i] = a0 + i < n ? in[a0 + i] : 0; mine += c[i]; }
    uint32_t inc = mine;
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) woff += wsum[i];
    uint64_t run = bsum[blockIdx.x] + woff + (inc - mine);
    for (int i = 0; i < 4; ++i) { if (a0 + i < n) out[a0 + i] = run; run += c[i]; }
}
__global__ void __launch_bounds__(T) win_index_kernel(const uint64_t* __restrict__ off, uint32_t n, uint32_t* __restrict__ idx) {
    const uint32_t w = blockIdx.x * T + threadIdx.x;
    if (w < n) idx[w] = (uint32_t)off[w];
}
}  // namespace

hipError_t scan32(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* bsum, uint64_t* total, hipStream_t st) {
    const uint64_t nb = (n + S_ITEMS - 1) / S_ITEMS;
    if (!n) return hipMemsetAsync(total, 0, 8, st);
    hipLaunchKernelGGL(scan32_partial, dim3((unsigned)nb), dim3(T), 0, st, in, n, bsum);
    hipLaunchKernelGGL(scan32_blocksums, dim3(1), dim3(1024), 0, st, bsum, nb, total);
    hipLaunchKernelGGL(scan32_final, dim3((unsigned)nb), dim3(T), 0, st, in, n, bsum, out);
    return hipGetLastError();
}
size_t scan32_scratch_bytes(uint64_t n) { return (((n + S_ITEMS - 1) / S_ITEMS) * 8 + 255) / 256 * 256 + 256; }

hipError_t arms_phase1(const ArmsIn& I, uint32_t* b_ind, uint32_t* ntouch, uint32_t* bad, hipStream_t st) {
    if (!I.n_alignments) return hipSuccess;
    hipLaunchKernelGGL(arms_span_kernel, dim3((I.n_alignments + T - 1) / T), dim3(T), 0, st, I, b_ind, ntouch, bad);
    return hipGetLastError();
}
hipError_t arms_phase2(const ArmsIn& I, const uint32_t* b_ind, const uint32_t* ntouch, const uint64_t* touch_off, uint32_t* bp, uint2* cand,
                       const ArmsOut& O, hipStream_t st) {
    if (I.n_alignments) hipLaunchKernelGGL(arms_walk_kernel, dim3((I.n_alignments + T - 1) / T), dim3(T), 0, st, I, b_ind, ntouch, touch_off, bp, cand, O);
    hipLaunchKernelGGL(arms_window_kernel<false>, dim3((I.n_regions + T - 1) / T), dim3(T), 0, st, I, b_ind, ntouch, touch_off, cand, O);
    return hipGetLastError();
}
hipError_t arms_long_minlen(const ArmsIn& I, const ArmsOut& O, hipStream_t st) {
    hipLaunchKernelGGL(arms_minlen_kernel, dim3((I.n_regions + T - 1) / T), dim3(T), 0, st, I, O);
    return hipGetLastError();
}
hipError_t arms_long_draftmin(const ArmsIn& I, const ArmsOut& O, hipStream_t st) {
    hipLaunchKernelGGL(arms_draftmin_kernel, dim3((I.n_regions + T - 1) / T), dim3(T), 0, st, I, O);
    return hipGetLastError();
}
hipError_t arms_phase3(const ArmsIn& I, const uint32_t* b_ind, const uint32_t* ntouch, const uint64_t* touch_off, const uint2* cand,
                       const ArmsOut& O, const uint64_t* win_off, hipStream_t st) {
    hipLaunchKernelGGL(win_index_kernel, dim3((I.n_regions + T - 1) / T), dim3(T), 0, st, win_off, I.n_regions, O.win_index);
    hipLaunchKernelGGL(arms_describe_kernel, dim3((I.n_regions + T - 1) / T), dim3(T), 0, st, I, O);
    hipLaunchKernelGGL(arms_window_kernel<true>, dim3((I.n_regions + T - 1) / T), dim3(T), 0, st, I, b_ind, ntouch, touch_off, cand, O);
    return hipGetLastError();
}

}  // namespace hypo
