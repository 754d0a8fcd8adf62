// scan_kernel.hip — solid-kmer scan of a 4-bit packed contig (replaces Contig::find_solid_pos,
// src/Contig.cpp:40-74, and suk::SolidKmers::is_solid, external/suk/include/suk/SolidKmers.hpp:119).
//
// MI355X mapping: HBM/L2-bound integer work, no MFMA.  ONE kernel (scan_fused_kernel) behind one small memset:
//   * a 256-lane workgroup takes the next tile of 4 096 positions (tiles = workgroups in launch order), stages its 2 KiB of packed
//     bases (+ 1 byte before, 16 after) into LDS with coalesced 16-byte loads; every lane rolls the 2-bit k-mer over its 16
//     positions (+ k - 1 bases) out of LDS and probes the 4^k-bit solid set for each (random 1-bit gathers, all in flight:
//     L2-resident for k <= 13 = 8 MiB, Infinity-Cache-resident for k = 15 = 128 MiB, HBM sectors for k = 17 = 2 GiB), with the
//     reference's two homopolymer-edge tests; four lanes put their 16 mark bits together into one output word;
//   * the rank directory behind the reference's sdsl rank/select support (Contig.cpp:72-73) is an exclusive prefix sum of the
//     words' popcounts: inside the tile by one wave; across tiles through a status word per tile and one per group of 256 tiles
//     (agent-scope atomics): a tile adds up the counts of its group's earlier tiles and the totals of the earlier groups — two
//     levels of waiting, no chain, whatever the contig's length;
//   * every marked position then re-reads its k bases from the tile's LDS copy and writes the k-mer id at its rank
//     (Contig::_kmerinfo order, Contig.cpp:67-68).
// Round 2 had five launches here (mark, three for the prefix sum, k-mer ids): 94 us for the 5 Mbp contig of the C2 batch, most of
// it the gaps between launches and a mark kernel with 1.2 workgroups per CU.
// The 4^k-bit set is NOT staged through LDS for the survey's configurations: k=11 needs 512 KiB,
// more than the 160 KiB of a CU, while it fits every XCD's 4 MiB L2.
#include <hip/hip_runtime.h>
#include "scan_kernel.hpp"

namespace hypo {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_POS_PER_LANE = 16;
constexpr int SCAN_POS_PER_BLOCK = SCAN_THREADS * SCAN_POS_PER_LANE;   // 4096 positions
constexpr int SCAN_BYTES_PER_BLOCK = SCAN_POS_PER_BLOCK / 2;            // 2048 bytes
constexpr int SCAN_WORDS_PER_BLOCK = SCAN_POS_PER_BLOCK / 64;           // 64 output words
constexpr int SCAN_LDS_BYTES = SCAN_BYTES_PER_BLOCK + 32;               // + 1 byte before, k+1 bases after
constexpr uint64_t SCAN_GROUP = SCAN_THREADS;                              // tiles per group of the two-level prefix (one tile per lane)
constexpr uint32_t kScanSpinLimit = 1u << 20;                               // polls of a status word (about a microsecond each) before a tile gives up
constexpr uint64_t ST_AGGREGATE = 1ull << 62, ST_VALUE = (1ull << 62) - 1ull;   // status word of a tile

// Stages the workgroup's 2 KiB of packed bases (+ one byte before, 16 after) into LDS: sb[16 + x] = packed4[blk_byte0 + x] for
// x in [-1, 2048 + 16); out of range -> 0x44 ("NN").
__device__ __forceinline__ void stage_block(const uint8_t* __restrict__ packed4, uint64_t n_bytes, uint64_t blk_byte0, uint8_t* sb) {
    const int t = threadIdx.x;
    for (int c = t; c < SCAN_BYTES_PER_BLOCK / 16 + 1; c += SCAN_THREADS) {
        const uint64_t gb = blk_byte0 + (uint64_t)c * 16;
        uint4 v;
        if (gb + 16 <= n_bytes && ((uintptr_t)(packed4 + gb) & 15) == 0) {
            v = *(const uint4*)(packed4 + gb);
        } else {
            uint8_t tmp[16];
            for (int b = 0; b < 16; ++b) tmp[b] = (gb + b < n_bytes) ? packed4[gb + b] : (uint8_t)0x44;
            v = *(uint4*)tmp;
        }
        *(uint4*)(sb + 16 + c * 16) = v;
    }
    if (t == 0) sb[15] = blk_byte0 > 0 ? packed4[blk_byte0 - 1] : (uint8_t)0x44;
}

// The k-mers that START at the lane's P positions (local nibble indices l0 .. l0 + P - 1 of the staged block), in position
// order: emit(p, kmer, ok) with p = 0..P-1, ok = the k-mer has no N, lies inside the contig and passes the reference's two
// homopolymer-edge tests (next base != last base, Contig.cpp:59; previous base != first base, Contig.cpp:63).  Bases come
// out of LDS 8 at a time (one 32-bit read); the 2-bit k-mer, the run length since the last N and a history of "equals its
// predecessor" bits roll in registers.  Fillers beyond the contig are N, so k-mers that would run over the end are never ok,
// and a filler never equals a real base; `n_left` = bases of the contig from the lane's first position on (the pad nibble of an
// odd-length contig is not a base).
template <int P, class Emit>
__device__ __forceinline__ void roll(const uint8_t* base, long l0, uint32_t k, int64_t n_left, Emit&& emit) {
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;     // k <= 31
    uint64_t kmer = 0;
    uint32_t klen = 0, eh = 0;
    unsigned pb = (base[(l0 - 1) >> 1] >> (4 - 4 * ((l0 - 1) & 1))) & 15u;        // base before the first position
    const uint8_t* src = base + (l0 >> 1);               // l0 is even: the lane's bases start on a byte
    const int last = P - 1 + (int)k;                     // step t looks at base t: the k-mer ending at t - 1 starts at t - k
    for (int w = 0; w * 8 <= last; ++w) {
        const uint32_t word = __builtin_bswap32(*(const uint32_t*)(src + 4 * w));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 8 * w + j;
            const unsigned b = t < n_left ? (word >> (28 - 4 * j)) & 15u : 4u;
            eh = (eh << 1) | (b == pb ? 1u : 0u);        // bit i: base t - i equals base t - i - 1
            const int p = t - (int)k;
            if (p >= 0 && p < P) {
                // reject: base t == base t - 1 (the k-mer's last), or base p == base p - 1 (bit t - p = k of the history)
                const bool ok = klen >= k && ((eh | (eh >> k)) & 1u) == 0;
                emit(p, kmer, ok);
            }
            if (b < 4) { kmer = ((kmer << 2) | b) & kmask; klen += klen < k ? 1u : 0u; }
            else { kmer = 0; klen = 0; }
            pb = b;
        }
    }
}

// 16 bases (one big-endian 64-bit word of nibbles) -> 32 bits, first base on top.  The nibbles of a marked k-mer are 0..3; the
// ones after it in the word may be N and are cut down to two bits so that they cannot spill into their neighbour.
__device__ __forceinline__ uint32_t squeeze16(uint64_t x) {
    x &= 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)x;
}


__device__ __forceinline__ uint64_t st_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// hdr[0] = number of marked positions (out), hdr[2] = 1 when a wait was abandoned; status[tile]; both zeroed before the launch.
__global__ void __launch_bounds__(SCAN_THREADS)
scan_fused_kernel(const uint8_t* __restrict__ packed4, uint64_t n_bases, uint32_t k, const uint64_t* __restrict__ bits,
                  uint64_t* __restrict__ words, uint64_t* __restrict__ word_rank, uint64_t* __restrict__ kids, uint64_t kids_cap,
                  uint64_t n_words, uint64_t n_tiles, uint64_t* __restrict__ hdr, uint64_t* __restrict__ status, uint64_t* __restrict__ n_solid,
                  uint32_t* __restrict__ kids32, uint32_t* __restrict__ spos) {
    __shared__ __attribute__((aligned(16))) uint8_t sb[SCAN_LDS_BYTES + 16];
    __shared__ uint32_t wex[SCAN_WORDS_PER_BLOCK];
    __shared__ uint32_t wc[SCAN_WORDS_PER_BLOCK];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_total;
    __shared__ uint64_t s_part[2 * SCAN_THREADS / 64];
    uint64_t* const gstatus = status + n_tiles;              // totals of the groups of SCAN_GROUP tiles, behind the tiles' own status words
    const int tid = threadIdx.x, lane = tid & 63;
    // Tiles are workgroups in launch order.  A tile waits for tiles with smaller indices only, and workgroups are dispatched in
    // index order, so what it waits for is resident or done.  (A ticket from one atomic counter would make that order explicit, as
    // rocPRIM's look-back scan does: 1 221 tiles of a 5 Mbp contig queueing on one address cost 25 us of a 44-us kernel.)  The waits
    // are bounded all the same: a tile that sees no progress for kScanSpinLimit polls gives up and raises hdr[2].
    const uint64_t tile = blockIdx.x;
    stage_block(packed4, (n_bases + 1) / 2, tile * SCAN_BYTES_PER_BLOCK, sb);
    __syncthreads();
    // ---- marks: 16 positions per lane, every probe issued without a branch in between (word 0 of the set where the k-mer is
    // not a candidate), so all of a lane's loads are in flight together
    const uint64_t pos0 = tile * SCAN_POS_PER_BLOCK + (uint64_t)tid * SCAN_POS_PER_LANE;
    uint32_t m = 0;
    if (pos0 < n_bases)
        roll<SCAN_POS_PER_LANE>(sb + 16, (long)tid * SCAN_POS_PER_LANE, k, (int64_t)(n_bases - pos0), [&](int p, uint64_t kmer, bool ok) {
            const uint64_t wd = bits[ok ? (kmer >> 6) : 0];
            m |= (uint32_t)(ok && ((wd >> (kmer & 63)) & 1ull)) << p;
        });
    // ---- four lanes -> one word
    const uint32_t m1 = __shfl_down(m, 1, 64), m2 = __shfl_down(m, 2, 64), m3 = __shfl_down(m, 3, 64);
    const int wl = tid >> 2;                                       // word of the tile
    const uint64_t wg = tile * SCAN_WORDS_PER_BLOCK + (uint64_t)wl;
    if ((tid & 3) == 0) {
        const uint64_t word = (uint64_t)m | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32) | ((uint64_t)m3 << 48);
        if (wg < n_words) words[wg] = word;
        wc[wl] = (uint32_t)__popcll(word);
    }
    __syncthreads();
    // ---- ranks: inside the tile (wave 0), then the tile's exclusive prefix over the tiles before it — without a chain.  Tiles
    // form groups of 256.  A tile publishes its own count (depends on nothing but its marks); the LAST tile of a group adds up the
    // group's counts and publishes the group's total; every tile then sums the counts of the tiles of its group before it (one
    // load per lane) and the totals of the groups before its own (n_tiles / 65 536 loads per lane): two levels of waiting whatever
    // the contig's length.  (Rounds 2-4 chained inclusive prefixes by a decoupled look-back once a contig had more than 4 096
    // tiles: the front of published prefixes advanced 64 tiles per round trip through L2, 0.4 of the 0.68 ms of a 100 Mbp scan and
    // 2 of the 6.4 ms of a 512 Mbp one.)  What a tile waits for was published by tiles with smaller indices, which were dispatched
    // before it: resident or done.
    if (tid < 64) {
        const uint32_t c = wc[tid];
        uint32_t inc = c;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        wex[tid] = inc - c;
        const uint32_t total = (uint32_t)__shfl((int)inc, 63, 64);
        if (lane == 0) { st_store(status + tile, ST_AGGREGATE | (uint64_t)total); s_total = total; }
    }
    {
        const uint64_t grp = tile / SCAN_GROUP, g0 = grp * SCAN_GROUP;
        auto wait_for = [&](const uint64_t* p) -> uint64_t {
            uint64_t st;
            uint32_t spins = 0;
            while (((st = st_load(p)) >> 62) == 0) {
                if (++spins > kScanSpinLimit) { hdr[2] = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            return st & ST_VALUE;
        };
        uint64_t vin = 0, vgr = 0;                            // counts of this group's tiles before this one; totals of the groups before
        if (g0 + (uint64_t)tid < tile) vin = wait_for(status + g0 + tid);
        for (int d = 32; d >= 1; d >>= 1) vin += __shfl_xor(vin, d, 64);
        if (lane == 0) s_part[tid >> 6] = vin;
        if (tile == g0 + SCAN_GROUP - 1) {                    // (block-uniform) the group's total goes out BEFORE this tile waits for the totals
            __syncthreads();                                  // of the groups before it: a group's total depends on its own tiles only, no chain
            if (tid == 0) st_store(gstatus + grp, ST_AGGREGATE | (s_part[0] + s_part[1] + s_part[2] + s_part[3] + (uint64_t)s_total));
        }
        for (uint64_t g = (uint64_t)tid; g < grp; g += SCAN_THREADS) vgr += wait_for(gstatus + g);
        for (int d = 32; d >= 1; d >>= 1) vgr += __shfl_xor(vgr, d, 64);
        if (lane == 0) s_part[4 + (tid >> 6)] = vgr;
        __syncthreads();
        if (tid == 0) s_prefix = s_part[0] + s_part[1] + s_part[2] + s_part[3] + s_part[4] + s_part[5] + s_part[6] + s_part[7];
        __syncthreads();
    }
    const uint64_t prefix = s_prefix;
    if ((tid & 3) == 0 && wg < n_words) word_rank[wg] = prefix + wex[wl];
    if (tile + 1 == n_tiles && tid == 0) {
        const uint64_t tot = prefix + s_total;
        word_rank[n_words] = tot;
        hdr[0] = tot;
        if (n_solid) *n_solid = tot;
    }
    // ---- k-mer ids of the lane's marked positions at their ranks, bases from the tile's LDS copy
    const uint32_t pc = (uint32_t)__popc(m);
    const uint32_t q1 = __shfl_up(pc, 1, 64), q2 = __shfl_up(pc, 2, 64), q3 = __shfl_up(pc, 3, 64);
    if ((!kids && !kids32) || !kids_cap || m == 0) return;
    const int sub = tid & 3;
    uint64_t r = prefix + wex[wl] + (sub >= 1 ? q1 : 0u) + (sub >= 2 ? q2 : 0u) + (sub >= 3 ? q3 : 0u);
    while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        if (r < kids_cap) {
            const int l = tid * SCAN_POS_PER_LANE + b;                          // nibble of the tile
            const uint8_t* q = sb + 16 + (l >> 1);
            uint64_t hi = 0, lo = 0;
            for (int i = 0; i < 8; ++i) { hi = (hi << 8) | q[i]; lo = (lo << 8) | q[8 + i]; }
            if (l & 1) { hi = (hi << 4) | (lo >> 60); lo = (lo << 4) | (uint64_t)(q[16] >> 4); }
            uint64_t kmer;
            if (k <= 16) kmer = (uint64_t)squeeze16(hi) >> (2 * (16 - k));
            else kmer = (((uint64_t)squeeze16(hi) << 32) | squeeze16(lo)) >> (2 * (32 - k));
            // (a resident scan, hypo_gpu_solid_scan_keep: k-mers of k <= 16 as 32-bit words, and the position next to each)
            if (kids32) kids32[r] = (uint32_t)kmer; else kids[r] = kmer;
            if (spos) spos[r] = (uint32_t)(tile * SCAN_POS_PER_BLOCK + (uint64_t)l);
        }
        ++r;
    }
}

size_t scan_workspace_bytes(uint64_t n_bases) {
    // (sized as in round 2, when the per-word counts and block sums of three prefix-sum kernels lived here: a caller's allocation
    // stays valid; the fused kernel needs 256 + 8 bytes per tile of 4 096 positions + the rank directory if the caller passes none)
    const uint64_t n_words = (n_bases + 63) / 64;
    const uint64_t n_blocks = (n_words + 1023) / 1024;
    size_t b = 256;
    b += (n_words * 4 + 255) / 256 * 256;
    b += (n_blocks * 8 + 255) / 256 * 256;
    b += (n_words + 1) * 8 + 256;
    return b;
}

hipError_t scan_run(const uint8_t* packed4, uint64_t n_bases, uint32_t k, const uint64_t* bits,
                    uint64_t* words, uint64_t* kids, uint64_t kids_cap, uint64_t* word_rank,
                    uint64_t* n_solid, void* workspace, size_t workspace_bytes, hipStream_t stream,
                    hipEvent_t* prof_ev, uint32_t* kids32, uint32_t* spos) {
    if (kids32 && k > 16) return hipErrorInvalidValue;
    if (workspace_bytes < scan_workspace_bytes(n_bases)) return hipErrorInvalidValue;
    const uint64_t n_words = (n_bases + 63) / 64;
    const uint64_t n_tiles = (n_words + SCAN_WORDS_PER_BLOCK - 1) / SCAN_WORDS_PER_BLOCK;
    char* ws = (char*)workspace;
    uint64_t* hdr = (uint64_t*)ws;
    uint64_t* status = (uint64_t*)(ws + 256);
    const size_t status_bytes = ((n_tiles + (n_tiles + SCAN_GROUP - 1) / SCAN_GROUP) * 8 + 255) / 256 * 256;   // tiles + groups; <= n_words * 4 rounded up (a tile has 64 words): inside round 2's wcount area
    uint64_t* own_rank = (uint64_t*)(ws + 256 + (n_words * 4 + 255) / 256 * 256 + ((n_words + 1023) / 1024 * 8 + 255) / 256 * 256);
    if (!word_rank) word_rank = own_rank;
    hipError_t e;
    if (n_words == 0) {
        if ((e = hipMemsetAsync(hdr, 0, 8, stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(word_rank, 0, 8, stream)) != hipSuccess) return e;
        if (n_solid) return hipMemsetAsync(n_solid, 0, 8, stream);
        return hipSuccess;
    }
    if (prof_ev) (void)hipEventRecord(prof_ev[0], stream);
    if ((e = hipMemsetAsync(ws, 0, 256 + status_bytes, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(scan_fused_kernel, dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, stream,
                       packed4, n_bases, k, bits, words, word_rank, (kids && kids_cap) ? kids : nullptr, kids_cap, n_words, n_tiles, hdr, status, n_solid,
                       kids_cap ? kids32 : nullptr, kids_cap ? spos : nullptr);
    if (prof_ev) { (void)hipEventRecord(prof_ev[1], stream); (void)hipEventRecord(prof_ev[2], stream); (void)hipEventRecord(prof_ev[3], stream); }
    return hipGetLastError();
}

}  // namespace hypo
