// scan_kernel.hip — solid-kmer scan of a 4-bit packed contig (replaces Contig::find_solid_pos,
// src/Contig.cpp:40-74, and suk::SolidKmers::is_solid, external/suk/include/suk/SolidKmers.hpp:119).
//
// MI355X mapping: HBM/L2-bound integer work, no MFMA.
//   pass 1  scan_mark_kernel   one lane owns 64 consecutive k-mer start positions (= one output word).
//           A 256-lane workgroup stages its 8 KiB of packed bases (+ k+1 bases of halo) into LDS with
//           coalesced 16-byte loads, every lane then rolls the 2-bit k-mer over its 64+k-1 bases out of
//           LDS and probes the 4^k-bit solid set (random 1-bit gathers: L2-resident for k<=13 = 8 MiB,
//           Infinity-Cache-resident for k=15 = 128 MiB, HBM sectors for k=17 = 2 GiB).  The reference's
//           two homopolymer-edge tests are applied, the lane writes its 64 mark bits as one coalesced
//           8-byte store and its popcount.
//   pass 2  scan_rank_*        exclusive prefix sum of the per-word popcounts = the rank directory
//           behind the reference's sdsl rank/select support (Contig.cpp:72-73).
//   pass 3  scan_kids_kernel   every marked position re-reads its k bases and writes the k-mer id at
//           rank order (Contig::_kmerinfo order, Contig.cpp:67-68).
// The 4^k-bit set is NOT staged through LDS for the survey's configurations: k=11 needs 512 KiB,
// more than the 160 KiB of a CU, while it fits every XCD's 4 MiB L2.
#include <hip/hip_runtime.h>
#include "scan_kernel.hpp"

namespace hypo {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_POS_PER_BLOCK = SCAN_THREADS * 64;             // 16384 positions
constexpr int SCAN_BYTES_PER_BLOCK = SCAN_POS_PER_BLOCK / 2;      // 8192 bytes
constexpr int SCAN_LDS_BYTES = SCAN_BYTES_PER_BLOCK + 32;         // + 1 byte before, k+1 bases after

__device__ __forceinline__ unsigned nib(const uint8_t* p, long i) { return (p[i >> 1] >> (4 - 4 * (i & 1))) & 15; }

// Stages the workgroup's 8 KiB of packed bases (+ one byte before, 16 after) into LDS: sb[16 + x] = packed4[blk_byte0 + x] for
// x in [-1, 8192 + 16); out of range -> 0x44 ("NN").
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

// The k-mers that START at the lane's 64 positions (local nibble indices l0 .. l0 + 63 of the staged block), in position order:
// emit(p, kmer, ok) with p = 0..63, ok = the k-mer has no N, lies inside the contig and passes the reference's two
// homopolymer-edge tests (next base != last base, Contig.cpp:59; previous base != first base, Contig.cpp:63).  Bases come
// out of LDS 8 at a time (one 32-bit read); the 2-bit k-mer, the run length since the last N and a history of "equals its
// predecessor" bits roll in registers.  Fillers beyond the contig are N, so k-mers that would run over the end are never ok,
// and a filler never equals a real base; `n_left` = bases of the contig from the lane's first position on (the pad nibble of an
// odd-length contig is not a base).
template <class Emit>
__device__ __forceinline__ void roll64(const uint8_t* base, long l0, uint32_t k, int64_t n_left, Emit&& emit) {
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;     // k <= 31
    uint64_t kmer = 0;
    uint32_t klen = 0, eh = 0;
    unsigned pb = (base[(l0 - 1) >> 1] >> (4 - 4 * ((l0 - 1) & 1))) & 15u;        // base before the first position
    const uint8_t* src = base + (l0 >> 1);               // l0 is even: the lane's bases start on a byte
    const int last = 63 + (int)k;                        // step t looks at base t: the k-mer ending at t - 1 starts at t - k
    for (int w = 0; w * 8 <= last; ++w) {
        const uint32_t word = __builtin_bswap32(*(const uint32_t*)(src + 4 * w));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 8 * w + j;
            const unsigned b = t < n_left ? (word >> (28 - 4 * j)) & 15u : 4u;
            eh = (eh << 1) | (b == pb ? 1u : 0u);        // bit i: base t - i equals base t - i - 1
            const int p = t - (int)k;
            if (p >= 0 && p < 64) {
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

__global__ void __launch_bounds__(SCAN_THREADS)
scan_mark_kernel(const uint8_t* __restrict__ packed4, uint64_t n_bases, uint32_t k,
                 const uint64_t* __restrict__ bits, uint64_t* __restrict__ words,
                 uint32_t* __restrict__ wcount, uint64_t n_words) {
    __shared__ __attribute__((aligned(16))) uint8_t sb[SCAN_LDS_BYTES + 16];
    stage_block(packed4, (n_bases + 1) / 2, (uint64_t)blockIdx.x * SCAN_BYTES_PER_BLOCK, sb);
    __syncthreads();
    const uint64_t word = (uint64_t)blockIdx.x * SCAN_THREADS + threadIdx.x;
    if (word >= n_words) return;
    uint64_t out = 0;
    // the probe is issued for every position (word 0 of the set where the k-mer is not a candidate): no branch between the
    // loads, so many of them are in flight per lane
    roll64(sb + 16, (long)threadIdx.x * 64, k, (int64_t)n_bases - (int64_t)(word * 64), [&](int p, uint64_t kmer, bool ok) {
        const uint64_t wd = bits[ok ? (kmer >> 6) : 0];
        out |= (uint64_t)(ok && ((wd >> (kmer & 63)) & 1ull)) << p;
    });
    words[word] = out;
    wcount[word] = (uint32_t)__popcll(out);
}

// ---- exclusive scan of per-word popcounts (three small kernels) ---------------------------------
constexpr int RANK_THREADS = 256;
constexpr int RANK_ITEMS = 1024;     // words per block

__global__ void __launch_bounds__(RANK_THREADS)
scan_rank_partial(const uint32_t* __restrict__ wcount, uint64_t n_words, uint64_t* __restrict__ bsum) {
    __shared__ uint32_t red[RANK_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RANK_ITEMS;
    uint32_t s = 0;
    for (int i = threadIdx.x; i < RANK_ITEMS; i += RANK_THREADS) {
        const uint64_t w = b0 + i;
        if (w < n_words) s += wcount[w];
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t t = 0; for (int i = 0; i < RANK_THREADS / 64; ++i) t += red[i]; bsum[blockIdx.x] = t; }
}

__global__ void __launch_bounds__(1024)
scan_rank_blocksums(uint64_t* __restrict__ bsum, uint64_t n_blocks, uint64_t* __restrict__ total) {
    // single workgroup: sequential chunks of 1024 block sums
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
            uint64_t add = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        const uint64_t incl = sh[threadIdx.x];
        if (i < n_blocks) bsum[i] = carry + incl - v;      // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ void __launch_bounds__(RANK_THREADS)
scan_rank_final(const uint32_t* __restrict__ wcount, uint64_t n_words, const uint64_t* __restrict__ bsum,
                uint64_t* __restrict__ word_rank, const uint64_t* __restrict__ total) {
    // each lane owns 4 consecutive words of the block's 1024
    __shared__ uint32_t wsum[RANK_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RANK_ITEMS;
    const uint64_t w0 = b0 + (uint64_t)threadIdx.x * 4;
    uint32_t c[4];
    uint32_t mine = 0;
    for (int i = 0; i < 4; ++i) { c[i] = (w0 + i < n_words) ? wcount[w0 + i] : 0; mine += c[i]; }
    // wave-inclusive scan
    uint32_t inc = mine;
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) woff += wsum[i];
    uint64_t run = bsum[blockIdx.x] + woff + (inc - mine);
    for (int i = 0; i < 4; ++i) { if (w0 + i < n_words) word_rank[w0 + i] = run; run += c[i]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) word_rank[n_words] = *total;
}

// ---- k-mer ids of the marked positions, in position order --------------------------------------
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

// One lane per output word.  A marked position reads the two or three aligned 8-byte words that hold its k bases (neighbouring
// marks share them in L2), lines the nibbles up with two funnel shifts and squeezes them to 2 bits each.  The words are
// aligned in memory: a contig that does not start on an 8-byte boundary is read from the boundary before it (same 8-byte
// word, so same page), the bytes after the last whole word one by one.
__device__ __forceinline__ uint64_t be_qword(const uint64_t* q8, uint64_t q, uint64_t n_full, uint64_t n_total_bytes) {
    if (q < n_full) return __builtin_bswap64(q8[q]);
    uint64_t v = 0;
    const uint8_t* p = (const uint8_t*)q8;
    for (int i = 0; i < 8; ++i) { const uint64_t at = q * 8 + (uint64_t)i; v = (v << 8) | (at < n_total_bytes ? p[at] : 0u); }
    return v;
}
__global__ void __launch_bounds__(256)
scan_kids_kernel(const uint8_t* __restrict__ packed4, uint64_t n_bytes, uint32_t k, const uint64_t* __restrict__ words,
                 const uint64_t* __restrict__ word_rank, uint64_t n_words,
                 uint64_t* __restrict__ kids, uint64_t kids_cap) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t m = words[w];
    uint64_t r = word_rank[w];
    const uint64_t mis = (uint64_t)((uintptr_t)packed4 & 7);
    const uint64_t* q8 = (const uint64_t*)(packed4 - mis);
    const uint64_t n_total = mis + n_bytes, n_full = n_total / 8;
    while (m) {
        const int b = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        if (r < kids_cap) {
            const uint64_t beg = w * 64 + (uint64_t)b + 2 * mis;          // nibble index of the first base, from the aligned boundary
            const uint64_t q = beg >> 4;                                 // 16 nibbles per 8-byte word
            const int sh = 4 * (int)(beg & 15);
            const uint64_t w0 = be_qword(q8, q, n_full, n_total);
            const uint64_t w1 = be_qword(q8, q + 1, n_full, n_total);
            const uint64_t hi = sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0;                 // bases 0..15
            uint64_t kmer;
            if (k <= 16) kmer = (uint64_t)squeeze16(hi) >> (2 * (16 - k));
            else {
                const uint64_t w2 = be_qword(q8, q + 2, n_full, n_total);
                const uint64_t lo = sh ? (w1 << sh) | (w2 >> (64 - sh)) : w1;             // bases 16..31
                kmer = (((uint64_t)squeeze16(hi) << 32) | squeeze16(lo)) >> (2 * (32 - k));
            }
            kids[r] = kmer;
        }
        ++r;
    }
}

size_t scan_workspace_bytes(uint64_t n_bases) {
    const uint64_t n_words = (n_bases + 63) / 64;
    const uint64_t n_blocks = (n_words + RANK_ITEMS - 1) / RANK_ITEMS;
    size_t b = 256;                                        // total
    b += (n_words * 4 + 255) / 256 * 256;                  // wcount
    b += (n_blocks * 8 + 255) / 256 * 256;                 // block sums
    b += (n_words + 1) * 8 + 256;                          // internal rank directory if the caller passes none
    return b;
}

hipError_t scan_run(const uint8_t* packed4, uint64_t n_bases, uint32_t k, const uint64_t* bits,
                    uint64_t* words, uint64_t* kids, uint64_t kids_cap, uint64_t* word_rank,
                    uint64_t* n_solid, void* workspace, size_t workspace_bytes, hipStream_t stream,
                    hipEvent_t* prof_ev) {
    if (workspace_bytes < scan_workspace_bytes(n_bases)) return hipErrorInvalidValue;
    const uint64_t n_words = (n_bases + 63) / 64;
    char* ws = (char*)workspace;
    uint64_t* total = (uint64_t*)ws;
    uint32_t* wcount = (uint32_t*)(ws + 256);
    const uint64_t n_rblocks = (n_words + RANK_ITEMS - 1) / RANK_ITEMS;
    uint64_t* bsum = (uint64_t*)(ws + 256 + (n_words * 4 + 255) / 256 * 256);
    uint64_t* own_rank = (uint64_t*)((char*)bsum + (n_rblocks * 8 + 255) / 256 * 256);
    if (!word_rank) word_rank = own_rank;
    hipError_t e;
    if (n_words == 0) {
        if ((e = hipMemsetAsync(total, 0, 8, stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(word_rank, 0, 8, stream)) != hipSuccess) return e;
        if (n_solid) return hipMemsetAsync(n_solid, 0, 8, stream);
        return hipSuccess;
    }
    const unsigned mark_blocks = (unsigned)((n_words + SCAN_THREADS - 1) / SCAN_THREADS);
    if (prof_ev) (void)hipEventRecord(prof_ev[0], stream);
    hipLaunchKernelGGL(scan_mark_kernel, dim3(mark_blocks), dim3(SCAN_THREADS), 0, stream,
                       packed4, n_bases, k, bits, words, wcount, n_words);
    if (prof_ev) (void)hipEventRecord(prof_ev[1], stream);
    hipLaunchKernelGGL(scan_rank_partial, dim3((unsigned)n_rblocks), dim3(RANK_THREADS), 0, stream, wcount, n_words, bsum);
    hipLaunchKernelGGL(scan_rank_blocksums, dim3(1), dim3(1024), 0, stream, bsum, n_rblocks, total);
    hipLaunchKernelGGL(scan_rank_final, dim3((unsigned)n_rblocks), dim3(RANK_THREADS), 0, stream, wcount, n_words, bsum, word_rank, total);
    if (prof_ev) (void)hipEventRecord(prof_ev[2], stream);
    if (kids && kids_cap)
        hipLaunchKernelGGL(scan_kids_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream,
                           packed4, (n_bases + 1) / 2, k, words, word_rank, n_words, kids, kids_cap);
    if (prof_ev) (void)hipEventRecord(prof_ev[3], stream);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (n_solid) return hipMemcpyAsync(n_solid, total, 8, hipMemcpyDeviceToDevice, stream);
    return hipSuccess;
}

}  // namespace hypo
