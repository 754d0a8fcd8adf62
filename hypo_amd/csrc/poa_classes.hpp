// poa_classes.hpp — size classes of the POA kernel (one template instantiation + launch each).
//
//   class lanes/window (windows/wave) cols/lane max seq nodes in-edges dir cells(bits) ring cells arm B seqs scores ids  memory / window
//   0     16 (4)                      4         47      48    4        2208 (4)       384        384   64   int16  u8   LDS  3.8 KB
//   1     64 (1)                      2         79      84    4        6720 (4)       640        640   64   int16  u8   LDS  8.0 KB
//   2     64 (1)                      2         127     126   6        13440 (4)      1024       1024  127  int16  u8   LDS  14.1 KB
//   3     64 (1)                      4         255     254   7        49152 (4)      2048       1024  192  int16  u8   LDS  40 KB (4 waves per CU): the wide SHORT windows (the reference cuts a weak region only
//                                                                                                                     above 2 x 100 bp, src/Contig.cpp:526-711) and small windows with large graphs
//   4     64 (1)                      10        639     2400  12       1536000 (8)    491520     16384 256  int16  u16  HBM scratch 3.3 MB / resident group: the LONG windows (<= 500 bp, arms ~ window length,
//                                                                                                                     graphs ~1.3 k nodes; src/Window.cpp:156-236), up to 2048 groups resident
//   5     64 (1)                      16        1023    32767 58       33554432 (8)   2097152    16384 16383 int32 u16  HBM scratch 61 MB / resident group: whatever overflows everything else (up to 32 groups)
// Class 0 carries 4 or 2 tiny windows per wavefront (16- / 32-lane groups with group-uniform control flow, one instruction
// stream advancing several windows); from class 1 on a window has the wave to itself.
// A window that does not fit class c (too many nodes / in-edges / cells, a predecessor row that already left
// the ring, sequence too long, or scores whose magnitude could overflow int16) is re-queued to class c+1 by
// the kernel itself.  int16 is exact iff max(|m|,|n|,|g|) * (nodes + len + 1) < 32767 (the guard spoa's SIMD
// path uses, external/spoa/src/simd_alignment_engine.cpp:660-665); Poa::align checks it before every alignment.
#pragma once
#include "poa_core.hpp"

namespace hypo {
//              GW CPL LCAP NMAX KIN DIRCELLS RINGCELLS ARMBYTES SEQMAX
typedef PoaCfg<16, 4, 47, 48, 4, 2208, 384, 384, 64, int16_t, uint8_t> PoaClass0;
// Class 0 again with two 32-lane groups per wave instead of four 16-lane ones: same capacities (the plan does not care), half
// the windows per wave but half the partners a group waits for when the windows of a wave differ.  poa_run picks one of the
// two per call: four groups when the batch is tiny windows almost only (dense short reads: 56 vs 46 M windows/s), two groups
// otherwise (C2: 3.89 vs 4.08 ms).
#ifndef HYPO_C0W_GW
#define HYPO_C0W_GW 32
#endif
// Class 1 runs ONE window per wave (64 lanes x 2 columns) since round 3: with two 32-lane groups per wave (x 4 columns, the
// geometry of rounds 1-2: -DHYPO_C1_GW=32 -DHYPO_C1_CPL=4) a window cost 345 k wave-cycles against 472 k for a class-2 window with
// four times the rows — the two windows of a wave wait for each other at every step and their group-uniform values are vector
// registers.  A window of its own per wave: class 1 alone 1.73 -> 1.33 ms on the C2 batch in 3/4 of the LDS, C2 call 2.90 -> 2.56
// ms, 1 % read error 11.3 -> 10.2 ms, dense short reads 59.6 -> 76 M windows/s (profiles/diag/r03_wave_wide_*.sh).
#define HYPO_C1_GW 64
#define HYPO_C1_CPL 2
typedef PoaCfg<HYPO_C0W_GW, 2, 47, 48, 4, 2208, 384, 384, 64, int16_t, uint8_t> PoaClass0W;
typedef PoaCfg<HYPO_C1_GW, HYPO_C1_CPL, 79, 84, 4, 6720, 640, 640, 64, int16_t, uint8_t> PoaClass1;     // (640 staged arm bytes: two groups + their stat blocks fill 32 LDS granules of 512 B exactly; copies of a neighbour take none)
#define HYPO_C2_GW 64
#define HYPO_C2_CPL 2
#define HYPO_C2_ARMBYTES 1024
typedef PoaCfg<HYPO_C2_GW, HYPO_C2_CPL, 127, 126, 6, 13440, 1024, HYPO_C2_ARMBYTES, 127, int16_t, uint8_t> PoaClass2;      // (direction codes in LDS: in HBM scratch like class 3's it was measured slower, round 3)
// class 3 keeps its direction codes (up to 254 x 256 cells) in HBM scratch (Cfg::DIRG): 16 KB of LDS per window instead of 40
typedef PoaCfg<64, 4, 255, 254, 7, 65536, 2048, 1024, 192, int16_t, uint8_t, 0, false, true> PoaClass3;
typedef PoaCfg<64, 10, 639, 2400, 12, 1536000, 491520, 16384, 256, int16_t, uint16_t, 1 << 18, true> PoaClass4;          // + 256 K path ids: runs LONG windows
// Last resort, also runs LONG windows.  The reference has no size limit (spoa's graph is a vector of heap nodes, external/spoa/
// src/graph.cpp:99-128); a window of a deep repeat (mito / rDNA: thousands of reads) must not "keep its draft", so this class is
// sized for what one window can plausibly be and then some: 16 382 sequences of up to 1 021 bases, 32 767 nodes, 58 in-edges per
// node (61 MB of HBM scratch per resident group, few groups: it is slow and rare).
typedef PoaCfg<64, 16, 1023, 32767, 58, 1 << 25, 1 << 21, 16384, 16383, int32_t, uint16_t, 1 << 22> PoaClass5;
constexpr int kNumPoaClasses = 6;
constexpr int kGiantClass = 6;          // poa_giant.hpp: its queue counter / head / statistics use slot 6 of the per-class arrays, its queue the region of class 0's
}  // namespace hypo

#define HYPO_FOR_EACH_CLASS(X) X(0, PoaClass0) X(1, PoaClass1) X(2, PoaClass2) X(3, PoaClass3) X(4, PoaClass4) X(5, PoaClass5)
// alternative geometries of a class (same id in queues and statistics; the emulator runs them under these ids)
#define HYPO_FOR_EACH_ALT_CLASS(X) X(6, PoaClass0W)
