// poa_classes.hpp — size classes of the POA kernel (one template instantiation + launch each).
//
//   class  lanes  cols/lane  max seq  max nodes  in-edges  matrix cells  scores  ids   memory
//   0      64     2          127      126        4         4096          int16   u8    LDS ~ 13.6 KB / window
//   1      64     2          127      254        8         16384         int16   u8    LDS ~ 45 KB / window
//   2      64     8          511      4000       16        1048576       int16   u16   global scratch ~ 2.6 MB / resident group
// A window that does not fit class c (too many nodes / in-edges / cells, sequence too long, or scores
// whose magnitude could overflow int16) is re-queued to class c+1 by the kernel itself.
// int16 is exact iff max(|m|,|n|,|g|) * (nodes + len + 1) < 32767 (same guard as spoa's SIMD path,
// external/spoa/src/simd_alignment_engine.cpp:660-665); poa_kernel.hip checks it per class.
#pragma once
#include "poa_core.hpp"

namespace hypo {
typedef PoaCfg<64, 2, 126, 4, 4096, int16_t, uint8_t> PoaClass0;
typedef PoaCfg<64, 2, 254, 8, 16384, int16_t, uint8_t> PoaClass1;
typedef PoaCfg<64, 8, 4000, 16, 1 << 20, int16_t, uint16_t> PoaClass2;
constexpr int kNumPoaClasses = 3;
}  // namespace hypo

#define HYPO_FOR_EACH_CLASS(X) X(0, PoaClass0) X(1, PoaClass1) X(2, PoaClass2)
