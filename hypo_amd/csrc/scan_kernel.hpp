// scan_kernel.hpp — host-visible interface of scan_kernel.hip (internal to libhypo_gpu.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hypo {
size_t scan_workspace_bytes(uint64_t n_bases);
// All pointers are device pointers; n_solid (optional) receives the number of marked positions.
hipError_t scan_run(const uint8_t* packed4, uint64_t n_bases, uint32_t k, const uint64_t* bits,
                    uint64_t* words, uint64_t* kids, uint64_t kids_cap, uint64_t* word_rank,
                    uint64_t* n_solid, void* workspace, size_t workspace_bytes, hipStream_t stream,
                    hipEvent_t* prof_ev /* 4 events or NULL */,
                    uint32_t* kids32 = nullptr /* k <= 16: the k-mer ids as 32-bit words INSTEAD of `kids` */,
                    uint32_t* spos = nullptr /* the marked positions themselves, at their ranks (at most kids_cap) */);
}  // namespace hypo
