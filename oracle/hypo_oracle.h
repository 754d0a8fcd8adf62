/*
 * hypo_oracle.h — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this library;
 * the product (hypo_amd/, libhypo_gpu.so) never links, loads or falls back to it.
 *
 * Pinning: the POA part is checked bit-for-bit against (a) the real reference classes compiled from
 * /root/reference into oracle/_ref/ (tests/test_oracle_vs_ref.py, this container only), (b) the
 * committed golden vectors in tests/golden/ that were produced by that build, and (c) spoa's own
 * GlobalConsensus known-answer string (external/spoa/test/spoa_test.cpp:220-239).
 * The solid scan has no buildable reference harness here (Contig.cpp needs sdsl's cmake-generated
 * sources) -> it is pinned only indirectly (SR starts of a real run's regions map): "parity partially
 * pinned" for oracle_solid_scan, see DESIGN.md.
 */
#ifndef HYPO_ORACLE_H
#define HYPO_ORACLE_H

#include "../include/hypo_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* alignment modes used by HyPo (external/spoa/include/spoa/alignment_engine.hpp:17-23) */
#define ORACLE_NW  1
#define ORACLE_LOV 3
#define ORACLE_ROV 4

/* Same contract as hypo_gpu_poa_batch (host pointers).  n_threads <= 0 -> all OpenMP threads.
 * cells_out / aligns_out (optional) receive sum (nodes+1)*(len+1) and the number of align() calls. */
int oracle_poa_batch(const HypoScoreParams* scores, const HypoWindowBatch* in,
                     HypoConsensusBatch* out, int n_threads,
                     uint64_t* cells_out, uint64_t* aligns_out);

/* Same contract as hypo_gpu_solid_scan (host pointers). */
int oracle_solid_scan(const uint8_t* packed4, uint64_t n_bases, uint32_t k,
                      const uint64_t* bitset_words,
                      uint64_t* solid_pos_words, uint64_t* kids, uint64_t kids_cap,
                      uint64_t* word_rank, uint64_t* n_solid);

/* Low-level replay used to localise bugs: adds n_seq text sequences with the given modes to one
 * graph (engine scores m,n,g), like repeated engine->align + graph->add_alignment.
 * Writes the alignment of the LAST sequence as (node,pos) int32 pairs into pairs_out (cap pairs),
 * the final rank_to_node order into rank_out (cap entries) and the heaviest-bundle consensus into
 * cons_out.  Returns 0 or <0. */
int oracle_replay(int m, int n, int g, int n_seq, const char* const* seqs, const int* modes,
                  int32_t* pairs_out, int pairs_cap, int* n_pairs,
                  int32_t* rank_out, int rank_cap, int* n_nodes,
                  char* cons_out, int cons_cap, int* cons_len);

/* PackedSeq helpers (src/PackedSeq.cpp:58-89,231-262): text <-> packed, MSB-first. */
void oracle_pack2(const char* s, uint32_t len, uint8_t* dst);   /* dst: ceil(len/4) bytes, zero-filled tail */
void oracle_pack4(const char* s, uint32_t len, uint8_t* dst);   /* dst: ceil(len/2) bytes */
void oracle_unpack2(const uint8_t* src, uint32_t len, char* dst);
void oracle_unpack4(const uint8_t* src, uint32_t len, char* dst);

int oracle_num_threads(void);
/* 1: kLOV end rows are ranked like the SIMD engine of a -march=native reference build ranks them (process-wide switch) */
void oracle_set_native_klov(int on);

#ifdef __cplusplus
}
#endif
#endif
