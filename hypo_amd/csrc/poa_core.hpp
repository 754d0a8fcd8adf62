// poa_core.hpp — one window's partial-order-alignment consensus, executed by one lane group.
//
// MI355X-first restructuring of the reference's per-window POA (no reference code is reused):
//   reference                                              here
//   ---------------------------------------------------   ------------------------------------------
//   Window::generate_consensus      src/Window.cpp:44-61   Poa::run (dispatch rules)
//   generate_consensus_short        src/Window.cpp:87-154  Poa::run_short (sequence order, markers)
//   SisdAlignmentEngine::linear     sisd..cpp:263-342      Poa::align, DP part: rows in rank order, the CPL
//                                                          columns of a lane in registers; vertical/diagonal
//                                                          terms from predecessor rows kept in a small LDS
//                                                          ring (only rows a later row can still reference);
//                                                          horizontal term = max-plus prefix scan over lanes
//                                                          with DPP (exact: integer max/+ is associative)
//   traceback                       sisd..cpp:344-438      the reference re-derives each move from H with a
//                                                          fixed preference order (diagonal pred 0,1.. ->
//                                                          vertical pred 0,1.. -> horizontal).  Here each lane
//                                                          resolves that preference for its cells while the
//                                                          row is still in registers and stores ONE direction
//                                                          code per cell (4 or 8 bits); the traceback walks the
//                                                          codes, and runs of "diagonal to the previous row"
//                                                          are consumed GW cells at a time with one ballot.
//   Graph::add_alignment            graph.cpp:154-271      lanes own sequence positions (every graph node
//                                                          and aligned clique occurs at most once on a
//                                                          path, so the updates are independent)
//   Graph::topological_sort         graph.cpp:293-353      runs of ready roots (and a ready aligned clique) are
//                                                          retired GW at a time; otherwise literal DFS replay,
//                                                          dependencies of the stack top checked by lanes in
//                                                          parallel; skipped when an alignment added no node
//                                                          and no edge
//   traverse_heaviest_bundle        graph.cpp:610-705      all lanes (pointer doubling over the chosen-pred
//                                                          forest) when no weight tie / branch completion is
//                                                          involved, else the literal pass on lane 0
//   generate_consensus_long, curate src/Window.cpp:156-254 Poa::run_long (HBM-scratch classes)
//
// All per-window state lives in the group's memory slice `mem` (LDS for the in-LDS size classes, HBM scratch
// for the last two, of which the hybrid LONG class keeps a second small slice in LDS): the full score matrix
// is never materialised — per window the slice holds (nodes x len) direction codes, a ring of score rows,
// the graph, and the window's packed arms staged once from HBM.
// Compiled by hipcc for gfx950 and, with HYPO_EMU, by g++ for the lockstep emulator used in tests/.
#pragma once
#include <math.h>
#include <type_traits>
#include "grp.hpp"
#include "../../include/hypo_gpu.h"

namespace hypo {
#ifdef HYPO_EMU_DBG
static unsigned long g_dbg_reason[16];
static unsigned long g_dbg_hist[64];
#define DBGH(d) do { if (g.lane == 0) g_dbg_hist[(d) < 63 ? (d) : 63]++; } while (0)
#else
#define DBGH(d) do { } while (0)
#endif

enum { MODE_NW = 1, MODE_LOV = 3, MODE_ROV = 4 };
enum { C_A = 0, C_C = 1, C_G = 2, C_T = 3, C_N = 4, C_J = 5, C_O = 6, C_NONE = 7 };
// Optional per-phase cycle accounting (diagnostic build only: make -C hypo_amd/csrc prof).
enum { PH_LOAD = 0, PH_DP = 1, PH_TRACE = 2, PH_ADD = 3, PH_TOPO = 4, PH_CONS = 5, PH_OUT = 6, PH_META = 7, PH_EXACT = 8, PH_N = 9 };
#if defined(HYPO_PHASE_TIMERS) && !defined(HYPO_EMU)
#define HYPO_TICK(k) do { const uint64_t t1_ = (uint64_t)clock64(); tphase[k] += t1_ - tlast; tlast = t1_; } while (0)
#define HYPO_TICK_RESET() do { tlast = (uint64_t)clock64(); } while (0)
#else
#define HYPO_TICK(k) do { } while (0)
#define HYPO_TICK_RESET() do { } while (0)
#endif

enum { RES_OK = 0, RES_OVERFLOW = 1, RES_UNDEFINED = 2, RES_CONS_OVERFLOW = 3, RES_UNSUPPORTED = 4, RES_INVALID = 5 };

struct PoaParams {
    const HypoWindow* windows;
    const uint8_t* draft4;
    const uint64_t* arm_off;
    const uint32_t* arm_len;
    const uint8_t* arms2;
    char* out_bases;
    const uint64_t* out_off;
    uint32_t* out_len;
    uint8_t* out_status;
    int sr_m, sr_n, sr_g, lr_m, lr_n, lr_g;
    uint64_t n_arms, draft4_bytes, arms2_bytes;          // buffer sizes: descriptors are checked against them (RES_INVALID)
    int flags;                                           // POA_NATIVE_KLOV
};
// opt-in: rank kLOV end rows by the maximum over the whole row, like the AVX2 / SSE4.1 engine of a -march=native build of the
// reference does (simd_alignment_engine.cpp:803,834-840; traceback still starts in column L, :859-861)
enum { POA_NATIVE_KLOV = 1, POA_MIN_CLASS_SHIFT = 8 };      // bits 8-9: smallest size class a SHORT window starts in (0 = the plan's choice)

// How the per-window code reaches PoaParams.  On the device it is a pointer into the kernel-argument segment
// (constant address space) that is made opaque at every use, so each use is a fresh scalar load instead of nine
// 64-bit pointers pinned in SGPRs for the lifetime of a persistent wave (the row loop needs those registers).
#ifdef HYPO_EMU
struct PoaParamRef {
    const PoaParams* p;
    HD const PoaParams* operator->() const { return p; }
};
#else
struct PoaParamRef {
    typedef const PoaParams __attribute__((address_space(4)))* cptr;
    cptr p;
    HD cptr operator->() const { cptr q = p; asm volatile("" : "+s"(q)); return q; }
};
#endif

// exact threading of sequences that spell a path (Poa::thread_cols); HYPO_EXACT=0 sends every alignment through the score rows
#ifndef HYPO_EXACT
#define HYPO_EXACT 1
#endif
// ... also in the class of wide windows (class 3: along the guide only, see Poa::align)
#define HYPO_EXACT_WIDE 1
// ... and kNW arms one substitution off the path of the arm before them (Poa::guided_one_sub); 0: they go through the score rows
#ifndef HYPO_ONE_SUB
#define HYPO_ONE_SUB 1
#endif
#ifndef HYPO_ONE_SUB_ROV
#define HYPO_ONE_SUB_ROV 1          // ... kROV and kLOV arms too (suffix arms end where the guide ends and may start anywhere; prefix arms start where it starts and may end anywhere)
#endif
// a new node in an old clique goes into the literal order without a sort (Poa::topo_insert); 0: every change of the graph is sorted
#ifndef HYPO_TOPO_INSERT
#define HYPO_TOPO_INSERT 1
#endif
// the packed classes build the row metadata when the score rows or Poa::thread_cols are reached, not at the start of every alignment
#ifndef HYPO_DEFER_META
#define HYPO_DEFER_META 1
#endif
// int16 score rows as packed pairs of columns (Poa::rows_pk); HYPO_PACKED=0 builds the one-column-per-register loop everywhere
#define HYPO_PACKED 1
template <int GW_, int CPL_, int LCAP_, int NMAX_, int KIN_, int DIRCELLS_, int RINGCELLS_, int ARMBYTES_,
          int SEQMAX_, class ScoreT, class IdT, int PATHCAP_ = 0, bool HYBRID_ = false, bool DIRG_ = false>
struct PoaCfg {
    // DIRG: an LDS class whose direction codes (the one array that grows with nodes x length) live in a per-group slice of HBM
    // scratch instead: they are written once per row (fire and forget) and read by the traceback only, so the window's LDS
    // footprint is the graph + the ring and the class can run next to the others (PoaLayout::DIRG_BYTES per resident group)
    static constexpr bool DIRG = DIRG_;
    // the packed row loop: int16 rows, 4-bit direction codes, an even number of columns per lane, state in LDS
    static constexpr bool PACKED = HYPO_PACKED && sizeof(ScoreT) == 2 && CPL_ % 2 == 0 && CPL_ <= 8 && !HYBRID_ && (KIN_ <= 7);
    static constexpr bool HYBRID = HYBRID_;     // state in HBM scratch except the arrays the graph walks hammer (PoaLayout::FAST_BYTES of LDS)
    // ... and the hybrid class runs its int16 rows on packed pairs as well (Poa::rows_pk_hyb: the dense ring, byte codes, any
    // in-degree); HYPO_PACKED_HYB=0 keeps the one-column-per-register loop there
#define HYPO_PACKED_HYB 1
    static constexpr bool PACKED_HYB = HYPO_PACKED_HYB && HYBRID_ && sizeof(ScoreT) == 2 && CPL_ % 2 == 0 && PATHCAP_ > 0;
    static constexpr int PATHCAP = PATHCAP_;    // node ids of the sequences' paths (LONG windows only; 0 = class cannot run them)
    static constexpr int GW = GW_;              // lanes per window
    static constexpr int CPL = CPL_;            // matrix columns per lane
    static constexpr int LMAX = LCAP_;          // longest sequence incl. markers
    static constexpr int NMAX = NMAX_;          // graph nodes
    static constexpr int KIN = KIN_;            // in-edges per node
    static constexpr int DIRCELLS = DIRCELLS_;  // direction codes (nodes x row stride)
    static constexpr int RINGCELLS = RINGCELLS_;// score cells of the row ring
    static constexpr int ARMBYTES = ARMBYTES_;  // packed arm bytes staged per window
    static constexpr int SEQMAX = SEQMAX_;      // sequences (arms + backbone) per window
    static constexpr int AL = 6;                // aligned clique partners (alphabet ACGTNJO -> at most 6)
#define HYPO_HYB_STK 1536
    static constexpr int STK = HYBRID_ ? HYPO_HYB_STK : 2 * NMAX_;   // DFS stack entries (hybrid: the LDS copy is smaller; deeper DFS -> next class)
#define HYPO_RING1 6
// LONG windows in the hybrid class: the rank order is kept valid incrementally and the literal DFS order of the reference is
// computed only where it can be observed (Poa::lazy_update); 0: literal sort after every alignment that changed the graph
#define HYPO_LAZY_TOPO 1
// ... and SHORT windows of the packed classes as well (round 5): HYPO_LAZY_PACKED
#define HYPO_LAZY_PACKED 1
    static constexpr int RING1 = HYBRID_ ? HYPO_RING1 : 0;
    // class 3 (wide windows, and every window re-queued from classes 0-2: the ones whose graphs keep changing): the lazy order's scratch
    // lives in the score ring (idle between row loops).  Classes 0-2 sort literally: their windows see one to three sorts, and carrying the
    // tie bookkeeping through their row loops cost 5-9 % at 0.2 % read error (measured, profiles/diag/r05_sched_experiments.txt)
    static constexpr bool LAZY_RING = HYPO_LAZY_PACKED && PACKED && DIRG_;
    static constexpr bool LAZY = (HYBRID_ && (HYPO_LAZY_TOPO != 0) && PATHCAP_ > 0) || LAZY_RING;  // hybrid: this many most recent score rows are also kept in LDS
    // direction codes: 4 bits when the pred index fits (diag p = p, vert p = 7+p, horiz = 14, fast = 15)
    static constexpr bool NIB = (KIN_ <= 7) && (CPL_ % 2 == 0);
    static constexpr int DIRBYTES = NIB ? DIRCELLS_ / 2 : DIRCELLS_;
    typedef ScoreT score_t;
    typedef IdT id_t;
    // edge weights grow by 2 per traversal: 8 bits suffice while a window has <= 127 sequences
    typedef typename std::conditional<(SEQMAX_ <= 127), uint8_t, uint16_t>::type wt_t;
    static constexpr int ID_NONE = (IdT)~(IdT)0;
    static_assert(LCAP_ <= GW_ * CPL_ - 1, "columns 0..L must fit the group");
    static_assert(LCAP_ <= 1023 && ARMBYTES_ <= 16384 && SEQMAX_ <= 16383, "sequence table entry is 32 bits");
    static_assert(KIN_ + 6 <= GW_, "dependency lanes");
    static_assert(KIN_ <= 62, "direction byte holds the pred index in 6 bits");
    static constexpr int DIRBYTES_LDS = DIRG_ ? 0 : DIRBYTES;      // what the direction codes take of the group's own slice
    static_assert((int)sizeof(ScoreT) * RINGCELLS_ + DIRBYTES_LDS >= 14 * NMAX_, "consensus scratch aliases ring+dir");
    static_assert(NMAX_ < ID_NONE, "id range");
    static_assert(SEQMAX_ <= STK && SEQMAX_ <= ID_NONE, "arm indices are parked in the DFS stack while staging");
};

template <int N> HD constexpr int align_up(int x) { return (x + N - 1) / N * N; }

template <class Cfg>
struct PoaLayout {   // byte offsets inside a group's memory slice
    typedef typename Cfg::score_t score_t;
    typedef typename Cfg::id_t id_t;
    static constexpr int oRing = 0;                                                     // ring, then dir: contiguous
    static constexpr int oDir = oRing + align_up<16>(Cfg::RINGCELLS * (int)sizeof(score_t));   // (consensus scratch aliases both)
    // posnode (alignment -> graph update) and the DFS stack (toposort, arm staging) are never live together, and neither is
    // live during the row loop: the packed loop lets lanes beyond a row's width store their (unused) direction codes, which
    // land in the following rows' cells or, behind the last row, at most GW * CPL / 2 bytes into this region
    static constexpr int POS_BYTES = (Cfg::LMAX + 1) * 2 > Cfg::STK * (int)sizeof(id_t) ? (Cfg::LMAX + 1) * 2 : Cfg::STK * (int)sizeof(id_t);
    static_assert(!Cfg::PACKED || POS_BYTES >= Cfg::GW * Cfg::CPL / 2, "slack behind the direction codes");
    static constexpr int oPosnode = oDir + align_up<16>(Cfg::DIRBYTES_LDS);
    // Cfg::DIRG: bytes of HBM scratch per resident group (the codes + the slack the packed loop's unguarded stores need)
    static constexpr int DIRG_BYTES = Cfg::DIRG ? align_up<256>(Cfg::DIRBYTES + Cfg::GW * Cfg::CPL + 16) : 0;
    static constexpr int oRowmeta = oPosnode + align_up<16>(POS_BYTES);
    static constexpr int oSeqtab = oRowmeta + align_up<16>(Cfg::NMAX * 4);
    static constexpr int oInw = oSeqtab + align_up<16>(Cfg::SEQMAX * 4);
    static constexpr int oInp = oInw + align_up<16>(Cfg::NMAX * Cfg::KIN * (int)sizeof(typename Cfg::wt_t));
    static constexpr int oAl = oInp + align_up<16>(Cfg::NMAX * Cfg::KIN * (int)sizeof(id_t));
    static constexpr int oR2n = oAl + align_up<16>(Cfg::NMAX * Cfg::AL * (int)sizeof(id_t));
    static constexpr int oN2r = oR2n + align_up<16>(Cfg::NMAX * (int)sizeof(id_t));
    static constexpr int oCode = oN2r + align_up<16>(Cfg::NMAX * (int)sizeof(id_t));
    static constexpr int oNin = oCode + align_up<16>(Cfg::NMAX);
    static constexpr int oNout = oNin + align_up<16>(Cfg::NMAX);
    static constexpr int oNal = oNout + align_up<16>(Cfg::NMAX);
    static constexpr int oMark = oNal + align_up<16>(Cfg::NMAX);
    static constexpr int oSeq = oMark + align_up<16>(Cfg::NMAX);
    static constexpr int oArms = oSeq + align_up<16>(Cfg::LMAX + 1);
    // LONG windows (generate_consensus_custom needs every sequence's path through the graph)
    static constexpr int LONGSEQ = Cfg::PATHCAP ? Cfg::SEQMAX : 0;
    static constexpr int LONGN = Cfg::PATHCAP ? Cfg::NMAX : 0;
    static constexpr int oPathNodes = oArms + align_up<16>(Cfg::ARMBYTES);
    static constexpr int oPathOff = oPathNodes + align_up<16>(Cfg::PATHCAP * (int)sizeof(id_t));
    static constexpr int oPathLen = oPathOff + align_up<16>(LONGSEQ * 4);
    static constexpr int oPathMult = oPathLen + align_up<16>(LONGSEQ * 2);
    static constexpr int oMsa = oPathMult + align_up<16>(LONGSEQ * 2);
    static constexpr int oDst = oMsa + align_up<16>(LONGN * 2);
    static constexpr int oCons = oDst + align_up<16>(LONGN * 4);
    // HBM-scratch classes: matrix row of every in-edge source by rank, so that the row loop needs one load per extra
    // predecessor instead of the r2n -> inp -> n2r chain (three dependent HBM round trips per predecessor)
    static constexpr int oPredRows = oCons + align_up<16>(LONGN);
    // lazy rank order (Cfg::LAZY): a second rank -> node array (the update writes the new order beside the old one; the literal
    // order of an end-row tie goes there as well), a second node -> rank array for the latter, and the new nodes of the
    // alignment in hand with the rank they are inserted behind
    static constexpr int LAZYN = (Cfg::LAZY && !Cfg::LAZY_RING) ? Cfg::NMAX : 0;
    static constexpr int LAZYL = (Cfg::LAZY && !Cfg::LAZY_RING) ? Cfg::LMAX + 1 : 0;
    static constexpr int oR2nAlt = oPredRows + align_up<16>(LONGN * Cfg::KIN * (int)sizeof(id_t));
    static constexpr int oN2rAlt = oR2nAlt + align_up<16>(LAZYN * (int)sizeof(id_t));
    static constexpr int oNewId = oN2rAlt + align_up<16>(LAZYN * (int)sizeof(id_t));
    static constexpr int oNewSlot = oNewId + align_up<16>(LAZYL * 2);
    // rarely touched group-uniform scalars (Poa::stat): per-window counters, what a re-queued window takes along, and the
    // per-wave totals of the kernel.  In registers they were live across the whole window — a vector register each in the
    // sub-wave classes, spilled scalars in the others; here they cost an LDS access where they change.
    static constexpr int STAT_BYTES = 208;
    static constexpr int oStat = oNewSlot + align_up<16>(LAZYL * 2);
    static constexpr int BYTES = oStat + STAT_BYTES;
    // Hybrid classes (Cfg::HYBRID) keep everything above in HBM scratch except what the topological sort and the graph update
    // chase with dependent loads: DFS stack / posnode, in-degree, clique size, marks, current sequence.  These live in a
    // second, small slice in LDS (their slots in the big slice stay unused).
    static constexpr int fPosnode = 0;
    static constexpr int fNin = fPosnode + align_up<16>(POS_BYTES);
    static constexpr int fNal = fNin + align_up<16>(Cfg::NMAX);
    static constexpr int fMark = fNal + align_up<16>(Cfg::NMAX);
    static constexpr int fSeq = fMark + align_up<16>(Cfg::NMAX);
    static constexpr int SMAX = (Cfg::LMAX + 1 + Cfg::CPL - 1) / Cfg::CPL * Cfg::CPL;          // largest row stride
    static constexpr int fRing1 = fSeq + align_up<16>(Cfg::LMAX + 1);                              // RING1 recent score rows
    static constexpr int fStat = fRing1 + align_up<16>(Cfg::RING1 * SMAX * (int)sizeof(score_t));
    static constexpr int FAST_BYTES = fStat + STAT_BYTES;   // (row metadata in LDS as well was measured: fewer resident
                                                                             // waves cost more than the shorter row loop gains)
};

// What the persistent kernel hangs on a Poa object (poa_kernel.hip: PoaPrefetch): where the queue of the launch is, so that a
// group can fetch its next window's descriptor, output range and carry word into its LDS block in one go (Poa::fetch_next) and
// a window opens without touching HBM for them.  The emulator and the HBM-scratch classes use PoaNoHook: one window per run()
// call, descriptor and output range read where they are needed.
struct PoaNoHook { static constexpr bool enabled = false; };

template <class Cfg, class Hook = PoaNoHook>
struct Poa {
    typedef typename Cfg::score_t score_t;
    typedef typename Cfg::id_t id_t;
    typedef typename Cfg::wt_t wt_t;
    typedef PoaLayout<Cfg> Lay;
    static constexpr int GW = Cfg::GW, CPL = Cfg::CPL, KIN = Cfg::KIN, NMAX = Cfg::NMAX, AL = Cfg::AL;
    static constexpr bool NIB = Cfg::NIB;
    static constexpr int NEG = -(1 << 29);
    static constexpr bool PK = Cfg::PACKED;
    static_assert(!PK || Cfg::NIB, "packed rows store nibble codes");
    static_assert(!PK || (int)sizeof(score_t) * Cfg::RINGCELLS >= Cfg::STK * (int)sizeof(id_t), "packed classes: the DFS stack fits the score ring");
    // direction codes
    static constexpr int DIR_FAST = NIB ? 15 : 0xFF;     // diagonal via pred 0 and pred 0 is the previous row
    static constexpr int DIR_HORIZ = NIB ? 14 : 0xFE;
    HD static int dir_diag(int p) { return NIB ? p : (p << 1); }
    HD static int dir_vert(int p) { return NIB ? 7 + p : ((p << 1) | 1); }
    HD static bool is_vert(int d) { return NIB ? d >= 7 : (d & 1); }
    HD static int dir_pred(int d) { return NIB ? (d >= 7 ? d - 7 : d) : (d >> 1); }

    // a lane's cells of one row move as one block; aligned to the largest power of two dividing its size (CPL = 10 -> 4 / 2 bytes)
    static constexpr int pow2_of(int x) { return x & -x; }
    struct alignas(pow2_of((int)sizeof(score_t) * CPL)) Pack { score_t v[CPL]; };
    struct alignas(pow2_of(NIB ? CPL / 2 : CPL)) DPack { uint8_t v[NIB ? CPL / 2 : CPL]; };
    // sequence table entry: bits 0-13 src (LDS offset of the staged bytes, or arm index), 14 unused, 15 "byte-identical to
    // the previous entry", 16-25 length, 26 head marker J, 27 tail marker O,
    // 28-29 mode (0 NW, 1 LOV, 2 ROV), 30-31 where (0 staged in LDS, 1 arms2 in HBM, 2 draft4 in HBM: 4-bit packed)
    HD static uint32_t seq_ent(uint32_t src, uint32_t len, bool head, bool tail, int mode, bool /*four*/, int where) {
        const uint32_t mc = mode == MODE_NW ? 0u : (mode == MODE_LOV ? 1u : 2u);
        return (src & 0x3fffu) | (len << 16) | ((head ? 1u : 0u) << 26) | ((tail ? 1u : 0u) << 27) | (mc << 28) |
               ((uint32_t)where << 30);
    }

    const Grp<GW>& g;
    const PoaParamRef P;
    // memory slice
    score_t* ring; uint8_t* dir; uint32_t* rowmeta; uint32_t* seqtab; wt_t* inw; int16_t* posnode;
    id_t *inp, *al, *r2n, *n2r, *stack;
    uint8_t *code, *nin, *nout, *nal, *mark, *seq, *armbuf;
    uint8_t* sidx;                                           // packed classes: SAVEd rows before each row (aliases mark: toposort and the row loop never overlap)
    id_t* pathnodes; uint32_t* pathoff; uint16_t *pathlen, *pathmult, *msa; uint32_t* dstcnt; uint8_t* consbuf; id_t* predrows; score_t* ring1;
    id_t *r2n_alt, *n2r_alt, *newid; int16_t* newslot;       // lazy rank order (Cfg::LAZY)
    bool lazy_on; int n_new;                                 // lazy_on: this window keeps its order lazily (LONG windows); n_new: new nodes of the alignment in hand
    int n_paths, path_used, head_first;
    // group-uniform state
    int n_nodes; int L; bool topo_dirty; bool meta_dirty;
    int tb_steps; int tb_fv;
    bool last_changed;         // did the most recent add_alignment change the graph topology?
    bool weights_done;         // ... and the weights along its path are already incremented (Poa::thread_guided)
    bool threaded;             // the alignment in hand was threaded (Poa::thread_cols): every position sits on a node of its own letter, along existing edges
    // Group-uniform scalars that change rarely live in a small block of the group's LDS slice (PoaLayout::oStat) instead of
    // registers; lane 0 writes, everybody may read after the next sync.
    //   ST_CELLS .. ST_CEXACT  per-window counters: reference-equivalent cells / alignments, reused alignments, threaded ones,
    //                          cells through the score rows / the one-bit rows
    //   ST_XT, ST_XH           threading attempts / hits of the window in hand (HYPO_EXACT_ADAPT)
    //   ST_NEED                after RES_OVERFLOW of a SHORT window: projected node count (0 = unknown)
    //   ST_CKIND .. ST_CPASS   after RES_OVERFLOW of a SHORT window: what of the work so far can travel to the next class (Poa::
    //                          spill).  CARRY_BEFORE: the graph is exactly what the sequences before ST_CS left (the failed step
    //                          changed nothing); CARRY_UNSORTED: sequence ST_CS - 1 is in the graph but its topological sort did not
    //                          fit (the next class sorts first); ST_CPASS: nothing newer to spill, but the spill the window came
    //                          with is still valid and travels on
    //   ACC_*                  per-wave totals of poa_class_kernel (LDS classes)
    //   ST_ARMB, ST_OLEN       packed bytes of the window's arms (build_seqtab) and the length it answered with (finish): the
    //                          algorithmic bytes of SURVEY.md 8(d) without reading the descriptor and the arm lengths again
    //   NX_*                   Hook::enabled: the NEXT window of this group (window index, the descriptor's ten words,
    //                          out_off[w], out_off[w + 1], carry word; Poa::fetch_next); CUR_OFF / CUR_STATIC: the output range of
    //                          the window in hand and the part of its algorithmic bytes the descriptor gives
    enum { ST_CELLS = 0, ST_ALIGNS, ST_REUSED, ST_XHITS, ST_CSCORED, ST_CEXACT, ST_XT, ST_XH, ST_NEED, ST_CKIND, ST_CS, ST_CCHAIN0, ST_CPASS, ST_ARMB, ST_OLEN, ST_N,
           ST_LASTX = ST_CPASS,   // while a window runs: did its latest alignment thread?  (ST_CPASS is written after the window's last step only)
           ST_MAXD = ST_N, ST_NSORT,  // ST_MAXD: of the rank order in hand (Poa::build_rowmeta), ring rows the furthest predecessor needs; ST_NSORT: literal sorts of the window in hand
           ACC_CELLS, ACC_ALIGNS, ACC_ABYTES, ACC_REUSED, ACC_THR, ACC_CSCORED, ACC_CTHR, ACC_NOK, ACC_NESC, ACC_NFAIL, ACC_NCARRIED, ACC_END,
           NX_W0 = ACC_END, NX_OFF = NX_W0 + 10, NX_WIDX = NX_OFF + 4, NX_CARRY, CUR_STATIC, CUR_OFF, NX_END = CUR_OFF + 4 };
    static constexpr uint32_t NX_NONE = 0xffffffffu;         // NX_WIDX: the queue is drained
    static_assert(sizeof(HypoWindow) == 40, "the prefetch moves the descriptor as ten words");
    static_assert(NX_END <= Lay::STAT_BYTES / 4 && ST_N <= GW && GW >= 16, "stat block");
    enum { CARRY_NONE = 0, CARRY_BEFORE = 1, CARRY_UNSORTED = 2 };
    static constexpr int RES_OVERFLOW_CLEAN = 64;            // add_alignment: RES_OVERFLOW before anything was changed (internal)
    uint32_t* stat;
    HD void stat_add(int k, uint32_t v) const { if (g.lane == 0) stat[k] += v; }
    HD void stat_set(int k, uint32_t v) const { if (g.lane == 0) stat[k] = v; }
    HD uint32_t stat_get(int k) const { return stat[k]; }
#if defined(HYPO_PHASE_TIMERS) || defined(HYPO_EMU)
#define HYPO_DIAG(x) do { x; } while (0)
    uint32_t rows_done, topo_runs, cons_serial, rows_slow, exact_tries, guided_hits, rows_scored_n, topo_dfs, topo_fast;
    uint32_t one_sub_hits, cols_hits, topo_inserts, lazy_updates, tie_sorts;      // hits of guided_one_sub / thread_cols / topo_insert, lazy_update calls, first_of_rows calls
#else
#define HYPO_DIAG(x) do { } while (0)
#endif
    uint64_t tphase[PH_N]; uint64_t tlast;

    // `fast`: the LDS slice of a hybrid class (ignored otherwise: one slice, LDS or HBM, holds everything)
    // `dirg`: the group's direction-code slice in HBM scratch (Cfg::DIRG; PoaLayout::DIRG_BYTES)
    HD Poa(const Grp<GW>& g_, const PoaParamRef& P_, char* mem, char* fast = nullptr, char* dirg = nullptr) : g(g_), P(P_) {
        ring = (score_t*)(mem + Lay::oRing); dir = Cfg::DIRG ? (uint8_t*)dirg : (uint8_t*)(mem + Lay::oDir);
        rowmeta = (uint32_t*)(mem + Lay::oRowmeta); seqtab = (uint32_t*)(mem + Lay::oSeqtab);
        inw = (wt_t*)(mem + Lay::oInw);
        constexpr bool HYB = Cfg::HYBRID;
        posnode = (int16_t*)(HYB ? fast + Lay::fPosnode : mem + Lay::oPosnode);
        inp = (id_t*)(mem + Lay::oInp); al = (id_t*)(mem + Lay::oAl);
        r2n = (id_t*)(mem + Lay::oR2n); n2r = (id_t*)(mem + Lay::oN2r);
        // DFS stack (toposort) / arm indices while staging: beside posnode in the hybrid classes; in the packed classes in the score ring
        // (idle outside the row loop), so that posnode[] — the path of the sequence before — survives a sort and can guide the next arm
        stack = (id_t*)(HYB ? fast + Lay::fPosnode : (PK ? mem + Lay::oRing : mem + Lay::oPosnode)); code = (uint8_t*)(mem + Lay::oCode);
        nin = (uint8_t*)(HYB ? fast + Lay::fNin : mem + Lay::oNin); nout = (uint8_t*)(mem + Lay::oNout);
        nal = (uint8_t*)(HYB ? fast + Lay::fNal : mem + Lay::oNal); mark = (uint8_t*)(HYB ? fast + Lay::fMark : mem + Lay::oMark);
        seq = (uint8_t*)(HYB ? fast + Lay::fSeq : mem + Lay::oSeq); armbuf = (uint8_t*)(mem + Lay::oArms);
        sidx = mark;
        pathnodes = (id_t*)(mem + Lay::oPathNodes); pathoff = (uint32_t*)(mem + Lay::oPathOff);
        pathlen = (uint16_t*)(mem + Lay::oPathLen); pathmult = (uint16_t*)(mem + Lay::oPathMult);
        msa = (uint16_t*)(mem + Lay::oMsa); dstcnt = (uint32_t*)(mem + Lay::oDst); consbuf = (uint8_t*)(mem + Lay::oCons);
        predrows = (id_t*)(mem + Lay::oPredRows);
        r2n_alt = (id_t*)(mem + Lay::oR2nAlt); n2r_alt = (id_t*)(mem + Lay::oN2rAlt); newid = (id_t*)(mem + Lay::oNewId); newslot = (int16_t*)(mem + Lay::oNewSlot);
        if constexpr (Cfg::LAZY_RING) {
            // score ring, between row loops: [shift table of lazy_update / DFS stack of a literal sort | r2n_alt | n2r_alt | newid | newslot]
            constexpr int oA = align_up<2>((NMAX + 1) * 2 > Cfg::STK * (int)sizeof(id_t) ? (NMAX + 1) * 2 : Cfg::STK * (int)sizeof(id_t));
            constexpr int oB = oA + NMAX * (int)sizeof(id_t), oC = oB + NMAX * (int)sizeof(id_t), oD = align_up<2>(oC + (Cfg::LMAX + 1) * (int)sizeof(id_t));
            static_assert(oD + (Cfg::LMAX + 1) * 2 <= (int)sizeof(score_t) * Cfg::RINGCELLS, "the lazy order's scratch fits the score ring");
            r2n_alt = (id_t*)(mem + Lay::oRing + oA); n2r_alt = (id_t*)(mem + Lay::oRing + oB); newid = (id_t*)(mem + Lay::oRing + oC); newslot = (int16_t*)(mem + Lay::oRing + oD);
        }
        lazy_on = false; n_new = 0; guide_len = -1; guide_mode = 0;
        stat = (uint32_t*)(HYB ? fast + Lay::fStat : mem + Lay::oStat);
        for (int t = g.lane; t < Lay::STAT_BYTES / 4; t += GW) stat[t] = 0;
        ring1 = (score_t*)(HYB ? fast + Lay::fRing1 : mem + Lay::oRing);
        n_paths = 0; path_used = 0; head_first = 0;
        n_nodes = 0; L = 0; topo_dirty = false; meta_dirty = true; tb_steps = 0; tb_fv = 0;
        last_changed = true; threaded = false;
        HYPO_DIAG(rows_done = 0; topo_runs = 0; cons_serial = 0; rows_slow = 0; exact_tries = 0; guided_hits = 0; rows_scored_n = 0; topo_dfs = 0; topo_fast = 0; one_sub_hits = 0; cols_hits = 0; topo_inserts = 0; lazy_updates = 0; tie_sorts = 0);
        for (int i = 0; i < PH_N; ++i) tphase[i] = 0;
        tlast = 0;
        HYPO_TICK_RESET();
    }

    // ---- the next window of this group (Hook::enabled) ----------------------------------------------------------------------
    // Behind every window the group claims its next queue slot (one atomic by lane 0), reads the slot's window index, then — side
    // by side, a word per lane — the descriptor (lanes 0-9), the output range out_off[w], out_off[w + 1] (lanes 10-13) and the
    // carry word (lane 15), and leaves all of it in the stat block (NX_*): three dependent round trips where a window used to
    // open with five (a look at the cursor, the atomic, the slot, the descriptor, later the output range) and to close with
    // three more for its statistics.
    // (Measured and dropped: claiming the next slot while the window in hand is still being staged, the three round trips riding
    // on those of build_seqtab — C2 call 3.45 ms against 2.94 ms with the claim behind the window, class 2 alone 2.4-2.8 ms
    // against 2.07: a claimed window waits for its group while others run dry.  profiles/diag/r03_prefetch_ab.sh)
    Hook hk;
    HD void fetch_next() const {
        if constexpr (Hook::enabled) {
            g.sync();
            uint32_t a = 0;
            if (g.lane == 0) a = hk.claim();
            const uint32_t count = hk.count();                  // (read beside the atomic: no round trip of its own)
            const uint32_t idx = (uint32_t)g.shfl((int)a, 0);
            const uint32_t wn = idx < count ? hk.item(idx) : NX_NONE;
            a = wn;                                            // (lane 14 keeps the window index: NX_WIDX)
            if (wn != NX_NONE) {
                if (g.lane < 10) a = ((const uint32_t*)(P->windows + wn))[g.lane];
                else if (g.lane < 14) a = ((const uint32_t*)(P->out_off + wn))[g.lane - 10];
                else if (g.lane == 15) a = idx >= hk.planned() ? hk.carry(wn) : 0u;      // slots from `planned` on hold re-queued windows
            }
            if (g.lane < 16) stat[NX_W0 + g.lane] = a;
            g.sync();
        }
    }
    // output range of window w (the hook keeps the one of the window in hand in LDS)
    HD void out_range(uint32_t w, uint64_t* o, uint64_t* cap) const {
        if constexpr (Hook::enabled) {
            const uint64_t o0 = (uint64_t)stat[CUR_OFF] | ((uint64_t)stat[CUR_OFF + 1] << 32), o1 = (uint64_t)stat[CUR_OFF + 2] | ((uint64_t)stat[CUR_OFF + 3] << 32);
            *o = o0; *cap = o1 - o0;
        } else { *o = P->out_off[w]; *cap = P->out_off[w + 1] - *o; }
    }

    // ---- sequence table: the window's sequences in the reference's consumption order -----------------
    // (Window.cpp:87-130: [draft if no internal arm] internal.. | prefix arms reversed | suffix arms;
    // zero-length arms are skipped).  Packed arm bytes are staged once into `armbuf` (one exposed HBM
    // latency per window instead of one per arm); what does not fit is read in place.
    HD int build_seqtab(const HypoWindow& W, bool is_long, int* n_seq_out, bool* added_out) {
        const uint32_t a0 = W.first_arm;
        const int ni = (int)W.n_internal, np = (int)W.n_prefix, ns = (int)W.n_suffix;
        const int narm = ni + np + ns;
        const int base = (ni == 0 || is_long) ? 1 : 0;     // slot 0 = draft backbone (LONG: always, Window.cpp:173-178)
        if (narm + base > Cfg::SEQMAX) return RES_OVERFLOW;
        if ((int)W.draft_len + (is_long ? 0 : 2) > Cfg::LMAX && base) return RES_OVERFLOW;
        g.sync();
        if (base && g.lane == 0) seqtab[0] = seq_ent(0, W.draft_len, !is_long, !is_long, MODE_NW, true, 2);
        bool over = false, any_len = false, bad = false;
        uint32_t armb = 0;
        if constexpr (STAGE_FUSED) {
            if (!is_long) return build_seqtab_fused(W, base, narm, n_seq_out, added_out);
        }
        for (int t = g.lane; t < narm; t += GW) {
            int a, mode; bool head, tail;                  // consumption slot t -> arm index
            if (is_long) { a = t; mode = MODE_NW; head = false; tail = false; }   // all kNW, insertion order, no markers (Window.cpp:179-206)
            else if (t < ni) { a = t; mode = MODE_NW; head = true; tail = true; }
            else if (t < ni + np) { a = ni + (np - 1 - (t - ni)); mode = MODE_LOV; head = true; tail = false; }
            else { a = t; mode = MODE_ROV; head = false; tail = true; }
            const uint32_t len = P->arm_len[a0 + a];
            { const uint64_t ao = P->arm_off[a0 + a], ab = P->arms2_bytes; if (ao > ab || ((uint64_t)len + 3) / 4 > ab - ao) { bad = true; continue; } }
            if (len + (is_long ? 0u : 2u) > (uint32_t)Cfg::LMAX) { over = true; continue; }
            if (len) any_len = true;
            armb += (len + 3) >> 2;
            seqtab[base + t] = seq_ent((uint32_t)a, len, head, tail, mode, false, 1);
        }
        if (g.any(bad)) return RES_INVALID;
        if (g.any(over)) return RES_OVERFLOW;
        any_len = g.any(any_len);
        if constexpr (Hook::enabled) { armb = (uint32_t)g.reduce_add((int)armb); stat_set(ST_ARMB, armb); }
        g.sync();
        // Arms that repeat their predecessor byte for byte (same length, markers, mode) are flagged first, on the packed bytes
        // where they lie in HBM (one lane per arm; neighbours in consumption order are neighbours in memory): four arms in five
        // of a 30x short-read window are such copies, they are never aligned (Poa::same_as_previous) and need no LDS of their own.
        for (int t = g.lane + 1; t < narm; t += GW) {
            const uint32_t a = seqtab[base + t], b = seqtab[base + t - 1];
            const int nb = (int)(((a >> 16) & 0x3ff) + 3) >> 2;
            if (((a ^ b) & 0x3fff0000u) == 0 && nb > 0) {
                const uint8_t* pa = P->arms2 + P->arm_off[a0 + (a & 0x3fff)];
                const uint8_t* pb = P->arms2 + P->arm_off[a0 + (b & 0x3fff)];
                bool same = true;
                for (int k = 0; k < nb; ++k) same &= pa[k] == pb[k];
                if (same) seqtab[base + t] = a | 0x8000u;           // (the neighbour's compare looks at bits 0-13 and 16-29 only)
            }
        }
        g.sync();
        if (g.lane == 0) {                                  // LDS offsets of the staged arms (serial prefix, <= SEQMAX small adds)
            int used = 0, prev_off = -1;                    // prev_off: where the predecessor's bytes were staged (-1: they were not)
            for (int t = 0; t < narm; ++t) {
                const uint32_t e = seqtab[base + t];
                const int nb = (int)(((e >> 16) & 0x3ff) + 3) >> 2;
                if (e & 0x8000u) {                          // a copy shares its predecessor's staged bytes
                    if (prev_off >= 0) seqtab[base + t] = (e & 0x3fff8000u) | (uint32_t)prev_off;     // where = 0
                    else prev_off = -1;
                } else if (used + nb <= Cfg::ARMBYTES) {
                    // keep the arm index in posnode-free scratch: staged entries remember it via `stack`
                    stack[t] = (id_t)(e & 0x3fff);
                    seqtab[base + t] = (e & 0x3fff0000u) | (uint32_t)used;        // where = 0 (bits 30-31 cleared)
                    prev_off = used;
                    used += nb;
                } else prev_off = -1;
            }
        }
        g.sync();
        for (int t = g.lane; t < narm; t += GW) {           // one lane copies one arm: the loads of a lane pipeline
            const uint32_t e = seqtab[base + t];
            if ((e >> 30) == 0 && !(e & 0x8000u)) {
                const int nb = (int)(((e >> 16) & 0x3ff) + 3) >> 2;
                const uint8_t* src = P->arms2 + P->arm_off[a0 + (uint32_t)stack[t]];
                uint8_t* dst = armbuf + (e & 0x3fff);
                for (int b = 0; b < nb; ++b) dst[b] = src[b];
            }
        }
        g.sync();
        *n_seq_out = narm + base;
        *added_out = any_len;
        return RES_OK;
    }

    // The same table in one pass over the arms (SHORT windows of the classes whose arms fit a few registers): a lane reads its
    // arm's length and offset, then the arm's packed bytes ONCE, as whole words into registers — two dependent round trips to HBM
    // for the whole window, where the passes above make five (lengths and offsets, the two byte ranges of every neighbour
    // compare, offsets and bytes again for staging) and read the bytes one by one.  Everything else happens on the registers:
    // the neighbour compare against the lane below (the last lane of the previous round for lane 0), the arm's hash, its place in
    // `armbuf` from a prefix sum over the lanes (copies take none; slots are word-aligned so that the words go to LDS as they
    // are), the stores.  What differs from the passes above is invisible to the alignment: once an arm does not fit `armbuf` no
    // later arm is staged either (the serial pass skipped it and went on), and the hash is a different function.
    static constexpr int STAGE_NB = (Cfg::LMAX + 3) / 4, STAGE_NDW = (STAGE_NB + 3) / 4;
#define HYPO_STAGE_FUSED 1
    static constexpr bool STAGE_FUSED = HYPO_STAGE_FUSED && PK && STAGE_NDW <= 8;
    HD static uint32_t load_u32(const uint8_t* p) {
#ifdef HYPO_EMU
        uint32_t v; __builtin_memcpy(&v, p, 4); return v;
#else
        typedef uint32_t __attribute__((aligned(1))) u32u;
        return *(const u32u*)p;
#endif
    }
    HD int build_seqtab_fused(const HypoWindow& W, int base, int narm, int* n_seq_out, bool* added_out) {
        constexpr int NDW = STAGE_NDW;
        const uint32_t a0 = W.first_arm;
        const int ni = (int)W.n_internal, np = (int)W.n_prefix;
        bool over = false, any_len = false, bad = false;
        uint32_t armb = 0;
        uint32_t prev_e = 0, prev_d[NDW];                      // the arm before this round's first (group-uniform)
        HYPO_UNROLL
        for (int i = 0; i < NDW; ++i) prev_d[i] = 0;
        int run = 0;                                           // bytes of armbuf taken so far (group-uniform)
        for (int t0 = 0; t0 < narm; t0 += GW) {
            const int t = t0 + g.lane;
            const bool on = t < narm;
            uint32_t e0 = 0, d[NDW];
            HYPO_UNROLL
            for (int i = 0; i < NDW; ++i) d[i] = 0;
            int nb = 0;
            if (on) {
                int a, mode; bool head, tail;                  // consumption slot t -> arm index (Window.cpp:87-130)
                if (t < ni) { a = t; mode = MODE_NW; head = true; tail = true; }
                else if (t < ni + np) { a = ni + (np - 1 - (t - ni)); mode = MODE_LOV; head = true; tail = false; }
                else { a = t; mode = MODE_ROV; head = false; tail = true; }
                uint32_t len = P->arm_len[a0 + a];
                const uint64_t ao = P->arm_off[a0 + a], ab = P->arms2_bytes;
                if (ao > ab || ((uint64_t)len + 3) / 4 > ab - ao) { bad = true; len = 0; }
                else if (len + 2u > (uint32_t)Cfg::LMAX) { over = true; len = 0; }
                if (len) any_len = true;
                nb = (int)((len + 3) >> 2);
                armb += (uint32_t)nb;
                e0 = seq_ent((uint32_t)a, len, head, tail, mode, false, 1);
                if (nb) {
                    const uint8_t* p = P->arms2 + ao;
                    const int ndw = (nb + 3) >> 2;
                    if ((uint64_t)ndw * 4 <= ab - ao) {        // whole words (the last one may reach past the arm, never past the buffer)
                        HYPO_UNROLL
                        for (int i = 0; i < NDW; ++i) if (i < ndw) d[i] = load_u32(p + 4 * i);
                    } else {
                        HYPO_UNROLL
                        for (int i = 0; i < NDW; ++i) {
                            HYPO_UNROLL
                            for (int j = 0; j < 4; ++j) if (4 * i + j < nb) d[i] |= (uint32_t)p[4 * i + j] << (8 * j);
                        }
                    }
                    if (nb & 3) {                              // bytes behind the arm do not count
                        const uint32_t keep = (1u << (8 * (nb & 3))) - 1u;
                        HYPO_UNROLL
                        for (int i = 0; i < NDW; ++i) if (i == ndw - 1) d[i] &= keep;
                    }
                }
            }
            // copy of the arm before it: same length, markers and mode, same bytes
            const uint32_t pe = g.shfl_up1(e0, prev_e);
            bool same = on && t > 0 && nb > 0 && ((e0 ^ pe) & 0x3fff0000u) == 0;
            HYPO_UNROLL
            for (int i = 0; i < NDW; ++i) {
                const uint32_t pd = g.shfl_up1(d[i], prev_d[i]);
                same &= pd == d[i];
            }
            // place in armbuf: copies take none, the others a word-aligned slot while the arms so far fit
            const int slot = (nb + 3) & ~3;
            const int incl = run + g.scan_add_incl(on && !same ? slot : 0);
            const bool staged = on && incl <= Cfg::ARMBYTES;
            const int off = incl - slot;
            if (on) {
                seqtab[base + t] = staged ? ((e0 & 0x3fff0000u) | (uint32_t)off | (same ? 0x8000u : 0u)) : (e0 | (same ? 0x8000u : 0u));
                if (staged && !same && nb) {
                    uint32_t* const dst = (uint32_t*)(armbuf + off);
                    const int ndw = (nb + 3) >> 2;
                    HYPO_UNROLL
                    for (int i = 0; i < NDW; ++i) if (i < ndw) dst[i] = d[i];
                }
            }
            if (t0 + GW < narm) {                              // what the next round's first lane compares with
                prev_e = (uint32_t)g.shfl((int)e0, GW - 1);
                HYPO_UNROLL
                for (int i = 0; i < NDW; ++i) prev_d[i] = (uint32_t)g.shfl((int)d[i], GW - 1);
                run = g.shfl(incl, GW - 1);
            }
        }
        if (g.any(bad)) return RES_INVALID;
        if (g.any(over)) return RES_OVERFLOW;
        any_len = g.any(any_len);
        if constexpr (Hook::enabled) { armb = (uint32_t)g.reduce_add((int)armb); stat_set(ST_ARMB, armb); }
        g.sync();
        *n_seq_out = narm + base;
        *added_out = any_len;
        return RES_OK;
    }

    HD int load_seq(const HypoWindow& W, int s, int* mode_out) {
        const uint32_t e = seqtab[s];
        const bool head = (e >> 26) & 1, tail = (e >> 27) & 1;
        const int mc = (int)((e >> 28) & 3), where = (int)(e >> 30);
        const bool four = where == 2;
        *mode_out = mc == 0 ? MODE_NW : (mc == 1 ? MODE_LOV : MODE_ROV);
        const int len = (int)((e >> 16) & 0x3ff);
        if (len == 0) { L = 0; return RES_OK; }
        L = g.uniform(len + (head ? 1 : 0) + (tail ? 1 : 0));   // group-uniform by construction; tells the compiler (scalar control flow for 64-lane groups)
        if (L > Cfg::LMAX) return RES_OVERFLOW;
        const uint8_t* p;
        if (where == 0) p = armbuf + (e & 0x3fff);
        else if (where == 1) p = P->arms2 + P->arm_off[W.first_arm + (e & 0x3fff)];
        else p = P->draft4 + W.draft_off;
        for (int t = g.lane; t < L; t += GW) {
            int c;
            if (head && t == 0) c = C_J;
            else if (tail && t == L - 1) c = C_O;
            else {
                const int b = t - (head ? 1 : 0);
                if (four) { c = (p[b >> 1] >> (4 - 4 * (b & 1))) & 15; c = c < 4 ? c : C_N; }
                else c = (p[b >> 2] >> (6 - 2 * (b & 3))) & 3;
            }
            seq[t] = (uint8_t)c;
        }
        g.sync();
        HYPO_TICK(PH_LOAD);
        return RES_OK;
    }

    // ---- per-row metadata in rank order (rebuilt only when the graph topology changed) -----------
    // rowmeta[r]: bits 0-2 node code, 3 SLOW, 4 SAVE, 5 sink, 8-15 in-degree (8-11 in the packed classes), 16-18 code again (the
    // packed row loop XORs the word, masked, with a pair of sequence codes), 19-30 matrix row of pred 0 (further preds are rare
    // and looked up through pred_row()), 31 DEEP (hybrid classes).  Packed classes (<= 254 nodes, <= 7 in-edges) also keep the
    // ring distances of preds 0 and 1 here: bits 27-31 and bits 12-15 + 6.
    // Packed classes: a row is FAST when its only predecessor is the previous row (whose scores are still in registers), it is
    // no end-cell candidate and no later row reads it from the ring; everything else is SLOW.  Only rows that a later row
    // reads as something other than "pred 0 = the previous row" are SAVEd to the ring, in order: the ring then has to span
    // the SAVEd rows between a row and its furthest predecessor (maxdelta), not all rows in between.  sidx[r] = number of
    // SAVEd rows before row r.
    static constexpr uint32_t META_SLOW = 8u, META_SAVE = 16u, META_SINK = 32u;
    static constexpr uint32_t META_DEEP = 0x80000000u;     // rowmeta bit 31: a row further than RING1 ahead reads this row
    HD static int meta_code(uint32_t meta) { return (int)(meta & 7u); }
    HD static int meta_k(uint32_t meta) { return (int)((meta >> 8) & (PK ? 0xfu : 0xffu)); }
    // packed classes: how many SAVEd rows back the rows of pred 0 / pred 1 sit in the ring (0 = not read from the ring)
    static constexpr int RING_BACK_MAX = 31;
    HD static int meta_back0(uint32_t meta) { return (int)(meta >> 27); }
    HD static int meta_back1(uint32_t meta) { return (int)(((meta >> 12) & 0xfu) | ((meta >> 2) & 0x10u)); }
    HD static uint32_t meta_backs(int b0, int b1) { return ((uint32_t)b0 << 27) | (((uint32_t)b1 & 0xfu) << 12) | (((uint32_t)b1 & 0x10u) << 2); }
    HD static bool meta_sink(uint32_t meta) { return (meta & META_SINK) != 0; }
    // (the packed classes keep the node code a second time in bits 16-18; the others use those bits for the row: 15 bits)
    static constexpr int P0_SHIFT = PK ? 19 : 16;
    HD static int meta_p0(uint32_t meta) { return (int)((meta >> P0_SHIFT) & (PK ? 0xffu : 0x7fffu)); }
    static_assert(NMAX <= (PK ? 255 : 32767) && (!PK || KIN <= 15), "field widths of rowmeta");
    HD void build_rowmeta() {
        int md = 0;
        for (int r = g.lane; r < n_nodes; r += GW) {
            const int u = r2n[r];
            const int k = nin[u];
            int p0 = 0;
            for (int p = 0; p < k; ++p) {
                const int pr = (int)n2r[inp[u * KIN + p]] + 1;
                if (PRED_TABLE) predrows[r * KIN + p] = (id_t)pr;
                if (p == 0) p0 = pr;
                const int d = r + 1 - pr;
                md = d > md ? d : md;
            }
            const bool sink = n_out(u) == 0;
            const bool slow = !(k == 1 && p0 == r) || sink;
            const uint32_t c = code[u];
            rowmeta[r] = c | (slow ? META_SLOW : 0u) | (sink ? META_SINK : 0u) | ((PK && n_out(u) == 1) ? META_OUT1 : 0u) | ((uint32_t)k << 8) | (PK ? (c << 16) : 0u) | ((uint32_t)p0 << P0_SHIFT);
        }
        meta_dirty = false;
        if (PK) {
            g.sync();
            // Plain read-modify-write on purpose (also below for DEEP): every writer ORs the same bits into a word nobody else
            // changes in this pass, so colliding lanes store identical values.
            for (int r = g.lane; r < n_nodes; r += GW) {
                const int k = meta_k(rowmeta[r]);
                for (int p = 0; p < k; ++p) {
                    const int pr = pred_row(r, p);
                    if (pr > 0 && !(p == 0 && pr == r)) rowmeta[pr - 1] |= META_SAVE | META_SLOW;
                }
            }
            g.sync();
            int total = 0;
            for (int base = 0; base < n_nodes; base += GW) {
                const int r = base + g.lane;
                const bool sv = r < n_nodes && (rowmeta[r] & META_SAVE);
                const uint64_t b = g.ballot(sv);
                if (r < n_nodes) sidx[r] = (uint8_t)(total + popc64(b & ((1ull << g.lane) - 1ull)));
                total += popc64(b);
            }
            g.sync();
            // ring distances of preds 0 and 1 into rowmeta[r] (0 = not in the ring: the previous row as pred 0, or the virtual
            // source row); the row loop needs no other lookup for these preds
            int mds = 0;
            for (int r = g.lane; r < n_nodes; r += GW) {
                const int k = meta_k(rowmeta[r]);
                int b0 = 0, b1 = 0;
                for (int p = 0; p < k; ++p) {
                    const int pr = pred_row(r, p);
                    if (pr > 0 && !(p == 0 && pr == r)) {
                        const int d = (int)sidx[r] - (int)sidx[pr - 1];
                        mds = d > mds ? d : mds;
                        if (p == 0) b0 = d; else if (p == 1) b1 = d;
                    }
                }
                if (b0 > RING_BACK_MAX || b1 > RING_BACK_MAX) mds = 1 << 20;       // does not fit the fields: the window moves up a class
                else rowmeta[r] |= meta_backs(b0, b1);
            }
            stat_set(ST_MAXD, (uint32_t)g.reduce_max(mds));
            g.sync();
            return;
        }
        const int maxdelta = g.reduce_max(md);
        stat_set(ST_MAXD, (uint32_t)maxdelta);
        g.sync();
        if (Cfg::RING1 > 0 && maxdelta > Cfg::RING1) {
            // hybrid classes: only rows that some later row reads from further back than the LDS ring reaches go to the HBM ring
            for (int r = g.lane; r < n_nodes; r += GW) {
                const int k = meta_k(rowmeta[r]);
                for (int p = 0; p < k; ++p) {
                    const int pr = pred_row(r, p);
                    // plain read-modify-write: an atomic would execute in L2 and leave a stale copy of the word in this CU's
                    // vector L1, which the row loop's ordinary loads could then hit (seen on hardware, never in the emulator)
                    if (pr > 0 && r + 1 - pr > Cfg::RING1) rowmeta[pr - 1] |= META_DEEP;
                }
            }
            g.sync();
        }
    }
    static constexpr bool PRED_TABLE = Cfg::PATHCAP > 0;    // the HBM-scratch classes tabulate pred rows in build_rowmeta
    HD int pred_row(int r, int p) const {                   // matrix row of pred p of rank r
        if (PRED_TABLE) return (int)predrows[r * KIN + p];
        return (int)n2r[inp[(int)r2n[r] * KIN + p]] + 1;
    }
    HD void load_ring_at(const score_t* base, int off, int S, int (&out)[CPL]) const {   // off = slot * S
        if (CPL * g.lane < S) {
            const Pack pk = *(const Pack*)(base + off + CPL * g.lane);
            HYPO_UNROLL
            for (int c = 0; c < CPL; ++c) out[c] = (int)pk.v[c];
        } else {
            HYPO_UNROLL
            for (int c = 0; c < CPL; ++c) out[c] = NEG;
        }
    }
    HD int read_dir(int cell) const {                      // cell = row_index * S + column
        if (NIB) { const int b = dir[cell >> 1]; return (cell & 1) ? (b >> 4) : (b & 15); }
        return dir[cell];
    }

    // ---- engine->align (sisd_alignment_engine.cpp:246-439), linear gaps -----------------------------
    // Leaves posnode[q] (graph node aligned to sequence position q, -1 = insertion) for q in
    // ---- the row loop on packed pairs of int16 columns (classes with int16 rows, nibble codes, even CPL) ------------
    // Same recurrence, same tie rules and same direction codes as the loop in align(); two neighbouring columns share one
    // register and every select is arithmetic (t = min_u16(a - b, 1) is 0 where a == b and 1 where a > b), so a row costs
    // about 2/3 of the one-column-per-register form.  Values stay exact because align() admits only windows whose scores
    // and `H - j*g` terms fit 16 bits.  NEG16 + a score never wraps and stays below every real cell.
    struct alignas(pow2_of(CPL * 2) < 4 ? 4 : pow2_of(CPL * 2)) PackP { P2 v[CPL / 2 ? CPL / 2 : 1]; };
    // Two bodies per row.  FAST (kNW, one predecessor = the previous row, not a sink, not read from the ring later: ~95 % of
    // the rows): everything the row needs is in registers or a constant, the only memory operation is the store of its
    // direction codes.  SLOW: any predecessors (ring reads located through sidx[]), end-cell bookkeeping, ring save.
    static constexpr int PK_TIECAP = 47;                    // rows remembered as tied for the end row (lazy rank order); count and list live in posnode[] during the row loop: tie[0] = count, tie[1 ..] = rows
    HD int rows_pk(int mode, int m, int n, int gp, int S, int R, int* ntie_out) {
        constexpr int NP = CPL / 2;
        const int j0 = CPL * g.lane;
        int amax = m < 0 ? -m : m; { const int b = n < 0 ? -n : n, c2 = gp < 0 ? -gp : gp; amax = amax > b ? amax : b; amax = amax > c2 ? amax : c2; }
        const int NEG16 = -32768 + amax;
        P2 SQ[NP], JG[NP], LAST[NP];
        HYPO_UNROLL
        for (int q = 0; q < NP; ++q) {
            const int j = j0 + 2 * q;
            const int s0 = (j >= 1 && j <= L) ? (int)seq[j - 1] : (int)C_NONE;
            const int s1 = (j + 1 <= L) ? (int)seq[j] : (int)C_NONE;
            SQ[q] = pk_make(s0, s1);
            JG[q] = pk_make(j * gp, (j + 1) * gp);           // row 0: H[0][j] = j*g (sisd..cpp:197-199,230-232)
            LAST[q] = JG[q];
        }
        int vM = pk_bits(pk_splat(m)), vMN = pk_bits(pk_splat(n - m)), vGP = pk_bits(pk_splat(gp)), vONE = pk_bits(pk_splat(1));
        HYPO_IN_VGPR(vM); HYPO_IN_VGPR(vMN); HYPO_IN_VGPR(vGP); HYPO_IN_VGPR(vONE);
        const P2 M = pk_from_bits(vM), MN = pk_from_bits(vMN), GP = pk_from_bits(vGP), ONE = pk_from_bits(vONE);
        // direction codes of a FAST row, the high column's already shifted into the upper nibble:
        // code = 15 (FAST) - 8 tD + 7 tD tU with tD = [v > D], tU = [v > U]  ->  15 diagonal, 7 vertical via pred 0, 14 horizontal
        int vK7 = pk_bits(pk_make(7, 7 << 4)), vKM8 = pk_bits(pk_make(-8, -(8 << 4))), vK15 = pk_bits(pk_make(15, 15 << 4));
        HYPO_IN_VGPR(vK7); HYPO_IN_VGPR(vKM8); HYPO_IN_VGPR(vK15);
        const P2 K7 = pk_from_bits(vK7), KM8 = pk_from_bits(vKM8), K15 = pk_from_bits(vK15);
        static_assert(DIR_FAST == 15 && DIR_HORIZ == 14, "fast-row code constants");
        const int negfill = pk_bits(pk_splat(NEG16));
        // kROV: first column is 0 (sisd..cpp:200-211,237-239): lane 0 clears the low half of its first pair
        const int keep0 = (g.lane == 0 && mode == MODE_ROV) ? (int)0xffff0000u : -1;

        const int le = L / CPL, ce = L % CPL;               // owner of the last column
        const int ce_shift = 16 * (ce & 1);
        int best = NEG, best_i = -1;
        int16_t* const tie = posnode;                        // (the guide has been tried by now; the traceback writes posnode[] after the ties are settled)
        const bool ties = Cfg::LAZY && lazy_on;
        static_assert((PK_TIECAP + 1) * 2 <= Lay::POS_BYTES, "the tie list fits posnode");
        if (ties && g.lane == 0) tie[0] = 0;
        g.sync();
        // native kLOV flavour: a row's end value is its maximum over columns 1..L (every lane then holds the same value)
        const bool native_lov = mode == MODE_LOV && (P->flags & POA_NATIVE_KLOV) != 0;
        auto end_value = [&](const P2 (&v)[NP]) -> int {
            if (native_lov) {
                int mx = NEG;
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {
                    const int c0 = j0 + 2 * q;
                    if (c0 >= 1 && c0 <= L) { const int x = pk_lo(v[q]); mx = x > mx ? x : mx; }
                    if (c0 + 1 >= 1 && c0 + 1 <= L) { const int x = pk_hi(v[q]); mx = x > mx ? x : mx; }
                }
                return g.reduce_max(mx);
            }
            int w = pk_bits(v[0]);
            HYPO_UNROLL
            for (int q = 1; q < NP; ++q) if (ce / 2 == q) w = pk_bits(v[q]);
            return (int)(int16_t)(uint16_t)((uint32_t)w >> ce_shift);     // column L: meaningful in lane `le` only
        };
        int wslotS = 0, scount = 0, rowS = 0;               // ring write slot * S, rows saved so far, r * S
        const int RS = R * S;
        const bool lov = mode == MODE_LOV;                   // kLOV: every row's column L is an end-cell candidate
        // loop-carried results of the two lane shifts: lane 0 of a group is never written, it keeps the fill value
        int nbreg = negfill, exreg = (int)0x80000000;
        constexpr bool META_CHUNKED = (GW == 64);
        uint32_t mchunk = 0u;
        uint32_t meta_a = META_CHUNKED ? 0u : rowmeta[0];
        uint32_t meta_b = (!META_CHUNKED && n_nodes > 1) ? rowmeta[1] : 0u;
        auto load_pk = [&](int off, P2 (&out)[NP]) {
            if (j0 < S) {
                const PackP pk = *(const PackP*)(ring + off + j0);
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) out[q] = pk.v[q];
            } else {
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) out[q] = pk_from_bits(negfill);
            }
        };
        // ring offset of the row saved `back` SAVEd rows ago
        auto ring_back = [&](int back) -> int {
            int ps = wslotS - back * S;
            return ps < 0 ? ps + RS : ps;
        };
        // horizontal term H[i][j] = max(H[i][j], H[i][j-1] + g): prefix max of H[i][j] - j*g.  The lane scan compares whole
        // registers: the running maximum sits in the high half and decides, the low half only orders equal maxima and is
        // dropped afterwards; INT_MIN, the scan's identity, reads as -32768 there
        auto hscan = [&](P2 (&v)[NP]) {
            P2 x[NP];
            x[0] = pk_fold_hi(pk_sub(v[0], JG[0]));
            HYPO_UNROLL
            for (int q = 1; q < NP; ++q) x[q] = pk_fold_hi(pk_max(pk_sub(v[q], JG[q]), pk_hi_splat(x[q - 1])));
            exreg = g.scan_max_excl_c(pk_bits(x[NP - 1]), exreg);
            const P2 EX = pk_hi_splat(pk_from_bits(exreg));
            HYPO_UNROLL
            for (int q = 0; q < NP; ++q) v[q] = pk_add(pk_max(x[q], EX), JG[q]);
        };
        for (int r = 0; r < n_nodes; ++r) {
            const int i = r + 1;
            uint32_t meta;
            if (META_CHUNKED) {
                if ((r & 63) == 0) { mchunk = r + g.lane < n_nodes ? rowmeta[r + g.lane] : 0u; HYPO_ARRIVED(mchunk); }
                meta = (uint32_t)g.shfl((int)mchunk, r & 63);
            } else {
                meta = meta_a;
                meta_a = meta_b;
                if (r + 2 < n_nodes) meta_b = rowmeta[r + 2];
            }
            if (!(meta & META_SLOW)) {
                // ---- FAST row ----
                const P2 CD = pk_from_bits((int)(meta & 0x00070007u));
                const int nb = nbreg = g.shfl_up1(pk_bits(LAST[NP - 1]), nbreg);
                P2 D[NP], U[NP], v[NP];
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {
                    const P2 MV = pk_mad(pk_minu(pk_xor(SQ[q], CD), ONE), MN, M);      // match / mismatch score per column
                    D[q] = pk_add(pk_shift_in(q ? LAST[q - 1] : pk_from_bits(nb), LAST[q]), MV);
                    U[q] = pk_add(LAST[q], GP);
                    v[q] = pk_max(D[q], U[q]);
                }
                v[0] = pk_from_bits(pk_bits(v[0]) & keep0);
                hscan(v);
                // the reference's traceback preference (sisd..cpp:370-428): diagonal, else vertical, else horizontal.  Lanes
                // beyond the row's width store into the slack behind the row (PoaLayout::oRowmeta).
                uint32_t codes = 0;
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {
                    const P2 tD = pk_minu(pk_sub(v[q], D[q]), ONE), tU = pk_minu(pk_sub(v[q], U[q]), ONE);
                    const uint32_t b = (uint32_t)pk_bits(pk_mad(tD, pk_mad(tU, K7, KM8), K15));
                    codes |= ((b | (b >> 16)) & 0xffu) << (8 * q);
                    LAST[q] = v[q];
                }
                uint8_t* dst = dir + (rowS >> 1) + (j0 >> 1);         // S and j0 are even
                if (NP == 1) *dst = (uint8_t)codes;
                else if (NP == 2) *(uint16_t*)dst = (uint16_t)codes;
                else *(uint32_t*)dst = codes;                // NP == 4 (NP == 3 is not instantiated)
                if (lov) {                                   // end cell: first strictly greater in rank order; only lane `le` counts
                    HYPO_NO_IFCVT();
                    const int val = end_value(v);
                    if (ties && g.lane == le) {              // lazy rank order: rows that tie for the end row are remembered
                        if (val > best) { tie[0] = 1; tie[1] = (int16_t)i; }
                        else if (val == best) { const int nt = tie[0]; if (nt < PK_TIECAP) tie[1 + nt] = (int16_t)i; tie[0] = (int16_t)(nt + 1); }
                    }
                    best_i = val > best ? i : best_i;
                    best = val > best ? val : best;
                }
                rowS += S;
                g.sync();
                continue;
            }
            // ---- SLOW row ----
            HYPO_NO_IFCVT();
            HYPO_DIAG(rows_slow += 1);
            const int cd = meta_code(meta), k = meta_k(meta);
            const int p0 = meta_p0(meta);                    // 0 when k == 0 (virtual source row)
            const bool fastrow = p0 == i - 1;
            const int fastcode = fastrow ? (int)DIR_FAST : dir_diag(0);
            P2 MV[NP];                                       // match / mismatch score per column
            {
                const P2 CD = pk_splat(cd);
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) MV[q] = pk_mad(pk_minu(pk_xor(SQ[q], CD), ONE), MN, M);
            }
            P2 D[NP], U[NP], cD[NP], cU[NP];
            {
                P2 hp[NP];
                if (fastrow) { HYPO_UNROLL for (int q = 0; q < NP; ++q) hp[q] = LAST[q]; }
                else if (p0 == 0) { HYPO_UNROLL for (int q = 0; q < NP; ++q) hp[q] = JG[q]; }
                else load_pk(ring_back(meta_back0(meta)), hp);
                const int nb = nbreg = g.shfl_up1(pk_bits(hp[NP - 1]), nbreg);
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {
                    D[q] = pk_add(pk_shift_in(q ? hp[q - 1] : pk_from_bits(nb), hp[q]), MV[q]);
                    U[q] = pk_add(hp[q], GP);
                }
            }
            if (k > 1) {                                     // several predecessors: remember which one reaches each maximum first
                P2 pD[NP], pU[NP];
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) { pD[q] = pk_splat(0); pU[q] = pk_splat(0); }
                for (int p = 1; p < k; ++p) {
                    P2 hp[NP];
                    // pred 1: distance from rowmeta; beyond that (0.02 % of the nodes) through the graph tables
                    const int back = p == 1 ? meta_back1(meta) : scount - (int)sidx[g.uniform(pred_row(r, p)) - 1];
                    load_pk(ring_back(back), hp);
                    const int nb = nbreg = g.shfl_up1(pk_bits(hp[NP - 1]), nbreg);
                    const P2 PP = pk_splat(p);
                    HYPO_UNROLL
                    for (int q = 0; q < NP; ++q) {
                        const P2 d = pk_add(pk_shift_in(q ? hp[q - 1] : pk_from_bits(nb), hp[q]), MV[q]);
                        const P2 u = pk_add(hp[q], GP);
                        const P2 nd = pk_max(D[q], d), nu = pk_max(U[q], u);
                        // strict: the first pred reaching the maximum wins
                        pD[q] = pk_mad(pk_minu(pk_sub(nd, D[q]), ONE), pk_sub(PP, pD[q]), pD[q]);
                        pU[q] = pk_mad(pk_minu(pk_sub(nu, U[q]), ONE), pk_sub(PP, pU[q]), pU[q]);
                        D[q] = nd; U[q] = nu;
                    }
                }
                const P2 FC = pk_splat(fastcode), V0 = pk_splat(dir_vert(0));
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {               // nibble codes: dir_diag(p) = p, dir_vert(p) = dir_vert(0) + p
                    cD[q] = pk_mad(pk_sub(ONE, pk_minu(pD[q], ONE)), FC, pD[q]);
                    cU[q] = pk_add(pU[q], V0);
                }
            } else {
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) { cD[q] = pk_splat(fastcode); cU[q] = pk_splat(dir_vert(0)); }
            }
            P2 v[NP];
            HYPO_UNROLL
            for (int q = 0; q < NP; ++q) v[q] = pk_max(D[q], U[q]);
            v[0] = pk_from_bits(pk_bits(v[0]) & keep0);
            hscan(v);
            if (j0 < S) {
                const P2 HZ = pk_splat(DIR_HORIZ);
                uint32_t codes = 0;
                PackP pk;
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {
                    const P2 tD = pk_minu(pk_sub(v[q], D[q]), ONE), tU = pk_minu(pk_sub(v[q], U[q]), ONE);
                    const P2 dc = pk_mad(tD, pk_mad(tU, pk_sub(HZ, cU[q]), pk_sub(cU[q], cD[q])), cD[q]);
                    const uint32_t b = (uint32_t)pk_bits(dc);
                    codes |= ((b | (b >> 12)) & 0xffu) << (8 * q);
                    pk.v[q] = v[q];
                }
                uint8_t* dst = dir + (rowS >> 1) + (j0 >> 1);         // S and j0 are even
                if (NP == 1) *dst = (uint8_t)codes;
                else if (NP == 2) *(uint16_t*)dst = (uint16_t)codes;
                else *(uint32_t*)dst = codes;
                if (meta & META_SAVE) *(PackP*)(ring + wslotS + j0) = pk;
            }
            if (meta & META_SAVE) { scount += 1; wslotS = wslotS + S == RS ? 0 : wslotS + S; }
            rowS += S;
            HYPO_UNROLL
            for (int q = 0; q < NP; ++q) LAST[q] = v[q];
            // end cell: first strictly greater in rank order (sisd..cpp:279-288,332-339)
            if (mode == MODE_LOV || meta_sink(meta)) {
                const int val = end_value(v);                // (all lanes keep score; lane `le` is the one that is read)
                if (ties && g.lane == le) {
                    if (val > best) { tie[0] = 1; tie[1] = (int16_t)i; }
                    else if (val == best) { const int nt = tie[0]; if (nt < PK_TIECAP) tie[1 + nt] = (int16_t)i; tie[0] = (int16_t)(nt + 1); }
                }
                best_i = val > best ? i : best_i;
                best = val > best ? val : best;
            }
            g.sync();
        }
        g.sync();
        *ntie_out = ties ? (int)tie[0] : 0;
        return g.shfl(best_i, le);
    }

    // ---- exact threading: the alignment of a sequence that spells a path of the graph, without scores ----------------------
    // With m > 0, n < m and g < 0 no cell can exceed H[i][j] <= m * j, and H[i][j] == m * j exactly when some path that ends
    // in node i spells seq[0..j) with j matches and nothing else (kNW / kLOV: starting at a node without in-edges, because
    // the first column costs g per node, sisd..cpp:200-211; kROV: starting anywhere, the first column being 0, :237-239).
    // Call such a cell PERFECT.  perfect(i, j) = letter(i) == seq[j-1] and perfect(p, j-1) for some pred p of i.  If an end-cell
    // candidate (kNW / kROV: a sink, kLOV: any node; column L) is perfect, its score m * L is the maximum, so the reference
    // starts its traceback at the FIRST such row in rank order (strictly-greater rule, sisd..cpp:279-288,332-339) and at
    // every perfect cell takes the diagonal through the first pred (in-edge order) whose cell (p, j-1) is perfect: diagonals
    // are tried before anything else (:370-428) and H[p][j-1] + m == m * j holds for exactly those preds.
    //
    // Computed COLUMN BY COLUMN with lanes = graph ranks (rounds 2-4 walked the rows in rank order, through the ring, 200-850
    // cycles a row): the perfect cells of column j are a set of ranks F_j, one bit per rank in XW 64-bit words, and
    //   F_1     = { sources (kROV: any node) whose letter is seq[0] }        (perfect(p, 0) holds for the virtual row 0 only; kROV: for every row)
    //   F_{j+1} = { v : letter(v) == seq[j] and a pred of v is in F_j }.
    // For a CHAIN rank (one in-edge, from the rank before it: ~90 % of a window's graph) "a pred is in F_j" is the mask shifted
    // by one, so a column costs a letter compare per owned rank, a few mask operations and no memory access at all; the other
    // ranks test their in-edge sources bit by bit, and only in columns where F_j holds a rank that has an out-edge other than
    // "the chain link to the next rank" (NCP).  The attempt ends in the column where F runs empty: a sequence that spells no path
    // costs the columns up to its first error.  Lane c - 1 keeps F_c; the alignment is then read off without a traceback:
    // the end row is the first rank of F_L (kNW / kROV: among the sinks), a column with ONE perfect cell must be that cell
    // (every cell of the walk is perfect), and the few columns with several (the first ones of a kROV arm, repeats inside a
    // bubble) take the first in-edge source, in order, of the cell chosen to their right that is in their F.
    // Leaves posnode[] (node of every position), tb_steps = L, tb_fv = 0 and returns 1; returns 0 when no end-cell candidate
    // is perfect (the caller runs the score rows).
    static constexpr int XRPL = (NMAX + GW - 1) / GW;       // ranks per lane: rank = q * GW + lane
    static constexpr int XSL = (Cfg::LMAX + GW - 1) / GW;   // columns per lane: column c = 1 + t * GW + lane
    static constexpr int XW = (XRPL * GW + 63) / 64;        // 64-bit words of a rank set (bit r of the concatenation = rank r)
    typedef typename std::conditional<(KIN <= 4), uint32_t, uint64_t>::type xpw_t;      // a rank's in-edge sources, a byte each
    static constexpr uint32_t META_OUT1 = 0x80u;            // rowmeta bit 7 (packed classes): the node has exactly one out-edge
    struct XSet { uint64_t w[XW]; };
    HD static bool x_in(const XSet& F, int rank) {
        uint64_t w = F.w[0];
        HYPO_UNROLL
        for (int t = 1; t < XW; ++t) if ((rank >> 6) == t) w = F.w[t];
        return ((w >> (rank & 63)) & 1ull) != 0;
    }
    HD static bool x_any(const XSet& F) { uint64_t o = F.w[0]; HYPO_UNROLL for (int t = 1; t < XW; ++t) o |= F.w[t]; return o != 0; }
    HD static int x_first(const XSet& F) {                  // lowest rank of a non-empty set
        int e = 0;
        HYPO_UNROLL
        for (int t = XW - 1; t >= 0; --t) if (F.w[t] != 0) e = t * 64 + ctz64(F.w[t]);
        return e;
    }
    // the set of ranks whose owner says yes for them (pred[q] of lane l speaks for rank q * GW + l)
    HD XSet x_ballot(const bool (&pred)[XRPL]) const {
        XSet S;
        HYPO_UNROLL
        for (int t = 0; t < XW; ++t) S.w[t] = 0;
        HYPO_UNROLL
        for (int q = 0; q < XRPL; ++q) S.w[(q * GW) / 64] |= g.ballot(pred[q]) << ((q * GW) % 64);
        return S;
    }
    HD XSet x_shfl(const XSet& v, int src) const {
        XSet r;
        HYPO_UNROLL
        for (int t = 0; t < XW; ++t) {
            const uint32_t lo = (uint32_t)g.shfl((int)(uint32_t)v.w[t], src), hi = (uint32_t)g.shfl((int)(uint32_t)(v.w[t] >> 32), src);
            r.w[t] = ((uint64_t)hi << 32) | lo;
        }
        return r;
    }
    // ---- threading along a guide: the path of the sequence before ----------------------------------------------------------
    // The arms of a window are reads of one locus: an arm differs from the arm before it in a base or two, and posnode[] still
    // holds that arm's path (the node of every position; Poa::add_alignment leaves it complete).  So the path this arm would
    // spell is known up to those bases — position q sits on the guide's node for it (kNW: position q, kROV: counted from the
    // end), or, where the letters differ, on the member of that node's aligned clique that carries the arm's letter — and all
    // that is left is to CHECK it, every position at once (lanes = positions, a handful of dependent LDS reads):
    //   * every node carries its position's letter and every pair of neighbours is joined by an in-edge (the path exists);
    //   * kNW: the first node has no in-edges (Poa::thread_cols' comment: only then is its cell perfect); the last node is a sink;
    //   * the reference's traceback is FORCED along this path: the last node is the only sink with the last letter (so it
    //     is the only end-cell candidate that can be perfect), and at every node the path's in-edge is the FIRST, in in-edge
    //     order, whose source carries the letter of the position before (a perfect predecessor must carry it; the walk takes
    //     the first perfect one; the path's own is perfect, so nothing behind it in the list matters).
    //     kLOV ends anywhere: see the backward check below.
    // Returns 1: threaded (posnode[] is the alignment, the edge weights along it are already incremented); 0: cannot tell
    // (no guide of this shape; a forced-ness check failed) — the caller asks Poa::thread_cols; -1: along a guide of its own
    // kind the arm spells no path — the caller goes to the score rows, which are exact whatever was tried before them.
#ifdef HYPO_EMU_DBG
#define DBGR(i) do { const int dbgr_i = (i); if (g.lane == 0) g_dbg_reason[dbgr_i]++; } while (0)
#else
#define DBGR(i) do { } while (0)
#endif
    // ---- one substitution off the guide --------------------------------------------------------------------------------------
    // A kNW arm as long as its guide whose letters the guide's nodes (or their aligned cliques) carry at every position but
    // ONE, q0: the alignment A* that walks the guide's path with a mismatch on the guide's own node x at q0 scores
    // m (L - 1) + n, i.e. it is m - n short of perfect.  With gp < n and m - n < -2 gp (Poa::align checks; the defaults 5 / -4 / -8
    // do) anything at least as good has no horizontal step and at most ONE edit, a mismatch or a vertical step, and is perfect
    // everywhere else.  A* is the reference's alignment if no other such alignment exists, and that is checked on the graph,
    // lanes = positions, without a score:
    //   (end)    the graph has one sink, the path's last node, and no other node carries the last letter: every candidate ends
    //            in a match on that node;
    //   (forced) of the in-edge sources of the path's node at q exactly one carries the letter of q - 1 (the path's own) — at
    //            q0 + 1 none does.  Walking back from the end, a candidate therefore stays on the path for as long as its steps
    //            are perfect, and cannot pass q0 perfectly: its edit is at q0 or behind it, and everything before the edit is a
    //            perfect prefix that starts at a node without in-edges;
    //   (entrances) an edit at t >= q0 leaves the path through an in-edge source z of the path's node at t + 1: as a mismatch
    //            on z (z is not the path's own; the prefix ends at a source of z that carries the letter of t - 1), or as a
    //            vertical step over z (any z; the prefix ends at a source of z that carries the letter of t).  Either prefix is
    //            followed backwards letter by letter (Poa::back_alive) and has to DIE: a handful of nodes, one in four
    //            survives a step.  A candidate that meets the path's own node of its position behind q0 is dead too (forced:
    //            it would have to pass q0 perfectly); one that meets it at or before q0, reaches a node without in-edges in
    //            column 1, or is still alive eight steps back fails the check;
    //   (prefix) up to q0 the path is forced the same way and starts at a node without in-edges.
    // kROV arms (suffix arms: they end at the sink like kNW, their first column is free, sisd..cpp:237-239): the same walk back from
    // the end; a rival's prefix may start at ANY node, so a candidate that reaches column 1 is alive whatever its in-degree, and the
    // path's first node needs no test.  kLOV arms (prefix arms: they start like kNW and end on any node): nothing is forced from the
    // end; two searches over node sets take its place (Poa::set_alive, comment in the code below).
    // Returns 2: posnode[] is the alignment (every position aligned; Poa::add_alignment adds the new node at q0); -1: cannot
    // tell, the score rows decide (posnode[] no longer holds the guide).
    HD bool back_alive(int z, int pos, int q0, bool rov) const {
        uint64_t cur = 0; int n = 0;                         // up to four candidates, 16 bits each
        bool alive = false;
        auto expand = [&](int u, int ps, uint64_t& dst, int& dn) {
            const int c = (int)seq[ps], k = (int)nin[u], own = (int)posnode[ps];
            for (int p = 0; p < k; ++p) {
                const int sx = (int)inp[u * KIN + p];
                if ((int)code[sx] != c) continue;
                if (sx == own) { if (ps <= q0) alive = true; continue; }
                if (ps == 0) { if (rov || nin[sx] == 0) alive = true; continue; }       // (column 1 is perfect for a node without in-edges; kROV: for any node)
                bool dup = false;
                for (int i = 0; i < dn; ++i) dup = dup || (int)((dst >> (16 * i)) & 0xffffu) == sx;
                if (dup) continue;
                if (dn == 4) { alive = true; continue; }
                dst |= (uint64_t)(uint32_t)sx << (16 * dn); ++dn;
            }
        };
        expand(z, pos, cur, n);
        for (int step = 0; step < 8 && n != 0 && !alive; ++step) {
            uint64_t nx = 0; int nn = 0;
            --pos;
            for (int i = 0; i < n; ++i) expand((int)((cur >> (16 * i)) & 0xffffu), pos, nx, nn);
            cur = nx; n = nn;
        }
        return alive || n != 0;
    }
    // A set of nodes, each the last node of a would-be perfect path whose last letter sits in column j, followed backwards column
    // by column (a byte per node in the score ring, idle here).  True if some chain may be real: it reaches column 1 at a node without
    // in-edges, is still alive eight columns back, is alive in column jstop, or — skip_path — meets the path: the set then starts
    // without the path's own last node (the caller accounts for that chain), and a candidate that reaches the path's node of its
    // column is a second way to spell the letters behind it.
    HD bool set_alive(bool (&in)[XRPL], int j, int jstop, bool skip_path) {
        uint8_t* const flag = (uint8_t*)ring;
        const int jtop = j;
        for (;; --j) {
            if (skip_path) {
                const int pn = (int)posnode[j - 1];
                bool on_path = false;
                HYPO_UNROLL
                for (int q = 0; q < XRPL; ++q) if (q * GW + g.lane == pn) { on_path = in[q]; in[q] = false; }
                if (j != jtop && g.any(on_path)) return true;
            }
            bool some = false, src = false;
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) { some = some || in[q]; src = src || (in[q] && nin[q * GW + g.lane] == 0); }
            if (!g.any(some)) return false;
            if (j == 1) return g.any(src);
            if (j == jstop || jtop - j >= 8) return true;
            const int cb = (int)seq[j - 2];
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) { const int u = q * GW + g.lane; if (u < n_nodes) flag[u] = 0; }
            g.sync();
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) {
                const int u = q * GW + g.lane;
                if (in[q]) { const int k = (int)nin[u]; for (int p = 0; p < k; ++p) { const int sx = (int)inp[u * KIN + p]; if ((int)code[sx] == cb) flag[sx] = 1; } }
            }
            g.sync();
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) { const int u = q * GW + g.lane; in[q] = u < n_nodes && flag[u] != 0; }
            g.sync();
        }
    }
    HD int guided_one_sub(int q0, int mode) {                        // (posnode[] holds the path; loops kept rolled: the code is cold next to the score rows it replaces, and must not set the kernel's register count)
        const int Lu = g.uniform(L);
        q0 = g.uniform(q0);
        if (q0 < 1 || q0 >= Lu - 1) return -1;
        const bool rov = mode == MODE_ROV;                  // kROV: the first column is free (sisd..cpp:237-239) — a prefix may start at any node, the path's first node too
        const bool lov = mode == MODE_LOV;                  // kLOV: the end is free — see below
        bool bad = false;
        HYPO_NOUNROLL
        for (int q = g.lane; q < Lu; q += GW) {
            const int u = (int)posnode[q];
            const int k = (int)nin[u];
            if (q == 0) { if (!rov && k != 0) bad = true; }
            else {
                const int prev = (int)posnode[q - 1], cp = (int)seq[q - 1];
                int same = 0; bool own = false;
                for (int p = 0; p < k; ++p) {
                    const int src = (int)inp[u * KIN + p];
                    own = own || src == prev;
                    same += (int)code[src] == cp ? 1 : 0;
                }
                if (!own || same != (q == q0 + 1 ? 0 : 1)) bad = true;
                if (!bad && (lov ? q == q0 + 1 : q > q0)) {
                    for (int p = 0; p < k; ++p) {
                        const int z = (int)inp[u * KIN + p];
                        if (back_alive(z, q - 1, q0, rov)) bad = true;                 // a vertical step over z
                        if (z != prev && back_alive(z, q - 2, q0, rov)) bad = true;    // a mismatch on z
                    }
                }
            }
            if (!lov && q == Lu - 1 && n_out(u) != 0) bad = true;
        }
        if constexpr (GW >= 32) if (lov) {        // (not in class 0's four-groups-per-wave geometry: the node sets would cost it a wave per SIMD)
            // kLOV ends on any node, so nothing is forced from the end.  Instead: (i) no path from a node without in-edges spells
            // seq[0 .. q0] — every node that carries the letter of q0, followed backwards, dies — so every rival has its edit at q0 or
            // before it and is perfect behind it; (ii) the only path that spells what is behind q0 is the guide's: every OTHER node that
            // carries the last letter, followed backwards, dies before column q0 + 2 without meeting the path (a chain that joins it is a
            // second end for the same letters: 9-10-11 beside 9-11-12 in a run of C), and the path's own chain ends
            // at q0 + 1, none of whose in-edge sources carries the letter of q0 — so the edit is AT q0, on or over an in-edge source of
            // the path's node at q0 + 1 (the entrance searches above).
            if (g.any(bad)) { DBGR(12); return -1; }
            bool in[XRPL];
            const int c0 = (int)seq[q0], cl = (int)seq[Lu - 1];
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) { const int u = q * GW + g.lane; in[q] = u < n_nodes && (int)code[u] == c0; }
            if (set_alive(in, q0 + 1, 0, false)) { DBGR(12); return -1; }
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) { const int u = q * GW + g.lane; in[q] = u < n_nodes && (int)code[u] == cl; }
            if (set_alive(in, Lu, q0 + 2, true)) { DBGR(12); return -1; }
            tb_steps = Lu; tb_fv = 0;
            g.sync();
            DBGR(13);
            return 2;
        }
        int ns = 0, nl = 0;                                  // sinks; nodes that carry the last letter
        {
            const int cl = (int)seq[Lu - 1];
            for (int u0 = 0; u0 < n_nodes; u0 += GW) {
                const int u = u0 + g.lane;
                ns += popc64(g.ballot(u < n_nodes && n_out(u) == 0));
                nl += popc64(g.ballot(u < n_nodes && (int)code[u] == cl));
            }
        }
        if (g.any(bad) || ns != 1 || nl != 1) { DBGR(12); return -1; }
        tb_steps = Lu; tb_fv = 0;
        g.sync();
        DBGR(13);
        return 2;
    }
    int guide_len, guide_mode;                              // posnode[0 .. guide_len) is the path of the last sequence added (its mode); -1: none
    HD int thread_guided(int mode, bool sub_ok) {
        const int Lu = g.uniform(L), Gl = g.uniform(guide_len);
        if (Gl < Lu) { DBGR(Gl < 0 ? 1 : 2); return 0; }
        const int d = mode == MODE_ROV ? Gl - Lu : 0;       // kROV arms end where the guide ends, the others start where it starts
        if (mode == MODE_NW && Gl != Lu) { DBGR(3); return 0; }
        const bool strong = guide_mode == mode;              // the guide is anchored like this arm: a letter it cannot place is an error of the arm
        int v[XSL];                                          // node of position t * GW + lane; from the edge check on: | in-edge index << 16
        bool bad = false, amb = false;
        HYPO_UNROLL
        for (int t = 0; t < XSL; ++t) {
            const int q = t * GW + g.lane;
            v[t] = -1;
            if (q < Lu) {
                const int g0 = (int)posnode[q + d], c = (int)seq[q];
                if ((int)code[g0] == c) v[t] = g0;
                else {
                    const int ka = (int)nal[g0];
                    for (int a = 0; a < ka; ++a) { const int x = (int)al[g0 * AL + a]; if ((int)code[x] == c) v[t] = x; }
                }
                if (v[t] < 0) bad = true;
            }
        }
        // (every lane has read its guide nodes before any lane writes posnode[] below: the collectives in between are rendezvous)
        if (g.any(bad)) {
            if (HYPO_ONE_SUB && strong && sub_ok && (mode == MODE_NW || (HYPO_ONE_SUB_ROV && (mode == MODE_ROV || (GW >= 32 && mode == MODE_LOV))))) {       // a single letter the guide cannot place: Poa::guided_one_sub
                int nbad = 0, q0 = 0;
                HYPO_UNROLL
                for (int t = 0; t < XSL; ++t) {
                    const uint64_t b = g.ballot(t * GW + g.lane < Lu && v[t] < 0);
                    if (b != 0) q0 = t * GW + ctz64(b);
                    nbad += popc64(b);
                }
                if (nbad == 1) {                             // the path, with the guide's own node at q0 (every lane has read its guide nodes: the ballots were rendezvous)
                    const int x0 = (int)posnode[q0 + d];
                    g.sync();
                    HYPO_UNROLL
                    for (int t = 0; t < XSL; ++t) { const int q = t * GW + g.lane; if (q < Lu) posnode[q] = (int16_t)(q == q0 ? x0 : v[t]); }
                    g.sync();
                    return guided_one_sub(q0, mode);
                }
            }
            DBGR(strong ? 4 : 5); return strong ? -1 : 0;
        }
        int carry = -1;
        HYPO_UNROLL
        for (int t = 0; t < XSL; ++t) {
            const int q = t * GW + g.lane;
            const int prev = g.shfl_up1(v[t], carry);        // (v[t] is still the plain node here)
            if (t + 1 < XSL) carry = g.shfl(v[t], GW - 1);
            if (q < Lu) {
                const int u = v[t];
                const int k = (int)nin[u];
                if (q == 0) { if (mode != MODE_ROV && k != 0) bad = true; }
                else {
                    const int cp = (int)seq[q - 1];
                    int first_same = -1, pi = -1;                      // first in-edge (the reference tries them in this order) whose source carries the letter before
                    for (int p = 0; p < k; ++p) {
                        const int src = (int)inp[u * KIN + p];
                        if (src == prev) pi = p;
                        if (k > 1 && first_same < 0 && (int)code[src] == cp) first_same = p;
                    }
                    if (pi < 0) bad = true;
                    if (k > 1 && first_same != pi) amb = true;
                    if (pi >= 0) v[t] |= pi << 16;
                }
                if (q == Lu - 1 && mode != MODE_LOV && n_out(u) != 0) bad = true;
            }
        }
        int ns = 1;                                          // sinks that carry the last letter (kNW / kROV)
        if (mode != MODE_LOV) {
            ns = 0;
            const int cl = (int)seq[Lu - 1];
            for (int u0 = 0; u0 < n_nodes; u0 += GW) { const int u = u0 + g.lane; ns += popc64(g.ballot(u < n_nodes && n_out(u) == 0 && (int)code[u] == cl)); }
        }
        if (g.any(bad)) { DBGR(strong ? 6 : 7); return strong ? -1 : 0; }
        if (g.any(amb) || ns != 1) { DBGR(g.any(amb) ? 8 : 9); return 0; }
        if (mode == MODE_LOV) {
            // kLOV's end row is the FIRST row in rank order whose cell in column L is perfect: no node that carries the last
            // letter and ranks before the path's last node may be the end of a perfect path.  Followed backwards — the in-edge
            // sources that carry the letter before, theirs, .. — such candidates die out within a few columns (one in four
            // survives a step); one that reaches a source in column 1, or is still alive eight columns back, is left to
            // Poa::thread_cols.  flag[]: a byte per node in the score ring (idle here).
            int vl = v[0];
            HYPO_UNROLL
            for (int t = 1; t < XSL; ++t) if ((Lu - 1) / GW == t) vl = v[t];
            vl = g.shfl(vl, (Lu - 1) & (GW - 1)) & 0xffff;
            uint8_t* const flag = (uint8_t*)ring;
            static_assert((int)sizeof(score_t) * Cfg::RINGCELLS >= NMAX, "a byte per node fits the ring");
            const int rv = Cfg::LAZY ? 0 : (int)n2r[vl], cl = (int)seq[Lu - 1];
            bool in[XRPL];
            HYPO_UNROLL
            // (every OTHER node that carries the letter, whatever its rank: under the lazy rank order n2r[] is not the reference's)
            for (int q = 0; q < XRPL; ++q) { const int u = q * GW + g.lane; in[q] = u < n_nodes && (int)code[u] == cl && (Cfg::LAZY ? u != vl : (int)n2r[u] < rv); }
            for (int j = Lu; ; --j) {                        // in[]: nodes whose cell in column j may be perfect
                bool some = false, src = false;
                HYPO_UNROLL
                for (int q = 0; q < XRPL; ++q) { some = some || in[q]; src = src || (in[q] && nin[q * GW + g.lane] == 0); }
                if (!g.any(some)) break;
                if (j == 1) { if (g.any(src)) { DBGR(10); return 0; } break; }      // (column 1 is perfect for a node without in-edges only)
                if (Lu - j >= 8) { DBGR(11); return 0; }
                const int cb = (int)seq[j - 2];
                HYPO_UNROLL
                for (int q = 0; q < XRPL; ++q) { const int u = q * GW + g.lane; if (u < n_nodes) flag[u] = 0; }
                g.sync();
                HYPO_UNROLL
                for (int q = 0; q < XRPL; ++q) {
                    const int u = q * GW + g.lane;
                    if (in[q]) { const int k = (int)nin[u]; for (int p = 0; p < k; ++p) { const int sx = (int)inp[u * KIN + p]; if ((int)code[sx] == cb) flag[sx] = 1; } }
                }
                g.sync();
                HYPO_UNROLL
                for (int q = 0; q < XRPL; ++q) { const int u = q * GW + g.lane; in[q] = u < n_nodes && flag[u] != 0; }
                g.sync();
            }
        }
        g.sync();
        HYPO_UNROLL
        for (int t = 0; t < XSL; ++t) {
            const int q = t * GW + g.lane;
            if (q < Lu) {
                const int u = v[t] & 0xffff, pi = v[t] >> 16;
                posnode[q] = (int16_t)u;
                if (q >= 1) inw[u * KIN + pi] = (wt_t)(inw[u * KIN + pi] + 2);      // (graph.cpp:104-109; a path visits a node once)
            }
        }
        tb_steps = Lu; tb_fv = 0;
        g.sync();
        return 1;
    }

    HD int thread_cols(int mode) {
        const bool rov = mode == MODE_ROV, lov = mode == MODE_LOV;
        const int Lu = g.uniform(L);                         // (a member the compiler cannot prove uniform: the column loop must not run under an exec mask)
        uint32_t mt[XRPL]; xpw_t pw[XRPL];                  // a lane's ranks: row metadata; in-edge sources as matrix rows, a byte each, in in-edge order
        HYPO_UNROLL
        for (int q = 0; q < XRPL; ++q) { const int r = q * GW + g.lane; mt[q] = r < n_nodes ? rowmeta[r] : 0u; }
        XSet CH, NCP, F;
        int kmax = 0;
        // the sequence: 64-lane groups keep it in a register (lane t: letters t, t + 64, ..) and read a column's letter with
        // v_readlane; narrower groups read it from LDS one column ahead
        uint32_t sreg = 0;
        if constexpr (GW == 64) {
            HYPO_UNROLL
            for (int t = 0; t < XSL; ++t) { const int c = t * GW + g.lane; if (c < Lu) sreg |= (uint32_t)seq[c] << (8 * t); }
        }
        static_assert(GW != 64 || XSL <= 4, "a lane's letters fit a register");
        auto letter = [&](int j) -> int {                    // seq[j] (64-lane groups)
            return (int)(((uint32_t)g.shfl((int)sreg, j & 63) >> (8 * (j >> 6))) & 0xffu);
        };
        {
            bool chain[XRPL], src1[XRPL], out1[XRPL], valid[XRPL];
            const int s0 = GW == 64 ? letter(0) : (int)seq[0];
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) {
                const int r = q * GW + g.lane;
                valid[q] = r < n_nodes;
                const int k = meta_k(mt[q]), p0 = meta_p0(mt[q]);
                chain[q] = valid[q] && k == 1 && p0 == r;
                out1[q] = valid[q] && (mt[q] & META_OUT1) != 0;
                // column 1: perfect(p, 0) holds for the virtual row 0 only (kROV: for every row)
                src1[q] = valid[q] && meta_code(mt[q]) == s0 && (rov || k == 0);
                pw[q] = (xpw_t)(uint32_t)p0;
                if (!chain[q] && k > kmax) kmax = k;
            }
            CH = x_ballot(chain);
            F = x_ballot(src1);
            // NCP: ranks with an out-edge other than "the chain link to the next rank" (only they can feed a rank that is no chain link)
            const XSet O1 = x_ballot(out1), AV = x_ballot(valid);
            HYPO_UNROLL
            for (int t = 0; t < XW; ++t) {
                const uint64_t nextch = (CH.w[t] >> 1) | (t + 1 < XW ? CH.w[t + 1 < XW ? t + 1 : t] << 63 : 0ull);
                NCP.w[t] = AV.w[t] & ~(O1.w[t] & nextch);
            }
        }
        if (!x_any(F)) return 0;
        kmax = g.reduce_max(kmax);
        HYPO_NOUNROLL
        for (int p = 1; p < kmax; ++p) {                     // further in-edges of the ranks that are no chain links (three dependent reads each)
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) {
                const int r = q * GW + g.lane;
                if (r < n_nodes && p < meta_k(mt[q])) pw[q] |= (xpw_t)((xpw_t)(uint32_t)pred_row(r, p) << (8 * p));
            }
        }
        static_assert(KIN <= 8 && NMAX <= 255, "in-edge sources of a rank fit eight bytes");
        XSet st[XSL];                                        // lane c - 1 (slot (c - 1) / GW) keeps F_c
        HYPO_UNROLL
        for (int t = 0; t < XSL; ++t) { HYPO_UNROLL for (int u = 0; u < XW; ++u) st[t].w[u] = 0; }
        if (g.lane == 0) st[0] = F;
        int s_next = (GW == 64 || Lu < 2) ? 0 : (int)seq[1];
        for (int j = 2; j <= Lu; ++j) {
            int s;
            if constexpr (GW == 64) s = letter(j - 1);
            else { s = s_next; if (j < Lu) s_next = (int)seq[j]; }
            bool mq[XRPL];
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) mq[q] = q * GW + g.lane < n_nodes && meta_code(mt[q]) == s;
            const XSet M = x_ballot(mq);
            XSet C;
            uint64_t trig = 0;
            HYPO_UNROLL
            for (int t = 0; t < XW; ++t) {
                C.w[t] = ((F.w[t] << 1) | (t ? F.w[t ? t - 1 : 0] >> 63 : 0ull)) & CH.w[t];
                trig |= F.w[t] & NCP.w[t];
            }
            if (trig != 0) {
                HYPO_NO_IFCVT();
                bool hit[XRPL];
                HYPO_UNROLL
                for (int q = 0; q < XRPL; ++q) hit[q] = false;
                HYPO_NOUNROLL
                for (int p = 0; p < kmax; ++p) {
                    HYPO_UNROLL
                    for (int q = 0; q < XRPL; ++q) {
                        const bool ch = x_in(CH, q * GW + g.lane);
                        if (!ch && p < meta_k(mt[q])) hit[q] = hit[q] || x_in(F, (int)((pw[q] >> (8 * p)) & 0xffu) - 1);
                    }
                }
                const XSet H = x_ballot(hit);
                HYPO_UNROLL
                for (int t = 0; t < XW; ++t) C.w[t] |= H.w[t];
            }
            HYPO_UNROLL
            for (int t = 0; t < XW; ++t) F.w[t] = C.w[t] & M.w[t];
            if (!x_any(F)) return 0;                         // no path of the graph spells seq[0..j)
            if (g.lane == ((j - 1) & (GW - 1))) {            // lane j - 1 keeps column j
                HYPO_UNROLL
                for (int t = 0; t < XSL; ++t) if ((j - 1) / GW == t) st[t] = F;
            }
        }
        // end row: first perfect candidate in rank order (kNW / kROV: among the sinks)
        if (!lov) {
            bool sk[XRPL];
            HYPO_UNROLL
            for (int q = 0; q < XRPL; ++q) sk[q] = q * GW + g.lane < n_nodes && meta_sink(mt[q]);
            const XSet SNK = x_ballot(sk);
            HYPO_UNROLL
            for (int t = 0; t < XW; ++t) F.w[t] &= SNK.w[t];
            if (!x_any(F)) return 0;
        }
        int e = x_first(F);
        if constexpr (Cfg::LAZY) {
            // several perfect end rows under the lazy rank order: the first in the REFERENCE's order (a literal sort into the ring)
            int cnt = 0;
            HYPO_UNROLL
            for (int t = 0; t < XW; ++t) cnt += popc64(F.w[t]);
            if (lazy_on && cnt > 1) {
                HYPO_NO_IFCVT();
                if (literal_order() != RES_OK) return 0;
                int key = 0x7fffffff;
                HYPO_UNROLL
                for (int q = 0; q < XRPL; ++q) {
                    const int r = q * GW + g.lane;
                    if (r < n_nodes && x_in(F, r)) { const int k2 = ((int)n2r_alt[r2n[r]] << 8) | r; key = k2 < key ? k2 : key; }
                }
                e = (-g.reduce_max(-key)) & 255;
            }
        }
        // the rank of every column: the end row, a column's only perfect cell, or (-1) to be settled from the right
        int rk[XSL];
        HYPO_UNROLL
        for (int t = 0; t < XSL; ++t) {
            const int c = 1 + t * GW + g.lane;
            int cnt = 0;
            HYPO_UNROLL
            for (int u = 0; u < XW; ++u) cnt += popc64(st[t].w[u]);
            rk[t] = c == Lu ? e : (cnt == 1 ? x_first(st[t]) : -1);
        }
        HYPO_UNROLL
        for (int t = XSL - 1; t >= 0; --t) {
            uint64_t amb = g.ballot(1 + t * GW + g.lane < Lu && rk[t] < 0);
            while (amb) {
                HYPO_NO_IFCVT();
                const int ln = 63 - clz64(amb);
                amb &= ~(1ull << ln);
                const int c = 1 + t * GW + ln;               // the column to settle; column c + 1 is settled (lane c % GW, slot c / GW)
                int rnv = rk[0];
                HYPO_UNROLL
                for (int u = 1; u < XSL; ++u) if (c / GW == u) rnv = rk[u];
                const int rn = g.shfl(rnv, c & (GW - 1));
                if (rn < 0) return 0;                         // (cannot happen)
                // in-edges of that cell's rank, from the lane that owns it
                uint32_t mv = mt[0]; xpw_t pv = pw[0];
                HYPO_UNROLL
                for (int u = 1; u < XRPL; ++u) if (rn / GW == u) { mv = mt[u]; pv = pw[u]; }
                const int own = rn & (GW - 1);
                const uint32_t mtn = (uint32_t)g.shfl((int)mv, own);
                uint64_t pwn = (uint64_t)(uint32_t)g.shfl((int)(uint32_t)pv, own);
                if constexpr (sizeof(xpw_t) == 8) pwn |= (uint64_t)(uint32_t)g.shfl((int)(uint32_t)((uint64_t)pv >> 32), own) << 32;
                const XSet Fc = x_shfl(st[t], ln);
                const int k = meta_k(mtn);
                int found = -1;
                HYPO_NOUNROLL
                for (int p = 0; p < k; ++p) {
                    const int x = (int)((pwn >> (8 * p)) & 0xffu) - 1;
                    if (found < 0 && x >= 0 && x_in(Fc, x)) found = x;
                }
                if (found < 0) return 0;                      // (cannot happen: the cell to the right is perfect through one of them)
                if (g.lane == ln) rk[t] = found;
            }
        }
        HYPO_UNROLL
        for (int t = 0; t < XSL; ++t) {
            const int c = 1 + t * GW + g.lane;
            if (c <= Lu) posnode[c - 1] = (int16_t)r2n[rk[t]];
        }
        tb_steps = Lu; tb_fv = 0;
        g.sync();
        return 1;
    }

    // ---- the hybrid class's traceback through a tile in LDS -----------------------------------------------------------------------
    // Every step of the plain traceback is a chain of dependent HBM reads in this class (code -> node of the row -> row metadata
    // -> pred row: 60 alignments x ~200 steps x ~2 000 cycles were a fifth of a LONG window).  The walk moves up and to the left
    // a step at a time, so a block of direction codes around the cell in hand, with the node and the metadata of its rows, is
    // fetched in one go (two rows per lane, 16-byte loads) into the LDS the recent score rows occupy
    // during the row loop (idle here), and the steps read LDS until the walk leaves the block.  Same moves, same result.
#define HYPO_TB_TILE 1
    HD int traceback_tiled(int mode, int best_i, int S) {
        // 128 rows x 48 columns: a LONG window's graph has ~2.6 rows per column (1 300 nodes for 500 bases), so the walk leaves a
        // block of that shape through its top and its left edge at about the same time (~5 blocks per alignment)
        constexpr int TR = 128, TC = 48;
        static_assert(GW == 64 && TR == 2 * GW, "two tile rows per lane");
        uint8_t* const tdir = (uint8_t*)ring1;                       // [TR][TC] codes
        uint16_t* const tr2n = (uint16_t*)(tdir + TR * TC);          // [TR] node of the row
        uint32_t* const tmeta = (uint32_t*)(tr2n + TR);              // [TR] row metadata (with the row of pred 0)
        uint16_t* const tp1 = (uint16_t*)(tmeta + TR);               // [TR] row of pred 1 (a merge node has two in-edges as a rule; preds 2.. are read in place)
        static_assert(!(Cfg::HYBRID && Cfg::PATHCAP > 0) || TR * TC + TR * 2 + TR * 4 + TR * 2 <= Cfg::RING1 * Lay::SMAX * (int)sizeof(score_t), "the tile fits the LDS row ring");
        int i = best_i > 0 ? best_i : 0, j = best_i > 0 ? L : 0;
        int steps = 0, guard = 0;
        int ti0 = -1, tj0 = 0;                                       // the tile holds matrix rows ti0 - 127 .. ti0, columns tj0 .. tj0 + 47
        while (mode == MODE_ROV ? (i != 0 && j != 0) : (i != 0 || j != 0)) {
            if (++guard > n_nodes + L + 4) return RES_UNDEFINED;      // cannot loop; protects the GPU from a hang
            if (i == 0) {                                   // only row 0 left: horizontal moves (insertions)
                for (int t = g.lane; t < j; t += GW) posnode[t] = -1;
                steps += j; j = 0;
                break;
            }
            if (i > ti0 || i < ti0 - (TR - 17) || j < tj0 || j >= tj0 + TC || (tj0 > 0 && j - tj0 < 8)) {
                g.sync();
                ti0 = i; tj0 = j > 31 ? ((j - 31) & ~15) : 0;
                HYPO_UNROLL
                for (int h = 0; h < 2; ++h) {
                    const int t = g.lane + h * GW, row = ti0 - t;
                    if (row >= 1) {
                        const uint8_t* src = dir + (size_t)(row - 1) * (size_t)S + tj0;
                        uint4v* dst = (uint4v*)(tdir + t * TC);
                        HYPO_UNROLL
                        for (int c = 0; c < TC / 16; ++c) if (tj0 + 16 * c < S) dst[c] = *(const uint4v*)(src + 16 * c);
                        tr2n[t] = (uint16_t)r2n[row - 1]; tmeta[t] = rowmeta[row - 1];
                        if (KIN >= 2) tp1[t] = (uint16_t)predrows[(row - 1) * KIN + 1];      // (garbage for rows with fewer than two in-edges: never used)
                    }
                }
                g.sync();
            }
            // run of FAST cells along the diagonal: lane t looks at (i-t, j-t), as far as the tile reaches
            const int ii = i - g.lane, jj = j - g.lane;
            const int tr = ti0 - ii;
            const bool inside = ii >= 1 && jj >= 1 && tr < TR && jj >= tj0;
            int dv = -1, nodev = 0;
            if (inside) { dv = tdir[tr * TC + (jj - tj0)]; nodev = (int)tr2n[tr]; }
            else if (g.lane == 0) dv = tdir[(ti0 - i) * TC + (j - tj0)];       // (i >= 1 here; j == 0: the code of column 0)
            const bool fast = inside && dv == DIR_FAST;
            const uint64_t stop = g.ballot(!fast);
            const int run = stop ? ctz64(stop) : GW;
            if (run >= 6) {
                if (g.lane < run) posnode[jj - 1] = (int16_t)nodev;
                i -= run; j -= run; steps += run;
                continue;
            }
            // No run worth a wave's iteration: a noisy read moves one cell at a time (its rows' first predecessor is rarely the row before: bubbles interleave
            // in rank order), and a full-wave iteration per move costs ~500 cycles.  Lane 0 walks on alone through the tile — code,
            // metadata and node of a cell are three LDS reads — until it leaves the tile, needs a predecessor beyond the second (its
            // row is read from HBM by everybody below) or has made 96 moves (the wave then looks for a long run of FAST cells again).
            int wi = i, wj = j, ws = 0, wstop = 0;                 // wstop: 1 = j ran out (undefined), 2 = needs pred p >= 1
            if (g.lane == 0) {
                for (int it = 0; it < 96; ++it) {
                    if (wi < 1 || wi > ti0 || wi <= ti0 - TR || wj < tj0 || wj >= tj0 + TC) break;
                    if (mode == MODE_ROV ? (wi == 0 || wj == 0) : (wi == 0 && wj == 0)) break;
                    // everything a move can need of its row is asked for together with the cell's code: one LDS round trip per
                    // move instead of a chain of two or three (code -> metadata -> second predecessor)
                    const int trow = ti0 - wi;
                    const int d = tdir[trow * TC + (wj - tj0)];
                    int nd = (int)tr2n[trow], p1r = KIN >= 2 ? (int)tp1[trow] : 0;
                    uint32_t mt = tmeta[trow];
                    HYPO_ARRIVED(d); HYPO_ARRIVED(nd); HYPO_ARRIVED(p1r); HYPO_ARRIVED(mt);
                    if (d == DIR_FAST) {                          // diagonal to the row before
                        if (wj == 0) { wstop = 1; break; }
                        posnode[wj - 1] = (int16_t)nd; --wj; --wi; ++ws;
                        continue;
                    }
                    if (d == DIR_HORIZ) {
                        if (wj == 0) { wstop = 1; break; }
                        posnode[wj - 1] = -1; --wj; ++ws;
                        continue;
                    }
                    const int p = dir_pred(d);
                    if (meta_k(mt) && p > 1) { wstop = 2; break; }
                    const int pi = meta_k(mt) ? (p == 0 ? meta_p0(mt) : p1r) : 0;
                    if (!is_vert(d)) {
                        if (wj == 0) { wstop = 1; break; }
                        posnode[wj - 1] = (int16_t)nd; --wj;
                    }
                    wi = pi; ++ws;
                }
            }
            wi = g.shfl(wi, 0); wj = g.shfl(wj, 0); ws = g.shfl(ws, 0); wstop = g.shfl(wstop, 0);
            g.sync();
            if (wstop == 1) return RES_UNDEFINED;
            i = wi; j = wj; steps += ws;
            if (wstop == 2) {                                     // a move through pred p >= 2: its row from the table in HBM
                const int d = g.shfl((int)tdir[(ti0 - i) * TC + (j - tj0)], 0);
                const int p = dir_pred(d);
                const int pi = pred_row(i - 1, p);
                if (!is_vert(d)) {
                    if (j == 0) return RES_UNDEFINED;
                    if (g.lane == 0) posnode[j - 1] = (int16_t)tr2n[ti0 - i];
                    --j;
                }
                i = pi;
                ++steps;
            } else if (ws == 0) return RES_UNDEFINED;              // (cannot happen: the cell in hand is inside the tile and not a run)
            guard += ws;
        }
        tb_steps = g.uniform(steps); tb_fv = g.uniform(j);
        g.sync();
        return RES_OK;
    }

    // ---- the hybrid class's score rows on packed pairs of int16 columns ------------------------------------------------------
    // Same recurrence, tie rules and byte codes as the one-column-per-register loop in align() (which classes without int16 rows
    // keep), in the arithmetic of rows_pk: two columns per register, every select a multiply-add on t = min_u16(a - b, 1).  The
    // hybrid class's own furniture stays: the dense ring (row i in slot i mod R of the HBM ring when a far row reads it, the
    // RING1 most recent rows in LDS), row metadata and the second predecessor's row refilled 64 rows at a time, 8-bit codes
    // (any in-degree), the list of rows tied for the end row that the lazy rank order needs (*ntie_out, newslot[]).
    // Returns the end row as lane `L / CPL` sees it (the caller broadcasts), like the loop it replaces.
    template <int NP>
    struct alignas(pow2_of(NP * 4)) PackPX { P2 v[NP]; };
    template <int NP>
    struct alignas(pow2_of(NP * 2)) DPackX { uint8_t v[2 * NP]; };
    template <int NP>
    HD int rows_pk_hyb_w(int mode, int m, int n, int gp, int S, int R, int* ntie_out) {
        constexpr int CPL = 2 * NP;                          // columns per lane of THIS alignment (shadows the class's)
        typedef PackPX<NP> PackP;
        typedef DPackX<NP> DPack;
        constexpr int TIECAP = 48;
        const int j0 = CPL * g.lane;
        int amax = m < 0 ? -m : m; { const int b = n < 0 ? -n : n, c2 = gp < 0 ? -gp : gp; amax = amax > b ? amax : b; amax = amax > c2 ? amax : c2; }
        const int NEG16 = -32768 + amax;
        P2 SQ[NP], JG[NP], LAST[NP];
        HYPO_UNROLL
        for (int q = 0; q < NP; ++q) {
            const int j = j0 + 2 * q;
            const int s0 = (j >= 1 && j <= L) ? (int)seq[j - 1] : (int)C_NONE;
            const int s1 = (j + 1 <= L) ? (int)seq[j] : (int)C_NONE;
            SQ[q] = pk_make(s0, s1);
            JG[q] = pk_make(j * gp, (j + 1) * gp);           // row 0: H[0][j] = j*g
            LAST[q] = JG[q];
        }
        int vM = pk_bits(pk_splat(m)), vMN = pk_bits(pk_splat(n - m)), vGP = pk_bits(pk_splat(gp)), vONE = pk_bits(pk_splat(1));
        HYPO_IN_VGPR(vM); HYPO_IN_VGPR(vMN); HYPO_IN_VGPR(vGP); HYPO_IN_VGPR(vONE);
        const P2 M = pk_from_bits(vM), MN = pk_from_bits(vMN), GP = pk_from_bits(vGP), ONE = pk_from_bits(vONE);
        const int negfill = pk_bits(pk_splat(NEG16));
        const int keep0 = (g.lane == 0 && mode == MODE_ROV) ? (int)0xffff0000u : -1;      // kROV: first column is 0
        const int le = L / CPL, ce = L % CPL;
        const int ce_shift = 16 * (ce & 1);
        int best = NEG, best_i = -1, ntie = 0;
        const bool native_lov = mode == MODE_LOV && (P->flags & POA_NATIVE_KLOV) != 0;
        auto end_value = [&](const P2 (&v)[NP]) -> int {
            if (native_lov) {
                int mx = NEG;
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {
                    const int c0 = j0 + 2 * q;
                    if (c0 >= 1 && c0 <= L) { const int x = pk_lo(v[q]); mx = x > mx ? x : mx; }
                    if (c0 + 1 >= 1 && c0 + 1 <= L) { const int x = pk_hi(v[q]); mx = x > mx ? x : mx; }
                }
                return g.reduce_max(mx);
            }
            // (every candidate passes through a register of its own first: a chain of selects over v[q] was turned into ONE load from v[ce / 2] —
            // the row's scores stored to private memory in every row, read back here behind an s_waitcnt vmcnt(0) that also waited for the
            // row's direction codes to reach HBM; rounds 2-6, found in round 6's disassembly)
            int w = pk_bits(v[0]);
            HYPO_IN_VGPR(w);
            HYPO_UNROLL
            for (int q = 1; q < NP; ++q) { int b = pk_bits(v[q]); HYPO_IN_VGPR(b); if (ce / 2 == q) w = b; }
            return (int)(int16_t)(uint16_t)((uint32_t)w >> ce_shift);     // column L: meaningful in lane `le` only
        };
        int slot = 0, slotS = 0, rowS = 0;
        const int RS = R * S;
        constexpr int R1 = Cfg::RING1;
        int slot1S = 0;
        const int R1S = R1 * S;
        int nbreg = negfill, exreg = (int)0x80000000;
        uint32_t mchunk = 0u;
        int p1chunk = 0;
        auto load_pk = [&](const score_t* base, int off, P2 (&out)[NP]) {
            if (j0 < S) {
                const PackP pk = *(const PackP*)(base + off + j0);
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) out[q] = pk.v[q];
            } else {
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) out[q] = pk_from_bits(negfill);
            }
        };
        auto load_row = [&](int i, int pr, P2 (&out)[NP]) {          // matrix row pr as row i sees it
            DBGH(i - pr);
            if (R1 > 0 && i - pr <= R1) { int ps = slot1S - (i - pr) * S; ps = ps < 0 ? ps + R1S : ps; load_pk(ring1, ps, out); }
            else {
                // a row from the HBM ring (rare: a predecessor further back than the LDS ring reaches).  Its wait stays INSIDE this
                // branch: left to the point where the two paths join, the compiler waits there for "whatever memory operation may
                // be pending" — s_waitcnt vmcnt(0) on every row, i.e. for the direction-code stores of the row before to be
                // acknowledged by HBM (2 000 cycles a row of the LONG class in rounds 2-4, profiles/r05_long_waitcnt.txt)
                HYPO_NO_IFCVT();
                int ps = slotS - (i - pr) * S; ps = ps < 0 ? ps + RS : ps; load_pk(ring, ps, out);
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) { int b = pk_bits(out[q]); HYPO_ARRIVED(b); (void)b; }
            }
        };
        auto hscan = [&](P2 (&v)[NP]) {
            P2 x[NP];
            x[0] = pk_fold_hi(pk_sub(v[0], JG[0]));
            HYPO_UNROLL
            for (int q = 1; q < NP; ++q) x[q] = pk_fold_hi(pk_max(pk_sub(v[q], JG[q]), pk_hi_splat(x[q - 1])));
            exreg = g.scan_max_excl_c(pk_bits(x[NP - 1]), exreg);
            const P2 EX = pk_hi_splat(pk_from_bits(exreg));
            HYPO_UNROLL
            for (int q = 0; q < NP; ++q) v[q] = pk_add(pk_max(x[q], EX), JG[q]);
        };
        for (int r = 0; r < n_nodes; ++r) {
            const int i = r + 1;
            if ((r & 63) == 0) {
                mchunk = r + g.lane < n_nodes ? rowmeta[r + g.lane] : 0u;
                if (KIN >= 2) { p1chunk = r + g.lane < n_nodes ? (int)predrows[(r + g.lane) * KIN + 1] : 0; HYPO_ARRIVED(p1chunk); }
                HYPO_ARRIVED(mchunk);
            }
            const uint32_t meta = (uint32_t)g.shfl((int)mchunk, r & 63);
            const int cd = meta_code(meta), k = meta_k(meta);
            const bool sink = meta_sink(meta);
            const int p0 = meta_p0(meta);                    // 0 when k == 0 (virtual source row)
            const bool fastrow = p0 == i - 1;
            const int fastcode = fastrow ? (int)DIR_FAST : dir_diag(0);
            HYPO_DIAG(rows_slow += k > 1; guided_hits += (!fastrow && p0 != 0); one_sub_hits += k > 2; cols_hits += (R1 > 0 && p0 != 0 && i - p0 > R1); exact_tries += (mode == MODE_LOV || sink));
            P2 MV[NP];
            {
                const P2 CD = pk_splat(cd);
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) MV[q] = pk_mad(pk_minu(pk_xor(SQ[q], CD), ONE), MN, M);
            }
            P2 D[NP], U[NP], cD[NP], cU[NP];
            {
                P2 hp[NP];
                if (fastrow) { HYPO_UNROLL for (int q = 0; q < NP; ++q) hp[q] = LAST[q]; }
                else if (p0 == 0) { HYPO_UNROLL for (int q = 0; q < NP; ++q) hp[q] = JG[q]; }
                else load_row(i, p0, hp);
                const int nb = nbreg = g.shfl_up1(pk_bits(hp[NP - 1]), nbreg);
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {
                    D[q] = pk_add(pk_shift_in(q ? hp[q - 1] : pk_from_bits(nb), hp[q]), MV[q]);
                    U[q] = pk_add(hp[q], GP);
                }
            }
            if (k > 1) {                                     // several predecessors: remember which one reaches each maximum first
                P2 pD[NP], pU[NP];
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) { pD[q] = pk_splat(0); pU[q] = pk_splat(0); }
                for (int p = 1; p < k; ++p) {
                    P2 hp[NP];
                    // (the third and later predecessors' rows come from HBM here, behind an s_waitcnt vmcnt(0); keeping them per 64-row chunk in
                    // registers like the second's was measured in round 6 and gained nothing: 14-17 % of a LONG window's rows have three or more)
                    const int pr = (KIN >= 2 && p == 1) ? g.shfl(p1chunk, r & 63) : g.uniform(pred_row(r, p));
                    load_row(i, pr, hp);
                    const int nb = nbreg = g.shfl_up1(pk_bits(hp[NP - 1]), nbreg);
                    const P2 PP = pk_splat(p);
                    HYPO_UNROLL
                    for (int q = 0; q < NP; ++q) {
                        const P2 d = pk_add(pk_shift_in(q ? hp[q - 1] : pk_from_bits(nb), hp[q]), MV[q]);
                        const P2 u = pk_add(hp[q], GP);
                        const P2 nd = pk_max(D[q], d), nu = pk_max(U[q], u);
                        pD[q] = pk_mad(pk_minu(pk_sub(nd, D[q]), ONE), pk_sub(PP, pD[q]), pD[q]);      // strict: the first pred reaching the maximum wins
                        pU[q] = pk_mad(pk_minu(pk_sub(nu, U[q]), ONE), pk_sub(PP, pU[q]), pU[q]);
                        D[q] = nd; U[q] = nu;
                    }
                }
                const P2 FC = pk_splat(fastcode);
                HYPO_UNROLL
                for (int q = 0; q < NP; ++q) {               // byte codes: dir_diag(p) = 2 p, dir_vert(p) = 2 p + 1
                    cD[q] = pk_mad(pk_sub(ONE, pk_minu(pD[q], ONE)), FC, pk_add(pD[q], pD[q]));
                    cU[q] = pk_add(pk_add(pU[q], pU[q]), ONE);
                }
            }
            P2 v[NP];
            HYPO_UNROLL
            for (int q = 0; q < NP; ++q) v[q] = pk_max(D[q], U[q]);
            v[0] = pk_from_bits(pk_bits(v[0]) & keep0);
            hscan(v);
            if (j0 < S) {
                const P2 HZ = pk_splat(DIR_HORIZ);
                DPack dk;
                PackP pk;
                // code = cD, unless the vertical term wins (cU), unless the horizontal one does (DIR_HORIZ): cD + tD ((cU - cD) + tU (HZ - cU))
                if (k > 1) {
                    HYPO_UNROLL
                    for (int q = 0; q < NP; ++q) {
                        const P2 tD = pk_minu(pk_sub(v[q], D[q]), ONE), tU = pk_minu(pk_sub(v[q], U[q]), ONE);
                        const uint32_t b = (uint32_t)pk_bits(pk_mad(tD, pk_mad(tU, pk_sub(HZ, cU[q]), pk_sub(cU[q], cD[q])), cD[q]));
                        dk.v[2 * q] = (uint8_t)b; dk.v[2 * q + 1] = (uint8_t)(b >> 16);
                        pk.v[q] = v[q];
                    }
                } else {                                     // one predecessor (most rows): the two codes are the same in every column
                    HYPO_NO_IFCVT();
                    int vCD = pk_bits(pk_splat(fastcode)), vY = pk_bits(pk_splat(dir_vert(0) - fastcode));
                    HYPO_IN_VGPR(vCD); HYPO_IN_VGPR(vY);
                    const P2 CDU = pk_from_bits(vCD), Y = pk_from_bits(vY), X = pk_splat(DIR_HORIZ - dir_vert(0));
                    HYPO_UNROLL
                    for (int q = 0; q < NP; ++q) {
                        const P2 tD = pk_minu(pk_sub(v[q], D[q]), ONE), tU = pk_minu(pk_sub(v[q], U[q]), ONE);
                        const uint32_t b = (uint32_t)pk_bits(pk_mad(tD, pk_mad(tU, X, Y), CDU));
                        dk.v[2 * q] = (uint8_t)b; dk.v[2 * q + 1] = (uint8_t)(b >> 16);
                        pk.v[q] = v[q];
                    }
                }
                *(DPack*)(dir + rowS + j0) = dk;
                if (R1 > 0) {
                    *(PackP*)(ring1 + slot1S + j0) = pk;
                    if (meta & META_DEEP) *(PackP*)(ring + slotS + j0) = pk;      // read again from further back than the LDS ring reaches
                } else {
                    *(PackP*)(ring + slotS + j0) = pk;
                }
            }
            slot = slot + 1 == R ? 0 : slot + 1;
            slotS = slot == 0 ? 0 : slotS + S;
            if (R1 > 0) slot1S = slot1S + S == R1S ? 0 : slot1S + S;
            rowS += S;
            HYPO_UNROLL
            for (int q = 0; q < NP; ++q) LAST[q] = v[q];
            if (mode == MODE_LOV || sink) {                  // end cell: first strictly greater in rank order (sisd..cpp:279-288,332-339)
                HYPO_NO_IFCVT();
                const int val = end_value(v);
                if (native_lov) { if (val > best) { best = val; best_i = i; } }
                else if (g.lane == le) {
                    if constexpr (Cfg::LAZY) {               // lazy rank order: rows that tie for the end row are remembered
                        if (lazy_on) {
                            if (val > best) { ntie = 1; newslot[0] = (int16_t)i; }
                            else if (val == best) { if (ntie < TIECAP) newslot[ntie] = (int16_t)i; ++ntie; }
                        }
                    }
                    if (val > best) { best = val; best_i = i; }
                }
            }
            g.sync();
        }
        *ntie_out = g.shfl(ntie, le);                        // (group-uniform, like the end row: the caller does not know this alignment's split)
        return g.shfl(best_i, le);
    }

    // Columns per lane follow the sequence in hand: the row loop is VALU bound (two waves per SIMD, both in it most of the time) and
    // its cost grows with the register pairs a lane carries whether the columns exist or not.  A LONG window's arms are 120-550
    // bases, the class is built for 639: rows_pk_hyb_w<2..5> = 4 / 6 / 8 / 10 columns per lane for up to 256 / 384 / 512 / 640
    // columns (Poa::hyb_pairs; the row stride is a multiple of the lane's columns, Poa::align).  Scores, codes and their places
    // in memory do not depend on the split.
    static constexpr int HYB_NP_MAX = Cfg::CPL / 2;
#define HYPO_HYB_ADAPT 1
    HD static constexpr int hyb_pairs(int W) {             // register pairs per lane for rows of W columns
        return !HYPO_HYB_ADAPT ? HYB_NP_MAX : (W <= 2 * GW * 2 && HYB_NP_MAX >= 2 ? 2 : (W <= 3 * GW * 2 && HYB_NP_MAX >= 3 ? 3 : (W <= 4 * GW * 2 && HYB_NP_MAX >= 4 ? 4 : HYB_NP_MAX)));
    }
    HD int rows_pk_hyb(int mode, int m, int n, int gp, int S, int R, int* ntie_out) {
        if constexpr (HYB_NP_MAX > 4) {
            const int np = g.uniform(hyb_pairs(L + 1));
            if (np == 2) return rows_pk_hyb_w<2>(mode, m, n, gp, S, R, ntie_out);
            if (np == 3) return rows_pk_hyb_w<3>(mode, m, n, gp, S, R, ntie_out);
            if (np == 4) return rows_pk_hyb_w<4>(mode, m, n, gp, S, R, ntie_out);
        }
        return rows_pk_hyb_w<HYB_NP_MAX>(mode, m, n, gp, S, R, ntie_out);
    }

    // [tb_fv, L) and tb_steps (number of traceback steps; 0 = "empty alignment").
    HD int align(int mode, int m, int n, int gp) {
        tb_steps = 0; tb_fv = L; threaded = false; weights_done = false;
        if (n_nodes == 0 || L == 0) return RES_OK;
        n_nodes = g.uniform(n_nodes);
        mode = g.uniform(mode);                             // group-uniform by construction (one sequence per group at a time)
        const int W = g.uniform(L) + 1;
        // row stride (even when NIB); the hybrid class rounds to a multiple of 16 as well, so that the traceback's tile loads of
        // direction codes (traceback_tiled) are 16-byte aligned
        constexpr int SQ = (Cfg::HYBRID && Cfg::PATHCAP > 0) ? (CPL % 16 == 0 ? CPL : (CPL % 8 == 0 ? 2 * CPL : (CPL % 4 == 0 ? 4 * CPL : (CPL % 2 == 0 ? 8 * CPL : 16 * CPL)))) : CPL;
        static_assert(Lay::SMAX % SQ == 0 || !(Cfg::HYBRID && Cfg::PATHCAP > 0), "largest row stride is a multiple of the stride quantum");
        int sq = SQ;
        if constexpr (Cfg::PACKED_HYB && CPL / 2 > 4) {     // columns per lane follow the sequence (Poa::rows_pk_hyb): stride = a multiple of them and of 16
            const int np = hyb_pairs(W);
            sq = np == 3 ? 48 : (np == CPL / 2 ? SQ : 16);
        }
        const int S = (W + sq - 1) / sq * sq;
        // (the packed classes ask only when direction codes are about to be written: need_meta below)
        const bool dir_fits = n_nodes * S <= Cfg::DIRCELLS;
        if (!(PK && HYPO_DEFER_META) && !dir_fits) return RES_OVERFLOW;
        if (sizeof(score_t) < 4) {                          // int16 rows are exact only below this bound
            int a = m < 0 ? -m : m, b = n < 0 ? -n : n, c2 = gp < 0 ? -gp : gp;
            a = a > b ? a : b; a = a > c2 ? a : c2;
            if (a * (n_nodes + L + 1) >= 32767) return RES_OVERFLOW;
            if ((PK || Cfg::PACKED_HYB) && a * (n_nodes + 2 * L + CPL + 8) >= 32767) return RES_OVERFLOW;   // rows_pk: H - j*g and NEG16 + score in 16 bits
        }
        // The row metadata (by rank: rebuilt after every change of the graph, four passes of dependent LDS reads) are what the score
        // rows and Poa::thread_cols read; threading along the guide does not, so the packed classes build them only when one of
        // those two is reached — between two arms that thread along the guide, or sit one substitution off it, nobody asks.
        const int R = g.uniform(Cfg::RINGCELLS / S);        // ring rows; row i can still see rows i-R .. i-1
        auto need_meta = [&](bool rows) -> bool {           // false (score rows only): no room for the direction codes, or the ring is too shallow for this rank order (RES_OVERFLOW)
            if (rows && !dir_fits) return false;
            if (meta_dirty) { build_rowmeta(); HYPO_TICK(PH_META); }
            return !rows || !(R < (int)stat[ST_MAXD] + 1 || R < 1);
        };
        if constexpr (!PK || !HYPO_DEFER_META) { if (!need_meta(true)) return RES_OVERFLOW; }
        if (g.lane == 0) { stat[ST_CELLS] += (uint32_t)((n_nodes + 1) * W); stat[ST_ALIGNS] += 1; } HYPO_DIAG(rows_done += (uint32_t)n_nodes);
        auto overflow_late = [&]() -> int {                 // (the class that takes the window over makes this alignment again and counts it there)
            if (g.lane == 0) { stat[ST_CELLS] -= (uint32_t)((n_nodes + 1) * W); stat[ST_ALIGNS] -= 1; }
            return RES_OVERFLOW;
        };

        int best_i = -1;
        if constexpr (PK) {
            // a sequence that spells a path of the graph (most reads do) is threaded without scores (Poa::thread_cols: no rows, no
            // traceback; posnode[] is final); what spells none goes through the score rows.  An attempt costs the columns up to
            // the sequence's first error, so every alignment starts with one.
            // (not in the class of wide windows, Cfg::LMAX > 127, unless HYPO_EXACT_WIDE)
            constexpr bool EXACT_HERE = HYPO_EXACT && (Cfg::LMAX <= 127 || HYPO_EXACT_WIDE);
            if (EXACT_HERE && m > 0 && n < m && gp < 0) {
                // first along the path of the sequence before (Poa::thread_guided), then, where that cannot tell, column by column
                int hit = thread_guided(mode, gp < n && m - n < -2 * gp);
                weights_done = hit == 1;
                // (column by column not in the wide class and not in four-groups-per-wave class 0, where every rank set is a vector register per lane: both would lose a wave per SIMD to it and thread along the guide only)
                if constexpr (Cfg::LMAX <= 127 && GW >= 32) {
                    if (hit == 0) {
                        HYPO_TICK(PH_EXACT);
                        need_meta(false);
                        hit = thread_cols(mode);
                    }
                }
                if (hit < 0) hit = 0;
                HYPO_TICK(PH_EXACT);
                if (g.lane == 0) {
                    stat[ST_CEXACT] += (uint32_t)((n_nodes + 1) * W);
                    stat[ST_LASTX] = hit ? 1u : 0u;
                    if (hit) stat[ST_XHITS] += 1;
                }
                HYPO_DIAG(exact_tries += 1; guided_hits += weights_done ? 1u : 0u; one_sub_hits += hit == 2 ? 1u : 0u; cols_hits += (hit == 1 && !weights_done) ? 1u : 0u);
                if (hit == 2) return RES_OK;                   // aligned, one substitution off the guide: posnode[] goes to add_alignment
                if (hit) { threaded = true; return RES_OK; }
            } else stat_set(ST_LASTX, 0u);
            int ntie_pk = 0;
            if (!need_meta(true)) return overflow_late();
            { best_i = rows_pk(mode, m, n, gp, S, R, &ntie_pk); stat_add(ST_CSCORED, (uint32_t)((n_nodes + 1) * W)); HYPO_DIAG(rows_scored_n += (uint32_t)n_nodes); }
            if constexpr (Cfg::LAZY) {
                if (lazy_on && ntie_pk > 1) {
                    // several rows share the best end value: the reference takes the first of them in ITS rank order
                    // (the class that takes the window over sorts first, makes this alignment again and counts it there)
                    if (ntie_pk > PK_TIECAP) { stat_add(ST_CSCORED, 0u - (uint32_t)((n_nodes + 1) * W)); return overflow_late(); }
                    g.sync();
                    HYPO_DIAG(tie_sorts += 1);
                    const int rc = first_of_rows(posnode + 1, ntie_pk, &best_i);
                    if (rc == RES_OVERFLOW) { stat_add(ST_CSCORED, 0u - (uint32_t)((n_nodes + 1) * W)); return overflow_late(); }
                    if (rc != RES_OK) return rc;
                }
            }
        } else {
        stat_add(ST_CSCORED, (uint32_t)((n_nodes + 1) * W));
        int ntie = 0;                                        // lazy rank order: rows tied for the end row (in newslot[], dead until add_alignment)
        constexpr int TIECAP = 48;
        static_assert(TIECAP <= 64 && TIECAP <= GW, "one lane per tied row, its index in 6 bits");
        const int le = L / CPL;                              // owner of the last column
        if constexpr (Cfg::PACKED_HYB) {
            best_i = rows_pk_hyb(mode, m, n, gp, S, R, &ntie);
            HYPO_DIAG(rows_scored_n += (uint32_t)n_nodes);
        } else {
        HYPO_IN_VGPR(m); HYPO_IN_VGPR(n); HYPO_IN_VGPR(gp);
        const int j0 = CPL * g.lane;
        int sq[CPL];                                        // sq[c] = code of seq[j-1] for column j = j0+c
        HYPO_UNROLL
        for (int c = 0; c < CPL; ++c) { const int j = j0 + c; sq[c] = (j >= 1 && j <= L) ? (int)seq[j - 1] : (int)C_NONE; }

        int jg[CPL], last[CPL];                              // j*g per column; most recently computed row (registers)
        HYPO_UNROLL
        for (int c = 0; c < CPL; ++c) { jg[c] = (j0 + c) * gp; last[c] = jg[c]; }   // row 0: H[0][j] = j*g (sisd..cpp:197-199,230-232)

        const int ce = L % CPL;
        int best = NEG;

        int slot = 0;                                        // ring slot of row i (no integer division in the loop)
        int slotS = 0, rowS = 0;                             // slot * S and r * S, advanced by addition (group-uniform)
        const int RS = R * S;
        constexpr int R1 = Cfg::RING1;                       // hybrid classes: LDS ring of the R1 most recent rows (0 = none)
        int slot1S = 0;                                      // (i mod R1) * S
        const int R1S = R1 * S;
        // Row metadata: full-wave groups keep it in registers (lane r holds row r, fetched with v_readlane, no
        // LDS latency in the row loop); narrower groups and the big classes prefetch it from LDS two rows ahead.
        constexpr int MREG = (NMAX + GW - 1) / GW;
        constexpr bool META_IN_REGS = (GW == 64) && (MREG <= 4);
        uint32_t mreg[META_IN_REGS ? MREG : 1];
        if (META_IN_REGS) {
            HYPO_UNROLL
            for (int q = 0; q < (META_IN_REGS ? MREG : 1); ++q) {
                const int rr = q * GW + g.lane;
                mreg[q] = rr < n_nodes ? rowmeta[rr] : 0u;
            }
            HYPO_UNROLL
            for (int q = 0; q < (META_IN_REGS ? MREG : 1); ++q) HYPO_ARRIVED(mreg[q]);
        }
        // Full-wave groups with more rows than that refill one register every 64 rows (lane t = row base + t) and wait for the
        // load right there: a per-row prefetch from HBM makes the compiler wait for ALL outstanding memory operations (the
        // row's own stores included) once per row, which was most of the row time of the HBM-scratch classes.
        constexpr bool META_CHUNKED = (GW == 64) && !META_IN_REGS;
        uint32_t mchunk = 0u;
        // ... and, in the classes that tabulate pred rows in HBM scratch, the matrix row of pred 1 of the same 64 rows: nearly every
        // row with several predecessors has two, and looking the second one up inside the loop was an HBM load whose wait also
        // drained the rows' outstanding stores
#define HYPO_P1_CHUNK 1
        constexpr bool P1_CHUNKED = (HYPO_P1_CHUNK != 0) && META_CHUNKED && PRED_TABLE && KIN >= 2;
        int p1chunk = 0;
        uint32_t meta_a = (META_IN_REGS || META_CHUNKED) ? 0u : rowmeta[0];
        uint32_t meta_b = (!META_IN_REGS && !META_CHUNKED && n_nodes > 1) ? rowmeta[1] : 0u;
        for (int r = 0; r < n_nodes; ++r) {
            const int i = r + 1;
            uint32_t meta;
            if (META_CHUNKED) {
                if ((r & 63) == 0) {
                    mchunk = r + g.lane < n_nodes ? rowmeta[r + g.lane] : 0u;
                    if (P1_CHUNKED) { p1chunk = r + g.lane < n_nodes ? (int)predrows[(r + g.lane) * KIN + 1] : 0; HYPO_ARRIVED(p1chunk); }
                    HYPO_ARRIVED(mchunk);
                }
                meta = (uint32_t)g.shfl((int)mchunk, r & 63);
            } else if (META_IN_REGS) {
                uint32_t mv = mreg[0];
                HYPO_UNROLL
                for (int q = 1; q < (META_IN_REGS ? MREG : 1); ++q) if ((r / GW) == q) mv = mreg[q];
                meta = (uint32_t)g.shfl((int)mv, r % GW);
            } else {
                meta = meta_a;
                meta_a = meta_b;
                if (r + 2 < n_nodes) meta_b = rowmeta[r + 2];
            }
            const int cd = meta_code(meta), k = meta_k(meta);
            const bool sink = meta_sink(meta);
            const int p0 = meta_p0(meta);                    // 0 when k == 0 (virtual source row)
            int D[CPL], U[CPL];
            int codeD[CPL], codeU[CPL];                      // direction codes if the cell's value comes from D / U
            const bool fastrow = p0 == i - 1;
            const int fastcode = fastrow ? (int)DIR_FAST : dir_diag(0);
            {
                int hp[CPL];
                if (fastrow) { HYPO_UNROLL for (int c = 0; c < CPL; ++c) hp[c] = last[c]; }
                else if (p0 == 0) { HYPO_UNROLL for (int c = 0; c < CPL; ++c) hp[c] = jg[c]; }
                else if (R1 > 0 && i - p0 <= R1) { int ps = slot1S - (i - p0) * S; ps = ps < 0 ? ps + R1S : ps; load_ring_at(ring1, ps, S, hp); }
                else { int ps = slotS - (i - p0) * S; ps = ps < 0 ? ps + RS : ps; load_ring_at(ring, ps, S, hp); }
                const int left = g.shfl_up1(hp[CPL - 1], NEG);
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) {
                    const int dsrc = c ? hp[c - 1] : left;
                    D[c] = dsrc + (sq[c] == cd ? m : n);
                    U[c] = hp[c] + gp;
                }
            }
            if (k > 1) {                                     // several predecessors: remember which one reaches each maximum first
                int pD[CPL], pU[CPL];
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) { pD[c] = 0; pU[c] = 0; }
                for (int p = 1; p < k; ++p) {
                    int hp[CPL];
                    const int pr = (P1_CHUNKED && p == 1) ? g.shfl(p1chunk, r & 63) : g.uniform(pred_row(r, p));
                    if (R1 > 0 && i - pr <= R1) { int ps = slot1S - (i - pr) * S; ps = ps < 0 ? ps + R1S : ps; load_ring_at(ring1, ps, S, hp); }
                    else { int ps = slotS - (i - pr) * S; ps = ps < 0 ? ps + RS : ps; load_ring_at(ring, ps, S, hp); }
                    const int left = g.shfl_up1(hp[CPL - 1], NEG);
                    HYPO_UNROLL
                    for (int c = 0; c < CPL; ++c) {
                        const int dsrc = c ? hp[c - 1] : left;
                        const int d = dsrc + (sq[c] == cd ? m : n);
                        const int u = hp[c] + gp;
                        if (d > D[c]) { D[c] = d; pD[c] = p; }   // strict: the first pred reaching the maximum wins
                        if (u > U[c]) { U[c] = u; pU[c] = p; }
                    }
                }
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) { codeD[c] = pD[c] == 0 ? fastcode : dir_diag(pD[c]); codeU[c] = dir_vert(pU[c]); }
            } else {
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) { codeD[c] = fastcode; codeU[c] = dir_vert(0); }
            }
            if (g.lane == 0) D[0] = NEG;                     // column 0 has no diagonal
            int v[CPL];
            HYPO_UNROLL
            for (int c = 0; c < CPL; ++c) v[c] = D[c] > U[c] ? D[c] : U[c];
            if (g.lane == 0 && mode == MODE_ROV) v[0] = 0;   // first column (sisd..cpp:200-211,237-239)
            // horizontal term H[i][j] = max(H[i][j], H[i][j-1] + g): prefix max of H[i][j] - j*g
            {
                int run = v[0] - jg[0];
                v[0] = run;
                HYPO_UNROLL
                for (int c = 1; c < CPL; ++c) {
                    const int x = v[c] - jg[c];
                    run = x > run ? x : run;
                    v[c] = run;
                }
                const int ex = g.scan_max_excl(run, NEG);
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) {
                    const int x = v[c] > ex ? v[c] : ex;
                    v[c] = x + jg[c];
                }
            }
            if (j0 < S) {
                // the reference's traceback preference (sisd..cpp:370-428), resolved per cell with selects
                int dc[CPL];
                Pack pk;
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) {
                    const int vtc = v[c] == U[c] ? codeU[c] : (int)DIR_HORIZ;
                    dc[c] = v[c] == D[c] ? codeD[c] : vtc;
                    pk.v[c] = (score_t)v[c];
                }
                DPack dk;
                if (NIB) {
                    HYPO_UNROLL
                    for (int c = 0; c < CPL; c += 2) dk.v[c / 2] = (uint8_t)(dc[c] | (dc[c + 1] << 4));
                    *(DPack*)(dir + (rowS >> 1) + (j0 >> 1)) = dk;      // S and j0 are even here
                } else {
                    HYPO_UNROLL
                    for (int c = 0; c < CPL; ++c) dk.v[c] = (uint8_t)dc[c];
                    *(DPack*)(dir + rowS + j0) = dk;
                }
                if (R1 > 0) {
                    *(Pack*)(ring1 + slot1S + j0) = pk;
                    if (meta & META_DEEP) *(Pack*)(ring + slotS + j0) = pk;      // read again from further back than the LDS ring reaches
                } else {
                    *(Pack*)(ring + slotS + j0) = pk;
                }
            }
            slot = slot + 1 == R ? 0 : slot + 1;
            slotS = slot == 0 ? 0 : slotS + S;
            if (R1 > 0) slot1S = slot1S + S == R1S ? 0 : slot1S + S;
            rowS += S;
            HYPO_UNROLL
            for (int c = 0; c < CPL; ++c) last[c] = v[c];
            // end cell: first strictly greater in rank order (sisd..cpp:279-288,332-339)
            if (mode == MODE_LOV || sink) {                  // group-uniform: most rows of kNW / kROV skip it
                HYPO_NO_IFCVT();
                if (mode == MODE_LOV && (P->flags & POA_NATIVE_KLOV)) {      // native flavour: maximum over columns 1..L of the row
                    int mx = NEG;
                    HYPO_UNROLL
                    for (int c = 0; c < CPL; ++c) if (j0 + c >= 1 && j0 + c <= L) mx = v[c] > mx ? v[c] : mx;
                    const int val = g.reduce_max(mx);
                    if (val > best) { best = val; best_i = i; }
                } else if (g.lane == le) {
                    int val = v[0];
                    HYPO_UNROLL
                    for (int c = 1; c < CPL; ++c) if (c == ce) val = v[c];
                    if constexpr (Cfg::LAZY) {               // lazy rank order: rows that tie for the end row are remembered
                        if (lazy_on) {
                            if (val > best) { ntie = 1; newslot[0] = (int16_t)i; }
                            else if (val == best) { if (ntie < TIECAP) newslot[ntie] = (int16_t)i; ++ntie; }
                        }
                    }
                    if (val > best) { best = val; best_i = i; }
                }
            }
            g.sync();
        }
        best_i = g.shfl(best_i, le);
        }   // !PACKED_HYB
        if constexpr (Cfg::LAZY) {
            if (lazy_on) {
                ntie = g.shfl(ntie, le);
                if (ntie > 1) {
                    // several sinks share the best score: the reference takes the first of them in ITS rank order
                    if (ntie > TIECAP) return RES_OVERFLOW;      // (the next class sorts literally every time)
                    g.sync();
                    // The usual tie: the tied sinks are members of ONE aligned clique all of whose members are sinks (the last
                    // column of the window).  Nothing depends on a sink, so the reference's DFS reaches that clique from its
                    // member with the smallest id, as a root, and emits that member followed by its aligned list in list order
                    // (graph.cpp:311-349): the winner follows from ids and that list, without sorting.
                    {
                        const int u0 = (int)r2n[(int)newslot[0] - 1];
                        const int ka0 = (int)nal[u0];
                        int cmin = u0; bool all_sinks = n_out(u0) == 0;
                        for (int a = 0; a < ka0; ++a) { const int x = (int)al[u0 * AL + a]; cmin = x < cmin ? x : cmin; all_sinks = all_sinks && n_out(x) == 0; }
                        int pos = 0x7fff; bool member = true;
                        if (g.lane < ntie) {
                            const int u = (int)r2n[(int)newslot[g.lane] - 1];
                            if (u == cmin) pos = 0;
                            else {
                                member = false;
                                const int kc = (int)nal[cmin];
                                for (int a = 0; a < kc; ++a) if ((int)al[cmin * AL + a] == u) { pos = 1 + a; member = true; }
                            }
                        }
                        if (all_sinks && !g.any(!member)) {
                            const int key = g.lane < ntie ? ((pos << 6) | g.lane) : 0x7fffffff;
                            const int kmin = -g.reduce_max(-key);
                            best_i = (int)newslot[kmin & 63];
                            ntie = 0;
                            g.sync();
                        }
                    }
                }
                if (ntie > 1) {
                    id_t* const sr = r2n; id_t* const sn = n2r;
                    const bool td = topo_dirty;
                    r2n = r2n_alt; n2r = n2r_alt;
                    const int rc = toposort();
                    r2n = sr; n2r = sn; topo_dirty = td;
                    HYPO_DIAG(topo_runs += 1);
                    if (rc != RES_OK) return rc;
                    int key = 0x7fffffff;
                    if (g.lane < ntie) key = ((int)n2r_alt[r2n[(int)newslot[g.lane] - 1]] << 6) | g.lane;
                    const int kmin = -g.reduce_max(-key);
                    best_i = (int)newslot[kmin & 63];
                    g.sync();
                }
            }
        }
        }
        HYPO_TICK(PH_DP);

        // ---- traceback over direction codes ----
        if constexpr (Cfg::HYBRID && Cfg::PATHCAP > 0 && HYPO_TB_TILE) {
            const int rc = traceback_tiled(mode, best_i, S);
            HYPO_TICK(PH_TRACE);
            return rc;
        }
        int i = best_i > 0 ? best_i : 0, j = best_i > 0 ? L : 0;
        int steps = 0, guard = 0;
        while (mode == MODE_ROV ? (i != 0 && j != 0) : (i != 0 || j != 0)) {
            if (++guard > n_nodes + L + 4) return RES_UNDEFINED;      // cannot loop; protects the GPU from a hang
            if (i == 0) {                                   // only row 0 left: horizontal moves (insertions)
                for (int t = g.lane; t < j; t += GW) posnode[t] = -1;
                steps += j; j = 0;
                break;
            }
            // run of FAST cells along the diagonal: lane t looks at (i-t, j-t)
            const int ii = i - g.lane, jj = j - g.lane;
            // HBM-scratch classes: every step of this loop is a chain of dependent HBM reads (code -> node of the row -> row
            // metadata -> pred row), so what the step may need is fetched together with the codes: one round trip for a run
            // instead of two, two for a single move instead of three
            constexpr bool AHEAD = Cfg::PATHCAP > 0;
            int dv = -1, nodev = 0; uint32_t meta0 = 0;
            if (AHEAD) {
                if (ii >= 1 && jj >= 1) { dv = read_dir((ii - 1) * S + jj); nodev = (int)r2n[ii - 1]; }
                else if (g.lane == 0) dv = read_dir((i - 1) * S + j);          // (i >= 1 here; j == 0: the code of column 0)
                meta0 = rowmeta[i - 1];
            }
            const bool fast = AHEAD ? ((ii >= 1 && jj >= 1) && dv == DIR_FAST) : ((ii >= 1 && jj >= 1) && read_dir((ii - 1) * S + jj) == DIR_FAST);
            const uint64_t stop = g.ballot(!fast);
            const int run = stop ? ctz64(stop) : GW;
            if (run > 0) {
                if (g.lane < run) posnode[jj - 1] = (int16_t)(AHEAD ? nodev : (int)r2n[ii - 1]);
                i -= run; j -= run; steps += run;
                continue;
            }
            const int d = AHEAD ? g.shfl(dv, 0) : read_dir((i - 1) * S + j);
            if (d == DIR_HORIZ) {
                if (j == 0) return RES_UNDEFINED;
                if (g.lane == 0) posnode[j - 1] = -1;
                --j;
            } else {
                const int p = dir_pred(d);
                const int k = meta_k(AHEAD ? meta0 : rowmeta[i - 1]);
                const int pi = k ? pred_row(i - 1, p) : 0;
                if (!is_vert(d)) {
                    if (j == 0) return RES_UNDEFINED;
                    if (g.lane == 0) posnode[j - 1] = (int16_t)(AHEAD ? nodev : (int)r2n[i - 1]);
                    --j;
                }
                i = pi;
            }
            ++steps;
        }
        tb_steps = g.uniform(steps); tb_fv = g.uniform(j);
        g.sync();
        HYPO_TICK(PH_TRACE);
        return RES_OK;
    }

    // ---- graph->add_alignment (graph.cpp:154-271) ---------------------------------------------------
    HD void new_node(int id, int c) {
        code[id] = (uint8_t)c; nin[id] = 0; nout[id] = 0; nal[id] = 0;
    }
    // nout[]: out-degree in bits 0-6 (only 0, 1 and "more" are ever asked); bit 7: the literal sort emitted this node from its
    // main loop — as a root whose in-edge sources, and its clique's, were all marked — and not inside a DFS (Poa::topo_insert)
    static constexpr int NOUT_RE = 0x80;
    HD int n_out(int u) const { return (int)nout[u] & 0x7f; }
    // adds edge prev->to (graph.cpp:99-115).  Returns 0 = existing edge, 1 = new edge, 2 = no room.
    HD int add_edge(int prev, int to) {
        const int k = nin[to];
        for (int p = 0; p < k; ++p)
            if ((int)inp[to * KIN + p] == prev) { inw[to * KIN + p] = (wt_t)(inw[to * KIN + p] + 2); return 0; }
        if (k == KIN) return 2;
        inp[to * KIN + k] = (id_t)prev; inw[to * KIN + k] = 2; nin[to] = (uint8_t)(k + 1);
        if (n_out(prev) != 127) nout[prev] = (uint8_t)(nout[prev] + 1);         // (seven bits, saturating; bit 7 is NOUT_RE)
        return 1;
    }
    HD int add_alignment() {
        if (L == 0) return RES_OK;
        const int fv = tb_steps == 0 ? L : tb_fv;          // empty alignment -> the whole sequence is a fresh chain
        if (tb_steps != 0 && fv == L) return RES_UNDEFINED; // graph.cpp:184-200: no sequence position aligned
        bool changed = false;
        const int n_old = n_nodes;                         // (lazy rank order: nodes from here on are new)
        n_new = 0;
        // Poa::topo_insert: up to TI_MAX new nodes, each aligned to an old clique: their positions (0xff: more, or another kind)
        const bool order_was_valid = !topo_dirty;
        uint32_t ti_q = 0; int ti_n = 0; bool ti_ok = true;
        // unaligned head [0, fv): new chain (graph.cpp:194-196,273-291)
        int head = -1;
        if (fv > 0) {
            if (n_nodes + fv > NMAX) return RES_OVERFLOW_CLEAN;           // nothing touched yet
            for (int t = g.lane; t < fv; t += GW) {
                const int id = n_nodes + t;
                new_node(id, seq[t]);
                if (t > 0) { nin[id] = 1; inp[id * KIN] = (id_t)(id - 1); inw[id * KIN] = 2; }
                if (t < fv - 1) nout[id] = 1;
                if constexpr (Cfg::LAZY) { if (lazy_on) { newid[t] = (id_t)id; newslot[t] = -1; } }    // the unaligned head goes in front of everything
                if constexpr (PK) posnode[t] = (int16_t)id;                                                 // (the guide of the next arm: Poa::thread_guided)
            }
            n_new = fv;
            head = n_nodes + fv - 1;
            head_first = n_nodes;
            n_nodes += fv;
            changed = true;
        }
        g.sync();
        // aligned part [fv, L): every position owns a distinct node / clique.  posnode[q] is rewritten
        // in place from "node the position is aligned to" to "node the position becomes".
        bool over = false, node_over = false;
        int slot_carry = -1;                               // lazy rank order: rank the new nodes of the positions so far go behind
        for (int base = fv; base < L; base += GW) {
            const int q = base + g.lane;
            const bool act = q < L;
            int kind = 0, tgt = -1, nd = -1, c = 0;      // kind 0 reuse, 1 new unaligned, 2 new aligned to nd
            int slot = -1;
            if constexpr (Cfg::LAZY) {
                if (lazy_on) {
                    // A position aligned to node nd sits in nd's clique, whose members are neighbours in the order; what the
                    // position and the unaligned positions after it add goes right behind that block (Poa::lazy_update).
                    int v = -1;
                    if (act) {
                        const int nd0 = posnode[q];
                        if (nd0 >= 0) {
                            v = (int)n2r[nd0];
                            const int ka0 = nal[nd0];
                            for (int a = 0; a < ka0; ++a) { const int rx = (int)n2r[al[nd0 * AL + a]]; v = rx > v ? rx : v; }
                        }
                    }
                    const int ex = g.scan_max_excl(v, (int)0x80000000);
                    slot = ex > v ? ex : v;
                    slot = slot > slot_carry ? slot : slot_carry;
                    slot_carry = g.shfl(slot, GW - 1);
                }
            }
            if (act) {
                nd = posnode[q]; c = seq[q];
                if (nd < 0) kind = 1;
                else if (code[nd] == c) tgt = nd;
                else {
                    kind = 2;
                    const int ka = nal[nd];
                    for (int a = 0; a < ka; ++a) {
                        const int x = al[nd * AL + a];
                        if (code[x] == c) { kind = 0; tgt = x; break; }
                    }
                }
            }
            const uint64_t nb = g.ballot(act && kind != 0);
            const int tot = popc64(nb);
            if (n_nodes + tot > NMAX) { node_over = true; break; }
            if constexpr (TOPO_INSERT) {
                if (tot) {
                    if (g.ballot(act && kind == 1) != 0 || ti_n + tot > TI_MAX) ti_ok = false;
                    else { uint64_t b2 = nb; while (b2) { ti_q |= (uint32_t)(base + ctz64(b2)) << (8 * ti_n); ++ti_n; b2 &= b2 - 1; } }
                }
            }
            if (act && kind != 0) {
                const int below = popc64(nb & ((1ull << g.lane) - 1ull));
                const int id = n_nodes + below;
                new_node(id, c);
                if constexpr (Cfg::LAZY) { if (lazy_on) { newid[n_new + below] = (id_t)id; newslot[n_new + below] = (int16_t)slot; } }
                if (kind == 2) {                           // join nd's clique (graph.cpp:229-238)
                    const int ka = nal[nd];
                    if (ka + 1 > AL) over = true;
                    else {
                        for (int a = 0; a < ka; ++a) {
                            const int x = al[nd * AL + a];
                            al[id * AL + a] = (id_t)x;
                            al[x * AL + nal[x]] = (id_t)id; nal[x] = (uint8_t)(nal[x] + 1);
                        }
                        al[id * AL + ka] = (id_t)nd; nal[id] = (uint8_t)(ka + 1);
                        al[nd * AL + ka] = (id_t)id; nal[nd] = (uint8_t)(ka + 1);
                    }
                }
                tgt = id;
            }
            if (act) posnode[q] = (int16_t)tgt;
            n_nodes = g.uniform(n_nodes + tot);
            n_new += tot;
            if (tot) changed = true;
        }
        if (g.any(over)) return RES_OVERFLOW;
        if (node_over) {
            // The node table is full.  What the positions so far did to the OLD graph is undone, so that the window can take this
            // graph to the next class (Poa::spill): the new nodes are dropped with the node count, and the only other change is
            // that new nodes joined cliques — their ids sit at the tail of the old members' aligned lists.
            n_nodes = g.uniform(n_old);
            g.sync();
            HYPO_NOUNROLL
            for (int u = g.lane; u < n_old; u += GW) {
                int k = nal[u];
                while (k > 0 && (int)al[u * AL + k - 1] >= n_old) --k;
                nal[u] = (uint8_t)k;
            }
            g.sync();
            return RES_OVERFLOW_CLEAN;
        }
        g.sync();
        // edges between consecutive positions (graph.cpp:250-258)
        int st = 0, ne = 0;
        for (int base = fv; base < L; base += GW) {
            const int q = base + g.lane;
            if (q < L) {
                const int prev = q == fv ? head : (int)posnode[q - 1];
                if (prev >= 0) { const int e = add_edge(prev, (int)posnode[q]); st = e > st ? e : st; ne += e == 1 ? 1 : 0; }
            }
        }
        const int sm = g.reduce_max(st);
        if (sm == 2) return RES_OVERFLOW;
        if (sm == 1) changed = true;
        g.sync();
        if (changed) { topo_dirty = true; meta_dirty = true; }
        if constexpr (TOPO_INSERT) {
            // the usual change — a new base or two, each a new node in an old clique with its two edges — keeps the literal order up
            // to where the new nodes go (Poa::topo_insert); anything else is sorted again
            if (changed && order_was_valid && ti_ok && ti_n > 0 && fv == 0 && !lazy_on_()) {
                if (topo_insert(ti_q, ti_n, n_old, g.reduce_add(ne))) { topo_dirty = false; HYPO_DIAG(topo_inserts += 1); }
            }
        }
        last_changed = changed;
        if constexpr (Cfg::LAZY) {
            if (lazy_on) {                                 // the order stays valid: new edges follow it, new nodes are slotted in
                if (n_new > 0) { lazy_update(n_old); HYPO_DIAG(lazy_updates += 1); }
                topo_dirty = false;
            }
        }
        return RES_OK;
    }

    // ---- lazy rank order (LONG windows of the hybrid class) --------------------------------------------------------------------
    // The reference sorts the graph again after every alignment (graph.cpp:293-353, a DFS whose order is 43 % of a LONG window
    // here), but the DP values and the traceback do not depend on WHICH topological order the rows are visited in: the
    // traceback prefers predecessors in in-edge order, not in rank order.  The reference's order can be observed in three
    // places only: the kNW end row (first sink in rank order among equal scores, sisd..cpp:279-288), the heaviest-bundle pass
    // and the MSA columns.  So a LONG window keeps A valid order with the cliques of aligned nodes as blocks of neighbours
    // (what the reference's order has too, and what makes "prev precedes the clique-mate the sequence continues on" true),
    // sorts literally before the consensus of a round, and into spare arrays when an end row is tied (Poa::align).
    // Update after add_alignment: a new node goes behind the block of the clique its position was aligned to; new nodes of
    // unaligned positions follow the previous position's.  New nodes come in sequence order with non-decreasing slots
    // (newslot[i] = rank they go behind, -1 = in front), so new rank of new node i = slot + 1 + i and an old rank r moves up
    // by the number of new nodes with slot < r: a running maximum over `cum` (kept where the recent score rows live, dead here).
    HD void lazy_update(int n_old) {
        static_assert(!Cfg::LAZY || Cfg::LAZY_RING || (Cfg::RING1 * Lay::SMAX * (int)sizeof(score_t)) / 2 >= NMAX + 1, "the shift table fits the LDS row ring");
        int16_t* cum = (int16_t*)ring1;
        const int k = n_new;
        for (int r = g.lane; r <= n_old; r += GW) cum[r] = 0;
        g.sync();
        for (int i = g.lane; i < k; i += GW) {
            const int sl = newslot[i];
            if (i == k - 1 || (int)newslot[i + 1] != sl) cum[sl + 1] = (int16_t)(i + 1);     // new nodes with slot <= sl
        }
        g.sync();
        const int chunk = (n_old + 1 + GW - 1) / GW;
        const int r0 = g.lane * chunk, r1 = r0 + chunk < n_old + 1 ? r0 + chunk : n_old + 1;
        int run = 0;
        for (int r = r0; r < r1; ++r) run = (int)cum[r] > run ? (int)cum[r] : run;
        const int ex = g.scan_max_excl(run, (int)0x80000000);
        run = ex > 0 ? ex : 0;
        for (int r = r0; r < r1; ++r) { run = (int)cum[r] > run ? (int)cum[r] : run; cum[r] = (int16_t)run; }
        g.sync();
        for (int r = g.lane; r < n_old; r += GW) {
            const int u = r2n[r];
            const int nr = r + (int)cum[r];
            r2n_alt[nr] = (id_t)u; n2r[u] = (id_t)nr;
        }
        for (int i = g.lane; i < k; i += GW) {
            const int u = newid[i];
            const int nr = (int)newslot[i] + 1 + i;
            r2n_alt[nr] = (id_t)u; n2r[u] = (id_t)nr;
        }
        if constexpr (Cfg::LAZY_RING) {                      // (the new order was written into the score ring: back into the window's own array)
            g.sync();
            for (int r = g.lane; r < n_old + k; r += GW) r2n[r] = r2n_alt[r];
        } else { id_t* t = r2n; r2n = r2n_alt; r2n_alt = t; }
        g.sync();
    }
    // The reference's own rank order where it can be observed in the middle of a window (a tied end row, several perfect end
    // rows): a literal sort into the spare arrays (packed classes: in the score ring, idle outside the row loop); n2r_alt[node] is
    // the node's rank in the reference's order afterwards.  The sort marks nodes in mark[], which the packed row loop reads as
    // sidx[]: the row metadata are rebuilt before the next row loop.
    HD int literal_order() {
        id_t* const sr = r2n; id_t* const sn = n2r;
        const bool td = topo_dirty;
        r2n = r2n_alt; n2r = n2r_alt;
        const int rc = toposort();
        r2n = sr; n2r = sn; topo_dirty = td;
        if (PK) meta_dirty = true;
        HYPO_DIAG(topo_runs += 1);
        return rc;
    }
    // of the matrix rows tie[0 .. ntie): the one the reference's order ranks first
    HD int first_of_rows(const int16_t* tie, int ntie, int* row_out) {
        const int rc = literal_order();
        if (rc != RES_OK) return rc;
        int key = 0x7fffffff;
        for (int t = g.lane; t < ntie; t += GW) { const int k2 = ((int)n2r_alt[r2n[(int)tie[t] - 1]] << 8) | t; key = k2 < key ? k2 : key; }
        const int kmin = -g.reduce_max(-key);
        *row_out = (int)tie[kmin & 255];
        g.sync();
        return RES_OK;
    }

    // ---- exact reuse of the previous alignment -------------------------------------------------------
    // The DP and its traceback depend only on the graph's topology (nodes, letters, in-edge order, rank
    // order) and on the sequence; edge weights never enter.  If sequence s is byte-identical to sequence
    // s-1 (same mode and markers) and adding s-1 created no node and no edge, the reference would compute
    // exactly the same alignment again: every position maps to the node it mapped to before.  Then only
    // the edge weights along that path are incremented (graph.cpp:104-109).  posnode[] still holds the
    // path of s-1 (rewritten in place to the nodes the positions became).
    HD bool same_as_previous(int s) const { return s > 0 && (seqtab[s] & 0x8000u) != 0; }
    // adds `count` more traversals of the path of the previous alignment: weights only
    HD int readd_alignment(int count) {
        bool bad = false;
        for (int base = 0; base < L; base += GW) {
            const int q = base + g.lane;
            if (q >= 1 && q < L) {
                const int prev = (int)posnode[q - 1], to = (int)posnode[q];
                const int k = nin[to];
                bool found = false;
                for (int p = 0; p < k; ++p)
                    if ((int)inp[to * KIN + p] == prev) { inw[to * KIN + p] = (wt_t)(inw[to * KIN + p] + 2 * count); found = true; break; }
                bad |= !found;
            }
        }
        g.sync();
        return g.any(bad) ? RES_UNDEFINED : RES_OK;
    }

    // ---- the literal order after a substitution, without sorting ------------------------------------------------------------
    // What most alignments that change the graph add is a base the column has not seen: ONE new node y (the largest id), aligned
    // to the clique of the node x its position was aligned to, with an in-edge from the node before it on the path and an
    // out-edge, appended to the in-edge list of the node after it.  If the last literal sort emitted x's clique from its main
    // loop (NOUT_RE: it reached the clique's first member as a root, found the in-edge sources of every member marked and emitted
    // member + aligned list at once, graph.cpp:311-349) and y's source is an in-edge source of x, a sort of the new graph runs
    // exactly as the old one did: nothing it visits before that root mentions y (only the clique's aligned lists and the node
    // after y do, and had the DFS reached one of those first the clique would have been emitted inside a DFS); at the root every
    // member's sources are marked, y's among them, and the clique is emitted with y LAST (add_alignment appends y to every
    // member's aligned list); from then on y is marked and never pushed.  So the new order is the old one with y behind its
    // clique, and y is NOUT_RE in turn.  Several such nodes of one alignment (not on neighbouring positions: then an edge joins
    // two of them) are the same graph as one alignment each, in position order.  Everything else — an unaligned base, a new edge
    // between old nodes, a clique emitted inside a DFS (the nodes behind an insertion) — is sorted literally.
    static constexpr bool TOPO_INSERT = HYPO_TOPO_INSERT && PK && !Cfg::LAZY;
    static constexpr int TI_MAX = 3;
    HD bool lazy_on_() const { if constexpr (Cfg::LAZY) return lazy_on; else return false; }
    HD bool topo_insert(uint32_t ti_q, int ti_n, int n_old, int new_edges) {
        // (group-uniform throughout: every lane reads the same few table entries)
        int want = 0, last_q = -2;
        for (int t = 0; t < ti_n; ++t) {
            const int q = (int)((ti_q >> (8 * t)) & 0xffu);
            if (q == last_q + 1) { DBGR(15); return false; }
            last_q = q;
            want += (q > 0 ? 1 : 0) + (q + 1 < L ? 1 : 0);
            const int y = (int)posnode[q];
            const int ka = (int)nal[y];
            const int x = (int)al[y * AL + ka - 1];           // (add_alignment: the node the position was aligned to closes y's list)
            if (y < n_old || !(nout[x] & NOUT_RE)) { DBGR(15); return false; }
            if (q > 0) {
                const int prev = (int)posnode[q - 1], k = (int)nin[x];
                bool found = false;
                for (int p = 0; p < k; ++p) found = found || (int)inp[x * KIN + p] == prev;
                if (!found || prev >= n_old) { DBGR(15); return false; }
            }
        }
        if (new_edges != want) { DBGR(15); return false; }
        for (int t = 0; t < ti_n; ++t) {
            const int q = (int)((ti_q >> (8 * t)) & 0xffu);
            const int y = (int)posnode[q];
            const int ka = (int)nal[y];
            int pos = 0;                                       // rank y takes: behind the last member of its clique
            for (int a = 0; a < ka; ++a) { const int r = (int)n2r[al[y * AL + a]]; pos = r + 1 > pos ? r + 1 : pos; }
            const int n_in = n_old + t;                        // nodes in the order so far
            id_t keep[XRPL];
            HYPO_UNROLL
            for (int i = 0; i < XRPL; ++i) { const int r = i * GW + g.lane; keep[i] = (r >= pos && r < n_in) ? r2n[r] : (id_t)0; }
            g.sync();
            HYPO_UNROLL
            for (int i = 0; i < XRPL; ++i) { const int r = i * GW + g.lane; if (r >= pos && r < n_in) { r2n[r + 1] = keep[i]; n2r[keep[i]] = (id_t)(r + 1); } }
            if (g.lane == 0) { r2n[pos] = (id_t)y; n2r[y] = (id_t)pos; nout[y] = (uint8_t)(nout[y] | NOUT_RE); }
            g.sync();
        }
        DBGR(14);
        return true;
    }

    // ---- Graph::topological_sort (graph.cpp:293-353) ---------------------------------------------
    // mark bit0 = permanently marked, bit1 = "aligned nodes already pushed by another clique member".
    HD int toposort() {
        for (int t = g.lane; t < n_nodes; t += GW) mark[t] = 0;
        g.sync();
        int cnt = 0, root = 0, guard = 0;
        while (root < n_nodes) {
            if (++guard > 8 * Cfg::STK + 64) return RES_UNDEFINED;       // cannot loop on a DAG; hang guard
            // Fast path: lane t looks at root+t.  A root that has no aligned nodes and whose in-edge sources are
            // all marked (or are lower roots of this very batch) is emitted at once by the reference's loop, in id
            // order; marked roots are skipped.  The leading run of such lanes is retired with one ballot.
            const int r = root + g.lane;
            bool pre = false, isdone = false, clq = false;
            int ka = 0;
            if (r < n_nodes) {
                isdone = mark[r] & 1;
                pre = isdone;
                if (!isdone) {
                    ka = nal[r];
                    const int k = nin[r];
                    bool ok = true;
                    for (int p = 0; p < k; ++p) {
                        const int d = inp[r * KIN + p];
                        ok &= (d >= root && d < r) || (mark[d] & 1);
                    }
                    if (ka == 0) pre = ok;
                    else if (ok) {
                        // A root with an aligned clique whose members' in-edge sources are all emitted: the reference's DFS pushes
                        // the members, finds each of them ready, marks them and emits root + members in aligned-list order
                        // (graph.cpp:311-349).  Such a root may END a run (it is the lane right after the leading plain roots).
                        bool cok = true;
                        for (int j = 0; j < ka; ++j) {
                            const int a = al[r * AL + j];
                            const int k2 = nin[a];
                            for (int p = 0; p < k2; ++p) {
                                const int d = inp[a * KIN + p];
                                cok &= (d >= root && d < r) || (mark[d] & 1);
                            }
                        }
                        clq = cok;
                    }
                }
            }
            const uint64_t full = GW == 64 ? ~0ull : ((1ull << (GW & 63)) - 1ull);
            const uint64_t bp = g.ballot(pre);
            const int run = bp == full ? GW : ctz64(~bp);
            const uint64_t cb = g.ballot(clq);
            const bool has_clq = run < GW && ((cb >> run) & 1ull);
            if (run > 0 || has_clq) {
                HYPO_DIAG(topo_fast += 1);
                const bool em = g.lane < run && !isdone;       // lanes < run have r < n_nodes (pre is false beyond)
                const uint64_t eb = g.ballot(em);
                if (em) { r2n[cnt + popc64(eb & ((1ull << g.lane) - 1ull))] = (id_t)r; mark[r] = 1; nout[r] = (uint8_t)(nout[r] | NOUT_RE); }
                cnt += popc64(eb);
                root += run;
                if (has_clq) {
                    const int kc = g.shfl(ka, run);
                    if (g.lane == run) {
                        r2n[cnt] = (id_t)r; mark[r] = 1; nout[r] = (uint8_t)(nout[r] | NOUT_RE);
                        for (int j = 0; j < ka; ++j) { const int a = al[r * AL + j]; r2n[cnt + 1 + j] = (id_t)a; mark[a] = 1; nout[a] = (uint8_t)(nout[a] | NOUT_RE); }
                    }
                    cnt += 1 + kc;
                    root += 1;
                }
                g.sync();
                continue;
            }
            // Slow path: literal DFS from `root` (unmarked, with a pending dependency or an aligned clique)
            if (g.lane == 0) stack[0] = (id_t)root;
            int sp = 1;
            g.sync();
            while (sp > 0) {
                if (++guard > 8 * Cfg::STK + 64) return RES_UNDEFINED;
                HYPO_DIAG(topo_dfs += 1);
                const int v = stack[sp - 1];
                const int mv = mark[v];
                if (mv & 1) { --sp; continue; }
                const int k = nin[v];
                const int ka = (mv & 2) ? 0 : (int)nal[v];
                int d = -1;
                if (g.lane < k) d = inp[v * KIN + g.lane];
                else if (g.lane >= KIN && g.lane - KIN < ka) d = al[v * AL + g.lane - KIN];
                const bool un = d >= 0 && !(mark[d] & 1);
                const uint64_t b = g.ballot(un);
                if (b == 0) {
                    if (g.lane == 0) {
                        mark[v] = (uint8_t)(mv | 1);
                        nout[v] = (uint8_t)n_out(v);             // (emitted inside a DFS: not NOUT_RE)
                        if (!(mv & 2)) r2n[cnt] = (id_t)v;
                    }
                    if (!(mv & 2)) {
                        if (g.lane >= KIN && g.lane - KIN < ka) r2n[cnt + 1 + g.lane - KIN] = (id_t)d;
                        cnt += 1 + ka;
                    }
                    --sp;
                } else {
                    const int np = popc64(b);
                    if (sp + np > Cfg::STK) return RES_OVERFLOW;
                    if (un) {
                        stack[sp + popc64(b & ((1ull << g.lane) - 1ull))] = (id_t)d;
                        if (g.lane >= KIN) mark[d] = (uint8_t)(mark[d] | 2);
                    }
                    sp += np;
                }
                g.sync();
            }
            root += 1;
        }
        for (int r = g.lane; r < n_nodes; r += GW) n2r[r2n[r]] = (id_t)r;
        topo_dirty = false;
        g.sync();
        return cnt == n_nodes ? RES_OK : RES_UNDEFINED;
    }

    HD int add_sequence_step(int mode, int m, int n, int gp) {
        int rc = align(mode, m, n, gp);
        if (rc != RES_OK) { if (rc == RES_OVERFLOW) stat_set(ST_CKIND, CARRY_BEFORE); return rc; }       // align() never changes the graph
        if (threaded) {
            // a threaded sequence adds no node and no edge (graph.cpp:154-271 would find every node and edge in place): only the
            // weights along its path grow, the rank order and the row metadata stay valid
            rc = weights_done ? (int)RES_OK : readd_alignment(1);
            last_changed = false;
            guide_len = L; guide_mode = mode;
            HYPO_TICK(PH_ADD);
            return rc;
        }
        guide_len = -1;
        rc = add_alignment();
        HYPO_TICK(PH_ADD);
        if (rc == RES_OVERFLOW_CLEAN) {
            // the alignment just made is made again by the class that takes the window over: it is counted there
            if (g.lane == 0) {
                stat[ST_CKIND] = CARRY_BEFORE;
                stat[ST_CELLS] -= (uint32_t)((n_nodes + 1) * (L + 1)); stat[ST_ALIGNS] -= 1; stat[ST_XHITS] -= stat[ST_LASTX];
            }
            return RES_OVERFLOW;
        }
        if (rc != RES_OK) return rc;
        if constexpr (PK) { guide_len = L; guide_mode = mode; }       // posnode[] = the node of every position (the sort below leaves it alone: its stack is in the ring)
        if (topo_dirty) {
            rc = toposort(); HYPO_DIAG(topo_runs += 1); HYPO_TICK(PH_TOPO);
            if constexpr (Cfg::LAZY_RING) {
                // A window whose graph keeps changing goes over to the lazy rank order (Poa::lazy_update) after its second literal sort:
                // the literal order it has by now is a valid one with the cliques as blocks, which is all the lazy order asks for.  One
                // or two sorts are cheaper than the updates plus the literal sort the consensus then needs (C2 at 0.2 % read error:
                // lazy from the first sequence on cost 1.90 -> 2.20 ms per call; from 3 % on it gains 25 %).
                if (g.lane == 0) stat[ST_NSORT] += 1;
                g.sync();
                if (stat[ST_NSORT] >= 2u && !(P->flags & POA_NATIVE_KLOV)) lazy_on = true;
            }
            if (rc == RES_OVERFLOW) stat_set(ST_CKIND, CARRY_UNSORTED);                                 // (the DFS stack: the graph itself is complete)
        }
        return rc;
    }

    // ---- carrying a SHORT window's graph to the class it is re-queued to ------------------------------------------------------
    // A window that outgrows its class used to start again from its first sequence in the next one.  The graph after the
    // sequences added so far is the same in every class (node ids, letters, in-edge order, aligned lists, rank order are what
    // the reference would have at that point), so it travels: spill() writes it in a class-independent form (16-bit ids and
    // weights, the source class's in-edge stride), restore() reads it into the receiving class's tables, and run_short()
    // continues with sequence `carry_s`.  Layout: 32-byte header ({magic, n_nodes, carry_s, chain0, kind | kin << 8, x_tries,
    // x_hits, 0} as uint16, {cells, alignments, reused, threaded} so far as uint32), then code / nin / nout / nal (n bytes each, the block padded to 16), r2n, n2r (n x u16 each), inp, inw
    // (n x kin x u16 each), al (n x AL x u16).
    static constexpr uint32_t CARRY_MAGIC = 0x4879u;
    HD static uint32_t spill_bytes(int n, int kin) {
        return (uint32_t)(32 + align_up<16>(4 * n) + 2 * align_up<16>(2 * n) + 2 * align_up<16>(2 * n * kin) + align_up<16>(2 * n * AL));
    }
    HD uint32_t spill_size() const { return spill_bytes(n_nodes, KIN); }
    HD void spill(uint8_t* out) const {
        // (every store device-coherent: the class that takes the window over may already be running on another XCD, HYPO_ST_DEV in grp.hpp)
        const int n = n_nodes;
        uint16_t* h = (uint16_t*)out;
        if (g.lane == 0) {
            HYPO_ST_DEV(h + 0, (uint16_t)CARRY_MAGIC); HYPO_ST_DEV(h + 1, (uint16_t)n); HYPO_ST_DEV(h + 2, (uint16_t)stat[ST_CS]); HYPO_ST_DEV(h + 3, (uint16_t)stat[ST_CCHAIN0]);
            // (bit 7 of the kind: r2n / n2r are A valid order kept lazily, not the reference's — a class that needs the latter sorts first)
            HYPO_ST_DEV(h + 4, (uint16_t)(stat[ST_CKIND] | ((Cfg::LAZY && lazy_on) ? 0x80u : 0u) | (KIN << 8))); HYPO_ST_DEV(h + 5, (uint16_t)stat[ST_XT]); HYPO_ST_DEV(h + 6, (uint16_t)stat[ST_XH]); HYPO_ST_DEV(h + 7, (uint16_t)0);
            // the reference-equivalent work of the sequences behind the cursor is accounted by whoever finishes the window
            uint32_t* c = (uint32_t*)(out + 16);
            HYPO_ST_DEV(c + 0, (uint32_t)stat[ST_CELLS]); HYPO_ST_DEV(c + 1, (uint32_t)stat[ST_ALIGNS]); HYPO_ST_DEV(c + 2, (uint32_t)stat[ST_REUSED]); HYPO_ST_DEV(c + 3, (uint32_t)stat[ST_XHITS]);
        }
        uint8_t* b = out + 32;
        HYPO_NOUNROLL
        for (int u = g.lane; u < n; u += GW) { HYPO_ST_DEV(b + u, (uint8_t)code[u]); HYPO_ST_DEV(b + n + u, (uint8_t)nin[u]); HYPO_ST_DEV(b + 2 * n + u, (uint8_t)n_out(u)); HYPO_ST_DEV(b + 3 * n + u, (uint8_t)nal[u]); }
        uint16_t* o_r2n = (uint16_t*)(b + align_up<16>(4 * n));
        uint16_t* o_n2r = o_r2n + align_up<16>(2 * n) / 2;
        uint16_t* o_inp = o_n2r + align_up<16>(2 * n) / 2;
        uint16_t* o_inw = o_inp + align_up<16>(2 * n * KIN) / 2;
        uint16_t* o_al = o_inw + align_up<16>(2 * n * KIN) / 2;
        HYPO_NOUNROLL
        for (int u = g.lane; u < n; u += GW) { HYPO_ST_DEV(o_r2n + u, (uint16_t)r2n[u]); HYPO_ST_DEV(o_n2r + u, (uint16_t)n2r[u]); }
        HYPO_NOUNROLL
        for (int t = g.lane; t < n * KIN; t += GW) { HYPO_ST_DEV(o_inp + t, (uint16_t)inp[t]); HYPO_ST_DEV(o_inw + t, (uint16_t)inw[t]); }   // (slots beyond nin[u] hold garbage: never read)
        HYPO_NOUNROLL
        for (int t = g.lane; t < n * AL; t += GW) HYPO_ST_DEV(o_al + t, (uint16_t)al[t]);
    }
    // RES_OK, or RES_OVERFLOW when the graph does not fit this class either (the caller passes the window on with the same spill)
    HD int restore(const uint8_t* in, int* s_out, int* chain0_out) {
        const uint16_t* h = (const uint16_t*)in;
        // (device-coherent loads: the spill may have been written a moment ago by a wave on another XCD, HYPO_LD_DEV in grp.hpp)
        if (HYPO_LD_DEV(h + 0) != (uint16_t)CARRY_MAGIC) return RES_INVALID;
        const int h4 = HYPO_LD_DEV(h + 4);
        const int n = HYPO_LD_DEV(h + 1), kin = h4 >> 8, kind = h4 & 0x7f;
        const bool lazy_order = (h4 & 0x80) != 0;
        if (n > NMAX || n < 1) return RES_OVERFLOW;
        const uint8_t* b = in + 32;
        bool over = false;
        HYPO_NOUNROLL
        for (int u = g.lane; u < n; u += GW) {
            const int k = HYPO_LD_DEV(b + n + u);
            if (k > KIN) over = true;
            code[u] = HYPO_LD_DEV(b + u); nin[u] = (uint8_t)k; nout[u] = HYPO_LD_DEV(b + 2 * n + u); nal[u] = HYPO_LD_DEV(b + 3 * n + u);
        }
        if (g.any(over)) return RES_OVERFLOW;
        const uint16_t* i_r2n = (const uint16_t*)(b + align_up<16>(4 * n));
        const uint16_t* i_n2r = i_r2n + align_up<16>(2 * n) / 2;
        const uint16_t* i_inp = i_n2r + align_up<16>(2 * n) / 2;
        const uint16_t* i_inw = i_inp + align_up<16>(2 * n * kin) / 2;
        const uint16_t* i_al = i_inw + align_up<16>(2 * n * kin) / 2;
        g.sync();
        HYPO_NOUNROLL
        for (int u = g.lane; u < n; u += GW) {
            r2n[u] = (id_t)HYPO_LD_DEV(i_r2n + u); n2r[u] = (id_t)HYPO_LD_DEV(i_n2r + u);
            const int k = nin[u];
            HYPO_NOUNROLL
            for (int p = 0; p < k; ++p) { inp[u * KIN + p] = (id_t)HYPO_LD_DEV(i_inp + u * kin + p); inw[u * KIN + p] = (wt_t)HYPO_LD_DEV(i_inw + u * kin + p); }
            const int ka = nal[u];
            HYPO_NOUNROLL
            for (int a = 0; a < ka; ++a) al[u * AL + a] = (id_t)HYPO_LD_DEV(i_al + u * AL + a);
        }
        n_nodes = g.uniform(n);
        *s_out = HYPO_LD_DEV(h + 2); *chain0_out = HYPO_LD_DEV(h + 3);
        if (g.lane == 0) {
            stat[ST_XT] = HYPO_LD_DEV(h + 5); stat[ST_XH] = HYPO_LD_DEV(h + 6);
            const uint32_t* c = (const uint32_t*)(in + 16);
            stat[ST_CELLS] = HYPO_LD_DEV(c + 0); stat[ST_ALIGNS] = HYPO_LD_DEV(c + 1); stat[ST_REUSED] = HYPO_LD_DEV(c + 2); stat[ST_XHITS] = HYPO_LD_DEV(c + 3);
        }
        if (lazy_order && Cfg::LAZY_RING && !(P->flags & POA_NATIVE_KLOV)) lazy_on = true;      // (a window that went lazy stays lazy in the class that takes it over)
        topo_dirty = kind == CARRY_UNSORTED || (lazy_order && !(Cfg::LAZY && lazy_on)); meta_dirty = true; last_changed = true;
        g.sync();
        return RES_OK;
    }

    // ---- heaviest bundle, all lanes (graph.cpp:610-658 when no tie rule and no branch completion is involved) ------
    // The reference scores nodes in rank order: score = weight of the heaviest in-edge + score of its source (sources
    // -1), and among in-edges of equal weight the one whose source scores >= wins.  When no node has two in-edges
    // sharing its maximum weight, the chosen in-edge does not depend on any score, so the scores are sums along the
    // chosen-predecessor forest: pointer doubling over ranks (log2(n) rounds, every lane busy) instead of one lane
    // walking n nodes with a dependent LDS round trip each.  The doubling table is kept: position t of the
    // consensus path is the t-th ancestor of the best node, found by all lanes at once.  Returns CONS_SERIAL when a
    // weight tie, or a best node that is not a sink (branch completion), needs the literal pass below.
    static constexpr int CONS_SERIAL = -2;
    static constexpr int NN = NMAX + 1;                     // ranks + one "no predecessor" sentinel
    static constexpr int LV = NN <= 64 ? 6 : (NN <= 128 ? 7 : 8);
    static constexpr int RPL = (NN + GW - 1) / GW;          // ranks per lane
    static constexpr bool FAST_CONS = sizeof(id_t) == 1 && NN <= 255 && RPL <= 16;
    static_assert(!FAST_CONS || 4 * NN + LV * NN + 1 + 2 * NMAX <= (int)sizeof(score_t) * Cfg::RINGCELLS + Cfg::DIRBYTES_LDS, "fast consensus scratch aliases ring+dir");
    HD int consensus_fast(int16_t** path_out) {
        int32_t* acc = (int32_t*)ring;                       // (sum of chosen weights << 8) + nodes on the chain, by rank
        uint8_t* up = (uint8_t*)(acc + NN);                  // up[lv][r]: 2^lv-th ancestor of rank r (n = none)
        int16_t* path = (int16_t*)(up + ((LV * NN + 1) & ~1));
        const int n = n_nodes;
        bool tie = false;
        for (int r = g.lane; r <= n; r += GW) {
            int a = 0, pr = n;
            if (r < n) {
                const uint32_t meta = rowmeta[r];
                const int k = meta_k(meta);
                a = -255;                                    // a source scores -1 and is one node
                if (k != 0) {
                    const int u = r2n[r];
                    int bw = inw[u * KIN], bp = meta_p0(meta);
                    bool tied = false;
                    for (int p = 1; p < k; ++p) {
                        const int w = inw[u * KIN + p];
                        if (w > bw) { bw = w; bp = pred_row(r, p); tied = false; }
                        else if (w == bw) tied = true;
                    }
                    tie |= tied;
                    a = (bw << 8) + 1; pr = bp - 1;
                }
            }
            acc[r] = a; up[r] = (uint8_t)pr;
        }
        g.sync();
        if (g.any(tie)) return CONS_SERIAL;
        for (int lv = 1; lv <= LV; ++lv) {                  // after round lv a rank has summed 2^lv chain nodes; tables 0..LV-1 are kept
            int na[RPL], nacc[RPL];
            const uint8_t* prev = up + (lv - 1) * NN;
            HYPO_UNROLL
            for (int q = 0; q < RPL; ++q) {
                const int r = g.lane + q * GW;
                if (r <= n) { const int a = prev[r]; na[q] = prev[a]; nacc[q] = acc[r] + acc[a]; }
            }
            g.sync();
            HYPO_UNROLL
            for (int q = 0; q < RPL; ++q) {
                const int r = g.lane + q * GW;
                if (r <= n) { if (lv < LV) up[lv * NN + r] = (uint8_t)na[q]; acc[r] = nacc[q]; }
            }
            g.sync();
        }
        // best node: first strictly greater score in rank order, node 0 if every score is -1 (graph.cpp:618,633-635)
        int key = 0;
        for (int r = g.lane; r < n; r += GW) {
            const int k2 = (((acc[r] >> 8) + 1) << 8) | (255 - r);
            key = k2 > key ? k2 : key;
        }
        key = g.reduce_max(key);
        const int best_r = (key >> 8) == 0 ? (int)n2r[0] : 255 - (key & 255);
        const int max_id = r2n[best_r];
        if (n_out(max_id) != 0) return CONS_SERIAL;
        const int len = g.uniform(acc[best_r] & 255);
        for (int t = g.lane; t < len; t += GW) {
            int x = best_r;
            HYPO_UNROLL
            for (int lv = 0; lv < LV; ++lv) if ((t >> lv) & 1) x = up[lv * NN + x];
            path[t] = (int16_t)r2n[x];
        }
        g.sync();
        *path_out = path;                                   // reversed: path[len-1] is the first node
        return len;
    }

    // ---- Graph::generate_consensus (graph.cpp:467-476,610-705) --------------------------------------
    // Scratch aliases ring + dir: score[NMAX] i32, pred[NMAX] i16, path[NMAX] i16, rs[NMAX] i32, w0r[NMAX] u16.
    HD int consensus(int16_t** path_out) {
        if (meta_dirty) build_rowmeta();
        if (FAST_CONS) {
            const int fl = consensus_fast(path_out);
            if (fl != CONS_SERIAL) return fl;
            HYPO_DIAG(cons_serial += 1);
        }
        int32_t* score = (int32_t*)ring;                   // by node id
        int16_t* pred = (int16_t*)(score + NMAX);
        int16_t* path = pred + NMAX;
        int32_t* rs = (int32_t*)(path + NMAX);             // by rank
        uint16_t* w0r = (uint16_t*)(rs + NMAX);            // weight of in-edge 0, by rank
        g.sync();
        for (int r = g.lane; r < n_nodes; r += GW) {
            const int u = r2n[r];
            score[u] = -1; pred[u] = -1;
            w0r[r] = nin[u] ? inw[u * KIN] : (uint16_t)0;
        }
        g.sync();
        int max_id = 0;
        if (g.lane == 0) {
            // One pass in rank order (graph.cpp:615-636).  For a node whose only predecessor is the previous
            // row the running score stays in a register; everything else it needs was prefetched.
            int best_val = -1, prev_s = 0;
            uint32_t meta_n = rowmeta[0]; int w_n = w0r[0]; int u_n = r2n[0];
            for (int r = 0; r < n_nodes; ++r) {
                const uint32_t meta = meta_n; const int w0 = w_n; const int u = u_n;
                if (r + 1 < n_nodes) { meta_n = rowmeta[r + 1]; w_n = w0r[r + 1]; u_n = r2n[r + 1]; }
                const int k = meta_k(meta), p0 = meta_p0(meta);
                int sc = -1, pd = -1;
                if (k != 0) {
                    int bw = w0, bs = p0 == r ? prev_s : rs[p0 - 1], bp = p0;
                    for (int p = 1; p < k; ++p) {
                        const int w = inw[u * KIN + p], pr = pred_row(r, p);
                        const int sp = rs[pr - 1];
                        if (bw < w || (bw == w && bs <= sp)) { bw = w; bs = sp; bp = pr; }
                    }
                    sc = bw + bs; pd = (int)r2n[bp - 1];
                }
                rs[r] = sc; score[u] = sc; pred[u] = (int16_t)pd;
                if (u == max_id) best_val = sc;             // scores[max_score_id] < scores[node_id], :633-635
                else if (best_val < sc) { max_id = u; best_val = sc; }
                prev_s = sc;
            }
        }
        max_id = g.shfl(max_id, 0);
        g.sync();
        // branch completion (graph.cpp:660-705) while the best node is not a sink
        int rounds = 0;
        while (n_out(max_id) != 0) {
            if (++rounds > n_nodes) return -1;             // hang guard (cannot happen on a DAG)
            // invalidate the other sources feeding max_id's successors
            for (int t = g.lane; t < n_nodes; t += GW) {
                const int k = nin[t];
                bool succ = false;
                for (int p = 0; p < k; ++p) succ |= ((int)inp[t * KIN + p] == max_id);
                if (succ) for (int p = 0; p < k; ++p) { const int b = inp[t * KIN + p]; if (b != max_id) score[b] = -1; }
            }
            g.sync();
            int nxt = 0;
            if (g.lane == 0) {
                int ms = 0;
                for (int r = (int)n2r[max_id] + 1; r < n_nodes; ++r) {
                    const int u = r2n[r];
                    const int k = nin[u];
                    // graph.cpp:683-696.  The chosen predecessor's score travels in a register: written as
                    // `s < w || (s == w && score[pd] <= score[b])` this loop came out of hipcc (ROCm 7.2, gfx950) keeping the
                    // first of two equal-weight, equal-score in-edges on the device while the same source is right on the CPU
                    // (found by the messy end-to-end seeds; pinned by tests/test_gpu_poa.py::test_branch_completion_tie).
                    int s = -1, pd = -1, spd = 0;
                    for (int p = 0; p < k; ++p) {
                        const int w = inw[u * KIN + p], b = inp[u * KIN + p];
                        const int sb = score[b];
                        if (sb == -1) continue;
                        bool take = s < w;
                        if (!take && s == w) take = spd <= sb;
                        if (take) { s = w; pd = b; spd = sb; }
                    }
                    if (pd != -1) s += spd;
                    score[u] = s; pred[u] = (int16_t)pd;
                    if (ms < s) { ms = s; nxt = u; }
                }
            }
            max_id = g.shfl(nxt, 0);
            g.sync();
        }
        int len = 0;
        if (g.lane == 0) {
            int u = max_id;
            while (pred[u] != -1 && len < n_nodes) { path[len++] = (int16_t)u; u = pred[u]; }
            path[len++] = (int16_t)u;
        }
        len = g.shfl(len, 0);
        g.sync();
        *path_out = path;          // reversed: path[len-1] is the first node
        return len;
    }

    // ---- LONG windows: Window::generate_consensus_long + curate (src/Window.cpp:156-254) ----------------
    // reads a counter that atomics have updated: the atomics execute in L2, an ordinary load could hit an older copy in this
    // CU's vector L1
    HD static uint32_t counter_load(const uint32_t* p) {
#ifdef HYPO_EMU
        return *p;
#else
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    HD static void atomic_inc(uint32_t* p, uint32_t v) {
#ifdef HYPO_EMU
        *p += v;
#else
        atomicAdd(p, v);
#endif
    }
    // path of the sequence just added (graph.cpp:30-41: successor(label) follows exactly this path)
    HD int record_path(int fv) {
        if (n_paths >= Cfg::SEQMAX || path_used + L > Cfg::PATHCAP) return RES_OVERFLOW;
        for (int q = g.lane; q < L; q += GW) pathnodes[path_used + q] = (id_t)(q < fv ? head_first + q : (int)posnode[q]);
        if (g.lane == 0) { pathoff[n_paths] = (uint32_t)path_used; pathlen[n_paths] = (uint16_t)L; pathmult[n_paths] = 1; }
        n_paths += 1; path_used += L;
        g.sync();
        return RES_OK;
    }
    HD int load_codes(const uint8_t* src, int len) {       // backbone of round 2: the curated consensus
        L = len;
        if (L > Cfg::LMAX) return RES_OVERFLOW;
        for (int t = g.lane; t < L; t += GW) seq[t] = src[t];
        g.sync();
        return RES_OK;
    }
    HD int long_step(int m, int n, int gp) {
        int rc = align(MODE_NW, m, n, gp);
        if (rc != RES_OK) return rc;
        if ((rc = add_alignment()) != RES_OK) return rc == RES_OVERFLOW_CLEAN ? (int)RES_OVERFLOW : rc;      // (LONG windows start over in the last class)
        if ((rc = record_path(tb_steps == 0 ? L : tb_fv)) != RES_OK) return rc;   // before toposort: its stack aliases posnode
        HYPO_TICK(PH_ADD);
        if (topo_dirty) { rc = toposort(); HYPO_DIAG(topo_runs += 1); HYPO_TICK(PH_TOPO); }
        return rc;
    }
    HD int run_long(uint32_t w, const HypoWindow& W) {
        const int m = P->lr_m, n = P->lr_n, gp = P->lr_g;
        const uint8_t* d4 = P->draft4 + W.draft_off;
        int n_seq = 0; bool added = false;
        int rc = build_seqtab(W, true, &n_seq, &added);
        if (rc != RES_OK) return rc;
        if (!added) return emit_draft(w, d4, (int)W.draft_len);      // Window.cpp:233-235
        const unsigned thr = (unsigned)floorf((float)W.n_internal * 0.4f);   // Window.cpp:28,245
        int conslen = 0;
        for (int round = 0; round < 2; ++round) {
            n_nodes = 0; topo_dirty = false; meta_dirty = true; n_paths = 0; path_used = 0; last_changed = true;
            lazy_on = Cfg::LAZY;
            bool prev_aligned = false;
            int s = 0;
            if (round == 1) {                                // backbone = round-1 consensus (skipped when empty)
                s = 1;
                if (conslen > 0) {
                    if ((rc = load_codes(consbuf, conslen)) != RES_OK) return rc;
                    if ((rc = long_step(m, n, gp)) != RES_OK) return rc;
                }
            }
            while (s < n_seq) {
                if (prev_aligned && !last_changed && tb_steps != 0 && tb_fv == 0) {
                    int c = 0;
                    while (s < n_seq && same_as_previous(s)) { ++c; ++s; }
                    if (c) {
                        if (g.lane == 0) { stat[ST_CELLS] += (uint32_t)(c * (n_nodes + 1) * (L + 1)); stat[ST_ALIGNS] += c; stat[ST_REUSED] += c; }
                        if ((rc = readd_alignment(c)) != RES_OK) return rc;
                        if (g.lane == 0) pathmult[n_paths - 1] = (uint16_t)(pathmult[n_paths - 1] + c);
                        g.sync();
                        if (s >= n_seq) break;
                    }
                }
                int mode;
                if ((rc = load_seq(W, s, &mode)) != RES_OK) return rc;
                ++s;
                if (L == 0) { prev_aligned = false; continue; }
                if ((rc = long_step(m, n, gp)) != RES_OK) return rc;
                prev_aligned = true;
            }
            // generate_consensus_custom (graph.cpp:533-568)
            if constexpr (Cfg::LAZY) {                       // the heaviest bundle and the MSA columns see the reference's own order
                if (n_nodes > 0) { if ((rc = toposort()) != RES_OK) return rc; meta_dirty = true; HYPO_DIAG(topo_runs += 1); HYPO_TICK(PH_TOPO); }   // (row metadata is by rank)
                lazy_on = false;
            }
            int16_t* path;
            const int len = consensus(&path);
            if (len < 1) return RES_UNDEFINED;
            // MSA column of every node: cliques share a column (graph.cpp:371-388)
            if (g.lane == 0) {
                int col = 0;
                for (int i = 0; i < n_nodes; ++i) {
                    const int u = r2n[i];
                    const int ka = nal[u];
                    msa[u] = (uint16_t)col;
                    for (int j = 0; j < ka; ++j) msa[r2n[++i]] = (uint16_t)col;
                    ++col;
                }
            }
            for (int c = g.lane; c < len; c += GW) dstcnt[c] = 0;
            g.sync();
            for (int sidx = g.lane; sidx < n_paths; sidx += GW) {   // one lane walks one sequence's path
                const id_t* pn = pathnodes + pathoff[sidx];
                const int pl = pathlen[sidx]; const uint32_t mult = pathmult[sidx];
                int c = 0;
                for (int q = 0; q < pl; ++q) {
                    const int v = pn[q];
                    const int mv = msa[v];
                    while (c < len && (int)msa[path[len - 1 - c]] < mv) ++c;
                    if (c >= len) break;
                    const int cn = path[len - 1 - c];
                    if ((int)msa[cn] == mv && code[v] == code[cn]) atomic_inc(&dstcnt[c], mult);
                }
            }
            g.sync();
            // curate (Window.cpp:239-254): keep position i iff dst[i] >= floor(n_internal * 0.4f)
            int o = 0;
            for (int base = 0; base < len; base += GW) {
                const int c = base + g.lane;
                const bool keep = c < len && counter_load(&dstcnt[c]) >= thr;
                const uint64_t kb = g.ballot(keep);
                // consbuf may alias nothing else; positions only move left, chunk by chunk
                const int cd = c < len ? (int)code[path[len - 1 - c]] : 0;
                g.sync();
                if (keep) consbuf[o + popc64(kb & ((1ull << g.lane) - 1ull))] = (uint8_t)cd;
                o += popc64(kb);
            }
            conslen = o;
            g.sync();
            HYPO_TICK(PH_CONS);
        }
        uint64_t oo, cap; out_range(w, &oo, &cap);
        if ((uint64_t)conslen > cap) { finish(w, HYPO_ST_CONS_OVERFLOW, (uint32_t)conslen); return RES_OK; }
        for (int t = g.lane; t < conslen; t += GW) P->out_bases[oo + t] = "ACGTNJO"[consbuf[t]];
        finish(w, HYPO_ST_OK, (uint32_t)conslen);
        return RES_OK;
    }

    // ---- outputs -----------------------------------------------------------------------------------
    HD void finish(uint32_t w, int status, uint32_t len) const {
        if (g.lane == 0) { P->out_len[w] = len; P->out_status[w] = (uint8_t)status; if constexpr (Hook::enabled) stat[ST_OLEN] = len; }
    }
    HD int emit_draft(uint32_t w, const uint8_t* d4, int dlen) const {
        uint64_t o, cap; out_range(w, &o, &cap);
        if ((uint64_t)dlen > cap) { finish(w, HYPO_ST_CONS_OVERFLOW, (uint32_t)dlen); return RES_OK; }
        for (int t = g.lane; t < dlen; t += GW) {
            int c = (d4[t >> 1] >> (4 - 4 * (t & 1))) & 15;
            P->out_bases[o + t] = "ACGTN"[c < 4 ? c : 4];
        }
        finish(w, HYPO_ST_OK, (uint32_t)dlen);
        return RES_OK;
    }

    // Window::generate_consensus_short (src/Window.cpp:87-154)
    HD int run_short(uint32_t w, const HypoWindow& W, const uint8_t* carry_in) {
        const int m = P->sr_m, n = P->sr_n, gp = P->sr_g;
        const uint8_t* d4 = P->draft4 + W.draft_off;
        n_nodes = 0; topo_dirty = false; meta_dirty = true;
        // SHORT windows that ended up in the hybrid class (wide windows of --ccs-windows, large graphs) keep their rank order
        // lazily as well: kLOV's end row (first row in rank order among equal maxima of the last column) goes through the same
        // list of tied rows as the sinks of kNW / kROV.  Not with the native kLOV flavour, whose end row is ranked differently.
        lazy_on = Cfg::LAZY && !Cfg::LAZY_RING && !(P->flags & POA_NATIVE_KLOV);      // (packed classes: after the window's second literal sort, Poa::add_sequence_step)
        stat_set(ST_NSORT, 0u);
        int n_seq = 0; bool added = false;
        int rc = build_seqtab(W, false, &n_seq, &added);
        if (rc != RES_OK) return rc;
        bool prev_aligned = false;                           // the previous non-reused sequence went through align()
        int s = 0;
        int chain0_in = 0;
        if (carry_in) {                                      // the graph another class built from sequences [0, s): continue from there
            if ((rc = restore(carry_in, &s, &chain0_in)) != RES_OK) return rc;
            if (s > n_seq) return RES_INVALID;
            if (topo_dirty) { if ((rc = toposort()) != RES_OK) return rc; HYPO_DIAG(topo_runs += 1); HYPO_TICK(PH_TOPO); }
            // (a window that outgrew another class keeps changing: it goes lazy at once)
            if constexpr (Cfg::LAZY_RING) { if (!(P->flags & POA_NATIVE_KLOV)) lazy_on = true; }
        }
        // A window that outgrows this class's node table says how many nodes it will probably need, so that it is re-queued
        // straight into a class that holds it instead of climbing one class at a time: the sequences added so far grew the
        // graph from its first chain of `chain0` nodes to n_nodes, the remaining ones are assumed to add as many each.
        int chain0 = chain0_in;
        auto project = [&](int rc_) -> int {
            if (rc_ == RES_OVERFLOW && s > 0 && n_nodes > 0) {
                const int grown = n_nodes - chain0 > 0 ? n_nodes - chain0 : 0;
                if (g.lane == 0) {
                    stat[ST_NEED] = (uint32_t)(n_nodes + (int)(((int64_t)grown * (n_seq - s) + s - 1) / s) + 4);
                    // what travels (Poa::spill): the sequence in hand is s - 1; it is aligned again by the receiving class unless
                    // only its topological sort is missing
                    const uint32_t kind = stat[ST_CKIND];
                    stat[ST_CS] = (uint32_t)(kind == CARRY_UNSORTED ? s : s - 1);
                    stat[ST_CCHAIN0] = (uint32_t)(chain0 ? chain0 : (kind == CARRY_UNSORTED ? n_nodes : 0));
                }
            } else stat_set(ST_CKIND, CARRY_NONE);
            g.sync();
            return rc_;
        };
        while (s < n_seq) {
            if (prev_aligned && !last_changed && tb_steps != 0 && tb_fv == 0) {
                // run of arms identical to the one just aligned: L and posnode[] are still its path
                int c = 0;
                while (s < n_seq && same_as_previous(s)) { ++c; ++s; }
                if (c) {
                    if (g.lane == 0) { stat[ST_CELLS] += (uint32_t)(c * (n_nodes + 1) * (L + 1)); stat[ST_ALIGNS] += c; stat[ST_REUSED] += c; }   // work the reference does
                    if ((rc = readd_alignment(c)) != RES_OK) return project(rc);
                    HYPO_TICK(PH_ADD);
                    if (s >= n_seq) break;
                }
            }
            int mode;
            if ((rc = load_seq(W, s, &mode)) != RES_OK) return rc;
            ++s;
            if (L == 0) { prev_aligned = false; continue; }  // zero-length arms are skipped (Window.cpp:100,113,124)
            if ((rc = add_sequence_step(mode, m, n, gp)) != RES_OK) return project(rc);
            if (chain0 == 0) chain0 = n_nodes;
            prev_aligned = true;
        }
        if (!added) return emit_draft(w, d4, (int)W.draft_len);
        if constexpr (Cfg::LAZY) {                           // the heaviest bundle sees the reference's own order
            if (lazy_on && n_nodes > 0) { if ((rc = toposort()) != RES_OK) return rc; meta_dirty = true; HYPO_TICK(PH_TOPO); }
            lazy_on = false;
        }
        int16_t* path;
        const int len = consensus(&path);
        HYPO_TICK(PH_CONS);
        if (len < 2) return RES_UNDEFINED;                  // Window.hpp:144 strips two markers
        const int olen = len - 2;
        uint64_t o, cap; out_range(w, &o, &cap);
        if ((uint64_t)olen > cap) { finish(w, HYPO_ST_CONS_OVERFLOW, (uint32_t)olen); return RES_OK; }
        for (int t = g.lane; t < olen; t += GW) P->out_bases[o + t] = "ACGTNJO"[code[path[len - 2 - t]]];
        finish(w, HYPO_ST_OK, (uint32_t)olen);
        HYPO_TICK(PH_OUT);
        return RES_OK;
    }

    // Window::generate_consensus (src/Window.cpp:44-61)
    HD int run(uint32_t w, const uint8_t* carry_in = nullptr) {
        const HypoWindow W = P->windows[w];
        return run_window(w, W, carry_in);
    }
    // ... with the descriptor in hand (the persistent kernel's prefetch has it in LDS when the window starts)
    HD int run_window(uint32_t w, const HypoWindow& W, const uint8_t* carry_in) {
        // the object outlives the window (one per persistent group): per-window counters and flags start over here
        last_changed = true;
        g.sync();
        if (g.lane < ST_N) stat[g.lane] = 0;
        g.sync();
        HYPO_DIAG(rows_done = 0; topo_runs = 0; cons_serial = 0; rows_slow = 0; exact_tries = 0; guided_hits = 0; rows_scored_n = 0; topo_dfs = 0; topo_fast = 0; one_sub_hits = 0; cols_hits = 0; topo_inserts = 0; lazy_updates = 0; tie_sorts = 0);
        n_paths = 0; path_used = 0; head_first = 0; L = 0; tb_steps = 0; tb_fv = 0;
        lazy_on = false; n_new = 0; guide_len = -1;
        for (int i = 0; i < PH_N; ++i) tphase[i] = 0;
        HYPO_TICK_RESET();
        const uint32_t ne = W.n_internal + W.n_prefix + W.n_suffix;
        // a descriptor that points outside the batch's buffers is answered with HYPO_ST_INVALID, never followed
        if ((uint64_t)W.n_internal + W.n_prefix + W.n_suffix > P->n_arms || (uint64_t)W.first_arm + ne > P->n_arms ||
            W.draft_off > P->draft4_bytes || ((uint64_t)W.draft_len + 1) / 2 > P->draft4_bytes - W.draft_off) return RES_INVALID;
        if constexpr (Hook::enabled) {
            // (windows answered without staging their arms: the packed arm bytes of the statistics are summed here, rare)
            if ((W.n_empty > ne || ne < 2) && g.lane == 0) {
                uint32_t a = 0;
                HYPO_NOUNROLL
                for (uint32_t t = 0; t < ne; ++t) a += (P->arm_len[W.first_arm + t] + 3) >> 2;
                stat[ST_ARMB] = a;
            }
        }
        if (W.n_empty > ne) { finish(w, HYPO_ST_OK, 0); return RES_OK; }
        if (ne < 2) return emit_draft(w, P->draft4 + W.draft_off, (int)W.draft_len);
        if (W.type != HYPO_WIN_SHORT) {
            if (Cfg::PATHCAP == 0) return RES_UNSUPPORTED;   // re-queued to a class that keeps sequence paths
            return run_long(w, W);
        }
        const int rc = run_short(w, W, carry_in);
        // a window that came with a spill and overflows where nothing newer can be saved (its graph does not fit this class either,
        // an in-edge list ran full halfway through an update) keeps the spill it came with: the graph after the sequences before
        // that spill's cursor is as valid as it was
        if (g.lane == 0) stat[ST_CPASS] = (rc == RES_OVERFLOW && carry_in && stat[ST_CKIND] == CARRY_NONE) ? 1u : 0u;
        g.sync();
        return rc;
    }
};

}  // namespace hypo
