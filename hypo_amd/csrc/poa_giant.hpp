// poa_giant.hpp — size class 6 of the POA kernel: the windows nothing else holds.
//
// Classes 0-5 (poa_core.hpp) keep a window's state in fixed tables: sequences of up to 1 021 bases, 16 382 of them, 32 767 nodes,
// 58 in-edges per node.  The reference has no such bound — spoa's graph is vectors of heap nodes and its score matrix is
// re-allocated to whatever (nodes + 1) x (length + 1) asks for (external/spoa/src/graph.cpp:99-128,
// sisd_alignment_engine.cpp:60-93) — so a window with a 1 500-base insertion in a long read, or the 20 000 reads of a collapsed
// repeat, used to come back with its draft and HYPO_ST_CAPACITY.  Such a window now ends up here: one wavefront, the window's whole
// state in a slice of HBM (PoaAux::giant_arena), and the reference's procedure carried out literally —
//   engine  SisdAlignmentEngine::linear, kNW / kLOV / kROV, the FULL int32 score matrix as the reference keeps it
//           (sisd_alignment_engine.cpp:95-439): rows in rank order with the 64 lanes over the columns (the horizontal term as a
//           max-plus prefix scan per strip of 64 columns, carried from strip to strip), end row and traceback by its rules (:279-288,
//           :338-339, :344-438: first in-edge whose diagonal, then whose vertical source explains the cell, then the horizontal step);
//   graph   Graph::add_alignment / add_edge / add_sequence (graph.cpp:93-128,154-291) with in-edge, out-edge and aligned-node lists in
//           creation order, Graph::topological_sort (:293-353, the literal DFS; skipped when the alignment added neither a node nor an
//           edge: the order is a function of the structure alone), traverse_heaviest_bundle + branch_completion (:610-705),
//           generate_consensus_custom's coverage summary (:533-568, :371-388) with the sequence labels of the edges (:30-41);
//   window  Window::generate_consensus_short / _long + curate (src/Window.cpp:87-254, include/Window.hpp:30-33,144).
// Everything that is not a score row runs on lane 0 (dependent loads from HBM: slow, and rare by construction); lanes meet at
// Grp::sync().  Capacity is what the slice holds: sequences of up to 65 535 bases, node / edge tables and the score matrix sized
// from the slice (hypo_gpu_set_option("giant_arena_mb")); a window that outgrows even that keeps HYPO_ST_CAPACITY.
// The same code runs in the test emulator (tests/emu, HYPO_EMU) against the oracle and the real reference.
#pragma once
#include "poa_core.hpp"
#include <math.h>

namespace hypo {

struct GiantStats { uint64_t cells, aligns; };

template <class G>
struct Giant {
    static constexpr int GNEG = (int)0x80000000 + 1024;     // kNegativeInfinity (sisd_alignment_engine.cpp:21)
    static constexpr int ID = (int)0x80000000;              // identity of the lane scans
    static constexpr int AL = 6;                            // aligned nodes of a node: at most the other letters of its column (A C G T N J O)
    static constexpr uint32_t LMAXG = 65535 + 2;            // longest sequence (markers included)

    // group-shared scalars (in the slice; written by lane 0, read by everybody behind a sync)
    struct Shared {
        int n_nodes, n_edges, n_labels, nseq, rank_n, cons_n, aln_n, changed, rc, L, max_i;
        int keep_labels, conslen, stack_cap;
    };

    const G& g;
    const PoaParamRef& P;
    // slice layout
    Shared* sh;
    uint8_t *code, *mark, *chk, *nal, *sq, *ctext, *ctext2;
    int *in_head, *in_tail, *out_head, *out_tail, *al, *rank, *n2r, *pred, *cons, *msa, *stack, *seqbeg, *alnN, *alnP;
    long long* score;
    uint32_t* dst;
    int *e_src, *e_dst, *e_w, *e_nin, *e_nout, *e_lab_head, *e_lab_tail, *lab_seq, *lab_next;
    int* H;
    uint32_t Ncap, Ecap, Pcap, Scap;
    uint64_t Hcap;
    uint64_t cells, aligns;
#if defined(HYPO_PHASE_TIMERS) && !defined(HYPO_EMU)
    uint64_t t_rows = 0, t_trace = 0, t_add = 0, t_cons = 0, t_stage = 0;      // diagnostic build: cycles per part of a window (printed by Giant::run)
#define HYPO_GT(var, t0) do { var += (uint64_t)clock64() - (t0); } while (0)
#define HYPO_GT0() ((uint64_t)clock64())
#else
#define HYPO_GT(var, t0) do { } while (0)
#define HYPO_GT0() 0ull
#endif

    HD Giant(const G& g_, const PoaParamRef& P_) : g(g_), P(P_), cells(0), aligns(0) {}

    // ---- slice layout ---------------------------------------------------------------------------------------------------------
    // total_pos: positions of all sequences of the window (markers included), lmax: the longest of them, nseq_max: how many
    HD bool layout(char* slice, uint64_t bytes, uint64_t total_pos, uint32_t lmax, uint32_t nseq_max, bool labels) {
        if (lmax > LMAXG) return false;
        // the graph gets at most half of the slice: nodes and edges are bounded by the positions the window has
        uint64_t per_node = 1 + 1 + 1 + 1 + 4 * 4 + 4 * AL + 4 * 7 + 8 + 4 + 2 + 4 * (AL + 1);      // code mark chk nal | heads / tails | al | rank n2r pred cons msa alnN alnP | score | dst | ctext x 2 | DFS stack (a node is pushed once per in-edge and aligned node that names it, and as a root)
        uint64_t per_edge = 4 * 7 + 4;                                                   // src dst w nin nout lab_head lab_tail | stack (1 per edge)
        uint64_t n = total_pos + 8;
        const uint64_t graph_budget = bytes / 2;
        const uint64_t fixed = 4096 + (uint64_t)lmax + 64 + 4ull * nseq_max + (labels ? 8ull * (total_pos + 8) : 0) + 8ull * lmax;
        if (fixed >= graph_budget) return false;
        const uint64_t fit = (graph_budget - fixed) / (per_node + per_edge);
        if (fit < 64) return false;
        if (n > fit) n = fit;
        if (n < 64) n = 64;
        Ncap = Ecap = (uint32_t)(n > 0x3fffffffu ? 0x3fffffffu : n);
        Pcap = labels ? (uint32_t)(total_pos + 8 > 0x7fffffffu ? 0x7fffffffu : total_pos + 8) : 0u;
        Scap = (AL + 1) * Ncap + Ecap + 8;
        uint64_t off = 0;
        auto take = [&](uint64_t b) -> char* { char* p = slice + off; off += (b + 15) / 16 * 16; return p; };
        sh = (Shared*)take(sizeof(Shared));
        code = (uint8_t*)take(Ncap); mark = (uint8_t*)take(Ncap); chk = (uint8_t*)take(Ncap); nal = (uint8_t*)take(Ncap);
        in_head = (int*)take(4ull * Ncap); in_tail = (int*)take(4ull * Ncap); out_head = (int*)take(4ull * Ncap); out_tail = (int*)take(4ull * Ncap);
        al = (int*)take(4ull * AL * Ncap);
        rank = (int*)take(4ull * Ncap); n2r = (int*)take(4ull * Ncap); pred = (int*)take(4ull * Ncap); cons = (int*)take(4ull * Ncap); msa = (int*)take(4ull * Ncap);
        alnN = (int*)take(4ull * (Ncap + lmax + 8)); alnP = (int*)take(4ull * (Ncap + lmax + 8));
        score = (long long*)take(8ull * Ncap); dst = (uint32_t*)take(4ull * Ncap);
        ctext = (uint8_t*)take(Ncap); ctext2 = (uint8_t*)take(Ncap);
        stack = (int*)take(4ull * Scap);
        e_src = (int*)take(4ull * Ecap); e_dst = (int*)take(4ull * Ecap); e_w = (int*)take(4ull * Ecap); e_nin = (int*)take(4ull * Ecap); e_nout = (int*)take(4ull * Ecap);
        e_lab_head = (int*)take(4ull * Ecap); e_lab_tail = (int*)take(4ull * Ecap);
        lab_seq = (int*)take(4ull * Pcap); lab_next = (int*)take(4ull * Pcap);
        seqbeg = (int*)take(4ull * nseq_max + 16);
        sq = (uint8_t*)take((uint64_t)lmax + 64);
        if (off + 4096 > bytes) return false;
        H = (int*)(slice + off);
        Hcap = (bytes - off) / 4;
        return true;
    }

    // ---- graph (lane 0) ---------------------------------------------------------------------------------------------------------
    HD void g_reset(bool keep_labels) { sh->n_nodes = 0; sh->n_edges = 0; sh->n_labels = 0; sh->nseq = 0; sh->rank_n = 0; sh->cons_n = 0; sh->keep_labels = keep_labels ? 1 : 0; }
    HD int g_add_node(int c) {                                   // graph.cpp:93-97; -1: the tables are full
        const int id = sh->n_nodes;
        if ((uint32_t)id >= Ncap) return -1;
        sh->n_nodes = id + 1;
        code[id] = (uint8_t)c; in_head[id] = in_tail[id] = out_head[id] = out_tail[id] = -1; nal[id] = 0;
        sh->changed = 1;
        return id;
    }
    HD bool g_label(int ed) {
        if (!sh->keep_labels) return true;
        const int l = sh->n_labels;
        if ((uint32_t)l >= Pcap) return false;
        sh->n_labels = l + 1;
        lab_seq[l] = sh->nseq; lab_next[l] = -1;
        if (e_lab_tail[ed] < 0) e_lab_head[ed] = l; else lab_next[e_lab_tail[ed]] = l;
        e_lab_tail[ed] = l;
        return true;
    }
    HD bool g_add_edge(int b, int e, int w) {                    // graph.cpp:99-115: search the begin node's out-edges, else append to both lists
        for (int ed = out_head[b]; ed >= 0; ed = e_nout[ed])
            if (e_dst[ed] == e) { e_w[ed] += w; return g_label(ed); }
        const int ed = sh->n_edges;
        if ((uint32_t)ed >= Ecap) return false;
        sh->n_edges = ed + 1;
        e_src[ed] = b; e_dst[ed] = e; e_w[ed] = w; e_nin[ed] = -1; e_nout[ed] = -1; e_lab_head[ed] = e_lab_tail[ed] = -1;
        if (out_tail[b] < 0) out_head[b] = ed; else e_nout[out_tail[b]] = ed;
        out_tail[b] = ed;
        if (in_tail[e] < 0) in_head[e] = ed; else e_nin[in_tail[e]] = ed;
        in_tail[e] = ed;
        sh->changed = 1;
        return g_label(ed);
    }
    // graph.cpp:273-291; first node id, -1 when the range is empty, -2 when the tables are full
    HD int g_add_sequence(int begin, int end) {
        if (begin == end) return -1;
        const int first = g_add_node(sq[begin]);
        if (first < 0) return -2;
        for (int i = begin + 1; i < end; ++i) {
            const int id = g_add_node(sq[i]);
            if (id < 0 || !g_add_edge(id - 1, id, 2)) return -2;
        }
        return first;
    }
    // graph.cpp:293-353
    HD int g_toposort() {
        const int n = sh->n_nodes;
        int rn = 0;
        for (int i = 0; i < n; ++i) { mark[i] = 0; chk[i] = 1; }
        int sp = 0;
        for (int i = 0; i < n; ++i) {
            if (mark[i] != 0) continue;
            stack[sp++] = i;
            while (sp != 0) {
                const int v = stack[sp - 1];
                bool valid = true;
                if (mark[v] != 2) {
                    for (int ed = in_head[v]; ed >= 0; ed = e_nin[ed]) {
                        const int b = e_src[ed];
                        if (mark[b] != 2) { if ((uint32_t)sp >= Scap) return RES_OVERFLOW; stack[sp++] = b; valid = false; }
                    }
                    if (chk[v]) {
                        for (int k = 0; k < (int)nal[v]; ++k) {
                            const int a = al[v * AL + k];
                            if (mark[a] != 2) { if ((uint32_t)sp >= Scap) return RES_OVERFLOW; stack[sp++] = a; chk[a] = 0; valid = false; }
                        }
                    }
                    if (valid) {
                        mark[v] = 2;
                        if (chk[v]) {
                            rank[rn++] = v;
                            for (int k = 0; k < (int)nal[v]; ++k) rank[rn++] = al[v * AL + k];
                        }
                    } else mark[v] = 1;
                }
                if (valid) --sp;
            }
        }
        sh->rank_n = rn;
        return RES_OK;
    }
    // graph.cpp:154-271 (the alignment is alnN / alnP[0 .. aln_n), the sequence sq[0 .. L))
    HD int g_add_alignment(int L) {
        if (L == 0) return RES_OK;
        sh->changed = 0;
        const int an = sh->aln_n;
        if (an == 0) {
            const int b = g_add_sequence(0, L);
            if (b == -2) return RES_OVERFLOW;
            seqbeg[sh->nseq] = b; sh->nseq += 1;
            return g_toposort();
        }
        int first_valid = -1, last_valid = -1;
        for (int i = 0; i < an; ++i) if (alnP[i] != -1) { if (first_valid < 0) first_valid = alnP[i]; last_valid = alnP[i]; }
        if (first_valid < 0) return RES_UNDEFINED;               // valid_seq_ids.front() of an empty vector in the reference
        const int before = sh->n_nodes;
        int begin_node = g_add_sequence(0, first_valid);
        if (begin_node == -2) return RES_OVERFLOW;
        int head = before == sh->n_nodes ? -1 : sh->n_nodes - 1;
        const int tail = g_add_sequence(last_valid + 1, L);
        if (tail == -2) return RES_OVERFLOW;
        int cur = -1;
        for (int i = 0; i < an; ++i) {
            const int pos = alnP[i], nd = alnN[i];
            if (pos == -1) continue;
            const int c = sq[pos];
            if (nd == -1) { cur = g_add_node(c); if (cur < 0) return RES_OVERFLOW; }
            else if ((int)code[nd] == c) cur = nd;
            else {
                int found = -1;
                for (int k = 0; k < (int)nal[nd]; ++k) if ((int)code[al[nd * AL + k]] == c) { found = al[nd * AL + k]; break; }
                if (found == -1) {
                    cur = g_add_node(c);
                    if (cur < 0) return RES_OVERFLOW;
                    const int cnt = nal[nd];                     // nd's list before it learns about cur
                    if (cnt + 1 > AL) return RES_OVERFLOW;
                    for (int k = 0; k < cnt; ++k) {
                        const int a = al[nd * AL + k];
                        al[cur * AL + (int)nal[cur]] = a; nal[cur] += 1;
                        if ((int)nal[a] >= AL) return RES_OVERFLOW;
                        al[a * AL + (int)nal[a]] = cur; nal[a] += 1;
                    }
                    al[cur * AL + (int)nal[cur]] = nd; nal[cur] += 1;
                    al[nd * AL + (int)nal[nd]] = cur; nal[nd] += 1;
                } else cur = found;
            }
            if (begin_node == -1) begin_node = cur;
            if (head != -1 && !g_add_edge(head, cur, 2)) return RES_OVERFLOW;
            head = cur;
        }
        if (tail != -1 && !g_add_edge(head, tail, 2)) return RES_OVERFLOW;
        seqbeg[sh->nseq] = begin_node; sh->nseq += 1;
        // (the order is a function of nodes, in-edge lists and aligned lists: an alignment that only raised weights leaves it what it is)
        return sh->changed ? g_toposort() : (int)RES_OK;
    }
    // graph.cpp:660-705
    HD int g_branch_completion(int rk) {
        const int v = rank[rk], rn = sh->rank_n;
        for (int ed = out_head[v]; ed >= 0; ed = e_nout[ed]) {
            const int t = e_dst[ed];
            for (int q = in_head[t]; q >= 0; q = e_nin[q]) { const int b = e_src[q]; if (b != v) score[b] = -1; }
        }
        long long max_score = 0; int max_id = 0;
        for (int i = rk + 1; i < rn; ++i) {
            const int u = rank[i];
            score[u] = -1; pred[u] = -1;
            for (int q = in_head[u]; q >= 0; q = e_nin[q]) {
                const int b = e_src[q];
                if (score[b] == -1) continue;
                if (score[u] < (long long)e_w[q] || (score[u] == (long long)e_w[q] && score[pred[u]] <= score[b])) { score[u] = e_w[q]; pred[u] = b; }
            }
            if (pred[u] != -1) score[u] += score[pred[u]];
            if (max_score < score[u]) { max_score = score[u]; max_id = u; }
        }
        return max_id;
    }
    // graph.cpp:610-658, :467-476: consensus node ids in cons[], their letters in ctext[]
    HD int g_consensus() {
        const int n = sh->n_nodes, rn = sh->rank_n;
        for (int i = 0; i < n; ++i) { pred[i] = -1; score[i] = -1; }
        int max_id = 0;
        for (int r = 0; r < rn; ++r) {
            const int u = rank[r];
            for (int q = in_head[u]; q >= 0; q = e_nin[q]) {
                const int b = e_src[q];
                if (score[u] < (long long)e_w[q] || (score[u] == (long long)e_w[q] && score[pred[u]] <= score[b])) { score[u] = e_w[q]; pred[u] = b; }
            }
            if (pred[u] != -1) score[u] += score[pred[u]];
            if (score[max_id] < score[u]) max_id = u;
        }
        if (out_head[max_id] >= 0) {
            for (int i = 0; i < rn; ++i) n2r[rank[i]] = i;
            while (out_head[max_id] >= 0) max_id = g_branch_completion(n2r[max_id]);
        }
        int cn = 0;
        while (pred[max_id] != -1) { cons[cn++] = max_id; max_id = pred[max_id]; }
        cons[cn++] = max_id;
        for (int a = 0, b = cn - 1; a < b; ++a, --b) { const int t = cons[a]; cons[a] = cons[b]; cons[b] = t; }
        for (int i = 0; i < cn; ++i) ctext[i] = code[cons[i]];
        sh->cons_n = cn;
        return cn;
    }
    HD int g_successor(int v, int label) const {                 // graph.cpp:30-41
        for (int ed = out_head[v]; ed >= 0; ed = e_nout[ed])
            for (int l = e_lab_head[ed]; l >= 0; l = lab_next[l]) if (lab_seq[l] == label) return e_dst[ed];
        return -1;
    }
    // graph.cpp:533-568 + :371-388: consensus + how many sequences carry each of its bases (dst[])
    HD int g_consensus_custom() {
        const int n = g_consensus();
        const int nn = sh->n_nodes;
        int msa_id = 0;
        for (int i = 0; i < nn; ++i) {
            const int u = rank[i];
            msa[u] = msa_id;
            for (int k = 0; k < (int)nal[u]; ++k) msa[rank[++i]] = msa_id;
            ++msa_id;
        }
        for (int i = 0; i < n; ++i) dst[i] = 0;
        for (int s = 0; s < sh->nseq; ++s) {
            int v = seqbeg[s], k = 0;
            for (;;) {
                while (k < n && msa[cons[k]] < msa[v]) ++k;
                if (k >= n) break;
                if (msa[cons[k]] == msa[v] && code[v] == ctext[k]) dst[k] += 1;
                const int nx = g_successor(v, s);
                if (nx < 0) break;
                v = nx;
            }
        }
        return n;
    }

    // ---- engine: all lanes -------------------------------------------------------------------------------------------------------
    // SisdAlignmentEngine::linear (sisd_alignment_engine.cpp:263-439) of sq[0 .. L) against the graph; the alignment goes to alnN / alnP
    HD int align(int L, int mode, int m, int n_, int gp) {
        if (g.lane == 0) sh->aln_n = 0;
        g.sync();
        const int nn = sh->n_nodes;
        if (nn == 0 || L == 0) return RES_OK;                    // :249-251
        const int W = L + 1;
        if ((uint64_t)(nn + 1) * (uint64_t)W > Hcap) return RES_OVERFLOW;
        cells += (uint64_t)(nn + 1) * (uint64_t)W; aligns += 1;
        for (int r = g.lane; r < nn; r += 64) n2r[rank[r]] = r;
        for (int j = g.lane; j < W; j += 64) H[j] = j * gp;      // row 0 (:197-199,230-232)
        g.sync();
        const bool native_lov = mode == MODE_LOV && (P->flags & POA_NATIVE_KLOV) != 0;
        int max_score = GNEG, max_i = -1;
        const uint64_t tr0 = HYPO_GT0(); (void)tr0;
        for (int r = 0; r < nn; ++r) {
            const int u = rank[r], i = r + 1;
            const int c = code[u];
            int* const row = H + (size_t)i * W;
            // the rows of the predecessors (in-edge order; a node without one: the virtual row 0, :296-298): the first PC of them by offset, a
            // node with more (rare) walks the rest of its list per strip
            constexpr int PC = 8;
            size_t poff[PC];
            int np = 0;
            for (int ed = in_head[u]; ed >= 0; ed = e_nin[ed]) { if (np < PC) poff[np] = (size_t)(n2r[e_src[ed]] + 1) * W; ++np; }
            HYPO_UNROLL
            for (int k = 0; k < PC; ++k) if (k >= np) poff[k] = 0;
            const int npc = np == 0 ? 1 : (np < PC ? np : PC);
            // column 0 (:163-243): kNW / kLOV: the best predecessor's + g (a source: 0 + g); kROV: 0
            int h0;
            if (mode == MODE_ROV) h0 = 0;
            else {
                int pen = np == 0 ? 0 : GNEG;
                for (int ed = in_head[u]; ed >= 0; ed = e_nin[ed]) { const int pv = H[(size_t)(n2r[e_src[ed]] + 1) * W]; pen = pv > pen ? pv : pen; }
                h0 = pen + gp;
            }
            if (g.lane == 0) row[0] = h0;
            int carry = h0;                                      // max over the columns so far of H[i][j] - j * g
            int row_max = ID;
            // CH strips of 64 columns at a time: the vertical / diagonal maxima of a chunk are loads that depend on nothing in this row, so the
            // NEXT chunk's are issued before this chunk's horizontal pass (strip by strip: H[i][j] = max(H[i][j], H[i][j-1] + g) <=> prefix
            // maximum of H[i][j] - j * g, :321-329) — the matrix is far larger than any cache and a load from it takes as long as a dozen scans
            constexpr int CH = 16;
            auto load_chunk = [&](int b0, int* x) {
                HYPO_UNROLL
                for (int q = 0; q < CH; ++q) {
                    const int j = b0 + 64 * q + g.lane;
                    x[q] = ID;
                    if (j < W) {
                        const int sc = (int)sq[j - 1] == c ? m : n_;
                        int v = ID;
                        HYPO_UNROLL
                        for (int k = 0; k < PC; ++k) if (k < npc) {
                            const int* const prow = H + poff[k];
                            const int a = prow[j - 1] + sc, d = prow[j] + gp;
                            const int t = a > d ? a : d;
                            v = t > v ? t : v;
                        }
                        if (np > PC) {
                            int k = 0;
                            for (int ed = in_head[u]; ed >= 0; ed = e_nin[ed], ++k) {
                                if (k < PC) continue;
                                const int* const prow = H + (size_t)(n2r[e_src[ed]] + 1) * W;
                                const int a = prow[j - 1] + sc, d = prow[j] + gp;
                                const int t = a > d ? a : d;
                                v = t > v ? t : v;
                            }
                        }
                        x[q] = v - j * gp;
                    }
                }
            };
            int xa[CH], xb[CH];
            load_chunk(1, xa);
            for (int b0 = 1; b0 < W; b0 += 64 * CH) {
                const bool more = b0 + 64 * CH < W;              // (group-uniform)
                if (more) load_chunk(b0 + 64 * CH, xb);
                HYPO_UNROLL
                for (int q = 0; q < CH; ++q) {
                    if (b0 + 64 * q < W) {                       // (group-uniform)
                        const int j = b0 + 64 * q + g.lane;
                        const int ex = g.scan_max_excl(xa[q], ID);
                        int inc = xa[q] > ex ? xa[q] : ex;
                        inc = inc > carry ? inc : carry;
                        if (j < W) { const int h = inc + j * gp; row[j] = h; if (native_lov) row_max = h > row_max ? h : row_max; }
                        carry = g.shfl(inc, 63);
                    }
                }
                if (more) { HYPO_UNROLL for (int q = 0; q < CH; ++q) xa[q] = xb[q]; }
            }
            g.sync();
            bool is_end = mode == MODE_LOV;                      // :338-339
            if (!is_end) is_end = out_head[u] < 0;               // :332-334
            if (is_end) {
                int endval = row[W - 1];
                if (native_lov) { const int rm = g.reduce_max(row_max); endval = rm > endval ? rm : endval; }
                if (max_score < endval) { max_score = endval; max_i = i; }
            } else if (native_lov) (void)g.reduce_max(row_max);
        }
        HYPO_GT(t_rows, tr0);
        const uint64_t tt0 = HYPO_GT0(); (void)tt0;
        // traceback (:344-438), lane 0
        if (g.lane == 0) {
            int i = max_i > 0 ? max_i : 0, j = max_i > 0 ? W - 1 : 0;
            int prev_i = 0, prev_j = 0, an = 0;
            for (;;) {
                if (mode == MODE_ROV) { if (i == 0 || j == 0) break; }
                else { if (i == 0 && j == 0) break; }
                const int Hij = H[(size_t)i * W + j];
                bool found = false;
                if (i != 0 && j != 0) {
                    const int u = rank[i - 1];
                    const int mc = (int)code[u] == (int)sq[j - 1] ? m : n_;
                    if (in_head[u] < 0) { if (Hij == H[j - 1] + mc) { prev_i = 0; prev_j = j - 1; found = true; } }
                    else for (int ed = in_head[u]; ed >= 0; ed = e_nin[ed]) {
                        const int pi = n2r[e_src[ed]] + 1;
                        if (Hij == H[(size_t)pi * W + j - 1] + mc) { prev_i = pi; prev_j = j - 1; found = true; break; }
                    }
                }
                if (!found && i != 0) {
                    const int u = rank[i - 1];
                    if (in_head[u] < 0) { if (Hij == H[j] + gp) { prev_i = 0; prev_j = j; found = true; } }
                    else for (int ed = in_head[u]; ed >= 0; ed = e_nin[ed]) {
                        const int pi = n2r[e_src[ed]] + 1;
                        if (Hij == H[(size_t)pi * W + j] + gp) { prev_i = pi; prev_j = j; found = true; break; }
                    }
                }
                if (!found && j != 0 && Hij == H[(size_t)i * W + j - 1] + gp) { prev_i = i; prev_j = j - 1; found = true; }
                alnN[an] = i == prev_i ? -1 : rank[i - 1];
                alnP[an] = j == prev_j ? -1 : j - 1;
                ++an;
                if (!found && i == prev_i && j == prev_j) break;   // (cannot happen for a consistent matrix)
                i = prev_i; j = prev_j;
            }
            for (int a = 0, b = an - 1; a < b; ++a, --b) { int t = alnN[a]; alnN[a] = alnN[b]; alnN[b] = t; t = alnP[a]; alnP[a] = alnP[b]; alnP[b] = t; }
            sh->aln_n = an;
        }
        g.sync();
        HYPO_GT(t_trace, tt0);
        return RES_OK;
    }

    // ---- sequences --------------------------------------------------------------------------------------------------------------
    HD int draft_code(const HypoWindow& W, uint32_t i) const { const uint32_t c = (P->draft4[W.draft_off + (i >> 1)] >> (4 - 4 * (i & 1))) & 15u; return c < 4u ? (int)c : (int)C_N; }
    // sq[] = [head marker] + the sequence + [tail marker]; which: -1 the draft, -2 the consensus of the round before (ctext2), else the arm
    HD int stage(const HypoWindow& W, int which, bool head, bool tail, int prev_len) {
        int len;
        const int o = head ? 1 : 0;
        if (which == -1) { len = (int)W.draft_len; for (int i = g.lane; i < len; i += 64) sq[o + i] = (uint8_t)draft_code(W, (uint32_t)i); }
        else if (which == -2) { len = prev_len; for (int i = g.lane; i < len; i += 64) sq[o + i] = ctext2[i]; }
        else {
            const uint64_t a = (uint64_t)W.first_arm + (uint64_t)which;
            len = (int)P->arm_len[a];
            const uint8_t* const src = P->arms2 + P->arm_off[a];
            for (int i = g.lane; i < len; i += 64) sq[o + i] = (uint8_t)((src[i >> 2] >> (6 - 2 * (i & 3))) & 3u);
        }
        if (g.lane == 0) { if (head) sq[0] = (uint8_t)C_J; if (tail) sq[o + len] = (uint8_t)C_O; }
        g.sync();
        return len + o + (tail ? 1 : 0);
    }
    // engine->align + graph->add_alignment
    HD int add(int L, int mode, int m, int n_, int gp) {
        int rc = align(L, mode, m, n_, gp);
        if (rc != RES_OK) return rc;
        g.sync();                                               // every lane has read what it needs of the graph before lane 0 changes it
        const uint64_t ta0 = HYPO_GT0(); (void)ta0;
        if (g.lane == 0) sh->rc = g_add_alignment(L);
        g.sync();
        HYPO_GT(t_add, ta0);
        return sh->rc;
    }

    HD static char letter(int c) { return c < 4 ? "ACGT"[c] : (c == C_N ? 'N' : (c == C_J ? 'J' : 'O')); }

    // Window::generate_consensus for window w: RES_OK (answered: status + length written), RES_OVERFLOW (the slice is too small), RES_INVALID
    HD int run(uint32_t w, char* slice, uint64_t slice_bytes) {
        const HypoWindow W = P->windows[w];
        const uint64_t narm64 = (uint64_t)W.n_internal + W.n_prefix + W.n_suffix;
        if (narm64 + W.first_arm > P->n_arms || W.draft_off + (W.draft_len + 1) / 2 > P->draft4_bytes) return RES_INVALID;
        const uint32_t narm = (uint32_t)narm64;
        const uint64_t cap = P->out_off[w + 1] - P->out_off[w];
        char* const out = P->out_bases + P->out_off[w];
        auto answer_draft = [&]() {
            if ((uint64_t)W.draft_len > cap) { if (g.lane == 0) { P->out_len[w] = W.draft_len; P->out_status[w] = HYPO_ST_CONS_OVERFLOW; } return; }
            for (uint32_t i = (uint32_t)g.lane; i < W.draft_len; i += 64) out[i] = letter(draft_code(W, i));
            if (g.lane == 0) { P->out_len[w] = W.draft_len; P->out_status[w] = HYPO_ST_OK; }
        };
        if (W.n_empty > narm) { if (g.lane == 0) { P->out_len[w] = 0; P->out_status[w] = HYPO_ST_OK; } return RES_OK; }     // src/Window.cpp:47-49
        if (narm < 2) { answer_draft(); return RES_OK; }                                                                       // :59-61
        const bool is_long = W.type != HYPO_WIN_SHORT;
        // sizes: every position the window can put into a graph, the longest sequence
        uint64_t total = (uint64_t)W.draft_len + 2;
        uint32_t lmax = W.draft_len + 2;
        {
            uint64_t t = 0; uint32_t lm = 0;
            for (uint32_t a = (uint32_t)g.lane; a < narm; a += 64) {
                const uint64_t ai = (uint64_t)W.first_arm + a;
                const uint32_t l = P->arm_len[ai];
                if (P->arm_off[ai] > P->arms2_bytes || (uint64_t)(l + 3) / 4 > P->arms2_bytes - P->arm_off[ai]) lm = 0xffffffffu;
                t += (uint64_t)l + 2; lm = (l + 2 > lm && lm != 0xffffffffu) ? l + 2 : lm;
            }
            // (sums and maxima over the lanes; the totals stay far below 2^31 for anything a slice can hold, larger ones fail the layout)
            const int bad = g.reduce_max(lm == 0xffffffffu ? 1 : 0);
            if (bad) return RES_INVALID;
            const int lm_all = g.reduce_max((int)lm);
            uint64_t hi = t >> 24, lo = t & 0xffffffu;
            const uint64_t sum = ((uint64_t)(uint32_t)g.reduce_add((int)hi) << 24) + (uint64_t)(uint32_t)g.reduce_add((int)lo);
            total += sum; lmax = (uint32_t)lm_all > lmax ? (uint32_t)lm_all : lmax;
        }
        if (!layout(slice, slice_bytes, total, lmax, narm + 2, is_long)) return RES_OVERFLOW;
        const uint32_t ni = W.n_internal, np = W.n_prefix, ns = W.n_suffix;
        int rc;
        if (!is_long) {                                                      // src/Window.cpp:87-154
            const int m = P->sr_m, n_ = P->sr_n, gp = P->sr_g;
            if (g.lane == 0) g_reset(false);
            g.sync();
            bool added = false;
            if (ni == 0) { const int L = stage(W, -1, true, true, 0); if ((rc = add(L, MODE_NW, m, n_, gp)) != RES_OK) return rc; }
            for (uint32_t i = 0; i < ni; ++i) if (P->arm_len[(uint64_t)W.first_arm + i] > 0) {
                const int L = stage(W, (int)i, true, true, 0); added = true;
                if ((rc = add(L, MODE_NW, m, n_, gp)) != RES_OK) return rc;
            }
            for (uint32_t i = np; i-- > 0;) if (P->arm_len[(uint64_t)W.first_arm + ni + i] > 0) {      // last one first (:111)
                const int L = stage(W, (int)(ni + i), true, false, 0); added = true;
                if ((rc = add(L, MODE_LOV, m, n_, gp)) != RES_OK) return rc;
            }
            for (uint32_t i = 0; i < ns; ++i) if (P->arm_len[(uint64_t)W.first_arm + ni + np + i] > 0) {
                const int L = stage(W, (int)(ni + np + i), false, true, 0); added = true;
                if ((rc = add(L, MODE_ROV, m, n_, gp)) != RES_OK) return rc;
            }
            if (!added) { answer_draft(); return RES_OK; }
            if (g.lane == 0) sh->conslen = g_consensus();
            g.sync();
            const int len = sh->conslen;
            if (len < 2) return RES_UNDEFINED;                               // include/Window.hpp:144 would be undefined
            const int olen = len - 2;
            if ((uint64_t)olen > cap) { if (g.lane == 0) { P->out_len[w] = (uint32_t)olen; P->out_status[w] = HYPO_ST_CONS_OVERFLOW; } return RES_OK; }
            for (int i = g.lane; i < olen; i += 64) out[i] = letter(ctext[1 + i]);
            if (g.lane == 0) { P->out_len[w] = (uint32_t)olen; P->out_status[w] = HYPO_ST_OK; }
            return RES_OK;
        }
        // LONG (src/Window.cpp:156-254): two rounds, every alignment kNW with the long-read scores, the consensus curated by coverage
        const int m = P->lr_m, n_ = P->lr_n, gp = P->lr_g;
        int conslen = 0;
        for (int round = 0; round < 2; ++round) {
            if (g.lane == 0) g_reset(true);
            g.sync();
            bool added = false;
            if (round == 0) { const int L = stage(W, -1, false, false, 0); if ((rc = add(L, MODE_NW, m, n_, gp)) != RES_OK) return rc; }
            else if (conslen > 0) { const int L = stage(W, -2, false, false, conslen); if ((rc = add(L, MODE_NW, m, n_, gp)) != RES_OK) return rc; }
            for (uint32_t a = 0; a < narm; ++a) if (P->arm_len[(uint64_t)W.first_arm + a] > 0) {
                const int L = stage(W, (int)a, false, false, 0); added = true;
                if ((rc = add(L, MODE_NW, m, n_, gp)) != RES_OK) return rc;
            }
            if (!added) { answer_draft(); return RES_OK; }
            const uint64_t tc0 = HYPO_GT0(); (void)tc0;
            if (g.lane == 0) {
                const int len = g_consensus_custom();
                const uint32_t thr = (uint32_t)floorf((float)ni * 0.4f);     // src/Window.cpp:28,245
                int o = 0;
                for (int i = 0; i < len; ++i) if (dst[i] >= thr) ctext2[o++] = ctext[i];
                sh->conslen = o;
            }
            g.sync();
            HYPO_GT(t_cons, tc0);
            conslen = sh->conslen;
        }
#if defined(HYPO_PHASE_TIMERS) && !defined(HYPO_EMU)
        if (g.lane == 0) printf("[giant] window %u: nodes %d, rows %llu Mcycles, traceback %llu, add_alignment + sort %llu, consensus %llu\n", w, sh->n_nodes, (unsigned long long)(t_rows >> 20), (unsigned long long)(t_trace >> 20), (unsigned long long)(t_add >> 20), (unsigned long long)(t_cons >> 20));
#endif
        if ((uint64_t)conslen > cap) { if (g.lane == 0) { P->out_len[w] = (uint32_t)conslen; P->out_status[w] = HYPO_ST_CONS_OVERFLOW; } return RES_OK; }
        for (int i = g.lane; i < conslen; i += 64) out[i] = letter(ctext2[i]);
        if (g.lane == 0) { P->out_len[w] = (uint32_t)conslen; P->out_status[w] = HYPO_ST_OK; }
        return RES_OK;
    }
};

}  // namespace hypo
