// poa_core.hpp — one window's partial-order-alignment consensus, executed by one lane group.
//
// MI355X-first restructuring of the reference's per-window POA (no reference code is reused):
//   reference                                              here
//   ---------------------------------------------------   ------------------------------------------
//   Window::generate_consensus      src/Window.cpp:44-61   Poa::run (dispatch rules)
//   generate_consensus_short        src/Window.cpp:87-154  Poa::run_short (sequence order, markers)
//   SisdAlignmentEngine::linear     sisd..cpp:263-439      Poa::align: rows in rank order, the CPL
//                                                          columns of a lane in registers, vertical and
//                                                          diagonal terms from predecessor rows in LDS,
//                                                          horizontal term = max-plus prefix scan over
//                                                          lanes (exact: integer max/+ is associative)
//   traceback                       sisd..cpp:344-438      lanes test the predecessors of a cell in
//                                                          parallel, ballot picks the reference's first hit
//   Graph::add_alignment            graph.cpp:154-271      lanes own sequence positions (every graph node
//                                                          and aligned clique occurs at most once on a
//                                                          path, so the updates are independent)
//   Graph::topological_sort         graph.cpp:293-353      literal DFS replay, dependencies of the stack
//                                                          top checked by lanes in parallel; skipped when
//                                                          an alignment added no node and no edge
//   traverse_heaviest_bundle        graph.cpp:610-705      lane 0 (once per window)
//
// All per-window state (score matrix, graph, order, stack) lives in the group's memory slice `mem`
// (LDS for the in-LDS size classes).  Compiled by hipcc for gfx950 and, with HYPO_EMU, by g++ for the
// lockstep emulator used in tests/.
#pragma once
#include "grp.hpp"
#include "../../include/hypo_gpu.h"

namespace hypo {

enum { MODE_NW = 1, MODE_LOV = 3, MODE_ROV = 4 };
enum { C_A = 0, C_C = 1, C_G = 2, C_T = 3, C_N = 4, C_J = 5, C_O = 6, C_NONE = 7 };
enum { RES_OK = 0, RES_OVERFLOW = 1, RES_UNDEFINED = 2, RES_CONS_OVERFLOW = 3, RES_UNSUPPORTED = 4 };

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
};

struct PoaCounters { uint64_t cells, aligns; };

template <int GW_, int CPL_, int NMAX_, int KIN_, int HCELLS_, class ScoreT, class IdT>
struct PoaCfg {
    static constexpr int GW = GW_;          // lanes per window
    static constexpr int CPL = CPL_;        // matrix columns per lane
    static constexpr int NMAX = NMAX_;      // graph nodes
    static constexpr int KIN = KIN_;        // in-edges per node
    static constexpr int HCELLS = HCELLS_;  // score-matrix capacity (cells)
    static constexpr int LMAX = GW_ * CPL_ - 1;   // longest sequence incl. markers
    static constexpr int AL = 6;            // aligned clique partners (alphabet ACGTNJO -> at most 6)
    static constexpr int STK = 2 * NMAX_;   // DFS stack
    typedef ScoreT score_t;
    typedef IdT id_t;
    static constexpr int ID_NONE = (IdT)~(IdT)0;
    static_assert(KIN_ + 6 <= GW_, "dependency lanes");
    static_assert((int)sizeof(ScoreT) * HCELLS_ >= 8 * NMAX_, "consensus scratch aliases the matrix");
    static_assert(NMAX_ < ID_NONE, "id range");
};

template <int N> HD constexpr int align16(int x) { return (x + N - 1) / N * N; }

template <class Cfg>
struct PoaLayout {   // byte offsets inside a group's memory slice
    typedef typename Cfg::score_t score_t;
    typedef typename Cfg::id_t id_t;
    static constexpr int oH = 0;
    static constexpr int oRowmeta = oH + align16<16>(Cfg::HCELLS * (int)sizeof(score_t));
    static constexpr int oInw = oRowmeta + align16<16>(Cfg::NMAX * 4);
    static constexpr int oPosnode = oInw + align16<16>(Cfg::NMAX * Cfg::KIN * 2);
    static constexpr int oCur = oPosnode + align16<16>((Cfg::LMAX + 1) * 2);
    static constexpr int oProw = oCur + align16<16>((Cfg::LMAX + 1) * 2);
    static constexpr int oInp = oProw + align16<16>(Cfg::NMAX * Cfg::KIN * (int)sizeof(id_t));
    static constexpr int oAl = oInp + align16<16>(Cfg::NMAX * Cfg::KIN * (int)sizeof(id_t));
    static constexpr int oR2n = oAl + align16<16>(Cfg::NMAX * Cfg::AL * (int)sizeof(id_t));
    static constexpr int oN2r = oR2n + align16<16>(Cfg::NMAX * (int)sizeof(id_t));
    static constexpr int oStack = oN2r + align16<16>(Cfg::NMAX * (int)sizeof(id_t));
    static constexpr int oCode = oStack + align16<16>(Cfg::STK * (int)sizeof(id_t));
    static constexpr int oNin = oCode + align16<16>(Cfg::NMAX);
    static constexpr int oNout = oNin + align16<16>(Cfg::NMAX);
    static constexpr int oNal = oNout + align16<16>(Cfg::NMAX);
    static constexpr int oMark = oNal + align16<16>(Cfg::NMAX);
    static constexpr int oSeq = oMark + align16<16>(Cfg::NMAX);
    static constexpr int BYTES = oSeq + align16<16>(Cfg::LMAX + 1);
};

template <class Cfg>
struct Poa {
    typedef typename Cfg::score_t score_t;
    typedef typename Cfg::id_t id_t;
    typedef PoaLayout<Cfg> Lay;
    static constexpr int GW = Cfg::GW, CPL = Cfg::CPL, KIN = Cfg::KIN, NMAX = Cfg::NMAX, AL = Cfg::AL;
    static constexpr int NEG = -(1 << 29);

    struct alignas(sizeof(score_t) * CPL) Pack { score_t v[CPL]; };

    const Grp<GW>& g;
    const PoaParams& P;
    // memory slice
    score_t* H; uint32_t* rowmeta; uint16_t* inw; int16_t* posnode; int16_t* cur;
    id_t *prow, *inp, *al, *r2n, *n2r, *stack;
    uint8_t *code, *nin, *nout, *nal, *mark, *seq;
    // group-uniform state
    int n_nodes; int L; bool topo_dirty; bool meta_dirty;
    int tb_steps; int tb_fv;
    uint64_t cells, aligns;

    HD Poa(const Grp<GW>& g_, const PoaParams& P_, char* mem) : g(g_), P(P_) {
        H = (score_t*)(mem + Lay::oH); rowmeta = (uint32_t*)(mem + Lay::oRowmeta);
        inw = (uint16_t*)(mem + Lay::oInw); posnode = (int16_t*)(mem + Lay::oPosnode);
        cur = (int16_t*)(mem + Lay::oCur); prow = (id_t*)(mem + Lay::oProw);
        inp = (id_t*)(mem + Lay::oInp); al = (id_t*)(mem + Lay::oAl);
        r2n = (id_t*)(mem + Lay::oR2n); n2r = (id_t*)(mem + Lay::oN2r);
        stack = (id_t*)(mem + Lay::oStack); code = (uint8_t*)(mem + Lay::oCode);
        nin = (uint8_t*)(mem + Lay::oNin); nout = (uint8_t*)(mem + Lay::oNout);
        nal = (uint8_t*)(mem + Lay::oNal); mark = (uint8_t*)(mem + Lay::oMark);
        seq = (uint8_t*)(mem + Lay::oSeq);
        n_nodes = 0; L = 0; topo_dirty = false; meta_dirty = true; tb_steps = 0; tb_fv = 0;
        cells = 0; aligns = 0;
    }

    // ---- sequence staging (PackedSeq<2>/<4> bytes -> codes in `seq`, markers J/O added) -----------
    HD int load_seq(const uint8_t* p, int len, bool four_bit, bool head, bool tail) {
        L = len + (head ? 1 : 0) + (tail ? 1 : 0);
        if (L > Cfg::LMAX) return RES_OVERFLOW;
        g.sync();
        for (int t = g.lane; t < L; t += GW) {
            int c;
            if (head && t == 0) c = C_J;
            else if (tail && t == L - 1) c = C_O;
            else {
                int b = t - (head ? 1 : 0);
                if (four_bit) { c = (p[b >> 1] >> (4 - 4 * (b & 1))) & 15; c = c < 4 ? c : C_N; }
                else c = (p[b >> 2] >> (6 - 2 * (b & 3))) & 3;
            }
            seq[t] = (uint8_t)c;
        }
        g.sync();
        return RES_OK;
    }

    // ---- per-row metadata in rank order (rebuilt only when the graph topology changed) -----------
    // rowmeta[r]: bits 0-7 code, 8-15 in-degree, 16 sink flag; prow[r*KIN+p] = matrix row of pred p.
    HD void build_rowmeta() {
        for (int r = g.lane; r < n_nodes; r += GW) {
            int u = r2n[r];
            int k = nin[u];
            for (int p = 0; p < k; ++p) prow[r * KIN + p] = (id_t)(n2r[inp[u * KIN + p]] + 1);
            rowmeta[r] = (uint32_t)code[u] | ((uint32_t)k << 8) | ((nout[u] == 0 ? 1u : 0u) << 16);
        }
        meta_dirty = false;
        g.sync();
    }

    HD void load_row(int row, int S, int (&out)[CPL]) const {
        if (CPL * g.lane < S) {
            Pack pk = *(const Pack*)(H + row * S + CPL * g.lane);
            HYPO_UNROLL
            for (int c = 0; c < CPL; ++c) out[c] = (int)pk.v[c];
        } else {
            HYPO_UNROLL
            for (int c = 0; c < CPL; ++c) out[c] = NEG;
        }
    }

    // ---- engine->align (sisd_alignment_engine.cpp:246-439), linear gaps -----------------------------
    // Leaves posnode[q] (graph node aligned to sequence position q, -1 = insertion) for q in
    // [tb_fv, L) and tb_steps (number of traceback steps; 0 = "empty alignment").
    HD int align(int mode, int m, int n, int gp) {
        tb_steps = 0; tb_fv = L;
        if (n_nodes == 0 || L == 0) return RES_OK;
        const int W = L + 1;
        const int S = (W + CPL - 1) / CPL * CPL;          // row stride
        if ((n_nodes + 1) * S > Cfg::HCELLS) return RES_OVERFLOW;
        if (meta_dirty) build_rowmeta();
        cells += (uint64_t)(n_nodes + 1) * W; aligns += 1;

        const int j0 = CPL * g.lane;
        int sq[CPL];                                       // sq[c] = code of seq[j-1] for column j = j0+c
        HYPO_UNROLL
        for (int c = 0; c < CPL; ++c) { int j = j0 + c; sq[c] = (j >= 1 && j <= L) ? (int)seq[j - 1] : (int)C_NONE; }

        int last[CPL];                                     // most recently computed row (registers)
        HYPO_UNROLL
        for (int c = 0; c < CPL; ++c) last[c] = (j0 + c) * gp;
        if (j0 < S) {
            Pack pk;
            HYPO_UNROLL
            for (int c = 0; c < CPL; ++c) pk.v[c] = (score_t)last[c];
            *(Pack*)(H + j0) = pk;                          // row 0: H[0][j] = j*g (sisd..cpp:197-199,230-232)
        }
        g.sync();

        const int le = L / CPL, ce = L % CPL;              // owner of the last column
        int best = NEG, best_i = -1;

        for (int r = 0; r < n_nodes; ++r) {
            const int i = r + 1;
            const uint32_t meta = rowmeta[r];
            const int cd = (int)(meta & 0xff), k = (int)((meta >> 8) & 0xff);
            const bool sink = (meta >> 16) & 1;
            int v[CPL];
            // first predecessor (row 0 when the node has no in-edge)
            {
                int hp[CPL];
                const int p0 = k ? (int)prow[r * KIN] : 0;
                if (p0 == i - 1) { HYPO_UNROLL for (int c = 0; c < CPL; ++c) hp[c] = last[c]; }
                else load_row(p0, S, hp);
                const int left = g.shfl_up1(hp[CPL - 1], NEG);
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) {
                    const int dsrc = c ? hp[c - 1] : left;
                    const int d = dsrc + (sq[c] == cd ? m : n);
                    const int up = hp[c] + gp;
                    v[c] = d > up ? d : up;
                }
                if (g.lane == 0) v[0] = (mode == MODE_ROV) ? 0 : hp[0] + gp;   // first column (sisd..cpp:200-211,237-239)
            }
            for (int p = 1; p < k; ++p) {
                int hp[CPL];
                load_row((int)prow[r * KIN + p], S, hp);
                const int left = g.shfl_up1(hp[CPL - 1], NEG);
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) {
                    const int dsrc = c ? hp[c - 1] : left;
                    const int d = dsrc + (sq[c] == cd ? m : n);
                    const int up = hp[c] + gp;
                    int x = d > up ? d : up;
                    if (g.lane == 0 && c == 0) x = (mode == MODE_ROV) ? 0 : up;
                    v[c] = x > v[c] ? x : v[c];
                }
            }
            // horizontal term H[i][j] = max(H[i][j], H[i][j-1] + g): prefix max of H[i][j] - j*g
            {
                int run = NEG;
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) {
                    int x = v[c] - (j0 + c) * gp;
                    run = x > run ? x : run;
                    v[c] = run;
                }
                const int ex = g.scan_max_excl(run, NEG);
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) {
                    int x = v[c] > ex ? v[c] : ex;
                    v[c] = x + (j0 + c) * gp;
                }
            }
            if (j0 < S) {
                Pack pk;
                HYPO_UNROLL
                for (int c = 0; c < CPL; ++c) pk.v[c] = (score_t)v[c];
                *(Pack*)(H + i * S + j0) = pk;
            }
            HYPO_UNROLL
            for (int c = 0; c < CPL; ++c) last[c] = v[c];
            // end cell: first strictly greater in rank order (sisd..cpp:279-288,332-339)
            if (g.lane == le && (mode == MODE_LOV || sink)) {
                int val = v[0];
                HYPO_UNROLL
                for (int c = 1; c < CPL; ++c) if (c == ce) val = v[c];
                if (val > best) { best = val; best_i = i; }
            }
            g.sync();
        }
        best_i = g.shfl(best_i, le);

        // ---- traceback (sisd..cpp:344-438) ----
        int i = best_i > 0 ? best_i : 0, j = best_i > 0 ? L : 0;
        int steps = 0;
        while (mode == MODE_ROV ? (i != 0 && j != 0) : (i != 0 || j != 0)) {
            const int hij = (int)H[i * S + j];
            int pi = 0, node = -1, k = 0, mc = 0;
            bool dg = false, vt = false;
            if (i != 0) {
                const uint32_t meta = rowmeta[i - 1];
                k = (int)((meta >> 8) & 0xff);
                node = (int)r2n[i - 1];
                const int np = k ? k : 1;
                if (g.lane < np) {
                    pi = k ? (int)prow[(i - 1) * KIN + g.lane] : 0;
                    if (j != 0) {
                        mc = (seq[j - 1] == (uint8_t)(meta & 0xff)) ? m : n;
                        dg = hij == (int)H[pi * S + j - 1] + mc;
                    }
                    vt = hij == (int)H[pi * S + j] + gp;
                }
            }
            const uint64_t bd = g.ballot(dg), bv = g.ballot(vt);
            int ni_ = i, nj_ = j;
            if (bd) { ni_ = g.shfl(pi, ctz64(bd)); nj_ = j - 1; }
            else if (bv) { ni_ = g.shfl(pi, ctz64(bv)); }
            else if (j != 0 && hij == (int)H[i * S + j - 1] + gp) { nj_ = j - 1; }
            else return RES_UNDEFINED;                     // inconsistent matrix: cannot happen
            if (nj_ != j && g.lane == 0) posnode[j - 1] = (int16_t)(ni_ != i ? node : -1);
            i = ni_; j = nj_; ++steps;
            if (steps > n_nodes + L + 2) return RES_UNDEFINED;   // bounded by construction; guards the GPU against a hang
        }
        tb_steps = steps; tb_fv = j;
        g.sync();
        return RES_OK;
    }

    // ---- graph->add_alignment (graph.cpp:154-271) ---------------------------------------------------
    HD void new_node(int id, int c) {
        code[id] = (uint8_t)c; nin[id] = 0; nout[id] = 0; nal[id] = 0;
    }
    // adds edge prev->to (graph.cpp:99-115).  Returns 0 = existing edge, 1 = new edge, 2 = no room.
    HD int add_edge(int prev, int to) {
        const int k = nin[to];
        for (int p = 0; p < k; ++p)
            if ((int)inp[to * KIN + p] == prev) { inw[to * KIN + p] = (uint16_t)(inw[to * KIN + p] + 2); return 0; }
        if (k == KIN) return 2;
        inp[to * KIN + k] = (id_t)prev; inw[to * KIN + k] = 2; nin[to] = (uint8_t)(k + 1);
        if (nout[prev] != 255) nout[prev] = (uint8_t)(nout[prev] + 1);
        return 1;
    }
    HD int add_alignment() {
        if (L == 0) return RES_OK;
        const int fv = tb_steps == 0 ? L : tb_fv;          // empty alignment -> the whole sequence is a fresh chain
        if (tb_steps != 0 && fv == L) return RES_UNDEFINED; // graph.cpp:184-200: no sequence position aligned
        bool changed = false;
        // unaligned head [0, fv): new chain (graph.cpp:194-196,273-291)
        int head = -1;
        if (fv > 0) {
            if (n_nodes + fv > NMAX) return RES_OVERFLOW;
            for (int t = g.lane; t < fv; t += GW) {
                const int id = n_nodes + t;
                new_node(id, seq[t]);
                if (t > 0) { nin[id] = 1; inp[id * KIN] = (id_t)(id - 1); inw[id * KIN] = 2; }
                if (t < fv - 1) nout[id] = 1;
            }
            head = n_nodes + fv - 1;
            n_nodes += fv;
            changed = true;
        }
        g.sync();
        // aligned part [fv, L): every position owns a distinct node / clique
        bool over = false;
        for (int base = fv; base < L; base += GW) {
            const int q = base + g.lane;
            const bool act = q < L;
            int kind = 0, tgt = -1, nd = -1, c = 0;      // kind 0 reuse, 1 new unaligned, 2 new aligned to nd
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
            if (n_nodes + tot > NMAX) { over = true; break; }
            if (act && kind != 0) {
                const int id = n_nodes + popc64(nb & ((1ull << g.lane) - 1ull));
                new_node(id, c);
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
            if (act) cur[q] = (int16_t)tgt;
            n_nodes += tot;
            if (tot) changed = true;
        }
        if (g.any(over)) return RES_OVERFLOW;
        g.sync();
        // edges between consecutive positions (graph.cpp:250-258)
        int st = 0;
        for (int base = fv; base < L; base += GW) {
            const int q = base + g.lane;
            if (q < L) {
                const int prev = q == fv ? head : (int)cur[q - 1];
                if (prev >= 0) { const int e = add_edge(prev, (int)cur[q]); st = e > st ? e : st; }
            }
        }
        const int sm = g.reduce_max(st);
        if (sm == 2) return RES_OVERFLOW;
        if (sm == 1) changed = true;
        g.sync();
        if (changed) { topo_dirty = true; meta_dirty = true; }
        return RES_OK;
    }

    // ---- Graph::topological_sort (graph.cpp:293-353) ---------------------------------------------
    // mark bit0 = permanently marked, bit1 = "aligned nodes already pushed by another clique member".
    HD int toposort() {
        for (int t = g.lane; t < n_nodes; t += GW) mark[t] = 0;
        g.sync();
        int cnt = 0, sp = 0;
        for (int root = 0; root < n_nodes; ++root) {
            if (mark[root] & 1) continue;
            if (g.lane == 0) stack[0] = (id_t)root;
            sp = 1;
            g.sync();
            int guard = 0;
            while (sp > 0) {
                if (++guard > 4 * Cfg::STK + 16) return RES_UNDEFINED;   // cannot loop on a DAG; hang guard
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
        }
        for (int r = g.lane; r < n_nodes; r += GW) n2r[r2n[r]] = (id_t)r;
        topo_dirty = false;
        g.sync();
        return cnt == n_nodes ? RES_OK : RES_UNDEFINED;
    }

    HD int add_sequence_step(int mode, int m, int n, int gp) {
        int rc = align(mode, m, n, gp);
        if (rc != RES_OK) return rc;
        rc = add_alignment();
        if (rc != RES_OK) return rc;
        if (topo_dirty) rc = toposort();
        return rc;
    }

    // ---- Graph::generate_consensus (graph.cpp:467-476,610-705) --------------------------------------
    // Scratch aliases the score matrix: score[NMAX] int32, pred[NMAX] int16, path[NMAX] int16.
    HD int consensus(int16_t** path_out) {
        int32_t* score = (int32_t*)H;
        int16_t* pred = (int16_t*)(score + NMAX);
        int16_t* path = pred + NMAX;
        g.sync();
        for (int t = g.lane; t < n_nodes; t += GW) { score[t] = -1; pred[t] = -1; }
        g.sync();
        int max_id = 0;
        if (g.lane == 0) {
            for (int r = 0; r < n_nodes; ++r) {
                const int u = r2n[r];
                const int k = nin[u];
                int s = -1, pd = -1;
                for (int p = 0; p < k; ++p) {
                    const int w = inw[u * KIN + p], b = inp[u * KIN + p];
                    if (s < w || (s == w && score[pd] <= score[b])) { s = w; pd = b; }
                }
                if (pd != -1) s += score[pd];
                score[u] = s; pred[u] = (int16_t)pd;
                if (score[max_id] < s) max_id = u;
            }
        }
        max_id = g.shfl(max_id, 0);
        g.sync();
        // branch completion (graph.cpp:660-705) while the best node is not a sink
        int rounds = 0;
        while (nout[max_id] != 0) {
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
                    int s = -1, pd = -1;
                    for (int p = 0; p < k; ++p) {
                        const int w = inw[u * KIN + p], b = inp[u * KIN + p];
                        if (score[b] == -1) continue;
                        if (s < w || (s == w && score[pd] <= score[b])) { s = w; pd = b; }
                    }
                    if (pd != -1) s += score[pd];
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

    // ---- outputs -----------------------------------------------------------------------------------
    HD void finish(uint32_t w, int status, uint32_t len) const {
        if (g.lane == 0) { P.out_len[w] = len; P.out_status[w] = (uint8_t)status; }
    }
    HD int emit_draft(uint32_t w, const uint8_t* d4, int dlen) const {
        const uint64_t o = P.out_off[w], cap = P.out_off[w + 1] - o;
        if ((uint64_t)dlen > cap) { finish(w, HYPO_ST_CONS_OVERFLOW, (uint32_t)dlen); return RES_OK; }
        for (int t = g.lane; t < dlen; t += GW) {
            int c = (d4[t >> 1] >> (4 - 4 * (t & 1))) & 15;
            P.out_bases[o + t] = "ACGTN"[c < 4 ? c : 4];
        }
        finish(w, HYPO_ST_OK, (uint32_t)dlen);
        return RES_OK;
    }

    // Window::generate_consensus_short (src/Window.cpp:87-154)
    HD int run_short(uint32_t w, const HypoWindow& W) {
        const int m = P.sr_m, n = P.sr_n, gp = P.sr_g;
        const uint8_t* d4 = P.draft4 + W.draft_off;
        n_nodes = 0; topo_dirty = false; meta_dirty = true;
        bool added = false;
        int rc;
        if (W.n_internal == 0) {                            // draft as backbone only without internal arms
            if ((rc = load_seq(d4, (int)W.draft_len, true, true, true)) != RES_OK) return rc;
            if ((rc = add_sequence_step(MODE_NW, m, n, gp)) != RES_OK) return rc;
        }
        const uint32_t a0 = W.first_arm;
        for (uint32_t a = 0; a < W.n_internal; ++a) {
            const int len = (int)P.arm_len[a0 + a];
            if (len == 0) continue;
            added = true;
            if ((rc = load_seq(P.arms2 + P.arm_off[a0 + a], len, false, true, true)) != RES_OK) return rc;
            if ((rc = add_sequence_step(MODE_NW, m, n, gp)) != RES_OK) return rc;
        }
        for (uint32_t t = 0; t < W.n_prefix; ++t) {        // reverse insertion order (Window.cpp:111)
            const uint32_t a = a0 + W.n_internal + (W.n_prefix - 1 - t);
            const int len = (int)P.arm_len[a];
            if (len == 0) continue;
            added = true;
            if ((rc = load_seq(P.arms2 + P.arm_off[a], len, false, true, false)) != RES_OK) return rc;
            if ((rc = add_sequence_step(MODE_LOV, m, n, gp)) != RES_OK) return rc;
        }
        for (uint32_t t = 0; t < W.n_suffix; ++t) {
            const uint32_t a = a0 + W.n_internal + W.n_prefix + t;
            const int len = (int)P.arm_len[a];
            if (len == 0) continue;
            added = true;
            if ((rc = load_seq(P.arms2 + P.arm_off[a], len, false, false, true)) != RES_OK) return rc;
            if ((rc = add_sequence_step(MODE_ROV, m, n, gp)) != RES_OK) return rc;
        }
        if (!added) return emit_draft(w, d4, (int)W.draft_len);
        int16_t* path;
        const int len = consensus(&path);
        if (len < 2) return RES_UNDEFINED;                  // Window.hpp:144 strips two markers
        const int olen = len - 2;
        const uint64_t o = P.out_off[w], cap = P.out_off[w + 1] - o;
        if ((uint64_t)olen > cap) { finish(w, HYPO_ST_CONS_OVERFLOW, (uint32_t)olen); return RES_OK; }
        for (int t = g.lane; t < olen; t += GW) P.out_bases[o + t] = "ACGTNJO"[code[path[len - 2 - t]]];
        finish(w, HYPO_ST_OK, (uint32_t)olen);
        return RES_OK;
    }

    // Window::generate_consensus (src/Window.cpp:44-61)
    HD int run(uint32_t w) {
        const HypoWindow W = P.windows[w];
        const uint32_t ne = W.n_internal + W.n_prefix + W.n_suffix;
        if (W.n_empty > ne) { finish(w, HYPO_ST_OK, 0); return RES_OK; }
        if (ne < 2) return emit_draft(w, P.draft4 + W.draft_off, (int)W.draft_len);
        if (W.type != HYPO_WIN_SHORT) return RES_UNSUPPORTED;
        return run_short(w, W);
    }
};

}  // namespace hypo
