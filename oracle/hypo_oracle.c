/*
 * hypo_oracle.c — CPU restatement (plain C) of HyPo's per-window polishing hot path.
 * TEST INFRASTRUCTURE ONLY (see hypo_oracle.h).  Written from the behaviour of the cited reference
 * lines; no reference source is copied.  Flat arrays instead of the reference's heap objects, but
 * every ordering that influences the result (in-edge creation order, aligned-list order, DFS order,
 * tie rules) is reproduced literally.
 *
 *   engine      : external/spoa/src/sisd_alignment_engine.cpp:95-439 (linear gap, kNW/kLOV/kROV)
 *   graph       : external/spoa/src/graph.cpp:93-128,154-353,371-388,467-476,533-568,610-705
 *   window      : src/Window.cpp:44-61,87-154,156-254 ; include/Window.hpp:30-33,144
 *   packed seq  : src/PackedSeq.cpp:58-89,231-262 ; include/PackedSeq.hpp:35-72
 *   solid scan  : src/Contig.cpp:40-74 ; external/suk/include/suk/SolidKmers.hpp:119
 */
#include "hypo_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>

static volatile int g_native_klov = 0;     /* oracle_set_native_klov */
#endif

/* ------------------------------------------------------------------------------------------------
 * small growable int vector
 * --------------------------------------------------------------------------------------------- */
typedef struct { int* d; int n, cap; } ivec;

static void iv_push(ivec* v, int x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 4;
        v->d = (int*)realloc(v->d, (size_t)v->cap * sizeof(int));
    }
    v->d[v->n++] = x;
}
static void iv_free(ivec* v) { free(v->d); v->d = NULL; v->n = v->cap = 0; }

/* ------------------------------------------------------------------------------------------------
 * POA graph (graph.cpp).  Nodes and edges are indices into flat arrays; per-node lists keep the
 * reference's insertion order (in_edges_/out_edges_/aligned_nodes_ids_, graph.cpp:111-114,229-238).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int n, cap;              /* nodes */
    char* letter;            /* decoder_[code_] of the node */
    ivec *in, *out, *al;     /* in: edge ids, out: edge ids, al: node ids */
    int ne, ecap;            /* edges */
    int *eb, *ee;            /* begin / end node */
    long long* ew;           /* total_weight_ (int64, graph.hpp) */
    ivec* elab;              /* sequence_labels_ (kept only if keep_labels) */
    int keep_labels;
    int nseq;                /* num_sequences_ */
    ivec seqbeg;             /* sequences_begin_nodes_ids_ */
    ivec rank;               /* rank_to_node_id_ */
    ivec cons;               /* consensus_ */
    /* scratch reused between calls */
    unsigned char* mark; unsigned char* chk; ivec stack;
    long long* score; int* pred; int* n2r; int scap;
} graph;

static void g_reset(graph* g, int keep_labels) {
    g->n = 0; g->ne = 0; g->nseq = 0; g->keep_labels = keep_labels;
    g->seqbeg.n = 0; g->rank.n = 0; g->cons.n = 0;
}
static void g_free(graph* g) {
    for (int i = 0; i < g->cap; ++i) { iv_free(&g->in[i]); iv_free(&g->out[i]); iv_free(&g->al[i]); }
    for (int i = 0; i < g->ecap; ++i) iv_free(&g->elab[i]);
    free(g->letter); free(g->in); free(g->out); free(g->al);
    free(g->eb); free(g->ee); free(g->ew); free(g->elab);
    iv_free(&g->seqbeg); iv_free(&g->rank); iv_free(&g->cons); iv_free(&g->stack);
    free(g->mark); free(g->chk); free(g->score); free(g->pred); free(g->n2r);
    memset(g, 0, sizeof(*g));
}
/* graph.cpp:93-97 */
static int g_add_node(graph* g, char letter) {
    if (g->n == g->cap) {
        int nc = g->cap ? 2 * g->cap : 256;
        g->letter = (char*)realloc(g->letter, (size_t)nc);
        g->in = (ivec*)realloc(g->in, (size_t)nc * sizeof(ivec));
        g->out = (ivec*)realloc(g->out, (size_t)nc * sizeof(ivec));
        g->al = (ivec*)realloc(g->al, (size_t)nc * sizeof(ivec));
        memset(g->in + g->cap, 0, (size_t)(nc - g->cap) * sizeof(ivec));
        memset(g->out + g->cap, 0, (size_t)(nc - g->cap) * sizeof(ivec));
        memset(g->al + g->cap, 0, (size_t)(nc - g->cap) * sizeof(ivec));
        g->cap = nc;
    }
    int id = g->n++;
    g->letter[id] = letter;
    g->in[id].n = 0; g->out[id].n = 0; g->al[id].n = 0;
    return id;
}
/* graph.cpp:99-115: linear search over the begin node's out-edges; new edges are appended to the
 * begin node's out list and the end node's in list (creation order). */
static void g_add_edge(graph* g, int b, int e, int w) {
    ivec* o = &g->out[b];
    for (int i = 0; i < o->n; ++i) {
        int ed = o->d[i];
        if (g->ee[ed] == e) {
            g->ew[ed] += w;
            if (g->keep_labels) iv_push(&g->elab[ed], g->nseq);
            return;
        }
    }
    if (g->ne == g->ecap) {
        int nc = g->ecap ? 2 * g->ecap : 512;
        g->eb = (int*)realloc(g->eb, (size_t)nc * sizeof(int));
        g->ee = (int*)realloc(g->ee, (size_t)nc * sizeof(int));
        g->ew = (long long*)realloc(g->ew, (size_t)nc * sizeof(long long));
        g->elab = (ivec*)realloc(g->elab, (size_t)nc * sizeof(ivec));
        memset(g->elab + g->ecap, 0, (size_t)(nc - g->ecap) * sizeof(ivec));
        g->ecap = nc;
    }
    int ed = g->ne++;
    g->eb[ed] = b; g->ee[ed] = e; g->ew[ed] = w;
    g->elab[ed].n = 0;
    if (g->keep_labels) iv_push(&g->elab[ed], g->nseq);
    iv_push(&g->out[b], ed);
    iv_push(&g->in[e], ed);
}
/* graph.cpp:273-291 (all per-base weights are 1 in HyPo: graph.cpp:117-128) */
static int g_add_sequence(graph* g, const char* s, int begin, int end) {
    if (begin == end) return -1;
    int first = g_add_node(g, s[begin]);
    for (int i = begin + 1; i < end; ++i) {
        int id = g_add_node(g, s[i]);
        g_add_edge(g, id - 1, id, 2);
    }
    return first;
}
static void g_scratch(graph* g) {
    if (g->scap < g->n + 1) {
        g->scap = 2 * (g->n + 1);
        g->mark = (unsigned char*)realloc(g->mark, (size_t)g->scap);
        g->chk = (unsigned char*)realloc(g->chk, (size_t)g->scap);
        g->score = (long long*)realloc(g->score, (size_t)g->scap * sizeof(long long));
        g->pred = (int*)realloc(g->pred, (size_t)g->scap * sizeof(int));
        g->n2r = (int*)realloc(g->n2r, (size_t)g->scap * sizeof(int));
    }
}
/* graph.cpp:293-353: iterative DFS over in-edges and aligned nodes; a node whose aligned nodes were
 * pushed by somebody else (chk == 0) is marked but emitted by the clique's checking member. */
static void g_toposort(graph* g) {
    g_scratch(g);
    g->rank.n = 0;
    memset(g->mark, 0, (size_t)g->n);
    memset(g->chk, 1, (size_t)g->n);
    g->stack.n = 0;
    for (int i = 0; i < g->n; ++i) {
        if (g->mark[i] != 0) continue;
        iv_push(&g->stack, i);
        while (g->stack.n != 0) {
            int v = g->stack.d[g->stack.n - 1];
            int valid = 1;
            if (g->mark[v] != 2) {
                for (int k = 0; k < g->in[v].n; ++k) {
                    int b = g->eb[g->in[v].d[k]];
                    if (g->mark[b] != 2) { iv_push(&g->stack, b); valid = 0; }
                }
                if (g->chk[v]) {
                    for (int k = 0; k < g->al[v].n; ++k) {
                        int a = g->al[v].d[k];
                        if (g->mark[a] != 2) { iv_push(&g->stack, a); g->chk[a] = 0; valid = 0; }
                    }
                }
                if (valid) {
                    g->mark[v] = 2;
                    if (g->chk[v]) {
                        iv_push(&g->rank, v);
                        for (int k = 0; k < g->al[v].n; ++k) iv_push(&g->rank, g->al[v].d[k]);
                    }
                } else {
                    g->mark[v] = 1;
                }
            }
            if (valid) g->stack.n--;
        }
    }
}

typedef struct { int node, pos; } apair;
typedef struct { apair* d; int n, cap; } pvec;
static void pv_push(pvec* v, int node, int pos) {
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 256;
        v->d = (apair*)realloc(v->d, (size_t)v->cap * sizeof(apair));
    }
    v->d[v->n].node = node; v->d[v->n].pos = pos; v->n++;
}

/* graph.cpp:154-271.  Returns 0, or -3 when the alignment has no sequence position at all
 * (valid_seq_ids.front() on an empty vector: undefined behaviour in the reference). */
static int g_add_alignment(graph* g, const pvec* aln, const char* s, int L) {
    if (L == 0) return 0;
    if (aln->n == 0) {
        int b = g_add_sequence(g, s, 0, L);
        g->nseq++;
        iv_push(&g->seqbeg, b);
        g_toposort(g);
        return 0;
    }
    int first_valid = -1, last_valid = -1;
    for (int i = 0; i < aln->n; ++i)
        if (aln->d[i].pos != -1) { if (first_valid < 0) first_valid = aln->d[i].pos; last_valid = aln->d[i].pos; }
    if (first_valid < 0) return -3;

    int before = g->n;
    int begin_node = g_add_sequence(g, s, 0, first_valid);
    int head = (before == g->n) ? -1 : g->n - 1;
    int tail = g_add_sequence(g, s, last_valid + 1, L);
    int cur = -1;
    for (int i = 0; i < aln->n; ++i) {
        int pos = aln->d[i].pos, nd = aln->d[i].node;
        if (pos == -1) continue;
        char c = s[pos];
        if (nd == -1) {
            cur = g_add_node(g, c);
        } else if (g->letter[nd] == c) {
            cur = nd;
        } else {
            int found = -1;
            for (int k = 0; k < g->al[nd].n; ++k)
                if (g->letter[g->al[nd].d[k]] == c) { found = g->al[nd].d[k]; break; }
            if (found == -1) {
                cur = g_add_node(g, c);
                int cnt = g->al[nd].n;           /* list of nd before it learns about cur */
                for (int k = 0; k < cnt; ++k) {
                    int a = g->al[nd].d[k];
                    iv_push(&g->al[cur], a);
                    iv_push(&g->al[a], cur);
                }
                iv_push(&g->al[cur], nd);
                iv_push(&g->al[nd], cur);
            } else {
                cur = found;
            }
        }
        if (begin_node == -1) begin_node = cur;
        if (head != -1) g_add_edge(g, head, cur, 2);
        head = cur;
    }
    if (tail != -1) g_add_edge(g, head, tail, 2);
    g->nseq++;
    iv_push(&g->seqbeg, begin_node);
    g_toposort(g);
    return 0;
}

/* graph.cpp:660-705 */
static int g_branch_completion(graph* g, int rank) {
    int v = g->rank.d[rank];
    for (int k = 0; k < g->out[v].n; ++k) {
        int t = g->ee[g->out[v].d[k]];
        for (int q = 0; q < g->in[t].n; ++q) {
            int b = g->eb[g->in[t].d[q]];
            if (b != v) g->score[b] = -1;
        }
    }
    long long max_score = 0; int max_id = 0;
    for (int i = rank + 1; i < g->rank.n; ++i) {
        int u = g->rank.d[i];
        g->score[u] = -1; g->pred[u] = -1;
        for (int q = 0; q < g->in[u].n; ++q) {
            int ed = g->in[u].d[q]; int b = g->eb[ed];
            if (g->score[b] == -1) continue;
            if (g->score[u] < g->ew[ed] ||
                (g->score[u] == g->ew[ed] && g->score[g->pred[u]] <= g->score[b])) {
                g->score[u] = g->ew[ed]; g->pred[u] = b;
            }
        }
        if (g->pred[u] != -1) g->score[u] += g->score[g->pred[u]];
        if (max_score < g->score[u]) { max_score = g->score[u]; max_id = u; }
    }
    return max_id;
}
/* graph.cpp:610-658 */
static void g_heaviest_bundle(graph* g) {
    g_scratch(g);
    for (int i = 0; i < g->n; ++i) { g->pred[i] = -1; g->score[i] = -1; }
    int max_id = 0;
    for (int r = 0; r < g->rank.n; ++r) {
        int u = g->rank.d[r];
        for (int q = 0; q < g->in[u].n; ++q) {
            int ed = g->in[u].d[q]; int b = g->eb[ed];
            if (g->score[u] < g->ew[ed] ||
                (g->score[u] == g->ew[ed] && g->score[g->pred[u]] <= g->score[b])) {
                g->score[u] = g->ew[ed]; g->pred[u] = b;
            }
        }
        if (g->pred[u] != -1) g->score[u] += g->score[g->pred[u]];
        if (g->score[max_id] < g->score[u]) max_id = u;
    }
    if (g->out[max_id].n != 0) {
        for (int i = 0; i < g->n; ++i) g->n2r[g->rank.d[i]] = i;
        while (g->out[max_id].n != 0) max_id = g_branch_completion(g, g->n2r[max_id]);
    }
    g->cons.n = 0;
    while (g->pred[max_id] != -1) { iv_push(&g->cons, max_id); max_id = g->pred[max_id]; }
    iv_push(&g->cons, max_id);
    for (int a = 0, b = g->cons.n - 1; a < b; ++a, --b) { int t = g->cons.d[a]; g->cons.d[a] = g->cons.d[b]; g->cons.d[b] = t; }
}
/* graph.cpp:30-41 */
static int g_successor(const graph* g, int v, int label) {
    for (int k = 0; k < g->out[v].n; ++k) {
        int ed = g->out[v].d[k];
        for (int q = 0; q < g->elab[ed].n; ++q) if (g->elab[ed].d[q] == label) return g->ee[ed];
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------------
 * alignment engine (sisd_alignment_engine.cpp), linear gaps only (HyPo always builds kLinear:
 * alignment_engine.cpp:18-22,52-60)
 * --------------------------------------------------------------------------------------------- */
typedef struct { int* H; size_t hcap; int* n2r; int ncap; uint64_t cells, aligns; } engine;

#define NEG_INF (INT_MIN + 1024)

static void eng_align(engine* E, const graph* g, const char* s, int L, int mode,
                      int m, int n_, int gp, pvec* out) {
    out->n = 0;
    if (g->n == 0 || L == 0) return;             /* sisd..cpp:249-251 */
    const int W = L + 1, Hh = g->n + 1;
    if (E->hcap < (size_t)W * Hh) { E->hcap = (size_t)W * Hh * 2; E->H = (int*)realloc(E->H, E->hcap * sizeof(int)); }
    if (E->ncap < g->n) { E->ncap = 2 * g->n; E->n2r = (int*)realloc(E->n2r, (size_t)E->ncap * sizeof(int)); }
    int* H = E->H; int* n2r = E->n2r;
    E->cells += (uint64_t)W * Hh; E->aligns++;
    for (int r = 0; r < g->n; ++r) n2r[g->rank.d[r]] = r;

    /* initialize, sisd..cpp:163-243 */
    H[0] = 0;
    for (int j = 1; j < W; ++j) H[j] = j * gp;
    if (mode == ORACLE_NW || mode == ORACLE_LOV) {
        for (int i = 1; i < Hh; ++i) {
            int u = g->rank.d[i - 1];
            int pen = g->in[u].n == 0 ? 0 : NEG_INF;
            for (int q = 0; q < g->in[u].n; ++q) {
                int pi = n2r[g->eb[g->in[u].d[q]]] + 1;
                if (H[(size_t)pi * W] > pen) pen = H[(size_t)pi * W];
            }
            H[(size_t)i * W] = pen + gp;
        }
    } else { /* ROV */
        for (int i = 1; i < Hh; ++i) H[(size_t)i * W] = 0;
    }

    int max_score = NEG_INF, max_i = -1, max_j = -1;
    /* row loop, sisd..cpp:291-342 */
    for (int r = 0; r < g->n; ++r) {
        int u = g->rank.d[r];
        int i = r + 1;
        char c = g->letter[u];
        int* row = H + (size_t)i * W;
        int pi = g->in[u].n == 0 ? 0 : n2r[g->eb[g->in[u].d[0]]] + 1;
        const int* prow = H + (size_t)pi * W;
        for (int j = 1; j < W; ++j) {
            int a = prow[j - 1] + (c == s[j - 1] ? m : n_);
            int b = prow[j] + gp;
            row[j] = a > b ? a : b;
        }
        for (int q = 1; q < g->in[u].n; ++q) {
            pi = n2r[g->eb[g->in[u].d[q]]] + 1;
            prow = H + (size_t)pi * W;
            for (int j = 1; j < W; ++j) {
                int a = prow[j - 1] + (c == s[j - 1] ? m : n_);
                int b = prow[j] + gp;
                int x = row[j] > b ? row[j] : b;
                row[j] = a > x ? a : x;
            }
        }
        for (int j = 1; j < W; ++j) {
            int h = row[j - 1] + gp;
            if (h > row[j]) row[j] = h;
        }
        int is_end = 0;
        if (mode == ORACLE_LOV) is_end = 1;                                     /* :338-339 */
        else if (g->out[u].n == 0) is_end = 1;                                   /* :332-334 */
        int endval = row[W - 1];
        if (mode == ORACLE_LOV && g_native_klov) {
            /* the AVX2 / SSE4.1 engine of a -march=native reference build ranks kLOV rows by the maximum over the whole row
             * (columns 1..L; its padding columns never exceed column L) and still starts the traceback in column L
             * (simd_alignment_engine.cpp:803,834-840,859-861) */
            for (int j = 1; j < W; ++j) if (row[j] > endval) endval = row[j];
        }
        if (is_end && max_score < endval) { max_score = endval; max_i = i; max_j = W - 1; }
    }

    /* backtrack, sisd..cpp:344-438 */
    int i = max_i > 0 ? max_i : 0, j = max_j > 0 ? max_j : 0;
    int prev_i = 0, prev_j = 0;
    for (;;) {
        if (mode == ORACLE_ROV) { if (i == 0 || j == 0) break; }
        else { if (i == 0 && j == 0) break; }
        int Hij = H[(size_t)i * W + j];
        int found = 0;
        if (i != 0 && j != 0) {
            int u = g->rank.d[i - 1];
            int mc = g->letter[u] == s[j - 1] ? m : n_;
            int pi = g->in[u].n == 0 ? 0 : n2r[g->eb[g->in[u].d[0]]] + 1;
            if (Hij == H[(size_t)pi * W + j - 1] + mc) { prev_i = pi; prev_j = j - 1; found = 1; }
            else for (int q = 1; q < g->in[u].n; ++q) {
                pi = n2r[g->eb[g->in[u].d[q]]] + 1;
                if (Hij == H[(size_t)pi * W + j - 1] + mc) { prev_i = pi; prev_j = j - 1; found = 1; break; }
            }
        }
        if (!found && i != 0) {
            int u = g->rank.d[i - 1];
            int pi = g->in[u].n == 0 ? 0 : n2r[g->eb[g->in[u].d[0]]] + 1;
            if (Hij == H[(size_t)pi * W + j] + gp) { prev_i = pi; prev_j = j; found = 1; }
            else for (int q = 1; q < g->in[u].n; ++q) {
                pi = n2r[g->eb[g->in[u].d[q]]] + 1;
                if (Hij == H[(size_t)pi * W + j] + gp) { prev_i = pi; prev_j = j; found = 1; break; }
            }
        }
        if (!found && j != 0 && Hij == H[(size_t)i * W + j - 1] + gp) { prev_i = i; prev_j = j - 1; found = 1; }
        pv_push(out, i == prev_i ? -1 : g->rank.d[i - 1], j == prev_j ? -1 : j - 1);
        if (!found && i == prev_i && j == prev_j) break;   /* cannot happen for a consistent matrix; avoids a hang */
        i = prev_i; j = prev_j;
    }
    for (int a = 0, b = out->n - 1; a < b; ++a, --b) { apair t = out->d[a]; out->d[a] = out->d[b]; out->d[b] = t; }
}

/* ------------------------------------------------------------------------------------------------
 * PackedSeq (src/PackedSeq.cpp)
 * --------------------------------------------------------------------------------------------- */
static unsigned char nt4(char c) {                     /* globalDefs.hpp:160-178 cNt4Table */
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': case 'U': case 'u': return 3; case 0: return 0; case 1: return 1; case 2: return 2; case 3: return 3;
                 default: return 4; }
}
void oracle_pack2(const char* s, uint32_t len, uint8_t* dst) {
    memset(dst, 0, (len + 3) / 4);
    for (uint32_t i = 0; i < len; ++i) dst[i >> 2] |= (uint8_t)((nt4(s[i]) & 3) << (6 - 2 * (i & 3)));
}
void oracle_pack4(const char* s, uint32_t len, uint8_t* dst) {
    memset(dst, 0, (len + 1) / 2);
    for (uint32_t i = 0; i < len; ++i) dst[i >> 1] |= (uint8_t)(nt4(s[i]) << (4 - 4 * (i & 1)));
}
void oracle_unpack2(const uint8_t* src, uint32_t len, char* dst) {
    for (uint32_t i = 0; i < len; ++i) dst[i] = "ACGT"[(src[i >> 2] >> (6 - 2 * (i & 3))) & 3];
}
void oracle_unpack4(const uint8_t* src, uint32_t len, char* dst) {   /* PackedSeq.hpp:54-72: codes > 3 -> 'N' */
    for (uint32_t i = 0; i < len; ++i) { unsigned c = (src[i >> 1] >> (4 - 4 * (i & 1))) & 15; dst[i] = c < 4 ? "ACGT"[c] : 'N'; }
}

/* ------------------------------------------------------------------------------------------------
 * Window (src/Window.cpp)
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    graph g; engine e; pvec aln;
    char* buf; size_t bufcap;         /* unpacked sequence + markers */
    char* cons; size_t conscap;       /* consensus text */
    char* cons2; unsigned* dst; int* msa;
} wctx;

static void ctx_free(wctx* c) {
    g_free(&c->g); free(c->e.H); free(c->e.n2r); free(c->aln.d); free(c->buf); free(c->cons); free(c->cons2); free(c->dst); free(c->msa);
}
static char* ctx_buf(wctx* c, size_t need) {
    if (c->bufcap < need) { c->bufcap = 2 * need + 64; c->buf = (char*)realloc(c->buf, c->bufcap); }
    return c->buf;
}
static void ctx_cons(wctx* c, size_t need) {
    if (c->conscap < need) {
        c->conscap = 2 * need + 64;
        c->cons = (char*)realloc(c->cons, c->conscap);
        c->cons2 = (char*)realloc(c->cons2, c->conscap);
        c->dst = (unsigned*)realloc(c->dst, c->conscap * sizeof(unsigned));
    }
}
/* engine->align + graph->add_alignment */
static int ctx_add(wctx* c, const char* s, int L, int mode, int m, int n, int gp) {
    eng_align(&c->e, &c->g, s, L, mode, m, n, gp, &c->aln);
    return g_add_alignment(&c->g, &c->aln, s, L);
}
/* graph.cpp:467-476 */
static int ctx_consensus(wctx* c) {
    g_heaviest_bundle(&c->g);
    ctx_cons(c, (size_t)c->g.cons.n + 1);
    for (int i = 0; i < c->g.cons.n; ++i) c->cons[i] = c->g.letter[c->g.cons.d[i]];
    return c->g.cons.n;
}
/* graph.cpp:533-568 + :371-388 */
static int ctx_consensus_custom(wctx* c) {
    int n = ctx_consensus(c);
    graph* g = &c->g;
    c->msa = (int*)realloc(c->msa, (size_t)(g->n + 1) * sizeof(int));
    int msa_id = 0;
    for (int i = 0; i < g->n; ++i) {
        int u = g->rank.d[i];
        c->msa[u] = msa_id;
        for (int k = 0; k < g->al[u].n; ++k) c->msa[g->rank.d[++i]] = msa_id;
        ++msa_id;
    }
    for (int i = 0; i < n; ++i) c->dst[i] = 0;
    for (int sidx = 0; sidx < g->nseq; ++sidx) {
        int v = g->seqbeg.d[sidx];
        int k = 0;
        for (;;) {
            while (k < n && c->msa[g->cons.d[k]] < c->msa[v]) ++k;
            if (k >= n) break;
            if (c->msa[g->cons.d[k]] == c->msa[v] && g->letter[v] == c->cons[k]) c->dst[k]++;
            int nx = g_successor(g, v, sidx);
            if (nx < 0) break;
            v = nx;
        }
    }
    return n;
}

typedef struct { const char* s; int len; } sview;

/* Window::generate_consensus (Window.cpp:44-61).  draft: text (ACGTN).  arms: text views in
 * insertion order.  Returns consensus length (written to c->cons) or <0 (HYPO_ST_* negated). */
static int window_consensus(wctx* c, const HypoScoreParams* sp, int type,
                            const char* draft, int dlen,
                            const sview* in, int ni, const sview* pre, int np, const sview* suf, int ns,
                            int n_empty) {
    int ne = ni + np + ns;
    if (n_empty > ne) return 0;                                        /* "" */
    if (ne < 2) { ctx_cons(c, (size_t)dlen + 1); memcpy(c->cons, draft, (size_t)dlen); return dlen; }

    if (type == HYPO_WIN_SHORT) {                                      /* Window.cpp:87-154 */
        const int m = sp->sr_match, n = sp->sr_mismatch, gp = sp->sr_gap;
        g_reset(&c->g, 0);
        int added = 0, rc;
        if (ni == 0) {
            char* b = ctx_buf(c, (size_t)dlen + 2);
            b[0] = 'J'; memcpy(b + 1, draft, (size_t)dlen); b[dlen + 1] = 'O';
            if ((rc = ctx_add(c, b, dlen + 2, ORACLE_NW, m, n, gp)) < 0) return rc;
        }
        for (int i = 0; i < ni; ++i) if (in[i].len > 0) {
            char* b = ctx_buf(c, (size_t)in[i].len + 2);
            b[0] = 'J'; memcpy(b + 1, in[i].s, (size_t)in[i].len); b[in[i].len + 1] = 'O';
            added = 1;
            if ((rc = ctx_add(c, b, in[i].len + 2, ORACLE_NW, m, n, gp)) < 0) return rc;
        }
        for (int i = np - 1; i >= 0; --i) if (pre[i].len > 0) {         /* reverse order, :111 */
            char* b = ctx_buf(c, (size_t)pre[i].len + 1);
            b[0] = 'J'; memcpy(b + 1, pre[i].s, (size_t)pre[i].len);
            added = 1;
            if ((rc = ctx_add(c, b, pre[i].len + 1, ORACLE_LOV, m, n, gp)) < 0) return rc;
        }
        for (int i = 0; i < ns; ++i) if (suf[i].len > 0) {
            char* b = ctx_buf(c, (size_t)suf[i].len + 1);
            memcpy(b, suf[i].s, (size_t)suf[i].len); b[suf[i].len] = 'O';
            added = 1;
            if ((rc = ctx_add(c, b, suf[i].len + 1, ORACLE_ROV, m, n, gp)) < 0) return rc;
        }
        if (!added) { ctx_cons(c, (size_t)dlen + 1); memcpy(c->cons, draft, (size_t)dlen); return dlen; }
        int len = ctx_consensus(c);
        if (len < 2) return -HYPO_ST_UNDEFINED;                         /* Window.hpp:144 would be UB */
        memmove(c->cons, c->cons + 1, (size_t)len - 2);                 /* strip markers */
        return len - 2;
    }

    /* LONG, Window.cpp:156-236: every alignment uses the long engine, which stays kNW (:36-38) */
    const int m = sp->lr_match, n = sp->lr_mismatch, gp = sp->lr_gap;
    int conslen = 0;
    for (int round = 0; round < 2; ++round) {
        g_reset(&c->g, 1);
        int added = 0, rc;
        if (round == 0) {
            if ((rc = ctx_add(c, draft, dlen, ORACLE_NW, m, n, gp)) < 0) return rc;
        } else if (conslen > 0) {
            char* b = ctx_buf(c, (size_t)conslen);
            memcpy(b, c->cons, (size_t)conslen);
            if ((rc = ctx_add(c, b, conslen, ORACLE_NW, m, n, gp)) < 0) return rc;
        }
        const sview* grp[3] = { in, pre, suf }; const int cnt[3] = { ni, np, ns };
        for (int k = 0; k < 3; ++k) for (int i = 0; i < cnt[k]; ++i) if (grp[k][i].len > 0) {
            added = 1;
            if ((rc = ctx_add(c, grp[k][i].s, grp[k][i].len, ORACLE_NW, m, n, gp)) < 0) return rc;
        }
        if (!added) { ctx_cons(c, (size_t)dlen + 1); memcpy(c->cons, draft, (size_t)dlen); return dlen; }
        int len = ctx_consensus_custom(c);
        unsigned thr = (unsigned)floorf((float)(unsigned)ni * 0.4f);     /* Window.cpp:28,245 */
        int o = 0;
        for (int i = 0; i < len; ++i) if (c->dst[i] >= thr) c->cons2[o++] = c->cons[i];
        memcpy(c->cons, c->cons2, (size_t)o);
        conslen = o;
    }
    return conslen;
}

/* opt-in "native flavour": reproduce the kLOV row choice of the reference's SIMD engine (see oracle_align) */
void oracle_set_native_klov(int on) { g_native_klov = on ? 1 : 0; }

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_poa_batch(const HypoScoreParams* sp, const HypoWindowBatch* in, HypoConsensusBatch* out,
                     int n_threads, uint64_t* cells_out, uint64_t* aligns_out) {
    if (!sp || !in || !out) return HYPO_E_INVALID;
    if (sp->sr_gap > 0 || sp->lr_gap > 0) return HYPO_E_INVALID;
    uint64_t cells = 0, aligns = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel num_threads(n_threads) reduction(+ : cells, aligns)
    {
        wctx c; memset(&c, 0, sizeof(c));
        char* text = NULL; size_t textcap = 0; sview* views = NULL; size_t viewcap = 0;
#pragma omp for schedule(static, 1)
        for (long long w = 0; w < (long long)in->n_windows; ++w) {
            const HypoWindow* W = &in->windows[w];
            int narm = (int)(W->n_internal + W->n_prefix + W->n_suffix);
            size_t need = (size_t)W->draft_len + 1;
            for (int a = 0; a < narm; ++a) need += in->arm_len[W->first_arm + a];
            if (textcap < need) { textcap = 2 * need; text = (char*)realloc(text, textcap); }
            if (viewcap < (size_t)narm + 1) { viewcap = 2 * ((size_t)narm + 1); views = (sview*)realloc(views, viewcap * sizeof(sview)); }
            char* p = text;
            oracle_unpack4(in->draft4 + W->draft_off, W->draft_len, p);
            const char* draft = p; p += W->draft_len;
            for (int a = 0; a < narm; ++a) {
                uint32_t len = in->arm_len[W->first_arm + a];
                oracle_unpack2(in->arms2 + in->arm_off[W->first_arm + a], len, p);
                views[a].s = p; views[a].len = (int)len; p += len;
            }
            int r = window_consensus(&c, sp, W->type, draft, (int)W->draft_len,
                                     views, (int)W->n_internal,
                                     views + W->n_internal, (int)W->n_prefix,
                                     views + W->n_internal + W->n_prefix, (int)W->n_suffix,
                                     (int)W->n_empty);
            uint64_t cap = out->off[w + 1] - out->off[w];
            if (r < 0) { out->len[w] = 0; out->status[w] = (uint8_t)(-r); }
            else if ((uint64_t)r > cap) { out->len[w] = (uint32_t)r; out->status[w] = HYPO_ST_CONS_OVERFLOW; }
            else { memcpy(out->bases + out->off[w], c.cons, (size_t)r); out->len[w] = (uint32_t)r; out->status[w] = HYPO_ST_OK; }
        }
        cells += c.e.cells; aligns += c.e.aligns;
        ctx_free(&c); free(text); free(views);
    }
    if (cells_out) *cells_out = cells;
    if (aligns_out) *aligns_out = aligns;
    return HYPO_OK;
}

int oracle_replay(int m, int n, int gp, int n_seq, const char* const* seqs, const int* modes,
                  int32_t* pairs_out, int pairs_cap, int* n_pairs,
                  int32_t* rank_out, int rank_cap, int* n_nodes,
                  char* cons_out, int cons_cap, int* cons_len) {
    wctx c; memset(&c, 0, sizeof(c));
    g_reset(&c.g, 1);
    int rc = 0;
    for (int i = 0; i < n_seq && rc == 0; ++i) {
        int L = (int)strlen(seqs[i]);
        eng_align(&c.e, &c.g, seqs[i], L, modes[i], m, n, gp, &c.aln);
        if (i == n_seq - 1 && n_pairs) {
            *n_pairs = c.aln.n;
            for (int k = 0; k < c.aln.n && k < pairs_cap; ++k) { pairs_out[2 * k] = c.aln.d[k].node; pairs_out[2 * k + 1] = c.aln.d[k].pos; }
        }
        rc = g_add_alignment(&c.g, &c.aln, seqs[i], L);
    }
    if (rc == 0) {
        if (n_nodes) *n_nodes = c.g.n;
        for (int k = 0; k < c.g.rank.n && k < rank_cap; ++k) rank_out[k] = c.g.rank.d[k];
        if (c.g.n > 0) {
            int len = ctx_consensus(&c);
            if (cons_len) *cons_len = len;
            for (int k = 0; k < len && k < cons_cap; ++k) cons_out[k] = c.cons[k];
        } else if (cons_len) *cons_len = 0;
    }
    ctx_free(&c);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Solid-kmer scan (src/Contig.cpp:40-74)
 * --------------------------------------------------------------------------------------------- */
static inline unsigned enc4(const uint8_t* p, uint64_t i) { return (p[i >> 1] >> (4 - 4 * (i & 1))) & 15; }

/* The reference's loop (Contig.cpp:46-71) restated once, for the k-mers that START in [beg0, beg1): the rolling state at
 * position i depends only on the last k bases (an N resets it), so a chunk that starts rolling at beg0 sees exactly the
 * k-mers the single loop sees there.  kids == NULL: count only. */
static uint64_t scan_range(const uint8_t* packed4, uint64_t n_bases, uint32_t k, const uint64_t* bits,
                           uint64_t beg0, uint64_t beg1, uint64_t* solid_pos_words, uint64_t* kids, uint64_t kids_at,
                           uint64_t kids_cap) {
    const uint64_t kmask = (1ULL << (2 * k)) - 1;
    uint64_t kmer = 0, cnt = 0; uint32_t klen = 0;
    uint64_t iend = beg1 + k - 1; if (iend > n_bases) iend = n_bases;
    for (uint64_t i = beg0; i < iend; ++i) {
        unsigned b = enc4(packed4, i);
        if (b < 4) { kmer = ((kmer << 2) | b) & kmask; if (klen < k) ++klen; }
        else { klen = 0; kmer = 0; }
        if (klen == k && ((bits[kmer >> 6] >> (kmer & 63)) & 1)) {
            int add = 1;
            if (i < n_bases - 1 && enc4(packed4, i + 1) == b) add = 0;
            uint64_t beg = i + 1 - k;
            if (beg > 0 && enc4(packed4, beg - 1) == enc4(packed4, beg)) add = 0;
            if (add) {
                if (solid_pos_words) solid_pos_words[beg >> 6] |= 1ULL << (beg & 63);
                if (kids && kids_at + cnt < kids_cap) kids[kids_at + cnt] = kmer;
                ++cnt;
            }
        }
    }
    return cnt;
}

int oracle_solid_scan(const uint8_t* packed4, uint64_t n_bases, uint32_t k,
                      const uint64_t* bits,
                      uint64_t* solid_pos_words, uint64_t* kids, uint64_t kids_cap,
                      uint64_t* word_rank, uint64_t* n_solid) {
    if (!packed4 || !bits || !solid_pos_words || k < 2 || k > 31) return HYPO_E_INVALID;
    uint64_t nw = (n_bases + 63) / 64;
    memset(solid_pos_words, 0, nw * 8);
    /* chunks own whole output words; small contigs run as one chunk (= the reference's single loop) */
    const uint64_t chunk_words = 1 << 14;
    const uint64_t nchunk = nw ? (nw + chunk_words - 1) / chunk_words : 0;
    uint64_t* ccnt = (uint64_t*)calloc(nchunk + 1, sizeof(uint64_t));
    if (!ccnt) return HYPO_E_INVALID;
    #pragma omp parallel for schedule(dynamic, 1)
    for (uint64_t c = 0; c < nchunk; ++c) {
        uint64_t b0 = c * chunk_words * 64, b1 = (c + 1) * chunk_words * 64;
        if (b1 > n_bases) b1 = n_bases;
        ccnt[c + 1] = scan_range(packed4, n_bases, k, bits, b0, b1, solid_pos_words, NULL, 0, 0);
    }
    for (uint64_t c = 0; c < nchunk; ++c) ccnt[c + 1] += ccnt[c];
    const uint64_t cnt = ccnt[nchunk];
    if (kids) {
        #pragma omp parallel for schedule(dynamic, 1)
        for (uint64_t c = 0; c < nchunk; ++c) {
            uint64_t b0 = c * chunk_words * 64, b1 = (c + 1) * chunk_words * 64;
            if (b1 > n_bases) b1 = n_bases;
            if (ccnt[c] < kids_cap) scan_range(packed4, n_bases, k, bits, b0, b1, NULL, kids, ccnt[c], kids_cap);
        }
    }
    free(ccnt);
    if (word_rank) {
        uint64_t acc = 0;
        for (uint64_t w = 0; w < nw; ++w) { word_rank[w] = acc; acc += (uint64_t)__builtin_popcountll(solid_pos_words[w]); }
        word_rank[nw] = acc;
    }
    if (n_solid) *n_solid = cnt;
    return HYPO_OK;
}
