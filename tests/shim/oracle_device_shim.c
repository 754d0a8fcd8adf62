/* oracle_device_shim.c — TEST INFRASTRUCTURE, never shipped, never on the product's library path.
 *
 * A stand-in for libhypo_gpu.so that answers the two data-path entry points with the CPU oracle, so that the
 * host pipeline above the C-ABI (hypo_amd/csrc/host: SAM parsing, support counting, segmentation, arms, FASTA
 * reassembly) can be exercised end to end in the GPU-less CI container against the reference's golden outputs.
 * Only tests/test_host_e2e_cpu.py puts this directory on LD_LIBRARY_PATH; the `-m gpu` e2e test runs the same
 * binary against the real library.  What it proves is the HOST logic; it says nothing about the HIP kernels.
 */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/hypo_gpu.h"
#include "../../oracle/hypo_oracle.h"

static int g_ready = 0;
/* arm_off == NULL (arms back to back): the oracle wants the offsets spelled out */
static int with_offsets(const HypoScoreParams* scores, const HypoWindowBatch* in, HypoConsensusBatch* out) {
    if (in->arm_off || !in->n_arms) return oracle_poa_batch(scores, in, out, 0, NULL, NULL);
    uint64_t* ao = (uint64_t*)malloc((size_t)in->n_arms * 8);
    uint64_t acc = 0;
    for (uint32_t a = 0; a < in->n_arms; ++a) { ao[a] = acc; acc += ((uint64_t)in->arm_len[a] + 3) / 4; }
    HypoWindowBatch b = *in;
    b.arm_off = ao;
    const int rc = oracle_poa_batch(scores, &b, out, 0, NULL, NULL);
    free(ao);
    return rc;
}
static const uint64_t* g_set = 0; static uint32_t g_set_k = 0; static int g_ndev = 1;
int hypo_gpu_init(const int* device_ids, int n_devices) { (void)device_ids; g_ndev = n_devices > 0 ? n_devices : 1; g_ready = 1; fprintf(stderr, "[oracle_device_shim] TEST SHIM in use: CPU oracle behind the C-ABI\n"); return HYPO_OK; }
int hypo_gpu_shutdown(void) { g_ready = 0; return HYPO_OK; }
int hypo_gpu_abi_version(void) { return HYPO_GPU_ABI_VERSION; }
const char* hypo_gpu_last_error(void) { return "oracle_device_shim"; }
int hypo_gpu_num_cus(void) { return 0; }

int hypo_gpu_solid_scan(const uint8_t* packed4, uint64_t n_bases, uint32_t k, const uint64_t* bitset_words,
                        uint64_t* solid_pos_words, uint64_t* kids, uint64_t kids_cap, uint64_t* word_rank, uint64_t* n_solid) {
    if (!g_ready) return HYPO_E_NOTINIT;
    if (!bitset_words) { if (g_set_k != k) return HYPO_E_INVALID; bitset_words = g_set; }
    return oracle_solid_scan(packed4, n_bases, k, bitset_words, solid_pos_words, kids, kids_cap, word_rank, n_solid);
}

/* the caller's buffer outlives the scans of a run (host/Hypo.cpp keeps the SolidKmers object alive) */
int hypo_gpu_solid_set_upload(const uint64_t* bitset_words, uint32_t k) { g_set = bitset_words; g_set_k = k; return HYPO_OK; }
int hypo_gpu_set_option(const char* name, int value) { if (name && !strcmp(name, "native_klov")) { oracle_set_native_klov(value); return HYPO_OK; } return HYPO_E_INVALID; }
int hypo_gpu_num_devices(void) { return g_ready ? g_ndev : 0; }
int hypo_gpu_use_device(int slot) { return slot >= 0 && slot < g_ndev ? HYPO_OK : HYPO_E_INVALID; }
const char* hypo_gpu_build_id(void) { return "oracle_device_shim"; }
/* arm selection has no CPU restatement behind the boundary: the host runs its own (reference) loops (host/DeviceArms.cpp) */
int hypo_gpu_arms_build(const HypoArmsRegions* r, const HypoArmsReads* a, uint8_t* v, HypoArmsSummary* s) { (void)r; (void)a; (void)v; (void)s; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_arms_download(HypoWindow* w, uint32_t* r, uint32_t* l, uint64_t* o, uint8_t* a, uint8_t* d) { (void)w; (void)r; (void)l; (void)o; (void)a; (void)d; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_arms_poa(const HypoScoreParams* sc, char* b, uint64_t* o, uint32_t* l, uint8_t* st) { (void)sc; (void)b; (void)o; (void)l; (void)st; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_reads_upload(const HypoArmsReads* a, const uint32_t* c, uint64_t t) { (void)a; (void)c; (void)t; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_support_kmers(uint32_t k, uint64_t n, const uint32_t* sp, const uint64_t* kd, uint32_t* cv, uint32_t* su) { (void)k; (void)n; (void)sp; (void)kd; (void)cv; (void)su; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_support_minimizers(const HypoMegaWindows* w, uint32_t* cv, uint32_t* su) { (void)w; (void)cv; (void)su; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_solid_scan_keep(uint32_t h, const uint8_t* p, uint64_t n, uint32_t k, uint64_t* w, uint64_t* r, uint64_t* ns) { (void)h; (void)p; (void)n; (void)k; (void)w; (void)r; (void)ns; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_solid_release(uint32_t h) { (void)h; return HYPO_OK; }
int hypo_gpu_support_kmers_kept(uint32_t k, uint32_t n, const uint32_t* h, const uint32_t* b, uint32_t* cv, uint32_t* su, uint64_t* t) { (void)k; (void)n; (void)h; (void)b; (void)cv; (void)su; (void)t; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_host_alloc(size_t bytes, void** out) { *out = malloc(bytes ? bytes : 16); return *out ? HYPO_OK : HYPO_E_HIP; }
int hypo_gpu_host_free(void* p) { free(p); return HYPO_OK; }
int hypo_gpu_host_register(void* p, size_t bytes) { (void)p; (void)bytes; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_host_unregister(void* p) { (void)p; return HYPO_OK; }
int hypo_gpu_poa_last_stats(HypoPoaStats* out) { (void)out; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_arms_build_long(const HypoArmsRegions* r, const HypoArmsReads* a, uint8_t* v, HypoArmsSummary* s) { (void)r; (void)a; (void)v; (void)s; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_arms_download_long(HypoWindow* w, uint32_t* r, uint32_t* l, uint64_t* o, uint8_t* a, uint8_t* d) { (void)w; (void)r; (void)l; (void)o; (void)a; (void)d; return HYPO_E_UNSUPPORTED; }
int hypo_gpu_arms_poa_long(const HypoScoreParams* sc, char* b, uint64_t* o, uint32_t* l, uint8_t* st) { (void)sc; (void)b; (void)o; (void)l; (void)st; return HYPO_E_UNSUPPORTED; }

int hypo_gpu_poa_slot_layout(const HypoWindowBatch* in, uint64_t* off) {
    uint64_t acc = 0;
    for (uint32_t w = 0; w < in->n_windows; ++w) {
        const HypoWindow* W = &in->windows[w];
        uint64_t longest = W->draft_len;
        const uint32_t narm = W->n_internal + W->n_prefix + W->n_suffix;
        for (uint32_t a = 0; a < narm; ++a) if (in->arm_len[W->first_arm + a] > longest) longest = in->arm_len[W->first_arm + a];
        off[w] = acc;
        acc += (longest + longest / 2 + 24 + 7) / 8 * 8;
    }
    off[in->n_windows] = acc;
    return HYPO_OK;
}

int hypo_gpu_poa_batch(const HypoScoreParams* scores, const HypoWindowBatch* in, HypoConsensusBatch* out) {
    if (!g_ready) return HYPO_E_NOTINIT;
    return with_offsets(scores, in, out);
}
static HypoConsensusBatch g_pending_out; static int g_pending = 0;
int hypo_gpu_poa_batch_begin(const HypoScoreParams* scores, const HypoWindowBatch* in, HypoConsensusBatch* out, int* ticket) {
    (void)g_pending_out; if (!g_ready) return HYPO_E_NOTINIT; if (g_pending) return HYPO_E_INVALID; g_pending = 1; *ticket = 0;
    return with_offsets(scores, in, out);
}
int hypo_gpu_poa_batch_end(int ticket) { if (ticket != 0 || !g_pending) return HYPO_E_INVALID; g_pending = 0; return HYPO_OK; }
/* two "devices": the window list in two contiguous halves, each answered separately into the caller's slots */
int hypo_gpu_poa_batch_sharded(const HypoScoreParams* scores, const HypoWindowBatch* in, HypoConsensusBatch* out) {
    if (!g_ready) return HYPO_E_NOTINIT;
    if (g_ndev < 2 || in->n_windows < 2 || !in->arm_off) return with_offsets(scores, in, out);
    fprintf(stderr, "[oracle_device_shim] sharded call over %d contexts\n", g_ndev);
    const uint32_t h = in->n_windows / 2;
    HypoWindowBatch a = *in, b = *in;
    a.n_windows = h;
    b.n_windows = in->n_windows - h; b.windows = in->windows + h;
    HypoConsensusBatch oa = *out, ob = *out;
    ob.off = out->off + h; ob.len = out->len + h; ob.status = out->status + h;
    int rc = oracle_poa_batch(scores, &a, &oa, 0, NULL, NULL);
    if (rc) return rc;
    return oracle_poa_batch(scores, &b, &ob, 0, NULL, NULL);
}
