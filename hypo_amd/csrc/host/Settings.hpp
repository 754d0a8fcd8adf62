// Settings.hpp — the reference's compile-time constants and flag structs (include/globalDefs.hpp:58-156,
// src/main.cpp:85-88) restated; same names so that the pipeline code reads like the reference's.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../../include/hypo_gpu.h"

namespace hypo {

using ScoreParams = HypoScoreParams;       // include/globalDefs.hpp:58-66

struct InputFlags {                        // include/globalDefs.hpp:68-87
    std::vector<std::string> sr_filenames;
    std::string sr_bam_filename, draft_filename, lr_bam_filename, output_filename;
    uint32_t k = 0, threads = 1, processing_batch_size = 0, map_qual_th = 2, norm_edit_th = 20, cov = 0, sz_in_gb = 0;
    ScoreParams score_params{5, -4, -8, 3, -5, -4};
    uint32_t done_stage = 0;
    bool intermed = false;
    int device = 0;                        // new, opt-in: --device
    int gpus = 1;                          // new, opt-in: --gpus N (devices 0..N-1)
    std::vector<int> devices;              // new, opt-in: --devices a,b,c
    bool native_klov = false;              // new, opt-in: --native-klov (match a -march=native build of the reference)
    bool require_device = false;           // new, opt-in: --require-device (a stage the device should run must not quietly fall back to the host loops: error instead)
    bool host_arms = false;                // new, opt-in: --host-arms (cut the short reads into arms on the host, not on the device)
    bool ccs_windows = false;              // new, opt-in: --ccs-windows (the window sizes -k ccs was meant to select)
};

enum class RegionType : uint8_t { SWS, SW, WS, MWM, MW, WM, SWM, MWS, OTHER, LONG, SR, MSR };   // globalDefs.hpp:95-108
inline const char* region_name(RegionType t) {
    static const char* n[] = {"SWS", "SW", "WS", "MWM", "MW", "WM", "SWM", "MWS", "OTH", "LNG", "SR", "MSR"};
    return n[(int)t];
}

#define HYPO_AUX_DIR "aux/"
#define HYPO_SKFILE "aux/solid_kmers.bvsd"
#define HYPO_STAGEFILE "aux/stage.txt"

struct SrSettings { uint32_t cov_th = 5; double supp_frac = 0.4; };
struct MinimizerSettings { uint32_t k = 10, w = 10, cov_th = 5; double supp_frac = 0.8;
                           uint32_t polyA = 0x000000u, polyC = 0x055555u, polyG = 0x0aaaaau, polyT = 0x0fffffu; };
struct WindowSettings { uint32_t ideal_swind_size = 100, ideal_lwind_size = 500, wind_size_search_th = 80; };
struct ArmsSettings { uint32_t min_short_num = 3, min_internal_num1 = 20, min_internal_num2 = 5, min_internal_num3 = 10,
                      min_contrib = 10; double min_internal_contrib = 0.4; uint32_t short_arm_coef = 10; };
constexpr uint32_t kMinimizerRingCap = 32;        // monotone-queue slots of the minimizer scans (need w + 1)
static const SrSettings Sr_settings;
static const MinimizerSettings Minimizer_settings;
// `-k ccs` never changes it in the reference (src/main.cpp:312 is a declaration, not a call); the opt-in --ccs-windows applies the
// sizes set_kind("ccs") was meant to set (src/main.cpp:572-585)
inline WindowSettings Window_settings;
static const ArmsSettings Arms_settings;

}  // namespace hypo
