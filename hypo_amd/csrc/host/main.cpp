// main.cpp — command line of the MI355X build: the reference's flags and defaults (src/main.cpp:46-67,100-113,
// 124-358) plus the opt-in flags --device / --gpus / --devices.  usage() keeps the reference's layout, the wording is this build's own.
#include <getopt.h>
#include <unistd.h>
#include <malloc.h>
#include <sys/stat.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include "Hypo.hpp"

namespace {

const struct option long_options[] = {
    {"reads-short", required_argument, nullptr, 'r'}, {"draft", required_argument, nullptr, 'd'},
    {"size-ref", required_argument, nullptr, 's'}, {"coverage-short", required_argument, nullptr, 'c'},
    {"bam-sr", required_argument, nullptr, 'b'}, {"bam-lr", required_argument, nullptr, 'B'},
    {"output", required_argument, nullptr, 'o'}, {"threads", required_argument, nullptr, 't'},
    {"processing-size", required_argument, nullptr, 'p'}, {"kind-sr", required_argument, nullptr, 'k'},
    {"match-sr", required_argument, nullptr, 'm'}, {"mismatch-sr", required_argument, nullptr, 'x'},
    {"gap-sr", required_argument, nullptr, 'g'}, {"match-lr", required_argument, nullptr, 'M'},
    {"mismatch-lr", required_argument, nullptr, 'X'}, {"gap-lr", required_argument, nullptr, 'G'},
    {"qual-map-th", required_argument, nullptr, 'q'}, {"ned-th", required_argument, nullptr, 'n'},
    {"intermed", no_argument, nullptr, 'i'}, {"help", no_argument, nullptr, 'h'},
    {"device", required_argument, nullptr, 1000}, {"gpus", required_argument, nullptr, 1001},
    {"devices", required_argument, nullptr, 1002}, {"native-klov", no_argument, nullptr, 1003},
    {"ccs-windows", no_argument, nullptr, 1004}, {"host-arms", no_argument, nullptr, 1005}, {"require-device", no_argument, nullptr, 1006}, {nullptr, 0, nullptr, 0}};

// Same layout as the reference's usage() (src/main.cpp:363-430): "Usage: hypo <args>", the mandatory block, the optional
// block, every flag as "-x, --long <type>" followed by what it does and its default.  The wording is this build's own.
struct HelpEntry { const char* flag; const char* what; const char* dflt; };
void usage() {
    static const HelpEntry mandatory[] = {
        {"-r, --reads-short <str>", "Short reads (fasta/fastq, plain or compressed), or @file holding one file name per line. Only its existence is checked: solid k-mers come from aux/solid_kmers.bvsd (see -i).", nullptr},
        {"-d, --draft <str>", "Draft contigs to polish (fasta/fastq, plain or compressed).", nullptr},
        {"-b, --bam-sr <str>", "Short reads aligned to the draft (bam, or sam plain/gzip; CIGAR required), sorted by coordinate.", nullptr},
        {"-c, --coverage-short <int>", "Approximate mean coverage of the short reads.", nullptr},
        {"-s, --size-ref <str>", "Approximate genome size: a number, optionally followed by k/m/g (10m, 2.3g). It fixes the solid k-mer length.", nullptr}};
    static const HelpEntry optional[] = {
        {"-B, --bam-lr <str>", "Long reads aligned to the draft (bam/sam with CIGAR).", "short-read polishing only"},
        {"-o, --output <str>", "Output file.", "hypo_<draft_file_name>.fasta in the working directory"},
        {"-t, --threads <int>", "Host threads.", "1"},
        {"-p, --processing-size <int>", "Contigs per batch; fewer contigs per batch need less memory.", "all contigs of the draft"},
        {"-k, --kind-sr <str>", "Kind of short reads: sr (Illumina-like) or ccs (HiFi). Accepted and, as in the reference (src/main.cpp:312), without effect.", "sr"},
        {"-m, --match-sr <int>", "Match score, short reads.", "5"},
        {"-x, --mismatch-sr <int>", "Mismatch score, short reads.", "-4"},
        {"-g, --gap-sr <int>", "Gap penalty, short reads (negative).", "-8"},
        {"-M, --match-lr <int>", "Match score, long reads.", "3"},
        {"-X, --mismatch-lr <int>", "Mismatch score, long reads.", "-5"},
        {"-G, --gap-lr <int>", "Gap penalty, long reads (negative).", "-4"},
        {"-n, --ned-th <int>", "Largest normalised edit distance (in %) of a long arm that may enter a window.", "20"},
        {"-q, --qual-map-th <int>", "Reads mapped with a quality below this are ignored.", "2"},
        {"-i, --intermed", "Keep and reuse intermediate files (aux/solid_kmers.bvsd). Needed here: k-mer counting (KMC) is not part of this build.", nullptr},
        {"    --device <int>", "[MI355X build] HIP device to run on.", "0"},
        {"    --gpus <int>", "[MI355X build] Use devices 0..N-1, one context each: the contigs of a batch are dealt out to them (a contig larger than its share is cut into coordinate ranges), every context votes, cuts arms and polishes its own resident windows; windows that exist as host objects are sharded with an RCCL gather.", "1"},
        {"    --devices <list>", "[MI355X build] Comma-separated HIP device ids instead of --device / --gpus.", nullptr},
        {"    --native-klov", "[MI355X build] Choose the end row of prefix arms like the AVX2 / SSE4.1 alignment engine of a -march=native build of the reference does (the default follows the scalar engine of the reference's default build; the two differ on noisy prefix arms).", "off"},
        {"    --ccs-windows", "[MI355X build] Cut windows with the sizes -k ccs was meant to select (ideal length 500, search threshold 400; the reference parses -k but never applies it).", "off"},
        {"    --host-arms", "[MI355X build] Cut the short reads into arms on the host (the reference's loops) instead of on the device.", "off"},
        {"    --require-device", "[MI355X build] Exit with an error, instead of an Info line and the host loops, when a stage the device should run (support votes, arm selection) cannot run there (also: HYPO_REQUIRE_DEVICE=1).", "off"},
        {"-h, --help", "Print the usage.", nullptr}};
    std::printf("\n Usage: hypo <args>\n\n ** Mandatory args:\n");
    for (const auto& e : mandatory) std::printf("\t%s\n\t%s\n\n", e.flag, e.what);
    std::printf("\n ** Optional args:\n");
    for (const auto& e : optional) {
        std::printf("\t%s\n\t%s\n", e.flag, e.what);
        if (e.dflt) std::printf("\t[Default] %s.\n", e.dflt);
        std::printf("\n");
    }
}

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// src/main.cpp:490-528: smallest odd k with 4^k >= genome size
unsigned get_kmer_len(const std::string& given) {
    size_t ind = 0;
    const float val = std::stof(given, &ind);
    unsigned power = 0;
    if (ind >= given.size()) {
        if (std::floor(val) != std::ceil(val)) { std::fprintf(stderr, "[Hypo::Utils] Error: Wrong format for genome-size: Genome-size with no units (K,M,G etc,) should be absolute number!\n"); std::exit(1); }
    } else {
        switch (std::toupper(given[ind])) {
            case 'K': power = 10; break; case 'M': power = 20; break; case 'G': power = 30; break; case 'T': power = 40; break;
            default: std::fprintf(stderr, "[Hypo::Utils] Error: Wrong format for genome-size: Allowed units for Genome-size are K (10^3),M (10^6),G (10^9),T (10^12)!\n"); std::exit(1);
        }
    }
    unsigned k = (unsigned)((double)power + std::ceil(std::log2(val)));
    k = (unsigned)std::ceil(k / 2);                 // integer division first, as in the reference
    if (k % 2 == 0) ++k;
    std::fprintf(stdout, "[Hypo::Utils] Info: Value of K chosen for the given genome size (%s): %u\n", given.c_str(), k);
    return k;
}

}  // namespace

int main(int argc, char** argv) {
    // The host stages allocate millions of small objects (alignments, arms) from all threads.  glibc grows a thread arena in
    // small mprotect steps and trims it eagerly; with > 32 threads those system calls serialise on the address-space lock
    // (measured: -t 64 took 2.1 s instead of 1.05 s on the 5 Mbp set).  Grow in 64 MB steps, give memory back late.
    // the device library runs up to seven HIP streams; a ROCm process gets four hardware queues unless told otherwise before the
    // runtime initialises (two streams sharing a queue serialise: DESIGN.md 3.1)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    // The OpenMP workers must sleep between parallel regions: the run alternates short parallel phases with serial stretches,
    // device calls and helper threads (record reader, releasers, the HIP runtime's own), and libgomp's default lets idle workers
    // spin — measured on the 100 x 1 Mbp set, -t 64: 8.98 s spinning, 4.77 s sleeping (profiles/diag/r03_c3_threads.sh).  libgomp
    // reads the policy when it is loaded, i.e. before main(): set it and start over once.
    // HYPO_NO_REEXEC=1 (or an OMP_WAIT_POLICY of the caller's own) opts out; a traced process (debugger, strace, profilers that
    // attach) is never re-executed either — it would lose its tracer's state.
    auto traced = [] {
        std::ifstream st("/proc/self/status");
        std::string key; long v = 0;
        while (st >> key) { if (key == "TracerPid:") { st >> v; return v != 0; } st.ignore(4096, '\n'); }
        return false;
    };
    if (!getenv("OMP_WAIT_POLICY") && !getenv("HYPO_NO_REEXEC") && !traced()) {
        setenv("OMP_WAIT_POLICY", "passive", 1);
        setenv("HYPO_NO_REEXEC", "1", 1);
        execv("/proc/self/exe", argv);                          // (if this fails the run goes on with the default policy)
    }
    mallopt(M_TOP_PAD, 64 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    hypo::InputFlags flags;
    bool is_sr = false, is_draft = false, is_size = false, is_cov = false, is_bamsr = false;
    std::string given_sz, kind = "sr";
    int opt;
    auto need_file = [](const char* what, const std::string& p) {
        if (!file_exists(p)) { std::fprintf(stderr, "[Hypo::] Error: File Error: %s file does not exist %s!\n", what, p.c_str()); std::exit(1); }
    };
    while ((opt = getopt_long(argc, argv, "r:d:s:c:b:B:o:t:p:k:m:x:g:M:X:G:q:n:ih", long_options, nullptr)) != -1) {
        switch (opt) {
            case 'r': {
                std::string in = optarg;
                if (!in.empty() && in[0] == '@') {
                    std::ifstream f(in.substr(1));
                    if (!f) { std::fprintf(stderr, "[Hypo::] Error: File Error: Reads file-list does not exist %s!\n", in.c_str()); std::exit(1); }
                    std::string name;
                    while (std::getline(f, name)) if (!name.empty()) { need_file("Reads", name); flags.sr_filenames.push_back(name); }
                } else { need_file("Reads", in); flags.sr_filenames.push_back(in); }
                is_sr = true; break;
            }
            case 'd': flags.draft_filename = optarg; need_file("Draft", flags.draft_filename); is_draft = true; break;
            case 's': given_sz = optarg; flags.k = std::max(2u, get_kmer_len(given_sz)); is_size = true; break;
            case 'c': flags.cov = (uint32_t)std::atoi(optarg); if (flags.cov == 0) { std::fprintf(stderr, "[Hypo::] Error: Arg Error: Coverage should be a positive integer %s!\n", optarg); std::exit(1); } is_cov = true; break;
            case 'b': flags.sr_bam_filename = optarg; need_file("Short reads BAM", flags.sr_bam_filename); is_bamsr = true; break;
            case 'B': flags.lr_bam_filename = optarg; need_file("Long reads BAM", flags.lr_bam_filename); break;
            case 'o': flags.output_filename = optarg; break;
            case 't': flags.threads = (uint32_t)std::max(1, std::atoi(optarg)); break;
            case 'p': flags.processing_batch_size = (uint32_t)std::max(0, std::atoi(optarg)); break;
            case 'k': kind = optarg; break;                               // parsed, never applied (src/main.cpp:312)
            case 'm': flags.score_params.sr_match = (int8_t)std::atoi(optarg); break;
            case 'x': flags.score_params.sr_mismatch = (int8_t)std::atoi(optarg); break;
            case 'g': flags.score_params.sr_gap = (int8_t)std::atoi(optarg);
                      if (flags.score_params.sr_gap >= 0) { std::fprintf(stderr, "[Hypo::] Error: Arg Error: Gap penalty must be negative %s!\n", optarg); std::exit(1); } break;
            case 'M': flags.score_params.lr_match = (int8_t)std::atoi(optarg); break;
            case 'X': flags.score_params.lr_mismatch = (int8_t)std::atoi(optarg); break;
            case 'G': flags.score_params.lr_gap = (int8_t)std::atoi(optarg);
                      if (flags.score_params.lr_gap >= 0) { std::fprintf(stderr, "[Hypo::] Error: Arg Error: Gap penalty must be negative %s!\n", optarg); std::exit(1); } break;
            case 'q': flags.map_qual_th = (uint32_t)std::max(0, std::atoi(optarg)); break;
            case 'n': flags.norm_edit_th = (uint32_t)std::max(0, std::atoi(optarg)); break;
            case 'i': flags.intermed = true; break;
            case 1000: flags.device = std::atoi(optarg); break;
            case 1001: flags.gpus = std::max(1, std::atoi(optarg)); break;
            case 1003: flags.native_klov = true; break;
            case 1004: flags.ccs_windows = true; break;
            case 1005: flags.host_arms = true; break;
            case 1006: flags.require_device = true; break;
            case 1002: {
                flags.devices.clear();
                for (const char* c = optarg; *c;) { flags.devices.push_back(std::atoi(c)); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
                break;
            }
            default: usage(); return 0;                                    // -h and unknown options alike (src/main.cpp:302-304)
        }
    }
    if (!(is_sr && is_draft && is_size && is_bamsr && is_cov)) {
        std::fprintf(stderr, "[Hypo::] Error: Invalid command: Too few arguments!\n");
        usage();
        return 1;
    }
    if (flags.output_filename.empty()) {                                  // src/main.cpp:317-323
        const size_t s = flags.draft_filename.find_last_of("(/\\");
        std::string full = flags.draft_filename.substr(s == std::string::npos ? 0 : s + 1);
        flags.output_filename = "hypo_" + full.substr(0, full.find_last_of(".")) + ".fasta";
    }
    mkdir(HYPO_AUX_DIR, 0777);
    flags.done_stage = 0;
    if (flags.intermed && file_exists(HYPO_STAGEFILE)) {                  // last stage number of aux/stage.txt (src/main.cpp:327-345)
        std::ifstream ifs(HYPO_STAGEFILE);
        std::string a, b, c; unsigned st = 0;
        while (ifs >> a >> b >> c >> st) {}
        flags.done_stage = st;
        std::cout << "Stagenum found is " << st << std::endl;
    }
    std::fprintf(stdout, "[Hypo::Utils] Info: Beginning from stage: %u\n", flags.done_stage);
    const bool timing = std::getenv("HYPO_HOST_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (flags.devices.empty()) {                                           // --device D, or --gpus N = devices 0..N-1
        if (flags.gpus > 1) for (int d = 0; d < flags.gpus; ++d) flags.devices.push_back(d);
        else flags.devices.push_back(flags.device);
    }
    // HYPO_GIANT_ARENA_MB: HBM per device context for the windows beyond the table-driven size classes (size class 6; default 1024, 0 = none:
    // such windows then keep their draft with a warning, as before round 6)
    if (const char* ga = std::getenv("HYPO_GIANT_ARENA_MB")) {
        if (hypo_gpu_set_option("giant_arena_mb", std::atoi(ga)) != HYPO_OK) { std::fprintf(stderr, "[Hypo::] Error: %s\n", hypo_gpu_last_error()); return 1; }
    }
    if (hypo_gpu_init(flags.devices.data(), (int)flags.devices.size()) != HYPO_OK) { std::fprintf(stderr, "[Hypo::] Error: %s\n", hypo_gpu_last_error()); return 1; }
    if (flags.native_klov && hypo_gpu_set_option("native_klov", 1) != HYPO_OK) { std::fprintf(stderr, "[Hypo::] Error: %s\n", hypo_gpu_last_error()); return 1; }
    if (flags.ccs_windows) { hypo::Window_settings.ideal_swind_size = 500; hypo::Window_settings.wind_size_search_th = 400; }   // src/main.cpp:577-580
    if (timing) std::fprintf(stderr, "[timing] main: device initialised at %.3f s\n", since());
    {
        hypo::Hypo h(flags);
        if (const char* d = std::getenv("HYPO_REGION_DUMP")) h.set_region_dump(d);
        h.polish();
        if (timing) std::fprintf(stderr, "[timing] main: polish() returned at %.3f s\n", since());
    }
    if (timing) std::fprintf(stderr, "[timing] main: pipeline objects released at %.3f s\n", since());
    return 0;
}
