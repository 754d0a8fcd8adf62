// main.cpp — command line of the MI355X build: the reference's flags and defaults (src/main.cpp:46-67,100-113,
// 124-358) plus one opt-in flag, --device.  Help text is this build's own.
#include <getopt.h>
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
    {"device", required_argument, nullptr, 1000}, {nullptr, 0, nullptr, 0}};

void usage() {
    std::puts(
        "hypo (MI355X build): polishes draft contigs with short reads (and optionally long reads)\n"
        "usage: hypo -r <reads|@list> -d <draft.fa> -b <short-reads.sam> -c <coverage> -s <genome size[k|m|g|t]> [options]\n"
        "  mandatory\n"
        "    -r, --reads-short FILE     short reads file or @file-of-names (only checked for existence here)\n"
        "    -d, --draft FILE           draft contigs, FASTA/FASTQ, plain or gzip\n"
        "    -b, --bam-sr FILE          short reads mapped to the draft, coordinate sorted (BAM, or SAM text plain/gzip)\n"
        "    -c, --coverage-short INT   approximate coverage of the short reads\n"
        "    -s, --size-ref STR         approximate genome size: number with unit k/m/g/t; fixes the solid k-mer length\n"
        "  optional\n"
        "    -B, --bam-lr FILE          long reads mapped to the draft (BAM or SAM)       [none]\n"
        "    -o, --output FILE          polished contigs                                  [hypo_<draft>.fasta]\n"
        "    -t, --threads INT          host threads                                      [1]\n"
        "    -p, --processing-size INT  contigs per batch, 0 = all                        [0]\n"
        "    -k, --kind-sr STR          sr | ccs (accepted and, as in the reference, without effect) [sr]\n"
        "    -m/-x/-g INT               short-read match / mismatch / gap (g < 0)         [5 / -4 / -8]\n"
        "    -M/-X/-G INT               long-read match / mismatch / gap (G < 0)          [3 / -5 / -4]\n"
        "    -n, --ned-th INT           long reads: max normalised edit distance          [20]\n"
        "    -q, --qual-map-th INT      minimum mapping quality                           [2]\n"
        "    -i, --intermed             keep / reuse aux/solid_kmers.bvsd (required here: k-mer counting is not built in)\n"
        "        --device INT           HIP device                                        [0]\n"
        "    -h, --help");
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
    if (hypo_gpu_init(&flags.device, 1) != HYPO_OK) { std::fprintf(stderr, "[Hypo::] Error: %s\n", hypo_gpu_last_error()); return 1; }
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
