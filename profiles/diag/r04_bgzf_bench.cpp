#include "../../hypo_amd/csrc/host/SeqIO.hpp"
#include <chrono>
#include <omp.h>
int main(int argc, char** argv) {
    hypo::SamReader sf(argv[1]);
    const int nt = atoi(argv[2]);
    sf.set_inflate_threads(nt);
    hypo::SamReader::RecordBlock b;
    size_t n = 0, bytes = 0; int blocks = 0; uint64_t sig = 1469598103934665603ull;
    auto t0 = std::chrono::steady_clock::now();
    double tcut = 0;
    while (blocks < 100000) {
        bool more = sf.read_block(b, 1 << 19, (size_t)128 << 20);
        n += b.n(); for (size_t i = 0; i < b.n(); i += 16) bytes += 16 * (b.len(i) + 4);
        for (size_t i = 0; i < b.n(); ++i) sig = (sig ^ (uint64_t)b.len(i) ^ ((uint64_t)(unsigned char)b.rec(i)[32] << 32)) * 1099511628211ull;
        ++blocks;
        if (!more && b.n() == 0) break;
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%zu records in %.2f s = %.1f M rec/s (%d blocks) ~%.2f GB/s inflated, signature %016llx\n", n, dt, n / dt / 1e6, blocks, (double)bytes / dt / 1e9, (unsigned long long)sig);
}
