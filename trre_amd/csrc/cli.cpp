// cli.cpp — `trre` / `trre_dft` work-alikes for scan mode, on top of the C ABI.
//
// Same command line as the reference (trre_nft.c:728-773, trre_dft.c:1217-1270,
// trre.1:8-28): `trre [-d] [-m] [-a] PATTERN [FILE]`, FILE defaults to stdin,
// errors go to stderr as "error: ..." with exit status 1.  Only scan mode runs
// here — it is the GPU hot path; -m / -a / -d belong to the reference's CPU
// binaries and are refused rather than emulated on the host.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/trre_mi355x.h"

#ifndef TRRE_CLI_ENGINE
#define TRRE_CLI_ENGINE TRRE_ENGINE_NFT
#endif

int main(int argc, char** argv) {
    int opt;
    while ((opt = getopt(argc, argv, "dma")) != -1) {
        switch (opt) {
        case 'd': case 'm': case 'a':
            std::fprintf(stderr, "error: -%c is not part of the GPU scan path; use the reference binary for it\n", opt);
            return EXIT_FAILURE;
        default:
            std::fprintf(stderr, TRRE_CLI_ENGINE == TRRE_ENGINE_NFT ? "Usage: %s [-d] [-m] expr [file]\n"
                                                                     : "Usage: %s [-dma] expr [file]\n", argv[0]);
            return EXIT_FAILURE;
        }
    }
    if (optind >= argc) {
        std::fprintf(stderr, "error: missing trre expression\n");
        return EXIT_FAILURE;
    }
    trre_prog* prog = nullptr;
    if (trre_compile(argv[optind], TRRE_CLI_ENGINE, &prog) != TRRE_OK) {
        std::fprintf(stderr, "%s\n", trre_last_error());
        return EXIT_FAILURE;
    }
    FILE* fp = stdin;
    if (optind == argc - 2) {
        fp = std::fopen(argv[optind + 1], "rb");
        if (!fp) {
            std::fprintf(stderr, "error: can not open file %s\n", argv[optind + 1]);
            return EXIT_FAILURE;
        }
    }
    std::vector<uint8_t> in;
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = std::fread(buf, 1, sizeof buf, fp)) > 0) in.insert(in.end(), buf, buf + k);
    std::vector<uint8_t> out(in.size() + 64);
    size_t m = 0;
    int rc = trre_scan_host(prog, in.data(), in.size(), out.data(), out.size(), &m, 0);
    if (rc == TRRE_E_CAPACITY) {
        out.resize(m + 64);
        rc = trre_scan_host(prog, in.data(), in.size(), out.data(), out.size(), &m, 0);
    }
    if (rc != TRRE_OK) {
        std::fprintf(stderr, "%s\n", trre_last_error());
        return EXIT_FAILURE;
    }
    std::fwrite(out.data(), 1, m, stdout);
    trre_free(prog);
    return 0;
}
