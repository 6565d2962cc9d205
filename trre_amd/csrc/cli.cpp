// cli.cpp — `trre` / `trre_dft` work-alikes for scan mode, on top of the C ABI.
//
// Same command line as the reference (trre_nft.c:728-773, trre_dft.c:1217-1270,
// trre.1:8-28): `trre [-d] [-m] [-a] PATTERN [FILE]`, FILE defaults to stdin,
// errors go to stderr as "error: ..." with exit status 1.  Scan mode (the GPU
// hot path) and, for the NFT engine, `-m` (whole-line match, first output: trre_nft.c:791-797) and the
// generator modes `-a` / `-ma` (every accepting path prints: trre_nft.c:640-641,647-648) run here; -d
// (Graphviz dumps) and trre_dft's -m (which only prints empty lines, trre_dft.c:1185-1190) belong to the
// reference's CPU binaries and are refused rather than emulated; trre_dft -a answers "Not supported
// yet" like the reference (trre_dft.c:1227-1229).
//
// Like the reference's getline loop the input is streamed: it is read in blocks
// of up to 256 MiB, every block is cut after its last '\n' (the rest is carried
// into the next block), scanned line-sharded on all visible GPUs
// (trre_scan_host_multi) and written out, so neither the input nor the output
// has to fit in host memory.  TRRE_DEVICES=<mask> restricts the GPUs used.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/trre_mi355x.h"

#ifndef TRRE_CLI_ENGINE
#define TRRE_CLI_ENGINE TRRE_ENGINE_NFT
#endif

int main(int argc, char** argv) {
    int opt;
    bool match = false, all = false;
    while ((opt = getopt(argc, argv, "dma")) != -1) {
        switch (opt) {
        case 'm':
            if (TRRE_CLI_ENGINE == TRRE_ENGINE_NFT) { match = true; break; }
            std::fprintf(stderr, "error: -m is not part of the GPU scan path; use the reference binary for it\n");
            return EXIT_FAILURE;
        case 'a':
            if (TRRE_CLI_ENGINE == TRRE_ENGINE_NFT) { all = true; break; }
            std::fprintf(stderr, "Not supported yet\n");               // trre_dft.c:1227-1229
            return EXIT_FAILURE;
        case 'd':
            std::fprintf(stderr, "error: -d is not part of the GPU scan path; use the reference binary for it\n");
            return EXIT_FAILURE;
        default:
            std::fprintf(stderr, TRRE_CLI_ENGINE == TRRE_ENGINE_NFT ? "Usage: %s [-d] [-m] expr [file]\n"
                                                                     : "Usage: %s [-dma] expr [file]\n", argv[0]);
            return EXIT_FAILURE;
        }
    }
    const int mode = all ? (match ? TRRE_MODE_MATCH_ALL : TRRE_MODE_SCAN_ALL) : (match ? TRRE_MODE_MATCH : TRRE_MODE_SCAN);
    if (optind >= argc) {
        std::fprintf(stderr, "error: missing trre expression\n");
        return EXIT_FAILURE;
    }
    trre_prog* prog = nullptr;
    if (trre_compile_mode(reinterpret_cast<const uint8_t*>(argv[optind]), std::strlen(argv[optind]), TRRE_CLI_ENGINE, mode, &prog) != TRRE_OK) {
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
    const uint32_t mask = std::getenv("TRRE_DEVICES") ? (uint32_t)std::strtoul(std::getenv("TRRE_DEVICES"), nullptr, 0) : 0u;
    const size_t block = std::getenv("TRRE_CLI_BLOCK") ? (size_t)std::strtoull(std::getenv("TRRE_CLI_BLOCK"), nullptr, 0) : (size_t)256 << 20;
    std::vector<uint8_t> in, out;
    size_t have = 0;                   // bytes of `in` that are filled (a carried partial line first)
    bool eof = false;
    while (!eof) {
        if (in.size() < have + block) in.resize(have + block);
        size_t k;
        while (have < in.size() && (k = std::fread(in.data() + have, 1, in.size() - have, fp)) > 0) have += k;
        eof = have < in.size();
        // scan up to the last record end; the very last block goes as it is (a final record without '\n'
        // loses its last byte, like every record: trre_nft.c:777)
        size_t n = have;
        if (!eof) {
            while (n > 0 && in[n - 1] != '\n') --n;
            if (n == 0) continue;      // one line longer than the block: keep reading
        }
        if (n) {
            if (out.size() < n + 64) out.resize(n + 64);
            size_t m = 0;
            int rc = trre_scan_host_multi(prog, in.data(), n, out.data(), out.size(), &m, mask);
            if (rc == TRRE_E_CAPACITY) {
                out.resize(m + 64);
                rc = trre_scan_host_multi(prog, in.data(), n, out.data(), out.size(), &m, mask);
            }
            if (rc == TRRE_E_DIVERGES && m) {
                // the reference has printed everything up to the attempt it does not come back from (exit() flushes stdout,
                // trre_nft.c:551-553): so has the library (NFT engine), m bytes
                (void)std::fwrite(out.data(), 1, m, stdout);
                std::fflush(stdout);
            }
            if (rc != TRRE_OK) {
                std::fprintf(stderr, "%s\n", trre_last_error());
                return EXIT_FAILURE;
            }
            if (std::fwrite(out.data(), 1, m, stdout) != m) {
                std::fprintf(stderr, "error: write failed\n");
                return EXIT_FAILURE;
            }
        }
        std::memmove(in.data(), in.data() + n, have - n);
        have -= n;
    }
    trre_free(prog);
    return 0;
}
