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
// Like the reference's getline loop (trre_nft.c:776-790) the input is streamed, neither the input nor the output has to
// fit in host memory — as a pipeline (round 5; round 4 read, scanned and wrote one block after the other):
//   input    a regular file is MAPPED and handed to the scan block by block as it lies in the page cache: no read(), no copy of the
//            input on the host but the library's own staging copy into pinned memory (a thread ahead of the scan touches the next
//            block's pages).  A pipe or a terminal is read() into block buffers — whatever has arrived goes on as soon as it ends
//            in a '\n', so that an interactive producer sees its lines answered (the reference prints per getline);
//   scan     every block is cut after its last '\n' (the rest goes with the next block) and scanned line-sharded on all visible
//            GPUs (trre_scan_host_multi);
//   writer   write()s the finished blocks in order while the next one is scanned.
// Buffers are reused, never zero-filled, sized by the input, and NOT pinned: pinning costs a second per gigabyte on this platform
// (measured: five pinned 256 MiB blocks took 1.1-1.4 s of a 1.3 s run; the library pins its 32 MiB staging slots once).
// TRRE_DEVICES=<mask> restricts the GPUs used, TRRE_CLI_BLOCK=<bytes> sets the block size (default 256 MiB), TRRE_TRACE=1 says
// where the time went.
#include <fcntl.h>
#include <poll.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/trre_mi355x.h"

#ifndef TRRE_CLI_ENGINE
#define TRRE_CLI_ENGINE TRRE_ENGINE_NFT
#endif

namespace {

// a block of input or output: a buffer of its own (reused, never zero-filled), or a window of the mapped input file
struct Block {
    uint8_t* p = nullptr;
    size_t cap = 0;
    bool mapped = false;   // p points into the mapping: nothing to free
    size_t n = 0;          // bytes that go to the scan (whole records)
    size_t have = 0;       // bytes filled (n + the carried partial line)
    bool last = false;
    void reserve(size_t want) {
        if (!mapped && cap >= want) return;
        uint8_t* q = static_cast<uint8_t*>(std::malloc(want));
        if (!q) { std::fprintf(stderr, "error: out of memory\n"); std::_Exit(EXIT_FAILURE); }
        if (!mapped && p && have) std::memcpy(q, p, have);
        release();
        p = q; cap = want; mapped = false;
    }
    void release() {
        if (p && !mapped) std::free(p);
        p = nullptr; cap = 0; mapped = false;
    }
};

template <class T>
class Channel {            // a bounded hand-over between two stages
public:
    void push(T v) {
        std::unique_lock<std::mutex> lk(mu_);
        q_.push_back(v);
        cv_.notify_all();
    }
    T pop() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !q_.empty(); });
        T v = q_.front();
        q_.pop_front();
        return v;
    }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool write_all(int fd, const uint8_t* p, size_t n) {
    while (n) {
        const ssize_t k = ::write(fd, p, n > ((size_t)1 << 30) ? (size_t)1 << 30 : n);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        p += k; n -= (size_t)k;
    }
    return true;
}

}  // namespace

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
    const double t_main = now_s();
    trre_prog* prog = nullptr;
    if (trre_compile_mode(reinterpret_cast<const uint8_t*>(argv[optind]), std::strlen(argv[optind]), TRRE_CLI_ENGINE, mode, &prog) != TRRE_OK) {
        std::fprintf(stderr, "%s\n", trre_last_error());
        return EXIT_FAILURE;
    }
    if (std::getenv("TRRE_TRACE")) std::fprintf(stderr, "trre: pattern compiled in %.1f ms\n", (now_s() - t_main) * 1e3);
    int fd = 0;
    if (optind == argc - 2) {
        fd = ::open(argv[optind + 1], O_RDONLY);
        if (fd < 0) {
            std::fprintf(stderr, "error: can not open file %s\n", argv[optind + 1]);
            return EXIT_FAILURE;
        }
    }
    const uint32_t mask = std::getenv("TRRE_DEVICES") ? (uint32_t)std::strtoul(std::getenv("TRRE_DEVICES"), nullptr, 0) : 0u;
    size_t block = std::getenv("TRRE_CLI_BLOCK") ? (size_t)std::strtoull(std::getenv("TRRE_CLI_BLOCK"), nullptr, 0) : (size_t)256 << 20;
    if (block < 64) block = 64;
    struct stat sb;
    // A regular file WITH a size is mapped; anything else is read to its end like a pipe — also files that report size 0 (/proc and sysfs
    // entries are S_ISREG with st_size 0 and have content: the reference reads them with getline) and files the map fails on.
    bool regular = ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    off_t file_off = regular ? ::lseek(fd, 0, SEEK_CUR) : 0;
    if (file_off < 0) file_off = 0;
    size_t file_left0 = regular && sb.st_size > file_off ? (size_t)(sb.st_size - file_off) : 0;
    if (!file_left0) regular = false;

    constexpr int kIn = 3, kOut = 2;
    Block inb[kIn], outb[kOut];
    Channel<int> in_free, in_full, out_free, out_full;      // indices; -1 ends a stage
    for (int k = 0; k < kIn; ++k) in_free.push(k);
    for (int k = 0; k < kOut; ++k) out_free.push(k);
    struct OutJob { int buf; size_t m; };
    Channel<OutJob> jobs;
    bool write_failed = false;
    // TRRE_TRACE=1: where the time of the three stages went, on stderr at the end
    const bool trace = std::getenv("TRRE_TRACE") != nullptr;
    double t_read = 0, t_alloc = 0, t_scan = 0, t_write = 0, t_wait_in = 0, t_wait_out = 0;
    const double t_begin = now_s();

    // ---- input -----------------------------------------------------------------------------------------------------------------
    const uint8_t* map = nullptr;
    if (regular && file_left0) {
        void* m = ::mmap(nullptr, file_left0 + (size_t)(file_off & 4095), PROT_READ, MAP_PRIVATE, fd, file_off & ~(off_t)4095);
        if (m == MAP_FAILED) {
            regular = false;                               // read() it instead
            file_left0 = 0;
        } else {
            (void)::madvise(m, file_left0 + (size_t)(file_off & 4095), MADV_SEQUENTIAL);
            map = static_cast<const uint8_t*>(m) + (file_off & 4095);
        }
    }
    std::thread reader([&] {
        if (regular) {
            // windows of the mapping, cut after their last '\n'; the pages of a window are touched here, a window or two ahead of the scan
            size_t pos = 0;
            while (pos < file_left0) {
                const int b = in_free.pop();
                Block& B = inb[b];
                B.release();
                const double t0 = now_s();
                size_t end = std::min(file_left0, pos + block);
                while (end < file_left0) {
                    const void* nl = ::memrchr(map + pos, '\n', end - pos);
                    if (nl) { end = (size_t)(static_cast<const uint8_t*>(nl) - map) + 1; break; }
                    end = std::min(file_left0, end + block);       // one line longer than the block: a longer window
                }
                B.p = const_cast<uint8_t*>(map + pos); B.mapped = true; B.cap = 0;
                B.n = B.have = end - pos;
                B.last = end == file_left0;
                // (the window's pages into this process's page tables before the scan's staging threads get there: one call where the kernel
                // has MADV_POPULATE_READ (5.14), else a byte of every page)
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
                const uintptr_t pa = reinterpret_cast<uintptr_t>(map + pos) & ~(uintptr_t)4095;
                if (::madvise(reinterpret_cast<void*>(pa), reinterpret_cast<uintptr_t>(map + end) - pa, MADV_POPULATE_READ) != 0) {
                    volatile uint8_t sink = 0;
                    for (size_t k = pos; k < end; k += 4096) sink = sink ^ map[k];
                }
                t_read += now_s() - t0;
                pos = end;
                in_full.push(b);
            }
            in_full.push(-1);
            return;
        }
        std::vector<uint8_t> carry;
        size_t cur = std::min(block, (size_t)1 << 20);     // a pipe starts small (its first answer is not 256 MiB away) and grows
        bool eof = false;
        while (!eof) {
            const int b = in_free.pop();
            Block& B = inb[b];
            B.have = 0; B.n = 0; B.last = false;
            for (;;) {
                double t0 = now_s();
                B.reserve(std::max<size_t>(carry.size() + cur + 64, 4096));
                t_alloc += now_s() - t0;
                t0 = now_s();
                if (!carry.empty()) { std::memcpy(B.p, carry.data(), carry.size()); B.have = carry.size(); carry.clear(); }
                // a pipe or a terminal: block for the first bytes, then take what is there without waiting — until the block is
                // full — and go on as soon as the bytes end in a record
                while (B.have < B.cap - 64) {
                    const ssize_t k = ::read(fd, B.p + B.have, std::min(B.cap - 64 - B.have, (size_t)1 << 30));
                    if (k < 0) { if (errno == EINTR) continue; std::fprintf(stderr, "error: read failed\n"); std::_Exit(EXIT_FAILURE); }
                    if (k == 0) { eof = true; break; }
                    B.have += (size_t)k;
                    struct pollfd pf{fd, POLLIN, 0};
                    if (B.p[B.have - 1] == '\n' && ::poll(&pf, 1, 0) <= 0) break;
                }
                if (B.have >= B.cap - 64) cur = std::min(block, cur * 4);
                // up to the last record end; the very last block goes as it is (a final record without '\n' loses its last byte,
                // like every record: trre_nft.c:777)
                size_t n = B.have;
                if (!eof) {
                    const void* nl = ::memrchr(B.p, '\n', n);
                    n = nl ? (size_t)(static_cast<const uint8_t*>(nl) - B.p) + 1 : 0;
                }
                if (n == 0 && !eof) {                  // one line longer than the block: keep reading into a larger buffer
                    carry.assign(B.p, B.p + B.have);
                    B.have = 0;
                    cur *= 2;
                    if (block < cur) block = cur;
                    continue;
                }
                B.n = n;
                carry.assign(B.p + n, B.p + B.have);
                B.last = eof;
                t_read += now_s() - t0;
                break;
            }
            in_full.push(b);
        }
        in_full.push(-1);
    });

    // ---- writer ----------------------------------------------------------------------------------------------------------------
    std::thread writer([&] {
        for (;;) {
            const OutJob j = jobs.pop();
            if (j.buf < 0) break;
            const double t0 = now_s();
            if (!write_failed && !write_all(1, outb[j.buf].p, j.m)) write_failed = true;
            t_write += now_s() - t0;
            out_free.push(j.buf);
        }
    });

    // ---- scan ------------------------------------------------------------------------------------------------------------------
    int status = 0, n_calls = 0;
    bool undecided = false;
    for (;;) {
        double t0 = now_s();
        const int b = in_full.pop();
        t_wait_in += now_s() - t0;
        if (b < 0) break;
        Block& B = inb[b];
        if (B.n && !status) {
            t0 = now_s();
            const int o = out_free.pop();
            t_wait_out += now_s() - t0;
            Block& O = outb[o];
            O.have = 0;
            t0 = now_s();
            O.reserve(B.n + 64);
            t_alloc += now_s() - t0;
            size_t m = 0;
            t0 = now_s();
            int rc = trre_scan_host_multi(prog, B.p, B.n, O.p, O.cap, &m, mask);
            if (rc == TRRE_E_CAPACITY) {
                O.reserve(m + 64);
                rc = trre_scan_host_multi(prog, B.p, B.n, O.p, O.cap, &m, mask);
            }
            t_scan += now_s() - t0;
            if (trace && n_calls++ < 4) std::fprintf(stderr, "trre: scan call %d: %zu bytes in %.1f ms\n", n_calls, B.n, (now_s() - t0) * 1e3);
            undecided = undecided || (trre_last_scan_flags() & TRRE_SCAN_GUARD_UNDECIDED);
            // a scan the reference does not survive: it has printed everything up to the attempt it does not come back from (exit()
            // flushes stdout, trre_nft.c:551-553) — so has the library (NFT engine), m bytes
            if (rc == TRRE_OK || (rc == TRRE_E_DIVERGES && m)) jobs.push(OutJob{o, m});
            else out_free.push(o);
            if (rc != TRRE_OK) {
                jobs.push(OutJob{-1, 0});
                writer.join();
                std::fprintf(stderr, "%s\n", trre_last_error());
                std::fflush(stderr);
                std::_Exit(EXIT_FAILURE);          // (the reader may be waiting for input that will never be looked at)
            }
        }
        in_free.push(b);
    }
    jobs.push(OutJob{-1, 0});
    writer.join();
    reader.join();
    if (write_failed) {
        std::fprintf(stderr, "error: write failed\n");
        status = EXIT_FAILURE;
    }
    if (undecided)
        std::fprintf(stderr, "trre: warning: a line long enough to exhaust the reference's stack was not searched to the end (step budget): "
                             "where the reference may have exited with \"stack max capacity reached\" the match is printed\n");
    if (trace)
        std::fprintf(stderr, "trre: %.3f s in all: reader %.3f s reading + %.3f s allocating (both stages), scan calls %.3f s (waited %.3f s for input, %.3f s for an output buffer), writer %.3f s\n",
                     now_s() - t_begin, t_read, t_alloc, t_scan, t_wait_in, t_wait_out, t_write);
    for (Block& B : inb) B.release();
    for (Block& B : outb) B.release();
    trre_free(prog);
    return status;
}
