// runtime.cpp — C ABI (include/trre_mi355x.h) and host runtime: pattern
// compilation, table upload, kernel-family selection, launches, status
// collection.  The scan always runs on the GPU; nothing here computes output
// bytes on the host.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../../include/trre_mi355x.h"
#include "device_blob.hpp"
#include "front.hpp"
#include "launch.hpp"
#include "scan_block.hpp"
#include "splice_block.hpp"
#include "gen_block.hpp"
#include "lazy_block.hpp"
#include "one_block.hpp"
#include "map_block.hpp"
#include "guard_block.hpp"

namespace {

thread_local std::string g_error;
thread_local uint32_t g_scan_flags = 0;       // TRRE_SCAN_*: what the last scan call on this thread has to add to its return code

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(TRRE_E_DEVICE, std::string("hip: ") + hipGetErrorString(e_) + " at " #expr); \
    } while (0)

// Pinned staging memory: anonymous pages (huge ones where the kernel gives them), touched by a few threads, then registered with the
// runtime.  hipHostMalloc of the host path's six 36 MiB buffers took 56 ms of a process's first scan call (the runtime allocates, clears
// and pins 4 KiB pages on one thread); this takes 3 ms and copies at the same 57 GB/s (tools/probes/pin_cost.hip, round 5).
std::mutex g_pin_mu;
std::map<void*, size_t> g_pin_len;
hipError_t pinned_get(uint8_t** out, size_t bytes) {
    const size_t len = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* m = ::mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return hipErrorOutOfMemory;
#ifdef MADV_HUGEPAGE
    (void)::madvise(m, len, MADV_HUGEPAGE);
#endif
    {
        const int ways = len >= ((size_t)8 << 20) ? 4 : 1;
        const size_t piece = (len / (size_t)ways + 4095) & ~(size_t)4095;
        auto touch = [m, len, piece](int w) {
            volatile uint8_t* q = static_cast<volatile uint8_t*>(m);
            for (size_t k = (size_t)w * piece; k < std::min(len, (size_t)(w + 1) * piece); k += 4096) q[k] = 0;
        };
        std::vector<std::thread> th;
        for (int w = 1; w < ways; ++w) th.emplace_back(touch, w);
        touch(0);
        for (auto& t : th) t.join();
    }
    const hipError_t e = hipHostRegister(m, len, hipHostRegisterDefault);
    if (e != hipSuccess) { ::munmap(m, len); return e; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_len[m] = len;
    *out = static_cast<uint8_t*>(m);
    return hipSuccess;
}
void pinned_put(uint8_t* p) {
    if (!p) return;
    size_t len = 0;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pin_len.find(p);
        if (it == g_pin_len.end()) return;
        len = it->second;
        g_pin_len.erase(it);
    }
    (void)hipHostUnregister(p);
    ::munmap(p, len);
}

struct Pending {
    bool active = false;
    int family = 0;         // what runs
    int asked = 0;          // ... and what the caller's family choice was (a length-preserving family on an input with long lines runs as its general sibling)
    const uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    size_t n = 0, cap = 0;
    hipStream_t stream = nullptr;
    bool launched = false;
    bool timed = false;
    int count = 0;          // launches in the current batch
    const uint64_t* total_at = nullptr;   // general families: where the launch leaves its output size
    bool patched = false;                 // the launch was the mark + splice pair of a large table's copy form, not a count / emit pair
    bool one = false;                     // the launch was the one-pass kernel (one_block.hpp), not a count / emit pair
    bool mapgen = false;                  // ... the memoryless one-pass kernel (map_block.hpp)
    // the stack guard found, before an in-place launch, a line on which the reference's search runs out of stack: nothing was launched
    // exact sub-ranges (scan_block.hpp: ScanArgs::exact): what finish() needs to run repair rounds and the emit pass again
    bool exact = false;
    trre::ScanArgs xargs{};
    int64_t x_lane_bytes = 0, x_n_chunks = 0, x_rev_lane_bytes = 0;
    int x_g16 = 0, x_sym = 0;
    bool x_slow = false, x_ent_lds = false;
    bool guard_hit = false;
    uint64_t guard_line = 0;              // where that line starts
    uint32_t guard_part = 0;              // bytes of its output the reference had printed (in ScanCtx::d_gout), or ~0u: not available
};

// Everything ONE in-flight scan needs on a device besides the tables: status words, the workspaces of the
// general families, scratch buffers, timing events.  trre_scan_device / enqueue+finish use the device's own
// context; trre_scan_host keeps several chunks in flight, each with the context of its slot.
struct ScanCtx {
    uint32_t* d_status = nullptr;     // [4]
    uint32_t* h_status = nullptr;     // pinned mirror: [0] status bits; [2..3] total (u64)
    uint32_t* d_lane_counts = nullptr;
    uint64_t* d_chunk_total = nullptr;
    uint64_t* d_chunk_base = nullptr;
    int64_t ws_chunks = 0;
    uint8_t* d_scratch = nullptr;     // NFT tile kernels: mask scratch for lines longer than the LDS tile
    size_t scratch_bytes = 0;
    uint32_t* d_redo = nullptr;       // window kernel redo list
    int64_t redo_lanes = 0;
    uint8_t* d_sym = nullptr;         // guided families: one symbol per input byte
    uint8_t* d_gen_out = nullptr;     // generator modes: the enumeration's output before it goes down
    size_t gen_out_cap = 0;
    size_t sym_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the copy form of a large table (scan_block.hpp fb_lane<3> / fb_copy_lane): events, lane headers
    uint32_t* d_cevents = nullptr;
    uint32_t* d_chdr = nullptr;
    int64_t copy_lanes = 0;
    // the stack guard (guard_block.hpp): window flags, runs of flagged windows, results, the pool of stacks, the one line's output
    uint64_t* d_gflags = nullptr;
    size_t gflags_words = 0;
    uint8_t* d_gruns = nullptr;       // GuardRun[gruns_cap], then GuardResult[gruns_cap]
    size_t gruns_cap = 0;
    uint32_t* d_gstack = nullptr;     // [kGuardSlots][65 536][3]
    uint8_t* d_gobuf = nullptr;
    size_t gobuf_cap = 0;
    uint8_t* d_gout = nullptr;
    size_t gout_cap = 0;
    // exact sub-ranges: entry / exit states and flags per lane; the long-line probe and what it said about which buffer
    uint32_t* d_spec = nullptr;
    int64_t spec_lanes = 0;
    uint32_t* d_probe = nullptr;
    const uint8_t* probe_in = nullptr;
    size_t probe_n = 0;
    bool long_lines = false;
    bool exact_off = false;           // finish() runs a scan again the old way (a diverging attempt: the first lane in stream order has to be named)
    // the one-pass kernel (one_block.hpp): look-back descriptors [tiles], then the ticket counter and the total
    uint64_t* d_one = nullptr;
    int64_t one_tiles = 0;
    bool one_off = false;             // finish() runs the scan again as a count / emit pair (a void one-pass launch)
    uint64_t* d_mg = nullptr;         // the memoryless kernel (map_block.hpp): descriptors, group sums and totals, the total
    int64_t mg_tiles = 0;
    bool mapgen_off = false;          // finish() runs the scan again on the general family (a NUL in the input)
    int mapgen_voids = 0;             // launches of the memoryless kernel that were void (a NUL; a grid that was not resident): after two, the pair from the start
    bool mapgen_dense = false;        // a program with longer texts printed 4 % more than it read: its texts are frequent ('a:xyz': the pair is 7 % faster there)
    uint32_t* d_miss = nullptr;       // lazy tables (lazy_block.hpp): [0] misses listed, then {row, class} pairs
    int bt_tier = 0;                  // the backtracking fallback: which size of stacks and path buffers the next launch uses (kBtTiers)
    bool guard_off = false;           // finish() scans the lines before a line the guard stopped at: not to be guarded again
    bool patch_off = false;           // finish() runs the scan again as a count / emit pair (diverged, or out of overflow records)
    int relaunches = 0;               // finish() ran the scan again (scratch, NUL, overflow): whatever was downloaded early is stale
    Pending pend;
};

constexpr int kHostSlots = 3;         // chunks in flight in trre_scan_host: staging in, on the device, staging out

struct DeviceState {
    std::mutex mu;                    // one scan call at a time per (prog, device): trre_scan_device, trre_scan_host
    int device = 0;
    uint8_t* d_blob = nullptr;
    uint8_t* d_sblob = nullptr;       // stream tables
    uint8_t* d_gblob = nullptr;       // guided families: forward tables (stream form) ...
    uint8_t* d_rblob = nullptr;       // ... and the backward DFA
    uint8_t* d_nblob = nullptr;       // generator modes: the enumeration's tables
    uint8_t* d_kblob = nullptr;       // the stack guard's tables (the NFT itself)
    // the deterministic engine's tables while they are being built (front.hpp: LazyDft): this device's copy and how much of the host's it holds.
    // A buffer that has to grow is replaced, not freed: scans of other chunks may still be reading it (freed with the state).
    uint64_t* d_lent = nullptr;
    uint8_t* d_lpool = nullptr;
    uint8_t* d_lcls = nullptr;
    size_t lent_cap = 0, lpool_cap = 0;        // rows / bytes
    uint32_t lazy_rows_up = 0, lazy_epoch_up = 0;
    size_t lazy_pool_up = 0;
    std::vector<void*> retired;
    ScanCtx ctx;
    struct HostSlot {
        uint8_t *pin_in = nullptr, *pin_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        size_t in_cap = 0, out_cap = 0;
        hipStream_t stream = nullptr;
        ScanCtx ctx;
    } slot[kHostSlots];
};

template <class T>
void put(std::vector<uint8_t>& b, size_t off, const T* src, size_t count) {
    if (b.size() < off + count * sizeof(T)) b.resize(off + count * sizeof(T));
    if (count) std::memcpy(b.data() + off, src, count * sizeof(T));
}
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

struct trre_prog {
    int engine = 0;
    int forced_family = TRRE_KERNEL_AUTO;
    uint32_t nft_states = 0, nft_cons = 0;
    trre::DftTables dt;
    trre::NftTables nt;
    trre::StreamTables stt;
    trre::GuidedTables gt;
    trre::GenTables gen;              // generator modes (`-a`): viability DFA for the device, follow lists for the host enumeration
    int mode = TRRE_MODE_SCAN;
    uint32_t nft_nodes = 0;
    bool has_engine_tables = false;   // tile kernels available (always for DFT; NFT: <= 64 nodes, no epsilon cycle)
    std::vector<uint8_t> blob;
    std::vector<uint8_t> sblob;
    std::vector<uint8_t> gblob, rblob;
    std::vector<uint8_t> nblob;       // generator modes: the enumeration's tables (gen_block.hpp); scan mode, NFT engine: the same lists for the backtracking fallback
    bool bt_ok = false;               // scan mode, NFT engine: nblob holds the backtracking fallback's tables
    trre::Nft dft_nft;                // DFT engine: the automaton the lazy tables are built from
    std::unique_ptr<trre::LazyDft> lazy;   // DFT engine: the tables one miss at a time (front.hpp) — a pattern beyond the eager caps, or on request
    bool lazy_only = false;           // ... the eager construction gave up: nothing else exists
    std::mutex lazy_mu;               // explore() and the uploads
    trre::GuardTables guard;          // scan mode, NFT engine: which lines can exhaust the reference's stack (stack_guard.cpp)
    std::vector<uint8_t> kblob;
    int mask_bytes = 0;
    bool profiling = false;
    std::atomic<float> last_ms{-1.f};
    std::atomic<bool> bounded_off{false};     // a launch on a bounded stream table met a run it was not built for: the guided / tile kernels from now on
    std::atomic<int> one_fails{0};            // void launches of the one-pass kernel (a region outgrown, a first guess wrong): after two the pair from now on
    std::atomic<bool> copy_form_off{false};   // a launch of the copy form met more texts than its event lists hold: the count / emit pair from now on
    std::mutex dev_mu;                                  // guards the map (not the states)
    std::map<int, std::unique_ptr<DeviceState>> dev;
};

namespace {

void serialize_dft(trre_prog& p) {
    using namespace trre;
    const DftTables& t = p.dt;
    DftBlobHeader h{};
    h.magic = kMagicDft;
    h.n_rows = t.n_rows;
    h.n_cls = t.n_cls;
    h.flags = t.flags;
    h.max_edge_out = t.max_edge_out;
    h.n_states = t.n_states;
    size_t off = sizeof h;
    h.off_ent0 = (uint32_t)off; off += 256 * 8;
    h.off_cls = (uint32_t)off; off += 256;
    h.off_bytemap = (uint32_t)off; off += 256;
    h.off_ent = (uint32_t)off; off += t.ent.size() * 8;
    h.off_pool = (uint32_t)off; h.pool_bytes = (uint32_t)t.pool.size(); off += t.pool.size();
    off = align_up(off + 16, 16);
    h.total_bytes = (uint32_t)off;
    std::vector<uint8_t>& b = p.blob;
    b.assign(off, 0);
    std::vector<uint64_t> ent0(256);
    for (int c = 0; c < 256; ++c) ent0[c] = t.ent[t.cls[c]];      // row 0, expanded over raw bytes
    put(b, 0, &h, 1);
    put(b, h.off_ent0, ent0.data(), 256);
    put(b, h.off_cls, t.cls.data(), 256);
    put(b, h.off_bytemap, t.bytemap.data(), 256);
    put(b, h.off_ent, t.ent.data(), t.ent.size());
    put(b, h.off_pool, t.pool.data(), t.pool.size());
}

void serialize_nft(trre_prog& p) {
    using namespace trre;
    const NftTables& t = p.nt;
    NftBlobHeader h{};
    h.magic = kMagicNft;
    h.n_cons = t.n_cons;
    h.flags = t.flags;
    h.n_follow = (uint32_t)t.follow.size();
    h.n_states = t.n_states;
    size_t off = sizeof h;
    h.off_cons_mask = (uint32_t)off; off += 256 * 8;
    h.off_pred = (uint32_t)off; off += (t.n_cons + 1) * 8;
    h.off_follow_off = (uint32_t)off; off += align_up((t.n_cons + 2) * 4, 8);
    h.off_follow = (uint32_t)off; off += t.follow.size() * sizeof(NftFollow);
    h.off_pool = (uint32_t)off; h.pool_bytes = (uint32_t)t.pool.size(); off += t.pool.size();
    off = align_up(off + 16, 16);
    h.total_bytes = (uint32_t)off;
    std::vector<uint8_t>& b = p.blob;
    b.assign(off, 0);
    std::vector<uint64_t> pred(t.pred);
    pred.push_back(t.to_final);
    put(b, 0, &h, 1);
    put(b, h.off_cons_mask, t.cons_mask.data(), 256);
    put(b, h.off_pred, pred.data(), pred.size());
    put(b, h.off_follow_off, t.follow_off.data(), t.follow_off.size());
    put(b, h.off_follow, t.follow.data(), t.follow.size());
    put(b, h.off_pool, t.pool.data(), t.pool.size());
}

void serialize_stream(const trre::StreamTables& t, std::vector<uint8_t>& b) {
    using namespace trre;
    StreamBlobHeader h{};
    h.magic = kMagicStream;
    h.n_states = t.n_states;
    h.n_cls = t.n_cls;
    h.flags = t.flags;
    h.max_out = t.max_out;
    size_t off = sizeof h;
    h.off_cls = (uint32_t)off; off += 256;
    h.off_ent = (uint32_t)off; h.ent_bytes = (uint32_t)(t.ent.size() * 8); off += t.ent.size() * 8;
    h.off_pool = (uint32_t)off; h.pool_bytes = (uint32_t)t.pool.size(); off += t.pool.size();
    off = align_up(off, 16);
    h.off_lpw = (uint32_t)off; h.lpw_bytes = (uint32_t)(t.lpw.size() * 4); h.lpw_delay = t.lpw_delay; off += t.lpw.size() * 4;
    off = align_up(off, 16);
    h.off_g16 = (uint32_t)off; h.g16_bytes = (uint32_t)(t.g16.size() * 4); off += t.g16.size() * 4;
    off = align_up(off, 16);
    h.off_p32 = (uint32_t)off; h.p32_bytes = (uint32_t)(t.p32.size() * 4); h.p32_slow = t.p32_slow ? 1u : 0u; off += t.p32.size() * 4;
    off = align_up(off, 16);
    h.off_lpw2 = (uint32_t)off; h.lpw2_bytes = (uint32_t)(t.lpw2.size() * 4); off += t.lpw2.size() * 4;
    off = align_up(off, 16);
    if (t.fb_ok) {
        h.fb_slots = (uint32_t)t.fb_comb.size(); h.off_fb_comb = (uint32_t)off; off = align_up(off + t.fb_comb.size() * 8, 16);
        h.fb_lits = (uint32_t)t.fb_lit.size(); h.off_fb_lit = (uint32_t)off; off = align_up(off + t.fb_lit.size() * 8, 16);
        h.fb_escs = (uint32_t)t.fb_esc_slot.size(); h.off_fb_esc_slot = (uint32_t)off; off = align_up(off + t.fb_esc_slot.size() * 4, 16);
        h.off_fb_esc = (uint32_t)off; off = align_up(off + t.fb_esc.size() * 4, 16);
        h.off_fb_pool = (uint32_t)off; off = align_up(off + t.fb_pool.size(), 16);
        if (t.fb_copy_ok) { h.off_fb_lit_meta = (uint32_t)off; off = align_up(off + t.fb_lit_meta.size() * 2, 16); }
        if (t.fb_mark4_ok) {
            h.fb4_slots = (uint32_t)t.fb_comb4.size(); h.off_fb_comb4 = (uint32_t)off; off = align_up(off + t.fb_comb4.size() * 4, 16);
            h.fb4_dense = (uint32_t)t.fb_dense_base.size(); h.off_fb_dense4 = (uint32_t)off; off = align_up(off + t.fb_dense4.size() * 4, 16);
            h.off_fb_dense_base = (uint32_t)off; off = align_up(off + t.fb_dense_base.size() * 2, 16);
            h.fb_pad = t.fb_pad;
            std::memcpy(h.fb_start4, t.fb_start4, sizeof h.fb_start4);
        }
        std::memcpy(h.fb_start, t.fb_start, sizeof h.fb_start);
    }
    if (t.mg_max) { h.off_mg = (uint32_t)off; h.mg_max = t.mg_max; off = align_up(off + t.mg.size() * 4, 16); }
    off = align_up(off + 16, 16);
    h.total_bytes = (uint32_t)off;
    b.assign(off, 0);
    put(b, 0, &h, 1);
    if (t.mg_max) put(b, h.off_mg, t.mg.data(), t.mg.size());
    if (t.fb_ok) {
        put(b, h.off_fb_comb, t.fb_comb.data(), t.fb_comb.size());
        put(b, h.off_fb_lit, t.fb_lit.data(), t.fb_lit.size());
        put(b, h.off_fb_esc_slot, t.fb_esc_slot.data(), t.fb_esc_slot.size());
        put(b, h.off_fb_esc, t.fb_esc.data(), t.fb_esc.size());
        put(b, h.off_fb_pool, t.fb_pool.data(), t.fb_pool.size());
        if (t.fb_copy_ok) put(b, h.off_fb_lit_meta, t.fb_lit_meta.data(), t.fb_lit_meta.size());
        if (t.fb_mark4_ok) {
            put(b, h.off_fb_comb4, t.fb_comb4.data(), t.fb_comb4.size());
            put(b, h.off_fb_dense4, t.fb_dense4.data(), t.fb_dense4.size());
            put(b, h.off_fb_dense_base, t.fb_dense_base.data(), t.fb_dense_base.size());
        }
    }
    put(b, h.off_cls, t.cls.data(), 256);
    put(b, h.off_ent, t.ent.data(), t.ent.size());
    put(b, h.off_pool, t.pool.data(), t.pool.size());
    put(b, h.off_lpw, t.lpw.data(), t.lpw.size());
    put(b, h.off_g16, t.g16.data(), t.g16.size());
    put(b, h.off_p32, t.p32.data(), t.p32.size());
    put(b, h.off_lpw2, t.lpw2.data(), t.lpw2.size());
}

void serialize_rev_table(uint32_t n_rev, uint32_t n_cls, uint32_t sym_bits, const std::array<uint8_t, 256>& cls, const std::vector<uint8_t>& rev,
                         std::vector<uint8_t>& b) {
    using namespace trre;
    RevBlobHeader h{};
    h.magic = kMagicRev;
    h.n_rev = n_rev;
    h.n_cls = n_cls;
    h.sym_bits = sym_bits;
    size_t off = sizeof h;
    h.off_cls = (uint32_t)off; off += 256;
    h.off_tab = (uint32_t)off; off += align_up(rev.size(), 16);
    h.off_wide = (uint32_t)off; off += (size_t)n_rev * 256;
    off = align_up(off + 16, 16);
    h.total_bytes = (uint32_t)off;
    b.assign(off, 0);
    std::vector<uint8_t> wide((size_t)n_rev * 256);
    for (uint32_t r = 0; r < n_rev; ++r)
        for (int c = 0; c < 256; ++c) wide[(size_t)r * 256 + c] = rev[(size_t)r * n_cls + cls[c]];
    put(b, 0, &h, 1);
    put(b, h.off_cls, cls.data(), 256);
    put(b, h.off_tab, rev.data(), rev.size());
    put(b, h.off_wide, wide.data(), wide.size());
}
void serialize_rev(const trre::GuidedTables& g, std::vector<uint8_t>& b) {
    if (!g.wide) { serialize_rev_table(g.n_rev, g.n_cls, g.sym_bits, g.cls, g.rev, b); return; }
    // more than 256 states: 16-bit entries, [state][raw byte], read through L1 / L2 by k_rev_wide
    using namespace trre;
    RevBlobHeader h{};
    h.magic = kMagicRev;
    h.n_rev = g.n_rev;
    h.n_cls = g.n_cls;
    h.sym_bits = 16;
    size_t off = sizeof h;
    h.off_cls = (uint32_t)off; off += 256;
    h.off_tab = (uint32_t)off; off += align_up(g.rev16.size() * 2, 16);
    h.off_wide = (uint32_t)off; off += (size_t)g.n_rev * 256 * 2;
    off = align_up(off + 16, 16);
    h.total_bytes = (uint32_t)off;
    b.assign(off, 0);
    std::vector<uint16_t> wide((size_t)g.n_rev * 256);
    for (uint32_t r = 0; r < g.n_rev; ++r)
        for (int c = 0; c < 256; ++c) wide[(size_t)r * 256 + c] = g.rev16[(size_t)r * g.n_cls + g.cls[c]];
    put(b, 0, &h, 1);
    put(b, h.off_cls, g.cls.data(), 256);
    put(b, h.off_tab, g.rev16.data(), g.rev16.size());
    put(b, h.off_wide, wide.data(), wide.size());
}

// generator modes: follow lists (every epsilon path), the nodes' byte sets, the viability sets per symbol (gen_block.hpp)
void serialize_gen(const trre::GenTables& g, std::vector<uint8_t>& b) {
    using namespace trre;
    const NftNodes& nd = g.nodes;
    const uint32_t n_nodes = (uint32_t)nd.node.size();
    std::vector<uint32_t> bytes((size_t)n_nodes * 8, 0), foff, follow;
    std::vector<uint8_t> echo(n_nodes, 0), pool;
    for (uint32_t t = 0; t < n_nodes; ++t) {
        for (int c = 0; c < 256; ++c)
            if (nd.node[t].reads((uint8_t)c)) bytes[(size_t)t * 8 + (c >> 5)] |= 1u << (c & 31);
        echo[t] = nd.node[t].echo ? 1 : 0;
    }
    for (size_t l = 0; l < nd.follow.size(); ++l) {
        foff.push_back((uint32_t)(follow.size() / 3));
        for (const NodeFollow& e : nd.follow[l]) {
            if (e.out.size() > 0xffff) throw Error(kErrTooBig, "error: an output of more than 64 KiB on one epsilon path (generator mode)");
            follow.push_back(e.target);
            follow.push_back((uint32_t)pool.size());
            follow.push_back((uint32_t)e.out.size() | (e.mute ? 1u << 16 : 0u));
            pool.insert(pool.end(), e.out.begin(), e.out.end());
        }
    }
    foff.push_back((uint32_t)(follow.size() / 3));
    GenBlobHeader h{};
    h.magic = kMagicGen;
    h.n_nodes = n_nodes;
    h.n_rev = g.n_rev;
    h.words = g.viable_words;
    h.match_mode = g.match_mode ? 1u : 0u;
    h.n_follow = (uint32_t)(follow.size() / 3);
    size_t off = align_up(sizeof h, 16);
    h.off_bytes = (uint32_t)off; off = align_up(off + bytes.size() * 4, 16);
    h.off_echo = (uint32_t)off; off = align_up(off + echo.size(), 16);
    h.off_foff = (uint32_t)off; off = align_up(off + foff.size() * 4, 16);
    h.off_follow = (uint32_t)off; off = align_up(off + follow.size() * 4, 16);
    h.off_pool = (uint32_t)off; h.pool_bytes = (uint32_t)pool.size(); off = align_up(off + pool.size() + 8, 16);
    h.off_viable = (uint32_t)off; off = align_up(off + g.viable.size() * 8, 16);
    h.total_bytes = (uint32_t)off;
    b.assign(off, 0);
    put(b, 0, &h, 1);
    put(b, h.off_bytes, bytes.data(), bytes.size());
    put(b, h.off_echo, echo.data(), echo.size());
    put(b, h.off_foff, foff.data(), foff.size());
    put(b, h.off_follow, follow.data(), follow.size());
    put(b, h.off_pool, pool.data(), pool.size());
    put(b, h.off_viable, g.viable.data(), g.viable.size());
}

bool is_generate(int mode) { return mode == TRRE_MODE_SCAN_ALL || mode == TRRE_MODE_MATCH_ALL; }
bool is_stream(int fam) { return fam == TRRE_KERNEL_STREAM_LP || fam == TRRE_KERNEL_STREAM_GEN; }
bool is_guided(int fam) { return fam == TRRE_KERNEL_GUIDED_LP || fam == TRRE_KERNEL_GUIDED_GEN; }
bool is_gen(int fam) {
    return fam == TRRE_KERNEL_TILE_GEN || fam == TRRE_KERNEL_STREAM_GEN || fam == TRRE_KERNEL_GUIDED_GEN || fam == TRRE_KERNEL_BACKTRACK || fam == TRRE_KERNEL_DFT_LAZY;
}
bool is_guided_wide(const trre_prog& p, int fam) { return (fam == TRRE_KERNEL_GUIDED_LP || fam == TRRE_KERNEL_GUIDED_GEN) && p.gt.wide; }
bool lp_inplace(uint32_t flags) { return (flags & trre::kFlagLengthPreserving) && (flags & trre::kFlagNoOverrun); }
// the family that takes over when a length-preserving launch met a NUL, or a bounded stream table a long run
int general_family(const trre_prog& p, bool stream_ok) {
    if (p.lazy_only) return TRRE_KERNEL_DFT_LAZY;
    if (stream_ok && p.stt.ok) return TRRE_KERNEL_STREAM_GEN;
    if (p.gt.ok) return TRRE_KERNEL_GUIDED_GEN;
    if (p.engine == TRRE_ENGINE_DFT && p.mode == TRRE_MODE_SCAN) return TRRE_KERNEL_DFT_LAZY;      // (round 6: 100 GB/s against the tile kernels' 48)
    return p.has_engine_tables ? TRRE_KERNEL_TILE_GEN : TRRE_KERNEL_BACKTRACK;
}

int auto_family(const trre_prog& p) {
    using namespace trre;
    if (p.lazy_only) return TRRE_KERNEL_DFT_LAZY;      // beyond the eager construction's caps: the tables grow with the input (lazy_block.hpp)
    if (p.engine == TRRE_ENGINE_DFT && (p.dt.flags & kFlagMemoryless)) return TRRE_KERNEL_BYTEMAP;
    // a stream table too large for LDS is walked through L1/L2 (<= 0.35 TB/s); guided tables with a small forward table
    // run at 0.6-0.8 TB/s: prefer them then (patterns with `.` or wide ranges before a literal fold into hundreds of
    // states x 256 classes)
    const bool stream_small = p.stt.ok && (p.stt.g16_ok || p.stt.lpw_ok);
    const bool guided_small = p.gt.ok && p.gt.fwd.g16_ok && !p.gt.wide;
    // (a bounded fold whose flushes do not fit the 16-byte entries — 'a*b:x': up to 64 pending bytes go out raw when the b does not
    // come — walks them on its slow path: 0.30 TB/s against 0.60 for the guided tables of the same pattern; ' +: ' has none: 0.72 / 0.59)
    const bool bounded_slow = p.stt.ok && p.stt.bounded && (p.stt.flags & kFlagG16Slow) && guided_small;
    if (p.stt.ok && !bounded_slow && (stream_small || !guided_small)) return lp_inplace(p.stt.flags) ? TRRE_KERNEL_STREAM_LP : TRRE_KERNEL_STREAM_GEN;
    // (wide guided tables — 16-bit symbols, both tables through L1 / L2 — still beat the bitmask tile kernels: 37 against 17 GB/s
    // on 'a(a|b|c){9}c:x', 256 MiB of printable lines)
    if (p.gt.ok) return lp_inplace(p.gt.fwd.flags) && !p.gt.wide ? TRRE_KERNEL_GUIDED_LP : TRRE_KERNEL_GUIDED_GEN;
    if (p.mode == TRRE_MODE_MATCH) return TRRE_KERNEL_BACKTRACK;
    if (p.engine == TRRE_ENGINE_DFT) {
        // A DFT pattern without a fold and beyond the guided tables (a reduced backward automaton of more than 16 384 states, or one that its
        // construction's budget does not reach).  Rounds 1-5: the tile kernels, 48 GB/s; round 6: the lazily determinised family, which runs any
        // DFT scan at twice that once its tables have grown — a first scan that needs more than kLazyAutoRounds rounds of
        // growing hands the buffer to the tile kernels (finish_inner) and the tables keep what they learnt.
        if (p.mode == TRRE_MODE_SCAN) return TRRE_KERNEL_DFT_LAZY;
        if ((p.dt.flags & kFlagLengthPreserving) && (p.dt.flags & kFlagNoOverrun)) return TRRE_KERNEL_TILE_LP;
        return TRRE_KERNEL_TILE_GEN;
    }
    // beyond every table form (more than 64 nodes, a backward automaton beyond the guided limits, no fold): the search itself
    if (!p.has_engine_tables) return TRRE_KERNEL_BACKTRACK;
    return (p.nt.flags & kFlagLengthPreserving) ? TRRE_KERNEL_TILE_LP : TRRE_KERNEL_TILE_GEN;
}

// the family a scan call launches: the forced one, else the automatic choice — which leaves a bounded stream table alone once
// a launch on it has overflowed
int scan_family(const trre_prog& p) {
    if (p.forced_family) return p.forced_family;
    const int fam = auto_family(p);
    if (is_stream(fam) && p.stt.bounded && p.bounded_off.load()) return general_family(p, false);
    return fam;
}

bool family_allowed(const trre_prog& p, int fam) {
    using namespace trre;
    if (fam == TRRE_KERNEL_DFT_LAZY) return p.engine == TRRE_ENGINE_DFT && p.mode == TRRE_MODE_SCAN;
    if (p.lazy_only) return false;
    if (fam == TRRE_KERNEL_STREAM_GEN) return p.stt.ok;
    if (fam == TRRE_KERNEL_STREAM_LP) return p.stt.ok && lp_inplace(p.stt.flags);
    if (fam == TRRE_KERNEL_GUIDED_GEN) return p.gt.ok;
    if (fam == TRRE_KERNEL_GUIDED_LP) return p.gt.ok && lp_inplace(p.gt.fwd.flags) && !p.gt.wide;
    if (fam == TRRE_KERNEL_BACKTRACK) return p.bt_ok;
    if (!p.has_engine_tables) return false;
    if (fam == TRRE_KERNEL_TILE_GEN) return true;
    if (p.engine == TRRE_ENGINE_DFT) {
        if (fam == TRRE_KERNEL_BYTEMAP) return (p.dt.flags & kFlagMemoryless) != 0;
        return (p.dt.flags & kFlagLengthPreserving) && (p.dt.flags & kFlagNoOverrun);
    }
    if (fam == TRRE_KERNEL_BYTEMAP) return false;
    return (p.nt.flags & kFlagLengthPreserving) != 0;
}

int ctx_init(ScanCtx& c) {
    if (c.d_status) return TRRE_OK;
    // status words [4], then the byte map's NUL list: a count, kNulCap offsets, kNulCap line ends (repair_bytemap_nuls)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c.d_status), 24 + (size_t)trre::kNulCap * 16));
    HIP_TRY(hipMemset(c.d_status, 0, 24));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c.h_status), 16, hipHostMallocDefault));
    HIP_TRY(hipEventCreate(&c.ev0));
    HIP_TRY(hipEventCreate(&c.ev1));
    return TRRE_OK;
}
void ctx_free(ScanCtx& c) {
    (void)hipFree(c.d_status);
    if (c.h_status) (void)hipHostFree(c.h_status);
    (void)hipFree(c.d_lane_counts);
    (void)hipFree(c.d_chunk_total);
    (void)hipFree(c.d_chunk_base);
    (void)hipFree(c.d_scratch);
    (void)hipFree(c.d_redo);
    (void)hipFree(c.d_sym);
    (void)hipFree(c.d_gen_out);
    (void)hipFree(c.d_gflags); (void)hipFree(c.d_gruns); (void)hipFree(c.d_gstack); (void)hipFree(c.d_gobuf); (void)hipFree(c.d_gout);
    (void)hipFree(c.d_miss);
    (void)hipFree(c.d_spec);
    (void)hipFree(c.d_one);
    (void)hipFree(c.d_mg);
    (void)hipFree(c.d_probe);
    (void)hipFree(c.d_cevents);
    (void)hipFree(c.d_chdr);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    c = ScanCtx();
}

// the prog's state on device `dev` (tables uploaded on first use); the calling thread's current device must be `dev`
int device_state(trre_prog* p, int dev, DeviceState** out) {
    std::lock_guard<std::mutex> lock(p->dev_mu);
    auto it = p->dev.find(dev);
    if (it == p->dev.end()) {
        static const bool trace_on = getenv("TRRE_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        std::unique_ptr<DeviceState> st(new DeviceState);
        st->device = dev;
        auto upload = [&](const std::vector<uint8_t>& b, uint8_t** d) -> int {
            if (b.empty()) return TRRE_OK;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(d), b.size()));
            HIP_TRY(hipMemcpy(*d, b.data(), b.size(), hipMemcpyHostToDevice));
            return TRRE_OK;
        };
        int rc = upload(p->blob, &st->d_blob);
        if (!rc) rc = upload(p->sblob, &st->d_sblob);
        if (!rc) rc = upload(p->gblob, &st->d_gblob);
        if (!rc) rc = upload(p->rblob, &st->d_rblob);
        if (!rc) rc = upload(p->nblob, &st->d_nblob);
        if (!rc) rc = upload(p->kblob, &st->d_kblob);
        if (!rc) rc = ctx_init(st->ctx);
        if (rc) return rc;
        if (trace_on) fprintf(stderr, "trre: device %d: tables up in %.1f ms\n", dev, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        it = p->dev.emplace(dev, std::move(st)).first;
    }
    *out = it->second.get();
    return TRRE_OK;
}

int ensure_workspace(ScanCtx* c, int64_t n_chunks, int threads) {
    n_chunks = n_chunks * ((threads + 255) / 256);      // sized in units of 256-lane chunks
    if (n_chunks <= c->ws_chunks) return TRRE_OK;
    if (c->d_lane_counts) { (void)hipFree(c->d_lane_counts); (void)hipFree(c->d_chunk_total); (void)hipFree(c->d_chunk_base); }
    c->d_lane_counts = nullptr; c->d_chunk_total = nullptr; c->d_chunk_base = nullptr;
    c->ws_chunks = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_lane_counts), (size_t)n_chunks * 256 * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_chunk_total), (size_t)n_chunks * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_chunk_base), (size_t)(n_chunks + 1) * 8));
    c->ws_chunks = n_chunks;
    return TRRE_OK;
}

constexpr uint32_t kCopyEvCap = 256;       // events per lane of the copy form (a 2 KiB sub-range of the dictionary corpus holds ~120)
// (n_lanes counts rows of kCopyEvCap events: a lane of more than 2 KiB takes several)
int ensure_copy_workspace(ScanCtx* c, int64_t n_lanes) {
    if (n_lanes <= c->copy_lanes) return TRRE_OK;
    (void)hipFree(c->d_cevents); (void)hipFree(c->d_chdr);
    c->d_cevents = nullptr; c->d_chdr = nullptr;
    c->copy_lanes = 0;
    // (events: a row of kCopyEvCap per lane, 4 bytes each)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_cevents), (size_t)n_lanes * kCopyEvCap * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_chdr), (size_t)n_lanes * 16));
    c->copy_lanes = n_lanes;
    return TRRE_OK;
}

// ---- the stack guard (guard_block.hpp) --------------------------------------------------------------------------------
constexpr int64_t kGuardSlots = 256;                       // lines searched at a time (a stack of 65 536 items each: 192 MiB in all)
constexpr size_t kGuardMaxRuns = 1024;                     // suspects per batch (the call's budget is looked at between batches)
constexpr uint64_t kGuardBudget = 1ull << 23;              // search steps per line; beyond: not decided (the scan's output stands)
// ... and per scan call, summed over its suspect lines in input order, batch by batch (TRRE_GUARD_CALL_BUDGET).  Steps, not seconds:
// the same input gives the same answer on a loaded host and on an idle one (round 4 stopped after 20 s of wall clock).  What the budgets
// leave undecided is SAID: TRRE_SCAN_GUARD_UNDECIDED in trre_last_scan_flags().
constexpr uint64_t kGuardCallBudget = 1ull << 35;
constexpr const char* kStackMsg = "error: stack max capacity reached";
bool guard_applies(const trre_prog& p, const ScanCtx& cx, size_t n) {
    static const bool off = getenv("TRRE_NO_STACK_GUARD") != nullptr;
    return !off && p.guard.on && !cx.guard_off && (p.mode == TRRE_MODE_SCAN || p.mode == TRRE_MODE_MATCH) && n + 1 >= p.guard.l_min;
}
struct GuardHit {
    bool hit = false;
    uint64_t line_start = 0;
    uint32_t part = 0;         // bytes of the line's output in cx->d_gout, or ~0u: they could not be produced
};
template <class T>
int guard_room(T** d, size_t* have, size_t want) {
    if (*have >= want) return TRRE_OK;
    if (*d) (void)hipFree(*d);
    *d = nullptr; *have = 0;
    if (hipMalloc(reinterpret_cast<void**>(d), want * sizeof(T)) != hipSuccess) return fail(TRRE_E_TOO_BIG, "error: out of device memory (stack guard)");
    *have = want;
    return TRRE_OK;
}
// Looks for lines long enough to exhaust the reference's stack and runs the reference's search on them.  Synchronous: it ends
// with the stream idle.  hit: the first line (in input order) on which the search overflows, and what the reference had printed of it.
int guard_check(trre_prog* p, DeviceState* st, ScanCtx* cx, const uint8_t* d_in, size_t n, hipStream_t stream, GuardHit* hit) {
    using namespace trre;
    const GuardTables& g = p->guard;
    const int64_t al = (int64_t)(reinterpret_cast<uintptr_t>(d_in) & 15u);
    ScanArgs args{};
    args.in_v0 = d_in - al;
    args.vbeg = al;
    args.vend = al + (int64_t)n;
    const int64_t n_win = ((int64_t)n + g.window - 1) / g.window;
    const size_t n_words = (size_t)(n_win + 255) / 256 * 4;
    int rc = guard_room(&cx->d_gflags, &cx->gflags_words, n_words);
    if (rc) return rc;
    args.blob = st->d_kblob;
    launch_guard_probe(args, g.window, n_win, cx->d_gflags, stream);
    HIP_TRY(hipGetLastError());
    std::vector<uint64_t> flags(n_words);
    HIP_TRY(hipMemcpyAsync(flags.data(), cx->d_gflags, n_words * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    // the runs of flagged windows, a batch at a time (input order: the first line that overflows is the one that counts)
    std::vector<GuardRun> runs;
    std::vector<GuardResult> res;
    GuardArgs ga{};
    GuardResult* d_res = nullptr;
    GuardRun* d_runs = nullptr;
    size_t bad = 0;
    bool found = false;
    static const uint64_t call_budget = getenv("TRRE_GUARD_CALL_BUDGET") ? (uint64_t)atoll(getenv("TRRE_GUARD_CALL_BUDGET")) : kGuardCallBudget;
    uint64_t steps_spent = 0;
    bool undecided = false;
    for (int64_t w = 0; w < n_win && !found;) {
        if (steps_spent > call_budget) {            // suspects are left and the call's budget is spent: they are not decided — and the caller is told
            for (int64_t x = w; x < n_win && !undecided; x = (x | 63) + 1)
                if (flags[(size_t)(x >> 6)] >> (x & 63)) undecided = true;
            break;
        }
        runs.clear();
        while (w < n_win && runs.size() < kGuardMaxRuns) {
            const uint64_t word = flags[(size_t)(w >> 6)] >> (w & 63);
            if (!word) { w = (w | 63) + 1; continue; }
            if (!(word & 1u)) { w += __builtin_ctzll(word); continue; }
            int64_t e = w;
            while (e + 1 < n_win && ((flags[(size_t)((e + 1) >> 6)] >> ((e + 1) & 63)) & 1u)) ++e;
            runs.push_back(GuardRun{(uint32_t)w, (uint32_t)e});
            w = e + 1;
        }
        if (runs.empty()) break;
        const size_t n_runs = runs.size();
        rc = guard_room(&cx->d_gruns, &cx->gruns_cap, n_runs * (sizeof(GuardRun) + sizeof(GuardResult)));      // (runs and results share one allocation)
        if (rc) return rc;
        if (!cx->d_gstack && hipMalloc(reinterpret_cast<void**>(&cx->d_gstack), (size_t)kGuardSlots * kGuardStackMax * 12) != hipSuccess)
            return fail(TRRE_E_TOO_BIG, "error: out of device memory (stack guard)");
        d_res = reinterpret_cast<GuardResult*>(cx->d_gruns);
        d_runs = reinterpret_cast<GuardRun*>(cx->d_gruns + n_runs * sizeof(GuardResult));
        HIP_TRY(hipMemcpyAsync(d_runs, runs.data(), n_runs * sizeof(GuardRun), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_res, 0, n_runs * sizeof(GuardResult), stream));
        ga = GuardArgs{};
        ga.blob = st->d_kblob;
        ga.runs = d_runs;
        ga.results = d_res;
        ga.stack = cx->d_gstack;
        static const uint64_t budget = getenv("TRRE_GUARD_BUDGET") ? (uint64_t)atoll(getenv("TRRE_GUARD_BUDGET")) : kGuardBudget;
        ga.budget = budget;
        ga.obuf = nullptr;                    // (the search alone: PROD writes nothing, FINAL prints nothing)
        ga.obuf_cap = 0xffffffffu;
        launch_guard(false, args, ga, (int64_t)n_runs, kGuardSlots, stream);
        HIP_TRY(hipGetLastError());
        res.resize(n_runs);
        HIP_TRY(hipMemcpyAsync(res.data(), d_res, n_runs * sizeof(GuardResult), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (size_t k = 0; k < n_runs && !found; ++k) {
            if (res[k].status == 1u) { bad = k; found = true; break; }
            if (res[k].status == 2u) undecided = true;          // this line's search outran its budget (or the line has more than 4 GiB)
            steps_spent += res[k].bad_at;                       // (a line that did not overflow leaves its step count there)
        }
    }
    // (lines BEFORE a line that overflows and not decided themselves: the answer given is the first overflow found — flagged too)
    if (undecided) g_scan_flags |= TRRE_SCAN_GUARD_UNDECIDED;
    if (!found) return TRRE_OK;
    hit->hit = true;
    hit->line_start = res[bad].line_start;
    hit->part = 0xffffffffu;
    // the same line again, printing: what the reference had printed of it when it gave up
    const uint64_t line_len = res[bad].out_len;             // (the search pass leaves the line's length here)
    const uint64_t obuf_want = 16 * (line_len + 1) + 65536, out_want = 16 * (res[bad].bad_at + 1) + 65536;
    if (obuf_want > (1ull << 30) || out_want > (1ull << 30)) return TRRE_OK;
    if (guard_room(&cx->d_gobuf, &cx->gobuf_cap, (size_t)obuf_want) || guard_room(&cx->d_gout, &cx->gout_cap, (size_t)out_want)) return TRRE_OK;
    ga.runs = d_runs + bad;
    ga.results = d_res + bad;
    ga.obuf = cx->d_gobuf;
    ga.obuf_cap = (uint32_t)std::min<uint64_t>(cx->gobuf_cap, 0xfffffff0u);
    ga.out = cx->d_gout;
    ga.out_cap = cx->gout_cap;
    launch_guard(true, args, ga, 1, 1, stream);
    HIP_TRY(hipGetLastError());
    GuardResult again{};
    HIP_TRY(hipMemcpyAsync(&again, d_res + bad, sizeof again, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (again.status == 1u) hit->part = again.out_len;
    return TRRE_OK;
}

// the backtracking fallback: sub-range per thread, frames (= bytes an attempt may consume) and path bytes per thread, workgroups in the pool, steps per sub-range
constexpr int64_t kBtLaneBytes = 1024;
constexpr uint32_t kBtBudget = 16u << 20;
// The pool of stacks and path buffers sizes the scratch, not the input: 3.4-4.6 GB in every tier.  A launch in which an attempt outgrew its
// stack or its path buffer runs again on the next tier (round 4 gave up at 4 096 bytes / 4 KiB of output).  The search waits on memory — a
// stack frame, list bounds, a follow entry, the byte, its set: dependent loads — so the first tier is sized for the MACHINE: 1 536 workgroups
// are six waves per SIMD (k_bt's registers allow six), and 512 frames are what such a pool can have; the second tier is round 4's
// (one wave per SIMD: a sixth of the rate).
struct BtTier { int64_t pool_blocks; uint32_t frames, path_cap; };
constexpr int kBtTierCount = 4;
constexpr BtTier kBtTiers[kBtTierCount] = {{1536, 512, 512}, {256, 4096, 4096}, {16, 65536, 65536}, {1, 1u << 20, 1u << 20}};
// ---- the deterministic engine on tables still being built (front.hpp: LazyDft, lazy_block.hpp) ----
constexpr int64_t kLazyLaneBytes = 1024;
constexpr uint32_t kLazyMissCap = 1u << 16;
static_assert(trre::kLazyMissWords == trre::kLazyRecWords, "miss record layout");
constexpr uint64_t kLazyBudget = 1ull << 32;             // table steps per sub-range (an attempt per byte of a long line is quadratic, as in the reference)

int lazy_ensure(trre_prog* p) {
    std::lock_guard<std::mutex> lock(p->lazy_mu);
    if (p->lazy) return TRRE_OK;
    try {
        trre::LazyLimits lim;
        if (const char* mb = getenv("TRRE_LAZY_MAX_BYTES")) lim.max_bytes = (size_t)atoll(mb);
        if (const char* ss = getenv("TRRE_LAZY_SEED_STATES")) lim.seed_states = (size_t)atoll(ss);
        p->lazy.reset(new trre::LazyDft(p->dft_nft, lim));
    } catch (const trre::Error& e) {
        return fail(e.code, e.what());
    } catch (const std::bad_alloc&) {
        return fail(TRRE_E_TOO_BIG, "error: out of memory while determinising");
    }
    return TRRE_OK;
}

// brings the device's copy of the tables up to the host's: the pool's new bytes, then the new rows, then the old rows that changed (an entry
// must never name a row or a text the device does not hold yet — scans of other chunks may be reading the tables meanwhile)
int lazy_sync(trre_prog* p, DeviceState* st) {
    std::lock_guard<std::mutex> lock(p->lazy_mu);
    const trre::LazyDft& z = *p->lazy;
    const uint32_t n_cls = z.n_cls(), rows = z.n_rows();
    const size_t pool = z.pool_bytes();
    if (!st->d_lcls) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st->d_lcls), 256));
        HIP_TRY(hipMemcpy(st->d_lcls, z.cls(), 256, hipMemcpyHostToDevice));
    }
    if (pool > st->lpool_cap) {
        size_t cap = st->lpool_cap ? st->lpool_cap : (size_t)1 << 20;
        while (cap < pool) cap *= 2;
        uint8_t* fresh = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&fresh), cap));
        if (st->d_lpool) st->retired.push_back(st->d_lpool);
        st->d_lpool = fresh;
        st->lpool_cap = cap;
        st->lazy_pool_up = 0;
    }
    if (rows > st->lent_cap) {
        size_t cap = st->lent_cap ? st->lent_cap : 4096;
        while (cap < rows) cap *= 2;
        uint64_t* fresh = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&fresh), cap * n_cls * 8));
        if (st->d_lent) st->retired.push_back(st->d_lent);
        st->d_lent = fresh;
        st->lent_cap = cap;
        st->lazy_rows_up = 0;
    }
    if (pool > st->lazy_pool_up) {
        HIP_TRY(hipMemcpy(st->d_lpool + st->lazy_pool_up, z.pool() + st->lazy_pool_up, pool - st->lazy_pool_up, hipMemcpyHostToDevice));
        st->lazy_pool_up = pool;
    }
    if (rows > st->lazy_rows_up) {
        HIP_TRY(hipMemcpy(st->d_lent + (size_t)st->lazy_rows_up * n_cls, z.ent() + (size_t)st->lazy_rows_up * n_cls, (size_t)(rows - st->lazy_rows_up) * n_cls * 8,
                          hipMemcpyHostToDevice));
    }
    const uint32_t dirty = z.first_dirty_row(st->lazy_epoch_up);
    if (dirty < st->lazy_rows_up)
        HIP_TRY(hipMemcpy(st->d_lent + (size_t)dirty * n_cls, z.ent() + (size_t)dirty * n_cls, (size_t)(st->lazy_rows_up - dirty) * n_cls * 8, hipMemcpyHostToDevice));
    st->lazy_rows_up = rows;
    st->lazy_epoch_up = z.epoch();
    return TRRE_OK;
}

// one round: the count pass (fresh: every lane; else the lanes that are still void), the exclusive sum, the emit pass (which leaves at once
// when the count pass met a miss)
int lazy_round(trre_prog* p, DeviceState* st, ScanCtx* cx, const trre::ScanArgs& args, int64_t n_chunks, bool fresh, hipStream_t stream) {
    using namespace trre;
    int rc = lazy_sync(p, st);
    if (rc) return rc;
    if (!cx->d_miss) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cx->d_miss), (size_t)(2 + trre::kLazyMissWords * kLazyMissCap) * 4));
    HIP_TRY(hipMemsetAsync(cx->d_miss, 0, 8, stream));
    if (fresh) HIP_TRY(hipMemsetAsync(cx->d_lane_counts, 0xff, (size_t)n_chunks * 256 * 4, stream));
    LazyArgs la{};
    la.cls = st->d_lcls;
    la.ent = st->d_lent;
    la.pool = st->d_lpool;
    la.n_cls = p->lazy->n_cls();
    la.n_rows = st->lazy_rows_up;
    la.miss = cx->d_miss;
    la.miss_cap = kLazyMissCap;
    static std::atomic<uint32_t> launch_id{0};
    la.gen = ++launch_id;                     // (lazy_block.hpp: marks of other launches — other chunks in flight on the same table — are listed again)
    static const uint64_t budget = getenv("TRRE_LAZY_BUDGET") ? (uint64_t)atoll(getenv("TRRE_LAZY_BUDGET")) : kLazyBudget;
    la.budget = budget;
    launch_lazy(1, args, la, kLazyLaneBytes, n_chunks, stream);
    launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, n_chunks, stream);
    launch_lazy(2, args, la, kLazyLaneBytes, n_chunks, stream);
    HIP_TRY(hipGetLastError());
    return TRRE_OK;
}

// Launches of the memoryless kernel wait for one another, device by device: the kernel wants its whole grid resident (a tile's look-back waits for
// the tiles of the other workgroups), and two of them on two streams — the host path's chunks in flight — would each hold a part of the machine
// and wait for the rest until their spins ran out.  (Other kernels beside it only delay it: they end, its workgroups move in.)
struct MapGenChain { std::mutex mu; hipEvent_t ev = nullptr; bool any = false; };
MapGenChain g_mapgen_chain[64];

// the memoryless kernel's workspace and launch (map_block.hpp); 0: launched (pd.total_at set), else the caller takes another route
int mapgen_launch(trre_prog* p, ScanCtx* cx, const trre::ScanArgs& args, const trre::StreamTables& stt, hipStream_t stream, Pending& pd) {
    using namespace trre;
    MapGenArgs oa{};
    oa.n_tiles = (args.vend + kMapGenTile - 1) / kMapGenTile;
    const size_t groups = (size_t)(oa.n_tiles / 64 + 1);
    static const char* dbg_env = getenv("TRRE_MAPGEN_DBG");                // (a file: every tile's total and place, written by finish())
    const size_t words = (size_t)oa.n_tiles + 2 * groups + 1 + 8 + (dbg_env ? 16 * (size_t)oa.n_tiles : 0);   // descriptors, group sums, group totals, the total, phase clocks
    if (cx->mg_tiles < oa.n_tiles || dbg_env) {
        if (cx->d_mg) (void)hipFree(cx->d_mg);
        cx->d_mg = nullptr; cx->mg_tiles = 0;
        if (hipMalloc(reinterpret_cast<void**>(&cx->d_mg), words * 8) != hipSuccess) { (void)hipGetLastError(); return -1; }
        cx->mg_tiles = oa.n_tiles;
    }
    oa.desc = cx->d_mg;
    oa.gsum = cx->d_mg + oa.n_tiles;
    oa.ginc = oa.gsum + groups;
    oa.total = oa.ginc + groups;
    // the windows (two: a tile is stored a trip after it was expanded): the whole output of a tile as a rule — 16 KiB of input and an eighth: four
    // workgroups' pairs of windows share a CU's 160 KiB; a tile that prints more takes the clipped path, window by window
    static const int window_env = getenv("TRRE_MAPGEN_WINDOW") ? atoi(getenv("TRRE_MAPGEN_WINDOW")) : 0;
    uint32_t window = window_env > 0 ? (uint32_t)window_env : (uint32_t)(kMapGenTile + kMapGenTile / 8);
    oa.window = (window + 15u) & ~15u;
    oa.spin = 1u << 16;
    if (getenv("TRRE_MAPGEN_NOLB")) oa.spin = 7u;      // (an experiment: no look-back — the pace of the rest; the output is void)
    static const bool mg_prof = getenv("TRRE_MAPGEN_PROF") != nullptr;       // (phase clocks of the kernel, printed by finish())
    oa.prof = mg_prof ? oa.total + 1 : nullptr;
    oa.dbg = dbg_env ? oa.total + 1 + 8 : nullptr;
    oa.longest = stt.mg_max;
    oa.first_lookup = 0;
    for (int c = 0; c < 256; ++c)
        if ((stt.mg[4 * c + 2] & (15u | kMgNul)) == 1u && (stt.mg[4 * c] & 0xffu) != (uint32_t)c) oa.first_lookup = 1;   // (a NUL voids the launch: whatever it prints)
    if (hipMemsetAsync(cx->d_mg, 0, words * 8, stream) != hipSuccess) { (void)hipGetLastError(); return -1; }
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        MapGenChain& ch = g_mapgen_chain[dev & 63];
        std::lock_guard<std::mutex> lk(ch.mu);
        if (!ch.ev && hipEventCreateWithFlags(&ch.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ch.ev = nullptr; return -1; }
        if (ch.any && hipStreamWaitEvent(stream, ch.ev, 0) != hipSuccess) { (void)hipGetLastError(); return -1; }
        if (launch_mapgen(args, oa, stream) != 0) return -1;
        if (hipEventRecord(ch.ev, stream) != hipSuccess) { (void)hipGetLastError(); return -1; }
        ch.any = true;
    }
    pd.total_at = oa.total;
    pd.mapgen = true;
    (void)p;
    return 0;
}

int enqueue(trre_prog* p, DeviceState* st, ScanCtx* cx, int family, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap,
            hipStream_t stream) {
    using namespace trre;
    Pending& pd = cx->pend;
    // Back-to-back enqueues without a finish() in between form one batch: status bits accumulate, the
    // timing events bracket the whole batch and finish() speaks for the last launch — so every launch of
    // a batch must be the same scan (benchmark loops); anything else has to be finished first.
    const bool batch = pd.active && pd.launched;
    if (pd.active && (pd.asked != family || pd.d_in != d_in || pd.d_out != d_out || pd.n != n || pd.cap != cap || pd.stream != stream))
        return fail(TRRE_E_ARG, "error: a different scan is still in flight on this device: call trre_scan_finish first");
    const int batch_count = batch ? pd.count : 0;
    // an in-place scan destroys what the stack guard would look at: it looks first (the first launch of a batch only — the rest
    // are the same scan).  A line the reference's search fails on: nothing is launched, finish() answers.
    if (!batch && n && d_in == d_out && guard_applies(*p, *cx, n)) {
        GuardHit hit;
        const int grc = guard_check(p, st, cx, d_in, n, stream, &hit);
        if (grc) return grc;
        if (hit.hit) {
            pd = Pending();
            pd.active = true;
            pd.family = family; pd.asked = family;
            pd.d_in = d_in; pd.d_out = d_out; pd.n = n; pd.cap = cap; pd.stream = stream;
            pd.guard_hit = true; pd.guard_line = hit.line_start; pd.guard_part = hit.part;
            return TRRE_OK;
        }
    }
    pd = Pending();
    pd.active = true;
    pd.count = batch_count;
    pd.family = family; pd.asked = family;
    pd.d_in = d_in; pd.d_out = d_out; pd.n = n; pd.cap = cap; pd.stream = stream;
    if (n == 0) return TRRE_OK;
    int rc;
    // Long lines (round 5): is there a sample point without a line end within 32 KiB?  Asked once per buffer (a tiny kernel and a wait), for the
    // families that have the exact sub-ranges to answer with; a length-preserving family then runs as its general sibling (count, exclusive sum,
    // emit: the exact sub-ranges live there), unless the scan is in place.
    static const int exact_env0 = getenv("TRRE_EXACT") ? atoi(getenv("TRRE_EXACT")) : -1;
    {
        const bool small_tables = (family == TRRE_KERNEL_STREAM_LP && p->stt.g16_ok) || (family == TRRE_KERNEL_GUIDED_LP && p->gt.fwd.g16_ok && !p->gt.wide);
        if (small_tables && !cx->exact_off && exact_env0 != 0 && d_in != d_out && cap >= n) {
            if (cx->probe_in != d_in || cx->probe_n != n) {
                const int64_t a0 = (int64_t)(reinterpret_cast<uintptr_t>(d_in) & 15u);
                ScanArgs pa{};
                pa.in_v0 = d_in - a0; pa.vbeg = a0; pa.vend = a0 + (int64_t)n;
                if (!cx->d_probe) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cx->d_probe), 16));
                uint32_t misses = 0;
                HIP_TRY(hipMemsetAsync(cx->d_probe, 0, 4, stream));
                launch_line_probe(pa, 32768, cx->d_probe, stream);
                HIP_TRY(hipMemcpyAsync(&misses, cx->d_probe, 4, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                cx->probe_in = d_in; cx->probe_n = n;
                cx->long_lines = misses != 0;
            }
            const bool longl = exact_env0 == 2 || cx->long_lines;       // (TRRE_EXACT=2: the switch whatever the lines, for A/B runs and tests)
            if (longl) {
                family = family == TRRE_KERNEL_STREAM_LP ? TRRE_KERNEL_STREAM_GEN : TRRE_KERNEL_GUIDED_GEN;
                pd.family = family;
            }
        }
    }

    const int64_t a = (int64_t)(reinterpret_cast<uintptr_t>(d_in) & 15u);
    ScanArgs args{};
    args.in_v0 = d_in - a;
    args.out_v0 = d_out - a;
    args.out = d_out;
    args.vbeg = a;
    args.vend = a + (int64_t)n;
    args.blob = is_stream(family) ? st->d_sblob : (is_guided(family) ? st->d_gblob : st->d_blob);
    const trre::StreamTables& stt = is_guided(family) ? p->gt.fwd : p->stt;       // the stream-form tables this launch walks
    args.status = cx->d_status;
    args.cap = cap;
    // a mask scratch left by an earlier, smaller scan must not be used: the kernel asks for one again
    args.gscratch = cx->scratch_bytes >= (n + 32) * (size_t)p->mask_bytes ? cx->d_scratch : nullptr;
    const bool backtrack = family == TRRE_KERNEL_BACKTRACK;
    const bool lazy = family == TRRE_KERNEL_DFT_LAZY;
    if (lazy) { rc = lazy_ensure(p); if (rc) return rc; }
    // (the stream and guided families run the direct kernels — a lane walks a long sub-range straight from memory —: their chunks are
    // counted below, once the sub-range is known.  Rounds 1-5 also kept LDS-tile kernels for the stream tables, TRRE_STREAM_IMPL=0:
    // 10-100 GB/s, no job since round 2, removed in round 6)
    const int chunk = backtrack ? (int)kBtLaneBytes * 256 : lazy ? (int)kLazyLaneBytes * 256
                      : is_stream(family) ? 2048 * direct_block_threads() : chunk_bytes(p->engine, p->mask_bytes);
    const int threads = backtrack || lazy ? 256 : is_stream(family) ? direct_block_threads() : block_threads(p->engine, p->mask_bytes);
    int64_t n_chunks = (args.vend + chunk - 1) / chunk;
    static const int64_t lane_bytes_env = getenv("TRRE_LANE_BYTES") ? atoll(getenv("TRRE_LANE_BYTES")) : 0;
    // the positional-window kernel (wave-tiled I/O) for length-preserving tables that have the window form, the direct walkers otherwise
    const bool window = is_stream(family) && family == TRRE_KERNEL_STREAM_LP && p->stt.lpw_ok &&
                        (reinterpret_cast<uintptr_t>(args.out_v0) & 15u) == 0;   // its 16-byte stores: in and out congruent mod 16 (unaligned they work, at the pace
                                                                                 // of the emit pass alone: 0.94 against 0.98 ms per GiB, round 4)
    // sub-range per lane: 2 KiB, growing with the input so that about half a million lanes (8192 waves)
    // walk it — per-lane costs (the skipped head, the tail beyond the sub-range, the wave waiting for its
    // slowest lane) shrink with longer lanes: cfg 4 at 8 GiB 1.72 TB/s with 2 KiB lanes, 2.06 with 16 KiB
    int64_t lane_auto = 2048;
    if (window)
        while (lane_auto < 16384 && (int64_t)n / (lane_auto * 2) >= 524288) lane_auto *= 2;
    // (the count / emit pair on the 16-byte entries gains too — 'a:xyz' at 8 GiB: 920 GB/s with 2 KiB lanes, 967 with 8 KiB, 946 with
    // 32 KiB; ' +: ' 896 / 947 / 805 — and so does the mark + splice pair of a large table, its event lists growing with the sub-range:
    // cfg 5 at 4 GiB 594 / 663 (4 KiB) / 675; the guided families do not: 647 / 641)
    // (the guided families' forward passes gain 2 % at 8 GiB, their backward pass loses 15 %: it keeps 2 KiB)
    if ((family == TRRE_KERNEL_STREAM_GEN && (p->stt.g16_ok || p->stt.fb_ok)) || (is_guided(family) && !p->gt.wide))
        while (lane_auto < 8192 && (int64_t)n / (lane_auto * 2) >= 262144) lane_auto *= 2;
    const int64_t lane_bytes = lane_bytes_env > 0 ? (lane_bytes_env + 127) / 128 * 128 : lane_auto;
    const bool direct = is_stream(family) || is_guided(family);
    const int64_t rev_lane_bytes = lane_bytes_env > 0 ? lane_bytes : 2048;
    const bool direct_ent_lds = stt.ok && stt.ent.size() * 8 <= (size_t)direct_ent_lds_bytes();
    const bool g16_slow = (stt.flags & kFlagG16Slow) != 0;
    static const bool no_g16_env = getenv("TRRE_NO_G16") != nullptr;       // A/B: the 8-byte entries
    static const bool no_fb_env = getenv("TRRE_NO_FB") != nullptr;         // A/B: large tables walk their 8-byte rows in both passes
    // memoryless programs in ONE pass (map_block.hpp) — `[aie]:` in 6.2 ms per 8 GiB where the pair takes 7.6, HTML escapes 8.4 against 11.0.
    // A program whose longer texts turn out to be frequent (finish(): the output 4 % longer than the input; `a:xyz` on text: 8.6 ms against 8.0)
    // goes back to the pair for the context's later scans.  TRRE_MAPGEN=1: every memoryless program, always; TRRE_MAPGEN=0: none
    static const int mapgen_env = getenv("TRRE_MAPGEN") ? atoi(getenv("TRRE_MAPGEN")) : -1;
    const bool mapgen_on = stt.mg_max != 0 && (mapgen_env < 0 ? !(stt.mg_max > 1u && cx->mapgen_dense) : mapgen_env != 0) && !cx->mapgen_off && cx->mapgen_voids < 2;
    const int sym_mode = !is_guided(family) ? 0 : (p->gt.sym_bits == 4 && !no_g16_env ? 2 : 1);
    if (direct) n_chunks = (args.vend + lane_bytes * direct_block_threads() - 1) / (lane_bytes * direct_block_threads());

    if (!is_gen(family) && cap < n) return TRRE_OK;   // finish() reports the capacity error
    if (is_gen(family)) {
        rc = ensure_workspace(cx, n_chunks, direct ? direct_block_threads() : threads);
        if (rc) return rc;
        args.lane_counts = cx->d_lane_counts;
        args.chunk_total = cx->d_chunk_total;
        args.chunk_base = cx->d_chunk_base;
    }
    if (is_guided(family)) {
        // one symbol per input byte, written by the backward pass (whole 64-byte pieces) and read by the forward pass
        const size_t need = ((size_t)((args.vend + 63) & ~(int64_t)63) + 256) * (p->gt.wide ? 2 : 1);
        if (cx->sym_bytes < need) {
            if (cx->d_sym) (void)hipFree(cx->d_sym);
            cx->d_sym = nullptr; cx->sym_bytes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cx->d_sym), need));
            cx->sym_bytes = need;
        }
        args.rblob = st->d_rblob;
        args.sym_v0 = cx->d_sym;
    }
    if (!batch) {
        HIP_TRY(hipMemsetAsync(cx->d_status, 0, 24, stream));          // (status words and the NUL count)
        if (p->profiling) HIP_TRY(hipEventRecord(cx->ev0, stream));
    }
    pd.timed = p->profiling;
    if (lazy) {
        // tables still being built (lazy_block.hpp): the first round; finish() builds what it missed and runs the next ones
        rc = lazy_round(p, st, cx, args, n_chunks, true, stream);
        if (rc) return rc;
        pd.total_at = cx->d_chunk_base + n_chunks;
    } else if (backtrack) {
        // the search itself (gen_block.hpp: bt_lane): a pool of workgroups takes the chunks of 256 sub-ranges in turn, each thread
        // with a stack and a path buffer of its own for the whole launch
        const BtTier& tier = kBtTiers[cx->bt_tier];
        const int64_t pool_blocks = n_chunks < tier.pool_blocks ? n_chunks : tier.pool_blocks;
        const uint32_t kBtFrames = tier.frames, kBtPathCap = tier.path_cap;
        const size_t need = (size_t)pool_blocks * 256 * ((size_t)kBtFrames * 16 + kBtPathCap);
        if (cx->scratch_bytes < need) {
            if (cx->d_scratch) (void)hipFree(cx->d_scratch);
            cx->d_scratch = nullptr; cx->scratch_bytes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cx->d_scratch), need));
            cx->scratch_bytes = need;
        }
        GenArgs ga{};
        ga.stack = reinterpret_cast<uint32_t*>(cx->d_scratch);
        ga.path = cx->d_scratch + (size_t)pool_blocks * 256 * kBtFrames * 16;
        ga.frames = kBtFrames;
        ga.path_cap = kBtPathCap;
        args.blob = st->d_nblob;
        static const uint32_t budget = getenv("TRRE_BT_BUDGET") ? (uint32_t)atoll(getenv("TRRE_BT_BUDGET")) : kBtBudget;
        launch_bt(1, args, ga, kBtLaneBytes, n_chunks, pool_blocks, budget, stream);
        launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, n_chunks, stream);
        launch_bt(2, args, ga, kBtLaneBytes, n_chunks, pool_blocks, budget, stream);
        pd.total_at = cx->d_chunk_base + n_chunks;
    } else if (family == TRRE_KERNEL_BYTEMAP) {
        args.nul_list = cx->d_status + 4;
        launch_bytemap(args, stream);
    } else if (family == TRRE_KERNEL_TILE_LP) {
        launch_tile_kernel(0, p->engine, p->mask_bytes, args, n_chunks, stream);
    } else if (window) {
        const int64_t n_lanes = (args.vend + lane_bytes - 1) / lane_bytes;
        if (cx->redo_lanes < n_lanes) {
            if (cx->d_redo) (void)hipFree(cx->d_redo);
            cx->d_redo = nullptr; cx->redo_lanes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cx->d_redo), (size_t)(n_lanes + 1) * 4));
            cx->redo_lanes = n_lanes;
        }
        HIP_TRY(hipMemsetAsync(cx->d_redo, 0, 4, stream));
        args.redo = cx->d_redo;
        // (the pair form of the window entries where the tables have it: two input bytes per table read.  TRRE_NO_LPW_PAIR=1: A/B runs)
        static const bool no_pair_env = getenv("TRRE_NO_LPW_PAIR") != nullptr;
        launch_lpw_kernel((int)(p->stt.lpw.size() * 4), p->stt.lpw_delay > 3, direct_ent_lds, args, lane_bytes, stream,
                          p->stt.lpw2_ok && !no_pair_env ? (int)(p->stt.lpw2.size() * 4) : 0);
    } else if (is_guided(family) && p->gt.wide) {
        // a backward DFA of more than 256 states: 16-bit symbols, both tables through L1 / L2 (scan_block.hpp: wide guided tables)
        launch_rev_wide(args, lane_bytes, stream);
        launch_wide_fwd(1, args, lane_bytes, n_chunks, stream);
        launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, n_chunks, stream);
        launch_wide_fwd(2, args, lane_bytes, n_chunks, stream);
        pd.total_at = cx->d_chunk_base + n_chunks;
    } else if (direct && !is_gen(family)) {
        // length-preserving without a window form: the emit pass alone, every lane writing its lines where it read them
        const int g16 = stt.g16_ok && !no_g16_env ? (int)(align_up(stt.g16.size() * 4, 16) + stt.p32.size() * 4) : 0;   // LDS room: 16-byte + pair forms
        if (is_guided(family)) launch_rev_sweep(args, (int)p->gt.n_rev * 256, rev_lane_bytes, stream, sym_mode == 2);
        args.lp_emit = 1;
        launch_direct_kernel(2, direct_ent_lds, args, lane_bytes, n_chunks, stream, g16, sym_mode, g16_slow);
    } else if (direct && family == TRRE_KERNEL_STREAM_GEN && mapgen_on &&
               mapgen_launch(p, cx, args, stt, stream, pd) == 0) {
        // a MEMORYLESS program (map_block.hpp; round 6): no state, so no walk — lengths, a prefix sum with look-back, the bytes' texts
        // at their places: ONE pass, one read of the input.  A NUL voids it (finish()).
    } else if (direct && !is_guided(family) && stt.fb_ok && !no_fb_env) {
        // a large table (a dictionary) in its fallback form: the count pass with every per-byte lookup in LDS (0.86 ms per
        // GiB against 1.62 on the 8-byte rows through L1/L2).  The emit pass over the same form (TRRE_FB_EMIT=1) is
        // correct but slower than the one over the 8-byte rows (3.8 against 2.9 ms): its tables leave LDS for only 512
        // staging rings per CU.  TRRE_NO_FB=1: both passes on the 8-byte rows, for A/B runs.
        static const bool fb_emit_env = getenv("TRRE_FB_EMIT") != nullptr;
        const bool fb_emit = fb_emit_env && fb_fits(p->sblob.data());         // (the form next to 512 staging rings)
        // The copy form (default where the tables have it): the comb walk also lists where the replacement texts go, the
        // second pass copies the input around them without walking any table (scan_block.hpp).  TRRE_NO_FB_COPY=1: the
        // count pass + the emit pass on the 8-byte rows, for A/B runs and what finish() falls back to (a NUL in the input,
        // more texts in a sub-range than its event list holds).
        static const bool no_copy_env = getenv("TRRE_NO_FB_COPY") != nullptr;
        // (the second pass is the wave-cooperative splice of splice_block.hpp; a table with an escape text longer than its tiles take — more than
        // 255 bytes — or too large for its LDS has no copy form: the pair below)
        uint32_t esc_max = 0;
        for (size_t k = 0; k + 3 < p->stt.fb_esc.size(); k += 4) esc_max = std::max(esc_max, p->stt.fb_esc[k + 1]);
        if (!no_copy_env && !fb_emit_env && !cx->patch_off && !p->copy_form_off.load() && lane_bytes % 64 == 0 && fb_copy_fits(p->sblob.data()) &&
            esc_max <= kSpMaxText && fb_splice_fits(p->sblob.data())) {
            const int64_t n_lanes = n_chunks * direct_block_threads();
            const int64_t ev_rows = (lane_bytes + 2047) / 2048;                  // (the event list grows with the sub-range)
            rc = ensure_copy_workspace(cx, n_lanes * ev_rows);
            if (rc) return rc;
            FbCopyArgs ca{};
            ca.events = cx->d_cevents;
            ca.lane_hdr = cx->d_chdr;
            ca.ev_cap = kCopyEvCap * (uint32_t)ev_rows;
            launch_fb_mark(args, ca, p->sblob.data(), lane_bytes, n_chunks, stream);
            launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, n_chunks, stream);
            launch_fb_splice(args, ca, p->sblob.data(), lane_bytes, n_chunks, stream);
            pd.total_at = cx->d_chunk_base + n_chunks;
            pd.patched = true;
        } else {
        launch_fb_kernel(1, args, p->sblob.data(), lane_bytes, n_chunks, stream);
        launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, n_chunks, stream);
        if (fb_emit) launch_fb_kernel(2, args, p->sblob.data(), lane_bytes, n_chunks, stream);
        else launch_direct_kernel(2, direct_ent_lds, args, lane_bytes, n_chunks, stream, 0, sym_mode, g16_slow);
        pd.total_at = cx->d_chunk_base + n_chunks;
        }
    } else if (direct) {
        const int g16 = stt.g16_ok && !no_g16_env ? (int)(align_up(stt.g16.size() * 4, 16) + stt.p32.size() * 4) : 0;   // LDS room: 16-byte + pair forms
        // Exact sub-ranges (round 5): every lane walks the bytes of its sub-range and nothing else, from the state the transducer is in
        // there — what makes a line of 400 KB as parallel as 4 000 lines of 100 bytes (rounds 1-4: a lane owns the lines that START in its
        // sub-range and walks them to their end alone, and the backward pass of the guided families has ONE thread carry the state through
        // a long line: 5.8-9.0 GB/s on such lines; 1 GiB of 400 KB lines now: ' +: ' 6.2 -> 700 GB/s, '(a|b)*c:x' 4.9 -> 520).  On ordinary text the
        // form costs nothing measurable (1 GiB: 745 against 733 GB/s), so the general families always run it — also what keeps ONE giant line in
        // a file of short ones from serialising the scan; TRRE_EXACT=0: the old ownership, for A/B runs.
        static const int exact_env = getenv("TRRE_EXACT") ? atoi(getenv("TRRE_EXACT")) : -1;
        bool use_exact = g16 > 0 && !cx->exact_off && exact_env != 0 && lane_bytes % 128 == 0 && rev_lane_bytes % 128 == 0 &&
                         !is_guided_wide(*p, family);
        const int64_t rev_lanes = ((((args.vend + 127) & ~(int64_t)127) + rev_lane_bytes - 1) / rev_lane_bytes + 255) / 256 * 256;
        if (use_exact) {
            const int64_t n_lanes = n_chunks * direct_block_threads();
            const int64_t want = 3 * n_lanes + (is_guided(family) ? 2 * rev_lanes : 0);
            if (cx->spec_lanes < want) {
                if (cx->d_spec) (void)hipFree(cx->d_spec);
                cx->d_spec = nullptr; cx->spec_lanes = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cx->d_spec), (size_t)want * 4));
                cx->spec_lanes = want;
            }
            args.entry_rows = cx->d_spec;
            args.exit_rows = cx->d_spec + n_lanes;
            args.spec_flags = cx->d_spec + 2 * n_lanes;
            args.rev_guess = cx->d_spec + 3 * n_lanes;
            args.rev_flags = cx->d_spec + 3 * n_lanes + rev_lanes;
            args.exact = 1;
            args.spec_look = (uint32_t)kSpecLook;
        }
        if (is_guided(family)) {
            launch_rev_sweep(args, (int)p->gt.n_rev * 256, rev_lane_bytes, stream, sym_mode == 2);
            if (use_exact) launch_rev_verify(args, rev_lane_bytes, stream, sym_mode == 2);
        }
        // ONE walk (round 6; one_block.hpp — SURVEY.md row f2): lanes of 128 bytes from guessed-and-verified states, a workgroup's whole output
        // staged in LDS, its place in the output by look-back over the workgroups' totals: the input is read once and walked once.  What the
        // kernel cannot answer in place — a lane whose output outgrows its LDS region, a workgroup whose first lane guessed wrong, a replacement
        // text of more than 8 bytes, a wrong guess of the backward pass — voids the launch (kStOneVoid) and finish() runs the pair below;
        // after two such launches the program stops trying.
        // MEASURED (profiles/r06_expand_one_*): bit-identical output, 1.05 GiB fetched and 1.06 GiB written per GiB (the pair:
        // 2.2 x that) — and HALF the pair's speed (8 GiB of 'a:xyz': 15.7 ms against 7.9): with a workgroup's output held in LDS only three
        // waves per SIMD are resident, and the walk is a chain of dependent table reads that wants twice that to hide.  So the pair stays
        // the default and this form is opt-in: TRRE_ONE=1.
        static const int one_env = getenv("TRRE_ONE") ? atoi(getenv("TRRE_ONE")) : 0;
        static const int one_lane_env = getenv("TRRE_ONE_LANE") ? atoi(getenv("TRRE_ONE_LANE")) : 0;
        static const int one_region_env = getenv("TRRE_ONE_REGION") ? atoi(getenv("TRRE_ONE_REGION")) : 0;
        static const int one_look_env = getenv("TRRE_ONE_LOOK") ? atoi(getenv("TRRE_ONE_LOOK")) : 0;
        if (use_exact && one_env == 1 && !cx->one_off && p->one_fails.load() < 2 && stt.max_out <= 8u) {
            OneArgs oa{};
            oa.lane_bytes = one_lane_env > 0 ? (uint32_t)((one_lane_env + 63) / 64 * 64) : 128u;
            oa.region = one_region_env > 0 ? (uint32_t)(one_region_env / 8 * 8 + 4) : 172u;          // (an odd number of dwords)
            oa.look = one_look_env > 0 ? (uint32_t)((one_look_env + 15) / 16 * 16) : 32u;
            oa.spin = 1u << 20;
            oa.n_tiles = (args.vend + (int64_t)oa.lane_bytes * kOneThreads - 1) / ((int64_t)oa.lane_bytes * kOneThreads);
            if (cx->one_tiles < oa.n_tiles) {
                if (cx->d_one) (void)hipFree(cx->d_one);
                cx->d_one = nullptr; cx->one_tiles = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cx->d_one), (size_t)(oa.n_tiles + 2 + 8 + 2 * (oa.n_tiles / 32 + 1)) * 8));
                cx->one_tiles = oa.n_tiles;
            }
            oa.desc = cx->d_one;
            oa.ticket = reinterpret_cast<uint32_t*>(cx->d_one + oa.n_tiles);
            oa.total = cx->d_one + oa.n_tiles + 1;
            oa.gsum = cx->d_one + oa.n_tiles + 2 + 8;
            oa.ginc = oa.gsum + (oa.n_tiles / 32 + 1);
            HIP_TRY(hipMemsetAsync(cx->d_one, 0, (size_t)(oa.n_tiles + 2 + 8 + 2 * (oa.n_tiles / 32 + 1)) * 8, stream));
            static const bool one_prof = getenv("TRRE_ONE_PROF") != nullptr;        // (phase clocks of the kernel, printed by finish())
            oa.prof = one_prof ? cx->d_one + oa.n_tiles + 2 : nullptr;
            ScanArgs wa = args;
            wa.spec_look = oa.look;                            // (the forward lanes' look-back; the backward pass above kept kSpecLook)
            if (launch_one(wa, oa, stream, (int)align_up(stt.g16.size() * 4, 16), sym_mode, g16_slow) == 0) {      // (the 16-byte form alone: no pair form in LDS)
                pd.total_at = oa.total;
                pd.one = true;
                HIP_TRY(hipGetLastError());
                pd.launched = true;
                pd.count += 1;
                return TRRE_OK;
            }
        }
        // (Rounds 3 and 4 had two one-walk forms here — record + patch, and mark + splice for small tables; both lost to this pair and were removed
        // in round 5: DESIGN.md §4.5.)
        launch_direct_kernel(1, direct_ent_lds, args, lane_bytes, n_chunks, stream, g16, sym_mode, g16_slow);
        if (use_exact) launch_spec_verify(args, (args.vend + lane_bytes - 1) / lane_bytes, stream);
        launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, n_chunks, stream);
        if (use_exact) args.exact = 2;
        launch_direct_kernel(2, direct_ent_lds, args, lane_bytes, n_chunks, stream, g16, sym_mode, g16_slow);
        pd.total_at = cx->d_chunk_base + n_chunks;
        if (use_exact) {
            pd.exact = true;
            pd.xargs = args; pd.x_lane_bytes = lane_bytes; pd.x_n_chunks = n_chunks; pd.x_g16 = g16; pd.x_sym = sym_mode; pd.x_slow = g16_slow; pd.x_ent_lds = direct_ent_lds;
            pd.x_rev_lane_bytes = is_guided(family) ? rev_lane_bytes : 0;
        }
    } else {
        launch_tile_kernel(1, p->engine, p->mask_bytes, args, n_chunks, stream);
        launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, n_chunks, stream);
        launch_tile_kernel(2, p->engine, p->mask_bytes, args, n_chunks, stream);
        pd.total_at = cx->d_chunk_base + n_chunks;
    }
    HIP_TRY(hipGetLastError());
    // (the status word, the output size and the closing timing event are fetched by finish(): nothing
    // sits between the kernels of back-to-back launches)
    pd.launched = true;
    pd.count += 1;
    return TRRE_OK;
}

// A byte map's launch met NUL bytes: a NUL cuts its line short (Q2) — the bytes before it are mapped, a '\n' stands at its
// position, the rest of the record is gone — so everything behind it moves up.  The kernel listed the NULs; the lines'
// ends are looked up, and the stretches between the cuts are mapped again from the input to where they belong (the
// first one is in place already).  Returns TRRE_OK with *out_len, an error, or -1: not a case for this (more NULs than the
// list holds, more than 256 cut lines, an in-place scan) — the caller runs the buffer through the general family.
int repair_bytemap_nuls(DeviceState* st, ScanCtx* cx, const Pending& was, size_t* out_len) {
    using namespace trre;
    static const bool off = getenv("TRRE_NO_NUL_REPAIR") != nullptr;
    if (off || was.d_in == was.d_out) return -1;
    uint32_t count = 0;
    HIP_TRY(hipMemcpyAsync(&count, cx->d_status + 4, 4, hipMemcpyDeviceToHost, was.stream));
    HIP_TRY(hipStreamSynchronize(was.stream));
    if (count == 0 || count > kNulCap) return -1;
    uint64_t* d_pos = reinterpret_cast<uint64_t*>(cx->d_status + 6);
    uint64_t* d_eol = d_pos + kNulCap;
    launch_nul_eol(was.d_in, (int64_t)was.n, d_pos, d_eol, count, was.stream);
    std::vector<uint64_t> pos(count), eol(count);
    HIP_TRY(hipMemcpyAsync(pos.data(), d_pos, (size_t)count * 8, hipMemcpyDeviceToHost, was.stream));
    HIP_TRY(hipMemcpyAsync(eol.data(), d_eol, (size_t)count * 8, hipMemcpyDeviceToHost, was.stream));
    HIP_TRY(hipStreamSynchronize(was.stream));
    std::vector<std::pair<uint64_t, uint64_t>> cuts;          // (the NUL, the end of its line): the first NUL of every line
    {
        std::vector<std::pair<uint64_t, uint64_t>> all(count);
        for (uint32_t i = 0; i < count; ++i) {
            if (eol[i] == ~0ull) return -1;                   // (a line of many megabytes behind a NUL)
            all[i] = {pos[i], eol[i]};
        }
        std::sort(all.begin(), all.end());
        bool any = false;
        uint64_t done_to = 0;
        for (const auto& c : all) {
            if (any && c.first <= done_to) continue;
            cuts.push_back(c);
            done_to = c.second;
            any = true;
        }
    }
    if (cuts.size() > 256) return -1;
    const uint8_t* blob = st->d_blob;
    // stretch by stretch: input [s, the next NUL) mapped to s - shift, then a '\n'; the last one ends at the last byte of
    // the input, which is a terminator whatever it holds (Q1)
    uint64_t s = 0, shift = 0;
    const uint64_t last_byte = (uint64_t)was.n - 1;
    for (size_t j = 0; j < cuts.size(); ++j) {
        const uint64_t nul = cuts[j].first;
        if (j == 0) launch_bytemap_shift(blob, was.d_in + nul, was.d_out + nul, 0, true, was.stream);       // (in place already: only its '\n')
        else launch_bytemap_shift(blob, was.d_in + s, was.d_out + (s - shift), (int64_t)(nul - s), true, was.stream);
        shift += cuts[j].second - nul;
        s = cuts[j].second + 1;
    }
    if (s <= last_byte) launch_bytemap_shift(blob, was.d_in + s, was.d_out + (s - shift), (int64_t)(last_byte - s), true, was.stream);
    HIP_TRY(hipStreamSynchronize(was.stream));
    cx->relaunches += 1;
    if (out_len) *out_len = (size_t)((uint64_t)was.n - shift);
    return TRRE_OK;
}

int finish_inner(trre_prog* p, DeviceState* st, ScanCtx* cx, size_t* out_len) {
    using namespace trre;
    Pending& pd = cx->pend;
    if (!pd.active) return fail(TRRE_E_ARG, "error: no scan in flight");
    const Pending was = pd;
    pd = Pending();
    if (was.n == 0) { if (out_len) *out_len = 0; return TRRE_OK; }
    if (!was.launched) {                                 // length-preserving family, buffer too small
        if (out_len) *out_len = was.n;
        return fail(TRRE_E_CAPACITY, "error: output buffer too small");
    }
    if (was.timed) HIP_TRY(hipEventRecord(cx->ev1, was.stream));
    HIP_TRY(hipMemcpyAsync(cx->h_status, cx->d_status, 8, hipMemcpyDeviceToHost, was.stream));
    if (was.total_at) HIP_TRY(hipMemcpyAsync(cx->h_status + 2, was.total_at, 8, hipMemcpyDeviceToHost, was.stream));
    HIP_TRY(hipStreamSynchronize(was.stream));
    if (was.timed) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, cx->ev0, cx->ev1));
        p->last_ms = ms / (float)(was.count > 0 ? was.count : 1);   // average per launch of the batch
    }
    uint32_t status = cx->h_status[0];
    if (was.family == TRRE_KERNEL_DFT_LAZY) {
        // Rounds (lazy_block.hpp): the edges the launch listed are built on the host (trre_dft.c:1135-1175: nft_step, truncate_lcp, the
        // lookup, the finality probe), the new rows go up, the lanes that were void walk again.  Every round explores at least one edge
        // of a finite table over a finite input, so this ends — with the answer, or with the memory limit of the tables.
        const int64_t a0 = (int64_t)(reinterpret_cast<uintptr_t>(was.d_in) & 15u);
        ScanArgs args{};
        args.in_v0 = was.d_in - a0;
        args.out_v0 = was.d_out - a0;
        args.out = was.d_out;
        args.vbeg = a0;
        args.vend = a0 + (int64_t)was.n;
        args.status = cx->d_status;
        args.cap = was.cap;
        args.lane_counts = cx->d_lane_counts;
        args.chunk_total = cx->d_chunk_total;
        args.chunk_base = cx->d_chunk_base;
        const int64_t n_chunks = (args.vend + kLazyLaneBytes * 256 - 1) / (kLazyLaneBytes * 256);
        std::vector<uint32_t> miss;
        size_t spec = 1024;
        int empty_rounds = 0;
        static const bool lazy_trace = getenv("TRRE_TRACE") != nullptr;
        for (int round = 1; (status & kStMiss) && !(status & (kStEditOverflow | kStDiverge)); ++round) {
            uint32_t head[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(head, cx->d_miss, 8, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipStreamSynchronize(was.stream));
            const uint32_t n_miss = head[0] < kLazyMissCap ? head[0] : kLazyMissCap;
            // a void lane lists the edge it stopped at unless this very launch has listed it (or the list was full: then it holds other
            // edges), so a round with a miss and an empty list cannot be — if it is, the device's rows go up afresh, and the scan gives up
            // rather than spin
            if (!n_miss) {
                if (++empty_rounds > 3) {
                    if (out_len) *out_len = 0;
                    return fail(TRRE_E_DEVICE, "error: the lazy tables make no progress (a miss with nothing listed, four rounds running)");
                }
                std::lock_guard<std::mutex> lock(p->lazy_mu);
                st->lazy_rows_up = 0;
            } else {
                empty_rounds = 0;
            }
            miss.resize((size_t)trre::kLazyMissWords * n_miss + 2);
            if (n_miss) HIP_TRY(hipMemcpyAsync(miss.data(), cx->d_miss + 2, (size_t)n_miss * trre::kLazyMissWords * 4, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipStreamSynchronize(was.stream));
            try {
                std::lock_guard<std::mutex> lock(p->lazy_mu);
                p->lazy->explore(miss.data(), n_miss, spec);
            } catch (const trre::Error& e) {
                if (out_len) *out_len = 0;
                return fail(e.code, e.what());
            } catch (const std::bad_alloc&) {
                if (out_len) *out_len = 0;
                return fail(TRRE_E_TOO_BIG, "error: out of memory while determinising");
            }
            if (lazy_trace) fprintf(stderr, "trre: lazy round %d: %u misses, %u rows, %u states\n", round, n_miss, p->lazy->n_rows(), p->lazy->n_states());
            // the automatic choice for a pattern that also has its eager tables: a scan that is still growing the lazy ones after kLazyAutoRounds
            // goes to the tile kernels (the next scan finds the tables that much further)
            constexpr int kLazyAutoRounds = 64;
            if (round >= kLazyAutoRounds && !p->forced_family && !p->lazy_only && p->has_engine_tables) {
                cx->relaunches += 1;
                const int rc = enqueue(p, st, cx, TRRE_KERNEL_TILE_GEN, was.d_in, was.n, was.d_out, was.cap, was.stream);
                return rc ? rc : finish_inner(p, st, cx, out_len);
            }
            if (spec < (1u << 16)) spec *= 2;                      // (the deeper the input digs, the further ahead the host looks)
            HIP_TRY(hipMemsetAsync(cx->d_status, 0, 8, was.stream));
            cx->relaunches += 1;
            const int rc = lazy_round(p, st, cx, args, n_chunks, false, was.stream);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(cx->h_status, cx->d_status, 8, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipMemcpyAsync(cx->h_status + 2, was.total_at, 8, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipStreamSynchronize(was.stream));
            status = cx->h_status[0];
        }
        if (status & kStDiverge) {      // an edge whose closure runs round an epsilon cycle: the reference dies there with its output unflushed
            if (out_len) *out_len = 0;
            return fail(TRRE_E_DIVERGES, "error: stack max capacity reached (the reference's search does not terminate on this input)");
        }
        if (status & kStEditOverflow) {
            if (out_len) *out_len = 0;
            return fail(TRRE_E_UNSUPPORTED, "error: a sub-range of 1 KiB takes more than 2^32 table steps (TRRE_LAZY_BUDGET): lines of megabytes with an attempt per byte");
        }
    }
    if (was.exact && !(status & (kStOverflow | kStDiverge))) {
        // exact sub-ranges: lanes whose guessed entry state was not the exit state of the lane before them walk again (and on, while
        // their exit keeps differing from what the next lane assumed), until k_spec_verify finds none; then the sizes are final
        uint32_t both[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(both, cx->d_status + 2, 8, hipMemcpyDeviceToHost, was.stream));
        HIP_TRY(hipStreamSynchronize(was.stream));
        uint32_t misses = both[1];
        static const bool spec_trace0 = getenv("TRRE_TRACE") != nullptr;
        if (both[0] && was.x_rev_lane_bytes) {
            // the backward pass guessed wrong somewhere: its flagged lanes sweep again (and on to the left while what they arrive with is not what
            // was assumed there) until every guess is what the lane to the right found; then the forward passes run — they had left at once
            ScanArgs xa = was.xargs;
            uint32_t rev_misses = both[0];
            for (int64_t round = 0; rev_misses; ++round) {
                if (spec_trace0) fprintf(stderr, "trre: exact sub-ranges, backward pass: round %lld, %u lane(s) to repair\n", (long long)round, rev_misses);
                HIP_TRY(hipMemsetAsync(cx->d_status + 2, 0, 4, was.stream));
                launch_rev_repair(xa, was.x_rev_lane_bytes, was.stream, was.x_sym == 2);
                launch_rev_verify(xa, was.x_rev_lane_bytes, was.stream, was.x_sym == 2);
                HIP_TRY(hipMemcpyAsync(&rev_misses, cx->d_status + 2, 4, hipMemcpyDeviceToHost, was.stream));
                HIP_TRY(hipStreamSynchronize(was.stream));
            }
            HIP_TRY(hipMemsetAsync(cx->d_status, 0, 16, was.stream));
            xa.exact = 1;
            launch_direct_kernel(1, was.x_ent_lds, xa, was.x_lane_bytes, was.x_n_chunks, was.stream, was.x_g16, was.x_sym, was.x_slow);
            launch_spec_verify(xa, (xa.vend + was.x_lane_bytes - 1) / was.x_lane_bytes, was.stream);
            launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, was.x_n_chunks, was.stream);
            xa.exact = 2;
            launch_direct_kernel(2, was.x_ent_lds, xa, was.x_lane_bytes, was.x_n_chunks, was.stream, was.x_g16, was.x_sym, was.x_slow);
            HIP_TRY(hipGetLastError());
            cx->relaunches += 1;
            HIP_TRY(hipMemcpyAsync(cx->h_status, cx->d_status, 8, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipMemcpyAsync(cx->h_status + 2, was.total_at, 8, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipMemcpyAsync(&misses, cx->d_status + 3, 4, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipStreamSynchronize(was.stream));
            status = cx->h_status[0];
        }
        if (misses && !(status & (kStOverflow | kStDiverge))) {
            ScanArgs xa = was.xargs;
            const int64_t n_lanes = (xa.vend + was.x_lane_bytes - 1) / was.x_lane_bytes;
            static const bool spec_trace = getenv("TRRE_TRACE") != nullptr;
            for (int64_t round = 0; misses; ++round) {
                if (spec_trace) fprintf(stderr, "trre: exact sub-ranges: round %lld, %u lane(s) to repair\n", (long long)round, misses);
                if (round > was.x_n_chunks + 64) {           // (every round settles at least the first flagged lane's workgroup: cannot happen)
                    cx->exact_off = true;
                    const int rc = enqueue(p, st, cx, was.family, was.d_in, was.n, was.d_out, was.cap, was.stream);
                    cx->exact_off = false;
                    cx->relaunches += 1;
                    return rc ? rc : finish_inner(p, st, cx, out_len);
                }
                HIP_TRY(hipMemsetAsync(cx->d_status + 3, 0, 4, was.stream));
                xa.exact = 3;
                launch_direct_kernel(1, was.x_ent_lds, xa, was.x_lane_bytes, was.x_n_chunks, was.stream, was.x_g16, was.x_sym, was.x_slow);
                launch_spec_verify(xa, n_lanes, was.stream);
                HIP_TRY(hipMemcpyAsync(&misses, cx->d_status + 3, 4, hipMemcpyDeviceToHost, was.stream));
                HIP_TRY(hipStreamSynchronize(was.stream));
            }
            launch_chunk_scan(cx->d_chunk_total, cx->d_chunk_base, was.x_n_chunks, was.stream);
            xa.exact = 2;
            launch_direct_kernel(2, was.x_ent_lds, xa, was.x_lane_bytes, was.x_n_chunks, was.stream, was.x_g16, was.x_sym, was.x_slow);
            HIP_TRY(hipGetLastError());
            cx->relaunches += 1;
            HIP_TRY(hipMemcpyAsync(cx->h_status, cx->d_status, 8, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipMemcpyAsync(cx->h_status + 2, was.total_at, 8, hipMemcpyDeviceToHost, was.stream));
            HIP_TRY(hipStreamSynchronize(was.stream));
            status = cx->h_status[0];
        }
    }
    if (was.one && getenv("TRRE_ONE_PROF")) {
        uint64_t pr[8] = {};
        HIP_TRY(hipMemcpy(pr, was.total_at + 1, sizeof(pr), hipMemcpyDeviceToHost));
        const double t = pr[7] ? (double)pr[7] : 1.0;
        fprintf(stderr, "trre: one-pass phases, shader clocks per tile (%llu tiles): ticket %.0f  walk %.0f  verify %.0f  sizes %.0f  look-back %.0f  store %.0f\n",
                (unsigned long long)pr[7], pr[0] / t, pr[1] / t, pr[2] / t, pr[3] / t, pr[4] / t, pr[5] / t);
    }
    if (was.mapgen && getenv("TRRE_MAPGEN_PROF")) {
        uint64_t pr[8] = {};
        HIP_TRY(hipMemcpy(pr, was.total_at + 1, sizeof(pr), hipMemcpyDeviceToHost));
        const double t = pr[7] ? (double)pr[7] : 1.0;
        fprintf(stderr, "trre: memoryless kernel, shader clocks per trip (%llu trips): look-back %.0f  expand %.0f  next tile counted %.0f  store %.0f  barrier %.0f\n",
                (unsigned long long)pr[7], pr[1] / t, pr[2] / t, pr[3] / t, pr[4] / t, pr[5] / t);
    }
    if (was.mapgen && getenv("TRRE_MAPGEN_DBG")) {
        const int64_t nt = ((int64_t)was.n + 15 + kMapGenTile) / kMapGenTile + 1;
        std::vector<uint64_t> d(16 * (size_t)cx->mg_tiles);
        HIP_TRY(hipMemcpy(d.data(), was.total_at + 1 + 8, d.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(getenv("TRRE_MAPGEN_DBG"), "wb")) { fwrite(d.data(), 8, d.size(), f); fclose(f); }
        (void)nt;
    }
    if (was.mapgen && (status & (kStNul | kStOneVoid))) {
        // a NUL cuts its record short — the rest of the record is swallowed: state after all — (or a look-back that gave up: a workgroup of
        // the grid was not resident — another process on the device —, tests: TRRE_MAPGEN_OVERSUB): the general small-table family takes the buffer
        static const bool mg_trace = getenv("TRRE_TRACE") != nullptr;
        if (mg_trace) fprintf(stderr, "trre: a one-pass launch of the memoryless kernel was void: status 0x%x\n", status);
        cx->mapgen_off = true;
        cx->mapgen_voids += 1;
        cx->relaunches += 1;
        int rc = enqueue(p, st, cx, was.family, was.d_in, was.n, was.d_out, was.cap, was.stream);
        if (!rc) rc = finish_inner(p, st, cx, out_len);
        cx->mapgen_off = false;
        return rc;
    }
    if (was.mapgen) {
        // (the longer texts are frequent in what this context scans: see enqueue())
        uint64_t m = 0;
        std::memcpy(&m, cx->h_status + 2, 8);
        if (m > (uint64_t)was.n + (uint64_t)was.n / 25) cx->mapgen_dense = true;
    }
    if (was.one && (status & kStOneVoid)) {
        // the one-pass kernel could not answer (one_block.hpp: what voids it): the count / emit pair takes the buffer — and, the corpus being
        // what it is, after two such launches the program's later scans too
        static const bool one_trace = getenv("TRRE_TRACE") != nullptr;
        if (one_trace) fprintf(stderr, "trre: a one-pass launch of family %d was void: status 0x%x\n", was.family, status);
        p->one_fails.fetch_add(1);
        cx->one_off = true;
        cx->relaunches += 1;
        int rc = enqueue(p, st, cx, was.family, was.d_in, was.n, was.d_out, was.cap, was.stream);
        if (!rc) rc = finish_inner(p, st, cx, out_len);
        cx->one_off = false;
        return rc;
    }
    if ((was.exact || was.one) && (status & kStDiverge)) {
        // an attempt that does not return: the lane bookkeeping of the error path (the first such lane in stream order, the sizes of the lanes
        // before it) is that of lanes that own lines — the same scan again, the old way
        cx->exact_off = true;
        const int rc = enqueue(p, st, cx, was.family, was.d_in, was.n, was.d_out, was.cap, was.stream);
        cx->exact_off = false;
        cx->relaunches += 1;
        return rc ? rc : finish_inner(p, st, cx, out_len);
    }
    auto again = [&](int family) -> int {
        cx->relaunches += 1;
        int rc = enqueue(p, st, cx, family, was.d_in, was.n, was.d_out, was.cap, was.stream);
        if (rc) return rc;
        return finish_inner(p, st, cx, out_len);
    };
    // (a length-preserving launch that met a NUL is void — it went on walking behind the NUL, where the reference never
    // looks — and so is whatever else it reports: the general family decides, below)
    const bool void_by_nul = !is_gen(was.family) && (status & kStNul);
    if ((status & kStDiverge) && !void_by_nul) {
        const char* msg = "error: stack max capacity reached (the reference's search does not terminate on this input)";
        if (out_len) *out_len = 0;
        // The NFT binary exits with everything it had printed so far (exit() flushes stdout): the lines before the bad one
        // and the bad line's output up to the attempt that does not return.  The guided tables stop a lane exactly there
        // (guided_build.cpp), the count pass names the first such lane in stream order and knows every lane's size: the
        // output up to that point is in the buffer and its length is reported with the error.  (The DFT binary dies of
        // unbounded recursion, its buffered output is lost: nothing to reproduce, *out_len = 0.)
        const bool knows_lane = (was.family == TRRE_KERNEL_GUIDED_GEN && !was.patched) || was.family == TRRE_KERNEL_BACKTRACK;
        if (p->engine != TRRE_ENGINE_NFT || (!p->gt.ok && !p->bt_ok)) return fail(TRRE_E_DIVERGES, msg);
        if (!knows_lane) {
            cx->patch_off = true;                              // (the count / emit pair knows every lane's size)
            const int rc = again(p->gt.ok ? TRRE_KERNEL_GUIDED_GEN : TRRE_KERNEL_BACKTRACK);      // (comes back here through the branch below)
            cx->patch_off = false;
            return rc == TRRE_OK ? fail(TRRE_E_DEVICE, "error: a diverging scan did not diverge when it was run again") : rc;
        }
        const uint32_t lane = 0xffffffffu - cx->h_status[1];
        const int64_t chunk = (int64_t)lane / 256;             // (workspace chunks are 256 lanes: ensure_workspace)
        std::vector<uint32_t> counts(256);
        uint64_t base = 0, chunk_total = 0;
        HIP_TRY(hipMemcpyAsync(counts.data(), cx->d_lane_counts + chunk * 256, 256 * 4, hipMemcpyDeviceToHost, was.stream));
        HIP_TRY(hipMemcpyAsync(&base, cx->d_chunk_base + chunk, 8, hipMemcpyDeviceToHost, was.stream));
        HIP_TRY(hipMemcpyAsync(&chunk_total, cx->d_chunk_total + chunk, 8, hipMemcpyDeviceToHost, was.stream));
        HIP_TRY(hipStreamSynchronize(was.stream));
        uint64_t upto = base;
        for (uint32_t k = 0; k <= lane % 256; ++k) upto += counts[k];
        if (base + chunk_total > was.cap) {                    // the emit pass skipped this chunk: ask for room, the retry reports
            if (out_len) *out_len = (size_t)(base + chunk_total);
            return fail(TRRE_E_CAPACITY, "error: output buffer too small");
        }
        if (out_len) *out_len = (size_t)upto;
        return fail(TRRE_E_DIVERGES, msg);
    }
    if (was.family == TRRE_KERNEL_BACKTRACK && (status & kStEditOverflow)) {
        // which limit: an attempt deeper than a lane's stack or with more output than its path buffer runs again with fewer, larger ones;
        // the step budget is final (the reference's search is exponential here too)
        uint32_t why = 0;
        HIP_TRY(hipMemcpyAsync(&why, cx->d_status + 2, 4, hipMemcpyDeviceToHost, was.stream));
        HIP_TRY(hipStreamSynchronize(was.stream));
        if (!(why & kBtWhyBudget) && cx->bt_tier < kBtTierCount - 1) {
            const int tier = cx->bt_tier;
            cx->bt_tier = tier + 1;
            const int rc = again(was.family);
            cx->bt_tier = tier;
            return rc;
        }
        if (out_len) *out_len = 0;
        return fail(TRRE_E_UNSUPPORTED, (why & kBtWhyBudget) ? "error: the search takes more than 16 M steps in 1 KiB of this input (TRRE_BT_BUDGET; the backtracking fallback)"
                                                              : "error: an attempt of the search consumes more than 1 M bytes or prints more than 1 MiB (the backtracking fallback)");
    }
    if (status & kStNeedScratch) {
        // a line longer than the LDS tile met the non-deterministic engine: give it
        // a mask scratch (one mask per input byte) and run again
        const size_t need = (was.n + 32) * (size_t)p->mask_bytes;
        if (cx->scratch_bytes < need) {
            if (cx->d_scratch) (void)hipFree(cx->d_scratch);
            cx->d_scratch = nullptr; cx->scratch_bytes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&cx->d_scratch), need));
            cx->scratch_bytes = need;
        }
        return again(was.family);
    }
    static const bool trace_void = getenv("TRRE_TRACE") != nullptr;      // (why a launch was void, on stderr)
    if (trace_void && was.patched && (status & (kStEditOverflow | kStNul | kStOverflow | kStDiverge)))
        fprintf(stderr, "trre: a mark + splice launch of family %d was void: status 0x%x\n", was.family, status);
    if (was.patched && (status & (kStEditOverflow | kStNul))) {
        // copy
        // form of a large table: more texts in a sub-range (or in 64 bytes of it) than its event list holds, or a NUL byte —
        // the count / emit pair
        if (status & kStEditOverflow) p->copy_form_off.store(true);      // (a property of the dictionary and its corpus: do not try again)
        cx->patch_off = true;
        const int rc = again(was.family);
        cx->patch_off = false;
        return rc;
    }
    // an undecided attempt outgrew the stream table (bounded fold): the guided (or the tile) kernels take the buffer — this one
    // and, the corpus being what it is, the ones after it (the count pass found out; the emit pass saw the mark and left at once)
    if (is_stream(was.family) && (status & kStOverflow)) {
        p->bounded_off.store(true);
        return again(general_family(*p, false));
    }
    if (!is_gen(was.family)) {
        // a NUL cuts its line short, so output positions no longer equal input positions: redo with a general family
        if ((status & kStNul) && was.family == TRRE_KERNEL_BYTEMAP) {
            const int rc = repair_bytemap_nuls(st, cx, was, out_len);
            if (rc >= 0) return rc;
        }
        // (the same repair for the other length-preserving families — the NULs looked up by a pass over the input, the stretches
        // scanned again by the window kernel with unaligned stores — was built and measured in round 4: 13.5 ms for 8 GiB with a
        // NUL per GiB against 12.9 ms for the general family over the whole buffer; not kept.  DESIGN.md §5)
        if (status & kStNul) return again(general_family(*p, !is_guided(was.family)));
        if (out_len) *out_len = was.n;
        return TRRE_OK;
    }
    uint64_t total;
    std::memcpy(&total, cx->h_status + 2, 8);
    if (out_len) *out_len = (size_t)total;
    if ((status & kStCapacity) || total > was.cap) return fail(TRRE_E_CAPACITY, "error: output buffer too small");
    return TRRE_OK;
}

// finish_inner, then the stack guard: a scan that went through may have met a line on which the reference's search runs out of
// stack (guard_block.hpp) — then the answer is what the reference gives: an error, and the output up to the attempt that
// overflowed (the lines before that line from a scan of exactly those lines, the line's own part from the guard's search).
int finish(trre_prog* p, DeviceState* st, ScanCtx* cx, size_t* out_len) {
    using namespace trre;
    const Pending was = cx->pend;
    // the long-line probe's verdict holds for the batch it was asked in (identical launches); the caller may put other bytes into the same
    // buffer afterwards — the host path's slots do for every chunk — so the next batch asks again
    cx->probe_in = nullptr; cx->probe_n = 0;
    GuardHit hit;
    if (was.active && was.guard_hit) {                     // found before an in-place launch: nothing has run yet
        cx->pend = Pending();
        hit.hit = true; hit.line_start = was.guard_line; hit.part = was.guard_part;
    } else {
        const int rc = finish_inner(p, st, cx, out_len);
        if (rc != TRRE_OK || !was.active || was.n == 0 || !guard_applies(*p, *cx, was.n) || was.d_in == was.d_out) return rc;
        const int grc = guard_check(p, st, cx, was.d_in, was.n, was.stream, &hit);
        if (grc) return grc;
        if (!hit.hit) return rc;
    }
    size_t pre = 0;
    if (hit.line_start > 0) {
        cx->guard_off = true;
        int rc = enqueue(p, st, cx, was.family, was.d_in, (size_t)hit.line_start, was.d_out, was.cap, was.stream);
        if (!rc) rc = finish_inner(p, st, cx, &pre);
        cx->guard_off = false;
        cx->relaunches += 1;
        if (rc) { if (out_len) *out_len = pre; return rc; }
    }
    const size_t part = hit.part == 0xffffffffu ? 0 : hit.part;
    if (out_len) *out_len = pre + part;
    if (pre + part > was.cap) return fail(TRRE_E_CAPACITY, "error: output buffer too small");
    if (part) {
        HIP_TRY(hipMemcpyAsync(was.d_out + pre, cx->d_gout, part, hipMemcpyDeviceToDevice, was.stream));
        HIP_TRY(hipStreamSynchronize(was.stream));
    }
    cx->relaunches += 1;
    return fail(TRRE_E_DIVERGES, kStackMsg);
}


int compile_impl(const std::string& pattern, int engine, trre_prog** out, int mode = TRRE_MODE_SCAN) {
    using namespace trre;
    if (!out) return fail(TRRE_E_ARG, "error: null output handle");
    *out = nullptr;
    if (engine != TRRE_ENGINE_NFT && engine != TRRE_ENGINE_DFT) return fail(TRRE_E_ARG, "error: unknown engine");
    if (mode != TRRE_MODE_SCAN && mode != TRRE_MODE_MATCH && !is_generate(mode)) return fail(TRRE_E_ARG, "error: unknown mode");
    if (mode == TRRE_MODE_MATCH && engine != TRRE_ENGINE_NFT)
        return fail(TRRE_E_UNSUPPORTED, "error: match mode is offered for the non-deterministic engine only (trre_dft -m prints empty lines)");
    if (is_generate(mode) && engine != TRRE_ENGINE_NFT) return fail(TRRE_E_UNSUPPORTED, "Not supported yet");   // trre_dft.c:1227-1229
    try {
        std::unique_ptr<trre_prog> p(new trre_prog);
        p->engine = engine;
        p->mode = mode;
        Ast ast = parse_pattern(pattern);
        Nft nft = build_nft(ast, engine == TRRE_ENGINE_DFT);
        p->nft_states = (uint32_t)nft.st.size();
        p->nft_cons = (uint32_t)nft.n_cons;
        // the stack guard (guard_block.hpp): the NFT itself, for the lines long enough to exhaust the reference's stack
        auto make_guard = [&](bool match) {
            p->guard = build_guard(nft);
            if (!p->guard.on) return;
            GuardBlobHeader gh{};
            gh.magic = kMagicGuard;
            gh.n_states = (uint32_t)nft.st.size();
            gh.start = p->guard.start;
            gh.d = p->guard.d;
            gh.l_min = p->guard.l_min;
            gh.window = p->guard.window;
            gh.off_states = (uint32_t)sizeof gh;
            gh.total_bytes = (uint32_t)(sizeof gh + p->guard.states.size() * 4);
            gh.match = match ? 1u : 0u;
            gh.n_once = p->guard.n_once;
            for (int k = 0; k < 8; ++k) gh.bset[k] = p->guard.bset[k];
            put(p->kblob, 0, &gh, 1);
            put(p->kblob, gh.off_states, p->guard.states.data(), p->guard.states.size());
        };
        if (engine == TRRE_ENGINE_DFT) {
            p->dft_nft = nft;
            std::unique_ptr<Dft> eager;
            try {
                eager.reset(new Dft(determinize(nft)));
            } catch (const Error& e) {
                if (e.code != kErrTooBig) throw;
                // The eager construction does not end within its caps — a state per run length ('((a:x)*b)|((a:y)*c)'), 2^19 states
                // ('(a|b)*a(a|b){18}:x').  The reference builds only the states its input visits (trre_dft.c:1135-1175) and so does
                // the lazy family (lazy_block.hpp; rounds 1-4: TRRE_E_TOO_BIG).
            }
            if (!eager) {
                p->lazy_only = true;
                const int lrc = lazy_ensure(p.get());
                if (lrc) return lrc;
                *out = p.release();
                return TRRE_OK;
            }
            Dft& dft = *eager;
            p->dt = flatten_dft(dft);
            serialize_dft(*p);
            p->has_engine_tables = true;
            p->stt = build_stream_dft(dft);
            // the guided route (guided_build.cpp: GuidedDftBuilder): what a pattern whose scan loop does not fold runs on ('[a-z]+ing:X':
            // a loop before the decision), and what takes over when a bounded fold overflows ('a*b:x' behind a run of 65 a's) — the
            // tile kernels remain for tables beyond its limits.  (A byte map needs neither.)
            if (!(p->dt.flags & kFlagMemoryless)) p->gt = build_guided_dft(dft);
        } else if (is_generate(mode)) {
            // trre -a / -ma: every accepting path prints (generate.cpp): a viability DFA for the device, the lists for the host
            p->gen = build_gen_tables(nft, mode == TRRE_MODE_MATCH_ALL);      // (always ok since round 5: beyond 256 viability states the filter lets everything through)
            p->nft_nodes = (uint32_t)p->gen.nodes.node.size();
            serialize_rev_table(p->gen.n_rev, p->gen.n_cls, 8, p->gen.cls, p->gen.rev, p->rblob);
            serialize_gen(p->gen, p->nblob);
        } else if (mode == TRRE_MODE_MATCH) {
            // trre -m: one attempt per line, accepted at its end only — the guided tables in match form
            const NftNodes nodes = build_nft_nodes(nft, true);
            p->nft_nodes = (uint32_t)nodes.node.size();
            p->gt = build_guided_nft(nodes);
            // beyond the guided tables' limits (a backward automaton of more than 16 384 states): the search itself in match form
            // (gen_block.hpp: bt_lane; round 4: TRRE_E_UNSUPPORTED) — and on request for any pattern (the parity tests)
            {
                GenTables lists;
                lists.nodes = nodes;
                lists.match_mode = true;
                lists.ok = true;
                serialize_gen(lists, p->nblob);
                p->bt_ok = true;
            }
            make_guard(true);
        } else {
            // TRRE_TRACE=1: the stages of the NFT compile on stderr as they start (to find the one a pattern is slow in)
            static const bool trace_on = getenv("TRRE_TRACE") != nullptr;
            auto trace = [&](const char* what) { if (trace_on) fprintf(stderr, "compile: %s\n", what); };
            trace("nodes");
            const NftNodes nodes = build_nft_nodes(nft);
            p->nft_nodes = (uint32_t)nodes.node.size();
            std::unique_ptr<Error> deferred;
            trace("bitmask tables");
            try {
                p->nt = build_nft_tables(nodes);
                p->mask_bytes = p->nt.n_cons <= 8 ? 1 : p->nt.n_cons <= 16 ? 2 : p->nt.n_cons <= 32 ? 4 : 8;
                serialize_nft(*p);
                p->has_engine_tables = true;
            } catch (const Error& e) {
                if (e.code != kErrUnsupported) throw;
                deferred.reset(new Error(e));            // too many nodes (or an epsilon cycle) for the bitmask kernels
            }
            // the fold walks the follow lists (TRRE_NFT_FOLD=states: the NFT's states as the reference does, =both: both, compared)
            const char* fold = getenv("TRRE_NFT_FOLD");
            trace("fold");
            if (fold && !strcmp(fold, "states")) {
                p->stt = build_stream_nft(nft);
            } else {
                p->stt = build_stream_nodes(nodes);
                if (fold && !strcmp(fold, "both")) {
                    const StreamTables ref = build_stream_nft(nft);
                    // (the walk over the states may run out of budget where this one does not: no table to compare then)
                    if (ref.ok && !p->stt.ok) throw Error(kErrArg, "error: the fold over the follow lists gave up where the one over the states did not (TRRE_NFT_FOLD=both)");
                    if (ref.ok)
                    if (ref.ent != p->stt.ent || ref.pool != p->stt.pool || ref.cls != p->stt.cls || ref.flags != p->stt.flags ||
                        ref.pending_len != p->stt.pending_len || ref.lpw != p->stt.lpw || ref.g16 != p->stt.g16 || ref.fb_comb != p->stt.fb_comb)
                        throw Error(kErrArg, "error: the two folds of the NFT scan loop disagree (TRRE_NFT_FOLD=both)");
                }
            }
            trace("guided tables");
            p->gt = build_guided_nft(nodes);
            trace("done");
            // The backtracking fallback (gen_block.hpp: bt_lane) runs what no table form holds — more than 64 nodes with a backward
            // automaton beyond the guided limits and no fold (round 3: TRRE_E_UNSUPPORTED) — and can be asked for on any pattern
            // (trre_set_kernel: the differential tests do).  Its tables are the follow lists themselves.
            {
                GenTables lists;
                lists.nodes = nodes;
                lists.ok = true;
                serialize_gen(lists, p->nblob);
                p->bt_ok = true;
            }
            make_guard(false);
        }
        if (p->stt.ok) serialize_stream(p->stt, p->sblob);
        if (p->gt.ok) { serialize_stream(p->gt.fwd, p->gblob); serialize_rev(p->gt, p->rblob); }
        *out = p.release();
        return TRRE_OK;
    } catch (const Error& e) {
        return fail(e.code, e.what());
    } catch (const std::bad_alloc&) {
        return fail(TRRE_E_TOO_BIG, "error: out of memory while compiling the pattern");
    }
}

}  // namespace

extern "C" {

int trre_compile(const char* pattern, int engine, trre_prog** out) {
    if (!pattern) return fail(TRRE_E_ARG, "error: missing trre expression");
    return compile_impl(std::string(pattern), engine, out);
}

int trre_compile_bytes(const uint8_t* pattern, size_t len, int engine, trre_prog** out) {
    if (!pattern) return fail(TRRE_E_ARG, "error: missing trre expression");
    return compile_impl(std::string(reinterpret_cast<const char*>(pattern), len), engine, out);
}

int trre_compile_mode(const uint8_t* pattern, size_t len, int engine, int mode, trre_prog** out) {
    if (!pattern) return fail(TRRE_E_ARG, "error: missing trre expression");
    return compile_impl(std::string(reinterpret_cast<const char*>(pattern), len), engine, out, mode);
}

void trre_free(trre_prog* p) {
    if (!p) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto& kv : p->dev) {
        DeviceState& st = *kv.second;
        (void)hipSetDevice(st.device);
        (void)hipFree(st.d_blob);
        (void)hipFree(st.d_sblob);
        (void)hipFree(st.d_gblob);
        (void)hipFree(st.d_rblob);
        (void)hipFree(st.d_nblob);
        (void)hipFree(st.d_kblob);
        (void)hipFree(st.d_lent); (void)hipFree(st.d_lpool); (void)hipFree(st.d_lcls);
        for (void* r : st.retired) (void)hipFree(r);
        ctx_free(st.ctx);
        for (auto& hs : st.slot) {
            pinned_put(hs.pin_in);
            pinned_put(hs.pin_out);
            (void)hipFree(hs.d_in);
            (void)hipFree(hs.d_out);
            if (hs.stream) (void)hipStreamDestroy(hs.stream);
            ctx_free(hs.ctx);
        }
    }
    if (have_cur) (void)hipSetDevice(cur);
    delete p;
}

const char* trre_last_error(void) { return g_error.c_str(); }

int trre_get_info(const trre_prog* p, trre_info* info) {
    if (!p || !info) return fail(TRRE_E_ARG, "error: null argument");
    std::memset(info, 0, sizeof *info);
    info->engine = p->engine;
    info->kernel = is_generate(p->mode) ? TRRE_KERNEL_GENERATE : scan_family(*p);
    info->nft_states = p->nft_states;
    info->nft_cons_states = p->nft_cons;
    if (p->engine == TRRE_ENGINE_DFT) {
        info->dft_states = p->dt.n_states;
        info->table_rows = p->dt.n_rows;
        info->table_classes = p->dt.n_cls;
        info->flags = p->dt.flags;
        if (p->lazy_only) {       // (the tables grow with the inputs: what exists now)
            info->dft_states = p->lazy->n_states();
            info->table_rows = p->lazy->n_rows();
            info->table_classes = p->lazy->n_cls();
        }
    } else {
        info->table_rows = p->nt.n_cons;
        info->flags = p->nt.flags;
    }
    info->table_bytes = (uint32_t)(p->blob.size() + p->sblob.size() + p->gblob.size() + p->rblob.size());
    info->chunk_bytes = (uint32_t)trre::chunk_bytes(p->engine, p->mask_bytes);
    if (p->stt.ok) { info->stream_states = p->stt.n_states; info->stream_classes = p->stt.n_cls; }
    info->nft_nodes = p->nft_nodes;
    if (p->gt.ok) { info->guided_rev_states = p->gt.n_rev; info->guided_fwd_states = p->gt.fwd.n_states; }
    if (p->gen.ok) info->guided_rev_states = p->gen.n_rev;
    return TRRE_OK;
}

int trre_set_kernel(trre_prog* p, int family) {
    if (!p) return fail(TRRE_E_ARG, "error: null argument");
    if (is_generate(p->mode)) return family == TRRE_KERNEL_AUTO ? TRRE_OK : fail(TRRE_E_UNSUPPORTED, "error: generator mode has one implementation");
    if (family != TRRE_KERNEL_AUTO && !family_allowed(*p, family))
        return fail(TRRE_E_UNSUPPORTED, "error: this kernel family cannot run these tables");
    p->forced_family = family;
    return TRRE_OK;
}

size_t trre_export_tables(const trre_prog* p, void* buf, size_t cap) {
    if (!p) return 0;
    if (buf && cap) std::memcpy(buf, p->blob.data(), cap < p->blob.size() ? cap : p->blob.size());
    return p->blob.size();
}

size_t trre_export_stream_tables(const trre_prog* p, void* buf, size_t cap) {
    if (!p) return 0;
    if (buf && cap) std::memcpy(buf, p->sblob.data(), cap < p->sblob.size() ? cap : p->sblob.size());
    return p->sblob.size();
}

size_t trre_export_guided_tables(const trre_prog* p, int which, void* buf, size_t cap) {
    if (!p) return 0;
    const std::vector<uint8_t>& b = which == 0 ? p->rblob : (which == 2 ? p->nblob : (which == 3 ? p->kblob : p->gblob));   // (2: the enumeration's / the backtracking fallback's tables, 3: the stack guard's)
    if (buf && cap) std::memcpy(buf, b.data(), cap < b.size() ? cap : b.size());
    return b.size();
}

// CPU test tier (tests/cpu_shim.cpp runs lazy_block.hpp's lane body on the host): the lazy tables as they stand — which 0: u32 n_cls, u32 rows,
// then cls[256]; 1: the entries; 2: the pool — and the exploration of a list of misses.  Host only.
size_t trre_debug_lazy_tables(trre_prog* p, int which, void* buf, size_t cap) {
    if (!p || p->engine != TRRE_ENGINE_DFT || p->mode != TRRE_MODE_SCAN || lazy_ensure(p)) return 0;
    std::lock_guard<std::mutex> lock(p->lazy_mu);
    const trre::LazyDft& z = *p->lazy;
    std::vector<uint8_t> head;
    const void* src;
    size_t n;
    if (which == 0) {
        const uint32_t w[2] = {z.n_cls(), z.n_rows()};
        put(head, 0, w, 2);
        put(head, 8, z.cls(), 256);
        src = head.data(); n = head.size();
    } else if (which == 1) {
        src = z.ent(); n = (size_t)z.n_rows() * z.n_cls() * 8;
    } else {
        src = z.pool(); n = z.pool_bytes();
    }
    if (buf && cap) std::memcpy(buf, src, cap < n ? cap : n);
    return n;
}
int trre_debug_lazy_explore(trre_prog* p, const uint32_t* misses, size_t n, size_t spec_states) {
    if (!p || p->engine != TRRE_ENGINE_DFT || p->mode != TRRE_MODE_SCAN) return fail(TRRE_E_ARG, "error: no lazy tables");
    const int rc = lazy_ensure(p);
    if (rc) return rc;
    try {
        std::lock_guard<std::mutex> lock(p->lazy_mu);
        p->lazy->explore(misses, n, spec_states);
    } catch (const trre::Error& e) {
        return fail(e.code, e.what());
    } catch (const std::bad_alloc&) {
        return fail(TRRE_E_TOO_BIG, "error: out of memory while determinising");
    }
    return TRRE_OK;
}

// the prog's state on the calling thread's current device
static int current_state(trre_prog* p, DeviceState** st) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    return device_state(p, dev, st);
}

int trre_scan_enqueue(trre_prog* p, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, void* stream) {
    g_scan_flags = 0;
    if (!p || (n && (!d_in || !d_out))) return fail(TRRE_E_ARG, "error: null argument");
    DeviceState* st;
    int rc = current_state(p, &st);
    if (rc) return rc;
    if (is_generate(p->mode)) return fail(TRRE_E_ARG, "error: generator mode has no split form: use trre_scan_device / trre_scan_host");
    const int fam = scan_family(*p);
    return enqueue(p, st, &st->ctx, fam, d_in, n, d_out, cap, static_cast<hipStream_t>(stream));
}

int trre_scan_finish(trre_prog* p, size_t* out_len) {
    if (!p) return fail(TRRE_E_ARG, "error: null argument");
    DeviceState* st;
    int rc = current_state(p, &st);
    if (rc) return rc;
    return finish(p, st, &st->ctx, out_len);
}

namespace {
int slot_reserve(DeviceState::HostSlot& hs, bool input, size_t bytes, bool need_pin = true);
constexpr size_t kGenChunk = (size_t)16 << 20;
constexpr int64_t kGenLaneBytes = 512;          // generator modes on the device: a lane per 512 bytes of input (the records that start there),
constexpr uint32_t kGenFrames = 512, kGenPathCap = 2048;   // a stack of 512 frames and 2 KiB of path output each (deeper / longer: the host enumeration)

// Generator modes on one device, host buffers: chunks cut at record ends go up, the backward kernel leaves one viability
// symbol per byte (k_rev_sweep, the guided families' backward pass with the tables of generate.cpp), the symbols come
// down and the accepting paths are enumerated on a few host threads.  `result` takes what the reference prints; false
// through `diverged`: a path ran into an epsilon cycle (what was printed before it stays).
int generate_on(trre_prog* p, DeviceState* st, const uint8_t* in, size_t n, std::vector<uint8_t>& result, bool& diverged) {
    using namespace trre;
    diverged = false;
    DeviceState::HostSlot& hs = st->slot[0];
    if (!hs.stream) HIP_TRY(hipStreamCreateWithFlags(&hs.stream, hipStreamNonBlocking));
    int rc = ctx_init(hs.ctx);
    if (rc) return rc;
    static const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 2));
    size_t off = 0;
    while (off < n && !diverged) {
        size_t len = n - off;
        if (len > kGenChunk) {
            const void* nl = std::memchr(in + off + kGenChunk - 1, '\n', n - off - (kGenChunk - 1));
            len = nl ? (size_t)(static_cast<const uint8_t*>(nl) - (in + off)) + 1 : n - off;
        }
        rc = slot_reserve(hs, true, len);
        if (!rc) rc = slot_reserve(hs, false, len + 512);                  // (pin_out / d_out of the slot carry the symbols)
        if (rc) return rc;
        std::memcpy(hs.pin_in, in + off, len);
        HIP_TRY(hipMemcpyAsync(hs.d_in, hs.pin_in, len, hipMemcpyHostToDevice, hs.stream));
        ScanArgs args{};
        args.in_v0 = hs.d_in;                                              // (hipMalloc'ed: 16-byte aligned, v == g)
        args.vbeg = 0;
        args.vend = (int64_t)len;
        args.rblob = st->d_rblob;
        args.sym_v0 = hs.d_out;
        args.status = hs.ctx.d_status;
        launch_rev_sweep(args, (int)p->gen.n_rev * 256, 2048, hs.stream, false);
        HIP_TRY(hipGetLastError());
        // The enumeration on the device (gen_block.hpp; TRRE_GEN_HOST=1: on host threads, as in round 3): count, exclusive
        // sum, emit.  A chunk on which a path never returns, a search goes deeper than a lane's stack or the sizes leave 32 bits
        // goes to the host enumeration below, which also knows what the reference had printed when it gave up.
        static const bool gen_host = getenv("TRRE_GEN_HOST") != nullptr;
        bool done = false;
        if (!gen_host) {
            const int64_t n_lanes_raw = ((int64_t)len + kGenLaneBytes - 1) / kGenLaneBytes;
            const int64_t n_chunks = (n_lanes_raw + 255) / 256, n_lanes = n_chunks * 256;
            rc = ensure_workspace(&hs.ctx, n_chunks, 256);
            const size_t need = (size_t)n_lanes * (kGenFrames * 16 + kGenPathCap);
            if (!rc && hs.ctx.scratch_bytes < need) {
                if (hs.ctx.d_scratch) (void)hipFree(hs.ctx.d_scratch);
                hs.ctx.d_scratch = nullptr; hs.ctx.scratch_bytes = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&hs.ctx.d_scratch), need));
                hs.ctx.scratch_bytes = need;
            }
            if (rc) return rc;
            GenArgs ga{};
            ga.stack = reinterpret_cast<uint32_t*>(hs.ctx.d_scratch);
            ga.path = hs.ctx.d_scratch + (size_t)n_lanes * kGenFrames * 16;
            ga.frames = kGenFrames;
            ga.path_cap = kGenPathCap;
            args.blob = st->d_nblob;
            args.lane_counts = hs.ctx.d_lane_counts;
            args.chunk_total = hs.ctx.d_chunk_total;
            args.chunk_base = hs.ctx.d_chunk_base;
            HIP_TRY(hipMemsetAsync(hs.ctx.d_status, 0, 24, hs.stream));
            launch_gen(1, args, ga, kGenLaneBytes, n_chunks, hs.stream);
            launch_chunk_scan(hs.ctx.d_chunk_total, hs.ctx.d_chunk_base, n_chunks, hs.stream);
            uint64_t total = 0;
            HIP_TRY(hipMemcpyAsync(hs.ctx.h_status, hs.ctx.d_status, 8, hipMemcpyDeviceToHost, hs.stream));
            HIP_TRY(hipMemcpyAsync(&total, hs.ctx.d_chunk_base + n_chunks, 8, hipMemcpyDeviceToHost, hs.stream));
            HIP_TRY(hipStreamSynchronize(hs.stream));
            if (!(hs.ctx.h_status[0] & (kStDiverge | kStEditOverflow | kStCapacity))) {
                // (the symbols sit in the slot's output buffer: the output gets a buffer of its own, kept by the context)
                if (hs.ctx.gen_out_cap < total + 64) {
                    if (hs.ctx.d_gen_out) (void)hipFree(hs.ctx.d_gen_out);
                    hs.ctx.d_gen_out = nullptr; hs.ctx.gen_out_cap = 0;
                    const size_t want = (size_t)total + (size_t)total / 4 + 4096;
                    if (hipMalloc(reinterpret_cast<void**>(&hs.ctx.d_gen_out), want) != hipSuccess)
                        return fail(TRRE_E_TOO_BIG, "error: out of device memory for the outputs of generator mode");
                    hs.ctx.gen_out_cap = want;
                }
                args.out = hs.ctx.d_gen_out;
                args.cap = hs.ctx.gen_out_cap;
                launch_gen(2, args, ga, kGenLaneBytes, n_chunks, hs.stream);
                HIP_TRY(hipGetLastError());
                try {
                    const size_t at = result.size();
                    result.resize(at + (size_t)total);
                    if (total) HIP_TRY(hipMemcpyAsync(result.data() + at, hs.ctx.d_gen_out, (size_t)total, hipMemcpyDeviceToHost, hs.stream));
                } catch (const std::bad_alloc&) {
                    return fail(TRRE_E_TOO_BIG, "error: out of host memory for the outputs of generator mode");
                }
                HIP_TRY(hipStreamSynchronize(hs.stream));
                done = true;
            }
        }
        if (!done) {
            HIP_TRY(hipMemcpyAsync(hs.pin_out, hs.d_out, len, hipMemcpyDeviceToHost, hs.stream));
            HIP_TRY(hipStreamSynchronize(hs.stream));
            try {
                if (!generate_buffer(p->gen, in + off, len, hs.pin_out, result, threads)) diverged = true;
            } catch (const std::bad_alloc&) {
                return fail(TRRE_E_TOO_BIG, "error: out of host memory for the outputs of generator mode");
            }
        }
        off += len;
    }
    return TRRE_OK;
}
constexpr const char* kDivergeMsg = "error: stack max capacity reached (the reference's search does not terminate on this input)";
}  // namespace

// host-only entry for the CPU test tier: the enumeration of generator mode with symbols computed elsewhere (tests/cpu_shim.cpp
// runs the backward kernel's per-thread body on the host).  Not part of the drop-in boundary.
int trre_debug_generate(trre_prog* p, const uint8_t* in, size_t n, const uint8_t* sym, uint8_t* out, size_t cap, size_t* out_len) {
    if (!p || !is_generate(p->mode) || (n && (!in || !sym))) return fail(TRRE_E_ARG, "error: bad argument");
    std::vector<uint8_t> result;
    bool ok;
    try {
        ok = trre::generate_buffer(p->gen, in, n, sym, result, 2);
    } catch (const std::bad_alloc&) {
        return fail(TRRE_E_TOO_BIG, "error: out of host memory for the outputs of generator mode");
    }
    if (out_len) *out_len = result.size();
    if (result.size() > cap) return fail(TRRE_E_CAPACITY, "error: output buffer too small");
    if (!result.empty()) std::memcpy(out, result.data(), result.size());
    return ok ? TRRE_OK : fail(TRRE_E_DIVERGES, kDivergeMsg);
}

uint32_t trre_last_scan_flags(void) { return g_scan_flags; }

int trre_scan_device(trre_prog* p, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, size_t* out_len,
                     void* stream) {
    g_scan_flags = 0;
    if (!p || (n && (!d_in || !d_out))) return fail(TRRE_E_ARG, "error: null argument");
    DeviceState* st;
    int rc = current_state(p, &st);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(st->mu);        // one scan call at a time per (prog, device); other devices run in parallel
    if (is_generate(p->mode)) {
        // the enumeration runs on the host (generate.cpp): input down, output up; the device computes the viability symbols
        if (out_len) *out_len = 0;
        if (n == 0) return TRRE_OK;
        std::vector<uint8_t> host, result;
        try {
            host.resize(n);
        } catch (const std::bad_alloc&) {
            return fail(TRRE_E_TOO_BIG, "error: out of host memory for generator mode");
        }
        hipStream_t s = static_cast<hipStream_t>(stream);
        HIP_TRY(hipMemcpyAsync(host.data(), d_in, n, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        bool diverged = false;
        rc = generate_on(p, st, host.data(), n, result, diverged);
        if (rc) return rc;
        if (out_len) *out_len = result.size();
        if (result.size() > cap) return fail(TRRE_E_CAPACITY, "error: output buffer too small");
        if (!result.empty()) HIP_TRY(hipMemcpyAsync(d_out, result.data(), result.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        return diverged ? fail(TRRE_E_DIVERGES, kDivergeMsg) : TRRE_OK;
    }
    const int fam = scan_family(*p);
    rc = enqueue(p, st, &st->ctx, fam, d_in, n, d_out, cap, static_cast<hipStream_t>(stream));
    if (rc) { st->ctx.pend = Pending(); return rc; }
    return finish(p, st, &st->ctx, out_len);
}

namespace {
// grow one direction of a host slot (pinned staging + device buffer); need_pin false: the caller's own buffer is pinned, no staging on this side
int slot_reserve(DeviceState::HostSlot& hs, bool input, size_t bytes, bool need_pin) {
    uint8_t*& pin = input ? hs.pin_in : hs.pin_out;
    uint8_t*& dev = input ? hs.d_in : hs.d_out;
    size_t& have = input ? hs.in_cap : hs.out_cap;
    if (have >= bytes && (pin || !need_pin)) return TRRE_OK;
    const bool grow = have < bytes;
    const size_t want = grow ? bytes + bytes / 8 : have;  // (head room: the next chunk is rarely the same size)
    if (grow) {
        if (dev) (void)hipFree(dev);
        pinned_put(pin);
        pin = nullptr; dev = nullptr; have = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), want + 64));
    }
    if (need_pin && !pin) HIP_TRY(pinned_get(&pin, want + 64));
    have = want;
    return TRRE_OK;
}
constexpr size_t kHostChunk = (size_t)32 << 20;

// Staging copies between the caller's pageable buffers and pinned memory are spread over a few threads:
// one thread moves ~10 GB/s, a PCIe 5 x16 link ~55 GB/s each way.
class CopyPool {
public:
    static CopyPool& get() { static CopyPool pool; return pool; }
    struct Job { int left = 0; };
    // copies n bytes with up to `ways` workers; returns when done
    void copy(void* dst, const void* src, size_t n, int ways) {
        if (n < ((size_t)4 << 20) || ways <= 1) { std::memcpy(dst, src, n); return; }
        Job job;
        start(job, dst, src, n, ways);
        wait(job);
    }
    // the same in two halves: queue the pieces, collect later (the job must stay alive until wait() returns)
    void start(Job& job, void* dst, const void* src, size_t n, int ways) {
        if (n == 0) return;
        const size_t piece = std::max<size_t>((n / (size_t)(ways > 0 ? ways : 1) + 4095) & ~(size_t)4095, (size_t)1 << 20);
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t off = 0; off < n; off += piece) {
            queue_.push_back(Piece{&job, static_cast<uint8_t*>(dst) + off, static_cast<const uint8_t*>(src) + off, std::min(piece, n - off)});
            ++job.left;
        }
        cv_.notify_all();
    }
    void wait(Job& job) {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return job.left == 0; });
    }

private:
    struct Piece { Job* job; uint8_t* dst; const uint8_t* src; size_t n; };
    CopyPool() {
        // (a copy asks for kCopyWays = 8 workers; trre_scan_host_multi runs one staging copy in and one out per device at a time: on a host with
        // the cores for it the pool holds 8 x 8 workers, so that eight devices do not queue behind two devices' worth of them)
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned n = hw >= 256 ? 64 : (hw >= 128 ? 32 : (hw >= 32 ? 16 : (hw >= 8 ? hw / 2 : 2)));
        for (unsigned i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    void run() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
            if (stop_) return;
            Piece pc = queue_.back();
            queue_.pop_back();
            lk.unlock();
            std::memcpy(pc.dst, pc.src, pc.n);
            lk.lock();
            if (--pc.job->left == 0) done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<Piece> queue_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};
constexpr int kCopyWays = 8;         // (4 and 16 measured the same through the command line, round 5)

// Host buffers on one device.  The input goes through in chunks cut at line ends (lines are independent, so
// the chunks' outputs simply concatenate), kHostSlots chunks in flight, each on its slot's stream: while one
// chunk is scanned the next is staged into pinned memory and copied up, and the previous one's output comes
// down and is copied out.  `out` may be null when cap == 0 (size query).
int scan_host_on(trre_prog* p, DeviceState* st, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len) {
    for (auto& hs : st->slot) {
        if (!hs.stream) HIP_TRY(hipStreamCreateWithFlags(&hs.stream, hipStreamNonBlocking));
        int rc = ctx_init(hs.ctx);
        if (rc) return rc;
    }
    const int fam = scan_family(*p);
    const bool fixed_len = !is_gen(fam);              // output size == input size (unless a NUL forces a general family)
    // A caller's buffer that is pinned already (hipHostMalloc / hipHostRegister) goes over the link as it
    // is: no staging copy on that side (round 5; the copies run at 116 GB/s on 8 threads of one pool, which eight devices share).
    auto is_pinned = [](const void* ptr, size_t len) -> bool {
        if (!ptr || !len) return false;
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (at.type != hipMemoryTypeHost) return false;
        if (hipPointerGetAttributes(&at, static_cast<const uint8_t*>(ptr) + len - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    };
    static const bool no_direct = getenv("TRRE_NO_PINNED_DIRECT") != nullptr;      // (A/B runs)
    const bool in_direct = !no_direct && is_pinned(in, n), out_direct = !no_direct && cap && is_pinned(out, cap);

    struct Chunk { size_t off = 0, len = 0, out_at = 0, m = 0, early_at = 0; bool submitted = false, copying = false, leaving = false, early = false; CopyPool::Job job; };
    Chunk ch[kHostSlots];
    size_t off = 0, total = 0;                        // input consumed, output produced (or needed)
    size_t total_bound = 0;                           // length-preserving: output of everything submitted so far
    bool overflow = false;                            // the caller's buffer is too small: keep counting only
    bool diverged = false;                            // a chunk reported TRRE_E_DIVERGES: its partial output is the last thing that counts
    std::string diverge_msg;
    int rc = TRRE_OK;
    auto abandon = [&](int code) -> int {             // leave nothing queued behind an error
        for (auto& hs : st->slot) { (void)hipStreamSynchronize(hs.stream); hs.ctx.pend = Pending(); }
        for (Chunk& c : ch)
            if (c.leaving) { CopyPool::get().wait(c.job); c.leaving = false; }
        return code;
    };
    // stage chunk k in and queue its upload + scan (and, when the output size is known beforehand, its download)
    static const bool trace_on = getenv("TRRE_TRACE") != nullptr;
    double t_reserve = 0, t_first = 0;
    bool first_done = false;
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    auto submit = [&](int b, size_t at, size_t len) -> int {
        DeviceState::HostSlot& hs = st->slot[b];
        const auto tr = std::chrono::steady_clock::now();
        int r = slot_reserve(hs, true, len, !in_direct);
        if (r) return r;
        r = slot_reserve(hs, false, fixed_len ? len : len + len / 2 + 4096, !out_direct);
        if (r) return r;
        t_reserve += ms_since(tr);
        if (in_direct) {
            HIP_TRY(hipMemcpyAsync(hs.d_in, in + at, len, hipMemcpyHostToDevice, hs.stream));
        } else {
            CopyPool::get().copy(hs.pin_in, in + at, len, kCopyWays);
            HIP_TRY(hipMemcpyAsync(hs.d_in, hs.pin_in, len, hipMemcpyHostToDevice, hs.stream));
        }
        ch[b].off = at; ch[b].len = len; ch[b].out_at = 0; ch[b].m = 0; ch[b].submitted = true; ch[b].early = false;
        hs.ctx.relaunches = 0;
        const auto te = std::chrono::steady_clock::now();
        r = enqueue(p, st, &hs.ctx, fam, hs.d_in, len, hs.d_out, hs.out_cap, hs.stream);
        if (r) return r;
        if (!first_done) { t_first = ms_since(te); first_done = true; }
        if (fixed_len && !overflow && total_bound + len <= cap) {
            // length-preserving: the output is `len` bytes unless a NUL turns up (then complete() downloads again).  (Straight to its
            // place when the caller's buffer is pinned — if the chunks before it come out shorter after all, complete() downloads again.)
            HIP_TRY(hipMemcpyAsync(out_direct ? out + total_bound : hs.pin_out, hs.d_out, len, hipMemcpyDeviceToHost, hs.stream));
            ch[b].early = true;
            ch[b].early_at = total_bound;
        }
        total_bound += len;
        return TRRE_OK;
    };
    // wait for chunk b's scan, learn its output size, queue the download (unless it is already there)
    auto complete = [&](int b) -> int {
        DeviceState::HostSlot& hs = st->slot[b];
        size_t m = 0;
        int r = finish(p, st, &hs.ctx, &m);
        bool again = false;
        while (r == TRRE_E_CAPACITY) {                // the general families report the size they need
            r = slot_reserve(hs, false, m + 4096, !out_direct);
            if (r) return r;
            r = enqueue(p, st, &hs.ctx, fam, hs.d_in, ch[b].len, hs.d_out, hs.out_cap, hs.stream);
            if (!r) r = finish(p, st, &hs.ctx, &m);
            again = true;
        }
        if (r == TRRE_E_DIVERGES) {                   // the reference stops inside this chunk: what it had printed (m bytes) still counts
            diverged = true;
            diverge_msg = g_error;
            again = true;
        } else if (r) {
            return r;
        }
        ch[b].m = m;
        ch[b].out_at = total;
        if (!overflow && total + m > cap) overflow = true;
        if (!overflow && m) {
            // (the early download holds the FIRST launch's bytes: a relaunch inside finish() — mask scratch for a long
            // line, a NUL, a bounded fold that overflowed — rewrote d_out afterwards)
            if (!(ch[b].early && m == ch[b].len && !again && hs.ctx.relaunches == 0 && (!out_direct || ch[b].early_at == total)))
                HIP_TRY(hipMemcpyAsync(out_direct ? out + total : hs.pin_out, hs.d_out, m, hipMemcpyDeviceToHost, hs.stream));
            ch[b].copying = true;
        }
        total += m;
        ch[b].submitted = false;
        return TRRE_OK;
    };
    // chunk output: pinned -> caller, once its download has finished.  The copy is only started here (it runs on the
    // pool's threads while this thread stages the next chunk in); settle() waits for it before the slot is used again.
    auto drain = [&](int b) -> int {
        if (!ch[b].copying) return TRRE_OK;
        HIP_TRY(hipStreamSynchronize(st->slot[b].stream));
        ch[b].copying = false;
        if (out_direct) return TRRE_OK;            // (the download went to its place)
        CopyPool::get().start(ch[b].job, out + ch[b].out_at, st->slot[b].pin_out, ch[b].m, kCopyWays);
        ch[b].leaving = true;
        return TRRE_OK;
    };
    auto settle = [&](int b) {
        if (ch[b].leaving) { CopyPool::get().wait(ch[b].job); ch[b].leaving = false; }
    };
    // software pipeline, per iteration: chunk k is staged in by this thread (while chunk k-1 is on the device and chunk
    // k-2 is copied out by the pool), uploaded and scanned; then chunk k-1 is collected and its copy-out started
    for (int64_t k = 0;; ++k) {
        const bool more = off < n;
        if (more) {
            const int b = (int)(k % kHostSlots);
            rc = drain(b);                            // the slot's previous output must have left its staging buffer
            if (rc) return abandon(rc);
            settle(b);
            size_t len = n - off;                     // this chunk: up to kHostChunk bytes, extended to the end of its last line
            if (len > kHostChunk) {
                const void* nl = std::memchr(in + off + kHostChunk - 1, '\n', n - off - (kHostChunk - 1));
                len = nl ? (size_t)(static_cast<const uint8_t*>(nl) - (in + off)) + 1 : n - off;
            }
            rc = submit(b, off, len);
            if (rc) return abandon(rc);
            off += len;
        }
        if (k >= 1) {
            const int b1 = (int)((k - 1) % kHostSlots);
            if (ch[b1].submitted) { rc = complete(b1); if (rc) return abandon(rc); }
            rc = drain(b1);
            if (rc) return abandon(rc);
            if (diverged) {
                // nothing after the diverging chunk counts: the chunk submitted in this iteration is dropped
                off = n;
                for (int b = 0; b < kHostSlots; ++b)
                    if (ch[b].submitted) {
                        (void)hipStreamSynchronize(st->slot[b].stream);
                        st->slot[b].ctx.pend = Pending();
                        ch[b].submitted = false;
                    }
            }
        }
        if (!more) {
            bool busy = false;
            for (const Chunk& c : ch) busy = busy || c.submitted || c.copying;
            if (!busy) break;
        }
    }
    for (int b = 0; b < kHostSlots; ++b) settle(b);
    if (trace_on && t_reserve + t_first > 1.0)
        fprintf(stderr, "trre: host call of %zu bytes: %.1f ms getting staging and device buffers, %.1f ms in the first chunk's launch calls\n", n, t_reserve, t_first);
    if (out_len) *out_len = total;
    if (overflow) return fail(TRRE_E_CAPACITY, "error: output buffer too small");
    if (diverged) return fail(TRRE_E_DIVERGES, diverge_msg);
    return TRRE_OK;
}

// RAII: make `device` current for the calling thread, restore the previous one on return
struct DeviceScope {
    int prev = -1;
    hipError_t err;
    explicit DeviceScope(int device) {
        static const bool trace_on = getenv("TRRE_TRACE") != nullptr;
        static std::atomic<bool> first{true};
        const auto t0 = std::chrono::steady_clock::now();
        err = hipGetDevice(&prev);
        if (err == hipSuccess) err = hipSetDevice(device);
        if (trace_on && first.exchange(false))
            fprintf(stderr, "trre: first HIP calls of the process (runtime start-up): %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

int trre_scan_host(trre_prog* p, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, int device) {
    g_scan_flags = 0;
    if (!p || (n && !in) || (cap && !out)) return fail(TRRE_E_ARG, "error: null argument");
    if (out_len) *out_len = 0;
    if (n == 0) return TRRE_OK;
    DeviceScope scope(device);
    HIP_TRY(scope.err);
    DeviceState* st;
    int rc = device_state(p, device, &st);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(st->mu);
    if (is_generate(p->mode)) {
        std::vector<uint8_t> result;
        bool diverged = false;
        rc = generate_on(p, st, in, n, result, diverged);
        if (rc) return rc;
        if (out_len) *out_len = result.size();
        if (result.size() > cap) return fail(TRRE_E_CAPACITY, "error: output buffer too small");
        if (!result.empty()) std::memcpy(out, result.data(), result.size());
        return diverged ? fail(TRRE_E_DIVERGES, kDivergeMsg) : TRRE_OK;
    }
    return scan_host_on(p, st, in, n, out, cap, out_len);
}

// Host buffers, line-sharded over several GPUs of this node: the reference's scan has no exchange step
// (a line's output depends on that line and the read-only program, trre_nft.c:776-790), so the input is
// cut at line ends into one contiguous shard per device (trre_shard_bounds), every device scans its shard on
// its own host thread and streams, and the outputs are concatenated in shard order — an exclusive sum of G
// sizes on the host, no collective.
namespace {
// The shards of one call: cut at line ends, every shard scanned by `scan_shard` on a host thread of its own, the outputs put together in
// shard order.  (A function of its own so that the CPU test tier can drive the reassembly — eight shards, one of them ending the output —
// with a stand-in for the device call: trre_debug_scan_host_multi.)
using ShardFn = std::function<int(int shard, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len)>;
int scan_shards(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, int G, bool fixed_len, const ShardFn& scan_shard) {
    std::vector<size_t> bounds((size_t)G + 1);
    int rc = trre_shard_bounds(in, n, G, bounds.data());
    if (rc) return rc;
    // A length-preserving program writes every shard straight to its place (output offset == input offset);
    // otherwise a shard's offset is known only when the shards before it are done: each goes to a buffer of its
    // own and is moved into place afterwards.
    struct Shard { int rc = TRRE_OK; size_t m = 0; std::string err; std::vector<uint8_t> buf; bool own = false; uint32_t flags = 0; };
    std::vector<Shard> sh((size_t)G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) {
        th.emplace_back([&, g] {
            Shard& s = sh[(size_t)g];
            try {
                const size_t lo = bounds[(size_t)g], len = bounds[(size_t)g + 1] - lo;
                if (len == 0) return;
                const bool direct = fixed_len && lo + len <= cap;
                uint8_t* dst = nullptr;
                size_t room = 0;
                // cap == 0 is a size query; otherwise a shard that cannot go straight to its place gets a buffer of its
                // own — also for a length-preserving program whose place lies beyond `cap`: NUL bytes may shorten the
                // shards before it, and the call must succeed whenever the TOTAL fits
                if (direct) { dst = out + lo; room = len; }
                else if (cap > 0) { s.buf.resize(fixed_len ? len + 64 : len + len / 2 + 4096); s.own = true; dst = s.buf.data(); room = s.buf.size(); }
                g_scan_flags = 0;
                s.rc = scan_shard(g, in + lo, len, dst, room, &s.m);
                if (s.rc == TRRE_E_CAPACITY && s.own) {          // the shard needs s.m bytes
                    s.buf.resize(s.m + 64);
                    s.rc = scan_shard(g, in + lo, len, s.buf.data(), s.buf.size(), &s.m);
                }
                if (s.rc && s.rc != TRRE_E_CAPACITY) s.err = trre_last_error();
                s.flags = g_scan_flags;                          // (this thread's: handed to the caller's below)
            } catch (const std::bad_alloc&) {                    // a multi-GB shard buffer: an error code, not std::terminate
                s.rc = TRRE_E_TOO_BIG;
                s.err = "error: out of host memory for a shard's output buffer";
            } catch (const std::exception& e) {
                s.rc = TRRE_E_DEVICE;
                s.err = std::string("error: ") + e.what();
            }
        });
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < G; ++g) g_scan_flags |= sh[(size_t)g].flags;
    size_t total = 0;
    bool short_cap = false;
    int last = G, diverged_at = -1;                   // shards [0, last) count; a shard on which the reference stops is the last one
    for (int g = 0; g < G; ++g) {
        const Shard& s = sh[(size_t)g];
        if (s.rc == TRRE_E_DIVERGES) { diverged_at = g; last = g + 1; total += s.m; break; }
        if (s.rc && s.rc != TRRE_E_CAPACITY) return fail(s.rc, s.err);
        if (s.rc == TRRE_E_CAPACITY) short_cap = true;
        total += s.m;
    }
    if (out_len) *out_len = total;
    if (short_cap || total > cap) return fail(TRRE_E_CAPACITY, "error: output buffer too small");   // (short_cap: a size query, cap == 0)
    // move the shards into place, in order (a direct shard that came out shorter — NUL bytes — moves down)
    size_t at = 0;
    for (int g = 0; g < last; ++g) {
        const Shard& s = sh[(size_t)g];
        const uint8_t* src = s.own ? s.buf.data() : out + bounds[(size_t)g];
        if (s.m && src != out + at) std::memmove(out + at, src, s.m);
        at += s.m;
    }
    if (diverged_at >= 0) return fail(TRRE_E_DIVERGES, sh[(size_t)diverged_at].err);
    return TRRE_OK;
}
}  // namespace

int trre_scan_host_multi(trre_prog* p, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, uint32_t device_mask) {
    g_scan_flags = 0;
    if (!p || (n && !in) || (cap && !out)) return fail(TRRE_E_ARG, "error: null argument");
    if (out_len) *out_len = 0;
    int n_dev = 0;
    {
        static const bool trace_on = getenv("TRRE_TRACE") != nullptr;
        static std::atomic<bool> first{true};
        const auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipGetDeviceCount(&n_dev));
        if (trace_on && first.exchange(false))
            fprintf(stderr, "trre: hipGetDeviceCount, the first HIP call of the process (runtime start-up): %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::vector<int> devs;
    for (int d = 0; d < n_dev && d < 32; ++d)
        if (device_mask == 0 || (device_mask >> d & 1u)) devs.push_back(d);
    if (devs.empty()) return fail(TRRE_E_ARG, "error: device_mask selects no visible device");
    if (n == 0) return TRRE_OK;
    // TRRE_SHARDS_PER_DEVICE=k (tests): k shards per selected device, so that the sharding, the per-shard threads and
    // the reassembly run on a box with a single GPU (shards on one device take turns: calls are serialised per device)
    static const int per_dev = getenv("TRRE_SHARDS_PER_DEVICE") ? atoi(getenv("TRRE_SHARDS_PER_DEVICE")) : 1;
    if (per_dev > 1) {
        std::vector<int> rep;
        for (int d : devs) rep.insert(rep.end(), (size_t)per_dev, d);
        devs.swap(rep);
    }
    const int G = (int)devs.size();
    if (G == 1) return trre_scan_host(p, in, n, out, cap, out_len, devs[0]);
    const int fam = scan_family(*p);
    const bool fixed_len = !is_gen(fam) && !is_generate(p->mode);
    const uint32_t flags0 = g_scan_flags;
    const int rc = scan_shards(in, n, out, cap, out_len, G, fixed_len,
                               [&](int g, const uint8_t* sin, size_t sn, uint8_t* sout, size_t scap, size_t* sm) { return trre_scan_host(p, sin, sn, sout, scap, sm, devs[(size_t)g]); });
    g_scan_flags |= flags0;
    return rc;
}

// CPU test tier: the sharding and the reassembly of trre_scan_host_multi with the caller's stand-in for the per-shard device call.
int trre_debug_scan_host_multi(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, int n_shards, int fixed_len, trre_debug_shard_fn fn, void* user) {
    g_scan_flags = 0;
    if ((n && !in) || (cap && !out) || n_shards < 1 || !fn) return fail(TRRE_E_ARG, "error: null argument");
    if (out_len) *out_len = 0;
    if (n == 0) return TRRE_OK;
    return scan_shards(in, n, out, cap, out_len, n_shards, fixed_len != 0,
                       [&](int g, const uint8_t* sin, size_t sn, uint8_t* sout, size_t scap, size_t* sm) { return fn(user, g, sin, sn, sout, scap, sm); });
}

int trre_set_profiling(trre_prog* p, int on) {
    if (!p) return fail(TRRE_E_ARG, "error: null argument");
    p->profiling = on != 0;
    return TRRE_OK;
}

int trre_last_kernel_ms(trre_prog* p, float* ms) {
    if (!p || !ms) return fail(TRRE_E_ARG, "error: null argument");
    *ms = p->last_ms.load();
    return *ms < 0 ? TRRE_E_ARG : TRRE_OK;
}

int trre_shard_bounds(const uint8_t* in, size_t n, int nshards, size_t* bounds) {
    if (nshards < 1 || !bounds || (n && !in)) return fail(TRRE_E_ARG, "error: bad shard request");
    bounds[0] = 0;
    for (int s = 1; s < nshards; ++s) {
        size_t cut = n / (size_t)nshards * (size_t)s;
        if (cut < bounds[s - 1]) cut = bounds[s - 1];
        const void* nl = cut < n ? std::memchr(in + cut, '\n', n - cut) : nullptr;
        bounds[s] = nl ? (size_t)(static_cast<const uint8_t*>(nl) - in) + 1 : n;
    }
    bounds[nshards] = n;
    return TRRE_OK;
}

}  // extern "C"
