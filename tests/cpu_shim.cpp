// cpu_shim.cpp — TEST INFRASTRUCTURE: runs the kernels' per-thread phase bodies
// (trre_amd/csrc/scan_block.hpp, scan_core.hpp) on the host, thread by thread,
// with barriers replaced by loop boundaries, so that the lane logic, the tile
// geometry, chunk ownership, the long-line slow path and the copy-out can be
// checked against the oracle in the CPU test tier (no GPU in the build
// container).  It is NOT a product path: libtrre_mi355x.so has no CPU scan and
// nothing in trre_amd/ links this file.
//
// Besides the production geometry it instantiates a tiny one (4 lanes, 64-byte
// chunks, 32-byte halo) so that small inputs cross chunk and tile boundaries.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../trre_amd/csrc/scan_block.hpp"
#include "../trre_amd/csrc/splice_block.hpp"
#include "../trre_amd/csrc/gen_block.hpp"
#include "../trre_amd/csrc/lazy_block.hpp"
#include "../trre_amd/csrc/one_block.hpp"
#include "../trre_amd/csrc/map_block.hpp"
#include "../trre_amd/csrc/guard_block.hpp"

using namespace trre;

namespace {

int g_last_rounds = 0;       // repair rounds of the last exact-sub-range run (shim_last_rounds)

constexpr uint32_t kFlagG16SlowBit = 1u << 3;    // front.hpp: kFlagG16Slow
using GeoTiny = Geometry<4, 64, 32>;

template <class G, class Engine>
struct Block {
    std::vector<uint8_t> tin, tout, tab, mask;
    Block() : tin(G::TILE_ALLOC + 16), tout(G::TILE_ALLOC + 16), tab(Engine::kLdsBytes + 16),
              mask((size_t)G::TILE_ALLOC * (Engine::kMaskBytes ? Engine::kMaskBytes : 1) + 16) {}
    uint8_t* al(std::vector<uint8_t>& v) { return reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(v.data()) + 15) & ~(uintptr_t)15); }
};

template <class G, class Engine>
void run_lp(const ScanArgs& a, int64_t n_chunks, uint32_t& status) {
    Block<G, Engine> blk;
    uint8_t *tin = blk.al(blk.tin), *tout = blk.al(blk.tout), *tab = blk.al(blk.tab), *mask = blk.al(blk.mask);
    for (int64_t b = 0; b < n_chunks; ++b) {
        const int64_t v0 = b * G::CHUNK - G::PRE;
        for (int t = 0; t < G::THREADS; ++t) { Engine::stage(a.blob, tab, t, G::THREADS); tile_load<G>(a, v0, tin, t); }
        std::memset(tout, 0xEE, G::TILE_ALLOC);          // poison: unwritten output must never be copied out
        const typename Engine::View T = Engine::view(a.blob, tab);
        int32_t first = 0x7fffffff, last = -1;
        for (int t = 0; t < G::THREADS; ++t) {
            typename Engine::Lane L = Engine::make_lane(mask);
            int32_t f, l;
            lane_walk_lp<G, Engine>(a, T, L, v0, tin, tout, t, f, l, status);
            if (f < first) first = f;
            if (l > last) last = l;
        }
        for (int t = 0; t < G::THREADS; ++t) tile_store_lp<G>(a, v0, tout, first, last, t);
    }
}

template <class G, class Engine>
void run_gen(ScanArgs a, int64_t n_chunks, uint32_t& status, uint64_t& total_out) {
    Block<G, Engine> blk;
    uint8_t *tin = blk.al(blk.tin), *tout = blk.al(blk.tout), *tab = blk.al(blk.tab), *mask = blk.al(blk.mask);
    std::vector<uint32_t> lane_counts((size_t)n_chunks * G::THREADS);
    std::vector<uint64_t> chunk_total(n_chunks), chunk_base(n_chunks + 1);
    for (int64_t b = 0; b < n_chunks; ++b) {             // pass 1: count
        const int64_t v0 = b * G::CHUNK - G::PRE;
        for (int t = 0; t < G::THREADS; ++t) { Engine::stage(a.blob, tab, t, G::THREADS); tile_load<G>(a, v0, tin, t); }
        const typename Engine::View T = Engine::view(a.blob, tab);
        uint64_t tot = 0;
        for (int t = 0; t < G::THREADS; ++t) {
            typename Engine::Lane L = Engine::make_lane(mask);
            CountSink s;
            lane_walk_gen<G, Engine>(a, T, L, v0, tin, t, s, status);
            lane_counts[(size_t)b * G::THREADS + t] = (uint32_t)s.n;
            tot += s.n;
        }
        chunk_total[b] = tot;
    }
    uint64_t run = 0;                                    // chunk scan
    for (int64_t b = 0; b < n_chunks; ++b) { chunk_base[b] = run; run += chunk_total[b]; }
    chunk_base[n_chunks] = run;
    total_out = run;
    if (run > a.cap) { status |= kStCapacity; return; }
    for (int64_t b = 0; b < n_chunks; ++b) {             // pass 2: emit
        const int64_t v0 = b * G::CHUNK - G::PRE;
        for (int t = 0; t < G::THREADS; ++t) { Engine::stage(a.blob, tab, t, G::THREADS); tile_load<G>(a, v0, tin, t); }
        std::memset(tout, 0xEE, G::TILE_ALLOC);
        const typename Engine::View T = Engine::view(a.blob, tab);
        const uint64_t total = chunk_total[b], gbase = chunk_base[b];
        const int shift = (int)((reinterpret_cast<uintptr_t>(a.out) + gbase) & 15u);
        const bool staged = (uint64_t)shift + total <= (uint64_t)G::TILE;
        uint64_t lane_base = 0;
        for (int t = 0; t < G::THREADS; ++t) {
            typename Engine::Lane L = Engine::make_lane(mask);
            ByteSink s{staged ? tout + shift + lane_base : a.out + gbase + lane_base};
            lane_walk_gen<G, Engine>(a, T, L, v0, tin, t, s, status);
            if (s.n != lane_counts[(size_t)b * G::THREADS + t]) status |= 1u << 30;   // count/emit disagree
            lane_base += lane_counts[(size_t)b * G::THREADS + t];
        }
        if (staged)
            for (int t = 0; t < G::THREADS; ++t) tile_store_seq<G>(a.out + gbase, tout, shift, (int64_t)total, t);
    }
}

// ---- direct stream kernels (no tile): lanes run one after the other --------------------------
StreamView direct_view(const ScanArgs& a) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    StreamView T;
    T.cls = a.blob + h.off_cls;
    T.ent = reinterpret_cast<const uint64_t*>(a.blob + h.off_ent);
    T.pool = a.blob + h.off_pool;
    T.long_pool = h.max_out >= 255u;
    if (h.g16_bytes) T.g16 = a.blob + h.off_g16;
    if (h.p32_bytes && !getenv("TRRE_NO_PAIRS")) { T.p32 = a.blob + h.off_p32; T.p32_slow = h.p32_slow; }
    return T;
}
template <int kSym = 0>
void run_direct_lp(const ScanArgs& a, int64_t lane_bytes, uint32_t& status) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const StreamView T = direct_view(a);
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    alignas(16) uint8_t ring[kRingStride];
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) {      // any order: lanes write disjoint bytes
        DirectLane L;
        stream_direct_lane<0, false, (kSym != 0)>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
    }
}
// length-preserving as the runtime launches it: the emit pass alone, lanes writing their lines in place
template <int kSym = 0>
void run_direct_lp_emit(ScanArgs a, int64_t lane_bytes, uint32_t& status, bool g16) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const StreamView T = direct_view(a);
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    alignas(16) uint8_t ring[kRingStride];
    a.lp_emit = 1;
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) {
        DirectLane L;
        if (g16 && (h.flags & kFlagG16SlowBit)) g16_lane<2, kSym, true>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
        else if (g16) g16_lane<2, kSym, false>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
        else stream_direct_lane<2, false, (kSym != 0)>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
    }
}
// backward pass of the guided families, lane by lane (any order: lanes write disjoint symbols)
void run_rev_sweep(const ScanArgs& a, int64_t lane_bytes, bool packed) {
    const RevBlobHeader& h = *reinterpret_cast<const RevBlobHeader*>(a.rblob);
    const RevView T{a.rblob + h.off_wide};
    const int64_t vtop = packed ? (a.vend + 127) & ~(int64_t)127 : (a.vend + 63) & ~(int64_t)63;
    const int64_t n_lanes = (vtop + lane_bytes - 1) / lane_bytes;
    for (int64_t lane = 0; lane < n_lanes; ++lane) {
        // (the tiny geometry bounds the look-ahead at 256 bytes: every line of a few hundred bytes has a walker)
        const int64_t max_look = lane_bytes <= 128 ? 256 : kRevMaxLook;
        if (packed) rev_sweep_lane<0, true>(a, T, lane, lane_bytes, nullptr, max_look);
        else rev_sweep_lane<0, false>(a, T, lane, lane_bytes, nullptr, max_look);
    }
}
// The backward pass with exact sub-ranges (round 5), as runtime.cpp drives it: every lane from a guessed (or known) state at the end of its
// sub-range, k_rev_verify's rule, repair rounds until every guess is what the lane to the right found.  Returns the repair rounds.
int run_rev_sweep_exact(ScanArgs a, int64_t lane_bytes, bool packed, uint32_t look) {
    const RevBlobHeader& h = *reinterpret_cast<const RevBlobHeader*>(a.rblob);
    const RevView T{a.rblob + h.off_wide};
    const int64_t vtop = packed ? (a.vend + 127) & ~(int64_t)127 : (a.vend + 63) & ~(int64_t)63;
    const int64_t n_lanes = (vtop + lane_bytes - 1) / lane_bytes;
    std::vector<uint32_t> guess(n_lanes, 0xEEEEEEEu), flags(n_lanes, 0);
    a.rev_guess = guess.data(); a.rev_flags = flags.data();
    a.exact = 1; a.spec_look = look;
    for (int64_t lane = 0; lane < n_lanes; ++lane) {
        if (packed) rev_sweep_lane<0, true>(a, T, lane, lane_bytes, nullptr);
        else rev_sweep_lane<0, false>(a, T, lane, lane_bytes, nullptr);
    }
    int rounds = 0;
    for (;;) {
        int64_t bad = 0;
        for (int64_t lane = 0; lane < n_lanes; ++lane) {
            flags[lane] = lane + 1 < n_lanes && (packed ? rev_guess_wrong<true>(a, lane, lane_bytes) : rev_guess_wrong<false>(a, lane, lane_bytes));
            bad += flags[lane];
        }
        if (!bad) return rounds;
        if (++rounds > n_lanes + 8) return -1;
        for (int64_t lane = 0; lane < n_lanes; ++lane) {
            if (!flags[lane]) continue;
            if (packed) rev_repair_lane<true>(a, T, lane, lane_bytes);
            else rev_repair_lane<false>(a, T, lane, lane_bytes);
        }
    }
}
// g16: walk the 16-byte entries (when the tables have them), like k_stream_g16
template <int kSym = 0>
void run_direct_gen(const ScanArgs& a, int64_t lane_bytes, uint32_t& status, uint64_t& total_out, bool g16) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const StreamView T = direct_view(a);
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    alignas(16) uint8_t ring[kRingStride];
    std::vector<uint64_t> cnt(n_lanes);
    for (int64_t lane = 0; lane < n_lanes; ++lane) {
        DirectLane L;
        if (g16 && (h.flags & kFlagG16SlowBit)) g16_lane<1, kSym, true>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
        else if (g16) g16_lane<1, kSym, false>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
        else stream_direct_lane<1, false, (kSym != 0)>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
        cnt[lane] = L.count;
    }
    uint64_t run = 0;
    std::vector<uint64_t> base(n_lanes);
    for (int64_t lane = 0; lane < n_lanes; ++lane) { base[lane] = run; run += cnt[lane]; }
    total_out = run;
    if (run > a.cap) { status |= kStCapacity; return; }
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) {
        DirectLane L;
        if (g16 && (h.flags & kFlagG16SlowBit)) g16_lane<2, kSym, true>(a, T, h.n_cls, lane, lane_bytes, ring, base[lane], L, status);
        else if (g16) g16_lane<2, kSym, false>(a, T, h.n_cls, lane, lane_bytes, ring, base[lane], L, status);
        else stream_direct_lane<2, false, (kSym != 0)>(a, T, h.n_cls, lane, lane_bytes, ring, base[lane], L, status);
    }
}

// The same with EXACT SUB-RANGES (round 5; scan_block.hpp: ScanArgs::exact), as runtime.cpp drives it: the count pass guesses every lane's
// entry state from `look` bytes before its sub-range, k_spec_verify's rule flags the lanes whose guess is not the exit state of the lane
// before them, repair rounds (a flagged lane walks again, and on into the lanes behind it within its workgroup of `group` lanes) until
// none is flagged, then the emit pass from the verified states.  *rounds: repair rounds it took.
template <int kSym = 0>
void run_direct_gen_exact(ScanArgs a, int64_t lane_bytes, uint32_t& status, uint64_t& total_out, uint32_t look, int64_t group, int& rounds) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const StreamView T = direct_view(a);
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    alignas(16) uint8_t ring[kRingStride];
    std::vector<uint64_t> cnt(n_lanes);
    std::vector<uint32_t> entry(n_lanes, 0xEEEEEEE0u), exits(n_lanes, 0xDDDDDDD0u), flags(n_lanes, 0);
    a.entry_rows = entry.data(); a.exit_rows = exits.data(); a.spec_flags = flags.data();
    a.spec_look = look;
    const bool slow = (h.flags & kFlagG16SlowBit) != 0;
    auto count_lane = [&](int64_t lane) {
        DirectLane L;
        if (slow) g16_lane<1, kSym, true>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
        else g16_lane<1, kSym, false>(a, T, h.n_cls, lane, lane_bytes, ring, 0, L, status);
        cnt[lane] = L.count;
    };
    a.exact = 1;
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) count_lane(lane);
    rounds = 0;
    for (;;) {
        int64_t bad = 0;
        for (int64_t lane = 1; lane < n_lanes; ++lane) {
            flags[lane] = !(entry[lane] & 1u) && (entry[lane] & ~1u) != exits[lane - 1];
            bad += flags[lane];
        }
        if (!bad) break;
        ++rounds;
        if (rounds > n_lanes + 8) { status |= 1u << 29; return; }      // must not happen
        a.exact = 3;
        const std::vector<uint32_t> exits_before = exits;               // (what the threads of one launch read of the lanes before them: maybe stale)
        for (int64_t lane = n_lanes - 1; lane >= 1; --lane) {
            if (!flags[lane]) continue;
            const int64_t block_end = std::min(n_lanes, (lane / group + 1) * group);
            uint32_t state = (lane % 2) ? exits[lane - 1] : exits_before[lane - 1];
            const uint32_t skip_row = kSkipState * h.n_cls * 16u;
            for (int64_t j = lane;;) {
                entry[j] = state;
                if (state == skip_row && (j + 1) * lane_bytes < a.vend - 1 && !rev_has_newline(a, j * lane_bytes, (j + 1) * lane_bytes)) {
                    exits[j] = state;                                    // (as in k_stream_g16's repair round: SKIP travels without a walk)
                    cnt[j] = 0;
                    ++j;
                    if (j >= block_end || flags[j]) break;
                    if ((entry[j] & 1u) || (entry[j] & ~1u) == state) break;
                    continue;
                }
                count_lane(j);
                state = exits[j];
                ++j;
                if (j >= block_end || flags[j]) break;
                if ((entry[j] & 1u) || (entry[j] & ~1u) == state) break;
            }
        }
    }
    uint64_t run = 0;
    std::vector<uint64_t> base(n_lanes);
    for (int64_t lane = 0; lane < n_lanes; ++lane) { base[lane] = run; run += cnt[lane]; }
    total_out = run;
    if (run > a.cap) { status |= kStCapacity; return; }
    a.exact = 2;
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) {
        DirectLane L;
        if (slow) g16_lane<2, kSym, true>(a, T, h.n_cls, lane, lane_bytes, ring, base[lane], L, status);
        else g16_lane<2, kSym, false>(a, T, h.n_cls, lane, lane_bytes, ring, base[lane], L, status);
    }
}

// ONE walk (round 6; one_block.hpp) as k_stream_one runs it: tiles of `nl` lanes of `S` bytes; per tile the walk from guessed entry
// states into private regions of R bytes, the verification against the lane before with repair rounds, the prefix sums, the tile's place
// in the output (the running sum — what the look-back finds) with the check of the tile's first guess against the exit of the tile before,
// then the output line by line through one_mark / one_store_line.  kStOneVoid: the kernel would have left the buffer to the count / emit pair.
template <int kSym = 0>
void run_direct_gen_one(ScanArgs a, uint32_t S, uint32_t R, uint32_t look, int nl, uint32_t& status, uint64_t& total_out, int& rounds) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const StreamView T = direct_view(a);
    const int64_t n_lanes = (a.vend + S - 1) / S;
    const int64_t n_tiles = (n_lanes + nl - 1) / nl;
    const bool slow = (h.flags & kFlagG16SlowBit) != 0;
    a.spec_look = look;
    std::vector<uint8_t> regions_raw((size_t)nl * R + 64 + 16);
    uint8_t* regions = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(regions_raw.data()) + 15) & ~(uintptr_t)15);
    std::vector<uint32_t> offs(nl + 1), exits(nl), used(nl), known(nl), lens(nl), sts(nl);
    std::vector<uint8_t> need(nl);
    std::vector<uint8_t> marks((size_t)nl * R / 16 + 32);
    uint64_t base = 0;
    uint32_t prev_exit = 0;
    rounds = 0;
    for (int64_t tile = 0; tile < n_tiles; ++tile) {
        std::fill(regions_raw.begin(), regions_raw.end(), 0xCD);
        auto walk = [&](int t, uint32_t entry) {
            DirectLane L;
            L.entry = entry;
            uint32_t st = 0;
            const int64_t lane = tile * nl + t;
            if (slow) g16_lane<3, kSym, true>(a, T, h.n_cls, lane, (int64_t)S, regions + (size_t)t * R, (uint64_t)R, L, st);
            else g16_lane<3, kSym, false>(a, T, h.n_cls, lane, (int64_t)S, regions + (size_t)t * R, (uint64_t)R, L, st);
            exits[t] = L.exit; used[t] = L.entry; lens[t] = (uint32_t)L.count; sts[t] = st;
            if (entry == kOneGuess) known[t] = L.known;
        };
        for (int t = nl - 1; t >= 0; --t) walk(t, kOneGuess);
        bool gave_up = false;
        for (int round = 0;; ++round) {
            bool any = false;
            const std::vector<uint32_t> exits_before = exits;            // (what the threads read before any of them walks again)
            for (int t = 1; t < nl; ++t) {
                const bool live = (tile * nl + t) * (int64_t)S < a.vend;
                need[t] = live && !known[t] && used[t] != exits_before[t - 1];
                any = any || need[t];
            }
            if (!any) break;
            if (round + 1 >= kOneRounds) { gave_up = true; break; }
            ++rounds;
            for (int t = nl - 1; t >= 1; --t) if (need[t]) walk(t, exits_before[t - 1]);
        }
        uint32_t st = 0;
        for (int t = 0; t < nl; ++t) st |= sts[t];
        if (gave_up) st |= kStOneVoid;
        uint32_t run = 0;
        for (int t = 0; t < nl; ++t) { offs[t] = run; run += (st & kStOneVoid) ? 0u : lens[t]; }      // (a void tile: an empty output, as in the kernel)
        offs[nl] = run;
        const uint32_t total = run;
        if (tile > 0 && !known[0] && used[0] != prev_exit) st |= kStOneVoid;      // the guess nobody repairs in place
        prev_exit = exits[nl - 1];
        status |= st;
        OneTile tv{regions, offs.data(), marks.data(), R};
        const uint32_t hh = (uint32_t)((reinterpret_cast<uintptr_t>(a.out) + base) & 15u);
        std::fill(marks.begin(), marks.end(), 0xEE);
        for (int t = 0; t < nl; ++t) one_mark(tv, t, hh);
        const bool write = base + total <= a.cap;
        if (!write) status |= kStCapacity;
        const uint32_t n_lines = (hh + total + 15u) >> 4;
        for (uint32_t c = 0; c < n_lines; c += 2) {
            const uint32_t cc[2] = {c, c + 1};
            one_store_lines<2>(tv, a.out, base, hh, cc, total, write);
        }
        base += total;
    }
    total_out = base;
}

// A memoryless program in one pass (round 6; map_block.hpp) as k_mapgen runs it: tiles of nw waves x 8 rows x nl lanes x 8 bytes; per tile the
// lengths, the prefix sums in (wave, row, lane) order, the tile's place (the running sum: what the look-back finds), the bytes' texts into a
// window of `window` bytes — as many windows as the tile's output needs —, the window out line by line.  Interior tiles whose output fits the
// window take the kernel's fast bodies (<kEdge = false, kClip = false>), the others the clipped ones.  kStNul: a NUL — the launch would be void.
template <bool kFirst, bool kMulti>
void run_mapgen_t(const ScanArgs& a, const MapGenView& T, uint32_t& status, uint64_t& total_out, uint32_t window, int nw, int nl) {
    const int64_t row_bytes = (int64_t)nl * kMgLaneBytes, wave_bytes = row_bytes * kMgRows, tile_bytes = wave_bytes * nw;
    const int64_t n_tiles = (a.vend + tile_bytes - 1) / tile_bytes;
    std::vector<uint8_t> win_raw(window + 64 + 8);
    uint8_t* win = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(win_raw.data()) + 15) & ~(uintptr_t)15);
    const uint32_t sink = window + 40;               // (one for all: the lanes run one after the other here)
    const int lanes = nw * kMgRows * nl;
    std::vector<uint32_t> cnt(lanes), off(lanes), dlo(lanes), dhi(lanes);
    const int64_t vl = (a.vend + 15) & ~(int64_t)15;
    uint64_t base = 0;
    for (int64_t tile = 0; tile < n_tiles; ++tile) {
        const int64_t t0 = tile * tile_bytes;
        const bool edge = t0 < a.vbeg || t0 + tile_bytes > a.vend - 1;
        uint32_t total = 0, orsum = 0;
        for (int i = 0; i < lanes; ++i) {                // (wave, row, lane): the order of the input
            const int64_t v = t0 + (int64_t)i * kMgLaneBytes;
            uint32_t lo = 0, hi = 0;
            if (v < vl) { std::memcpy(&lo, a.in_v0 + v, 4); std::memcpy(&hi, a.in_v0 + v + 4, 4); }
            dlo[i] = lo; dhi[i] = hi;
            const uint32_t n = edge ? mg_count8<true>(T, lo, hi, mg_edge(v, a.vbeg, a.vend)) : mg_count8<false>(T, lo, hi, MgEdge{});
            orsum |= n;
            cnt[i] = n; off[i] = total; total += n;
        }
        if (orsum >= kMgNul) { status |= kStNul; continue; }      // (the kernel: nothing expanded, nothing stored; the launch is void)
        const bool write = base + total <= a.cap;
        if (!write) status |= kStCapacity;
        for (uint32_t wlo = 0; wlo < total; wlo += window) {
            const uint32_t whi = wlo + window < total ? wlo + window : total;
            const uint32_t hh = (uint32_t)((reinterpret_cast<uintptr_t>(a.out) + base + wlo) & 15u);
            std::fill(win_raw.begin(), win_raw.end(), 0xCD);
            for (int i = lanes - 1; i >= 0; --i) {       // (lanes in another order than the positions': nobody may lean on its neighbour's stores)
                const int64_t v = t0 + (int64_t)i * kMgLaneBytes;
                uint32_t end;
                if (total <= window && !edge) end = mg_expand8<kFirst, kMulti, false, false>(T, dlo[i], dhi[i], MgEdge{}, win, sink, off[i], 0u, 0u);
                else end = mg_expand8<kFirst, kMulti, true, true>(T, dlo[i], dhi[i], mg_edge(v, a.vbeg, a.vend), win, sink, off[i], wlo, whi - wlo);
                if (end != off[i] + cnt[i]) status |= 1u << 30;          // count and expand disagree
            }
            const uint32_t n_lines = (hh + (whi - wlo) + 15u) >> 4;
            for (uint32_t c = 0; c < n_lines; ++c) mg_store_line(win, a.out, base, wlo, whi - wlo, hh, c, write);
        }
        base += total;
    }
    total_out = base;
}
void run_mapgen(const ScanArgs& a, uint32_t& status, uint64_t& total_out, uint32_t window, int nw, int nl) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const uint32_t* mg = reinterpret_cast<const uint32_t*>(a.blob + h.off_mg);
    std::vector<uint8_t> len(256), first(256);
    std::vector<uint64_t> text(256);
    bool first_lookup = false;
    for (int k = 0; k < 256; ++k) {
        len[k] = (uint8_t)mg[4 * k + 2]; first[k] = (uint8_t)mg[4 * k]; text[k] = (uint64_t)mg[4 * k + 1] << 32 | mg[4 * k];
        if ((len[k] & (15u | kMgNul)) == 1u && first[k] != k) first_lookup = true;        // (runtime.cpp: MapGenArgs::first_lookup)
    }
    const MapGenView T{len.data(), first.data(), text.data()};
    const bool multi = h.mg_max > 1;
    if (first_lookup) multi ? run_mapgen_t<true, true>(a, T, status, total_out, window, nw, nl) : run_mapgen_t<true, false>(a, T, status, total_out, window, nw, nl);
    else multi ? run_mapgen_t<false, true>(a, T, status, total_out, window, nw, nl) : run_mapgen_t<false, false>(a, T, status, total_out, window, nw, nl);
}


// large tables in their fallback form (k_stream_fb): count, scan, emit — lane by lane
FbView fb_view(const ScanArgs& a) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    FbView T;
    T.cls = a.blob + h.off_cls;
    T.comb = reinterpret_cast<const uint64_t*>(a.blob + h.off_fb_comb);
    T.lit = reinterpret_cast<const uint64_t*>(a.blob + h.off_fb_lit);
    T.lit_meta = h.off_fb_lit_meta ? reinterpret_cast<const uint16_t*>(a.blob + h.off_fb_lit_meta) : nullptr;
    T.esc_slot = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc_slot);
    T.esc = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc);
    T.pool = a.blob + h.off_fb_pool;
    T.n_esc = h.fb_escs;
    for (int i = 0; i < 3; ++i) { T.start[i][0] = h.fb_start[i][0]; T.start[i][1] = h.fb_start[i][1]; }
    return T;
}
// emit8: the emit pass on the 8-byte rows (what the runtime launches: the two forms must agree lane by lane)
void run_fb_gen(const ScanArgs& a, int64_t lane_bytes, uint32_t& status, uint64_t& total_out, bool emit8) {
    const FbView T = fb_view(a);
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    alignas(16) uint8_t ring_room[kBRingStride];
    uint8_t* ring = ring_room + kBRingPad;
    std::vector<uint64_t> cnt(n_lanes), base(n_lanes);
    for (int64_t lane = 0; lane < n_lanes; ++lane) {
        DirectLane L;
        fb_lane<1>(a, T, lane, lane_bytes, ring, 0, L, status);
        cnt[lane] = L.count;
    }
    uint64_t run = 0;
    for (int64_t lane = 0; lane < n_lanes; ++lane) { base[lane] = run; run += cnt[lane]; }
    total_out = run;
    if (run > a.cap) { status |= kStCapacity; return; }
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const StreamView T8 = direct_view(a);
    alignas(16) uint8_t ring8[kRingStride];
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) {
        DirectLane L;
        if (emit8) stream_direct_lane<2, false, false>(a, T8, h.n_cls, lane, lane_bytes, ring8, base[lane], L, status);
        else fb_lane<2>(a, T, lane, lane_bytes, ring, base[lane], L, status);
    }
}

// The copy form of a large table (k_fb_mark / k_chunk_scan / k_fb_splice): the comb walk marks where the replacement texts go,
// the second pass — the wave-cooperative splice — walks no automaton.  ev_cap: ids per lane (small in the tests: the overflow route runs too).
void run_fb_copy(const ScanArgs& a, int64_t lane_bytes, uint32_t& status, uint64_t& total_out, uint32_t ev_cap, bool mark8 = false) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const FbView T = fb_view(a);
    const uint16_t* lit_meta = reinterpret_cast<const uint16_t*>(a.blob + h.off_fb_lit_meta);
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    std::vector<uint32_t> hdr((size_t)n_lanes * 4, 0xEEEEEEEEu);
    std::vector<uint32_t> events((size_t)n_lanes * ev_cap + 4, 0xEEEEEEEEu);
    FbCopyArgs ca{};
    ca.events = events.data();
    ca.lane_hdr = hdr.data();
    ca.ev_cap = ev_cap;
    std::vector<uint64_t> cnt(n_lanes), base(n_lanes);
    uint32_t stage[kMarkStageStride];
    // (the mark form of the comb where the tables have it — what the runtime launches —, or mark8: the 8-byte comb)
    Fb4View T4{};
    std::vector<uint8_t> cls4(256);
    if (h.fb4_slots && !mark8) {
        for (int c = 0; c < 256; ++c) cls4[c] = (uint8_t)(a.blob[h.off_cls + c] << 2);
        T4.cls4 = cls4.data();
        T4.comb4 = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_comb4);
        T4.dense4 = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_dense4);
        T4.lit_meta = lit_meta;
        T4.dense_base = reinterpret_cast<const uint16_t*>(a.blob + h.off_fb_dense_base);
        T4.esc_slot = T.esc_slot; T4.esc = T.esc; T4.n_esc = T.n_esc;
        T4.pad = h.fb_pad;
        for (int i = 0; i < 3; ++i) T4.start[i] = h.fb_start4[i];
    }
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) {      // any order
        DirectLane L;
        if (T4.comb4) fb_mark4_lane(a, T4, lane, lane_bytes, reinterpret_cast<uint8_t*>(stage), L, status, ca);
        else fb_lane<3>(a, T, lane, lane_bytes, reinterpret_cast<uint8_t*>(stage), 0, L, status, nullptr, &ca);
        cnt[lane] = L.count;
    }
    if (status & (kStEditOverflow | kStNul)) return;
    uint64_t run = 0;
    for (int64_t lane = 0; lane < n_lanes; ++lane) { base[lane] = run; run += cnt[lane]; }
    total_out = run;
    if (run > a.cap) { status |= kStCapacity; return; }
    std::vector<U128> lit(h.fb_lits + 1);
    const uint64_t* tx = reinterpret_cast<const uint64_t*>(a.blob + h.off_fb_lit);
    for (uint32_t k = 0; k < h.fb_lits; ++k) lit[k] = U128{(uint32_t)tx[k], (uint32_t)(tx[k] >> 32), (uint32_t)lit_meta[k], 0u};
    std::memset(a.out, 0xEE, (size_t)run);
    {
        // the wave-cooperative second pass (k_fb_splice): one emulated wave per sub-range, its LDS carve poisoned every time
        // (as the kernel deals them out: a wave takes every fourth sub-range of a chunk of 256)
        alignas(16) static uint8_t lds[kSpLdsPerWave];
        for (int64_t first = 0; first < n_lanes; first += 256) {
            for (int w = 3; w >= 0; --w) {
                std::memset(lds, 0xEE, sizeof lds);
                const int64_t left = n_lanes - (first + w);
                if (left <= 0) continue;
                const SpliceWork W{first + w, 4, (int)std::min<int64_t>(64, (left + 3) / 4), base.data() + first + w, 4};
                SpliceTables ST;
                ST.lit = lit.data();
                ST.esc = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc);
                ST.pool = a.blob + h.off_fb_pool;
                fb_splice_ranges<2>(a, ST, ca, W, lane_bytes, SpliceLds{lds});
            }
        }
    }
}

// Host emulation of the window kernel (k_stream_lpw): 64 lanes in lockstep over an
// emulated pair of LDS tiles, the same lane / mover code as the device.
template <bool kWide, bool kPair = false>
void run_lpw_t(const ScanArgs& a, int64_t lane_bytes, uint32_t& status) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    LpwView T;
    T.cls = a.blob + h.off_cls;
    T.ent = reinterpret_cast<const U128*>(a.blob + (kPair ? h.off_lpw2 : h.off_lpw));
    T.delay = h.lpw_delay;
    T.n_cls = h.n_cls;
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    const int64_t n_waves = (n_lanes + 63) / 64;
    std::vector<uint32_t> redo(n_waves * 64 + 1, 0);
    ScanArgs b = a;
    b.redo = redo.data();
    alignas(16) static uint8_t tin[kWtTile], tout[kWtOutTile];
    const int64_t vhi = (a.vend - 16) & ~(int64_t)15;
    for (int64_t wv = n_waves - 1; wv >= 0; --wv) {
        WtLane<kWide, kPair> L[64];
        WtMover M[64];
        const int64_t lane0 = wv * 64;
        for (int lid = 0; lid < 64; ++lid) L[lid].init(b, T, h.n_cls, lane0 + lid, lane_bytes);
        for (int lid = 0; lid < 64; ++lid) {
            for (int i = 0; i < 4; ++i) M[lid].set(lid, i, lane_bytes, L[WtMover::row_of(lid, i)].rv);
            for (int i = 0; i < 8; ++i) M[lid].set_out(lid, i, lane_bytes, L[WtMover::out_row_of(lid, i)].rv);
        }
        const int64_t wave_lo = lane0 * lane_bytes;
        const uint8_t* win_in = a.in_v0 + wave_lo;
        uint8_t* win_out = a.out_v0 + wave_lo;
        const int64_t room0 = vhi - wave_lo;
        auto ballot = [&]() { uint64_t m = 0; for (int lid = 0; lid < 64; ++lid) if (L[lid].active) m |= 1ull << lid; return m; };
        auto fetch = [&](int32_t k64) {
            for (int lid = 0; lid < 64; ++lid)
                for (int i = 0; i < 4; ++i) std::memcpy(tin + i * 1024 + lid * 16, win_in + M[lid].load_off(i, k64, room0), 16);
        };
        if (!ballot()) continue;
        std::memset(tout, 0xEE, sizeof(tout));
        fetch(0);
        uint64_t rows1 = 0, rows2 = 0;
        for (int32_t k64 = 0;; k64 += kWtPiece) {
            for (int lid = 0; lid < 64; ++lid) L[lid].check(b, lane0 + lid);
            const uint64_t rows = ballot();
            if (!rows && !rows1 && !rows2) break;
            U128 blk[64][4];
            int md[64];
            for (int lid = 0; lid < 64; ++lid) {
                const WtRow irow{tin + lid * kWtPiece, (uint32_t)((lid >> 1) & 3) << 4};
                for (int q = 0; q < 4; ++q) blk[lid][q] = irow.load(q);
            }
            fetch(k64 + kWtPiece);
            for (int lid = 0; lid < 64; ++lid) {
                const WtOutRow orow{tout + lid * kWtOutRow, (uint32_t)(lid & 7) << 4};
                md[lid] = L[lid].mode(k64);
                if (L[lid].active) L[lid].front(T, md[lid], blk[lid][0], blk[lid][1].x, orow, a.out_v0);
            }
            if ((k64 & (kWtOutRow - 1)) == 0) {
                for (int lid = 0; lid < 64; ++lid)
                    for (int i = 0; i < 8; ++i)
                        if (WtMover::stores(i, k64, rows, rows1, rows2, lid, L[WtMover::out_row_of(lid, i)].rfs))
                            std::memcpy(win_out + M[lid].store_off(i, k64), tout + i * 1024 + lid * 16, 16);
            }
            for (int lid = 0; lid < 64; ++lid) {
                const WtOutRow orow{tout + lid * kWtOutRow, (uint32_t)(lid & 7) << 4};
                if (L[lid].active) L[lid].back(T, md[lid], blk[lid][1], blk[lid][2], blk[lid][3], orow, a.out_v0);
            }
            rows2 = rows1;
            rows1 = rows;
        }
        for (int lid = 0; lid < 64; ++lid) {
            if (L[lid].seen & kLpwNul) status |= kStNul;
            if (L[lid].seen & kLpwDiv) status |= kStDiverge;
        }
    }
    // second launch: the lanes that touch an end of the input
    const StreamView TS = direct_view(a);
    alignas(16) uint8_t ring[kRingStride];
    const int64_t sub = lane_bytes / 64;
    for (int64_t k = 0; k < (int64_t)redo[0] * sub; ++k) {
        DirectLane L;
        stream_direct_lane<0>(a, TS, h.n_cls, (int64_t)redo[1 + k / sub] * sub + k % sub, 64, ring, 0, L, status);
    }
}

// pair: the pair form of the window entries (what the runtime launches when the tables have it)
void run_lpw(const ScanArgs& a, int64_t lane_bytes, uint32_t& status, bool pair = false) {
    if (pair) run_lpw_t<false, true>(a, lane_bytes, status);
    else if (reinterpret_cast<const StreamBlobHeader*>(a.blob)->lpw_delay > 3) run_lpw_t<true>(a, lane_bytes, status);
    else run_lpw_t<false>(a, lane_bytes, status);
}

void run_bytemap(const ScanArgs& a, uint32_t& status) {
    const DftBlobHeader& h = *reinterpret_cast<const DftBlobHeader*>(a.blob);
    const uint8_t* map = a.blob + h.off_bytemap;
    const bool aligned = (reinterpret_cast<uintptr_t>(a.out_v0) & 15u) == 0;
    const int64_t vfirst = a.vbeg & ~(int64_t)15;
    const int64_t nvec = (a.vend - vfirst + 15) / 16;
    uint32_t zero = 0;
    for (int64_t k = 0; k < nvec; ++k) {
        U128 w = *reinterpret_cast<const U128*>(a.in_v0 + vfirst + k * 16);
        bytemap_vec(a, map, w, vfirst + k * 16, aligned, zero);
    }
    if (zero) status |= kStNul;
}

template <class G, class Engine>
int run_family(int family, ScanArgs& a, uint32_t& status, uint64_t& total) {
    const int64_t n_chunks = (a.vend + G::CHUNK - 1) / G::CHUNK;
    if (family == 2) { run_lp<G, Engine>(a, n_chunks, status); total = (uint64_t)(a.vend - a.vbeg); }
    else run_gen<G, Engine>(a, n_chunks, status, total);
    return 0;
}

}  // namespace

extern "C" {

// family: 1 bytemap, 2 tile LP, 3 tile general (4 / 5, the LDS-tile walkers of the stream tables, went in round 6).
// 20 / 21 stream LP by the emit pass alone (16-byte / 8-byte entries),
// 8 positional-window stream LP, 9 direct stream general on the 8-byte entries (7 prefers the 16-byte ones), 6 direct stream LP, 7 direct stream general (geo: 0 -> 2048-byte lanes, 1 -> 48-byte lanes).
// geo: 0 production, 1 tiny.
// in_mis/out_mis: address misalignment (0..15) to give the staged buffers.
// want_scratch: pass a mask scratch to the NFT long-line path.
int shim_scan(const uint8_t* blob, int engine, int mask_bytes, int family, int geo, const uint8_t* in, size_t n,
              int in_mis, uint8_t* out, size_t cap, int out_mis, int want_scratch, size_t* m, uint32_t* status_out) {
    if (n == 0) { *m = 0; *status_out = 0; return 0; }
    std::vector<uint8_t> ibuf(n + 64, 0xAA), obuf(cap + 64, 0xEE);
    uint8_t* ia = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ibuf.data()) + 15) & ~(uintptr_t)15) + in_mis;
    uint8_t* oa = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(obuf.data()) + 15) & ~(uintptr_t)15) + out_mis;
    std::memcpy(ia, in, n);
    std::vector<uint8_t> scratch(want_scratch ? (n + 64) * 8 : 0);
    ScanArgs a{};
    const int64_t al = (int64_t)(reinterpret_cast<uintptr_t>(ia) & 15u);
    a.in_v0 = ia - al;
    a.out_v0 = oa - al;
    a.out = oa;
    a.vbeg = al;
    a.vend = al + (int64_t)n;
    a.blob = blob;
    a.cap = cap;
    a.gscratch = want_scratch ? scratch.data() : nullptr;
    uint32_t status = 0;
    uint64_t total = 0;
    if (family != 3 && family != 7 && family != 9 && family != 22 && family != 23 && family != 24 && family != 34 && family != 35 && family != 36 && family != 37 && family != 38 && cap < n) return -9;
    if (family == 1) { run_bytemap(a, status); total = n; }
    else if (family == 8) {
        if (reinterpret_cast<const StreamBlobHeader*>(blob)->lpw_bytes == 0) return -5;
        // like the runtime: buffers that are not congruent mod 16 go to the direct walker
        if (reinterpret_cast<uintptr_t>(a.out_v0) & 15u) run_direct_lp<>(a, geo == 0 ? 2048 : 48, status);
        else run_lpw(a, geo == 0 ? 2048 : 64, status);
        total = n;
    }
    else if (family == 26) {                          // the window kernel on the pair form of its entries
        if (reinterpret_cast<const StreamBlobHeader*>(blob)->lpw2_bytes == 0) return -5;
        if (reinterpret_cast<uintptr_t>(a.out_v0) & 15u) run_direct_lp<>(a, geo == 0 ? 2048 : 48, status);
        else run_lpw(a, geo == 0 ? 2048 : 64, status, true);
        total = n;
    }
    else if (family == 6) { run_direct_lp<>(a, geo == 0 ? 2048 : 48, status); total = n; }
    else if (family == 20 || family == 21) {          // stream LP as the runtime launches it without a window form: emit pass alone
        const bool g16 = family == 20 && reinterpret_cast<const StreamBlobHeader*>(blob)->g16_bytes != 0;
        run_direct_lp_emit<>(a, geo == 0 ? 2048 : 64, status, g16);
        total = n;
    }
    else if (family == 22 || family == 23) {          // stream general on the fallback form of a large table (23: count pass only)
        if (reinterpret_cast<const StreamBlobHeader*>(blob)->fb_slots == 0) return -5;
        run_fb_gen(a, geo == 0 ? 2048 : 48, status, total, family == 23);
    }
    else if (family == 7 || family == 9) {
        const bool g16 = family == 7 && reinterpret_cast<const StreamBlobHeader*>(blob)->g16_bytes != 0;
        run_direct_gen<>(a, geo == 0 ? 2048 : 48, status, total, g16);
    }
    else if (family == 32 || family == 33) {          // ... with exact sub-ranges (33: a look-back of 4 bytes: wrong guesses, repair rounds)
        if (reinterpret_cast<const StreamBlobHeader*>(blob)->g16_bytes == 0) return -5;
        int rounds = 0;
        run_direct_gen_exact<>(a, geo == 0 ? 2048 : 64, status, total, family == 33 ? 4u : (uint32_t)kSpecLook, geo == 0 ? 256 : 3, rounds);
        g_last_rounds = rounds;
    }
    else if (family == 37 || family == 38) {          // a memoryless program in one pass (map_block.hpp); 38: tiles of 3 threads, windows of 48 bytes
        if (reinterpret_cast<const StreamBlobHeader*>(blob)->mg_max == 0) return -5;
        if (family == 37) run_mapgen(a, status, total, geo == 0 ? 40960u : 256u, geo == 0 ? kMapGenThreads / kMgLanes : 2, geo == 0 ? kMgLanes : 3);
        else run_mapgen(a, status, total, 48u, 1, 1);
    }
    else if (family == 34 || family == 35 || family == 36) {   // ... in ONE walk (one_block.hpp); 35: a look-back of 4 bytes and tiles of 3 lanes (wrong guesses: repair
        if (reinterpret_cast<const StreamBlobHeader*>(blob)->g16_bytes == 0) return -5;     // rounds, void tiles); 36: regions of 76 bytes for 64 of input
        int rounds = 0;
        if (family == 34) run_direct_gen_one<>(a, geo == 0 ? 128u : 64u, geo == 0 ? 172u : 100u, 32u, geo == 0 ? 256 : 5, status, total, rounds);
        else if (family == 35) run_direct_gen_one<>(a, 64u, 140u, 4u, 3, status, total, rounds);
        else run_direct_gen_one<>(a, 64u, 76u, 32u, 4, status, total, rounds);
        g_last_rounds = rounds;
    }
    else if (family == 27) {
        // ... its second pass by the wave-cooperative splice (what the runtime launches by default)
        const StreamBlobHeader& sh = *reinterpret_cast<const StreamBlobHeader*>(blob);
        if (!sh.fb_slots || !sh.off_fb_lit_meta) return -5;
        run_fb_copy(a, geo == 0 ? 2048 : 128, status, total, geo == 0 ? 256u : 64u);
    }
    else if (family == 29) {
        // ... with the first pass on the 8-byte comb (round 3's mark pass; what tables without the mark form run)
        const StreamBlobHeader& sh = *reinterpret_cast<const StreamBlobHeader*>(blob);
        if (!sh.fb_slots || !sh.off_fb_lit_meta) return -5;
        run_fb_copy(a, geo == 0 ? 2048 : 128, status, total, geo == 0 ? 256u : 64u, true);
    }
    else if (engine == 1) {
        if (geo == 0) run_family<GeoDft, DftEngine>(family, a, status, total);
        else run_family<GeoTiny, DftEngine>(family, a, status, total);
    } else if (mask_bytes == 1) {
        if (geo == 0) run_family<GeoNft8, NftEngine<uint8_t>>(family, a, status, total);
        else run_family<GeoTiny, NftEngine<uint8_t>>(family, a, status, total);
    } else if (mask_bytes == 2) {
        if (geo == 0) run_family<GeoNft16, NftEngine<uint16_t>>(family, a, status, total);
        else run_family<GeoTiny, NftEngine<uint16_t>>(family, a, status, total);
    } else if (mask_bytes == 4) {
        if (geo == 0) run_family<GeoNft32, NftEngine<uint32_t>>(family, a, status, total);
        else run_family<GeoTiny, NftEngine<uint32_t>>(family, a, status, total);
    } else {
        if (geo == 0) run_family<GeoNft64, NftEngine<uint64_t>>(family, a, status, total);
        else run_family<GeoTiny, NftEngine<uint64_t>>(family, a, status, total);
    }
    *status_out = status;
    *m = (size_t)total;
    if (total <= cap) std::memcpy(out, oa, (size_t)total);
    return 0;
}

// Guided families (backward DFA sweep + forward transducer over its symbols).
// family: 10 length-preserving (emit pass alone; 14: on the 8-byte entries; 13: the LDS-ring walker),
// 11 general on the 16-byte entries when the tables have them, 12 general on the 8-byte entries.  geo: 0 -> 2048-byte lanes, 1 -> 64-byte lanes.
int shim_scan_guided(const uint8_t* rblob, const uint8_t* gblob, int family, int geo, const uint8_t* in, size_t n, int in_mis,
                     uint8_t* out, size_t cap, int out_mis, size_t* m, uint32_t* status_out) {
    if (n == 0) { *m = 0; *status_out = 0; return 0; }
    std::vector<uint8_t> ibuf(n + 64, 0xAA), obuf(cap + 64, 0xEE);
    uint8_t* ia = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ibuf.data()) + 15) & ~(uintptr_t)15) + in_mis;
    uint8_t* oa = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(obuf.data()) + 15) & ~(uintptr_t)15) + out_mis;
    std::memcpy(ia, in, n);
    ScanArgs a{};
    const int64_t al = (int64_t)(reinterpret_cast<uintptr_t>(ia) & 15u);
    a.in_v0 = ia - al;
    a.out_v0 = oa - al;
    a.out = oa;
    a.vbeg = al;
    a.vend = al + (int64_t)n;
    a.blob = gblob;
    a.rblob = rblob;
    a.cap = cap;
    std::vector<uint8_t> sym(2 * n + 1024, 0xDD);   // poison: every symbol the forward pass walks must have been written
    a.sym_v0 = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(sym.data()) + 63) & ~(uintptr_t)63);
    uint32_t status = 0;
    uint64_t total = 0;
    const int64_t lane_bytes = geo == 0 ? 2048 : 128;
    if ((family == 10 || family == 13 || family == 14) && cap < n) return -9;
    // like the runtime: symbols are packed two per byte when the backward DFA allows it and the walk uses the 16-byte entries
    const bool has_g16 = reinterpret_cast<const StreamBlobHeader*>(gblob)->g16_bytes != 0;
    const bool packed = reinterpret_cast<const RevBlobHeader*>(rblob)->sym_bits == 4 && has_g16 &&
                        (family == 10 || family == 11 || family == 17 || family == 18 || family == 40 || family == 41);
    if (reinterpret_cast<const RevBlobHeader*>(rblob)->sym_bits == 16) {
        // wide guided tables (more than 256 backward states): k_rev_wide, k_wide_fwd<count>, scan, k_wide_fwd<emit>
        if (family == 10 || family == 13 || family == 14) return -5;
        const RevBlobHeader& rh = *reinterpret_cast<const RevBlobHeader*>(rblob);
        const RevWideView RT{reinterpret_cast<const uint16_t*>(rblob + rh.off_wide)};
        const int64_t max_look = lane_bytes <= 128 ? 256 : kRevMaxLook;
        const int64_t vtop = (a.vend + 63) & ~(int64_t)63;
        for (int64_t lane = 0; lane < (vtop + lane_bytes - 1) / lane_bytes; ++lane) rev_wide_lane(a, RT, lane, lane_bytes, max_look);
        const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(gblob);
        const StreamView T = direct_view(a);
        const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
        std::vector<uint64_t> cnt(n_lanes), base(n_lanes);
        for (int64_t lane = 0; lane < n_lanes; ++lane) { DirectLane L; wide_fwd_lane<1>(a, T, h.n_cls, lane, lane_bytes, 0, L, status); cnt[lane] = L.count; }
        uint64_t run = 0;
        for (int64_t lane = 0; lane < n_lanes; ++lane) { base[lane] = run; run += cnt[lane]; }
        total = run;
        if (run > cap) { status |= kStCapacity; *status_out = status; *m = (size_t)total; return 0; }
        for (int64_t lane = n_lanes - 1; lane >= 0; --lane) { DirectLane L; wide_fwd_lane<2>(a, T, h.n_cls, lane, lane_bytes, base[lane], L, status); }
        *status_out = status;
        *m = (size_t)total;
        std::memcpy(out, oa, (size_t)total);
        return 0;
    }
    int rev_rounds = 0;
    if (family == 17 || family == 18 || family == 40 || family == 41) {
        rev_rounds = run_rev_sweep_exact(a, lane_bytes, packed, family == 18 || family == 41 ? 4u : (uint32_t)kSpecLook);
        if (rev_rounds < 0) { *status_out = 1u << 29; *m = 0; return 0; }
    } else {
        run_rev_sweep(a, lane_bytes, packed);
    }
    if (family == 10 || family == 13 || family == 14) {
        if (family == 13) run_direct_lp<1>(a, lane_bytes, status);                  // the LDS-ring walker (A/B variant)
        else if (packed) run_direct_lp_emit<2>(a, lane_bytes, status, true);
        else run_direct_lp_emit<1>(a, lane_bytes, status, family == 10 && has_g16);  // 14: on the 8-byte entries
        total = n;
    }
    else if (family == 17 || family == 18) {
        // general guided family with exact sub-ranges (18: a look-back of 4 bytes: wrong guesses, repair rounds)
        if (!has_g16) return -5;
        int rounds = 0;
        if (packed) run_direct_gen_exact<2>(a, lane_bytes, status, total, family == 18 ? 4u : (uint32_t)kSpecLook, geo == 0 ? 256 : 3, rounds);
        else run_direct_gen_exact<1>(a, lane_bytes, status, total, family == 18 ? 4u : (uint32_t)kSpecLook, geo == 0 ? 256 : 3, rounds);
        g_last_rounds = rounds + rev_rounds;
    }
    else if (family == 40 || family == 41) {
        // the general guided family in ONE forward walk behind the backward pass (one_block.hpp); 41: look-backs of 4 bytes, tiles of 3 lanes
        if (!has_g16) return -5;
        int rounds = 0;
        const uint32_t S = family == 41 ? 64u : (geo == 0 ? 128u : 64u), R = family == 41 ? 140u : (geo == 0 ? 172u : 100u);
        const int nl = family == 41 ? 3 : (geo == 0 ? 256 : 5);
        if (packed) run_direct_gen_one<2>(a, S, R, family == 41 ? 4u : 32u, nl, status, total, rounds);
        else run_direct_gen_one<1>(a, S, R, family == 41 ? 4u : 32u, nl, status, total, rounds);
        g_last_rounds = rounds + rev_rounds;
    }
    else if (packed) run_direct_gen<2>(a, lane_bytes, status, total, true);
    else run_direct_gen<1>(a, lane_bytes, status, total, family == 11 && has_g16);
    *status_out = status;
    *m = (size_t)total;
    if (total <= cap) std::memcpy(out, oa, (size_t)total);
    return 0;
}

// The backward pass alone (k_rev_sweep's per-thread body): one symbol per input byte into sym_out[0, n).  Used for the
// guided families' symbols and for the viability symbols of the generator modes (generate.cpp).
// Generator modes as the runtime runs them on the device: the backward sweep (viability symbols), then the enumeration
// kernel's lane body — count, exclusive sum, emit (gen_block.hpp).  status: kStDiverge / kStEditOverflow mean "this input goes
// to the host enumeration".  frames / path_cap: a lane's stack and path buffer (small in the tests: the overflow route runs too).
int shim_generate(const uint8_t* rblob, const uint8_t* nblob, int geo, const uint8_t* in, size_t n, int in_mis, uint8_t* out, size_t cap,
                  uint32_t frames, uint32_t path_cap, size_t* m, uint32_t* status_out) {
    *m = 0; *status_out = 0;
    if (n == 0) return 0;
    std::vector<uint8_t> ibuf(n + 64, 0xAA), sym(n + 1024, 0xEE);
    uint8_t* ia = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ibuf.data()) + 15) & ~(uintptr_t)15) + in_mis;
    std::memcpy(ia, in, n);
    ScanArgs a{};
    const int64_t al = (int64_t)(reinterpret_cast<uintptr_t>(ia) & 15u);
    a.in_v0 = ia - al;
    a.vbeg = al;
    a.vend = al + (int64_t)n;
    a.rblob = rblob;
    a.sym_v0 = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(sym.data()) + 15) & ~(uintptr_t)15);
    uint32_t status = 0;
    a.status = &status;
    run_rev_sweep(a, geo == 0 ? 2048 : 128, false);
    a.blob = nblob;
    a.out = out;
    a.cap = cap;
    const GenView G = gen_view(nblob);
    const int64_t lane_bytes = geo == 0 ? 512 : 64;
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    std::vector<uint32_t> stack((size_t)n_lanes * frames * 4 + 4, 0xEEEEEEEEu);
    std::vector<uint8_t> path((size_t)n_lanes * path_cap + 4, 0xEE);
    GenArgs ga{stack.data(), path.data(), frames, path_cap};
    std::vector<uint64_t> cnt(n_lanes), base(n_lanes);
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) { DirectLane L; gen_lane<1>(a, G, ga, lane, lane_bytes, 0, L, status); cnt[lane] = L.count; }
    *status_out = status;
    if (status & (kStDiverge | kStEditOverflow)) return 0;
    uint64_t run = 0;
    for (int64_t lane = 0; lane < n_lanes; ++lane) { base[lane] = run; run += cnt[lane]; }
    *m = (size_t)run;
    if (run > cap) { *status_out = status | kStCapacity; return 0; }
    for (int64_t lane = 0; lane < n_lanes; ++lane) {
        DirectLane L;
        gen_lane<2>(a, G, ga, lane, lane_bytes, base[lane], L, status);
        if (L.count != cnt[lane]) status |= 1u << 30;                     // count and emit passes disagree
    }
    *status_out = status;
    return 0;
}

// The backtracking fallback as the runtime runs it: count per sub-range, exclusive sum, emit (gen_block.hpp: bt_lane), with a pool
// of `pool` stacks the sub-ranges take in turn.  status: kStDiverge (with *m = the output up to the attempt that does not return),
// kStEditOverflow (the limits).
int shim_backtrack(const uint8_t* nblob, int geo, const uint8_t* in, size_t n, int in_mis, uint8_t* out, size_t cap, uint32_t frames,
                   uint32_t path_cap, uint32_t budget, size_t* m, uint32_t* status_out) {
    *m = 0; *status_out = 0;
    if (n == 0) return 0;
    std::vector<uint8_t> ibuf(n + 64, 0xAA);
    uint8_t* ia = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ibuf.data()) + 15) & ~(uintptr_t)15) + in_mis;
    std::memcpy(ia, in, n);
    ScanArgs a{};
    const int64_t al = (int64_t)(reinterpret_cast<uintptr_t>(ia) & 15u);
    a.in_v0 = ia - al;
    a.vbeg = al;
    a.vend = al + (int64_t)n;
    uint32_t status = 0;
    a.status = &status;
    a.blob = nblob;
    a.out = out;
    a.cap = cap;
    const GenView G = gen_view(nblob);
    const int64_t lane_bytes = geo == 0 ? 1024 : 64;
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    const int64_t pool = 3;
    std::vector<uint32_t> stack((size_t)pool * frames * 4 + 4, 0xEEEEEEEEu);
    std::vector<uint8_t> path((size_t)pool * path_cap + 4, 0xEE);
    GenArgs ga{stack.data(), path.data(), frames, path_cap};
    std::vector<uint64_t> cnt(n_lanes), base(n_lanes);
    int64_t first_div = -1;
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) {
        DirectLane L;
        uint32_t lst = 0;
        uint32_t why = 0;
        bt_lane<1>(a, G, ga, lane % pool, lane, lane_bytes, 0, budget, L, lst, why);
        cnt[lane] = L.count;
        if (lst & kStDiverge) first_div = lane;
        status |= lst;
    }
    *status_out = status;
    if (status & kStEditOverflow) return 0;
    uint64_t run = 0;
    for (int64_t lane = 0; lane < n_lanes; ++lane) { base[lane] = run; run += cnt[lane]; }
    if (first_div >= 0) run = base[first_div] + cnt[first_div];
    *m = (size_t)run;
    if (run > cap) { *status_out = status | kStCapacity; return 0; }
    for (int64_t lane = 0; lane < n_lanes && (first_div < 0 || lane <= first_div); ++lane) {
        DirectLane L;
        uint32_t lst = 0;
        uint32_t why = 0;
        bt_lane<2>(a, G, ga, lane % pool, lane, lane_bytes, base[lane], budget, L, lst, why);
        if (L.count != cnt[lane]) status |= 1u << 30;                     // count and emit passes disagree
    }
    *status_out = status;
    return 0;
}

// One round of the lazy family as the runtime runs it (lazy_block.hpp; runtime.cpp: lazy_round): the count pass for the lanes that have no
// result yet (lane_counts[lane] == kLazyVoid), and — when no lane met an unexplored edge — the exclusive sum and the emit pass.  The caller
// (tests/shim_lib.py) has the library explore the listed misses and comes back with the grown tables.  ent: the caller's copy (lanes mark
// the misses they list).
int shim_lazy_round(const uint8_t* cls, uint64_t* ent, const uint8_t* pool, uint32_t n_cls, int geo, const uint8_t* in, size_t n, int in_mis,
                    uint32_t* lane_counts, uint32_t* miss, uint32_t miss_cap, uint64_t budget, uint8_t* out, size_t cap, size_t* m, uint32_t* status_out,
                    size_t ent_words, int foreign_marks) {
    *m = 0; *status_out = 0;
    miss[0] = 0;
    if (n == 0) return 0;
    // foreign_marks: every unexplored edge carries the mark of ANOTHER launch (a chunk in flight on the same table that listed it in ITS
    // miss list, runtime.cpp: the host path's slots) — this launch must list the edges it needs all the same
    if (foreign_marks)
        for (size_t k = 0; k < ent_words; ++k)
            if (ent[k] == kLazyUnexplored) ent[k] = kLazyNoted | (uint64_t)0xdead << 32;
    std::vector<uint8_t> ibuf(n + 64, 0xAA);
    uint8_t* ia = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ibuf.data()) + 15) & ~(uintptr_t)15) + in_mis;
    std::memcpy(ia, in, n);
    ScanArgs a{};
    const int64_t al = (int64_t)(reinterpret_cast<uintptr_t>(ia) & 15u);
    a.in_v0 = ia - al;
    a.vbeg = al;
    a.vend = al + (int64_t)n;
    uint32_t status = 0;
    a.status = &status;
    a.out = out;
    a.cap = cap;
    LazyArgs la{cls, ent, pool, n_cls, miss, miss_cap, budget};
    la.gen = 1;
    const int64_t lane_bytes = geo == 0 ? 1024 : 64;
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    for (int64_t lane = n_lanes - 1; lane >= 0; --lane) {
        if (lane_counts[lane] != kLazyVoid) continue;
        DirectLane L;
        uint32_t lst = 0;
        bool voided = false;
        lazy_lane<1>(a, la, lane, lane_bytes, 0, L, lst, voided);
        status |= lst;
        if (!voided && !(lst & (kStEditOverflow | kStDiverge))) lane_counts[lane] = (uint32_t)L.count;
    }
    *status_out = status;
    if (status & (kStMiss | kStEditOverflow | kStDiverge)) return 0;
    uint64_t run = 0;
    std::vector<uint64_t> base(n_lanes);
    for (int64_t lane = 0; lane < n_lanes; ++lane) { base[lane] = run; run += lane_counts[lane]; }
    *m = (size_t)run;
    if (run > cap) { *status_out = status | kStCapacity; return 0; }
    for (int64_t lane = 0; lane < n_lanes; ++lane) {
        DirectLane L;
        uint32_t lst = 0;
        bool voided = false;
        lazy_lane<2>(a, la, lane, lane_bytes, base[lane], L, lst, voided, lane_counts[lane]);
        if (L.count != lane_counts[lane] || voided) status |= 1u << 30;    // count and emit passes disagree
    }
    *status_out = status;
    return 0;
}

// The stack guard as the runtime drives it (guard_block.hpp): probe the windows, the runs of flagged windows, the reference's
// search on the lines that cover them; the first line that overflows again with its output.  *hit: 0 none, 1 a line overflowed
// (*line_start, out[0, *part) = what the reference had printed of it), 2 a line was not decided.
int shim_guard(const uint8_t* kblob, const uint8_t* in, size_t n, int in_mis, uint64_t budget, int* hit, uint64_t* line_start, uint8_t* out, size_t cap,
               size_t* part) {
    *hit = 0; *line_start = 0; *part = 0;
    if (n == 0) return 0;
    const GuardBlobHeader& h = *reinterpret_cast<const GuardBlobHeader*>(kblob);
    std::vector<uint8_t> ibuf(n + 64, 0xAA);
    uint8_t* ia = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ibuf.data()) + 15) & ~(uintptr_t)15) + in_mis;
    std::memcpy(ia, in, n);
    ScanArgs a{};
    const int64_t al = (int64_t)(reinterpret_cast<uintptr_t>(ia) & 15u);
    a.in_v0 = ia - al;
    a.vbeg = al;
    a.vend = al + (int64_t)n;
    const int64_t n_win = ((int64_t)n + h.window - 1) / h.window;
    std::vector<GuardRun> runs;
    for (int64_t w = 0; w < n_win;) {
        if (!guard_probe(a, h.bset, a.vbeg + w * h.window, a.vbeg + (w + 1) * h.window)) { ++w; continue; }
        int64_t e = w;
        while (e + 1 < n_win && guard_probe(a, h.bset, a.vbeg + (e + 1) * h.window, a.vbeg + (e + 2) * h.window)) ++e;
        runs.push_back(GuardRun{(uint32_t)w, (uint32_t)e});
        w = e + 1;
    }
    if (runs.empty()) return 0;
    std::vector<GuardResult> res(runs.size());
    std::vector<uint32_t> stack((size_t)kGuardStackMax * 3 * 2);
    GuardArgs ga{};
    ga.blob = kblob;
    ga.runs = runs.data();
    ga.results = res.data();
    ga.stack = stack.data();
    ga.budget = budget;
    ga.obuf = nullptr;
    ga.obuf_cap = 0xffffffffu;
    for (size_t r = 0; r < runs.size(); ++r) guard_line<false>(a, ga, (int64_t)(r & 1), (int64_t)r);
    size_t bad = runs.size();
    for (size_t r = 0; r < runs.size(); ++r) {
        if (res[r].status == 1u) { bad = r; break; }
        if (res[r].status == 2u) *hit = 2;
    }
    if (bad == runs.size()) return 0;
    *hit = 1;
    *line_start = res[bad].line_start;
    std::vector<uint8_t> obuf(16 * ((size_t)res[bad].out_len + 1) + 65536);
    ga.runs = runs.data() + bad;
    ga.results = res.data() + bad;
    ga.obuf = obuf.data();
    ga.obuf_cap = (uint32_t)obuf.size();
    ga.out = out;
    ga.out_cap = cap;
    guard_line<true>(a, ga, 0, 0);
    if (res[bad].status != 1u) return 3;
    *part = res[bad].out_len;
    return 0;
}

int shim_last_rounds() { return g_last_rounds; }

int shim_rev_sweep(const uint8_t* rblob, int geo, const uint8_t* in, size_t n, int in_mis, uint8_t* sym_out) {
    if (n == 0) return 0;
    std::vector<uint8_t> ibuf(n + 64, 0xAA);
    uint8_t* ia = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ibuf.data()) + 15) & ~(uintptr_t)15) + in_mis;
    std::memcpy(ia, in, n);
    ScanArgs a{};
    const int64_t al = (int64_t)(reinterpret_cast<uintptr_t>(ia) & 15u);
    a.in_v0 = ia - al;
    a.vbeg = al;
    a.vend = al + (int64_t)n;
    a.rblob = rblob;
    std::vector<uint8_t> sym(n + 512, 0xDD);
    a.sym_v0 = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(sym.data()) + 63) & ~(uintptr_t)63);
    run_rev_sweep(a, geo == 0 ? 2048 : 128, false);
    std::memcpy(sym_out, a.sym_v0 + al, n);
    return 0;
}

}  // extern "C"
