// lazy_block.hpp — the deterministic engine on tables that are still being built (round 5; front.hpp: LazyDft).
//
// The reference determinises on the fly: infer_dft builds a dstate when the input first takes an edge to it
// (trre_dft.c:1135-1175), so it runs every pattern on every finite input — also '((a:x)*b)|((a:y)*c)', whose eager
// construction never ends (a state per run length), and '(a|b)*a(a|b){18}:x' with its 2^19 states.  Rounds 1-4 determinised
// eagerly and refused what did not fit (TRRE_E_TOO_BIG).  This family is the reference's scheme with the roles split: a
// lane walks the scan loop (trre_dft.c:1272-1286, 1110-1196) over the rows that exist, in HBM; an edge nobody has explored yet
// is listed {row, class} — the first lane that meets it marks the entry so that it is listed once — and the lane's result is
// void; the host builds the listed edges, uploads the new rows and the launch runs again for the void lanes.  A round without
// misses is the answer.  Count, exclusive sum, emit as in every general family; a thread per sub-range (the lines that start
// in it), byte loads — a last resort like the backtracking fallback of the other engine, not a fast path.
//
// The lane body is TRRE_HD: tests/cpu_shim.cpp runs it lane by lane on the host, with the library's explore() in between.
#pragma once
#include "scan_block.hpp"

namespace trre {

constexpr uint32_t kStMiss = 1u << 7;          // a lane met an unexplored edge: the launch is not final
constexpr uint32_t kLazyVoid = 0xffffffffu;    // lane_counts: this lane has no result yet
constexpr uint64_t kLazyUnexplored = 6u << 2, kLazyNoted = 5u << 2;   // (front.hpp: kEntUnexplored, kEntMissNoted)
#ifndef TRRE_LAZY_MISS_WORDS
#define TRRE_LAZY_MISS_WORDS 16
#endif
constexpr uint32_t kLazyRecWords = TRRE_LAZY_MISS_WORDS;            // (front.hpp: kLazyMissWords — this header does not include the front end)

struct LazyArgs {
    const uint8_t* cls;        // [256]
    uint64_t* ent;             // [n_rows][n_cls] (the device's copy: lanes mark the misses they list)
    const uint8_t* pool;
    uint32_t n_cls;
    uint32_t* miss;            // [0] = misses listed, [1] = unused, then records of kLazyRecWords (16) words: {row, class, n, 0, the n <= 48 bytes
                               // of the line behind the byte that missed} — the host builds the edge and follows the attempt along those bytes
                               // (trre_dft.c:1135-1175 goes on from a miss the same way), so that a deep walk costs a round per 49 states, not one each
    uint32_t miss_cap;
    uint64_t budget;           // table steps per sub-range
    // the kernel's copy in LDS: the classes and the first rows_l rows (the root row — every attempt starts there and most end there — and
    // whatever else fits; a row beyond is read from `ent`).  Stale entries are harmless: the tables only grow, an entry that reads as
    // unexplored here is looked at again in `ent` by lazy_note_miss's compare-and-swap.  (Host: not used.)
    uint32_t n_rows = 0;
    const uint8_t* cls_l = nullptr;
    const uint64_t* ent_l = nullptr;
    uint32_t rows_l = 0;
    uint32_t gen = 0;          // this launch's id: a mark is kLazyNoted | gen << 32.  Several scan contexts share one table (the host path's
                               // chunks in flight): a mark left by ANOTHER launch says nothing about this launch's list, so the edge is
                               // listed again (without the id a chunk could wait for ever on an edge only a later chunk had listed)
};

constexpr uint32_t kLazyFollow = 48;     // (kLazyMissWords: front.hpp / below)
template <class ByteAt>
TRRE_HD void lazy_note_miss(const LazyArgs& la, uint32_t row, uint32_t k, int64_t behind, ByteAt byte_at) {
    uint64_t* e = la.ent + (uint64_t)row * la.n_cls + k;
    const uint64_t mine = kLazyNoted | (uint64_t)la.gen << 32;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long cur = __hip_atomic_load(reinterpret_cast<unsigned long long*>(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (cur == mine) return;                                               // listed by this launch already
        if (cur != kLazyUnexplored && (uint32_t)cur != (uint32_t)kLazyNoted) return;   // built meanwhile
        const unsigned long long was = atomicCAS(reinterpret_cast<unsigned long long*>(e), cur, (unsigned long long)mine);
        if (was == cur) break;
        cur = was;
    }
    const uint32_t at = atomicAdd(la.miss, 1u);
#else
    if (*e == mine || (*e != kLazyUnexplored && (uint32_t)*e != (uint32_t)kLazyNoted)) return;
    *e = mine;
    const uint32_t at = la.miss[0]++;
#endif
    if (at < la.miss_cap) {
        uint32_t* r = la.miss + 2 + (size_t)kLazyRecWords * at;
        uint32_t n = 0, w = 0;
        for (; n < kLazyFollow; ++n) {
            const uint8_t c = byte_at(behind + n);
            if (c == (uint8_t)'\n' || c == 0) break;
            w |= (uint32_t)c << (8u * (n & 3u));
            if ((n & 3u) == 3u) { r[4 + (n >> 2)] = w; w = 0; }
        }
        if (n & 3u) r[4 + (n >> 2)] = w;
        r[0] = row; r[1] = k; r[2] = n; r[3] = 0;
        return;
    }
    // the list is full: the edge stays unexplored for the next round
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_store(e, kLazyUnexplored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *e = kLazyUnexplored;
#endif
}

// kMode 1: count (L.count; void: kLazyVoid is the caller's business — `voided` says so); 2: emit at a.out + out_base, lane_total bytes
template <int kMode>
TRRE_HD void lazy_lane(const ScanArgs& a, const LazyArgs& la, int64_t lane, int64_t lane_bytes, uint64_t out_base, DirectLane& L, uint32_t& status,
                       bool& voided, uint64_t lane_total = 0) {
    const int64_t lo = lane * lane_bytes;
    int64_t hi = lo + lane_bytes;
    if (hi > a.vend) hi = a.vend;
    uint64_t cnt = 0, steps = 0;
    L.count = 0;
    voided = false;
    if (lo >= hi) return;
    uint8_t* const op = kMode == 2 ? a.out + out_base : nullptr;
    // the emit pass's output: up to 8 bytes — the positions [cnt - on, cnt) — wait in a register and leave as one store (a lane's region is its
    // own up to lane_total; what a failing attempt has stored is written over)
    uint64_t ow = 0;
    uint32_t on = 0;
    // n <= 8 bytes (the low ones of `bytes`, zero above them) behind the output so far
    auto append = [&](uint64_t bytes, uint32_t n) {
        cnt += n;
        if (kMode == 2) {
            ow |= bytes << (8u * on);
            const uint32_t tot = on + n;
            if (tot >= 8u) {                                // the positions [cnt - tot, cnt - tot + 8) leave
                const uint64_t at = cnt - tot;
                if (at + 8 <= lane_total) __builtin_memcpy(op + at, &ow, 8);
                else for (uint32_t b = 0; b < 8u; ++b) if (at + b < lane_total) op[at + b] = (uint8_t)(ow >> (8u * b));
                ow = on ? bytes >> (8u * (8u - on)) : 0;
                on = tot - 8u;
            } else {
                on = tot;
            }
        }
    };
    auto put1 = [&](uint8_t c) { append((uint64_t)c, 1u); };
    auto flush_out = [&]() {
        if (kMode == 2) {
            for (uint32_t b = 0; b < on; ++b) if (cnt - on + b < lane_total) op[cnt - on + b] = (uint8_t)(ow >> (8u * b));
            ow = 0; on = 0;
        }
    };
    auto rewind_out = [&](uint64_t to) {                    // a failed attempt: back to `to` (<= cnt); bytes below `to` that have left stay
        if (kMode == 2) {
            const uint64_t wstart = cnt - on;
            on = to >= wstart ? (uint32_t)(to - wstart) : 0u;
            ow = on ? ow & (~0ull >> (64u - 8u * on)) : 0;
        }
        cnt = to;
    };
    // a record's content ends at its '\n', at a NUL before it (Q2) or at the last byte of the input, which ends its record
    // whatever it is (Q1).  The input through an 8-byte window (aligned: the block that holds a byte of the input is readable).
    uint64_t iw = 0;
    int64_t iwb = -8;
    auto byte_at = [&](int64_t v) -> uint8_t {
        if (v >= a.vend - 1) return (uint8_t)'\n';
        const int64_t b = v & ~(int64_t)7;
        if (b != iwb) { iw = *reinterpret_cast<const uint64_t*>(a.in_v0 + b); iwb = b; }
        return (uint8_t)(iw >> (8u * (uint32_t)(v & 7)));
    };
    auto entry = [&](uint32_t row, uint8_t c, uint32_t& k) -> uint64_t {
#if defined(__HIP_DEVICE_COMPILE__)
        k = la.cls_l[c];
        if (row < la.rows_l) return la.ent_l[row * la.n_cls + k];
        return __hip_atomic_load(la.ent + (uint64_t)row * la.n_cls + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        k = la.cls[c];
        return la.ent[(uint64_t)row * la.n_cls + k];
#endif
    };
    auto out_len = [&](uint64_t e) -> uint32_t {
        const uint32_t il = ent_ilen(e);
        if (il != 7u) return il;
        const uint8_t* r = la.pool + ent_hi(e);
        return (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24;
    };
    int64_t v = lo;
    if (!(lo == a.vbeg || (lo > a.vbeg && a.in_v0[lo - 1] == (uint8_t)'\n'))) v = lo < a.vbeg ? a.vbeg : first_line_start_safe(a, lo, hi);
    // The lines that START in [lo, hi), as ONE loop that takes one table step per turn: an attempt (infer_dft) that goes on, the start of the
    // next one, the rest of a record nobody looks at — all are states of the lane, so the lanes of a wave, each somewhere else in its own
    // text, run the same few instructions per turn.  (The first version nested the attempt's loop in the position's in the line's: every
    // wave ran the longest attempt of its 64 lanes at every position, and 375 scalar instructions of branch bookkeeping per step.)
    int64_t i = v;                 // the attempt's cursor (== v: an attempt starts here)
    uint32_t row = 0;
    uint64_t acc = 0, attempt_at = 0;
    uint8_t c0 = 0;
    bool at_start = true;          // v is a line start
    bool skip = false;             // behind a NUL or a miss: the rest of the record is nobody's
    // One turn per byte looked at, written with selects and no way out of the middle of the loop: the lanes of a wave are each somewhere
    // else — an attempt starting, one going on, one dying, a line ending, a record's rest being skipped — and every `if` on such a
    // condition is a stretch of code the whole wave walks through, every `continue` a page of scalar mask bookkeeping (the first flat
    // version: 140 scalar instructions per turn).  Branches are left where a lane rarely goes: a miss, an escape to the pool, a full
    // output window.
    bool stop = false;
    while (!stop && !(at_start && v >= hi)) {
        const uint8_t c = byte_at(i);                      // (i == v while skipping)
        const bool in_skip = skip;
        const bool starting = !in_skip && i == v;
        // (the attempt on the empty tail — trre_dft.c:1284 — looks at no byte and the start state is never final: it prints nothing)
        const bool nul = starting && c == 0, eol = starting && c == (uint8_t)'\n', nl = nul || eol;
        // one attempt from v (infer_dft): walk until the first final state; a dead edge or the end of the line discards it
        c0 = starting ? c : c0;
        attempt_at = starting ? cnt : attempt_at;
        acc = starting ? 0 : acc;
        row = (starting || in_skip) ? 0u : row;
        const bool walk = !nl && !in_skip;
        steps += walk ? 1u : 0u;
        uint32_t k;
        const uint64_t e = entry(row, c, k);
        uint32_t kind = walk ? ent_kind(e) : 0u;
        bool missed = false;
        if (walk && (e == kLazyUnexplored || (uint32_t)e == (uint32_t)kLazyNoted)) {
            if (e != (kLazyNoted | (uint64_t)la.gen << 32)) lazy_note_miss(la, row, k, i + 1, byte_at);
            voided = true;
            status |= kStMiss;
            missed = true;                                 // nothing more to learn from this line
            kind = 0u;
        }
        if (steps > la.budget) { status |= kStEditOverflow; stop = true; }
        if (kind == 3u) { status |= kStDiverge; stop = true; kind = 0u; }     // the reference's closure never returns from this edge
        const bool go = kind != 0u;                        // the attempt takes this byte
        const bool one = !go && !in_skip;                  // one byte goes out: '\n' (a line end, a NUL, a miss) or the dead attempt's first byte (trre_dft.c:1281-1282)
        uint32_t il = go ? ent_ilen(e) : 0u;
        if (kMode == 1) {
            if (il == 7u) il = out_len(e);
            acc += il;
            cnt += go ? (kind == 2u ? acc : 0u) : (one ? 1u : 0u);
        } else {
            // the emit pass writes as it walks — a lane's output region is its own, an attempt that fails is written over, and one that
            // would run past the region's end is failing: its bytes beyond `lane_total` are not stored
            rewind_out(one ? attempt_at : cnt);
            if (il != 7u) {
                const uint64_t bytes = one ? (uint64_t)((nl || missed) ? (uint8_t)'\n' : c0) : ((uint64_t)ent_hi(e) & (il ? ~0ull >> (64u - 8u * il) : 0ull));
                append(bytes, one ? 1u : il);
            } else {
                const uint8_t* r = la.pool + ent_hi(e);
                const uint32_t len = (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24;
                for (uint32_t b = 0; b < len; ++b) put1(r[4 + b]);
            }
        }
        // where the lane stands next: behind the attempt's byte (it goes on, or is done: the next one starts there), or one byte on (a dead
        // edge, a line end, a skipped byte); a NUL or a miss stays where it is: the skipping starts there
        const bool to_skip = nul || missed;
        const int64_t v1 = in_skip ? v + 1 : (kind == 2u ? i + 1 : ((go || to_skip) ? v : v + 1));
        i = go ? i + 1 : v1;
        v = v1;
        row = ent_next(e);                                 // (kind 1; any other: reset when the next attempt starts)
        skip = in_skip ? c != (uint8_t)'\n' : to_skip;
        at_start = in_skip ? c == (uint8_t)'\n' : eol;
    }
    if (stop) { L.count = cnt; return; }
    flush_out();
    L.count = cnt;
}

}  // namespace trre
