// The stack guard (round 4): the one place where the reference's search FAILS although a match exists.
//
// infer_backtrack keeps the untried alternatives of the attempt it is in on a stack that may hold 65 536 items
// (trre_nft.c:35-36,548-556: the capacity doubles until the next doubling would pass 100 000); the push after that prints
// "error: stack max capacity reached" and exits 1 with what it had printed.  A loop over a run of 65 536 bytes gets there —
// ' +: ' on 70 000 spaces — and the table kernels, which never hold failing alternatives, print the match instead.
//
// Only a long line can do it: an attempt that has consumed k bytes holds at most D * (k + 1) items, D = the deepest nest of
// first-tried branches between two reads (a pattern constant, stack_guard.cpp) — and all of those k bytes but a handful were read
// inside loops: the line holds a long run of bytes the pattern's loops can read.  So the runtime looks for windows that consist of
// such bytes only (guard_probe: on ordinary text a window is left after a few bytes) and, for the lines that hold one, runs the
// reference's search as it is, state by state over the NFT the front end
// built (front.hpp: Nft mirrors create_nft), counting the stack: guard_line.  An overflow is reported like the reference
// reports it — TRRE_E_DIVERGES, "stack max capacity reached", the output up to the attempt that overflowed.
//
// Dual-compiled: the kernels of scan_kernels.hip and the host shim of the CPU test tier (tests/cpu_shim.cpp) run the same
// bodies.
#pragma once
#include "scan_block.hpp"

namespace trre {

constexpr uint32_t kMagicGuard = 0x314b5254u;   // "TRK1"
struct GuardBlobHeader {
    uint32_t magic, n_states, start, d;      // d: items per consumed byte, at most
    uint32_t l_min;                          // lines shorter than this cannot overflow
    uint32_t window;                         // the probe's window (a multiple of 16, <= l_min / 2)
    uint32_t off_states;                     // u32[n_states][4]: {kind | val << 8, a, b, 0}
    uint32_t total_bytes;
    uint32_t match;                          // trre -m: one attempt per line, FINAL accepts at the end of the line only
    uint32_t n_once;                         // CONS states on no cycle: the bytes of an attempt that need not be of the set
    uint32_t bset[8];                        // the bytes a window must consist of to be a suspect (never '\n', never NUL)
};
static_assert(sizeof(GuardBlobHeader) == 72, "header layout");
constexpr uint32_t kGuardProd = 0, kGuardCons = 1, kGuardSplit = 2, kGuardSplitNg = 3, kGuardJoin = 4, kGuardFinal = 5;   // (front.hpp: NKind)
constexpr uint32_t kGuardStackMax = 65536;   // live items (trre_nft.c:551: capacity * 2 > STACK_MAX_CAPACITY at capacity 65 536)

// a window of the input consists of bytes of the set only (the set never holds '\n'): leaving at the first other byte — on
// ordinary text after a few bytes
TRRE_HD bool guard_probe(const ScanArgs& a, const uint32_t (&bset)[8], int64_t lo, int64_t hi) {
    if (hi > a.vend - 1) hi = a.vend - 1;                         // (the last byte ends its record whatever it is: Q1)
    for (int64_t v = lo; v < hi; ++v) {
        const uint32_t c = a.in_v0[v];
        uint32_t w = bset[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) w = (c >> 5) == (uint32_t)k ? bset[k] : w;
        if (!((w >> (c & 31u)) & 1u)) return false;
    }
    return lo < hi;
}

struct GuardRun {            // a stretch of windows without a '\n': [first, last] in units of the window
    uint32_t first, last;
};
struct GuardResult {
    uint64_t line_start;     // (v coordinates minus vbeg: offsets into the caller's buffer)
    uint64_t bad_at;         // status 1: offset, in the line, of the attempt that overflowed; otherwise: the search steps the line took (the runtime
                             // adds them up against the call's budget)
    uint32_t status;         // 0 fine (or not a line of l_min bytes), 1 overflow, 2 gave up (budget / output room)
    uint32_t out_len;        // kOut: bytes of the line's output before that attempt; the search alone: the line's length
};
struct GuardArgs {
    const uint8_t* blob;
    const GuardRun* runs;
    GuardResult* results;
    uint32_t* stack;         // [slots][kGuardStackMax][3]: state, input offset, output offset
    uint8_t* obuf;           // [slots][obuf_cap]: the attempt's output buffer (PROD writes, FINAL prints)
    uint32_t obuf_cap;
    uint8_t* out;            // kOut: where the line's output goes
    uint64_t out_cap;
    uint64_t budget;         // search steps per line
};

// One run of windows: the line that covers it, and — when it is long enough — the reference's scan of that line with its stack
// counted.  kOut: the same again for the one line that overflowed, printing what the reference had printed of it.
template <bool kOut>
TRRE_HD void guard_line(const ScanArgs& a, const GuardArgs& ga, int64_t slot, int64_t run_index) {
    const GuardBlobHeader& h = *reinterpret_cast<const GuardBlobHeader*>(ga.blob);
    const uint32_t* const S = reinterpret_cast<const uint32_t*>(ga.blob + h.off_states);
    const GuardRun run = ga.runs[run_index];
    GuardResult R{};
    // the line: from behind the last '\n' before the run to the first one after it (the last byte of the input ends its record)
    // (the windows next to a run need not hold a '\n' — only a byte outside the set: the walk to the line's ends is as long as the line
    // — up to 4 GiB either way: a longer line is not decided)
    int64_t ls = a.vbeg + (int64_t)run.first * h.window;
    const int64_t ls_stop = ls - 0xffffffffll > a.vbeg ? ls - 0xffffffffll : a.vbeg;
    while (ls > ls_stop && a.in_v0[ls - 1] != (uint8_t)'\n') --ls;
    int64_t le = a.vbeg + ((int64_t)run.last + 1) * h.window;
    if (le > a.vend - 1) le = a.vend - 1;
    const int64_t le_stop = le + 0xffffffffll < a.vend - 1 ? le + 0xffffffffll : a.vend - 1;
    while (le < le_stop && a.in_v0[le] != (uint8_t)'\n') ++le;
    R.line_start = (uint64_t)(ls - a.vbeg);
    int64_t len = le - ls;
    for (int64_t v = ls; v < le; ++v)
        if (a.in_v0[v] == 0) { len = v - ls; break; }              // the record is a C string: a NUL ends it (Q2)
    if (len + 1 < (int64_t)h.l_min || len > 0xfffffff0ll) {         // too short to overflow (or beyond 32-bit offsets: not decided)
        R.status = len > 0xfffffff0ll ? 2u : 0u;
        ga.results[run_index] = R;
        return;
    }
    uint32_t* const stk = ga.stack + (size_t)slot * kGuardStackMax * 3;
    uint8_t* const ob = ga.obuf + (size_t)slot * ga.obuf_cap;
    const uint8_t* const in = a.in_v0 + ls;
    // Where an attempt that overflows can begin: it consumes at least l_min bytes of the line, all but n_once of them bytes of the
    // set — the last such position (one pass over the line, the window's far end running l_min bytes ahead of its near end).
    // None: the line cannot overflow, nothing is searched.  Attempts that begin behind the last one are not searched either.
    int64_t last_start = -1;
    {
        const uint32_t bs[8] = {h.bset[0], h.bset[1], h.bset[2], h.bset[3], h.bset[4], h.bset[5], h.bset[6], h.bset[7]};
        auto outside = [&](int64_t k) -> uint32_t { const uint32_t c = in[k]; return ((bs[c >> 5] >> (c & 31u)) & 1u) ^ 1u; };
        const int64_t w = (int64_t)h.l_min;
        uint64_t bad = 0;
        for (int64_t k = 0; k < len; ++k) {
            bad += outside(k);
            if (k >= w) bad -= outside(k - w);
            if (k + 1 >= w && bad <= (uint64_t)h.n_once) last_start = k + 1 - w;
        }
    }
    if (last_start < 0) {
        R.out_len = kOut ? 0u : (uint32_t)len;
        ga.results[run_index] = R;
        return;
    }
    uint64_t steps = 0, printed = 0;
    auto put = [&](uint8_t c) {
        if (kOut) {
            if (printed < ga.out_cap) ga.out[printed] = c;
            else R.status = 2u;
        }
        ++printed;
    };
    // infer_backtrack (trre_nft.c:593-657) at offset p: > 0 consumed, 0 / -1 no advance, -2 overflow, -3 gave up
    auto attempt = [&](uint32_t p) -> int64_t {
        uint32_t n_items = 0, i = p, o = 0;
        int32_t s = (int32_t)h.start;
        while (n_items || s >= 0) {
            if (++steps > ga.budget) return -3;
            if (s < 0) {
                --n_items;
                s = (int32_t)stk[3 * (size_t)n_items]; i = stk[3 * (size_t)n_items + 1]; o = stk[3 * (size_t)n_items + 2];
                if (s < 0) continue;
            }
            const uint32_t kv = S[4 * (size_t)s], kind = kv & 0xffu;
            const int32_t sa = (int32_t)S[4 * (size_t)s + 1], sb = (int32_t)S[4 * (size_t)s + 2];
            if (kind == kGuardCons) {
                if (i < (uint32_t)len && (uint8_t)(kv >> 8) == in[i]) { ++i; s = sa; }
                else s = -1;
            } else if (kind == kGuardProd) {
                if (kOut) {                                                  // (the search alone needs no output, only its length)
                    if (o + 1u >= ga.obuf_cap) return -3;
                    ob[o] = (uint8_t)(kv >> 8);
                }
                ++o;
                s = sa;
            } else if (kind == kGuardSplit || kind == kGuardSplitNg) {
                if (n_items >= kGuardStackMax) return -2;                   // spush: "stack max capacity reached"
                stk[3 * (size_t)n_items] = (uint32_t)(kind == kGuardSplit ? sa : sb);
                stk[3 * (size_t)n_items + 1] = i; stk[3 * (size_t)n_items + 2] = o;
                ++n_items;
                s = kind == kGuardSplit ? sb : sa;
            } else if (kind == kGuardJoin) {
                s = sa;
            } else if (h.match && i != (uint32_t)len) {                     // trre -m (trre_nft.c:635-642): not at the end of the line — on
                s = -1;
            } else {                                                        // FINAL: fputs(output), return i
                if (kOut) { for (uint32_t k = 0; k < o && ob[k]; ++k) put(ob[k]); }
                return (int64_t)i - (int64_t)p;
            }
        }
        return -1;
    };
    uint32_t p = 0;
    int64_t r = 0;
    if (h.match) {                                                          // trre_nft.c:790-793: one attempt; what it prints is the whole line's output
        r = attempt(0);
        R.bad_at = steps;
        if (r == -2) { R.status = 1u; R.bad_at = 0; }
        else if (r == -3) R.status = 2u;
        R.out_len = kOut ? 0u : (uint32_t)len;
        ga.results[run_index] = R;
        return;
    }
    while (p < (uint32_t)len) {                                             // trre_nft.c:778-784
        if (!kOut && (int64_t)p > last_start) { r = 0; break; }             // (no attempt from here on can hold that many items)
        r = attempt(p);
        if (r <= -2) break;
        if (r > 0) p += (uint32_t)r;
        else { put(in[p]); ++p; }
    }
    if (r > -2 && p >= (uint32_t)len) r = attempt((uint32_t)len);            // the empty tail (trre_nft.c:786)
    R.bad_at = steps;
    if (r == -2) { R.status = 1u; R.bad_at = p; }
    else if (r == -3) R.status = 2u;
    R.out_len = kOut ? (uint32_t)(printed > 0xffffffffull ? 0xffffffffull : printed) : (uint32_t)len;    // (the search alone: the line's length)
    if (kOut && printed > ga.out_cap) R.status = 2u;
    ga.results[run_index] = R;
}

}  // namespace trre
