// gen_block.hpp — generator mode (`trre -a`, `trre -ma`) on the device: the enumeration of ALL accepting paths
// (infer_backtrack with all = 1, trre_nft.c:593-657: FINAL prints and the search goes on, :640-641, :647-648; the line
// loops trre_nft.c:775-797).
//
// Round 3 ran the viability filter on the device (k_rev_sweep over the tables of generate.cpp) and the enumeration on
// host threads.  Here the enumeration is a kernel too: a lane owns the records that start in its sub-range and walks, for
// every attempt, the follow lists depth first with an explicit stack in its own stretch of scratch memory, entering a node
// only if the symbol of the byte says some path from it prints (or never returns): the work is proportional to what is
// printed.  Two passes like every general family — count, exclusive sum, emit — because the amount of output is unbounded
// in the input.  What the kernel does not do itself: a path that runs into an epsilon cycle (the reference exits 1 with what
// it had printed), a line whose search goes deeper than the lane's stack or builds a longer output than its path buffer —
// it says so (kStDiverge / kStEditOverflow) and the runtime hands that chunk of the input to the host enumeration of
// generate.cpp, which is also what the tests check the kernel against.
//
// The lane body is TRRE_HD: tests/cpu_shim.cpp runs it lane by lane on the host.
#pragma once
#include "scan_block.hpp"

namespace trre {

constexpr uint32_t kMagicGen = 0x31475254u;   // "TRG1"
struct GenBlobHeader {
    uint32_t magic, n_nodes, n_rev, words, match_mode, n_follow;
    uint32_t off_bytes;      // u32[n_nodes][8]: the bytes a node reads
    uint32_t off_echo;       // u8[n_nodes]: copy mode — reading a byte also produces it
    uint32_t off_foff;       // u32[n_nodes + 2]: follow lists; [n_nodes] = the start's
    uint32_t off_follow;     // u32[n_follow][3]: {target, offset of the bytes produced on the way, their number | mute << 16}
    uint32_t off_pool, pool_bytes;
    uint32_t off_viable;     // u64[n_rev][words]: nodes worth entering at a byte with that symbol
    uint32_t total_bytes;
};
static_assert(sizeof(GenBlobHeader) == 56, "header layout");

struct GenView {
    const uint32_t* bytes;
    const uint8_t* echo;
    const uint32_t* foff;
    const uint32_t* follow;
    const uint8_t* pool;
    const uint64_t* viable;
    uint32_t n_nodes, words, match;
};
TRRE_HD GenView gen_view(const uint8_t* blob) {
    const GenBlobHeader& h = *reinterpret_cast<const GenBlobHeader*>(blob);
    GenView G;
    G.bytes = reinterpret_cast<const uint32_t*>(blob + h.off_bytes);
    G.echo = blob + h.off_echo;
    G.foff = reinterpret_cast<const uint32_t*>(blob + h.off_foff);
    G.follow = reinterpret_cast<const uint32_t*>(blob + h.off_follow);
    G.pool = blob + h.off_pool;
    G.viable = reinterpret_cast<const uint64_t*>(blob + h.off_viable);
    G.n_nodes = h.n_nodes; G.words = h.words; G.match = h.match_mode;
    return G;
}

// a lane's scratch: its stack of frames {list, next entry, input position, output length | muted << 31} and the output of
// the path it is on
struct GenArgs {
    uint32_t* stack;         // [n_lanes][frames][4]
    uint8_t* path;           // [n_lanes][path_cap]
    uint32_t frames, path_cap;
};
constexpr uint32_t kGenTgtFinal = 0xFFFFFFFFu, kGenTgtDiverge = 0xFFFFFFFEu;    // (front.hpp: kNodeFinal, kNodeDiverge)

// kMode 1: count; 2: emit at a.out + out_base
template <int kMode>
TRRE_HD void gen_lane(const ScanArgs& a, const GenView& G, const GenArgs& ga, int64_t lane, int64_t lane_bytes, uint64_t out_base, DirectLane& L,
                      uint32_t& status) {
    const int64_t lo = lane * lane_bytes;
    int64_t hi = lo + lane_bytes;
    if (hi > a.vend) hi = a.vend;
    uint64_t cnt = 0;
    L.count = 0;
    if (lo >= hi) return;
    uint32_t* const stack = ga.stack + (size_t)lane * ga.frames * 4;
    uint8_t* const path = ga.path + (size_t)lane * ga.path_cap;
    uint8_t* op = kMode == 2 ? a.out + out_base : nullptr;
    auto byte_at = [&](int64_t v) -> uint8_t { return v >= a.vend - 1 ? (uint8_t)'\n' : a.in_v0[v]; };   // the last byte ends its record (Q1)
    auto put = [&](const uint8_t* src, uint32_t n) {
        if (kMode == 2) {
            // (eight bytes at a time: neither side is aligned, global memory does not mind)
            uint32_t i = 0;
            for (; i + 8u <= n; i += 8u) reinterpret_cast<UnalignedU64*>(op + cnt + i)->v = reinterpret_cast<const UnalignedU64*>(src + i)->v;
            for (; i < n; ++i) op[cnt + i] = src[i];
        }
        cnt += n;
    };
    auto put1 = [&](uint8_t c) {
        if (kMode == 2) op[cnt] = c;
        cnt += 1;
    };
    // all accepting paths of ONE attempt at position p of the record [rec, rec + len)
    auto attempt = [&](int64_t rec, uint32_t len, uint32_t p) -> bool {
        uint32_t sp = 1;
        stack[0] = G.n_nodes; stack[1] = 0; stack[2] = p; stack[3] = 0;
        while (sp) {
            uint32_t* f = stack + 4 * (sp - 1);
            const uint32_t list = f[0], idx = f[1], fi = f[2], fo = f[3];
            const uint32_t beg = G.foff[list], end = G.foff[list + 1];
            if (beg + idx >= end) { --sp; continue; }
            f[1] = idx + 1;
            const uint32_t* e = G.follow + 3 * (size_t)(beg + idx);
            const uint32_t target = e[0], out_off = e[1], out_len = e[2] & 0xffffu, mute = e[2] >> 16;
            const uint32_t olen = fo & 0x7fffffffu, muted = fo >> 31;
            if (target == kGenTgtDiverge) { status |= kStDiverge; return false; }
            if (target == kGenTgtFinal) {
                if (G.match && fi != len) continue;                                 // trre_nft.c:636: only with the whole line consumed
                put(path, olen);
                if (!muted) put(G.pool + out_off, out_len);
                if (G.match) put1((uint8_t)'\n');
                continue;
            }
            if (fi >= len) continue;
            const uint8_t c = a.in_v0[rec + fi];
            if (!((G.bytes[8 * (size_t)target + (c >> 5)] >> (c & 31u)) & 1u)) continue;
            const uint64_t v = G.viable[(size_t)a.sym_v0[rec + fi] * G.words + (target >> 6)];
            if (!((v >> (target & 63u)) & 1u)) continue;                            // nothing printed, nothing entered below: skip
            uint32_t nlen = olen, nmuted = muted;
            if (!muted) {
                if (olen + out_len + 1u > ga.path_cap) { status |= kStEditOverflow; return false; }
                for (uint32_t i = 0; i < out_len; ++i) path[olen + i] = G.pool[out_off + i];
                nlen = olen + out_len;
                if (mute) nmuted = 1;
                else if (G.echo[target]) path[nlen++] = c;
            }
            if (sp >= ga.frames) { status |= kStEditOverflow; return false; }
            uint32_t* nf = stack + 4 * sp;
            nf[0] = target; nf[1] = 0; nf[2] = fi + 1; nf[3] = nlen | nmuted << 31;
            ++sp;
        }
        return true;
    };
    // the records that start in [lo, hi)
    int64_t pos = lo;
    if (!(lo == a.vbeg || (lo > a.vbeg && a.in_v0[lo - 1] == (uint8_t)'\n'))) pos = lo < a.vbeg ? a.vbeg : first_line_start_safe(a, lo, hi);
    while (pos < hi) {
        int64_t e = pos;
        uint32_t len = 0xffffffffu;
        for (;; ++e) {
            const uint8_t c = byte_at(e);
            if (c == (uint8_t)'\n') break;
            if (c == 0 && len == 0xffffffffu) len = (uint32_t)(e - pos);            // cut at the first NUL (Q2)
        }
        if (len == 0xffffffffu) {
            if (e - pos > 0x7ffffff0ll) { status |= kStEditOverflow; break; }
            len = (uint32_t)(e - pos);
        }
        bool ok = true;
        if (G.match) {
            ok = attempt(pos, len, 0);                                              // trre_nft.c:791-797
        } else {
            for (uint32_t p = 0; p < len && ok; ++p) {                              // trre_nft.c:780-786 with all = 1: no attempt returns > 0
                ok = attempt(pos, len, p);
                if (ok) put1(a.in_v0[pos + p]);
            }
            if (ok) ok = attempt(pos, len, len);                                    // the empty tail (trre_nft.c:788)
            if (ok) put1((uint8_t)'\n');
        }
        if (!ok) break;                                                             // (the chunk goes to the host enumeration)
        pos = e + 1;
    }
    L.count = cnt;
}

// =============================================================================================
// The search itself, for the patterns nothing else runs (round 4): infer_backtrack with all = 0 (trre_nft.c:593-657: depth
// first, the first path that reaches FINAL wins) and the scan line loop around it (trre_nft.c:775-790), a lane per sub-range
// and an explicit stack in scratch memory — what the reference does, on every line at once.  The guided families answer
// every attempt in time linear in the line; this walker is exponential where the reference is.  It is what an NFT pattern
// runs on when its backward automaton has more than 16 384 states, it does not fold and it has more than 64 nodes (round 3:
// TRRE_E_UNSUPPORTED): the engine then runs every pattern the reference runs.  Tables: the follow lists of nft_tables.cpp
// (first occurrence of a target per list, a list ends at its first FINAL) in the generator modes' blob form; no symbols.
// An attempt that takes more than `budget` steps, goes deeper than the lane's stack or builds more output than its path
// buffer gives up (kStCapacity is not it: kStEditOverflow) — an error, not a hang; an epsilon cycle is kStDiverge.  `why` says
// which limit it was (kBtWhy*): the runtime runs the buffer again with fewer, larger stacks and path buffers when it was one of
// those two (round 5: attempts of up to 1 M bytes / 1 MiB of output; round 4 gave up at 4 096 / 4 KiB).
// With tables in match form (G.match, round 5: `trre -m` on a pattern beyond the guided tables' limits): one attempt per line
// from its first byte, FINAL accepts only with the whole line consumed and the search goes on past it otherwise
// (trre_nft.c:635-642); an accepted line prints its output and '\n', a rejected one nothing (trre_nft.c:791-797).
// =============================================================================================
constexpr uint32_t kBtWhyBudget = 1u, kBtWhyFrames = 2u, kBtWhyPath = 4u;
#ifndef TRRE_BT_SCAN_BYTES
#define TRRE_BT_SCAN_BYTES 4
#endif
constexpr int kBtScanBytes = TRRE_BT_SCAN_BYTES;
template <int kMode>
TRRE_HD void bt_lane(const ScanArgs& a, const GenView& G, const GenArgs& ga, int64_t slot, int64_t lane, int64_t lane_bytes, uint64_t out_base,
                     uint32_t budget, DirectLane& L, uint32_t& status, uint32_t& why) {
    const int64_t lo = lane * lane_bytes;
    int64_t hi = lo + lane_bytes;
    if (hi > a.vend) hi = a.vend;
    uint64_t cnt = 0;
    L.count = 0;
    if (lo >= hi) return;
    uint32_t* const stack = ga.stack + (size_t)slot * ga.frames * 4;
    uint8_t* const path = ga.path + (size_t)slot * ga.path_cap;
    uint8_t* op = kMode == 2 ? a.out + out_base : nullptr;
    auto put = [&](const uint8_t* src, uint32_t n) {
        if (kMode == 2) { for (uint32_t i = 0; i < n; ++i) op[cnt + i] = src[i]; }
        cnt += n;
    };
    auto put1 = [&](uint8_t c) {
        if (kMode == 2) op[cnt] = c;
        cnt += 1;
    };
    // the first accepting path of ONE attempt at position p: prints its output; returns the bytes it consumed, -1: no path, -2: gave up
    uint32_t steps = 0;                                                             // (of the whole sub-range: the launch ends in bounded time)
    // (a record's content ends at its '\n', at a NUL before it (Q2) or at the last byte of the input, which ends its record
    // whatever it is (Q1): looked at as the walk gets there — no pass over the line beforehand)
    auto ends = [&](int64_t v, uint8_t c) -> bool { return c == (uint8_t)'\n' || c == 0 || v >= a.vend - 1; };
    auto attempt = [&](int64_t v0) -> int64_t {                                     // at position v0 of the input
        uint32_t sp = 1;
        stack[0] = G.n_nodes; stack[1] = 0; stack[2] = 0; stack[3] = 0;
        while (sp) {
            if (++steps > budget) { status |= kStEditOverflow; why |= kBtWhyBudget; return -2; }
            uint32_t* f = stack + 4 * (sp - 1);
            const uint32_t list = f[0], idx = f[1], fi = f[2], fo = f[3];
            const uint32_t beg = G.foff[list], end = G.foff[list + 1];
            if (beg + idx >= end) { --sp; continue; }
            f[1] = idx + 1;
            const uint32_t* e = G.follow + 3 * (size_t)(beg + idx);
            const uint32_t target = e[0], out_off = e[1], out_len = e[2] & 0xffffu, mute = e[2] >> 16;
            const uint32_t olen = fo & 0x7fffffffu, muted = fo >> 31;
            if (target == kGenTgtDiverge) { status |= kStDiverge; return -2; }
            if (target == kGenTgtFinal) {                                           // trre_nft.c:643-648: print, return the offset
                if (G.match && !ends(v0 + fi, a.in_v0[v0 + fi])) continue;          // trre_nft.c:636: only with the whole line consumed
                put(path, olen);
                if (!muted) put(G.pool + out_off, out_len);
                return (int64_t)fi;
            }
            const uint8_t c = a.in_v0[v0 + fi];
            if (ends(v0 + fi, c)) continue;
            if (!((G.bytes[8 * (size_t)target + (c >> 5)] >> (c & 31u)) & 1u)) continue;
            uint32_t nlen = olen, nmuted = muted;
            if (!muted) {
                if (olen + out_len + 1u > ga.path_cap) { status |= kStEditOverflow; why |= kBtWhyPath; return -2; }
                for (uint32_t i = 0; i < out_len; ++i) path[olen + i] = G.pool[out_off + i];
                nlen = olen + out_len;
                if (mute) nmuted = 1;
                else if (G.echo[target]) path[nlen++] = c;
            }
            if (sp >= ga.frames) { status |= kStEditOverflow; why |= kBtWhyFrames; return -2; }
            uint32_t* nf = stack + 4 * sp;
            nf[0] = target; nf[1] = 0; nf[2] = fi + 1; nf[3] = nlen | nmuted << 31;
            ++sp;
        }
        return -1;
    };
    if (G.match) {
        // `trre -m`: the lines that START in [lo, hi), one attempt each
        int64_t v = lo;
        if (!(lo == a.vbeg || (lo > a.vbeg && a.in_v0[lo - 1] == (uint8_t)'\n'))) v = lo < a.vbeg ? a.vbeg : first_line_start_safe(a, lo, hi);
        while (v < hi) {
            const int64_t r = attempt(v);
            if (r == -2) break;
            if (r >= 0) put1((uint8_t)'\n');
            while (v < a.vend - 1 && a.in_v0[v] != (uint8_t)'\n') ++v;
            ++v;
        }
        L.count = cnt;
        return;
    }
    // the bytes an attempt can begin with (all of them when the start's list reaches FINAL or a cycle without reading): elsewhere the
    // attempt is known to fail and the byte is copied without a search
    uint32_t first[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = G.foff[G.n_nodes]; k < G.foff[G.n_nodes + 1]; ++k) {
        const uint32_t target = G.follow[3 * (size_t)k];
        for (int w = 0; w < 8; ++w) first[w] |= target >= kGenTgtDiverge ? 0xffffffffu : G.bytes[8 * (size_t)target + w];
    }
    int64_t v = lo;
    if (!(lo == a.vbeg || (lo > a.vbeg && a.in_v0[lo - 1] == (uint8_t)'\n'))) v = lo < a.vbeg ? a.vbeg : first_line_start_safe(a, lo, hi);
    // The lines that START in [lo, hi), byte by byte; 16 input bytes at a time in registers.  ONE loop: a turn is either one byte of the
    // line loop (trre_nft.c:780-788: most bytes cannot begin a match and are copied) or one step of the search an earlier turn began —
    // the lanes of a wave are each somewhere else in their own text, and with the search as a loop of its own inside the line loop every
    // wave ran the longest search of its 64 lanes at every position (round 5's counters: 1 170 vector and 1 000 scalar instructions per
    // byte looked at).  No `continue`, one way out (see lazy_block.hpp on what those cost).
    U128 blk{};
    int64_t blk_at = -16;
    bool open = false;                                                              // inside a line of this lane's
    uint32_t mode = 0;                                                              // 0: the line loop; 1: searching from v (c0 could begin a match); 2: searching the empty tail at v
    uint32_t sp = 0;
    uint8_t c0 = 0;
    bool stop = false;
    while (!stop) {
        if (mode == 0u) {
            // (up to kBtScanBytes bytes of the line loop per turn: a search step waits on memory — a frame, list bounds, a follow entry, the
            // byte, its set — and nearly every turn has some lane of the wave in one; the lanes that copy do not wait with it byte by byte)
#pragma clang loop unroll(disable)
            for (int t = 0; t < kBtScanBytes && mode == 0u && !stop; ++t) {
                if (!open && v >= hi) {
                    stop = true;
                } else {
                    open = true;
                    if ((v & ~(int64_t)15) != blk_at) { blk_at = v & ~(int64_t)15; blk = *reinterpret_cast<const U128*>(a.in_v0 + blk_at); }
                    const uint32_t q = (uint32_t)(v >> 2) & 3u;
                    const uint32_t dw = q == 0 ? blk.x : (q == 1 ? blk.y : (q == 2 ? blk.z : blk.w));
                    c0 = (uint8_t)(dw >> (8u * ((uint32_t)v & 3u)));
                    uint32_t fw = first[0];
#pragma unroll
                    for (int w = 1; w < 8; ++w) fw = (c0 >> 5) == w ? first[w] : fw;
                    const bool tail = ends(v, c0);                                  // the empty tail is searched too (trre_nft.c:788)
                    if (tail || ((fw >> (c0 & 31u)) & 1u)) {
                        mode = tail ? 2u : 1u;
                        sp = 1;
                        stack[0] = G.n_nodes; stack[1] = 0; stack[2] = 0; stack[3] = 0;
                    } else {
                        put1(c0);                                                   // trre_nft.c:780-786
                        ++v;
                    }
                }
            }
        } else {
            // one step of the search (infer_backtrack, all = 0): r — the bytes the first accepting path consumed, -1: no path; -3: not done
            int64_t r = -3;
            if (++steps > budget) {
                status |= kStEditOverflow; why |= kBtWhyBudget; stop = true;
            } else {
                uint32_t* f = stack + 4 * (sp - 1);
                const uint32_t list = f[0], idx = f[1], fi = f[2], fo = f[3];
                const uint32_t beg = G.foff[list], end = G.foff[list + 1];
                if (beg + idx >= end) {
                    --sp;
                    if (sp == 0) r = -1;
                } else {
                    f[1] = idx + 1;
                    const uint32_t* e = G.follow + 3 * (size_t)(beg + idx);
                    const uint32_t target = e[0], out_off = e[1], out_len = e[2] & 0xffffu, mute = e[2] >> 16;
                    const uint32_t olen = fo & 0x7fffffffu, muted = fo >> 31;
                    if (target == kGenTgtDiverge) {
                        status |= kStDiverge; stop = true;
                    } else if (target == kGenTgtFinal) {                            // trre_nft.c:643-648: print, return the offset
                        put(path, olen);
                        if (!muted) put(G.pool + out_off, out_len);
                        r = (int64_t)fi;
                    } else {
                        const uint8_t c = a.in_v0[v + fi];
                        if (!ends(v + fi, c) && ((G.bytes[8 * (size_t)target + (c >> 5)] >> (c & 31u)) & 1u)) {
                            uint32_t nlen = olen, nmuted = muted;
                            bool room = true;
                            if (!muted) {
                                if (olen + out_len + 1u > ga.path_cap) {
                                    status |= kStEditOverflow; why |= kBtWhyPath; stop = true; room = false;
                                } else {
                                    for (uint32_t k = 0; k < out_len; ++k) path[olen + k] = G.pool[out_off + k];
                                    nlen = olen + out_len;
                                    if (mute) nmuted = 1;
                                    else if (G.echo[target]) path[nlen++] = c;
                                }
                            }
                            if (room) {
                                if (sp >= ga.frames) {
                                    status |= kStEditOverflow; why |= kBtWhyFrames; stop = true;
                                } else {
                                    uint32_t* nf = stack + 4 * sp;
                                    nf[0] = target; nf[1] = 0; nf[2] = fi + 1; nf[3] = nlen | nmuted << 31;
                                    ++sp;
                                }
                            }
                        }
                    }
                }
            }
            if (r != -3) {
                if (mode == 2u) {
                    put1((uint8_t)'\n');
                    if (c0 != (uint8_t)'\n') {                                      // behind a NUL: the rest of the record is nobody's
                        while (v < a.vend - 1 && a.in_v0[v] != (uint8_t)'\n') ++v;
                    }
                    ++v;
                    open = false;
                } else if (r > 0) {
                    v += r;
                } else {
                    put1(c0);                                                       // no match, or an empty one (its output is printed: Q3)
                    ++v;
                }
                mode = 0u;
            }
        }
    }
    L.count = cnt;
}

}  // namespace trre
