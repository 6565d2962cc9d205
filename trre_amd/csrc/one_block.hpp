// one_block.hpp — ONE walk for programs whose output is not as long as their input (round 6).
//
// The count / emit pair of the general families reads the input twice and walks it twice (three times with the backward pass of the
// guided families): a lane's place in the output is the sum of the sizes of ALL lanes before it, so the count walk had to finish before
// the emit walk could store a byte.  Rounds 3 and 4 tried "walk once, list the edits, assemble the output in a second pass" and lost
// (DESIGN.md §4.5).  What round 5's exact sub-ranges make possible instead: a lane may own ANY stretch of bytes — it starts from a
// guessed-and-verified state — so lanes can be SHORT (128 bytes), and then a workgroup's whole output fits its LDS:
//
//   walk      every lane walks its 128 bytes once (g16_lane<3>: the emit walk, appending to a private linear region of LDS) from the state
//             it guesses after `look` bytes of context;
//   verify    lane i's entry must be lane i - 1's exit — checked in LDS, a lane that guessed wrong walks again from the right state
//             (and on, round by round, while exits keep changing; a workgroup that does not settle in kOneRounds gives up);
//   sizes     wave + workgroup prefix sums of the lanes' sizes (LDS);
//   base      decoupled look-back over the workgroups' totals (one 8-byte descriptor per tile of input: {state, exit row of the tile's
//             last lane, total or running total}): the tile's place in the output, and the check of its FIRST lane's guess against the
//             exit of the tile before it — the one guess nobody can repair in place; a wrong one voids the launch (kStOneVoid) and the
//             count / emit pair runs the buffer;
//   store     output-parallel: a thread per 16-byte line of the output gathers it from the regions of the (one or two, rarely more)
//             lanes it spans and stores it — whole aligned 16-byte lines, 1 KiB per wave instruction.
//
// The input is read once, nothing but the output is written, and no line index or edit list passes through HBM.
// The per-thread bodies below are TRRE_HD: tests/cpu_shim.cpp runs them thread by thread with the barriers as loops.
// Matches: the scan loops trre_nft.c:775-790 / trre_dft.c:1272-1286 (framing), the emits trre_dft.c:1121-1122 / trre_nft.c:645.
#pragma once
#include "scan_block.hpp"

namespace trre {

constexpr int kOneThreads = 256;            // lanes per tile
constexpr int kOneRounds = 8;               // repair rounds a tile may take before it gives up
// look-back descriptors: [63:62] 0 nothing yet, 1 the tile's own total, 2 the running total up to and including the tile;
// [61:42] the exit row of the tile's last lane (a row offset: below 1 MiB); [41:0] the total
constexpr uint64_t kOneDescAgg = 1ull << 62, kOneDescInc = 2ull << 62;
constexpr uint64_t kOneValMask = (1ull << 42) - 1;
TRRE_HD uint64_t one_desc(uint64_t kind, uint32_t exit_row, uint64_t value) { return kind | (uint64_t)(exit_row & 0xfffffu) << 42 | (value & kOneValMask); }
TRRE_HD uint32_t one_desc_exit(uint64_t d) { return (uint32_t)(d >> 42) & 0xfffffu; }

struct OneArgs {
    uint64_t* desc;          // [n_tiles] look-back descriptors, zeroed before the launch
    uint64_t* gsum;          // [n_tiles / 32 + 1] per group of 32 tiles: [63:58] how many of them have added their total, [57:0] the sum (zeroed)
    uint64_t* ginc;          // [n_tiles / 32 + 1] [63] set: [62:0] the running total up to the end of the group (zeroed)
    uint32_t* ticket;        // [1] the next tile, zeroed before the launch (a workgroup takes tiles in the order it asks: every tile before
                             // the one it holds is in some workgroup's hands, so a look-back never waits for work that has not started)
    uint64_t* total;         // [1] the size of the whole output (written by the last tile)
    int64_t n_tiles;
    uint32_t lane_bytes;     // S: input bytes per lane (a multiple of 64)
    uint32_t region;         // R: bytes of a lane's LDS region (a multiple of 4, R / 4 odd: lanes in step hit distinct banks)
    uint32_t look;           // bytes of context a lane guesses its entry state from
    uint32_t spin;           // look-back polls before a tile gives up (the launch is void then — never a hang)
    uint64_t* prof;          // TRRE_ONE_PROF=1: [8] shader clocks per phase summed over the tiles (thread 0's) — ticket, walk, verify, sizes, look-back,
                             // store, and [7] the tiles; else null
};

// the tile's lanes' regions in LDS: lane t at regions + t * region; sizes[t] bytes used; offs[t] = exclusive prefix sum (offs[nl] = total)
struct OneTile {
    const uint8_t* regions;
    const uint32_t* offs;    // [kOneThreads + 1]
    uint8_t* mark;           // [chunks]: the lane whose bytes hold the first byte of output line c
    uint32_t region;
};

// Output lines: the tile's output starts at global address g0 = out + base (any alignment); line c covers the tile-relative positions
// [16 c - h, 16 c - h + 16) with h = g0 & 15.  Lane t fills the marks of the lines whose first byte is one of its own.
TRRE_HD void one_mark(const OneTile& t, int lane, uint32_t h) {
    const uint32_t lo = t.offs[lane], hi = t.offs[lane + 1];
    if (lo >= hi) return;
    for (uint32_t c = (lo + h + 15u) >> 4; (c << 4) < hi + h; ++c) t.mark[c] = (uint8_t)lane;
}
// 16 bytes of the tile's output from tile-relative position p (>= 0, < total), gathered from the regions — for kLines lines at once (two
// independent chains of LDS round trips per thread instead of one).  Positions at or beyond `total` come out as whatever lies there.
// Source k + 1 is read from "f bytes before its region" so that its bytes fall into place, and merged under a byte mask.
TRRE_HD U128 one_read16(const uint8_t* src) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(src);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3u);
    const uint32_t r0 = w[0], r1 = w[1], r2 = w[2], r3 = w[3], r4 = w[4];
    return U128{alignbyte_b32(r1, r0, sh), alignbyte_b32(r2, r1, sh), alignbyte_b32(r3, r2, sh), alignbyte_b32(r4, r3, sh)};
}
template <int kLines>
TRRE_HD void one_gather(const OneTile& t, const uint32_t (&p)[kLines], uint32_t (&lane)[kLines], const uint32_t (&total)[kLines], U128 (&d)[kLines]) {
    uint32_t f[kLines];                                  // bytes of the line that are in place
#pragma unroll
    for (int j = 0; j < kLines; ++j) {
        d[j] = one_read16(t.regions + (size_t)lane[j] * t.region + (p[j] - t.offs[lane[j]]));
        f[j] = t.offs[lane[j] + 1] - p[j];
    }
    // (the lanes of a wave go round together: one more round whenever some line of the wave spans one more lane)
    for (;;) {
        bool more = false;
#pragma unroll
        for (int j = 0; j < kLines; ++j) more = more || (f[j] < 16u && p[j] + f[j] < total[j]);
        if (!TRRE_WAVE_ANY(more)) break;
#pragma unroll
        for (int j = 0; j < kLines; ++j) {
            if (f[j] < 16u && p[j] + f[j] < total[j]) {
                ++lane[j];
                const uint32_t have = t.offs[lane[j] + 1] - t.offs[lane[j]];
                if (have) {
                    const U128 s = one_read16(t.regions + (size_t)lane[j] * t.region - f[j]);
                    // bytes [f, 16) from s
                    const uint32_t fb = f[j] << 3;       // bit position of byte f
                    const uint32_t m0 = fb >= 32u ? 0u : 0xffffffffu << fb;
                    const uint32_t m1 = fb >= 64u ? 0u : (fb <= 32u ? 0xffffffffu : 0xffffffffu << (fb - 32u));
                    const uint32_t m2 = fb >= 96u ? 0u : (fb <= 64u ? 0xffffffffu : 0xffffffffu << (fb - 64u));
                    const uint32_t m3 = fb <= 96u ? 0xffffffffu : 0xffffffffu << (fb - 96u);
                    d[j].x = (d[j].x & ~m0) | (s.x & m0);
                    d[j].y = (d[j].y & ~m1) | (s.y & m1);
                    d[j].z = (d[j].z & ~m2) | (s.z & m2);
                    d[j].w = (d[j].w & ~m3) | (s.w & m3);
                    f[j] += have;
                }
            }
        }
    }
}
// Lines c[0 .. kLines) of the tile (see one_mark) to memory: whole when all 16 bytes of a line are the tile's, byte by byte at the tile's two
// ends, whose lines it shares with its neighbours.  write = false: nothing is stored (the output does not fit the caller's buffer).  Every
// thread of a wave calls this together (one_gather's loop is the wave's); a line beyond the tile's output is nobody's.
template <int kLines>
TRRE_HD void one_store_lines(const OneTile& t, uint8_t* out, uint64_t base, uint32_t h, const uint32_t (&c)[kLines], uint32_t total, bool write) {
    uint32_t p[kLines], lane[kLines], tot[kLines];
    bool head[kLines], active[kLines];
    U128 d[kLines];
#pragma unroll
    for (int j = 0; j < kLines; ++j) {
        head[j] = c[j] == 0 && h != 0;                   // the line the tile's first byte lies in, not at its start
        const uint32_t pj = head[j] ? 0u : (c[j] << 4) - h;
        active[j] = pj < total;
        p[j] = active[j] ? pj : 0u;
        lane[j] = active[j] && !head[j] ? (uint32_t)t.mark[c[j]] : 0u;
        tot[j] = active[j] ? total : 0u;
    }
    one_gather<kLines>(t, p, lane, tot, d);
#pragma unroll
    for (int j = 0; j < kLines; ++j) {
        if (!active[j] || !write) continue;
        uint8_t* g = out + base + p[j];
        const uint32_t room = head[j] ? 16u - h : 16u;
        const uint32_t n = total - p[j] < room ? total - p[j] : room;
        if (n == 16u) {
            *reinterpret_cast<U128*>(g) = d[j];
            continue;
        }
        const uint32_t wd[4] = {d[j].x, d[j].y, d[j].z, d[j].w};
        for (uint32_t k = 0; k < n; ++k) g[k] = (uint8_t)(wd[k >> 2] >> (8u * (k & 3u)));
    }
}

}  // namespace trre
