// map_block.hpp — MEMORYLESS programs of any output length in one pass at the pace of memory (round 6).
//
// The byte map (k_bytemap) runs the programs whose every attempt is decided by ONE byte and prints one byte for it ('[a:A-z:Z]', Caesar):
// output position == input position, a pure streaming map.  The same programs with another output length per byte — 'a:xyz', '[aie]:',
// '(<:&lt;|>:&gt;)' — were walked by the general small-table family like any transducer: a count walk and an emit walk, a dependent table
// read per byte and lane (1.08 TB/s).  But a program whose folded scan loop never leaves the root state (stream_build.cpp:
// every cell of the root row leads back to it) has NO state to carry: what byte i becomes is a function of byte i alone, and where it goes
// is a prefix sum of lengths — stream compaction / expansion, the textbook single-pass scan:
//
//   tile     4 waves x 8 ROWS of 512 bytes (16 KiB of input); workgroup w of G takes the tiles w, w + G, w + 2 G, .. (no ticket: its next tiles
//            are known ahead, their bytes can be on their way).  A lane holds 8 consecutive
//            bytes of each of its wave's rows (a `global_load_dwordx2`: 512 bytes side by side per instruction) — the lanes of a wave then write
//            to LDS 8-odd bytes apart: two lanes per bank, which a byte store's own cost (4 cycles) covers.  64 consecutive bytes per lane, the
//            first form of this kernel, put sixteen lanes on one bank: it ran at the pace of those conflicts;
//   lengths  a lane sums the output lengths of its 8 bytes per row (a 256-byte table of lengths in LDS);
//   places   the rows' sums two to a word through DPP prefix sums over the wave, the rows' and waves' totals, then a two-level decoupled
//            look-back over the tiles' totals (groups of 64 tiles: one round trip reaches 8 192 tiles back);
//   expand   a lane writes its bytes' texts at their places in a WINDOW of the tile's output in LDS: one byte store per input byte (a byte
//            that prints nothing stores into a sink: a select instead of a branch) and, for the lanes whose byte prints a longer text, the
//            text's other bytes — in as many rounds as the output needs windows (one, as a rule);
//   store    the window leaves as whole aligned 16-byte lines, the two ends of a tile byte by byte.
// The loop is a software pipeline (map_kernels.hip): a tile's bytes are asked for two tiles ahead, its total is out a whole trip before anybody
// looks back at it, it is expanded into one of TWO windows and stored a trip later — when its look-back, whose loads went out a trip before,
// has its answers (and if they are not enough, the second round is hidden behind the next tile's count).  Two barriers per tile.
//
// The input is read once, the output written once, nothing else touches memory but 24 bytes per 16 KiB tile.  A NUL (the rest of its record
// is swallowed: state after all) voids the launch and the general family runs the buffer, as for the byte map.
// Measured (round 6), 8 GiB: '[aie]:' 6.2 ms (1.39 TB/s; the count / emit pair that reads the input twice: 7.6), '(a:b|e:)' 6.5 (7.7),
// HTML escapes 8.4 (11.0), 'a:xyz' — a longer text every 35 bytes — 8.5 (8.0).  On by default for the programs it takes (runtime.cpp: a context whose
// longer texts turn out frequent goes back to the pair); TRRE_MAPGEN=0 / 1: never / always.
// Matches: the scan loops trre_dft.c:1272-1286 / trre_nft.c:775-790 with infer_* deciding after one byte; emits trre_dft.c:1121-1122, trre_nft.c:645.
// The per-lane bodies are TRRE_HD: tests/cpu_shim.cpp runs them lane by lane.
#pragma once
#include "scan_block.hpp"

namespace trre {

constexpr int kMapGenThreads = 256;
constexpr int kMgLaneBytes = 8;                  // bytes of a row per lane
constexpr int kMgRows = 8;                       // rows per wave
constexpr int kMgLanes = 64;                     // lanes of a wave
constexpr int kMgRowBytes = kMgLanes * kMgLaneBytes;
constexpr int kMgWaveBytes = kMgRows * kMgRowBytes;
constexpr int kMapGenTile = (kMapGenThreads / kMgLanes) * kMgWaveBytes;
constexpr uint32_t kMgNul = 0x80u;               // length table: the byte cuts its record short (a NUL): the launch is void

// the tables (StreamTables::mg, 256 x 16 bytes in the blob: {text lo, text hi, length | kMgNul, 0}) as the kernel keeps them in LDS
struct MapGenView {
    const uint8_t* len;      // [256] the text's length (0..8) | kMgNul
    const uint8_t* first;    // [256] the text's first byte
    const uint64_t* text;    // [256] the bytes, zero beyond the length
};

struct MapGenArgs {
    uint64_t* desc;          // [n_tiles] look-back descriptors (one_block.hpp: kind, -, total), zeroed before the launch
    uint64_t* gsum;          // [n_tiles / 64 + 1] per group of 64 tiles: count and sum (zeroed)
    uint64_t* ginc;          // [n_tiles / 64 + 1] the running total at the end of a group (zeroed)
    uint64_t* total;         // [1] the size of the whole output (written by the last tile)
    int64_t n_tiles;
    uint32_t window;         // bytes of ONE of the two output windows in LDS (a multiple of 16)
    uint32_t spin;           // look-back polls before a tile gives up (void launch — never a hang)
    uint32_t first_lookup;   // some byte prints ONE byte that is not itself: the expansion looks every first byte up (else: the byte itself,
                             // and the texts of two bytes and more bring their own)
    uint32_t longest;        // the longest text (StreamTables::mg_max)
    uint64_t* prof;          // TRRE_MAPGEN_PROF=1: [8] shader clocks per phase summed over the trips (thread 0's): [1] the look-back's answers, [2] expand,
                             // [3] the next tile counted and published, [4] store, [5] the last barrier; [7] the trips; else null
    uint64_t* dbg;           // TRRE_MAPGEN_DBG=<file>: [n_tiles][16] every tile's total, place, workgroup and clock (tools/mapgen_diff.py checks them); else null
};

// output bytes of a lane's 8 bytes of a row (lo, hi: the bytes at v .. v + 7, v-space).  kEdge: the row may touch an end of the input — the
// bytes j outside [jlo, jhi) count nothing and byte jnl (the input's last position, vend - 1) is a '\n' (mg_edge: the three from v).  A NUL
// leaves kMgNul in the sum (at most 64 otherwise).
struct MgEdge { int jlo, jhi, jnl; };
TRRE_HD MgEdge mg_edge(int64_t v, int64_t vbeg, int64_t vend) {
    const int64_t a = vbeg - v, b = vend - v;
    return MgEdge{a < 0 ? 0 : (a > 8 ? 8 : (int)a), b < 0 ? 0 : (b > 8 ? 8 : (int)b), b - 1 >= 0 && b - 1 < 8 ? (int)(b - 1) : -1};
}
template <bool kEdge>
TRRE_HD uint32_t mg_count8(const MapGenView& T, uint32_t lo, uint32_t hi, MgEdge e) {
    uint32_t n = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t b = ((j < 4 ? lo : hi) >> (8 * (j & 3))) & 0xffu;
        if (kEdge) {
            if (j < e.jlo || j >= e.jhi) continue;
            if (j == e.jnl) b = (uint32_t)'\n';
        }
        n += T.len[b];
    }
    return n;
}
// the same bytes into the window: win[p - wlo] for the output positions p in [wlo, wlo + wsize) (tile-relative); pos: where the lane's output
// begins, returns where it ends.  kClip = false: the whole tile fits the window (wlo = 0, nothing to test).  kFirst: MapGenArgs::first_lookup;
// kMulti: some text has two bytes or more — the lanes of a wave that meet one write its other bytes, the others wait.
template <bool kFirst, bool kMulti, bool kEdge, bool kClip>
TRRE_HD uint32_t mg_expand8(const MapGenView& T, uint32_t lo, uint32_t hi, MgEdge e, uint8_t* win, uint32_t sink, uint32_t pos, uint32_t wlo, uint32_t wsize) {
    // (the eight lookups first — independent reads —, then the eight stores.  A byte that prints nothing stores into the lane's SINK, a byte
    // of its own behind the window (win[sink]): a select, not a branch — a predicated store is a compare, two scalar operations on the
    // execution mask and the store, and the scalar unit of a CU is one)
    uint32_t bb[8], l[8], f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t b = ((j < 4 ? lo : hi) >> (8 * (j & 3))) & 0xffu;
        bool inside = true;
        if (kEdge) {
            inside = j >= e.jlo && j < e.jhi;
            if (j == e.jnl) b = (uint32_t)'\n';
        }
        bb[j] = b;
        l[j] = inside ? (uint32_t)T.len[b] : 0u;
        f[j] = kFirst ? (uint32_t)T.first[b] : b;
    }
    // (pointers, not indices: the window's address is added once, not per store)
    uint8_t* p = win + (int32_t)(pos - wlo);       // (window-relative; beyond wsize — or before the window — outside it)
    uint8_t* const sk = win + sink;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t x = f[j];
        if (kMulti && l[j] > 1u) {                 // (the lanes that meet a longer text; the others wait — or, no such lane, skip)
            const uint64_t t = T.text[bb[j]];
            const uint32_t t4 = (uint32_t)t;
            x = t4;
            if (!kClip || (uint32_t)(p + 1 - win) < wsize) p[1] = (uint8_t)(t4 >> 8);
            *(l[j] > 2u && (!kClip || (uint32_t)(p + 2 - win) < wsize) ? p + 2 : sk) = (uint8_t)(t4 >> 16);
#pragma clang loop vectorize(disable) unroll(disable)
            for (uint32_t q = 3; q < l[j]; ++q)              // (a text of four bytes and more: rare)
                if (!kClip || (uint32_t)(p + q - win) < wsize) p[q] = (uint8_t)(t >> (8 * q));
        }
        *(l[j] != 0u && (!kClip || (uint32_t)(p - win) < wsize) ? p : sk) = (uint8_t)x;
        p += l[j];
    }
    const uint32_t at = (uint32_t)(p - win);
    return at + wlo;
}

// Window lines to memory.  The window holds the tile's output positions [wlo, wlo + n) from its first byte on (the expansion does not wait for
// the tile's place in the output); output line c is the 16 aligned bytes at (address of out + base + wlo - h) + 16 c, h = that address & 15:
// read from the window at 16 c - h through a funnel shift (aligned dwords), stored whole when all its bytes are this window's, byte by byte
// at the two ends.  write = false: nothing is stored (the output does not fit the caller's buffer).
// kSel: the line's first dword within its 16-byte block of the window, (16 - h) & 15 >> 2 — the same for every line of a tile (the kernel branches on it
// once per window instead of selecting per line); -1: worked out here.
template <int kSel = -1>
TRRE_HD void mg_store_line(const uint8_t* win, uint8_t* out, uint64_t base, uint32_t wlo, uint32_t n, uint32_t h, uint32_t c, bool write) {
    const int32_t lo = (int32_t)(c << 4) - (int32_t)h;   // window offset of the line's first byte (negative: the tile's first line)
    if (lo >= (int32_t)n || !write) return;
    uint8_t* g = out + base + wlo + lo;                  // the line's address: aligned
    if (lo >= 0 && (uint32_t)lo + 16u <= n) {
        // two aligned 16-byte reads (no bank conflicts: a wave's lines lie side by side) and a funnel over their eight dwords
        const U128 x = *reinterpret_cast<const U128*>(win + ((uint32_t)lo & ~15u)), y = *reinterpret_cast<const U128*>(win + ((uint32_t)lo & ~15u) + 16);
        const uint32_t dsel = kSel >= 0 ? (uint32_t)kSel : ((uint32_t)lo >> 2) & 3u, sh = (uint32_t)lo & 3u;
        const uint32_t r0 = dsel == 0 ? x.x : (dsel == 1 ? x.y : (dsel == 2 ? x.z : x.w));
        const uint32_t r1 = dsel == 0 ? x.y : (dsel == 1 ? x.z : (dsel == 2 ? x.w : y.x));
        const uint32_t r2 = dsel == 0 ? x.z : (dsel == 1 ? x.w : (dsel == 2 ? y.x : y.y));
        const uint32_t r3 = dsel == 0 ? x.w : (dsel == 1 ? y.x : (dsel == 2 ? y.y : y.z));
        const uint32_t r4 = dsel == 0 ? y.x : (dsel == 1 ? y.y : (dsel == 2 ? y.z : y.w));
        *reinterpret_cast<U128*>(g) = U128{alignbyte_b32(r1, r0, sh), alignbyte_b32(r2, r1, sh), alignbyte_b32(r3, r2, sh), alignbyte_b32(r4, r3, sh)};
        return;
    }
    const int32_t a = lo < 0 ? 0 : lo, b = lo + 16 < (int32_t)n ? lo + 16 : (int32_t)n;
    for (int32_t k = a; k < b; ++k) out[base + wlo + (uint32_t)k] = win[k];
}

}  // namespace trre
