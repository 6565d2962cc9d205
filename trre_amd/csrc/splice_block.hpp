// splice_block.hpp — the copy form's second pass as a WAVE-COOPERATIVE splice (round 4).
//
// The mark pass (scan_block.hpp: fb_lane<3>) leaves, per 2 KiB sub-range of the input, the list of its edits: where a
// replacement text goes, which text, how many input bytes it stands for.  The first version of the second pass
// (fb_copy_lane) had every lane copy its own sub-range through a staging ring, byte range by byte range: 35 VALU
// instructions per input byte, because some lane of a wave meets an edit in nearly every dword and all lanes pay for the
// edit path.  Here a WAVE takes a sub-range and its 64 lanes work on one window of it at a time — up to
// 2304 input bytes and 128 edits: as a rule the whole sub-range:
//
//   edits     two per lane (two batches of 64): decoded, then prefix sums over the lanes give every text its place in the
//             window's output and every raw run between two texts its displacement D (output position = input position + D);
//   markers   the output dword in which a run begins gets the run's index (one byte per dword), a prefix maximum over
//             the dwords turns that into "the run the dword's first byte belongs to";
//   phase A   output-parallel: a lane assembles 16 output bytes as four funnel shifts of the staged input, each dword with
//             the displacement of its first byte's run — right for all bytes of that run, scratch for the others;
//   phase B   edit-parallel: a lane writes its text over the scratch, byte by byte, and the (at most three) raw bytes
//             between the text's end and the next dword boundary;
//   store     the window's output leaves as whole 16-byte lines of the output buffer (the tile is kept congruent to the
//             output address mod 16; a partial last line is carried into the next window; only the two ends of the
//             sub-range's output, whose lines it shares with its neighbours, go byte by byte).
//
// No per-lane trip counts, no staging rings, no branch on the data but "an escape record in this window" (rare).  The
// input bytes and the edits of a window are requested while the window before it is worked on (across sub-ranges too): the
// passes of a window are a chain of dependent steps, and a wave has nothing else to do while it waits.
//
// The body is written once for the device (a lane per thread, DPP collectives) and for tests/cpu_shim.cpp (the 64 lanes
// as arrays, the collectives as loops): per-lane variables are declared with SPV and touched inside SP_FOR blocks; between
// the blocks control flow is wave-uniform.
#pragma once
#include "scan_block.hpp"

namespace trre {

#ifndef TRRE_SP_WIN
#define TRRE_SP_WIN 2304
#define TRRE_SP_GROW 640
#endif
constexpr uint32_t kSpWin = TRRE_SP_WIN;   // input bytes per window
constexpr uint32_t kSpEdits = 128;         // edits per window: two per lane
constexpr uint32_t kSpIn = (15 + kSpWin + 8 + 15) / 16 * 16 + 16;   // staged input (at tin + 16): at most 15 + the window and the second dword of a funnel read are looked at
constexpr uint32_t kSpGrow = TRRE_SP_GROW; // what the texts of one window may add
constexpr uint32_t kSpOut = (15 + kSpWin + kSpGrow + 15) / 16 * 16;   // tile of output bytes
constexpr uint32_t kSpMk = (kSpOut / 4 + 15) / 16 * 16;               // markers: a byte per output dword
constexpr uint32_t kSpMaxText = 255;       // longest escape text a table may have for this pass (runtime.cpp checks)
// per wave: staged input | output tile | markers | displacement table [129] | the carried line
constexpr uint32_t kSpOffOut = 16 + kSpIn, kSpOffMk = kSpOffOut + kSpOut, kSpOffTab = kSpOffMk + kSpMk,
                   kSpOffCarry = kSpOffTab + 528, kSpLdsPerWave = kSpOffCarry + 16;
static_assert(kSpLdsPerWave % 16 == 0 && (kSpEdits + 1) * 4 <= 528, "per-wave LDS carve");
static_assert(15 + kSpWin + kSpGrow <= kSpOut && 15 + kSpWin + 8 <= kSpIn && kSpOut <= 3 * 1024 && kSpIn <= 3 * 1024, "tile sizes");

#if defined(__HIP_DEVICE_COMPILE__)
#define SPV(T, x) T x
#define SP(x) x
#define SP_FOR
#define SP_LANE ((uint32_t)__lane_id())
// inclusive scans over the 64 lanes: row_shr 1/2/4/8 inside the rows of 16, row_bcast 15/31 across them (gfx9 DPP)
#define SP_DPP_STEP(op, v, ctrl, rows) v = op(v, (decltype(v))__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, 0xf, false))
__device__ __forceinline__ int32_t sp_add(int32_t a, int32_t b) { return a + b; }
__device__ __forceinline__ int32_t sp_max(int32_t a, int32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int32_t sp_scan_add(int32_t v) {
    SP_DPP_STEP(sp_add, v, 0x111, 0xf); SP_DPP_STEP(sp_add, v, 0x112, 0xf); SP_DPP_STEP(sp_add, v, 0x114, 0xf); SP_DPP_STEP(sp_add, v, 0x118, 0xf);
    SP_DPP_STEP(sp_add, v, 0x142, 0xa); SP_DPP_STEP(sp_add, v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ int32_t sp_scan_max(int32_t v) {      // (values >= 0)
    SP_DPP_STEP(sp_max, v, 0x111, 0xf); SP_DPP_STEP(sp_max, v, 0x112, 0xf); SP_DPP_STEP(sp_max, v, 0x114, 0xf); SP_DPP_STEP(sp_max, v, 0x118, 0xf);
    SP_DPP_STEP(sp_max, v, 0x142, 0xa); SP_DPP_STEP(sp_max, v, 0x143, 0xc);
    return v;
}
#define SP_SCAN_ADD(x) x = sp_scan_add(x)
#define SP_SCAN_MAX(x) x = sp_scan_max(x)
#define SP_BALLOT(dst, expr) dst = __ballot(expr)
#define SP_FROM_NEXT(dst, src, fill) dst = __builtin_amdgcn_update_dpp((int)(fill), (int)(src), 0x130, 0xf, 0xf, false)   /* wave_shl:1 */
#define SP_FROM_PREV(dst, src, fill) dst = __builtin_amdgcn_update_dpp((int)(fill), (int)(src), 0x138, 0xf, 0xf, false)   /* wave_shr:1 */
#define SP_BCAST(src, l) __builtin_amdgcn_readlane((int)(src), (int)(l))
#define SP_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define SP_ALIGNBYTE(hi, lo, sh) __builtin_amdgcn_alignbyte(hi, lo, sh)
#define SP_ANY(dst, expr) dst = __any(expr)
#else
#define SPV(T, x) T x[64]
#define SP(x) x[spl_]
#define SP_FOR for (uint32_t spl_ = 0; spl_ < 64u; ++spl_)
#define SP_LANE spl_
#define SP_SCAN_ADD(x) do { for (int i_ = 1; i_ < 64; ++i_) x[i_] += x[i_ - 1]; } while (0)
#define SP_SCAN_MAX(x) do { for (int i_ = 1; i_ < 64; ++i_) x[i_] = x[i_] > x[i_ - 1] ? x[i_] : x[i_ - 1]; } while (0)
#define SP_BALLOT(dst, expr) do { dst = 0; SP_FOR { if (expr) dst |= 1ull << spl_; } } while (0)
#define SP_FROM_NEXT(dst, src, fill) do { for (int i_ = 0; i_ < 63; ++i_) dst[i_] = src[i_ + 1]; dst[63] = (fill); } while (0)
#define SP_FROM_PREV(dst, src, fill) do { for (int i_ = 63; i_ > 0; --i_) dst[i_] = src[i_ - 1]; dst[0] = (fill); } while (0)
#define SP_BCAST(src, l) src[l]
#define SP_WAVE_SYNC() ((void)0)
#define SP_ALIGNBYTE(hi, lo, sh) alignbyte_b32(hi, lo, sh)
#define SP_ANY(dst, expr) do { dst = false; SP_FOR { if (expr) dst = true; } } while (0)
#endif

TRRE_HD uint32_t sp_ctz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return x ? (uint32_t)__ffsll((unsigned long long)x) - 1u : 64u;
#else
    return x ? (uint32_t)__builtin_ctzll(x) : 64u;
#endif
}

// what the edits' texts are looked up in (the copy form of a large table, scan_block.hpp: fb_lane<3> / fb_mark4_lane): an edit names a
// literal {text, length, input bytes it stands for} or an escape record.  (Round 4 also spliced SMALL tables' edits — an edit named the
// 16-byte entry of its transition; removed in round 5: DESIGN.md §4.5a.)
struct SpliceTables {
    const U128* lit = nullptr;          // [fb_lits] {text lo, text hi, n | kb << 8, -}  (LDS), or null: ...
    const uint64_t* lit_text = nullptr; // ... the texts and their {n | kb << 8} from memory (the table stays in L1 / L2: more LDS for tiles)
    const uint16_t* lit_meta = nullptr;
    const uint32_t* esc = nullptr;      // escape records (global) ...
    const uint8_t* pool = nullptr;      // ... and their texts
};

struct SpliceLds {            // one wave's share of the workgroup's LDS (kSpLdsPerWave bytes, 16-byte aligned)
    uint8_t* base;
    TRRE_HD uint8_t* tin() const { return base; }
    TRRE_HD uint8_t* tout() const { return base + kSpOffOut; }
    TRRE_HD uint8_t* mk() const { return base + kSpOffMk; }
    TRRE_HD int32_t* tab() const { return reinterpret_cast<int32_t*>(base + kSpOffTab); }
    TRRE_HD uint8_t* carry() const { return base + kSpOffCarry; }
};

// what a wave is to splice: `count` sub-ranges of the mark pass, lanes (of the mark pass) first, first + stride, ...;
// out_base[k * base_stride]: where the k-th one's output begins (offset into a.out)
struct SpliceWork {
    int64_t first, stride;
    int count;
    const uint64_t* out_base;
    int base_stride;
};

// a sub-range with lines of its own
struct SpliceSub { int k; int64_t lo; uint32_t n_ev, b_rel, end_rel; const uint32_t* evp; };
// the first one at or behind `from` (k == W.count: none)
TRRE_HD SpliceSub splice_open(const ScanArgs& a, const FbCopyArgs& ca, const SpliceWork& W, int64_t lane_bytes, int from) {
    SpliceSub s{};
    for (int k = from; k < W.count; ++k) {
        const int64_t lane = W.first + (int64_t)k * W.stride, lo = lane * lane_bytes;
        if (lo >= a.vend) continue;
        const uint32_t* hdr = ca.lane_hdr + (size_t)lane * 4;
        const uint32_t n_ev = hdr[0], b_rel = hdr[1];
        uint32_t end_rel = hdr[2];
        if (end_rel <= b_rel) continue;                                  // no line starts there
        if (lo + (int64_t)end_rel > a.vend) end_rel = (uint32_t)(a.vend - lo);
        s.k = k; s.lo = lo; s.n_ev = n_ev; s.b_rel = b_rel; s.end_rel = end_rel; s.evp = copy_event_row(ca, lane);
        return s;
    }
    s.k = W.count;
    return s;
}

// a.dbg & 1: no global stores (timing experiments).
// kBatches: edits per lane and window (2: 128 edits per window, the dictionary's density; 1: sparse edits — the window's
// fixed costs, two prefix sums and a phase B per batch, are paid once)
template <int kBatches = 2>
TRRE_HD void fb_splice_ranges(const ScanArgs& a, const SpliceTables& T, const FbCopyArgs& ca, const SpliceWork& W, int64_t lane_bytes,
                              const SpliceLds& L) {
    uint8_t* const tin = L.tin();
    uint8_t* const tout = L.tout();
    uint8_t* const mk = L.mk();
    int32_t* const tab = L.tab();
    uint8_t* const carry = L.carry();
    const uint32_t* tin32 = reinterpret_cast<const uint32_t*>(tin);
    const int64_t vlast = (a.vend - 1) & ~(int64_t)15;                    // the last readable aligned block
    constexpr bool kB2 = kBatches == 2;
    constexpr uint32_t kLimit = 64u * (uint32_t)kBatches;

    // ---- a window's input bytes and raw edits, requested one window ahead -------------------------------------------------
    SPV(uint32_t, praw0); SPV(uint32_t, praw1);                          // raw edits of the two batches (all ones: none)
    SPV(U128, pin0); SPV(U128, pin1); SPV(U128, pin2);                   // input blocks tin0 + 16 (lane + 64 r)
#define SP_REQUEST(s, at, from_ev)                                                                                   \
    do {                                                                                                             \
        const int64_t rq0_ = ((s).lo + (int64_t)(at)) & ~(int64_t)15;                                                \
        SP_FOR {                                                                                                     \
            const uint32_t j0 = (from_ev) + SP_LANE, j1 = j0 + 64u;                                                  \
            SP(praw0) = j0 < (s).n_ev ? (s).evp[j0] : 0xffffffffu;                                                   \
            SP(praw1) = kB2 && j1 < (s).n_ev ? (s).evp[j1] : 0xffffffffu;                                            \
            const int64_t v0 = rq0_ + 16 * (int64_t)SP_LANE, v1 = v0 + 1024, v2 = v0 + 2048;                         \
            SP(pin0) = *reinterpret_cast<const U128*>(a.in_v0 + (v0 < vlast ? v0 : vlast));                          \
            SP(pin1) = *reinterpret_cast<const U128*>(a.in_v0 + (v1 < vlast ? v1 : vlast));                          \
            SP(pin2) = *reinterpret_cast<const U128*>(a.in_v0 + (v2 < vlast ? v2 : vlast));                          \
        }                                                                                                            \
    } while (0)

    SpliceSub cur = splice_open(a, ca, W, lane_bytes, 0);
    if (cur.k >= W.count) return;
    uint32_t pos = cur.b_rel, skip = 0, ei = 0;   // next input byte, bytes of it an earlier text stands for, next edit
    uint8_t* gout = a.out + W.out_base[(size_t)cur.k * W.base_stride];    // where the next output byte goes
    bool head_open = true;                        // the first line of the sub-range's output is shared with the sub-range before:
    uint32_t head0 = (uint32_t)(reinterpret_cast<uintptr_t>(gout) & 15u);   // ... its first head0 bytes are not ours
    bool carried = false;                         // the tile's first bytes come from the window before
    SP_REQUEST(cur, pos, ei);

    for (;;) {
        const int64_t lo = cur.lo;
        const uint32_t end_rel = cur.end_rel;
        const uint32_t w0 = pos;
        uint32_t w1 = w0 + kSpWin < end_rel ? w0 + kSpWin : end_rel;
        // ---- stage the input that was requested a window ago, clear the markers ------------------------------------------
        const int64_t tin0 = (lo + (int64_t)w0) & ~(int64_t)15;                          // v of tin[16]
        {
            const uint32_t need = (uint32_t)(lo + (int64_t)w1 - tin0) + 8u;              // staged bytes that may be looked at
            bool edge;
            SP_ANY(edge, tin0 + 16 * (int64_t)SP_LANE < a.vbeg || tin0 + 16 * (int64_t)SP_LANE + 2048 + 16 > a.vend - 1);
            if (edge) {                                                                  // the ends of the input: filler, the last byte reads as '\n'
                SP_FOR {
                    const int64_t v0 = tin0 + 16 * (int64_t)SP_LANE;
                    SP(pin0) = direct_load(a, v0); SP(pin1) = direct_load(a, v0 + 1024); SP(pin2) = direct_load(a, v0 + 2048);
                }
            }
            SP_FOR {
                *reinterpret_cast<U128*>(tin + 16u + 16u * SP_LANE) = SP(pin0);
                if (need > 1024u) *reinterpret_cast<U128*>(tin + 16u + 1024u + 16u * SP_LANE) = SP(pin1);
                if (need > 2048u && 2048u + 16u * SP_LANE < kSpIn) *reinterpret_cast<U128*>(tin + 16u + 2048u + 16u * SP_LANE) = SP(pin2);
                if (SP_LANE < kSpMk / 16u) reinterpret_cast<U128*>(mk)[SP_LANE] = U128{0, 0, 0, 0};
            }
        }
        // ---- the window's edits, two per lane -----------------------------------------------------------------------
        SPV(uint32_t, fp0); SPV(uint32_t, n0); SPV(uint32_t, kb0); SPV(uint32_t, tlo0); SPV(uint32_t, thi0); SPV(uint32_t, esc0);
        SPV(uint32_t, fp1); SPV(uint32_t, n1); SPV(uint32_t, kb1); SPV(uint32_t, tlo1); SPV(uint32_t, thi1); SPV(uint32_t, esc1);
        SPV(int32_t, cum0); SPV(int32_t, cum1);
#define SP_DECODE(raw, fp, n, kb, tlo, thi, ex, cum)                                                               \
        SP(fp) = 0xffffffffu; SP(n) = 0; SP(kb) = 0; SP(tlo) = 0; SP(thi) = 0; SP(ex) = 0;                      \
        if (SP(raw) != 0xffffffffu) {                                                                             \
            const uint32_t id = SP(raw) >> 16, p = SP(raw) & 0xffffu;                                             \
            if (!(id & 0x8000u)) {                                                                                \
                U128 r;                                                                                           \
                if (T.lit) r = T.lit[id];                                                                         \
                else { const uint64_t tx = T.lit_text[id]; r.x = (uint32_t)tx; r.y = (uint32_t)(tx >> 32); r.z = T.lit_meta[id]; r.w = 0; } \
                SP(tlo) = r.x; SP(thi) = r.y; SP(n) = r.z & 255u; SP(kb) = r.z >> 8; SP(fp) = p - SP(kb);          \
            } else {                                      /* a text spelled out in memory (rare) */               \
                const uint32_t* r = T.esc + 4u * (id & 0x7fffu);                                                  \
                SP(ex) = 1u + r[0]; SP(n) = r[1]; SP(kb) = r[3] & 255u; SP(fp) = p - (r[3] >> 8);                  \
            }                                                                                                     \
        }                                                                                                         \
        SP(cum) = (SP(fp) < w1 && SP(n) > SP(kb)) ? (int32_t)(SP(n) - SP(kb)) : 0;
        SP_FOR {
            SP_DECODE(praw0, fp0, n0, kb0, tlo0, thi0, esc0, cum0)
            if (kB2) { SP_DECODE(praw1, fp1, n1, kb1, tlo1, thi1, esc1, cum1) }
            else { SP(fp1) = 0xffffffffu; SP(n1) = 0; SP(kb1) = 0; SP(tlo1) = 0; SP(thi1) = 0; SP(esc1) = 0; SP(cum1) = 0; }
        }
#undef SP_DECODE
        SP_SCAN_ADD(cum0);
        uint64_t ok0, ok1 = 0, in0, in1 = 0;
        SP_BALLOT(in0, SP(fp0) < w1);
        SP_BALLOT(ok0, SP(fp0) < w1 && SP(cum0) <= (int32_t)kSpGrow);
        if (kB2) {
            SP_SCAN_ADD(cum1);
            const int32_t grow0 = (int32_t)SP_BCAST(cum0, 63);
            SP_BALLOT(in1, SP(fp1) < w1);
            SP_BALLOT(ok1, SP(fp1) < w1 && grow0 + SP(cum1) <= (int32_t)kSpGrow);
        }
        uint32_t m = sp_ctz64(~ok0);                                       // the edits this window takes: the first m of the kLimit
        if (kB2 && m == 64u) m += sp_ctz64(~ok1);
        const uint32_t m0 = m < 64u ? m : 64u, m1 = m - m0;               // ... of the first / second batch
        if (m < kLimit) {
            const bool next_in = m < 64u ? (in0 >> m) & 1u : (in1 >> (m - 64u)) & 1u;
            if (next_in) w1 = m < 64u ? (uint32_t)SP_BCAST(fp0, m) : (uint32_t)SP_BCAST(fp1, m - 64u);   // in the window but does not fit: end before it
        } else {
            // (there may be more edits in the window than a window takes: end where the last one taken ends — the next begins no earlier)
            const uint32_t r_end = kB2 ? (uint32_t)SP_BCAST(fp1, 63) + (uint32_t)SP_BCAST(kb1, 63) : (uint32_t)SP_BCAST(fp0, 63) + (uint32_t)SP_BCAST(kb0, 63);
            if (r_end < w1) w1 = r_end;
        }
        const bool last = w1 >= end_rel;
        // ---- what comes after this window: ask for it now ------------------------------------------------------------------
        SpliceSub nxt = cur;
        uint32_t npos = w1, nei = ei + m;
        if (last) { nxt = splice_open(a, ca, W, lane_bytes, cur.k + 1); npos = nxt.b_rel; nei = 0; }
        const bool more = nxt.k < W.count;
        if (more) SP_REQUEST(nxt, npos, nei);
        // ---- where everything goes ----------------------------------------------------------------------------------
        // run j = the raw bytes in front of text j (run m: behind the last text); an input byte x of run j lands at x + D_j
        SPV(int32_t, d0); SPV(int32_t, d1); SPV(int32_t, S0); SPV(int32_t, S1); SPV(int32_t, P0); SPV(int32_t, P1); SPV(int32_t, C0); SPV(int32_t, C1);
        SP_FOR {
            SP(d0) = SP_LANE < m0 ? (int32_t)SP(n0) - (int32_t)SP(kb0) : 0; SP(S0) = SP(d0);
            SP(d1) = SP_LANE < m1 ? (int32_t)SP(n1) - (int32_t)SP(kb1) : 0; SP(S1) = SP(d1);
        }
        SP_SCAN_ADD(S0);
        if (kB2) SP_SCAN_ADD(S1);
        const int32_t sum0 = (int32_t)SP_BCAST(S0, 63), sum1 = kB2 ? (int32_t)SP_BCAST(S1, 63) : 0;
        const int32_t D0 = -(int32_t)(w0 + skip);
        const int32_t Dm = D0 + sum0 + sum1;
        uint32_t rm = w0 + skip;
        if (kB2 && m1) rm = (uint32_t)SP_BCAST(fp1, m1 - 1) + (uint32_t)SP_BCAST(kb1, m1 - 1);
        else if (m0) rm = (uint32_t)SP_BCAST(fp0, m0 - 1) + (uint32_t)SP_BCAST(kb0, m0 - 1);
        const uint32_t top = w1 > rm ? w1 : rm;
        const uint32_t out_len = (uint32_t)((int32_t)top + Dm);
        const uint32_t skip_out = top - w1;
        const uint32_t oa = (uint32_t)(reinterpret_cast<uintptr_t>(gout) & 15u);       // tile index of output position 0
        const int32_t cbase = 16 + (int32_t)(lo - tin0) - (int32_t)oa;                  // tin index = tile index + cbase - D (the input is staged
                                                                                         // 16 bytes in: a funnel read may begin one dword before it)
        SP_FOR {
            const int32_t Dj0 = D0 + SP(S0) - SP(d0), Dj1 = D0 + sum0 + SP(S1) - SP(d1);
            SP(P0) = (int32_t)SP(fp0) + Dj0;                                             // output position of the text
            SP(P1) = (int32_t)SP(fp1) + Dj1;
            SP(C0) = cbase - Dj0;
            SP(C1) = cbase - Dj1;
            if (SP_LANE < m0) tab[SP_LANE] = SP(C0);
            if (kB2 && SP_LANE < m1) tab[64u + SP_LANE] = SP(C1);
            if (SP_LANE == (m & 63u)) tab[m] = cbase - Dm;
        }
        // the dword whose FIRST byte is at or behind the start of text j belongs to run j + 1 (or a later one) from there on
        SPV(uint32_t, q0); SPV(uint32_t, q1); SPV(uint32_t, qn0); SPV(uint32_t, qn1);
        SP_FOR {
            SP(q0) = SP_LANE < m0 ? (oa + (uint32_t)SP(P0) + 3u) >> 2 : 0xffffffffu;
            SP(q1) = SP_LANE < m1 ? (oa + (uint32_t)SP(P1) + 3u) >> 2 : 0xffffffffu;
        }
        const uint32_t q1_first = kB2 ? (uint32_t)SP_BCAST(q1, 0) : 0xffffffffu;
        SP_FROM_NEXT(qn0, q0, q1_first);
        SP_FOR { if (SP_LANE < m0 && SP(qn0) != SP(q0)) mk[SP(q0)] = (uint8_t)(SP_LANE + 1u); }
        SP_WAVE_SYNC();
        if (kB2) {
            SP_FROM_NEXT(qn1, q1, 0xffffffffu);
            SP_FOR { if (SP_LANE < m1 && SP(qn1) != SP(q1)) mk[SP(q1)] = (uint8_t)(SP_LANE + 65u); }
            SP_WAVE_SYNC();
        }
        // ---- phase A: 16 output bytes per lane and round ---------------------------------------------------------------
        const uint32_t total = oa + out_len;
        const uint32_t rounds = (total + 1023u) >> 10;
        int32_t before = 0;                                                              // the largest marker of the rounds before
        for (uint32_t rd = 0; rd < rounds; ++rd) {
            SPV(int32_t, k0); SPV(int32_t, k1); SPV(int32_t, k2); SPV(int32_t, k3); SPV(int32_t, inc); SPV(int32_t, exc);
            SP_FOR {
                const uint32_t ci = rd * 64u + SP_LANE;
                const uint32_t mk4 = ci < kSpOut / 16u ? reinterpret_cast<const uint32_t*>(mk)[ci] : 0u;
                SP(k0) = (int32_t)(mk4 & 255u);
                const int32_t b1 = (int32_t)((mk4 >> 8) & 255u), b2 = (int32_t)((mk4 >> 16) & 255u), b3 = (int32_t)(mk4 >> 24);
                SP(k1) = SP(k0) > b1 ? SP(k0) : b1;
                SP(k2) = SP(k1) > b2 ? SP(k1) : b2;
                SP(k3) = SP(k2) > b3 ? SP(k2) : b3;
                SP(inc) = SP(k3);
            }
            SP_SCAN_MAX(inc);
            SP_FROM_PREV(exc, inc, 0);
            SP_FOR {
                const int32_t e = SP(exc) > before ? SP(exc) : before;
                const int32_t r0 = SP(k0) > e ? SP(k0) : e, r1 = SP(k1) > e ? SP(k1) : e, r2 = SP(k2) > e ? SP(k2) : e,
                              r3 = SP(k3) > e ? SP(k3) : e;
                const int32_t at = (int32_t)(rd * 1024u + 16u * SP_LANE);
                const int32_t rr[4] = {r0, r1, r2, r3};
                uint32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int32_t s = at + 4 * k + tab[rr[k]];
                    s = s < 0 ? 0 : (s > (int32_t)kSpIn + 8 ? (int32_t)kSpIn + 8 : s);   // (scratch dwords may point anywhere)
                    const uint32_t lo32 = tin32[s >> 2], hi32 = tin32[(s >> 2) + 1];
                    w[k] = SP_ALIGNBYTE(hi32, lo32, (uint32_t)s & 3u);
                }
                if ((uint32_t)at < total) *reinterpret_cast<U128*>(tout + at) = U128{w[0], w[1], w[2], w[3]};
            }
            before = (int32_t)SP_BCAST(inc, 63) > before ? (int32_t)SP_BCAST(inc, 63) : before;
        }
        SP_WAVE_SYNC();
        // ---- the line carried over from the window before, then phase B: the texts -------------------------------------
        SPV(int32_t, Pn0); SPV(int32_t, Pn1);
        const int32_t P1_first = kB2 ? (int32_t)SP_BCAST(P1, 0) : 0;
        SP_FROM_NEXT(Pn0, P0, P1_first);
        uint64_t escmask0, escmask1 = 0;
        SP_BALLOT(escmask0, SP_LANE < m0 && SP(esc0) != 0u);
        if (kB2) {
            SP_FROM_NEXT(Pn1, P1, 0);
            SP_BALLOT(escmask1, SP_LANE < m1 && SP(esc1) != 0u);
        }
#define SP_TEXT(mb, idx0, P, Pn, n, tlo, thi, esc, Cj, d)                                                              \
        if (SP_LANE < (mb)) {                                                                                         \
            uint8_t* t = tout + oa + (uint32_t)SP(P);                                                                 \
            if (!SP(esc)) {                                                                                           \
                _Pragma("unroll") for (uint32_t i = 0; i < 8u; ++i)                                                   \
                    if (i < SP(n)) t[i] = (uint8_t)((i < 4u ? SP(tlo) >> (8u * i) : SP(thi) >> (8u * (i - 4u))));     \
            }                                                                                                         \
            /* the raw bytes between the text's end and the next dword boundary (or the next text) belong to run j + 1 */ \
            const uint32_t be = oa + (uint32_t)SP(P) + SP(n);                                                         \
            const uint32_t next_at = (idx0) + SP_LANE + 1u < m ? (uint32_t)SP(Pn) : out_len;                          \
            uint32_t cnt = (4u - (be & 3u)) & 3u;                                                                     \
            const uint32_t room = oa + next_at - be;                                                                  \
            if (room < cnt) cnt = room;                                                                               \
            const int32_t cn = SP(Cj) - SP(d);              /* run j + 1: displaced by what text j added */           \
            _Pragma("unroll") for (uint32_t i = 0; i < 3u; ++i) {                                                     \
                int32_t s = (int32_t)(be + i) + cn;                                                                   \
                s = s < 0 ? 0 : (s > (int32_t)kSpIn + 15 ? (int32_t)kSpIn + 15 : s);                                  \
                if (i < cnt) tout[be + i] = tin[s];                                                                   \
            }                                                                                                         \
        }
        SP_FOR {
            if (carried && SP_LANE < oa) tout[SP_LANE] = carry[SP_LANE];
            SP_TEXT(m0, 0u, P0, Pn0, n0, tlo0, thi0, esc0, C0, d0)
        }
        if (kB2 && m1) {
            SP_WAVE_SYNC();
            SP_FOR { SP_TEXT(m1, 64u, P1, Pn1, n1, tlo1, thi1, esc1, C1, d1) }
        }
#undef SP_TEXT
        if (escmask0 | escmask1) {                                                       // texts from memory (rare; they may be longer than 8 bytes)
            SP_WAVE_SYNC();
            SP_FOR {
                if (SP_LANE < m0 && SP(esc0)) {
                    const uint8_t* text = T.pool + (SP(esc0) - 1u);
                    uint8_t* t = tout + oa + (uint32_t)SP(P0);
                    for (uint32_t i = 0; i < SP(n0); ++i) t[i] = text[i];
                }
                if (kB2 && SP_LANE < m1 && SP(esc1)) {
                    const uint8_t* text = T.pool + (SP(esc1) - 1u);
                    uint8_t* t = tout + oa + (uint32_t)SP(P1);
                    for (uint32_t i = 0; i < SP(n1); ++i) t[i] = text[i];
                }
            }
        }
        SP_WAVE_SYNC();
        // ---- the tile leaves: whole lines of the output buffer -------------------------------------------------------
        uint8_t* const g0 = gout - oa;                                                   // 16-byte aligned
        const uint32_t n_full = total >> 4, tail = total & 15u;
        const bool stores = !(a.dbg & 1u);
        for (uint32_t rd = 0; rd < rounds; ++rd) {
            SP_FOR {
                const uint32_t c = rd * 64u + SP_LANE;
                if (c < n_full && !(c == 0u && head_open) && stores)
                    *reinterpret_cast<U128*>(g0 + 16u * c) = *reinterpret_cast<const U128*>(tout + 16u * c);
            }
        }
        SP_FOR {
            const uint32_t t = SP_LANE;
            if (t < 16u && stores) {
                // the sub-range's first line: only the bytes that are ours
                if (head_open && n_full > 0u && t >= head0) g0[t] = tout[t];
                // its last line
                const uint32_t from = (n_full == 0u && head_open) ? head0 : 0u;
                if (last && t < tail && t >= from) g0[16u * n_full + t] = tout[16u * n_full + t];
            }
            if (t < 16u && !last && t < tail) carry[t] = tout[16u * n_full + t];
        }
        SP_WAVE_SYNC();
        if (!more) break;
        if (last) {                                                                      // on to the next sub-range
            cur = nxt;
            pos = cur.b_rel; skip = 0; ei = 0;
            gout = a.out + W.out_base[(size_t)cur.k * W.base_stride];
            head_open = true;
            head0 = (uint32_t)(reinterpret_cast<uintptr_t>(gout) & 15u);
            carried = false;
        } else {
            if (n_full > 0u) head_open = false;
            carried = true;
            gout += out_len;
            pos = w1;
            skip = skip_out;
            ei += m;
        }
    }
#undef SP_REQUEST
}

}  // namespace trre
