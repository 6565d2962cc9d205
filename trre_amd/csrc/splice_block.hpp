// splice_block.hpp — the copy form's second pass as a WAVE-COOPERATIVE splice (round 4).
//
// The mark pass (scan_block.hpp: fb_lane<3>) leaves, per 2 KiB sub-range of the input, the list of its edits: where a
// replacement text goes, which text, how many input bytes it stands for.  The first version of the second pass
// (fb_copy_lane) had every lane copy its own sub-range through a staging ring, byte range by byte range: 35 VALU
// instructions per input byte, because some lane of a wave meets an edit in nearly every dword and all lanes pay for the
// edit path.  Here a WAVE takes a sub-range and its 64 lanes work on one window of it at a time
// (up to 1000 input bytes and 64 edits):
//
//   edits     one per lane: decoded, then two prefix sums over the lanes give every text its place in the window's
//             output and every raw run between two texts its displacement D (output position = input position + D);
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
// No per-lane trip counts, no staging rings, no branch on the data but "an escape record in this window" (rare): ~230 wave
// instructions per window, 13-15 lane instructions per input byte.
//
// The body is written once for the device (a lane per thread, DPP collectives) and for tests/cpu_shim.cpp (the 64 lanes
// as arrays, the collectives as loops): per-lane variables are declared with SPV and touched inside SP_FOR blocks; between
// the blocks control flow is wave-uniform.
#pragma once
#include "scan_block.hpp"

namespace trre {

constexpr uint32_t kSpWin = 944;           // input bytes per window (with what the texts add and the carried bytes: one round of phase A as a rule)
constexpr uint32_t kSpIn = 1024;           // staged input: 16-byte aligned start, at tin + 16
constexpr uint32_t kSpGrow = 512;          // what the texts of one window may add
constexpr uint32_t kSpOut = 1536;          // tile of output bytes: 15 + 1000 + 512, rounded up
constexpr uint32_t kSpMaxText = 255;       // longest escape text a table may have for this pass (runtime.cpp checks)
// per wave: staged input | output tile | markers | displacement table [65] | the carried line
constexpr uint32_t kSpOffOut = kSpIn + 16, kSpOffMk = kSpOffOut + kSpOut, kSpOffTab = kSpOffMk + kSpOut / 4,
                   kSpOffCarry = kSpOffTab + 272, kSpLdsPerWave = kSpOffCarry + 16;
static_assert(kSpLdsPerWave % 16 == 0, "per-wave LDS carve");

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
#endif

TRRE_HD uint32_t sp_ctz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return x ? (uint32_t)__ffsll((unsigned long long)x) - 1u : 64u;
#else
    return x ? (uint32_t)__builtin_ctzll(x) : 64u;
#endif
}

struct SpliceLds {            // one wave's share of the workgroup's LDS (kSpLdsPerWave bytes, 16-byte aligned)
    uint8_t* base;
    TRRE_HD uint8_t* tin() const { return base; }
    TRRE_HD uint8_t* tout() const { return base + kSpOffOut; }
    TRRE_HD uint8_t* mk() const { return base + kSpOffMk; }
    TRRE_HD int32_t* tab() const { return reinterpret_cast<int32_t*>(base + kSpOffTab); }
    TRRE_HD uint8_t* carry() const { return base + kSpOffCarry; }
};

// One sub-range of the mark pass (`lane` is the mark pass's lane index), spliced by one wave.  out_base: where the
// sub-range's output begins (offset into a.out).  dbg & 1: no global stores (timing experiments).
template <class Dummy = void>
TRRE_HD void fb_splice_range(const ScanArgs& a, const FbCopyTables& T, const FbCopyArgs& ca, int64_t lane, int64_t lane_bytes,
                             uint64_t out_base, const SpliceLds& L) {
    const int64_t lo = lane * lane_bytes;
    if (lo >= a.vend) return;
    const uint32_t* hdr = ca.lane_hdr + (size_t)lane * 4;
    const uint32_t n_ev = hdr[0], b_rel = hdr[1];
    uint32_t end_rel = hdr[2];
    if (end_rel <= b_rel) return;                                         // no line starts here: nothing to copy
    if (lo + (int64_t)end_rel > a.vend) end_rel = (uint32_t)(a.vend - lo);
    const uint32_t* evp = copy_event_row(ca, lane);
    uint8_t* const tin = L.tin();
    uint8_t* const tout = L.tout();
    uint8_t* const mk = L.mk();
    int32_t* const tab = L.tab();
    uint8_t* const carry = L.carry();
    const uint32_t* tin32 = reinterpret_cast<const uint32_t*>(tin);

    uint32_t pos = b_rel, skip = 0, ei = 0;       // next input byte, bytes of it an earlier text stands for, next edit
    uint8_t* gout = a.out + out_base;             // where the next output byte goes
    bool head_open = true;                        // the first line of the sub-range's output is shared with the sub-range before:
    uint32_t head0 = (uint32_t)(reinterpret_cast<uintptr_t>(gout) & 15u);   // ... its first head0 bytes are not ours
    bool carried = false;                         // the tile's first bytes come from the window before

    while (pos < end_rel) {
        const uint32_t w0 = pos;
        uint32_t w1 = w0 + kSpWin < end_rel ? w0 + kSpWin : end_rel;
        // ---- the window's edits, one per lane -------------------------------------------------------------------
        SPV(uint32_t, fp); SPV(uint32_t, n); SPV(uint32_t, kb); SPV(uint32_t, tlo); SPV(uint32_t, thi); SPV(uint32_t, esc);
        SPV(int32_t, cum);
        SP_FOR {
            const uint32_t j = ei + SP_LANE;
            SP(fp) = 0xffffffffu; SP(n) = 0; SP(kb) = 0; SP(tlo) = 0; SP(thi) = 0; SP(esc) = 0;
            if (j < n_ev) {
                const uint32_t raw = evp[j], id = raw >> 16, p = raw & 0xffffu;
                if (!(id & 0x8000u)) {
                    const U128 r = T.lit[id];
                    SP(tlo) = r.x; SP(thi) = r.y;
                    SP(n) = r.z & 255u;
                    SP(kb) = r.z >> 8;
                    SP(fp) = p - SP(kb);
                } else {                                                   // a text spelled out in memory (rare)
                    const uint32_t* r = T.esc + 4u * (id & 0x7fffu);
                    SP(esc) = 1u + r[0];
                    SP(n) = r[1];
                    SP(kb) = r[3] & 255u;
                    SP(fp) = p - (r[3] >> 8);
                }
            }
            SP(cum) = (SP(fp) < w1 && SP(n) > SP(kb)) ? (int32_t)(SP(n) - SP(kb)) : 0;
        }
        SP_SCAN_ADD(cum);
        uint64_t okmask, inmask;
        SP_BALLOT(inmask, SP(fp) < w1);
        SP_BALLOT(okmask, SP(fp) < w1 && SP(cum) <= (int32_t)kSpGrow);
        const uint32_t m = sp_ctz64(~okmask);                              // the edits this window takes: lanes 0..m-1
        if (m < 64u) {
            if ((inmask >> m) & 1u) w1 = (uint32_t)SP_BCAST(fp, m);        // the next one is in the window but does not fit: end before it
        } else {
            // (there may be more edits in the window than lanes: end where the last one taken ends — the next begins no earlier)
            const uint32_t r64 = (uint32_t)SP_BCAST(fp, 63) + (uint32_t)SP_BCAST(kb, 63);
            if (r64 < w1) w1 = r64;
        }
        // ---- where everything goes ----------------------------------------------------------------------------------
        // run j = the raw bytes in front of text j (run m: behind the last text); an input byte x of run j lands at x + D_j
        SPV(int32_t, d); SPV(int32_t, S); SPV(int32_t, P); SPV(int32_t, Cj);
        SP_FOR { SP(d) = SP_LANE < m ? (int32_t)SP(n) - (int32_t)SP(kb) : 0; SP(S) = SP(d); }
        SP_SCAN_ADD(S);
        const int32_t D0 = -(int32_t)(w0 + skip);
        const int32_t Dm = D0 + (int32_t)SP_BCAST(S, 63);
        const uint32_t rm = m ? (uint32_t)SP_BCAST(fp, m - 1) + (uint32_t)SP_BCAST(kb, m - 1) : w0 + skip;
        const uint32_t top = w1 > rm ? w1 : rm;
        const uint32_t out_len = (uint32_t)((int32_t)top + Dm);
        const uint32_t skip_out = top - w1;
        const uint32_t oa = (uint32_t)(reinterpret_cast<uintptr_t>(gout) & 15u);       // tile index of output position 0
        const int64_t tin0 = (lo + (int64_t)w0) & ~(int64_t)15;                          // v of tin[0]
        const int32_t cbase = 16 + (int32_t)(lo - tin0) - (int32_t)oa;                  // tin index = tile index + cbase - D (the input is staged
                                                                                         // 16 bytes in: a funnel read may begin one dword before it)
        SP_FOR {
            const int32_t Dj = D0 + SP(S) - SP(d);
            SP(P) = (int32_t)SP(fp) + Dj;                                                // output position of text j
            SP(Cj) = cbase - Dj;
            if (SP_LANE < m) tab[SP_LANE] = SP(Cj);
            if (SP_LANE == m || (SP_LANE == 63u && m == 64u)) tab[m] = cbase - Dm;
        }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(SP_TRACE)
        fprintf(stderr, "win lane %lld w0 %u w1 %u skip %u ei %u m %u out_len %u oa %u Dm %d rm %u head_open %d\n", (long long)lane, w0, w1, skip, ei, m, out_len, oa, Dm, rm, (int)head_open);
#endif
        // ---- stage the input, clear the markers ------------------------------------------------------------------
        SP_FOR {
            *reinterpret_cast<U128*>(tin + 16u + 16u * SP_LANE) = direct_load(a, tin0 + 16 * (int64_t)SP_LANE);
            if (SP_LANE < kSpOut / 32u) reinterpret_cast<uint64_t*>(mk)[SP_LANE] = 0;
        }
        SP_WAVE_SYNC();
        // the dword whose FIRST byte is at or behind the start of text j belongs to run j + 1 (or a later one) from there on
        SPV(uint32_t, q); SPV(uint32_t, qn);
        SP_FOR { SP(q) = SP_LANE < m ? (oa + (uint32_t)SP(P) + 3u) >> 2 : 0xffffffffu; }
        SP_FROM_NEXT(qn, q, 0xffffffffu);
        SP_FOR { if (SP_LANE < m && SP(qn) != SP(q)) mk[SP(q)] = (uint8_t)(SP_LANE + 1u); }
        SP_WAVE_SYNC();
        // ---- phase A: 16 output bytes per lane --------------------------------------------------------------------
        const uint32_t total = oa + out_len;
        const uint32_t rounds = (total + 1023u) >> 10;
        int32_t before = 0;                                                              // the largest marker of the rounds before
        for (uint32_t rd = 0; rd < rounds; ++rd) {
            SPV(int32_t, m0); SPV(int32_t, m1); SPV(int32_t, m2); SPV(int32_t, m3); SPV(int32_t, inc); SPV(int32_t, exc);
            SP_FOR {
                const uint32_t ci = rd * 64u + SP_LANE;
                const uint32_t mk4 = ci < kSpOut / 16u ? reinterpret_cast<const uint32_t*>(mk)[ci] : 0u;
                SP(m0) = (int32_t)(mk4 & 255u);
                const int32_t b1 = (int32_t)((mk4 >> 8) & 255u), b2 = (int32_t)((mk4 >> 16) & 255u), b3 = (int32_t)(mk4 >> 24);
                SP(m1) = SP(m0) > b1 ? SP(m0) : b1;
                SP(m2) = SP(m1) > b2 ? SP(m1) : b2;
                SP(m3) = SP(m2) > b3 ? SP(m2) : b3;
                SP(inc) = SP(m3);
            }
            SP_SCAN_MAX(inc);
            SP_FROM_PREV(exc, inc, 0);
            SP_FOR {
                const int32_t e = SP(exc) > before ? SP(exc) : before;
                const int32_t r0 = SP(m0) > e ? SP(m0) : e, r1 = SP(m1) > e ? SP(m1) : e, r2 = SP(m2) > e ? SP(m2) : e,
                              r3 = SP(m3) > e ? SP(m3) : e;
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
        SPV(int32_t, Pn);
        SP_FROM_NEXT(Pn, P, 0);
        uint64_t escmask;
        SP_BALLOT(escmask, SP_LANE < m && SP(esc) != 0u);
        SP_FOR {
            if (carried && SP_LANE < oa) tout[SP_LANE] = carry[SP_LANE];
            if (SP_LANE < m) {
                uint8_t* t = tout + oa + (uint32_t)SP(P);
                if (!SP(esc)) {
#pragma unroll
                    for (uint32_t i = 0; i < 8u; ++i)
                        if (i < SP(n)) t[i] = (uint8_t)((i < 4u ? SP(tlo) >> (8u * i) : SP(thi) >> (8u * (i - 4u))));
                }
                // the raw bytes between the text's end and the next dword boundary (or the next text) belong to run j + 1
                const uint32_t be = oa + (uint32_t)SP(P) + SP(n);
                const uint32_t next_at = SP_LANE + 1u < m ? (uint32_t)SP(Pn) : out_len;
                uint32_t cnt = (4u - (be & 3u)) & 3u;
                const uint32_t room = oa + next_at - be;
                if (room < cnt) cnt = room;
                const int32_t cn = SP(Cj) - SP(d);                                       // run j + 1: displaced by what text j added
#pragma unroll
                for (uint32_t i = 0; i < 3u; ++i) {
                    int32_t s = (int32_t)(be + i) + cn;
                    s = s < 0 ? 0 : (s > (int32_t)kSpIn + 15 ? (int32_t)kSpIn + 15 : s);
                    if (i < cnt) tout[be + i] = tin[s];
                }
            }
        }
        if (escmask) {                                                                   // texts from memory (rare; they may be longer than 8 bytes)
            SP_WAVE_SYNC();
            SP_FOR {
                if (SP_LANE < m && SP(esc)) {
                    const uint8_t* text = T.pool + (SP(esc) - 1u);
                    uint8_t* t = tout + oa + (uint32_t)SP(P);
                    for (uint32_t i = 0; i < SP(n); ++i) t[i] = text[i];
                }
            }
        }
        SP_WAVE_SYNC();
#if !defined(__HIP_DEVICE_COMPILE__) && defined(SP_TRACE)
        fprintf(stderr, "  tile: tab0 %d tin[12..20) %02x %02x %02x %02x %02x %02x %02x %02x  tout[0..4) %02x %02x %02x %02x\n", tab[0], tin[12], tin[13], tin[14], tin[15], tin[16], tin[17], tin[18], tin[19], tout[0], tout[1], tout[2], tout[3]);
#endif
        // ---- the tile leaves: whole lines of the output buffer -------------------------------------------------------
        uint8_t* const g0 = gout - oa;                                                   // 16-byte aligned
        const uint32_t n_full = total >> 4, tail = total & 15u;
        const bool last = w1 >= end_rel;
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
        if (n_full > 0u) head_open = false;
        carried = true;
        gout += out_len;
        pos = w1;
        skip = skip_out;
        ei += m;
    }
}

}  // namespace trre
