// patch_block.hpp — the "record + patch" form of the general (variable-length output) families.
//
// The count / emit pair walks the input twice and the emit walk appends its output byte by byte through a per-lane
// staging ring: 24 (small tables) to 70 (large tables) VALU instructions per input byte, the slowest kernels of the
// engine.  But a scan's output is the input with EDITS: between two matches the bytes are copied.
// So the walk is done once, by the RECORD pass (g16_lane<3> / fb_lane<3> in scan_block.hpp: the count walk plus a
// list of the transitions that do not simply emit the byte they read), and the output is produced by a PATCH pass
// that does not walk the transducer at all — every 64-byte piece of the input is patched independently:
//
//   record   per piece q (64 input bytes at v-space offset 64 q) a 32-byte SLOT:
//              word 0   [15:0] delta = bytes the piece's edits add (signed)   [23:16] number of edits (255: see below)
//                       [24] the slot was written
//              1..7     the edits, in input order:  [6:0] p = input bytes of the piece consumed up to and including the
//                       edit (1..64)   [8:7] kind   [31:9] reference (kind 0: index of the 16-byte table entry whose
//                       output replaces input byte p - 1; kinds 1, 2: large tables, see fb_lane<3>)
//              more than 7 edits: words 1, 2 = index and length of a 64-word overflow record that holds all of them.
//            Every piece is recorded by exactly ONE lane although lanes own lines, not pieces: a lane records from
//            the first piece boundary at or after its first line start up to the first piece boundary at or after the
//            end of its last line (it simply keeps walking to the end of that piece: it is in the root state there,
//            exactly the state in which the next lane starts), and adds the pieces' output sizes (64 + delta) to the
//            total of their BLOCK (256 pieces, 16 KiB of input) with one atomic per lane and block.
//   scan     exclusive sum of the block totals (k_chunk_scan).
//   patch    one workgroup per block, one thread per piece: output offset of the piece = block base + prefix sum of
//            the pieces' sizes; the edits' texts and the copied bytes between them go to an LDS tile, the tile goes
//            out in aligned 16-byte rows.  No table walk, no sequential dependency between pieces.
//
// HBM traffic per input byte: record 1 read + 0.5 written (slots), patch 1 + 0.5 read + the output written.
#pragma once
#include <cstdint>

#include "scan_block.hpp"

namespace trre {

// ---- patch pass ---------------------------------------------------------------------------------------------------
// LDS tiles of a patch workgroup.  Input: 256 rows of 64 bytes, 68 bytes apart (17 dwords: the rows of 32 consecutive
// threads start in distinct banks).  Output: logical offset L (from a 64-byte aligned output address at or below the
// block's first byte) lives at L + 4 * (L / 64): the same skew for whatever the threads write in step.
constexpr int kPatchInStride = 68;
constexpr int kPatchInBytes = kBlockPieces * kPatchInStride;
constexpr int kPatchOutLogical = 24 * 1024;       // blocks that produce more go straight to memory, byte by byte
constexpr int kPatchOutBytes = kPatchOutLogical + kPatchOutLogical / 16 + 64;
TRRE_HD uint32_t patch_phys(uint32_t L) { return L + ((L >> 6) << 2); }

struct PatchTables {
    const uint8_t* g16;        // 16-byte entries (kind 0)
    const uint64_t* ent;       // 8-byte entries (slow ones: more than 4 bytes or pooled text)
    const uint8_t* pool;
};
// where a thread's bytes go: the block's LDS tile, or memory when the block's output does not fit the tile
struct TileSink {
    uint8_t* tile;             // LDS
    TRRE_HD void put(uint32_t L, uint32_t b) const { tile[patch_phys(L)] = (uint8_t)b; }
};
struct MemSink {
    uint8_t* out;              // a.out + block base - L0  (logical offset -> address)
    TRRE_HD void put(uint32_t L, uint32_t b) const { out[L] = (uint8_t)b; }
};
// length of the text of an edit of kind 0
TRRE_HD uint32_t patch_len_g16(const PatchTables& T, uint32_t idx) {
    const uint32_t meta = *reinterpret_cast<const uint32_t*>(T.g16 + ((size_t)idx << 4) + 4);
    if (!(meta & 128u)) return meta & 7u;
    const uint64_t e = T.ent[idx];                    // a slow entry: the 8-byte form (same index: both tables are [state][column])
    const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
    const uint32_t ol = (elo >> 24) & 7u;
    uint32_t len = ol;
    if (ol == 7u) __builtin_memcpy(&len, T.pool + str_pool_off(ehi), 4);
    return len + ((elo >> 27) & 1u);
}
// text of an edit of kind 0 whose entry is slow (more than 4 bytes or pooled text): from the 8-byte form
template <class Sink>
TRRE_HD uint32_t patch_text_slow(const PatchTables& T, const Sink& S, uint32_t idx, uint32_t c, uint32_t L) {
    const uint64_t e = T.ent[idx];
    const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
    const uint32_t ol = (elo >> 24) & 7u, cc = (elo >> 27) & 1u;
    uint32_t n = 0;
    if (ol != 7u) {
        for (uint32_t k = 0; k < ol; ++k) S.put(L + n++, (ehi >> (8u * k)) & 0xffu);
    } else {
        const uint8_t* rec = T.pool + str_pool_off(ehi);
        uint32_t len;
        __builtin_memcpy(&len, rec, 4);
        for (uint32_t k = 0; k < len; ++k) S.put(L + n++, rec[4 + k]);
    }
    if (cc) S.put(L + n++, c);
    return n;
}

// One piece: the thread-private part of the patch pass.  in_row: the piece's 64 input bytes (an LDS row); valid: how many
// of them are input; slot: the piece's slot (LDS); L: logical output offset of the piece's first byte; g16: the table of the
// edits' texts (LDS when it is small).
// Phase A walks the edits (their texts go out, their places are noted in a 64-bit mask, their lengths in nibbles), phase B
// the 64 positions without a branch (a copied byte goes to its own position plus what the edits before it added).
// Returns false for a piece this does not handle — more than 7 edits (an overflow record) or a text of more than 14 bytes:
// patch_piece_any does those.
template <class Sink>
TRRE_HD bool patch_piece(const PatchTables& T, const uint8_t* g16, const Sink& S, const uint8_t* in_row, uint32_t valid, const uint32_t* slot,
                         uint32_t L) {
    const uint32_t cnt = (slot[0] >> 16) & 255u;
    if (cnt == kSlotOverflow) return false;
    uint32_t ed_lo = 0, ed_hi = 0;     // bit i: input byte i is replaced by an edit's text
    uint32_t nib = 0;                  // edit e adds ((nib >> 4 e) & 15) - 1 bytes
    int32_t shift = 0;                 // what the edits so far have added
    bool ok = true;
    for (uint32_t e = 0; e < cnt; ++e) {
        const uint32_t w = slot[1 + e];
        const uint32_t p = (w & 127u) - 1u;           // the edit replaces input byte p
        const U128 g = *reinterpret_cast<const U128*>(g16 + ((size_t)(w >> 9) << 4));
        const uint32_t c = in_row[p];
        const uint32_t at = (uint32_t)((int32_t)(L + p) + shift);
        uint32_t n;
        if (!(g.y & 128u)) {
            n = g.y & 7u;
            const uint32_t bytes = perm_b32(c, g.z, g.w);
            for (uint32_t k = 0; k < n; ++k) S.put(at + k, (bytes >> (8u * k)) & 0xffu);
        } else {
            n = patch_len_g16(T, w >> 9);
            if (n > 14u) ok = false;
            else patch_text_slow(T, S, w >> 9, c, at);
        }
        ed_lo |= p < 32u ? 1u << p : 0u;
        ed_hi |= p >= 32u ? 1u << (p - 32u) : 0u;
        nib |= (n & 15u) << (4u * e);
        shift += (int32_t)n - 1;
    }
    if (!ok) return false;
    int32_t sh = 0;
#pragma clang loop unroll(disable)
    for (uint32_t d = 0; d < 16u; ++d) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(in_row + 4u * d);
        const uint32_t m4 = ((d < 8u ? ed_lo : ed_hi) >> ((4u * d) & 31u)) & 15u;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t i = 4u * d + j;
            const uint32_t is_ed = (m4 >> j) & 1u;
            if (!is_ed && i < valid) S.put((uint32_t)((int32_t)(L + i) + sh), (w >> (8u * j)) & 0xffu);
            sh += is_ed ? (int32_t)(nib & 15u) - 1 : 0;
            nib = is_ed ? nib >> 4 : nib;
        }
    }
    return true;
}
// the same for any piece, byte by byte with every length looked up (overflow records, long texts: rare)
template <class Sink>
TRRE_HD void patch_piece_any(const PatchArgs& pa, const PatchTables& T, const Sink& S, const uint8_t* in_row, uint32_t valid,
                             const uint32_t* slot, uint32_t L) {
    uint32_t cnt = (slot[0] >> 16) & 255u;
    const uint32_t* ev = slot + 1;
    if (cnt == kSlotOverflow) { cnt = slot[2]; ev = pa.ovf + (size_t)slot[1] * kOvfWords; }
    int32_t sh = 0;
    uint32_t e = 0;
    for (uint32_t i = 0; i < valid; ++i) {
        if (e < cnt && (ev[e] & 127u) == i + 1u) {
            const uint32_t idx = ev[e] >> 9;
            const uint32_t at = (uint32_t)((int32_t)(L + i) + sh);
            const U128 g = *reinterpret_cast<const U128*>(T.g16 + ((size_t)idx << 4));
            uint32_t n;
            if (!(g.y & 128u)) {
                n = g.y & 7u;
                const uint32_t bytes = perm_b32(in_row[i], g.z, g.w);
                for (uint32_t k = 0; k < n; ++k) S.put(at + k, (bytes >> (8u * k)) & 0xffu);
            } else {
                n = patch_text_slow(T, S, idx, in_row[i], at);
            }
            sh += (int32_t)n - 1;
            ++e;
        } else {
            S.put((uint32_t)((int32_t)(L + i) + sh), in_row[i]);
        }
    }
}

}  // namespace trre
