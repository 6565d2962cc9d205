// device_blob.hpp — byte layout of the compiled tables as uploaded to HBM.
// Shared by the host serialiser (runtime.cpp), the kernels (scan_kernels.hip)
// and the test shim.  All offsets are from the start of the blob.
#pragma once
#include <cstdint>

namespace trre {

constexpr uint32_t kMagicDft = 0x31445254u;   // "TRD1"
constexpr uint32_t kMagicNft = 0x314e5254u;   // "TRN1"

struct DftBlobHeader {
    uint32_t magic, n_rows, n_cls, flags;
    uint32_t off_ent0;       // u64[256]  start row by raw byte
    uint32_t off_cls;        // u8[256]
    uint32_t off_bytemap;    // u8[256]   (kFlagMemoryless)
    uint32_t off_ent;        // u64[n_rows][n_cls]
    uint32_t off_pool, pool_bytes;
    uint32_t total_bytes, max_edge_out, n_states;
    uint32_t pad[3];
};
static_assert(sizeof(DftBlobHeader) == 64, "header layout");

struct NftBlobHeader {
    uint32_t magic, n_cons, flags, n_follow;
    uint32_t off_cons_mask;  // u64[256]
    uint32_t off_pred;       // u64[n_cons + 1]
    uint32_t off_follow_off; // u32[n_cons + 2]
    uint32_t off_follow;     // NftFollow[n_follow] (8 bytes each)
    uint32_t off_pool, pool_bytes;
    uint32_t total_bytes, n_states;
    uint32_t pad[4];
};
static_assert(sizeof(NftBlobHeader) == 64, "header layout");

constexpr uint32_t kMagicStream = 0x31535254u;   // "TRS1"
struct StreamBlobHeader {
    uint32_t magic, n_states, n_cls, flags;
    uint32_t off_cls;        // u8[256]
    uint32_t off_ent;        // u64[n_states][n_cls]
    uint32_t ent_bytes;
    uint32_t off_pool, pool_bytes;
    uint32_t total_bytes, max_out;
    uint32_t off_lpw, lpw_bytes, lpw_delay;   // window form (0 bytes when not available)
    uint32_t off_g16, g16_bytes;              // 16-byte count / emit entries (0 bytes when not available)
    uint32_t off_p32, p32_bytes;              // pair form (0 bytes when not available)
    uint32_t p32_slow;                        // some pair entry is "slow"
    uint32_t off_lpw2;                        // pair form of the window entries (lpw2_bytes: 0 when not available)
    // fallback form of a large table (front.hpp, StreamTables::fb_*): 0 slots when not available
    uint32_t fb_slots, off_fb_comb;           // u64[fb_slots]
    uint32_t fb_lits, off_fb_lit;             // u64[fb_lits]
    uint32_t fb_escs, off_fb_esc_slot;        // u32[fb_escs] ascending
    uint32_t off_fb_esc, off_fb_pool;         // escape records (4 words each) and their texts
    uint32_t fb_start[3][2];                  // root, SKIP, DONE: {descriptor, next-state bits of an entry's hi}
    uint32_t off_fb_lit_meta;                 // u16[fb_lits]: the copy form (front.hpp); 0: the tables do not have it
    uint32_t lpw2_bytes;
    // the mark form of the comb (front.hpp, StreamTables::fb_comb4): 0 slots when not available
    uint32_t fb4_slots, off_fb_comb4;         // u32[fb4_slots]
    uint32_t fb4_dense, off_fb_dense4;        // u32[fb4_dense][32]
    uint32_t off_fb_dense_base;               // u16[fb4_dense]
    uint32_t fb_pad;
    uint32_t fb_start4[3];
    uint32_t off_mg, mg_max;                  // memoryless programs (map_block.hpp): u32[256][4] = {text lo, text hi, length | kMgNul, 0}; mg_max = the longest text, 0: not one
    uint32_t pad4[1];
};
static_assert(sizeof(StreamBlobHeader) == 192, "header layout");

// entry bits of the fallback form (front.hpp)
constexpr uint32_t kFbCc = 1u << 17, kFbNl = 1u << 18, kFbEol = 1u << 19;      // (kFbCc and kFbNl both: an escape)
constexpr uint32_t kFbNoTag = 0x3fffu;

// backward pass of the guided families (guided_build.cpp): a DFA read right to left
constexpr uint32_t kMagicRev = 0x31525254u;   // "TRR1"
struct RevBlobHeader {
    uint32_t magic, n_rev, n_cls;
    uint32_t off_cls;        // u8[256] byte -> class            } the compact form (inspection, host tests)
    uint32_t off_tab;        // u8[n_rev][n_cls] next state      }
    uint32_t off_wide;       // u8[n_rev][256] next state by raw byte (= the symbol left at the byte's position): what the kernel walks
    uint32_t total_bytes;
    uint32_t sym_bits;       // 8: one symbol per input byte; 4: two per byte (n_rev <= 16)
};
static_assert(sizeof(RevBlobHeader) == 32, "header layout");

}  // namespace trre
