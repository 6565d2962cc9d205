// scan_core.hpp — the per-line scan functions each GPU lane runs.
//
// These are the hot path: the scan line loop (trre_dft.c:1272-1286 /
// trre_nft.c:775-790) and the per-attempt state-transition + emit loop
// (infer_dft trre_dft.c:1110-1196, infer_backtrack trre_nft.c:593-657),
// restated over flat tables.  They are written once as TRRE_HD functions so
// that scan_kernels.hip instantiates them for the device and tests/cpu_shim.cpp
// can drive the very same lane logic on the host against the oracle when no GPU
// is present (test infrastructure only — the product has no CPU scan path).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define TRRE_HD __host__ __device__ __forceinline__
#else
#define TRRE_HD inline
#endif

namespace trre {

// status bits reported by a launch (device word, read back by the host)
enum : uint32_t {
    kStNul = 1u << 0,        // a NUL byte was seen (length-preserving launch is void)
    kStDiverge = 1u << 1,    // non-deterministic search would not terminate (reference hangs)
    kStCapacity = 1u << 2,   // output does not fit the caller's buffer
    kStLongLine = 1u << 3,   // informational: a line left the LDS tile (slow path taken)
    kStOverflow = 1u << 5    // a stream table met an attempt longer than it was built for: the launch is void
};

// ---- deterministic tables as the kernel sees them --------------------------------
struct DftView {
    const uint64_t* ent0;   // [256] start row indexed by raw byte
    const uint8_t* cls;     // [256] byte -> column class
    const uint64_t* ent;    // [n_rows][n_cls]
    const uint8_t* pool;    // pooled outputs: {u32 len, bytes}
    uint32_t n_cls;
};

TRRE_HD uint32_t ent_kind(uint64_t e) { return (uint32_t)e & 3u; }
// kind 3: the reference never returns from this edge (epsilon cycle) — report it and treat the edge as dead
TRRE_HD uint32_t ent_kind_st(uint64_t e, uint32_t& status) {
    const uint32_t k = (uint32_t)e & 3u;
    if (k == 3u) { status |= kStDiverge; return 0u; }
    return k;
}
TRRE_HD uint32_t ent_ilen(uint64_t e) { return ((uint32_t)e >> 2) & 7u; }
TRRE_HD uint32_t ent_next(uint64_t e) { return ((uint32_t)e >> 5); }
TRRE_HD uint32_t ent_hi(uint64_t e) { return (uint32_t)(e >> 32); }

TRRE_HD uint32_t ent_out_len(const DftView& T, uint64_t e) {
    uint32_t il = ent_ilen(e);
    if (il != 7u) return il;
    const uint8_t* r = T.pool + ent_hi(e);
    return (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24;
}

// ---- byte sources ----------------------------------------------------------------
// Positions are int64 in whatever coordinate system the source defines.  Every
// source guarantees a '\n' at or before its end, so a walk always terminates.
struct TileIn {            // LDS tile; tile[tile_len] holds a '\n' sentinel
    const uint8_t* t;
    TRRE_HD uint8_t operator()(int64_t p) const { return t[p]; }
};
struct GlobalIn {          // whole input in HBM; the last byte of the input acts as '\n'
    const uint8_t* in;     // (a record without '\n' loses its last byte: trre_nft.c:777)
    int64_t last;          // n - 1
    TRRE_HD uint8_t operator()(int64_t g) const { return g >= last ? (uint8_t)'\n' : in[g]; }
};

// ---- byte sinks --------------------------------------------------------------------
struct PosOut {            // length-preserving: output position == input position
    uint8_t* o;
    TRRE_HD void put(int64_t p, uint8_t c) const { o[p] = c; }
};
struct CountSink {
    uint64_t n = 0;
    static constexpr bool kCountOnly = true;
    TRRE_HD void put(uint8_t) { ++n; }
    TRRE_HD void add(uint64_t k) { n += k; }
};
struct ByteSink {          // sequential writer (LDS staging tile or HBM)
    uint8_t* o;
    uint64_t n = 0;
    static constexpr bool kCountOnly = false;
    TRRE_HD void put(uint8_t c) { o[n++] = c; }
    TRRE_HD void add(uint64_t) {}
};

template <class Sink>
TRRE_HD void sink_entry(const DftView& T, Sink& s, uint64_t e) {
    uint32_t il = ent_ilen(e);
    if (il != 7u) {
        uint32_t w = ent_hi(e);
        for (uint32_t k = 0; k < il; ++k) { s.put((uint8_t)w); w >>= 8; }
    } else {
        const uint8_t* r = T.pool + ent_hi(e);
        uint32_t len = (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24;
        for (uint32_t k = 0; k < len; ++k) s.put(r[4 + k]);
    }
}

// -------------------------------------------------------------------------------------
// Deterministic engine, length-preserving tables (kFlagLengthPreserving |
// kFlagNoOverrun): one line, output written at the input's own positions.
// Attempt output is written speculatively; a failed attempt is simply
// overwritten by the raw byte copy (trre_dft.c:1193-1195,1281-1282) because the
// pending output never runs ahead of the bytes consumed.
// Returns the position of the line's terminating '\n'.
// -------------------------------------------------------------------------------------
template <class In, class Out>
TRRE_HD int64_t dft_line_lp(const DftView& T, In in, Out out, int64_t p, uint32_t& status) {
    for (;;) {
        const uint8_t c0 = in(p);
        uint64_t e = T.ent0[c0];
        uint32_t kind = ent_kind_st(e, status);
        if (kind == 0u) {
            if (c0 == (uint8_t)'\n') { out.put(p, (uint8_t)'\n'); return p; }
            if (c0 == 0) status |= kStNul;
            out.put(p, c0);
            ++p;
            continue;
        }
        int64_t i = p, oo = p;
        for (;;) {
            uint32_t il = ent_ilen(e);
            if (il != 7u) {
                uint32_t w = ent_hi(e);
                for (uint32_t k = 0; k < il; ++k) { out.put(oo++, (uint8_t)w); w >>= 8; }
            } else {
                const uint8_t* r = T.pool + ent_hi(e);
                uint32_t len = (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24;
                for (uint32_t k = 0; k < len; ++k) out.put(oo++, r[4 + k]);
            }
            ++i;
            if (kind == 2u) { p = i; break; }                 // first final state: shortest match
            const uint8_t c = in(i);
            e = T.ent[(uint64_t)ent_next(e) * T.n_cls + T.cls[c]];
            kind = ent_kind_st(e, status);
            if (kind == 0u) { out.put(p, c0); ++p; break; }   // dead (or end of line): copy one raw byte
        }
    }
}

// -------------------------------------------------------------------------------------
// Deterministic engine, any tables: one line into a sequential sink.  An attempt
// is first verified without writing (its output must be discarded if it dies),
// then replayed into the sink.  A NUL ends the line's content; the rest of the
// record up to '\n' is dropped (C-string semantics, trre_dft.c:1277,1118).
// Returns the position of the record's terminating '\n'.
// -------------------------------------------------------------------------------------
template <class In, class Sink>
TRRE_HD int64_t dft_line_gen(const DftView& T, In in, Sink& sink, int64_t p, uint32_t& status) {
    for (;;) {
        const uint8_t c0 = in(p);
        const uint64_t e0 = T.ent0[c0];
        if (ent_kind_st(e0, status) == 0u) {
            if (c0 == (uint8_t)'\n') { sink.put((uint8_t)'\n'); return p; }
            if (c0 == 0) {
                sink.put((uint8_t)'\n');
                do { ++p; } while (in(p) != (uint8_t)'\n');
                return p;
            }
            sink.put(c0);
            ++p;
            continue;
        }
        // verify
        int64_t i = p;
        uint64_t e = e0, acc = 0;
        bool ok;
        for (;;) {
            if (Sink::kCountOnly) acc += ent_out_len(T, e);
            ++i;
            if (ent_kind(e) == 2u) { ok = true; break; }
            e = T.ent[(uint64_t)ent_next(e) * T.n_cls + T.cls[in(i)]];
            if (ent_kind_st(e, status) == 0u) { ok = false; break; }
        }
        if (!ok) { sink.put(c0); ++p; continue; }
        if (Sink::kCountOnly) {
            sink.add(acc);
        } else {                                              // replay p..i into the sink
            e = e0;
            int64_t j = p;
            for (;;) {
                sink_entry(T, sink, e);
                ++j;
                if (ent_kind(e) == 2u) break;
                e = T.ent[(uint64_t)ent_next(e) * T.n_cls + T.cls[in(j)]];
            }
        }
        p = i;
    }
}

// ---- stream tables (the scan loop folded into one transducer, stream_build.cpp) -----------
struct StreamView {
    const uint8_t* cls;     // [256]
    const uint64_t* ent;    // [n_states][n_cls]
    const uint8_t* pool;
    const uint8_t* pool_fast = nullptr;   // a copy of the pool in LDS when it fits (emit pass), else null
    const uint8_t* g16 = nullptr;         // 16-byte count / emit entries (front.hpp), when the walker uses them
    const uint8_t* p32 = nullptr;         // pair form of the same table (front.hpp), or null
    uint32_t p32_slow = 0;                // some pair entry is "slow"
    uint32_t long_pool = 1;  // 0: no pooled text reaches 255 bytes, i.e. every pooled entry carries its exact length
};
TRRE_HD uint64_t str_entry(const StreamView& T, uint32_t idx) { return T.ent[idx]; }
constexpr uint32_t kStrCopyC = 1u << 27, kStrEol = 1u << 28;
constexpr uint32_t kStrOvf = 1u << 30;      // the attempt outgrew the table (bounded fold): the result is void, see kStOverflow

TRRE_HD uint32_t str_olen(uint32_t lo) { return (lo >> 24) & 7u; }
TRRE_HD uint32_t str_next(uint32_t lo) { return lo & 0xffffffu; }
TRRE_HD uint32_t str_pool_off(uint32_t hi) { return (hi & 0xffffffu) << 2; }   // records are 4-byte aligned: {u32 len, bytes}
TRRE_HD uint32_t str_pool_len(const StreamView& T, uint32_t hi) {
    const uint32_t l = hi >> 24;
    return l != 255u ? l : *reinterpret_cast<const uint32_t*>(T.pool + str_pool_off(hi));
}
// number of bytes one transition emits
TRRE_HD uint32_t str_count(const StreamView& T, uint32_t lo, uint32_t hi) {
    const uint32_t ol = str_olen(lo);
    return (ol == 7u ? str_pool_len(T, hi) : ol) + ((lo >> 27) & 1u);
}
// write one transition's bytes at o (in-place or staging buffer); returns the new cursor
TRRE_HD int64_t str_emit(const StreamView& T, uint8_t* dst, int64_t o, uint32_t lo, uint32_t hi, uint8_t c) {
    const uint32_t ol = str_olen(lo);
    const uint32_t cc = (lo >> 27) & 1u;
    if (ol != 7u) {
        const uint32_t n = ol + cc;
        if (n) dst[o] = ol ? (uint8_t)hi : c;                  // the common case: at most one byte
        if (n >= 2u) {                                        // rare: replacement text / flushed pending bytes
            uint32_t w = hi >> 8;
            for (uint32_t k = 1; k < ol; ++k) { dst[o + k] = (uint8_t)w; w >>= 8; }
            if (cc) dst[o + ol] = c;
        }
        return o + n;
    }
    const uint8_t* r = T.pool + str_pool_off(hi);
    const uint32_t len = str_pool_len(T, hi);
    for (uint32_t k = 0; k < len; ++k) dst[o + k] = r[4 + k];
    if (cc) dst[o + len] = c;
    return o + len + cc;
}

// ---- byte permute / funnel shifts (v_perm_b32, v_alignbit_b32, v_alignbyte_b32) -------------
TRRE_HD uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(s0, s1, sel);
#else
    const uint64_t src = (uint64_t)s0 << 32 | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = (sel >> (8 * i)) & 0xffu;
        const uint32_t byte = b <= 7u ? (uint32_t)(src >> (8 * b)) & 0xffu : (b >= 0x0du ? 0xffu : 0u);
        r |= byte << (8 * i);
    }
    return r;
#endif
}
// a * b for operands below 2^24 (table offsets, class counts): v_mul_u32_u24 issues at full rate, v_mul_lo_u32 at a quarter of it — and the
// pair walks of the small tables multiply twice per step
TRRE_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
TRRE_HD uint32_t alignbit_b32(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)(((uint64_t)hi << 32 | lo) >> (sh & 31u));
#endif
}
TRRE_HD uint32_t alignbyte_b32(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (uint32_t)(((uint64_t)hi << 32 | lo) >> (8u * (sh & 3u)));
#endif
}

// ---- non-deterministic tables as the kernel sees them ---------------------------------
struct NftFollowDev {      // mirrors trre::NftFollow (flags: 1 mute, 2 the target echoes the byte it reads)
    uint8_t target, flags;
    uint16_t out_len;
    uint32_t out_off;
};
struct NftView {
    const uint64_t* cons_mask;    // [256]
    const uint64_t* pred;         // [n_cons + 1]; [n_cons] = states that can reach FINAL directly
    const uint32_t* follow_off;   // [n_cons + 2]; [n_cons] = start
    const NftFollowDev* follow;
    const uint8_t* pool;
    uint32_t n_cons;
};

// One backward step of the co-reachability sweep:  given X = G[i+1] (CONS states
// that are alive on the next byte) return G[i] for byte c.
TRRE_HD uint64_t nft_back(const NftView& T, uint64_t next_alive, uint8_t c) {
    uint64_t m = T.pred[T.n_cons];
    uint64_t x = next_alive;
    while (x) {
#if defined(__HIP_DEVICE_COMPILE__)
        int t = __ffsll((unsigned long long)x) - 1;
#else
        int t = __builtin_ctzll(x);
#endif
        x &= x - 1;
        m |= T.pred[t];
    }
    return m & T.cons_mask[c];
}

// -------------------------------------------------------------------------------------
// Non-deterministic engine: one line.  `G` gives the backward-sweep mask for an
// input position (callers fill it before the walk).  Semantics per attempt
// (trre_nft.c:593-657 + 775-790): the first path in priority order that reaches
// FINAL wins and its own output is printed (cut at a NUL, like fputs); if it
// consumed nothing the raw byte is copied as well; one more attempt runs on the
// empty tail of the line.  `len_end` is the position of the line's terminator.
// -------------------------------------------------------------------------------------
template <class In, class GMask, class Sink>
TRRE_HD void nft_line(const NftView& T, In in, GMask G, Sink& sink, int64_t p, int64_t end, uint32_t& status) {
    for (;;) {
        // one attempt starting at p (p == end: the empty tail)
        uint32_t s = T.n_cons;          // start pseudo-state
        int64_t i = p;
        bool muted = false, accepted = false;
        // Pass 1 decides acceptance without writing when the sink must not see a
        // failed attempt's bytes: with the guided walk an attempt that leaves the
        // start state always accepts, so only the first hop can fail.
        for (;;) {
            const uint64_t alive = i < end ? G(i) : 0ull;
            uint32_t k = T.follow_off[s];
            const uint32_t k_end = T.follow_off[s + 1];
            bool moved = false;
            for (; k < k_end; ++k) {
                const NftFollowDev f = T.follow[k];
                const bool fin = f.target == 0xFFu;
                if (fin || (alive >> f.target & 1ull)) {
                    if (!muted) {
                        if (Sink::kCountOnly) sink.add(f.out_len);
                        else for (uint32_t b = 0; b < f.out_len; ++b) sink.put(T.pool[f.out_off + b]);
                    }
                    if (f.flags & 1u) muted = true;
                    if (fin) accepted = true;
                    else {
                        if ((f.flags & 2u) && !muted) sink.put(in(i));     // a byte range in copy mode
                        s = f.target; ++i;
                    }
                    moved = true;
                    break;
                }
            }
            if (!moved || accepted) break;
        }
        if (p >= end) break;                       // that was the attempt on the empty tail
        if (accepted && i > p) p = i;
        else { sink.put(in(p)); ++p; }             // no match, or an empty match: copy one raw byte
    }
    sink.put((uint8_t)'\n');
}

}  // namespace trre
