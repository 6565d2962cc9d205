// scan_block.hpp — what one workgroup does with one chunk of the input.
//
// Data layout.  The input is cut into CHUNK-byte chunks in
// "v-space": v = g + a where g is the byte offset in the caller's buffer and
// a = (address of in) & 15, so every tile row is a 16-byte aligned HBM
// address.  Workgroup b owns the lines that START inside chunk b.  It stages
// the v-range [b*CHUNK - PRE, b*CHUNK + CHUNK + HALO) into an LDS tile with
// coalesced 16-byte loads (the PRE bytes give the left context that decides
// whether the chunk begins at a line start, HALO lets the last owned line run
// past the chunk end).  Lane t then owns the lines that start inside its SUB =
// CHUNK/THREADS byte sub-range — one lane per line at a time, lanes walking
// disjoint lines from LDS.  Bytes outside the input are staged as '\n', and so
// is the input's very last byte (a final record without '\n' loses its last
// byte: trre_nft.c:777, trre_dft.c:1274), hence every record ends in '\n'.
// A line that leaves the tile (longer than HALO) is redone by its lane straight
// from HBM (slow path, correctness only).
//
// The functions here are the per-thread bodies of the kernel phases; the
// kernels in scan_kernels.hip put barriers and wave reductions between them,
// tests/cpu_shim.cpp runs them thread by thread on the host.
#pragma once
#include <cstddef>
#include <cstdint>
#include <type_traits>

#include "device_blob.hpp"
#include "scan_core.hpp"

namespace trre {

struct alignas(16) U128 { uint32_t x, y, z, w; };


template <int THREADS_, int CHUNK_, int HALO_>
struct Geometry {
    static constexpr int THREADS = THREADS_;
    static constexpr int CHUNK = CHUNK_;
    static constexpr int HALO = HALO_;
    static constexpr int PRE = 16;
    static constexpr int TILE = PRE + CHUNK + HALO;   // staged bytes; tile[TILE] = '\n' sentinel
    static constexpr int TILE_ALLOC = TILE + 16;
    static constexpr int SUB = CHUNK / THREADS;
    static_assert(CHUNK % THREADS == 0 && TILE % 16 == 0, "geometry");
};

struct ScanArgs {
    const uint8_t* in_v0;    // in - a   (16-byte aligned)
    uint8_t* out_v0;         // out - a  (length-preserving launches: out position == in position)
    uint8_t* out;            // general launches: sequential output
    int64_t vbeg, vend;      // valid input is v in [vbeg, vend)
    const uint8_t* blob;     // compiled tables in HBM
    uint32_t* status;        // status bits (kSt*)
    uint32_t* lane_counts;   // [n_chunks][THREADS] output bytes per lane (general path)
    uint64_t* chunk_total;   // [n_chunks]
    uint64_t* chunk_base;    // [n_chunks + 1] exclusive scan of chunk_total; [n_chunks] = total
    uint64_t cap;            // capacity of out
    uint8_t* gscratch;       // NFT long-line mask scratch (or null)
    uint32_t* redo;          // window kernel: [0] = count, [1..] = lanes to redo with the general direct walker
    const uint8_t* rblob;    // guided families: tables of the backward pass (RevBlobHeader)
    uint32_t dbg;            // experiments (TRRE_EMIT_DBG): 1 no global stores in the emit pass, 2 no ring writes; the output is void
    uint32_t lp_emit;        // emit pass of a length-preserving program without a count pass: a lane's output starts at
                             // the position of its first line start (output position == input position at line starts)
    uint8_t* sym_v0;         // guided families: one symbol per input byte, indexed like in_v0 (v-space); the
                             // backward pass fills [0, round_up(vend, 64)), the forward pass reads it
    uint32_t* nul_list;      // byte map: [0] = NUL bytes met, from [2] on their offsets in the input (64 bits each, the first
                             // kNulCap of them), or null: what finish() repairs the output from (runtime.cpp)
    // Exact sub-ranges (round 5; g16_lane): a lane walks the bytes [lo, hi) of its sub-range and nothing else, from the state the
    // transducer is in at lo — not "the lines that start in the sub-range", whose walk is as long as the longest of them (a line of
    // 400 KB: one lane walks it all, its neighbours idle).  The state at lo is what the count pass GUESSES from the kSpecLook bytes
    // before lo (exact when a line starts among them) and k_spec_verify / k_spec_repair make true: lane i's entry must be lane
    // i - 1's exit.  exact: 0 off; 1 the count pass (guesses its entry, leaves entry / exit / count); 2 the emit pass (entry given);
    // 3 a repair walk (entry given, leaves exit / count)
    uint32_t exact;
    uint32_t spec_look;      // bytes before lo the count pass looks at for its guess (kSpecLook; the tests use less, to provoke wrong guesses)
    uint32_t* entry_rows;    // [lanes]: row offset of the state at lo | 1 when it is known to be exact
    uint32_t* exit_rows;     // [lanes]: row offset of the state at hi
    uint32_t* spec_flags;    // [lanes]: 1 = the lane's entry is not the exit of the lane before it (k_spec_verify); status[3] counts them
    // ... and for the backward pass of the guided families (rev_sweep_lane): a lane starts from the state at the end of its sub-range — exact
    // when a line ends within spec_look bytes behind it, else guessed (as if the line ended spec_look bytes on) and checked against the symbol
    // the lane to its right leaves at that position (k_rev_verify / k_rev_repair; status[2] counts the wrong guesses)
    uint32_t* rev_guess;     // [backward lanes]: the state the lane started from | 1 << 31 when it is known to be exact
    uint32_t* rev_flags;     // [backward lanes]
};
constexpr int64_t kSpecLook = 256;
constexpr uint32_t kNulCap = 1024;

// ---- phase: stage the tile -------------------------------------------------------------
template <class G>
TRRE_HD void tile_load(const ScanArgs& a, int64_t v0, uint8_t* tin, int tid) {
    const uint32_t nl4 = 0x0a0a0a0au;
    for (int j = tid * 16; j < G::TILE; j += G::THREADS * 16) {
        const int64_t v = v0 + j;
        U128 w;
        if (v >= a.vend) {
            w.x = w.y = w.z = w.w = nl4;
        } else if (v + 16 <= a.vbeg) {                     // before the input: filler, '\n' only right before it
            w.x = w.y = w.z = 0x78787878u;
            w.w = v + 16 == a.vbeg ? 0x0a787878u : 0x78787878u;
        } else {
            w = *reinterpret_cast<const U128*>(a.in_v0 + v);
            if (v < a.vbeg || v + 16 > a.vend - 1) {       // vector straddles an end of the input
                uint8_t* b = reinterpret_cast<uint8_t*>(&w);
                for (int k = 0; k < 16; ++k) {
                    const int64_t vv = v + k;
                    if (vv >= a.vend - 1) b[k] = (uint8_t)'\n';
                    else if (vv < a.vbeg) b[k] = vv == a.vbeg - 1 ? (uint8_t)'\n' : (uint8_t)'x';
                }
            }
        }
        *reinterpret_cast<U128*>(tin + j) = w;
    }
    if (tid == 0) tin[G::TILE] = (uint8_t)'\n';
}

// first line start inside [lo, hi) (tile coordinates), or hi if none
TRRE_HD int64_t first_line_start(const uint8_t* tin, int64_t lo, int64_t hi) {
    int64_t q = lo;
    while (q < hi && tin[q - 1] != (uint8_t)'\n') ++q;
    return q;
}

template <class G>
TRRE_HD void lane_range(const ScanArgs& a, int64_t v0, int tid, int64_t& lo, int64_t& hi) {
    lo = G::PRE + (int64_t)tid * G::SUB;
    hi = lo + G::SUB;
    const int64_t lim_lo = a.vbeg - v0, lim_hi = a.vend - v0;
    if (lo < lim_lo) lo = lim_lo;
    if (hi > lim_hi) hi = lim_hi;
}

// ---- phase: length-preserving walk -------------------------------------------------------
// Writes the lane's lines into tout at the input's own tile positions and
// reports the tile range [first, last) it produced.
template <class G, class Engine>
TRRE_HD void lane_walk_lp(const ScanArgs& a, const typename Engine::View& T, typename Engine::Lane& L, int64_t v0,
                          const uint8_t* tin, uint8_t* tout, int tid, int32_t& first, int32_t& last, uint32_t& status) {
    int64_t lo, hi;
    lane_range<G>(a, v0, tid, lo, hi);
    first = 0x7fffffff;
    last = -1;
    if (lo >= hi) return;
    int64_t q = first_line_start(tin, lo, hi);
    if (q >= hi) return;
    first = (int32_t)q;
    while (q < hi) {
        const int64_t e = Engine::line_lp_tile(T, L, tin, tout, q, G::TILE, status);
        if (e >= G::TILE) {                   // reached the sentinel: the line leaves the tile
            status |= kStLongLine;
            Engine::line_lp_global(T, L, a, v0 + q, status);
            last = (int32_t)q;
            return;
        }
        q = e + 1;
    }
    last = (int32_t)q;
}

// ---- phase: write the produced tile range back, coalesced ---------------------------------
template <class G>
TRRE_HD void tile_store_lp(const ScanArgs& a, int64_t v0, const uint8_t* tout, int first, int last, int tid) {
    if (first >= last) return;
    const bool aligned = (reinterpret_cast<uintptr_t>(a.out_v0) & 15u) == 0;
    for (int j = (first & ~15) + tid * 16; j < last; j += G::THREADS * 16) {
        if (aligned && j >= first && j + 16 <= last) {
            *reinterpret_cast<U128*>(a.out_v0 + v0 + j) = *reinterpret_cast<const U128*>(tout + j);
        } else {
            for (int k = 0; k < 16; ++k) {
                const int jj = j + k;
                if (jj >= first && jj < last) a.out_v0[v0 + jj] = tout[jj];
            }
        }
    }
}

// ---- phase: general walk (count or emit into a sequential sink) ---------------------------
template <class G, class Engine, class Sink>
TRRE_HD void lane_walk_gen(const ScanArgs& a, const typename Engine::View& T, typename Engine::Lane& L, int64_t v0,
                           const uint8_t* tin, int tid, Sink& sink, uint32_t& status) {
    int64_t lo, hi;
    lane_range<G>(a, v0, tid, lo, hi);
    if (lo >= hi) return;
    int64_t q = first_line_start(tin, lo, hi);
    while (q < hi) {
        const uint64_t mark = sink.n;
        const int64_t e = Engine::line_gen_tile(T, L, tin, sink, q, G::TILE, status);
        if (e >= G::TILE) {
            status |= kStLongLine;
            sink.n = mark;                    // forget the partial line, redo it from HBM
            Engine::line_gen_global(T, L, a, sink, v0 + q, status);
            return;
        }
        q = e + 1;
    }
}

// copy `total` staged bytes (tout[shift .. shift+total)) to dst, where
// (dst - shift) is 16-byte aligned
template <class G>
TRRE_HD void tile_store_seq(uint8_t* dst, const uint8_t* tout, int shift, int64_t total, int tid) {
    const int64_t end = shift + total;
    uint8_t* base = dst - shift;
    for (int64_t j = (int64_t)tid * 16; j < end; j += G::THREADS * 16) {
        if (j >= shift && j + 16 <= end) {
            *reinterpret_cast<U128*>(base + j) = *reinterpret_cast<const U128*>(tout + j);
        } else {
            for (int k = 0; k < 16; ++k) {
                const int64_t jj = j + k;
                if (jj >= shift && jj < end) base[jj] = tout[jj];
            }
        }
    }
}

// =============================================================================================
// Deterministic engine
// =============================================================================================
struct DftEngine {
    static constexpr int kLdsEntBytes = 8192;   // rows kept in LDS when the whole table fits
    using View = DftView;
    struct Lane {};
    static constexpr int kMaskBytes = 0;
    TRRE_HD static Lane make_lane(uint8_t*) { return Lane{}; }

    // LDS carve for the tables: ent0[256] u64, cls[256], ent[...]
    static constexpr int kLdsBytes = 2048 + 256 + kLdsEntBytes;

    TRRE_HD static bool ent_fits(const DftBlobHeader& h) { return (uint64_t)h.n_rows * h.n_cls * 8u <= (uint64_t)kLdsEntBytes; }

    // cooperative copy of the hot tables into LDS
    TRRE_HD static void stage(const uint8_t* blob, uint8_t* lds, int tid, int nthreads) {
        const DftBlobHeader& h = *reinterpret_cast<const DftBlobHeader*>(blob);
        const uint64_t* e0 = reinterpret_cast<const uint64_t*>(blob + h.off_ent0);
        uint64_t* d0 = reinterpret_cast<uint64_t*>(lds);
        for (int k = tid; k < 256; k += nthreads) d0[k] = e0[k];
        const uint8_t* c = blob + h.off_cls;
        for (int k = tid; k < 256; k += nthreads) lds[2048 + k] = c[k];
        if (ent_fits(h)) {
            const uint64_t* e = reinterpret_cast<const uint64_t*>(blob + h.off_ent);
            uint64_t* d = reinterpret_cast<uint64_t*>(lds + 2048 + 256);
            const int n = (int)(h.n_rows * h.n_cls);
            for (int k = tid; k < n; k += nthreads) d[k] = e[k];
        }
    }
    TRRE_HD static View view(const uint8_t* blob, const uint8_t* lds) {
        const DftBlobHeader& h = *reinterpret_cast<const DftBlobHeader*>(blob);
        View v;
        v.ent0 = reinterpret_cast<const uint64_t*>(lds);
        v.cls = lds + 2048;
        v.ent = ent_fits(h) ? reinterpret_cast<const uint64_t*>(lds + 2048 + 256)
                            : reinterpret_cast<const uint64_t*>(blob + h.off_ent);
        v.pool = blob + h.off_pool;
        v.n_cls = h.n_cls;
        return v;
    }

    TRRE_HD static int64_t line_lp_tile(const View& T, Lane&, const uint8_t* tin, uint8_t* tout, int64_t q, int64_t, uint32_t& st) {
        return dft_line_lp(T, TileIn{tin}, PosOut{tout}, q, st);
    }
    TRRE_HD static void line_lp_global(const View& T, Lane&, const ScanArgs& a, int64_t v, uint32_t& st) {
        dft_line_lp(T, GlobalIn{a.in_v0, a.vend - 1}, PosOut{a.out_v0}, v, st);
    }
    template <class Sink>
    TRRE_HD static int64_t line_gen_tile(const View& T, Lane&, const uint8_t* tin, Sink& s, int64_t q, int64_t, uint32_t& st) {
        return dft_line_gen(T, TileIn{tin}, s, q, st);
    }
    template <class Sink>
    TRRE_HD static void line_gen_global(const View& T, Lane&, const ScanArgs& a, Sink& s, int64_t v, uint32_t& st) {
        dft_line_gen(T, GlobalIn{a.in_v0, a.vend - 1}, s, v, st);
    }
};

// =============================================================================================
// Non-deterministic engine.  MaskT is the narrowest unsigned type that holds
// one bit per CONS state; the G tile holds one MaskT per input byte.
// =============================================================================================
template <class MaskT>
struct NftEngine {
    using View = NftView;
    struct Lane { MaskT* gt; };                  // G tile (LDS), indexed by tile position
    static constexpr int kMaskBytes = (int)sizeof(MaskT);
    TRRE_HD static Lane make_lane(uint8_t* lds) { return Lane{reinterpret_cast<MaskT*>(lds)}; }
    static constexpr int kMaxConsLds = 64;
    // LDS carve: cons_mask[256] u64 + pred[65] u64 + follow_off[66] u32 (+ pad) ; follow lists stay in HBM/L2
    static constexpr int kLdsBytes = 2048 + 65 * 8 + 66 * 4 + 8;

    TRRE_HD static void stage(const uint8_t* blob, uint8_t* lds, int tid, int nthreads) {
        const NftBlobHeader& h = *reinterpret_cast<const NftBlobHeader*>(blob);
        const uint64_t* cm = reinterpret_cast<const uint64_t*>(blob + h.off_cons_mask);
        uint64_t* d = reinterpret_cast<uint64_t*>(lds);
        for (int k = tid; k < 256; k += nthreads) d[k] = cm[k];
        const uint64_t* pr = reinterpret_cast<const uint64_t*>(blob + h.off_pred);
        for (int k = tid; k <= (int)h.n_cons; k += nthreads) d[256 + k] = pr[k];
        const uint32_t* fo = reinterpret_cast<const uint32_t*>(blob + h.off_follow_off);
        uint32_t* df = reinterpret_cast<uint32_t*>(lds + 2048 + 65 * 8);
        for (int k = tid; k < (int)h.n_cons + 2; k += nthreads) df[k] = fo[k];
    }
    TRRE_HD static View view(const uint8_t* blob, const uint8_t* lds) {
        const NftBlobHeader& h = *reinterpret_cast<const NftBlobHeader*>(blob);
        View v;
        v.cons_mask = reinterpret_cast<const uint64_t*>(lds);
        v.pred = reinterpret_cast<const uint64_t*>(lds) + 256;
        v.follow_off = reinterpret_cast<const uint32_t*>(lds + 2048 + 65 * 8);
        v.follow = reinterpret_cast<const NftFollowDev*>(blob + h.off_follow);
        v.pool = blob + h.off_pool;
        v.n_cons = h.n_cons;
        return v;
    }

    struct TileMask { const MaskT* g; TRRE_HD uint64_t operator()(int64_t i) const { return (uint64_t)g[i]; } };
    struct GlobalMask { const MaskT* g; TRRE_HD uint64_t operator()(int64_t i) const { return (uint64_t)g[i]; } };

    // line content ends at the first '\n' or NUL; returns that position and the
    // position of the record's '\n' through rec_end
    template <class In>
    TRRE_HD static int64_t line_end(In in, int64_t q, int64_t limit, int64_t& rec_end) {
        int64_t e = q;
        uint8_t c;
        while ((c = in(e)) != (uint8_t)'\n' && c != 0 && e < limit) ++e;
        rec_end = e;
        if (c == 0 && e < limit) { do { ++rec_end; } while (in(rec_end) != (uint8_t)'\n' && rec_end < limit); }
        return e;
    }

    template <class Sink>
    TRRE_HD static int64_t line_tile(const View& T, Lane& L, const uint8_t* tin, Sink& s, int64_t q, int64_t tile_len, uint32_t& st) {
        TileIn in{tin};
        int64_t rec_end;
        const int64_t end = line_end(in, q, tile_len, rec_end);
        if (rec_end >= tile_len) return tile_len;          // leaves the tile: caller takes the slow path
        uint64_t alive = 0;
        for (int64_t i = end - 1; i >= q; --i) {           // backward co-reachability sweep
            alive = nft_back(T, alive, tin[i]);
            L.gt[i] = (MaskT)alive;
        }
        nft_line(T, in, TileMask{L.gt}, s, q, end, st);
        return rec_end;
    }
    template <class Sink>
    TRRE_HD static int64_t line_global(const View& T, Lane&, const ScanArgs& a, Sink& s, int64_t v, uint32_t& st) {
        GlobalIn in{a.in_v0, a.vend - 1};
        int64_t rec_end;
        const int64_t end = line_end(in, v, a.vend, rec_end);
        MaskT* g = reinterpret_cast<MaskT*>(a.gscratch);   // one mask per input byte, indexed by v
        uint64_t alive = 0;
        for (int64_t i = end - 1; i >= v; --i) {
            alive = nft_back(T, alive, in(i));
            g[i] = (MaskT)alive;
        }
        nft_line(T, in, GlobalMask{g}, s, v, end, st);
        return rec_end;
    }

    // length-preserving: the guided walk never discards output, so a line is
    // written front to back starting at its own position
    TRRE_HD static int64_t line_lp_tile(const View& T, Lane& L, const uint8_t* tin, uint8_t* tout, int64_t q, int64_t tile_len, uint32_t& st) {
        ByteSink s{tout + q};
        const int64_t e = line_tile(T, L, tin, s, q, tile_len, st);
        if (e < tile_len && (int64_t)s.n != e - q + 1) st |= kStNul;   // a NUL shortened the record
        return e;
    }
    TRRE_HD static void line_lp_global(const View& T, Lane& L, const ScanArgs& a, int64_t v, uint32_t& st) {
        if (!a.gscratch) { st |= kStNeedScratch(); return; }
        ByteSink s{a.out_v0 + v};
        const int64_t e = line_global(T, L, a, s, v, st);
        if ((int64_t)s.n != e - v + 1) st |= kStNul;
    }
    template <class Sink>
    TRRE_HD static int64_t line_gen_tile(const View& T, Lane& L, const uint8_t* tin, Sink& s, int64_t q, int64_t tile_len, uint32_t& st) {
        return line_tile(T, L, tin, s, q, tile_len, st);
    }
    template <class Sink>
    TRRE_HD static void line_gen_global(const View& T, Lane& L, const ScanArgs& a, Sink& s, int64_t v, uint32_t& st) {
        if (!a.gscratch) { st |= kStNeedScratch(); return; }
        line_global(T, L, a, s, v, st);
    }
    TRRE_HD static constexpr uint32_t kStNeedScratch() { return 1u << 4; }
};
constexpr uint32_t kStNeedScratch = 1u << 4;   // NFT long line met without mask scratch: relaunch with it

#if defined(__HIP_DEVICE_COMPILE__)
#define TRRE_WAVE_ANY(x) __any(x)
#define TRRE_WAVE_ALL(x) __all(x)
#define TRRE_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   // keep the scheduler from interleaving blocks
#define TRRE_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#define TRRE_TOUCH(x) asm volatile("" ::"v"(x))                // force x to be materialised here
#define TRRE_PIN(x) asm volatile("" : "+v"(x))                 // x is computed by here (stops reassociation into late trees)
#else
#define TRRE_NT_STORE(v, p) (*(p) = (v))
#define TRRE_TOUCH(x) ((void)(x))
#define TRRE_PIN(x) ((void)0)
#define TRRE_WAVE_ANY(x) (x)
#define TRRE_WAVE_ALL(x) (x)
#define TRRE_SCHED_FENCE() ((void)0)
#endif

constexpr uint32_t kStrNul = 1u << 29;
constexpr uint32_t kStrDiv = 1u << 31;       // guided tables: the reference's search would not terminate on this input
constexpr uint32_t kSkipState = 1, kDoneState = 2;

// =============================================================================================
// Direct stream walk: no LDS tile.  Each lane walks a long contiguous sub-range (lane_bytes, a few
// KiB) of the input straight from HBM/L2, 16-byte loads held in registers, so a lane's tail — the
// part of its last line beyond its sub-range — is small against its sub-range.  The count and emit
// passes fetch 64 bytes at a time (one sector, once) and the emit pass assembles the lane's output in
// a per-lane LDS staging buffer behind a 64-bit register window (see Stage); the in-place walk
// (kMode 0: tables without a window form, and the window kernel's edge lanes) keeps a 64-byte LDS
// ring per lane and stores aligned 16-byte chunks.
//   kMode 0: length-preserving (output address = input address), 1: count, 2: emit
// =============================================================================================
// ---- device-only: direct global -> LDS loads and explicit waits ------------------------------------
#if defined(__HIPCC__)
// 16 bytes per lane from global memory straight into LDS at lds_dst + 16 * lane id (lds_dst: wave-uniform
// LDS byte address).  The compiler neither counts this load nor knows that it writes LDS: the waits
// around it are explicit.
__device__ __forceinline__ void wt_glds16(const uint8_t* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t wt_lds_addr(const uint8_t* p) {
    return (uint32_t)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*)p);
}
#define TRRE_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define TRRE_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

// a lane's row of a wave tile (64 rows x 64 bytes; see "wave tiles" below)
struct WtRow {
    uint8_t* row;        // tile + r * 64
    uint32_t swz16;      // ((r >> 1) & 3) << 4
    TRRE_HD U128 load(int b) const { return *reinterpret_cast<const U128*>(row + ((uint32_t)(b << 4) ^ swz16)); }
    TRRE_HD void store(int b, const U128& v) const { *reinterpret_cast<U128*>(row + ((uint32_t)(b << 4) ^ swz16)) = v; }
};

struct DirectLane {
    uint64_t count = 0;        // kMode 1: bytes this lane emits
    // the one-pass walk (g16_lane<3>, one_block.hpp): in — the row offset of the state at lo, or kOneGuess: look back and guess it;
    // out — the state it walked from, whether that is exact by construction, and the state at hi
    uint32_t entry = 0xffffffffu, known = 0, exit = 0;
};
constexpr uint32_t kOneGuess = 0xffffffffu;
constexpr uint32_t kStOneVoid = 1u << 8;       // the one-pass kernel could not answer (a lane's output outgrew its LDS region, a workgroup's first guess was wrong,
                                               // a text too long for the staging): the count / emit pair runs the buffer

// 16 input bytes at v (16-byte aligned in v-space) as the walkers see them: bytes from the last one of
// the input on read as '\n' (every record ends in '\n', Q1), the byte right before the input as '\n'
// and anything before that as filler.  Word-wise, so that inlining it several times stays cheap.
TRRE_HD U128 direct_load(const ScanArgs& a, int64_t v) {
    const uint32_t nl4 = 0x0a0a0a0au;
    U128 w;
    if (v >= a.vend) { w.x = w.y = w.z = w.w = nl4; return w; }
    w = *reinterpret_cast<const U128*>(a.in_v0 + v);
    if (v < a.vbeg || v + 16 > a.vend - 1) {
        uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int64_t p = v + 4 * d;
            // bytes at positions >= vend - 1: the top n_hi bytes of the dword
            int64_t n_hi = p + 4 - (a.vend - 1);
            n_hi = n_hi < 0 ? 0 : (n_hi > 4 ? 4 : n_hi);
            const uint32_t m_hi = n_hi >= 4 ? 0xffffffffu : ~(0xffffffffu >> (8 * (int)n_hi));
            // bytes at positions < vbeg: the low n_lo bytes; the last of them (position vbeg - 1) is '\n'
            int64_t n_lo = a.vbeg - p;
            n_lo = n_lo < 0 ? 0 : (n_lo > 4 ? 4 : n_lo);
            const uint32_t m_lo = n_lo >= 4 ? 0xffffffffu : ~(0xffffffffu << (8 * (int)n_lo));
            uint32_t fill_lo = 0x78787878u;
            if (a.vbeg - 1 >= p && a.vbeg - 1 < p + 4) fill_lo = (fill_lo & ~(0xffu << (8 * (int)(a.vbeg - 1 - p)))) | (0x0au << (8 * (int)(a.vbeg - 1 - p)));
            uint32_t x = wd[d];
            x = (x & ~m_lo) | (fill_lo & m_lo);
            x = (x & ~m_hi) | (nl4 & m_hi);
            wd[d] = x;
        }
        w.x = wd[0]; w.y = wd[1]; w.z = wd[2]; w.w = wd[3];
    }
    return w;
}

// ... as a call (the one-pass kernel: the care for the input's two ends, inlined ten times, was a thousand instructions and fifty
// spilled registers on a path that two tiles of a launch take)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline)) U128 direct_load_cold(const uint8_t* in_v0, int64_t vbeg, int64_t vend, int64_t v) {
    ScanArgs a{};
    a.in_v0 = in_v0; a.vbeg = vbeg; a.vend = vend;
    return direct_load(a, v);
}
#else
inline U128 direct_load_cold(const uint8_t* in_v0, int64_t vbeg, int64_t vend, int64_t v) {
    ScanArgs a{};
    a.in_v0 = in_v0; a.vbeg = vbeg; a.vend = vend;
    return direct_load(a, v);
}
#endif

// flush the ring: bytes [of, o) of the lane's output (offsets from obase) are in ring[x & 63]
template <bool kAll>
TRRE_HD void direct_flush(uint8_t* obase, const uint8_t* ring, uint32_t& of, uint32_t o) {
    // head: single bytes up to the first 16-byte boundary of the output address
    while (of < o && (of & 15u)) { obase[of] = ring[of & 63u]; ++of; }
    while (o - of >= 16u) {
        const uint32_t r = of & 63u;
        U128 q;
        q.x = *reinterpret_cast<const uint32_t*>(ring + r);
        q.y = *reinterpret_cast<const uint32_t*>(ring + r + 4);
        q.z = *reinterpret_cast<const uint32_t*>(ring + r + 8);
        q.w = *reinterpret_cast<const uint32_t*>(ring + r + 12);
        *reinterpret_cast<U128*>(obase + of) = q;
        of += 16u;
    }
    if (kAll) while (of < o) { obase[of] = ring[of & 63u]; ++of; }
}

// ---- output staging of the emit pass ---------------------------------------------------------------
// A lane's output is a contiguous run of the output buffer that starts at an arbitrary byte.  The dword
// being filled lives in a 64-bit register window `acc`; a transition ORs its bytes in at the fill position
// and only a COMPLETED dword goes to the lane's 128-byte LDS ring (one aligned ds_write_b32, and only the
// lanes that completed one take part); nothing is ever moved inside the ring.  After every eight input
// bytes the complete 64-byte UNITS leave the ring, and they leave it through the wave: the lanes that have
// a unit ready post {ring position, output address} in a small per-wave LDS table, and four adjacent lanes
// store one unit — 16 bytes each, one 64-byte request — sixteen units per store instruction.
// History: (1) the window's two dwords stored on every transition and the remainder moved to the front of a
// linear buffer at every flush: PMC showed the LDS array busy 63 % of the kernel's time, a third of it bank
// conflicts.  (2) Ring, every lane storing its own 32-byte sectors as two 16-byte stores: PMC counted 86 M
// L2 write requests per GiB of output (13 bytes per request), four fifths of all the kernel's L2 traffic, and
// 1.38x write amplification in HBM — the pass ran at the pace of those requests, not of its instructions.
// Stream offsets count from the 64-byte aligned output address g0 at or below the lane's first byte; the
// ring holds offsets [fp, wp): 63 + 8 * 5 bytes at most between two flushes (tables with slow entries — up
// to 9 bytes per transition — flush after every four input bytes).  Rings are 132 bytes apart (33 dwords,
// odd: lanes that run in step store to distinct banks), so a ring is only 4-byte aligned.
constexpr int kRingStride = 132;
constexpr uint32_t kRingBytes = 128;
constexpr uint32_t kUnitBytes = 64;
constexpr int kWaveScratchBytes = 16 * 16;     // per wave: 16 posted units x {ring position, address lo, address hi, -}
// (the ablation switches of TRRE_EMIT_DBG sit on the per-byte path: a scalar test and a branch per append — they exist in builds
// with -DTRRE_DBG_SWITCHES only)
#if defined(TRRE_DBG_SWITCHES)
#define TRRE_STAGE_DBG(s, bit) ((s).dbg & (bit))
#else
#define TRRE_STAGE_DBG(s, bit) 0u
#endif
struct Stage {
    uint8_t* buf;        // the lane's ring: kRingBytes, 4-byte aligned
    uint8_t* g0;         // 64-byte aligned output address of stream offset 0
    uint32_t lo;         // the dword being filled: bytes [wp, wp + sh / 8) of the stream, zero above
    uint32_t wp;         // stream offset of the dword being filled (multiple of 4)
    uint32_t sh;         // 8 x the bytes of it that are filled (0, 8, 16, 24)
    uint32_t fp;         // everything below this stream offset has left for memory (multiple of 64)
    uint32_t skip;       // leading bytes of unit 0 that belong to whoever wrote before this lane's first byte
    uint32_t dbg;
    uint32_t* wsc;       // the wave's posting table (LDS), or null: every lane stores its own units (host shim)
};
// start (or restart) at an arbitrary output address; bytes below it in its 64-byte unit are not ours
TRRE_HD void stage_begin(Stage& s, uint8_t* buf, uint8_t* first_out_byte) {
    const uintptr_t start = reinterpret_cast<uintptr_t>(first_out_byte);
    s.buf = buf;
    s.g0 = reinterpret_cast<uint8_t*>(start & ~(uintptr_t)(kUnitBytes - 1u));
    s.skip = (uint32_t)(start & (kUnitBytes - 1u));
    s.wp = s.skip & ~3u;
    s.sh = (s.skip & 3u) << 3;
    s.fp = 0;
    s.lo = 0;
}
// The one-pass walk's staging (one_block.hpp): the lane's WHOLE output goes to a private linear region of LDS — its place in the output
// is not known before every lane of the workgroup has walked — through the same register window as the ring's.  A write beyond the
// region lands on its last dword instead (lim = region bytes - 4): the walk goes on counting, the caller finds wp > lim and gives up.
struct PStage {
    uint8_t* buf;        // the region (4-byte aligned)
    uint32_t lo, wp, sh; // as in Stage: the dword being filled, its offset in the region, 8 x its filled bytes
    uint32_t lim;        // offset of the region's last dword
    uint32_t dbg;
    uint32_t* wsc;
};
TRRE_HD void stage_begin(PStage& s, uint8_t* buf, uint8_t*) { s.buf = buf; s.lo = 0; s.wp = 0; s.sh = 0; }
TRRE_HD uint32_t stage_fill_end(const PStage& s) { return s.wp + (s.sh >> 3); }
TRRE_HD void stage_append_bits(PStage& s, uint32_t v, uint32_t n8) {
    const uint64_t vv = (uint64_t)v << s.sh;
    const uint32_t x = s.lo | (uint32_t)vv;
    *reinterpret_cast<uint32_t*>(s.buf + (s.wp < s.lim ? s.wp : s.lim)) = x;
    const uint32_t t = s.sh + n8;
    s.lo = t >= 32u ? (uint32_t)(vv >> 32) : x;
    s.wp += (t >> 5) << 2;
    s.sh = t & 31u;
}

// The byte-granular variant (fb_lane): no register window — a transition stores its 8 bytes straight into the ring at the
// byte position (LDS takes unaligned 8-byte stores on gfx950) and moves on by as many as count; what lies beyond is
// overwritten by the next one.  A store that runs over the end of the ring is repeated 128 bytes lower, so the ring has
// 8 spare bytes on either side.  wp is a byte offset here.
constexpr int kBRingPad = 8;
constexpr int kBRingStride = 148;              // 8 + 128 + 8, rounded to an odd number of dwords
struct BStage {
    uint8_t* buf;        // the lane's ring (kRingBytes, 4-byte aligned; 8 spare bytes below and above)
    uint8_t* g0;
    uint32_t wp;         // stream offset of the next byte
    uint32_t fp, skip, dbg;
    uint32_t* wsc;
};
TRRE_HD void stage_begin(BStage& s, uint8_t* buf, uint8_t* first_out_byte) {
    const uintptr_t start = reinterpret_cast<uintptr_t>(first_out_byte);
    s.buf = buf;
    s.g0 = reinterpret_cast<uint8_t*>(start & ~(uintptr_t)(kUnitBytes - 1u));
    s.skip = (uint32_t)(start & (kUnitBytes - 1u));
    s.wp = s.skip;
    s.fp = 0;
}
struct __attribute__((packed)) UnalignedU64 { uint64_t v; };
// 8 bytes at the fill position, of which the first n (0..8) count
TRRE_HD void bstage_put8(BStage& s, uint64_t v, uint32_t n) {
    const uint32_t o = s.wp & (kRingBytes - 1u);
    if (!TRRE_STAGE_DBG(s, 2u)) {
        // (an LDS store off its natural alignment is replayed at 64 cycles per wave instruction — two 4-byte stores
        // instead of one 8-byte store were measured slower still: 5.4 ms against 3.8 ms for the pass)
        reinterpret_cast<UnalignedU64*>(s.buf + o)->v = v;
        if (TRRE_WAVE_ANY(o > kRingBytes - 8u)) {
            if (o > kRingBytes - 8u) reinterpret_cast<UnalignedU64*>(s.buf + o - kRingBytes)->v = v;     // the part beyond the end, at the start
        }
    }
    s.wp += n;
}
// one byte at the fill position, counted or not
TRRE_HD void bstage_put1(BStage& s, uint32_t b, uint32_t n) {
    if (!TRRE_STAGE_DBG(s, 2u)) s.buf[s.wp & (kRingBytes - 1u)] = (uint8_t)b;
    s.wp += n;
}
TRRE_HD uint32_t stage_fill_end(const Stage& s) { return s.wp + (s.sh >> 3); }     // stream offset of the next byte
TRRE_HD uint32_t stage_fill_end(const BStage& s) { return s.wp; }
TRRE_HD void stage_spill(Stage& s) { *reinterpret_cast<uint32_t*>(s.buf + (s.wp & (kRingBytes - 1u))) = s.lo; }   // the dword being filled
TRRE_HD void stage_spill(BStage&) {}
TRRE_HD void stage_spill(PStage& s) { *reinterpret_cast<uint32_t*>(s.buf + (s.wp < s.lim ? s.wp : s.lim)) = s.lo; }
template <class St>
TRRE_HD uint8_t* stage_out_ptr(const St& s) { return s.g0 + stage_fill_end(s); }      // where the next byte goes
// append the low n (0..4) bytes of v; the bytes of v above n must be zero.
// No branch: the dword being filled goes to the ring on every append (complete or not — it is written again until it
// is), and what it could not take starts the next one.  (The first version kept a 64-bit window and stored a dword when
// it was complete: a compare, an exec mask and two moves more per append — 15 instructions against 11.)
TRRE_HD void stage_append_bits(Stage& s, uint32_t v, uint32_t n8) {     // n8 = 8 x the bytes that count
    const uint64_t vv = (uint64_t)v << s.sh;
    const uint32_t x = s.lo | (uint32_t)vv;
    if (!TRRE_STAGE_DBG(s, 2u)) *reinterpret_cast<uint32_t*>(s.buf + (s.wp & (kRingBytes - 1u))) = x;
    const uint32_t t = s.sh + n8;         // <= 56: at most one dword completed
    s.lo = t >= 32u ? (uint32_t)(vv >> 32) : x;
    s.wp += (t >> 5) << 2;
    s.sh = t & 31u;
}
TRRE_HD void stage_append_n4(Stage& s, uint32_t v, uint32_t n) { stage_append_bits(s, v, 8u * n); }
TRRE_HD void stage_append4(Stage& s, uint32_t v, uint32_t n) { stage_append_n4(s, v, n); }
// the same for up to 8 bytes (the second half only when some lane of the wave has more than 4)
TRRE_HD void stage_append(Stage& s, uint64_t v, uint32_t n) {
    const uint32_t n1 = n < 4u ? n : 4u;
    stage_append_n4(s, (uint32_t)v, n1);
    if (TRRE_WAVE_ANY(n > 4u)) stage_append_n4(s, (uint32_t)(v >> 32), n - n1);
}
// a pooled text of 5..8 bytes, then maybe the input byte c
TRRE_HD void stage_append_text_c(Stage& s, uint64_t text, uint32_t len, uint32_t c, uint32_t cc) {
    stage_append_n4(s, (uint32_t)text, 4u);
    const uint32_t rest = len - 4u;                                     // 1..4 bytes
    const uint32_t hi = (uint32_t)(text >> 32) & (0xffffffffu >> (32u - 8u * rest));
    stage_append_n4(s, hi, rest);
    stage_append_n4(s, cc ? c : 0u, cc);
}
TRRE_HD void stage_append_n4(PStage& s, uint32_t v, uint32_t n) { stage_append_bits(s, v, 8u * n); }
TRRE_HD void stage_append(PStage& s, uint64_t v, uint32_t n) {
    const uint32_t n1 = n < 4u ? n : 4u;
    stage_append_n4(s, (uint32_t)v, n1);
    if (TRRE_WAVE_ANY(n > 4u)) stage_append_n4(s, (uint32_t)(v >> 32), n - n1);
}
TRRE_HD void stage_append_text_c(PStage& s, uint64_t text, uint32_t len, uint32_t c, uint32_t cc) {
    stage_append_n4(s, (uint32_t)text, 4u);
    const uint32_t rest = len - 4u;
    const uint32_t hi = (uint32_t)(text >> 32) & (0xffffffffu >> (32u - 8u * rest));
    stage_append_n4(s, hi, rest);
    stage_append_n4(s, cc ? c : 0u, cc);
}
// stream bytes [from, to) from the ring to memory: single bytes up to a dword boundary, dwords, single bytes
template <class St>
TRRE_HD void stage_store_span(St& s, uint32_t from, uint32_t to) {
    uint32_t i = from;
    for (; i < to && (i & 3u); ++i) s.g0[i] = s.buf[i & (kRingBytes - 1u)];
    for (; i + 4u <= to; i += 4u)
        *reinterpret_cast<uint32_t*>(s.g0 + i) = *reinterpret_cast<const uint32_t*>(s.buf + (i & (kRingBytes - 1u)));
    for (; i < to; ++i) s.g0[i] = s.buf[i & (kRingBytes - 1u)];
}
template <class St>
TRRE_HD void stage_store_own_unit(St& s) {        // the unit at stream offset fp is complete: the lane stores it itself
    if (s.fp == 0 && s.skip) {
        stage_store_span(s, s.skip, kUnitBytes);     // once per lane: the unit it shares with its predecessor
    } else {
        const uint32_t* s32 = reinterpret_cast<const uint32_t*>(s.buf + (s.fp & (kRingBytes - 1u)));
        U128* dst = reinterpret_cast<U128*>(s.g0 + s.fp);
        if (!TRRE_STAGE_DBG(s, 1u)) {
            dst[0] = U128{s32[0], s32[1], s32[2], s32[3]};
            dst[1] = U128{s32[4], s32[5], s32[6], s32[7]};
            dst[2] = U128{s32[8], s32[9], s32[10], s32[11]};
            dst[3] = U128{s32[12], s32[13], s32[14], s32[15]};
        }
    }
    s.fp += kUnitBytes;
}
// everything the lane holds goes to memory, by the lane itself: for the divergent slow paths (only some lanes of the
// wave get here, so the wave cannot store their units for them)
template <class St>
TRRE_HD void stage_flush_solo(St& s) {
    while (s.wp - s.fp >= kUnitBytes) stage_store_own_unit(s);
    stage_spill(s);
    const uint32_t end = stage_fill_end(s);
    stage_store_span(s, s.fp == 0 ? s.skip : s.fp, end);
    s.skip = end & (kUnitBytes - 1u);
}
// kAll = false: the complete units (all lanes of the wave must take part); true: what is left as well (end of the lane)
// (kAll = false is called after every few input bytes, but stores only once some lane's ring could not take another such
// interval: kFlushAt bytes held, at most 36..40 more arrive between two calls.  By then about a third of the wave's lanes
// have a unit ready and one store instruction moves sixteen of them; flushing whenever any lane had one — the first
// version — ran this path five times as often for three or four units each.)
constexpr uint32_t kFlushAt = 84;
template <bool kAll, class St>
TRRE_HD void stage_flush(St& s) {
    if (!kAll) {
        if (!TRRE_WAVE_ANY(s.wp - s.fp >= kFlushAt)) return;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (s.wsc) {
        // a lane's first unit may share its 64 bytes with the lane before it: that one it stores itself
        if (TRRE_WAVE_ANY(s.fp == 0 && s.skip != 0 && s.wp >= kUnitBytes)) {
            if (s.fp == 0 && s.skip != 0 && s.wp >= kUnitBytes) stage_store_own_unit(s);
        }
        const uint32_t lid = __lane_id();
        uint64_t ready = __ballot(s.wp - s.fp >= kUnitBytes);
        while (ready) {
            const bool mine = (ready >> lid) & 1ull;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ready >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ready, 0u));
            const bool post = mine && rank < 16u;
            if (post) {
                const uintptr_t dst = reinterpret_cast<uintptr_t>(s.g0 + s.fp);
                uint32_t* slot = s.wsc + 4u * rank;
                slot[0] = (uint32_t)(int32_t)((s.buf + (s.fp & (kRingBytes - 1u))) - reinterpret_cast<uint8_t*>(s.wsc));
                slot[1] = (uint32_t)dst;
                slot[2] = (uint32_t)((uint64_t)dst >> 32);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t n_units = (uint32_t)__popcll(ready);
            const uint32_t u = lid >> 2;
            if (u < (n_units < 16u ? n_units : 16u)) {
                const uint32_t* slot = s.wsc + 4u * u;
                // (the ring position was posted as a signed distance from the posting table: both live in this workgroup's LDS)
                const uint32_t* src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s.wsc) + (ptrdiff_t)(int32_t)slot[0] + 16 * (ptrdiff_t)(lid & 3u));
                uint8_t* dst = reinterpret_cast<uint8_t*>((uintptr_t)((uint64_t)slot[2] << 32 | slot[1])) + 16u * (lid & 3u);
                const U128 q{src[0], src[1], src[2], src[3]};
                if (!TRRE_STAGE_DBG(s, 1u)) *reinterpret_cast<U128*>(dst) = q;
            }
            __builtin_amdgcn_wave_barrier();
            if (post) s.fp += kUnitBytes;
            ready &= ~__ballot(post);
            ready |= __ballot(post && s.wp - s.fp >= kUnitBytes);       // (a lane may hold more than one complete unit)
        }
    } else
#endif
    {
        while (TRRE_WAVE_ANY(s.wp - s.fp >= kUnitBytes)) {
            if (s.wp - s.fp >= kUnitBytes) stage_store_own_unit(s);
        }
    }
    if (kAll) {
        // the rest: what is in the ring and the partial dword of the window
        stage_spill(s);
        const uint32_t end = stage_fill_end(s);
        stage_store_span(s, s.fp == 0 ? s.skip : s.fp, end);
        s.skip = end & (kUnitBytes - 1u);     // (a caller that goes on restarts with stage_begin at stage_out_ptr)
    }
}

template <bool kAll>
TRRE_HD void stage_flush(PStage&) {}           // nothing leaves a private region before the workgroup knows where it goes

// position after the first '\n' at or after lo - 1, anywhere in the input (reads through direct_load); >= hi: none
TRRE_HD int64_t first_line_start_safe(const ScanArgs& a, int64_t lo, int64_t hi) {
    if (lo <= a.vbeg) return a.vbeg;
    for (int64_t v = (lo - 1) & ~(int64_t)15; v < hi; v += 16) {
        const U128 q = direct_load(a, v);
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma clang loop unroll(disable)
        for (int k = 0; k < 16; ++k)
            if (v + k >= lo - 1 && ((wd[k >> 2] >> (8 * (k & 3))) & 0xffu) == 0x0au) return v + k + 1;
    }
    return hi;
}

// kG16 (count and emit passes of small tables): the walk uses the 16-byte entries T.g16 (front.hpp) —
// bytes to append and their count come ready-made (v_perm selector, count field), rows are byte offsets.
// kSym (guided families): a transition's column is the symbol the backward pass left at the byte's position
// (a.sym_v0) instead of the byte's class.
template <int kMode, bool kG16 = false, bool kSym = false>
TRRE_HD void stream_direct_lane(const ScanArgs& a, const StreamView& T, uint32_t n_cls, int64_t lane, int64_t lane_bytes,
                                uint8_t* ring, uint64_t out_base, DirectLane& L, uint32_t& status, uint32_t* wave_scratch = nullptr) {
    const uint32_t rs = kG16 ? 16u : 1u;                              // row unit
    const uint32_t done_row = kDoneState * n_cls * rs;
    int64_t lo = lane * lane_bytes, hi = lo + lane_bytes;
    if (hi > a.vend) hi = a.vend;
    // end of the lane's sub-range as a 32-bit offset from lo (the count and emit passes stop branch-free)
    const uint32_t rhi = (uint32_t)(hi > lo ? hi - lo : 0);
    uint32_t row;
    if (lo >= hi) row = done_row;
    else if (lo < a.vbeg) row = kSkipState * n_cls * rs;             // filler then '\n' right before the input
    else row = (lo == a.vbeg || a.in_v0[lo - 1] == (uint8_t)'\n') ? 0u : kSkipState * n_cls * rs;

    // kMode 0: output cursor as a 32-bit offset from a 64-byte aligned base address, through a 64-byte ring
    uint8_t* obase = nullptr;
    uint32_t o = 0;
    if (kMode == 0) {
        const uintptr_t start = reinterpret_cast<uintptr_t>(a.out_v0 + lo);
        obase = reinterpret_cast<uint8_t*>(start & ~(uintptr_t)63);
        o = (uint32_t)(start & 63u);
    }
    const uint32_t o_of_lo = o;           // kMode 0: offset that corresponds to position lo
    uint32_t of = o;                      // everything below `of` has left the ring
    Stage S{};                            // kMode 2
    S.dbg = a.dbg;
    S.wsc = wave_scratch;
    if (kMode == 2) {
        if (a.lp_emit) {
            // no count pass: this lane's lines are written where they were read
            const int64_t fs = row == done_row ? hi : first_line_start_safe(a, lo, hi);
            if (fs >= hi) row = done_row;
            stage_begin(S, ring, a.out_v0 + fs);
        } else {
            stage_begin(S, ring, a.out + out_base);
        }
    }
    uint64_t cnt = 0;
    uint32_t seen = 0;

    const int64_t vlast = (a.vend - 1) & ~(int64_t)15;        // the last readable aligned block
    const int64_t slast = ((a.vend + 63) & ~(int64_t)63) - 16;   // the last block of symbols
    auto sym_at = [&](int64_t vn) -> U128 {
        if (!kSym) return U128{};
        return *reinterpret_cast<const U128*>(a.sym_v0 + (vn < slast ? vn : slast));
    };
    // One 16-byte block at a time from two alternating register buffers: a buffer is refilled right after
    // its block has been walked and used one block later, so the load has a whole block of walking to
    // land and no register copies (which would wait for it) are needed.  Away from the two ends of the
    // input a refill is a bare load; only waves that reach an end take the version that patches bytes.
    auto fetch = [&](int64_t vn) -> U128 {
        U128 q = *reinterpret_cast<const U128*>(a.in_v0 + (vn < vlast ? vn : vlast));
        if (TRRE_WAVE_ANY(vn < a.vbeg || vn + 16 > a.vend - 1)) q = direct_load(a, vn);
        return q;
    };
    // one dword (4 input bytes) at position v + 4 d of the block at v
    auto walk_dword = [&](const uint32_t w, const uint32_t sw, const int64_t v, const int d) {
        {
            const uint8_t kk[4] = {kSym ? (uint8_t)sw : T.cls[w & 0xffu], kSym ? (uint8_t)(sw >> 8) : T.cls[(w >> 8) & 0xffu],
                                   kSym ? (uint8_t)(sw >> 16) : T.cls[(w >> 16) & 0xffu], kSym ? (uint8_t)(sw >> 24) : T.cls[w >> 24]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint8_t c = (uint8_t)(w >> (8 * j));
                if (kG16) {
                    const U128 g = *reinterpret_cast<const U128*>(T.g16 + row + ((uint32_t)kk[j] << 4));
                    const uint32_t n = g.y & 7u;
                    if (kMode == 1) { cnt += n; seen |= g.y; }
                    else { stage_append4(S, perm_b32(w >> (8 * j), g.z, g.w), n); seen |= g.y; }
                    if (TRRE_WAVE_ANY(g.y & 128u)) {
                        if (g.y & 128u) {
                            // more than four bytes, or pooled text: from the 8-byte entry
                            const uint64_t e = T.ent[(row >> 4) + kk[j]];
                            const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
                            const uint32_t ol = str_olen(elo), cc = (elo >> 27) & 1u;
                            if (kMode == 1) {
                                cnt += (ol == 7u ? str_pool_len(T, ehi) : ol) + cc;
                            } else if (ol != 7u) {
                                stage_append(S, (uint64_t)ehi | (uint64_t)(cc ? c : 0u) << (8u * ol), ol + cc);
                            } else {
                                uint32_t len = ehi >> 24;
                                if (len <= 8u) {
                                    uint64_t text;
                                    if (T.pool_fast) __builtin_memcpy(&text, T.pool_fast + str_pool_off(ehi) + 4, 8);
                                    else __builtin_memcpy(&text, T.pool + str_pool_off(ehi) + 4, 8);
                                    stage_append_text_c(S, text, len, c, cc);
                                } else {
                                    const uint8_t* r = T.pool + str_pool_off(ehi);
                                    if (len == 255u) len = str_pool_len(T, ehi);
                                    stage_flush_solo(S);
                                    uint8_t* gp = stage_out_ptr(S);
                                    for (uint32_t i = 0; i < len; ++i) gp[i] = r[4 + i];
                                    stage_begin(S, S.buf, gp + len);
                                    stage_append(S, (uint64_t)(cc ? c : 0u), cc);
                                }
                            }
                        }
                    }
                    const uint32_t p1 = (uint32_t)(v - lo) + 4u * d + j + 1u;
                    row = ((g.y & 32u) && p1 >= rhi) ? done_row : g.x;
                    continue;
                }
                const uint64_t e = T.ent[row + kk[j]];
                const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
                const uint32_t ol = str_olen(elo), cc = (elo >> 27) & 1u;
                if (kMode == 1) {
                    uint32_t n = (ol == 7u ? ehi >> 24 : ol) + cc;
                    if (T.long_pool) {
                        if (TRRE_WAVE_ANY(ol == 7u && n - cc == 255u)) { if (ol == 7u && n - cc == 255u) n = str_pool_len(T, ehi) + cc; }
                    }
                    cnt += n;
                } else if (kMode == 2) {
                    // the common transition: at most 4 inline bytes, then maybe the input byte
                    const bool pooled = ol == 7u;
                    const uint32_t il = pooled ? 0u : ol;
                    const uint64_t v = (uint64_t)(pooled ? 0u : ehi) | (uint64_t)(cc && !pooled ? c : 0u) << (8u * il);
                    stage_append(S, v, pooled ? 0u : ol + cc);
                    if (TRRE_WAVE_ANY(pooled)) {
                        if (pooled) {
                            uint32_t len = ehi >> 24;
                            if (len <= 8u) {
                                // (pooled texts are longer than the 4 bytes an entry holds inline)
                                uint64_t text;
                                if (T.pool_fast) __builtin_memcpy(&text, T.pool_fast + str_pool_off(ehi) + 4, 8);
                                else __builtin_memcpy(&text, T.pool + str_pool_off(ehi) + 4, 8);
                                stage_append_text_c(S, text, len, c, cc);
                            } else {
                                // long replacement text: empty the staging buffer, write straight to memory
                                const uint8_t* r = T.pool + str_pool_off(ehi);
                                if (len == 255u) len = str_pool_len(T, ehi);
                                stage_flush_solo(S);
                                uint8_t* g = stage_out_ptr(S);
                                for (uint32_t i = 0; i < len; ++i) g[i] = r[4 + i];
                                stage_begin(S, S.buf, g + len);
                                stage_append(S, (uint64_t)(cc ? c : 0u), cc);
                            }
                        }
                    }
                } else {
                    uint32_t n = ol + cc;
                    if (n) ring[o & 63u] = ol ? (uint8_t)ehi : c;
                    if (TRRE_WAVE_ANY(n >= 2u)) {
                        if (n >= 2u) {
                            if (ol != 7u) {
                                uint32_t x = ehi >> 8;
                                for (uint32_t i = 1; i < ol; ++i) { ring[(o + i) & 63u] = (uint8_t)x; x >>= 8; }
                                if (cc) ring[(o + ol) & 63u] = c;
                            } else {
                                const uint8_t* r = T.pool + str_pool_off(ehi);
                                const uint32_t len = str_pool_len(T, ehi);
                                if (len <= 40u) {
                                    // replacement text of ordinary length goes through the ring like any output
                                    if (o - of + len + 1u > 60u) direct_flush<false>(obase, ring, of, o);
                                    for (uint32_t i = 0; i < len; i += 4) {                  // dword reads from the pool record
                                        uint32_t x = *reinterpret_cast<const uint32_t*>(r + 4 + i);
                                        const uint32_t m = len - i < 4u ? len - i : 4u;
                                        for (uint32_t b2 = 0; b2 < m; ++b2) { ring[(o + i + b2) & 63u] = (uint8_t)x; x >>= 8; }
                                    }
                                    if (cc) ring[(o + len) & 63u] = c;
                                } else {
                                    // very long replacement text: empty the ring, then write straight to HBM
                                    direct_flush<true>(obase, ring, of, o);
                                    for (uint32_t i = 0; i < len; ++i) obase[o + i] = r[4 + i];
                                    if (cc) obase[o + len] = c;
                                    of = o + len + cc;
                                }
                                n = len + cc;
                            }
                        }
                    }
                    o += n;
                }
                row = str_next(elo);
                seen |= elo;
                if (kMode == 0) {
                    if (elo & kStrEol) {
                        const int64_t p1 = v + 4 * d + j + 1;
                        const uint32_t sync = o_of_lo + (uint32_t)(p1 - lo);
                        if (o != sync) { o = sync; of = sync; }       // leaving SKIP (or after a NUL: launch is void)
                        if (p1 >= hi) row = done_row;
                    }
                } else {
                    // a record end at or beyond the end of the sub-range ends the lane (a lane walks far less
                    // than 4 GiB: its sub-range plus the rest of one line, or it stops at the end of the input)
                    const uint32_t p1 = (uint32_t)(v - lo) + 4u * d + j + 1u;
                    row = ((elo & kStrEol) && p1 >= rhi) ? done_row : row;
                }
            }
            if (kMode == 2) stage_flush<false>(S);         // (up to 9 bytes per transition: 63 + 4 * 9 fits the ring)
            if (kMode != 0) {
                // one dword at a time: interleaving the walks of several dwords only costs registers, and left
                // alone the compiler turns the running sums into trees evaluated at the end of the piece
                TRRE_PIN(seen);
                if (kMode == 1) TRRE_PIN(cnt);
                TRRE_SCHED_FENCE();
            }
        }
    };
    // One 16-byte block.  The emit pass walks its dwords (and a piece's blocks) with real loops: fully
    // unrolled, a piece was 19 000 instructions, several times the instruction cache; the count pass is
    // small enough to keep its dwords unrolled.
    auto walk = [&](const U128& cur, const U128& sy, const int64_t v) {
        if (kMode == 2) {
#pragma clang loop unroll(disable)
            for (int d = 0; d < 4; ++d)
                walk_dword(d == 0 ? cur.x : (d == 1 ? cur.y : (d == 2 ? cur.z : cur.w)), d == 0 ? sy.x : (d == 1 ? sy.y : (d == 2 ? sy.z : sy.w)), v, d);
        } else {
            walk_dword(cur.x, sy.x, v, 0);
            walk_dword(cur.y, sy.y, v, 1);
            walk_dword(cur.z, sy.z, v, 2);
            walk_dword(cur.w, sy.w, v, 3);
        }
        if (kMode == 0) direct_flush<false>(obase, ring, of, o);
    };
    if (kMode != 0) {
        // Count and emit passes: 64 bytes at a time.  A lane's four 16-byte loads of a piece are issued
        // together, one piece ahead: per-lane 16-byte loads spread over time fetch every 64-byte sector
        // four times (the lanes of a wave lie lane_bytes apart; measured 4.3 bytes of HBM reads per
        // input byte), four back-to-back loads of one sector fetch it once.
        // (named registers, selected by compares: an indexed array would be kept in scratch memory)
        U128 c0 = direct_load(a, lo), c1 = direct_load(a, lo + 16), c2 = direct_load(a, lo + 32), c3 = direct_load(a, lo + 48);
        U128 s0 = sym_at(lo), s1 = sym_at(lo + 16), s2 = sym_at(lo + 32), s3 = sym_at(lo + 48);
        for (int64_t v = lo;; v += 64) {
            if (!TRRE_WAVE_ANY(row != done_row)) break;
            const int64_t vn = v + 64;
            const int64_t x0 = vn < vlast ? vn : vlast, x1 = vn + 16 < vlast ? vn + 16 : vlast,
                          x2 = vn + 32 < vlast ? vn + 32 : vlast, x3 = vn + 48 < vlast ? vn + 48 : vlast;
            U128 n0 = *reinterpret_cast<const U128*>(a.in_v0 + x0), n1 = *reinterpret_cast<const U128*>(a.in_v0 + x1),
                 n2 = *reinterpret_cast<const U128*>(a.in_v0 + x2), n3 = *reinterpret_cast<const U128*>(a.in_v0 + x3);
            if (TRRE_WAVE_ANY(vn < a.vbeg || vn + 64 > a.vend - 1)) {
                n0 = direct_load(a, vn); n1 = direct_load(a, vn + 16); n2 = direct_load(a, vn + 32); n3 = direct_load(a, vn + 48);
            }
            const U128 t0 = sym_at(vn), t1 = sym_at(vn + 16), t2 = sym_at(vn + 32), t3 = sym_at(vn + 48);
#pragma clang loop unroll(disable)
            for (int q = 0; q < 4; ++q) {
                U128 b, y{};
                b.x = q == 0 ? c0.x : (q == 1 ? c1.x : (q == 2 ? c2.x : c3.x));
                b.y = q == 0 ? c0.y : (q == 1 ? c1.y : (q == 2 ? c2.y : c3.y));
                b.z = q == 0 ? c0.z : (q == 1 ? c1.z : (q == 2 ? c2.z : c3.z));
                b.w = q == 0 ? c0.w : (q == 1 ? c1.w : (q == 2 ? c2.w : c3.w));
                if (kSym) {
                    y.x = q == 0 ? s0.x : (q == 1 ? s1.x : (q == 2 ? s2.x : s3.x));
                    y.y = q == 0 ? s0.y : (q == 1 ? s1.y : (q == 2 ? s2.y : s3.y));
                    y.z = q == 0 ? s0.z : (q == 1 ? s1.z : (q == 2 ? s2.z : s3.z));
                    y.w = q == 0 ? s0.w : (q == 1 ? s1.w : (q == 2 ? s2.w : s3.w));
                }
                walk(b, y, v + 16 * q);
            }
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            if (kSym) { s0 = t0; s1 = t1; s2 = t2; s3 = t3; }
        }
    } else {
        U128 blk0 = direct_load(a, lo), blk1 = direct_load(a, lo + 16);
        U128 sy0 = sym_at(lo), sy1 = sym_at(lo + 16);
        for (int64_t v = lo; TRRE_WAVE_ANY(row != done_row); v += 32) {
            walk(blk0, sy0, v);
            blk0 = fetch(v + 32);
            sy0 = sym_at(v + 32);
            if (!TRRE_WAVE_ANY(row != done_row)) break;
            walk(blk1, sy1, v + 16);
            blk1 = fetch(v + 48);
            sy1 = sym_at(v + 48);
        }
    }
    if (kMode == 0) direct_flush<true>(obase, ring, of, o);
    if (kMode == 2) stage_flush<true>(S);
    if ((kMode == 0 || (kMode == 2 && a.lp_emit)) && (seen & (kG16 ? 8u : kStrNul))) status |= kStNul;
    if (kMode == 1 && (seen & (kG16 ? 64u : kStrOvf))) status |= kStOverflow;     // bounded fold: the launch is void
    if ((kMode != 2 || a.lp_emit) && (seen & (kG16 ? 16u : kStrDiv))) status |= kStDiverge;      // guided tables: the reference's search never returns
    L.count = cnt;
}

constexpr uint32_t kStEditOverflow = 1u << 6;   // a list is full (the copy form's events, a search's stack): the launch is void, another family or a larger tier runs

// =============================================================================================
// Small tables (16-byte entries, whole table in LDS): the count and emit passes written for instruction
// count — these passes are bound by VALU issue (a wave64 integer instruction occupies its SIMD for 4
// cycles), not by memory.  Differences from stream_direct_lane<.., kG16>:
//   * pieces that lie wholly inside the lane's sub-range are walked without any end-of-lane test (no lane
//     of the wave can finish there); only the pieces of the tail — the rest of the lane's last line —
//     carry it;
//   * the count pass sums the entries' byte counts per dword in 32 bits and looks at the "slow" flag once
//     per dword (a dword that met one is simply counted again the careful way);
//   * the emit pass appends through a 64-bit window that moves on by at most one dword per transition
//     (an entry of this form emits at most 4 bytes): OR, store, one variable 64-bit shift, no selects;
//   * tables without slow entries (kHasSlow = false: no transition emits more than 4 bytes) have no slow
//     path at all.
// =============================================================================================
// kSym: 0 columns are byte classes; 1 / 2 (guided families) columns are the symbols the backward pass left, one per
// byte / packed two per byte (backward DFAs of at most 16 states: half the symbol traffic).
// (Rounds 3 and 4 had two more modes of this walk — a record pass that listed the edits of every 64-byte piece for a patch pass, and a mark
// pass that listed a lane's edits for a wave-cooperative splice: ONE walk instead of two.  Both were correct and both lost to the count /
// emit pair on small tables — listing the edits doubles the count walk: DESIGN.md §4.5 — and were removed in round 5.)
struct FbCopyArgs;
TRRE_HD uint32_t* copy_event_row(const FbCopyArgs& ca, int64_t lane);
TRRE_HD uint32_t copy_event_cap(const FbCopyArgs& ca);
TRRE_HD uint32_t* copy_lane_hdr(const FbCopyArgs& ca, int64_t lane);
constexpr int kMarkStage = 16;                 // events a lane can collect per 64 input bytes (LDS, 17 dwords apart)
constexpr int kMarkStageStride = 17;
template <int kMode, int kSym, bool kHasSlow>
TRRE_HD void g16_lane(const ScanArgs& a, const StreamView& T, uint32_t n_cls, int64_t lane, int64_t lane_bytes, uint8_t* ring,
                      uint64_t out_base, DirectLane& L, uint32_t& status, uint32_t* wave_scratch = nullptr) {
    // kMode 3 (one_block.hpp): ONE walk — an emit walk of exactly [lo, hi) into the lane's private LDS region `ring` (out_base: its size),
    // from the state L.entry (kOneGuess: guessed from a.spec_look bytes of context, like the count pass of the exact sub-ranges); leaves
    // L.count, L.entry / L.known, L.exit
    static_assert(kMode == 1 || kMode == 2 || kMode == 3, "count, emit or both at once");
    const uint32_t done_row = kDoneState * n_cls * 16u;
    const int64_t lo = lane * lane_bytes;
    int64_t hi = lo + lane_bytes;
    if (hi > a.vend) hi = a.vend;
    const uint32_t rhi = (uint32_t)(hi > lo ? hi - lo : 0);
    uint32_t row;
    const uint32_t exact = kMode == 3 ? 1u : a.exact;                // (uniform)
    // the one-pass walk asks for everything it reads at once — the first piece and the 32 bytes before it (a tile's time is
    // made of memory latencies: one round trip instead of five)
    U128 c0{}, c1{}, c2{}, c3{};
    if constexpr (kMode == 3) {
        U128 wb0{}, wb1{}, ws0{}, ws1{};
        const bool warm = lo < hi && lo > a.vbeg && L.entry == kOneGuess;
        // (direct_load's care for the two ends of the input is a few hundred instructions when it is inlined six times — a third of what
        // a lane of 128 bytes has to do at all: a wave that lies inside the input loads plainly)
        if (TRRE_WAVE_ALL(lo - 32 >= a.vbeg && lo + 64 < a.vend - 1)) {
            if (warm) {
                wb0 = *reinterpret_cast<const U128*>(a.in_v0 + lo - 32); wb1 = *reinterpret_cast<const U128*>(a.in_v0 + lo - 16);
            }
            c0 = *reinterpret_cast<const U128*>(a.in_v0 + lo); c1 = *reinterpret_cast<const U128*>(a.in_v0 + lo + 16);
            c2 = *reinterpret_cast<const U128*>(a.in_v0 + lo + 32); c3 = *reinterpret_cast<const U128*>(a.in_v0 + lo + 48);
        } else {
            if (warm) { wb0 = direct_load_cold(a.in_v0, a.vbeg, a.vend, lo - 32); wb1 = direct_load_cold(a.in_v0, a.vbeg, a.vend, lo - 16); }
            c0 = direct_load_cold(a.in_v0, a.vbeg, a.vend, lo); c1 = direct_load_cold(a.in_v0, a.vbeg, a.vend, lo + 16);
            c2 = direct_load_cold(a.in_v0, a.vbeg, a.vend, lo + 32); c3 = direct_load_cold(a.in_v0, a.vbeg, a.vend, lo + 48);
        }
        if (warm) {
            if (kSym == 1) { ws0 = *reinterpret_cast<const U128*>(a.sym_v0 + lo - 32); ws1 = *reinterpret_cast<const U128*>(a.sym_v0 + lo - 16); }
            if (kSym == 2) ws0 = *reinterpret_cast<const U128*>(a.sym_v0 + ((lo - 32) >> 1));
        }
        L.known = 1u;
        if (lo >= hi) row = done_row;
        else if (L.entry != kOneGuess) { row = L.entry; L.known = 0u; }
        else if (lo < a.vbeg) row = kSkipState * n_cls * 16u;
        else if (lo == a.vbeg) row = 0u;
        else {
            // the state at lo: the root state a.spec_look (16 or 32) bytes back, walked up to lo.  A '\n' among those bytes makes it exact
            // whatever the walk began in (every state goes to the root there: meta bit 5), and so does the start of the input; else it is
            // a guess — transducers of this kind forget — that the caller checks against the lane before.
            const int64_t s0 = lo - (int64_t)a.spec_look > a.vbeg ? lo - (int64_t)a.spec_look : a.vbeg;
            uint32_t r = 0u, seen_eol = s0 == a.vbeg ? 32u : 0u;
            // one 16-byte block at v: its bytes w, its symbols y (per byte) / yn (packed)
            auto warm_block = [&](const int64_t v, const U128& w, const U128& y, const uint64_t yn) {
                const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
                const uint32_t yd[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    uint32_t kk;
                    if (kSym == 2) kk = (uint32_t)(yn >> (4 * k)) & 15u;
                    else if (kSym == 1) kk = (yd[k >> 2] >> (8 * (k & 3))) & 0xffu;
                    else kk = T.cls[(wd[k >> 2] >> (8 * (k & 3))) & 0xffu];
                    const uint64_t g = *reinterpret_cast<const uint64_t*>(T.g16 + r + (kk << 4));
                    const bool take = v + k >= s0;
                    r = take ? (uint32_t)g : r;
                    seen_eol |= take ? (uint32_t)(g >> 32) : 0u;
                }
            };
            if (TRRE_WAVE_ANY(lo - 16 > s0)) warm_block(lo - 32, wb0, ws0, (uint64_t)ws0.y << 32 | ws0.x);
            warm_block(lo - 16, wb1, ws1, (uint64_t)ws0.w << 32 | ws0.z);
            row = r;
            L.known = (seen_eol >> 5) & 1u;
        }
    } else {
    if (lo >= hi) row = done_row;
    else if (lo < a.vbeg) row = kSkipState * n_cls * 16u;             // filler then '\n' right before the input
    else row = (lo == a.vbeg || a.in_v0[lo - 1] == (uint8_t)'\n') ? 0u : kSkipState * n_cls * 16u;
    }
    uint32_t xrow = row;                                             // exact sub-ranges: the state at hi
    if (kMode != 3 && exact && lo < hi) {
        if (exact >= 2u) {
            row = a.entry_rows[lane] & ~1u;
        } else if (lo > a.vbeg && row != 0u) {
            // the state at lo: from the last line start among the kSpecLook bytes before lo (exact), else a guess — the root state
            // kSpecLook bytes back, walked up to lo (transducers of this kind forget: the guess is right unless the bytes in between
            // leave a memory of what was before them; k_spec_verify checks it against the lane before)
            // (16 bytes at a time: the search for the line start first — highest block first —, then the walk; lo is a multiple of 16 and
            // direct_load shows a '\n' right before the input)
            int64_t s0 = lo - (int64_t)a.spec_look > a.vbeg ? lo - (int64_t)a.spec_look : a.vbeg;
            uint32_t known = s0 == a.vbeg ? 1u : 0u;
            for (int64_t v = lo - 16; v + 16 > s0; v -= 16) {
                const U128 q = direct_load(a, v);
                const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
                int found = -1;
                for (int d = 3; d >= 0 && found < 0; --d) {
                    const uint32_t x = wd[d] ^ 0x0a0a0a0au;
                    if ((x - 0x01010101u) & ~x & 0x80808080u) {          // some byte of the dword is a '\n' (which one: looked at byte by byte)
                        for (int k = 3; k >= 0 && found < 0; --k)
                            if (((wd[d] >> (8 * k)) & 0xffu) == 0x0au && v + 4 * d + k >= s0 - 1) found = 4 * d + k;
                    }
                }
                if (found >= 0) { s0 = v + found + 1; known = 1u; break; }
            }
            uint32_t r = 0u;
            for (int64_t v = s0 & ~(int64_t)15; v < lo; v += 16) {
                const U128 q = direct_load(a, v);
                const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
                U128 y{};
                if (kSym == 1) y = *reinterpret_cast<const U128*>(a.sym_v0 + v);
                uint64_t yn = 0;
                if (kSym == 2) yn = *reinterpret_cast<const uint64_t*>(a.sym_v0 + (v >> 1));
                const uint32_t yd[4] = {y.x, y.y, y.z, y.w};
#pragma clang loop unroll(disable)
                for (int k = 0; k < 16; ++k) {
                    if (v + k < s0) continue;
                    uint32_t kk;
                    if (kSym == 2) kk = (uint32_t)(yn >> (4 * k)) & 15u;
                    else if (kSym == 1) kk = (yd[k >> 2] >> (8 * (k & 3))) & 0xffu;
                    else kk = T.cls[(wd[k >> 2] >> (8 * (k & 3))) & 0xffu];
                    r = *reinterpret_cast<const uint32_t*>(T.g16 + r + (kk << 4));
                }
            }
            row = r;
            a.entry_rows[lane] = r | known;
        } else {
            a.entry_rows[lane] = row | 1u;                           // the buffer's first lane, or a line starts exactly at lo
        }
        xrow = row;
    }
    if (kMode == 3) { L.entry = row; xrow = row; }
    std::conditional_t<kMode == 3, PStage, Stage> S{};
    S.dbg = a.dbg;
    S.wsc = wave_scratch;
    if (kMode == 3) {
        stage_begin(S, ring, nullptr);
        if constexpr (kMode == 3) S.lim = (uint32_t)out_base - 4u;
    }
    if (kMode == 2) {
        if (a.lp_emit) {
            // no count pass: this lane's lines are written where they were read
            const int64_t fs = row == done_row ? hi : first_line_start_safe(a, lo, hi);
            if (fs >= hi) row = done_row;
            stage_begin(S, ring, a.out_v0 + fs);
        } else {
            stage_begin(S, ring, a.out + out_base);
        }
    }
    uint64_t cnt = 0;
    uint32_t seen = 0;
    const int64_t vlast = (a.vend - 1) & ~(int64_t)15;               // the last readable aligned block
    // symbols: per byte — 16 bytes per 16-byte block; packed — 16 bytes per 32 input bytes (fetched for offsets 0 and 32 of a piece)
    const int64_t slast = kSym == 2 ? (((a.vend + 127) & ~(int64_t)127) >> 1) - 16 : ((a.vend + 63) & ~(int64_t)63) - 16;
    auto sym_at = [&](int64_t vn) -> U128 {
        if (!kSym) return U128{};
        const int64_t at = kSym == 2 ? vn >> 1 : vn;
        return *reinterpret_cast<const U128*>(a.sym_v0 + (at < slast ? at : slast));
    };
    // the careful version of one transition (a "slow" entry: more than 4 bytes or pooled text, from the 8-byte entry)
    auto slow_count = [&](uint32_t r, uint32_t k) -> uint32_t {
        const uint64_t e = T.ent[(r >> 4) + k];
        const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
        const uint32_t ol = str_olen(elo);
        return (ol == 7u ? str_pool_len(T, ehi) : ol) + ((elo >> 27) & 1u);
    };
    auto slow_emit = [&](uint32_t r, uint32_t k, uint8_t c) {
        const uint64_t e = T.ent[(r >> 4) + k];
        const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
        const uint32_t ol = str_olen(elo), cc = (elo >> 27) & 1u;
        if (ol != 7u) {
            stage_append(S, (uint64_t)ehi | (uint64_t)(cc ? c : 0u) << (8u * ol), ol + cc);
            return;
        }
        uint32_t len = ehi >> 24;
        if (len <= 8u) {
            uint64_t text;
            if (T.pool_fast) __builtin_memcpy(&text, T.pool_fast + str_pool_off(ehi) + 4, 8);
            else __builtin_memcpy(&text, T.pool + str_pool_off(ehi) + 4, 8);
            stage_append_text_c(S, text, len, c, cc);
            return;
        }
        // long replacement text: empty the staging buffer, write straight to memory
        const uint8_t* rec = T.pool + str_pool_off(ehi);
        if (len == 255u) len = str_pool_len(T, ehi);
        if constexpr (kMode == 3) {
            status |= kStOneVoid;                                    // (no memory to write to yet: the pair takes the buffer)
        } else {
            stage_flush_solo(S);
            uint8_t* gp = stage_out_ptr(S);
            for (uint32_t i = 0; i < len; ++i) gp[i] = rec[4 + i];
            stage_begin(S, S.buf, gp + len);
            stage_append(S, (uint64_t)(cc ? c : 0u), cc);
        }
    };
    // one dword (4 input bytes) whose first byte lies rp bytes into the sub-range; kEnd: a lane may finish in it
    auto dword = [&](auto end_tag, const uint32_t w, const uint32_t sw, const uint32_t rp) {
        constexpr bool kEnd = decltype(end_tag)::value;
        uint32_t kk[4];
        if (kSym == 2) { kk[0] = sw & 15u; kk[1] = (sw >> 4) & 15u; kk[2] = (sw >> 8) & 15u; kk[3] = (sw >> 12) & 15u; }
        else if (kSym == 1) { kk[0] = sw & 0xffu; kk[1] = (sw >> 8) & 0xffu; kk[2] = (sw >> 16) & 0xffu; kk[3] = sw >> 24; }
        else { kk[0] = T.cls[w & 0xffu]; kk[1] = T.cls[(w >> 8) & 0xffu]; kk[2] = T.cls[(w >> 16) & 0xffu]; kk[3] = T.cls[w >> 24]; }
        if (kMode == 1) {
            const uint32_t row0 = row;
            uint32_t c = 0, fl = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t g = *reinterpret_cast<const uint64_t*>(T.g16 + row + (kk[j] << 4));       // {next row, meta}
                const uint32_t meta = (uint32_t)(g >> 32);
                if (kEnd && exact) {                                     // exact sub-ranges: the bytes from hi on are the next lane's
                    const bool take = rp + (uint32_t)j < rhi;
                    c += take ? (meta & 7u) : 0u;
                    fl |= take ? meta : 0u;
                    row = take ? (uint32_t)g : row;
                    continue;
                }
                c += meta & 7u;
                fl |= meta;
                row = (kEnd && (meta & 32u) && rp + (uint32_t)j + 1u >= rhi) ? done_row : (uint32_t)g;
            }
            if (kHasSlow) {
                if (TRRE_WAVE_ANY(fl & 128u)) {
                    if (fl & 128u) {              // count this dword again, slow entries from their 8-byte form
                        row = row0;
                        c = 0;
                        for (int j = 0; j < 4; ++j) {
                            const uint64_t g = *reinterpret_cast<const uint64_t*>(T.g16 + row + (kk[j] << 4));
                            const uint32_t meta = (uint32_t)(g >> 32);
                            if (kEnd && exact) {
                                if (rp + (uint32_t)j < rhi) { c += (meta & 128u) ? slow_count(row, kk[j]) : (meta & 7u); row = (uint32_t)g; }
                                continue;
                            }
                            c += (meta & 128u) ? slow_count(row, kk[j]) : (meta & 7u);
                            row = (kEnd && (meta & 32u) && rp + (uint32_t)j + 1u >= rhi) ? done_row : (uint32_t)g;
                        }
                    }
                }
            }
            seen |= fl;
            cnt += c;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const U128 g = *reinterpret_cast<const U128*>(T.g16 + row + (kk[j] << 4));
                if (kEnd && exact) {                                     // exact sub-ranges: the bytes from hi on are the next lane's
                    const bool take = rp + (uint32_t)j < rhi;
                    stage_append_bits(S, take ? perm_b32(w >> (8 * j), g.z, g.w) : 0u, take ? g.y >> 24 : 0u);
                    seen |= take ? g.y : 0u;
                    if (kHasSlow) {
                        if (TRRE_WAVE_ANY(take && (g.y & 128u))) {
                            if (take && (g.y & 128u)) slow_emit(row, kk[j], (uint8_t)(w >> (8 * j)));
                        }
                    }
                    row = take ? g.x : row;
                    continue;
                }
                stage_append_bits(S, perm_b32(w >> (8 * j), g.z, g.w), g.y >> 24);      // ([31:24] of the entry: 8 x the bytes it emits)
                seen |= g.y;
                if (kHasSlow) {
                    if (TRRE_WAVE_ANY(g.y & 128u)) {
                        if (g.y & 128u) slow_emit(row, kk[j], (uint8_t)(w >> (8 * j)));
                    }
                }
                row = (kEnd && (g.y & 32u) && rp + (uint32_t)j + 1u >= rhi) ? done_row : g.x;
            }
        }
    };
    // Interior pieces of tables that have the pair form: two input bytes per table step (half the steps of a pass that is
    // bound by instruction issue).  A pair whose bytes do not fit its entry ("slow") is walked as two single steps.
    const uint32_t pair_mul = 2u * n_cls;         // pair row offset = 16-byte row offset * 2 n_cls
    uint32_t seen2 = 0;                           // metas of pair entries (their flag bits differ from the 16-byte form's)
    auto single_emit = [&](const uint32_t wj, const uint32_t k) {       // one byte through the 16-byte entries, no end-of-lane test
        const U128 g = *reinterpret_cast<const U128*>(T.g16 + row + (k << 4));
        stage_append_bits(S, perm_b32(wj, g.z, g.w), g.y >> 24);
        seen |= g.y;
        if (kHasSlow) {
            if (TRRE_WAVE_ANY(g.y & 128u)) {
                if (g.y & 128u) slow_emit(row, k, (uint8_t)wj);
            }
        }
        row = g.x;
    };
    auto dword_pairs = [&](const uint32_t w, const uint32_t sw, const uint32_t rp) {
        uint32_t kk[4];
        if (kSym == 2) { kk[0] = sw & 15u; kk[1] = (sw >> 4) & 15u; kk[2] = (sw >> 8) & 15u; kk[3] = (sw >> 12) & 15u; }
        else if (kSym == 1) { kk[0] = sw & 0xffu; kk[1] = (sw >> 8) & 0xffu; kk[2] = (sw >> 16) & 0xffu; kk[3] = sw >> 24; }
        else { kk[0] = T.cls[w & 0xffu]; kk[1] = T.cls[(w >> 8) & 0xffu]; kk[2] = T.cls[(w >> 16) & 0xffu]; kk[3] = T.cls[w >> 24]; }
        if (kMode == 1) {
            const uint32_t row0 = row;
            uint32_t c = 0, fl = 0, r = row;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint64_t g = *reinterpret_cast<const uint64_t*>(T.p32 + mul24(r, pair_mul) + ((mul24(kk[2 * h], n_cls) + kk[2 * h + 1]) << 5));
                const uint32_t meta = (uint32_t)(g >> 32);
                c += meta & 15u;
                fl |= meta;
                r = (uint32_t)g;
            }
            if (T.p32_slow) {
                if (TRRE_WAVE_ANY(fl & 128u)) {
                    if (fl & 128u) {              // this dword again, byte by byte
                        row = row0;
                        dword(std::false_type{}, w, sw, 0u);
                        return;
                    }
                }
            }
            row = r;
            seen2 |= fl;
            cnt += c;
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint8_t* ep = T.p32 + mul24(row, pair_mul) + ((mul24(kk[2 * h], n_cls) + kk[2 * h + 1]) << 5);
                const U128 e = *reinterpret_cast<const U128*>(ep);
                const uint32_t ws = w >> (16 * h);
                bool as_pair = true;
                if (T.p32_slow) {
                    if (TRRE_WAVE_ANY(e.y & 128u)) {
                        if (e.y & 128u) {
                            single_emit(ws, kk[2 * h]);
                            single_emit(ws >> 8, kk[2 * h + 1]);
                            as_pair = false;
                        }
                    }
                }
                if (as_pair) {
                    const uint32_t n = e.y & 15u;
                    uint32_t hi4 = 0;
                    if (TRRE_WAVE_ANY(n > 4u)) {
                        const uint64_t e2 = *reinterpret_cast<const uint64_t*>(ep + 16);
                        hi4 = perm_b32(ws, (uint32_t)e2, (uint32_t)(e2 >> 32));
                    }
                    // ([31:24] of the entry: 8 x the bytes of its first half that count)
                    stage_append_bits(S, perm_b32(ws, e.z, e.w), e.y >> 24);
                    if (TRRE_WAVE_ANY(n > 4u)) stage_append_n4(S, hi4, n > 4u ? n - 4u : 0u);
                    seen2 |= e.y;
                    row = e.x;
                }
            }
        }
    };
    // one 16-byte block; y: its symbols — four dwords (per byte) or the two dwords x, y (packed)
    auto block = [&](auto end_tag, const U128& b, const U128& y, const uint32_t rp) {
        constexpr bool kEnd = decltype(end_tag)::value;
        const uint32_t y0 = kSym == 2 ? (y.x & 0xffffu) : y.x, y1 = kSym == 2 ? (y.x >> 16) : y.y,
                       y2 = kSym == 2 ? (y.y & 0xffffu) : y.z, y3 = kSym == 2 ? (y.y >> 16) : y.w;
        // (between two flushes at most 65 bytes may arrive: 8 transitions of up to 5 bytes, or 4 of up to 9 with slow entries)
        if (!kEnd && T.p32) {
            dword_pairs(b.x, y0, rp);
            if (kMode == 2 && (kHasSlow || T.p32_slow)) stage_flush<false>(S);
            dword_pairs(b.y, y1, rp + 4u);
            if (kMode == 2) stage_flush<false>(S);
            dword_pairs(b.z, y2, rp + 8u);
            if (kMode == 2 && (kHasSlow || T.p32_slow)) stage_flush<false>(S);
            dword_pairs(b.w, y3, rp + 12u);
        } else {
            dword(end_tag, b.x, y0, rp);
            if (kMode == 2 && kHasSlow) stage_flush<false>(S);
            dword(end_tag, b.y, y1, rp + 4u);
            if (kMode == 2) stage_flush<false>(S);
            dword(end_tag, b.z, y2, rp + 8u);
            if (kMode == 2 && kHasSlow) stage_flush<false>(S);
            dword(end_tag, b.w, y3, rp + 12u);
        }
        if (kMode == 2) stage_flush<false>(S);
        TRRE_PIN(seen);
        TRRE_PIN(seen2);
        if (kMode == 1) TRRE_PIN(cnt);
        TRRE_SCHED_FENCE();
    };
    // 64 bytes at a time; a lane's four 16-byte loads of a piece are issued together, one piece ahead
    if (kMode != 3) { c0 = direct_load(a, lo); c1 = direct_load(a, lo + 16); c2 = direct_load(a, lo + 32); c3 = direct_load(a, lo + 48); }
    U128 s0 = sym_at(lo), s1 = kSym == 2 ? sym_at(lo + 32) : sym_at(lo + 16), s2 = kSym == 1 ? sym_at(lo + 32) : U128{},
         s3 = kSym == 1 ? sym_at(lo + 48) : U128{};
    // the symbols of block q of the piece
    auto sym_of = [&](int q) -> U128 {
        U128 y{};
        if (kSym == 1) {
            y.x = q == 0 ? s0.x : (q == 1 ? s1.x : (q == 2 ? s2.x : s3.x));
            y.y = q == 0 ? s0.y : (q == 1 ? s1.y : (q == 2 ? s2.y : s3.y));
            y.z = q == 0 ? s0.z : (q == 1 ? s1.z : (q == 2 ? s2.z : s3.z));
            y.w = q == 0 ? s0.w : (q == 1 ? s1.w : (q == 2 ? s2.w : s3.w));
        } else if (kSym == 2) {
            y.x = q == 0 ? s0.x : (q == 1 ? s0.z : (q == 2 ? s1.x : s1.z));
            y.y = q == 0 ? s0.y : (q == 1 ? s0.w : (q == 2 ? s1.y : s1.w));
        }
        return y;
    };
    for (int64_t v = lo;; v += 64) {
        if (!TRRE_WAVE_ANY(row != done_row)) break;
        const int64_t vn = v + 64;
        const int64_t x0 = vn < vlast ? vn : vlast, x1 = vn + 16 < vlast ? vn + 16 : vlast,
                      x2 = vn + 32 < vlast ? vn + 32 : vlast, x3 = vn + 48 < vlast ? vn + 48 : vlast;
        U128 n0 = *reinterpret_cast<const U128*>(a.in_v0 + x0), n1 = *reinterpret_cast<const U128*>(a.in_v0 + x1),
             n2 = *reinterpret_cast<const U128*>(a.in_v0 + x2), n3 = *reinterpret_cast<const U128*>(a.in_v0 + x3);
        if (TRRE_WAVE_ANY(vn < a.vbeg || vn + 64 > a.vend - 1)) {
            if (kMode == 3) {
                n0 = direct_load_cold(a.in_v0, a.vbeg, a.vend, vn); n1 = direct_load_cold(a.in_v0, a.vbeg, a.vend, vn + 16);
                n2 = direct_load_cold(a.in_v0, a.vbeg, a.vend, vn + 32); n3 = direct_load_cold(a.in_v0, a.vbeg, a.vend, vn + 48);
            } else {
                n0 = direct_load(a, vn); n1 = direct_load(a, vn + 16); n2 = direct_load(a, vn + 32); n3 = direct_load(a, vn + 48);
            }
        }
        const U128 t0 = sym_at(vn), t1 = kSym == 2 ? sym_at(vn + 32) : sym_at(vn + 16), t2 = kSym == 1 ? sym_at(vn + 32) : U128{},
                   t3 = kSym == 1 ? sym_at(vn + 48) : U128{};
        const uint32_t rp = (uint32_t)(v - lo);
        // (the one-pass walk owns exactly [lo, hi): a piece that ends AT hi is interior too)
        if (kMode == 3 ? TRRE_WAVE_ALL(rp + 64u <= rhi) : TRRE_WAVE_ALL(rp + 64u < rhi)) {
            // interior piece: a lane finishes at the first record end whose '\n' is the last byte of its sub-range or lies
            // beyond it, and every byte of this piece lies before that last byte
#pragma clang loop unroll(disable)
            for (int q = 0; q < 4; ++q) {
                U128 b;
                b.x = q == 0 ? c0.x : (q == 1 ? c1.x : (q == 2 ? c2.x : c3.x));
                b.y = q == 0 ? c0.y : (q == 1 ? c1.y : (q == 2 ? c2.y : c3.y));
                b.z = q == 0 ? c0.z : (q == 1 ? c1.z : (q == 2 ? c2.z : c3.z));
                b.w = q == 0 ? c0.w : (q == 1 ? c1.w : (q == 2 ? c2.w : c3.w));
                block(std::false_type{}, b, sym_of(q), rp + 16u * (uint32_t)q);
            }
        } else {
#pragma clang loop unroll(disable)
            for (int q = 0; q < 4; ++q) {
                U128 b;
                b.x = q == 0 ? c0.x : (q == 1 ? c1.x : (q == 2 ? c2.x : c3.x));
                b.y = q == 0 ? c0.y : (q == 1 ? c1.y : (q == 2 ? c2.y : c3.y));
                b.z = q == 0 ? c0.z : (q == 1 ? c1.z : (q == 2 ? c2.z : c3.z));
                b.w = q == 0 ? c0.w : (q == 1 ? c1.w : (q == 2 ? c2.w : c3.w));
                block(std::true_type{}, b, sym_of(q), rp + 16u * (uint32_t)q);
            }
        }
        if (exact) {
            if (rp + 64u >= rhi && row != done_row) { xrow = row; row = done_row; }     // exact sub-ranges: the lane ends at hi
        }
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        if (kSym) { s0 = t0; s1 = t1; s2 = t2; s3 = t3; }
    }
    if (kMode == 1 && exact && lo < hi) a.exit_rows[lane] = xrow;
    if (kMode == 2) stage_flush<true>(S);
    if (kMode == 2 && a.lp_emit && ((seen & 8u) || (seen2 & 256u))) status |= kStNul;
    if ((kMode == 1 || kMode == 3) && ((seen | seen2) & 64u)) status |= kStOverflow;            // bounded fold: the launch is void
    if ((kMode == 1 || kMode == 3 || a.lp_emit) && ((seen | seen2) & 16u)) status |= kStDiverge;   // guided tables: the reference's search never returns
    if constexpr (kMode == 3) {
        L.exit = xrow;
        stage_spill(S);                                              // the dword being filled
        cnt = stage_fill_end(S);
        if (S.wp > S.lim) status |= kStOneVoid;                      // the region could not hold it all
    }
    L.count = cnt;
}

// =============================================================================================
// Large tables in their fallback form (front.hpp, StreamTables::fb_*): the count and emit passes of a
// dictionary-like program with every per-byte lookup in LDS.  The 8-byte rows of such a program (~0.9 MB for 1000
// keys) are a gather through L1/L2 on every byte: the passes ran at the pace of the L1's tag lookups (one address per
// clock and CU; count 1.65 ms and emit 2.9 ms per GiB, whatever the occupancy).  Here a state travels as a descriptor
// {base of its own slots, base of its fallback row} and a step reads both candidate entries at once and keeps the own
// one if its tag matches: ONE LDS round trip per input byte.  (A first version — a record per state with a mask of its
// exceptional classes, then the entry — was two dependent reads and ~700 clocks per step: 1.16 ms count, 4.5 ms emit.)
// What a transition emits is a prefix of the lane's last 7 input bytes (kept in a register pair) or an owed
// replacement text (8 bytes from a small LDS table), then maybe the input byte or '\n'; it goes into the lane's ring by
// one unaligned 8-byte store (BStage).  The odd cell that is none of that leaves the fast path through an escape record
// in global memory.
// =============================================================================================
struct FbView {
    const uint8_t* cls;        // [256] (LDS)
    const uint64_t* comb;      // the slots (LDS)
    const uint64_t* lit;       // literal texts (LDS; emit pass)
    const uint16_t* lit_meta;  // per literal: length | input bytes it stands for << 8 (LDS; mark pass)
    const uint32_t* esc_slot;  // slots of the escape entries, ascending (global)
    const uint32_t* esc;       // their records, 4 words each (global)
    const uint8_t* pool;       // their texts (global)
    uint32_t n_esc;
    uint32_t start[3][2];      // root, SKIP, DONE: {descriptor, about bits}
};
// the mark pass's product (the copy form, below): events and lane headers
struct FbCopyArgs {
    uint32_t* events;          // [n_lanes][ev_cap]: a lane's events side by side (ev_cap: a multiple of 4 — the copy pass reads them 16 bytes at a time)
    uint32_t* lane_hdr;        // [n_lanes][4]: {events, first line start, end of the last line (offsets from the sub-range's start), -}
    uint32_t ev_cap;
};
TRRE_HD uint32_t* copy_event_row(const FbCopyArgs& ca, int64_t lane) { return ca.events + (size_t)lane * ca.ev_cap; }
TRRE_HD uint32_t copy_event_cap(const FbCopyArgs& ca) { return ca.ev_cap; }
TRRE_HD uint32_t* copy_lane_hdr(const FbCopyArgs& ca, int64_t lane) { return ca.lane_hdr + (size_t)lane * 4; }
// the record of the escape entry in slot `slot`
TRRE_HD const uint32_t* fb_esc_record(const FbView& T, uint32_t slot) {
    uint32_t lo = 0, hi = T.n_esc;
    while (lo + 1u < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (T.esc_slot[mid] <= slot) lo = mid; else hi = mid;
    }
    return T.esc + 4u * lo;
}
template <int kMode>
TRRE_HD void fb_lane(const ScanArgs& a, const FbView& T, int64_t lane, int64_t lane_bytes, uint8_t* ring, uint64_t out_base, DirectLane& L,
                     uint32_t& status, uint32_t* wave_scratch = nullptr, const FbCopyArgs* ca = nullptr) {
    static_assert(kMode == 1 || kMode == 2 || kMode == 3, "count, emit, or mark (count + the copy form's events)");
    const int64_t lo = lane * lane_bytes;
    int64_t hi = lo + lane_bytes;
    if (hi > a.vend) hi = a.vend;
    const uint32_t rhi = (uint32_t)(hi > lo ? hi - lo : 0);
    const uint32_t done_st = T.start[2][0], done_ab = T.start[2][1];
    int first;                                                         // 0 root, 1 SKIP, 2 DONE
    if (lo >= hi) first = 2;
    else if (lo < a.vbeg) first = 1;                                   // filler then '\n' right before the input
    else first = (lo == a.vbeg || a.in_v0[lo - 1] == (uint8_t)'\n') ? 0 : 1;
    uint32_t st = T.start[first][0], ab = T.start[first][1];          // the state's descriptor, and the entry bits that came with it
    BStage S{};
    S.dbg = a.dbg;
    S.wsc = wave_scratch;
    if (kMode == 2) stage_begin(S, ring, a.out + out_base);
    uint64_t cnt = 0;
    uint64_t hist = 0;                                                // the last 7 input bytes: byte 6 = the one before the current
    const int64_t vlast = (a.vend - 1) & ~(int64_t)15;               // the last readable aligned block
    // mark: the lane's events, where its first line starts and where its last line ends
    // (events are collected in the lane's stage — LDS, kMarkStage dwords — and leave for memory after every 64 input bytes:
    // a store to memory per event, some lane of the wave has one at almost every byte, held the walk up for the store's
    // round trip: 2.25 ms per GiB against 0.86 for the count walk alone)
    uint32_t* evp = nullptr;
    uint32_t* const stage0 = reinterpret_cast<uint32_t*>(ring);
    uint32_t* sp = stage0;
    uint32_t n_ev = 0, b_rel = 0, e_rel = 0, nul = 0, far = 0;
    int64_t delta = 0;
    if (kMode == 3) {
        evp = copy_event_row(*ca, lane);
        if (first == 1) b_rel = (uint32_t)(first_line_start_safe(a, lo, hi) - lo);
    }
    // an escape entry: the output spelled out in global memory (rare)
    auto esc_count = [&](uint32_t slot) -> uint32_t {
        const uint32_t* r = fb_esc_record(T, slot);
        return r[1] + r[2];
    };
    auto esc_emit = [&](uint32_t slot, uint32_t c) {
        const uint32_t* r = fb_esc_record(T, slot);
        const uint8_t* text = T.pool + r[0];
        const uint32_t len = r[1];
        stage_flush_solo(S);
        uint8_t* gp = stage_out_ptr(S);
        for (uint32_t i = 0; i < len; ++i) gp[i] = text[i];
        if (r[2]) gp[len] = (uint8_t)c;
        stage_begin(S, S.buf, gp + len + r[2]);
    };
    // one dword (4 input bytes) whose first byte lies rp bytes into the sub-range; kEnd: a lane may finish in it
    auto dword = [&](auto end_tag, const uint32_t w, const uint32_t rp) {
        constexpr bool kEnd = decltype(end_tag)::value;
        const uint32_t kk[4] = {T.cls[w & 0xffu], T.cls[(w >> 8) & 0xffu], T.cls[(w >> 16) & 0xffu], T.cls[w >> 24]};
        if (kMode == 3) nul |= (w - 0x01010101u) & ~w & 0x80808080u;          // a zero byte in these four
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k = kk[j], c = (w >> (8 * j)) & 0xffu;
            const uint32_t base = st & 0x3fffu, fbase = (st >> 14) & 0x3fffu;
            const uint64_t e1 = T.comb[base + k], e2 = T.comb[fbase + k];      // the state's own slot and its fallback row's: one round trip
            const bool mine = ((uint32_t)(e1 >> 32) & 0x3fffu) == base;
            const uint64_t e = mine ? e1 : e2;
            const uint32_t eh = (uint32_t)(e >> 32);
            const uint32_t owed = st >> 31;
            const uint32_t n_rec = mine ? 0u : ((st >> 28) & 7u) + owed;
            const bool esc = (eh & (kFbCc | kFbNl)) == (kFbCc | kFbNl);
            if (kMode == 1) {
                uint32_t add = n_rec + ((eh >> 14) & 7u) + ((eh >> 17) & 1u) + ((eh >> 18) & 1u);
                if (TRRE_WAVE_ANY(esc)) {
                    if (esc) add = n_rec + esc_count(mine ? base + k : fbase + k);
                }
                cnt += add;
            } else if (kMode == 3) {
                // an owed state left through its fallback row emits its text now (it stands for the bytes right before this
                // one).  No byte count here: every other transition copies what it reads, the events' texts are accounted
                // for when they leave the stage.
                bool ev = owed && !mine;
                uint32_t id = ab >> 20;
                if (TRRE_WAVE_ANY(esc)) {
                    if (esc) {
                        ev = true;
                        id = 0x8000u | (uint32_t)((fb_esc_record(T, mine ? base + k : fbase + k) - T.esc) >> 2);
                    }
                }
                if (ev) {
                    *sp = (rp + (uint32_t)j) | id << 16;
                    sp = sp + 1 < stage0 + (kMarkStage - 1) ? sp + 1 : stage0 + (kMarkStage - 1);      // (a full stage: the launch is void)
                }
                if (kEnd && ev && rp + (uint32_t)j > 0xffffu) far = 1;
            } else {
                const uint32_t about = ab >> 20;                      // of the current state: pending length, or the owed text's index
                const uint32_t n_tot = n_rec + ((eh >> 14) & 7u);     // 0..8 bytes of prefix (an escape entry has none of its own)
                const uint64_t text = T.lit[owed ? about : 0u];
                bstage_put8(S, owed ? text : hist >> (8u * (7u - (about & 7u))), n_tot);
                bstage_put1(S, (eh & kFbCc) ? c : (uint32_t)'\n', esc ? 0u : ((eh >> 17) | (eh >> 18)) & 1u);
                if (TRRE_WAVE_ANY(esc)) {
                    if (esc) esc_emit(mine ? base + k : fbase + k, c);
                }
            }
            hist = (hist >> 8) | (uint64_t)c << 48;
            const bool fin = kEnd && (eh & kFbEol) && rp + (uint32_t)j + 1u >= rhi;
            if (kMode == 3 && kEnd && fin && st != done_st) e_rel = rp + (uint32_t)j + 1u;
            st = fin ? done_st : (uint32_t)e;
            ab = fin ? done_ab : eh;
        }
    };
    auto block = [&](auto end_tag, const U128& b, const uint32_t rp) {
        // (between two flushes at most 36 bytes arrive: 4 transitions of up to 9 bytes.  Not unrolled: the rare paths —
        // escapes, the unit stores — would be there sixteen times, and the loop should stay in the instruction cache)
#pragma clang loop unroll(disable)
        for (int d = 0; d < 4; ++d) {
            dword(end_tag, d == 0 ? b.x : (d == 1 ? b.y : (d == 2 ? b.z : b.w)), rp + 4u * (uint32_t)d);
            if (kMode == 2) stage_flush<false>(S);
            if (kMode == 1) TRRE_PIN(cnt);
            TRRE_SCHED_FENCE();
        }
    };
    U128 c0 = direct_load(a, lo), c1 = direct_load(a, lo + 16), c2 = direct_load(a, lo + 32), c3 = direct_load(a, lo + 48);
    for (int64_t v = lo;; v += 64) {
        if (!TRRE_WAVE_ANY(st != done_st)) break;
        const int64_t vn = v + 64;
        const int64_t x0 = vn < vlast ? vn : vlast, x1 = vn + 16 < vlast ? vn + 16 : vlast,
                      x2 = vn + 32 < vlast ? vn + 32 : vlast, x3 = vn + 48 < vlast ? vn + 48 : vlast;
        U128 n0 = *reinterpret_cast<const U128*>(a.in_v0 + x0), n1 = *reinterpret_cast<const U128*>(a.in_v0 + x1),
             n2 = *reinterpret_cast<const U128*>(a.in_v0 + x2), n3 = *reinterpret_cast<const U128*>(a.in_v0 + x3);
        if (TRRE_WAVE_ANY(vn < a.vbeg || vn + 64 > a.vend - 1)) {
            n0 = direct_load(a, vn); n1 = direct_load(a, vn + 16); n2 = direct_load(a, vn + 32); n3 = direct_load(a, vn + 48);
        }
        const uint32_t rp = (uint32_t)(v - lo);
        if (TRRE_WAVE_ALL(rp + 64u < rhi)) {
            // interior piece: no lane of the wave can finish in it (g16_lane)
#pragma clang loop unroll(disable)
            for (int q = 0; q < 4; ++q) {
                U128 b;
                b.x = q == 0 ? c0.x : (q == 1 ? c1.x : (q == 2 ? c2.x : c3.x));
                b.y = q == 0 ? c0.y : (q == 1 ? c1.y : (q == 2 ? c2.y : c3.y));
                b.z = q == 0 ? c0.z : (q == 1 ? c1.z : (q == 2 ? c2.z : c3.z));
                b.w = q == 0 ? c0.w : (q == 1 ? c1.w : (q == 2 ? c2.w : c3.w));
                block(std::false_type{}, b, rp + 16u * (uint32_t)q);
            }
        } else {
#pragma clang loop unroll(disable)
            for (int q = 0; q < 4; ++q) {
                U128 b;
                b.x = q == 0 ? c0.x : (q == 1 ? c1.x : (q == 2 ? c2.x : c3.x));
                b.y = q == 0 ? c0.y : (q == 1 ? c1.y : (q == 2 ? c2.y : c3.y));
                b.z = q == 0 ? c0.z : (q == 1 ? c1.z : (q == 2 ? c2.z : c3.z));
                b.w = q == 0 ? c0.w : (q == 1 ? c1.w : (q == 2 ? c2.w : c3.w));
                block(std::true_type{}, b, rp + 16u * (uint32_t)q);
            }
        }
        if (kMode == 3) {
            // the piece's events, to the end of the lane's row
            const uint32_t n_loc = (uint32_t)(sp - stage0);
            if (n_loc >= (uint32_t)kMarkStage - 1u) far = 1;
            for (uint32_t k = 0; TRRE_WAVE_ANY(k < n_loc); ++k) {
                if (k < n_loc) {
                    const uint32_t ev = stage0[k], id = ev >> 16;
                    if (n_ev + k < ca->ev_cap) evp[k] = ev;
                    // what the text adds: its length - the input bytes it stands for
                    if (!(id & 0x8000u)) {
                        const uint32_t m = T.lit_meta[id];
                        delta += (int32_t)(m & 255u) - (int32_t)(m >> 8);
                    } else {
                        const uint32_t* r = T.esc + 4u * (id & 0x7fffu);
                        delta += (int32_t)r[1] - (int32_t)(r[3] & 255u);
                    }
                }
            }
            evp += n_loc;
            n_ev += n_loc;
            sp = stage0;
        }
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    if (kMode == 2) stage_flush<true>(S);
    if (kMode == 3) {
        if (n_ev > ca->ev_cap || far) status |= kStEditOverflow;
        if (nul) status |= kStNul;
        uint32_t* hdr = ca->lane_hdr + (size_t)lane * 4;
        hdr[0] = n_ev < ca->ev_cap ? n_ev : ca->ev_cap;
        hdr[1] = b_rel;
        hdr[2] = b_rel < rhi ? e_rel : b_rel;          // (no line starts in the sub-range: the lane has nothing to copy)
        hdr[3] = 0;
        cnt = 0;
        if (b_rel < rhi && e_rel > b_rel) {
            const int64_t end = lo + (int64_t)e_rel < a.vend ? lo + (int64_t)e_rel : a.vend;
            const int64_t bytes = end - (lo + (int64_t)b_rel) + delta;
            cnt = bytes > 0 ? (uint64_t)bytes : 0;
        }
    }
    (void)status;
    L.count = cnt;
}

// =============================================================================================
// The copy form's first pass on the MARK FORM of the comb (round 4; front.hpp: StreamTables::fb_comb4): the same walk and
// the same product as fb_lane<3> — events, lane header, the lane's output size — from 32-bit entries.  A step reads the
// state's own slot and its fallback row's (two 4-byte LDS reads, half the bytes of the 8-byte comb), "mine" is one byte
// compare (the slot's class against the byte's), the event word is the state itself next to the position (an owed state's
// base field names its text) and is stored at the stage's fill position whatever happens — the position moves on only
// for an owed state: no branch in the step.  Escape entries are looked for once per dword (on the cfg 5 corpus one dword in
// 10 000 holds one); a dword that met one is walked again the careful way.
// =============================================================================================
struct Fb4View {
    const uint8_t* cls4;         // [256] 4 x class (LDS)
    const uint32_t* comb4;       // the slots and the never-owned stretch behind them (LDS)
    const uint32_t* dense4;      // [dense states][32] (LDS)
    const uint16_t* lit_meta;    // per literal: length | input bytes it stands for << 8 (LDS)
    const uint16_t* dense_base;  // per dense state: its base in the comb (global; escapes only)
    const uint32_t* esc_slot;    // slots of the escape entries, ascending (global)
    const uint32_t* esc;         // their records, 4 words each (global)
    uint32_t n_esc, pad;
    uint32_t start[3];           // root, SKIP, DONE
};
constexpr uint32_t kFb4Owed = 1u << 21, kFb4Esc = 1u << 22, kFb4Eol = 1u << 23, kFb4State = 0x3fffffu;
template <class Dummy = void>
TRRE_HD void fb_mark4_lane(const ScanArgs& a, const Fb4View& T, int64_t lane, int64_t lane_bytes, uint8_t* stage_mem, DirectLane& L, uint32_t& status,
                           const FbCopyArgs& ca) {
    const int64_t lo = lane * lane_bytes;
    int64_t hi = lo + lane_bytes;
    if (hi > a.vend) hi = a.vend;
    const uint32_t rhi = (uint32_t)(hi > lo ? hi - lo : 0);
    const uint32_t done_st = T.start[2];
    int first;                                                         // 0 root, 1 SKIP, 2 DONE
    if (lo >= hi) first = 2;
    else if (lo < a.vbeg) first = 1;
    else first = (lo == a.vbeg || a.in_v0[lo - 1] == (uint8_t)'\n') ? 0 : 1;
    uint32_t st = T.start[first];
    const int64_t vlast = (a.vend - 1) & ~(int64_t)15;
    uint32_t* const stage0 = reinterpret_cast<uint32_t*>(stage_mem);
    uint32_t* evp = copy_event_row(ca, lane);
    uint32_t si = 0, escmask = 0, n_ev = 0, b_rel = 0, e_rel = 0, nul = 0, far = 0;
    int64_t delta = 0;
    if (first == 1) b_rel = (uint32_t)(first_line_start_safe(a, lo, hi) - lo);
    const uint8_t* const comb_b = reinterpret_cast<const uint8_t*>(T.comb4);
    const uint8_t* const dense_b = reinterpret_cast<const uint8_t*>(T.dense4);
    // an escape entry's record, by the slot it sits in (the 8-byte comb's numbering)
    auto esc_index = [&](uint32_t slot) -> uint32_t {
        uint32_t l = 0, h = T.n_esc;
        while (l + 1u < h) {
            const uint32_t mid = (l + h) >> 1;
            if (T.esc_slot[mid] <= slot) l = mid; else h = mid;
        }
        return l;
    };
    // one dword (4 input bytes) whose first byte lies rp bytes into the sub-range; kEnd: a lane may finish in it
    auto dword = [&](auto end_tag, const uint32_t w, const uint32_t rp) {
        constexpr bool kEnd = decltype(end_tag)::value;
        const uint32_t kk[4] = {T.cls4[w & 0xffu], T.cls4[(w >> 8) & 0xffu], T.cls4[(w >> 16) & 0xffu], T.cls4[w >> 24]};
        nul |= (w - 0x01010101u) & ~w & 0x80808080u;                              // a zero byte in these four
        const uint32_t st0 = st, si0 = si, e_rel0 = e_rel, far0 = far;
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k4 = kk[j];
            const uint32_t e1 = *reinterpret_cast<const uint32_t*>(comb_b + ((st & 0x3fffu) << 2) + k4);
            const uint32_t e2 = *reinterpret_cast<const uint32_t*>(dense_b + (((st >> 14) & 0x7fu) << 7) + k4);
            const uint32_t e = (e1 >> 24) == k4 ? e1 : e2;
            const uint32_t ev = (st >> 21) & 1u;                                  // an owed state is left: its text goes here
            stage0[si] = (rp + (uint32_t)j) | st << 16;
            si = si + ev < (uint32_t)kMarkStage - 2u ? si + ev : (uint32_t)kMarkStage - 2u;   // (a full stage: the launch is void)
            acc |= e;
            if (kEnd) {
                if (ev && rp + (uint32_t)j > 0xffffu) far = 1;
                const bool fin = (e & kFb4Eol) && rp + (uint32_t)j + 1u >= rhi;
                if (fin && (st & kFb4State) != done_st) e_rel = rp + (uint32_t)j + 1u;
                st = fin ? done_st : e;
            } else {
                st = e;
            }
        }
        if (TRRE_WAVE_ANY(acc & kFb4Esc)) {
            if (acc & kFb4Esc) {                                                  // an escape entry in these four: again, looking
                st = st0; si = si0; e_rel = e_rel0; far = far0;
                for (int j = 0; j < 4; ++j) {
                    const uint32_t k4 = kk[j];
                    const uint32_t base = st & 0x3fffu, dn = (st >> 14) & 0x7fu;
                    const uint32_t e1 = *reinterpret_cast<const uint32_t*>(comb_b + (base << 2) + k4);
                    const uint32_t e2 = *reinterpret_cast<const uint32_t*>(dense_b + (dn << 7) + k4);
                    const bool mine = (e1 >> 24) == k4;
                    const uint32_t e = mine ? e1 : e2;
                    uint32_t ev = (st >> 21) & 1u;
                    uint32_t word = (rp + (uint32_t)j) | st << 16;
                    if (e & kFb4Esc) {
                        const uint32_t slot = (mine ? base : (uint32_t)T.dense_base[dn]) + (k4 >> 2);
                        word = (rp + (uint32_t)j) | (0x8000u | esc_index(slot)) << 16;
                        escmask |= 1u << si;
                        ev = 1u;
                    }
                    if (ev) stage0[si] = word;
                    si = si + ev < (uint32_t)kMarkStage - 2u ? si + ev : (uint32_t)kMarkStage - 2u;
                    if (kEnd && ev && rp + (uint32_t)j > 0xffffu) far = 1;
                    const bool fin = kEnd && (e & kFb4Eol) && rp + (uint32_t)j + 1u >= rhi;
                    if (fin && (st & kFb4State) != done_st) e_rel = rp + (uint32_t)j + 1u;
                    st = fin ? done_st : e;
                }
            }
        }
    };
    auto block = [&](auto end_tag, const U128& b, const uint32_t rp) {
#pragma clang loop unroll(disable)
        for (int d = 0; d < 4; ++d) {
            dword(end_tag, d == 0 ? b.x : (d == 1 ? b.y : (d == 2 ? b.z : b.w)), rp + 4u * (uint32_t)d);
            TRRE_SCHED_FENCE();
        }
    };
    U128 c0 = direct_load(a, lo), c1 = direct_load(a, lo + 16), c2 = direct_load(a, lo + 32), c3 = direct_load(a, lo + 48);
    for (int64_t v = lo;; v += 64) {
        if (!TRRE_WAVE_ANY((st & kFb4State) != done_st)) break;
        const int64_t vn = v + 64;
        const int64_t x0 = vn < vlast ? vn : vlast, x1 = vn + 16 < vlast ? vn + 16 : vlast,
                      x2 = vn + 32 < vlast ? vn + 32 : vlast, x3 = vn + 48 < vlast ? vn + 48 : vlast;
        U128 n0 = *reinterpret_cast<const U128*>(a.in_v0 + x0), n1 = *reinterpret_cast<const U128*>(a.in_v0 + x1),
             n2 = *reinterpret_cast<const U128*>(a.in_v0 + x2), n3 = *reinterpret_cast<const U128*>(a.in_v0 + x3);
        if (TRRE_WAVE_ANY(vn < a.vbeg || vn + 64 > a.vend - 1)) {
            n0 = direct_load(a, vn); n1 = direct_load(a, vn + 16); n2 = direct_load(a, vn + 32); n3 = direct_load(a, vn + 48);
        }
        const uint32_t rp = (uint32_t)(v - lo);
        if (TRRE_WAVE_ALL(rp + 64u < rhi)) {
#pragma clang loop unroll(disable)
            for (int q = 0; q < 4; ++q) {
                U128 b;
                b.x = q == 0 ? c0.x : (q == 1 ? c1.x : (q == 2 ? c2.x : c3.x));
                b.y = q == 0 ? c0.y : (q == 1 ? c1.y : (q == 2 ? c2.y : c3.y));
                b.z = q == 0 ? c0.z : (q == 1 ? c1.z : (q == 2 ? c2.z : c3.z));
                b.w = q == 0 ? c0.w : (q == 1 ? c1.w : (q == 2 ? c2.w : c3.w));
                block(std::false_type{}, b, rp + 16u * (uint32_t)q);
            }
        } else {
#pragma clang loop unroll(disable)
            for (int q = 0; q < 4; ++q) {
                U128 b;
                b.x = q == 0 ? c0.x : (q == 1 ? c1.x : (q == 2 ? c2.x : c3.x));
                b.y = q == 0 ? c0.y : (q == 1 ? c1.y : (q == 2 ? c2.y : c3.y));
                b.z = q == 0 ? c0.z : (q == 1 ? c1.z : (q == 2 ? c2.z : c3.z));
                b.w = q == 0 ? c0.w : (q == 1 ? c1.w : (q == 2 ? c2.w : c3.w));
                block(std::true_type{}, b, rp + 16u * (uint32_t)q);
            }
        }
        // the piece's events, to the end of the lane's row: an owed state's base field becomes the index of its text
        const uint32_t n_loc = si;
        if (n_loc >= (uint32_t)kMarkStage - 2u) far = 1;
        for (uint32_t k = 0; TRRE_WAVE_ANY(k < n_loc); ++k) {
            if (k < n_loc) {
                uint32_t ev = stage0[k];
                if ((escmask >> k) & 1u) {
                    const uint32_t* r = T.esc + 4u * ((ev >> 16) & 0x7fffu);
                    delta += (int32_t)r[1] - (int32_t)(r[3] & 255u);
                } else {
                    const uint32_t id = ((ev >> 16) & 0x3fffu) - T.pad;
                    const uint32_t m = T.lit_meta[id];
                    delta += (int32_t)(m & 255u) - (int32_t)(m >> 8);
                    ev = (ev & 0xffffu) | id << 16;
                }
                if (n_ev + k < ca.ev_cap) evp[k] = ev;
            }
        }
        evp += n_loc;
        n_ev += n_loc;
        si = 0;
        escmask = 0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    if (n_ev > ca.ev_cap || far) status |= kStEditOverflow;
    if (nul) status |= kStNul;
    uint32_t* hdr = ca.lane_hdr + (size_t)lane * 4;
    hdr[0] = n_ev < ca.ev_cap ? n_ev : ca.ev_cap;
    hdr[1] = b_rel;
    hdr[2] = b_rel < rhi ? e_rel : b_rel;          // (no line starts in the sub-range: the lane has nothing to copy)
    hdr[3] = 0;
    uint64_t cnt = 0;
    if (b_rel < rhi && e_rel > b_rel) {
        const int64_t end = lo + (int64_t)e_rel < a.vend ? lo + (int64_t)e_rel : a.vend;
        const int64_t bytes = end - (lo + (int64_t)b_rel) + delta;
        cnt = bytes > 0 ? (uint64_t)bytes : 0;
    }
    L.count = cnt;
}

// =============================================================================================
// The copy form of a large table (front.hpp, StreamTables::fb_copy_ok): the dictionary's second pass without a table
// walk.  The emit pass over the 8-byte rows of such a table costs 70 instructions per byte: most of it the
// walk itself.  But the automaton's output is the input with EDITS — a replacement text where a key stood — and the count
// walk over the comb in LDS (fb_lane<1>: 29 instructions per byte) sees every edit go by:
//
//   fb_lane<3>     (mark) the count walk, plus one 4-byte EVENT per edit, in input order: [15:0] the position (from the
//                  start of the lane's sub-range) of the byte on which the text came out — a text is known only when its
//                  key is complete: it stands for the kb bytes right before that byte — and [31:16] the text's id
//                  (0x8000 | index: an escape record).  Plus the lane's first line start and the end of its last line.
//   second pass    no automaton: input and events in, copy the bytes, insert the texts, skip what they stand for — the
//                  wave-cooperative splice of splice_block.hpp (round 4).  (Round 3's lane-sequential copy pass, fb_copy_lane / k_fb_copy —
//                  35 instructions per byte, every lane copying its own sub-range through a staging ring — stayed as the route for
//                  escape texts of more than 255 bytes until round 6; such tables run the count / emit pair now.)
//
// A NUL ends a line early (the rest of the record is swallowed, not passed through): the launch is void and the count /
// emit pair runs (kStNul), as for the length-preserving kernels.  So does a lane with more than ev_cap events, or with an
// event more than 64 KiB behind its start (a very long last line): kStEditOverflow.
// =============================================================================================


// =============================================================================================
// Positional-window walk for length-preserving stream tables (window form of
// stream_build.cpp).  Per byte: one class lookup (prefetched), one 16-byte table
// entry, and five register operations — no branch, no output cursor:
//
//   seq  = v_perm_b32(input byte, entry.bytes, entry.selector)   the bytes this step emits
//   win |= seq << entry.shift        they land at (delay - pending) bytes past the release point
//   R    = alignbit(win, R, 8)       the byte for position p - delay is final: release it
//   win >>= 8 ; row = entry.next
//
// With 4..7 bytes pending (keys of 5..8 bytes) the window is 64 bits wide and an entry carries a
// second {bytes, selector} pair for the bytes 4..7 of its sequence (kWide).
//
// Output position == input position, so released bytes are packed into aligned
// dwords statically, 16 bytes at a time.  Each lane walks a long sub-range of the
// input (lane_bytes); it starts at its first line start and runs to the end of its
// last line.
// =============================================================================================
struct LpwView {
    const uint8_t* cls;      // [256]
    const U128* ent;         // [n_states][n_cls] entries of 16 bytes (delay <= 3) or 32 bytes; pair form: [n_states][n_cls][n_cls] of 32 bytes
    uint32_t delay;
    uint32_t n_cls;          // (pair form: the first byte's class times this, plus the second byte's, is the column)
};
constexpr uint32_t kLpwEol = 64u, kLpwNul = 128u, kLpwDiv = 256u, kLpwEol2 = 512u;

// position after the first '\n' at or after lo - 1 (the lane's first line start); >= hi: none
TRRE_HD int64_t first_line_start_global(const ScanArgs& a, int64_t lo, int64_t hi) {
    if (a.in_v0[lo - 1] == (uint8_t)'\n') return lo;
    for (int64_t v = lo & ~(int64_t)15; v < hi; v += 16) {
        const U128 q = *reinterpret_cast<const U128*>(a.in_v0 + v);     // callers keep [lo, hi + 16) inside the input
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
        for (int d = 0; d < 4; ++d) {
            const uint32_t x = wd[d] ^ 0x0a0a0a0au;
            const uint32_t m = (x - 0x01010101u) & ~x & 0x80808080u;   // the lowest flag is exact
            if (m) {
#if defined(__HIP_DEVICE_COMPILE__)
                const int bit = __ffs((int)m) - 1;
#else
                const int bit = __builtin_ctz(m);
#endif
                return v + 4 * d + (bit >> 3) + 1;
            }
        }
    }
    return hi;
}

// The window kernel's hot loop has no code for the two ends of the whole input (filler before
// it, the last byte acting as '\n', nothing readable after it).  A lane whose pieces would touch
// an end hands itself over to the general direct walker (stream_direct_lane<0>, run by a small
// second launch): outputs are position-determined, so whatever the lane already wrote is simply
// written again with the same bytes.
TRRE_HD void lpw_redo(const ScanArgs& a, int64_t lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t k = atomicAdd(a.redo, 1u);
#else
    const uint32_t k = a.redo[0]++;
#endif
    a.redo[1 + k] = (uint32_t)lane;
}

// =============================================================================================
// How the window walk gets its bytes: wave tiles.
//
// A lane never touches global memory for its pieces.  Per-lane 16-byte accesses are one L2 request
// each (the lanes of a wave are lane_bytes apart) and complete a line's 64-byte sectors piecemeal;
// measured that way the store path, not the walk, set the pace.  Instead the wave moves the
// 64-byte pieces of its 64 lanes together as 4 KiB tiles: four adjacent lanes cover one piece, so
// a wave instruction touches 16 rows x 64 contiguous, 64-byte aligned bytes.  The input tile goes
// straight from global memory to LDS (global_load_lds_dwordx4: no staging registers); every lane
// takes its row into registers, at which point the next tile is already requested into the same
// buffer.  Output blocks are collected in a second tile of 128-byte rows, the aligned 128 bytes
// BEHIND the lane's position (the output lags the input by the window delay, so the last block of
// a row is only known after the first block of the piece after it), and every second iteration the
// wave stores that tile as whole 128-byte lines, eight adjacent lanes per line (the compute-free
// probe tools/probes/io_probe: 64-byte rows both ways 1.83 TB/s, 64-byte loads with 128-byte
// stores 2.47 TB/s, 128-byte rows both ways 2.52 TB/s).  All lanes start walking at a multiple of
// 128 bytes (in SKIP state up to their first line start), so their lines complete in step.
//
// Tile layouts: input row r (= lane r of the wave) at r * 64, logical block b in physical 16-byte
// slot b ^ ((r >> 1) & 3); output row r at r * 128, block b in slot b ^ (r & 7).  That makes both
// the row-wise accesses of 8 consecutive lanes and the slot-linear accesses of the tile moves
// conflict-free.  The LDS side of a direct load is linear in the lane id, so the permutation is
// applied to the global address instead.
// =============================================================================================
constexpr int kWtBlocks = 4;
constexpr int kWtPiece = 16 * kWtBlocks;
constexpr int kWtTile = 64 * kWtPiece;             // one piece of every lane of a wave
constexpr int kWtOutRow = 128;                     // output rows: whole cache lines
constexpr int kWtOutTile = 64 * kWtOutRow;

// a lane's row of the output tile
struct WtOutRow {
    uint8_t* row;        // tile + r * 128
    uint32_t swz16;      // (r & 7) << 4
    TRRE_HD uint8_t* slot(int b) const { return row + ((uint32_t)(b << 4) ^ swz16); }
};

// byte classes of one dword
TRRE_HD void lpw_classes(const LpwView& T, uint32_t w, uint32_t (&kk)[4]) {
    kk[0] = T.cls[w & 0xffu]; kk[1] = T.cls[(w >> 8) & 0xffu]; kk[2] = T.cls[(w >> 16) & 0xffu]; kk[3] = T.cls[w >> 24];
}

// One 16-byte block of the walk.  The class lookups are taken off the row chain: kk holds the
// classes of the block's first dword on entry and those of next_w on return.  rv: the block's offset
// in the lane's sub-range, rhi: the sub-range's length.
template <bool kCheckEnd, bool kWide>
TRRE_HD void wt_block(const LpwView& T, const U128& cur, uint32_t next_w, uint32_t (&kk)[4], int32_t rv, int32_t rhi, uint32_t& row,
                      uint32_t& win, uint32_t& win_hi, uint32_t& seen, uint32_t (&Rm)[4], uint32_t& done, int32_t& rend) {
    const uint32_t wd[5] = {cur.x, cur.y, cur.z, cur.w, next_w};
    uint32_t R = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t w = wd[d];
        const uint32_t kc[4] = {kk[0], kk[1], kk[2], kk[3]};
        lpw_classes(T, wd[d + 1], kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint8_t* ep = reinterpret_cast<const uint8_t*>(T.ent) + row + (kc[j] << (kWide ? 5 : 4));
            const U128 e = *reinterpret_cast<const U128*>(ep);
            const uint32_t seq = perm_b32(w, e.z, e.w);
            if (!kWide) {
                win |= seq << (e.y & 63u);
                R = alignbit_b32(win, R, 8);
                win >>= 8;
            } else {
                const uint64_t e2 = *reinterpret_cast<const uint64_t*>(ep + 16);       // bytes 4..7: {bytes, selector}
                const uint32_t seq2 = perm_b32(w, (uint32_t)e2, (uint32_t)(e2 >> 32));
                uint64_t w64 = (uint64_t)win_hi << 32 | win;
                w64 |= ((uint64_t)seq2 << 32 | seq) << (e.y & 63u);
                R = alignbit_b32((uint32_t)w64, R, 8);
                w64 >>= 8;
                win = (uint32_t)w64;
                win_hi = (uint32_t)(w64 >> 32);
            }
            row = e.x;
            seen |= e.y;
            if (kCheckEnd) {
                // the first record end at or beyond the end of the sub-range is where the lane's own lines end
                // (branch-free; the walk itself simply goes on to the end of the piece)
                const int32_t p1 = rv + 4 * d + j + 1;
                const uint32_t hit = ((e.y >> 6) & 1u) & (uint32_t)(p1 >= rhi) & (done ^ 1u);
                rend = hit ? p1 : rend;
                done |= hit;
            }
            w >>= 8;
        }
        Rm[d] = R;
        // the running flags are not needed before the end of the piece; left alone the compiler turns
        // their chains into trees evaluated there, with every entry's meta word live until then
        TRRE_PIN(seen);
        if (kCheckEnd) { TRRE_PIN(done); TRRE_PIN(rend); }
        TRRE_SCHED_FENCE();
    }
}

// The same with the pair form of the entries (front.hpp, lpw2): one table read per TWO input bytes.  The walk is a chain
// of dependent LDS reads — with three waves per SIMD (the tiles fill the LDS) each step took ~260 clocks, twice what its
// ten instructions and its 16 bytes of LDS traffic need — so the way to go faster is fewer, fatter steps: an entry holds
// what two steps put into the window, already in place, and the window releases two bytes at a time.
template <bool kCheckEnd>
TRRE_HD void wt_block_pair(const LpwView& T, const U128& cur, uint32_t next_w, uint32_t (&kk)[4], int32_t rv, int32_t rhi, uint32_t& row,
                           uint32_t& win, uint32_t& win_hi, uint32_t& seen, uint32_t (&Rm)[4], uint32_t& done, int32_t& rend) {
    const uint32_t wd[5] = {cur.x, cur.y, cur.z, cur.w, next_w};
    uint32_t R = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t w = wd[d];
        const uint32_t kc[4] = {kk[0], kk[1], kk[2], kk[3]};
        lpw_classes(T, wd[d + 1], kk);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t col = __umul24(kc[2 * j], T.n_cls) + kc[2 * j + 1];     // (a full 32-bit multiply is quarter rate)
#else
            const uint32_t col = kc[2 * j] * T.n_cls + kc[2 * j + 1];
#endif
            const uint8_t* ep = reinterpret_cast<const uint8_t*>(T.ent) + row + (col << 5);
            const U128 e = *reinterpret_cast<const U128*>(ep);
            const uint64_t e2 = *reinterpret_cast<const uint64_t*>(ep + 16);
            const uint32_t w2 = w >> (16 * j);                                   // the pair's bytes in bytes 0 and 1
            win |= perm_b32(w2, e.z, e.w);
            win_hi |= perm_b32(w2, (uint32_t)e2, (uint32_t)(e2 >> 32));
            R = alignbit_b32(win, R, 16);
            win = alignbit_b32(win_hi, win, 16);
            win_hi >>= 16;
            row = e.x;
            seen |= e.y;
            if (kCheckEnd) {
                const int32_t p1 = rv + 4 * d + 2 * j + 1;
                const uint32_t hit1 = ((e.y >> 6) & 1u) & (uint32_t)(p1 >= rhi) & (done ^ 1u);
                rend = hit1 ? p1 : rend;
                done |= hit1;
                const uint32_t hit2 = ((e.y >> 9) & 1u) & (uint32_t)(p1 + 1 >= rhi) & (done ^ 1u);
                rend = hit2 ? p1 + 1 : rend;
                done |= hit2;
            }
        }
        Rm[d] = R;
        TRRE_PIN(seen);
        if (kCheckEnd) { TRRE_PIN(done); TRRE_PIN(rend); }
        TRRE_SCHED_FENCE();
    }
}

template <bool kWide, bool kPair = false>
struct WtLane {
    int64_t lo;
    int32_t rhi, rfs, rlimit, rv, rend;
    uint32_t D, row, win, win_hi, seen, Rprev, done;
    uint32_t kk[4];
    U128 carry;          // the output block being assembled (its last dword needs the next block)
    bool active;

    TRRE_HD void init(const ScanArgs& a, const LpwView& T, uint32_t n_cls, int64_t lane, int64_t lane_bytes) {
        active = false;
        seen = 0; rv = 0; rfs = 0;
        lo = lane * lane_bytes;                          // lane_bytes is a multiple of the piece size
        int64_t hi = lo + lane_bytes;
        if (hi > a.vend) hi = a.vend;
        if (lo >= hi) return;
        // keep slack to both ends of the input; lanes that need the edge fix-ups are redone
        if (lo < a.vbeg + kWtPiece || hi + 3 * kWtPiece > a.vend) { lpw_redo(a, lane); return; }
        const int64_t fs = first_line_start_global(a, lo, hi);
        if (fs >= hi) return;                                     // no line starts in this sub-range
        D = T.delay & 3u;
        rhi = (int32_t)(hi - lo);
        rfs = (int32_t)(fs - lo);
        const int64_t room = a.vend - lo - 3 * kWtPiece;          // pieces are fetched one ahead: never run into
        rlimit = room < 0x40000000 ? (int32_t)room : 0x40000000;  // the end of the input; keep 32-bit offsets exact
        rv = rfs & ~(kWtOutRow - 1);                              // (output rows of all lanes complete in step)
        row = rv == rfs ? 0u : (kPair ? kSkipState * n_cls * n_cls * 32u : kSkipState * n_cls * (kWide ? 32u : 16u));   // the byte before fs is '\n': SKIP reaches root exactly at fs
        win = 0; win_hi = 0; Rprev = 0; done = 0;
        rend = 0x7fffffff;
        carry = U128{};
        active = true;
    }
    // top of an iteration: a very long last line is handed over before the lane would read beyond its slack
    TRRE_HD void check(const ScanArgs& a, int64_t lane) {
        if (active && rv > rlimit) { lpw_redo(a, lane); active = false; }
    }
    // 0: one of the lane's first three pieces (may hold offsets below its first line start)
    // 1: a piece well inside the sub-range   2: a piece in which the lane's last line may end
    TRRE_HD int mode(int32_t k64) const { return k64 < 3 * kWtPiece ? 0 : (rv + kWtPiece < rhi ? 1 : 2); }

    // The output block `carry` = offsets [r0, r0 + 16) is complete: it goes to its slot of the lane's
    // output row.  kHead: the block that contains the lane's first line start is written to memory
    // from here, bytewise from that offset (the tile store skips it; blocks below are nobody's).
    template <bool kHead>
    TRRE_HD void emit(const WtOutRow& orow, int32_t r0, uint8_t* out_v0) {
        uint8_t* blk = orow.slot((r0 >> 4) & 7);
        *reinterpret_cast<U128*>(blk) = carry;
        if (kHead && r0 < rfs && r0 + 16 > rfs) {
            // (a real loop over the bytes just put in the tile: this runs once per lane, keep it small)
#pragma clang loop unroll(disable)
            for (int i = rfs - r0; i < 16; ++i) out_v0[lo + r0 + i] = blk[i];
        }
    }
    // One input block.  The released bytes lag the input by `delay` bytes: the output block below this one
    // is complete after this block's first dword (delay <= 3) or its first two (delay 4..7).
    template <bool kCheckEnd, bool kHead>
    TRRE_HD void step(const LpwView& T, const U128& blk, uint32_t next_w, int q, const WtOutRow& orow, uint8_t* out_v0) {
        uint32_t Rm[4];
        if (kPair) wt_block_pair<kCheckEnd>(T, blk, next_w, kk, rv + 16 * q, rhi, row, win, win_hi, seen, Rm, done, rend);
        else wt_block<kCheckEnd, kWide>(T, blk, next_w, kk, rv + 16 * q, rhi, row, win, win_hi, seen, Rm, done, rend);
        if (!kWide) {
            carry.w = alignbyte_b32(Rm[0], Rprev, D);
        } else {
            carry.z = alignbyte_b32(Rm[0], Rprev, D);
            carry.w = alignbyte_b32(Rm[1], Rm[0], D);
        }
        emit<kHead>(orow, rv - 16 + 16 * q, out_v0);
        if (!kWide) {
            carry.x = alignbyte_b32(Rm[1], Rm[0], D);
            carry.y = alignbyte_b32(Rm[2], Rm[1], D);
            carry.z = alignbyte_b32(Rm[3], Rm[2], D);
        } else {
            carry.x = alignbyte_b32(Rm[2], Rm[1], D);
            carry.y = alignbyte_b32(Rm[3], Rm[2], D);
        }
        Rprev = Rm[3];
    }
    // First block of the piece at rv: the output block below rv (in every second piece it completes an output row).
    TRRE_HD void front(const LpwView& T, int md, const U128& b0, uint32_t next_w, const WtOutRow& orow, uint8_t* out_v0) {
        lpw_classes(T, b0.x, kk);
        if (md == 0) step<true, true>(T, b0, next_w, 0, orow, out_v0);
        else if (md == 1) step<false, false>(T, b0, next_w, 0, orow, out_v0);
        else step<true, false>(T, b0, next_w, 0, orow, out_v0);
    }
    // The other three: the output blocks [rv, rv + 48).
    template <bool kCheckEnd, bool kHead>
    TRRE_HD void back_t(const LpwView& T, const U128& b1, const U128& b2, const U128& b3, const WtOutRow& orow, uint8_t* out_v0) {
        step<kCheckEnd, kHead>(T, b1, b2.x, 1, orow, out_v0);
        step<kCheckEnd, kHead>(T, b2, b3.x, 2, orow, out_v0);
        step<kCheckEnd, kHead>(T, b3, 0u, 3, orow, out_v0);
    }
    TRRE_HD void back(const LpwView& T, int md, const U128& b1, const U128& b2, const U128& b3, const WtOutRow& orow, uint8_t* out_v0) {
        if (md == 0) back_t<true, true>(T, b1, b2, b3, orow, out_v0);
        else if (md == 1) back_t<false, false>(T, b1, b2, b3, orow, out_v0);
        else back_t<true, false>(T, b1, b2, b3, orow, out_v0);
        // The lane's own lines end at the first record end at or beyond the end of its sub-range; the
        // automaton has simply kept going to the end of the piece, where what it emits is the head of
        // the next lane's first line, byte for byte what that lane writes itself: whole blocks are stored.
        if (done && rend <= rv + kWtPiece - 16) active = false;   // every offset below `rend` is in the output tile
        else rv += kWtPiece;
    }
};

// What lane `lid` of a wave needs to move tiles.  Offsets are relative to the start of the wave's first
// sub-range (a wave spans 64 * lane_bytes < 2^31 bytes).
//   input  (4 instructions of 16 rows x 64 bytes): instruction i, logical block (lid & 3) ^ ((lid >> 3) & 3)
//          of row 16 i + (lid >> 2)
//   output (8 instructions of 8 rows x 128 bytes): instruction i, logical block (lid & 7) ^ ((lid >> 3) & 7)
//          of row 8 i + (lid >> 3)
struct WtMover {
    int32_t src[4];       // input: offset of that block in the row's piece 0
    int32_t dst[8];       // output: offset of that block in the row's output row 0 (= [rv0, rv0 + 128))
    TRRE_HD static int row_of(int lid, int i) { return 16 * i + (lid >> 2); }
    TRRE_HD static int block_of(int lid) { return (lid & 3) ^ ((lid >> 3) & 3); }
    TRRE_HD static int out_row_of(int lid, int i) { return 8 * i + (lid >> 3); }
    TRRE_HD static int out_block_of(int lid) { return (lid & 7) ^ ((lid >> 3) & 7); }
    // row_rv0: the 128-aligned offset at which the row's lane starts walking
    TRRE_HD void set(int lid, int i, int64_t lane_bytes, int32_t row_rv0) {
        src[i] = (int32_t)(row_of(lid, i) * lane_bytes) + row_rv0 + 16 * block_of(lid);
    }
    TRRE_HD void set_out(int lid, int i, int64_t lane_bytes, int32_t row_rv0) {
        dst[i] = (int32_t)(out_row_of(lid, i) * lane_bytes) + row_rv0 + 16 * out_block_of(lid);
    }
    // where to fetch the block of piece k from (rows that are not walking get any readable address);
    // room = offset of the last readable 16 bytes, relative like src
    TRRE_HD int32_t load_off(int i, int32_t k64, int64_t room) const {
        const int64_t v = (int64_t)src[i] + k64;
        return (int32_t)(v < room ? v : room);
    }
    // An iteration with k64 a multiple of 128 stores the output rows [rv - 128, rv): blocks 0..2 were
    // written two iterations ago, 3..6 in the previous one, 7 in this one (rows2 / rows1 / rows: the
    // lanes that walked then).  Blocks below the row's first line start (row_rfs, relative to its
    // sub-range) are nobody's; the one containing it is written by the row's lane itself.
    TRRE_HD static bool stores(int i, int32_t k64, uint64_t rows, uint64_t rows1, uint64_t rows2, int lid, int32_t row_rfs) {
        const int b = out_block_of(lid);
        const uint64_t m = b == 7 ? rows : (b >= 3 ? rows1 : rows2);
        return ((m >> out_row_of(lid, i)) & 1u) && k64 - kWtOutRow + 16 * b >= (row_rfs & (kWtOutRow - 1));
    }
    TRRE_HD int32_t store_off(int i, int32_t k64) const { return dst[i] + k64 - kWtOutRow; }
};

// =============================================================================================
// Backward pass of the guided families (tables: guided_build.cpp).  A DFA reads the input right to
// left, '\n' resets it, and its state after byte p — what the reference's backtracking search would
// find on the rest of the line — is stored as the symbol of position p.  Lane `lane` produces the
// symbols of exactly the positions [lane * lane_bytes, (lane + 1) * lane_bytes) (lane_bytes: a multiple
// of 64): it first runs, without storing, from the end of the line that crosses the end of its
// sub-range, then walks its own bytes, 64 at a time, highest first.  Positions outside the input get
// symbols too (they read as the walkers see them: direct_load), up to the next multiple of 64.
// =============================================================================================
struct RevView {
    const uint8_t* tab;      // [n_rev][256] next state by raw byte (the class-compressed table expanded: one lookup per byte)
};
TRRE_HD uint32_t rev_step4(const RevView& T, uint32_t& r, uint32_t w) {
    // bytes 3, 2, 1, 0 of w in that order; returns their four symbols packed like w
    uint32_t y;
    r = T.tab[(r << 8) | (w >> 24)]; y = r << 24;
    r = T.tab[(r << 8) | ((w >> 16) & 0xffu)]; y |= r << 16;
    r = T.tab[(r << 8) | ((w >> 8) & 0xffu)]; y |= r << 8;
    r = T.tab[(r << 8) | (w & 0xffu)]; y |= r;
    return y;
}
// four symbols packed into 16 bits (backward DFAs of at most 16 states)
TRRE_HD uint32_t rev_step4n(const RevView& T, uint32_t& r, uint32_t w) {
    uint32_t y;
    r = T.tab[(r << 8) | (w >> 24)]; y = r << 12;
    r = T.tab[(r << 8) | ((w >> 16) & 0xffu)]; y |= r << 8;
    r = T.tab[(r << 8) | ((w >> 8) & 0xffu)]; y |= r << 4;
    r = T.tab[(r << 8) | (w & 0xffu)]; y |= r;
    return y;
}
// symbols of a 16-byte block, packed: dword x = bytes 0..7, dword y = bytes 8..15
TRRE_HD void rev_block_n(const RevView& T, uint32_t& r, const U128& b, uint32_t& x, uint32_t& y) {
    const uint32_t h3 = rev_step4n(T, r, b.w), h2 = rev_step4n(T, r, b.z), h1 = rev_step4n(T, r, b.y), h0 = rev_step4n(T, r, b.x);
    x = h0 | h1 << 16;
    y = h2 | h3 << 16;
}
// one lane's sub-range [lo, hi) by a thread on its own (no wave-level stores, every load patched at the ends of the input):
// the long-line walker below continues with this through the sub-ranges of the lanes it stands in for
template <bool kNib>
TRRE_HD void rev_pieces_solo(const ScanArgs& a, const RevView& T, int64_t lo, int64_t hi, uint32_t& r) {
    if (kNib) {
        for (int64_t v = hi - 128; v >= lo; v -= 128) {
            U128 b[8], y[4];
            for (int k = 0; k < 8; ++k) b[k] = direct_load(a, v + 16 * k);
            rev_block_n(T, r, b[7], y[3].z, y[3].w); rev_block_n(T, r, b[6], y[3].x, y[3].y);
            rev_block_n(T, r, b[5], y[2].z, y[2].w); rev_block_n(T, r, b[4], y[2].x, y[2].y);
            rev_block_n(T, r, b[3], y[1].z, y[1].w); rev_block_n(T, r, b[2], y[1].x, y[1].y);
            rev_block_n(T, r, b[1], y[0].z, y[0].w); rev_block_n(T, r, b[0], y[0].x, y[0].y);
            U128* dst = reinterpret_cast<U128*>(a.sym_v0 + (v >> 1));
            dst[0] = y[0]; dst[1] = y[1]; dst[2] = y[2]; dst[3] = y[3];
        }
        return;
    }
    for (int64_t v = hi - 64; v >= lo; v -= 64) {
        const U128 b0 = direct_load(a, v), b1 = direct_load(a, v + 16), b2 = direct_load(a, v + 32), b3 = direct_load(a, v + 48);
        U128 y0, y1, y2, y3;
        y3.w = rev_step4(T, r, b3.w); y3.z = rev_step4(T, r, b3.z); y3.y = rev_step4(T, r, b3.y); y3.x = rev_step4(T, r, b3.x);
        y2.w = rev_step4(T, r, b2.w); y2.z = rev_step4(T, r, b2.z); y2.y = rev_step4(T, r, b2.y); y2.x = rev_step4(T, r, b2.x);
        y1.w = rev_step4(T, r, b1.w); y1.z = rev_step4(T, r, b1.z); y1.y = rev_step4(T, r, b1.y); y1.x = rev_step4(T, r, b1.x);
        y0.w = rev_step4(T, r, b0.w); y0.z = rev_step4(T, r, b0.z); y0.y = rev_step4(T, r, b0.y); y0.x = rev_step4(T, r, b0.x);
        U128* dst = reinterpret_cast<U128*>(a.sym_v0 + v);
        dst[0] = y0; dst[1] = y1; dst[2] = y2; dst[3] = y3;
    }
}
// the input byte at v as the walkers see it
TRRE_HD uint32_t rev_byte_at(const ScanArgs& a, int64_t v) {
    const U128 q = direct_load(a, v & ~(int64_t)15);
    const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
    return (wd[(v & 15) >> 2] >> (8 * (int)(v & 3))) & 0xffu;
}
// is there a '\n' in [lo, hi) (16-byte aligned bounds)
TRRE_HD bool rev_has_newline(const ScanArgs& a, int64_t lo, int64_t hi) {
    for (int64_t v = lo; v < hi; v += 16) {
        const U128 q = direct_load(a, v);
        const uint32_t wd[4] = {q.x ^ 0x0a0a0a0au, q.y ^ 0x0a0a0a0au, q.z ^ 0x0a0a0a0au, q.w ^ 0x0a0a0a0au};
        for (int d = 0; d < 4; ++d)
            if ((wd[d] - 0x01010101u) & ~wd[d] & 0x80808080u) return true;
    }
    return false;
}
// A lane needs the DFA's state at the end of its sub-range, i.e. it first runs from the end of the line that crosses that
// end — for a line of megabytes every lane inside it would rescan the rest of the line: quadratic.  So the
// look-ahead is bounded by max_look: of the lanes inside a longer line exactly one lies between max_look and max_look +
// lane_bytes from the line's end; it becomes the line's WALKER — after its own sub-range it carries the state on through
// the sub-ranges of the lanes to its left, one after the other, down to the lane in which the line starts —, the others
// stand back and write nothing.  A line of n bytes costs its walker n sequential steps (one thread, ~0.5 GB/s: the reference
// itself is a single thread), everything else stays parallel.
constexpr int64_t kRevMaxLook = 64 * 1024;
template <int kDbg = 0, bool kNib = false>
TRRE_HD void rev_sweep_lane(const ScanArgs& a, const RevView& T, int64_t lane, int64_t lane_bytes, uint8_t* wave_tile = nullptr,
                            int64_t max_look = kRevMaxLook) {
    const int64_t lo = lane * lane_bytes;
    const int64_t vtop = kNib ? (a.vend + 127) & ~(int64_t)127 : (a.vend + 63) & ~(int64_t)63;
    int64_t hi = lo + lane_bytes;
    if (hi > vtop) hi = vtop;
    const bool empty = lo >= hi;
    uint32_t r = 1;            // kSymEol, the state right of a '\n': nothing alive (every byte from vend - 1 on reads as '\n')
    int role = 0;              // 0 an ordinary lane, 1 the walker of a long line, 2 inside a long line: the walker does its bytes
    if (a.exact) {
        // exact sub-ranges (round 5): no walkers — every lane does its own bytes, from the state at hi: exact when the line that crosses hi
        // ends within spec_look bytes, else the state it would have if the line ended there (k_rev_verify checks it against the symbol
        // the lane to the right leaves at hi)
        uint32_t known = 1u;
        if (!empty && hi < a.vend - 1 && rev_byte_at(a, hi - 1) != (uint32_t)'\n') {
            // (16 bytes at a time; hi is a multiple of 16, direct_load shows every byte from the input's last one on as '\n')
            int64_t e = hi + (int64_t)a.spec_look;
            known = 0u;
            for (int64_t v = hi; v < e; v += 16) {
                const U128 q = direct_load(a, v);
                const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
                int found = -1;
                for (int d = 0; d < 4 && found < 0; ++d) {
                    const uint32_t x = wd[d] ^ 0x0a0a0a0au;
                    const uint32_t m = (x - 0x01010101u) & ~x & 0x80808080u;   // the lowest flag is exact
                    if (m) {
#if defined(__HIP_DEVICE_COMPILE__)
                        found = 4 * d + ((__ffs((int)m) - 1) >> 3);
#else
                        found = 4 * d + (__builtin_ctz(m) >> 3);
#endif
                    }
                }
                if (found >= 0 && v + found < e) { e = v + found; known = 1u; break; }
            }
            for (int64_t v = (e - 1) & ~(int64_t)15; v >= hi && e > hi; v -= 16) {
                const U128 q = direct_load(a, v);
                const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma clang loop unroll(disable)
                for (int k = 15; k >= 0; --k)
                    if (v + k < e) r = T.tab[(r << 8) | ((wd[k >> 2] >> (8 * (k & 3))) & 0xffu)];
            }
        }
        if (!empty) a.rev_guess[lane] = r | known << 31;
    } else
    if (!empty && hi < a.vend - 1 && rev_byte_at(a, hi - 1) != (uint32_t)'\n') {      // (right of a '\n' the state is known)
        // the line that crosses hi: find its end e (first '\n' at or after hi), then run e - 1 .. hi
        int64_t e = hi;        // hi is a multiple of 16
        const int64_t limit = hi + max_look + lane_bytes;
        bool found_end = false;
        for (; e <= limit; e += 16) {
            const U128 q = direct_load(a, e);
            const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
            int found = -1;
            for (int d = 0; d < 4 && found < 0; ++d) {
                const uint32_t x = wd[d] ^ 0x0a0a0a0au;
                const uint32_t m = (x - 0x01010101u) & ~x & 0x80808080u;   // the lowest flag is exact
                if (m) {
#if defined(__HIP_DEVICE_COMPILE__)
                    found = 4 * d + ((__ffs((int)m) - 1) >> 3);
#else
                    found = 4 * d + (__builtin_ctz(m) >> 3);
#endif
                }
            }
            if (found >= 0) { e += found; found_end = true; break; }
        }
        if (!found_end || e - hi > max_look + lane_bytes) role = 2;
        else if (e - hi > max_look) role = 1;
        // blocks from the one holding e - 1 down to hi; bytes at or beyond e are skipped
        if (role != 2)
        for (int64_t v = (e - 1) & ~(int64_t)15; v >= hi && e > hi; v -= 16) {
            const U128 q = direct_load(a, v);
            const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma clang loop unroll(disable)
            for (int k = 15; k >= 0; --k) {
                const int64_t p = v + k;
                if (p < e) r = T.tab[(r << 8) | ((wd[k >> 2] >> (8 * (k & 3))) & 0xffu)];
            }
        }
    }
    // (a wave with a lane that is not ordinary does without the stores through the wave's tile: they need all 64 lanes)
    const bool odd_wave = TRRE_WAVE_ANY(role != 0);
    if (empty || role == 2) return;
    do {
    // The lane's own pieces, highest first.  A wave whose 64 sub-ranges lie wholly inside the input takes the loop
    // whose loads and stores are unconditional — the piece below is requested before this one is walked, and with
    // a fixed number of memory operations per iteration the wait for it can leave the previous piece's stores in
    // flight (s_waitcnt vmcnt(N) counts instructions: a conditional one forces vmcnt(0), i.e. every piece would
    // wait for its own stores).  Waves at an end of the input take the patched loads (direct_load).
    if (kNib) {
        // packed symbols: 128 input bytes -> 64 bytes of symbols per piece (lane_bytes is a multiple of 128)
        auto fetch8 = [&](int64_t v, U128 (&b)[8], bool interior) {
            const int64_t vv = v >= lo ? v : lo;
            if (interior) {
                const U128* src = reinterpret_cast<const U128*>(a.in_v0 + vv);
#pragma unroll
                for (int k = 0; k < 8; ++k) b[k] = src[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) b[k] = direct_load(a, vv + 16 * k);
            }
        };
        const bool interior = !odd_wave && TRRE_WAVE_ALL(lo >= a.vbeg && lo + lane_bytes + 128 <= a.vend - 1 && hi == lo + lane_bytes);
        U128 b[8];
        fetch8(hi - 128, b, interior);
        for (int64_t v = hi - 128; v >= lo; v -= 128) {
            U128 nx[8];
            fetch8(v - 128, nx, interior);
            U128 y[4];
            rev_block_n(T, r, b[7], y[3].z, y[3].w); rev_block_n(T, r, b[6], y[3].x, y[3].y);
            rev_block_n(T, r, b[5], y[2].z, y[2].w); rev_block_n(T, r, b[4], y[2].x, y[2].y);
            rev_block_n(T, r, b[3], y[1].z, y[1].w); rev_block_n(T, r, b[2], y[1].x, y[1].y);
            rev_block_n(T, r, b[1], y[0].z, y[0].w); rev_block_n(T, r, b[0], y[0].x, y[0].y);
#if defined(__HIP_DEVICE_COMPILE__)
            if (interior && wave_tile) {
                // The 64 bytes of a lane are four 16-byte stores of its own: 64 requests per instruction for 64 different
                // lines, each line written in four parts (PMC: 1.84x write amplification).  Through a 4 KiB tile of the
                // wave instead: four adjacent lanes store one lane's 64 bytes as one request (all lanes of an interior
                // wave are at the same offset of their sub-ranges).
                const int lid = (int)__lane_id();
                const WtRow mine{wave_tile + lid * 64, (uint32_t)((lid >> 1) & 3) << 4};
                mine.store(0, y[0]); mine.store(1, y[1]); mine.store(2, y[2]); mine.store(3, y[3]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                uint8_t* wave_dst = a.sym_v0 + (((lane - lid) * lane_bytes + (v - lo)) >> 1) + 16 * (lid & 3);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * q + (lid >> 2);
                    const WtRow row{wave_tile + r * 64, (uint32_t)((r >> 1) & 3) << 4};
                    *reinterpret_cast<U128*>(wave_dst + (int64_t)r * (lane_bytes >> 1)) = row.load(lid & 3);
                }
                __builtin_amdgcn_wave_barrier();
            } else
#endif
            {
                U128* dst = reinterpret_cast<U128*>(a.sym_v0 + (v >> 1));
                dst[0] = y[0]; dst[1] = y[1]; dst[2] = y[2]; dst[3] = y[3];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = nx[k];
        }
        break;
    }
    auto walk_piece = [&](const U128& b0, const U128& b1, const U128& b2, const U128& b3, int64_t v) {
        U128 y0, y1, y2, y3;
        if (kDbg != 2) {
        y3.w = rev_step4(T, r, b3.w); y3.z = rev_step4(T, r, b3.z); y3.y = rev_step4(T, r, b3.y); y3.x = rev_step4(T, r, b3.x);
        y2.w = rev_step4(T, r, b2.w); y2.z = rev_step4(T, r, b2.z); y2.y = rev_step4(T, r, b2.y); y2.x = rev_step4(T, r, b2.x);
        y1.w = rev_step4(T, r, b1.w); y1.z = rev_step4(T, r, b1.z); y1.y = rev_step4(T, r, b1.y); y1.x = rev_step4(T, r, b1.x);
        y0.w = rev_step4(T, r, b0.w); y0.z = rev_step4(T, r, b0.z); y0.y = rev_step4(T, r, b0.y); y0.x = rev_step4(T, r, b0.x);
        }
        U128* dst = reinterpret_cast<U128*>(a.sym_v0 + v);
        if (kDbg == 1) { if (y0.x == 0x12345678u && y3.w == 0x9abcdef0u) dst[0] = y1; }      // experiment: no stores
        else if (kDbg == 2) { dst[0] = b0; dst[1] = b1; dst[2] = b2; dst[3] = b3; }               // experiment: no walk
        else { dst[0] = y0; dst[1] = y1; dst[2] = y2; dst[3] = y3; }
    };
    if (TRRE_WAVE_ALL(lo >= a.vbeg && lo + lane_bytes + 64 <= a.vend - 1 && hi == lo + lane_bytes)) {
        const U128* src = reinterpret_cast<const U128*>(a.in_v0 + hi - 64);
        U128 b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
        for (int64_t v = hi - 64; v >= lo; v -= 64) {
            const U128* nx = reinterpret_cast<const U128*>(a.in_v0 + (v - 64 >= lo ? v - 64 : lo));     // (the last fetch is not used)
            const U128 n0 = nx[0], n1 = nx[1], n2 = nx[2], n3 = nx[3];
            walk_piece(b0, b1, b2, b3, v);
            b0 = n0; b1 = n1; b2 = n2; b3 = n3;
        }
        break;
    }
    auto fetch = [&](int64_t v, U128& b0, U128& b1, U128& b2, U128& b3) {
        const int64_t vv = v >= lo ? v : lo;              // (the fetch below the lane's first piece is not used)
        b0 = direct_load(a, vv); b1 = direct_load(a, vv + 16); b2 = direct_load(a, vv + 32); b3 = direct_load(a, vv + 48);
    };
    U128 b0, b1, b2, b3;
    fetch(hi - 64, b0, b1, b2, b3);
    for (int64_t v = hi - 64; v >= lo; v -= 64) {
        U128 n0, n1, n2, n3;
        fetch(v - 64, n0, n1, n2, n3);
        walk_piece(b0, b1, b2, b3, v);
        b0 = n0; b1 = n1; b2 = n2; b3 = n3;
    }
    } while (false);
    if (role == 1) {
        // the walker: on through the lanes to the left that stood back, down to the one in which the long line starts
        for (int64_t k = lane; k > 0;) {
            const int64_t lok = k * lane_bytes;
            // the long line starts inside lane k if the lane holds a '\n' at all (every '\n' in it lies before that start),
            // or right at its first byte
            if (lok <= a.vbeg || rev_byte_at(a, lok - 1) == (uint32_t)'\n' || rev_has_newline(a, lok, lok + lane_bytes)) break;
            --k;
            rev_pieces_solo<kNib>(a, T, k * lane_bytes, (k + 1) * lane_bytes, r);
        }
    }
}

// exact sub-ranges, the backward pass: the symbol at position v as the sweep left it
template <bool kNib>
TRRE_HD uint32_t rev_symbol_at(const ScanArgs& a, int64_t v) {
    return kNib ? ((uint32_t)a.sym_v0[v >> 1] >> (4u * ((uint32_t)v & 1u))) & 15u : a.sym_v0[v];
}
// ... is lane's guess what the lane to its right found?
template <bool kNib>
TRRE_HD bool rev_guess_wrong(const ScanArgs& a, int64_t lane, int64_t lane_bytes) {
    const uint32_t g = a.rev_guess[lane];
    if (g >> 31) return false;
    return (g & 0x7fffffffu) != rev_symbol_at<kNib>(a, (lane + 1) * lane_bytes);
}
// ... a flagged lane sweeps its sub-range again from the symbol at its end, and on through the lanes to its left for as long as what it
// arrives with is not what they had assumed (T: the table in global memory — a handful of lanes)
template <bool kNib>
TRRE_HD void rev_repair_lane(const ScanArgs& a, const RevView& T, int64_t lane, int64_t lane_bytes) {
    uint32_t r = rev_symbol_at<kNib>(a, (lane + 1) * lane_bytes);
    for (int64_t k = lane;; --k) {
        a.rev_guess[k] = r;                                            // (open to the next verification)
        rev_pieces_solo<kNib>(a, T, k * lane_bytes, (k + 1) * lane_bytes, r);
        if (k == 0 || a.rev_flags[k - 1]) break;
        const uint32_t g = a.rev_guess[k - 1];
        if ((g >> 31) || (g & 0x7fffffffu) == r) break;
    }
}

// =============================================================================================
// Wide guided tables: a backward DFA of more than 256 states (e.g. 'a(a|b|c|d|e|f|g|h){9}c:x': which of the next ten bytes
// is a 'c' — 4 604 states).  Symbols are 16 bits, neither table fits LDS: both passes go through L1 / L2, one lookup per
// byte and lane, no staging — the simplest correct walkers (an order of magnitude slower than the byte-symbol kernels, far
// faster than refusing the pattern).  Same lane ownership, same count / scan / emit plumbing as k_stream_direct.
// =============================================================================================
struct RevWideView { const uint16_t* tab; };     // [n_rev][256] next state by raw byte
// symbols of the positions [lo, hi) (multiples of 16), highest first; r: the state right of hi - 1 on entry, of lo on return
TRRE_HD void rev_wide_span(const ScanArgs& a, const RevWideView& T, int64_t lo, int64_t hi, uint32_t& r) {
    uint16_t* sym = reinterpret_cast<uint16_t*>(a.sym_v0);
    for (int64_t v = hi - 16; v >= lo; v -= 16) {
        const U128 q = direct_load(a, v);
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
        uint32_t y[8];
#pragma clang loop unroll(disable)
        for (int k = 15; k >= 0; --k) {
            r = T.tab[(r << 8) | ((wd[k >> 2] >> (8 * (k & 3))) & 0xffu)];
            if (k & 1) y[k >> 1] = r << 16; else y[k >> 1] |= r;
        }
        U128* dst = reinterpret_cast<U128*>(sym + v);
        dst[0] = U128{y[0], y[1], y[2], y[3]};
        dst[1] = U128{y[4], y[5], y[6], y[7]};
    }
}
// first '\n' at or after `from` (a multiple of 16), looking no further than `limit`; false: none up to there
TRRE_HD bool rev_line_end(const ScanArgs& a, int64_t from, int64_t limit, int64_t& e) {
    for (e = from; e <= limit; e += 16) {
        const U128 q = direct_load(a, e);
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
        for (int d = 0; d < 4; ++d) {
            const uint32_t x = wd[d] ^ 0x0a0a0a0au;
            const uint32_t m = (x - 0x01010101u) & ~x & 0x80808080u;   // the lowest flag is exact
            if (m) {
#if defined(__HIP_DEVICE_COMPILE__)
                e += 4 * d + ((__ffs((int)m) - 1) >> 3);
#else
                e += 4 * d + (__builtin_ctz(m) >> 3);
#endif
                return true;
            }
        }
    }
    return false;
}
// the backward pass of one lane, with the long-line walker of rev_sweep_lane
TRRE_HD void rev_wide_lane(const ScanArgs& a, const RevWideView& T, int64_t lane, int64_t lane_bytes, int64_t max_look = kRevMaxLook) {
    const int64_t lo = lane * lane_bytes;
    const int64_t vtop = (a.vend + 63) & ~(int64_t)63;
    int64_t hi = lo + lane_bytes;
    if (hi > vtop) hi = vtop;
    if (lo >= hi) return;
    uint32_t r = 1;            // kSymEol
    int role = 0;
    if (hi < a.vend - 1 && rev_byte_at(a, hi - 1) != (uint32_t)'\n') {
        int64_t e;
        const bool found = rev_line_end(a, hi, hi + max_look + lane_bytes, e);
        if (!found || e - hi > max_look + lane_bytes) role = 2;
        else if (e - hi > max_look) role = 1;
        if (role != 2)
            for (int64_t v = (e - 1) & ~(int64_t)15; v >= hi && e > hi; v -= 16) {
                const U128 q = direct_load(a, v);
                const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma clang loop unroll(disable)
                for (int k = 15; k >= 0; --k)
                    if (v + k < e) r = T.tab[(r << 8) | ((wd[k >> 2] >> (8 * (k & 3))) & 0xffu)];
            }
    }
    if (role == 2) return;
    rev_wide_span(a, T, lo, hi, r);
    if (role == 1)
        for (int64_t k = lane; k > 0;) {
            const int64_t lok = k * lane_bytes;
            if (lok <= a.vbeg || rev_byte_at(a, lok - 1) == (uint32_t)'\n' || rev_has_newline(a, lok, lok + lane_bytes)) break;
            --k;
            rev_wide_span(a, T, k * lane_bytes, (k + 1) * lane_bytes, r);
        }
}
// the forward pass of one lane: kMode 1 count, 2 emit (straight to memory at its offset)
template <int kMode>
TRRE_HD void wide_fwd_lane(const ScanArgs& a, const StreamView& T, uint32_t n_cls, int64_t lane, int64_t lane_bytes, uint64_t out_base, DirectLane& L,
                           uint32_t& status) {
    const uint32_t done_row = kDoneState * n_cls;
    const int64_t lo = lane * lane_bytes;
    int64_t hi = lo + lane_bytes;
    if (hi > a.vend) hi = a.vend;
    uint32_t row;
    if (lo >= hi) row = done_row;
    else if (lo < a.vbeg) row = kSkipState * n_cls;
    else row = (lo == a.vbeg || a.in_v0[lo - 1] == (uint8_t)'\n') ? 0u : kSkipState * n_cls;
    const uint16_t* sym = reinterpret_cast<const uint16_t*>(a.sym_v0);
    uint64_t cnt = 0;
    int64_t o = (int64_t)out_base;
    uint32_t seen = 0;
    for (int64_t v = lo; row != done_row; v += 16) {
        const U128 q = direct_load(a, v);
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma clang loop unroll(disable)
        for (int k = 0; k < 16 && row != done_row; ++k) {
            const uint8_t c = (uint8_t)(wd[k >> 2] >> (8 * (k & 3)));
            const uint64_t e = T.ent[row + sym[v + k]];
            const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
            if (kMode == 1) cnt += str_count(T, elo, ehi);
            else o = str_emit(T, a.out, o, elo, ehi, c);
            seen |= elo;
            row = str_next(elo);
            if ((elo & kStrEol) && v + k + 1 >= hi) row = done_row;
        }
    }
    if (kMode == 1 && (seen & kStrDiv)) status |= kStDiverge;
    L.count = cnt;
}

// =============================================================================================
// Memoryless tables: out[v] = map[in[v]] for one 16-byte vector at v.
// =============================================================================================
TRRE_HD uint32_t map4(const uint8_t* m, uint32_t w) {
    return (uint32_t)m[w & 0xffu] | (uint32_t)m[(w >> 8) & 0xffu] << 8 | (uint32_t)m[(w >> 16) & 0xffu] << 16 |
           (uint32_t)m[w >> 24] << 24;
}
TRRE_HD uint32_t has_zero_byte(uint32_t w) { return (w - 0x01010101u) & ~w & 0x80808080u; }

// the NUL bytes of a vector, for the repair of the output (rare: a list in memory, appended to with an atomic)
TRRE_HD void nul_record(const ScanArgs& a, const U128& w, int64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!a.nul_list) return;
    const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
    for (int b = 0; b < 16; ++b) {
        const int64_t vv = v + b;
        if (vv < a.vbeg || vv >= a.vend - 1 || ((wd[b >> 2] >> (8 * (b & 3))) & 0xffu) != 0u) continue;
        const uint32_t k = atomicAdd(a.nul_list, 1u);
        if (k < kNulCap) reinterpret_cast<uint64_t*>(a.nul_list + 2)[k] = (uint64_t)(vv - a.vbeg);
    }
#else
    (void)a; (void)w; (void)v;
#endif
}
TRRE_HD void bytemap_vec(const ScanArgs& a, const uint8_t* map, const U128& w, int64_t v, bool aligned, uint32_t& zero) {
    U128 r;
    r.x = map4(map, w.x); r.y = map4(map, w.y); r.z = map4(map, w.z); r.w = map4(map, w.w);
    if (v >= a.vbeg && v + 16 <= a.vend - 1) {            // interior vector
        const uint32_t z = has_zero_byte(w.x) | has_zero_byte(w.y) | has_zero_byte(w.z) | has_zero_byte(w.w);
        if (z) nul_record(a, w, v);
        zero |= z;
        if (aligned) { *reinterpret_cast<U128*>(a.out_v0 + v) = r; return; }
    } else {
        nul_record(a, w, v);
    }
    const uint8_t* src = reinterpret_cast<const uint8_t*>(&w);
    const uint8_t* dst = reinterpret_cast<const uint8_t*>(&r);
    for (int b = 0; b < 16; ++b) {
        const int64_t vv = v + b;
        if (vv < a.vbeg || vv >= a.vend) continue;
        if (vv == a.vend - 1) { a.out_v0[vv] = (uint8_t)'\n'; continue; }   // the last byte is a terminator
        if (src[b] == 0) zero = 1;
        a.out_v0[vv] = dst[b];
    }
}

// =============================================================================================
// Production geometry per engine: tiles sized so that two workgroups share a
// CU's 160 KiB of LDS (one for 64-bit masks).
// =============================================================================================
using GeoDft = Geometry<256, 32768, 2032>;
using GeoNft8 = Geometry<256, 16384, 2032>;
using GeoNft16 = Geometry<256, 16384, 2032>;
using GeoNft32 = Geometry<256, 8192, 2032>;
using GeoNft64 = Geometry<256, 8192, 2032>;

}  // namespace trre
