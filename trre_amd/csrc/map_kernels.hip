// map_kernels.hip — the memoryless one-pass kernel (map_block.hpp) for gfx950: a translation unit of its own (scan_kernels.hip takes two
// minutes to compile; this one seconds).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "launch.hpp"
#include "scan_block.hpp"
#include "one_block.hpp"
#include "map_block.hpp"
#include "device_blob.hpp"

namespace trre {
namespace {

constexpr int kWave = 64;
constexpr int kLdsLimit = 160 * 1024;

__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    for (int d = 32; d; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d, kWave);
    return v;
}

}  // namespace

// ---- memoryless programs of any output length: one pass, no state (map_block.hpp) ---------------------------------------------------
// The two-level look-back of k_stream_one without the exit rows, in groups of 64: a tile publishes its total (descriptor) and adds it to its
// group's {count, sum} word; the last tile of a group leaves the group's running total.  Wave 0 finds the tile's place: lane l asks for the
// tile l + 1 before this one (while that is in this tile's group) AND for the group l + 1 before this tile's — three loads, one round trip,
// a reach of 4 096 tiles: more than a full machine holds.  The first round's loads are issued BEFORE the tile is expanded (mg_poll) and
// looked at after: under load a round trip to the L2 is 3 us — 7 000 clocks —, and by then the tiles before have their totals out.
struct MgPoll { uint64_t d, gs, gi; };
__device__ __forceinline__ MgPoll mg_poll(const uint64_t* desc, const uint64_t* gsum, const uint64_t* ginc, int64_t tile, int64_t gtop, bool tiles_done) {
    const int lid = threadIdx.x & (kWave - 1);
    const int r = (int)(tile & 63);
    MgPoll p;
    p.d = (lid < r && !tiles_done) ? __hip_atomic_load(desc + (tile - 1 - lid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const int64_t gq = gtop - lid;
    p.gs = gq >= 0 ? __hip_atomic_load(gsum + gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    p.gi = gq >= 0 ? __hip_atomic_load(ginc + gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 63);   // (before the first group: 0)
    return p;
}
constexpr uint64_t kMgGroupOne = 1ull << 57, kMgGroupSum = kMgGroupOne - 1;    // a group's word: tiles counted (7 bits) | their sum
// A look-back in steps: mg_lb_step looks at one round's answers — it adds what they say, and either is done, or moves on to the 64 groups
// before (64 complete groups and no running total among them), or leaves everything as it was (a tile or a group it needs is not out yet).
struct MgLb { uint64_t base; int64_t gtop; bool tiles_done, done; };
__device__ __forceinline__ bool mg_lb_step(MgLb& s, int64_t tile, const MgPoll& p) {      // false: wait and ask again
    const int lid = threadIdx.x & (kWave - 1);
    const int r = (int)(tile & 63);
    const bool t_mine = lid < r && !s.tiles_done;
    const uint32_t kind = (uint32_t)(p.d >> 62);
    bool wait = false, finished = false;
    uint64_t v = 0;
    if (!s.tiles_done) {
        const uint64_t zero = __ballot(t_mine && kind == 0u), inc = __ballot(t_mine && kind == 2u);
        const int jinc = inc ? __builtin_ctzll(inc) : 64;
        const uint64_t upto = jinc >= 63 ? ~0ull : (1ull << (jinc + 1)) - 1ull;
        if (zero & upto) wait = true;
        else {
            v = (t_mine && ((upto >> lid) & 1ull)) ? (p.d & kOneValMask) : 0ull;
            if (jinc < 64) finished = true;
        }
    }
    if (!wait && !finished) {
        const uint64_t ginc_at = __ballot((p.gi >> 63) != 0);
        const uint64_t gfull = __ballot((p.gs >> 57) == 64u);
        const int qinc = ginc_at ? __builtin_ctzll(ginc_at) : 64;
        const uint64_t above = qinc >= 64 ? ~0ull : (1ull << qinc) - 1ull;
        if ((gfull & above) != above) wait = true;
        else {
            if (lid < qinc) v += p.gs & kMgGroupSum;
            if (lid == qinc) v += p.gi & ~(1ull << 63);
            if (qinc < 64) finished = true;
        }
    }
    if (wait) return false;
    for (int dd = 32; dd; dd >>= 1) v += __shfl_xor(v, dd, kWave);
    s.base += v;
    s.done = finished;
    s.tiles_done = true;
    s.gtop -= 64;
    return true;
}
// the rounds after the first (mg_lb_step with the answers asked for before the tile was expanded), then the tile's own running total out.
// Returns the tile's place; ok = false: gave up after `spin` polls (the launch is void).
__device__ __forceinline__ uint64_t mg_lb_finish(MgLb s, uint64_t* desc, uint64_t* gsum, uint64_t* ginc, int64_t tile, uint64_t total, uint32_t spin, uint32_t* status, bool& ok) {
    const int lid = threadIdx.x & (kWave - 1);
    ok = true;
    uint32_t polls = 0;
    while (!s.done) {
        const MgPoll p = mg_poll(desc, gsum, ginc, tile, s.gtop, s.tiles_done);
        if (!mg_lb_step(s, tile, p)) {
            // (somebody gave up — a workgroup of the grid is not resident, or the spin ran out here: the launch is void, and nobody waits on)
            if ((__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStOneVoid) || ++polls > spin) {
                if (lid == 0) atomicOr(status, kStOneVoid);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (lid == 0) {
        if (tile > 0) __hip_atomic_store(desc + tile, kOneDescInc | ((s.base + total) & kOneValMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((tile & 63) == 63) __hip_atomic_store(ginc + (tile >> 6), (1ull << 63) | (s.base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return s.base;
}
// an inclusive prefix sum over the wave's lanes through DPP (shifts within the rows of 16, then the rows' last lanes broadcast): six VALU
// operations and no LDS (__shfl_up is a ds_bpermute per step)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// (a barrier with the wave's LDS traffic drained by hand: round 6 met a loop whose head's s_barrier the compiler had left without the
// s_waitcnt for a store at the loop's end — one tile in 6 000 was counted twice by a wave that read the index of the tile before it)
// And: __syncthreads() is a fence too — s_waitcnt vmcnt(0) — which would wait for the bytes asked for a tile ahead; the barriers here order
// LDS traffic only.
#define MG_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
//   smem: len[256] | first[256] | text[256 x 8] | misc[32 x 4] | 2 x window[oa.window + 32] | sinks[kMapGenThreads x 4]
// Workgroup w of G takes the tiles w, w + G, w + 2 G, .. — no ticket to wait for, the next tiles known ahead — and its loop is a pipeline in
// which every trip to memory is asked for a phase or a whole tile before its answer is needed, and a tile's total is out a whole trip before
// anyone looks back at it (a look-back waits for the SLOWEST of the thousand tiles before it: with totals published just in time — the ticket
// forms of this kernel — that was the tail of the memory's latency under load, 60 000 clocks per tile; tickets asked for ahead hold a tile
// unpublished behind its holder's waits, and their answers come back through the same in-order counter as the bytes asked for ahead):
//   holding: `prv` expanded into one window, its look-back's loads out; `cur` counted, its total out; the bytes of the tile after `cur` on their way
//   `prv`'s place from the look-back's answers (wave 0; not enough: asked again, closed behind the count below) - `cur` into the OTHER window -
//   the next tile's bytes are here: the tile after it asked for, the next one counted, its total out - `prv`'s window stored - `cur`'s look-back asked for.
// Two barriers per trip (raw s_barrier with the LDS counter drained: __syncthreads() is a fence and would wait for the bytes asked for ahead).
// All G workgroups must be resident (launch_mapgen asks the runtime how many fit; launches wait for one another per device: runtime.cpp); a
// look-back that finds a tile untouched for `spin` polls gives up, says so in the status word — whoever else is waiting leaves at its next
// poll — and the launch is void (the pair runs the buffer): never a hang.
template <bool kFirst, bool kMulti>
__global__ __launch_bounds__(kMapGenThreads, 4) void k_mapgen(ScanArgs a, MapGenArgs oa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const int tid = threadIdx.x, lid = tid & (kWave - 1), wave = tid / kWave;
    uint8_t* lenp = smem;
    uint8_t* firstp = smem + 256;
    uint64_t* textp = reinterpret_cast<uint64_t*>(smem + 512);
    uint32_t* misc = reinterpret_cast<uint32_t*>(smem + 512 + 2048);     // [0..7] the waves' totals ([31]: a NUL), [10..11] the base, [12] the look-back gave up, [14] it is still open
    uint8_t* win = smem + 512 + 2048 + 128;
    const uint32_t W = oa.window;
    {
        const uint32_t* mg = reinterpret_cast<const uint32_t*>(a.blob + h.off_mg);
        for (int k = tid; k < 256; k += kMapGenThreads) {
            lenp[k] = (uint8_t)mg[4 * k + 2];
            firstp[k] = (uint8_t)mg[4 * k];
            textp[k] = (uint64_t)mg[4 * k + 1] << 32 | mg[4 * k];
        }
    }
    const MapGenView T{lenp, firstp, textp};
    uint32_t st_all = 0;
    if (tid == 0) misc[12] = misc[14] = 0;
    const bool prof = oa.prof != nullptr && tid == 0;
    auto stamp = [&](uint64_t& t, int slot) {
        if (prof) {
            const uint64_t now = clock64();
            atomicAdd(reinterpret_cast<unsigned long long*>(oa.prof + slot), (unsigned long long)(now - t));
            t = now;
        }
    };
    const int64_t vl = (a.vend + 15) & ~(int64_t)15;        // (16-byte blocks that lie beyond the input are not read)
    const int64_t vlane = (int64_t)wave * kMgWaveBytes + (int64_t)lid * kMgLaneBytes;
    const int64_t G = (int64_t)gridDim.x;
    auto is_edge = [&](int64_t tile) { return tile * kMapGenTile < a.vbeg || (tile + 1) * (int64_t)kMapGenTile > a.vend - 1; };
    // the bytes of a tile in flight (interior tiles; an edge tile's are read where they are used)
    uint32_t xlo[kMgRows], xhi[kMgRows], ylo[kMgRows], yhi[kMgRows];
    auto ask = [&](int64_t tile, uint32_t (&l)[kMgRows], uint32_t (&hh)[kMgRows]) {
        if (tile < oa.n_tiles && !is_edge(tile)) {
#pragma unroll
            for (int r = 0; r < kMgRows; ++r) {
                const uint2 d = *reinterpret_cast<const uint2*>(a.in_v0 + tile * kMapGenTile + vlane + r * kMgRowBytes);
                l[r] = d.x; hh[r] = d.y;
            }
        }
    };
    // a tile counted: the lanes' places within the wave (rows two to a word), the wave's offset in the tile, the tile's total
    uint32_t u[kMgRows / 2], inc[kMgRows / 2], tot[kMgRows / 2], woff = 0, total = 0;
    // count `tile` (its bytes in xlo / xhi when it is an interior tile), leave u / inc / tot / woff / total, publish the total; one barrier
    auto count_and_publish = [&](int64_t tile) -> uint32_t {
        const bool edge = is_edge(tile);
        const int64_t vw = tile * kMapGenTile + vlane;
        uint32_t s[kMgRows];
        uint32_t wtotal = 0, orsum = 0;
        if (!edge) {
#pragma unroll
            for (int r = 0; r < kMgRows; ++r) {
                s[r] = mg_count8<false>(T, xlo[r], xhi[r], MgEdge{});
                TRRE_SCHED_FENCE();                        // (row by row: eight rows' lookups in flight at once are 64 registers of addresses)
            }
        } else {
            // (an end of the input — rare, as is a tile whose output outgrows the window: row by row, in a loop that is not unrolled, the
            // bytes read again when they are expanded)
#pragma unroll
            for (int r = 0; r < kMgRows; ++r) s[r] = 0;
#pragma clang loop unroll(disable)
            for (int r = 0; r < kMgRows; ++r) {
                const int64_t v = vw + r * kMgRowBytes;
                uint2 d{0u, 0u};
                if (v < vl) d = *reinterpret_cast<const uint2*>(a.in_v0 + v);
                const uint32_t n = mg_count8<true>(T, d.x, d.y, mg_edge(v, a.vbeg, a.vend));
                orsum |= n;
                wtotal += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(n), kWave - 1);
            }
        }
        // the rows' sums two to a word (a row of a wave prints 64 x 64 bytes at most), prefix sums over the wave, the rows' totals from lane 63
#pragma unroll
        for (int k = 0; k < kMgRows / 2; ++k) {
            orsum |= s[2 * k] | s[2 * k + 1];
            u[k] = s[2 * k] | s[2 * k + 1] << 16;
            inc[k] = wave_incl_scan_dpp(u[k]);
            tot[k] = (uint32_t)__builtin_amdgcn_readlane((int)inc[k], kWave - 1);
            wtotal += (tot[k] & 0xffffu) + (tot[k] >> 16);
        }
        const bool nul = __any(orsum >= kMgNul);
        if (lid == 0) misc[wave] = (wtotal & 0x7fffffffu) | (nul ? 1u << 31 : 0u);
        MG_SYNC();
        uint32_t seen = 0;
        woff = 0; total = 0;
        for (int w = 0; w < kMapGenThreads / kWave; ++w) {
            const uint32_t m = misc[w];
            seen |= m;
            if (w < wave) woff += m & 0x7fffffffu;
            total += m & 0x7fffffffu;
        }
        const uint32_t st = (seen >> 31) ? kStNul : 0u;
        if (st) total = 0;                                 // (a NUL: the launch is void — nothing is expanded, nothing stored)
        if (tid == 0) {
            __hip_atomic_store(oa.desc + tile, (tile == 0 ? kOneDescInc : kOneDescAgg) | ((uint64_t)total & kOneValMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(oa.gsum + (tile >> 6), kMgGroupOne | (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return st;
    };
    // ---- the pipeline.  Per trip: tile `cur` (counted, its total out, its bytes in x) is expanded into its window; tile `nxt`'s bytes have
    // come (y): they move to x, the tile after it is asked for, `nxt` is counted and its total goes out; tile `prv` — expanded a trip ago
    // into the OTHER window — gets its place from a look-back whose loads were issued a trip ago, and is stored.  Two barriers per trip.
    auto win_of = [&](int64_t k) { return win + (size_t)((k / G) & 1) * (W + 32u); };
    auto sink_of = [&](int64_t k) { return (uint32_t)(2u * (W + 32u) - ((k / G) & 1) * (W + 32u)) + 4u * (uint32_t)tid; };   // (the sinks lie behind both windows)
    auto expand_clipped = [&](int64_t tile, uint32_t woff_k, uint32_t wlo, uint32_t wsize) {
        const int64_t vw = tile * kMapGenTile + vlane;
        uint32_t rowbase = woff_k;
#pragma clang loop unroll(disable)
        for (int r = 0; r < kMgRows; ++r) {
            const int64_t v = vw + r * kMgRowBytes;
            uint2 d{0u, 0u};
            if (v < vl) d = *reinterpret_cast<const uint2*>(a.in_v0 + v);
            const MgEdge e = mg_edge(v, a.vbeg, a.vend);
            const uint32_t n = mg_count8<true>(T, d.x, d.y, e);
            const uint32_t in = wave_incl_scan_dpp(n);
            mg_expand8<kFirst, kMulti, true, true>(T, d.x, d.y, e, win_of(tile), sink_of(tile), rowbase + in - n, wlo, wsize);
            rowbase += (uint32_t)__builtin_amdgcn_readlane((int)in, kWave - 1);
        }
    };
    auto store_window = [&](int64_t tile, uint64_t base, uint32_t wlo, uint32_t wsize, bool write) {
        // (hh is the same for every lane — and so is the dword a line starts at in its 16-byte block: mg_store_line's selects are scalar)
        const uint32_t hh = (uint32_t)__builtin_amdgcn_readfirstlane((int)((reinterpret_cast<uintptr_t>(a.out) + base + wlo) & 15u));
        const uint32_t n_lines = (hh + wsize + 15u) >> 4;
        for (uint32_t c = (uint32_t)tid; c < n_lines; c += kMapGenThreads) mg_store_line(win_of(tile), a.out, base, wlo, wsize, hh, c, write);
    };
    int64_t cur = (int64_t)blockIdx.x, prv = -1;
    uint32_t st_cur = 0, st_prv = 0, total_prv = 0, woff_prv = 0;
    bool fits_prv = false;                                 // `prv` lies expanded in its window (else: an edge tile or one of several windows — at its store)
    MgPoll poll_prv{0, 0, 0}, poll_prv2{0, 0, 0};         // (the tile's own group and the 64 groups before it; the 64 before those: a machine full of
                                                           // workgroups holds three tiles each — 48 groups and more between a tile and the newest running total)
    if (cur < oa.n_tiles) {
        ask(cur, xlo, xhi);
        ask(cur + G, ylo, yhi);
        st_cur = count_and_publish(cur);                   // (the barrier inside: the tables are staged)
    } else {
        cur = -1;
    }
    while (cur >= 0 || prv >= 0) {
        uint64_t tclk = prof ? clock64() : 0;
        // ---- `prv`'s place: the look-back's loads went out a trip ago (wave 0).  As a rule their answers are enough; if not — some tile before
        // `prv` had not published when they were asked for — the loads go out again and are looked at behind the next tile's count: a round trip
        // hidden, where waiting here would hold the whole workgroup at the barrier
        MgLb lb{0, (prv >> 6) - 1, false, prv <= 0};
        MgPoll again{0, 0, 0};
        bool lb_pending = false;
        const uint64_t lb_t0 = oa.dbg ? clock64() : 0;
        auto lb_close = [&](bool first) {                  // the rest of the look-back (polls until it is done), `prv`'s running total out, its place into LDS
            bool ok = true;
            uint64_t b;
            if (oa.spin == 7u) b = (uint64_t)prv * (kMapGenTile + kMapGenTile / 16);      // EXPERIMENT (TRRE_MAPGEN_NOLB): no look-back, the output is void
            else b = mg_lb_finish(lb, oa.desc, oa.gsum, oa.ginc, prv, (uint64_t)total_prv, oa.spin, a.status, ok);
            if (lid == 0) {
                misc[12] = ok ? 0u : 1u;
                *reinterpret_cast<uint64_t*>(misc + 10) = b;
                if (prv == oa.n_tiles - 1) *oa.total = b + total_prv;
                if (oa.dbg) { oa.dbg[16 * prv] = total_prv; oa.dbg[16 * prv + 1] = b; oa.dbg[16 * prv + 10] = blockIdx.x; oa.dbg[16 * prv + 11] = clock64(); oa.dbg[16 * prv + 12] = first; oa.dbg[16 * prv + 13] = clock64() - lb_t0; }
            }
        };
        if (wave == 0 && prv >= 0) {
            if (prv > 0 && mg_lb_step(lb, prv, poll_prv) && !lb.done) (void)mg_lb_step(lb, prv, poll_prv2);
            if (lb.done || oa.spin == 7u) {
                lb_close(true);
                if (lid == 0) misc[14] = 0;
            } else {
                again = mg_poll(oa.desc, oa.gsum, oa.ginc, prv, lb.gtop, lb.tiles_done);
                lb_pending = true;
                if (lid == 0) misc[14] = 1;
            }
        }
        stamp(tclk, 1);
        // ---- `cur` into its window (it does not need its place) ---------------------------------------------------------------------------
        const uint32_t total_cur = total, woff_cur = woff;
        const bool fits_cur = cur >= 0 && total_cur != 0u && total_cur <= W && !is_edge(cur);
        if (fits_cur) {
            uint8_t* const wk = win_of(cur);
            const uint32_t sk = sink_of(cur);
            uint32_t rowbase = woff_cur;
#pragma unroll
            for (int k = 0; k < kMgRows / 2; ++k) {
                const uint32_t e = inc[k] - u[k];
                // (the rows' bytes are picked apart anew: what the count pass extracted is not kept — 64 registers, spilled)
                asm volatile("" : "+v"(xlo[2 * k]), "+v"(xhi[2 * k]), "+v"(xlo[2 * k + 1]), "+v"(xhi[2 * k + 1]));
                mg_expand8<kFirst, kMulti, false, false>(T, xlo[2 * k], xhi[2 * k], MgEdge{}, wk, sk, rowbase + (e & 0xffffu), 0u, 0u);
                rowbase += tot[k] & 0xffffu;
                TRRE_SCHED_FENCE();
                mg_expand8<kFirst, kMulti, false, false>(T, xlo[2 * k + 1], xhi[2 * k + 1], MgEdge{}, wk, sk, rowbase + (e >> 16), 0u, 0u);
                rowbase += tot[k] >> 16;
                TRRE_SCHED_FENCE();
            }
        }
        stamp(tclk, 2);
        // ---- the next tile: its bytes are here (asked for two trips ago); the one after it is asked for, the next one counted and its total
        // out — a whole trip before this workgroup, and the others, look back at it -----------------------------------------------------------
        int64_t nxt = cur >= 0 && cur + G < oa.n_tiles ? cur + G : -1;
        uint32_t st_nxt = 0;
        if (nxt >= 0) {
#pragma unroll
            for (int r = 0; r < kMgRows; ++r) { xlo[r] = ylo[r]; xhi[r] = yhi[r]; }
            if (!lb_pending) ask(nxt + G, ylo, yhi);       // (wave 0 with a look-back still open asks below: nothing of its own shall lie between it and the answers)
            st_nxt = count_and_publish(nxt);               // (barrier)
        } else {
            MG_SYNC();
        }
        stamp(tclk, 3);
        // ---- `prv` leaves: its window as aligned lines (a tile of several windows, or at an end of the input: expanded here, window by window)
        if (prv >= 0 && misc[14]) {                        // (uniform) the look-back's second answers, and whatever else it takes
            if (wave == 0) {
                (void)mg_lb_step(lb, prv, again);
                lb_close(false);
                if (nxt >= 0) ask(nxt + G, ylo, yhi);
            }
            MG_SYNC();
        }
        if (prv >= 0) {
            if (misc[12]) { st_all |= kStOneVoid; break; } // (uniform: a look-back gave up, the launch is void — the pair will run the buffer)
            const uint64_t base = *reinterpret_cast<const uint64_t*>(misc + 10);
            const bool write = base + total_prv <= a.cap;
            if (!write) st_prv |= kStCapacity;
            if (fits_prv) {
                store_window(prv, base, 0u, total_prv, write);
            } else {
                for (uint32_t wlo = 0; wlo < total_prv; wlo += W) {
                    const uint32_t wsize = wlo + W < total_prv ? W : total_prv - wlo;
                    if (wlo) MG_SYNC();                    // (the window before has left)
                    expand_clipped(prv, woff_prv, wlo, wsize);
                    MG_SYNC();
                    store_window(prv, base, wlo, wsize, write);
                }
            }
            st_all |= st_prv;
        }
        // ---- `cur`'s look-back is asked for: looked at a trip from now ---------------------------------------------------------------------
        if (wave == 0 && cur > 0) {
            poll_prv = mg_poll(oa.desc, oa.gsum, oa.ginc, cur, (cur >> 6) - 1, false);
            poll_prv2 = mg_poll(oa.desc, oa.gsum, oa.ginc, cur, (cur >> 6) - 65, true);
        }
        stamp(tclk, 4);
        MG_SYNC();                                         // (`prv`'s window has left: the tile after `cur` will be expanded into it)
        stamp(tclk, 5);
        if (prof) atomicAdd(reinterpret_cast<unsigned long long*>(oa.prof + 7), 1ull);
        prv = cur; st_prv = st_cur; total_prv = total_cur; woff_prv = woff_cur; fits_prv = fits_cur;
        cur = nxt; st_cur = st_nxt;
    }
    st_all = wave_or(st_all);
    if (st_all && lid == 0) atomicOr(a.status, st_all);
}
template <bool kFirst, bool kMulti>
int launch_mapgen_t(const ScanArgs& a, const MapGenArgs& oa, hipStream_t s, int lds, int cus) {
    static std::atomic<int> allowed{0}, per_cu_cache{0};
    if (allowed.load() < lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mapgen<kFirst, kMulti>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        allowed.store(lds);
        per_cu_cache.store(0);
    }
    // every workgroup of the grid must be resident (a tile's look-back waits for the tiles of the others): as many as the runtime says fit
    if (!per_cu_cache.load()) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mapgen<kFirst, kMulti>, kMapGenThreads, (size_t)lds) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            return -1;
        }
        per_cu_cache.store(n);
    }
    const int per_cu = per_cu_cache.load();
    int64_t blocks = (int64_t)cus * per_cu;
    static const int oversub_env = getenv("TRRE_MAPGEN_OVERSUB") ? atoi(getenv("TRRE_MAPGEN_OVERSUB")) : 0;   // (tests: a grid that is NOT resident — the launch must give up, not hang)
    if (oversub_env > 1) blocks *= oversub_env;
    if (blocks > oa.n_tiles) blocks = oa.n_tiles;
    hipLaunchKernelGGL((k_mapgen<kFirst, kMulti>), dim3((unsigned)blocks), dim3(kMapGenThreads), lds, s, a, oa);
    return 0;
}
int launch_mapgen(const ScanArgs& a, const MapGenArgs& oa, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int lds = 512 + 2048 + 128 + 2 * ((int)oa.window + 32) + 4 * kMapGenThreads + 32;
    if (lds > kLdsLimit - 1024) return -1;
    static std::atomic<int> cus_cache{0};
    if (!cus_cache.load()) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t prop;
        cus_cache.store(hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    }
    const int cus = cus_cache.load();
    const bool multi = oa.longest > 1u;
    if (oa.first_lookup) return multi ? launch_mapgen_t<true, true>(a, oa, s, lds, cus) : launch_mapgen_t<true, false>(a, oa, s, lds, cus);
    return multi ? launch_mapgen_t<false, true>(a, oa, s, lds, cus) : launch_mapgen_t<false, false>(a, oa, s, lds, cus);
}

}  // namespace trre
