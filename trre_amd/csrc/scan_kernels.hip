// scan_kernels.hip — gfx950 kernels of the transducer scan engine.
//
// Branchy byte/integer work: no MFMA; the governing roofline is HBM bandwidth
// (1 byte read + ~1 byte written per input byte).  Kernels:
//
//   k_bytemap     memoryless tables (every attempt is one byte -> one byte):
//                 a pure streaming byte map, 16 bytes per lane per load.
//   k_scan_lp     length-preserving tables: one launch, output position ==
//                 input position; tile in LDS, one lane per line, speculative
//                 in-position writes, coalesced 16-byte copy-out.
//   k_scan_count  general tables, pass 1: output bytes per lane and per chunk.
//   k_chunk_scan  exclusive scan of the chunk totals (one workgroup).
//   k_scan_emit   general tables, pass 2: lane offsets by workgroup scan, lines
//                 emitted into an LDS staging tile, coalesced copy-out.
//
// The per-thread phase bodies live in scan_block.hpp / scan_core.hpp.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>

#include "launch.hpp"
#include "scan_block.hpp"
#include "splice_block.hpp"
#include "one_block.hpp"
#include "gen_block.hpp"
#include "lazy_block.hpp"
#include "guard_block.hpp"

namespace trre {
namespace {

constexpr int kWave = 64;

// Kernels that take more than 64 KiB of dynamic LDS need the limit raised — once per kernel and device, not per launch
// (a launch is ~5 us of host time; the call is another 2-3): the limit is set to the CU's whole 160 KiB.
constexpr int kLdsLimit = 160 * 1024;
template <auto Kernel>
void allow_big_lds() {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    done.fetch_or(bit, std::memory_order_relaxed);
}

__device__ __forceinline__ int wave_min(int v) {
    for (int d = 32; d; d >>= 1) v = min(v, __shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
    for (int d = 32; d; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    for (int d = 32; d; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d, kWave);
    return v;
}
__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
    for (int d = 32; d; d >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, kWave);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, kWave);
        v += (uint64_t)hi << 32 | lo;
    }
    return v;
}
// inclusive scan inside a wave
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    const int lane = threadIdx.x & (kWave - 1);
    for (int d = 1; d < kWave; d <<= 1) {
        uint32_t u = (uint32_t)__shfl_up((int)v, d, kWave);
        if (lane >= d) v += u;
    }
    return v;
}

// LDS carve shared by the tile kernels
template <class G, class Engine>
struct Carve {
    static constexpr int kTab = (Engine::kLdsBytes + 15) & ~15;
    static constexpr int kMask = G::TILE_ALLOC * Engine::kMaskBytes;
    static constexpr int off_tin = 0;
    static constexpr int off_tout = G::TILE_ALLOC;
    static constexpr int off_tab = 2 * G::TILE_ALLOC;
    static constexpr int off_mask = off_tab + kTab;
    static constexpr int off_red = off_mask + ((kMask + 15) & ~15);
    static constexpr int kBytes = off_red + 64 + 16 * 8;   // reduction words + per-wave partials
};

// ------------------------------------------------------------------------------------------
template <class G, class Engine>
__global__ __launch_bounds__(G::THREADS) void k_scan_lp(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using C = Carve<G, Engine>;
    uint8_t* tin = smem + C::off_tin;
    uint8_t* tout = smem + C::off_tout;
    uint8_t* tab = smem + C::off_tab;
    int32_t* red = reinterpret_cast<int32_t*>(smem + C::off_red);
    const int tid = threadIdx.x;
    const int64_t v0 = (int64_t)blockIdx.x * G::CHUNK - G::PRE;

    Engine::stage(a.blob, tab, tid, G::THREADS);
    tile_load<G>(a, v0, tin, tid);
    if (tid == 0) { red[0] = 0x7fffffff; red[1] = -1; }
    __syncthreads();

    const typename Engine::View T = Engine::view(a.blob, tab);
    typename Engine::Lane L = Engine::make_lane(smem + C::off_mask);
    int32_t first, last;
    uint32_t st = 0;
    lane_walk_lp<G, Engine>(a, T, L, v0, tin, tout, tid, first, last, st);

    first = wave_min(first);
    last = wave_max(last);
    st = wave_or(st);
    if ((tid & (kWave - 1)) == 0) {
        atomicMin(&red[0], first);
        atomicMax(&red[1], last);
        if (st) atomicOr(a.status, st);
    }
    __syncthreads();
    tile_store_lp<G>(a, v0, tout, red[0], red[1], tid);
}

// ------------------------------------------------------------------------------------------
template <class G, class Engine>
__global__ __launch_bounds__(G::THREADS) void k_scan_count(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using C = Carve<G, Engine>;
    uint8_t* tin = smem + C::off_tin;
    uint8_t* tab = smem + C::off_tab;
    uint64_t* part = reinterpret_cast<uint64_t*>(smem + C::off_red + 64);
    const int tid = threadIdx.x;
    const int64_t v0 = (int64_t)blockIdx.x * G::CHUNK - G::PRE;

    Engine::stage(a.blob, tab, tid, G::THREADS);
    tile_load<G>(a, v0, tin, tid);
    __syncthreads();

    const typename Engine::View T = Engine::view(a.blob, tab);
    typename Engine::Lane L = Engine::make_lane(smem + C::off_mask);
    CountSink sink;
    uint32_t st = 0;
    lane_walk_gen<G, Engine>(a, T, L, v0, tin, tid, sink, st);
    if (sink.n > 0xffffffffull) { st |= kStCapacity; sink.n = 0xffffffffull; }
    a.lane_counts[(size_t)blockIdx.x * G::THREADS + tid] = (uint32_t)sink.n;

    const uint64_t wsum = wave_sum(sink.n);
    st = wave_or(st);
    if ((tid & (kWave - 1)) == 0) {
        part[tid / kWave] = wsum;
        if (st) atomicOr(a.status, st);
    }
    __syncthreads();
    if (tid == 0) {
        uint64_t t = 0;
        for (int w = 0; w < G::THREADS / kWave; ++w) t += part[w];
        a.chunk_total[blockIdx.x] = t;
    }
}

// exclusive scan of chunk_total[0..n) into chunk_base[0..n], chunk_base[n] = total
__global__ __launch_bounds__(1024) void k_chunk_scan(const uint64_t* total, uint64_t* base, int64_t n) {
    __shared__ uint64_t seg[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint64_t s = 0;
    for (int64_t k = lo; k < hi; ++k) s += total[k];
    seg[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = 0;
        for (int k = 0; k < 1024; ++k) { uint64_t t = seg[k]; seg[k] = run; run += t; }
        base[n] = run;
    }
    __syncthreads();
    uint64_t run = seg[tid];
    for (int64_t k = lo; k < hi; ++k) { base[k] = run; run += total[k]; }
}

// ------------------------------------------------------------------------------------------
template <class G, class Engine>
__global__ __launch_bounds__(G::THREADS) void k_scan_emit(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using C = Carve<G, Engine>;
    uint8_t* tin = smem + C::off_tin;
    uint8_t* tout = smem + C::off_tout;
    uint8_t* tab = smem + C::off_tab;
    uint32_t* wpart = reinterpret_cast<uint32_t*>(smem + C::off_red);
    const int tid = threadIdx.x;
    const int64_t v0 = (int64_t)blockIdx.x * G::CHUNK - G::PRE;

    Engine::stage(a.blob, tab, tid, G::THREADS);
    tile_load<G>(a, v0, tin, tid);

    // lane offsets: workgroup-wide exclusive scan of the counts from pass 1
    const uint32_t mine = a.lane_counts[(size_t)blockIdx.x * G::THREADS + tid];
    const uint32_t incl = wave_scan_incl(mine);
    if ((tid & (kWave - 1)) == kWave - 1) wpart[tid / kWave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < tid / kWave; ++w) wbase += wpart[w];
    const uint64_t lane_base = (uint64_t)wbase + incl - mine;
    const uint64_t total = a.chunk_total[blockIdx.x];
    const uint64_t gbase = a.chunk_base[blockIdx.x];
    if (gbase + total > a.cap) {                       // uniform for the workgroup
        if (tid == 0) atomicOr(a.status, kStCapacity);
        return;
    }
    const int shift = (int)((reinterpret_cast<uintptr_t>(a.out) + gbase) & 15u);
    const bool staged = (uint64_t)shift + total <= (uint64_t)G::TILE;

    const typename Engine::View T = Engine::view(a.blob, tab);
    typename Engine::Lane L = Engine::make_lane(smem + C::off_mask);
    ByteSink sink{staged ? tout + shift + lane_base : a.out + gbase + lane_base};
    uint32_t st = 0;
    lane_walk_gen<G, Engine>(a, T, L, v0, tin, tid, sink, st);
    st = wave_or(st);
    if (st && (tid & (kWave - 1)) == 0) atomicOr(a.status, st);
    if (staged) {
        __syncthreads();
        tile_store_seq<G>(a.out + gbase, tout, shift, (int64_t)total, tid);
    }
}

// ------------------------------------------------------------------------------------------
// Direct stream kernels: no tile, one long sub-range per lane (see stream_direct_lane).
constexpr int kDirectThreads = 256;
constexpr int kDirectEntBytes = 2048;      // table rows in LDS when they all fit ...
constexpr int kDirectPoolSmall = 2048;     // ... next to this much of pooled text
constexpr int kDirectPoolBytes = 4096;     // larger tables stay in global memory (L1/L2); LDS takes their pooled texts when small
constexpr int kDirectTabSmall = kDirectEntBytes + kDirectPoolSmall;
constexpr int kDirectWsc = (kDirectThreads / kWave) * kWaveScratchBytes;       // posting tables of the emit pass's unit stores
constexpr int kDirectLds = 256 + kDirectTabSmall + kDirectThreads * kRingStride + 64 + kDirectWsc;
constexpr int kDirectLdsHot = 256 + kDirectPoolBytes + kDirectThreads * kRingStride + 64 + kDirectWsc;

template <bool kLdsEnt>
__device__ __forceinline__ StreamView direct_stage(const ScanArgs& a, uint8_t* smem) {
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    for (int k = threadIdx.x; k < 256; k += kDirectThreads) smem[k] = a.blob[h.off_cls + k];
    if (kLdsEnt) {
        const uint64_t* e = reinterpret_cast<const uint64_t*>(a.blob + h.off_ent);
        uint64_t* d = reinterpret_cast<uint64_t*>(smem + 256);
        for (int k = threadIdx.x; k < (int)(h.ent_bytes / 8); k += kDirectThreads) d[k] = e[k];
    }
    StreamView T;
    // pooled replacement texts next to the table (the pool is a multiple of 4 bytes)
    uint8_t* pool_lds = smem + 256 + (kLdsEnt ? kDirectEntBytes : 0);
    if ((int)h.pool_bytes <= (kLdsEnt ? kDirectPoolSmall : kDirectPoolBytes)) {
        const uint32_t* e = reinterpret_cast<const uint32_t*>(a.blob + h.off_pool);
        uint32_t* d = reinterpret_cast<uint32_t*>(pool_lds);
        for (int k = threadIdx.x; k < (int)(h.pool_bytes / 4); k += kDirectThreads) d[k] = e[k];
        T.pool_fast = pool_lds;
    }
    __syncthreads();
    T.cls = smem;
    T.ent = kLdsEnt ? reinterpret_cast<const uint64_t*>(smem + 256) : reinterpret_cast<const uint64_t*>(a.blob + h.off_ent);
    T.pool = a.blob + h.off_pool;
    T.long_pool = h.max_out >= 255u;
    return T;
}

template <int kMode, bool kLdsEnt, bool kSym>
__global__ __launch_bounds__(kDirectThreads) void k_stream_direct(ScanArgs a, int64_t lane_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamView T = direct_stage<kLdsEnt>(a, smem);
    const uint32_t n_cls = reinterpret_cast<const StreamBlobHeader*>(a.blob)->n_cls;
    constexpr int kTab = kLdsEnt ? kDirectTabSmall : kDirectPoolBytes;
    constexpr int kLds = kLdsEnt ? kDirectLds : kDirectLdsHot;
    uint8_t* ring = smem + 256 + kTab + threadIdx.x * kRingStride;
    const int64_t lane = (int64_t)blockIdx.x * kDirectThreads + threadIdx.x;
    DirectLane L;
    uint32_t st = 0;
    uint64_t base = 0;
    // (a bounded fold that overflowed in the count pass: the launch is void, finish() runs the buffer on another family)
    if (kMode == 2 && !a.lp_emit && (*a.status & kStOverflow)) return;
    if (kMode == 2 && !a.lp_emit) {
        // lane offsets: workgroup-wide exclusive scan of the counts from the count launch
        uint32_t* wpart = reinterpret_cast<uint32_t*>(smem + kLds - kDirectWsc - 64);
        const uint32_t mine = a.lane_counts[lane];
        const uint32_t incl = wave_scan_incl(mine);
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) wpart[threadIdx.x / kWave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < (int)threadIdx.x / kWave; ++w) wbase += wpart[w];
        base = a.chunk_base[blockIdx.x] + wbase + incl - mine;
        if (a.chunk_base[blockIdx.x] + a.chunk_total[blockIdx.x] > a.cap) {
            if (threadIdx.x == 0) atomicOr(a.status, kStCapacity);
            return;
        }
    }
    uint32_t* wsc = reinterpret_cast<uint32_t*>(smem + kLds - kDirectWsc + (threadIdx.x / kWave) * kWaveScratchBytes);
    stream_direct_lane<kMode, false, kSym>(a, T, n_cls, lane, lane_bytes, ring, base, L, st, kMode == 2 ? wsc : nullptr);
    // the first lane (in stream order) on which the reference's search does not return: everything up to the point where
    // it stopped is what the reference had printed (status[1] = ~lane, the launch zeroes it)
    if (kMode == 1 && kSym && (st & kStDiverge)) atomicMax(a.status + 1, 0xffffffffu - (uint32_t)lane);
    if (kMode == 1) {
        uint64_t* part = reinterpret_cast<uint64_t*>(smem + kLds - kDirectWsc - 64);
        if (L.count > 0xffffffffull) { st |= kStCapacity; L.count = 0xffffffffull; }
        a.lane_counts[lane] = (uint32_t)L.count;
        const uint64_t wsum = wave_sum(L.count);
        if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t t = 0;
            for (int w = 0; w < kDirectThreads / kWave; ++w) t += part[w];
            a.chunk_total[blockIdx.x] = t;
        }
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}

// Count / emit passes over the 16-byte entries of a small table (front.hpp): the whole table in LDS.
//   smem: cls[256] | g16[g16_room] | pooled text (2 KiB, when the pool fits) | staging[threads] | 64
template <int kMode, int kSym, bool kHasSlow>
__global__ __launch_bounds__(kDirectThreads) void k_stream_g16(ScanArgs a, int64_t lane_bytes, int g16_room) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    for (int k = threadIdx.x; k < 256; k += kDirectThreads) smem[k] = a.blob[h.off_cls + k];
    {
        const U128* e = reinterpret_cast<const U128*>(a.blob + h.off_g16);
        U128* d = reinterpret_cast<U128*>(smem + 256);
        for (int k = threadIdx.x; k < (int)(h.g16_bytes / 16); k += kDirectThreads) d[k] = e[k];
    }
    {
        // (g16_room holds the 16-byte form and, behind it, the pair form when the tables have one)
        const U128* e = reinterpret_cast<const U128*>(a.blob + h.off_p32);
        U128* d = reinterpret_cast<U128*>(smem + 256 + ((h.g16_bytes + 15u) & ~15u));
        for (int k = threadIdx.x; k < (int)(h.p32_bytes / 16); k += kDirectThreads) d[k] = e[k];
    }
    StreamView T;
    uint8_t* pool_lds = smem + 256 + g16_room;
    if (kMode == 2 && (int)h.pool_bytes <= kDirectPoolSmall) {
        const uint32_t* e = reinterpret_cast<const uint32_t*>(a.blob + h.off_pool);
        uint32_t* d = reinterpret_cast<uint32_t*>(pool_lds);
        for (int k = threadIdx.x; k < (int)(h.pool_bytes / 4); k += kDirectThreads) d[k] = e[k];
        T.pool_fast = pool_lds;
    }
    __syncthreads();
    T.cls = smem;
    T.g16 = smem + 256;
    if (h.p32_bytes) { T.p32 = smem + 256 + ((h.g16_bytes + 15u) & ~15u); T.p32_slow = h.p32_slow; }
    T.ent = reinterpret_cast<const uint64_t*>(a.blob + h.off_ent);
    T.pool = a.blob + h.off_pool;
    T.long_pool = h.max_out >= 255u;
    // (the count pass has neither pooled texts nor staging buffers in LDS: more resident waves.  With a
    // large table in global memory that does not pay: k_stream_direct's count pass is fastest at the
    // 16 waves per CU its emit-sized LDS allows — 1.63 ms against 1.9 ms at 8 or 28 waves: cache capacity)
    uint8_t* ring = kMode == 1 ? pool_lds : pool_lds + kDirectPoolSmall + threadIdx.x * kRingStride;
    uint8_t* tail = kMode == 1 ? pool_lds : pool_lds + kDirectPoolSmall + kDirectThreads * kRingStride;       // 64 bytes
    const int64_t lane = (int64_t)blockIdx.x * kDirectThreads + threadIdx.x;
    DirectLane L;
    uint32_t st = 0;
    uint64_t base = 0;
    // (a bounded fold that overflowed in the count pass: the launch is void, finish() runs the buffer on another family)
    if (kMode == 2 && !a.lp_emit && (*a.status & kStOverflow)) return;
    // (exact sub-ranges: some lane's guessed entry state was wrong — finish() repairs the lanes, then this pass runs again; with a wrong
    // guess in the backward pass the symbols are not final: both forward passes wait for finish())
    if (kMode == 2 && a.exact == 2u && a.status[3] != 0u) return;
    if (kSym != 0 && a.exact != 0u && a.exact != 3u && a.status[2] != 0u) return;
    if (kMode == 1 && a.exact == 3u) {
        // A repair round (exact sub-ranges): a flagged lane walks again from the exit state of the lane before it, and on into the
        // lanes behind it for as long as its exit state is not what they had assumed (within this workgroup's lanes: the next round
        // sees to the rest).  Everything it leaves is open to the next k_spec_verify.
        const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
        const int64_t block_end = ((int64_t)blockIdx.x + 1) * kDirectThreads < n_lanes ? ((int64_t)blockIdx.x + 1) * kDirectThreads : n_lanes;
        if (lane > 0 && lane < n_lanes && a.spec_flags[lane]) {
            uint32_t state = a.exit_rows[lane - 1];
            const uint32_t skip_row = kSkipState * h.n_cls * 16u;
            for (int64_t j = lane;;) {
                a.entry_rows[j] = state;
                DirectLane Lj;
                // (behind a NUL the rest of the record is swallowed: a lane without a line end inside stays in SKIP and prints nothing — no walk)
                if (state == skip_row && (j + 1) * lane_bytes < a.vend - 1 && !rev_has_newline(a, j * lane_bytes, (j + 1) * lane_bytes)) {
                    a.exit_rows[j] = state;
                    a.lane_counts[j] = 0u;
                    ++j;
                    if (j >= block_end || a.spec_flags[j]) break;
                    const uint32_t e0 = a.entry_rows[j];
                    if ((e0 & 1u) || (e0 & ~1u) == state) break;
                    continue;
                }
                g16_lane<1, kSym, kHasSlow>(a, T, h.n_cls, j, lane_bytes, ring, 0, Lj, st);
                if (Lj.count > 0xffffffffull) { st |= kStCapacity; Lj.count = 0xffffffffull; }
                a.lane_counts[j] = (uint32_t)Lj.count;
                if (kSym != 0 && (st & kStDiverge)) atomicMax(a.status + 1, 0xffffffffu - (uint32_t)j);
                state = a.exit_rows[j];
                ++j;
                if (j >= block_end || a.spec_flags[j]) break;
                const uint32_t e = a.entry_rows[j];
                if ((e & 1u) || (e & ~1u) == state) break;
            }
        }
        __syncthreads();
        L.count = lane < n_lanes ? a.lane_counts[lane] : 0u;
    } else
    if (kMode == 2 && !a.lp_emit) {
        uint32_t* wpart = reinterpret_cast<uint32_t*>(tail);
        const uint32_t mine = a.lane_counts[lane];
        const uint32_t incl = wave_scan_incl(mine);
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) wpart[threadIdx.x / kWave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < (int)threadIdx.x / kWave; ++w) wbase += wpart[w];
        base = a.chunk_base[blockIdx.x] + wbase + incl - mine;
        if (a.chunk_base[blockIdx.x] + a.chunk_total[blockIdx.x] > a.cap) {
            if (threadIdx.x == 0) atomicOr(a.status, kStCapacity);
            return;
        }
    }
    uint32_t* wsc = reinterpret_cast<uint32_t*>(tail + 64) + (threadIdx.x / kWave) * (kWaveScratchBytes / 4);
    if (!(kMode == 1 && a.exact == 3u)) {
        g16_lane<kMode, kSym, kHasSlow>(a, T, h.n_cls, lane, lane_bytes, ring, base, L, st, kMode == 2 ? wsc : nullptr);
        if (kMode == 1 && kSym != 0 && (st & kStDiverge)) atomicMax(a.status + 1, 0xffffffffu - (uint32_t)lane);     // (see k_stream_direct)
    }
    if (kMode == 1) {
        uint64_t* part = reinterpret_cast<uint64_t*>(tail);
        if (L.count > 0xffffffffull) { st |= kStCapacity; L.count = 0xffffffffull; }
        a.lane_counts[lane] = (uint32_t)L.count;
        const uint64_t wsum = wave_sum(L.count);
        if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t t = 0;
            for (int w = 0; w < kDirectThreads / kWave; ++w) t += part[w];
            a.chunk_total[blockIdx.x] = t;
        }
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}

// Count / emit passes over the fallback form of a large table (front.hpp, scan_block.hpp: fb_lane): the comb of entries
// (and, for the emit pass, the literal texts) in LDS — ~75 KB for a 1000-key dictionary; the emit pass's staging rings take
// the rest of the 160 KB.  Lanes are numbered as everywhere (256 per chunk of the workspace), a workgroup covers
// kThreads / 256 chunks.
//   smem: cls[256] | comb | emit: lit | rings[kThreads] | 64 x (kThreads / 256) | posting tables
template <int kMode, int kThreads>
__global__ __launch_bounds__(kThreads, (kMode == 1 ? 8 : 2)) void k_stream_fb(ScanArgs a, int64_t lane_bytes, int64_t n_chunks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const uint32_t comb_bytes = (h.fb_slots * 8u + 15u) & ~15u, lit_bytes = kMode == 2 ? (h.fb_lits * 8u + 15u) & ~15u : 0u;
    for (int k = threadIdx.x; k < 256; k += kThreads) smem[k] = a.blob[h.off_cls + k];
    {
        const U128* e = reinterpret_cast<const U128*>(a.blob + h.off_fb_comb);
        U128* d = reinterpret_cast<U128*>(smem + 256);
        for (int k = threadIdx.x; k < (int)(comb_bytes / 16); k += kThreads) d[k] = e[k];
        e = reinterpret_cast<const U128*>(a.blob + h.off_fb_lit);
        d = reinterpret_cast<U128*>(smem + 256 + comb_bytes);
        for (int k = threadIdx.x; k < (int)(lit_bytes / 16); k += kThreads) d[k] = e[k];
    }
    __syncthreads();
    FbView T;
    T.cls = smem;
    T.comb = reinterpret_cast<const uint64_t*>(smem + 256);
    T.lit = reinterpret_cast<const uint64_t*>(smem + 256 + comb_bytes);
    T.lit_meta = nullptr;
    T.esc_slot = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc_slot);
    T.esc = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc);
    T.pool = a.blob + h.off_fb_pool;
    T.n_esc = h.fb_escs;
    for (int i = 0; i < 3; ++i) { T.start[i][0] = h.fb_start[i][0]; T.start[i][1] = h.fb_start[i][1]; }
    uint8_t* top = smem + 256 + comb_bytes + lit_bytes;
    constexpr int kGroups = kThreads / kDirectThreads;                 // chunks of the workspace per workgroup
    uint8_t* ring = top + threadIdx.x * kBRingStride + kBRingPad;      // (byte-granular staging: 8 spare bytes on either side)
    uint8_t* tail = kMode == 1 ? top : top + kThreads * kBRingStride;  // 64 bytes per group
    uint32_t* wsc = reinterpret_cast<uint32_t*>(tail + 64 * kGroups) + (threadIdx.x / kWave) * (kWaveScratchBytes / 4);
    const int group = threadIdx.x / kDirectThreads, gtid = threadIdx.x % kDirectThreads;
    const int64_t chunk = (int64_t)blockIdx.x * kGroups + group;
    const int64_t lane = chunk * kDirectThreads + gtid;
    const bool live = chunk < n_chunks;
    DirectLane L;
    uint32_t st = 0;
    uint64_t base = 0;
    if (kMode == 2) {
        uint32_t* wpart = reinterpret_cast<uint32_t*>(tail + 64 * group);
        const uint32_t mine = live ? a.lane_counts[lane] : 0u;
        const uint32_t incl = wave_scan_incl(mine);
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) wpart[gtid / kWave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < gtid / kWave; ++w) wbase += wpart[w];
        if (live) base = a.chunk_base[chunk] + wbase + incl - mine;
        // (bases grow with the chunk index: if the last chunk of this workgroup does not fit, the output is void anyway)
        const int64_t last = ((int64_t)blockIdx.x + 1) * kGroups - 1 < n_chunks - 1 ? ((int64_t)blockIdx.x + 1) * kGroups - 1 : n_chunks - 1;
        if (a.chunk_base[last] + a.chunk_total[last] > a.cap) {
            if (threadIdx.x == 0) atomicOr(a.status, kStCapacity);
            return;
        }
    }
    // (a lane of a chunk beyond the workspace starts beyond the input: it is done before it begins)
    fb_lane<kMode>(a, T, live ? lane : (a.vend + lane_bytes - 1) / lane_bytes, lane_bytes, ring, base, L, st, kMode == 2 ? wsc : nullptr);
    if (kMode == 1) {
        uint64_t* part = reinterpret_cast<uint64_t*>(tail + 64 * group);
        if (L.count > 0xffffffffull) { st |= kStCapacity; L.count = 0xffffffffull; }
        if (live) a.lane_counts[lane] = (uint32_t)L.count;
        const uint64_t wsum = wave_sum(L.count);
        if ((threadIdx.x & (kWave - 1)) == 0) part[gtid / kWave] = wsum;
        __syncthreads();
        if (gtid == 0 && live) {
            uint64_t t = 0;
            for (int w = 0; w < kDirectThreads / kWave; ++w) t += part[w];
            a.chunk_total[chunk] = t;
        }
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}
// The copy form of a large table (scan_block.hpp: fb_lane<3> / fb_copy_lane).
// Mark pass: the comb walk of k_stream_fb<1> plus the events; one 1024-lane workgroup per CU (the event stages take the
// LDS the count pass's second workgroup has).   smem: cls[256] | comb | literals' meta (u16) | event stages[1024 x 68] | 64 x groups
constexpr int kFbMarkThreads = 1024;
__global__ __launch_bounds__(kFbMarkThreads, 4) void k_fb_mark(ScanArgs a, FbCopyArgs ca, int64_t lane_bytes, int64_t n_chunks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const uint32_t comb_bytes = (h.fb_slots * 8u + 15u) & ~15u, meta_bytes = (h.fb_lits * 2u + 15u) & ~15u;
    for (int k = threadIdx.x; k < 256; k += kFbMarkThreads) smem[k] = a.blob[h.off_cls + k];
    {
        const U128* e = reinterpret_cast<const U128*>(a.blob + h.off_fb_comb);
        U128* d = reinterpret_cast<U128*>(smem + 256);
        for (int k = threadIdx.x; k < (int)(comb_bytes / 16); k += kFbMarkThreads) d[k] = e[k];
        const uint16_t* m = reinterpret_cast<const uint16_t*>(a.blob + h.off_fb_lit_meta);
        uint16_t* dm = reinterpret_cast<uint16_t*>(smem + 256 + comb_bytes);
        for (int k = threadIdx.x; k < (int)h.fb_lits; k += kFbMarkThreads) dm[k] = m[k];
    }
    __syncthreads();
    FbView T;
    T.cls = smem;
    T.comb = reinterpret_cast<const uint64_t*>(smem + 256);
    T.lit = nullptr;
    T.lit_meta = reinterpret_cast<const uint16_t*>(smem + 256 + comb_bytes);
    T.esc_slot = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc_slot);
    T.esc = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc);
    T.pool = a.blob + h.off_fb_pool;
    T.n_esc = h.fb_escs;
    for (int i = 0; i < 3; ++i) { T.start[i][0] = h.fb_start[i][0]; T.start[i][1] = h.fb_start[i][1]; }
    uint8_t* stage = smem + 256 + comb_bytes + meta_bytes + threadIdx.x * (kMarkStageStride * 4);
    uint8_t* tail = smem + 256 + comb_bytes + meta_bytes + kFbMarkThreads * (kMarkStageStride * 4);
    constexpr int kGroups = kFbMarkThreads / kDirectThreads;
    const int group = threadIdx.x / kDirectThreads, gtid = threadIdx.x % kDirectThreads;
    const int64_t chunk = (int64_t)blockIdx.x * kGroups + group;
    const int64_t lane = chunk * kDirectThreads + gtid;
    const bool live = chunk < n_chunks;
    DirectLane L;
    uint32_t st = 0;
    // (a lane of a chunk beyond the workspace is not walked: it has no header and no events)
    if (live) fb_lane<3>(a, T, lane, lane_bytes, stage, 0, L, st, nullptr, &ca);
    uint64_t* part = reinterpret_cast<uint64_t*>(tail + 64 * group);
    if (L.count > 0xffffffffull) { st |= kStCapacity; L.count = 0xffffffffull; }
    if (live) a.lane_counts[lane] = (uint32_t)L.count;
    const uint64_t wsum = wave_sum(live ? L.count : 0ull);
    if ((threadIdx.x & (kWave - 1)) == 0) part[gtid / kWave] = wsum;
    __syncthreads();
    if (gtid == 0 && live) {
        uint64_t t = 0;
        for (int w = 0; w < kDirectThreads / kWave; ++w) t += part[w];
        a.chunk_total[chunk] = t;
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}
// The same pass on the mark form of the comb (scan_block.hpp: fb_mark4_lane; front.hpp: fb_comb4): 32-bit entries.
//   smem: 4 x cls[256] | comb4 | dense4 | literals' meta (u16) | event stages[1024 x 68] | 64 x groups
__global__ __launch_bounds__(kFbMarkThreads, 4) void k_fb_mark4(ScanArgs a, FbCopyArgs ca, int64_t lane_bytes, int64_t n_chunks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const uint32_t comb_bytes = (h.fb4_slots * 4u + 15u) & ~15u, dense_bytes = h.fb4_dense * 128u, meta_bytes = (h.fb_lits * 2u + 15u) & ~15u;
    for (int k = threadIdx.x; k < 256; k += kFbMarkThreads) smem[k] = (uint8_t)(a.blob[h.off_cls + k] << 2);
    {
        const U128* e = reinterpret_cast<const U128*>(a.blob + h.off_fb_comb4);
        U128* d = reinterpret_cast<U128*>(smem + 256);
        for (int k = threadIdx.x; k < (int)(comb_bytes / 16); k += kFbMarkThreads) d[k] = e[k];
        e = reinterpret_cast<const U128*>(a.blob + h.off_fb_dense4);
        d = reinterpret_cast<U128*>(smem + 256 + comb_bytes);
        for (int k = threadIdx.x; k < (int)(dense_bytes / 16); k += kFbMarkThreads) d[k] = e[k];
        const uint16_t* m = reinterpret_cast<const uint16_t*>(a.blob + h.off_fb_lit_meta);
        uint16_t* dm = reinterpret_cast<uint16_t*>(smem + 256 + comb_bytes + dense_bytes);
        for (int k = threadIdx.x; k < (int)h.fb_lits; k += kFbMarkThreads) dm[k] = m[k];
    }
    __syncthreads();
    Fb4View T;
    T.cls4 = smem;
    T.comb4 = reinterpret_cast<const uint32_t*>(smem + 256);
    T.dense4 = reinterpret_cast<const uint32_t*>(smem + 256 + comb_bytes);
    T.lit_meta = reinterpret_cast<const uint16_t*>(smem + 256 + comb_bytes + dense_bytes);
    T.dense_base = reinterpret_cast<const uint16_t*>(a.blob + h.off_fb_dense_base);
    T.esc_slot = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc_slot);
    T.esc = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc);
    T.n_esc = h.fb_escs;
    T.pad = h.fb_pad;
    for (int i = 0; i < 3; ++i) T.start[i] = h.fb_start4[i];
    uint8_t* top = smem + 256 + comb_bytes + dense_bytes + meta_bytes;
    uint8_t* stage = top + threadIdx.x * (kMarkStageStride * 4);
    uint8_t* tail = top + kFbMarkThreads * (kMarkStageStride * 4);
    constexpr int kGroups = kFbMarkThreads / kDirectThreads;
    const int group = threadIdx.x / kDirectThreads, gtid = threadIdx.x % kDirectThreads;
    const int64_t chunk = (int64_t)blockIdx.x * kGroups + group;
    const int64_t lane = chunk * kDirectThreads + gtid;
    const bool live = chunk < n_chunks;
    DirectLane L;
    uint32_t st = 0;
    if (live) fb_mark4_lane(a, T, lane, lane_bytes, stage, L, st, ca);
    uint64_t* part = reinterpret_cast<uint64_t*>(tail + 64 * group);
    if (L.count > 0xffffffffull) { st |= kStCapacity; L.count = 0xffffffffull; }
    if (live) a.lane_counts[lane] = (uint32_t)L.count;
    const uint64_t wsum = wave_sum(live ? L.count : 0ull);
    if ((threadIdx.x & (kWave - 1)) == 0) part[gtid / kWave] = wsum;
    __syncthreads();
    if (gtid == 0 && live) {
        uint64_t t = 0;
        for (int w = 0; w < kDirectThreads / kWave; ++w) t += part[w];
        a.chunk_total[chunk] = t;
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}
// The second pass — no automaton — as a wave-cooperative splice (splice_block.hpp): a workgroup takes kThreads / 256 chunks of the workspace (256
// sub-ranges of the mark pass each), the four waves of a chunk take its sub-ranges in turn, a whole wave on each.  The pass
// is a chain of dependent steps per window (edits -> offsets -> markers -> phase A -> phase B -> store), so it wants waves to
// switch between: 1024 threads share one copy of the literals, two such workgroups fill a CU's 32 wave slots.
//   smem: literals[fb_lits x 16] | output bases of the sub-ranges (u64) | 64 per chunk | the per-wave carves
template <int kThreads, bool kLitLds>
__global__ __launch_bounds__(kThreads) void k_fb_splice(ScanArgs a, FbCopyArgs ca, int64_t lane_bytes, int64_t n_chunks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    U128* lit = reinterpret_cast<U128*>(smem);
    const uint64_t* tx = reinterpret_cast<const uint64_t*>(a.blob + h.off_fb_lit);
    const uint16_t* tm = reinterpret_cast<const uint16_t*>(a.blob + h.off_fb_lit_meta);
    if (kLitLds)
        for (int k = threadIdx.x; k < (int)h.fb_lits; k += kThreads) lit[k] = U128{(uint32_t)tx[k], (uint32_t)(tx[k] >> 32), (uint32_t)tm[k], 0u};
    constexpr int kGroups = kThreads / kDirectThreads;
    uint64_t* sbase = reinterpret_cast<uint64_t*>(smem + (kLitLds ? h.fb_lits * 16u : 0u));
    uint32_t* wparts = reinterpret_cast<uint32_t*>(sbase + kThreads);
    uint8_t* carve = reinterpret_cast<uint8_t*>(wparts + 16 * kGroups);
    const int group = threadIdx.x / kDirectThreads, gtid = threadIdx.x % kDirectThreads;
    const int64_t chunk = (int64_t)blockIdx.x * kGroups + group;
    const bool live = chunk < n_chunks;
    const int64_t lane0 = chunk * kDirectThreads;
    uint32_t* wpart = wparts + 16 * group;
    const uint32_t mine = live ? a.lane_counts[lane0 + gtid] : 0u;
    const uint32_t incl = wave_scan_incl(mine);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) wpart[gtid / kWave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < gtid / kWave; ++w) wbase += wpart[w];
    sbase[threadIdx.x] = live ? a.chunk_base[chunk] + wbase + incl - mine : 0ull;
    // (bases grow with the chunk index: if the last chunk of this workgroup does not fit, the output is void anyway)
    const int64_t last = ((int64_t)blockIdx.x + 1) * kGroups - 1 < n_chunks - 1 ? ((int64_t)blockIdx.x + 1) * kGroups - 1 : n_chunks - 1;
    if (a.chunk_base[last] + a.chunk_total[last] > a.cap) {               // (uniform for the workgroup)
        if (threadIdx.x == 0) atomicOr(a.status, kStCapacity);
        return;
    }
    // a void launch (the mark pass met a NUL, or more events than a row holds): nothing is written, finish() runs the count / emit pair
    if (*a.status & (kStEditOverflow | kStNul)) return;
    __syncthreads();
    if (!live) return;
    SpliceTables T;
    if (kLitLds) T.lit = lit;
    else { T.lit_text = tx; T.lit_meta = tm; }
    T.esc = reinterpret_cast<const uint32_t*>(a.blob + h.off_fb_esc);
    T.pool = a.blob + h.off_fb_pool;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);             // in the workgroup
    const int gwave = wave % (kDirectThreads / kWave);                                     // in its chunk
    const SpliceLds L{carve + wave * kSpLdsPerWave};
    // (the chunk's waves take neighbouring sub-ranges at the same time: neighbouring lines of the output)
    const SpliceWork W{lane0 + gwave, kDirectThreads / kWave, kWave, sbase + group * kDirectThreads + gwave, kDirectThreads / kWave};
    fb_splice_ranges<2>(a, T, ca, W, lane_bytes, L);
}
// Generator modes (gen_block.hpp): count / emit passes of the enumeration, a lane per sub-range of lane_bytes.  Lanes are
// numbered as everywhere (256 per chunk of the workspace).
constexpr int kGenThreads = kDirectThreads;
template <int kMode>
__global__ __launch_bounds__(kGenThreads) void k_gen(ScanArgs a, GenArgs ga, int64_t lane_bytes) {
    __shared__ uint64_t part[kGenThreads / kWave];
    __shared__ uint32_t wpart[kGenThreads / kWave];
    const int64_t lane = (int64_t)blockIdx.x * kGenThreads + threadIdx.x;
    const GenView G = gen_view(a.blob);
    uint64_t base = 0;
    if (kMode == 2) {
        const uint32_t mine = a.lane_counts[lane];
        const uint32_t incl = wave_scan_incl(mine);
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) wpart[threadIdx.x / kWave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < (int)threadIdx.x / kWave; ++w) wbase += wpart[w];
        base = a.chunk_base[blockIdx.x] + wbase + incl - mine;
        if (a.chunk_base[blockIdx.x] + a.chunk_total[blockIdx.x] > a.cap) {
            if (threadIdx.x == 0) atomicOr(a.status, kStCapacity);
            return;
        }
    }
    DirectLane L;
    uint32_t st = 0;
    gen_lane<kMode>(a, G, ga, lane, lane_bytes, base, L, st);
    if (kMode == 1) {
        if (L.count > 0xffffffffull) { st |= kStCapacity; L.count = 0xffffffffull; }
        a.lane_counts[lane] = (uint32_t)L.count;
        const uint64_t wsum = wave_sum(L.count);
        if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t t = 0;
            for (int w = 0; w < kGenThreads / kWave; ++w) t += part[w];
            a.chunk_total[blockIdx.x] = t;
        }
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}

// The backtracking fallback (gen_block.hpp: bt_lane): persistent workgroups, one chunk of 256 lanes per turn — a thread's
// stack and path buffer are its own for the whole launch.
template <int kMode>
__global__ __launch_bounds__(kGenThreads) void k_bt(ScanArgs a, GenArgs ga, int64_t lane_bytes, int64_t n_chunks, uint32_t budget) {
    __shared__ uint64_t part[kGenThreads / kWave];
    __shared__ uint32_t wpart[kGenThreads / kWave];
    const GenView G = gen_view(a.blob);
    const int64_t slot = (int64_t)blockIdx.x * kGenThreads + threadIdx.x;
    __shared__ uint32_t seen_status;
    uint32_t st = 0;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        // (a sub-range gave up: the launch is an error whatever the others find — leave)
        if (threadIdx.x == 0) seen_status = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (seen_status & kStEditOverflow) break;
        const int64_t lane = chunk * kGenThreads + threadIdx.x;
        uint64_t base = 0;
        bool skip = false;
        if (kMode == 2) {
            const uint32_t mine = a.lane_counts[lane];
            const uint32_t incl = wave_scan_incl(mine);
            __syncthreads();
            if ((threadIdx.x & (kWave - 1)) == kWave - 1) wpart[threadIdx.x / kWave] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (int w = 0; w < (int)threadIdx.x / kWave; ++w) wbase += wpart[w];
            base = a.chunk_base[chunk] + wbase + incl - mine;
            if (a.chunk_base[chunk] + a.chunk_total[chunk] > a.cap) { st |= kStCapacity; skip = true; }
        }
        DirectLane L;
        uint32_t lst = 0;
        uint32_t why = 0;
        if (!skip) bt_lane<kMode>(a, G, ga, slot, lane, lane_bytes, base, budget, L, lst, why);
        if (lst & kStEditOverflow) { atomicOr(a.status + 2, why); atomicOr(a.status, kStEditOverflow); }      // (status[2]: which limit, for the runtime's next try)
        if (kMode == 1 && (lst & kStDiverge)) atomicMax(a.status + 1, 0xffffffffu - (uint32_t)lane);     // (see k_stream_direct)
        st |= lst;
        if (kMode == 1) {
            if (L.count > 0xffffffffull) { st |= kStCapacity; L.count = 0xffffffffull; }
            a.lane_counts[lane] = (uint32_t)L.count;
            const uint64_t wsum = wave_sum(L.count);
            __syncthreads();
            if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = wsum;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint64_t t = 0;
                for (int w = 0; w < kGenThreads / kWave; ++w) t += part[w];
                a.chunk_total[chunk] = t;
            }
        }
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}

// The deterministic engine on tables still being built (lazy_block.hpp): a thread per sub-range; a lane with a result from an
// earlier round of the same scan keeps it (the tables only grow), the others walk again.
template <int kMode>
__global__ __launch_bounds__(kGenThreads) void k_lazy(ScanArgs a, LazyArgs la, int64_t lane_bytes, int rows_l) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];    // cls[256] | the first rows_l rows
    __shared__ uint64_t part[kGenThreads / kWave];
    __shared__ uint32_t wpart[kGenThreads / kWave];
    for (int k = threadIdx.x; k < 256; k += kGenThreads) smem[k] = la.cls[k];
    {
        uint64_t* d = reinterpret_cast<uint64_t*>(smem + 256);
        const int n = rows_l * (int)la.n_cls;
        for (int k = threadIdx.x; k < n; k += kGenThreads) d[k] = __hip_atomic_load(la.ent + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    la.cls_l = smem;
    la.ent_l = reinterpret_cast<const uint64_t*>(smem + 256);
    la.rows_l = (uint32_t)rows_l;
    const int64_t lane = (int64_t)blockIdx.x * kGenThreads + threadIdx.x;
    uint64_t base = 0;
    if (kMode == 2) {
        // the count pass met an unexplored edge (or gave up): its sizes are not final, nothing is written — finish() runs the next round
        if (*a.status & (kStMiss | kStEditOverflow | kStDiverge)) return;
        const uint32_t mine = a.lane_counts[lane];
        const uint32_t incl = wave_scan_incl(mine);
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) wpart[threadIdx.x / kWave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < (int)threadIdx.x / kWave; ++w) wbase += wpart[w];
        base = a.chunk_base[blockIdx.x] + wbase + incl - mine;
        if (a.chunk_base[blockIdx.x] + a.chunk_total[blockIdx.x] > a.cap) {
            if (threadIdx.x == 0) atomicOr(a.status, kStCapacity);
            return;
        }
    }
    DirectLane L;
    uint32_t st = 0;
    bool voided = false;
    if (kMode == 1) {
        const uint32_t have = a.lane_counts[lane];
        if (have != kLazyVoid) {
            L.count = have;
        } else {
            lazy_lane<1>(a, la, lane, lane_bytes, 0, L, st, voided);
            if (L.count >= kLazyVoid) { st |= kStCapacity; L.count = kLazyVoid - 1u; }
            a.lane_counts[lane] = voided || (st & (kStEditOverflow | kStDiverge)) ? kLazyVoid : (uint32_t)L.count;
            if (voided) L.count = 0;
        }
        const uint64_t wsum = wave_sum(L.count);
        if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t t = 0;
            for (int w = 0; w < kGenThreads / kWave; ++w) t += part[w];
            a.chunk_total[blockIdx.x] = t;
        }
    } else {
        lazy_lane<2>(a, la, lane, lane_bytes, base, L, st, voided, a.lane_counts[lane]);
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}

// Exact sub-ranges (scan_block.hpp: ScanArgs::exact): lane i's entry state must be the exit state of lane i - 1 — an entry that was
// derived from a line start inside the look-back window is right by construction (bit 0), a guessed one is checked here.
__global__ __launch_bounds__(256) void k_spec_verify(ScanArgs a, int64_t n_lanes) {
    const int64_t lane = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (lane < n_lanes) {
        const uint32_t e = a.entry_rows[lane];
        bad = lane > 0 && !(e & 1u) && (e & ~1u) != a.exit_rows[lane - 1];
        a.spec_flags[lane] = bad ? 1u : 0u;
    }
    const uint32_t n = (uint32_t)__popcll(__ballot(bad));
    if (n && (threadIdx.x & (kWave - 1)) == 0) atomicAdd(a.status + 3, n);
}
// Are there long lines?  4 096 samples spread over the input, a WAVE each: its lanes look at 1 KiB of the `window` bytes behind the sample
// point at a time and leave at the first '\n' (on text: the first step); out[0] counts the samples that find none.  (A thread per sample took
// 553 us — 2 048 dependent loads each; this form: a few microseconds.)
__global__ __launch_bounds__(256) void k_line_probe(ScanArgs a, int64_t window, uint32_t* out) {
    const int64_t t = ((int64_t)blockIdx.x * 256 + threadIdx.x) / kWave, n_samples = (int64_t)gridDim.x * 256 / kWave;
    const int lid = threadIdx.x & (kWave - 1);
    const int64_t n = a.vend - a.vbeg;
    const int64_t v0 = (a.vbeg + (int64_t)(((__int128)n * (2 * t + 1)) / (2 * n_samples))) & ~(int64_t)15;
    bool found = v0 + window > a.vend;                // (a window that runs into the end of the input says nothing)
    for (int64_t v = v0; v < v0 + window && !found; v += 16 * kWave) {
        const U128 q = direct_load(a, v + 16 * lid);
        const uint32_t x0 = q.x ^ 0x0a0a0a0au, x1 = q.y ^ 0x0a0a0a0au, x2 = q.z ^ 0x0a0a0a0au, x3 = q.w ^ 0x0a0a0a0au;
        const bool mine = ((((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1) | ((x2 - 0x01010101u) & ~x2) | ((x3 - 0x01010101u) & ~x3)) & 0x80808080u) != 0u &&
                          v + 16 * lid < v0 + window;
        found = __ballot(mine) != 0ull;
    }
    if (!found && lid == 0) atomicAdd(out, 1u);
}

// The stack guard (guard_block.hpp): windows without a '\n' (a bit per window, a wave per 64 of them), then the reference's
// search itself on the lines that cover them, a thread per line from a pool of stacks.
__global__ __launch_bounds__(256) void k_guard_probe(ScanArgs a, int64_t window, int64_t n_windows, uint64_t* flags) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const GuardBlobHeader& h = *reinterpret_cast<const GuardBlobHeader*>(a.blob);
    const uint32_t bset[8] = {h.bset[0], h.bset[1], h.bset[2], h.bset[3], h.bset[4], h.bset[5], h.bset[6], h.bset[7]};
    bool none = false;
    if (w < n_windows) none = guard_probe(a, bset, a.vbeg + w * window, a.vbeg + (w + 1) * window);
    const uint64_t m = __ballot(none);
    if ((threadIdx.x & (kWave - 1)) == 0) flags[w >> 6] = m;
}
template <bool kOut>
__global__ __launch_bounds__(64) void k_guard(ScanArgs a, GuardArgs ga, int64_t n_runs) {
    const int64_t slot = (int64_t)blockIdx.x * 64 + threadIdx.x, stride = (int64_t)gridDim.x * 64;
    for (int64_t r = slot; r < n_runs; r += stride) guard_line<kOut>(a, ga, slot, r);
}

// ------------------------------------------------------------------------------------------
// memoryless byte map: out[v] = map[in[v]], 16 bytes per lane per step
constexpr int kMapThreads = 256;

template <int kUnroll, bool kNonTemporal>
__global__ __launch_bounds__(kMapThreads) void k_bytemap(ScanArgs a, int64_t nvec) {
    __shared__ uint8_t map[256];
    const DftBlobHeader& h = *reinterpret_cast<const DftBlobHeader*>(a.blob);
    map[threadIdx.x] = a.blob[h.off_bytemap + threadIdx.x];
    __syncthreads();
    const bool aligned = (reinterpret_cast<uintptr_t>(a.out_v0) & 15u) == 0;
    const int64_t vfirst = a.vbeg & ~(int64_t)15;
    uint32_t zero = 0;
    const int64_t stride = (int64_t)gridDim.x * kMapThreads * kUnroll;
    for (int64_t base = (int64_t)blockIdx.x * kMapThreads * kUnroll; base < nvec; base += stride) {
        U128 w[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t k = base + u * kMapThreads + threadIdx.x;
            if (k < nvec) {
                const U128* src = reinterpret_cast<const U128*>(a.in_v0 + vfirst + k * 16);
                if (kNonTemporal) {
                    w[u].x = __builtin_nontemporal_load(&src->x); w[u].y = __builtin_nontemporal_load(&src->y);
                    w[u].z = __builtin_nontemporal_load(&src->z); w[u].w = __builtin_nontemporal_load(&src->w);
                } else {
                    w[u] = *src;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t k = base + u * kMapThreads + threadIdx.x;
            if (k >= nvec) continue;
            const int64_t v = vfirst + k * 16;
            if (kNonTemporal && aligned && v >= a.vbeg && v + 16 <= a.vend - 1) {
                U128 r;
                r.x = map4(map, w[u].x); r.y = map4(map, w[u].y); r.z = map4(map, w[u].z); r.w = map4(map, w[u].w);
                const uint32_t z = has_zero_byte(w[u].x) | has_zero_byte(w[u].y) | has_zero_byte(w[u].z) | has_zero_byte(w[u].w);
                if (z) nul_record(a, w[u], v);
                zero |= z;
                U128* dst = reinterpret_cast<U128*>(a.out_v0 + v);
                __builtin_nontemporal_store(r.x, &dst->x); __builtin_nontemporal_store(r.y, &dst->y);
                __builtin_nontemporal_store(r.z, &dst->z); __builtin_nontemporal_store(r.w, &dst->w);
                continue;
            }
            bytemap_vec(a, map, w[u], v, aligned, zero);
        }
    }
    if (zero) atomicOr(a.status, kStNul);
}

// ------------------------------------------------------------------------------------------
template <class G, class Engine>
void launch3(int which, const ScanArgs& a, int64_t n_chunks, hipStream_t s) {
    using C = Carve<G, Engine>;
    allow_big_lds<&k_scan_lp<G, Engine>>();
    allow_big_lds<&k_scan_count<G, Engine>>();
    allow_big_lds<&k_scan_emit<G, Engine>>();
    const dim3 grid((unsigned)n_chunks), block(G::THREADS);
    if (which == 0) hipLaunchKernelGGL((k_scan_lp<G, Engine>), grid, block, C::kBytes, s, a);
    else if (which == 1) hipLaunchKernelGGL((k_scan_count<G, Engine>), grid, block, C::kBytes, s, a);
    else hipLaunchKernelGGL((k_scan_emit<G, Engine>), grid, block, C::kBytes, s, a);
}

}  // namespace

int chunk_bytes(int engine, int mask_bytes) {
    if (engine == kEngineDft) return GeoDft::CHUNK;
    switch (mask_bytes) {
    case 1: return GeoNft8::CHUNK;
    case 2: return GeoNft16::CHUNK;
    case 4: return GeoNft32::CHUNK;
    default: return GeoNft64::CHUNK;
    }
}
int block_threads(int, int) { return 256; }

void launch_tile_kernel(int which, int engine, int mask_bytes, const ScanArgs& a, int64_t n_chunks, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (engine == kEngineDft) { launch3<GeoDft, DftEngine>(which, a, n_chunks, s); return; }
    switch (mask_bytes) {
    case 1: launch3<GeoNft8, NftEngine<uint8_t>>(which, a, n_chunks, s); break;
    case 2: launch3<GeoNft16, NftEngine<uint16_t>>(which, a, n_chunks, s); break;
    case 4: launch3<GeoNft32, NftEngine<uint32_t>>(which, a, n_chunks, s); break;
    default: launch3<GeoNft64, NftEngine<uint64_t>>(which, a, n_chunks, s); break;
    }
}

// ---- positional-window kernel for length-preserving stream tables in window form, wave-tiled I/O (see scan_block.hpp)
constexpr int kWtThreads = 256;
constexpr int kWtWaves = kWtThreads / kWave;
constexpr int kWtEntMax = 32768;          // larger window tables stay in global memory (L1/L2)

// kPair: the pair form of the entries (always in LDS)
template <bool kLdsEnt, bool kWide, bool kPair = false>
__global__ __launch_bounds__(kWtThreads) void k_stream_lpw(ScanArgs a, int64_t lane_bytes, int ent_room) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];    // cls[256] | entries[ent_room] | tiles[waves][in 4 KiB, out 8 KiB]
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    for (int k = threadIdx.x; k < 256; k += kWtThreads) smem[k] = a.blob[h.off_cls + k];
    const uint32_t off_ent = kPair ? h.off_lpw2 : h.off_lpw, ent_bytes = kPair ? h.lpw2_bytes : h.lpw_bytes;
    if (kLdsEnt) {
        const U128* e = reinterpret_cast<const U128*>(a.blob + off_ent);
        U128* d = reinterpret_cast<U128*>(smem + 256);
        for (int k = threadIdx.x; k < (int)(ent_bytes / 16); k += kWtThreads) d[k] = e[k];
    }
    __syncthreads();
    LpwView T;
    T.cls = smem;
    T.ent = kLdsEnt ? reinterpret_cast<const U128*>(smem + 256) : reinterpret_cast<const U128*>(a.blob + off_ent);
    T.delay = h.lpw_delay;
    T.n_cls = h.n_cls;
    const int lid = threadIdx.x & (kWave - 1);
    uint8_t* tin = smem + 256 + ent_room + (threadIdx.x / kWave) * (kWtTile + kWtOutTile);
    uint8_t* tout = tin + kWtTile;
    const int64_t lane = (int64_t)blockIdx.x * kWtThreads + threadIdx.x;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    WtLane<kWide, kPair> L;
    L.init(a, T, h.n_cls, lane, lane_bytes);
    WtMover M;
#pragma unroll
    for (int i = 0; i < 4; ++i) M.set(lid, i, lane_bytes, __shfl(L.rv, WtMover::row_of(lid, i)));
#pragma unroll
    for (int i = 0; i < 8; ++i) M.set_out(lid, i, lane_bytes, __shfl(L.rv, WtMover::out_row_of(lid, i)));
    // the wave's window of the buffers (uniform) and how far a fetch may reach in it
    const int64_t wave_lo = (lane - lid) * lane_bytes;
    const uint8_t* win_in = a.in_v0 + wave_lo;
    uint8_t* win_out = a.out_v0 + wave_lo;
    const int64_t room0 = ((a.vend - 16) & ~(int64_t)15) - wave_lo;
    const WtRow irow{tin + lid * kWtPiece, (uint32_t)((lid >> 1) & 3) << 4};
    const WtOutRow orow{tout + lid * kWtOutRow, (uint32_t)(lid & 7) << 4};
    uint64_t rows1 = 0, rows2 = 0;            // the lanes that walked in the previous iteration / the one before
    if (__ballot(L.active)) {
        const uint32_t t0 = __builtin_amdgcn_readfirstlane(wt_lds_addr(tin));
#pragma unroll
        for (int i = 0; i < 4; ++i) wt_glds16(win_in + M.load_off(i, 0, room0), t0 + i * 1024);
        bool st1 = false;                     // the previous iteration issued all eight tile stores, unconditionally
        for (int32_t k64 = 0;; k64 += kWtPiece) {       // 64 k (a sub-range never walks 2^31 bytes: see rlimit)
            L.check(a, lane);
            const uint64_t rows = __ballot(L.active);
            if (!rows && !rows1 && !rows2) break;
            // Input tile k must have landed.  The memory counter retires in order and the only operations
            // issued after the tile's loads are the previous iteration's stores, which count only when
            // they were certainly issued (an all-lanes-off store may be branched over).
            if (st1) TRRE_WAIT_VM(8);
            else TRRE_WAIT_VM(0);
            const U128 b0 = irow.load(0), b1 = irow.load(1), b2 = irow.load(2), b3 = irow.load(3);
            TRRE_WAIT_LGKM0();                // the rows are in registers: the buffer can take tile k + 1
#pragma unroll
            for (int i = 0; i < 4; ++i) wt_glds16(win_in + M.load_off(i, k64 + kWtPiece, room0), t0 + i * 1024);
            const int md = L.mode(k64);
            if (L.active) L.front(T, md, b0, b1.x, orow, a.out_v0);
            st1 = false;
            if ((k64 & (kWtOutRow - 1)) == 0) {
                // the output rows [rv - 128, rv) are complete: out they go, eight lanes per 128-byte line
                const bool steady = (rows & rows1 & rows2) == ~0ull && k64 >= 2 * kWtOutRow;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const uint8_t* slot = tout + hf * 4096 + lid * 16;
                    const u32x4 ov0 = *reinterpret_cast<const u32x4*>(slot), ov1 = *reinterpret_cast<const u32x4*>(slot + 1024),
                                ov2 = *reinterpret_cast<const u32x4*>(slot + 2048), ov3 = *reinterpret_cast<const u32x4*>(slot + 3072);
                    if (steady) {
                        *reinterpret_cast<u32x4*>(win_out + M.store_off(4 * hf + 0, k64)) = ov0;
                        *reinterpret_cast<u32x4*>(win_out + M.store_off(4 * hf + 1, k64)) = ov1;
                        *reinterpret_cast<u32x4*>(win_out + M.store_off(4 * hf + 2, k64)) = ov2;
                        *reinterpret_cast<u32x4*>(win_out + M.store_off(4 * hf + 3, k64)) = ov3;
                    } else {
                        // (the first pieces and the tail: which blocks are whose is worked out on the spot)
                        if (WtMover::stores(4 * hf + 0, k64, rows, rows1, rows2, lid, __shfl(L.rfs, WtMover::out_row_of(lid, 4 * hf + 0))))
                            *reinterpret_cast<u32x4*>(win_out + M.store_off(4 * hf + 0, k64)) = ov0;
                        if (WtMover::stores(4 * hf + 1, k64, rows, rows1, rows2, lid, __shfl(L.rfs, WtMover::out_row_of(lid, 4 * hf + 1))))
                            *reinterpret_cast<u32x4*>(win_out + M.store_off(4 * hf + 1, k64)) = ov1;
                        if (WtMover::stores(4 * hf + 2, k64, rows, rows1, rows2, lid, __shfl(L.rfs, WtMover::out_row_of(lid, 4 * hf + 2))))
                            *reinterpret_cast<u32x4*>(win_out + M.store_off(4 * hf + 2, k64)) = ov2;
                        if (WtMover::stores(4 * hf + 3, k64, rows, rows1, rows2, lid, __shfl(L.rfs, WtMover::out_row_of(lid, 4 * hf + 3))))
                            *reinterpret_cast<u32x4*>(win_out + M.store_off(4 * hf + 3, k64)) = ov3;
                    }
                }
                st1 = steady;
            }
            if (L.active) L.back(T, md, b1, b2, b3, orow, a.out_v0);
            rows2 = rows1;
            rows1 = rows;
        }
        TRRE_WAIT_VM(0);                      // no load may still be writing LDS when the wave ends
    }
    if (L.seen & kLpwNul) atomicOr(a.status, kStNul);
    if (L.seen & kLpwDiv) atomicOr(a.status, kStDiverge);
}

// second launch of the window path: the few lanes that touch an end of the input, redone by the
// general direct walker (grid-stride over the redo list)
template <bool kLdsEnt>
__global__ __launch_bounds__(kDirectThreads) void k_stream_redo(ScanArgs a, int64_t lane_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamView T = direct_stage<kLdsEnt>(a, smem);
    const uint32_t n_cls = reinterpret_cast<const StreamBlobHeader*>(a.blob)->n_cls;
    uint8_t* ring = smem + 256 + (kLdsEnt ? kDirectTabSmall : kDirectPoolBytes) + threadIdx.x * kRingStride;
    // every listed lane is split into 64-byte sub-lanes (same ownership rule, same positional
    // output) so that the few redone lanes do not serialise a whole sub-range each
    const int64_t sub = lane_bytes / 64;
    const int64_t n = (int64_t)a.redo[0] * sub;
    uint32_t st = 0;
    for (int64_t k = (int64_t)blockIdx.x * kDirectThreads + threadIdx.x; k < n; k += (int64_t)gridDim.x * kDirectThreads) {
        DirectLane L;
        stream_direct_lane<0>(a, T, n_cls, (int64_t)a.redo[1 + k / sub] * sub + k % sub, 64, ring, 0, L, st);
    }
    if (st) atomicOr(a.status, st);
}
void launch_lpw_kernel(int ent_bytes, bool wide, bool direct_ent_in_lds, const ScanArgs& a, int64_t lane_bytes, void* stream, int pair_bytes) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n_lanes = (a.vend + lane_bytes - 1) / lane_bytes;
    const dim3 grid((unsigned)((n_lanes + kWtThreads - 1) / kWtThreads));
    const bool pair = pair_bytes > 0 && pair_bytes <= kWtEntMax && !wide;
    const bool ent_in_lds = pair || ent_bytes <= kWtEntMax;
    const int ent_room = pair ? (pair_bytes + 15) / 16 * 16 : (ent_in_lds ? (ent_bytes + 15) / 16 * 16 : 0);
    const int lds = 256 + ent_room + kWtWaves * (kWtTile + kWtOutTile);
    if (pair) {
        allow_big_lds<&k_stream_lpw<true, false, true>>();
        hipLaunchKernelGGL((k_stream_lpw<true, false, true>), grid, dim3(kWtThreads), lds, s, a, lane_bytes, ent_room);
    } else if (ent_in_lds && wide) {
        allow_big_lds<&k_stream_lpw<true, true>>();
        hipLaunchKernelGGL((k_stream_lpw<true, true>), grid, dim3(kWtThreads), lds, s, a, lane_bytes, ent_room);
    } else if (ent_in_lds) {
        allow_big_lds<&k_stream_lpw<true, false>>();
        hipLaunchKernelGGL((k_stream_lpw<true, false>), grid, dim3(kWtThreads), lds, s, a, lane_bytes, ent_room);
    } else if (wide) {
        hipLaunchKernelGGL((k_stream_lpw<false, true>), grid, dim3(kWtThreads), lds, s, a, lane_bytes, ent_room);
    } else {
        hipLaunchKernelGGL((k_stream_lpw<false, false>), grid, dim3(kWtThreads), lds, s, a, lane_bytes, ent_room);
    }
    const bool ring_lds = reinterpret_cast<const void*>(a.blob) != nullptr && direct_ent_in_lds;
    if (ring_lds) hipLaunchKernelGGL((k_stream_redo<true>), dim3(64), dim3(kDirectThreads), kDirectLds, s, a, lane_bytes);
    else hipLaunchKernelGGL((k_stream_redo<false>), dim3(64), dim3(kDirectThreads), kDirectLdsHot, s, a, lane_bytes);
}
template <int kMode, bool kSym>
void launch_direct_t(bool ent_lds, const ScanArgs& a, int64_t lane_bytes, int64_t n_blocks, hipStream_t s) {
    if (ent_lds) hipLaunchKernelGGL((k_stream_direct<kMode, true, kSym>), dim3((unsigned)n_blocks), dim3(kDirectThreads), kDirectLds, s, a, lane_bytes);
    else hipLaunchKernelGGL((k_stream_direct<kMode, false, kSym>), dim3((unsigned)n_blocks), dim3(kDirectThreads), kDirectLdsHot, s, a, lane_bytes);
}
int direct_ent_lds_bytes() { return kDirectEntBytes; }
int direct_block_threads() { return kDirectThreads; }
template <int kSym, bool kHasSlow>
void launch_g16(int which, const ScanArgs& a, int64_t lane_bytes, int64_t n_blocks, hipStream_t s, int g16_bytes) {
    const int room = (g16_bytes + 15) / 16 * 16;
    const int lds = 256 + room + kDirectPoolSmall + kDirectThreads * kRingStride + 64 + kDirectWsc;
    const int lds_count = 256 + room + 64;
    allow_big_lds<&k_stream_g16<1, kSym, kHasSlow>>();
    allow_big_lds<&k_stream_g16<2, kSym, kHasSlow>>();
    if (which == 1) hipLaunchKernelGGL((k_stream_g16<1, kSym, kHasSlow>), dim3((unsigned)n_blocks), dim3(kDirectThreads), lds_count, s, a, lane_bytes, room);
    else hipLaunchKernelGGL((k_stream_g16<2, kSym, kHasSlow>), dim3((unsigned)n_blocks), dim3(kDirectThreads), lds, s, a, lane_bytes, room);
}
// ONE walk for the general families on small tables (one_block.hpp): short lanes from guessed-and-verified states, the
// workgroup's whole output in LDS, its place in the output by decoupled look-back over the tiles' totals, stores as whole 16-byte lines.
// Workgroups are persistent: each takes tiles (kOneThreads lanes of oa.lane_bytes bytes) from a ticket counter until none is left.
//   smem: cls[256] | g16[g16_room] | pooled text (2 KiB) | regions[kOneThreads x R] + 32 | offs[kOneThreads + 4] | exits[kOneThreads] |
//         misc[16] | marks[kOneThreads x R / 16 + 16]
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    const int lid = threadIdx.x & (kWave - 1);
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, kWave);
        if (lid >= d) v += o;
    }
    return v;
}
// the tile's tail: sizes, look-back, stores (a function of its own: its registers are not the walk's)
// (everything by value: a reference to the kernel's arguments would put a copy of them on the stack — scratch memory per lane, and the
// scratch ring's size is what the dispatcher admits waves by)
struct OneTailArgs {
    uint8_t* out;
    uint64_t cap;
    uint32_t* status;
    uint64_t *desc, *gsum, *ginc, *total;
    int64_t n_tiles;
    uint32_t spin;
};
__device__ __forceinline__ uint32_t one_tail(OneTailArgs q, const uint8_t* regions, uint32_t* offs, uint8_t* marks, uint32_t region, const uint32_t* exits,
                                                       uint32_t* misc, int64_t tile, uint32_t count, uint32_t flags, uint32_t used, uint32_t st_in) {
    const int tid = threadIdx.x;
    const int wave = tid / kWave, lid = tid & (kWave - 1);
    uint32_t st = st_in;
    const bool tile_void = flags & 1u, live = flags & 2u;
    const uint32_t known = (flags >> 2) & 1u;
    const OneTile tv{regions, offs, marks, region};
    struct { uint8_t* out; uint64_t cap; uint32_t* status; } a{q.out, q.cap, q.status};
    struct { uint64_t *desc, *gsum, *ginc, *total; int64_t n_tiles; uint32_t spin; } oa{q.desc, q.gsum, q.ginc, q.total, q.n_tiles, q.spin};
        // ---- sizes: the lanes' places in the tile's output ---------------------------------------------------------------------------
        const uint32_t len = tile_void ? 0u : count;
        const uint32_t incl = wave_incl_scan_u32(len);
        if (lid == kWave - 1) misc[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += misc[w];
        offs[tid] = woff + incl - len;
        const uint32_t total = misc[0] + misc[1] + misc[2] + misc[3];
        if (tid == kOneThreads - 1) offs[kOneThreads] = total;
        // ---- base: look back over the tiles before this one (wave 0) ------------------------------------------------------------------
        // Two levels (the resident workgroups move in step: when a tile looks back, the hundreds of tiles before it have their totals out
        // and none its running total — tile by tile that was a dozen rounds of 64 polls, most of the tile's time): tiles also add their
        // total to their GROUP of 32 (one atomic: count and sum together), and the last tile of a group leaves the group's running total.
        // Lanes 0..31 poll the tiles before this one in its own group, lanes 32..63 the 32 groups before it: one round as a rule.
        if (wave == 0) {
            const uint32_t exit_last = exits[kOneThreads - 1];
            const int64_t grp = tile >> 5;
            const int r = (int)(tile & 31);
            if (lid == 0) {
                __hip_atomic_store(oa.desc + tile, one_desc(tile == 0 ? kOneDescInc : kOneDescAgg, exit_last, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(oa.gsum + grp, (1ull << 58) | (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint64_t base = 0;
            uint32_t lb_st = 0;
            if (tile > 0) {
                uint32_t polls = 0;
                int64_t gtop = grp - 1;                   // the newest group the upper lanes look at
                uint64_t part = 0;                        // what the windows so far have added up
                bool tiles_done = false;
                for (;;) {
                    // lanes 0..31: the tile lid + 1 before this one (lane 0 always: its exit row is checked), while it is in this group
                    const bool t_mine = lid < 32 && (lid < r || lid == 0) && !tiles_done;
                    const uint64_t d = t_mine ? __hip_atomic_load(oa.desc + (tile - 1 - lid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    // lanes 32..63: group gtop - (lid - 32): its count and sum, and its running total if it has one
                    const int64_t gq = gtop - (lid - 32);
                    const bool g_mine = lid >= 32 && gq >= 0;
                    const uint64_t gs = g_mine ? __hip_atomic_load(oa.gsum + gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    const uint64_t gi = g_mine ? __hip_atomic_load(oa.ginc + gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (lid >= 32 ? (1ull << 63) : 0ull);
                    const uint32_t kind = (uint32_t)(d >> 62);
                    bool wait = false;
                    uint64_t v = 0;
                    bool finished = false;
                    if (!tiles_done) {
                        const uint64_t in_grp = __ballot(lid < r);                                  // the tiles of this group before this one
                        const uint64_t zero = __ballot(t_mine && kind == 0u), inc = __ballot(lid < r && kind == 2u) & in_grp;
                        const int jinc = inc ? __builtin_ctzll(inc) : 64;
                        const uint64_t upto = (jinc >= 63 ? ~0ull : (1ull << (jinc + 1)) - 1ull) & (in_grp | 1ull);
                        if (zero & upto) wait = true;
                        else {
                            // the one guess nobody can repair in place: this tile's first lane against the exit of the tile before it
                            const uint32_t pred_exit = __shfl(one_desc_exit(d), 0, kWave);
                            if (lid == 0 && !known && live && used != pred_exit) lb_st |= kStOneVoid;
                            v = (lid < r && ((upto >> lid) & 1ull)) ? (d & kOneValMask) : 0ull;
                            if (jinc < 64) finished = true;
                        }
                    }
                    if (!wait && !finished) {
                        // the groups: down to the first one with a running total, every one above it complete
                        const uint64_t ginc_at = __ballot(lid >= 32 && (gi >> 63)) >> 32;          // bit q: group gtop - q has its running total
                        const uint64_t gfull = __ballot(lid >= 32 && (gs >> 58) == 32u) >> 32;
                        const int qinc = ginc_at ? __builtin_ctzll(ginc_at) : 32;
                        const uint64_t above = qinc >= 32 ? 0xffffffffull : (1ull << qinc) - 1ull;
                        if ((gfull & above) != above) wait = true;
                        else {
                            const int q = lid - 32;
                            if (lid >= 32 && q < qinc) v += gs & ((1ull << 58) - 1ull);
                            if (lid >= 32 && q == qinc) v += gi & ~(1ull << 63);
                            if (qinc < 32) finished = true;
                        }
                    }
                    if (wait) {
                        if (++polls > oa.spin) { lb_st |= kStOneVoid; break; }                     // (cannot be: tickets are taken in order)
                        __builtin_amdgcn_s_sleep(2);
                        continue;
                    }
                    for (int dd = 32; dd; dd >>= 1) v += __shfl_xor(v, dd, kWave);
                    part += v;
                    if (finished) break;
                    tiles_done = true;                     // 32 complete groups and no running total among them: the 32 before those
                    gtop -= 32;
                }
                base = part;
            }
            if (lid == 0) {
                if (tile > 0) __hip_atomic_store(oa.desc + tile, one_desc(kOneDescInc, exit_last, base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r == 31) __hip_atomic_store(oa.ginc + grp, (1ull << 63) | (base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *reinterpret_cast<uint64_t*>(misc + 6) = base;
                if (tile == oa.n_tiles - 1) *oa.total = base + total;
                misc[5] = (lb_st | st | __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & kStOneVoid;
            }
            st |= lb_st;
        }
        __syncthreads();
        const uint64_t base = *reinterpret_cast<const uint64_t*>(misc + 6);
        // ---- store: a thread per 16-byte line of the output, two at a time -----------------------------------------------------------
        const uint32_t hh = (uint32_t)((reinterpret_cast<uintptr_t>(a.out) + base) & 15u);
        one_mark(tv, tid, hh);
        __syncthreads();
        const bool write = base + total <= a.cap;
        if (!write) st |= kStCapacity;
        const uint32_t n_lines = (hh + total + 15u) >> 4;
        for (uint32_t c0 = 0; c0 < n_lines; c0 += 2 * kOneThreads) {
            const uint32_t cc[2] = {c0 + (uint32_t)tid, c0 + kOneThreads + (uint32_t)tid};
            one_store_lines<2>(tv, a.out, base, hh, cc, total, write);
        }
    return st;
}
template <int kSym, bool kHasSlow>
__global__ __launch_bounds__(kOneThreads, 3) void k_stream_one(ScanArgs a, OneArgs oa, int g16_room) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    const int tid = threadIdx.x;
    for (int k = tid; k < 256; k += kOneThreads) smem[k] = a.blob[h.off_cls + k];
    {
        const U128* e = reinterpret_cast<const U128*>(a.blob + h.off_g16);
        U128* d = reinterpret_cast<U128*>(smem + 256);
        for (int k = tid; k < (int)(h.g16_bytes / 16); k += kOneThreads) d[k] = e[k];
    }
    StreamView T;
    uint8_t* pool_lds = smem + 256 + g16_room;
    if ((int)h.pool_bytes <= kDirectPoolSmall) {
        const uint32_t* e = reinterpret_cast<const uint32_t*>(a.blob + h.off_pool);
        uint32_t* d = reinterpret_cast<uint32_t*>(pool_lds);
        for (int k = tid; k < (int)(h.pool_bytes / 4); k += kOneThreads) d[k] = e[k];
        T.pool_fast = pool_lds;
    }
    T.cls = smem;
    T.g16 = smem + 256;
    // (the pair form of the table is not walked here: measured, no difference — at three waves per SIMD the walk runs at the pace of its chain of
    // dependent table reads, not of its instruction count)
    T.ent = reinterpret_cast<const uint64_t*>(a.blob + h.off_ent);
    T.pool = a.blob + h.off_pool;
    T.long_pool = h.max_out >= 255u;
    const uint32_t R = oa.region;
    uint8_t* regions = pool_lds + kDirectPoolSmall;
    uint8_t* my = regions + (size_t)tid * R;
    uint32_t* offs = reinterpret_cast<uint32_t*>(regions + (((size_t)kOneThreads * R + 32 + 15) & ~(size_t)15));
    uint32_t* exits = offs + kOneThreads + 4;
    uint32_t* misc = exits + kOneThreads;                 // [0..3] the waves' totals, [4] the tile, [5] void seen, [6..7] the tile's base
    uint8_t* marks = reinterpret_cast<uint8_t*>(misc + 16);
    // (the backward pass of a guided family guessed wrong somewhere: its symbols are not final, finish() repairs them and runs the pair)
    if (kSym != 0 && __hip_atomic_load(a.status + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        if (tid == 0 && blockIdx.x == 0) atomicOr(a.status, kStOneVoid);
        return;
    }
    const int lid = tid & (kWave - 1);
    uint32_t st_all = 0;
    const bool prof = oa.prof != nullptr && tid == 0;
    auto stamp = [&](uint64_t& t, int slot) {
        if (prof) {
            const uint64_t now = clock64();
            atomicAdd(reinterpret_cast<unsigned long long*>(oa.prof + slot), (unsigned long long)(now - t));
            t = now;
        }
    };
    // A tile is taken when its walk begins, never earlier: a ticket held while the tile before it is still being worked on is a tile that
    // every later tile's look-back waits for (asking ahead was tried: the look-backs took a whole tile's time).
    if (tid == 0) {
        misc[4] = atomicAdd(oa.ticket, 1u);
        misc[5] = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStOneVoid;
    }
    __syncthreads();
    int64_t tile = (int64_t)misc[4];
    uint32_t void_seen = misc[5];
    for (;;) {
        if (tile >= oa.n_tiles) break;
        uint64_t tclk = prof ? clock64() : 0;
        if (void_seen) {
            // the launch is void already: nothing to walk, but whoever looks back at this tile must not wait for it
            if (tid == 0) {
                __hip_atomic_store(oa.desc + tile, one_desc(kOneDescInc, 0u, 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(oa.gsum + (tile >> 5), 1ull << 58, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((tile & 31) == 31) __hip_atomic_store(oa.ginc + (tile >> 5), 1ull << 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                misc[4] = atomicAdd(oa.ticket, 1u);
            }
            __syncthreads();
            tile = (int64_t)misc[4];
            __syncthreads();
            continue;
        }
        const int64_t lane = tile * kOneThreads + tid;
        const bool live = lane * (int64_t)oa.lane_bytes < a.vend;
        // ---- walk, verify against the lane before, walk again where the guess was wrong --------------------------------------------
        DirectLane L;
        uint32_t st = 0, entry = kOneGuess, used = 0, known = 0;
        bool need = true, gave_up = false;
        for (int round = 0;; ++round) {
            if (need) {
                L.entry = entry;
                st = 0;
                g16_lane<3, kSym, kHasSlow>(a, T, h.n_cls, lane, (int64_t)oa.lane_bytes, my, (uint64_t)R, L, st);
                exits[tid] = L.exit;
                used = L.entry;
                if (round == 0) { known = L.known; stamp(tclk, 1); }
            }
            __syncthreads();
            need = tid > 0 && live && !known && used != exits[tid - 1];
            if (need) entry = exits[tid - 1];
            if (!__syncthreads_or(need ? 1 : 0)) break;
            if (round + 1 >= kOneRounds) { gave_up = true; break; }
        }
        if (gave_up) st |= kStOneVoid;
        // (a tile that cannot answer — a region outgrown, no agreement on the states — keeps the protocol going with an empty output: the
        // launch is void, and sizes that are not backed by bytes in the regions must not reach the gather)
        const bool tile_void = __syncthreads_or((st & kStOneVoid) ? 1 : 0) != 0;
        stamp(tclk, 2);
        st = one_tail(OneTailArgs{a.out, a.cap, a.status, oa.desc, oa.gsum, oa.ginc, oa.total, oa.n_tiles, oa.spin}, regions, offs, marks, R, exits, misc, tile,
                      (uint32_t)L.count, (tile_void ? 1u : 0u) | (live ? 2u : 0u) | (known ? 4u : 0u), used, st);
        void_seen = misc[5];
        stamp(tclk, 5);
        if (prof) atomicAdd(reinterpret_cast<unsigned long long*>(oa.prof + 7), 1ull);
        st_all |= st;
        if (tid == 0) misc[4] = atomicAdd(oa.ticket, 1u);  // the next tile, now that this one is out
        __syncthreads();                                   // (... and every thread is done with the regions, the offsets and the marks)
        tile = (int64_t)misc[4];
        stamp(tclk, 0);
    }
    st_all = wave_or(st_all);
    if (st_all && lid == 0) atomicOr(a.status, st_all);
}
template <int kSym, bool kHasSlow>
int launch_one_t(const ScanArgs& a, const OneArgs& oa, hipStream_t s, int g16_bytes) {
    const int room = (g16_bytes + 15) / 16 * 16;
    const int lds = 256 + room + kDirectPoolSmall + ((kOneThreads * (int)oa.region + 32 + 15) & ~15) + (kOneThreads + 4 + kOneThreads + 16) * 4 +
                    kOneThreads * (int)oa.region / 16 + 32;
    // (the kernel has 256 bytes of static LDS — __syncthreads_or's — so the whole 160 KiB cannot be asked for as dynamic: ask for what it needs)
    if (lds > kLdsLimit - 1024) return -1;
    static std::atomic<int> allowed{0};
    if (allowed.load() < lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stream_one<kSym, kHasSlow>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        allowed.store(lds);
    }
    // persistent workgroups: about as many as the device holds at once — 3 waves per SIMD by the kernel's registers (launch bounds), and what
    // the LDS admits.  (More would only queue for a CU and find the tickets gone; tiles are handed out by the ticket counter, not by block.)
    static std::atomic<int> cus_cache{0};
    if (!cus_cache.load()) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t prop;
        cus_cache.store(hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    }
    const int cus = cus_cache.load();
    int per_cu = kLdsLimit / lds;
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)cus * per_cu;
    if (blocks > oa.n_tiles) blocks = oa.n_tiles;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_stream_one<kSym, kHasSlow>), dim3((unsigned)blocks), dim3(kOneThreads), lds, s, a, oa, room);
    return 0;
}
int launch_one(const ScanArgs& a, const OneArgs& oa, void* stream, int g16_bytes, int sym, bool g16_slow) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sym == 2) return g16_slow ? launch_one_t<2, true>(a, oa, s, g16_bytes) : launch_one_t<2, false>(a, oa, s, g16_bytes);
    if (sym == 1) return g16_slow ? launch_one_t<1, true>(a, oa, s, g16_bytes) : launch_one_t<1, false>(a, oa, s, g16_bytes);
    return g16_slow ? launch_one_t<0, true>(a, oa, s, g16_bytes) : launch_one_t<0, false>(a, oa, s, g16_bytes);
}
template <int kSym>
void launch_direct_sym(int which, bool ent_in_lds, const ScanArgs& a, int64_t lane_bytes, int64_t n_blocks, hipStream_t s, int g16_bytes, bool g16_slow) {
    if (g16_bytes > 0 && which != 0) {
        if (g16_slow) launch_g16<kSym, true>(which, a, lane_bytes, n_blocks, s, g16_bytes);
        else launch_g16<kSym, false>(which, a, lane_bytes, n_blocks, s, g16_bytes);
        return;
    }
    constexpr bool kS = kSym != 0;       // (the 8-byte-entry walkers know one symbol per byte only: the runtime packs symbols for 16-byte tables)
    if (which == 0) launch_direct_t<0, kS>(ent_in_lds, a, lane_bytes, n_blocks, s);
    else if (which == 1) launch_direct_t<1, kS>(ent_in_lds, a, lane_bytes, n_blocks, s);
    else launch_direct_t<2, kS>(ent_in_lds, a, lane_bytes, n_blocks, s);
}
void launch_direct_kernel(int which, bool ent_in_lds, const ScanArgs& a, int64_t lane_bytes, int64_t n_blocks, void* stream, int g16_bytes, int sym,
                          bool g16_slow) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sym == 2 && g16_bytes > 0 && which != 0) launch_direct_sym<2>(which, ent_in_lds, a, lane_bytes, n_blocks, s, g16_bytes, g16_slow);
    else if (sym) launch_direct_sym<1>(which, ent_in_lds, a, lane_bytes, n_blocks, s, g16_bytes, g16_slow);
    else launch_direct_sym<0>(which, ent_in_lds, a, lane_bytes, n_blocks, s, g16_bytes, g16_slow);
}

// ---- backward pass of the guided families: one symbol per input byte (rev_sweep_lane) -------------------
constexpr int kRevThreads = 256;

template <int kDbg, bool kNib>
__global__ __launch_bounds__(kRevThreads) void k_rev_sweep(ScanArgs a, int64_t lane_bytes, int tile_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];    // tab[n_rev][256]: at most 64 KiB
    const RevBlobHeader& h = *reinterpret_cast<const RevBlobHeader*>(a.rblob);
    const U128* e = reinterpret_cast<const U128*>(a.rblob + h.off_wide);
    U128* d = reinterpret_cast<U128*>(smem);
    for (int k = threadIdx.x; k < (int)h.n_rev * 16; k += kRevThreads) d[k] = e[k];
    __syncthreads();
    const RevView T{smem};
    // (packed symbols: a 4 KiB tile per wave behind the table, for the unit stores of interior waves)
    uint8_t* tile = (kNib && tile_bytes) ? smem + (((int)h.n_rev * 256 + 15) & ~15) + (threadIdx.x / kWave) * 4096 : nullptr;
    rev_sweep_lane<kDbg, kNib>(a, T, (int64_t)blockIdx.x * kRevThreads + threadIdx.x, lane_bytes, tile);
}
// exact sub-ranges, the backward pass (scan_block.hpp: rev_guess_wrong, rev_repair_lane)
template <bool kNib>
__global__ __launch_bounds__(256) void k_rev_verify(ScanArgs a, int64_t lane_bytes, int64_t n_lanes) {
    const int64_t lane = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (lane < n_lanes) {
        bad = lane + 1 < n_lanes && rev_guess_wrong<kNib>(a, lane, lane_bytes);
        a.rev_flags[lane] = bad ? 1u : 0u;
    }
    const uint32_t n = (uint32_t)__popcll(__ballot(bad));
    if (n && (threadIdx.x & (kWave - 1)) == 0) atomicAdd(a.status + 2, n);
}
template <bool kNib>
__global__ __launch_bounds__(256) void k_rev_repair(ScanArgs a, int64_t lane_bytes, int64_t n_lanes) {
    const int64_t lane = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const RevBlobHeader& h = *reinterpret_cast<const RevBlobHeader*>(a.rblob);
    const RevView T{a.rblob + h.off_wide};
    if (lane < n_lanes && a.rev_flags[lane]) rev_repair_lane<kNib>(a, T, lane, lane_bytes);
}
void launch_rev_verify(const ScanArgs& a, int64_t lane_bytes, void* stream, bool packed) {
    const int64_t vtop = packed ? (a.vend + 127) & ~(int64_t)127 : (a.vend + 63) & ~(int64_t)63;
    const int64_t n_lanes = (vtop + lane_bytes - 1) / lane_bytes;
    const dim3 grid((unsigned)((n_lanes + 255) / 256));
    if (packed) hipLaunchKernelGGL(k_rev_verify<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, lane_bytes, n_lanes);
    else hipLaunchKernelGGL(k_rev_verify<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, lane_bytes, n_lanes);
}
void launch_rev_repair(const ScanArgs& a, int64_t lane_bytes, void* stream, bool packed) {
    const int64_t vtop = packed ? (a.vend + 127) & ~(int64_t)127 : (a.vend + 63) & ~(int64_t)63;
    const int64_t n_lanes = (vtop + lane_bytes - 1) / lane_bytes;
    const dim3 grid((unsigned)((n_lanes + 255) / 256));
    if (packed) hipLaunchKernelGGL(k_rev_repair<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, lane_bytes, n_lanes);
    else hipLaunchKernelGGL(k_rev_repair<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, lane_bytes, n_lanes);
}
void launch_rev_sweep(const ScanArgs& a, int tab_bytes, int64_t lane_bytes, void* stream, bool packed) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t vtop = packed ? (a.vend + 127) & ~(int64_t)127 : (a.vend + 63) & ~(int64_t)63;
    const int64_t n_lanes = (vtop + lane_bytes - 1) / lane_bytes;
    const dim3 grid((unsigned)((n_lanes + kRevThreads - 1) / kRevThreads));
    allow_big_lds<&k_rev_sweep<0, false>>();
    allow_big_lds<&k_rev_sweep<0, true>>();
    const int tiles = (kRevThreads / kWave) * 4096;      // (packed symbols: interior waves store through a 4 KiB tile each)
    if (packed) hipLaunchKernelGGL((k_rev_sweep<0, true>), grid, dim3(kRevThreads), ((tab_bytes + 15) & ~15) + tiles, s, a, lane_bytes, tiles);
    else hipLaunchKernelGGL((k_rev_sweep<0, false>), grid, dim3(kRevThreads), tab_bytes, s, a, lane_bytes, 0);
}

// ---- wide guided tables (16-bit symbols, tables through L1 / L2): scan_block.hpp ------------------------------
__global__ __launch_bounds__(kRevThreads) void k_rev_wide(ScanArgs a, int64_t lane_bytes) {
    const RevBlobHeader& h = *reinterpret_cast<const RevBlobHeader*>(a.rblob);
    const RevWideView T{reinterpret_cast<const uint16_t*>(a.rblob + h.off_wide)};
    rev_wide_lane(a, T, (int64_t)blockIdx.x * kRevThreads + threadIdx.x, lane_bytes);
}
template <int kMode>
__global__ __launch_bounds__(kDirectThreads) void k_wide_fwd(ScanArgs a, int64_t lane_bytes) {
    __shared__ uint64_t part[kDirectThreads / kWave];
    __shared__ uint32_t wpart[kDirectThreads / kWave];
    const StreamBlobHeader& h = *reinterpret_cast<const StreamBlobHeader*>(a.blob);
    StreamView T;
    T.cls = a.blob + h.off_cls;
    T.ent = reinterpret_cast<const uint64_t*>(a.blob + h.off_ent);
    T.pool = a.blob + h.off_pool;
    T.long_pool = h.max_out >= 255u;
    const int64_t lane = (int64_t)blockIdx.x * kDirectThreads + threadIdx.x;
    DirectLane L;
    uint32_t st = 0;
    uint64_t base = 0;
    if (kMode == 2) {
        const uint32_t mine = a.lane_counts[lane];
        const uint32_t incl = wave_scan_incl(mine);
        if ((threadIdx.x & (kWave - 1)) == kWave - 1) wpart[threadIdx.x / kWave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < (int)threadIdx.x / kWave; ++w) wbase += wpart[w];
        base = a.chunk_base[blockIdx.x] + wbase + incl - mine;
        if (a.chunk_base[blockIdx.x] + a.chunk_total[blockIdx.x] > a.cap) {
            if (threadIdx.x == 0) atomicOr(a.status, kStCapacity);
            return;
        }
    }
    wide_fwd_lane<kMode>(a, T, h.n_cls, lane, lane_bytes, base, L, st);
    if (kMode == 1 && (st & kStDiverge)) atomicMax(a.status + 1, 0xffffffffu - (uint32_t)lane);     // (see k_stream_direct)
    if (kMode == 1) {
        if (L.count > 0xffffffffull) { st |= kStCapacity; L.count = 0xffffffffull; }
        a.lane_counts[lane] = (uint32_t)L.count;
        const uint64_t wsum = wave_sum(L.count);
        if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t t = 0;
            for (int w = 0; w < kDirectThreads / kWave; ++w) t += part[w];
            a.chunk_total[blockIdx.x] = t;
        }
    }
    st = wave_or(st);
    if (st && (threadIdx.x & (kWave - 1)) == 0) atomicOr(a.status, st);
}
void launch_rev_wide(const ScanArgs& a, int64_t lane_bytes, void* stream) {
    const int64_t vtop = (a.vend + 63) & ~(int64_t)63;
    const int64_t n_lanes = (vtop + lane_bytes - 1) / lane_bytes;
    hipLaunchKernelGGL(k_rev_wide, dim3((unsigned)((n_lanes + kRevThreads - 1) / kRevThreads)), dim3(kRevThreads), 0, static_cast<hipStream_t>(stream), a, lane_bytes);
}
void launch_wide_fwd(int which, const ScanArgs& a, int64_t lane_bytes, int64_t n_blocks, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (which == 1) hipLaunchKernelGGL((k_wide_fwd<1>), dim3((unsigned)n_blocks), dim3(kDirectThreads), 0, s, a, lane_bytes);
    else hipLaunchKernelGGL((k_wide_fwd<2>), dim3((unsigned)n_blocks), dim3(kDirectThreads), 0, s, a, lane_bytes);
}

void launch_fb_mark(const ScanArgs& a, const FbCopyArgs& ca, const void* hdr, int64_t lane_bytes, int64_t n_chunks, void* stream) {
    const StreamBlobHeader& h = *static_cast<const StreamBlobHeader*>(hdr);
    constexpr int kG = kFbMarkThreads / kDirectThreads;
    // the mark form of the comb where the tables have it (TRRE_NO_FB_MARK4=1: the 8-byte comb, for A/B runs)
    static const bool no_mark4 = getenv("TRRE_NO_FB_MARK4") != nullptr;
    if (h.fb4_slots && !no_mark4) {
        const int lds4 = 256 + (int)((h.fb4_slots * 4u + 15u) & ~15u) + (int)h.fb4_dense * 128 + (int)((h.fb_lits * 2u + 15u) & ~15u) +
                         kFbMarkThreads * kMarkStageStride * 4 + 64 * kG;
        allow_big_lds<&k_fb_mark4>();
        hipLaunchKernelGGL(k_fb_mark4, dim3((unsigned)((n_chunks + kG - 1) / kG)), dim3(kFbMarkThreads), lds4, static_cast<hipStream_t>(stream), a, ca, lane_bytes, n_chunks);
        return;
    }
    const int lds = 256 + (int)((h.fb_slots * 8u + 15u) & ~15u) + (int)((h.fb_lits * 2u + 15u) & ~15u) + kFbMarkThreads * kMarkStageStride * 4 + 64 * kG;
    allow_big_lds<&k_fb_mark>();
    hipLaunchKernelGGL(k_fb_mark, dim3((unsigned)((n_chunks + kG - 1) / kG)), dim3(kFbMarkThreads), lds, static_cast<hipStream_t>(stream), a, ca, lane_bytes, n_chunks);
}
// workgroup size of the copy pass: the largest whose rings fit next to the literals
int fb_splice_lds(const StreamBlobHeader& h, int threads, bool lit_lds) {
    return (lit_lds ? (int)h.fb_lits * 16 : 0) + threads * 8 + 64 * (threads / kDirectThreads) + (threads / kWave) * (int)kSpLdsPerWave;
}
void launch_fb_splice(const ScanArgs& a, const FbCopyArgs& ca, const void* hdr, int64_t lane_bytes, int64_t n_chunks, void* stream) {
    const StreamBlobHeader& h = *static_cast<const StreamBlobHeader*>(hdr);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // workgroups of 512: eight waves share the literals and the sub-range bases of two chunks; two of them per CU.  The literals
    // in LDS — or, for tables whose literals do not fit, read from memory (for every table that made room for a third workgroup per CU with
    // windows of 2 176 bytes and bought nothing: 2.04 against 1.94 ms per GiB for the whole scan, round 4)
    if (fb_splice_lds(h, 512, true) <= kLdsLimit) {
        allow_big_lds<&k_fb_splice<512, true>>();
        hipLaunchKernelGGL((k_fb_splice<512, true>), dim3((unsigned)((n_chunks + 1) / 2)), dim3(512), fb_splice_lds(h, 512, true), s, a, ca, lane_bytes, n_chunks);
    } else {
        allow_big_lds<&k_fb_splice<512, false>>();
        hipLaunchKernelGGL((k_fb_splice<512, false>), dim3((unsigned)((n_chunks + 1) / 2)), dim3(512), fb_splice_lds(h, 512, false), s, a, ca, lane_bytes, n_chunks);
    }
}
bool fb_splice_fits(const void* hdr) { return fb_splice_lds(*static_cast<const StreamBlobHeader*>(hdr), 512, false) <= kLdsLimit; }
void launch_gen(int which, const ScanArgs& a, const GenArgs& ga, int64_t lane_bytes, int64_t n_chunks, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (which == 1) hipLaunchKernelGGL(k_gen<1>, dim3((unsigned)n_chunks), dim3(kGenThreads), 0, s, a, ga, lane_bytes);
    else hipLaunchKernelGGL(k_gen<2>, dim3((unsigned)n_chunks), dim3(kGenThreads), 0, s, a, ga, lane_bytes);
}
void launch_lazy(int which, const ScanArgs& a, const LazyArgs& la, int64_t lane_bytes, int64_t n_chunks, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the table's first rows in LDS: all of them when they fit in 32 KiB (4 workgroups per CU), else as many as do
    const int row_bytes = (int)la.n_cls * 8;
    const int rows_l = (int)std::min<int64_t>((int64_t)la.n_rows, (32 * 1024) / row_bytes);
    const int lds = 256 + rows_l * row_bytes;
    if (which == 1) hipLaunchKernelGGL(k_lazy<1>, dim3((unsigned)n_chunks), dim3(kGenThreads), lds, s, a, la, lane_bytes, rows_l);
    else hipLaunchKernelGGL(k_lazy<2>, dim3((unsigned)n_chunks), dim3(kGenThreads), lds, s, a, la, lane_bytes, rows_l);
}
void launch_spec_verify(const ScanArgs& a, int64_t n_lanes, void* stream) {
    hipLaunchKernelGGL(k_spec_verify, dim3((unsigned)((n_lanes + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a, n_lanes);
}
void launch_line_probe(const ScanArgs& a, int64_t window, uint32_t* out, void* stream) {
    hipLaunchKernelGGL(k_line_probe, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), a, window, out);      // 4 096 waves
}
void launch_guard_probe(const ScanArgs& a, int64_t window, int64_t n_windows, uint64_t* flags, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_guard_probe, dim3((unsigned)((n_windows + 255) / 256)), dim3(256), 0, s, a, window, n_windows, flags);
}
void launch_guard(bool out, const ScanArgs& a, const GuardArgs& ga, int64_t n_runs, int64_t slots, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((std::min<int64_t>(n_runs, slots) + 63) / 64);
    if (out) hipLaunchKernelGGL(k_guard<true>, dim3(grid), dim3(64), 0, s, a, ga, n_runs);
    else hipLaunchKernelGGL(k_guard<false>, dim3(grid), dim3(64), 0, s, a, ga, n_runs);
}
void launch_bt(int which, const ScanArgs& a, const GenArgs& ga, int64_t lane_bytes, int64_t n_chunks, int64_t pool_blocks, uint32_t budget, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)(n_chunks < pool_blocks ? n_chunks : pool_blocks);
    if (which == 1) hipLaunchKernelGGL(k_bt<1>, dim3(grid), dim3(kGenThreads), 0, s, a, ga, lane_bytes, n_chunks, budget);
    else hipLaunchKernelGGL(k_bt<2>, dim3(grid), dim3(kGenThreads), 0, s, a, ga, lane_bytes, n_chunks, budget);
}
// the copy form's LDS fits
bool fb_copy_fits(const void* hdr) {
    const StreamBlobHeader& h = *static_cast<const StreamBlobHeader*>(hdr);
    if (!h.off_fb_lit_meta) return false;
    const int mark = 256 + (int)((h.fb_slots * 8u + 15u) & ~15u) + (int)((h.fb_lits * 2u + 15u) & ~15u) + kFbMarkThreads * kMarkStageStride * 4 + 256;
    return mark <= kLdsLimit;
}

constexpr int kFbCountThreads = 1024, kFbEmitThreads = 512;
int fb_lds_bytes(const StreamBlobHeader& h, int which) {
    const int tables = 256 + (int)((h.fb_slots * 8u + 15u) & ~15u);
    if (which == 1) return tables + 64 * (kFbCountThreads / kDirectThreads);
    return tables + (int)((h.fb_lits * 8u + 15u) & ~15u) + kFbEmitThreads * kBRingStride + 64 * (kFbEmitThreads / kDirectThreads) + (kFbEmitThreads / kWave) * kWaveScratchBytes;
}
// which: 1 count, 2 emit; `hdr`: the host's copy of the blob header (table sizes)
void launch_fb_kernel(int which, const ScanArgs& a, const void* hdr, int64_t lane_bytes, int64_t n_chunks, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int lds = fb_lds_bytes(*static_cast<const StreamBlobHeader*>(hdr), which);
    if (which == 1) {
        constexpr int kG = kFbCountThreads / kDirectThreads;
        allow_big_lds<&k_stream_fb<1, kFbCountThreads>>();
        hipLaunchKernelGGL((k_stream_fb<1, kFbCountThreads>), dim3((unsigned)((n_chunks + kG - 1) / kG)), dim3(kFbCountThreads), lds, s, a, lane_bytes, n_chunks);
    } else {
        constexpr int kG = kFbEmitThreads / kDirectThreads;
        allow_big_lds<&k_stream_fb<2, kFbEmitThreads>>();
        hipLaunchKernelGGL((k_stream_fb<2, kFbEmitThreads>), dim3((unsigned)((n_chunks + kG - 1) / kG)), dim3(kFbEmitThreads), lds, s, a, lane_bytes, n_chunks);
    }
}
// the fallback form fits next to the emit pass's rings
bool fb_fits(const void* hdr) { return fb_lds_bytes(*static_cast<const StreamBlobHeader*>(hdr), 2) <= 160 * 1024; }

void launch_chunk_scan(const uint64_t* total, uint64_t* base, int64_t n_chunks, void* stream) {
    hipLaunchKernelGGL(k_chunk_scan, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), total, base, n_chunks);
}

// ---- repair of a byte map's output after NUL bytes (runtime.cpp, repair_bytemap_nuls) ----------------------------------
// where the line of each listed NUL ends: the first '\n' at or behind it, or the last byte of the input (a terminator, Q1)
__global__ void k_nul_eol(const uint8_t* in, int64_t n, const uint64_t* pos, uint64_t* eol, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int64_t j = (int64_t)pos[i];
    const int64_t limit = j + (4 << 20);                   // (one thread per NUL: a line of many megabytes is not for this path)
    while (j < n - 1 && j < limit && in[j] != (uint8_t)'\n') ++j;
    eol[i] = (j < n - 1 && in[j] != (uint8_t)'\n') ? ~0ull : (uint64_t)j;
}
// dst[0, len) = map[src[0, len)], then (nl) a '\n' at dst[len]: any alignment of either side (aligned 16-byte stores, the
// loads as they fall)
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__global__ __launch_bounds__(kMapThreads) void k_bytemap_shift(const uint8_t* blob, const uint8_t* src, uint8_t* dst, int64_t len, int nl) {
    __shared__ uint8_t map[256];
    const DftBlobHeader& h = *reinterpret_cast<const DftBlobHeader*>(blob);
    map[threadIdx.x] = blob[h.off_bytemap + threadIdx.x];
    __syncthreads();
    int64_t head = (int64_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u);
    if (head > len) head = len;
    const int64_t nvec = (len - head) >> 4;
    const int64_t gid = (int64_t)blockIdx.x * kMapThreads + threadIdx.x, stride = (int64_t)gridDim.x * kMapThreads;
    for (int64_t k = gid; k < nvec; k += stride) {
        const u32_unaligned* s4 = reinterpret_cast<const u32_unaligned*>(src + head + 16 * k);
        U128 r;
        r.x = map4(map, s4[0]); r.y = map4(map, s4[1]); r.z = map4(map, s4[2]); r.w = map4(map, s4[3]);
        *reinterpret_cast<U128*>(dst + head + 16 * k) = r;
    }
    if (gid < head) dst[gid] = map[src[gid]];
    const int64_t tail0 = head + 16 * nvec;
    if (gid < len - tail0) dst[tail0 + gid] = map[src[tail0 + gid]];
    if (nl && gid == 0) dst[len] = (uint8_t)'\n';
}
void launch_nul_eol(const uint8_t* in, int64_t n, const uint64_t* pos, uint64_t* eol, uint32_t count, void* stream) {
    hipLaunchKernelGGL(k_nul_eol, dim3((count + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), in, n, pos, eol, count);
}
void launch_bytemap_shift(const uint8_t* blob, const uint8_t* src, uint8_t* dst, int64_t len, bool nl, void* stream) {
    int64_t blocks = (len / 16 + kMapThreads - 1) / kMapThreads;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_bytemap_shift, dim3((unsigned)blocks), dim3(kMapThreads), 0, static_cast<hipStream_t>(stream), blob, src, dst, len, nl ? 1 : 0);
}

void launch_bytemap(const ScanArgs& a, void* stream) {
    const int64_t vfirst = a.vbeg & ~(int64_t)15;
    const int64_t nvec = (a.vend - vfirst + 15) / 16;
    // Four 16-byte vectors per lane, non-temporal stores, one pass over the grid.  (Rounds 1-4 kept knobs for the alternatives; measured on
    // 1 GiB, ms per launch: this form 0.361; plain stores with 16 workgroups per CU 0.399; unroll 8 and unroll 2 slower in every combination.)
    constexpr int kUnroll = 4;
    const int64_t per_block = (int64_t)kMapThreads * kUnroll;
    int64_t blocks = (nvec + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_bytemap<kUnroll, true>), dim3((unsigned)blocks), dim3(kMapThreads), 0, static_cast<hipStream_t>(stream), a, nvec);
}

}  // namespace trre
