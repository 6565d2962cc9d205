// launch.hpp — host-callable launchers implemented in scan_kernels.hip.
#pragma once
#include <cstdint>

namespace trre {

struct ScanArgs;
struct FbCopyArgs;
struct GenArgs;
struct GuardArgs;
struct LazyArgs;
struct OneArgs;
struct MapGenArgs;

constexpr int kEngineNft = 0, kEngineDft = 1;

// bytes of input owned by one workgroup / threads per workgroup for an engine
int chunk_bytes(int engine, int mask_bytes);
int block_threads(int engine, int mask_bytes);

// which: 0 length-preserving scan, 1 count pass, 2 emit pass
void launch_tile_kernel(int which, int engine, int mask_bytes, const ScanArgs& a, int64_t n_chunks, void* stream);
// stream families (tables from stream_build.cpp): the direct kernels — a lane walks a long sub-range straight from memory; which: 1 count, 2 emit
// (0: the in-place ring walker, kept for the window kernel's redo lanes only)
int direct_ent_lds_bytes();
int direct_block_threads();
// sym: guided families — columns are the symbols of the backward pass (a.sym_v0): 1 one per byte, 2 packed two per byte
// (16-byte entries only)
void launch_direct_kernel(int which, bool ent_in_lds, const ScanArgs& a, int64_t lane_bytes, int64_t n_blocks, void* stream, int g16_bytes = 0,
                          int sym = 0, bool g16_slow = true);
// the one-pass form of the general families on small tables (one_block.hpp: one walk, the workgroup's output in LDS, look-back for its place);
// oa.desc / oa.ticket zeroed by the caller; -1: the tables and regions do not fit the LDS
int launch_one(const ScanArgs& a, const OneArgs& oa, void* stream, int g16_bytes, int sym, bool g16_slow);
// memoryless programs of any output length in one pass (map_block.hpp); a.blob: the stream blob with its mg table; oa's arrays zeroed by the caller;
// -1: the window does not fit the LDS
int launch_mapgen(const ScanArgs& a, const MapGenArgs& oa, void* stream);
// backward pass of the guided families: fills a.sym_v0 for positions [0 .. round_up(a.vend, 64)) (packed: round_up(.., 128), two per byte)
void launch_rev_sweep(const ScanArgs& a, int tab_bytes, int64_t lane_bytes, void* stream, bool packed);
// wide guided tables (more than 256 backward states: 16-bit symbols at a.sym_v0, both tables through L1 / L2); which: 1 count, 2 emit
void launch_rev_wide(const ScanArgs& a, int64_t lane_bytes, void* stream);
void launch_wide_fwd(int which, const ScanArgs& a, int64_t lane_bytes, int64_t n_blocks, void* stream);
void launch_lpw_kernel(int ent_bytes, bool wide, bool direct_ent_in_lds, const ScanArgs& a, int64_t lane_bytes, void* stream, int pair_bytes = 0);
// count / emit passes over the fallback form of a large table (StreamTables::fb_*); hdr: the host's copy of the stream
// blob's header.  Chunks are those of the direct kernels (direct_block_threads() lanes each).
void launch_fb_kernel(int which, const ScanArgs& a, const void* hdr, int64_t lane_bytes, int64_t n_chunks, void* stream);
bool fb_fits(const void* hdr);
// the copy form of a large table (scan_block.hpp): mark pass (count + events); the second pass is the splice below
void launch_fb_mark(const ScanArgs& a, const FbCopyArgs& ca, const void* hdr, int64_t lane_bytes, int64_t n_chunks, void* stream);
bool fb_copy_fits(const void* hdr);
// ... its second pass as a wave-cooperative splice (splice_block.hpp); the workspace's chunks must be full (the mark pass fills every lane's header)
void launch_fb_splice(const ScanArgs& a, const FbCopyArgs& ca, const void* hdr, int64_t lane_bytes, int64_t n_chunks, void* stream);
bool fb_splice_fits(const void* hdr);
// generator modes (gen_block.hpp): which 1 count, 2 emit; a.blob = the tables of serialize_gen (runtime.cpp), chunks of 256 lanes
void launch_gen(int which, const ScanArgs& a, const GenArgs& ga, int64_t lane_bytes, int64_t n_chunks, void* stream);
// the stack guard (guard_block.hpp): flags = a bit per window of `window` bytes without a '\n' (u64 per 64 windows, room for a multiple
// of 256 windows); then the reference's search on the lines that cover the runs of such windows — `slots` stacks, a thread each
void launch_guard_probe(const ScanArgs& a, int64_t window, int64_t n_windows, uint64_t* flags, void* stream);
void launch_guard(bool out, const ScanArgs& a, const GuardArgs& ga, int64_t n_runs, int64_t slots, void* stream);
// the backtracking fallback (gen_block.hpp: bt_lane): which 1 count, 2 emit; pool_blocks workgroups of 256 threads take the chunks in turn (ga holds
// pool_blocks * 256 stacks and path buffers)
void launch_bt(int which, const ScanArgs& a, const GenArgs& ga, int64_t lane_bytes, int64_t n_chunks, int64_t pool_blocks, uint32_t budget, void* stream);
// the deterministic engine on tables still being built (lazy_block.hpp): which 1 count (lanes whose lane_counts entry is not kLazyVoid keep
// it), 2 emit (leaves at once when the count pass was not final); chunks of 256 lanes
void launch_lazy(int which, const ScanArgs& a, const LazyArgs& la, int64_t lane_bytes, int64_t n_chunks, void* stream);
// exact sub-ranges (scan_block.hpp: ScanArgs::exact): flags the lanes whose guessed entry state is not the exit state of the lane before
// them (a.spec_flags, a.status[3] counts them); the repair round is launch_direct_kernel(1, ...) with a.exact == 3
void launch_spec_verify(const ScanArgs& a, int64_t n_lanes, void* stream);
// ... and of the backward pass: a.rev_guess against the symbols the lanes to the right left (a.rev_flags, a.status[2] counts the wrong ones); the
// flagged lanes sweep again (and on to the left while their result is not what was assumed there)
void launch_rev_verify(const ScanArgs& a, int64_t lane_bytes, void* stream, bool packed);
void launch_rev_repair(const ScanArgs& a, int64_t lane_bytes, void* stream, bool packed);
// 4 096 samples: how many find no '\n' within `window` bytes (added to *out)
void launch_line_probe(const ScanArgs& a, int64_t window, uint32_t* out, void* stream);
void launch_chunk_scan(const uint64_t* total, uint64_t* base, int64_t n_chunks, void* stream);
void launch_bytemap(const ScanArgs& a, void* stream);
void launch_nul_eol(const uint8_t* in, int64_t n, const uint64_t* pos, uint64_t* eol, uint32_t count, void* stream);
void launch_bytemap_shift(const uint8_t* blob, const uint8_t* src, uint8_t* dst, int64_t len, bool nl, void* stream);

}  // namespace trre
