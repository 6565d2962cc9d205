/*
 * trre_mi355x.h — C ABI of the MI355X-native transducer scan engine.
 *
 * Drop-in boundary for the scan-mode hot path of c0stya/trre.  The reference
 * has no library interface: both engines are main() programs whose only seam
 * is the per-position call inside the scan line loop
 *
 *     ioffset = infer_backtrack(start, ch, stack, mode, all);   trre_nft.c:781
 *     ioffset = infer_dft(dstart, (unsigned char*)ch, dcache, mode);  trre_dft.c:1278
 *
 * (contract: NUL-terminated bytes in -> bytes appended to stdout, returns the
 * number of bytes consumed or <= 0).  One device call per input position is
 * far too fine a grain, so the boundary sits one level up: a whole
 * '\n'-delimited input buffer in, the whole output buffer out, same per-line
 * function, i.e. what main()'s scan branch computes for a FILE
 * (trre_nft.c:775-790, trre_dft.c:1272-1286).  INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Plain C types only.  The scan itself always runs on the GPU; there is no CPU
 * fallback in this library (a missing/failed device is an error).
 */
#ifndef TRRE_MI355X_H
#define TRRE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* engines: which reference binary's semantics to reproduce */
#define TRRE_ENGINE_NFT 0 /* ./trre      priority backtracking, trre_nft.c:593-657 */
#define TRRE_ENGINE_DFT 1 /* ./trre_dft  shortest-match determinised, trre_dft.c:1110-1196 */

/* modes: which branch of the reference's main() to reproduce */
#define TRRE_MODE_SCAN 0  /* default: every line is scanned for matches, the rest is copied (trre_nft.c:775-790) */
#define TRRE_MODE_MATCH 1 /* `trre -m`: the whole line must match; its output and '\n' are printed, a line that does not
                             match prints nothing (trre_nft.c:791-797, 635-642).  NFT engine only: the reference's
                             trre_dft -m prints an empty line per record (its emit is commented out, trre_dft.c:1185-1190) */
#define TRRE_MODE_SCAN_ALL 2  /* `trre -a` (generator mode, trre_nft.c:736-738 with 647-648): at every position of a line the
                                 outputs of ALL accepting paths that start there are printed, in the search's depth-first
                                 priority order, then that position's raw byte (no attempt returns a match: trre_nft.c:780-786) */
#define TRRE_MODE_MATCH_ALL 3 /* `trre -ma` (what the reference's own test.sh runs, test.sh:4): one output + '\n' per path
                                 that accepts at the end of the line (trre_nft.c:635-642).
                                 Both generator modes: NFT engine only (trre_dft -a prints "Not supported yet",
                                 trre_dft.c:1227-1229).  The amount of output is unbounded in the input; the device computes
                                 the viability filter (one symbol per input byte: which nodes have an accepting or a
                                 non-terminating continuation) and, since round 4, enumerates the accepting paths itself —
                                 count, exclusive sum, emit: trre_amd/csrc/gen_block.hpp —; the host enumeration of round 3
                                 (trre_amd/csrc/generate.cpp) takes a chunk on which a path never returns or a search
                                 outgrows a lane's stack. */

/* return codes */
#define TRRE_OK 0
#define TRRE_E_SYNTAX (-1)      /* the reference prints "error: ..." and exits 1 (message kept) */
#define TRRE_E_UNDEFINED (-2)   /* the reference reads outside its buffers on this pattern */
#define TRRE_E_EPS_CYCLE (-3)   /* epsilon cycle: the reference recurses without bound (DFT) */
#define TRRE_E_TOO_BIG (-4)     /* DFT engine, at run time: the determinised states THIS INPUT visits do not fit the memory limit of the lazy
                                   tables (TRRE_LAZY_MAX_BYTES, 16 GiB; the reference keeps every state it meets, too).  Rounds 1-4 returned it
                                   at compile time for every pattern beyond the eager construction's caps. */
#define TRRE_E_UNSUPPORTED (-5) /* legal pattern, outside this engine's GPU limits (modes other than scan: a backward DFA beyond the guided
                                   families' limits; scan mode, NFT engine, at run time: an attempt beyond the backtracking fallback's limits) */
#define TRRE_E_DEVICE (-6)      /* HIP runtime failure / no GPU */
#define TRRE_E_ARG (-7)
#define TRRE_E_DIVERGES (-8)    /* the reference does not survive this input: an epsilon cycle is entered (NFT: "error: stack max
                                   capacity reached", exit 1; DFT: unbounded recursion, SIGSEGV).  NFT engine: the reference exits
                                   with everything it had printed so far — the lines before the bad one and the bad line's output up
                                   to the attempt that does not return (trre_nft.c:551-553, exit() flushes stdout) — and so does the
                                   scan: those bytes are in the output buffer and *out_len is their count.  DFT engine: the
                                   reference's buffered output dies with it; *out_len = 0.  The same error, with the same bytes, when
                                   an attempt of the NFT engine's search would hold more than 65 536 untried alternatives
                                   (trre_nft.c:35-36,548-556: a greedy loop over a run of 65 536 bytes): the stack guard (round 4;
                                   rounds 1-3 printed the match) finds the lines long enough for that and runs the reference's search
                                   on them, scan and match modes; TRRE_NO_STACK_GUARD=1 switches it off.  Not decided, and left as
                                   the table kernels print it: a line whose search takes more than 8 M steps (TRRE_GUARD_BUDGET), the
                                   suspect lines behind the first 2^35 steps of such searches in one call (TRRE_GUARD_CALL_BUDGET) —
                                   step counts, not a clock (round 4: 20 s): the same input gives the same answer whatever the host is
                                   busy with, and the call SAYS so: TRRE_SCAN_GUARD_UNDECIDED in trre_last_scan_flags() —, patterns
                                   whose loops nest more than 64 first-tried branches between two reads, generator modes */
#define TRRE_E_CAPACITY (-9)    /* output buffer too small; *out_len holds the size needed */

/* kernel families (trre_info.kernel, trre_set_kernel) */
#define TRRE_KERNEL_AUTO 0
#define TRRE_KERNEL_BYTEMAP 1   /* memoryless tables: streaming byte map */
#define TRRE_KERNEL_TILE_LP 2   /* length-preserving tables: one lane per line, single launch */
#define TRRE_KERNEL_TILE_GEN 3  /* any tables: count + scan + emit */
#define TRRE_KERNEL_STREAM_LP 4 /* scan loop folded into the tables, length-preserving: in-place, single launch */
#define TRRE_KERNEL_STREAM_GEN 5 /* scan loop folded into the tables, any output length: count + scan + emit */
#define TRRE_KERNEL_GUIDED_LP 6  /* NFT engine, any pattern (round 4: DFT engine too, any pattern that is not a byte map): backward DFA sweep (one symbol per byte) + guided forward transducer, in place */
#define TRRE_KERNEL_GUIDED_GEN 7 /* the same, any output length: backward sweep, count + scan + emit */

#define TRRE_KERNEL_GENERATE 8    /* generator modes: backward viability sweep and enumeration (count, exclusive sum, emit) on the device;
                                     a chunk on which a path never returns or a search outgrows a lane's stack: the host enumeration */

#define TRRE_KERNEL_BACKTRACK 9   /* NFT engine, scan mode, any pattern: the reference's depth-first search itself, a lane per sub-range with an
                                     explicit stack (round 4).  What a pattern beyond the limits of every other family runs on (round 3:
                                     TRRE_E_UNSUPPORTED); exponential where the reference is.  A 1 KiB sub-range whose search takes more than 16 M steps, an
                                     attempt that consumes more than 4 096 bytes or builds more than 4 KiB of output: TRRE_E_UNSUPPORTED at run time */

#define TRRE_KERNEL_DFT_LAZY 10   /* DFT engine, scan mode, any pattern: determinisation on the fly as in the reference (trre_dft.c:1135-1175) — a lane per
                                     sub-range walks the rows that exist, an edge nobody has explored yet is listed and built on the host, the lanes that met
                                     it run again (round 5).  What a pattern beyond the eager construction's caps runs on ('((a:x)*b)|((a:y)*c)',
                                     '(a|b)*a(a|b){18}:x'; rounds 1-4: TRRE_E_TOO_BIG at compile time) */

typedef struct trre_prog trre_prog;

typedef struct trre_info {
    int32_t engine;
    int32_t kernel;            /* family AUTO resolves to */
    uint32_t nft_states;       /* states of the compiled NFT (trre_nft.c:334-340) */
    uint32_t nft_cons_states;
    uint32_t dft_states;       /* determinised states incl. final ones (DFT engine) */
    uint32_t table_rows;       /* rows kept on the device (non-final states) */
    uint32_t table_classes;    /* byte classes (columns) */
    uint32_t table_bytes;      /* size of the device blob */
    uint32_t flags;            /* bit0 length-preserving, bit1 memoryless, bit2 no-overrun */
    uint32_t chunk_bytes;      /* input bytes owned by one workgroup */
    uint32_t stream_states;    /* states of the folded scan transducer (0 = pattern does not fold) */
    uint32_t stream_classes;
    uint32_t nft_nodes;         /* consuming nodes of the NFT engine's tables (a byte range is one node) */
    uint32_t guided_rev_states; /* states of the backward DFA of the guided families (0 = not available) */
    uint32_t guided_fwd_states;
} trre_info;

/* Replaces parse() + create_nft() (trre_nft.c:752-754) and, for the DFT engine,
 * the lazy table construction of infer_dft (trre_dft.c:1135-1175) run to
 * completion.  Host-only; no GPU needed.  On failure returns a negative code
 * and *out = NULL; trre_last_error() then holds the reference's stderr text. */
int trre_compile(const char* pattern, int engine, trre_prog** out);
int trre_compile_bytes(const uint8_t* pattern, size_t len, int engine, trre_prog** out);
int trre_compile_mode(const uint8_t* pattern, size_t len, int engine, int mode, trre_prog** out);
void trre_free(trre_prog* p);
const char* trre_last_error(void); /* thread-local */
int trre_get_info(const trre_prog* p, trre_info* info);
int trre_set_kernel(trre_prog* p, int kernel_family); /* force a family (benchmarks/tests) */

/* Copy of the device table blob (for offline inspection and the host-side table
 * tests).  Returns the blob size; copies min(size, cap) bytes. */
size_t trre_export_tables(const trre_prog* p, void* buf, size_t cap);
size_t trre_export_stream_tables(const trre_prog* p, void* buf, size_t cap); /* 0 if the pattern does not fold */
size_t trre_export_guided_tables(const trre_prog* p, int which, void* buf, size_t cap); /* which: 0 backward DFA, 1 forward tables; 0 if none */

/* Replaces the scan branch of main() (trre_nft.c:775-790 / trre_dft.c:1272-1286)
 * for a whole buffer that is already resident in HBM.
 *   d_in/d_out : device pointers (any alignment; 16-byte aligned is the fast path)
 *   n          : input bytes        cap : capacity of d_out in bytes
 *   out_len    : bytes produced (or needed, with TRRE_E_CAPACITY)
 *   stream     : hipStream_t (NULL = default stream)
 * Semantics, byte for byte: output = concat over getline() records of
 * scan_line(record minus its last byte, cut at the first NUL) + "\n".
 * Synchronous with respect to `stream` on return. */
int trre_scan_device(trre_prog* p, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, size_t* out_len,
                     void* stream);

/* What the last trre_scan_* call on the calling thread has to say beside its return code (thread-local, like trre_last_error;
 * trre_scan_finish adds to what its trre_scan_enqueue found). */
#define TRRE_SCAN_GUARD_UNDECIDED 1u /* NFT engine: the input holds a line long enough to exhaust the reference's 65 536-item stack
                                        (trre_nft.c:35-36,548-556) whose search the stack guard did not finish within its step budgets: the
                                        output is what the table kernels print — the match — where the reference MAY have exited 1 */
uint32_t trre_last_scan_flags(void);

/* Split form for back-to-back launches: enqueue only (no host sync), then collect status/size once.
 * One scan may be in flight per (prog, device); enqueues repeated before the finish must be the same
 * scan (same buffers, size and stream: a benchmark loop) — anything else returns TRRE_E_ARG.  The split
 * form uses the calling thread's current device and is not serialised against other threads. */
int trre_scan_enqueue(trre_prog* p, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, void* stream);
int trre_scan_finish(trre_prog* p, size_t* out_len);

/* Host buffers on `device`: what the scan branch of the reference's main() does with a FILE* (the
 * getline loop of trre_nft.c:776-790 / trre_dft.c:1272-1286).  The input goes through in 64 MiB chunks
 * cut at line ends, three in flight on their own streams (staging copy, H2D, scan, D2H and the copy out
 * overlap); records are independent, so the chunks' outputs concatenate to exactly the output of one scan.
 * On TRRE_E_CAPACITY *out_len is the size the whole output needs (cap 0 / out NULL: a size query).  The
 * caller's current device is left as it was. */
int trre_scan_host(trre_prog* p, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, int device);

/* The same over several GPUs of this node, line-sharded: the input is cut at line ends into one contiguous
 * shard per selected device (trre_shard_bounds), each shard runs trre_scan_host on its device from a host
 * thread of its own, the outputs are concatenated in shard order (an exclusive sum of the shard sizes on
 * the host — the path has no exchange step, hence no collective).  device_mask: bit d selects visible
 * device d; 0 selects all of them.
 *
 * Threading: a compiled program may be used from several host threads at once; calls that target the same
 * device are serialised per (prog, device), calls on different devices run in parallel.  Compilation
 * (trre_compile) is single-threaded host work and trre_last_error() is per thread. */
int trre_scan_host_multi(trre_prog* p, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len,
                         uint32_t device_mask);

/* Kernel timing for the last finished scan on this prog (HIP events recorded on
 * the launch stream around the scan kernels).  Enable first. */
int trre_set_profiling(trre_prog* p, int on);
int trre_last_kernel_ms(trre_prog* p, float* ms);

/* Diagnostics / CPU test tier: the enumeration of the generator modes with viability symbols computed elsewhere (sym[i] for
 * byte i; tests/cpu_shim.cpp runs the backward kernel's per-thread body on the host).  Host-only, no device involved; not a
 * replacement for trre_scan_host, which computes the symbols on the GPU. */
int trre_debug_generate(trre_prog* p, const uint8_t* in, size_t n, const uint8_t* sym, uint8_t* out, size_t cap, size_t* out_len);

/* Diagnostics / CPU test tier: the lazily determinised tables as they stand (which 0: u32 n_cls, u32 rows, u8 cls[256]; 1: the entries
 * [rows][n_cls] of 8 bytes; 2: the pool of texts) and the exploration of a list of n miss records (16 words: row, class, m, 0, the m <= 48 bytes behind the byte that missed) plus up to spec_states states
 * ahead — what trre_scan_* does between two rounds of a launch of TRRE_KERNEL_DFT_LAZY.  Host-only. */
size_t trre_debug_lazy_tables(trre_prog* p, int which, void* buf, size_t cap);
int trre_debug_lazy_explore(trre_prog* p, const uint32_t* misses, size_t n, size_t spec_states);

/* Diagnostics / CPU test tier: the sharding and reassembly of trre_scan_host_multi — n_shards shards cut at line ends, a host thread each, outputs
 * concatenated in shard order, a shard that returns TRRE_E_DIVERGES ends the output, TRRE_E_CAPACITY asks for room — with the caller's function in
 * place of the per-shard device call (fixed_len: the stand-in is length-preserving, shards go straight to their place).  Host-only. */
typedef int (*trre_debug_shard_fn)(void* user, int shard, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len);
int trre_debug_scan_host_multi(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, int n_shards, int fixed_len,
                               trre_debug_shard_fn fn, void* user);

/* Line sharding (multi-GPU, trre has no exchange step: lines are independent).
 * Fills bounds[0..nshards] with byte offsets such that every shard but the
 * last ends just past a '\n'.  `in` is a host pointer. */
int trre_shard_bounds(const uint8_t* in, size_t n, int nshards, size_t* bounds);

#ifdef __cplusplus
}
#endif
#endif /* TRRE_MI355X_H */
