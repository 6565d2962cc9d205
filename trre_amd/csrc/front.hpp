// front.hpp — host-side pattern front end of the MI355X transducer scan engine.
//
// pattern string -> AST -> NFT (consume/produce automaton) -> device tables.
// Everything here runs once per pattern on the host; the per-byte work is in
// scan_kernels.hip.  The language and its corner cases are those of the
// reference (c0stya/trre); citations are file:line into /root/reference.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace trre {

// ---- error codes shared with include/trre_mi355x.h -------------------------
enum : int {
    kOk = 0,
    kErrSyntax = -1,        // the reference prints "error: ..." and exits 1
    kErrUndefined = -2,     // the reference runs into undefined behaviour here
    kErrEpsCycle = -3,      // epsilon cycle: the reference recurses/loops without bound
    kErrTooBig = -4,        // determinisation exceeds the state / residual caps
    kErrUnsupported = -5,   // legal pattern, not supported by this engine on the GPU
    kErrDevice = -6,        // HIP runtime failure
    kErrArg = -7,
    kErrDiverges = -8,      // at run time: the reference would not terminate on this input
    kErrCapacity = -9       // output buffer too small (needed size is reported)
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---- AST --------------------------------------------------------------------
// type: '|' '.' ':' '-' '*' '+' '?' 'I' (iteration) 'c' (byte) 'e' (epsilon
// placeholder); the bounds node under an 'I' stores {lower, upper} in
// {type, val}, both reduced mod 256 (trre_nft.c:142).
struct AstNode {
    uint8_t type = 0;
    uint8_t val = 0;
    int32_t l = -1, r = -1;
};
struct Ast {
    std::vector<AstNode> nodes;
    int32_t root = -1;
    int32_t add(uint8_t type, int32_t l = -1, int32_t r = -1, uint8_t val = 0) {
        AstNode n; n.type = type; n.val = val; n.l = l; n.r = r;
        nodes.push_back(n);
        return (int32_t)nodes.size() - 1;
    }
};
Ast parse_pattern(const std::string& pattern);

// ---- NFT ----------------------------------------------------------------------
enum class NKind : uint8_t { Prod, Cons, Split, SplitNg, Join, Final };
struct NState {
    NKind kind;
    uint8_t val = 0;
    int32_t a = -1;   // primary successor ("nexta")
    int32_t b = -1;   // secondary successor of a split ("nextb")
};
// A byte range outside ':' ("[a-z]", ".") or on its left side is a chain of SPLITNG states with one
// CONS branch per byte (trre_nft.c:426-435).  The branches read distinct bytes, are entered only through
// the chain's head and all continue at the same JOIN, so the table builders treat the whole chain as ONE
// consuming node with a byte set (`echo`: each CONS is followed by a PROD of the same byte — copy mode).
struct NGroup {
    int32_t head = -1;              // first SPLITNG of the chain
    int32_t join = -1;              // where every branch continues
    uint8_t lo = 0, hi = 0;         // bytes lo..hi
    bool echo = false;
    std::vector<int32_t> members;   // the CONS states, lowest byte first
};
struct Nft {
    std::vector<NState> st;
    std::vector<NGroup> groups;
    int32_t start = -1;
    int32_t n_cons = 0;
    int32_t add(NKind k, int32_t a = -1, int32_t b = -1, uint8_t val = 0) {
        NState s; s.kind = k; s.a = a; s.b = b; s.val = val;
        st.push_back(s);
        return (int32_t)st.size() - 1;
    }
    // preferred / fallback successor of a split in depth-first priority order
    // (trre_nft.c:623-630): SPLIT takes b (the loop body) first, SPLITNG takes a.
    int32_t first(int32_t s) const { return st[s].kind == NKind::Split ? st[s].b : st[s].a; }
    int32_t second(int32_t s) const { return st[s].kind == NKind::Split ? st[s].a : st[s].b; }
};
// with_initial_join: the deterministic engine's automaton starts with an extra
// JOIN so that the start state is never a CONS (trre_dft.c:516-523).
Nft build_nft(const Ast& ast, bool with_initial_join);

// ---- deterministic transducer (eager subset construction) ---------------------
constexpr int32_t kEdgeDead = -1, kEdgeDiverge = -2;
struct DftEdge {
    int32_t to = -1;          // -1 = dead; -2 = exploring this edge runs into an epsilon cycle: the reference's closure
                              // recursion (trre_dft.c:874-907) never returns
    std::string out;          // output factored onto the edge (longest common prefix)
};
struct DftState {
    bool final = false;
    bool diverges = false;           // its finality probe runs into an epsilon cycle: every edge into it diverges
    std::string final_out;
    std::array<DftEdge, 256> edge;   // only filled for non-final states
    bool expanded = false;
};
struct Dft {
    std::vector<DftState> st;     // st[0] = start (never final: trre_dft.c:938,1120)
};
struct DftLimits {
    size_t max_states = 60000;
    size_t max_residual = 4096;
    uint64_t max_work = 40000000;   // epsilon-closure steps over the whole construction
};
Dft determinize(const Nft& nft, const DftLimits& lim = DftLimits());

// ---- device tables --------------------------------------------------------------
// Deterministic engine.  Rows exist only for non-final states (a final state is
// left immediately: shortest match, trre_dft.c:1120-1125).  Bytes with identical
// columns share a class.  Entry (64 bit):
//   [1:0]   kind      0 dead, 1 goto, 2 accept (edge output already includes the
//                     target's final_out), 3 the reference does not return from this edge (epsilon cycle)
//   [4:2]   ilen      0..4 output bytes held inline in [63:32]; 7 = pooled:
//                     [63:32] is a byte offset into the pool of a record
//                     {u32 len, bytes...} (4-byte aligned)
//   [31:5]  next row  (goto only)
constexpr uint32_t kEntDead = 0, kEntGoto = 1, kEntAccept = 2, kEntDiverge = 3;
constexpr uint32_t kIlenPooled = 7;
constexpr uint32_t kClassEol = 0;      // '\n' and NUL: never inside a line

enum : uint32_t {
    kFlagLengthPreserving = 1u << 0,   // every accepted attempt emits exactly what it consumed
    kFlagMemoryless = 1u << 1,         // every start edge is dead or accepts with one output byte
    kFlagNoOverrun = 1u << 2,          // pending output never exceeds consumed input inside an attempt
    kFlagG16Slow = 1u << 3             // stream tables: some 16-byte entry is "slow" (more than 4 bytes or pooled text)
};

struct DftTables {
    uint32_t n_rows = 0, n_cls = 0, flags = 0, max_edge_out = 0;
    std::array<uint8_t, 256> cls{};
    std::vector<uint64_t> ent;           // [n_rows][n_cls]
    std::vector<uint8_t> pool;
    std::array<uint8_t, 256> bytemap{};  // valid when kFlagMemoryless
    uint32_t n_states = 0;               // determinised states incl. final ones
};
DftTables flatten_dft(const Dft& dft);

// ---- the same construction one table miss at a time (dft_build.cpp: LazyDft; round 5) --------------------------------
// What the reference does (trre_dft.c:1135-1175: a dstate is built when the input first takes an edge to it), for the
// patterns whose eager construction does not end within DftLimits — '((a:x)*b)|((a:y)*c)' has a state per run length,
// '(a|b)*a(a|b){18}:x' 2^19 of them — and, on request, for any pattern (the parity tests).  The tables are the device's
// (same entries as DftTables, rows of n_cls entries, classes fixed up front: a class per byte some CONS state reads, one
// for all the others, one for the line terminators) plus a third kind of dead entry:
constexpr uint64_t kEntUnexplored = 6u << 2;   // nobody has looked at this edge yet (kind "dead", ilen 6)
constexpr uint64_t kEntMissNoted = 5u << 2;    // device copy only: a lane has listed the edge among its launch's misses
// A scan walks what exists; a lane that meets an unexplored edge lists it {row, class} and is void; explore() builds the
// listed edges (and, breadth first from the states they led to, up to `spec_states` more: fewer rounds), the runtime
// uploads rows [first_dirty_row(), n_rows()) and the pool's new bytes and runs the void lanes again.
constexpr uint32_t kLazyMissWords = 16;
struct LazyLimits {
    size_t max_residual = (size_t)1 << 20;      // bytes of pending output in one item
    uint64_t max_edge_work = 400000000ull;      // closure steps for ONE edge the input takes
    uint64_t spec_edge_work = 200000;           // ... for one edge explored ahead of the input (beyond: left unexplored)
    size_t max_bytes = (size_t)16 << 30;        // host memory of the tables and the item lists (TRRE_LAZY_MAX_BYTES)
    size_t seed_states = 2048;                  // states built at compile time, breadth first from the start
};
class LazyDft {
public:
    explicit LazyDft(const Nft& nft, const LazyLimits& lim = LazyLimits());
    ~LazyDft();
    LazyDft(const LazyDft&) = delete;
    LazyDft& operator=(const LazyDft&) = delete;
    uint32_t n_cls() const;
    const uint8_t* cls() const;                 // [256]; class 0 = '\n' and NUL
    uint32_t n_rows() const;
    uint32_t n_states() const;                  // determinised states incl. final ones
    const uint64_t* ent() const;                // [n_rows][n_cls]
    const uint8_t* pool() const;
    size_t pool_bytes() const;
    // misses: n records of kLazyMissWords words {row, class, m, 0, the m <= 48 bytes of the line behind the byte that missed}: the edge is
    // built, then the attempt is followed along those bytes.  Throws Error(kErrTooBig) when an edge the INPUT takes cannot be built (the tables the
    // reference itself would need do not fit); edges explored ahead of the input never throw.
    void explore(const uint32_t* misses, size_t n, size_t spec_states);
    uint32_t epoch() const;                     // counts explore() calls that changed something
    uint32_t first_dirty_row(uint32_t since_epoch) const;   // first row changed after that epoch (n_rows(): none)
private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// Stream tables: the scan line loop folded into one deterministic transducer over
// the raw byte stream (stream_build.cpp).  Entry (64 bit):
//   [23:0]  next state's row offset (state * n_cls)
//   [26:24] olen   0..4 bytes held inline in [63:32]; 7 = pooled: [55:32] is the
//                  offset / 4 of a {u32 len, bytes} record in the pool and [63:56]
//                  min(len, 255) (255: read the record's length)
//   [27]    after those bytes, also emit the input byte itself
//   [28]    this byte ends the record ('\n')
//   [30]    the undecided attempt outgrew max_pending: the table does not know how it ends.  A launch
//           that meets such a transition reports it and the runtime runs the tile kernels instead
struct StreamLimits {
    size_t max_states = 16384;     // (a 4000-key dictionary folds into ~15 000 states: 3.5 MB of 8-byte rows)
    size_t max_pending = 64;
    size_t max_out = 4096;
};
struct StreamTables {
    bool ok = false;
    bool bounded = false;                   // some transitions carry the overflow mark (entry bit 30)
    uint32_t n_states = 0, n_cls = 0, flags = 0, max_out = 0;
    std::array<uint8_t, 256> cls{};
    std::vector<uint64_t> ent;              // [n_states][n_cls]
    std::vector<uint8_t> pool;
    std::vector<uint32_t> pending_len;      // bytes consumed but not yet emitted, per state
    // "window" form for length-preserving programs whose pending string never exceeds 7 bytes.
    // Up to 3 pending (transitions emit at most 4 bytes): 16 bytes per entry
    //   x  byte offset of the next state's row (state * n_cls * entry size)
    //   y  [5:0] insert shift = 8 * (delay - pending(source state)), [6] record end, [7] NUL
    //   z  up to four inline output bytes
    //   w  v_perm_b32 selector that builds the emitted byte sequence from {input byte, z}
    // 4..7 pending (at most 8 bytes): 32 bytes per entry, the same followed by a second {z, w} pair for
    // bytes 4..7 and two unused words; the kernel keeps a 64-bit window
    bool lpw_ok = false;
    uint32_t lpw_delay = 0;                 // max pending length
    std::vector<uint32_t> lpw;              // [n_states][n_cls][4 or 8]
    // pair form of the window entries (delay <= 3, at most 32 KiB): [n_states][n_cls][n_cls][8] = {next row (byte offset),
    // flags (64 / 512: the first / second byte ends a record, 128 NUL, 256 diverges), bytes 0..3, selector 0..3, bytes 4..7,
    // selector 4..7, -, -}: what two steps put into the window, already in place (the second step's bytes one byte up);
    // selectors as for the single entries, 5 = the second input byte.  Two input bytes per table read.
    bool lpw2_ok = false;
    std::vector<uint32_t> lpw2;
    // 16-byte entries for the count / emit passes of small tables (any output length): like the window
    // form without the shift, {next row offset, meta, inline bytes, v_perm selector}; meta [2:0] = bytes
    // the transition appends (inline bytes, then maybe the input byte), [3] a NUL cut a line short, [4] the
    // reference's search diverges (guided tables), [5] record end, [6] bounded-fold overflow, [7] "slow":
    // more than 4 bytes or pooled text — handled from the 8-byte entry, [8] identity: the transition emits exactly the byte
    // it reads; [9] every other transition: an edit for the patch path, [10] a transition of SKIP / DONE, [11] an edit for the
    // mark pass of the splice form ([9] and not [10]), [23:16] bytes emitted - 1 (signed, 0 bytes for a slow entry)
    bool g16_ok = false;
    std::vector<uint32_t> g16;              // [n_states][n_cls][4]
    // A MEMORYLESS program (stream_build.cpp: every cell of the root row leads back to the root, a NUL aside): what a byte becomes is a
    // function of that byte alone — [256][4] = {text lo, text hi, length (0..8) | 0x80: the byte cuts its record short, 0}, by raw byte.
    // mg_max: the longest text; 0: the program is not one (or prints more than 8 bytes for some byte).  map_block.hpp runs it in one pass.
    std::vector<uint32_t> mg;
    uint32_t mg_max = 0;
    // Pair form of a small table: one entry per (state, class of byte 0, class of byte 1) = the two transitions composed,
    // 32 bytes: {next row offset (in the 16-byte form), meta, bytes 0..3, selector, bytes 4..7, selector, 0, 0}; meta [3:0] =
    // bytes the pair appends (0..8: literal bytes and the two input bytes, v_perm picks 4 / 5), [4] diverge, [5] a record
    // ends inside the pair, [6] overflow, [7] "slow" (more than 8 bytes or pooled text: walked as two single steps),
    // [8] a NUL cut a line short, [9] one of the two transitions is an edit for the mark pass of the splice form, [10] the pair begins in
    // SKIP / DONE.  The count and emit passes walk two input bytes per table step wherever no lane of the
    // wave can finish (the passes are bound by instruction issue, §4.2 of DESIGN.md).
    bool p32_ok = false, p32_slow = false;
    std::vector<uint32_t> p32;              // [n_states][n_cls][n_cls][8]
    // Fallback form of a LARGE table (a dictionary: thousands of states, almost all of them deep inside a key with one
    // or two ways on) — everything the count and emit passes touch per byte fits in LDS (the 8-byte rows of the same
    // program are ~0.9 MB: a gather through L1/L2 on every byte, and the passes run at the pace of the L1's tag lookups).
    // The folded scan loop has the Aho-Corasick shape: a failed attempt flushes its first bytes RAW and what is left is a
    // shorter pending string, so the row of a state s is, on most classes, "the first bytes of the pending string, then
    // the row of the state f of a suffix of it" — only the ways deeper into a key are the state's own ("exceptions").
    // The exceptions of all states are interleaved in one array of 8-byte entries (row displacement, the "comb" of
    // compiler parse tables): state s owns the slots base(s) + k of its exceptional classes k, f(s) is a DENSE state
    // (root, SKIP, DONE, the one-byte pending strings: every class is its own), and a state travels as a descriptor
    //   [13:0] base   [27:14] base of f   [30:28] plen: bytes of the pending string in front of f's   [31] owed
    // so that one step reads fb_comb[base + k] and fb_comb[fbase + k] AT ONCE and takes the first if its tag says
    // "mine" — one LDS round trip per input byte, no loop, no divergence.  Entries:
    //   lo       the next state's descriptor
    //   hi [13:0] tag = base of the owning state    [16:14] n: leading bytes of the owner's pending string to emit
    //      [17] then the input byte   [18] then '\n'  ([17] and [18] both: escape, below)   [19] record end
    //      [31:20] about the next state: the length of its pending string, or (owed) the index of its text in fb_lit
    // No literal bytes in the entries: what a transition emits is a prefix of the pending string, i.e. of the input the
    // lane has just read (the kernels keep the last 7 input bytes in a register pair) — except replacement texts.  Those
    // are OWED: a completed key emits nothing and enters a copy of the root state (no slots of its own, f = root) whose
    // descriptor says "emit the text (plen + 1 = 1..8 bytes) in front of whatever the root row emits"; a key that is
    // complete but waits for a longer one to fail is the same with the ways on as exceptions.  The odd cell that is none
    // of this (raw bytes, then a replacement, then more; a replacement of more than 8 bytes) is an ESCAPE: its output is
    // spelled out in a record in global memory, found by the slot's index (fb_esc_slot, sorted).  Tables with more
    // than 7 bytes pending, more than 31 classes or overflow / diverge marks have no fallback form.  The relation
    // between the rows is verified cell by cell by the builder, never assumed.
    bool fb_ok = false;
    uint32_t fb_states = 0, fb_dense = 0;   // statistics: states (regular + owed-text), dense states
    uint32_t fb_start[3][2] = {};           // {descriptor, next-state bits [31:20] of an entry's hi} of root, SKIP, DONE
    std::vector<uint64_t> fb_comb;          // the slots; unused ones carry tag 0x3fff
    std::vector<uint64_t> fb_lit;           // literal texts, little-endian
    std::vector<uint32_t> fb_esc_slot;      // slots whose entry is an escape, ascending
    std::vector<uint32_t> fb_esc;           // their records, 4 words each: {offset in fb_pool, length, 1 = then the input byte, 0}
    std::vector<uint8_t> fb_pool;           // texts of the escape records
    // The same automaton read as NET EDITS (the "copy form": scan_block.hpp fb_lane<3> / fb_copy_lane): the output is the
    // input with a replacement text put where a key stood.  A text is emitted exactly when an owed state is left through
    // its fallback row, and it then stands for the fb_lit_meta[id] >> 8 input bytes right before the current one (the key,
    // or the whole pending string of a state that waited for a longer key); an escape record's text stands for the bytes
    // esc[4 i + 3] says ([7:0] how many, [15:8] how far back the first one lies).  Every other transition only passes
    // input bytes through, late.  The builder checks this reading cell by cell (fb_copy_ok).
    bool fb_copy_ok = false;
    std::vector<uint16_t> fb_lit_meta;      // per literal: [7:0] length of the text, [15:8] input bytes it stands for
    // The MARK FORM of the comb (round 4): what the copy form's first pass needs of an entry fits 32 bits, so its walk reads half
    // the LDS bytes and spends half the instructions.  Same slots as fb_comb; an entry:
    //   [13:0]  the next state's own base — for an owed-text state (no slots of its own) fb_pad + the index of its text: a stretch
    //           of never-owned slots behind the comb, so that the walk needs no special case and the event can name the text
    //   [20:14] the next state's fallback state, as an index into fb_dense4 (rows of 32 entries: the dense states)
    //   [21]    the next state is owed      [22] escape (the slot's record is looked up)      [23] record end
    //   [31:24] 4 x the class that owns the slot (124: nobody) — "mine" is one byte compare with the class the walk looks up
    // A state travels as bits [21:0].  Only for tables whose owed states have no slots of their own (a prefix-free dictionary:
    // no key waits for a longer one) and at most 128 dense states.
    bool fb_mark4_ok = false;
    uint32_t fb_pad = 0;                    // first slot of the stretch behind the comb
    uint32_t fb_start4[3] = {};             // descriptors of root, SKIP, DONE
    std::vector<uint32_t> fb_comb4;         // [fb_pad + literals + 32]
    std::vector<uint32_t> fb_dense4;        // [dense states][32]
    std::vector<uint16_t> fb_dense_base;    // per dense state: its base in the comb (escape records are keyed by slot)
};
StreamTables build_stream_dft(const Dft& dft, const StreamLimits& lim = StreamLimits());
StreamTables build_stream_nft(const Nft& nft, const StreamLimits& lim = StreamLimits());

// Non-deterministic engine (priority-exact).  Only consuming NODES (a CONS state, or a whole byte-range
// chain: NGroup) and FINAL are materialised; for each node s (and for the start) `follow` lists, in the
// reference's depth-first priority order, the nodes / FINAL reachable from s's successor through epsilon
// states, each with the bytes produced on the way.  A list stops at its first FINAL (which always accepts
// in scan mode, trre_nft.c:643-648) or with a Diverge marker where the reference's search would run
// around an epsilon cycle for ever ("stack max capacity reached", trre_nft.c:551-553).
constexpr uint32_t kNodeFinal = 0xFFFFFFFFu, kNodeDiverge = 0xFFFFFFFEu;
struct NodeFollow {
    uint32_t target = 0;   // node index, kNodeFinal or kNodeDiverge
    bool mute = false;     // the bytes produced on the way contain a NUL: `out` stops before it and the
                           // rest of the attempt's output is invisible (fputs, trre_nft.c:645)
    std::string out;
};
struct NftNodes {
    struct Node {
        std::array<uint64_t, 4> bytes{};   // the bytes it reads
        bool echo = false;                 // copy mode: reading a byte also produces it
        bool reads(uint8_t c) const { return (bytes[c >> 6] >> (c & 63)) & 1u; }
    };
    std::vector<Node> node;
    std::vector<std::vector<NodeFollow>> follow;   // [n_nodes + 1]; the last one is the start's
    bool has_diverge = false;
    bool match_mode = false;
    uint32_t n_states = 0;                         // states of the NFT they were made from
};
// match_mode (trre -m): FINAL accepts only at the end of the line and the search goes on past it otherwise
// (trre_nft.c:635-642), so a list does not stop at its first FINAL (it still holds FINAL at most once: the first
// occurrence in search order is the one that accepts).
// all_paths (generator mode, `-a`): EVERY epsilon path is listed, in search order, nothing is deduplicated and the list goes
// on past FINAL — each occurrence is a path of its own that prints its own output (trre_nft.c:640-641,647-648).
NftNodes build_nft_nodes(const Nft& nft, bool match_mode = false, bool all_paths = false);
// the scan loop folded over the follow lists (stream_build.cpp: NodeModel) — same tables as build_stream_nft, built faster
StreamTables build_stream_nodes(const NftNodes& nodes, const StreamLimits& lim = StreamLimits());

// Tables of the bitmask tile kernels: at most 64 nodes, no Diverge marker (their backward sweep is
// two-valued; patterns with epsilon cycles need the guided tables below).
constexpr uint8_t kTgtFinal = 0xFF, kTgtDiverge = 0xFE;
constexpr uint8_t kFollowMute = 1, kFollowEcho = 2;
struct NftFollow {
    uint8_t target;        // node index 0..n_cons-1 or kTgtFinal
    uint8_t flags;         // kFollowMute: output contained a NUL (emit up to it, then mute the attempt);
                           // kFollowEcho: the target also produces the byte it reads
    uint16_t out_len;
    uint32_t out_off;      // into pool
};
struct NftTables {
    uint32_t n_cons = 0;                    // nodes, <= 64
    std::array<uint64_t, 256> cons_mask{};  // nodes that read byte c
    std::vector<uint64_t> pred;             // [n_cons]: nodes whose follow list holds t
    uint64_t to_final = 0;                  // nodes whose follow list holds FINAL
    std::vector<uint32_t> follow_off;       // [n_cons + 2]; index n_cons = start
    std::vector<NftFollow> follow;
    std::vector<uint8_t> pool;
    uint32_t flags = 0;                     // kFlagLengthPreserving
    uint32_t n_states = 0;
};
NftTables build_nft_tables(const NftNodes& nodes);

// Guided tables ("bimachine"): the backtracking search of trre_nft.c:593-657 as two deterministic passes.
//   backward  a DFA over the input read right to left whose state at position i is, for every node, what
//             the reference's search from that node at i would end in (fail / accept / run for ever);
//             it leaves one symbol per input byte (its state id);
//   forward   a transducer over (symbol, byte) pairs in stream-table form whose state is the node the
//             first accepting path is in: at every step it takes the first follow entry that does not fail.
// '\n' and NUL reset the backward DFA (symbols kSymEol / kSymNul); symbol 0 is "nothing alive".
constexpr uint32_t kSymDead = 0, kSymEol = 1, kSymNul = 2;
struct GuidedLimits {
    size_t max_rev_states = 16384;    // up to 256: symbols are bytes and the backward table lives in LDS; beyond: 16-bit symbols,
                                      // both tables through L1 / L2 (the "wide" kernels: correct, an order of magnitude slower)
    size_t max_fwd_states = 4096;
    size_t max_out = 4096;
    size_t max_fwd_cells = 2u << 20;  // forward states x symbols (8 bytes each: 16 MB)
};
struct GuidedTables {
    bool ok = false;
    uint32_t n_rev = 0, n_cls = 0;          // backward DFA: states (= symbols) x byte classes
    uint32_t sym_bits = 8;                  // 4: at most 16 states and a small forward table — symbols are stored two per byte;
                                            // 16: more than 256 states (wide)
    bool wide = false;
    std::array<uint8_t, 256> cls{};         // byte -> class; class 0 = '\n', class 1 = NUL
    std::vector<uint8_t> rev;               // [n_rev][n_cls] next state (n_rev <= 256)
    std::vector<uint16_t> rev16;            // the same when wide
    StreamTables fwd;                       // columns = symbols (fwd.cls is unused)
};
// With nodes built for match mode the tables compute `trre -m` (trre_nft.c:791-797): one attempt per line from its first
// byte, accepted only if it ends exactly at the end of the line; an accepted line prints its output and '\n', a
// rejected one prints nothing.
GuidedTables build_guided_nft(const NftNodes& nodes, const GuidedLimits& lim = GuidedLimits());
// The deterministic engine in the same form (trre_dft.c:1110-1196): the backward DFA says at every position what becomes of an
// attempt from START there (and which class the byte has), the forward transducer walks the determinised tables only through
// attempts that succeed.  For patterns whose scan loop does not fold into a stream table (a loop before the decision:
// 'a*b:x', '[a-z]+ing:X') and as what takes over when a bounded fold overflows.
GuidedTables build_guided_dft(const Dft& dft, const GuidedLimits& lim = GuidedLimits());

// Generator mode (`trre -a` / `trre -ma`): generate.cpp.  The backward DFA (one symbol per input byte, computed on the
// device by the guided families' backward kernel) says which nodes are worth entering — SOME path accepts, or SOME path
// runs into an epsilon cycle —, the enumeration of the accepting paths over the un-deduplicated follow lists runs on the host.
struct GenTables {
    bool ok = false;
    bool match_mode = false;
    bool pass_all = false;                    // the viability automaton has more than 256 states: the filter lets every node through (generate.cpp)
    NftNodes nodes;                           // follow lists with every epsilon path
    uint32_t n_rev = 0, n_cls = 0;            // backward DFA: states (= symbols; 0 dead, 1 at '\n', 2 at a NUL) x byte classes
    std::array<uint8_t, 256> cls{};
    std::vector<uint8_t> rev;                 // [n_rev][n_cls]
    uint32_t viable_words = 0;
    std::vector<uint64_t> viable;             // [n_rev][viable_words]: bit t = node t is worth entering at a position with this symbol
};
GenTables build_gen_tables(const Nft& nft, bool match_mode);

// ---- the stack guard (stack_guard.cpp, guard_block.hpp): which lines can exhaust the reference's 65 536-item stack ----
struct GuardTables {
    bool on = false;                 // the pattern can fill the stack on a long enough line
    bool too_deep = false;           // ... but lines of under 1 KiB could: not guarded (documented deviation)
    uint32_t d = 0;                  // items per consumed byte, at most
    uint32_t l_min = 0, window = 0;  // lines shorter than l_min cannot overflow; the probe's window
    uint32_t run_min = 0;            // ... nor lines without a run of this many bytes of `bset`
    uint32_t n_once = 0;             // CONS states on no cycle
    uint32_t bset[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the bytes the pattern's loops read (never '\n', never NUL)
    uint32_t start = 0;
    std::vector<uint32_t> states;    // [n][4]: kind | val << 8, a, b, 0 (NKind order)
};
GuardTables build_guard(const Nft& nft);
// in[0, n): the whole buffer (records end at '\n'; the last byte of the buffer ends its record whatever it is, trre_nft.c:777);
// sym[i]: the backward pass's symbol of byte i.  Appends what the reference prints; false: a path ran into an epsilon cycle
// (the reference exits 1 there, `out` holds what it had printed).
bool generate_buffer(const GenTables& g, const uint8_t* in, size_t n, const uint8_t* sym, std::vector<uint8_t>& out, int threads);

}  // namespace trre
