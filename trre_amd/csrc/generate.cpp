// generate.cpp — generator mode (`trre -a`, `trre -ma`): every accepting path prints.
//
// The reference's search (infer_backtrack, trre_nft.c:593-657) does not stop at the first FINAL when `all` is set
// (trre_nft.c:640-641, 647-648): it prints that path's output, takes the next alternative from its stack and goes on
// until the stack is empty, then reports "no match" — so in scan mode a position prints the outputs of ALL accepting
// paths that start there (search order), then its raw byte (trre_nft.c:780-786); in match mode (`-ma`, what the
// reference's own test.sh runs) a line prints one output + '\n' per path that accepts at its end.
//
// The amount of output is unbounded in the input (every start position, every parse), so this is not a streaming
// kernel.  Split as follows:
//   device   the VIABILITY FILTER: a backward DFA over the line (same kernel as the guided families' backward pass,
//            k_rev_sweep) whose state at position i says, for every consuming node t, whether SOME path from t at i
//            accepts or SOME path runs into an epsilon cycle.  One symbol per input byte.
//   host     the enumeration over the follow lists (all epsilon paths, no dedup, search order), entering a node only
//            if the symbol says it is viable: branches that neither print nor diverge are never walked, so the work
//            is proportional to what is printed, not to the (possibly exponential) number of failing alternatives.
// A path that runs into an epsilon cycle ends the scan the way the reference ends ("error: stack max capacity
// reached", exit 1, what was printed stays printed).  Not modelled, as in scan mode: the reference's limit of
// 65 536 live stack items.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

#include "front.hpp"

namespace trre {
namespace {

struct AnySets {
    std::vector<uint32_t> acc;   // sorted: nodes with an accepting continuation
    std::vector<uint32_t> div;   // sorted: nodes with a continuation that never returns
};
bool has(const std::vector<uint32_t>& v, uint32_t x) { return std::binary_search(v.begin(), v.end(), x); }

}  // namespace

namespace {
// The filter that filters nothing: one state inside a line, every node "viable" — what a pattern gets whose viability automaton has more
// than 256 states (symbols are bytes).  The enumeration then walks the failing branches too, exactly as the reference's search does
// (trre_nft.c:593-657); what it prints is the same (a branch the filter skips prints nothing).  Round 4: TRRE_E_UNSUPPORTED.
void pass_all_filter(GenTables& g) {
    const size_t words = (g.nodes.node.size() + 63) / 64;
    g.n_rev = 3;
    g.rev.assign((size_t)3 * g.n_cls, (uint8_t)kSymDead);
    for (uint32_t r = 0; r < 3; ++r) { g.rev[(size_t)r * g.n_cls + 0] = (uint8_t)kSymEol; g.rev[(size_t)r * g.n_cls + 1] = (uint8_t)kSymNul; }
    g.viable_words = (uint32_t)words;
    g.viable.assign(3 * words, ~0ull);
    g.pass_all = true;
    g.ok = true;
}
}  // namespace

GenTables build_gen_tables(const Nft& nft, bool match_mode) {
    GenTables g;
    g.match_mode = match_mode;
    g.nodes = build_nft_nodes(nft, match_mode, /*all_paths=*/true);
    const NftNodes& nd = g.nodes;
    const uint32_t n_nodes = (uint32_t)nd.node.size();
    // byte classes: bytes read by the same nodes; class 0 = '\n', class 1 = NUL (never inside a line)
    std::map<std::vector<uint32_t>, uint8_t> index;
    std::vector<std::vector<uint32_t>> readers(2);
    g.cls['\n'] = 0;
    g.cls[0] = 1;
    for (int c = 1; c < 256; ++c) {
        if (c == '\n') continue;
        std::vector<uint32_t> sig;
        for (uint32_t t = 0; t < n_nodes; ++t)
            if (nd.node[t].reads((uint8_t)c)) sig.push_back(t);
        auto hit = index.find(sig);
        if (hit == index.end()) {
            if (readers.size() >= 256) throw Error(kErrTooBig, "error: more than 256 byte classes (generator mode)");   // (2 + 254 at most: not reachable)
            hit = index.emplace(sig, (uint8_t)readers.size()).first;
            readers.push_back(sig);
        }
        g.cls[c] = hit->second;
    }
    g.n_cls = (uint32_t)readers.size();
    // (TRRE_GEN_MAX_REV: a smaller limit, for the tests of the filter that lets everything through)
    const size_t max_rev = getenv("TRRE_GEN_MAX_REV") ? (size_t)std::max(3, atoi(getenv("TRRE_GEN_MAX_REV"))) : 256;
    // subset construction, right to left.  States 0..2: nothing viable — inside a line, at its '\n', at a NUL
    std::vector<AnySets> rev(3);
    std::map<std::vector<uint32_t>, uint32_t> rev_index;
    rev_index.emplace(std::vector<uint32_t>{0xffffffffu}, kSymDead);
    std::vector<std::vector<uint8_t>> rows;
    for (uint32_t r = 0; r < rev.size(); ++r) {
        const bool final_ok = !match_mode || r == kSymEol || r == kSymNul;   // what lies right of the byte being read
        std::vector<uint8_t> row(g.n_cls, 0);
        row[0] = (uint8_t)kSymEol;
        row[1] = (uint8_t)kSymNul;
        for (uint32_t k = 2; k < g.n_cls; ++k) {
            AnySets nx;
            for (uint32_t t : readers[k]) {
                bool acc = false, div = false;
                for (const NodeFollow& e : nd.follow[t]) {
                    if (e.target == kNodeFinal) acc = acc || final_ok;
                    else if (e.target == kNodeDiverge) div = true;
                    else { acc = acc || has(rev[r].acc, e.target); div = div || has(rev[r].div, e.target); }
                }
                if (acc) nx.acc.push_back(t);
                if (div) nx.div.push_back(t);
            }
            std::vector<uint32_t> key(nx.acc);
            key.push_back(0xffffffffu);
            key.insert(key.end(), nx.div.begin(), nx.div.end());
            auto hit = rev_index.find(key);
            if (hit == rev_index.end()) {
                if (rev.size() >= max_rev) { pass_all_filter(g); return g; } // symbols are bytes
                hit = rev_index.emplace(std::move(key), (uint32_t)rev.size()).first;
                rev.push_back(std::move(nx));
            }
            row[k] = (uint8_t)hit->second;
        }
        rows.push_back(std::move(row));
    }
    g.n_rev = (uint32_t)rev.size();
    g.rev.resize((size_t)g.n_rev * g.n_cls);
    for (uint32_t r = 0; r < g.n_rev; ++r) std::copy(rows[r].begin(), rows[r].end(), g.rev.begin() + (size_t)r * g.n_cls);
    const size_t words = (n_nodes + 63) / 64;
    g.viable_words = (uint32_t)words;
    g.viable.assign((size_t)g.n_rev * words, 0);
    for (uint32_t r = 0; r < g.n_rev; ++r) {
        for (uint32_t t : rev[r].acc) g.viable[r * words + t / 64] |= 1ull << (t % 64);
        for (uint32_t t : rev[r].div) g.viable[r * words + t / 64] |= 1ull << (t % 64);
    }
    g.ok = true;
    return g;
}

namespace {

// the outputs of all accepting paths of ONE attempt at position p of `line` (content bytes [0, len)); sym: the
// backward pass's symbols of the line's bytes.  Returns false when a path runs into an epsilon cycle.
struct Enumerator {
    const GenTables& g;
    struct Frame { uint32_t list; uint32_t idx; size_t i, olen; bool muted; };
    std::vector<Frame> stack;
    std::string cur;

    explicit Enumerator(const GenTables& gt) : g(gt) {}

    bool attempt(const uint8_t* line, size_t len, const uint8_t* sym, size_t p, std::vector<uint8_t>& out) {
        const NftNodes& nd = g.nodes;
        stack.clear();
        cur.clear();
        stack.push_back(Frame{(uint32_t)nd.node.size(), 0, p, 0, false});
        while (!stack.empty()) {
            Frame& f = stack.back();
            const std::vector<NodeFollow>& list = nd.follow[f.list];
            if (f.idx >= list.size()) { stack.pop_back(); continue; }
            const NodeFollow& e = list[f.idx++];
            if (e.target == kNodeDiverge) return false;
            if (e.target == kNodeFinal) {
                if (g.match_mode && f.i != len) continue;                    // trre_nft.c:636: only with the whole line consumed
                out.insert(out.end(), cur.begin(), cur.begin() + (ptrdiff_t)f.olen);
                if (!f.muted) out.insert(out.end(), e.out.begin(), e.out.end());
                if (g.match_mode) out.push_back('\n');
                continue;
            }
            if (f.i >= len) continue;
            const uint8_t c = line[f.i];
            if (!nd.node[e.target].reads(c)) continue;
            const uint64_t* v = g.viable.data() + (size_t)sym[f.i] * g.viable_words;
            if (!((v[e.target / 64] >> (e.target % 64)) & 1u)) continue;     // nothing printed, nothing entered below: skip
            Frame nx{e.target, 0, f.i + 1, f.olen, f.muted};
            cur.resize(f.olen);
            if (!nx.muted) {
                cur += e.out;
                if (e.mute) nx.muted = true;
                else if (nd.node[e.target].echo) cur.push_back((char)c);
            }
            nx.olen = cur.size();
            stack.push_back(nx);                                             // (invalidates f)
        }
        return true;
    }

    // one record: [line, line + len) is its content (cut at the first NUL, without the terminator)
    bool record(const uint8_t* line, size_t len, const uint8_t* sym, std::vector<uint8_t>& out) {
        if (g.match_mode) return attempt(line, len, sym, 0, out);            // trre_nft.c:791-797
        for (size_t p = 0; p < len; ++p) {                                   // trre_nft.c:780-786 with all = 1: no attempt returns > 0
            if (!attempt(line, len, sym, p, out)) return false;
            out.push_back(line[p]);
        }
        if (!attempt(line, len, sym, len, out)) return false;                // the empty tail (trre_nft.c:788)
        out.push_back('\n');
        return true;
    }
};

// records of in[lo, hi) (lo at a record start, hi at a record end or the end of the buffer)
bool generate_range(const GenTables& g, const uint8_t* in, const uint8_t* sym, size_t lo, size_t hi, std::vector<uint8_t>& out) {
    Enumerator en(g);
    size_t pos = lo;
    while (pos < hi) {
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(in + pos, '\n', hi - pos));
        const size_t reclen = nl ? (size_t)(nl - (in + pos)) + 1 : hi - pos;
        size_t len = reclen - 1;                                             // line[read - 1] = '\0' (trre_nft.c:777)
        const uint8_t* z = static_cast<const uint8_t*>(memchr(in + pos, 0, len));
        if (z) len = (size_t)(z - (in + pos));
        if (!en.record(in + pos, len, sym + pos, out)) return false;
        pos += reclen;
    }
    return true;
}

}  // namespace

bool generate_buffer(const GenTables& g, const uint8_t* in, size_t n, const uint8_t* sym, std::vector<uint8_t>& out, int threads) {
    if (threads < 1) threads = 1;
    if (n < ((size_t)1 << 20) || threads == 1) return generate_range(g, in, sym, 0, n, out);
    // records are independent: contiguous ranges of them on a few host threads, outputs concatenated in order
    std::vector<size_t> cut(threads + 1, n);
    cut[0] = 0;
    for (int t = 1; t < threads; ++t) {
        size_t c = n / (size_t)threads * (size_t)t;
        if (c < cut[t - 1]) c = cut[t - 1];
        const void* nl = c < n ? memchr(in + c, '\n', n - c) : nullptr;
        cut[t] = nl ? (size_t)(static_cast<const uint8_t*>(nl) - in) + 1 : n;
    }
    std::vector<std::vector<uint8_t>> part(threads);
    std::vector<char> ok(threads, 1), oom(threads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            try {
                ok[t] = generate_range(g, in, sym, cut[t], cut[t + 1], part[t]) ? 1 : 0;
            } catch (const std::bad_alloc&) {      // (the outputs of generator mode are unbounded: an error, not std::terminate)
                oom[t] = 1;
            }
        });
    for (auto& x : th) x.join();
    for (int t = 0; t < threads; ++t)
        if (oom[t]) throw std::bad_alloc();
    for (int t = 0; t < threads; ++t) {
        out.insert(out.end(), part[t].begin(), part[t].end());
        if (!ok[t]) return false;                                            // what was printed before the cycle stays printed
    }
    return true;
}

}  // namespace trre
