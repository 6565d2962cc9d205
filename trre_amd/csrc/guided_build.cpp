// guided_build.cpp — the backtracking search of the reference's NFT engine
// (infer_backtrack, trre_nft.c:593-657, driven by the scan line loop trre_nft.c:775-790)
// as two deterministic passes over a line.
//
// What the reference's depth-first search from a consuming node t at input position i ends in
// depends only on the rest of the line:  val(t, i) = FAIL if t does not read line[i]; otherwise the
// search walks t's follow list (nft_tables.cpp) in priority order at position i + 1 — an entry FINAL
// accepts, an entry Diverge (an epsilon cycle) never returns, an entry t' recurses into val(t', i + 1)
// and on FAIL goes on to the next entry; the end of the list is FAIL.  So the vector val(., i) is a
// function of val(., i + 1) and line[i]:
//
//   backward pass   a DFA read right to left; its state at i is ({t : val(t,i) != FAIL},
//                   {t : val(t,i) = DIVERGE}).  Subset construction here, one table lookup per byte
//                   on the GPU, which stores the state id at every position (one byte: the "symbol").
//   forward pass    the search's first non-failing path, without the search: in node s at position
//                   i take the first follow entry that is FINAL, Diverge or alive in the symbol at
//                   i.  Entries FINAL end the attempt and the next attempt starts at the same
//                   position (or, if the attempt consumed nothing, after one raw byte: trre_nft.c:782-785).
//                   All of that is a function of (s, symbol): a transducer in stream-table form
//                   whose columns are symbols; echo bytes of copy-mode ranges and raw bytes are "the
//                   input byte" (copy flag), so the kernel needs the byte only to emit it.
//
// Both passes are exact for every pattern; what can fail is the size: more than 256 backward states
// (symbols are bytes) — then the bitmask tile kernels run the same two sweeps on masks.
#include <algorithm>
#include <map>

#include "front.hpp"
#include "stream_pack.hpp"

namespace trre {
namespace {

struct RevSets {
    std::vector<uint32_t> alive;   // sorted node ids with val != FAIL
    std::vector<uint32_t> div;     // sorted, subset of alive: val == DIVERGE
};

bool has(const std::vector<uint32_t>& v, uint32_t x) { return std::binary_search(v.begin(), v.end(), x); }

// index of the first entry of `list` the search does not get past, or -1 (the search fails).  final_ok: FINAL accepts
// here (always in scan mode; in match mode only when nothing of the line is left)
int decisive(const std::vector<NodeFollow>& list, const RevSets& r, bool final_ok) {
    for (size_t e = 0; e < list.size(); ++e) {
        const uint32_t t = list[e].target;
        if (t == kNodeFinal ? final_ok : (t == kNodeDiverge || has(r.alive, t))) return (int)e;
    }
    return -1;
}
bool diverges(const NodeFollow& f, const RevSets& r) {
    return f.target == kNodeDiverge || (f.target != kNodeFinal && has(r.div, f.target));
}

class GuidedBuilder {
public:
    GuidedBuilder(const NftNodes& nd, const GuidedLimits& lim)
        : nd_(nd), lim_(lim), n_nodes_((uint32_t)nd.node.size()), match_(nd.match_mode) {}

    GuidedTables run() {
        GuidedTables g;
        byte_classes(g);
        backward(g);
        forward(g);
        g.wide = g.n_rev > 256;
        g.sym_bits = g.wide ? 16 : ((g.n_rev <= 16 && g.fwd.g16_ok) ? 4 : 8);
        g.ok = true;
        return g;
    }

private:
    // bytes read by the same nodes behave alike; '\n' and NUL never occur inside a line
    void byte_classes(GuidedTables& g) {
        std::map<std::vector<uint32_t>, uint8_t> index;
        g.cls['\n'] = 0;
        g.cls[0] = 1;
        rep_ = {'\n', 0};
        for (int c = 1; c < 256; ++c) {
            if (c == '\n') continue;
            std::vector<uint32_t> sig;
            for (uint32_t t = 0; t < n_nodes_; ++t)
                if (nd_.node[t].reads((uint8_t)c)) sig.push_back(t);
            auto hit = index.find(sig);
            if (hit == index.end()) {
                if (rep_.size() >= 256) throw StreamGiveUp();
                hit = index.emplace(sig, (uint8_t)rep_.size()).first;
                rep_.push_back(c);
                readers_.resize(rep_.size());
                readers_.back() = sig;
            }
            g.cls[c] = hit->second;
        }
        readers_.resize(rep_.size());
        g.n_cls = (uint32_t)rep_.size();
    }

    uint32_t intern_rev(RevSets&& r) {
        std::vector<uint32_t> key(r.alive);
        key.push_back(0xffffffffu);
        key.insert(key.end(), r.div.begin(), r.div.end());
        auto hit = rev_index_.find(key);
        if (hit != rev_index_.end()) return hit->second;
        if (rev_.size() >= lim_.max_rev_states) throw StreamGiveUp();
        const uint32_t id = (uint32_t)rev_.size();
        rev_.push_back(std::move(r));
        rev_index_.emplace(std::move(key), id);
        return id;
    }

    void backward(GuidedTables& g) {
        // ids 0..2 all stand for "nothing alive": 0 inside a line, 1 at its '\n', 2 at a NUL (the symbols
        // differ because the forward pass treats the three positions differently; in match mode FINAL accepts
        // right of a line's last byte — states 1 and 2 — and nowhere else)
        rev_.assign(3, RevSets());
        rev_index_.emplace(std::vector<uint32_t>{0xffffffffu}, kSymDead);
        std::vector<std::vector<uint16_t>> rows;
        for (uint32_t r = 0; r < rev_.size(); ++r) {        // (rev_ grows while we go)
            std::vector<uint16_t> row(g.n_cls, 0);
            row[0] = (uint16_t)kSymEol;
            row[1] = (uint16_t)kSymNul;
            for (uint32_t k = 2; k < g.n_cls; ++k) {
                RevSets nx;
                for (uint32_t t : readers_[k]) {
                    // (a budget on the construction's work: up to 16 384 states x classes x nodes x follow entries would be hours)
                    work_ += nd_.follow[t].size() + 1;
                    if (work_ > kGuidedWork) throw StreamGiveUp();
                    const int e = decisive(nd_.follow[t], rev_[r], final_ok(r));
                    if (e < 0) continue;
                    nx.alive.push_back(t);
                    if (diverges(nd_.follow[t][e], rev_[r])) nx.div.push_back(t);
                }
                row[k] = (uint16_t)intern_rev(std::move(nx));
            }
            rows.push_back(std::move(row));
        }
        g.n_rev = (uint32_t)rev_.size();
        if (g.n_rev <= 256) {
            g.rev.resize((size_t)g.n_rev * g.n_cls);
            for (uint32_t r = 0; r < g.n_rev; ++r)
                for (uint32_t k = 0; k < g.n_cls; ++k) g.rev[(size_t)r * g.n_cls + k] = (uint8_t)rows[r][k];
        } else {
            g.rev16.resize((size_t)g.n_rev * g.n_cls);
            for (uint32_t r = 0; r < g.n_rev; ++r) std::copy(rows[r].begin(), rows[r].end(), g.rev16.begin() + (size_t)r * g.n_cls);
        }
    }

    bool final_ok(uint32_t sym) const { return !match_ || sym == kSymEol || sym == kSymNul; }
    static constexpr uint64_t kGuidedWork = 400u * 1000u * 1000u;      // follow entries looked at (a few seconds)
    uint64_t work_ = 0;

    // forward states: 0 root, 1 SKIP, 2 DONE (the stream kernels' conventions), then (node, muted)
    uint32_t intern_fwd(uint32_t node, bool muted) {
        const uint64_t key = (uint64_t)node * 2 + (muted ? 1 : 0);
        auto hit = fwd_index_.find(key);
        if (hit != fwd_index_.end()) return hit->second;
        if (fwd_.size() >= lim_.max_fwd_states) throw StreamGiveUp();
        const uint32_t id = (uint32_t)fwd_.size();
        fwd_.push_back(key);
        fwd_index_.emplace(key, id);
        return id;
    }

    StreamCell cell(uint32_t s, uint32_t y) {
        StreamCell c;
        if (s == 1) { c.next = y == kSymEol ? 0u : 1u; c.eol = y == kSymEol; return c; }
        if (s == 2) { c.next = 2; return c; }
        const RevSets& r = rev_[y];
        const bool at_end = y == kSymEol || y == kSymNul;
        if (match_) return match_cell(s, y, r, at_end);
        const std::vector<NodeFollow>& start = nd_.follow[n_nodes_];
        bool fresh = s == 0;
        bool muted = fresh ? false : (fwd_[s] & 1) != 0;
        const std::vector<NodeFollow>* cur = fresh ? &start : &nd_.follow[fwd_[s] / 2];
        bool ended = false;          // the line's attempts are over (at_end only)
        for (;;) {
            const int e = decisive(*cur, r, true);
            if (e < 0) {
                if (!fresh) {        // cannot happen: a node is only entered when its search does not fail
                    c.diverge = true; c.next = 1; c.out.clear();
                    return c;
                }
                if (at_end) { ended = true; break; }
                c.copy_c = true;     // no match here: one raw byte (trre_nft.c:784-785)
                c.next = 0;
                break;
            }
            const NodeFollow& f = (*cur)[e];
            if (diverges(f, r)) {
                // "error: stack max capacity reached" in the reference, which exits with what it has printed so far: the
                // outputs of the attempts that ended at this position stay (c.out), nothing of this attempt does (its
                // bytes would only be printed at FINAL, trre_nft.c:643-645), and the lane is finished (DONE: absorbing)
                c.diverge = true; c.next = 2;
                return c;
            }
            if (f.target == kNodeFinal) {
                if (!muted) c.out += f.out;
                if (fresh) {         // an attempt that consumed nothing
                    if (at_end) { ended = true; break; }     // ... on the empty tail (trre_nft.c:788)
                    c.copy_c = true;                         // ... inside the line: the raw byte follows (trre_nft.c:782-785)
                    c.next = 0;
                    break;
                }
                fresh = true;        // the next attempt starts at this very position
                muted = false;
                cur = &start;
                continue;
            }
            // this byte is consumed by node f.target
            if (!muted) c.out += f.out;
            if (f.mute) muted = true;
            if (nd_.node[f.target].echo && !muted) c.copy_c = true;
            c.next = intern_fwd(f.target, muted);
            break;
        }
        if (ended) {
            c.out.push_back('\n');
            c.next = y == kSymNul ? 1u : 0u;
            c.eol = y == kSymEol;
        }
        if (c.out.size() > lim_.max_out) throw StreamGiveUp();
        return c;
    }

    // trre -m (trre_nft.c:791-797, 635-642): ONE attempt per line from its first byte, accepted only at the end of the
    // line; an accepted line prints its output and '\n', a rejected one nothing at all.  Root is "at the start of a
    // line"; a rejected line is swallowed by SKIP, whose '\n' transition is silent.
    StreamCell match_cell(uint32_t s, uint32_t y, const RevSets& r, bool at_end) {
        StreamCell c;
        const bool fresh = s == 0;
        bool muted = fresh ? false : (fwd_[s] & 1) != 0;
        const std::vector<NodeFollow>& list = fresh ? nd_.follow[n_nodes_] : nd_.follow[fwd_[s] / 2];
        const int e = decisive(list, r, at_end);
        if (e < 0) {
            if (!fresh) { c.diverge = true; c.next = 1; return c; }        // cannot happen: a node is entered only when alive
            c.next = y == kSymEol ? 0u : 1u;                               // no match: the line prints nothing
            c.eol = y == kSymEol;
            return c;
        }
        const NodeFollow& f = list[e];
        if (diverges(f, r)) { c.diverge = true; c.next = 2; return c; }      // (nothing of this line is printed; the lane is finished)
        if (!muted) c.out += f.out;
        if (f.target == kNodeFinal) {                                      // (only at the end of the line)
            c.out.push_back('\n');
            c.next = y == kSymNul ? 1u : 0u;
            c.eol = y == kSymEol;
        } else {
            if (f.mute) muted = true;
            if (nd_.node[f.target].echo && !muted) c.copy_c = true;
            c.next = intern_fwd(f.target, muted);
        }
        if (c.out.size() > lim_.max_out) throw StreamGiveUp();
        return c;
    }

    void forward(GuidedTables& g) {
        StreamPackInput in;
        fwd_.assign(3, 0);
        in.wide_cols = g.n_rev > 256;
        for (uint32_t s = 0; s < fwd_.size(); ++s) {        // (fwd_ grows while we go)
            if ((uint64_t)(s + 1) * g.n_rev > lim_.max_fwd_cells) throw StreamGiveUp();
            std::vector<StreamCell> row;
            row.reserve(g.n_rev);
            for (uint32_t y = 0; y < g.n_rev; ++y) row.push_back(cell(s, y));
            in.rows.push_back(std::move(row));
        }
        for (uint32_t y = 0; y < g.n_rev; ++y) in.col_kind.push_back(y == kSymEol ? kColNewline : (y == kSymNul ? kColNul : kColPlain));
        in.skip = 1;
        in.done = 2;
        // pending bytes per state (consumed, not yet emitted) if that is a function of the state
        const uint32_t n = (uint32_t)fwd_.size();
        std::vector<int64_t> pend(n, INT64_MIN);
        std::vector<uint32_t> work{0};
        pend[0] = 0;
        bool lp = true;
        while (!work.empty() && lp) {
            const uint32_t s = work.back();
            work.pop_back();
            for (uint32_t y = 0; y < g.n_rev && lp; ++y) {
                const StreamCell& x = in.rows[s][y];
                if (y == kSymNul || x.diverge) continue;
                const int64_t p = pend[s] + 1 - (int64_t)x.out.size() - (x.copy_c ? 1 : 0);
                if (p < 0) { lp = false; break; }
                if (pend[x.next] == INT64_MIN) { pend[x.next] = p; work.push_back(x.next); }
                else if (pend[x.next] != p) lp = false;
            }
        }
        in.never_lp = !lp || match_;
        in.pending_len.assign(n, 0);
        if (lp)
            for (uint32_t s = 0; s < n; ++s) in.pending_len[s] = pend[s] == INT64_MIN || s == 1 || s == 2 ? 0u : (uint32_t)pend[s];
        g.fwd = pack_stream_tables(in);
    }

    const NftNodes& nd_;
    GuidedLimits lim_;
    uint32_t n_nodes_;
    bool match_;
    std::vector<int> rep_;                          // class -> a representative byte
    std::vector<std::vector<uint32_t>> readers_;    // class -> nodes that read its bytes
    std::vector<RevSets> rev_;
    std::map<std::vector<uint32_t>, uint32_t> rev_index_;
    std::vector<uint64_t> fwd_;                     // forward state -> node * 2 + muted (ids 0..2 are special)
    std::map<uint64_t, uint32_t> fwd_index_;
};

}  // namespace

GuidedTables build_guided_nft(const NftNodes& nodes, const GuidedLimits& lim) {
    try {
        return GuidedBuilder(nodes, lim).run();
    } catch (const StreamGiveUp&) {
        return GuidedTables();        // ok == false
    }
}

}  // namespace trre
