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

// columns, pending bytes per state (consumed, not yet emitted) if that is a function of the state, and the packer:
// shared by the two builders of forward tables (rows of cells over symbols; states 0 root, 1 SKIP, 2 DONE)
StreamTables pack_forward(StreamPackInput& in, uint32_t n_rev, bool never_lp) {
    for (uint32_t y = 0; y < n_rev; ++y) in.col_kind.push_back(y == kSymEol ? kColNewline : (y == kSymNul ? kColNul : kColPlain));
    in.skip = 1;
    in.done = 2;
    const uint32_t n = (uint32_t)in.rows.size();
    std::vector<int64_t> pend(n, INT64_MIN);
    std::vector<uint32_t> work{0};
    pend[0] = 0;
    bool lp = true;
    while (!work.empty() && lp) {
        const uint32_t s = work.back();
        work.pop_back();
        for (uint32_t y = 0; y < n_rev && lp; ++y) {
            const StreamCell& x = in.rows[s][y];
            if (y == kSymNul || x.diverge) continue;
            const int64_t p = pend[s] + 1 - (int64_t)x.out.size() - (x.copy_c ? 1 : 0);
            if (p < 0) { lp = false; break; }
            if (pend[x.next] == INT64_MIN) { pend[x.next] = p; work.push_back(x.next); }
            else if (pend[x.next] != p) lp = false;
        }
    }
    in.never_lp = !lp || never_lp;
    in.pending_len.assign(n, 0);
    if (lp)
        for (uint32_t s = 0; s < n; ++s) in.pending_len[s] = pend[s] == INT64_MIN || s == 1 || s == 2 ? 0u : (uint32_t)pend[s];
    return pack_stream_tables(in);
}

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
        g.fwd = pack_forward(in, g.n_rev, match_);
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

// ---- the deterministic engine's attempts as the same two passes ----------------------------------------------
// infer_dft (trre_dft.c:1110-1196) walks the determinised tables from START and succeeds at the FIRST final state it
// enters (:1120-1125); a dead edge or the end of the line discards the attempt (:1132-1134, :1193-1195) and the line loop
// (:1277-1283) copies one raw byte.  Whether the walk from a table state d at position i ends in success, failure or on an
// edge the reference's closure never returns from (kEdgeDiverge) depends only on the rest of the line, and as a vector over
// the states it is a function of the same vector at i + 1 and of line[i]: the backward DFA's raw state is (class of line[i],
// {d that do not fail}, {d that diverge}).  The forward pass needs two things of it: the byte's class (an attempt under way
// takes edge[d][class]: the walk is deterministic, and it is under way only if it succeeds) and what becomes of an attempt
// from START — so the raw automaton is reduced to the coarsest one that still tells these (Moore's partition refinement,
// output = (class, START's value)), and its states are the symbols.
class GuidedDftBuilder {
public:
    GuidedDftBuilder(const Dft& d, const GuidedLimits& lim) : d_(d), lim_(lim) {}

    GuidedTables run() {
        GuidedTables g;
        rows();
        byte_classes(g);
        raw_backward();
        reduce(g);
        forward(g);
        g.wide = g.n_rev > 256;
        g.sym_bits = g.wide ? 16 : ((g.n_rev <= 16 && g.fwd.g16_ok) ? 4 : 8);
        g.ok = true;
        return g;
    }

private:
    struct Raw { uint32_t k = 0; std::vector<uint32_t> alive, div; };      // (sorted row indices)
    enum : uint32_t { kFail = 0, kOk = 1, kDiv = 2 };

    // what reading byte c in state s does, with the target named by `name_of` (final targets by their output)
    std::string edge_sig(uint32_t s, int c, const std::vector<int32_t>& name_of) const {
        const DftEdge& e = d_.st[s].edge[c];
        if (e.to == kEdgeDiverge) return "D";
        if (e.to < 0) return "-";
        std::string sig = d_.st[e.to].final ? "F" : "G" + std::to_string(name_of[e.to]);
        sig.push_back(':');
        sig += std::to_string(e.out.size() + (d_.st[e.to].final ? d_.st[e.to].final_out.size() : 0));
        sig.push_back(':');
        sig += e.out;
        if (d_.st[e.to].final) sig += d_.st[e.to].final_out;
        return sig;
    }

    // table rows: the states an attempt can be in (a final state is left at once, a diverging one never entered), reduced:
    // the reference's determinisation keeps one state per item list ('[a-z]+' has a state per letter); states that emit the
    // same bytes and go to equivalent states on every byte are one row here (partition refinement; START stays alone at 0)
    void rows() {
        std::vector<uint32_t> live;
        for (size_t s = 0; s < d_.st.size(); ++s)
            if (!d_.st[s].final && !d_.st[s].diverges) live.push_back((uint32_t)s);
        if (live.empty() || live[0] != 0) throw StreamGiveUp();
        std::vector<int32_t> block(d_.st.size(), -1);
        for (uint32_t s : live) block[s] = s == 0 ? 0 : 1;
        size_t n_blocks = live.size() > 1 ? 2 : 1;
        for (;;) {
            std::map<std::string, int32_t> ids;
            std::vector<int32_t> nb(d_.st.size(), -1);
            nb[0] = 0;
            for (uint32_t s : live) {
                if (s == 0) continue;
                work_ += 256;
                if (work_ > kWork) throw StreamGiveUp();
                std::string sig = std::to_string(block[s]);
                for (int c = 1; c < 256; ++c) {
                    if (c == '\n') continue;
                    sig.push_back('|');
                    sig += edge_sig(s, c, block);
                }
                nb[s] = ids.emplace(std::move(sig), (int32_t)ids.size() + 1).first->second;
            }
            block.swap(nb);
            if (ids.size() + 1 == n_blocks) break;             // (a refinement only splits: the same count is the same partition)
            n_blocks = ids.size() + 1;
        }
        row_of_ = block;
        state_of_.assign(n_blocks, 0xffffffffu);
        for (uint32_t s : live)
            if (state_of_[row_of_[s]] == 0xffffffffu) state_of_[row_of_[s]] = s;
    }

    // bytes with the same edges out of every row behave alike; '\n' and NUL never occur inside a line
    void byte_classes(GuidedTables& g) {
        std::map<std::string, uint8_t> index;
        g.cls['\n'] = 0;
        g.cls[0] = 1;
        rep_ = {'\n', 0};
        for (int c = 1; c < 256; ++c) {
            if (c == '\n') continue;
            std::string sig;
            for (uint32_t s : state_of_) { sig += edge_sig(s, c, row_of_); sig.push_back('|'); }
            auto hit = index.find(sig);
            if (hit == index.end()) {
                if (rep_.size() >= 256) throw StreamGiveUp();
                hit = index.emplace(std::move(sig), (uint8_t)rep_.size()).first;
                rep_.push_back(c);
            }
            g.cls[c] = hit->second;
        }
        g.n_cls = (uint32_t)rep_.size();
    }

    uint32_t intern_raw(Raw&& r) {
        if (r.alive.empty()) return kSymDead;
        std::vector<uint32_t> key{r.k};
        key.insert(key.end(), r.alive.begin(), r.alive.end());
        key.push_back(0xffffffffu);
        key.insert(key.end(), r.div.begin(), r.div.end());
        auto hit = raw_index_.find(key);
        if (hit != raw_index_.end()) return hit->second;
        if (raw_.size() >= kMaxRaw) throw StreamGiveUp();
        const uint32_t id = (uint32_t)raw_.size();
        raw_.push_back(std::move(r));
        raw_index_.emplace(std::move(key), id);
        return id;
    }

    void raw_backward() {
        raw_.assign(3, Raw());                               // 0 nothing alive, 1 at a '\n', 2 at a NUL
        const uint32_t C = (uint32_t)rep_.size();
        for (uint32_t r = 0; r < raw_.size(); ++r) {         // (raw_ grows while we go)
            std::vector<uint32_t> row(C, 0);
            row[0] = kSymEol;
            row[1] = kSymNul;
            for (uint32_t k = 2; k < C; ++k) {
                work_ += state_of_.size();
                if (work_ > kWork) throw StreamGiveUp();
                Raw nx;
                nx.k = k;
                for (uint32_t q = 0; q < state_of_.size(); ++q) {
                    const DftEdge& e = d_.st[state_of_[q]].edge[rep_[k]];
                    if (e.to == kEdgeDiverge) { nx.alive.push_back(q); nx.div.push_back(q); continue; }
                    if (e.to < 0) continue;
                    if (d_.st[e.to].final) { nx.alive.push_back(q); continue; }
                    const uint32_t t = (uint32_t)row_of_[e.to];
                    if (!has(raw_[r].alive, t)) continue;
                    nx.alive.push_back(q);
                    if (has(raw_[r].div, t)) nx.div.push_back(q);
                }
                row[k] = intern_raw(std::move(nx));
            }
            raw_next_.push_back(std::move(row));
        }
    }

    uint32_t start_value(const Raw& r) const { return !has(r.alive, 0) ? kFail : (has(r.div, 0) ? kDiv : kOk); }

    // Moore reduction of the raw automaton; blocks 0..2 stay the three special symbols
    void reduce(GuidedTables& g) {
        const uint32_t n = (uint32_t)raw_.size(), C = (uint32_t)rep_.size();
        std::vector<uint32_t> block(n);
        {
            std::map<std::pair<uint32_t, uint32_t>, uint32_t> first;
            uint32_t next_id = 3;
            for (uint32_t r = 0; r < n; ++r) {
                if (r < 3) { block[r] = r; continue; }
                auto hit = first.emplace(std::make_pair(raw_[r].k, start_value(raw_[r])), next_id);
                if (hit.second) ++next_id;
                block[r] = hit.first->second;
            }
        }
        for (;;) {
            std::map<std::vector<uint32_t>, uint32_t> ids;
            std::vector<uint32_t> nb(n);
            uint32_t next_id = 3;
            std::vector<uint32_t> sig(C + 1);
            for (uint32_t r = 0; r < n; ++r) {
                if (r < 3) { nb[r] = r; continue; }
                work_ += C;
                if (work_ > kWork) throw StreamGiveUp();
                sig[0] = block[r];
                for (uint32_t k = 0; k < C; ++k) sig[k + 1] = block[raw_next_[r][k]];
                auto hit = ids.emplace(sig, next_id);
                if (hit.second) ++next_id;
                nb[r] = hit.first->second;
            }
            uint32_t before = 0, after = 0;
            for (uint32_t r = 0; r < n; ++r) { before = std::max(before, block[r]); after = std::max(after, nb[r]); }
            block.swap(nb);
            if (after == before) break;                       // (a refinement never merges: same count, same partition)
        }
        uint32_t n_blocks = 0;
        for (uint32_t r = 0; r < n; ++r) n_blocks = std::max(n_blocks, block[r] + 1);
        if (n_blocks > lim_.max_rev_states) throw StreamGiveUp();
        sym_k_.assign(n_blocks, 0);
        sym_start_.assign(n_blocks, kFail);
        std::vector<std::vector<uint16_t>> tab(n_blocks);
        for (uint32_t r = 0; r < n; ++r) {
            const uint32_t b = block[r];
            if (!tab[b].empty()) continue;
            tab[b].resize(C);
            for (uint32_t k = 0; k < C; ++k) tab[b][k] = (uint16_t)block[raw_next_[r][k]];
            sym_k_[b] = raw_[r].k;
            sym_start_[b] = r < 3 ? (uint32_t)kFail : start_value(raw_[r]);
        }
        g.n_rev = n_blocks;
        if (n_blocks <= 256) {
            g.rev.resize((size_t)n_blocks * C);
            for (uint32_t b = 0; b < n_blocks; ++b)
                for (uint32_t k = 0; k < C; ++k) g.rev[(size_t)b * C + k] = (uint8_t)tab[b][k];
        } else {
            g.rev16.resize((size_t)n_blocks * C);
            for (uint32_t b = 0; b < n_blocks; ++b) std::copy(tab[b].begin(), tab[b].end(), g.rev16.begin() + (size_t)b * C);
        }
    }

    // forward states: 0 root, 1 SKIP, 2 DONE (the stream kernels' conventions), then the table rows but START's (3 + row - 1)
    StreamCell cell(uint32_t s, uint32_t y) const {
        StreamCell c;
        if (s == 1) { c.next = y == kSymEol ? 0u : 1u; c.eol = y == kSymEol; return c; }
        if (s == 2) { c.next = 2; return c; }
        const bool at_end = y == kSymEol || y == kSymNul;
        if (s == 0) {
            if (at_end) {                                     // (no attempt on the empty tail: START is never final, trre_dft.c:938)
                c.out.push_back('\n');
                c.next = y == kSymNul ? 1u : 0u;
                c.eol = y == kSymEol;
                return c;
            }
            if (sym_start_[y] == kDiv) { c.diverge = true; c.next = 2; return c; }     // the reference does not come back from this attempt
            if (sym_start_[y] == kFail) { c.copy_c = true; c.next = 0; return c; }     // one raw byte (trre_dft.c:1281-1282)
        }
        // an attempt under way (or begun here) reads this byte: it is one that succeeds, so the edge exists
        const uint32_t q = s == 0 ? 0u : s - 2;
        const DftEdge* e = (at_end || y == kSymDead) ? nullptr : &d_.st[state_of_[q]].edge[rep_[sym_k_[y]]];
        if (!e || e->to < 0) { c.diverge = true; c.next = 1; return c; }               // cannot happen (a pair the passes never meet)
        c.out = e->out;
        if (d_.st[e->to].final) { c.out += d_.st[e->to].final_out; c.next = 0; }       // first final state: the attempt ends (trre_dft.c:1120-1125)
        else if (row_of_[e->to] <= 0) throw StreamGiveUp();                             // (START is never re-entered: its item list is the initial JOIN alone)
        else c.next = (uint32_t)row_of_[e->to] + 2;
        if (c.out.size() > lim_.max_out) throw StreamGiveUp();
        return c;
    }

    void forward(GuidedTables& g) {
        StreamPackInput in;
        in.wide_cols = g.n_rev > 256;
        const uint32_t n_fwd = (uint32_t)state_of_.size() + 2;      // (START is the root's attempt: rows 1.. are states 3..)
        if (n_fwd > lim_.max_fwd_states || (uint64_t)n_fwd * g.n_rev > lim_.max_fwd_cells) throw StreamGiveUp();
        for (uint32_t s = 0; s < n_fwd; ++s) {
            std::vector<StreamCell> row;
            row.reserve(g.n_rev);
            for (uint32_t y = 0; y < g.n_rev; ++y) row.push_back(cell(s, y));
            in.rows.push_back(std::move(row));
        }
        g.fwd = pack_forward(in, g.n_rev, false);
    }

    static constexpr size_t kMaxRaw = 65536;                  // raw backward states before the reduction
    static constexpr uint64_t kWork = 100u * 1000u * 1000u;      // row visits (a second or so: the 1000-key dictionary gives up here)
    const Dft& d_;
    GuidedLimits lim_;
    uint64_t work_ = 0;
    std::vector<int32_t> row_of_;                             // determinised state -> row, or -1
    std::vector<uint32_t> state_of_;
    std::vector<int> rep_;                                    // class -> a representative byte
    std::vector<Raw> raw_;
    std::map<std::vector<uint32_t>, uint32_t> raw_index_;
    std::vector<std::vector<uint32_t>> raw_next_;             // [raw state][class]
    std::vector<uint32_t> sym_k_, sym_start_;                 // per symbol: the byte class it stands on, START's value there
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

namespace trre {
GuidedTables build_guided_dft(const Dft& dft, const GuidedLimits& lim) {
    try {
        return GuidedDftBuilder(dft, lim).run();
    } catch (const StreamGiveUp&) {
        return GuidedTables();        // ok == false
    }
}
}  // namespace trre
