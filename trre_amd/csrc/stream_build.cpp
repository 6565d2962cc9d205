// stream_build.cpp — folds the reference's scan line loop INTO the tables.
//
// The reference scans a line by repeated attempts: run the automaton from the
// current position; on success print the attempt's output and skip what it
// consumed, otherwise copy one raw byte and retry one position further
// (trre_dft.c:1277-1283, trre_nft.c:780-786).  A failed attempt therefore
// re-reads input ("rollback"), which on a GPU means divergent, latency-bound
// lanes.  When every attempt is decided after a bounded number of bytes, the
// whole loop is itself a deterministic transducer over the raw byte stream:
//
//   state    = the bytes consumed since the start of the still-undecided
//              attempt (the "pending" string; root = empty);
//   on byte c: pending += c, then resolve from the left exactly as the line
//              loop would — decided success: emit its output, drop what it
//              consumed (or, if it consumed nothing, emit one raw byte: the NFT
//              empty-match rule, trre_nft.c:782-785); decided failure: emit one
//              raw byte; undecided: stop — what is left is the next state;
//   on '\n'  : the rest of the line is known to be empty, so everything pending
//              is resolved (plus the NFT's extra attempt on the empty tail,
//              trre_nft.c:788), '\n' is emitted and the state returns to root;
//   on NUL   : like '\n', then a SKIP state swallows the rest of the record
//              (C-string semantics, trre_nft.c:780 / trre_dft.c:1277,1118).
//
// One table lookup per input byte, no rollback, no special cases in the kernel.
// "Decided" is engine specific and exact: the deterministic engine accepts at
// the first final state (trre_dft.c:1120-1125); the backtracking engine takes the
// first path in priority order, so an attempt is undecided as long as a path
// that is still waiting for input precedes every accepting one
// (trre_nft.c:593-657).  Patterns whose attempts need unbounded look-ahead
// (loops before a decision) make the state set infinite; the builder then gives
// up and the launcher uses the general tile kernels instead.
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <unordered_map>

#include "device_blob.hpp"
#include "front.hpp"
#include "stream_pack.hpp"

namespace trre {
namespace {

struct Outcome {
    enum Kind { Undecided, Fail, Accept, Diverge } kind = Fail;      // Diverge: the reference does not return from this attempt
    std::string out;
    size_t consumed = 0;
};

using GiveUp = StreamGiveUp;   // the pattern does not fold into a bounded stream table

class AttemptModel {
public:
    virtual ~AttemptModel() {}
    virtual Outcome attempt(const std::string& w, bool at_eol) const = 0;
    virtual bool tries_empty_tail() const = 0;   // the NFT's extra attempt at end of line
    // bytes the automaton can consume at all; every other byte behaves like any other such byte
    virtual void alphabet(bool (&used)[256]) const = 0;
};

// ---- deterministic engine: infer_dft, trre_dft.c:1110-1196 --------------------------------
class DftModel : public AttemptModel {
public:
    explicit DftModel(const Dft& d) : d_(d) {}
    Outcome attempt(const std::string& w, bool at_eol) const override {
        Outcome r;
        int32_t s = 0;
        for (size_t i = 0; i < w.size(); ++i) {
            const DftEdge& e = d_.st[s].edge[(uint8_t)w[i]];
            if (e.to == kEdgeDiverge) { r.kind = Outcome::Diverge; r.out.clear(); return r; }
            if (e.to < 0) { r.kind = Outcome::Fail; r.out.clear(); return r; }
            r.out += e.out;
            s = e.to;
            if (d_.st[s].final) {
                r.kind = Outcome::Accept;
                r.out += d_.st[s].final_out;
                r.consumed = i + 1;
                return r;
            }
        }
        r.out.clear();
        r.kind = at_eol ? Outcome::Fail : Outcome::Undecided;
        return r;
    }
    bool tries_empty_tail() const override { return false; }   // never accepts: the start state is not final
    void alphabet(bool (&used)[256]) const override {
        for (int c = 0; c < 256; ++c) used[c] = false;
        for (const DftState& st : d_.st)
            for (int c = 0; c < 256; ++c)
                if (st.edge[c].to >= 0 || st.edge[c].to == kEdgeDiverge) used[c] = true;
    }

private:
    const Dft& d_;
};

// ---- backtracking engine: infer_backtrack, trre_nft.c:593-657 ------------------------------
class NftModel : public AttemptModel {
public:
    explicit NftModel(const Nft& n) : n_(n) {
        // whole-build search budget: grows with the automaton (a 1000-key dictionary needs ~5e8 steps)
        budget_ = std::min<uint64_t>(1500000000ull, std::max<uint64_t>(30000000ull, 60000ull * n.st.size()));
    }
    Outcome attempt(const std::string& w, bool at_eol) const override {
        struct Item { int32_t s; size_t i, o; };
        std::vector<Item> stack;
        std::string out;
        Outcome r;
        int32_t s = n_.start;
        size_t i = 0, o = 0, steps = 0;
        while (!stack.empty() || s >= 0) {
            if (++steps > 2000000 || ++work_ > budget_) throw GiveUp();   // per-attempt and whole-build budgets
            if (s < 0) {
                s = stack.back().s; i = stack.back().i; o = stack.back().o;
                stack.pop_back();
                if (s < 0) continue;
            }
            const NState& st = n_.st[s];
            switch (st.kind) {
            case NKind::Cons:
                if (i < w.size()) {
                    if (st.val == (uint8_t)w[i]) { ++i; s = st.a; } else s = -1;
                } else if (at_eol) {
                    s = -1;
                } else {
                    // this path outranks everything explored later and needs input we
                    // have not seen: the attempt cannot be decided yet
                    r.kind = Outcome::Undecided;
                    return r;
                }
                break;
            case NKind::Prod:
                if (out.size() <= o) out.resize(o + 1);
                out[o++] = (char)st.val;
                if (o > (1u << 16)) throw GiveUp();
                s = st.a;
                break;
            case NKind::Split:
            case NKind::SplitNg:
                if (stack.size() >= 65536) throw GiveUp();   // "stack max capacity reached" in the reference
                stack.push_back(Item{n_.second(s), i, o});
                s = n_.first(s);
                break;
            case NKind::Join:
                s = st.a;
                break;
            case NKind::Final: {
                r.kind = Outcome::Accept;
                r.out.assign(out.data(), o);
                size_t nul = r.out.find('\0');                // fputs stops at a NUL
                if (nul != std::string::npos) r.out.resize(nul);
                r.consumed = i;
                return r;
            }
            }
        }
        r.kind = Outcome::Fail;
        return r;
    }
    bool tries_empty_tail() const override { return true; }
    void alphabet(bool (&used)[256]) const override {
        for (int c = 0; c < 256; ++c) used[c] = false;
        for (const NState& st : n_.st)
            if (st.kind == NKind::Cons) used[st.val] = true;
    }

private:
    const Nft& n_;
    mutable uint64_t work_ = 0;
    uint64_t budget_ = 30000000ull;
};

// ---- the same search over the follow lists of nft_tables.cpp ---------------------------------
// A node's list names, in the reference's depth-first priority order, the consuming nodes / FINAL reachable from it
// through epsilon states, with the bytes produced on the way; the first occurrence of a target stands for all (a search
// from the same node at the same position fails again).  Walking the lists instead of the states gives the same
// outcome, and with the lists indexed by the next input byte an attempt on an alternation of thousands of keys visits
// only the keys that go on with that byte (the state-by-state walk above visits every SPLIT of the chain: a 1000-key
// dictionary needed 5e8 steps, larger ones ran out of budget).  TRRE_NFT_FOLD=states selects the walk above, =both
// builds the tables both ways and compares them.
class NodeModel : public AttemptModel {
public:
    explicit NodeModel(const NftNodes& n) : n_(n), index_(n.follow.size()) {
        size_t entries = 0;
        for (const auto& l : n.follow) entries += l.size();
        budget_ = std::min<uint64_t>(1500000000ull, std::max<uint64_t>(30000000ull, 60000ull * entries));
    }
    Outcome attempt(const std::string& w, bool at_eol) const override {
        struct Frame { uint32_t list; size_t next, i, o; bool muted; };    // next: where to go on in the list (or in its index for this byte)
        std::vector<Frame> stack;
        std::string out;
        Outcome r;
        size_t steps = 0;
        auto accept = [&](const Frame& f, const NodeFollow& e) {
            Outcome a;
            a.kind = Outcome::Accept;
            out.resize(f.o);
            if (!f.muted) out += e.out;                                    // (cut at a NUL already: fputs stops there)
            a.out = out;
            a.consumed = f.i;
            return a;
        };
        stack.push_back(Frame{(uint32_t)n_.node.size(), 0, 0, 0, false});
        while (!stack.empty()) {
            Frame& f = stack.back();
            const std::vector<NodeFollow>& list = n_.follow[f.list];
            const NodeFollow* e = nullptr;
            if (f.i >= w.size()) {
                // no input left: a node would need a byte we have not seen (at the end of the line: it fails), FINAL accepts
                for (; f.next < list.size() && !e; ++f.next) {
                    if (++steps > 2000000 || ++work_ > budget_) throw GiveUp();
                    const NodeFollow& x = list[f.next];
                    if (x.target == kNodeFinal || x.target == kNodeDiverge) e = &x;
                    else if (!at_eol) { r.kind = Outcome::Undecided; return r; }
                }
            } else {
                const uint8_t c = (uint8_t)w[f.i];
                if (++steps > 2000000 || ++work_ > budget_) throw GiveUp();
                if (list.size() <= kShortList) {
                    for (; f.next < list.size() && !e; ++f.next) {
                        const NodeFollow& x = list[f.next];
                        if (x.target == kNodeFinal || x.target == kNodeDiverge || n_.node[x.target].reads(c)) e = &x;
                    }
                } else {
                    const std::vector<uint32_t>& idx = entries_for(f.list, c);
                    if (f.next < idx.size()) e = &list[idx[f.next++]];
                }
            }
            if (!e) { stack.pop_back(); continue; }
            if (e->target == kNodeDiverge) throw GiveUp();                 // the reference's search does not come back from here
            if (e->target == kNodeFinal) return accept(f, *e);
            // the node consumes w[f.i]
            Frame g{e->target, 0, f.i + 1, f.o, f.muted};
            out.resize(f.o);
            if (!g.muted) {
                out += e->out;
                if (e->mute) g.muted = true;
                else if (n_.node[e->target].echo) out.push_back(w[f.i]);
            }
            g.o = out.size();
            if (g.o > (1u << 16) || stack.size() >= 65536) throw GiveUp();
            stack.push_back(g);
        }
        r.kind = Outcome::Fail;
        return r;
    }
    bool tries_empty_tail() const override { return true; }
    void alphabet(bool (&used)[256]) const override {
        for (int c = 0; c < 256; ++c) {
            used[c] = false;
            for (const NftNodes::Node& nd : n_.node)
                if (nd.reads((uint8_t)c)) { used[c] = true; break; }
        }
    }

private:
    static constexpr size_t kShortList = 8;
    using ByByte = std::array<std::vector<uint32_t>, 256>;
    // the entries of a long list that can be taken on byte c, in list order
    const std::vector<uint32_t>& entries_for(uint32_t l, uint8_t c) const {
        if (!index_[l]) {
            index_[l].reset(new ByByte());
            const std::vector<NodeFollow>& list = n_.follow[l];
            for (uint32_t k = 0; k < list.size(); ++k)
                for (int b = 0; b < 256; ++b)
                    if (list[k].target == kNodeFinal || list[k].target == kNodeDiverge || n_.node[list[k].target].reads((uint8_t)b))
                        (*index_[l])[b].push_back(k);
        }
        return (*index_[l])[c];
    }
    const NftNodes& n_;
    mutable std::vector<std::unique_ptr<ByByte>> index_;
    mutable uint64_t work_ = 0;
    uint64_t budget_ = 30000000ull;
};

class StreamBuilder {
public:
    StreamBuilder(const AttemptModel& m, const StreamLimits& lim) : m_(m), lim_(lim) {}

    StreamTables run() {
        m_.alphabet(used_);
        intern("");                       // 0 = root
        skip_ = (uint32_t)names_.size();  // 1 = SKIP: swallow the rest of the record (after a NUL; also a
        names_.push_back(std::string("\0skip", 5));   //     lane's state before its first line start)
        rows_.emplace_back();
        done_ = (uint32_t)names_.size();  // 2 = DONE: absorbing, silent (a lane that has finished its lines)
        names_.push_back(std::string("\0done", 5));
        rows_.emplace_back();
        for (uint32_t s = 0; s < names_.size(); ++s) {
            rows_[s].resize(256);
            if (s == skip_) {
                for (int c = 0; c < 256; ++c) rows_[s][c] = Cell{c == '\n' ? 0u : skip_, std::string(), false, c == '\n'};
                continue;
            }
            if (s == done_) {
                for (int c = 0; c < 256; ++c) rows_[s][c] = Cell{done_, std::string(), false, false};
                continue;
            }
            const std::string w = names_[s];
            // a byte no state can consume ends every pending attempt and is copied raw: all such
            // bytes share one transition (its output ends with the byte itself -> copy flag)
            int other = -1;
            for (int c = 0; c < 256; ++c) {
                if (c != 0 && c != '\n' && !used_[c]) {
                    if (other < 0) { other = c; rows_[s][c] = transition(w, c); }
                    else rows_[s][c] = rows_[s][other];
                } else {
                    rows_[s][c] = transition(w, c);
                }
            }
        }
        first_copy_ = (uint32_t)names_.size();
        spread_long_outputs();
        return pack();
    }

private:
    using Cell = StreamCell;

    uint32_t intern(const std::string& w) {
        auto hit = index_.find(w);
        if (hit != index_.end()) return hit->second;
        if (w.size() > lim_.max_pending) return kOverflow;     // bounded fold: the caller marks the transition
        if (names_.size() >= lim_.max_states) throw GiveUp();
        uint32_t id = (uint32_t)names_.size();
        names_.push_back(w);
        rows_.emplace_back();
        index_.emplace(w, id);
        return id;
    }

    // resolve pending attempts from the left, exactly like the scan line loop
    std::string resolve(std::string w, bool at_eol, std::string& out, bool& diverges) {
        while (!w.empty()) {
            Outcome r = m_.attempt(w, at_eol);
            if (r.kind == Outcome::Undecided) break;
            if (r.kind == Outcome::Diverge) { diverges = true; return std::string(); }
            if (r.kind == Outcome::Accept) {
                out += r.out;
                if (r.consumed > 0) { w.erase(0, r.consumed); continue; }
            }
            out.push_back(w[0]);          // no match here (or an empty one): one raw byte
            w.erase(0, 1);
        }
        if (out.size() > lim_.max_out) throw GiveUp();
        return w;
    }

    Cell transition(const std::string& w, int c) {
        Cell cell;
        bool diverges = false;
        if (c == '\n' || c == 0) {
            std::string rest = resolve(w, true, cell.out, diverges);
            if (diverges) { cell.out.clear(); cell.diverge = true; cell.next = skip_; return cell; }
            if (!rest.empty()) throw GiveUp();           // cannot happen: at end of line everything is decided
            if (m_.tries_empty_tail()) {
                Outcome r = m_.attempt(std::string(), true);
                if (r.kind == Outcome::Accept) cell.out += r.out;
            }
            cell.out.push_back('\n');
            cell.next = c == 0 ? skip_ : 0u;
            cell.eol = c == '\n';          // record end (a NUL only ends the line's content)
            return cell;
        }
        std::string rest = resolve(w + (char)c, false, cell.out, diverges);
        if (diverges) { cell.out.clear(); cell.diverge = true; cell.next = skip_; return cell; }
        cell.next = intern(rest);
        if (cell.next == kOverflow) {
            // An attempt that is still undecided after max_pending bytes (a long run under a greedy loop).
            // The table stops following it: the transition swallows the rest of the record and is marked,
            // a launch that takes it is void and the runtime falls back to the tile kernels.
            cell.next = skip_;
            cell.out.clear();
            cell.ovf = true;
            bounded_ = true;
            return cell;
        }
        // express "... then the input byte itself" through the copy flag so that
        // bytes the pattern never mentions share one column
        // (an exact re-encoding of this cell: the last emitted byte equals the byte read)
        if (!cell.out.empty() && (uint8_t)cell.out.back() == (uint8_t)c) {
            cell.out.pop_back();
            cell.copy_c = true;
        }
        return cell;
    }

    // An entry holds 4 output bytes inline; longer outputs are pooled and cost the kernels a slow path
    // (with a dictionary, one lane or another of a wave takes it on most steps), and texts of more than
    // 8 bytes a very slow one.  A completed match that emits 5..12 literal bytes is therefore split: it
    // emits the first 4 and enters a copy of the root state whose every transition first emits the rest
    // (at most 8 bytes: what the kernels append without leaving their fast paths).  The copy counts the
    // owed bytes as pending, so the length bookkeeping of pack() still holds; transitions that end a
    // record or copy the input byte are left alone (a record's output must be complete when it ends),
    // and so are flushes of a long failed prefix (one per deep state and byte, rare at run time).
    void spread_long_outputs() {
        std::map<std::string, uint32_t> made;
        for (uint32_t s = 0; s < (uint32_t)names_.size(); ++s) {       // (the copies are visited too: they may owe again)
            if (s == skip_ || s == done_) continue;
            for (int c = 0; c < 256; ++c) {
                if (rows_[s][c].out.size() < 5 || rows_[s][c].out.size() > 12 || rows_[s][c].copy_c || rows_[s][c].eol ||
                    rows_[s][c].next != 0u)
                    continue;
                const std::string rest = rows_[s][c].out.substr(4);
                auto hit = made.find(rest);
                if (hit == made.end()) {
                    if (names_.size() >= 2 * lim_.max_states) return;      // no room: the remaining ones stay pooled
                    const uint32_t id = (uint32_t)names_.size();
                    names_.push_back(rest);                                 // (only its length is used from here on)
                    std::vector<Cell> row = rows_[0];
                    for (Cell& y : row) y.out = rest + y.out;
                    rows_.push_back(std::move(row));
                    hit = made.emplace(rest, id).first;
                }
                Cell& x = rows_[s][c];
                if (s < first_copy_) undo_.push_back(Undo{s, c, x});
                x.out.resize(4);
                x.next = hit->second;
            }
        }
    }

    // byte classes (identical columns over all states), then the generic packer
    StreamTables pack() {
        const uint32_t n = (uint32_t)names_.size();
        StreamPackInput in;
        in.pending_len.resize(n);
        for (uint32_t s = 0; s < n; ++s) in.pending_len[s] = (s == skip_ || s == done_) ? 0 : (uint32_t)names_[s].size();
        std::map<std::vector<std::string>, uint32_t> col_index;
        std::vector<int> rep;
        std::array<uint8_t, 256> cls{};
        for (int c = 0; c < 256; ++c) {
            std::vector<std::string> key;
            key.reserve(n);
            for (uint32_t s = 0; s < n; ++s) {
                const Cell& x = rows_[s][c];
                key.push_back(std::to_string(x.next) + (x.copy_c ? "C" : "-") + (x.eol ? "E" : "-") + (x.ovf ? "O" : "-") + (x.diverge ? "D" : "-") + x.out);
            }
            auto hit = col_index.find(key);
            if (hit == col_index.end()) {
                hit = col_index.emplace(std::move(key), (uint32_t)rep.size()).first;
                rep.push_back(c);
            }
            cls[c] = (uint8_t)hit->second;
        }
        in.rows.resize(n);
        for (uint32_t s = 0; s < n; ++s) {
            in.rows[s].reserve(rep.size());
            for (int c : rep) in.rows[s].push_back(rows_[s][c]);
        }
        for (int c : rep) in.col_kind.push_back(c == 0 ? kColNul : (c == '\n' ? kColNewline : kColPlain));
        in.skip = skip_;
        in.done = done_;
        in.bounded = bounded_;
        StreamTables t = pack_stream_tables(in);
        t.cls = cls;
        build_mapgen(t);
        // (first without "a literal in front of the root row" as a way to describe a state: such a state is owed AND has slots of
        // its own, which the 32-bit mark form cannot express; if the comb does not come out with that form, as before)
        StreamTables t2 = t;
        build_fallback(t2, in, cls, false);
        if (!t2.fb_mark4_ok) { t2 = t; build_fallback(t2, in, cls, true); }
        return t2;
    }

    // The memoryless form (front.hpp: StreamTables::mg; map_block.hpp): the root row as it was before spread_long_outputs, by raw byte — if
    // every cell of it leads back to the root (a NUL may cut its record short: SKIP), nothing diverges or overflows and no text exceeds 8 bytes.
    void build_mapgen(StreamTables& t) const {
        std::vector<Cell> root = rows_[0];
        for (const Undo& u : undo_)
            if (u.s == 0) root[u.c] = u.cell;
        std::vector<uint32_t> v(256 * 4, 0);
        uint32_t longest = 0;
        for (int c = 0; c < 256; ++c) {
            const Cell& x = root[c];
            if (x.ovf || x.diverge) return;
            const bool cut = c == 0 && x.next == skip_;
            if (x.next != 0u && !cut) return;
            std::string text = x.out;
            if (x.copy_c) text.push_back((char)c);
            if (text.size() > 8) return;
            uint64_t bytes = 0;
            for (size_t k = 0; k < text.size(); ++k) bytes |= (uint64_t)(uint8_t)text[k] << (8 * k);
            v[4 * c] = (uint32_t)bytes;
            v[4 * c + 1] = (uint32_t)(bytes >> 32);
            v[4 * c + 2] = (uint32_t)text.size() | (cut ? 0x80u : 0u);
            longest = std::max<uint32_t>(longest, (uint32_t)text.size());
        }
        if (!longest) return;
        t.mg = std::move(v);
        t.mg_max = longest;
    }

    // Fallback form of a large table (front.hpp, StreamTables::fb_*), built from the rows as they were before
    // spread_long_outputs (replacement texts are owed whole here).  Which classes of a state are exceptions to "the
    // first bytes of the pending string, then the row of the state of its last byte (or of the root)" is found by
    // comparing the cells, never assumed.
    struct FbCell {
        uint32_t next = 0, n = 0;
        bool cc = false, nl = false, eol = false;
        int esc = -1;                     // index of an escape record: the cell's output spelled out
    };
    void build_fallback(StreamTables& t, const StreamPackInput& in, const std::array<uint8_t, 256>& cls, bool allow_literal_plans) {
        const uint32_t n0 = first_copy_, C = t.n_cls;
        const bool dbg = getenv("TRRE_TRACE") != nullptr;             // (why a table gets no fallback form, on stderr)
        if (t.g16_ok || bounded_ || C > 31 || n0 > 8000 || n0 < 64) return;   // small tables have the 16-byte form; 5-bit classes, 13-bit ids
        std::vector<std::vector<Cell>> rows(in.rows.begin(), in.rows.begin() + n0);
        for (const Undo& u : undo_) rows[u.s][cls[u.c]] = u.cell;
        auto pending = [&](uint32_t s) { return (s == skip_ || s == done_) ? std::string() : names_[s]; };
        for (uint32_t s = 0; s < n0; ++s)
            if (pending(s).size() > 7) return;                                 // the kernels keep 8 bytes of history; a transition appends at most 9
        // owed texts and literal prefixes, each with the number of input bytes it stands for (the copy form reads the
        // automaton as net edits: front.hpp) — the same text for another number of bytes is another literal
        std::map<std::pair<std::string, uint32_t>, uint32_t> lits;
        std::vector<std::string> lit_text;
        std::vector<uint32_t> lit_kb;
        auto lit_id = [&](const std::string& L, uint32_t kb) {
            auto hit = lits.find(std::make_pair(L, kb));
            if (hit == lits.end()) {
                hit = lits.emplace(std::make_pair(L, kb), (uint32_t)lit_text.size()).first;
                lit_text.push_back(L);
                lit_kb.push_back(kb);
            }
            return hit->second;
        };
        std::map<std::pair<std::string, uint32_t>, uint32_t> owed;            // (text, bytes it stands for) -> owed-text state (ids from n0 on)
        std::vector<std::string> owed_text;
        std::vector<uint32_t> owed_kb;
        std::vector<std::string> esc_text;
        std::vector<bool> esc_cc;
        std::vector<uint32_t> esc_net;                                         // [7:0] input bytes the text stands for, [15:8] how far back the first lies
        bool copy_ok = true;
        size_t hot_escapes = 0;
        uint32_t class_size[32] = {}, class_byte[32] = {};
        for (int c = 0; c < 256; ++c) { ++class_size[cls[c]]; class_byte[cls[c]] = (uint32_t)c; }
        // (the builder writes "ends with the input byte" as a flag: spelled out where the byte is known, so that equal
        // outputs compare equal)
        for (uint32_t s = 0; s < n0; ++s)
            for (uint32_t k = 0; k < C; ++k)
                if (rows[s][k].copy_c && class_size[k] == 1) { rows[s][k].out.push_back((char)class_byte[k]); rows[s][k].copy_c = false; }
        // a cell as an entry: leading bytes of the state's pending string, then maybe the input byte or '\n'; a whole
        // replacement text followed by the root state is owed
        auto decompose = [&](uint32_t s, uint32_t k, FbCell& y) -> bool {
            const Cell& x = rows[s][k];
            if (x.ovf || x.diverge || x.next >= n0) return false;
            const std::string w = pending(s);
            size_t n = 0;
            while (n < x.out.size() && n < w.size() && x.out[n] == w[n]) ++n;
            std::string lit = x.out.substr(n);
            bool cc = x.copy_c;
            if (lit.size() == 1 && class_size[k] == 1 && (uint8_t)lit[0] == class_byte[k]) { lit.clear(); cc = true; }   // the input byte itself
            y.next = x.next; y.n = (uint32_t)n; y.cc = cc; y.nl = false; y.eol = x.eol;
            if (lit.empty()) return true;
            if (lit == "\n" && !cc) { y.nl = true; return true; }
            if (x.next == 0 && !cc && !x.eol && lit.size() <= 8) {
                const uint32_t kb = (uint32_t)(w.size() - n + 1);              // the key: what is left of the pending string, and this byte
                auto hit = owed.find(std::make_pair(lit, kb));
                if (hit == owed.end()) {
                    hit = owed.emplace(std::make_pair(lit, kb), (uint32_t)owed_text.size()).first;
                    owed_text.push_back(lit);
                    owed_kb.push_back(kb);
                }
                y.next = n0 + hit->second;
                return true;
            }
            // anything else is spelled out in an escape record (rare at run time: the kernels leave their fast path for it)
            y.n = 0; y.cc = false; y.nl = false;
            y.esc = (int)esc_text.size();
            if (n == 0 && x.next == 0) ++hot_escapes;                          // a completed key with a long replacement: every hit would leave the fast path
            esc_text.push_back(x.out);
            esc_cc.push_back(x.copy_c);
            {
                // as a net edit: the text stands for the first bytes of (pending + this byte) that are not pending afterwards;
                // with the copy flag the byte itself passes through behind the text
                const size_t left = pending(x.next).size();
                const size_t kb = x.copy_c ? w.size() : w.size() + 1 - left;
                if ((x.copy_c && left != 0) || w.size() + 1 < left || kb == 0 || kb > 15 || w.size() > 8) {
                    if (dbg && copy_ok) fprintf(stderr, "copy form: escape of state '%s' class %u: copy %d left %zu kb %zu\n", w.c_str(), k, (int)x.copy_c, left, kb);
                    copy_ok = false;
                }
                esc_net.push_back((uint32_t)kb | (uint32_t)w.size() << 8);
            }
            return true;
        };
        // the same classification without creating anything (the copy form's check below)
        auto decompose_again = [&](uint32_t s, uint32_t k, FbCell& y) -> bool {
            const Cell& x = rows[s][k];
            const std::string w = pending(s);
            size_t n = 0;
            while (n < x.out.size() && n < w.size() && x.out[n] == w[n]) ++n;
            std::string lit = x.out.substr(n);
            bool cc = x.copy_c;
            if (lit.size() == 1 && class_size[k] == 1 && (uint8_t)lit[0] == class_byte[k]) { lit.clear(); cc = true; }
            y.next = x.next; y.esc = -1;
            if (lit.empty() || (lit == "\n" && !cc)) return true;
            if (x.next == 0 && !cc && !x.eol && lit.size() <= 8) { y.next = n0; return true; }
            y.esc = 0;
            return true;
        };
        auto same = [](const Cell& a, const Cell& b, const std::string& P) {
            return a.next == b.next && a.copy_c == b.copy_c && a.eol == b.eol && a.ovf == b.ovf && a.diverge == b.diverge &&
                   a.out.size() == P.size() + b.out.size() && a.out.compare(0, P.size(), P) == 0 &&
                   a.out.compare(P.size(), std::string::npos, b.out) == 0;
        };
        // dense: root, SKIP, DONE, one-byte pending strings; every other state is "a prefix, then the row of a dense state":
        // the first bytes of its pending string in front of the row of the state of its last byte or of the root, or a
        // literal text in front of the root row (a completed key that waits for a longer one to fail)
        struct Plan { bool dense = false, literal = false; uint32_t f = 0; std::string P; std::vector<uint32_t> exc; };
        std::vector<Plan> plan(n0);
        uint32_t n_dense = 0;
        std::vector<uint32_t> row_at(n0, 0);
        for (uint32_t s = 0; s < n0; ++s)
            if (s == 0 || s == skip_ || s == done_ || names_[s].size() == 1) { plan[s].dense = true; row_at[s] = C * n_dense++; }
        std::vector<std::vector<FbCell>> fr(n0, std::vector<FbCell>(C));
        for (uint32_t s = 0; s < n0; ++s)
            if (plan[s].dense)
                for (uint32_t k = 0; k < C; ++k)
                    if (!decompose(s, k, fr[s][k])) return;
        size_t n_exc = 0;
        for (uint32_t s = 0; s < n0; ++s) {
            if (plan[s].dense) continue;
            const std::string& w = names_[s];
            std::vector<Plan> tries;
            for (size_t keep = 0; keep <= 1; ++keep) {
                auto hit = index_.find(w.substr(w.size() - keep));
                if (hit == index_.end() || hit->second >= n0 || !plan[hit->second].dense) continue;
                Plan p;
                p.f = hit->second;
                p.P = w.substr(0, w.size() - keep);
                tries.push_back(std::move(p));
            }
            for (uint32_t k = 0; k < C && allow_literal_plans; ++k) {          // a literal in front of the root row: read it off any cell
                const Cell& x = rows[s][k];
                const Cell& r = rows[0][k];
                if (x.out.size() <= r.out.size() || x.out.size() - r.out.size() > 8) continue;
                Plan p;
                p.literal = true;
                p.f = 0;
                p.P = x.out.substr(0, x.out.size() - r.out.size());
                tries.push_back(std::move(p));
                break;
            }
            int best = -1;
            for (size_t i = 0; i < tries.size(); ++i) {
                Plan& p = tries[i];
                for (uint32_t k = 0; k < C; ++k)
                    if (!same(rows[s][k], rows[p.f][k], p.P)) p.exc.push_back(k);
                if (best < 0 || p.exc.size() < tries[best].exc.size()) best = (int)i;
            }
            if (best < 0) return;
            plan[s] = tries[best];
            for (uint32_t k : plan[s].exc)
                if (!decompose(s, k, fr[s][k])) return;
            n_exc += plan[s].exc.size();
        }
        const uint32_t n_all = n0 + (uint32_t)owed_text.size();
        for (size_t i = 0; i < owed_text.size(); ++i) lit_id(owed_text[i], owed_kb[i]);
        for (uint32_t s = 0; s < n0; ++s) if (plan[s].literal) lit_id(plan[s].P, (uint32_t)names_[s].size());
        if (n_all > 16000 || lit_text.size() > 4095 || esc_text.size() > 4095 || hot_escapes > 16) return;   // escapes are for the odd cell, not for every completed key
        // row displacement: every state with exceptions gets a base of its own such that its slots base + k are free
        // (first fit, the states with the most exceptions first); states without exceptions share the tail of the array,
        // where no slot is ever owned
        std::vector<std::vector<uint32_t>> exc(n0);
        for (uint32_t s = 0; s < n0; ++s) {
            if (plan[s].dense) for (uint32_t k = 0; k < C; ++k) exc[s].push_back(k);
            else exc[s] = plan[s].exc;
        }
        std::vector<uint32_t> order;
        for (uint32_t s = 0; s < n0; ++s) if (!exc[s].empty()) order.push_back(s);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return exc[x].size() > exc[y].size(); });
        std::vector<char> used, is_base;
        std::vector<uint32_t> base(n0, 0);
        size_t probes = 0;
        for (uint32_t s : order) {
            for (size_t b0 = 0;; ++b0) {
                if (++probes > 100000000u || b0 > 16383) return;              // (not that kind of table: give the form up rather than search on)
                if (used.size() < b0 + 32) { used.resize(b0 + 32, 0); is_base.resize(b0 + 32, 0); }
                if (is_base[b0]) continue;
                bool fits = true;
                for (uint32_t k : exc[s]) if (used[b0 + k]) { fits = false; break; }
                if (!fits) continue;
                base[s] = (uint32_t)b0;
                is_base[b0] = 1;
                for (uint32_t k : exc[s]) used[b0 + k] = 1;
                break;
            }
        }
        size_t top = used.size();
        while (top > 0 && !used[top - 1]) --top;
        const uint32_t shared = (uint32_t)top;                                 // base of the states without exceptions
        const size_t n_slots = top + 32;
        const size_t lds = 256 + n_slots * 8 + lit_text.size() * 8;
        if (dbg)
            fprintf(stderr, "fallback form: %u states + %zu owed texts, %u classes, %u dense, %zu exceptions in %zu slots, %zu literals, %zu escapes, %zu bytes of LDS\n",
                    n0, owed_text.size(), C, n_dense, n_exc, n_slots, lit_text.size(), esc_text.size(), lds);
        if (n_slots > 16383 || lds > kFallbackLdsBytes) return;
        for (uint32_t s = 0; s < n0; ++s) if (exc[s].empty()) base[s] = shared;
        // descriptors and the bits an entry carries about its target state
        auto desc_of = [&](uint32_t s) -> uint32_t {
            if (s >= n0) return shared | base[0] << 14 | (uint32_t)(owed_text[s - n0].size() - 1) << 28 | 1u << 31;
            if (plan[s].dense) return base[s] | base[s] << 14;
            if (plan[s].literal) return base[s] | base[0] << 14 | (uint32_t)(plan[s].P.size() - 1) << 28 | 1u << 31;
            return base[s] | base[plan[s].f] << 14 | (uint32_t)plan[s].P.size() << 28;
        };
        auto about_of = [&](uint32_t s) -> uint32_t {
            if (s >= n0) return lit_id(owed_text[s - n0], owed_kb[s - n0]);
            if (!plan[s].dense && plan[s].literal) return lit_id(plan[s].P, (uint32_t)names_[s].size());
            return (uint32_t)pending(s).size();
        };
        t.fb_comb.assign(n_slots, (uint64_t)kFbNoTag << 32);
        t.fb_lit.assign(lit_text.size(), 0);
        for (size_t i = 0; i < lit_text.size(); ++i)
            for (size_t b2 = 0; b2 < lit_text[i].size(); ++b2) t.fb_lit[i] |= (uint64_t)(uint8_t)lit_text[i][b2] << (8 * b2);
        std::vector<std::pair<uint32_t, int>> esc_at;                          // (slot, escape record)
        for (uint32_t s = 0; s < n0; ++s) {
            for (uint32_t k : exc[s]) {
                const FbCell& y = fr[s][k];
                uint32_t hi = base[s] | about_of(y.next) << 20 | (y.eol ? kFbEol : 0u);
                if (y.esc >= 0) { hi |= kFbCc | kFbNl; esc_at.emplace_back(base[s] + k, y.esc); }
                else hi |= y.n << 14 | (y.cc ? kFbCc : 0u) | (y.nl ? kFbNl : 0u);
                t.fb_comb[base[s] + k] = (uint64_t)desc_of(y.next) | (uint64_t)hi << 32;
            }
        }
        // escape records in slot order: {offset of the text in fb_pool, its length, 1 = then the input byte, 0}
        std::sort(esc_at.begin(), esc_at.end());
        for (auto& se : esc_at) {
            t.fb_esc_slot.push_back(se.first);
            t.fb_esc.push_back((uint32_t)t.fb_pool.size());
            t.fb_esc.push_back((uint32_t)esc_text[se.second].size());
            t.fb_esc.push_back(esc_cc[se.second] ? 1u : 0u);
            t.fb_esc.push_back(esc_net[se.second]);
            t.fb_pool.insert(t.fb_pool.end(), esc_text[se.second].begin(), esc_text[se.second].end());
        }
        const uint32_t starts[3] = {0u, skip_, done_};
        for (int i = 0; i < 3; ++i) { t.fb_start[i][0] = desc_of(starts[i]); t.fb_start[i][1] = about_of(starts[i]) << 20; }
        t.fb_states = n_all;
        t.fb_dense = n_dense;
        t.fb_ok = true;
        // The copy form: is every transition that is neither an owed text nor an escape a pure pass-through of input bytes?
        // Checked on the cells themselves: what a cell emits must be exactly the bytes of (pending + this byte) that are no
        // longer pending afterwards.  (SKIP swallows what it reads and a NUL ends a line early: the kernels leave the copy
        // form on a NUL; DONE is silent.)
        for (uint32_t s = 0; s < n0 && copy_ok; ++s) {
            if (s == skip_ || s == done_) continue;
            const std::string w = pending(s);
            for (uint32_t k = 0; k < C && copy_ok; ++k) {
                if (in.col_kind[k] == kColNul) continue;
                const Cell& x = rows[s][k];
                if (x.ovf || x.diverge || x.next >= n0) { copy_ok = false; break; }
                const std::string rest = pending(x.next);
                FbCell y;
                if (!decompose_again(s, k, y)) { copy_ok = false; break; }
                if (y.esc >= 0 || y.next >= n0) continue;                      // an edit: spelled out, or owed
                // (a class of several bytes: any of them; the copy flag stands for the byte)
                const std::string wc = w + (char)class_byte[k];
                const std::string full = x.out + (x.copy_c ? std::string(1, (char)class_byte[k]) : std::string());
                if (wc.size() < rest.size() || full != wc.substr(0, wc.size() - rest.size()) || wc.substr(wc.size() - rest.size()) != rest) {
                    if (dbg) fprintf(stderr, "copy form: state '%s' class %u emits '%s' and leaves '%s'\n", w.c_str(), k, full.c_str(), rest.c_str());
                    copy_ok = false;
                }
            }
        }
        for (size_t i = 0; i < lit_text.size(); ++i)
            if (lit_kb[i] > 15 || lit_text[i].size() > 8) {
                if (dbg && copy_ok) fprintf(stderr, "copy form: literal '%s' stands for %u bytes\n", lit_text[i].c_str(), lit_kb[i]);
                copy_ok = false;
            }
        if (copy_ok) {
            t.fb_lit_meta.resize(lit_text.size());
            for (size_t i = 0; i < lit_text.size(); ++i) t.fb_lit_meta[i] = (uint16_t)(lit_text[i].size() | lit_kb[i] << 8);
            t.fb_copy_ok = true;
        }
        // The mark form (front.hpp): 32-bit entries for the copy form's first pass
        bool any_literal_state = false;
        for (uint32_t s = 0; s < n0; ++s) any_literal_state = any_literal_state || (!plan[s].dense && plan[s].literal);
        if (dbg) fprintf(stderr, "mark form: copy %d literal states %d dense %u slots %zu lits %zu\n", (int)t.fb_copy_ok, (int)any_literal_state, n_dense, n_slots, lit_text.size());
        if (t.fb_copy_ok && !any_literal_state && n_dense <= 128 && n_slots + lit_text.size() + 32 <= 16383) {
            std::vector<uint32_t> dense_idx(n0, 0);
            t.fb_dense_base.clear();
            for (uint32_t s = 0; s < n0; ++s)
                if (plan[s].dense) { dense_idx[s] = (uint32_t)t.fb_dense_base.size(); t.fb_dense_base.push_back((uint16_t)base[s]); }
            const uint32_t pad = (uint32_t)n_slots;
            auto desc4 = [&](uint32_t s) -> uint32_t {
                if (s >= n0) return (pad + lit_id(owed_text[s - n0], owed_kb[s - n0])) | dense_idx[0] << 14 | 1u << 21;
                if (plan[s].dense) return base[s] | dense_idx[s] << 14;
                return base[s] | dense_idx[plan[s].f] << 14;
            };
            t.fb_comb4.assign(n_slots + lit_text.size() + 32, 124u << 24);
            t.fb_dense4.assign((size_t)t.fb_dense_base.size() * 32, 124u << 24);
            for (uint32_t s = 0; s < n0; ++s) {
                for (uint32_t k : exc[s]) {
                    const FbCell& y = fr[s][k];
                    const uint32_t e = desc4(y.next) | (y.esc >= 0 ? 1u << 22 : 0u) | (y.eol ? 1u << 23 : 0u) | (4u * k) << 24;
                    t.fb_comb4[base[s] + k] = e;
                    if (plan[s].dense) t.fb_dense4[(size_t)dense_idx[s] * 32 + k] = e;
                }
            }
            const uint32_t starts4[3] = {0u, skip_, done_};
            for (int i = 0; i < 3; ++i) t.fb_start4[i] = desc4(starts4[i]);
            t.fb_pad = pad;
            t.fb_mark4_ok = true;
        }
    }
    static constexpr size_t kFallbackLdsBytes = 90 * 1024;   // the tables' share of the 160 KB (the rest: the emit pass's staging rings)
    struct Undo { uint32_t s; int c; Cell cell; };
    std::vector<Undo> undo_;          // the cells spread_long_outputs changed, as they were
    uint32_t first_copy_ = 0;         // states from here on are the "owed bytes" copies of the root row (spread_long_outputs)

    const AttemptModel& m_;
    StreamLimits lim_;
    std::vector<std::string> names_;
    std::vector<std::vector<Cell>> rows_;
    std::unordered_map<std::string, uint32_t> index_;
    uint32_t skip_ = 1, done_ = 2;
    static constexpr uint32_t kOverflow = 0xffffffffu;
    bool bounded_ = false;
    bool used_[256];
};

StreamTables build(const AttemptModel& m, const StreamLimits& lim) {
    try {
        return StreamBuilder(m, lim).run();
    } catch (const GiveUp&) {
        return StreamTables();        // ok == false
    }
}

}  // namespace

StreamTables build_stream_dft(const Dft& dft, const StreamLimits& lim) { return build(DftModel(dft), lim); }
StreamTables build_stream_nft(const Nft& nft, const StreamLimits& lim) { return build(NftModel(nft), lim); }
StreamTables build_stream_nodes(const NftNodes& nodes, const StreamLimits& lim) { return build(NodeModel(nodes), lim); }

}  // namespace trre
