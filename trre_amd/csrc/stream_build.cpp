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
#include <map>
#include <unordered_map>

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
        return t;
    }

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

}  // namespace trre
