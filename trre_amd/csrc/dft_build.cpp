// dft_build.cpp — eager determinisation of the NFT into a string-weighted
// deterministic transducer, and its flattening into device tables.
//
// The reference builds these states lazily, one table miss at a time
// (trre_dft.c:1135-1175); a GPU needs the finished tables, so the same
// subset construction is run here as a worklist over every reachable state and
// every byte.  A determinised state is an ORDERED list of (CONS state, residual
// output) pairs; the identity of a state, the order inside it, the way pending
// output is carried through the epsilon closure and the longest-common-prefix
// factoring all decide which bytes come out, so they follow the reference:
//   closure order + first-writer-wins ......... trre_dft.c:874-907
//   residual inherited across a split .......... trre_dft.c:880-881, 884-885
//   step over all items, then clear marks ...... trre_dft.c:910-923
//   LCP moved onto the edge .................... trre_dft.c:988-1010
//   state identity (count, states, residuals) .. trre_dft.c:1028-1049
//   finality probe with byte 0, first item ..... trre_dft.c:1164-1174
#include <algorithm>
#include <unordered_map>

#include "front.hpp"

namespace trre {
namespace {

struct Item {
    int32_t st;
    std::string res;
    bool operator==(const Item& o) const { return st == o.st && res == o.res; }
};
using ItemList = std::vector<Item>;

struct ListHash {
    size_t operator()(const ItemList& l) const {
        uint64_t h = 1469598103934665603ull;
        for (const Item& it : l) {
            h = (h ^ (uint32_t)it.st) * 1099511628211ull;
            for (unsigned char ch : it.res) h = (h ^ ch) * 1099511628211ull;
            h = (h ^ 0x1ffu) * 1099511628211ull;
        }
        return (size_t)h;
    }
};

class Determinizer {
public:
    Determinizer(const Nft& nft, const DftLimits& lim)
        : nft_(nft), lim_(lim), mark_(nft.st.size(), 0) {}

    Dft run() {
        ItemList init{Item{nft_.start, std::string()}};   // trre_dft.c:1255-1258
        lists_.push_back(init);
        index_.emplace(init, 0);
        dft_.st.emplace_back();                           // start: finality is never evaluated
        for (size_t cur = 0; cur < lists_.size(); ++cur) {
            if (dft_.st[cur].final || dft_.st[cur].diverges) continue;   // left at once / never entered: its edges are never used
            expand((int32_t)cur);
        }
        return std::move(dft_);
    }

private:
    // Priority-ordered epsilon closure from state s with the pending output in
    // `pending`.  `pending` is one buffer shared with the PREFERRED branch of
    // every split; the other branch continues from a copy taken after the
    // preferred branch returned, i.e. it inherits the bytes the preferred
    // branch's Prod states appended.  This mirrors the reference, where the
    // first recursive call receives the caller's string object itself.
    void closure(int32_t s, std::string& pending, int c, ItemList& out, size_t hops) {
        while (s >= 0) {
            if (++hops > nft_.st.size() + 1)
                throw Error(kErrEpsCycle, "error: epsilon cycle in the pattern (unbounded recursion in the reference)");
            if (++work_ > lim_.max_work)
                throw Error(kErrTooBig, "error: pattern is too expensive to determinise eagerly (closure work cap)");
            const NState& st = nft_.st[s];
            switch (st.kind) {
            case NKind::Split:
            case NKind::SplitNg: {
                closure(nft_.first(s), pending, c, out, hops);
                std::string fork = pending;
                closure(nft_.second(s), fork, c, out, hops);
                return;
            }
            case NKind::Join:
                s = st.a;
                break;
            case NKind::Prod:
                if (pending.size() > lim_.max_residual)
                    throw Error(kErrTooBig, "error: residual output grows without bound (pattern is not determinisable)");
                pending.push_back((char)st.val);
                s = st.a;
                break;
            case NKind::Cons:
                if (c == st.val && !mark_[s]) { mark_[s] = 1; out.push_back(Item{s, pending}); }
                return;
            case NKind::Final:
                if (c == 0 && !mark_[s]) { mark_[s] = 1; out.push_back(Item{s, pending}); }
                return;
            }
        }
    }

    void step(const ItemList& from, int c, ItemList& out) {
        for (const Item& it : from) {
            std::string pending = it.res;
            closure(nft_.st[it.st].a, pending, c, out, 0);
        }
        for (const Item& it : out) mark_[it.st] = 0;
    }

    static std::string strip_common_prefix(ItemList& l) {
        std::string prefix;
        for (;;) {
            if (l[0].res.size() <= prefix.size()) break;
            const char ch = l[0].res[prefix.size()];
            bool all = true;
            for (const Item& it : l)
                if (it.res.size() <= prefix.size() || it.res[prefix.size()] != ch) { all = false; break; }
            if (!all) break;
            prefix.push_back(ch);
        }
        if (!prefix.empty())
            for (Item& it : l) it.res.erase(0, prefix.size());
        return prefix;
    }

    int32_t intern(ItemList& l) {
        auto hit = index_.find(l);
        if (hit != index_.end()) return hit->second;
        if (lists_.size() >= lim_.max_states)
            throw Error(kErrTooBig, "error: too many determinised states (pattern is not determinisable or too large)");
        const int32_t id = (int32_t)lists_.size();
        lists_.push_back(l);
        index_.emplace(l, id);
        dft_.st.emplace_back();
        // finality: closure with the probe byte 0; a Cons state that reads byte 0
        // answers the probe too (the "." quirk) and the FIRST item's residual is
        // the final output
        ItemList probe;
        try {
            step(l, 0, probe);
        } catch (const Error& e) {
            if (e.code != kErrEpsCycle) throw;
            // The reference creates the state and then probes its finality (trre_dft.c:1148-1174): with an epsilon
            // cycle behind one of its items that probe recurses for ever.  Reaching this state is what fails.
            std::fill(mark_.begin(), mark_.end(), 0);
            dft_.st[id].diverges = true;
            return id;
        }
        if (!probe.empty()) {
            dft_.st[id].final = true;
            dft_.st[id].final_out = probe[0].res;
        }
        return id;
    }

    void expand(int32_t id) {
        for (int c = 1; c < 256; ++c) {
            if (c == '\n') continue;      // a line never holds NUL or '\n' (getline + C string)
            ItemList next;
            try {
                step(lists_[id], c, next);    // lists_ may reallocate inside intern(): index, don't hold refs
            } catch (const Error& e) {
                if (e.code != kErrEpsCycle) throw;
                // the closure of this very step runs round an epsilon cycle: a table miss here never returns in the
                // reference (the lazy construction only gets here when the input makes it take this edge)
                std::fill(mark_.begin(), mark_.end(), 0);
                dft_.st[id].edge[c].to = kEdgeDiverge;
                continue;
            }
            if (next.empty()) continue;   // dead edge
            std::string prefix = strip_common_prefix(next);
            const int32_t to = intern(next);
            DftEdge& e = dft_.st[id].edge[c];
            if (dft_.st[to].diverges) { e.to = kEdgeDiverge; continue; }
            e.to = to;
            e.out = std::move(prefix);
        }
        dft_.st[id].expanded = true;
    }

    const Nft& nft_;
    DftLimits lim_;
    std::vector<uint8_t> mark_;
    uint64_t work_ = 0;
    std::vector<ItemList> lists_;
    std::unordered_map<ItemList, int32_t, ListHash> index_;
    Dft dft_;
};

}  // namespace

Dft determinize(const Nft& nft, const DftLimits& lim) { return Determinizer(nft, lim).run(); }

// -----------------------------------------------------------------------------
// flattening
// -----------------------------------------------------------------------------
namespace {

struct PoolBuilder {
    std::vector<uint8_t> bytes;
    std::unordered_map<std::string, uint32_t> seen;
    uint32_t put(const std::string& s) {
        auto hit = seen.find(s);
        if (hit != seen.end()) return hit->second;
        while (bytes.size() % 4) bytes.push_back(0);
        const uint32_t off = (uint32_t)bytes.size();
        const uint32_t len = (uint32_t)s.size();
        for (int k = 0; k < 4; ++k) bytes.push_back((uint8_t)(len >> (8 * k)));
        bytes.insert(bytes.end(), s.begin(), s.end());
        seen.emplace(s, off);
        return off;
    }
};

uint64_t encode(uint32_t kind, uint32_t next_row, const std::string& out, PoolBuilder& pool) {
    uint64_t lo = kind | (uint64_t)next_row << 5;
    uint64_t hi;
    if (out.size() <= 4) {
        lo |= (uint64_t)out.size() << 2;
        hi = 0;
        for (size_t k = 0; k < out.size(); ++k) hi |= (uint64_t)(uint8_t)out[k] << (8 * k);
    } else {
        lo |= (uint64_t)kIlenPooled << 2;
        hi = pool.put(out);
    }
    return lo | hi << 32;
}

}  // namespace

DftTables flatten_dft(const Dft& dft) {
    DftTables t;
    t.n_states = (uint32_t)dft.st.size();
    // rows: non-final states in discovery order (start = row 0)
    std::vector<int32_t> row_of(dft.st.size(), -1);
    std::vector<int32_t> state_of_row;
    for (size_t s = 0; s < dft.st.size(); ++s)
        if (!dft.st[s].final && !dft.st[s].diverges) { row_of[s] = (int32_t)state_of_row.size(); state_of_row.push_back((int32_t)s); }
    t.n_rows = (uint32_t)state_of_row.size();
    if (t.n_rows >= (1u << 27)) throw Error(kErrTooBig, "error: too many table rows");

    // full-width entries first, then merge identical columns into classes
    PoolBuilder pool;
    std::vector<std::array<uint64_t, 256>> full(t.n_rows);
    for (uint32_t r = 0; r < t.n_rows; ++r) {
        const DftState& s = dft.st[state_of_row[r]];
        for (int c = 0; c < 256; ++c) {
            const DftEdge& e = s.edge[c];
            if (e.to == kEdgeDiverge) { full[r][c] = kEntDiverge; continue; }
            if (e.to < 0) { full[r][c] = kEntDead; continue; }
            const DftState& tgt = dft.st[e.to];
            if (tgt.final) full[r][c] = encode(kEntAccept, 0, e.out + tgt.final_out, pool);
            else full[r][c] = encode(kEntGoto, (uint32_t)row_of[e.to], e.out, pool);
            t.max_edge_out = std::max<uint32_t>(t.max_edge_out, (uint32_t)(e.out.size() + (tgt.final ? tgt.final_out.size() : 0)));
        }
    }
    // class 0 is reserved for the line terminators; every other byte is classed
    // by its column
    std::vector<std::vector<uint64_t>> cols;   // representative column per class
    cols.emplace_back(t.n_rows, (uint64_t)kEntDead);
    std::unordered_map<std::string, uint32_t> col_index;
    for (int c = 0; c < 256; ++c) {
        if (c == 0 || c == '\n') { t.cls[c] = kClassEol; continue; }
        std::string key((size_t)t.n_rows * 8, '\0');
        for (uint32_t r = 0; r < t.n_rows; ++r)
            for (int k = 0; k < 8; ++k) key[(size_t)r * 8 + k] = (char)(full[r][c] >> (8 * k));
        auto hit = col_index.find(key);
        if (hit == col_index.end()) {
            if (cols.size() >= 256) throw Error(kErrTooBig, "error: too many byte classes");
            hit = col_index.emplace(key, (uint32_t)cols.size()).first;
            std::vector<uint64_t> col(t.n_rows);
            for (uint32_t r = 0; r < t.n_rows; ++r) col[r] = full[r][c];
            cols.push_back(std::move(col));
        }
        t.cls[c] = (uint8_t)hit->second;
    }
    t.n_cls = (uint32_t)cols.size();
    t.ent.resize((size_t)t.n_rows * t.n_cls);
    for (uint32_t r = 0; r < t.n_rows; ++r)
        for (uint32_t k = 0; k < t.n_cls; ++k) t.ent[(size_t)r * t.n_cls + k] = cols[k][r];
    t.pool = std::move(pool.bytes);

    // ---- static properties the launcher specialises on -----------------------
    // delta(state) = output emitted so far minus input consumed so far inside an
    // attempt.  Length-preserving: delta is a function of the state alone and
    // every accepting edge closes at delta 0.  No-overrun: delta never rises
    // above 0 (pending output always fits under the bytes already consumed).
    auto out_len = [&](uint64_t e) -> int64_t {
        uint32_t il = (uint32_t)(e >> 2) & 7u;
        if (il != kIlenPooled) return il;
        uint32_t off = (uint32_t)(e >> 32), len = 0;
        for (int k = 0; k < 4; ++k) len |= (uint32_t)t.pool[off + k] << (8 * k);
        return len;
    };
    bool lp = true, no_overrun = true, memoryless = true;
    std::vector<int64_t> delta(t.n_rows, INT64_MIN);
    std::vector<uint32_t> work{0};
    delta[0] = 0;
    while (!work.empty() && lp) {
        const uint32_t r = work.back();
        work.pop_back();
        for (uint32_t k = 0; k < t.n_cls && lp; ++k) {
            const uint64_t e = t.ent[(size_t)r * t.n_cls + k];
            const uint32_t kind = (uint32_t)e & 3u;
            if (kind == kEntDead || kind == kEntDiverge) continue;
            const int64_t d = delta[r] + out_len(e) - 1;
            if (d > 0) no_overrun = false;
            if (kind == kEntAccept) {
                if (d != 0) lp = false;
            } else {
                const uint32_t nr = (uint32_t)(e >> 5) & 0x7ffffffu;
                if (delta[nr] == INT64_MIN) { delta[nr] = d; work.push_back(nr); }
                else if (delta[nr] != d) lp = false;
            }
        }
    }
    for (int c = 0; c < 256; ++c) {
        t.bytemap[c] = (uint8_t)c;
        const uint64_t e = t.ent[t.cls[c]];
        const uint32_t kind = (uint32_t)e & 3u;
        if (kind == kEntDead) continue;
        if (kind == kEntDiverge) { memoryless = false; continue; }
        if (kind == kEntAccept && ((e >> 2) & 7u) == 1) t.bytemap[c] = (uint8_t)(e >> 32);
        else memoryless = false;
    }
    if (lp) t.flags |= kFlagLengthPreserving;
    if (lp && no_overrun) t.flags |= kFlagNoOverrun;
    if (memoryless) t.flags |= kFlagMemoryless | kFlagLengthPreserving | kFlagNoOverrun;
    return t;
}

}  // namespace trre
