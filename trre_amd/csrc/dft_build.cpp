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
//
// Round 5: the same construction one table miss at a time (LazyDft, below) — what a pattern runs on whose eager
// construction does not end within its caps ('((a:x)*b)|((a:y)*c)': a state per run length; '(a|b)*a(a|b){18}:x': 2^19
// states).  The device walks the tables that exist, lists the edges nobody has looked at yet, the host builds exactly
// those (trre_dft.c:1135-1175) and the lanes that met them run again: every legal pattern runs on every finite input, as
// in the reference.
#include <algorithm>
#include <cstring>
#include <deque>
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

// What both constructions share: the closure with its pending output, the step over an item list, the factoring.
class ClosureCore {
protected:
    ClosureCore(const Nft& nft, uint64_t max_work, size_t max_residual)
        : nft_(nft), max_work_(max_work), max_residual_(max_residual), mark_(nft.st.size(), 0) {
        for (const NState& s : nft.st)
            if (s.kind == NKind::Cons) reads_[s.val] = true;
    }

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
            if (++work_ > max_work_)
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
                if (pending.size() > max_residual_)
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

    // (a throw leaves marks behind: the catcher calls clear_marks())
    void step(const ItemList& from, int c, ItemList& out) {
        for (const Item& it : from) {
            std::string pending = it.res;
            closure(nft_.st[it.st].a, pending, c, out, 0);
        }
        for (const Item& it : out) mark_[it.st] = 0;
    }
    void clear_marks() { std::fill(mark_.begin(), mark_.end(), 0); }

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

    const Nft& nft_;
    uint64_t max_work_;
    size_t max_residual_;
    std::vector<uint8_t> mark_;
    std::array<bool, 256> reads_{};     // bytes some CONS state reads: only they can leave a state
    uint64_t work_ = 0;
};

class Determinizer : ClosureCore {
public:
    Determinizer(const Nft& nft, const DftLimits& lim) : ClosureCore(nft, lim.max_work, lim.max_residual), lim_(lim) {}

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
            clear_marks();
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
            if (!reads_[c]) continue;     // no CONS state reads it: the step is empty whatever the items (dead edge)
            ItemList next;
            try {
                step(lists_[id], c, next);    // lists_ may reallocate inside intern(): index, don't hold refs
            } catch (const Error& e) {
                if (e.code != kErrEpsCycle) throw;
                // the closure of this very step runs round an epsilon cycle: a table miss here never returns in the
                // reference (the lazy construction only gets here when the input makes it take this edge)
                clear_marks();
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

    DftLimits lim_;
    std::vector<ItemList> lists_;
    std::unordered_map<ItemList, int32_t, ListHash> index_;
    Dft dft_;
};

}  // namespace

Dft determinize(const Nft& nft, const DftLimits& lim) { return Determinizer(nft, lim).run(); }

// -----------------------------------------------------------------------------
// lazy construction (front.hpp: LazyDft)
// -----------------------------------------------------------------------------
namespace {

struct LazyPool {
    std::vector<uint8_t> bytes;
    std::unordered_map<std::string, uint32_t> seen;
    uint32_t put(const std::string& s) {
        auto hit = seen.find(s);
        if (hit != seen.end()) return hit->second;
        while (bytes.size() % 4) bytes.push_back(0);
        if (bytes.size() + s.size() + 8 > 0xfffffff0ull) throw Error(kErrTooBig, "error: the outputs of the determinised tables exceed 4 GiB");
        const uint32_t off = (uint32_t)bytes.size();
        const uint32_t len = (uint32_t)s.size();
        for (int k = 0; k < 4; ++k) bytes.push_back((uint8_t)(len >> (8 * k)));
        bytes.insert(bytes.end(), s.begin(), s.end());
        if (s.size() <= 64) seen.emplace(s, off);     // (long texts are the residuals of run-length states: each occurs once)
        return off;
    }
};

uint64_t lazy_encode(uint32_t kind, uint32_t next_row, const std::string& out, LazyPool& pool) {
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

struct LazyNftCopy { Nft nft; };              // (a copy: the tables outlive the compile call; a base, so that it exists before ClosureCore looks at it)
struct LazyDft::Impl : LazyNftCopy, ClosureCore {
    LazyLimits lim;
    uint32_t n_cls = 0;
    std::array<uint8_t, 256> cls{};
    std::vector<uint8_t> cls_byte;             // the byte a class stands for (classes 0, 1: none)
    std::vector<uint64_t> ent;
    LazyPool pool;
    // a state: its item list, serialised {state, length of the residual, residual}*, is the key of the index; non-final
    // states have a row
    struct State { const std::string* key; int32_t row; bool final, diverges; std::string final_out; };
    std::vector<State> states;
    std::unordered_map<std::string, int32_t> index;
    std::vector<int32_t> state_of_row;
    std::vector<uint32_t> row_epoch;
    uint32_t epoch = 0;
    size_t key_bytes = 0;
    std::deque<uint32_t> fresh;                // rows made by the current explore() call, for the look-ahead

    Impl(const Nft& n, const LazyLimits& l) : LazyNftCopy{n}, ClosureCore(nft, l.max_edge_work, l.max_residual), lim(l) {
        cls_byte = {0, 0};
        for (int c = 0; c < 256; ++c) {
            if (c == 0 || c == '\n') { cls[c] = kClassEol; continue; }
            if (!reads_[c]) { cls[c] = 1; continue; }
            cls[c] = (uint8_t)cls_byte.size();
            cls_byte.push_back((uint8_t)c);
        }
        n_cls = (uint32_t)cls_byte.size();
        ItemList init{Item{nft.start, std::string()}};        // trre_dft.c:1255-1258
        const std::string key = serialise(init);
        add_state(key, false, false, std::string());          // start: finality is never evaluated
    }

    static std::string serialise(const ItemList& l) {
        std::string k;
        size_t need = 0;
        for (const Item& it : l) need += 8 + it.res.size();
        k.reserve(need);
        for (const Item& it : l) {
            const uint32_t w[2] = {(uint32_t)it.st, (uint32_t)it.res.size()};
            k.append(reinterpret_cast<const char*>(w), 8);
            k.append(it.res);
        }
        return k;
    }
    static ItemList parse(const std::string& k) {
        ItemList l;
        for (size_t at = 0; at < k.size();) {
            uint32_t w[2];
            std::memcpy(w, k.data() + at, 8);
            l.push_back(Item{(int32_t)w[0], k.substr(at + 8, w[1])});
            at += 8 + w[1];
        }
        return l;
    }
    size_t bytes_held() const { return ent.size() * 8 + pool.bytes.size() + key_bytes + states.size() * (sizeof(State) + 64); }

    int32_t add_state(const std::string& key, bool final, bool diverges, std::string final_out) {
        if (bytes_held() + key.size() + (size_t)n_cls * 8 > lim.max_bytes)
            throw Error(kErrTooBig, "error: the determinised tables this input needs exceed the memory limit (TRRE_LAZY_MAX_BYTES; the reference keeps every state it meets, too)");
        // (every check before the index learns the key: a throw that a caller swallows must not leave an id without a state — ADVICE r5)
        if (!final && !diverges && state_of_row.size() >= (1u << 27) - 1) throw Error(kErrTooBig, "error: too many table rows");
        const int32_t id = (int32_t)states.size();
        auto ins = index.emplace(key, id);
        key_bytes += key.size();
        State s{&ins.first->first, -1, final, diverges, std::move(final_out)};
        if (!final && !diverges) {
            s.row = (int32_t)state_of_row.size();
            state_of_row.push_back(id);
            row_epoch.push_back(epoch);
            ent.resize(ent.size() + n_cls, kEntUnexplored);
            ent[(size_t)s.row * n_cls + 0] = kEntDead;       // line terminators
            ent[(size_t)s.row * n_cls + 1] = kEntDead;       // bytes no CONS state reads
            fresh.push_back((uint32_t)s.row);
        }
        states.push_back(std::move(s));
        return id;
    }

    // trre_dft.c:1148-1174: look the list up, else create the state and probe its finality with byte 0 (a Cons state that reads
    // byte 0 answers the probe too; the FIRST item's residual is the final output)
    int32_t intern(ItemList& l) {
        const std::string key = serialise(l);
        auto hit = index.find(key);
        if (hit != index.end()) return hit->second;
        ItemList probe;
        bool diverges = false;
        try {
            step(l, 0, probe);
        } catch (const Error& e) {
            clear_marks();
            if (e.code != kErrEpsCycle) throw;
            diverges = true;                                  // (the probe recurses for ever: reaching this state is what fails)
        }
        const bool final = !diverges && !probe.empty();
        return add_state(key, final, diverges, final ? probe[0].res : std::string());
    }

    // one edge, trre_dft.c:1135-1175.  ahead: explored before the input asked for it — whatever goes wrong leaves it unexplored
    void explore_edge(uint32_t row, uint32_t k, bool ahead) {
        if (ent[(size_t)row * n_cls + k] != kEntUnexplored) return;
        const ItemList from = parse(*states[state_of_row[row]].key);
        ItemList next;
        work_ = 0;
        max_work_ = ahead ? lim.spec_edge_work : lim.max_edge_work;
        uint64_t e;
        try {
            step(from, cls_byte[k], next);
            if (next.empty()) {
                e = kEntDead;
            } else {
                std::string prefix = strip_common_prefix(next);
                const int32_t to = intern(next);
                const State& t = states[to];
                if (t.diverges) e = kEntDiverge;
                else if (t.final) e = lazy_encode(kEntAccept, 0, prefix + t.final_out, pool);
                else e = lazy_encode(kEntGoto, (uint32_t)t.row, prefix, pool);
            }
        } catch (const Error& err) {
            clear_marks();
            if (err.code == kErrEpsCycle) e = kEntDiverge;    // a table miss here never returns in the reference
            else if (ahead) return;
            else if (err.code == kErrTooBig && work_ > max_work_)
                throw Error(kErrTooBig, "error: one step of the determinisation takes more than 4e8 closure steps on this input");
            else throw;
        }
        ent[(size_t)row * n_cls + k] = e;
        row_epoch[row] = epoch;
    }

    void explore(const uint32_t* misses, size_t n, size_t spec_states) {
        ++epoch;
        fresh.clear();
        for (size_t i = 0; i < n; ++i) {
            const uint32_t* r = misses + kLazyMissWords * i;
            uint32_t row = r[0], k = r[1];
            const uint32_t n_follow = r[2] < 48u ? r[2] : 48u;
            if (row >= state_of_row.size() || k >= n_cls) throw Error(kErrArg, "error: a miss record names an edge that does not exist");
            explore_edge(row, k, false);
            // ... and the attempt goes on along the bytes the lane sent with it, as infer_dft does after a miss
            for (uint32_t j = 0; j < n_follow; ++j) {
                const uint64_t e = ent[(size_t)row * n_cls + k];
                if (((uint32_t)e & 3u) != kEntGoto) break;
                row = (uint32_t)e >> 5;
                k = cls[(uint8_t)(r[4 + (j >> 2)] >> (8u * (j & 3u)))];
                explore_edge(row, k, false);
            }
        }
        look_ahead(spec_states);
    }
    // breadth first from the rows this call has made, until `spec_states` more states exist (or their memory would not fit)
    void look_ahead(size_t spec_states) {
        const size_t stop = states.size() + spec_states, stop_bytes = bytes_held() + ((size_t)64 << 20);
        while (!fresh.empty() && states.size() < stop && bytes_held() < stop_bytes) {
            const uint32_t row = fresh.front();
            fresh.pop_front();
            for (uint32_t k = 2; k < n_cls && states.size() < stop; ++k) explore_edge(row, k, true);
        }
        fresh.clear();
    }
};

LazyDft::LazyDft(const Nft& nft, const LazyLimits& lim) : impl_(new Impl(nft, lim)) {
    impl_->epoch = 0;
    impl_->look_ahead(lim.seed_states);
}
LazyDft::~LazyDft() = default;
uint32_t LazyDft::n_cls() const { return impl_->n_cls; }
const uint8_t* LazyDft::cls() const { return impl_->cls.data(); }
uint32_t LazyDft::n_rows() const { return (uint32_t)impl_->state_of_row.size(); }
uint32_t LazyDft::n_states() const { return (uint32_t)impl_->states.size(); }
const uint64_t* LazyDft::ent() const { return impl_->ent.data(); }
const uint8_t* LazyDft::pool() const { return impl_->pool.bytes.data(); }
size_t LazyDft::pool_bytes() const { return impl_->pool.bytes.size(); }
void LazyDft::explore(const uint32_t* misses, size_t n, size_t spec_states) { impl_->explore(misses, n, spec_states); }
uint32_t LazyDft::epoch() const { return impl_->epoch; }
uint32_t LazyDft::first_dirty_row(uint32_t since_epoch) const {
    const std::vector<uint32_t>& e = impl_->row_epoch;
    for (size_t r = 0; r < e.size(); ++r)
        if (e[r] > since_epoch) return (uint32_t)r;
    return (uint32_t)e.size();
}

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
