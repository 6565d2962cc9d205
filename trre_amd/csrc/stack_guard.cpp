// Tables of the stack guard (guard_block.hpp): the NFT as the reference built it, and the bound that says which lines are long
// enough for its search to run out of stack.
#include "front.hpp"

#include <algorithm>
#include <functional>

namespace trre {

// D: the most items one consumed byte can leave on the reference's stack.  An item stays while the search is inside the branch
// it tried first (trre_nft.c:623-630: SPLIT pushes nexta and goes on at nextb, SPLITNG the other way round); it is gone once
// the search has fallen back to the other branch.  So on any epsilon path from one read to the next the live items are the
// splits passed by their first branch: the longest such path, over all places a path can start (the start state, the state
// behind every CONS).  Epsilon cycles are left out of the count — the search never returns from one, which the table kernels
// report on their own (TRRE_E_DIVERGES).
GuardTables build_guard(const Nft& nft) {
    GuardTables g;
    const int32_t n = (int32_t)nft.st.size();
    if (n == 0 || nft.start < 0) return g;
    std::vector<int32_t> memo(n, -1);
    std::vector<uint8_t> on_path(n, 0);
    // (iterative DFS with an explicit stack: patterns of 100 000 states exist — the dictionary)
    struct Frame { int32_t s; int stage; int32_t acc; };
    auto depth_from = [&](int32_t root) -> int32_t {
        if (root < 0) return 0;
        if (memo[root] >= 0) return memo[root];
        std::vector<Frame> st;
        int32_t ret = 0;
        // go into `child`: its value lands in ret at once (nothing there, an epsilon cycle — not counted —, known already) or
        // after its frame has run
        auto descend = [&](int32_t child) {
            if (child < 0 || on_path[child]) { ret = 0; return; }
            if (memo[child] >= 0) { ret = memo[child]; return; }
            on_path[child] = 1;
            st.push_back({child, 0, 0});
        };
        on_path[root] = 1;
        st.push_back({root, 0, 0});
        while (!st.empty()) {
            const size_t at = st.size() - 1;
            const int32_t s = st[at].s;
            const NState& x = nft.st[s];
            const bool split = x.kind == NKind::Split || x.kind == NKind::SplitNg;
            bool done = false;
            int32_t value = 0;
            if (x.kind == NKind::Cons || x.kind == NKind::Final) {
                done = true;
            } else if (st[at].stage == 0) {
                st[at].stage = 1;
                descend(split ? nft.first(s) : x.a);
            } else if (st[at].stage == 1) {
                if (split) { st[at].acc = ret + 1; st[at].stage = 2; descend(nft.second(s)); }
                else { value = ret; done = true; }
            } else {
                value = std::max(st[at].acc, ret);
                done = true;
            }
            if (done) {
                memo[s] = value;
                on_path[s] = 0;
                ret = value;
                st.pop_back();
            }
        }
        return ret;
    };
    int32_t d = depth_from(nft.start);
    bool loop = false;
    for (int32_t s = 0; s < n; ++s)
        if (nft.st[s].kind == NKind::Cons) d = std::max(d, depth_from(nft.st[s].a));
    // a consuming cycle: some CONS reaches itself (without one an attempt reads at most n_cons bytes)
    {
        std::vector<uint8_t> color(n, 0);
        std::vector<std::pair<int32_t, int>> st;
        for (int32_t r = 0; r < n && !loop; ++r) {
            if (color[r]) continue;
            st.push_back({r, 0});
            color[r] = 1;
            while (!st.empty() && !loop) {
                auto& [s, k] = st.back();
                const NState& x = nft.st[s];
                const int32_t succ[2] = {x.a, (x.kind == NKind::Split || x.kind == NKind::SplitNg) ? x.b : -1};
                if (k < 2) {
                    const int32_t t = succ[k++];
                    if (t < 0) continue;
                    if (color[t] == 1) { loop = true; break; }
                    if (color[t] == 0) { color[t] = 1; st.push_back({t, 0}); }
                } else {
                    color[s] = 2;
                    st.pop_back();
                }
            }
            st.clear();
        }
    }
    g.d = (uint32_t)d;
    const uint64_t reach = loop ? ~0ull : (uint64_t)d * ((uint64_t)nft.n_cons + 1);      // items an attempt can hold at all
    if (d <= 0 || reach <= 65536) return g;                                            // the stack cannot fill: no guard
    if (d > 64) { g.too_deep = true; return g; }                                       // (lines of under 1 KiB would be suspects: the deviation stays)
    g.on = true;
    g.l_min = 65536u / (uint32_t)d - 1u;
    // The bytes of such an attempt are consecutive bytes of the line, and all but a few were read by CONS states that lie on a
    // cycle: a state on no cycle is passed once.  So the line holds l_min consecutive bytes of which at most n_once are not
    // bytes such a state reads — a run of (l_min - n_once) / (n_once + 1) bytes of the set somewhere — and a run that long
    // covers a whole window of half its size.  (Strongly connected components, Tarjan's, iteratively.)
    std::vector<uint8_t> cyclic(n, 0);
    {
        std::vector<int32_t> idx(n, -1), low(n, 0), comp_stack;
        std::vector<uint8_t> on_cs(n, 0);
        std::vector<std::pair<int32_t, int>> st;
        int32_t counter = 0;
        for (int32_t r = 0; r < n; ++r) {
            if (idx[r] >= 0) continue;
            st.push_back({r, 0});
            idx[r] = low[r] = counter++;
            comp_stack.push_back(r); on_cs[r] = 1;
            while (!st.empty()) {
                const int32_t s = st.back().first;
                const NState& x = nft.st[s];
                const int32_t succ[2] = {x.a, (x.kind == NKind::Split || x.kind == NKind::SplitNg) ? x.b : -1};
                if (st.back().second < 2) {
                    const int32_t t = succ[st.back().second++];
                    if (t < 0) continue;
                    if (t == s) cyclic[s] = 1;
                    if (idx[t] < 0) {
                        idx[t] = low[t] = counter++;
                        comp_stack.push_back(t); on_cs[t] = 1;
                        st.push_back({t, 0});
                    } else if (on_cs[t]) {
                        low[s] = std::min(low[s], idx[t]);
                    }
                } else {
                    if (low[s] == idx[s]) {
                        size_t first = comp_stack.size();
                        while (comp_stack[first - 1] != s) --first;
                        --first;
                        const bool many = comp_stack.size() - first > 1;
                        for (size_t k = first; k < comp_stack.size(); ++k) { if (many) cyclic[comp_stack[k]] = 1; on_cs[comp_stack[k]] = 0; }
                        comp_stack.resize(first);
                    }
                    st.pop_back();
                    if (!st.empty()) low[st.back().first] = std::min(low[st.back().first], low[s]);
                }
            }
        }
    }
    uint64_t n_once = 0;
    for (int32_t s = 0; s < n; ++s) {
        if (nft.st[s].kind != NKind::Cons) continue;
        const uint8_t c = nft.st[s].val;
        if (!cyclic[s]) { ++n_once; continue; }
        if (c != 0 && c != (uint8_t)'\n') g.bset[c >> 5] |= 1u << (c & 31);
    }
    g.n_once = (uint32_t)std::min<uint64_t>(n_once, 0xffffffffu);
    const uint64_t run_min = n_once >= g.l_min ? 0 : ((uint64_t)g.l_min - n_once) / (n_once + 1);
    if (run_min >= 64) {
        g.run_min = (uint32_t)run_min;
        g.window = ((uint32_t)run_min / 2u) & ~15u;
    } else {
        // (so many states outside the loops that short runs would do: any line of l_min bytes is a suspect — windows without a '\n')
        for (int k = 0; k < 8; ++k) g.bset[k] = 0xffffffffu;
        g.bset[0] &= ~((1u << 10) | 1u);
        g.run_min = g.l_min;
        g.window = (g.l_min / 2u) & ~15u;
    }
    g.start = (uint32_t)nft.start;
    g.states.resize((size_t)n * 4);
    for (int32_t s = 0; s < n; ++s) {
        const NState& x = nft.st[s];
        g.states[4 * (size_t)s] = (uint32_t)x.kind | (uint32_t)x.val << 8;
        g.states[4 * (size_t)s + 1] = (uint32_t)x.a;
        g.states[4 * (size_t)s + 2] = (uint32_t)x.b;
        g.states[4 * (size_t)s + 3] = 0;
    }
    return g;
}

}  // namespace trre
