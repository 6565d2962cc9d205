// nft_tables.cpp — NFT -> priority-exact tables for the non-deterministic engine.
//
// The reference resolves non-determinism by depth-first backtracking and takes
// the FIRST path that reaches FINAL (trre_nft.c:593-657).  On the GPU the same
// answer is obtained without a backtracking stack:
//
//   backward sweep   for every input position, which consuming nodes can still
//                    reach FINAL on the rest of the line;
//   guided walk      from the start, always take the first entry of the current
//                    node's follow list whose target is FINAL or still alive.
//
// Because a backtracking search from (state, i) succeeds iff FINAL is reachable
// from (state, i), and explores alternatives in list order, the guided walk ends
// on exactly the path the reference prints.  The lists are computed here
// (build_nft_nodes); build_nft_tables packs them for the bitmask tile kernels,
// guided_build.cpp determinises both passes.
#include <functional>
#include <unordered_map>

#include "front.hpp"

namespace trre {

NftNodes build_nft_nodes(const Nft& nft, bool match_mode, bool all_paths) {
    NftNodes t;
    t.match_mode = match_mode;
    t.n_states = (uint32_t)nft.st.size();
    if (nft.st.size() > 2000000) throw Error(kErrTooBig, "error: NFT too large for the non-deterministic GPU engine");

    // nodes in creation order of their (first) CONS state; a byte-range chain is one node
    std::vector<int32_t> node_of(nft.st.size(), -1);     // CONS state (or chain head) -> node
    std::vector<int32_t> member_group(nft.st.size(), -1);
    for (size_t g = 0; g < nft.groups.size(); ++g)
        for (int32_t m : nft.groups[g].members) member_group[m] = (int32_t)g;
    std::vector<int32_t> group_node(nft.groups.size(), -1);
    std::vector<int32_t> node_succ;                       // where a node's follow list starts
    for (size_t s = 0; s < nft.st.size(); ++s) {
        if (nft.st[s].kind != NKind::Cons) continue;
        const int32_t g = member_group[s];
        if (g >= 0) {
            if (group_node[g] < 0) {
                const NGroup& G = nft.groups[g];
                NftNodes::Node nd;
                for (int c = G.lo; c <= (int)G.hi; ++c) nd.bytes[c >> 6] |= 1ull << (c & 63);
                nd.echo = G.echo;
                group_node[g] = (int32_t)t.node.size();
                t.node.push_back(nd);
                node_succ.push_back(G.join);
                node_of[G.head] = group_node[g];
            }
            continue;
        }
        NftNodes::Node nd;
        nd.bytes[nft.st[s].val >> 6] |= 1ull << (nft.st[s].val & 63);
        node_of[s] = (int32_t)t.node.size();
        t.node.push_back(nd);
        node_succ.push_back(nft.st[s].a);
    }
    const uint32_t n_nodes = (uint32_t)t.node.size();

    std::vector<uint8_t> on_path(nft.st.size(), 0);
    // Epsilon states already walked for the list in hand.  The first occurrence of every target wins, so a state reached
    // again (off the current path) has nothing to add: every target behind it is listed, every cycle behind it was met
    // the first time (a cycle through a state that is only now on the path would have run into the path of the first
    // walk).  Without this the walk follows every PATH: nested optional groups — (a?b?c?)??{,2} — made it exponential
    // (found by tools/gpu_fuzz.py, seed 33: a compile that did not end).
    std::vector<uint32_t> walked(nft.st.size(), 0);
    uint32_t epoch = 0;
    size_t n_entries = 0;
    auto list_for = [&](int32_t from_state, std::vector<NodeFollow>& list) {
        // depth-first, priority order, first occurrence of each target wins,
        // stop at FINAL or when an epsilon cycle closes
        ++epoch;
        std::vector<uint8_t> seen(n_nodes, 0);
        bool done = false, seen_final = false;
        std::function<void(int32_t, std::string&)> visit = [&](int32_t s, std::string& out) {
            std::vector<int32_t> entered;
            while (s >= 0 && !done) {
                const NState& st = nft.st[s];
                if (node_of[s] >= 0 || st.kind == NKind::Final) {
                    const bool fin = st.kind == NKind::Final;
                    if (all_paths || (fin ? !seen_final : !seen[node_of[s]])) {
                        NodeFollow f;
                        f.target = fin ? kNodeFinal : (uint32_t)node_of[s];
                        const size_t nul = out.find('\0');
                        f.out = nul == std::string::npos ? out : out.substr(0, nul);
                        f.mute = nul != std::string::npos;
                        list.push_back(std::move(f));
                        if (all_paths && ++n_entries > (4u << 20)) throw Error(kErrTooBig, "error: too many epsilon paths for generator mode");
                        if (fin) { seen_final = true; if (!match_mode && !all_paths) done = true; }   // scan mode: FINAL always accepts
                        else seen[node_of[s]] = 1;
                    }
                    break;
                }
                if (st.kind == NKind::Cons) throw Error(kErrUndefined, "error: byte-range branch entered from outside its chain");
                if (on_path[s]) {               // the search would go round this cycle for ever
                    NodeFollow f;
                    f.target = kNodeDiverge;
                    list.push_back(std::move(f));
                    t.has_diverge = true;
                    done = true;
                    break;
                }
                if (!all_paths && walked[s] == epoch) break;
                walked[s] = epoch;
                on_path[s] = 1;
                entered.push_back(s);
                if (st.kind == NKind::Prod) {
                    out.push_back((char)st.val);
                    if (out.size() > (1u << 20)) throw Error(kErrTooBig, "error: epsilon output too long");
                    s = st.a;
                } else if (st.kind == NKind::Join) {
                    s = st.a;
                } else {                        // Split / SplitNg
                    std::string branch = out;
                    visit(nft.first(s), branch);
                    s = nft.second(s);          // continue with the fallback on this frame
                }
            }
            for (int32_t e : entered) on_path[e] = 0;
        };
        std::string out;
        visit(from_state, out);
    };

    t.follow.resize(n_nodes + 1);
    for (uint32_t k = 0; k <= n_nodes; ++k) list_for(k < n_nodes ? node_succ[k] : nft.start, t.follow[k]);
    return t;
}

NftTables build_nft_tables(const NftNodes& nd) {
    NftTables t;
    t.n_states = nd.n_states;
    t.n_cons = (uint32_t)nd.node.size();
    if (t.n_cons > 64)
        throw Error(kErrUnsupported, "error: pattern has more than 64 consuming nodes and its search does not determinise; "
                                     "use the deterministic engine for it");
    if (nd.has_diverge)
        throw Error(kErrUnsupported, "error: pattern has an epsilon cycle and its search does not determinise");
    for (uint32_t k = 0; k < t.n_cons; ++k)
        for (int c = 0; c < 256; ++c)
            if (nd.node[k].reads((uint8_t)c)) t.cons_mask[c] |= 1ull << k;
    t.pred.assign(t.n_cons, 0);

    std::unordered_map<std::string, uint32_t> pool_index;
    auto pool_put = [&](const std::string& s) -> uint32_t {
        if (s.empty()) return 0;
        auto hit = pool_index.find(s);
        if (hit != pool_index.end()) return hit->second;
        uint32_t off = (uint32_t)t.pool.size();
        t.pool.insert(t.pool.end(), s.begin(), s.end());
        pool_index.emplace(s, off);
        return off;
    };
    t.follow_off.assign(t.n_cons + 2, 0);
    for (uint32_t k = 0; k <= t.n_cons; ++k) {
        t.follow_off[k] = (uint32_t)t.follow.size();
        for (const NodeFollow& e : nd.follow[k]) {
            NftFollow f{};
            const bool fin = e.target == kNodeFinal;
            f.target = fin ? kTgtFinal : (uint8_t)e.target;
            f.flags = (e.mute ? kFollowMute : 0) | (!fin && nd.node[e.target].echo ? kFollowEcho : 0);
            if (e.out.size() > 0xffff) throw Error(kErrTooBig, "error: output between two consumed bytes exceeds 65535 bytes");
            f.out_len = (uint16_t)e.out.size();
            f.out_off = pool_put(e.out);
            t.follow.push_back(f);
            if (k < t.n_cons) {
                if (fin) t.to_final |= 1ull << k;
                else t.pred[e.target] |= 1ull << k;
            }
        }
    }
    t.follow_off[t.n_cons + 1] = (uint32_t)t.follow.size();

    // length-preserving: D(s) = bytes emitted minus bytes consumed once s has
    // consumed its byte must be a function of s, and every FINAL entry closes at 0
    bool lp = true;
    std::vector<int64_t> delta(t.n_cons + 1, INT64_MIN);   // [n_cons] = start
    std::vector<uint32_t> work{t.n_cons};
    delta[t.n_cons] = 0;
    while (!work.empty() && lp) {
        uint32_t s = work.back();
        work.pop_back();
        for (uint32_t e = t.follow_off[s]; e < t.follow_off[s + 1] && lp; ++e) {
            const NftFollow& f = t.follow[e];
            if (f.flags & kFollowMute) { lp = false; break; }
            if (f.target == kTgtFinal) { if (delta[s] + f.out_len != 0) lp = false; continue; }
            const int64_t d = delta[s] + f.out_len + ((f.flags & kFollowEcho) ? 1 : 0) - 1;
            if (delta[f.target] == INT64_MIN) { delta[f.target] = d; work.push_back(f.target); }
            else if (delta[f.target] != d) lp = false;
        }
    }
    if (lp) t.flags |= kFlagLengthPreserving;
    return t;
}

}  // namespace trre
