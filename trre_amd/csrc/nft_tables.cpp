// nft_tables.cpp — NFT -> priority-exact device tables for the
// non-deterministic engine.
//
// The reference resolves non-determinism by depth-first backtracking and takes
// the FIRST path that reaches FINAL (trre_nft.c:593-657).  On the GPU the same
// answer is obtained without a backtracking stack:
//
//   backward sweep   G[i] = set of CONS states that read line[i] AND from which
//                    FINAL is still reachable on line[i+1..]  (one bitmask per
//                    input byte, advanced right-to-left with the tables below);
//   guided walk      from the start, always take the first entry of the current
//                    state's follow list whose target is FINAL or lies in G[i].
//
// Because a backtracking search from (state, i) succeeds iff FINAL is reachable
// from (state, i), and explores alternatives in list order, the guided walk ends
// on exactly the path the reference prints.  The lists are computed here.
#include <functional>
#include <unordered_map>

#include "front.hpp"

namespace trre {

NftTables build_nft_tables(const Nft& nft) {
    NftTables t;
    t.n_states = (uint32_t)nft.st.size();
    if (nft.st.size() > 2000000) throw Error(kErrTooBig, "error: NFT too large for the non-deterministic GPU engine");

    // CONS states get dense indices in creation order
    std::vector<int32_t> cons_id(nft.st.size(), -1);
    std::vector<int32_t> cons_state;
    for (size_t s = 0; s < nft.st.size(); ++s)
        if (nft.st[s].kind == NKind::Cons) { cons_id[s] = (int32_t)cons_state.size(); cons_state.push_back((int32_t)s); }
    t.n_cons = (uint32_t)cons_state.size();
    if (t.n_cons > 64)
        throw Error(kErrUnsupported,
                    "error: pattern has more than 64 consuming states; use the deterministic engine for it");
    for (uint32_t k = 0; k < t.n_cons; ++k) t.cons_mask[nft.st[cons_state[k]].val] |= 1ull << k;
    t.pred.assign(t.n_cons, 0);

    std::vector<uint8_t> on_path(nft.st.size(), 0);
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

    bool lp = true;
    std::vector<int64_t> delta(t.n_cons + 1, INT64_MIN);   // [n_cons] = start

    auto list_for = [&](int32_t from_state, uint32_t owner) {
        // depth-first, priority order, first occurrence of each target wins,
        // stop at FINAL or when an epsilon cycle closes
        uint64_t seen = 0;
        bool done = false;
        std::function<void(int32_t, std::string&)> visit = [&](int32_t s, std::string& out) {
            std::vector<int32_t> entered;
            while (s >= 0 && !done) {
                const NState& st = nft.st[s];
                if (st.kind == NKind::Cons || st.kind == NKind::Final) {
                    const bool fin = st.kind == NKind::Final;
                    if (fin || !(seen >> cons_id[s] & 1)) {
                        NftFollow f{};
                        f.target = fin ? kTgtFinal : (uint8_t)cons_id[s];
                        size_t nul = out.find('\0');
                        std::string eff = nul == std::string::npos ? out : out.substr(0, nul);
                        f.mute = nul != std::string::npos;
                        if (eff.size() > 0xffff) throw Error(kErrTooBig, "error: output between two consumed bytes exceeds 65535 bytes");
                        f.out_len = (uint16_t)eff.size();
                        f.out_off = pool_put(eff);
                        t.follow.push_back(f);
                        if (fin) { done = true; if (owner < t.n_cons) t.to_final |= 1ull << owner; }
                        else { seen |= 1ull << cons_id[s]; if (owner < t.n_cons) t.pred[cons_id[s]] |= 1ull << owner; }
                    }
                    break;
                }
                if (on_path[s]) {               // the search would go round this cycle for ever
                    NftFollow f{};
                    f.target = kTgtDiverge;
                    t.follow.push_back(f);
                    done = true;
                    break;
                }
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

    t.follow_off.assign(t.n_cons + 2, 0);
    for (uint32_t k = 0; k <= t.n_cons; ++k) {
        t.follow_off[k] = (uint32_t)t.follow.size();
        list_for(k < t.n_cons ? nft.st[cons_state[k]].a : nft.start, k);
    }
    t.follow_off[t.n_cons + 1] = (uint32_t)t.follow.size();

    // length-preserving: D(s) = bytes emitted minus bytes consumed once s has
    // consumed its byte must be a function of s, and every FINAL entry closes at 0
    std::vector<uint32_t> work{t.n_cons};
    delta[t.n_cons] = 0;
    while (!work.empty() && lp) {
        uint32_t s = work.back();
        work.pop_back();
        for (uint32_t e = t.follow_off[s]; e < t.follow_off[s + 1] && lp; ++e) {
            const NftFollow& f = t.follow[e];
            if (f.target == kTgtDiverge || f.mute) { lp = false; break; }
            if (f.target == kTgtFinal) { if (delta[s] + f.out_len != 0) lp = false; continue; }
            const int64_t d = delta[s] + f.out_len - 1;
            if (delta[f.target] == INT64_MIN) { delta[f.target] = d; work.push_back(f.target); }
            else if (delta[f.target] != d) lp = false;
        }
    }
    if (lp) t.flags |= kFlagLengthPreserving;
    return t;
}

}  // namespace trre
