// nft_build.cpp — AST -> consume/produce automaton (NFT).
//
// Thompson-style construction in which every state either consumes one input
// byte (Cons), produces one output byte (Prod) or is an epsilon state
// (Split/SplitNg/Join/Final).  Fragment shapes and — decisive for the scan
// result — the priority order of every split are those of the reference's
// nft() (trre_nft.c:375-511; trre_dft.c:375-523 adds the initial JOIN).
#include "front.hpp"

namespace trre {
namespace {

struct Frag {
    int32_t head = -1;   // -1: the fragment is unreachable (an empty byte range)
    int32_t tail = -1;   // state whose primary successor is still open
};

class NftBuilder {
public:
    NftBuilder(const Ast& ast, Nft& out) : ast_(ast), nft_(out) {}

    // side: 0 copy (consume and reproduce), 1 left of ':' (consume), 2 right of ':' (produce)
    Frag emit(int32_t id, int side) {
        if (id < 0) return Frag{};
        if (nft_.st.size() > kMaxStates) throw Error(kErrTooBig, "error: pattern expands to too many NFT states");
        const AstNode n = ast_.nodes[id];
        switch (n.type) {
        case '.': {
            Frag l = emit(n.l, side), r = emit(n.r, side);
            connect(l.tail, r.head);
            return Frag{l.head, r.tail};
        }
        case '|': {   // the left alternative has priority (trre_nft.c:391-398)
            Frag l = emit(n.l, side), r = emit(n.r, side);
            int32_t fork = nft_.add(NKind::SplitNg, l.head, r.head);
            int32_t join = nft_.add(NKind::Join);
            connect(l.tail, join);
            connect(r.tail, join);
            return Frag{fork, join};
        }
        case '*': {
            Frag body = emit(n.l, side);
            int32_t loop = nft_.add(n.val ? NKind::SplitNg : NKind::Split, -1, body.head);
            connect(body.tail, loop);
            return Frag{loop, loop};
        }
        case '?': {
            Frag body = emit(n.l, side);
            int32_t join = nft_.add(NKind::Join);
            int32_t fork = nft_.add(n.val ? NKind::SplitNg : NKind::Split, join, body.head);
            connect(body.tail, join);
            return Frag{fork, join};
        }
        case '+': {
            Frag body = emit(n.l, side);
            int32_t loop = nft_.add(n.val ? NKind::SplitNg : NKind::Split, -1, body.head);
            connect(body.tail, loop);
            return Frag{body.head, loop};
        }
        case ':': {
            if (ast_.nodes[n.l].type == 'e') return emit(n.r, 2);
            if (ast_.nodes[n.r].type == 'e') return emit(n.l, 1);
            Frag l = emit(n.l, 1), r = emit(n.r, 2);
            connect(l.tail, r.head);
            return Frag{l.head, r.tail};
        }
        case '-':
            return range(n, side);
        case 'I':
            return repeat(n, side);
        default:
            // a byte; an epsilon placeholder that ends up outside a ':' is treated
            // as the byte it was synthesised at, exactly like the reference
            return byte(n.val, side);
        }
    }

    void connect(int32_t tail, int32_t to) {
        if (tail < 0) throw Error(kErrUndefined, "error: null tail (undefined in the reference)");
        nft_.st[tail].a = to;
    }

private:
    static constexpr size_t kMaxStates = 4000000;

    Frag byte(uint8_t v, int side) {
        if (side == 0) {
            int32_t c = nft_.add(NKind::Cons, -1, -1, v);
            int32_t p = nft_.add(NKind::Prod, -1, -1, v);
            nft_.st[c].a = p;
            return Frag{c, p};
        }
        int32_t s = nft_.add(side == 1 ? NKind::Cons : NKind::Prod, -1, -1, v);
        return Frag{s, s};
    }
    Frag pair(uint8_t in, uint8_t out) {   // "x:y" for single bytes, both sides present
        int32_t c = nft_.add(NKind::Cons, -1, -1, in);
        int32_t p = nft_.add(NKind::Prod, -1, -1, out);
        nft_.st[c].a = p;
        return Frag{c, p};
    }

    // Ranges become a chain of SplitNg states, one branch per byte, built from
    // the top byte down so that the LOWEST byte has the highest priority
    // (trre_nft.c:426-452).
    Frag range(const AstNode& n, int side) {
        const AstNode lo = ast_.nodes[n.l], hi = ast_.nodes[n.r];
        int32_t chain = -1;
        if (lo.type == 'c' && hi.type == 'c') {
            int32_t join = nft_.add(NKind::Join);
            NGroup g;
            for (int c = hi.val; c >= (int)lo.val; --c) {
                Frag f = byte((uint8_t)c, side);
                int32_t fork = nft_.add(NKind::SplitNg, f.head, chain);
                connect(f.tail, join);
                chain = fork;
                if (side != 2) g.members.insert(g.members.begin(), f.head);
            }
            if (g.members.size() >= 2) {
                g.head = chain; g.join = join; g.lo = lo.val; g.hi = hi.val; g.echo = side == 0;
                nft_.groups.push_back(std::move(g));
            }
            return Frag{chain, join};
        }
        if (lo.type == ':' && hi.type == ':') {
            // "a:x-c:z": for k = 0..(c-a): (a+k):(x+k); the right end's output is ignored
            const int in0 = ast_.nodes[lo.l].val, out0 = ast_.nodes[lo.r].val, in1 = ast_.nodes[hi.l].val;
            int32_t join = nft_.add(NKind::Join);
            for (int k = in1 - in0; k >= 0; --k) {
                // the reference rebuilds a ':' node over two byte nodes here, so the
                // pair is consume+produce regardless of the enclosing side
                Frag f = pair((uint8_t)(in0 + k), (uint8_t)(out0 + k));
                int32_t fork = nft_.add(NKind::SplitNg, f.head, chain);
                connect(f.tail, join);
                chain = fork;
            }
            return Frag{chain, join};
        }
        throw Error(kErrSyntax, "error: unexpected range syntax");
    }

    // "{lb,rb}" (trre_nft.c:458-485): lb mandatory copies, then a greedy star when
    // rb == 0, otherwise (rb-lb) optional copies that all exit to one join.
    Frag repeat(const AstNode& n, int side) {
        const int lb = ast_.nodes[n.r].type, rb = ast_.nodes[n.r].val;
        int32_t head = nft_.add(NKind::Join);
        int32_t tail = head;
        for (int i = 0; i < lb; ++i) {
            Frag f = emit(n.l, side);
            connect(tail, f.head);
            tail = f.tail;
        }
        if (rb == 0) {
            Frag body = emit(n.l, side);
            int32_t loop = nft_.add(NKind::Split, -1, body.head);   // always greedy here
            connect(body.tail, loop);
            connect(tail, loop);
            tail = loop;
        } else {
            int32_t exit = nft_.add(NKind::Join);
            for (int i = lb; i < rb; ++i) {
                Frag f = emit(n.l, side);
                connect(tail, nft_.add(n.val ? NKind::SplitNg : NKind::Split, exit, f.head));
                tail = f.tail;
            }
            connect(tail, exit);
            tail = exit;
        }
        return Frag{head, tail};
    }

    const Ast& ast_;
    Nft& nft_;
};

}  // namespace

Nft build_nft(const Ast& ast, bool with_initial_join) {
    Nft nft;
    int32_t fin = nft.add(NKind::Final);
    NftBuilder b(ast, nft);
    Frag f = b.emit(ast.root, 0);
    b.connect(f.tail, fin);
    nft.start = with_initial_join ? nft.add(NKind::Join, f.head) : f.head;
    for (const NState& s : nft.st) nft.n_cons += (s.kind == NKind::Cons);
    return nft;
}

}  // namespace trre
