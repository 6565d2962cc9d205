// stream_pack.cpp — see stream_pack.hpp.  Entry formats are documented in front.hpp (StreamTables).
#include "stream_pack.hpp"

#include <algorithm>
#include <unordered_map>

namespace trre {
namespace {

// entries for the positional-window kernel (see front.hpp): 16 bytes when no state has more than
// 3 bytes pending (at most 4 bytes per transition, 32-bit window), 32 bytes up to 7 pending
// (at most 8 bytes per transition, 64-bit window)
void build_window_form(StreamTables& t, const StreamPackInput& in) {
    uint32_t delay = 0;
    for (uint32_t s = 0; s < t.n_states; ++s) delay = std::max(delay, t.pending_len[s]);
    if (delay > 7) return;
    const bool wide = delay > 3;
    const uint32_t words = wide ? 8u : 4u;
    if ((size_t)t.n_states * t.n_cls * words * 4 > 32768) return;    // the table lives in LDS
    std::vector<uint32_t> v((size_t)t.n_states * t.n_cls * words, 0);
    for (uint32_t s = 0; s < t.n_states; ++s) {
        for (uint32_t k = 0; k < t.n_cls; ++k) {
            const StreamCell& x = in.rows[s][k];
            const size_t n = x.out.size() + (x.copy_c ? 1 : 0);
            if (n > (wide ? 8u : 4u)) return;        // (a NUL flushes pending + '\n': at most delay + 1)
            uint32_t* e = &v[((size_t)s * t.n_cls + k) * words];
            e[0] = x.next * t.n_cls * words * 4u;
            const bool silent = s == in.skip || s == in.done;
            e[1] = (silent ? 0u : 8u * (delay - t.pending_len[s])) | (x.eol ? 64u : 0u) |
                   ((in.col_kind[k] == kColNul && !silent) ? 128u : 0u) | (x.diverge ? 256u : 0u);
            for (uint32_t half = 0; half < words / 4; ++half) {
                uint32_t bytes = 0, sel = 0;
                for (size_t b = 0; b < 4; ++b) {
                    const size_t pos = 4 * half + b;
                    uint32_t pick = 0x0cu;                                   // constant 0x00
                    if (pos < x.out.size()) { bytes |= (uint32_t)(uint8_t)x.out[pos] << (8 * b); pick = (uint32_t)b; }
                    else if (pos == x.out.size() && x.copy_c) pick = 4u;     // byte 0 of the input register
                    sel |= pick << (8 * b);
                }
                e[2 + 2 * half] = bytes;
                e[3 + 2 * half] = sel;
            }
        }
    }
    t.lpw = std::move(v);
    t.lpw_delay = delay;
    t.lpw_ok = true;
}

// pair form of the window entries (front.hpp): the composite of two steps, for tables whose pair form fits LDS
void build_window_pair_form(StreamTables& t, const StreamPackInput& in) {
    if (!t.lpw_ok || t.lpw_delay > 3) return;
    const size_t C = t.n_cls, n = t.n_states;
    if (n * C * C * 32 > 32768) return;
    const uint32_t delay = t.lpw_delay;
    std::vector<uint32_t> v(n * C * C * 8, 0);
    for (uint32_t s = 0; s < n; ++s) {
        for (uint32_t k0 = 0; k0 < C; ++k0) {
            const StreamCell& a = in.rows[s][k0];
            for (uint32_t k1 = 0; k1 < C; ++k1) {
                const StreamCell& b = in.rows[a.next][k1];
                const bool silent_a = s == in.skip || s == in.done, silent_b = a.next == in.skip || a.next == in.done;
                // window positions (bytes past the release point before the first step): a step's bytes land at
                // delay - pending of its state; the second step happens one release later
                const uint32_t off_a = silent_a ? 0u : delay - t.pending_len[s];
                const uint32_t off_b = (silent_b ? 0u : delay - t.pending_len[a.next]) + 1u;
                int what[8];                                  // -1 nothing, 0..255 a literal, 256 / 257 the first / second input byte
                for (int& w : what) w = -1;
                bool ok = true;
                auto place = [&](const StreamCell& x, uint32_t off, int input) {
                    uint32_t pos = off;
                    for (unsigned char ch : x.out) { if (pos >= 8 || what[pos] != -1) { ok = false; return; } what[pos++] = ch; }
                    if (x.copy_c) { if (pos >= 8 || what[pos] != -1) { ok = false; return; } what[pos++] = 256 + input; }
                };
                place(a, off_a, 0);
                place(b, off_b, 1);
                if (!ok) return;                              // (not a table the window walk can take two steps at a time)
                uint32_t* e = &v[((size_t)s * C * C + (size_t)k0 * C + k1) * 8];
                e[0] = (uint32_t)(b.next * C * C * 32u);
                e[1] = (a.eol ? 64u : 0u) | (b.eol ? 512u : 0u) | ((a.diverge || b.diverge) ? 256u : 0u) |
                       (((in.col_kind[k0] == kColNul && !silent_a) || (in.col_kind[k1] == kColNul && !silent_b)) ? 128u : 0u);
                for (int half = 0; half < 2; ++half) {
                    uint32_t bytes = 0, sel = 0;
                    for (int i = 0; i < 4; ++i) {
                        const int w = what[4 * half + i];
                        uint32_t pick = 0x0cu;                                   // constant 0x00
                        if (w >= 256) pick = 4u + (uint32_t)(w - 256);           // byte 0 / 1 of the input register
                        else if (w >= 0) { bytes |= (uint32_t)w << (8 * i); pick = (uint32_t)i; }
                        sel |= pick << (8 * i);
                    }
                    e[2 + 2 * half] = bytes;
                    e[3 + 2 * half] = sel;
                }
            }
        }
    }
    t.lpw2 = std::move(v);
    t.lpw2_ok = true;
}

// 16-byte entries for the count and emit passes (see front.hpp); small tables only (they live in LDS)
void build_gen16(StreamTables& t, const StreamPackInput& in) {
    if ((size_t)t.n_states * t.n_cls > 2048) return;
    std::vector<uint32_t> v((size_t)t.n_states * t.n_cls * 4, 0);
    for (uint32_t s = 0; s < t.n_states; ++s) {
        for (uint32_t k = 0; k < t.n_cls; ++k) {
            const StreamCell& x = in.rows[s][k];
            const size_t n = x.out.size() + (x.copy_c ? 1 : 0);
            const bool slow = n > 4;
            if (slow) t.flags |= kFlagG16Slow;
            uint32_t* e = &v[((size_t)s * t.n_cls + k) * 4];
            e[0] = x.next * t.n_cls * 16u;
            const bool silent = s == in.skip || s == in.done;
            // [8] identity: the transition emits exactly the byte it reads (the record pass of the patch path lists every
            // other transition as an edit: scan_block.hpp g16_lane<3>); a record's '\n' emitted on the '\n' column is the byte read
            const bool ident = (x.out.empty() && x.copy_c) || (in.col_kind[k] == kColNewline && !x.copy_c && x.out == "\n");
            e[1] = (slow ? 128u : (uint32_t)n) | (x.eol ? 32u : 0u) | (x.ovf ? 64u : 0u) | (x.diverge ? 16u : 0u) |
                   ((in.col_kind[k] == kColNul && !silent) ? 8u : 0u) | (ident && !x.ovf && !x.diverge ? 256u : 512u) |
                   // [10] a transition of SKIP / DONE (nobody's bytes), [11] an edit for the mark pass of the splice form: [9] and not [10]
                   (silent ? 1024u : 0u) | (!silent && !(ident && !x.ovf && !x.diverge) ? 2048u : 0u) |
                   // [9] an edit (not the identity), [23:16] what it adds: bytes emitted - 1 (signed; a slow entry counts as 0 here)
                   (((uint32_t)(int32_t)((slow ? 0 : (int)n) - 1) & 0xffu) << 16) |
                   // [31:24] 8 x the bytes a fast entry emits (the emit walk adds it to its bit position as it is)
                   ((slow ? 0u : 8u * (uint32_t)n) << 24);
            uint32_t bytes = 0, sel = 0x0c0c0c0cu;                       // constant 0x00 everywhere
            if (!slow) {
                sel = 0;
                for (size_t b = 0; b < 4; ++b) {
                    uint32_t pick = 0x0cu;
                    if (b < x.out.size()) { bytes |= (uint32_t)(uint8_t)x.out[b] << (8 * b); pick = (uint32_t)b; }
                    else if (b == x.out.size() && x.copy_c) pick = 4u;   // byte 0 of the input register
                    sel |= pick << (8 * b);
                }
            }
            e[2] = bytes;
            e[3] = sel;
        }
    }
    t.g16 = std::move(v);
    t.g16_ok = true;
}

// pair form (see front.hpp); tables whose 16-byte and pair forms together stay small (they share LDS with the staging rings)
void build_pairs(StreamTables& t, const StreamPackInput& in) {
    const size_t C = t.n_cls, n = t.n_states;
    if (!t.g16_ok || n * C * C * 32 + t.g16.size() * 4 > 12288) return;
    std::vector<uint32_t> v(n * C * C * 8, 0);
    for (uint32_t s = 0; s < n; ++s) {
        for (uint32_t k0 = 0; k0 < C; ++k0) {
            const StreamCell& a = in.rows[s][k0];
            for (uint32_t k1 = 0; k1 < C; ++k1) {
                const StreamCell& b = in.rows[a.next][k1];
                // the appended bytes: literal, or 0x100 + i for input byte i of the pair
                std::vector<uint32_t> seq;
                for (unsigned char ch : a.out) seq.push_back(ch);
                if (a.copy_c) seq.push_back(0x100);
                for (unsigned char ch : b.out) seq.push_back(ch);
                if (b.copy_c) seq.push_back(0x101);
                const bool silent_a = s == in.skip || s == in.done, silent_b = a.next == in.skip || a.next == in.done;
                const bool slow = seq.size() > 8;
                uint32_t* e = &v[((size_t)s * C * C + (size_t)k0 * C + k1) * 8];
                e[0] = b.next * t.n_cls * 16u;
                // [9] one of the two is an edit for the mark pass of the splice form (as bit 11 of the 16-byte entries: not the identity,
                // not a transition of SKIP / DONE), [10] the pair begins in SKIP / DONE (with [9]: the mark pass walks it as two steps)
                auto ident = [&](const StreamCell& x, uint32_t k) {
                    return ((x.out.empty() && x.copy_c) || (in.col_kind[k] == kColNewline && !x.copy_c && x.out == "\n")) && !x.ovf && !x.diverge;
                };
                const bool edit = (!silent_a && !ident(a, k0)) || (!silent_b && !ident(b, k1));
                e[1] = (slow ? 128u : (uint32_t)seq.size()) | ((a.diverge || b.diverge) ? 16u : 0u) | ((a.eol || b.eol) ? 32u : 0u) |
                       ((a.ovf || b.ovf) ? 64u : 0u) | (edit ? 512u : 0u) | (silent_a ? 1024u : 0u) |
                       (((in.col_kind[k0] == kColNul && !silent_a) || (in.col_kind[k1] == kColNul && !silent_b)) ? 256u : 0u) |
                       // [31:24] 8 x the bytes of the first half that count (the emit walk adds it to its bit position as it is)
                       ((slow ? 0u : 8u * (uint32_t)std::min<size_t>(seq.size(), 4)) << 24);
                if (slow) t.p32_slow = true;
                for (int half = 0; half < 2; ++half) {
                    uint32_t bytes = 0, sel = 0;
                    for (size_t i = 0; i < 4; ++i) {
                        const size_t pos = 4 * half + i;
                        uint32_t pick = 0x0cu;                                   // constant 0x00
                        if (!slow && pos < seq.size()) {
                            if (seq[pos] >= 0x100) pick = 4u + (seq[pos] - 0x100);   // input byte 0 / 1 of the pair
                            else { bytes |= seq[pos] << (8 * i); pick = (uint32_t)i; }
                        }
                        sel |= pick << (8 * i);
                    }
                    e[2 + 2 * half] = bytes;
                    e[3 + 2 * half] = sel;
                }
            }
        }
    }
    t.p32 = std::move(v);
    t.p32_ok = true;
}

}  // namespace

StreamTables pack_stream_tables(const StreamPackInput& in) {
    StreamTables t;
    const uint32_t n = (uint32_t)in.rows.size();
    t.n_states = n;
    t.pending_len = in.pending_len;
    t.n_cls = (uint32_t)in.col_kind.size();
    if ((t.n_cls > 256 && !in.wide_cols) || t.n_cls > 65535 || (uint64_t)n * t.n_cls >= (1u << 24)) throw StreamGiveUp();
    t.ent.resize((size_t)n * t.n_cls);
    std::unordered_map<std::string, uint32_t> pooled;
    bool lp = !in.never_lp, inplace_ok = true;
    for (uint32_t s = 0; s < n; ++s) {
        for (uint32_t k = 0; k < t.n_cls; ++k) {
            const StreamCell& x = in.rows[s][k];
            const bool nul_col = in.col_kind[k] == kColNul;
            uint64_t lo = (uint64_t)x.next * t.n_cls;
            uint64_t hi = 0;
            if (x.out.size() <= 4) {
                lo |= (uint64_t)x.out.size() << 24;
                for (size_t b = 0; b < x.out.size(); ++b) hi |= (uint64_t)(uint8_t)x.out[b] << (8 * b);
            } else {
                lo |= 7ull << 24;
                auto hit = pooled.find(x.out);
                if (hit == pooled.end()) {
                    while (t.pool.size() % 4) t.pool.push_back(0);
                    hit = pooled.emplace(x.out, (uint32_t)t.pool.size()).first;
                    uint32_t len = (uint32_t)x.out.size();
                    for (int b = 0; b < 4; ++b) t.pool.push_back((uint8_t)(len >> (8 * b)));
                    t.pool.insert(t.pool.end(), x.out.begin(), x.out.end());
                }
                if (hit->second >= (1u << 26)) throw StreamGiveUp();
                hi = (uint64_t)(hit->second >> 2) | (uint64_t)std::min<size_t>(x.out.size(), 255) << 24;
            }
            if (x.copy_c) lo |= 1ull << 27;
            if (x.eol) lo |= 1ull << 28;
            if (x.ovf) lo |= 1ull << 30;
            if (x.diverge) lo |= 1ull << 31;
            if (nul_col && s != in.skip && s != in.done) lo |= 1ull << 29;   // a NUL cut a line short
            // in-place safety: an emitted '\n' may only be the last byte of a record-end transition
            if (!nul_col) {   // (a NUL voids the in-place launch anyway: kStNul)
                const size_t nl = x.out.find('\n');
                if (nl != std::string::npos && !(x.eol && nl + 1 == x.out.size())) inplace_ok = false;
                if (x.copy_c && in.col_kind[k] == kColNewline) inplace_ok = false;
            }
            t.ent[(size_t)s * t.n_cls + k] = lo | hi << 32;
            t.max_out = std::max<uint32_t>(t.max_out, (uint32_t)x.out.size() + (x.copy_c ? 1 : 0));
            // length-preserving: bytes emitted = pending released + the byte read
            if (s != in.skip && s != in.done && !nul_col && !x.diverge) {
                const int64_t emitted = (int64_t)x.out.size() + (x.copy_c ? 1 : 0);
                const int64_t expect = (int64_t)t.pending_len[s] + 1 - (int64_t)t.pending_len[x.next];
                if (emitted != expect) lp = false;
            }
        }
    }
    while (t.pool.size() % 4) t.pool.push_back(0);
    for (int k = 0; k < 8; ++k) t.pool.push_back(0);   // 8-byte reads of a record's text stay inside the pool
    t.ok = true;
    t.bounded = in.bounded;
    if (in.bounded) lp = false;         // (a void launch must be noticed: only the count pass reports it)
    if (lp && !in.wide_cols) build_window_form(t, in);
    if (lp && !in.wide_cols) build_window_pair_form(t, in);
    if (!in.wide_cols) build_gen16(t, in);
    if (!in.wide_cols) build_pairs(t, in);
    if (lp) t.flags |= kFlagLengthPreserving;
    if (lp && inplace_ok) t.flags |= kFlagNoOverrun;    // the in-place kernel may run
    return t;
}

}  // namespace trre
