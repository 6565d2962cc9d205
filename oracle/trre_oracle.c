/*
 * trre_oracle.c — CPU oracle for the scan-mode hot path.  TEST INFRASTRUCTURE.
 *
 * Plain-C restatement of what c0stya/trre computes in scan mode, written from
 * the behaviour of the reference (file:line citations are into /root/reference)
 * with its own data structures (index-based arenas, byte vectors, a hashed
 * state cache) — no reference source is reproduced here.
 *
 * Nothing in the product path may link, import or execute this file; see
 * trre_oracle.h for who may use it and how its parity is pinned.
 */
#define _GNU_SOURCE
#include "trre_oracle.h"

#include <pthread.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ errors */

enum {
    ORC_OK = 0,
    ORC_E_SYNTAX = -1,     /* the reference prints "error: ..." and exit(1)      */
    ORC_E_UNDEFINED = -2,  /* the reference would run into undefined behaviour   */
    ORC_E_STACK = -3,      /* "error: stack max capacity reached" nft.c:551-553  */
    ORC_E_DIVERGE = -4,    /* the reference would never terminate / exhaust RAM  */
    ORC_E_NOMEM = -5
};

/* ------------------------------------------------------------------- AST   */

typedef struct {
    unsigned char type; /* '|' '.' ':' '-' '*' '+' '?' 'I' 'c' 'e', or a count  */
    unsigned char val;
    int l, r;           /* children, -1 = none                                  */
} anode;

/* ------------------------------------------------------------------- NFT   */

enum { K_PROD, K_CONS, K_SPLIT, K_SPLITNG, K_JOIN, K_FINAL }; /* nft.c:323-330 */

typedef struct {
    unsigned char kind;
    unsigned char val;
    unsigned char mark; /* the DFT closure's "visited" flag, dft.c:339          */
    int a, b;           /* nexta / nextb, -1 = NULL                             */
} nstate;

typedef struct { int head, tail; } chunk;

/* ------------------------------------------------------------ byte vectors */

typedef struct { unsigned char *b; size_t n, cap; } bvec;

static void bv_reserve(bvec *v, size_t need)
{
    if (need <= v->cap) return;
    size_t c = v->cap ? v->cap : 16;
    while (c < need) c *= 2;
    v->b = realloc(v->b, c);
    if (!v->b) abort();
    v->cap = c;
}
static void bv_push(bvec *v, unsigned char c) { bv_reserve(v, v->n + 1); v->b[v->n++] = c; }
static void bv_append(bvec *v, const unsigned char *s, size_t n)
{
    if (!n) return;
    bv_reserve(v, v->n + n);
    memcpy(v->b + v->n, s, n);
    v->n += n;
}
static bvec bv_copy(const bvec *s)
{
    bvec d = {0, 0, 0};
    bv_append(&d, s->b, s->n);
    return d;
}
static void bv_free(bvec *v) { free(v->b); v->b = NULL; v->n = v->cap = 0; }

/* ------------------------------------------------------ determinised state */

typedef struct { int st; bvec suf; } ditem;      /* (nft state, residual output) */
typedef struct { ditem *it; int n, cap; } dlist;

typedef struct {
    dlist states;             /* ordered list, dft.c:926-927                    */
    signed char fin;          /* -1 unexplored / 0 / 1, dft.c:929,938           */
    bvec fout;                /* final_out                                      */
    int next[256];            /* -1 = not a live edge                           */
    unsigned char seen[256];  /* 0 unexplored, 1 live edge, 2 explored-dead     */
    bvec out[256];
    int hnext;                /* hash chain                                     */
    unsigned hash;
} dstate;

#define ORC_HASH_BUCKETS 16384

struct trre_oracle_prog {
    int engine;
    /* AST arena */
    anode *ast; int nast, cast;
    /* NFT arena */
    nstate *ns; int nns, cns;
    int start;
    /* parser stacks (1024 deep like nft.c:38-39, but bounds-checked) */
    unsigned char ops[1024]; int nops;
    int opd[1024]; int nopd;
    /* backtracking stack, nft.c:513-523 */
    struct bt_item { int s; size_t i, o; } *bt; size_t nbt, cbt;
    bvec attempt_out;          /* the global `output`, nft.c:44-45              */
    bvec scan_out;             /* whole-buffer output under construction        */
    int all;                   /* `-a`: every accepting path prints (generator mode, NFT engine) */
    /* lazy DFT */
    dstate **ds; int nds, cds;
    int buckets[ORC_HASH_BUCKETS];
    /* error plumbing */
    jmp_buf jb;
    int ecode;
    char emsg[160];
};
typedef struct trre_oracle_prog P;

static void fail(P *p, int code, const char *msg)
{
    p->ecode = code;
    snprintf(p->emsg, sizeof p->emsg, "%s", msg);
    longjmp(p->jb, 1);
}

/* ================================================================== parser */

/* operator precedence, nft.c:11-22 */
static int prec(int c)
{
    switch (c) {
    case '|': return 1;
    case '-': return 2;
    case ':': return 3;
    case '.': return 4;
    case '?': case '*': case '+': case 'I': return 5;
    case '\\': return 6;
    }
    return -1;
}

static int ast_new(P *p, unsigned char type, int l, int r)
{
    if (p->nast == p->cast) {
        p->cast = p->cast ? p->cast * 2 : 64;
        p->ast = realloc(p->ast, (size_t)p->cast * sizeof *p->ast);
        if (!p->ast) abort();
    }
    anode *n = &p->ast[p->nast];
    n->type = type; n->val = 0; n->l = l; n->r = r;
    return p->nast++;
}
static int ast_newv(P *p, unsigned char type, unsigned char val)
{
    int id = ast_new(p, type, -1, -1);
    p->ast[id].val = val;
    return id;
}

static void push_op(P *p, unsigned char c)
{
    if (p->nops == 1024) fail(p, ORC_E_UNDEFINED, "error: operator stack overflow (undefined in the reference)");
    p->ops[p->nops++] = c;
}
static void push_opd(P *p, int n)
{
    if (p->nopd == 1024) fail(p, ORC_E_UNDEFINED, "error: operand stack overflow (undefined in the reference)");
    p->opd[p->nopd++] = n;
}
static int pop_opd(P *p)
{
    if (p->nopd == 0) fail(p, ORC_E_UNDEFINED, "error: operand stack underflow (undefined in the reference)");
    return p->opd[--p->nopd];
}

/* one reduction step, nft.c:93-108 */
static void reduce(P *p)
{
    unsigned char op = p->ops[--p->nops];
    if (op == '|' || op == '.' || op == ':' || op == '-') {
        int r = pop_opd(p);
        int l = pop_opd(p);
        push_opd(p, ast_new(p, op, l, r));
    } else if (op == '(') {
        fail(p, ORC_E_SYNTAX, "error: unmached parenthesis");
    }
    /* any other stacked symbol is dropped silently, as in the reference */
}

/* nft.c:111-115 */
static void reduce_op(P *p, unsigned char op)
{
    while (p->nops && prec(p->ops[p->nops - 1]) >= prec(op)) reduce(p);
    push_op(p, op);
}

/* postfix * + ?, nft.c:76-90 */
static void reduce_postfix(P *p, unsigned char op, int ng)
{
    int l = pop_opd(p);
    int n = ast_new(p, op, l, -1);
    p->ast[n].val = (unsigned char)ng;
    push_opd(p, n);
}

/* {m,n} — nft.c:117-156.  Returns the index of the last byte it consumed. */
static size_t parse_curly(P *p, const char *e, size_t i)
{
    int commas = 0, ng = 0, count = 0, lv = 0;
    for (; e[i]; i++) {
        unsigned char c = (unsigned char)e[i];
        if (c >= '0' && c <= '9') {
            count = count * 10 + (c - '0');
        } else if (c == ',') {
            lv = count; count = 0; commas++;
        } else if (c == '}') {
            if (e[i + 1] == '?') { ng = 1; i++; }
            if (commas == 0) lv = count;
            else if (commas > 1) fail(p, ORC_E_SYNTAX, "error: more then one comma in curly brackets");
            /* bounds live in two unsigned-char fields: wrap mod 256, nft.c:142 */
            int bounds = ast_newv(p, (unsigned char)lv, (unsigned char)count);
            int it = ast_new(p, 'I', pop_opd(p), bounds);
            p->ast[it].val = (unsigned char)ng;
            push_opd(p, it);
            return i;
        } else {
            char m[96];
            snprintf(m, sizeof m, "error: unexpected symbol in curly brackets: %c", c);
            fail(p, ORC_E_SYNTAX, m);
        }
    }
    fail(p, ORC_E_SYNTAX, "error: unmached curly brackets");
    return i;
}

/* [...] — nft.c:158-193.  Returns the index of the closing bracket. */
static size_t parse_square(P *p, const char *e, size_t i)
{
    int want_operand = 1;
    while (e[i]) {
        unsigned char c = (unsigned char)e[i];
        if (want_operand) {
            if (c == ':' || c == '-' || c == '[' || c == ']') {
                char m[96];
                snprintf(m, sizeof m, "error: unexpected symbol in square brackets: %c", c);
                fail(p, ORC_E_SYNTAX, m);
            }
            push_opd(p, ast_newv(p, 'c', c));
            want_operand = 0;
        } else if (c == ':' || c == '-') {
            reduce_op(p, c);
            want_operand = 1;
        } else if (c == ']') {
            while (p->nops && p->ops[p->nops - 1] != '[') reduce(p);
            if (!p->nops) fail(p, ORC_E_UNDEFINED, "error: bracket marker lost (undefined in the reference)");
            p->nops--;
            return i;
        } else {              /* juxtaposition inside brackets = alternation */
            reduce_op(p, '|');
            want_operand = 1;
            continue;         /* re-read this byte as an operand */
        }
        i++;
    }
    fail(p, ORC_E_SYNTAX, "error: unmached square brackets");
    return i;
}

/* nft.c:196-288 */
static int parse(P *p, const char *e)
{
    int want_operand = 1;
    size_t i = 0;
    while (e[i]) {
        unsigned char c = (unsigned char)e[i];
        if (want_operand) {
            switch (c) {
            case '(':
                push_op(p, c);
                break;
            case '[':
                push_op(p, c);
                i = parse_square(p, e, i + 1);
                want_operand = 0;
                break;
            case '\\':
                if (!e[i + 1]) fail(p, ORC_E_UNDEFINED, "error: trailing backslash (reads past the pattern in the reference)");
                i++;
                push_opd(p, ast_newv(p, 'c', (unsigned char)e[i]));
                want_operand = 0;
                break;
            case '.':          /* any byte = range 0..255, nft.c:215-221 */
                push_opd(p, ast_new(p, '-', ast_newv(p, 'c', 0), ast_newv(p, 'c', 255)));
                want_operand = 0;
                break;
            case ':':          /* implicit epsilon on the left, nft.c:222-225 */
                push_opd(p, ast_newv(p, 'e', c));
                want_operand = 0;
                continue;
            case '|': case '*': case '+': case '?': case ')': case '{': case '}':
                if (p->nops && p->ops[p->nops - 1] == ':') { /* epsilon on the right */
                    push_opd(p, ast_newv(p, 'e', c));
                    want_operand = 0;
                    continue;
                } else {
                    char m[64];
                    snprintf(m, sizeof m, "error: unexpected symbol %c", c);
                    fail(p, ORC_E_SYNTAX, m);
                }
                break;
            default:
                push_opd(p, ast_newv(p, 'c', c));
                want_operand = 0;
            }
        } else {
            switch (c) {
            case '*': case '+': case '?':
                if (e[i + 1] == '?') { reduce_postfix(p, c, 1); i++; }
                else reduce_postfix(p, c, 0);
                break;
            case '|':
                reduce_op(p, c);
                want_operand = 1;
                break;
            case ':':
                /* a ':' that ends the pattern gets its epsilon pushed BEFORE the
                 * pending operators are reduced (nft.c:254-260) */
                if (!e[i + 1]) push_opd(p, ast_newv(p, 'e', c));
                reduce_op(p, c);
                want_operand = 1;
                break;
            case '{':
                i = parse_curly(p, e, i + 1);
                break;
            case ')':
                while (p->nops && p->ops[p->nops - 1] != '(') reduce(p);
                if (!p->nops) fail(p, ORC_E_SYNTAX, "error: unmached parenthesis");
                p->nops--;
                break;
            default:           /* implicit concatenation */
                reduce_op(p, '.');
                want_operand = 1;
                continue;
            }
        }
        i++;
    }
    while (p->nops) reduce(p);
    if (!p->nopd) fail(p, ORC_E_UNDEFINED, "error: empty expression (assertion failure in the reference)");
    return p->opd[--p->nopd];
}

/* ============================================================ NFT building */

static int ns_new(P *p, int kind, int a, int b)
{
    if (p->nns == p->cns) {
        p->cns = p->cns ? p->cns * 2 : 128;
        p->ns = realloc(p->ns, (size_t)p->cns * sizeof *p->ns);
        if (!p->ns) abort();
    }
    if (p->nns > 4000000) fail(p, ORC_E_NOMEM, "error: nft too large");
    nstate *s = &p->ns[p->nns];
    s->kind = (unsigned char)kind; s->val = 0; s->mark = 0; s->a = a; s->b = b;
    return p->nns++;
}
static chunk mkchunk(int h, int t) { chunk c; c.head = h; c.tail = t; return c; }

static void link_tail(P *p, int tail, int to)
{
    if (tail < 0) fail(p, ORC_E_UNDEFINED, "error: null tail (undefined in the reference)");
    p->ns[tail].a = to;
}

/* Thompson-style construction, nft.c:375-504.  mode 0 = copy (CONS+PROD),
 * 1 = consume only (left of ':'), 2 = produce only (right of ':'). */
static chunk build(P *p, int n, int mode)
{
    if (n < 0) return mkchunk(-1, -1);
    anode an = p->ast[n];
    chunk l, r;
    int split, join;

    switch (an.type) {
    case '.':
        l = build(p, an.l, mode);
        r = build(p, an.r, mode);
        link_tail(p, l.tail, r.head);
        return mkchunk(l.head, r.tail);
    case '|':                                   /* left alternative first */
        l = build(p, an.l, mode);
        r = build(p, an.r, mode);
        split = ns_new(p, K_SPLITNG, l.head, r.head);
        join = ns_new(p, K_JOIN, -1, -1);
        link_tail(p, l.tail, join);
        link_tail(p, r.tail, join);
        return mkchunk(split, join);
    case '*':
        l = build(p, an.l, mode);
        split = ns_new(p, an.val ? K_SPLITNG : K_SPLIT, -1, l.head);
        link_tail(p, l.tail, split);
        return mkchunk(split, split);
    case '?':
        l = build(p, an.l, mode);
        join = ns_new(p, K_JOIN, -1, -1);
        split = ns_new(p, an.val ? K_SPLITNG : K_SPLIT, join, l.head);
        link_tail(p, l.tail, join);
        return mkchunk(split, join);
    case '+':
        l = build(p, an.l, mode);
        split = ns_new(p, an.val ? K_SPLITNG : K_SPLIT, -1, l.head);
        link_tail(p, l.tail, split);
        return mkchunk(l.head, split);
    case ':':
        if (p->ast[an.l].type == 'e') return build(p, an.r, 2);
        if (p->ast[an.r].type == 'e') return build(p, an.l, 1);
        l = build(p, an.l, 1);
        r = build(p, an.r, 2);
        link_tail(p, l.tail, r.head);
        return mkchunk(l.head, r.tail);
    case '-': {
        anode ln = p->ast[an.l], rn = p->ast[an.r];
        int prev = -1;
        if (ln.type == 'c' && rn.type == 'c') {           /* byte range, lowest byte first */
            join = ns_new(p, K_JOIN, -1, -1);
            for (int c = rn.val; c >= (int)ln.val; c--) {
                l = build(p, ast_newv(p, 'c', (unsigned char)c), mode);
                split = ns_new(p, K_SPLITNG, l.head, prev);
                link_tail(p, l.tail, join);
                prev = split;
            }
            return mkchunk(prev, join);
        } else if (ln.type == ':' && rn.type == ':') {    /* range of pairs, nft.c:436-452 */
            int llv = p->ast[ln.l].val, lrv = p->ast[ln.r].val, rlv = p->ast[rn.l].val;
            join = ns_new(p, K_JOIN, -1, -1);
            for (int c = rlv - llv; c >= 0; c--) {
                int pair = ast_new(p, ':', ast_newv(p, 'c', (unsigned char)(llv + c)),
                                           ast_newv(p, 'c', (unsigned char)(lrv + c)));
                l = build(p, pair, mode);
                split = ns_new(p, K_SPLITNG, l.head, prev);
                link_tail(p, l.tail, join);
                prev = split;
            }
            return mkchunk(prev, join);
        }
        fail(p, ORC_E_SYNTAX, "error: unexpected range syntax");
        return mkchunk(-1, -1);
    }
    case 'I': {                                   /* {lb,rb}, nft.c:458-485 */
        int lb = p->ast[an.r].type, rb = p->ast[an.r].val;
        int head = ns_new(p, K_JOIN, -1, -1), tail = head;
        for (int i = 0; i < lb; i++) {
            l = build(p, an.l, mode);
            link_tail(p, tail, l.head);
            tail = l.tail;
        }
        if (rb == 0) {                            /* upper bound 0 = unbounded, always greedy */
            l = build(p, ast_new(p, '*', an.l, -1), mode);
            link_tail(p, tail, l.head);
            tail = l.tail;
        } else {
            int fin = ns_new(p, K_JOIN, -1, -1);
            for (int i = lb; i < rb; i++) {
                l = build(p, an.l, mode);
                link_tail(p, tail, ns_new(p, an.val ? K_SPLITNG : K_SPLIT, fin, l.head));
                tail = l.tail;
            }
            link_tail(p, tail, fin);
            tail = fin;
        }
        return mkchunk(head, tail);
    }
    default: {                                    /* a byte ('c'), or a stray epsilon node */
        int c, o;
        if (mode == 0) {
            c = ns_new(p, K_CONS, -1, -1);
            o = ns_new(p, K_PROD, -1, -1);
            p->ns[c].val = an.val; p->ns[o].val = an.val;
            p->ns[c].a = o;
            return mkchunk(c, o);
        }
        c = ns_new(p, mode == 1 ? K_CONS : K_PROD, -1, -1);
        p->ns[c].val = an.val;
        return mkchunk(c, c);
    }
    }
}

/* nft.c:506-511; the DFT variant prepends a JOIN, dft.c:516-523 */
static void create_nft(P *p, int root)
{
    int fin = ns_new(p, K_FINAL, -1, -1);
    chunk ch = build(p, root, 0);
    link_tail(p, ch.tail, fin);
    if (p->engine == TRRE_ORACLE_DFT) p->start = ns_new(p, K_JOIN, ch.head, -1);
    else p->start = ch.head;
}

/* ==================================================== NFT attempt (backtrack) */

/* The reference doubles its stack 32,64,... and aborts when the next doubling
 * would exceed 100000 (nft.c:35-36,548-556): at most 65536 live items. */
#define ORC_BT_MAX 65536u

static void bt_push(P *p, int s, size_t i, size_t o)
{
    if (p->nbt == p->cbt) {
        if (p->cbt >= ORC_BT_MAX) fail(p, ORC_E_STACK, "error: stack max capacity reached");
        p->cbt = p->cbt ? p->cbt * 2 : 32;
        p->bt = realloc(p->bt, p->cbt * sizeof *p->bt);
        if (!p->bt) abort();
    }
    p->bt[p->nbt].s = s; p->bt[p->nbt].i = i; p->bt[p->nbt].o = o;
    p->nbt++;
}

/* One attempt at `in` (len bytes remain on the line).  On the first FINAL in
 * depth-first priority order the attempt's output — cut at its first NUL, as
 * fputs would (nft.c:644-645) — is appended to dst and the consumed count is
 * returned; -1 when every path dies.  nft.c:593-657, scan mode, all=0. */
static long nft_attempt_mode(P *p, const unsigned char *in, size_t len, bvec *dst, int match);
static long nft_attempt(P *p, const unsigned char *in, size_t len, bvec *dst)
{
    return nft_attempt_mode(p, in, len, dst, p->all ? 2 : 0);
}
/* match bit 0: `trre -m`, nft.c:635-642 — FINAL accepts only with the whole line consumed (the output is followed by
 * '\n'); anywhere else the search goes on with the next alternative.
 * match bit 1: `-a` (generator mode, nft.c:640-641,647-648: `if (!all) return i;`) — the search does not stop at an
 * accepting path: every one prints its output, in depth-first priority order, and the attempt returns -1. */
static long nft_attempt_mode(P *p, const unsigned char *in, size_t len, bvec *dst, int match)
{
    const int all = match & 2;
    match &= 1;
    size_t i = 0, o = 0;
    int s = p->start;
    bvec *out = &p->attempt_out;
    /* an attempt that never goes round an epsilon cycle produces at most one
     * cycle-free epsilon path (< nns bytes) per consumed byte; far beyond that
     * the reference is in an output-producing loop that only ends with RAM */
    const size_t out_max = 8 * (len + 1) * ((size_t)p->nns + 1) + 65536;
    p->nbt = 0;
    while (p->nbt || s >= 0) {
        if (s < 0) {
            p->nbt--;
            s = p->bt[p->nbt].s; i = p->bt[p->nbt].i; o = p->bt[p->nbt].o;
            if (s < 0) continue;
        }
        const nstate *st = &p->ns[s];
        switch (st->kind) {
        case K_CONS:
            if (i < len && st->val == in[i]) { i++; s = st->a; }
            else s = -1;
            break;
        case K_PROD:
            if (o >= out_max) fail(p, ORC_E_DIVERGE, "error: attempt output diverges");
            bv_reserve(out, o + 1);
            out->b[o++] = st->val;
            s = st->a;
            break;
        case K_SPLIT:   bt_push(p, st->a, i, o); s = st->b; break;   /* greedy: body first  */
        case K_SPLITNG: bt_push(p, st->b, i, o); s = st->a; break;   /* nexta has priority  */
        case K_JOIN:    s = st->a; break;
        case K_FINAL: {
            size_t k = 0;
            if (match && i < len) { s = -1; break; }
            while (k < o && out->b[k]) k++;
            bv_append(dst, out->b, k);
            if (match) bv_push(dst, '\n');
            if (!all) return (long)i;
            s = -1;                                    /* nft.c:650: on to the next alternative */
            break;
        }
        }
    }
    return -1;
}

/* nft.c:775-790 for one line (record minus its last byte, cut at first NUL) */
static void scan_line_nft(P *p, const unsigned char *line, size_t len, bvec *dst)
{
    size_t pos = 0;
    while (pos < len) {
        long r = nft_attempt(p, line + pos, len - pos, dst);
        if (r > 0) pos += (size_t)r;
        else bv_push(dst, line[pos++]);
    }
    nft_attempt(p, line + len, 0, dst);   /* the extra attempt on the empty tail, nft.c:788 */
    bv_push(dst, '\n');
}

/* ===================================================== lazy determinisation */

static void dl_push(dlist *l, int st, const bvec *suf)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 4;
        l->it = realloc(l->it, (size_t)l->cap * sizeof *l->it);
        if (!l->it) abort();
    }
    l->it[l->n].st = st;
    l->it[l->n].suf = bv_copy(suf);
    l->n++;
}
static void dl_free(dlist *l)
{
    for (int i = 0; i < l->n; i++) bv_free(&l->it[i].suf);
    free(l->it);
    l->it = NULL; l->n = l->cap = 0;
}

#define ORC_SUFFIX_MAX ((size_t)1 << 20)

/* Priority-ordered epsilon closure carrying the pending output, dft.c:874-907.
 * `o` is ONE mutable buffer shared with the preferred branch of every split:
 * the other branch starts from a copy taken AFTER the preferred branch has
 * returned, so it inherits whatever the preferred branch's PROD states
 * appended (dft.c:880-881, 884-885 pass `o` first and `str_copy(o)` second).
 * CONS/FINAL targets are first-writer-wins through `mark` (dft.c:893-903). */
static void closure(P *p, int s, bvec *o, int c, dlist *sl, int depth)
{
    /* a closure path longer than the automaton has gone round an epsilon cycle */
    if (depth > p->nns + 1) fail(p, ORC_E_DIVERGE, "error: epsilon cycle (unbounded recursion in the reference)");
    while (s >= 0) {
        nstate *st = &p->ns[s];
        switch (st->kind) {
        case K_SPLIT:
        case K_SPLITNG: {
            int first = st->kind == K_SPLIT ? st->b : st->a;
            int second = st->kind == K_SPLIT ? st->a : st->b;
            closure(p, first, o, c, sl, depth + 1);
            bvec cp = bv_copy(o);
            closure(p, second, &cp, c, sl, depth + 1);
            bv_free(&cp);
            return;
        }
        case K_JOIN:
            s = st->a;
            break;
        case K_PROD:
            if (o->n > ORC_SUFFIX_MAX) fail(p, ORC_E_DIVERGE, "error: closure output diverges");
            bv_push(o, st->val);
            s = st->a;
            break;
        case K_CONS:
            if (c == st->val && !st->mark) { st->mark = 1; dl_push(sl, s, o); }
            return;
        case K_FINAL:
            if (c == 0 && !st->mark) { st->mark = 1; dl_push(sl, s, o); }
            return;
        }
    }
}

/* dft.c:910-923 */
static void nft_step(P *p, const dlist *from, int c, dlist *sl)
{
    for (int k = 0; k < from->n; k++) {
        bvec o = bv_copy(&from->it[k].suf);
        closure(p, p->ns[from->it[k].st].a, &o, c, sl, 0);
        bv_free(&o);
    }
    for (int k = 0; k < sl->n; k++) p->ns[sl->it[k].st].mark = 0;
}

/* move the longest common prefix of all residuals onto the edge, dft.c:988-1010 */
static void take_lcp(dlist *sl, bvec *prefix)
{
    for (;;) {
        if (sl->it[0].suf.n == 0) return;
        unsigned char ch = sl->it[0].suf.b[0];
        for (int k = 0; k < sl->n; k++)
            if (sl->it[k].suf.n == 0 || sl->it[k].suf.b[0] != ch) return;
        bv_push(prefix, ch);
        for (int k = 0; k < sl->n; k++) {
            bvec *s = &sl->it[k].suf;
            memmove(s->b, s->b + 1, s->n - 1);
            s->n--;
        }
    }
}

static unsigned dl_hash(const dlist *l)
{
    unsigned h = 2166136261u;
    for (int k = 0; k < l->n; k++) {
        h = (h ^ (unsigned)l->it[k].st) * 16777619u;
        for (size_t j = 0; j < l->it[k].suf.n; j++) h = (h ^ l->it[k].suf.b[j]) * 16777619u;
        h = (h ^ 0xffu) * 16777619u;
    }
    return h;
}
/* identity of a determinised state = same length, same NFT states in the same
 * order, same residuals (dft.c:1028-1049 used as an equality) */
static int dl_equal(const dlist *a, const dlist *b)
{
    if (a->n != b->n) return 0;
    for (int k = 0; k < a->n; k++) {
        if (a->it[k].st != b->it[k].st) return 0;
        if (a->it[k].suf.n != b->it[k].suf.n) return 0;
        if (a->it[k].suf.n && memcmp(a->it[k].suf.b, b->it[k].suf.b, a->it[k].suf.n)) return 0;
    }
    return 1;
}

static int ds_new(P *p, dlist *states)   /* takes ownership of *states */
{
    if (p->nds == p->cds) {
        p->cds = p->cds ? p->cds * 2 : 64;
        p->ds = realloc(p->ds, (size_t)p->cds * sizeof *p->ds);
        if (!p->ds) abort();
    }
    if (p->nds > 2000000) fail(p, ORC_E_NOMEM, "error: dft too large");
    dstate *d = calloc(1, sizeof *d);
    if (!d) abort();
    d->states = *states;
    d->fin = -1;
    for (int c = 0; c < 256; c++) d->next[c] = -1;
    d->hash = dl_hash(&d->states);
    unsigned b = d->hash % ORC_HASH_BUCKETS;
    d->hnext = p->buckets[b];
    p->buckets[b] = p->nds;
    p->ds[p->nds] = d;
    memset(states, 0, sizeof *states);
    return p->nds++;
}
static int ds_lookup(P *p, const dlist *l)
{
    unsigned h = dl_hash(l);
    for (int k = p->buckets[h % ORC_HASH_BUCKETS]; k >= 0; k = p->ds[k]->hnext)
        if (p->ds[k]->hash == h && dl_equal(&p->ds[k]->states, l)) return k;
    return -1;
}

/* explore edge (d, c) — the cold branch of infer_dft, dft.c:1135-1175 */
static void explore(P *p, int di, int c)
{
    dlist sl = {0, 0, 0};
    nft_step(p, &p->ds[di]->states, c, &sl);
    if (sl.n == 0) {                       /* dead edge, dft.c:1139-1143 */
        p->ds[di]->seen[c] = 2;
        dl_free(&sl);
        return;
    }
    bvec prefix = {0, 0, 0};
    take_lcp(&sl, &prefix);
    int to = ds_lookup(p, &sl);
    if (to < 0) to = ds_new(p, &sl);
    dl_free(&sl);
    dstate *d = p->ds[di];                 /* re-read: ds_new may have moved the table */
    d->next[c] = to;
    d->out[c] = prefix;
    d->seen[c] = 1;
    dstate *t = p->ds[to];
    if (t->fin < 0) {                      /* finality probe with byte 0, dft.c:1164-1174 */
        dlist fl = {0, 0, 0};
        nft_step(p, &t->states, 0, &fl);
        if (fl.n) { t->fin = 1; t->fout = bv_copy(&fl.it[0].suf); }
        else t->fin = 0;
        dl_free(&fl);
    }
}

/* dft.c:1110-1196 (scan mode): shortest match, the start state is never final */
static long dft_attempt(P *p, const unsigned char *in, size_t len, bvec *dst)
{
    int d = 0;
    size_t i = 0;
    size_t mark = dst->n;                  /* attempt output accumulates after `mark` */
    for (; i < len; i++) {
        if (p->ds[d]->fin == 1) {
            bv_append(dst, p->ds[d]->fout.b, p->ds[d]->fout.n);
            return (long)i;
        }
        unsigned c = in[i];
        if (!p->ds[d]->seen[c]) explore(p, d, (int)c);
        dstate *ds = p->ds[d];
        if (ds->seen[c] != 1) break;
        bv_append(dst, ds->out[c].b, ds->out[c].n);
        d = ds->next[c];
    }
    if (p->ds[d]->fin == 1) {
        bv_append(dst, p->ds[d]->fout.b, p->ds[d]->fout.n);
        return (long)i;
    }
    dst->n = mark;                         /* discard, dft.c:1193-1195 */
    return -1;
}

/* dft.c:1272-1286 for one line */
static void scan_line_dft(P *p, const unsigned char *line, size_t len, bvec *dst)
{
    size_t pos = 0;
    while (pos < len) {
        long r = dft_attempt(p, line + pos, len - pos, dst);
        if (r > 0) pos += (size_t)r;
        else bv_push(dst, line[pos++]);
    }
    /* the trailing attempt on "" can never accept (start state), dft.c:1284 */
    bv_push(dst, '\n');
}

/* ================================================================ public API */

int trre_oracle_compile(const char *pattern, int engine, trre_oracle_prog **out,
                        char *err, size_t errcap)
{
    P *p = calloc(1, sizeof *p);
    if (!p) return ORC_E_NOMEM;
    p->engine = engine;
    for (int i = 0; i < ORC_HASH_BUCKETS; i++) p->buckets[i] = -1;
    if (setjmp(p->jb)) {
        int code = p->ecode;
        if (err && errcap) snprintf(err, errcap, "%s", p->emsg);
        trre_oracle_free(p);
        return code;
    }
    int root = parse(p, pattern);
    create_nft(p, root);
    if (engine == TRRE_ORACLE_DFT) {       /* dstart = [(start, "")], dft.c:1255-1259 */
        dlist init = {0, 0, 0};
        bvec empty = {0, 0, 0};
        dl_push(&init, p->start, &empty);
        ds_new(p, &init);
    }
    *out = p;
    return ORC_OK;
}

int trre_oracle_scan(trre_oracle_prog *p, const uint8_t *in, size_t n,
                     uint8_t **out, size_t *m)
{
    /* lives in the program object so that a longjmp out of an attempt cannot
     * leave a stale local copy of the buffer header */
    p->scan_out.b = NULL; p->scan_out.n = p->scan_out.cap = 0;
    bv_reserve(&p->scan_out, n + 16);
    if (setjmp(p->jb)) {
        /* the NFT binary exits through exit(), which flushes stdout: what had been printed stays
         * printed (nft.c:551-553).  The DFT binary dies of a signal with its buffer unflushed. */
        if (p->engine == TRRE_ORACLE_DFT) bv_free(&p->scan_out);
        *out = p->scan_out.b; *m = p->scan_out.n;
        p->scan_out.b = NULL; p->scan_out.n = p->scan_out.cap = 0;
        return p->ecode;
    }
    bvec *dst = &p->scan_out;
    size_t pos = 0;
    while (pos < n) {
        /* one getline() record: up to and including '\n', or to end of input */
        const unsigned char *nl = memchr(in + pos, '\n', n - pos);
        size_t reclen = nl ? (size_t)(nl - (in + pos)) + 1 : n - pos;
        size_t len = reclen - 1;                       /* line[read-1] = '\0', nft.c:777 */
        const unsigned char *z = memchr(in + pos, 0, len);
        if (z) len = (size_t)(z - (in + pos));         /* C-string walk stops at NUL   */
        if (p->engine == TRRE_ORACLE_DFT) scan_line_dft(p, in + pos, len, dst);
        else scan_line_nft(p, in + pos, len, dst);
        pos += reclen;
    }
    *out = dst->b;
    *m = dst->n;
    dst->b = NULL; dst->n = dst->cap = 0;
    return ORC_OK;
}

/* `trre -m PATTERN` (nft.c:791-797): one attempt per record over the whole line; a line that does not match
 * prints nothing.  NFT engine only. */
int trre_oracle_match(trre_oracle_prog *p, const uint8_t *in, size_t n, uint8_t **out, size_t *m)
{
    if (p->engine != TRRE_ORACLE_NFT) return ORC_E_SYNTAX;
    p->scan_out.b = NULL; p->scan_out.n = p->scan_out.cap = 0;
    bv_reserve(&p->scan_out, n + 16);
    if (setjmp(p->jb)) {
        *out = p->scan_out.b; *m = p->scan_out.n;      /* (what had been printed: see trre_oracle_scan) */
        p->scan_out.b = NULL; p->scan_out.n = p->scan_out.cap = 0;
        return p->ecode;
    }
    bvec *dst = &p->scan_out;
    size_t pos = 0;
    while (pos < n) {
        const unsigned char *nl = memchr(in + pos, '\n', n - pos);
        size_t reclen = nl ? (size_t)(nl - (in + pos)) + 1 : n - pos;
        size_t len = reclen - 1;                       /* line[read-1] = '\0', nft.c:793 */
        const unsigned char *z = memchr(in + pos, 0, len);
        if (z) len = (size_t)(z - (in + pos));
        nft_attempt_mode(p, in + pos, len, dst, p->all ? 3 : 1);
        pos += reclen;
    }
    *out = dst->b;
    *m = dst->n;
    dst->b = NULL; dst->n = dst->cap = 0;
    return ORC_OK;
}

/* `-a` (nft.c:736-738): the following scans / matches print the output of EVERY accepting path.  NFT engine only
 * (trre_dft -a prints "Not supported yet", dft.c:1227-1229). */
int trre_oracle_set_all(trre_oracle_prog *p, int all)
{
    if (p->engine != TRRE_ORACLE_NFT) return ORC_E_SYNTAX;
    p->all = all != 0;
    return ORC_OK;
}

void trre_oracle_release(uint8_t *buf) { free(buf); }

void trre_oracle_free(trre_oracle_prog *p)
{
    if (!p) return;
    for (int k = 0; k < p->nds; k++) {
        dstate *d = p->ds[k];
        dl_free(&d->states);
        bv_free(&d->fout);
        for (int c = 0; c < 256; c++) bv_free(&d->out[c]);
        free(d);
    }
    free(p->ds);
    free(p->ast);
    free(p->ns);
    free(p->bt);
    bv_free(&p->attempt_out);
    free(p);
}

int trre_oracle_nft_states(const trre_oracle_prog *p) { return p->nns; }
int trre_oracle_dft_states(const trre_oracle_prog *p) { return p->nds; }

/* ---------------------------------------------------- line-sharded threads */

typedef struct {
    const char *pattern; int engine;
    const uint8_t *in; size_t n;
    uint8_t *out; size_t m;
    int rc;
} shard;

static void *shard_main(void *arg)
{
    shard *s = arg;
    trre_oracle_prog *p = NULL;
    char err[160];
    s->rc = trre_oracle_compile(s->pattern, s->engine, &p, err, sizeof err);
    if (s->rc) return NULL;
    s->rc = trre_oracle_scan(p, s->in, s->n, &s->out, &s->m);
    trre_oracle_free(p);
    return NULL;
}

int trre_oracle_scan_mt(const char *pattern, int engine, int threads,
                        const uint8_t *in, size_t n, uint8_t **out, size_t *m)
{
    if (threads < 1) threads = 1;
    shard *sh = calloc((size_t)threads, sizeof *sh);
    pthread_t *th = calloc((size_t)threads, sizeof *th);
    size_t begin = 0;
    for (int t = 0; t < threads; t++) {
        size_t end = (t == threads - 1) ? n : n / (size_t)threads * (size_t)(t + 1);
        if (end < begin) end = begin;
        if (end < n) {                     /* move the cut to just past the next '\n' */
            const uint8_t *nl = memchr(in + end, '\n', n - end);
            end = nl ? (size_t)(nl - in) + 1 : n;
        }
        sh[t].pattern = pattern; sh[t].engine = engine;
        sh[t].in = in + begin; sh[t].n = end - begin;
        begin = end;
    }
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, shard_main, &sh[t]);
    int rc = 0;
    size_t total = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        if (sh[t].rc && !rc) rc = sh[t].rc;
        total += sh[t].m;
    }
    uint8_t *dst = NULL;
    if (!rc) {
        dst = malloc(total ? total : 1);
        size_t off = 0;
        for (int t = 0; t < threads; t++) {
            if (sh[t].m) memcpy(dst + off, sh[t].out, sh[t].m);
            off += sh[t].m;
        }
    }
    for (int t = 0; t < threads; t++) free(sh[t].out);
    free(sh); free(th);
    *out = dst; *m = rc ? 0 : total;
    return rc;
}

/* ------------------------------------------------------------------- CLI   */
#ifdef TRRE_ORACLE_MAIN
/* usage: trre_oracle [-e nft|dft] [-t threads] PATTERN [FILE]   (scan mode only) */
int main(int argc, char **argv)
{
    int engine = TRRE_ORACLE_NFT, threads = 1, a = 1;
    while (a < argc && argv[a][0] == '-' && argv[a][1] && a + 1 < argc) {
        if (!strcmp(argv[a], "-e")) { engine = !strcmp(argv[a + 1], "dft"); a += 2; }
        else if (!strcmp(argv[a], "-t")) { threads = atoi(argv[a + 1]); a += 2; }
        else break;
    }
    if (a >= argc) { fprintf(stderr, "error: missing trre expression\n"); return 1; }
    const char *pattern = argv[a++];
    FILE *fp = a < argc ? fopen(argv[a], "rb") : stdin;
    if (!fp) { fprintf(stderr, "error: can not open file %s\n", argv[a]); return 1; }
    bvec in = {0, 0, 0};
    unsigned char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, fp)) > 0) bv_append(&in, buf, k);
    uint8_t *out = NULL; size_t m = 0;
    int rc;
    if (threads > 1) {
        rc = trre_oracle_scan_mt(pattern, engine, threads, in.b, in.n, &out, &m);
        if (rc) { fprintf(stderr, "error: oracle failed (%d)\n", rc); return 1; }
    } else {
        trre_oracle_prog *p = NULL;
        char err[160];
        rc = trre_oracle_compile(pattern, engine, &p, err, sizeof err);
        if (rc) { fprintf(stderr, "%s\n", err); return 1; }
        rc = trre_oracle_scan(p, in.b, in.n, &out, &m);
        if (rc) { fprintf(stderr, "error: scan failed (%d)\n", rc); return 1; }
    }
    fwrite(out, 1, m, stdout);
    return 0;
}
#endif
