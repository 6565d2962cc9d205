"""Patterns and inputs shared by the golden generator and the parity tests."""
import random

# --- the reference's own scan-mode test table (test.sh:40-132, `S` rows) -----------------
REF_S_CASES = [
    ("xor", "x:", "or"),
    ("Mary had a little lamb", "a:", "Mry hd  little lmb"),
    ("or", ":=", "=o=r="),
    ("a", "a*", "a"),
    ("aaa", "a*", "aaa"),
    ("", ":a{,3}", "aaa"),
    ("", ":a{,3}?", ""),
    ("aaa", "(.:x)*.*", "xxx"),
    ("aaa", "(.:x)*?.*", "aaa"),
    (".", "\\.", "."),
    ("+", "\\+", "+"),
    ("?", "\\?", "?"),
    (":", "\\:", ":"),
    (".a", "\\.a", ".a"),
    ("<cat><dog>", "<(.:)*>", "<>"),
    ("<cat><dog>", "<(.:)*?>", "<><>"),
    ("<cat><dog>", "<(.:)+>", "<>"),
    ("<cat><dog>", "<(.:)+?>", "<><>"),
]

# --- the reference's match-mode table (test.sh:30-126, `M` rows = `trre -ma`): (input, pattern, FIRST expected output
# or None when nothing matches) — what `trre -m` prints for it, verified against the compiled binary by make_golden.py
REF_M_CASES = [
    ("a", "a:x", "x"), ("ab", "ab:xy", "xy"), ("ab", "(a:x)(b:y)", "xy"), ("cat", "cat:dog", "dog"), ("cat", "(cat):(dog)", "dog"),
    ("cat", "(c:d)(a:o)(t:g)", "dog"), ("mat", "c:da:ot:g", None), ("xor", "(x:)or", "or"), ("or", "(:x)or", "xor"),
    ("a", "a|b|c", "a"), ("b", "a|b|c", "b"), ("c", "a|b|c", "c"), ("b", "a*", None), ("bbb", "a*", None),
    ("abab", "(ab)*", "abab"), ("ababa", "(ab)*", None), ("a", "[a-c]", "a"), ("b", "[a-c]", "b"), ("c", "[a-c]", "c"),
    ("d", "[a-c]", None), ("a", "[a:x]", "x"), ("a", "[a:xb:y]", "x"), ("b", "[a:xb:y]", "y"), ("c", "[a:xb:y]", None),
    ("a", "[a:x-c:z]", "x"), ("b", "[a:x-c:z]", "y"), ("c", "[a:x-c:z]", "z"), ("d", "[a:x-c:z]", None), ("a", ".", "a"),
    ("b", ".", "b"), ("abc", "...", "abc"), ("aa", "a{2}", "aa"), ("a", "a{2}", None), ("aaa", "a{2}", None),
    ("", "a{,2}", ""), ("a", "a{,2}", "a"), ("aa", "a{,2}", "aa"), ("aaa", "a{,2}", None), ("", "a{1,2}", None),
    ("a", "a{1,2}", "a"), ("aa", "a{1,2}", "aa"), ("aaa", "a{1,2}", None), ("", "a{2,}", None), ("a", "a{2,}", None),
    ("aa", "a{2,}", "aa"), ("aaa", "a{2,}", "aaa"), ("", ":a{,3}", "aaa"), ("", ":a{,3}?", ""), ("aaa", "(.:x)*.*", "xxx"),
    ("aaa", "(.:x)*?.*", "aaa"), (".c", "[.]c", ".c"),
]

# whole-line patterns for match mode on the multi-line inputs
MATCH_PATTERNS = ["(cat:dog|dog:cat| |[a-z]|[A-Z]|.)*", ".*(cat:dog).*", "(.:x)*?.*", "[a-z ]*", "(a:x|b|c:|d:yy| )*", ".*", "a:*",
                  "([a-z]:W| :_)*", "\xe9*.*:!"]

# generator mode (`-a`): patterns with several parses.  Whole-line (`-ma`) and scan (`-a`) sets; inputs are short (see
# make_golden.py) because the number of outputs multiplies with every ambiguity.
GEN_MATCH_PATTERNS = ["(a|a:x)*", "(a:x|a:y|b)*", "a*a*", "(:x|:y)(a|b)*", "(a|ab)(b|:z)*", ".*", "(.:x)*?.*", "(cat:dog|cat:cow|.)*",
                      "(:a){,2}(:b)?[a-z ]*", "a{,2}a{,2}", "a:*", "(ab|a)(b|:q)|a.?"]
GEN_SCAN_PATTERNS = ["a*", "(a:x|a:y)", "(cat:dog|cat:cow|ca:C)", ":=", "(a|ab)(b|:z)", "[ab]+:N", "(:x|:y)a", "a:*", "a??b?:(1|2)",
                     "b*c?"]

# --- README examples with a stated result (README.md:39-44,57-62,123-128,180-203) ---------
README_CASES = [
    ("cat", "cat:dog", "dog"),
    ("Mary had a little lamb.", "lamb:cat", "Mary had a little cat."),
    ("cat dog", "(cat:dog|dog:cat)", "dog cat"),
    ("hello", "[a:A-z:Z]", "HELLO"),
    ("hello, world", "[a:b-y:zz:a]", "ifmmp, xpsme"),
]

# --- the BASELINE.json configuration patterns ----------------------------------------------
CONFIG_PATTERNS = ["cat:dog", "[a:A-z:Z]", "[a:b-y:zz:a]", "(cat:dog|dog:cat)"]

# --- quirk probes (SURVEY.md Q1-Q11 and the ones found while pinning the oracle) -------------
QUIRK_PATTERNS = [
    "a:xyz",                 # expansion
    "[aie]:",                # deletion
    "had a (:little )lamb",  # context insert
    "abc:2|ab:1",            # prefix overlap: NFT takes the first listed, DFT the shortest
    "ab:1|abc:2",
    "a.c:X",                 # DFT '.' finality quirk
    "x.*:y",
    "a+:x",                  # DFT shortest match
    "(c:d)(a:o)(t:g)",       # dead transition discards the attempt
    "a((:x)c|(:y)b)",        # DFT closure: fallback branch inherits the preferred branch's output
    "ab:",                   # trailing ':' binds its epsilon before reducing: a -> "b:"
    "a:b:c",
    "a{2}", "a{,2}", "a{1,2}", "a{2,}", "a{0}", "a{300}b",
    "(a|b)*c", "(ab)+:x", "a?b", "a??b", "[abc]:x", "[a-c][x-z]",
    "(a:x)*b", "((a:x)|(a:y))c", ".", "...:x", ":x", "a|:x",
    "\\[a\\]", "\xe9:e", "[\xe0:a-\xe5:a]",
    # classes of tables the kernels treat differently: keys of 5..8 bytes (64-bit window), replacement
    # texts of 5..12 and more bytes (split / pooled outputs), greedy loops (bounded fold)
    "hello:world|world:hello", "little:LITTLE", "(quick:slow|brown:red|fox:elephant)", "cat:elephants!",
    "dog:a-replacement-of-more-than-forty-bytes-0123456789", " +: ", "a*b:x", "o+:0",
]

WORDS = ["cat", "dog", "ca", "do", "cadog", "catdog", "lamb", "Mary", "had", "a", "little", "the", "quick",
         "brown", "fox", "abc", "abcd", "ab", "xab", "aaa", "xyz", "hello", "world", "<cat>", "<dog>", "or", "xor"]


def word_soup(rng, nbytes, min_len=0, max_len=120, trailing_newline=True):
    out = bytearray()
    while len(out) < nbytes:
        target = rng.randint(min_len, max_len)
        line = bytearray()
        while len(line) < target:
            line += rng.choice(WORDS).encode("latin-1")
            line += b" " if rng.random() < 0.8 else rng.choice([b",", b".", b"  ", b";", b"\t"])
        out += line[:target] + b"\n"
    if not trailing_newline and out:
        out = out[:-1]
    return bytes(out)


def printable_lines(rng, nbytes, min_len=32, max_len=160, letters=0.8):
    """the bench distribution: printable ASCII, ~80 % letters, uniform line lengths"""
    out = bytearray()
    lower = b"abcdefghijklmnopqrstuvwxyz"
    other = bytes(range(0x20, 0x61)) + b"{|}~"
    while len(out) < nbytes:
        n = rng.randint(min_len, max_len)
        out += bytes(rng.choice(lower) if rng.random() < letters else rng.choice(other) for _ in range(n)) + b"\n"
    return bytes(out)


def edge_inputs():
    rng = random.Random(7)
    return {
        "empty": b"",
        "one_newline": b"\n",
        "one_byte": b"a",
        "no_trailing_newline": b"the cat\nsat on the dog",
        "only_newlines": b"\n\n\n\n",
        "embedded_nul": b"cat\0dog cat\ndog cat\0\n\0\ncat\n",
        "nul_last_line": b"cat dog\ncat\0dog",
        "crlf": b"cat dog\r\ndog cat\r\n",
        "high_bytes": bytes(range(0x80, 0x100)) + b"\ncat\xe9dog\xe0\xe5\n",
        "all_bytes": bytes(b for b in range(1, 256) if b != 10) + b"\n",
        "long_line_5k": (b"cat dog " * 640) + b"\n" + b"tail cat\n",
        "long_line_100k": word_soup(rng, 100000, 100000, 100000)[:100000].replace(b"\n", b" ") + b"\ncat\n",
        "many_short": b"a\nb\n\nc\ncat\n" * 200,
        "words": word_soup(rng, 6000),
        "words_no_nl": word_soup(rng, 3000, trailing_newline=False),
        "printable": printable_lines(rng, 8000),
    }
