"""Seeded synthetic corpora of SURVEY.md §8(d), generated ON THE DEVICE (torch is plumbing here).

    printable_lines   cfg 2 / cfg 3: printable ASCII 0x20-0x7E, ~80 % letters, lines of 32..160 bytes
    token_soup        cfg 4 / cfg 5: space-separated tokens drawn from a weighted vocabulary, a newline
                      instead of the space after every 8..24 tokens (>= 64 M lines in 8 GiB)
    cat_dog_soup      cfg 4: ~10 % of the tokens are 'cat' / 'dog', plus the near-misses 'ca', 'do', 'cadog', ...
    dictionary_soup   cfg 5: 30 % of the tokens are keys of the seeded 1000-entry dictionary (tools/dictgen.py)

Every buffer ends with '\\n' and holds no NUL (Q1/Q2 are covered by the fixtures, not by the timed path).
Seeds follow SURVEY §8(d): 0x7472726531 + config index (+ rank for per-rank shards)."""
import torch

SEED0 = 0x7472726531

FILLER = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "lorem", "ipsum", "dolor", "sit", "amet", "a", "of",
          "and", "mary", "had", "little", "lamb", "concatenate", "dogma", "scatter", "x"]
CAT_DOG = [("cat", 5.0), ("dog", 5.0), ("ca", 2.0), ("do", 2.0), ("cadog", 1.0), ("catdog", 1.0), ("dogcat", 0.5),
           ("og", 0.5), ("at", 0.5)]


def printable_lines(n, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    data = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 28
    for lo in range(0, n, step):
        k = min(step, n - lo)
        kind = torch.randint(0, 100, (k,), dtype=torch.uint8, device=device, generator=g)
        lower = torch.randint(97, 123, (k,), dtype=torch.uint8, device=device, generator=g)
        other = torch.randint(0x20, 0x7f, (k,), dtype=torch.uint8, device=device, generator=g)
        data[lo:lo + k] = torch.where(kind < 70, lower, torch.where(kind < 80, lower - 32, other))
        del kind, lower, other
    lens = torch.randint(33, 162, (n // 64 + 2,), device=device, generator=g)   # line length + newline
    ends = torch.cumsum(lens, 0) - 1
    data[ends[ends < n]] = 10
    data[n - 1] = 10
    return data


def token_soup(n, seed, device, vocab, weights, block_tokens=1 << 24):
    """n bytes of `vocab` tokens (list of str/bytes) drawn with `weights`, see module docstring."""
    g = torch.Generator(device=device).manual_seed(seed)
    toks = [t.encode() if isinstance(t, str) else bytes(t) for t in vocab]
    width = max(len(t) for t in toks) + 1
    table = torch.zeros((len(toks), width), dtype=torch.uint8)
    lens = torch.zeros(len(toks), dtype=torch.int64)
    for i, t in enumerate(toks):
        table[i, :len(t)] = torch.frombuffer(bytearray(t), dtype=torch.uint8)
        table[i, len(t)] = 32
        lens[i] = len(t) + 1
    table, lens = table.to(device), lens.to(device)
    w = torch.tensor(weights, dtype=torch.float32, device=device)
    cols = torch.arange(width, device=device)[None, :]
    data = torch.empty(n, dtype=torch.uint8, device=device)
    filled = 0
    while filled < n:
        ids = torch.multinomial(w, block_tokens, replacement=True, generator=g)
        rows = table[ids]
        ln = lens[ids]
        gaps = torch.randint(8, 25, (block_tokens // 8 + 2,), device=device, generator=g)
        ends = torch.cumsum(gaps, 0)
        ends = ends[ends < block_tokens]
        rows[ends, ln[ends] - 1] = 10
        flat = rows[cols < ln[:, None]]
        k = min(flat.numel(), n - filled)
        data[filled:filled + k] = flat[:k]
        filled += k
        del ids, rows, ln, flat
    data[n - 1] = 10
    return data


def cat_dog_soup(n, seed, device):
    vocab = [t for t, _ in CAT_DOG] + FILLER
    rest = 100.0 - sum(wt for _, wt in CAT_DOG)
    weights = [wt for _, wt in CAT_DOG] + [rest / len(FILLER)] * len(FILLER)
    return token_soup(n, seed, device, vocab, weights)


def dictionary_soup(n, seed, device, keys, key_fraction=0.3):
    vocab = list(keys) + FILLER
    weights = [key_fraction / len(keys)] * len(keys) + [(1.0 - key_fraction) / len(FILLER)] * len(FILLER)
    return token_soup(n, seed, device, vocab, weights)


def long_lines(n, seed, device, line_len):
    """cfg 2's text with most of its line ends turned into spaces: one line end survives per `line_len` bytes (JSON lines, minified files)"""
    data = printable_lines(n, seed, device)
    idx = (data == 10).nonzero().flatten()
    w = idx // line_len
    drop = torch.ones_like(idx, dtype=torch.bool)
    drop[1:] = w[1:] == w[:-1]
    drop[0] = False
    data[idx[drop]] = 32
    data[n - 1] = 10
    return data


def by_name(name, n, seed, device):
    if name == "printable":
        return printable_lines(n, seed, device)
    if name.startswith("long"):
        return long_lines(n, seed, device, int(name[4:]))
    if name == "catdog":
        return cat_dog_soup(n, seed, device)
    if name.startswith("dict"):
        import dictgen
        keys, _ = dictgen.make_dictionary(int(name[4:] or 1000))
        return dictionary_soup(n, seed, device, keys)
    raise ValueError(name)
