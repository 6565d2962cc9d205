"""Seeded 1k-entry key:value dictionary pattern and matching corpus (BASELINE config 5):
1000 random lowercase keys of length 3-8 made prefix-free (so that the NFT and DFT engines
agree), values of length 3-8, joined by '|'."""
import random


def make_dictionary(n=1000, seed=0x7472726535):
    rng = random.Random(seed)
    keys = []
    seen = set()
    while len(keys) < n:
        k = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(3, 8)))
        if any(k.startswith(s) or s.startswith(k) for s in seen):
            continue
        seen.add(k)
        keys.append(k)
    vals = ["".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(3, 8))) for _ in keys]
    return keys, vals


def pattern(keys, vals):
    return "|".join("%s:%s" % kv for kv in zip(keys, vals))


def corpus(keys, nbytes, seed=1, key_fraction=0.3):
    rng = random.Random(seed)
    out = bytearray()
    words = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "lorem", "ipsum", "dolor", "sit"]
    while len(out) < nbytes:
        line = bytearray()
        target = rng.randint(32, 160)
        while len(line) < target:
            w = rng.choice(keys) if rng.random() < key_fraction else rng.choice(words)
            line += w.encode() + b" "
        out += line[:target] + b"\n"
    return bytes(out)


def corpus_fast(keys, nbytes, seed=1, key_fraction=0.3):
    """numpy-vectorised variant for large inputs: tokens (keys 30 %, filler words 70 %) separated
    by spaces, every 8th..24th separator a newline; ends with a newline."""
    import numpy as np
    rng = np.random.default_rng(seed)
    words = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "lorem", "ipsum", "dolor", "sit"]
    toks = [k.encode() for k in keys] + [w.encode() for w in words]
    L = max(len(t) for t in toks) + 1
    M = np.zeros((len(toks), L), dtype=np.uint8)
    ln = np.zeros(len(toks), dtype=np.int64)
    for i, t in enumerate(toks):
        M[i, :len(t)] = np.frombuffer(t, dtype=np.uint8)
        M[i, len(t)] = 32
        ln[i] = len(t) + 1
    n_tok = int(nbytes / 6.2) + 16
    is_key = rng.random(n_tok) < key_fraction
    ids = np.where(is_key, rng.integers(0, len(keys), n_tok), len(keys) + rng.integers(0, len(words), n_tok))
    rows = M[ids]
    lens = ln[ids]
    # newline instead of the separating space after a random 8..24 tokens
    gaps = rng.integers(8, 25, n_tok // 8 + 2)
    ends = np.cumsum(gaps)
    ends = ends[ends < n_tok]
    rows[ends, lens[ends] - 1] = 10
    mask = np.arange(L)[None, :] < lens[:, None]
    flat = rows[mask]
    flat = flat[:nbytes].copy()
    flat[-1] = 10
    return flat.tobytes()
