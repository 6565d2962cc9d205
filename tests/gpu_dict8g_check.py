#!/usr/bin/env python3
"""Run by tests/test_gpu_fullsize.py::test_dictionary_8gib in a subprocess (TRRE_NO_FB_COPY / TRRE_NO_FB / TRRE_FB_EMIT select the
walkers of the large table; the library reads them once per process).

8 GiB of config 5's corpus through the 1000-entry dictionary, both engines.  No CPU can check 8 GiB (the
reference does 0.08 GB/s with the DFT binary and 0.3 MB/s with the NFT one), so:
  * slices of >= 4 MiB at the head, around the 4 GiB mark of the INPUT and at the tail are scanned on their own,
    checked against the oracle, and must reappear in the full output at the offset the scan of everything before
    them produces (prefix scans give the offsets; lines are independent, so the pieces concatenate);
  * the keys are prefix-free, so the two engines print the same bytes: the NFT engine's output is
    compared with the DFT oracle on those slices (the NFT oracle covers 128 KiB of the head), and the two engines'
    full outputs with each other."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import corpora  # noqa: E402
import dictgen  # noqa: E402
import trre_amd  # noqa: E402
from oracle_lib import Oracle, scan_mt  # noqa: E402

N = 8 << 30
SLICE = 4 << 20


def line_start(inp, pos):
    if pos <= 0:
        return 0
    nl = (inp[pos - 1: pos - 1 + (1 << 20)] == 10).nonzero()
    return pos + int(nl[0]) if nl.numel() else inp.numel()


def main():
    keys, vals = dictgen.make_dictionary(1000)
    pat = dictgen.pattern(keys, vals)
    inp = corpora.dictionary_soup(N, corpora.SEED0 + 5, "cuda", keys)
    out = torch.empty(N + N // 8, dtype=torch.uint8, device="cuda")
    tmp = torch.empty(N // 2 + N // 16 + (8 << 20), dtype=torch.uint8, device="cuda")
    dft_oracle = Oracle(pat, "dft")
    bad = 0
    first_total = None
    digest = None
    for eng in ("dft", "nft"):
        p = trre_amd.Program(pat, eng)
        assert p.info.kernel == trre_amd.KERNEL_STREAM_GEN, p.info.kernel
        m = p.scan_tensor(inp, out=out).numel()
        # slices: [s, e) of the input -> where they sit in the output = size of the scan of inp[:s]
        marks = [0, line_start(inp, (4 << 30) - SLICE // 2), line_start(inp, (4 << 30) + 12345), line_start(inp, N - SLICE)]
        for s in marks:
            e = line_start(inp, min(N, s + SLICE)) if s + SLICE < N else N
            at = 0 if s == 0 else None
            if at is None:
                # the prefix does not fit `tmp` in one piece beyond half of the buffer: sum two scans
                at = 0
                for lo, hi in ((0, min(s, line_start(inp, N // 2))), (min(s, line_start(inp, N // 2)), s)):
                    if hi > lo:
                        at += p.scan_tensor(inp[lo:hi], out=tmp).numel()
            piece = p.scan_tensor(inp[s:e], out=tmp)
            want = dft_oracle.scan(inp[s:e].cpu().numpy().tobytes())
            if piece.cpu().numpy().tobytes() != want:
                print("MISMATCH oracle", eng, s)
                bad += 1
            if not torch.equal(out[at:at + piece.numel()], piece):
                print("MISMATCH placement", eng, s, at)
                bad += 1
            if s == marks[-1] and at + piece.numel() != m:
                print("MISMATCH total", eng, at + piece.numel(), m)
                bad += 1
        if eng == "nft" and not os.environ.get("TRRE_NO_FB") and not os.environ.get("TRRE_FB_EMIT"):
            # (the NFT engine was checked against the DFT oracle and a 128 KiB head of its own.)  64 MiB right above the
            # 4 GiB mark against the NFT oracle on all host cores (0.3 MB/s per core on this 24 kB pattern: line-sharded threads), placed
            # in the full output by a prefix scan — once per process family (the alternative walkers of the large table: the slices above)
            s = line_start(inp, (4 << 30) + (64 << 20))
            e2 = line_start(inp, s + (64 << 20))
            at = 0
            for lo, hi in ((0, line_start(inp, N // 2)), (line_start(inp, N // 2), s)):
                if hi > lo:
                    at += p.scan_tensor(inp[lo:hi], out=tmp).numel()
            want = scan_mt(pat, "nft", os.cpu_count() or 1, inp[s:e2].cpu().numpy().tobytes())
            if out[at:at + len(want)].cpu().numpy().tobytes() != want:
                print("MISMATCH nft all-cores oracle", s, at)
                bad += 1
        if eng == "nft":
            e = line_start(inp, 128 << 10)
            if out[:len(Oracle(pat, "nft").scan(inp[:e].cpu().numpy().tobytes()))].cpu().numpy().tobytes() != Oracle(pat, "nft").scan(inp[:e].cpu().numpy().tobytes()):
                print("MISMATCH nft oracle head")
                bad += 1
        # the two engines agree on the whole output (sizes, and a checksum of 64-bit words of the first m bytes)
        words = out[: m & ~7].view(torch.int64)
        d = (int(words.sum().item()), int((words[::3] ^ (words[::3] >> 7)).sum().item()), m)
        if first_total is None:
            first_total, digest = m, d
        elif d != digest:
            print("MISMATCH engines", d, digest)
            bad += 1
        p.close()
    print("dict 8 GiB check: %s (output %d bytes)" % ("ok" if not bad else "%d mismatches" % bad, first_total))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
