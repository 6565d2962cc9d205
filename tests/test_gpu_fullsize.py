"""BASELINE.json's full sizes on the GPU: 8 GiB buffers, i.e. byte offsets beyond 2^32 through every kernel
family that accepts the pattern.  No full-size CPU run is possible (the reference does ~0.1 GB/s), so the
checks are size-independent properties and oracle-checked slices — at the head, right above the 4 GiB
mark and at the tail of the buffer — as tests/test_gpu_parity.py does at 1 GiB for configs[1]."""
import os
import sys

import pytest

import trre_amd
from oracle_lib import Oracle, scan_mt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
N = 8 << 30
SLICE = 2 << 20
LP_FAMILIES = (trre_amd.KERNEL_BYTEMAP, trre_amd.KERNEL_TILE_LP, trre_amd.KERNEL_STREAM_LP, trre_amd.KERNEL_GUIDED_LP)


def line_start(inp, pos):
    if pos <= 0:
        return 0
    nl = (inp[pos - 1: pos - 1 + (1 << 20)] == 10).nonzero()
    return pos + int(nl[0]) if nl.numel() else inp.numel()


def check_slices(p, oracle, inp, out, m, fam, full_oracle=None):
    """length-preserving families: output offset == input offset at line starts, so slices compare directly;
    general families: a slice of the input is scanned on its own and must reappear in the full output at the
    offset the scan of everything before it produces — checked for the two halves around the 4 GiB mark"""
    import torch
    n = inp.numel()
    cut = line_start(inp, (4 << 30) + 777)
    if fam in LP_FAMILIES:
        assert m == n
        for s in (0, line_start(inp, (4 << 30) - SLICE), cut, line_start(inp, n - SLICE)):
            e = min(n, line_start(inp, min(n, s + SLICE)))
            want = oracle.scan(inp[s:e].cpu().numpy().tobytes())
            assert out[s:e].cpu().numpy().tobytes() == want, (fam, s)
        return
    tmp = torch.empty(inp.numel() // 2 + inp.numel() // 8 + (4 << 20), dtype=torch.uint8, device="cuda")
    at = 0
    second_half_at = None
    for lo, hi in ((0, cut), (cut, n)):
        part = p.scan_tensor(inp[lo:hi], out=tmp)
        k = part.numel()
        assert torch.equal(out[at:at + k], part), (fam, lo)
        e = line_start(inp, lo + SLICE)
        want = oracle.scan(inp[lo:e].cpu().numpy().tobytes())
        assert part[:len(want)].cpu().numpy().tobytes() == want, (fam, lo)
        if lo:
            second_half_at = at
        at += k
    assert at == m
    # ... and the FULL output's second half against the oracle itself, not against the engine: every byte that the first
    # GiB of input above the 4 GiB mark becomes, checked by the oracle on all host cores (line-sharded threads; two slabs of 512 MiB)
    if full_oracle is not None:
        pattern, engine = full_oracle
        pos, opos = cut, second_half_at
        for _ in range(2):
            end = line_start(inp, min(n, pos + (512 << 20)))
            want = scan_mt(pattern, engine, os.cpu_count() or 1, inp[pos:end].cpu().numpy().tobytes())
            assert out[opos:opos + len(want)].cpu().numpy().tobytes() == want, (fam, "all-cores oracle", pos)
            pos, opos = end, opos + len(want)


def test_caesar_8gib_all_families():
    """BASELINE configs[2]: '[a:b-y:zz:a]' DFT over 8 GiB; the whole output against an independent torch byte map"""
    import torch
    import corpora
    inp = corpora.printable_lines(N, corpora.SEED0 + 3, "cuda")
    want = torch.where((inp >= 97) & (inp <= 121), inp + 1, torch.where(inp == 122, torch.full_like(inp, 97), inp))
    p = trre_amd.Program("[a:b-y:zz:a]", "dft")
    out = torch.empty(N + N // 8, dtype=torch.uint8, device="cuda")
    for fam in [trre_amd.KERNEL_AUTO] + p.allowed_kernels():
        if fam == trre_amd.KERNEL_TILE_GEN:
            continue                       # (40 GB/s class and two more 8 GiB workspaces: covered at 1 GiB)
        p.set_kernel(fam)
        got = p.scan_tensor(inp, out=out)
        assert got.numel() == N and torch.equal(got, want), fam
    p.set_kernel(trre_amd.KERNEL_AUTO)


def test_cfg4_8gib_all_families():
    """BASELINE configs[3]: '(cat:dog|dog:cat)' NFT over 8 GiB of its own corpus (word soup, ~10 % cat/dog tokens and
    near-misses, > 64 M lines), every family the pattern admits, oracle-checked slices around the 4 GiB mark"""
    import torch
    import corpora
    inp = corpora.cat_dog_soup(N, corpora.SEED0 + 4, "cuda")
    assert int((inp == 10).sum()) >= 64 << 20
    p = trre_amd.Program("(cat:dog|dog:cat)", "nft")
    oracle = Oracle("(cat:dog|dog:cat)", "nft")
    out = torch.empty(N + N // 8, dtype=torch.uint8, device="cuda")
    fams = [trre_amd.KERNEL_AUTO] + [f for f in p.allowed_kernels() if f != trre_amd.KERNEL_TILE_GEN]
    for fam in fams:
        p.set_kernel(fam)
        got = p.scan_tensor(inp, out=out)
        check_slices(p, oracle, inp, out, got.numel(), p.info.kernel, ("(cat:dog|dog:cat)", "nft") if fam == trre_amd.KERNEL_AUTO else None)
    p.set_kernel(trre_amd.KERNEL_AUTO)


def test_general_families_8gib():
    """variable-length output beyond 2^32 bytes: an expanding DFT pattern (stream family; a memoryless program: its first scan — the full
    buffer, compared with the oracle's slices — is k_mapgen's, the half scans after it the pair's, map_block.hpp), an NFT pattern that only the
    guided family runs, and a deleting memoryless program (k_mapgen throughout)"""
    import torch
    import corpora
    inp = corpora.printable_lines(N, corpora.SEED0 + 2, "cuda")
    out = torch.empty(N + N // 4, dtype=torch.uint8, device="cuda")
    for pat, eng in (("a:xyz", "dft"), ("(a|b)*c:x", "nft"), ("[aie]:", "dft")):
        p = trre_amd.Program(pat, eng)
        got = p.scan_tensor(inp, out=out)
        check_slices(p, Oracle(pat, eng), inp, out, got.numel(), p.info.kernel, (pat, eng))


@pytest.mark.parametrize("env", [{}, {"TRRE_NO_FB_MARK4": "1"}, {"TRRE_NO_FB_COPY": "1"}, {"TRRE_NO_FB": "1"}, {"TRRE_FB_EMIT": "1"}])
def test_dictionary_8gib(env):
    """BASELINE configs[4] at its per-GPU size: the 1000-entry dictionary over 8 GiB of its own corpus (offsets
    beyond 2^32 through the large-table kernels), both engines, the automatic choice (the copy form: mark pass on the 32-bit comb, wave-cooperative
    splice), round 3's first pass (TRRE_NO_FB_MARK4) and the three alternative walkers of the large table (the library reads the environment once per process: subprocess)"""
    import subprocess
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_dict8g_check.py")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, script], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert r.returncode == 0, r.stdout.decode("latin-1")[-3000:]
