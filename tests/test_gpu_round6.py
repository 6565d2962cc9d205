"""GPU tier, round 6: the general families in ONE walk (trre_amd/csrc/one_block.hpp) — opt-in (TRRE_ONE=1, read once per
process: the scans run in a child) —, the lazy family's miss marks with several chunks in flight on one table, and the 8 GiB
outputs of the general families against the oracle on all host cores (the second half of the buffer was only compared with
the engine itself)."""
import os
import subprocess
import sys

import pytest

import golden_lib
import trre_amd
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ONE_SCRIPT = r'''
import hashlib, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, trre_amd, corpora, golden_lib
dev = torch.device("cuda", 0)
res = {}
# 1. every golden vector whose program has a small table, through the device path (aligned and not)
bad = []
n = 0
for k, (pat, name, data, eng, exp) in enumerate(golden_lib.cases()):
    if exp is None or not data or k %% 3:
        continue
    p = trre_amd.Program(pat, eng)
    if trre_amd.KERNEL_NAMES[p.info.kernel] not in ("stream_gen", "guided_gen", "stream_lp", "guided_lp"):
        continue
    gen = {"stream_lp": "stream_gen", "guided_lp": "guided_gen"}.get(trre_amd.KERNEL_NAMES[p.info.kernel], trre_amd.KERNEL_NAMES[p.info.kernel])
    fam = {v: kk for kk, v in trre_amd.KERNEL_NAMES.items()}[gen]
    if fam not in p.allowed_kernels():
        continue
    p.set_kernel(fam)
    buf = torch.frombuffer(bytearray(b"#" * 3 + data), dtype=torch.uint8).to(dev)
    for off in (3, 0):
        view = buf[off:] if off else torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        try:
            got = p.scan_tensor(view).cpu().numpy().tobytes()
        except trre_amd.TrreError as e:
            got = ("err", e.code)
        if got != exp:
            bad.append((pat, name, eng, off))
        n += 1
res["golden"] = (n, bad[:5])
# 2. 256 MiB of text: checksums of the outputs (the parent compares them with the pair's, and heads with the oracle)
inp = corpora.printable_lines(256 << 20, corpora.SEED0 + 2, dev)
out = torch.empty(inp.numel() * 2 + 4096, dtype=torch.uint8, device=dev)
for pat, eng in [("a:xyz", "dft"), (" +: ", "nft"), ("(a|b)*c:x", "nft"), ("[aie]:", "nft"), ("(a|b)*c:x", "dft"), ("[a-z]+ing:X", "dft"), (".:xy", "dft"), ("a:0123456789", "dft")]:
    p = trre_amd.Program(pat, eng)
    for off in (0, 7):
        view = inp[off:]
        p.enqueue(view, out[off:]); m = p.finish()
        res[(pat, eng, off)] = (m, hashlib.md5(out[off:off + m].cpu().numpy().tobytes()).hexdigest())
# 3. a NUL inside a long line, a buffer without a final newline, a capacity one byte short
ll = corpora.long_lines(8 << 20, corpora.SEED0 + 2, dev, 400000)
ll[3 << 20] = 0
p = trre_amd.Program(" +: ", "nft")
res["nul_long"] = hashlib.md5(p.scan_tensor(ll).cpu().numpy().tobytes()).hexdigest()
p = trre_amd.Program("a:xyz", "dft")
tail = inp[:(1 << 20) + 37]
m_full = p.scan_tensor(tail).numel()
small = torch.empty(m_full - 1, dtype=torch.uint8, device=dev)
try:
    p.enqueue(tail, small); p.finish()
    res["capacity"] = "no error"
except trre_amd.TrreError as e:
    res["capacity"] = (e.code, e.needed if hasattr(e, "needed") else None, m_full)
res["tail"] = hashlib.md5(p.scan_tensor(tail).cpu().numpy().tobytes()).hexdigest()
print(repr(res))
'''


def _child(env):
    e = dict(os.environ)
    e.pop("TRRE_ONE", None)
    e["TRRE_MAPGEN"] = "0"                # (the one-walk form and the pair: not the memoryless kernel, which takes the programs it can by default)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", ONE_SCRIPT % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"))], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=e, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return eval(r.stdout.decode().strip().splitlines()[-1]), r.stderr.decode()


def test_the_general_families_in_one_walk():
    """one_block.hpp on the device: every third golden vector with a small table (device buffers, aligned and not); 256 MiB of text
    through eight patterns — expanding, deleting, guided on both engines, one that outgrows the lanes' LDS regions ('.:xy': the launch is
    void, finish() runs the pair) and one with texts of more than 8 bytes (never tried) — byte for byte what the count / emit pair prints
    (a process of its own, TRRE_ONE unset); a NUL inside a 400 KB line; a buffer one byte short.  TRRE_TRACE shows that the form really ran:
    void launches are reported, and only the patterns that must void it do."""
    one, err1 = _child({"TRRE_ONE": "1", "TRRE_TRACE": "1"})
    pair, _ = _child({})
    assert one["golden"][0] > 300 and one["golden"][1] == [] and pair["golden"][1] == [], (one["golden"], pair["golden"])
    assert set(one) == set(pair)
    for k in one:
        if k != "golden":
            assert one[k] == pair[k], k
    assert one["capacity"][0] == trre_amd.api.E_CAPACITY
    # the void launches: '.:xy' triples every byte (a lane's 128 bytes need 384: its region holds 168); nothing else of the 256 MiB runs
    voids = [ln for ln in err1.splitlines() if "one-pass launch" in ln and "void" in ln]
    assert 1 <= len(voids) <= 40, voids[:5]
    # heads against the oracle
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import corpora
    inp = corpora.printable_lines(256 << 20, corpora.SEED0 + 2, torch.device("cuda", 0))
    head = inp[:1 << 20].cpu().numpy().tobytes()
    head = head[:head.rfind(b"\n") + 1]
    for pat, eng in [("a:xyz", "dft"), (" +: ", "nft"), ("(a|b)*c:x", "nft"), ("[aie]:", "nft")]:
        want = Oracle(pat, eng).scan(head)
        p = trre_amd.Program(pat, eng)
        got = p.scan_tensor(inp[:len(head)]).cpu().numpy().tobytes()
        assert got == want, (pat, eng)


LAZY_SCRIPT = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import random, trre_amd
from oracle_lib import Oracle
rng = random.Random(4)
# chunk 0: a few long lines that dig deep; chunks 1, 2: many short lines of other shapes — three chunks in flight on one lazily built table
deep = b"".join(bytes(rng.choice(b"ab") for _ in range(rng.randint(20000, 60000))) + b"\n" for _ in range(700))
wide = b"".join(bytes(rng.choice(b"abab c") for _ in range(rng.randint(1, 60))) + b"\n" for _ in range(2400000))
data = deep[:33 << 20] + wide[:70 << 20]
pat = "(a|b)*a(a|b){18}:x"
p = trre_amd.Program(pat, "dft")
p.set_kernel({v: k for k, v in trre_amd.KERNEL_NAMES.items()}["dft_lazy"])
got = p.scan(data)
sample = data[:2 << 20]
sample = sample[:sample.rfind(b"\n") + 1]
want = Oracle(pat, "dft").scan(sample)
tail0 = data.rfind(b"\n", 0, len(data) - (1 << 20)) + 1
want_tail = Oracle(pat, "dft").scan(data[tail0:])
print(repr((len(data), len(got), got[:len(want)] == want, got[len(got) - len(want_tail):] == want_tail)))
'''


def test_lazy_tables_with_several_chunks_in_flight():
    """trre_scan_host keeps three 32 MiB chunks in flight on ONE lazily built table; an edge that a later chunk's launch had
    marked as listed voided an earlier chunk's lanes without being listed there, and that chunk's rounds never ended.  Marks carry their
    launch's id now.  96 MiB through the host path — heterogeneous chunks, tables grown from the misses — within the timeout, head and tail
    against the oracle."""
    e = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", LAZY_SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    n, m, head_ok, tail_ok = eval(r.stdout.decode().strip().splitlines()[-1])
    assert n > (90 << 20) and m > 0 and head_ok and tail_ok


def test_dft_patterns_beyond_the_guided_tables_run_lazily():
    """a DFT pattern with eager tables, no fold and a backward automaton beyond the guided limits ran on the tile kernels
    (48 GB/s); the automatic choice is the lazily determinised family now.  The same bytes as the tile kernels and as the oracle, device and
    host paths."""
    import random
    import torch
    pat = "a(a|b|c|d|e|f|g|h){12}c:x"
    rng = random.Random(12)
    data = b"".join(bytes(rng.choice(b"abcdefgh abc") for _ in range(rng.randint(0, 300))) + b"\n" for _ in range(30000))
    p = trre_amd.Program(pat, "dft")
    assert p.info.kernel == trre_amd.KERNEL_DFT_LAZY
    want = Oracle(pat, "dft").scan(data)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    assert p.scan_tensor(t).cpu().numpy().tobytes() == want
    assert p.scan(data) == want
    p.set_kernel(trre_amd.KERNEL_TILE_GEN)
    assert p.scan_tensor(t).cpu().numpy().tobytes() == want


MAPGEN_SCRIPT = r'''
import hashlib, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, trre_amd, corpora
dev = torch.device("cuda", 0)
res = {}
inp = corpora.printable_lines(256 << 20, corpora.SEED0 + 2, dev)
out = torch.empty(inp.numel() * 3 + 4096, dtype=torch.uint8, device=dev)
for pat, eng in [("a:xyz", "dft"), ("a:xyz", "nft"), ("[aie]:", "nft"), ("(a:xyz|e:)|.:uv", "dft"), ("(<:&lt;|>:&gt;|&:&amp;)", "nft"), ("[a-z]:", "dft"), (":x", "nft"), ("e:12345678", "dft")]:
    p = trre_amd.Program(pat, eng)
    for off, size in ((0, inp.numel()), (7, (64 << 20) + 12345), (0, 1), (3, 70000)):
        view = inp[off:off + size]
        for rep in range(2):
            p.enqueue(view, out[off:]); m = p.finish()
        res[(pat, eng, off, size)] = (m, hashlib.md5(out[off:off + m].cpu().numpy().tobytes()).hexdigest())
# a NUL (the launch is void: the general family answers), a capacity one byte short
p = trre_amd.Program("a:xyz", "dft")
nul = inp[:8 << 20].clone()
nul[5 << 20] = 0
res["nul"] = hashlib.md5(p.scan_tensor(nul).cpu().numpy().tobytes()).hexdigest()
tail = inp[:(1 << 20) + 37]
m_full = p.scan_tensor(tail).numel()
small = torch.empty(m_full - 1, dtype=torch.uint8, device=dev)
try:
    p.enqueue(tail, small); p.finish()
    res["capacity"] = "no error"
except trre_amd.TrreError as e:
    res["capacity"] = e.code
print(repr(res))
'''


def test_memoryless_programs_in_one_pass_on_gpu():
    """map_block.hpp on the device (k_mapgen; TRRE_MAPGEN=1: every memoryless program, by default those that print one byte or none per byte — lengths, prefix sum with look-back, the texts at their places, one
    read and one write): eight memoryless programs on 256 MiB, 64 MiB unaligned, one byte and 70 000 bytes, each twice (the descriptors of
    the launch before are gone), a NUL in the input, a buffer one byte short — byte for byte what the count / emit pair prints (a process
    of its own with TRRE_MAPGEN=0); heads against the oracle.  (Round 6 met, here, one tile in 6 000 counted twice: the compiler had left
    the s_barrier at a loop's head without the s_waitcnt for an LDS store at the loop's end — tools/barrier_audit.py looks for that.)"""
    def child(env):
        e = dict(os.environ)
        e.pop("TRRE_MAPGEN", None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", MAPGEN_SCRIPT % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"))], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=e, timeout=1200)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        return eval(r.stdout.decode().strip().splitlines()[-1]), r.stderr.decode()
    (new, err1), (old, err0) = child({"TRRE_MAPGEN": "1", "TRRE_MAPGEN_PROF": "1"}), child({"TRRE_MAPGEN": "0", "TRRE_MAPGEN_PROF": "1"})
    # (the kernel's phase clocks, printed by finish(): it ran — and only there)
    assert err1.count("memoryless kernel") >= 60 and "memoryless kernel" not in err0
    assert set(new) == set(old) and len(new) == 34
    for k in new:
        assert new[k] == old[k], k
    assert new["capacity"] == trre_amd.api.E_CAPACITY
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import corpora
    inp = corpora.printable_lines(256 << 20, corpora.SEED0 + 2, torch.device("cuda", 0))
    head = inp[:1 << 20].cpu().numpy().tobytes()
    head = head[:head.rfind(b"\n") + 1]
    for pat, eng in [("a:xyz", "dft"), ("[aie]:", "nft"), ("(<:&lt;|>:&gt;|&:&amp;)", "nft")]:
        assert trre_amd.Program(pat, eng).scan_tensor(inp[:len(head)]).cpu().numpy().tobytes() == Oracle(pat, eng).scan(head), (pat, eng)


GIVEUP_SCRIPT = r'''
import hashlib, sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, trre_amd, corpora
dev = torch.device("cuda", 0)
inp = corpora.printable_lines(256 << 20, corpora.SEED0 + 2, dev)
p = trre_amd.Program("[aie]:", "nft")
out = p.scan_tensor(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = p.scan_tensor(inp)
torch.cuda.synchronize()
print(repr((out.numel(), hashlib.md5(out.cpu().numpy().tobytes()).hexdigest(), time.perf_counter() - t0)))
'''


def test_a_grid_that_is_not_resident_gives_up_and_the_pair_answers():
    """k_mapgen's workgroups take their tiles in turn and a tile's look-back waits for the tiles of the others: every workgroup of the grid must be
    resident.  launch_mapgen asks the runtime how many fit; if they are not (another process on the device — here: TRRE_MAPGEN_OVERSUB=3, a grid
    three times the machine), a look-back finds a tile untouched, gives up after its spin, says so in the status word, everybody still waiting
    leaves at the next poll, and finish() runs the buffer on the count / emit pair: the same bytes, a fraction of a second later — never a hang."""
    def child(env):
        e = dict(os.environ)
        for k in ("TRRE_MAPGEN", "TRRE_MAPGEN_OVERSUB"):
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", GIVEUP_SCRIPT % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"))], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=e, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        return eval(r.stdout.decode().strip().splitlines()[-1]), r.stderr.decode()
    (m1, sum1, dt1), err1 = child({"TRRE_MAPGEN": "1", "TRRE_MAPGEN_OVERSUB": "3", "TRRE_TRACE": "1"})
    (m0, sum0, dt0), err0 = child({"TRRE_MAPGEN": "0"})
    assert "memoryless kernel was void" in err1 and "was void" not in err0
    assert (m1, sum1) == (m0, sum0)
    assert dt1 < 20.0, dt1


CLIP_SCRIPT = r'''
import hashlib, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, trre_amd, corpora
dev = torch.device("cuda", 0)
inp = corpora.printable_lines(64 << 20, corpora.SEED0 + 5, dev)
res = {}
for pat, eng in [("a:xyz", "dft"), ("(<:&lt;|>:&gt;|&:&amp;)", "nft"), (".:uv", "dft"), ("e:12345678", "dft"), ("[aie]:", "nft")]:
    for off, size in ((0, inp.numel()), (5, (8 << 20) + 777)):
        p = trre_amd.Program(pat, eng)             # (a program of its own per scan: none has gone back to the pair yet)
        out = p.scan_tensor(inp[off:off + size])
        res[(pat, eng, off, size)] = (out.numel(), hashlib.md5(out.cpu().numpy().tobytes()).hexdigest())
print(repr(res))
'''


def test_tiles_that_outgrow_the_window_on_gpu():
    """k_mapgen's clipped path on the device — a tile whose output does not fit its window is expanded and stored window by window, its bytes read
    again: the first scan of a program that doubles its input ('.:uv'), and, with a window of exactly the tile's size (TRRE_MAPGEN_WINDOW=16384),
    every tile of 'a:xyz' (two windows each) — byte for byte what the pair prints."""
    def child(env):
        e = dict(os.environ)
        for k in ("TRRE_MAPGEN", "TRRE_MAPGEN_WINDOW"):
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", CLIP_SCRIPT % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"))], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=e, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        return eval(r.stdout.decode().strip().splitlines()[-1])
    pair = child({"TRRE_MAPGEN": "0"})
    assert child({"TRRE_MAPGEN": "1"}) == pair
    assert child({"TRRE_MAPGEN": "1", "TRRE_MAPGEN_WINDOW": "16384"}) == pair
    assert child({"TRRE_MAPGEN": "1", "TRRE_MAPGEN_WINDOW": "8192"}) == pair
