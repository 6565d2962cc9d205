"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups run the
line-sharded scan with the oracle standing in for the per-GPU scan (no GPU in
this tier); the concatenation of the shard outputs must equal the unsharded
result, for inputs with and without a trailing newline."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import corpus


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data, pattern, engine, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle_lib import Oracle
    from trre_amd.sharded import scan_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = Oracle(pattern, engine)
        off, out, total = scan_sharded(data, o.scan)
        q.put((rank, off, out, total))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_line_sharded_scan_equals_unsharded(world):
    from oracle_lib import Oracle
    rng = random.Random(17)
    cases = [("(cat:dog|dog:cat)", "nft", corpus.word_soup(rng, 20000)),
             ("a:xyz", "dft", corpus.word_soup(rng, 9000, trailing_newline=False)),
             ("[aie]:", "nft", b"one line only, no newline")]
    ctx = mp.get_context("spawn")
    for pattern, engine, data in cases:
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, data, pattern, engine, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        want = Oracle(pattern, engine).scan(data)
        buf = bytearray(len(want))
        for rank, off, out, total in got:
            assert total == len(want)
            buf[off:off + len(out)] = out
        assert bytes(buf) == want, (pattern, engine, world)


def test_reassembly_of_eight_shards_inside_the_c_abi():
    """trre_scan_host_multi's own sharding and reassembly (round 5: a function of its own, driven here with a stand-in for the
    per-shard device call — the oracle): 8 and 24 shards (8 devices x TRRE_SHARDS_PER_DEVICE 3), variable-length and
    length-preserving outputs, a shard that asks for room, a shard on which the reference stops (TRRE_E_DIVERGES: its partial output is
    the last thing that counts), a size query."""
    import ctypes
    import threading
    import trre_amd
    from trre_amd import api
    from oracle_lib import Oracle
    L = api.lib()
    FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_size_t, ctypes.POINTER(ctypes.c_ubyte),
                          ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t))
    L.trre_debug_scan_host_multi.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int,
                                             ctypes.c_int, FN, ctypes.c_void_p]
    rng = random.Random(23)
    data = corpus.word_soup(rng, 200000) + b"cat\0dog\nlast line without newline"

    def run(pattern, engine, n_shards, fixed_len, stop_in=None, cap=None):
        o = Oracle(pattern, engine)
        seen = []
        lock = threading.Lock()                  # (the shards arrive on threads of their own; the oracle is one C object with lazily built tables)

        def shard(user, g, inp, n, out, cap_, m):
            seen.append(g)
            chunk = ctypes.string_at(inp, n)
            with lock:
                res = o.scan(chunk)
            rc = 0
            if stop_in is not None and g == stop_in:             # the reference stops in the middle of this shard
                res = res[:len(res) // 2]
                rc = api.E_DIVERGES
            m[0] = len(res)
            if len(res) > cap_:
                return api.E_CAPACITY
            ctypes.memmove(out, res, len(res))
            return rc
        want_cap = cap if cap is not None else 4 * len(data)
        out = ctypes.create_string_buffer(max(want_cap, 1))
        m = ctypes.c_size_t()
        rc = L.trre_debug_scan_host_multi(data, len(data), out, want_cap, ctypes.byref(m), n_shards, 1 if fixed_len else 0, FN(shard), None)
        return rc, out.raw[:min(m.value, want_cap)], m.value, seen

    for n_shards in (8, 24):
        for pattern, engine, fixed in [("a:xyz", "dft", False), ("(cat:dog|dog:cat)", "nft", True), ("[aie]:", "nft", False)]:
            want = Oracle(pattern, engine).scan(data)
            rc, got, m, seen = run(pattern, engine, n_shards, fixed)
            assert rc == 0 and got == want and sorted(set(seen)) == list(range(n_shards)), (pattern, n_shards)
            # a size query, then a buffer that is too small by one byte
            rc, _, m, _ = run(pattern, engine, n_shards, fixed, cap=0)
            assert rc == api.E_CAPACITY and m == len(want)
            rc, _, m, _ = run(pattern, engine, n_shards, fixed, cap=len(want) - 1)
            assert rc == api.E_CAPACITY and m == len(want)
            rc, got, m, _ = run(pattern, engine, n_shards, fixed, cap=len(want))
            assert rc == 0 and got == want
            # the reference stops inside shard 5: shards 0..4 whole, half of shard 5's output, nothing behind it
            b = trre_amd.shard_bounds(data, n_shards)
            o = Oracle(pattern, engine)
            head = b"".join(o.scan(data[b[g]:b[g + 1]]) for g in range(5))
            part = o.scan(data[b[5]:b[6]])
            rc, got, m, _ = run(pattern, engine, n_shards, fixed, stop_in=5)
            assert rc == api.E_DIVERGES and got == head + part[:len(part) // 2], (pattern, n_shards)
