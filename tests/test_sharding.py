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
