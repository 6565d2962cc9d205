"""Multi-GPU inside the C ABI and thread safety of a compiled program: line sharding,
one host thread + streams per device, per-device scan state.  The boxes these run on have ONE GPU, so the
sharding is forced to several shards per device (TRRE_SHARDS_PER_DEVICE); the collective-free N-process
path is covered on CPU by tests/test_sharding.py."""
import os
import random
import subprocess
import sys
import threading

import pytest

import corpus
import trre_amd
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("shards", ["1", "4", "7"])
def test_scan_host_multi_shards_and_reassembles(shards):
    e = dict(os.environ, TRRE_SHARDS_PER_DEVICE=shards)
    r = subprocess.run([sys.executable, os.path.join(HERE, "gpu_multi_check.py")], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode("latin-1")[-2000:]


def test_two_threads_one_handle():
    """several host threads drive ONE compiled program on the same device at once (host buffers and device
    buffers): calls are serialised per (prog, device) and every result is right"""
    import torch
    rng = random.Random(8)
    datas = [corpus.word_soup(rng, 200000 + 50000 * k) for k in range(4)]
    for pat, eng in [("(cat:dog|dog:cat)", "nft"), ("a:xyz", "dft"), ("(a|b)*c:x", "nft")]:
        p = trre_amd.Program(pat, eng)
        wants = [Oracle(pat, eng).scan(d) for d in datas]
        errors = []

        def host_worker(k):
            try:
                for _ in range(6):
                    if p.scan(datas[k]) != wants[k]:
                        errors.append(("host", k))
            except Exception as ex:       # noqa: BLE001
                errors.append(("host", k, repr(ex)))

        def device_worker(k):
            try:
                s = torch.cuda.Stream()
                t = torch.frombuffer(bytearray(datas[k]), dtype=torch.uint8).cuda()
                torch.cuda.synchronize()
                for _ in range(6):
                    with torch.cuda.stream(s):
                        got = p.scan_tensor(t, stream=s.cuda_stream)
                    s.synchronize()
                    if got.cpu().numpy().tobytes() != wants[k]:
                        errors.append(("device", k))
            except Exception as ex:       # noqa: BLE001
                errors.append(("device", k, repr(ex)))

        threads = [threading.Thread(target=host_worker, args=(k,)) for k in range(2)]
        threads += [threading.Thread(target=device_worker, args=(k,)) for k in range(2, 4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, (pat, eng, errors[:3])


def test_scan_sharded_with_the_real_program():
    """trre_amd.sharded.scan_sharded (the one-process-per-GPU path of bench.py --gpus N) with the real
    Program.scan_tensor as the per-rank scan: world 1 here, the N > 1 arithmetic is in tests/test_sharding.py"""
    import socket
    import torch
    import torch.distributed as dist
    from trre_amd.sharded import scan_sharded
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rng = random.Random(2)
        data = corpus.word_soup(rng, 500000)
        for pat, eng in [("(cat:dog|dog:cat)", "nft"), ("a:xyz", "dft")]:
            p = trre_amd.Program(pat, eng)

            def scan_fn(b):
                t = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
                return p.scan_tensor(t).cpu().numpy().tobytes()

            off, out, total = scan_sharded(data, scan_fn)
            assert off == 0 and out == Oracle(pat, eng).scan(data) and total == len(out)
    finally:
        dist.destroy_process_group()


def test_enqueue_rejects_a_different_scan_in_flight():
    """the split form batches identical launches; a different scan before the finish is an error"""
    import torch
    p = trre_amd.Program("[a:A-z:Z]", "dft")
    a = torch.frombuffer(bytearray(b"hello\nworld\n" * 1000), dtype=torch.uint8).cuda()
    b = torch.frombuffer(bytearray(b"other\nbytes\n" * 1000), dtype=torch.uint8).cuda()
    out = torch.empty(a.numel() + 64, dtype=torch.uint8, device="cuda")
    p.enqueue(a, out)
    p.enqueue(a, out)                       # the same scan again: a batch
    with pytest.raises(trre_amd.TrreError) as e:
        p.enqueue(b, out)
    assert e.value.code == trre_amd.api.E_ARG
    assert p.finish() == a.numel()
    assert out[:a.numel()].cpu().numpy().tobytes() == b"HELLO\nWORLD\n" * 1000
    p.enqueue(b, out)
    assert p.finish() == b.numel()
