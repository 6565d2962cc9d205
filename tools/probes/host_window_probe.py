#!/usr/bin/env python3
"""Why do the command line's 256 MiB scan calls run at half the rate of one 1 GiB trre_scan_host call?  The same calls from here, by
source (a window of a mapped /dev/shm file, pages populated / a private array) and destination (a buffer used before / a fresh one)."""
import ctypes, mmap, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import trre_amd, corpora

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2 << 30
W = 256 << 20
f = "/dev/shm/trre_window_probe.txt"
corpora.printable_lines(N, corpora.SEED0 + 2, torch.device("cuda", 0)).cpu().numpy().tofile(f)
L = trre_amd.api.lib()
p = trre_amd.Program("[a:A-z:Z]", "dft")
m = ctypes.c_size_t()
libc = ctypes.CDLL(None, use_errno=True)

def call(src_addr, n, dst):
    t0 = time.perf_counter()
    rc = L.trre_scan_host(p._h, ctypes.c_char_p(src_addr), n, dst.ctypes.data_as(ctypes.c_char_p), dst.size, ctypes.byref(m), 0)
    assert rc == 0, rc
    return time.perf_counter() - t0

fd = os.open(f, os.O_RDONLY)
mm = mmap.mmap(fd, N, flags=mmap.MAP_PRIVATE, prot=mmap.PROT_READ)
base = ctypes.addressof(ctypes.c_char.from_buffer_copy(b"x"))  # dummy
arr = np.frombuffer(mm, dtype=np.uint8)
src0 = arr.ctypes.data
priv = np.fromfile(f, dtype=np.uint8)            # a private copy (malloc: huge pages likely)
out = np.zeros(W + 4096, dtype=np.uint8); out[:] = 1
call(priv.ctypes.data, W, out)                   # warm: slots, pinned staging
for name, src, populate in (("private array", priv.ctypes.data, False), ("mapped file, untouched", src0, False), ("private array again", priv.ctypes.data, False)):
    ts = [call(src + k * W, W, out) for k in range(N // W)]
    print("%-28s -> used buffer : %s ms per 256 MiB call  (%.1f GB/s)" % (name, " ".join("%.1f" % (t * 1e3) for t in ts), W * len(ts) / sum(ts) / 1e9), flush=True)
# mapped file with the window populated first (MADV_POPULATE_READ = 22), as the command line's reader does
mm2 = mmap.mmap(fd, N, flags=mmap.MAP_PRIVATE, prot=mmap.PROT_READ)
arr2 = np.frombuffer(mm2, dtype=np.uint8)
ts, tp = [], []
for k in range(N // W):
    t0 = time.perf_counter()
    r = libc.madvise(ctypes.c_void_p(arr2.ctypes.data + k * W), ctypes.c_size_t(W), 22)
    tp.append(time.perf_counter() - t0)
    ts.append(call(arr2.ctypes.data + k * W, W, out))
print("%-28s -> used buffer : %s ms  (%.1f GB/s; populate %s ms, rc %d)" % ("mapped file, populated", " ".join("%.1f" % (t * 1e3) for t in ts), W * len(ts) / sum(ts) / 1e9,
                                                                     " ".join("%.1f" % (t * 1e3) for t in tp), r), flush=True)
ts = []
for k in range(N // W):
    fresh = np.empty(W + 4096 + k * 8192, dtype=np.uint8)
    ts.append(call(priv.ctypes.data + k * W, W, fresh))
print("%-28s -> fresh buffer: %s ms  (%.1f GB/s)" % ("private array", " ".join("%.1f" % (t * 1e3) for t in ts), W * len(ts) / sum(ts) / 1e9), flush=True)
def call_multi(src_addr, n, dst):
    t0 = time.perf_counter()
    rc = L.trre_scan_host_multi(p._h, ctypes.c_char_p(src_addr), n, dst.ctypes.data_as(ctypes.c_char_p), dst.size, ctypes.byref(m), 0)
    assert rc == 0, rc
    return time.perf_counter() - t0
ts = [call_multi(priv.ctypes.data + k * W, W, out) for k in range(N // W)]
print("%-28s -> used buffer : %s ms  (%.1f GB/s)" % ("trre_scan_host_multi, private", " ".join("%.1f" % (t * 1e3) for t in ts), W * len(ts) / sum(ts) / 1e9), flush=True)
t = call(priv.ctypes.data, N, np.zeros(N + 4096, dtype=np.uint8))
print("one call of %.1f GiB: %.1f ms (%.1f GB/s)" % (N / 2**30, t * 1e3, N / t / 1e9))
os.unlink(f)
