import time, torch, numpy as np, threading
n = 1 << 30
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pin2 = torch.empty(n, dtype=torch.uint8).pin_memory()
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
dev2 = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, f in (("H2D", lambda: dev.copy_(pin, non_blocking=True)), ("D2H", lambda: pin2.copy_(dev2, non_blocking=True))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "%.1f GB/s" % (n / dt / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(s1): dev.copy_(pin, non_blocking=True)
with torch.cuda.stream(s2): pin2.copy_(dev2, non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("H2D+D2H concurrent: %.1f GB/s each" % (n / dt / 1e9))
# host memcpy speeds
a = np.ones(n, dtype=np.uint8); b = np.empty(n, dtype=np.uint8); b[:] = 0
for ways in (1, 4, 8, 16, 32):
    def work(i):
        lo = n * i // ways; hi = n * (i + 1) // ways
        np.copyto(b[lo:hi], a[lo:hi])
    ths = [threading.Thread(target=work, args=(i,)) for i in range(ways)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("memcpy %d threads: %.1f GB/s" % (ways, n / dt / 1e9))
pa = pin.numpy()
for ways in (8, 16):
    def work(i):
        lo = n * i // ways; hi = n * (i + 1) // ways
        np.copyto(pa[lo:hi], a[lo:hi])
    ths = [threading.Thread(target=work, args=(i,)) for i in range(ways)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("memcpy pageable->pinned %d threads: %.1f GB/s" % (ways, n / dt / 1e9))
