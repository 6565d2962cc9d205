#!/usr/bin/env python3
"""bench.py — scan-mode throughput of the MI355X transducer scan engine.

One "step" = one pass of the hot path (the scan of a whole '\\n'-delimited
buffer, i.e. the scan branch of the reference's main(): trre_dft.c:1272-1286)
over one batch of synthetic input that is already resident in HBM.

Workload (BASELINE.json configs[1], the one the metric is quoted on):
  '[a:A-z:Z]' uppercase DFT scan over 1 GiB of synthetic ASCII lines per GPU.
N GPUs = N line shards of the same size (weak scaling, no data-path collective).

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # oracle bindings: cpu_baseline leg only

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def synth_lines(n, seed, device):
    """printable ASCII 0x20-0x7E, ~80 % letters, '\\n'-terminated lines of 32..160 bytes,
    last byte '\\n', no NUL."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    data = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 28
    for lo in range(0, n, step):
        k = min(step, n - lo)
        kind = torch.randint(0, 100, (k,), dtype=torch.uint8, device=device, generator=g)
        lower = torch.randint(97, 123, (k,), dtype=torch.uint8, device=device, generator=g)
        other = torch.randint(0x20, 0x7f, (k,), dtype=torch.uint8, device=device, generator=g)
        part = torch.where(kind < 70, lower, torch.where(kind < 80, lower - 32, other))
        data[lo:lo + k] = part
        del kind, lower, other, part
    lens = torch.randint(33, 162, (n // 64 + 2,), device=device, generator=g)   # line length + newline
    ends = torch.cumsum(lens, 0) - 1
    data[ends[ends < n]] = 10
    data[n - 1] = 10
    return data


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same
    command (profiles/*_bench_pmc_{FETCH,WRITE}_SIZE.txt, newest round).  FETCH_SIZE / WRITE_SIZE
    are in KB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read
    (MI355X_MICROARCH.md, HBM section), so it is doubled."""
    import glob
    import re
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_pmc_%s.txt" % c)))
        if not files:
            return None
        text = open(files[-1]).read()
        if kernel_name not in text:
            return None
        m = re.search(r"%s\s+([0-9.]+)" % c, text)
        if not m:
            return None
        vals[c] = float(m.group(1))
    return int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)


def cpu_baseline(pattern, engine, sample, cores_mt):
    """Time the reference's CPU path on the host cores over a bounded sample.
    Prefers the compiled reference binary (kind "reference"); falls back to the
    oracle's C restatement (kind "port")."""
    from oracle_lib import Oracle, REF_DIR, ref_available, scan_mt
    res = {}
    nbytes = len(sample)
    if ref_available():
        binary = os.path.join(REF_DIR, "trre" if engine == "nft" else "trre_dft")
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
        with tempfile.NamedTemporaryFile(dir=tmpdir) as tf:
            tf.write(sample)
            tf.flush()
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as devnull:
                subprocess.run([binary, pattern, tf.name], stdout=devnull, check=True)
            dt = time.perf_counter() - t0
        res = {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference"}
    else:
        o = Oracle(pattern, engine)
        t0 = time.perf_counter()
        o.scan(sample)
        dt = time.perf_counter() - t0
        res = {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port"}
    res["sample"] = "%d MiB of the same synthetic workload, file -> /dev/null, %.1f s" % (nbytes >> 20, dt)
    t0 = time.perf_counter()
    scan_mt(pattern, engine, cores_mt, sample)
    dt = time.perf_counter() - t0
    res["port_all_cores"] = {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": cores_mt, "kind": "port",
                             "sample": "same sample, line-sharded threads, %.1f s" % dt}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=1 << 30, help="input bytes per GPU")
    ap.add_argument("--pattern", default="[a:A-z:Z]")
    ap.add_argument("--engine", default="dft", choices=["dft", "nft"])
    ap.add_argument("--kernel", default="auto", choices=["auto", "bytemap", "tile_lp", "tile_gen", "stream_lp", "stream_gen"])
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    import torch
    import trre_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scan has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    n = args.bytes
    fam = {"auto": 0, "bytemap": 1, "tile_lp": 2, "tile_gen": 3, "stream_lp": 4, "stream_gen": 5}[args.kernel]
    prog = trre_amd.Program(args.pattern, args.engine)
    prog.set_kernel(fam)
    info = prog.info
    inp = synth_lines(n, 0x7472726531 + rank, dev)          # this rank's line shard
    out = torch.empty(n + 64, dtype=torch.uint8, device=dev)

    def run_steps(p, k):
        for _ in range(k):
            p.enqueue(inp, out)
        return p.finish()

    # warmup + one verified pass: size-independent property for the headline pattern
    m = run_steps(prog, max(args.warmup, 1))
    verified = None
    if args.pattern == "[a:A-z:Z]":
        low = (inp >= 97) & (inp <= 122)
        verified = bool(m == n and torch.equal(out[:n], torch.where(low, inp - 32, inp)))
        del low
        if not verified:
            raise SystemExit("bench: output does not match the uppercase property")

    prog.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    m = run_steps(prog, args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    kernel_ms = prog.last_kernel_ms()                        # HIP events on the launch stream, avg per launch
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = world * n / (elapsed / args.steps) / 1e9          # whole-job input GB/s
    achieved = n / (kernel_ms * 1e-3) / 1e9                   # algorithmic: 1 byte read per input byte
    line = {
        "metric": "input GB/s (scan mode)",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "'%s' %s scan over %.3f GiB synthetic ASCII lines per GPU%s"
                        % (args.pattern, args.engine.upper(), n / 2**30,
                           " (BASELINE.json configs[1])" if (args.pattern, args.engine, n) == ("[a:A-z:Z]", "dft", 1 << 30) else ""),
            "pattern": args.pattern, "engine": args.engine, "bytes_per_gpu": n, "output_bytes_per_gpu": m,
            "kernel": trre_amd.KERNEL_NAMES[info.kernel], "table_rows": info.table_rows,
            "parallelism": "line-sharded x%d, no collective" % world,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": pmc_traffic("k_bytemap") if (info.kernel == 1 and n == 1 << 30) else None,
            "kernel": {"bytemap": "k_bytemap", "tile_lp": "k_scan_lp", "tile_gen": "k_scan_emit", "stream_lp": "k_stream_lpw",
                       "stream_gen": "k_stream_g16<emit> (small tables) / k_stream_direct<emit>"}[trre_amd.KERNEL_NAMES[info.kernel]],
            "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes_per_launch": n,
            "achieved_read_plus_write": round((n + m) / (kernel_ms * 1e-3) / 1e9, 1),
        },
        "verified": verified,
    }

    if rank == 0 and world == 1 and not args.no_extras:
        # the other kernel families on the same input (fewer steps): the general
        # lane-per-line kernels are what non-memoryless patterns run on
        extra = {}
        q = trre_amd.Program(args.pattern, args.engine)
        for f in q.allowed_kernels():
            name = trre_amd.KERNEL_NAMES[f]
            if f == info.kernel:
                continue
            q.set_kernel(f)
            run_steps(q, 1)
            q.set_profiling(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(q, 5)
            dt = (time.perf_counter() - t0) / 5
            extra[name] = {"input_GBps": round(n / dt / 1e9, 1), "launch_batch_ms": round(q.last_kernel_ms(), 4)}
        line["other_kernel_families"] = extra
        # PCIe-inclusive rate through trre_scan_host, timed around the C call (never `value`)
        import ctypes
        import numpy as np
        host = inp.cpu().numpy()
        hout = np.empty(n + 4096, dtype=np.uint8)
        hm = ctypes.c_size_t()
        L = trre_amd.api.lib()
        for _ in range(2):                                   # first call: pinned staging buffers are allocated
            t0 = time.perf_counter()
            rc = L.trre_scan_host(prog._h, host.ctypes.data_as(ctypes.c_char_p), n, hout.ctypes.data_as(ctypes.c_char_p),
                                  hout.size, ctypes.byref(hm), local)
            dt = time.perf_counter() - t0
        line["pcie_inclusive_GBps"] = round(n / dt / 1e9, 2) if rc == 0 else None
        del host, hout

    if rank == 0 and world == 1 and not args.no_cpu:
        cut = min(n, args.cpu_sample_mib << 20)
        sample = inp[:cut].cpu().numpy().tobytes()
        sample = sample[: sample.rfind(b"\n") + 1]
        line["cpu_baseline"] = cpu_baseline(args.pattern, args.engine, sample, os.cpu_count() or 1)
    elif rank == 0:
        line["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
