"""ctypes binding of libtrre_mi355x.so (include/trre_mi355x.h).

Host-side mirror of the reference's command-line interface for the scan path:
``Program(pattern, engine)`` plays ``./trre PATTERN`` (engine="nft",
trre_nft.c:713-790) or ``./trre_dft PATTERN`` (engine="dft",
trre_dft.c:1199-1286); ``Program.scan(data)`` is the scan branch of main() over
a whole buffer.  The scan always runs on the GPU through the C ABI — there is no
Python or CPU implementation of it here, and a missing library or device raises.

PyTorch is used only as plumbing (device buffers, streams); it is imported
lazily so that pattern compilation and table inspection work without it.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRRE_LIB_PATH") or os.path.join(_HERE, "lib", "libtrre_mi355x.so")   # override: A/B builds only

ENGINE_NFT, ENGINE_DFT = 0, 1
MODE_SCAN, MODE_MATCH, MODE_SCAN_ALL, MODE_MATCH_ALL = 0, 1, 2, 3
_ENGINES = {"nft": ENGINE_NFT, "dft": ENGINE_DFT, ENGINE_NFT: ENGINE_NFT, ENGINE_DFT: ENGINE_DFT}

KERNEL_AUTO, KERNEL_BYTEMAP, KERNEL_TILE_LP, KERNEL_TILE_GEN, KERNEL_STREAM_LP, KERNEL_STREAM_GEN = 0, 1, 2, 3, 4, 5
KERNEL_GUIDED_LP, KERNEL_GUIDED_GEN = 6, 7
KERNEL_GENERATE = 8
KERNEL_BACKTRACK = 9
KERNEL_DFT_LAZY = 10
KERNEL_NAMES = {0: "auto", 1: "bytemap", 2: "tile_lp", 3: "tile_gen", 4: "stream_lp", 5: "stream_gen", 6: "guided_lp",
                7: "guided_gen", 8: "generate", 9: "backtrack", 10: "dft_lazy"}

FLAG_LENGTH_PRESERVING, FLAG_MEMORYLESS, FLAG_NO_OVERRUN = 1, 2, 4

E_SYNTAX, E_UNDEFINED, E_EPS_CYCLE, E_TOO_BIG, E_UNSUPPORTED = -1, -2, -3, -4, -5
E_DEVICE, E_ARG, E_DIVERGES, E_CAPACITY = -6, -7, -8, -9


class TrreError(RuntimeError):
    """A failed library call; .code is the TRRE_E_* value, .message the
    reference-style 'error: ...' text."""

    def __init__(self, code, message, partial=None):
        super().__init__("%s (code %d)" % (message, code))
        self.code = code
        self.message = message
        self.partial = partial      # E_DIVERGES: what the reference had printed before it failed (bytes / tensor), or None


class Info(ctypes.Structure):
    _fields_ = [("engine", ctypes.c_int32), ("kernel", ctypes.c_int32), ("nft_states", ctypes.c_uint32),
                ("nft_cons_states", ctypes.c_uint32), ("dft_states", ctypes.c_uint32),
                ("table_rows", ctypes.c_uint32), ("table_classes", ctypes.c_uint32),
                ("table_bytes", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("chunk_bytes", ctypes.c_uint32),
                ("stream_states", ctypes.c_uint32), ("stream_classes", ctypes.c_uint32),
                ("nft_nodes", ctypes.c_uint32), ("guided_rev_states", ctypes.c_uint32),
                ("guided_fwd_states", ctypes.c_uint32)]


def build_library(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "all"], check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TrreError(E_DEVICE, "error: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                                      "(the scan has no CPU fallback)" % LIB_PATH)
        try:
            import torch  # noqa: F401  (load torch's HIP runtime first so both share one libamdhip64)
        except Exception:  # pragma: no cover - torch is optional for host-only use
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.trre_compile_bytes.argtypes = [ctypes.c_char_p, sz, ctypes.c_int, ctypes.POINTER(vp)]
        L.trre_compile_mode.argtypes = [ctypes.c_char_p, sz, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
        L.trre_free.argtypes = [vp]
        L.trre_free.restype = None
        L.trre_last_error.restype = ctypes.c_char_p
        L.trre_get_info.argtypes = [vp, ctypes.POINTER(Info)]
        L.trre_set_kernel.argtypes = [vp, ctypes.c_int]
        L.trre_export_tables.argtypes = [vp, vp, sz]
        L.trre_export_tables.restype = sz
        L.trre_export_stream_tables.argtypes = [vp, vp, sz]
        L.trre_export_stream_tables.restype = sz
        L.trre_export_guided_tables.argtypes = [vp, ctypes.c_int, vp, sz]
        L.trre_export_guided_tables.restype = sz
        L.trre_scan_device.argtypes = [vp, vp, sz, vp, sz, ctypes.POINTER(sz), vp]
        L.trre_scan_enqueue.argtypes = [vp, vp, sz, vp, sz, vp]
        L.trre_scan_finish.argtypes = [vp, ctypes.POINTER(sz)]
        L.trre_scan_host.argtypes = [vp, ctypes.c_char_p, sz, vp, sz, ctypes.POINTER(sz), ctypes.c_int]
        L.trre_scan_host_multi.argtypes = [vp, ctypes.c_char_p, sz, vp, sz, ctypes.POINTER(sz), ctypes.c_uint32]
        L.trre_set_profiling.argtypes = [vp, ctypes.c_int]
        L.trre_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.trre_debug_generate.argtypes = [vp, ctypes.c_char_p, sz, ctypes.c_char_p, vp, sz, ctypes.POINTER(sz)]
        L.trre_debug_lazy_tables.argtypes = [vp, ctypes.c_int, vp, sz]
        L.trre_debug_lazy_tables.restype = sz
        L.trre_debug_lazy_explore.argtypes = [vp, vp, sz, sz]
        L.trre_shard_bounds.argtypes = [ctypes.c_char_p, sz, ctypes.c_int, ctypes.POINTER(sz)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise TrreError(rc, lib().trre_last_error().decode("latin-1"))


def _bytes(x):
    return x.encode("latin-1") if isinstance(x, str) else bytes(x)


class Program:
    """A compiled pattern bound to one engine."""

    def __init__(self, pattern, engine="nft", mode="scan"):
        """mode "scan" (default), "match" (`trre -m`: whole-line matches only, NFT engine), "scan_all" (`trre -a`) or
        "match_all" (`trre -ma`): generator modes, every accepting path prints (NFT engine)"""
        self.pattern = _bytes(pattern)
        self.engine = _ENGINES[engine]
        self.mode = {"scan": MODE_SCAN, "match": MODE_MATCH, "scan_all": MODE_SCAN_ALL, "match_all": MODE_MATCH_ALL}.get(mode, mode)
        self._h = ctypes.c_void_p()
        _check(lib().trre_compile_mode(self.pattern, len(self.pattern), self.engine, self.mode, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().trre_free(self._h)
            self._h = None

    __del__ = close

    @property
    def info(self):
        i = Info()
        _check(lib().trre_get_info(self._h, ctypes.byref(i)))
        return i

    def set_kernel(self, family):
        _check(lib().trre_set_kernel(self._h, family))

    def export_tables(self):
        n = lib().trre_export_tables(self._h, None, 0)
        buf = ctypes.create_string_buffer(n)
        lib().trre_export_tables(self._h, buf, n)
        return buf.raw

    def export_stream_tables(self):
        n = lib().trre_export_stream_tables(self._h, None, 0)
        buf = ctypes.create_string_buffer(max(n, 1))
        lib().trre_export_stream_tables(self._h, buf, n)
        return buf.raw[:n]

    def export_guided_tables(self):
        """(backward DFA blob, forward tables blob) of the guided families, or (b"", b"")"""
        blobs = []
        for which in (0, 1):
            n = lib().trre_export_guided_tables(self._h, which, None, 0)
            buf = ctypes.create_string_buffer(max(n, 1))
            lib().trre_export_guided_tables(self._h, which, buf, n)
            blobs.append(buf.raw[:n])
        return tuple(blobs)

    def export_gen_tables(self):
        """generator modes: the tables of the device enumeration (gen_block.hpp), or b"" """
        n = lib().trre_export_guided_tables(self._h, 2, None, 0)
        buf = ctypes.create_string_buffer(max(n, 1))
        lib().trre_export_guided_tables(self._h, 2, buf, n)
        return buf.raw[:n]

    def export_guard_tables(self):
        """the stack guard's tables (guard_block.hpp), or b"" when the pattern cannot exhaust the reference's stack"""
        n = lib().trre_export_guided_tables(self._h, 3, None, 0)
        buf = ctypes.create_string_buffer(max(n, 1))
        lib().trre_export_guided_tables(self._h, 3, buf, n)
        return buf.raw[:n]

    def lazy_tables(self):
        """DFT engine (CPU test tier): the lazily determinised tables as they stand: (header, entries, pool)"""
        parts = []
        for which in (0, 1, 2):
            n = lib().trre_debug_lazy_tables(self._h, which, None, 0)
            buf = ctypes.create_string_buffer(max(n, 1))
            lib().trre_debug_lazy_tables(self._h, which, buf, n)
            parts.append(buf.raw[:n])
        return tuple(parts)

    def lazy_explore(self, misses, spec_states=0):
        """... and the exploration of a list of misses (numpy uint32, 16 words per record)"""
        import numpy as np
        m = np.ascontiguousarray(misses, dtype=np.uint32)
        _check(lib().trre_debug_lazy_explore(self._h, m.ctypes.data, len(m) // 16, spec_states))

    def allowed_kernels(self):
        ok = []
        for fam in (KERNEL_BYTEMAP, KERNEL_TILE_LP, KERNEL_TILE_GEN, KERNEL_STREAM_LP, KERNEL_STREAM_GEN, KERNEL_GUIDED_LP,
                    KERNEL_GUIDED_GEN, KERNEL_BACKTRACK, KERNEL_DFT_LAZY):
            if lib().trre_set_kernel(self._h, fam) == 0:
                ok.append(fam)
        lib().trre_set_kernel(self._h, KERNEL_AUTO)
        return ok

    # ---- device path ---------------------------------------------------------------
    def scan_tensor(self, inp, out=None, stream=None):
        """inp: 1-D uint8 CUDA tensor.  Returns a uint8 CUDA tensor view of the output."""
        import torch
        assert inp.is_cuda and inp.dtype == torch.uint8 and inp.dim() == 1 and inp.is_contiguous()
        n = inp.numel()
        if out is None:
            out = torch.empty(max(n, 1) + 16, dtype=torch.uint8, device=inp.device)
        s = stream if stream is not None else torch.cuda.current_stream(inp.device).cuda_stream
        m = ctypes.c_size_t()
        with torch.cuda.device(inp.device):
            rc = lib().trre_scan_device(self._h, inp.data_ptr(), n, out.data_ptr(), out.numel(), ctypes.byref(m), s)
            if rc == E_CAPACITY:                       # variable-length output: retry with the size asked for
                out = torch.empty(m.value + 16, dtype=torch.uint8, device=inp.device)
                rc = lib().trre_scan_device(self._h, inp.data_ptr(), n, out.data_ptr(), out.numel(), ctypes.byref(m), s)
        if rc == E_DIVERGES:
            raise TrreError(rc, lib().trre_last_error().decode("latin-1"), out[:m.value])
        _check(rc)
        return out[:m.value]

    def enqueue(self, inp, out, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream(inp.device).cuda_stream
        _check(lib().trre_scan_enqueue(self._h, inp.data_ptr(), inp.numel(), out.data_ptr(), out.numel(), s))

    def finish(self):
        m = ctypes.c_size_t()
        _check(lib().trre_scan_finish(self._h, ctypes.byref(m)))
        return m.value

    def scan(self, data, device=0, device_mask=None):
        """bytes in -> bytes out through trre_scan_host (H2D, GPU scan, D2H); with device_mask (0 = every
        visible GPU) through trre_scan_host_multi, line-sharded over the selected GPUs."""
        import numpy as np
        data = _bytes(data)
        cap = len(data) + 64
        for _ in range(2):
            out = np.empty(cap, dtype=np.uint8)              # (not zero-filled: the library writes what it reports)
            m = ctypes.c_size_t()
            if device_mask is None:
                rc = lib().trre_scan_host(self._h, data, len(data), out.ctypes.data_as(ctypes.c_char_p), cap, ctypes.byref(m), device)
            else:
                rc = lib().trre_scan_host_multi(self._h, data, len(data), out.ctypes.data_as(ctypes.c_char_p), cap, ctypes.byref(m),
                                                device_mask)
            if rc == E_CAPACITY:
                cap = m.value + 64
                continue
            if rc == E_DIVERGES:
                raise TrreError(rc, lib().trre_last_error().decode("latin-1"), out[:m.value].tobytes())
            _check(rc)
            return out[:m.value].tobytes()
        _check(rc)

    def generate_with_symbols(self, data, sym):
        """generator modes, host only (CPU test tier): the enumeration with viability symbols computed elsewhere"""
        import numpy as np
        data, sym = _bytes(data), _bytes(sym)
        cap = 1 << 16
        for _ in range(2):
            out = np.empty(cap, dtype=np.uint8)
            m = ctypes.c_size_t()
            rc = lib().trre_debug_generate(self._h, data, len(data), sym, out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(m))
            if rc == E_CAPACITY:
                cap = m.value + 64
                continue
            if rc == E_DIVERGES:
                raise TrreError(rc, lib().trre_last_error().decode("latin-1"), out[:m.value].tobytes())
            _check(rc)
            return out[:m.value].tobytes()
        _check(rc)

    def set_profiling(self, on=True):
        _check(lib().trre_set_profiling(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        _check(lib().trre_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value


def shard_bounds(data, nshards):
    """Byte offsets that cut `data` into nshards pieces at '\\n' boundaries."""
    data = _bytes(data)
    b = (ctypes.c_size_t * (nshards + 1))()
    _check(lib().trre_shard_bounds(data, len(data), nshards, b))
    return list(b)
