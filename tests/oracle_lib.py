"""ctypes binding of oracle/libtrre_oracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
this module.  The product package (trre_amd) never imports it.
"""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
LIB_PATH = os.path.join(ORACLE_DIR, "libtrre_oracle.so")

ENGINES = {"nft": 0, "dft": 1}


class OracleError(RuntimeError):
    def __init__(self, code, msg, partial=b""):
        super().__init__("oracle error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg
        self.partial = partial      # what the reference had printed when it failed (NFT engine; exit() flushes stdout)


def build_oracle():
    """(Re)build the C restatement, and the reference binaries when
    /root/reference is present.  Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "port", "ref"], check=True,
                   stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_oracle()
        L = ctypes.CDLL(LIB_PATH)
        L.trre_oracle_compile.restype = ctypes.c_int
        L.trre_oracle_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                          ctypes.c_char_p, ctypes.c_size_t]
        L.trre_oracle_scan.restype = ctypes.c_int
        L.trre_oracle_scan.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        L.trre_oracle_match.restype = ctypes.c_int
        L.trre_oracle_match.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        L.trre_oracle_scan_mt.restype = ctypes.c_int
        L.trre_oracle_scan_mt.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                          ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p),
                                          ctypes.POINTER(ctypes.c_size_t)]
        L.trre_oracle_set_all.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.trre_oracle_release.argtypes = [ctypes.c_void_p]
        L.trre_oracle_release.restype = None
        L.trre_oracle_free.argtypes = [ctypes.c_void_p]
        L.trre_oracle_free.restype = None
        L.trre_oracle_nft_states.argtypes = [ctypes.c_void_p]
        L.trre_oracle_dft_states.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def _as_bytes(x):
    return x.encode("latin-1") if isinstance(x, str) else bytes(x)


class Oracle:
    """One compiled pattern.  engine: 'nft' (./trre) or 'dft' (./trre_dft)."""

    def __init__(self, pattern, engine="nft", all_outputs=False):
        """all_outputs: `-a`, generator mode — scan() / match() print the output of every accepting path"""
        self.pattern = _as_bytes(pattern)
        self.engine = ENGINES[engine]
        self._h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(200)
        rc = lib().trre_oracle_compile(self.pattern, self.engine, ctypes.byref(self._h), err, 200)
        if rc:
            self._h = None
            raise OracleError(rc, err.value.decode("latin-1"))
        if all_outputs and lib().trre_oracle_set_all(self._h, 1):
            raise OracleError(-1, "generator mode is an NFT feature")

    def scan(self, data):
        data = _as_bytes(data)
        out = ctypes.c_void_p()
        m = ctypes.c_size_t()
        rc = lib().trre_oracle_scan(self._h, data, len(data), ctypes.byref(out), ctypes.byref(m))
        try:
            if rc:
                raise OracleError(rc, "scan failed", ctypes.string_at(out, m.value) if out else b"")
            return ctypes.string_at(out, m.value)
        finally:
            lib().trre_oracle_release(out)

    def match(self, data):
        """`trre -m`: whole-line matches only (NFT engine)"""
        data = _as_bytes(data)
        out = ctypes.c_void_p()
        m = ctypes.c_size_t()
        rc = lib().trre_oracle_match(self._h, data, len(data), ctypes.byref(out), ctypes.byref(m))
        try:
            if rc:
                raise OracleError(rc, "match failed", ctypes.string_at(out, m.value) if out else b"")
            return ctypes.string_at(out, m.value)
        finally:
            lib().trre_oracle_release(out)

    def scan_buffer(self, addr, n):
        """scan n bytes at raw address addr (e.g. a numpy buffer); returns bytes"""
        out = ctypes.c_void_p()
        m = ctypes.c_size_t()
        rc = lib().trre_oracle_scan(self._h, ctypes.cast(addr, ctypes.c_char_p), n,
                                    ctypes.byref(out), ctypes.byref(m))
        if rc:
            raise OracleError(rc, "scan failed")
        try:
            return ctypes.string_at(out, m.value)
        finally:
            lib().trre_oracle_release(out)

    @property
    def nft_states(self):
        return lib().trre_oracle_nft_states(self._h)

    @property
    def dft_states(self):
        return lib().trre_oracle_dft_states(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().trre_oracle_free(self._h)
            self._h = None


def scan_mt(pattern, engine, threads, data):
    data = _as_bytes(data)
    out = ctypes.c_void_p()
    m = ctypes.c_size_t()
    rc = lib().trre_oracle_scan_mt(_as_bytes(pattern), ENGINES[engine], threads, data, len(data),
                                   ctypes.byref(out), ctypes.byref(m))
    if rc:
        raise OracleError(rc, "scan_mt failed")
    try:
        return ctypes.string_at(out, m.value)
    finally:
        lib().trre_oracle_release(out)


def ref_available():
    return os.access(os.path.join(REF_DIR, "trre"), os.X_OK) and os.access(os.path.join(REF_DIR, "trre_dft"), os.X_OK)


def ref_scan(pattern, engine, data, timeout=60):
    """Run the compiled reference binary in scan mode on `data` (via a temp file)."""
    import tempfile
    binary = os.path.join(REF_DIR, "trre" if engine == "nft" else "trre_dft")
    with tempfile.NamedTemporaryFile() as tf:
        tf.write(_as_bytes(data))
        tf.flush()
        p = subprocess.run([binary, _as_bytes(pattern), tf.name], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError("reference exited %d: %s" % (p.returncode, p.stderr.decode("latin-1")))
    return p.stdout
