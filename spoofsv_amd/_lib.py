"""ctypes binding of libssv_hip.so.

The prototypes are read from ``include/ssv_hip.h`` itself, so the header is the single source of
truth for the C ABI: every declared symbol must exist in the library (checked at load) and gets
exact ``argtypes``/``restype``.  There is NO fallback: if the shared object is missing or a call
fails, a ``RuntimeError`` is raised -- the product path never silently runs on stock torch ops.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "ssv_hip.h")
LIBPATH = os.environ.get("SSV_HIP_LIB") or os.path.join(_PKG, "libssv_hip.so")   # override: tuning builds only


class AdamChunk(ctypes.Structure):
    """Mirror of ``ssv_adam_chunk``."""
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p),
                ("v", ctypes.c_void_p), ("n", ctypes.c_long)]


class PackJob(ctypes.Structure):
    """Mirror of ``ssv_pack_job``."""
    _fields_ = [("w", ctypes.c_void_p), ("planes", ctypes.c_void_p), ("M", ctypes.c_int), ("K", ctypes.c_int),
                ("Kpad", ctypes.c_int), ("KT", ctypes.c_int), ("sm", ctypes.c_long), ("sk", ctypes.c_long),
                ("first_block", ctypes.c_int), ("pad_", ctypes.c_int), ("inv_out", ctypes.c_void_p)]


class WgradJob(ctypes.Structure):
    """Mirror of ``ssv_wgrad_job``."""
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("part", ctypes.c_void_p),
                ("pgrads", ctypes.c_void_p), ("shift", ctypes.c_int * 3), ("pad_", ctypes.c_int),
                ("dy_amax", ctypes.c_void_p), ("x_amax", ctypes.c_void_p), ("dy_namax", ctypes.c_int), ("x_namax", ctypes.c_int)]


def _ctype(decl):
    d = decl.strip()
    if "*" in d:
        return ctypes.c_char_p if d.replace(" ", "") == "constchar*" else ctypes.c_void_p
    base = d.split()[0] if d.split() else d
    if base == "unsigned":
        return ctypes.c_uint
    return {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
            "ssv_stream_t": ctypes.c_void_p, "void": None}[base]


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[\w]+\s*\*?)\s*(ssv_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, argnames = [], []
        if args.strip() not in ("", "void"):
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                argtypes.append(_ctype(mm.group(1)))
                argnames.append(mm.group(2))
        protos[name] = (_ctype(ret), argtypes, argnames)
    return protos


_lib = None
_protos = None


def lib():
    """Load (once) and return the ctypes library with typed prototypes."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise RuntimeError(
            "libssv_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C spoofsv_amd/csrc`).  There is no CPU/torch fallback for the HIP hot path." % LIBPATH)
    # PyTorch ships a HIP runtime of its own (torch/lib/libamdhip64.so); libssv_hip.so names the same SONAME.  Whichever is
    # loaded first serves both -- and a process where the system runtime came first and torch initialised the device second
    # fails its first kernel launch with "no ROCm-capable device".  Load torch's first, always.
    import torch  # noqa: F401
    e = os.environ.get("SSV_PRECISION")
    if e and e not in ("fp32", "0", "bf16x3", "1", "f16x2", "2"):
        # the library aborts on an unknown value (a typo must not run a job in another arithmetic than the one asked for)
        raise RuntimeError("SSV_PRECISION=%r is not one of fp32|0, bf16x3|1, f16x2|2" % e)
    L = ctypes.CDLL(LIBPATH)
    _protos = parse_header()
    missing = [n for n in _protos if not hasattr(L, n)]
    if missing:
        raise RuntimeError("libssv_hip.so lacks symbols declared in include/ssv_hip.h: %s" % ", ".join(missing))
    for name, (ret, argtypes, _) in _protos.items():
        fn = getattr(L, name)
        fn.restype = ret
        fn.argtypes = argtypes
    _lib = L
    return L


def precision():
    """Arithmetic mode of the conv GEMMs: 0 fp32, 1 split-bf16, 2 split-fp16 (include/ssv_hip.h, ssv_set_precision)."""
    return lib().ssv_get_precision()


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib().ssv_last_error().decode()))


def query(name, *args):
    """Call a size_t-returning *_workspace query."""
    return int(getattr(lib(), name)(*args))
