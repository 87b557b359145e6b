"""ctypes binding of libssg_hip.so (the C ABI declared in include/ssg_hip.h).

The prototypes are parsed from the header itself, so the binding cannot drift from the
declared ABI.  There is NO CPU fallback: if the HIP library is missing the import of any
compute entry point raises (the oracle under oracle/ is test infrastructure only)."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "ssg_hip.h")
SO_PATH = os.environ.get("SSG_LIB_PATH") or os.path.join(_HERE, "libssg_hip.so")     # SSG_LIB_PATH: A/B builds of the same ABI (development)

_SCALARS = {
    "int": ctypes.c_int, "double": ctypes.c_double, "uint64_t": ctypes.c_uint64, "int64_t": ctypes.c_int64,
    "size_t": ctypes.c_size_t, "uint16_t": ctypes.c_uint16, "uint32_t": ctypes.c_uint32, "ssg_stream_t": ctypes.c_void_p,
    "float": ctypes.c_float,
}


class SSGError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ssg_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or name == "ssg_stream_t":
            continue

        def ctype(decl, is_ret=False):
            decl = decl.strip()
            if "*" in decl:
                return ctypes.c_char_p if (is_ret and "char" in decl) else ctypes.c_void_p
            base = decl.replace("const", "").split()
            tname = base[0] if is_ret else (base[0] if len(base) == 1 else base[-2] if base[-2] in _SCALARS else base[0])
            return _SCALARS[tname]

        argtypes = [] if args in ("", "void") else [ctype(a) for a in args.split(",")]
        protos[name] = (ctype(ret, True), argtypes)
    return protos


_lib = None


def available():
    return os.path.exists(SO_PATH)


def lib():
    """Load libssg_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise SSGError("libssg_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback for the grouping path" % SO_PATH)
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in parse_header().items():
            fn = getattr(L, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ssg_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError("%s: %s" % (what, msg))
        raise SSGError("%s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
