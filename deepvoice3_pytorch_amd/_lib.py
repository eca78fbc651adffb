# coding: utf-8
"""ctypes binding of libdv3hip.so (the C ABI declared in include/dv3hip.h).

The descriptor structs and prototypes are parsed from the header itself, so this mirror can
not drift from the C side; `dv3_sizeof` double-checks every struct at load time.

The product path has NO fallback: if the shared library is missing or was built for another
ABI, importing an op raises (the reference's torch ops are only ever used by `oracle/`).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "dv3hip.h")
# DV3_LIBPATH: developer scripts load the experiment build (csrc: `make EXP=1` -> libdv3hip_exp.so) instead
_LIBPATH = os.environ.get("DV3_LIBPATH") or os.path.join(_HERE, "libdv3hip.so")
if not os.path.isabs(_LIBPATH):
    _LIBPATH = os.path.join(_HERE, _LIBPATH)

_CTYPES = {
    "float": ctypes.c_float,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "uint32_t": ctypes.c_uint32,
    "uint64_t": ctypes.c_uint64,
    "int": ctypes.c_int,
    "char": ctypes.c_char,
    "void": None,
}


def _strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def _field_ctype(base, stars):
    if stars:
        return ctypes.c_void_p
    if base in _CTYPES:
        return _CTYPES[base]
    return ("struct", base)          # a descriptor nested by value: resolved when the classes are made


def parse_header(path=_HEADER):
    """-> (structs: {name: [(field, ctype)]}, funcs: {name: (restype, [argtypes])}, consts)"""
    src = _strip_comments(open(path).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            dm = re.match(r"(?:const\s+)?(\w+)\s*(\**)\s*(.*)$", decl, flags=re.S)
            base, stars0, rest = dm.group(1), dm.group(2), dm.group(3)
            for var in rest.split(","):
                var = var.strip()
                stars = stars0 + "".join(c for c in var if c == "*")
                vname = var.replace("*", "").strip()
                fields.append((vname, _field_ctype(base, stars)))
        structs[name] = fields
    funcs = {}
    nostruct = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    nostruct = re.sub(r"enum\s*\{.*?\}\s*;", " ", nostruct, flags=re.S)
    for m in re.finditer(r"(?:^|\n)\s*(const\s+char\s*\*|int)\s+(dv3_\w+)\s*\(([^)]*)\)\s*;", nostruct):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = re.match(r"(?:const\s+)?(\w+)", a).group(1)
                    argtypes.append(_CTYPES[base])
        funcs[name] = (restype, argtypes)
    consts = {k: int(v) for k, v in re.findall(r"#define\s+(DV3_\w+)\s+\(?(-?\d+)\)?", src)}
    for em in re.finditer(r"enum\s*\{(.*?)\}\s*;", src, flags=re.S):
        nxt = 0
        for item in em.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = [s.strip() for s in item.split("=")]
                nxt = int(v)
            else:
                k = item
            consts[k] = nxt
            nxt += 1
    return structs, funcs, consts


STRUCT_FIELDS, FUNCS, CONSTS = parse_header()


def _make_structs(all_fields):
    out = {}
    for name, fields in all_fields.items():      # header order: a nested struct is declared before its user
        resolved = [(f, out[t[1]] if isinstance(t, tuple) else t) for f, t in fields]
        out[name] = type(name, (ctypes.Structure,), {"_fields_": resolved})
    return out


STRUCTS = _make_structs(STRUCT_FIELDS)
globals().update(CONSTS)


class Dv3LibraryError(RuntimeError):
    pass


_lib = None


def _preload_torch_hip():
    """Make sure the HIP runtime torch uses is the one libdv3hip resolves to (same soname,
    libamdhip64.so.7): torch must be imported first so its bundled runtime is already mapped."""
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so, soname libamdhip64.so.7)
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        try:
            ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly when it is not usable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBPATH):
        raise Dv3LibraryError(
            "libdv3hip.so not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the product path." % _LIBPATH)
    _preload_torch_hip()
    h = ctypes.CDLL(_LIBPATH)
    for name, (restype, argtypes) in FUNCS.items():
        try:
            fn = getattr(h, name)
        except AttributeError:
            raise Dv3LibraryError("libdv3hip.so does not export %s (stale build?)" % name)
        fn.restype = restype
        fn.argtypes = argtypes
    if h.dv3_abi_version() != CONSTS["DV3_ABI_VERSION"]:
        raise Dv3LibraryError("libdv3hip.so ABI %d != header ABI %d: rebuild" %
                              (h.dv3_abi_version(), CONSTS["DV3_ABI_VERSION"]))
    for name, cls in STRUCTS.items():
        n = h.dv3_sizeof(name.encode())
        if n != ctypes.sizeof(cls):
            raise Dv3LibraryError("struct %s: C sizeof %d != ctypes %d" % (name, n, ctypes.sizeof(cls)))
    _lib = h
    # developer knobs (include/dv3hip.h: dv3_debug_set) from the environment, for A/B runs
    for what, var in ((1, "DV3_X3_ABLATE"), (2, "DV3_WGRAD_TILE"), (3, "DV3_X3_PINGPONG")):
        if os.environ.get(var) is not None and os.environ[var] != "":
            h.dv3_debug_set(what, int(os.environ[var]))
    return h


def check(rc, what=""):
    if rc != 0:
        msg = lib().dv3_last_error()
        raise RuntimeError("dv3hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def call(name, *args):
    """Call an int-returning entry point and raise on error."""
    check(getattr(lib(), name)(*args), name)
