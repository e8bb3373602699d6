"""ctypes binding of libpyflyt_amd.so (include/pyflyt_amd.h). There is no fallback: if the HIP
extension is missing or cannot be loaded this module raises."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PF_LIB_PATH: an alternative build of the same ABI, for A/B experiments: scratch/variants2/)
LIB_PATH = os.environ.get("PF_LIB_PATH") or os.path.join(_HERE, "libpyflyt_amd.so")

QUADX, FIXEDWING, ROCKET = 0, 1, 2
TASK_NONE, TASK_HOVER, TASK_WAYPOINTS, TASK_MA_HOVER, TASK_DOGFIGHT = 0, 1, 2, 3, 4
NOISE_OFF, NOISE_INJECT, NOISE_PHILOX = 0, 1, 2
AUTORESET_OFF, AUTORESET_NEXT_STEP, AUTORESET_SAME_STEP = 0, 1, 2
F_TERMINATED, F_TRUNCATED, F_CONTACT, F_INFO_COLLISION, F_INFO_OOB, F_INFO_COMPLETE, F_NONFINITE = 1, 2, 4, 8, 16, 32, 64

# --------------------------------------------------------------------------- structs from the header
# The ctypes mirrors of pf_pid / pf_box / pf_surface / pf_rocket / pf_params / pf_buffers are GENERATED
# from include/pyflyt_amd.h at import time (one source of truth: a reordered or retyped field cannot
# drift between the header and this binding); the sizes are still cross-checked against the compiled
# library in lib().
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "include", "pyflyt_amd.h"))
_SCALARS = {"float": C.c_float, "int32_t": C.c_int32, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64,
            "int64_t": C.c_int64, "uint8_t": C.c_uint8, "int": C.c_int, "double": C.c_double, "size_t": C.c_size_t}


def _parse_header(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    defines = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(PF_[A-Z_0-9]+)\s+(-?\d+)\b", text)}
    enums = {}
    for m in re.finditer(r"enum\s+\w+\s*\{(.*?)\}\s*;", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = (s.strip() for s in item.split("="))
                nxt = int(val, 0)
            else:
                name = item
            enums[name] = nxt
            nxt += 1
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl)
            base, ptr, rest = mm.group(2), mm.group(3), mm.group(4)
            for item in rest.split(","):
                item = item.strip()
                is_ptr = bool(ptr) or item.startswith("*")
                item = item.lstrip("* ")
                nm = re.match(r"^(\w+)((?:\s*\[\s*\w+\s*\])*)$", item)
                if not nm:
                    raise ValueError(f"cannot parse field {item!r} of {m.group(3)} in {path}")
                dims = [defines[d] if d in defines else int(d) for d in re.findall(r"\[\s*(\w+)\s*\]", nm.group(2))]
                if is_ptr:
                    ct = C.c_void_p
                else:
                    ct = _SCALARS.get(base) or structs[base]
                    for d in reversed(dims):
                        ct = ct * d
                fields.append((nm.group(1), ct))
        structs[m.group(3)] = type(m.group(3), (C.Structure,), {"_fields_": fields})
    protos = sorted(set(re.findall(r"\b(pf_[a-z_0-9]+)\s*\(", text)))
    return defines, enums, structs, protos


_DEFINES, _ENUMS, _STRUCTS, _PROTOS = _parse_header(HEADER_PATH)
PF_ABI_VERSION = _DEFINES["PF_ABI_VERSION"]
PfPid, PfBox, PfSurface, PfRocket = _STRUCTS["pf_pid"], _STRUCTS["pf_box"], _STRUCTS["pf_surface"], _STRUCTS["pf_rocket"]
PfParams, PfBuffers = _STRUCTS["pf_params"], _STRUCTS["pf_buffers"]
for _k, _v in _ENUMS.items():  # PF_F_TERMINATED -> F_TERMINATED etc. stay spelled out above; expose the rest as PF_*
    globals().setdefault(_k, _v)


EXPORTS = tuple(_PROTOS)  # every function the header declares
PF_MAX_BOXES, PF_MAX_SURF, PF_MAX_TARGETS = _DEFINES["PF_MAX_BOXES"], _DEFINES["PF_MAX_SURF"], _DEFINES["PF_MAX_TARGETS"]
assert (QUADX, FIXEDWING, ROCKET) == (_ENUMS["PF_QUADX"], _ENUMS["PF_FIXEDWING"], _ENUMS["PF_ROCKET"])
assert (F_CONTACT, F_INFO_COMPLETE) == (_ENUMS["PF_F_CONTACT"], _ENUMS["PF_F_INFO_COMPLETE"])

_lib = None


class PyFlytAmdError(RuntimeError):
    pass


def lib():
    """Load the HIP extension; fail loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PyFlytAmdError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). pyflyt_amd has no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise PyFlytAmdError(f"{LIB_PATH} does not export {name}")
    L.pf_abi_version.restype = C.c_int
    L.pf_last_error.restype = C.c_char_p
    L.pf_last_error.argtypes = [C.c_void_p]
    L.pf_ctx_create.argtypes = [C.POINTER(PfParams), C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    L.pf_ctx_destroy.argtypes = [C.c_void_p]
    L.pf_ctx_destroy.restype = None
    for f in ("pf_state_groups", "pf_obs_dim", "pf_n_lanes", "pf_ctx_is_specialised"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.pf_env_reset.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_void_p, C.c_void_p]
    L.pf_env_step.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_void_p]
    L.pf_aviary_reset.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_void_p]
    L.pf_aviary_set_mode.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_int, C.c_void_p, C.c_void_p]
    L.pf_aviary_step.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_int, C.c_void_p]
    L.pf_sample_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.pf_aviary_tick.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_int, C.c_void_p]
    L.pf_wind_links.argtypes = [C.c_void_p]
    L.pf_rollout.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_int, C.c_uint32, C.c_void_p]
    L.pf_body_tick.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_int, C.c_void_p]
    L.pf_sizeof_params.restype = C.c_size_t
    L.pf_sizeof_buffers.restype = C.c_size_t
    if L.pf_sizeof_params() != C.sizeof(PfParams) or L.pf_sizeof_buffers() != C.sizeof(PfBuffers):
        raise PyFlytAmdError("struct layout mismatch between pyflyt_amd/_lib.py and include/pyflyt_amd.h")
    if L.pf_abi_version() != PF_ABI_VERSION:
        raise PyFlytAmdError("ABI version mismatch between pyflyt_amd/_lib.py and libpyflyt_amd.so")
    _lib = L
    return L


ERR_ARG, ERR_UNSUPPORTED, ERR_NO_DEVICE = _ENUMS["PF_ERR_ARG"], _ENUMS["PF_ERR_UNSUPPORTED"], _ENUMS["PF_ERR_NO_DEVICE"]
PfError = PyFlytAmdError  # (the library's status code rides on the exception: e.code)


def check(rc: int, ctx=None):
    if rc != 0:
        msg = lib().pf_last_error(ctx)
        err = PyFlytAmdError(f"pyflyt_amd call failed (code {rc}): {msg.decode() if msg else '?'}")
        err.code = rc
        raise err
