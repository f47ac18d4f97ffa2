"""Prints the ctypes mirror of include/mgp.h's structs exactly as cggp/_hip.py declares them (the block between
the GENERATED markers of INTEGRATION.md; tests/test_abi.py checks that the document carries this text, and that
_hip.py itself matches the header as compiled by gcc)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd"))


def _tname(t):
    if hasattr(t, "_length_") and hasattr(t, "_type_") and not isinstance(t, type(ctypes.c_char_p)):
        if getattr(t, "_type_", None) is not None and hasattr(t, "_length_"):
            return f"ctypes.{t._type_.__name__} * {t._length_}"
    if t.__name__.startswith("LP_"):
        return f"ctypes.POINTER({t._type_.__name__})"
    if t.__name__ == "CFunctionType":
        return "FNPTR"
    return f"ctypes.{t.__name__}"


def generate():
    from cggp import _hip
    lines = ["import ctypes", "", f"MGP_VERSION, MGP_MAX_D, MGP_COMM_ID_BYTES = {_hip.MGP_VERSION}, {_hip.MGP_MAX_D}, "
             f"{_hip.MGP_COMM_ID_BYTES}", ""]
    for cls, cname in ((_hip.MgpKernel, "mgp_kernel"), (_hip.MgpOperator, "mgp_operator"),
                       (_hip.MgpPrecond, "mgp_precond"), (_hip.MgpCgStats, "mgp_cg_stats")):
        lines.append(f"class {cls.__name__}(ctypes.Structure):  # {cname}, {ctypes.sizeof(cls)} bytes")
        lines.append("    _fields_ = [")
        for name, typ in cls._fields_:
            tn = _tname(typ)
            if tn == "FNPTR":
                lines.append(f'        ("{name}", ctypes.c_void_p),  # function pointer: a ctypes.CFUNCTYPE object goes here')
            else:
                lines.append(f'        ("{name}", {tn}),')
        lines.append("    ]")
        lines.append("")
    return "\n".join(lines).rstrip() + "\n"


if __name__ == "__main__":
    sys.stdout.write(generate())
