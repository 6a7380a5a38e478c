"""CPU tests of the boundary: libccal_hip.so builds (hipcc cross-compiles gfx950 without a
GPU), loads, exports every symbol include/ccal.h declares, and fails loudly without a device.
No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from cameracalibrations_amd import _ffi as F
from cameracalibrations_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    B.build()
    return F.lib()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ccal.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ccal_[a-z_0-9]+)\s*\(", hdr)) - {"ccal_allreduce_fn"}
    assert declared == set(F.SIGNATURES), declared ^ set(F.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ccal_version() == 105


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.ccal_device_count() == 0
    h = C.c_void_p()
    assert lib.ccal_create(0, C.byref(h)) == F.E_DEVICE
    assert b"no CPU fallback" in lib.ccal_last_error()
    from cameracalibrations_amd.solver import Solver
    with pytest.raises(F.CcalError):
        Solver(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cameracalibrations_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "ccal_oracle" not in text, f


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors of ccal_stats / ccal_trace against the header itself: gcc compiles include/ccal.h and prints
    the size of the two structs and the offset of every field."""
    import shutil
    import subprocess
    assert C.sizeof(F.Trace) == 32
    assert C.sizeof(F.Stats) == 7 * 8 + 8 * 4 + 8
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "ccal.h"', 'int main(void) {',
             'printf("%zu %zu\\n", sizeof(ccal_stats), sizeof(ccal_trace));']
    for struct, cls in (("ccal_stats", F.Stats), ("ccal_trace", F.Trace)):
        for name, _ in cls._fields_:
            lines.append(f'printf("{struct}.{name} %zu\\n", offsetof({struct}, {name}));')
    lines.append("return 0; }")
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert out[0].split() == [str(C.sizeof(F.Stats)), str(C.sizeof(F.Trace))]
    got = dict(ln.split() for ln in out[1:] if ln)
    for struct, cls in (("ccal_stats", F.Stats), ("ccal_trace", F.Trace)):
        for name, _ in cls._fields_:
            assert int(got[f"{struct}.{name}"]) == getattr(cls, name).offset, (struct, name)


def test_plain_c_example_compiles_and_links(tmp_path):
    """include/ccal.h is valid C99 and examples/fit_from_c.c links against the shared library alone."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "cameracalibrations_amd")
    from cameracalibrations_amd import build
    build.build()
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "fit_from_c.c"), "-o", str(tmp_path / "fit_from_c"),
                           "-L" + libdir, "-lccal_hip", "-Wl,-rpath," + libdir, "-lm"])


def test_argument_checks_need_no_device(lib):
    """Bad arguments of the multi-device and exchange entry points are refused before any HIP call; without
    a GPU the multi-device fit fails like the single-device one (no CPU path)."""
    import numpy as np
    from cameracalibrations_amd.solver import calibrate_multi
    obj = np.zeros((54, 3), np.float32)
    img = np.zeros((4, 54, 2), np.float32)
    with pytest.raises(F.CcalError) as e:
        calibrate_multi(obj, img, (1500, 2000), [])                   # no devices
    assert e.value.code == -1
    with pytest.raises(F.CcalError) as e:
        calibrate_multi(obj, img, (1500, 2000), [0] * 5)              # more devices than images
    assert e.value.code == -1
    with pytest.raises(F.CcalError) as e:
        calibrate_multi(obj, img, (1500, 2000), [99])                 # device id out of range
    assert e.value.code == -1
    assert lib.ccal_p2p_export(None, 2, 0, C.create_string_buffer(64)) == -1
    assert lib.ccal_p2p_connect(None, b"\0" * 128, 8) == -1
    assert lib.ccal_p2p_selftest(None, 4) == -1
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(F.CcalError) as e:
            calibrate_multi(obj, img, (1500, 2000), [0, 1])
        assert e.value.code == F.E_DEVICE or e.value.code == -6


def _julia_ccalls(text):
    """(symbol, return type, [argument types]) of every `ccall((:sym, LIB), Ret, (types...), args...)` in a Julia source"""
    out, pos = [], 0
    while True:
        k = text.find("ccall((:", pos)
        if k < 0:
            return out
        depth, j = 0, k + len("ccall")
        while True:                                    # the balanced extent of ccall( ... )
            depth += text[j] == "("
            depth -= text[j] == ")"
            j += 1
            if depth == 0:
                break
        body = text[k + len("ccall("):j - 1]
        sym = body[body.index(":") + 1:body.index(",")].strip()
        rest = body[body.index(")") + 1:].lstrip(", \n")         # after (:sym, LIB)
        ret, rest = rest.split(",", 1)
        rest = rest.lstrip()
        assert rest[0] == "(", (sym, rest[:40])
        depth, e = 0, 0
        while True:
            depth += rest[e] == "("
            depth -= rest[e] == ")"
            e += 1
            if depth == 0:
                break
        types = [t.strip() for t in rest[1:e - 1].replace("\n", " ").split(",") if t.strip()]
        args, d, cur = [], 0, ""
        for ch in rest[e:].lstrip(", \n"):              # top-level commas of the argument list
            d += ch in "([{"
            d -= ch in ")]}"
            if ch == "," and d == 0:
                args.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            args.append(cur.strip())
        out.append((sym, ret.strip(), types, args))
        pos = j


def test_julia_binding_matches_the_c_abi():
    """julia/CameraCalibrationsAMD.jl cannot be run here (no Julia in the image); what can be checked is that every `ccall`
    names a symbol of the C ABI and declares -- and passes -- exactly the arguments cameracalibrations_amd/_ffi.py declares
    for it, with the same C widths (pointer / Int32 / UInt32 / Int64 / Float64)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "julia", "CameraCalibrationsAMD.jl")).read()
    calls = _julia_ccalls(text)
    assert len(calls) >= 10

    def kind_jl(t):
        if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
            return "ptr"
        return {"Int32": "i32", "UInt32": "u32", "Int64": "i64", "Float64": "f64", "Cvoid": "void"}[t]

    def kind_c(t):
        if t is None:
            return "void"
        if t in (C.c_int, C.c_int32):
            return "i32"
        if t is C.c_uint32:
            return "u32"
        if t is C.c_int64:
            return "i64"
        if t is C.c_double:
            return "f64"
        return "ptr"                                   # c_void_p, c_char_p, POINTER(...), callbacks
    seen = set()
    for sym, ret, types, args in calls:
        assert sym in F.SIGNATURES, sym
        res, argtypes = F.SIGNATURES[sym]
        assert kind_jl(ret) == kind_c(res), (sym, ret)
        assert [kind_jl(t) for t in types] == [kind_c(t) for t in argtypes], (sym, types)
        assert len(args) == len(types), (sym, len(args), len(types))
        seen.add(sym)
    assert {"ccal_calibrate", "ccal_calibrate_multi", "ccal_project", "ccal_unproject", "ccal_errors", "ccal_detect_corners",
            "ccal_project_many", "ccal_unproject_many", "ccal_create", "ccal_destroy", "ccal_last_error"} <= seen
