"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/nodal_hip.h declares; the product fails loudly without a device."""
import ctypes
import os
import re

import pytest

from nodal_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nodal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nodal_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_ffi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in _ffi.load().nodal_version()


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import nodal_amd as n
    with pytest.raises(_ffi.NodalHipError, match="no CPU fallback"):
        n.Circuit(n.Netlist.from_rows([["r1", "R", "1", "1", "g"]]))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "nodal_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(base, f)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), f


def test_every_launch_of_the_library_passes_through_the_launch_log():
    """The diagnostics of a host wait that times out name the last kernel enqueued on the stream (csrc/wait.hip):
    the link step wraps hipLaunchKernel / hipExtLaunchKernel, so the library defines the wrappers and still
    imports the runtime's functions (called by the wrappers only)."""
    import subprocess
    out = subprocess.run(["nm", "-D", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = {line.split()[-1].split("@")[0] for line in out.splitlines() if " T " in line}
    undefined = {line.split()[-1].split("@")[0] for line in out.splitlines() if " U " in line}
    assert {"__wrap_hipLaunchKernel", "__wrap_hipExtLaunchKernel"} <= defined
    assert {"hipLaunchKernel", "hipExtLaunchKernel"} <= undefined
