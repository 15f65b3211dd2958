"""CPU-side checks of the drop-in boundary: the product library loads and exports every symbol that
include/lbmpc.h declares, struct layouts agree, and without a GPU every entry point fails loudly
(LBMPC_ERR_NODEVICE) instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from lbmpc import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lbmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbmpc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _abi.load()
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert lib.lbmpc_version() == 111


def test_struct_layouts_match_c():
    """compile a tiny C program against include/lbmpc.h and compare sizeof/offsetof with the ctypes mirror"""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "lbmpc.h"
int main(void){
 printf("%zu %zu %zu %zu\n", sizeof(lbmpc_sampler), sizeof(lbmpc_params), sizeof(lbmpc_mc_cfg), sizeof(lbmpc_mc_out));
 printf("%zu %zu %zu %zu\n", offsetof(lbmpc_sampler, EV_A), offsetof(lbmpc_params, A), offsetof(lbmpc_mc_cfg, x0), offsetof(lbmpc_mc_cfg, sampler_max_attempts));
 return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    got = [int(x) for x in out]
    want = [C.sizeof(_abi.Sampler), C.sizeof(_abi.Params), C.sizeof(_abi.McCfg), C.sizeof(_abi.McOut),
            _abi.Sampler.EV_A.offset, _abi.Params.A.offset, _abi.McCfg.x0.offset, _abi.McCfg.sampler_max_attempts.offset]
    assert got == want


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _abi.load()
    assert lib.lbmpc_init(0) == _abi.ERR_NODEVICE
    WA = _abi.fcol(np.eye(2))
    s = _abi.fcol(np.zeros((2, 4)))
    m = np.zeros(2)
    rc = lib.lbmpc_halfspace_max(_abi.dptr(WA), 2, 2, _abi.dptr(s), 4, 1, 8, _abi.dptr(m))
    assert rc == _abi.ERR_NODEVICE and b"no CPU path" in lib.lbmpc_last_error()
