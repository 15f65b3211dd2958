"""ctypes binding of oracle/liblbmpc_oracle.so (the CPU checker; tests and bench.py's cpu_baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

from lbmpc import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblbmpc_oracle.so")

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        lib = _abi.declare(C.CDLL(ORACLE_SO), "orc_")
        dp, ci, ip, up = _abi.c_double_p, C.c_int, _abi.c_int_p, _abi.c_uint32_p
        lib.orc_homothety_lp_enum.argtypes = [dp, dp, ci, ci, dp, dp, dp, dp]
        lib.orc_homothety_lp_enum.restype = ci
        lib.orc_ctx_export.argtypes = [C.c_void_p, dp, dp, dp, dp]
        lib.orc_ctx_export.restype = ci
        lib.orc_mc_run_omp.argtypes = [C.c_void_p, C.POINTER(_abi.Sampler), C.POINTER(_abi.McCfg), C.POINTER(_abi.McOut), ci]
        lib.orc_mc_run_omp.restype = ci
        lib.orc_set_certificate.argtypes = [C.c_void_p, dp]
        lib.orc_set_certificate.restype = ci
        _lib = lib
    return _lib
