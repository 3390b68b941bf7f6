"""ctypes loader for the CPU parity oracle (oracle/liboracle.so). TEST INFRASTRUCTURE: import only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        import shutil
        if not os.path.exists(path) or (shutil.which("make") and shutil.which("g++") and os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "oracle.cpp"))):
            build()   # never check against a stale build of the checker
        lib = C.CDLL(path)
        lib.oracle_graph_compute.restype = C.c_int
        lib.oracle_graph_compute.argtypes = [C.c_void_p, C.c_int]
        lib.oracle_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        lib.oracle_quantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        lib.oracle_mul_mat_vec.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        lib.oracle_fp16_to_fp32.restype = C.c_float
        lib.oracle_fp16_to_fp32.argtypes = [C.c_uint16]
        lib.oracle_fp32_to_fp16.restype = C.c_uint16
        lib.oracle_fp32_to_fp16.argtypes = [C.c_float]
        lib.oracle_fp32_to_bf16.restype = C.c_uint16
        lib.oracle_fp32_to_bf16.argtypes = [C.c_float]
        lib.oracle_gelu.restype = C.c_float
        lib.oracle_gelu.argtypes = [C.c_float]
        _lib = lib
    return _lib
