"""ctypes front-end of oracle/nw_oracle.c (TEST INFRASTRUCTURE ONLY).

Restates textSeqCompare.perform_alignment (reference textSeqCompare.py:13-177) on the
CPU.  `perform_alignment` here has the reference's signature and return value so the
parity tests can call the oracle and the HIP path the same way.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnw_oracle.so")
_lib = None

DEFAULT_SYS = [8, -4, -7, -7, -3, 0]          # textSeqCompare.py:10


def build(force=False):
    src = os.path.join(_HERE, "nw_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libnw_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.nw_oracle_align.restype = ctypes.c_int
        _lib.nw_oracle_align.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
            ctypes.c_void_p, ctypes.c_void_p]
        _lib.nw_oracle_fill_only.restype = ctypes.c_double
        _lib.nw_oracle_fill_only.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return _lib


def parse_scoring(scoring_system):
    """Scoring-system forms of textSeqCompare.py:24-42 -> (params[6] | None, callable | None)."""
    if scoring_system is None:
        scoring_system = DEFAULT_SYS
    if len(scoring_system) == 5 and callable(scoring_system[0]):
        gox, goy, gex, gey = scoring_system[-4:]
        return [0.0, 0.0, gox, goy, gex, gey], scoring_system[0]
    if len(scoring_system) == 6:
        return [scoring_system[k] for k in range(6)], None
    if len(scoring_system) == 4:
        ma, mi, go, ge = (scoring_system[k] for k in range(4))
        return [ma, mi, go, go, ge, ge], None
    raise ValueError('scoring_system {} invalid'.format(scoring_system))


def encode_tokens(transcript, ocr):
    """Map arbitrary hashable tokens to dense int32 ids preserving equality."""
    ids = {}
    def enc(seq):
        out = np.empty(len(seq), dtype=np.int32)
        for k, tok in enumerate(seq):
            out[k] = ids.setdefault(tok, len(ids))
        return out
    return enc(transcript), enc(ocr), ids


def align_ids(t_ids, o_ids, params, table=None, want_ptr=False):
    """Run the C oracle on id arrays; returns ops (uint8 array) [, ptr matrix, final scores]."""
    t_ids = np.ascontiguousarray(t_ids, dtype=np.int32)
    o_ids = np.ascontiguousarray(o_ids, dtype=np.int32)
    n, m = len(t_ids), len(o_ids)
    p = np.asarray(params, dtype=np.float64)
    ops = np.empty(max(n + m, 1), dtype=np.uint8)
    ln = ctypes.c_int(0)
    ptr = np.zeros((n + 1, m + 1), dtype=np.uint8) if want_ptr else None
    sc = np.zeros(3, dtype=np.float64)
    tb = None
    tn = tm = 0
    if table is not None:
        tb = np.ascontiguousarray(table, dtype=np.float64)
        tn, tm = tb.shape
    rc = lib().nw_oracle_align(
        t_ids.ctypes.data, n, o_ids.ctypes.data, m, p.ctypes.data,
        tb.ctypes.data if tb is not None else None, tn, tm,
        ops.ctypes.data, ctypes.byref(ln),
        ptr.ctypes.data if ptr is not None else None, sc.ctypes.data)
    if rc != 0:
        raise RuntimeError("nw_oracle_align failed: %d" % rc)
    ops = ops[:ln.value].copy()
    if want_ptr:
        return ops, ptr, sc
    return ops


def ops_to_alignment(ops, transcript, ocr):
    """ops (0 pair / 1 transcript-vs-gap / 2 gap-vs-ocr) -> (tra_align, ocr_align) token lists."""
    tra, oc = [], []
    i = j = 0
    for op in ops:
        if op == 0:
            tra.append(transcript[i]); oc.append(ocr[j]); i += 1; j += 1
        elif op == 1:
            tra.append(transcript[i]); oc.append('_'); i += 1
        else:
            tra.append('_'); oc.append(ocr[j]); j += 1
    return tra, oc


def perform_alignment(transcript, ocr, scoring_system=None, verbose=False):
    """Oracle with the reference's call surface (textSeqCompare.py:13, :177)."""
    params, fn = parse_scoring(scoring_system)
    transcript = list(transcript)
    ocr = list(ocr)
    t_ids, o_ids, ids = encode_tokens(transcript, ocr)
    table = None
    if fn is not None:
        toks = [None] * len(ids)
        for tok, k in ids.items():
            toks[k] = tok
        table = np.empty((len(ids), len(ids)), dtype=np.float64)
        used_t = set(int(v) for v in t_ids)
        used_o = set(int(v) for v in o_ids)
        table[:] = 0.0
        for a in used_t:
            for b in used_o:
                table[a, b] = fn(toks[a], toks[b])
    ops = align_ids(t_ids, o_ids, params, table)
    return ops_to_alignment(ops, transcript, ocr)


def fill_only_rate(t_ids, o_ids, params=None):
    """Time the C restatement's fill on one problem; returns cells/s (single thread)."""
    import time
    t_ids = np.ascontiguousarray(t_ids, dtype=np.int32)
    o_ids = np.ascontiguousarray(o_ids, dtype=np.int32)
    n, m = len(t_ids), len(o_ids)
    p = np.asarray(params or DEFAULT_SYS, dtype=np.float64)
    t0 = time.perf_counter()
    lib().nw_oracle_fill_only(t_ids.ctypes.data, n, o_ids.ctypes.data, m, p.ctypes.data)
    dt = time.perf_counter() - t0
    return n * m / dt
