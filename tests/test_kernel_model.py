"""CPU check of the kernel's arithmetic and layout: the host lane simulator
(tests/native/sim_nw.cpp) compiles the same nw_cell.h as the HIP kernel and must reproduce
the oracle's alignment bit for bit (reference behaviour: textSeqCompare.py:53-170)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO
from oracle import nw_oracle
from tools.synth import synth_pair_ids

_SRC = os.path.join(REPO, "tests", "native", "sim_nw.cpp")
_SO = os.path.join(REPO, "tests", "native", "libsim_nw.so")


@pytest.fixture(scope="module")
def sim():
    hdr = os.path.join(REPO, "text_alignment_amd", "csrc", "nw_cell.h")
    if (not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(_SRC), os.path.getmtime(hdr))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", _SO, _SRC])
    lib = ctypes.CDLL(_SO)
    lib.sim_nw.restype = ctypes.c_int
    lib.sim_nw.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                           ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.sim_nw2.restype = ctypes.c_int
    lib.sim_nw2.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                            ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    return lib


def _sim2_ops(lib, t, o, params, R, kcg, gspan):
    t = np.ascontiguousarray(t, dtype=np.int32)
    o = np.ascontiguousarray(o, dtype=np.int32)
    p = np.asarray(params, dtype=np.int32)
    ops = np.zeros(len(t) + len(o) + 1, dtype=np.uint8)
    ln = ctypes.c_int(0)
    rc = lib.sim_nw2(t.ctypes.data, len(t), o.ctypes.data, len(o), p.ctypes.data, R, kcg, gspan,
                     ops.ctypes.data, ctypes.byref(ln))
    assert rc == 0, rc
    return ops[:ln.value]


def _sim_ops(lib, t, o, params, R):
    t = np.ascontiguousarray(t, dtype=np.int32)
    o = np.ascontiguousarray(o, dtype=np.int32)
    p = np.asarray(params, dtype=np.int32)
    ops = np.zeros(len(t) + len(o) + 1, dtype=np.uint8)
    ln = ctypes.c_int(0)
    rc = lib.sim_nw(t.ctypes.data, len(t), o.ctypes.data, len(o), p.ctypes.data, R,
                    ops.ctypes.data, ctypes.byref(ln))
    assert rc == 0, rc
    return ops[:ln.value]


SYSTEMS = [[8, -4, -7, -7, -3, 0], [10, -5, -7, -7, -7, -7], [5, -10, -2, -7, 0, -5],
           [11, -4, -2, -2, 0, 0], [1, -1, -1, -1, -1, -1], [3, -3, 0, 0, 0, 0],
           [2, -1, 1, -3, -1, 1], [0, 0, 0, 0, 0, 0], [4, -6, -9, -1, -2, -4], [7, 7, 3, 2, 1, 1]]


@pytest.mark.parametrize("R", [4, 8, 16])
def test_sim_matches_oracle_random(sim, R):
    rng = np.random.default_rng(100 + R)
    for k in range(120):
        asz = [2, 4, 27][k % 3]
        n = int(rng.integers(0, 90)) if k % 4 else int(rng.integers(200, 700))
        m = int(rng.integers(0, 90)) if k % 5 else int(rng.integers(100, 400))
        t = rng.integers(0, asz, size=n)
        o = rng.integers(0, asz, size=m)
        if k % 2 == 0 and n and m:
            o[:min(n, m)] = np.where(rng.random(min(n, m)) < 0.8, t[:min(n, m)], o[:min(n, m)])
        sc = SYSTEMS[k % len(SYSTEMS)]
        want = nw_oracle.align_ids(t, o, sc)
        got = _sim_ops(sim, t, o, sc, R)
        assert got.tolist() == want.tolist(), (R, n, m, sc)


@pytest.mark.parametrize("R", [4, 8])
def test_sim_matches_oracle_synth(sim, R):
    for n, m, seed in [(500, 500, 1234), (300, 700, 1236), (1030, 515, 7), (257, 255, 1238)]:
        t, o = synth_pair_ids(n, m, seed)
        want = nw_oracle.align_ids(t, o, [8, -4, -7, -7, -3, 0])
        got = _sim_ops(sim, t, o, [8, -4, -7, -7, -3, 0], R)
        assert got.tolist() == want.tolist(), (R, n, m, seed)


@pytest.mark.parametrize("kcg,gspan", [(1, 1), (2, 3), (4, 8), (12, 12), (16, 16), (16, 32), (32, 96),
                                       (1, 1064), (3, 1064), (16, 1064), (1, 2064), (4, 2064), (16, 2064)])
def test_two_phase_sim_matches_oracle(sim, kcg, gspan):
    """Checkpointed score-only fill + chunked tagged re-fill (the two-phase aligner's data flow): tiny
    checkpoint periods force many restarts and halo steps; `gspan` below 64 is the number of lanes of a
    chunk whose pointer bytes are kept (nw_trace2_kernel keeps 32 at a checkpoint period of 16): windows
    of 1 .. 16 lanes make the walk run off the window's top lane all the time, and the chunk is then
    re-filled around the new position.  gspan = 1064: the flow of the several-waves-per-problem kernel
    (nw_trace2w_kernel): whole chunks kept, every chunk after a strip's first entered through its halo in a buffer
    re-filled for exactly that, positions in a chunk's first two steps sent to the chunk before at once.
    gspan = 2064: the half-strip flow of the large-batch kernel (nw_trace2h_kernel): strips restart in halves of 32
    lanes from lane 31's bottom row, pending states at half-strip borders, the start probe at n = 128 h + 1."""
    rng = np.random.default_rng(500 + kcg)
    for k in range(60):
        asz = [2, 4, 27][k % 3]
        n = int(rng.integers(1, 90)) if k % 4 else int(rng.integers(200, 700))
        m = int(rng.integers(1, 90)) if k % 5 else int(rng.integers(100, 500))
        t = rng.integers(0, asz, size=n)
        o = rng.integers(0, asz, size=m)
        if k % 2 == 0:
            kk = min(n, m)
            o[:kk] = np.where(rng.random(kk) < 0.8, t[:kk], o[:kk])
        sc = SYSTEMS[k % len(SYSTEMS)]
        want = nw_oracle.align_ids(t, o, sc)
        got = _sim2_ops(sim, t, o, sc, 4, kcg, gspan)
        assert got.tolist() == want.tolist(), (n, m, sc, kcg, gspan)
    for n, m, seed in [(500, 500, 1234), (300, 700, 1236), (1030, 515, 7)]:
        t, o = synth_pair_ids(n, m, seed)
        want = nw_oracle.align_ids(t, o, SYSTEMS[0])
        assert _sim2_ops(sim, t, o, SYSTEMS[0], 4, kcg, gspan).tolist() == want.tolist()
    # long runs of transcript-side gaps: the walk climbs many lanes inside one chunk
    for runlen, at in [(150, 100), (400, 0), (300, 250)]:
        base = rng.integers(0, 27, size=400)
        junk = 27 + rng.integers(0, 3, size=runlen)
        t = np.concatenate([base[:at], junk, base[at:]])
        for sc in ([8, -4, -7, -7, 0, 0], SYSTEMS[0]):
            want = nw_oracle.align_ids(t, base, sc)
            assert _sim2_ops(sim, t, base, sc, 4, kcg, gspan).tolist() == want.tolist(), (runlen, at, sc, kcg, gspan)
    # walks that START in a strip's first row (n = 256 s + 1: the start state is a tag of the strip
    # above) and that cross strip borders in every state, under every scoring system
    for k, (n, m) in enumerate([(257, 300), (513, 1), (257, 1), (513, 2), (769, 640), (257, 64), (512, 300),
                                (258, 257), (513, 513), (257, 5), (129, 200), (385, 1), (385, 2), (129, 64), (641, 700)]):
        for sc in SYSTEMS[k % 3::3]:
            t = rng.integers(0, 3, size=n)
            o = rng.integers(0, 3, size=m)
            if k % 2:
                kk = min(n, m)
                o[:kk] = np.where(rng.random(kk) < 0.7, t[:kk], o[:kk])
            want = nw_oracle.align_ids(t, o, sc)
            assert _sim2_ops(sim, t, o, sc, 4, kcg, gspan).tolist() == want.tolist(), (n, m, sc, kcg)
