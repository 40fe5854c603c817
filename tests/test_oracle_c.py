"""Pins the C restatement (oracle/mulaw_oracle.c) to the reference's golden vectors.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return C.CDLL(os.path.join(ROOT, "oracle", "_build", "libmulaw_oracle.so"))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("tag", ["c1", "c05"])
def test_c_mulaw_matches_reference(lib, tag):
    g = H.golden(f"mulaw_{tag}.npz")
    x = np.ascontiguousarray(g["x"], dtype=np.float32)
    comp = float(g["compression"])
    codes = np.empty(x.shape, dtype=np.int64)
    lib.oracle_mulaw_compress(ptr(x), ptr(codes), C.c_int64(x.size), 256, C.c_float(comp))
    n_near = int(g["n_near"])
    probes = slice(x.size - 4 - n_near, x.size - 4)        # the edge-adjacent inputs of the fixture
    away = np.ones(x.size, dtype=bool)
    away[probes] = False
    away[-4:] = False
    assert np.array_equal(codes[away], g["codes"][away])          # exact away from bin edges
    assert np.abs(codes[probes] - g["codes"][probes]).max() <= 1  # at most the neighbouring code within an ulp of an edge
    assert (codes[probes] != g["codes"][probes]).mean() < 0.05
    assert np.abs(codes[-4:] - g["codes"][-4:]).max() <= 1         # out-of-range inputs: no clamp
    allc = np.ascontiguousarray(g["all_codes"], dtype=np.int64)
    out = np.empty(allc.shape, dtype=np.float32)
    lib.oracle_mulaw_expand(ptr(allc), ptr(out), C.c_int64(allc.size), 256, C.c_float(comp))
    assert np.allclose(out, g["expanded"], rtol=3e-7, atol=1e-9)


def test_c_argmax_matches_reference(lib):
    g = H.golden("sampler.npz")
    logits = np.ascontiguousarray(g["logits"][:, 0], dtype=np.float32)
    out = np.empty(logits.shape[0], dtype=np.int64)
    lib.oracle_argmax_rows(ptr(logits), C.c_int64(logits.shape[0]), C.c_int64(256), C.c_int64(256), ptr(out))
    assert np.array_equal(out, g["argmax"][:, 0])
    ties = np.zeros((2, 256), dtype=np.float32)
    ties[1, [5, 200]] = 1.0
    lib.oracle_argmax_rows(ptr(ties), C.c_int64(2), C.c_int64(256), C.c_int64(256), ptr(out[:2].copy()))
