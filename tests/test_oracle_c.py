"""The plain-C oracle against the reference-made pack goldens and the numpy oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import c_oracle, gptq_oracle as O


@pytest.mark.parametrize("bits", [3, 4])
def test_c_pack_matches_reference_golden(bits):
    g = load_golden("g4_pack")
    tag = f"b{bits}_"
    iw = O.intweight(torch.from_numpy(g[tag + "W"]), torch.from_numpy(g[tag + "scale"]), torch.from_numpy(g[tag + "zero"]))
    assert np.array_equal(c_oracle.pack(iw, bits), g[tag + "qweight"])


def test_c_pack3_wraparound_matches_numpy():
    rng = np.random.default_rng(1)
    iw = rng.integers(0, 2 ** 32, size=(96, 7), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(c_oracle.pack(iw, 3), O.pack3(iw))
    assert np.array_equal(c_oracle.pack(iw, 4), O.pack4(iw))


def test_c_matvec3_matches_fp64_formula():
    rng = np.random.default_rng(2)
    n_in, n_out = 512, 96
    iw = rng.integers(0, 8, size=(n_in, n_out), dtype=np.uint32)
    qw = O.pack3(iw)
    scales = (rng.random(n_out) * 0.02 + 0.001).astype(np.float32)
    zeros = (rng.integers(0, 8, n_out) * scales).astype(np.float32)
    x = rng.standard_normal(n_in).astype(np.float32)
    bias = rng.standard_normal(n_out).astype(np.float32)
    y = c_oracle.vecquant3matmul(x, qw, bias.copy(), scales, zeros)
    ref = O.dequant_matvec(x, qw, bias, scales, zeros, 3)
    assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()
