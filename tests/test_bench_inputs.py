"""bench.py generates its synthetic inputs with its own vectorised PCG32 (the product and the
bench never import the oracle for inputs); it must agree bit for bit with the oracle's
restatement of legacy/PCG.jl:7-22."""
import numpy as np

import bench
from oracle import oracle as orc


def test_bench_pcg32_equals_oracle_pcg32():
    for seed in (0, 5, 6, 1000, 2**33 + 1):
        assert np.array_equal(bench.pcg32_uniform(70_001, seed), orc.pcg_fill(70_001, seed))


def test_bench_start_point_equals_oracle_definition():
    assert np.array_equal(bench.rosenbrock_chain_x0(4097, 5), orc.rosenbrock_chain_x0(4097))


def test_algorithmic_byte_model():
    n, k = 10_000_000, 20
    two_loop = bench._kernel_bytes("lbfgs_gram_pass", n, k, 8) + bench._kernel_bytes("lbfgs_combine", n, k, 8)
    assert two_loop == (4 * k + 2) * n * 8 == 6_560_000_000
