"""Shared helpers for the tests: build oracle problems from host NLLSProblem objects."""
import numpy as np

from oracle import oracle as O


def oracle_problem(problem):
    """The same packed description that crosses the C ABI, handed to the CPU oracle."""
    op = O.OracleProblem(problem.var_kind, problem.var_dim, problem.groups())
    op.set_variables(problem.variables)
    return op


def blockindices(problem, unfixed=None):
    """linearsystem.jl:93-102: 1-based block number per variable, 0 = fixed."""
    n = problem.nvariables
    unfixed = np.ones(n, bool) if unfixed is None else np.asarray(unfixed, bool)
    bi = np.zeros(n, np.uint64)
    bi[unfixed] = np.arange(1, unfixed.sum() + 1, dtype=np.uint64)
    return bi
