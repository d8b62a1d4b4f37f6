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


def bsm_to_csr(index, data, ndof):
    """The symmetric H of a block-sparse linear system as scipy CSR, from the BlockSparseMatrix layout (src/BlockSparseMatrix.jl:30-47): index = (colptr, rowval,
    nzval, boffsets) as nlls_get_bsm_index returns them (1-based), data = A.data.  Block rows hold the blocks at or left of the diagonal; diagonal blocks full."""
    import scipy.sparse as sp
    cp, rv, nz, bo = (np.asarray(a, np.int64) for a in index); nb = len(cp) - 1
    bo0 = bo - 1; bs = np.diff(np.r_[bo0, ndof])
    rows = np.repeat(np.arange(nb), np.diff(cp)); cols = rv - 1; offs = nz - 1
    I, J, V = [], [], []
    for (br, bc) in {(int(a), int(b)) for a, b in zip(bs[rows], bs[cols])}:
        m = (bs[rows] == br) & (bs[cols] == bc)
        r0, c0, o0 = bo0[rows[m]], bo0[cols[m]], offs[m]
        ii, jj = np.meshgrid(np.arange(br), np.arange(bc), indexing="ij")          # a block is column-major br x bc
        idx = o0[:, None, None] + ii[None] + br * jj[None]
        I.append((r0[:, None, None] + ii[None]).ravel()); J.append((c0[:, None, None] + jj[None]).ravel()); V.append(data[idx].ravel())
    L = sp.coo_matrix((np.concatenate(V), (np.concatenate(I), np.concatenate(J))), shape=(ndof, ndof)).tocsr()
    strict = sp.tril(L, -1)
    return (strict + strict.T + sp.diags(L.diagonal())).tocsr()


def device_solve_residual(ctx, lam):
    """|| (H + lam I) x + g || / || g ||  of the device's own damped solve -- H, g as its gradient sweep left them, x as nlls_solve returns it: a check of the
    linear solve that needs no second solver (sizes the oracle's factorisation would take minutes for)."""
    H = bsm_to_csr(ctx.bsm_index(), ctx.get_bsm_data(), ctx.info.ndof); g = ctx.get_grad()
    ctx.damp(lam); x = ctx.solve(want_x=True)
    return float(np.linalg.norm(H @ x + lam * x + g) / np.linalg.norm(g))
