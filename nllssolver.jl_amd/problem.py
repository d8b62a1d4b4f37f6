"""Host-side problem container: the NLLSProblem / addvariable! / addcost! API of
src/problem.jl:5-20,90-122, kept as a thin Python mirror (the host language of the reference,
Julia, is not available in this image; the Julia shim that binds the same C ABI is
julia/NLLSsolverAMD.jl).

Indices are 1-BASED, as in the reference: addvariable returns length(problem.variables)
(src/problem.jl:114-122) and cost blocks store 1-based varind (src/residual.jl:4-7); they cross
the C ABI unchanged.  Costs are stored struct-of-arrays per cost type -- the analogue of the
reference's VectorRepo (src/VectorRepo.jl:1-7): one group per (residual kind, robustifier).
"""
import numpy as np

from . import kinds as K


class CostGroup:
    """One VectorRepo entry: all cost blocks of one type (src/VectorRepo.jl:1-7)."""

    def __init__(self, res_kind, robust):
        self.res_kind = int(res_kind)
        self.robust = robust
        self._vi = []      # chunks (n x ndeps) int64, 1-based
        self._da = []      # chunks (n x ndata) f64
        self._cache = None

    def append(self, varind, data):
        self._vi.append(varind)
        self._da.append(data)
        self._cache = None

    def arrays(self):
        if self._cache is None:
            nd, ndata = K.res_ndeps(self.res_kind), K.res_ndata(self.res_kind)
            if ndata < 0:                      # dynamic-size kind: the width is whatever the blocks came with (1 + n)
                ndata = self._da[0].shape[1] if self._da else 0
            vi = np.concatenate(self._vi, axis=0) if self._vi else np.zeros((0, nd), np.int64)
            da = np.concatenate(self._da, axis=0) if self._da else np.zeros((0, ndata), np.float64)
            self._vi, self._da = [vi], [da]
            self._cache = (np.ascontiguousarray(vi, np.int64), np.ascontiguousarray(da, np.float64))
        return self._cache

    def set_arrays(self, varind, data):
        self._vi, self._da, self._cache = [varind], [data], None

    def __len__(self):
        return sum(v.shape[0] for v in self._vi)

    def as_dict(self):
        vi, da = self.arrays()
        return dict(res_kind=self.res_kind, robust_kind=self.robust.kind, robust_params=self.robust.params,
                    varind=vi, data=da)


class NLLSProblem:
    """src/problem.jl:5-20."""

    def __init__(self):
        self._kind = []
        self._dim = []
        self._chunks = []          # storage chunks (1-D f64)
        self._packed = None
        self.costs = {}            # insertion-ordered: key -> CostGroup   (CostStruct = VectorRepo)
        self.varnext = None        # packed trial point (problem.varnext, src/problem.jl:11): filled before a user callback runs, read back after it
        self._gpu = None           # cached device context (see linearsystem.py)

    # ---- variables ---------------------------------------------------------------------------
    def addvariable(self, value, kind=K.VAR_EUCLIDEAN):
        """addvariable!(problem, variable) -> 1-based index   src/problem.jl:114-122"""
        v = np.atleast_1d(np.asarray(value, dtype=np.float64)).ravel()
        dim = v.size if kind in (K.VAR_EUCLIDEAN, K.VAR_DYNAMIC) else K.var_dof(kind, 0)
        assert v.size == K.var_storage(kind, dim), "storage size does not match the variable kind"
        assert K.var_dof(kind, dim) > 0, "Problem with nvars()"
        self._kind.append(kind); self._dim.append(dim); self._chunks.append(v.copy())
        self._invalidate()
        return len(self._kind)

    def addvariables(self, values, kind=K.VAR_EUCLIDEAN):
        """Bulk addvariable!: values is (n x storage); returns the 1-based index of the first."""
        values = np.ascontiguousarray(values, dtype=np.float64)
        n, st = values.shape
        dim = st if kind in (K.VAR_EUCLIDEAN, K.VAR_DYNAMIC) else K.var_dof(kind, 0)
        assert st == K.var_storage(kind, dim)
        first = len(self._kind) + 1
        self._kind += [kind] * n; self._dim += [dim] * n; self._chunks.append(values.ravel().copy())
        self._invalidate()
        return first

    def copy_variables_from(self, other, values=None):
        """The variables of `other` (kinds, sizes, and their values -- or the packed `values`): the start of a sub-problem over the same variables
        (subproblem(problem, ...), src/problem.jl:73-109, keeps the variable vector and selects costs)."""
        self._kind = list(other._kind); self._dim = list(other._dim)
        self._chunks = [np.array(other.variables if values is None else values, dtype=np.float64)]
        self._invalidate()

    def _invalidate(self):
        self._packed = None
        self._gpu = None

    @property
    def nvariables(self):
        return len(self._kind)

    @property
    def var_kind(self):
        return np.asarray(self._kind, dtype=np.int32)

    @property
    def var_dim(self):
        return np.asarray(self._dim, dtype=np.int32)

    @property
    def var_offsets(self):
        st = np.array([K.var_storage(k, d) for k, d in zip(self._kind, self._dim)], dtype=np.int64) \
            if len(set(zip(self._kind, self._dim))) > 8 else None
        if st is None:
            st = np.zeros(len(self._kind), np.int64)
            kk, dd = self.var_kind, self.var_dim
            for (k, d) in set(zip(self._kind, self._dim)):
                st[(kk == k) & (dd == d)] = K.var_storage(k, d)
        off = np.zeros(len(self._kind) + 1, np.int64)
        np.cumsum(st, out=off[1:])
        return off

    @property
    def variables(self):
        """Packed current variable values (problem.variables, src/problem.jl:8)."""
        if self._packed is None:
            self._packed = np.concatenate(self._chunks) if self._chunks else np.zeros(0)
            self._chunks = [self._packed]
        return self._packed

    def variable(self, index):
        """View of variable `index` (1-based) in the packed storage."""
        off = self.var_offsets
        return self.variables[off[index - 1]:off[index]]

    # ---- costs -------------------------------------------------------------------------------
    def addcosts(self, res_kind, varind, data, robust=None):
        """Bulk addcost!: `varind` (n x ndeps, 1-based), `data` (n x ndata).  src/problem.jl:90-107"""
        robust = robust or K.NoRobust()
        nd, ndata = K.res_ndeps(res_kind), K.res_ndata(res_kind)
        varind = np.ascontiguousarray(np.asarray(varind, dtype=np.int64).reshape(-1, nd))
        if ndata < 0:                          # dynamic-size kind (src/autodiff.jl:96-121): 1 + n doubles per block, n = the variable's length
            n_ = int(self.var_dim[varind[0, 0] - 1]); ndata = {-1: 1 + n_, -2: n_ + n_ * n_, -3: n_}[ndata]
        data = np.ascontiguousarray(np.asarray(data, dtype=np.float64).reshape(varind.shape[0], ndata))
        assert varind.shape[0] == data.shape[0]
        assert 0 < nd <= 10, "Problem with ndeps()"            # MAX_ARGS, src/NLLSsolver.jl:28
        if varind.size:
            assert varind.min() >= 1 and varind.max() <= self.nvariables, "Problem with varindices()"
            kk, dd = self.var_kind, self.var_dim
            for s, (sk, sd) in enumerate(K.RES_TABLE[res_kind][4]):      # getvars() type annotations
                vk = kk[varind[:, s] - 1]
                assert np.all(vk == sk), f"slot {s + 1}: variable kind does not match the residual"
                if sk == K.VAR_EUCLIDEAN:
                    assert np.all(dd[varind[:, s] - 1] == sd), f"slot {s + 1}: variable dimension mismatch"
        key = (int(res_kind), robust.key()) if res_kind not in K.DYN_KINDS else (int(res_kind), robust.key(), int(self.var_dim[varind[0, 0] - 1]))   # dynamic: one group per variable length
        if key not in self.costs:
            self.costs[key] = CostGroup(res_kind, robust)
        self.costs[key].append(varind, data)
        self._gpu = None

    def addcost(self, cost):
        """addcost!(problem, cost) for a single residual object (residuals.py)."""
        self.addcosts(cost.res_kind, [cost.varind], [cost.data], cost.robust)

    def ncosts(self):
        """countcosts(costnum, problem.costs)   src/problem.jl:201-207"""
        return sum(len(g) for g in self.costs.values())

    def nresiduals(self):
        """countcosts(resnum, problem.costs)"""
        return sum(len(g) * K.res_nres(g.res_kind) for g in self.costs.values())

    def groups(self):
        return [g.as_dict() for g in self.costs.values()]

    def varcostmap(self):
        """getvarcostmap(problem): (nvar x ncost) boolean CSC as (colptr, rowval), 1-based
        src/problem.jl:124-175."""
        rows, cols = [], []
        c0 = 0
        for g in self.costs.values():
            vi, _ = g.arrays()
            nd = vi.shape[1]
            rows.append(np.sort(vi, axis=1).ravel())
            cols.append(np.repeat(np.arange(c0, c0 + vi.shape[0]), nd))
            c0 += vi.shape[0]
        rowval = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        counts = np.bincount(np.concatenate(cols), minlength=c0) if cols else np.zeros(0, np.int64)
        colptr = np.ones(c0 + 1, np.int64)
        np.cumsum(counts, out=colptr[1:]); colptr[1:] += 1
        return colptr, rowval

    def costlists(self, indices, check=True):
        """For each variable in `indices` (1-based): the cost blocks that depend on it, as the CSR lists
        (cptr, cgroup, cindex, cslot) that nlls_optimize_singles takes -- sparse(getvarcostmap(problem)') restricted to
        the listed variables, src/optimize.jl:62.  check: raise if a block holds two listed variables (the subproblems of one
        nlls_optimize_singles call must be independent; see singles_levels)."""
        indices = np.asarray(indices, dtype=np.int64)
        pos = np.full(self.nvariables + 1, -1, np.int64); pos[indices] = np.arange(indices.size)
        owners, groups, cidx, slots = [], [], [], []
        for gi, g in enumerate(self.costs.values()):
            vi, _ = g.arrays()
            hit = pos[vi]                                       # (n x ndeps): position in `indices` or -1
            assert not check or np.all((hit >= 0).sum(axis=1) <= 1), "a cost block depends on two of the listed variables"
            k, s = np.nonzero(hit >= 0)
            owners.append(hit[k, s]); groups.append(np.full(k.size, gi, np.int32)); cidx.append(k.astype(np.int64)); slots.append(s.astype(np.int32))
        owners = np.concatenate(owners) if owners else np.zeros(0, np.int64)
        order = np.argsort(owners, kind="stable")
        cptr = np.zeros(indices.size + 1, np.int64); np.cumsum(np.bincount(owners, minlength=indices.size), out=cptr[1:])
        cat = lambda xs, dt: (np.concatenate(xs)[order] if xs else np.zeros(0, dt))
        return cptr, cat(groups, np.int32), cat(cidx, np.int64), cat(slots, np.int32)

    def singles_levels(self, indices):
        """optimizesingles! relaxes the listed variables ONE AFTER THE OTHER (src/optimize.jl:183-205), so a variable sees the updated
        values of every listed variable before it.  Variables that share no cost block commute: the sequential result is reproduced by
        launches of independent sets -- level(v) = 1 + the highest level among the EARLIER listed variables v shares a block with.
        Returns the level (0-based) of each entry of `indices` (which must already be in the reference's processing order)."""
        indices = np.asarray(indices, dtype=np.int64)
        cptr, cgroup, cindex, _ = self.costlists(indices, check=False)
        level = np.zeros(indices.size, np.int64)
        if indices.size == 0:
            return level
        # global block number of every (group, index) pair
        gbase = np.concatenate([[0], np.cumsum([len(g) for g in self.costs.values()])])
        blk = gbase[cgroup] + cindex
        if np.unique(blk).size == blk.size:                    # no block holds two listed variables: one level (the common case)
            return level
        next_level = np.zeros(int(gbase[-1]), np.int64)        # per block: 1 + level of the last listed variable that touched it
        for i in range(indices.size):
            b = blk[cptr[i]:cptr[i + 1]]
            lv = int(next_level[b].max()) if b.size else 0
            level[i] = lv
            next_level[b] = lv + 1
        return level

    def reordercostsforschur(self, schurvars):
        """reordercostsforschur!(problem, schurvars): group each cost type's blocks by the Schur
        variable they touch (0 = none).  Returns {key: run indices (1-based)}.
        src/problem.jl:177-199, src/utils.jl:38-52."""
        schurvars = np.asarray(schurvars, dtype=bool)
        runs = {}
        for key, g in self.costs.items():
            vi, da = g.arrays()
            if vi.shape[0] == 0:
                continue
            mask = schurvars[vi - 1]
            assert np.all(mask.sum(axis=1) <= 1), "Each cost block can only depend on one schur variable at most"
            # index of the Schur variable among the Schur variables (costvarmap.rowval), 0 if none
            rank = np.cumsum(schurvars)
            per = np.where(mask.any(axis=1), rank[(vi * mask).max(axis=1) - 1], 0)
            order = np.argsort(per, kind="stable")
            per = per[order]
            runs[key] = runlengthencodesortedints(per)
            g.set_arrays(vi[order], da[order])
        self._gpu = None
        return runs


def runlengthencodesortedints(sortedints):
    """src/utils.jl:38-52 (returns 1-based run start indices, as the reference does):
    out[v+1] = 1-based position of the first element >= v, out[end] = length+1."""
    sortedints = np.asarray(sortedints, dtype=np.int64)
    out = np.empty(sortedints[-1] + 2, np.int64)
    out[0] = 1
    vals = np.arange(0, sortedints[-1] + 1)
    if len(vals) > 1:
        out[1:-1] = np.searchsorted(sortedints, vals, side="left")[1:] + 1
    out[-1] = len(sortedints) + 1
    return out
