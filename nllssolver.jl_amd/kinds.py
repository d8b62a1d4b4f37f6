"""Registry of variable / residual / robustifier kinds (mirrors include/nlls_amd.h).

The reference lets users write arbitrary Julia residuals differentiated by ForwardDiff
(src/autodiff.jl:81-93); a HIP kernel cannot call those, so the accelerated path is a closed
registry.  Anything outside it is declined (NLLS_ERR_UNSUPPORTED).
"""
# variable kinds: nvars()/update() of src/variable.jl:3-32, src/robustadaptive.jl:3-23
VAR_EUCLIDEAN = 1
VAR_ZERO_TO_INF = 2
VAR_ZERO_TO_ONE = 3
VAR_CONTAMINATED_GAUSSIAN = 4
VAR_POSE_SO3 = 5
VAR_DYNAMIC = 6             # DynamicVector{Float64}: run-time length (src/variable.jl), only under the RES_DYN_* kinds

# residual kinds
RES_BA_AFFINE = 1         # test/optimizeba.jl:4
RES_ROSENBROCK_A = 2      # test/functional.jl:5-16
RES_ROSENBROCK_B = 3      # test/functional.jl:18-25
RES_ROSENBROCK_2D = 4     # examples/rosenbrock.jl:10-20
RES_CURVE_EXP4 = 5        # BASELINE.json config 2
RES_ADAPTIVE_MEAN = 6     # test/adaptivecost.jl:3-13
RES_BA_SO3 = 7            # new
RES_BA_SO3_ADAPTIVE = 8   # new
RES_LINEAR3 = 9           # test/nonsquaredcost.jl:4-14: X w - y
COST_LINEAR3 = 10         # test/nonsquaredcost.jl:28-37: non-squared AbstractCost y'w (value, gradient, Hessian by second-order duals)
RES_DYN_LINEAR = 11       # test/dynamicvars.jl:3-11: X'w - y over one dynamic-size variable; data = (y, X[n])
RES_DYN_NORM = 12         # test/dynamicvars.jl:13-21: w (nres = n); no data
RES_DYN_LINEARSQ = 13     # test/nonsquaredcost.jl:16-26: X*w - y with a square X over a dynamic-size variable; data = (y[n], X[n*n] column-major)
COST_DYN_LINEAR = 14      # test/nonsquaredcost.jl:39-46: non-squared cost y'w over a dynamic-size variable; data = y[n]
RES_SCALE_MIX = 15        # s (w a + (1 - w) b) - y over a standalone ZeroToInfScalar and a standalone ZeroToOneScalar (src/variable.jl:18-32)
DYN_KINDS = (RES_DYN_LINEAR, RES_DYN_NORM, RES_DYN_LINEARSQ, COST_DYN_LINEAR)
ADAPTIVE_KINDS = (RES_ADAPTIVE_MEAN, RES_BA_SO3_ADAPTIVE)      # residual kinds whose slot 0 is the adaptive kernel's variable (src/residual.jl:46-47)

# robust kernels: src/robust.jl:7-77
ROBUST_NONE = 0
ROBUST_HUBER = 1
ROBUST_HUBER2O = 2
ROBUST_GEMAN_MCCLURE = 3
ROBUST_SCALED = 0x10

# (ndeps, nres, ndata, adaptive, ((slot kind, slot dim), ...))
RES_TABLE = {
    RES_BA_AFFINE: (2, 2, 2, False, ((VAR_EUCLIDEAN, 6), (VAR_EUCLIDEAN, 3))),
    RES_ROSENBROCK_A: (1, 1, 1, False, ((VAR_EUCLIDEAN, 1),)),
    RES_ROSENBROCK_B: (2, 1, 1, False, ((VAR_EUCLIDEAN, 1), (VAR_EUCLIDEAN, 1))),
    RES_ROSENBROCK_2D: (1, 2, 2, False, ((VAR_EUCLIDEAN, 2),)),
    RES_CURVE_EXP4: (4, 1, 2, False, ((VAR_EUCLIDEAN, 1),) * 4),
    RES_ADAPTIVE_MEAN: (2, 1, 1, True, ((VAR_CONTAMINATED_GAUSSIAN, 3), (VAR_EUCLIDEAN, 1))),
    RES_BA_SO3: (2, 2, 2, False, ((VAR_POSE_SO3, 6), (VAR_EUCLIDEAN, 3))),
    RES_BA_SO3_ADAPTIVE: (3, 2, 2, True, ((VAR_CONTAMINATED_GAUSSIAN, 3), (VAR_POSE_SO3, 6), (VAR_EUCLIDEAN, 3))),
    RES_LINEAR3: (1, 3, 12, False, ((VAR_EUCLIDEAN, 3),)),
    COST_LINEAR3: (1, 0, 3, False, ((VAR_EUCLIDEAN, 3),)),
    RES_DYN_LINEAR: (1, 1, -1, False, ((VAR_DYNAMIC, 0),)),    # ndata = 1 + n (the variable's run-time length)
    RES_DYN_NORM: (1, -1, 0, False, ((VAR_DYNAMIC, 0),)),      # nres = n
    RES_DYN_LINEARSQ: (1, -1, -2, False, ((VAR_DYNAMIC, 0),)),  # nres = n, ndata = n + n*n
    COST_DYN_LINEAR: (1, 0, -3, False, ((VAR_DYNAMIC, 0),)),   # ndata = n
    RES_SCALE_MIX: (2, 1, 3, False, ((VAR_ZERO_TO_INF, 1), (VAR_ZERO_TO_ONE, 1))),
}


RES_USER0 = 100           # .. 107: residual kinds a USER header adds at build time (include/nlls_amd.h; `make -C nllssolver.jl_amd/csrc user USER_KINDS=...`)


def register_user_kind(kind, ndeps, nres, ndata, slots, adaptive=False):
    """Tell the host mirror about a residual kind of a library built with a user header (ids 100 .. 107): what Res<kind> declares there -- NDEPS, M, NDATA and
    the slots' (variable kind, dimension).  The library itself knows the kind from its build; this table only sizes and checks the host-side arrays."""
    assert 100 <= kind <= 107 and len(slots) == ndeps <= 10
    RES_TABLE[kind] = (int(ndeps), int(nres), int(ndata), bool(adaptive), tuple((int(k), int(d)) for k, d in slots))


def res_ndeps(kind):
    return RES_TABLE[kind][0]


def res_nres(kind):
    return RES_TABLE[kind][1]


def res_ndata(kind):
    return RES_TABLE[kind][2]


def var_storage(kind, dim):
    """Storage length of a variable (may exceed its dof, src/docstrings.jl:11-14)."""
    return {VAR_EUCLIDEAN: dim, VAR_DYNAMIC: dim, VAR_ZERO_TO_INF: 1, VAR_ZERO_TO_ONE: 1,
            VAR_CONTAMINATED_GAUSSIAN: 3, VAR_POSE_SO3: 12}[kind]


def var_dof(kind, dim):
    """nvars(): src/variable.jl:4,9,21,28; src/robustadaptive.jl:21."""
    return {VAR_EUCLIDEAN: dim, VAR_DYNAMIC: dim, VAR_ZERO_TO_INF: 1, VAR_ZERO_TO_ONE: 1,
            VAR_CONTAMINATED_GAUSSIAN: 3, VAR_POSE_SO3: 6}[kind]


class Robustifier:
    """A fixed-parameter robust kernel: (kind, params) as passed in nlls_cost_group."""

    def __init__(self, kind=ROBUST_NONE, params=()):
        self.kind = int(kind)
        self.params = tuple(float(p) for p in params) + (0.0,) * (4 - len(params))

    def key(self):
        return (self.kind, self.params)


def NoRobust():                       # src/robust.jl:7-12
    return Robustifier(ROBUST_NONE)


def HuberKernel(w):                   # src/robust.jl:40-45
    return Robustifier(ROBUST_HUBER, (w,))


def Huber2oKernel(w):                 # src/robust.jl:46
    return Robustifier(ROBUST_HUBER2O, (w,))


def GemanMcclureKernel(w):            # src/robust.jl:63-69
    return Robustifier(ROBUST_GEMAN_MCCLURE, (w,))


def Scaled(inner, height):            # src/robust.jl:22-31
    assert not (inner.kind & ROBUST_SCALED), "nested Scaled is not a registered kernel"
    return Robustifier(inner.kind | ROBUST_SCALED, (inner.params[0], height))
