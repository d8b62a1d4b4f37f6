// tests/user_kinds/radial_ba.hpp -- a worked example of USER residual kinds (include/nlls_amd.h, NLLS_RES_USER0 .. 7; `make user USER_KINDS=...`).
// What a user of the reference writes as a Julia `computeresidual` (README.md:36-46 of the reference, differentiated by ForwardDiff: src/autodiff.jl:81-93) is here ONE
// templated eval<T>, generic in the scalar type: the library's dual numbers give the Jacobian w.r.t. the tangent of update(), and every kernel of the path (accumulate
// sweep in all its forms, cost sweep, optimizesingles, Schur elimination of the points) is instantiated for the kind like for a built-in one.
#pragma once
namespace nlls {
// USER0: an affine camera with one radial distortion coefficient.  Slots: camera (EuclideanVector{7}: two rows of the affine projection + k1), point (EuclideanVector{3}).
//   (u, v) = (c[0:3] . X, c[3:6] . X),   r = (1 + k1 (u^2 + v^2)) (u, v) - measurement
template <> struct Res<NLLS_RES_USER0> {
    static constexpr int NDEPS = 2, M = 2, NDATA = 2, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, 0, 0};
    static constexpr int SD[4] = {7, 3, 0, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        const T* c = sv[0]; const T* X = sv[1];
        const T u = c[0] * X[0] + c[1] * X[1] + c[2] * X[2], v = c[3] * X[0] + c[4] * X[1] + c[5] * X[2];
        const T s = c[6] * (u * u + v * v) + 1.0;
        r[0] = s * u - data[0]; r[1] = s * v - data[1];
    }
};
// USER1: THREE slots -- a focal length shared by every block (one EuclideanVector{1} variable: a single very long row, like the adaptive kernel's), an affine camera, a point:
//   r = f (c[0:3] . X, c[3:6] . X) - measurement.   Three-slot groups take the folded accumulate sweep: a user kind goes through it like a built-in one.
template <> struct Res<NLLS_RES_USER1> {
    static constexpr int NDEPS = 3, M = 2, NDATA = 2, ADAPT = 0;
    static constexpr int SK[4] = {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, 0};
    static constexpr int SD[4] = {1, 6, 3, 0};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        const T f = sv[0][0]; const T* c = sv[1]; const T* X = sv[2];
        r[0] = f * (c[0] * X[0] + c[1] * X[1] + c[2] * X[2]) - data[0];
        r[1] = f * (c[3] * X[0] + c[4] * X[1] + c[5] * X[2]) - data[1];
    }
};
// USER2: FIVE slots (the built-in kinds have at most four; the reference allows MAX_ARGS = 10 variables per cost, src/NLLSsolver.jl:28): a quartic through five scalar
// coefficients, each a variable of its own:  r = a + b t + c t^2 + d t^3 + e t^4 - y,  data = (t, y).  Five degrees of freedom in all: the dense linear system.
template <> struct Res<NLLS_RES_USER2> {
    static constexpr int NDEPS = 5, M = 1, NDATA = 2, ADAPT = 0;
    static constexpr int SK[5] = {NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN, NLLS_VAR_EUCLIDEAN};
    static constexpr int SD[5] = {1, 1, 1, 1, 1};
    template <class T> static NLLS_DEV void eval(const double* data, const T (*sv)[MAXST], T* r) {
        const double t = data[0];
        r[0] = sv[0][0] + sv[1][0] * t + sv[2][0] * (t * t) + sv[3][0] * (t * t * t) + sv[4][0] * (t * t * t * t) - data[1];
    }
};
}  // namespace nlls
#define NLLS_USER_RES(X) X(NLLS_RES_USER0) X(NLLS_RES_USER1) X(NLLS_RES_USER2)
