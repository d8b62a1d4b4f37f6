// Stand-alone check + timing of the block cyclic reduction solver (nlls_bcr.hip) against a CPU bordered-band LDL'.
// build: hipcc -O3 -std=c++20 --offload-arch=gfx950 -o tools/bcr/bcr_test tools/bcr/bcr_test.hip ; run on the GPU box.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../nllssolver.jl_amd/csrc/nlls_bcr.hip"

using namespace nlls;

static double cpu_solve(int n, int bw, int nbd, int H, const std::vector<double>& Sb, std::vector<double>& x) {
    const int nbr = nbd + 1;
    std::vector<double> B(Sb.begin(), Sb.begin() + (size_t)n * H);
    std::vector<double> C((size_t)nbr * nbr);
    for (int i = 0; i < nbr * nbr; ++i) C[i] = Sb[(size_t)n * H + i];
    for (int j = 0; j < n; ++j) {
        double* cj = &B[(size_t)j * H]; const double d = cj[0];
        const int em = std::min(bw, n - 1 - j);
        for (int e = 1; e <= em; ++e) {
            const double l = cj[e] / d; double* ce = &B[(size_t)(j + e) * H];
            for (int e2 = e; e2 <= em; ++e2) ce[e2 - e] -= l * cj[e2];
            for (int q = 0; q < nbr; ++q) ce[bw + 1 + q] -= l * cj[bw + 1 + q];
        }
        for (int q = 0; q < nbd; ++q) { const double l = cj[bw + 1 + q] / d; for (int q2 = q; q2 < nbr; ++q2) C[q2 + nbr * q] -= l * cj[bw + 1 + q2]; }
    }
    std::vector<double> xb(nbd + 1, 0.0);
    for (int j = 0; j < nbd; ++j) { const double d = C[j + nbr * j];
        for (int c2 = j + 1; c2 < nbd; ++c2) { const double f = C[c2 + nbr * j] / d; for (int i = c2; i < nbr; ++i) C[i + nbr * c2] -= C[i + nbr * j] * f; }
        for (int i = j + 1; i < nbr; ++i) C[i + nbr * j] /= d; }
    for (int r = nbd - 1; r >= 0; --r) { double v = C[nbd + nbr * r]; for (int r2 = r + 1; r2 < nbd; ++r2) v -= C[r2 + nbr * r] * xb[r2]; xb[r] = v; }
    // note: after the LDL' above row nbd of C holds z = D^-1 L^-1 rhs
    x.assign(n + nbd, 0.0);
    for (int q = 0; q < nbd; ++q) x[n + q] = xb[q];
    for (int j = n - 1; j >= 0; --j) { const double* cj = &B[(size_t)j * H]; double v = cj[bw + 1 + nbd];
        const int em = std::min(bw, n - 1 - j);
        for (int e = 1; e <= em; ++e) v -= cj[e] * x[j + e];
        for (int q = 0; q < nbd; ++q) v -= cj[bw + 1 + q] * xb[q];
        x[j] = v / cj[0]; }
    return 0;
}

static int run_case(int n, int bw, int nbd, int reps, unsigned seed) {
    const int H = bw + 1 + nbd + 1, nbr = nbd + 1;
    std::mt19937_64 rng(seed); std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> Sb((size_t)n * H + (size_t)nbr * nbr, 0.0);
    std::vector<double> rowsum(n + nbd, 0.0);
    for (int j = 0; j < n; ++j) for (int e = 1; e <= bw && j + e < n; ++e) { const double v = U(rng); Sb[(size_t)j * H + e] = v; rowsum[j] += std::fabs(v); rowsum[j + e] += std::fabs(v); }
    for (int j = 0; j < n; ++j) for (int q = 0; q < nbd; ++q) { const double v = 0.3 * U(rng); Sb[(size_t)j * H + bw + 1 + q] = v; rowsum[j] += std::fabs(v); rowsum[n + q] += std::fabs(v); }
    for (int j = 0; j < nbd; ++j) for (int i = j + 1; i < nbd; ++i) { const double v = U(rng); Sb[(size_t)n * H + i + nbr * j] = v; rowsum[n + i] += std::fabs(v); rowsum[n + j] += std::fabs(v); }
    for (int j = 0; j < n; ++j) { Sb[(size_t)j * H] = rowsum[j] * (0.6 + 0.2 * U(rng)) + 0.5; Sb[(size_t)j * H + bw + 1 + nbd] = U(rng); }   // (not diagonally dominant: 0.6 x)
    for (int j = 0; j < nbd; ++j) { Sb[(size_t)n * H + j + nbr * j] = rowsum[n + j] + 1.0; Sb[(size_t)n * H + nbd + nbr * j] = U(rng); }
    // make it safely positive definite: add a multiple of the identity found from Gershgorin
    for (int j = 0; j < n; ++j) Sb[(size_t)j * H] += 0.45 * rowsum[j];
    std::vector<double> xc; cpu_solve(n, bw, nbd, H, Sb, xc);
    if (!BcrSolver::supports(n, bw, nbd)) { printf("case n=%d bw=%d nbd=%d: unsupported\n", n, bw, nbd); return 0; }
    BcrSolver S; std::string err;
    if (S.build(n, bw, nbd, H, &err) != 0) { printf("build failed: %s\n", err.c_str()); return 1; }
    double *dS, *dx; int* dst;
    hipMalloc(&dS, Sb.size() * 8); hipMalloc(&dx, (n + nbd + 16) * 8); hipMalloc(&dst, 512);
    hipMemcpy(dS, Sb.data(), Sb.size() * 8, hipMemcpyHostToDevice); hipMemset(dst, 0, 512); hipMemset(dx, 0, (n + nbd + 16) * 8);
    hipStream_t st; hipStreamCreate(&st);
    S.enqueue(st, dS, dx, dst);
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 1; }
    std::vector<double> xg(n + nbd); int status[80];
    hipMemcpy(xg.data(), dx, (n + nbd) * 8, hipMemcpyDeviceToHost); hipMemcpy(status, dst, 320, hipMemcpyDeviceToHost);
#ifdef BCR_STAMPS
    for (int w = 0; w < 2; ++w) { printf("   stamps wave %d:", w); for (int i = 0; i < 20; ++i) printf(" %d", status[16 + 20 * w + i]); printf("\n"); }
    printf("   arrivals at barrier B, J=0:"); for (int i = 0; i < 8; ++i) printf(" %d", status[56 + i]); printf("   J=1:"); for (int i = 0; i < 8; ++i) printf(" %d", status[64 + i]); printf("\n");
#endif
    double num = 0, den = 0; int worst = -1;
    for (int i = 0; i < n + nbd; ++i) { const double d = std::fabs(xg[i] - xc[i]); if (d > num || d != d) { num = d; worst = i; } den = std::max(den, std::fabs(xc[i])); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) S.enqueue(st, dS, dx, dst);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) S.enqueue(st, dS, dx, dst);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const bool ok = num <= 1e-9 * den && status[0] == 0;
    printf("case n=%5d bw=%3d nbd=%2d  N=%4d NT=%d levels=%2zu launches=%2d  err=%.3e (|x|=%.3e, worst %d) status=%d  %.1f us/solve  %s\n",
           n, bw, nbd, S.N, S.NT, S.levels.size(), S.launches, num, den, worst, status[0], 1e3 * ms / reps, ok ? "OK" : "FAIL");
    hipFree(dS); hipFree(dx); hipFree(dst); hipStreamDestroy(st);
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    int fails = 0;
    const int cases[][3] = {{16, 5, 0}, {80, 65, 0}, {81, 65, 0}, {160, 65, 0}, {200, 17, 3}, {128, 5, 0}, {333, 40, 1}, {600, 65, 0}, {1000, 80, 15},
                            {3000, 65, 1}, {6000, 65, 0}, {6000, 65, 2}, {60000, 65, 0}, {5000, 33, 0}, {777, 1, 0}, {4096, 64, 7}};
    unsigned seed = 1;
    for (auto& c : cases) fails += run_case(c[0], c[1], c[2], 20, seed++);
    printf(fails ? "FAILED %d\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
