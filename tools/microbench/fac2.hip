// variants of the 16x16 diagonal-tile LDL' (+ inv(L)) on v_mfma_f64_16x16x4_f64: cycles and accuracy (scratch tool)
// build: hipcc -O3 --offload-arch=gfx950 -o fac2.out fac2.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
DEV double readlane_d(double x, int k) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k)); }
DEV double rcp_nr(double d) { double r = __builtin_amdgcn_rcp(d); r = fma(fma(-d, r, 1.0), r, r); r = fma(fma(-d, r, 1.0), r, r); return r; }
constexpr int P = 17;

template <int VARIANT>
__global__ void k(const double* tile_in, double* Wout, double* Liout, double* dout, long long* cyc) {
    __shared__ double tile[16 * P], Wb[16 * P], Li[16 * P], db[32];
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    for (int i = lane; i < 256; i += 64) tile[(i >> 4) * P + (i & 15)] = tile_in[i];
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 8; ++rep) {
    double4_t A, Bt;
#pragma unroll
    for (int r = 0; r < 4; ++r) { A[r] = tile[(lk + 4 * r) * P + li]; Bt[r] = (lk + 4 * r == li) ? 1.0 : 0.0; }
    if (VARIANT == 0) {                 // baseline
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r], bt = Bt[r];
            const double dk = readlane_d(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            const double bm = rowq ? bt : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            db[k] = dk; db[16 + k] = rdk;
            if (k < 15) {
                A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
                Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
            }
        }
    } else if (VARIANT == 2) {          // Bt's MFMA of pivot k-1 issued behind the readlane of pivot k
        double pam = 0, pbm = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r];
            const double dk = readlane_d(w, 16 * q + k);
            __builtin_amdgcn_sched_barrier(0);
            if (k > 0) Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(pam, pbm, Bt, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            db[k] = dk; db[16 + k] = rdk;
            if (k < 15) {
                A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // row k of Bt for its own update: Bt after pivots < k -- the pending update (pivot k-1) has been issued above
                const double bt = Bt[r]; pam = am; pbm = (rowq ? bt : 0.0) * -rdk;
            }
        }
        Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(pam, pbm, Bt, 0, 0, 0);
    } else if (VARIANT == 3 || VARIANT == 4) {   // pivot look-ahead: 1/d_{k+1} is formed from the tile BEFORE pivot k's update, beside its MFMA
        double dk = readlane_d(A[0], 0), rdk = rcp_nr(dk);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r];
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            db[k] = dk; db[16 + k] = rdk;
            double dn = 1.0, rdn = 1.0;
            if (k < 15) {
                // c = A_k[k][k+1] (row k), e = A_k[k+1][k+1] (row k+1)
                const int q1 = (k + 1) & 3, r1 = (k + 1) >> 2;
                const double c = readlane_d(w, 16 * q + k + 1), e = readlane_d(A[r1], 16 * q1 + k + 1);
                if (VARIANT == 4) { const double bt = Bt[r]; Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, (rowq ? bt : 0.0) * -rdk, Bt, 0, 0, 0); }
                A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
                dn = fma(-c * rdk, c, e); rdn = rcp_nr(dn);
            }
            dk = dn; rdk = rdn;
        }
    } else if (VARIANT == 5) {          // baseline + zero/NaN pivot tracking (what the kernel runs)
        int badk = 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r], bt = Bt[r];
            const double dk = readlane_d(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            const double bm = rowq ? bt : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            if (!(fabs(dk) > 0.0)) badk = badk < k ? badk : k;
            db[k] = dk; db[16 + k] = rdk;
            if (k < 15) {
                A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
                Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
            }
        }
        if (badk < 16 && lane == 0) cyc[1] = badk;
    } else if (VARIANT == 6) {          // compact code: four pivots unrolled, registers rotated after each group
#pragma unroll 1
        for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * g4 + q;
                const double w = A[0], bt = Bt[0];
                const double dk = readlane_d(w, 16 * q + k);
                double rdk = __builtin_amdgcn_rcp(dk);
                const bool rowq = lk == q;
                const double am = (rowq && li > k) ? w : 0.0;
                const double bm = rowq ? bt : 0.0;
                rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
                db[k] = dk; db[16 + k] = rdk;
                A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
                Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
            }
            A = double4_t{A[1], A[2], A[3], A[0]}; Bt = double4_t{Bt[1], Bt[2], Bt[3], Bt[0]};
        }
    } else if (VARIANT == 8 || VARIANT == 9) {          // no per-pivot LDS writes: Delta is the diagonal of the finished tile, 1/Delta recomputed by 16 lanes at the end
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r], bt = Bt[r];
            const double dk = readlane_d(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            const double bm = rowq ? bt : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            if (k < 15) {
                A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
                if (VARIANT == 8) Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
            }
        }
        {
            double dsel = A[0];
#pragma unroll
            for (int r = 1; r < 4; ++r) dsel = (li >> 2) == r ? A[r] : dsel;
            if ((li & 3) == lk) { db[li] = dsel; db[16 + li] = rcp_nr(dsel); }
        }
    } else if (VARIANT == 7) {          // one Newton step
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r], bt = Bt[r];
            const double dk = readlane_d(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            const double bm = rowq ? bt : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            db[k] = dk; db[16 + k] = rdk;
            if (k < 15) {
                A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
                Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
            }
        }
    } else if (VARIANT == 1) {          // A only (no inverse)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r];
            const double dk = readlane_d(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            db[k] = dk; db[16 + k] = rdk;
            if (k < 15) A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { Wb[(lk + 4 * r) * P + li] = A[r]; Li[li * P + (lk + 4 * r)] = Bt[r]; }
    asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[0] = (t1 - t0) / 8;
    __syncthreads();
    for (int i = lane; i < 256; i += 64) { Wout[i] = Wb[(i >> 4) * P + (i & 15)]; Liout[i] = Li[(i >> 4) * P + (i & 15)]; }
    if (lane < 32) dout[lane] = db[lane];
}

__global__ void rcpk(double* o) {
    const int i = threadIdx.x; const double x = 1.0 + i * (1.0 / 1024) + 1e-7 * i;
    double r = __builtin_amdgcn_rcp(x); o[i] = r; r = fma(fma(-x, r, 1.0), r, r); o[1024 + i] = r; r = fma(fma(-x, r, 1.0), r, r); o[2048 + i] = r;
}
static double h[256], dref[16], Lref[16][16], Linv[16][16];
static void reference() {
    double A[16][16]; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i][j] = h[i * 16 + j];
    for (int k = 0; k < 16; ++k) { dref[k] = A[k][k]; for (int i = k + 1; i < 16; ++i) { double l = A[i][k] / dref[k]; Lref[i][k] = l; for (int j = k + 1; j < 16; ++j) A[i][j] -= l * A[k][j]; } }
    for (int i = 0; i < 16; ++i) { Lref[i][i] = 1; for (int j = i + 1; j < 16; ++j) Lref[i][j] = 0; }
    for (int c = 0; c < 16; ++c) for (int i = 0; i < 16; ++i) { double v = (i == c); for (int k = 0; k < i; ++k) v -= Lref[i][k] * Linv[k][c]; Linv[i][c] = v; }
}
template <int V> static void run(const char* name, double* tin, double* W, double* Li, double* d, long long* cyc, bool has_inv) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, tin, W, Li, d, cyc);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double hd[32], hW[256], hL[256]; hipMemcpy(hd, d, 256, hipMemcpyDeviceToHost); hipMemcpy(hW, W, 2048, hipMemcpyDeviceToHost); hipMemcpy(hL, Li, 2048, hipMemcpyDeviceToHost);
    double ed = 0, ew = 0, el = 0;
    for (int k2 = 0; k2 < 16; ++k2) ed = fmax(ed, fabs(hd[k2] - dref[k2]) / fabs(dref[k2]));
    for (int i = 0; i < 16; ++i) for (int j = 0; j < i; ++j) ew = fmax(ew, fabs(hW[i * 16 + j] - Lref[i][j] * dref[j]));
    if (has_inv) for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) el = fmax(el, fabs(hL[j * 16 + i] - Linv[i][j]));   // Li[j][i] = inv(L)[i][j]
    printf("%-44s %6lld cycles per tile   err D %.1e  W %.1e  inv(L) %.1e\n", name, c, ed, ew, el);
}
int main() {
    srand(1);
    double M[16][16]; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) M[i][j] = (rand() % 1000) / 1000.0 - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k2 = 0; k2 < 16; ++k2) s += M[i][k2] * M[j][k2]; h[i * 16 + j] = s + (i == j ? 1.0 : 0.0); }
    reference();
    double *tin, *W, *Li, *d; long long* cyc; hipMalloc(&tin, 2048); hipMalloc(&W, 2048); hipMalloc(&Li, 2048); hipMalloc(&d, 256); hipMalloc(&cyc, 64);
    hipMemcpy(tin, h, 2048, hipMemcpyHostToDevice);
    run<0>("0 baseline (A + Bt mfma)", tin, W, Li, d, cyc, true);
    run<1>("1 A only", tin, W, Li, d, cyc, false);
    run<2>("2 Bt deferred behind the next readlane", tin, W, Li, d, cyc, true);
    run<3>("3 pivot look-ahead, A only", tin, W, Li, d, cyc, false);
    run<4>("4 pivot look-ahead, A + Bt", tin, W, Li, d, cyc, true);
    run<5>("5 baseline + zero-pivot tracking", tin, W, Li, d, cyc, true);
    run<6>("6 compact (4 unrolled, rotating registers)", tin, W, Li, d, cyc, true);
    run<7>("7 one Newton step", tin, W, Li, d, cyc, true);
    run<8>("8 no LDS writes in the pivot loop (A + Bt)", tin, W, Li, d, cyc, true);
    run<9>("9 no LDS writes in the pivot loop (A only)", tin, W, Li, d, cyc, false);
    {   // accuracy of v_rcp_f64 alone and with one / two Newton steps
        double* o; hipMalloc(&o, 3 * 1024 * 8);
        hipLaunchKernelGGL(rcpk, dim3(1), dim3(1024), 0, 0, o);
        static double ho[3 * 1024]; hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        double e0 = 0, e1 = 0, e2 = 0;
        for (int i = 0; i < 1024; ++i) { const double x = 1.0 + i * (1.0 / 1024) + 1e-7 * i, ex = 1.0 / x; e0 = fmax(e0, fabs(ho[i] - ex) / ex); e1 = fmax(e1, fabs(ho[1024 + i] - ex) / ex); e2 = fmax(e2, fabs(ho[2048 + i] - ex) / ex); }
        printf("v_rcp_f64 max rel err: raw %.2e, one Newton step %.2e, two %.2e\n", e0, e1, e2);
    }
    return 0;
}
