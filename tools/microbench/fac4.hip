// 16x16 diagonal-tile LDL' (+ inv(L)) with 4x4 PIVOT BLOCKS on v_mfma_f64_16x16x4_f64: cycles and accuracy (scratch tool)
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o fac4.out fac4.hip
//
// The tile sits in the accumulator layout (register r of lane (li, lk) = T[4 r + lk][li], kept symmetric).  Register p over the whole
// wavefront IS the 4 x 16 row panel P of pivot block p, and it is already a valid B operand (lane (j, kk) = P[kk][j]).  What a rank-4 update
// T -= P' D^-1 P needs besides is U = L_pp^-1 P in lane (i, kk) -- a combination over kk, i.e. over the four lanes i, i + 16, i + 32, i + 48:
// gfx950's v_permlane32_swap / v_permlane16_swap gather those four values into every lane (three swaps per 32-bit half).  The 4 x 4 diagonal
// block is broadcast with v_readlane and factored redundantly by every lane (a chain of four reciprocals), then ONE MFMA whose four
// K-slices all carry data applies the four pivots: 3 + 3 MFMAs per tile instead of 15 + 15, one trip through the matrix pipe per four pivots.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
DEV double readlane_d(double x, int k) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k)); }
DEV double rcp_nr(double d) { double r = __builtin_amdgcn_rcp(d); r = fma(fma(-d, r, 1.0), r, r); r = fma(fma(-d, r, 1.0), r, r); return r; }
// one cubic step: r (1 + e + e^2), e = 1 - d r  (error e^3: 4e-8 -> 6e-23), three dependent operations behind v_rcp_f64 instead of four
DEV double rcp3(double d) { const double r = __builtin_amdgcn_rcp(d); const double e = fma(-d, r, 1.0); const double e2 = fma(e, e, e); return fma(r, e2, r); }
constexpr int P = 17;

// the value of x in lanes li, li + 16, li + 32, li + 48 -> c0 .. c3 in every lane
DEV void gather4_u32(unsigned v, unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3) {
    const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);          // s[0] = rows 0 1 0 1, s[1] = rows 2 3 2 3
    const auto a = __builtin_amdgcn_permlane16_swap(s[0], s[0], false, false);    // a[0] = row 0 everywhere, a[1] = row 1
    const auto b = __builtin_amdgcn_permlane16_swap(s[1], s[1], false, false);
    c0 = a[0]; c1 = a[1]; c2 = b[0]; c3 = b[1];
}
DEV void gather4(double x, double& c0, double& c1, double& c2, double& c3) {
    unsigned l0, l1, l2, l3, h0, h1, h2, h3;
    gather4_u32((unsigned)__double2loint(x), l0, l1, l2, l3); gather4_u32((unsigned)__double2hiint(x), h0, h1, h2, h3);
    c0 = __hiloint2double((int)h0, (int)l0); c1 = __hiloint2double((int)h1, (int)l1); c2 = __hiloint2double((int)h2, (int)l2); c3 = __hiloint2double((int)h3, (int)l3);
}
// per-lane select by lk WITHOUT control flow: nested ?: on lk are turned into branches (with the candidates' arithmetic sunk into them) by
// the compiler; bit-selects through opaque masks stay three v_bfi_b32 per 32-bit half
struct LkMask { unsigned m1, m2, m3; };
DEV LkMask lkmask(int lk) { LkMask m{lk == 1 ? ~0u : 0u, lk == 2 ? ~0u : 0u, lk == 3 ? ~0u : 0u}; asm volatile("" : "+v"(m.m1), "+v"(m.m2), "+v"(m.m3)); return m; }
DEV unsigned bfi(unsigned m, unsigned a, unsigned b) { return (a & m) | (b & ~m); }
DEV double sel4(const LkMask& m, double a, double b, double c, double d) {
    const unsigned lo = bfi(m.m3, (unsigned)__double2loint(d), bfi(m.m2, (unsigned)__double2loint(c), bfi(m.m1, (unsigned)__double2loint(b), (unsigned)__double2loint(a))));
    const unsigned hi = bfi(m.m3, (unsigned)__double2hiint(d), bfi(m.m2, (unsigned)__double2hiint(c), bfi(m.m1, (unsigned)__double2hiint(b), (unsigned)__double2hiint(a))));
    return __hiloint2double((int)hi, (int)lo);
}

// 4x4 pivot blocks.  WITH_INV: the inverse accumulator too.
template <bool WITH_INV>
DEV void factor4(double4_t& A, double4_t& Bt, double* db) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    double dv[4], rv[4];
    const LkMask lm = lkmask(lk);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double x = A[p];
        // the diagonal block, uniform: D[i][j] = T[4p + i][4p + j] sits in lane 16 i + 4p + j
        const double D00 = readlane_d(x, 4 * p), D10 = readlane_d(x, 16 + 4 * p), D20 = readlane_d(x, 32 + 4 * p), D30 = readlane_d(x, 48 + 4 * p);
        const double D11 = readlane_d(x, 16 + 4 * p + 1), D21 = readlane_d(x, 32 + 4 * p + 1), D31 = readlane_d(x, 48 + 4 * p + 1);
        const double D22 = readlane_d(x, 32 + 4 * p + 2), D32 = readlane_d(x, 48 + 4 * p + 2), D33 = readlane_d(x, 48 + 4 * p + 3);
        const double r0 = rcp3(D00), l10 = D10 * r0, l20 = D20 * r0, l30 = D30 * r0;
        const double d1 = fma(-l10, D10, D11), e21 = fma(-l20, D10, D21), e31 = fma(-l30, D10, D31), e22 = fma(-l20, D20, D22), e32 = fma(-l30, D20, D32), e33 = fma(-l30, D30, D33);
        const double r1 = rcp3(d1), l21 = e21 * r1, l31 = e31 * r1;
        const double d2 = fma(-l21, e21, e22), f32 = fma(-l31, e21, e32), f33 = fma(-l31, e31, e33);
        const double r2 = rcp3(d2), l32 = f32 * r2;
        const double d3 = fma(-l32, f32, f33), r3 = rcp3(d3);
        double c0, c1, c2, c3; gather4(x, c0, c1, c2, c3);
        const double u0 = c0, u1 = fma(-l10, u0, c1), u2 = fma(-l21, u1, fma(-l20, u0, c2)), u3 = fma(-l32, u2, fma(-l31, u1, fma(-l30, u0, c3)));
        const double usel = sel4(lm, u0, u1, u2, u3), rsel = sel4(lm, r0, r1, r2, r3);
        const double nb = -usel * rsel;
        A[p] = usel;
        double vsel = 0.0;
        if (WITH_INV) {
            double g0, g1, g2, g3; gather4(Bt[p], g0, g1, g2, g3);
            const double v0 = g0, v1 = fma(-l10, v0, g1), v2 = fma(-l21, v1, fma(-l20, v0, g2)), v3 = fma(-l32, v2, fma(-l31, v1, fma(-l30, v0, g3)));
            vsel = sel4(lm, v0, v1, v2, v3); Bt[p] = vsel;
        }
        dv[p] = sel4(lm, D00, d1, d2, d3); rv[p] = rsel;
        if (p < 3) {
            const bool below = li >= 4 * (p + 1);
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(below ? usel : 0.0, nb, A, 0, 0, 0);
            if (WITH_INV) Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(below ? nb : 0.0, vsel, Bt, 0, 0, 0);
        }
    }
    if (li == 0) {          // lane (0, lk) holds Delta and 1 / Delta of pivots lk, 4 + lk, 8 + lk, 12 + lk
#pragma unroll
        for (int p = 0; p < 4; ++p) { db[4 * p + lk] = dv[p]; db[16 + 4 * p + lk] = rv[p]; }
    }
}

// 2x2 pivot blocks: the two rows of a pivot pair sit in lanes that differ by 16 (lk = q, q + 1, q in {0, 2}) -- ONE v_permlane16_swap per
// 32-bit half brings the partner row in; three broadcasts, two dependent reciprocals, one rank-2 MFMA (two of the four K-slices carry data).
DEV void pair_swap(double x, double& e, double& o) {      // e: the even row of the lane's pair, o: the odd one
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    e = __hiloint2double((int)h[0], (int)l[0]); o = __hiloint2double((int)h[1], (int)l[1]);
}
DEV double selm(unsigned m, double a, double b) {      // m ? a : b without control flow
    return __hiloint2double((int)bfi(m, (unsigned)__double2hiint(a), (unsigned)__double2hiint(b)), (int)bfi(m, (unsigned)__double2loint(a), (unsigned)__double2loint(b)));
}
// the same with the two reciprocals of a pair formed SIDE BY SIDE: 1 / d0 and 1 / det (det = D00 D11 - D10^2 = d0 d1), then d1 = det / d0 and
// 1 / d1 = d0 / det are products -- one reciprocal chain per pair on the critical path instead of two
template <bool WITH_INV>
DEV void factor2c(double4_t& A, double4_t& Bt) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const bool odd = lk & 1, hi = lk >> 1;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int r = p >> 1, q = 2 * (p & 1), k = 4 * r + q;
        const double x = A[r];
        const double D00 = readlane_d(x, 16 * q + k), D10 = readlane_d(x, 16 * q + k + 1), D11 = readlane_d(x, 16 * (q + 1) + k + 1);
        const double det = fma(D00, D11, -D10 * D10);
        const double r0 = rcp3(D00), rdet = rcp3(det);
        const double l10 = D10 * r0, r1 = D00 * rdet;
        double e, o; pair_swap(x, e, o);
        const double u1 = fma(-l10, e, o);
        const bool pair = hi == (bool)(p & 1), row1 = pair && odd, below = pair && li > k + 1;
        const double usel = odd ? u1 : e, rsel = odd ? r1 : r0;
        A[r] = row1 ? u1 : x;
        const double nb = -usel * rsel;
        double vsel = 0.0;
        if (WITH_INV) {
            const double y = Bt[r]; double ye, yo; pair_swap(y, ye, yo);
            const double v1 = fma(-l10, ye, yo);
            Bt[r] = row1 ? v1 : y; vsel = odd ? v1 : ye;
        }
        if (p < 7) {
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(below ? usel : 0.0, nb, A, 0, 0, 0);
            if (WITH_INV) Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(below ? nb : 0.0, vsel, Bt, 0, 0, 0);
        }
    }
}
template <bool WITH_INV>
DEV void factor2(double4_t& A, double4_t& Bt) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const bool odd = lk & 1, hi = lk >> 1;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int r = p >> 1, q = 2 * (p & 1), k = 4 * r + q;
        const double x = A[r];
        const double D00 = readlane_d(x, 16 * q + k), D10 = readlane_d(x, 16 * q + k + 1), D11 = readlane_d(x, 16 * (q + 1) + k + 1);
        const double r0 = rcp3(D00), l10 = D10 * r0, d1 = fma(-l10, D10, D11), r1 = rcp3(d1);
        double e, o; pair_swap(x, e, o);
        const double u1 = fma(-l10, e, o);
        const bool pair = hi == (bool)(p & 1), row1 = pair && odd, below = pair && li > k + 1;
        const double usel = odd ? u1 : e, rsel = odd ? r1 : r0;
        A[r] = row1 ? u1 : x;
        const double nb = -usel * rsel;
        double vsel = 0.0;
        if (WITH_INV) {
            const double y = Bt[r]; double ye, yo; pair_swap(y, ye, yo);
            const double v1 = fma(-l10, ye, yo);
            Bt[r] = row1 ? v1 : y; vsel = odd ? v1 : ye;
        }
        if (p < 7) {
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(below ? usel : 0.0, nb, A, 0, 0, 0);
            if (WITH_INV) Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(below ? nb : 0.0, vsel, Bt, 0, 0, 0);
        }
    }
}

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
    if (VARIANT == 0) {                 // the library's kernel at the end of round 2: 15 pivots x (tile + inverse)
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            const int q = k & 3, r = k >> 2;
            const double w = A[r], bt = Bt[r];
            const double dk = readlane_d(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            const double bm = rowq ? bt : 0.0;
            rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk); rdk = fma(fma(-dk, rdk, 1.0), rdk, rdk);
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
            Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
        }
        double dsel = A[0];
#pragma unroll
        for (int r = 1; r < 4; ++r) dsel = (li >> 2) == r ? A[r] : dsel;
        if ((li & 3) == lk) { db[li] = dsel; db[16 + li] = rcp_nr(dsel); }
    } else if (VARIANT == 1) factor4<true>(A, Bt, db);
    else if (VARIANT == 2) factor4<false>(A, Bt, db);
    else if (VARIANT >= 3 && VARIANT <= 6) {
        if (VARIANT == 3) factor2<true>(A, Bt); else if (VARIANT == 4) factor2<false>(A, Bt); else if (VARIANT == 5) factor2c<true>(A, Bt); else factor2c<false>(A, Bt);
        double dsel = A[0];
#pragma unroll
        for (int r = 1; r < 4; ++r) dsel = (li >> 2) == r ? A[r] : dsel;
        if ((li & 3) == lk) { db[li] = dsel; db[16 + li] = rcp_nr(dsel); }
    }
    // lower triangle of W (row j >= column k) from the symmetric upper entries (row k = lk + 4 r, column j = li)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int kk = lk + 4 * r; if (VARIANT == 0 || li >= kk) Wb[li * P + kk] = A[r]; Li[li * P + kk] = Bt[r]; }
    asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[0] = (t1 - t0) / 8;
    __syncthreads();
    for (int i = lane; i < 256; i += 64) { Wout[i] = Wb[(i >> 4) * P + (i & 15)]; Liout[i] = Li[(i >> 4) * P + (i & 15)]; }
    if (lane < 32) dout[lane] = db[lane];
}

// what the two swaps do to a register that holds the lane number (printed, so that the gather above is checked and not assumed)
__global__ void swapk(unsigned* o) {
    const unsigned v = threadIdx.x;
    const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    o[v] = s[0]; o[64 + v] = s[1]; o[128 + v] = a[0]; o[192 + v] = a[1];
    unsigned c0, c1, c2, c3; gather4_u32(v, c0, c1, c2, c3);
    o[256 + v] = c0; o[320 + v] = c1; o[384 + v] = c2; o[448 + v] = c3;
}
static double h[256], dref[16], Lref[16][16], Linv[16][16];
static void reference() {
    double A[16][16]; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i][j] = h[i * 16 + j];
    for (int k = 0; k < 16; ++k) { dref[k] = A[k][k]; for (int i = k + 1; i < 16; ++i) { double l = A[i][k] / dref[k]; Lref[i][k] = l; for (int j = k + 1; j < 16; ++j) A[i][j] -= l * A[k][j]; } }
    for (int i = 0; i < 16; ++i) { Lref[i][i] = 1; for (int j = i + 1; j < 16; ++j) Lref[i][j] = 0; }
    for (int c = 0; c < 16; ++c) for (int i = 0; i < 16; ++i) { double v = (i == c); for (int k = 0; k < i; ++k) v -= Lref[i][k] * Linv[k][c]; Linv[i][c] = v; }
}
template <int V> static void run(const char* name, double* tin, double* W, double* Li, double* d, long long* cyc, bool has_inv) {
    hipMemset(W, 0, 2048); hipMemset(Li, 0, 2048);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, tin, W, Li, d, cyc);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double hd[32], hW[256], hL[256]; hipMemcpy(hd, d, 256, hipMemcpyDeviceToHost); hipMemcpy(hW, W, 2048, hipMemcpyDeviceToHost); hipMemcpy(hL, Li, 2048, hipMemcpyDeviceToHost);
    double ed = 0, er = 0, ew = 0, el = 0;
    for (int k2 = 0; k2 < 16; ++k2) { ed = fmax(ed, fabs(hd[k2] - dref[k2]) / fabs(dref[k2])); er = fmax(er, fabs(hd[16 + k2] * dref[k2] - 1.0)); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < i; ++j) ew = fmax(ew, fabs(hW[i * 16 + j] - Lref[i][j] * dref[j]));
    if (has_inv) for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) el = fmax(el, fabs(hL[j * 16 + i] - Linv[i][j]));   // Li[j][i] = inv(L)[i][j]
    printf("%-44s %6lld cycles per tile   err D %.1e  1/D %.1e  W %.1e  inv(L) %.1e\n", name, c, ed, er, ew, el);
}
int main() {
    {
        unsigned* o; hipMalloc(&o, 512 * 4); hipLaunchKernelGGL(swapk, dim3(1), dim3(64), 0, 0, o);
        unsigned ho[512]; hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        const char* nm[8] = {"permlane32_swap(v,v)[0]", "permlane32_swap(v,v)[1]", "permlane16_swap(v,v)[0]", "permlane16_swap(v,v)[1]", "gather c0", "gather c1", "gather c2", "gather c3"};
        bool ok = true;
        for (int q = 0; q < 8; ++q) { printf("%-24s rows:", nm[q]); for (int row = 0; row < 4; ++row) printf(" %2u..%2u", ho[64 * q + 16 * row], ho[64 * q + 16 * row + 15]); printf("\n"); }
        for (int m = 0; m < 4; ++m) for (int l = 0; l < 64; ++l) ok = ok && ho[256 + 64 * m + l] == (unsigned)(16 * m + (l & 15));
        printf("gather4: %s\n", ok ? "OK" : "WRONG");
    }
    srand(1);
    double M[16][16]; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) M[i][j] = (rand() % 1000) / 1000.0 - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k2 = 0; k2 < 16; ++k2) s += M[i][k2] * M[j][k2]; h[i * 16 + j] = s + (i == j ? 1.0 : 0.0); }
    reference();
    double *tin, *W, *Li, *d; long long* cyc; hipMalloc(&tin, 2048); hipMalloc(&W, 2048); hipMalloc(&Li, 2048); hipMalloc(&d, 256); hipMalloc(&cyc, 64);
    hipMemcpy(tin, h, 2048, hipMemcpyHostToDevice);
    run<0>("0 round-2 kernel: 15 pivots x (A + Bt)", tin, W, Li, d, cyc, true);
    run<1>("1 4x4 pivot blocks, A + Bt", tin, W, Li, d, cyc, true);
    run<2>("2 4x4 pivot blocks, A only", tin, W, Li, d, cyc, false);
    run<3>("3 2x2 pivot blocks, A + Bt", tin, W, Li, d, cyc, true);
    run<4>("4 2x2 pivot blocks, A only", tin, W, Li, d, cyc, false);
    run<5>("5 2x2 blocks, reciprocals side by side, A + Bt", tin, W, Li, d, cyc, true);
    run<6>("6 2x2 blocks, reciprocals side by side, A only", tin, W, Li, d, cyc, false);
    return 0;
}
