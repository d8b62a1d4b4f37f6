// Where do the ~221 cycles per pivot of the 16x16 tile LDL' (csrc/nlls_bcr.hip: bcr_pivot_chain) go?  One wavefront, the chain with parts switched off, s_memtime around it.
//   hipcc -O3 --offload-arch=gfx950 -o pivot_chain tools/microbench/pivot_chain.hip && ./pivot_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double bdouble4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rl(double x, int k) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), k), __builtin_amdgcn_readlane(__double2loint(x), k)); }
__device__ __forceinline__ double refine(double d, double r) { const double e = fma(-d, r, 1.0); return fma(r, fma(e, e, e), r); }
// V: 0 full; 1 no Bt MFMA; 2 no refinement (raw rcp); 3 no rcp at all (a constant reciprocal); 4 no MFMA at all (vector FMA on one register instead); 5 pivot by DPP-free ds_bpermute broadcast instead of readlane
template <int V>
__device__ __forceinline__ void chain(bdouble4_t& A, bdouble4_t& Bt, int li, int lk) {
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int q = k & 3, r = k >> 2;
        const double w = A[r], bt = Bt[r];
        double dk;
        if constexpr (V == 5) dk = __hiloint2double(__builtin_amdgcn_ds_bpermute(4 * (16 * q + k), __double2hiint(w)), __builtin_amdgcn_ds_bpermute(4 * (16 * q + k), __double2loint(w)));
        else dk = rl(w, 16 * q + k);
        double rdk = V == 3 ? 0.5 : __builtin_amdgcn_rcp(dk);
        const bool rowq = lk == q;
        const double am = (rowq && li > k) ? w : 0.0;
        const double bm = rowq ? bt : 0.0;
        if constexpr (V != 2 && V != 3) rdk = refine(dk, rdk);
        if constexpr (V == 4) { A[r] = fma(am, am * -rdk, A[r]); A[(r + 1) & 3] = fma(am, bm * -rdk, A[(r + 1) & 3]); }
        else {
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
            if constexpr (V != 1) Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
        }
    }
}
// V == 6: the inverse's MFMA of pivot k - 1 DEFERRED behind the read-lane of pivot k (it then runs in the matrix pipe while the vector chain of pivot k -- reciprocal, refinement --
// is busy, instead of standing between two updates of the tile)
__device__ __forceinline__ void chain_deferred(bdouble4_t& A, bdouble4_t& Bt, int li, int lk) {
    double am_p = 0.0, bop_p = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int q = k & 3, r = k >> 2;
        const double w = A[r];
        const double dk = rl(w, 16 * q + k);
        if (k > 0) Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am_p, bop_p, Bt, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        double rdk = __builtin_amdgcn_rcp(dk);
        const bool rowq = lk == q;
        const double am = (rowq && li > k) ? w : 0.0;
        rdk = refine(dk, rdk);
        const double nr = -rdk;
        A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * nr, A, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const double bm = rowq ? Bt[r] : 0.0;          // (row k of the inverse after update k - 1: in flight until about now)
        am_p = am; bop_p = bm * nr;
    }
    Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am_p, bop_p, Bt, 0, 0, 0);
}
// V == 7: FOUR pivots per matrix-core instruction.  The four pivot columns of a group (rows 4g .. 4g + 3 of the symmetric tile) are register g of the whole wavefront -- lane (li, lk)
// holds entry li of column 4g + lk -- i.e. already the four K-slots of the A operand.  Inside the group the columns are eliminated against each other on the vector side (a column
// fetched into the other three row groups through the LDS crossbar, one FMA per pivot), then ONE instruction applies the group's four rank-1 updates to the rest of the tile and one
// to the inverse: 6 MFMAs per tile instead of 30 (the fp64 matrix core and the fp64 vector unit are one resource on this chip: an MFMA in flight is 64 cycles nothing else runs in).
__device__ __forceinline__ double bperm(double x, int byteidx) { return __hiloint2double(__builtin_amdgcn_ds_bpermute(byteidx, __double2hiint(x)), __builtin_amdgcn_ds_bpermute(byteidx, __double2loint(x))); }
__device__ __forceinline__ void chain_blocked(bdouble4_t& A, bdouble4_t& Bt, int li, int lk) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        double Ag = A[g], Bg = Bt[g], nrsel = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int k = 4 * g + p;
            if (k == 15) break;
            const double dk = rl(Ag, 16 * p + k);
            const double cp = bperm(Ag, 4 * (16 * p + li)), bp = bperm(Bg, 4 * (16 * p + li));      // column k / row k of the inverse, in every row group
            double rdk = __builtin_amdgcn_rcp(dk); rdk = refine(dk, rdk);
            const double nr = -rdk;
            nrsel = lk == p ? nr : nrsel;
            double l = 0.0;
#pragma unroll
            for (int pp = p + 1; pp < 4; ++pp) { const double lpp = rl(Ag, 16 * p + 4 * g + pp) * nr; l = lk == pp ? lpp : l; }
            Ag = fma(li > k ? cp : 0.0, l, Ag);
            Bg = fma(bp, l, Bg);
        }
        A[g] = Ag; Bt[g] = Bg;
        if (g < 3) {
            const double aop = li > 4 * g + 3 ? Ag : 0.0;
            const double bopA = li > 4 * g + lk ? Ag * nrsel : 0.0, bopB = Bg * nrsel;
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bopA, A, 0, 0, 0);
            Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bopB, Bt, 0, 0, 0);
        }
    }
}
// V == 8: as 7, but the four columns of the group (and the four rows of the inverse) are fetched into EVERY row group once, before the group's eliminations -- one LDS-crossbar
// round trip per group instead of one per pivot on the dependent chain -- and every row group eliminates all four redundantly (the same vector instructions)
__device__ __forceinline__ void chain_blocked2(bdouble4_t& A, bdouble4_t& Bt, int li, int lk) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        double c[4], b[4], nr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = bperm(A[g], 4 * (16 * q + li)); b[q] = bperm(Bt[g], 4 * (16 * q + li)); }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int k = 4 * g + p;
            if (k == 15) break;
            const double dk = rl(c[p], k);
            double rdk = __builtin_amdgcn_rcp(dk); rdk = refine(dk, rdk);
            nr[p] = -rdk;
            const double cm = li > k ? c[p] : 0.0;
#pragma unroll
            for (int pp = p + 1; pp < 4; ++pp) { const double l = rl(c[p], 4 * g + pp) * nr[p]; c[pp] = fma(cm, l, c[pp]); b[pp] = fma(b[p], l, b[pp]); }
        }
        const double Ag = lk == 0 ? c[0] : lk == 1 ? c[1] : lk == 2 ? c[2] : c[3], Bg = lk == 0 ? b[0] : lk == 1 ? b[1] : lk == 2 ? b[2] : b[3];
        const double nrsel = lk == 0 ? nr[0] : lk == 1 ? nr[1] : lk == 2 ? nr[2] : nr[3];
        A[g] = Ag; Bt[g] = Bg;
        if (g < 3) {
            const double aop = li > 4 * g + 3 ? Ag : 0.0;
            const double bopA = li > 4 * g + lk ? Ag * nrsel : 0.0, bopB = Bg * nrsel;
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bopA, A, 0, 0, 0);
            Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bopB, Bt, 0, 0, 0);
        }
    }
}
// V == 9: a REAL all-vector rank-1 update of the tile and of its inverse (no matrix-core instruction at all): row k of the symmetric tile sits in register k / 4 of the sixteen lanes
// with lk = k % 4; every lane fetches its column's entry T[k][li] and its four rows' entries T[k][lk + 4 r] (= T[lk + 4 r][k]) through the LDS crossbar (ds_bpermute_b32 x 2 per
// double: 12 per pivot with the inverse's row), then four FMAs per accumulator.  Same arithmetic per entry as the MFMA form (one product, one fused add).
__device__ __forceinline__ void chain_valu(bdouble4_t& A, bdouble4_t& Bt, int li, int lk) {
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int q = k & 3, r0 = k >> 2;
        const double w = A[r0], bt = Bt[r0];
        const double dk = rl(w, 16 * q + k);
        const double colv = bperm(w, 4 * (16 * q + li)), colb = bperm(bt, 4 * (16 * q + li));
        double rowv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rowv[r] = bperm(w, 4 * (16 * q + lk + 4 * r));
        double rdk = __builtin_amdgcn_rcp(dk); rdk = refine(dk, rdk);
        const double ca = li > k ? colv * -rdk : 0.0, cb = colb * -rdk;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const double a = lk + 4 * r > k ? rowv[r] : 0.0; A[r] = fma(a, ca, A[r]); Bt[r] = fma(a, cb, Bt[r]); }
    }
}
// V == 10: hybrid -- the tile's update on the matrix core, the inverse's on the vector side (its operands through the LDS crossbar: 10 ds_bpermute per pivot)
__device__ __forceinline__ void chain_hybrid(bdouble4_t& A, bdouble4_t& Bt, int li, int lk) {
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int q = k & 3, r0 = k >> 2;
        const double w = A[r0], bt = Bt[r0];
        const double dk = rl(w, 16 * q + k);
        const double colb = bperm(bt, 4 * (16 * q + li));
        double rowv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rowv[r] = bperm(w, 4 * (16 * q + lk + 4 * r));
        double rdk = __builtin_amdgcn_rcp(dk); rdk = refine(dk, rdk);
        const bool rowq = lk == q;
        const double am = (rowq && li > k) ? w : 0.0;
        A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
        const double cb = colb * -rdk;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const double a = lk + 4 * r > k ? rowv[r] : 0.0; Bt[r] = fma(a, cb, Bt[r]); }
    }
}
// V == 11: the library's chain ROLLED over the four pivots of a register group (the register A[r] of pivot k = 4 r + q must be a compile-time choice, the lane group q need not):
// four pivot bodies of code instead of fifteen -- a kernel launched once per level starts with a cold instruction cache, and the chain is the first thing wave 0 runs
__device__ __forceinline__ void chain_rolled(bdouble4_t& A, bdouble4_t& Bt, int li, int lk) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll 1
        for (int q = 0; q < (r == 3 ? 3 : 4); ++q) {
            const int k = 4 * r + q;
            const double w = A[r], bt = Bt[r];
            const double dk = rl(w, 16 * q + k);
            double rdk = __builtin_amdgcn_rcp(dk);
            const bool rowq = lk == q;
            const double am = (rowq && li > k) ? w : 0.0;
            const double bm = rowq ? bt : 0.0;
            rdk = refine(dk, rdk);
            A = __builtin_amdgcn_mfma_f64_16x16x4f64(am, am * -rdk, A, 0, 0, 0);
            Bt = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm * -rdk, Bt, 0, 0, 0);
        }
    }
}
template <int V>
__global__ void k(const double* T, double* out, long long* cyc, int reps) {
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    bdouble4_t A0, B0;
    for (int r = 0; r < 4; ++r) { A0[r] = T[(lk + 4 * r) * 16 + li]; B0[r] = (lk + 4 * r == li) ? 1.0 : 0.0; }
    double acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
        bdouble4_t A = A0, Bt = B0; A[0] += acc * 1e-300;        // (a dependence from one repetition to the next)
        if constexpr (V == 6) chain_deferred(A, Bt, li, lk); else if constexpr (V == 9) chain_valu(A, Bt, li, lk); else if constexpr (V == 10) chain_hybrid(A, Bt, li, lk); else if constexpr (V == 11) chain_rolled(A, Bt, li, lk); else if constexpr (V == 7) chain_blocked(A, Bt, li, lk); else if constexpr (V == 8) chain_blocked2(A, Bt, li, lk); else chain<V>(A, Bt, li, lk);
        acc += A[3] + Bt[3];
    }
    const long long t1 = __builtin_readcyclecounter();
    out[lane] = acc; if (lane == 0) cyc[0] = t1 - t0;
}
template <int V>
__global__ void dump(const double* T, double* outA, double* outB) {
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    bdouble4_t A, B; for (int r = 0; r < 4; ++r) { A[r] = T[(lk + 4 * r) * 16 + li]; B[r] = (lk + 4 * r == li) ? 1.0 : 0.0; }
    if constexpr (V == 7) chain_blocked(A, B, li, lk); else if constexpr (V == 8) chain_blocked2(A, B, li, lk); else if constexpr (V == 9) chain_valu(A, B, li, lk); else chain<0>(A, B, li, lk);
    for (int r = 0; r < 4; ++r) { outA[(lk + 4 * r) * 16 + li] = A[r]; outB[(lk + 4 * r) * 16 + li] = B[r]; }
}
int main() {
    double h[256]; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double *T, *out; long long* cyc; hipMalloc(&T, sizeof h); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMemcpy(T, h, sizeof h, hipMemcpyHostToDevice);
    const int reps = 2000; const char* names[] = {"full chain (as in the library)", "without the inverse's MFMA", "raw v_rcp_f64 (no refinement)", "no v_rcp_f64 (constant)", "no MFMA (vector FMAs)", "pivot by ds_bpermute instead of v_readlane", "the inverse's MFMA deferred behind the next read-lane", "four pivots per MFMA (in-group elimination on the vector side)", "... the group's columns fetched once, eliminated in every row group", "ALL-VECTOR rank-1 updates (12 ds_bpermute + 8 FMA per pivot, no MFMA)", "hybrid: tile on the matrix core, inverse on the vector side (10 ds_bpermute + 4 FMA)", "library chain rolled over the four pivots of a register group (4 bodies of code, not 15)"};
    auto run = [&](auto kern, int v) { long long c = 0; for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, T, out, cyc, reps); hipDeviceSynchronize(); } hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-48s %8.1f shader-clock cycles per pivot\n", names[v], (double)c / reps / 15.0); };
    run(k<0>, 0); run(k<1>, 1); run(k<2>, 2); run(k<3>, 3); run(k<4>, 4); run(k<5>, 5); run(k<6>, 6); run(k<7>, 7); run(k<8>, 8); run(k<9>, 9); run(k<10>, 10); run(k<11>, 11);
    { double *dA, *dB; hipMalloc(&dA, 256 * 8); hipMalloc(&dB, 256 * 8); double a0[256], b0[256], a7[256], b7[256];
      hipLaunchKernelGGL(dump<0>, dim3(1), dim3(64), 0, 0, T, dA, dB); hipMemcpy(a0, dA, sizeof a0, hipMemcpyDeviceToHost); hipMemcpy(b0, dB, sizeof b0, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(dump<8>, dim3(1), dim3(64), 0, 0, T, dA, dB); hipMemcpy(a7, dA, sizeof a7, hipMemcpyDeviceToHost); hipMemcpy(b7, dB, sizeof b7, hipMemcpyDeviceToHost);
      double ea = 0, eb = 0, eal = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { const double da = fabs(a0[i * 16 + j] - a7[i * 16 + j]) / (fabs(a0[i * 16 + j]) + 1e-300), db = fabs(b0[i * 16 + j] - b7[i * 16 + j]) / (fabs(b0[i * 16 + j]) + 1e-300); if (da > ea) ea = da; if (i >= j && da > eal) eal = da; if (db > eb && fabs(b0[i * 16 + j]) > 1e-300) eb = db; }
      printf("blocked against the library chain: max rel diff of the factored tile %.2e (lower triangle %.2e), of the inverse %.2e\n", ea, eal, eb); }
    { double *dA, *dB; hipMalloc(&dA, 256 * 8); hipMalloc(&dB, 256 * 8); double a0[256], b0[256], a9[256], b9[256];
      hipLaunchKernelGGL(dump<0>, dim3(1), dim3(64), 0, 0, T, dA, dB); hipMemcpy(a0, dA, sizeof a0, hipMemcpyDeviceToHost); hipMemcpy(b0, dB, sizeof b0, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(dump<9>, dim3(1), dim3(64), 0, 0, T, dA, dB); hipMemcpy(a9, dA, sizeof a9, hipMemcpyDeviceToHost); hipMemcpy(b9, dB, sizeof b9, hipMemcpyDeviceToHost);
      int same = 1; for (int i = 0; i < 256; ++i) same &= (a0[i] == a9[i]) && (b0[i] == b9[i]);
      printf("all-vector == library chain, bit for bit: %s\n", same ? "yes" : "NO"); }
    { double o0[64], o11[64]; hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, T, out, cyc, 1); hipMemcpy(o0, out, sizeof o0, hipMemcpyDeviceToHost); long long c0 = 0, c11 = 0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(k<11>, dim3(1), dim3(64), 0, 0, T, out, cyc, 1); hipMemcpy(o11, out, sizeof o11, hipMemcpyDeviceToHost); hipMemcpy(&c11, cyc, 8, hipMemcpyDeviceToHost);
      int same = 1; for (int i = 0; i < 64; ++i) same &= o0[i] == o11[i]; printf("rolled == library chain, bit for bit: %s;  ONE chain in a fresh launch (cold instruction cache): library %lld cycles, rolled %lld\n", same ? "yes" : "NO", c0, c11); }
    { double o0[64], o6[64]; hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, T, out, cyc, 1); hipMemcpy(o0, out, sizeof o0, hipMemcpyDeviceToHost); hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, T, out, cyc, 1); hipMemcpy(o6, out, sizeof o6, hipMemcpyDeviceToHost);
      int same = 1; for (int i = 0; i < 64; ++i) same &= o0[i] == o6[i]; printf("deferred == library chain, bit for bit: %s\n", same ? "yes" : "NO"); }
    return 0;
}
