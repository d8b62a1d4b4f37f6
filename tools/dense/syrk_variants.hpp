// variants under test (syrk_test.hip)
// w8 (the library's kernel) with the operand loads TWO chunks ahead (two register sets)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void syrk128w8pf2(double* __restrict__ S, const double* __restrict__ W0, const double* __restrict__ W1, int npad, int k0, int jb0) {
    __shared__ double As[2][S128_KC * S128_LD], Bs[2][S128_KC * S128_LD];
    int ti, tj;
    { const int tix = blockIdx.x; ti = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > tix) --ti; while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti; tj = tix - ti * (ti + 1) / 2; }
    const int I0 = jb0 * NB + 128 * ti, J0 = jb0 * NB + 128 * tj;
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, lk = lane >> 4;
    const int r0w = (w & 1) * 64, c0w = (w >> 1) * 32;
    const bool active = I0 + r0w < npad && J0 + c0w < npad && !(ti == tj && c0w >= r0w + 64);
    double4_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = double4_t{0, 0, 0, 0};
    const int cr = t & 127, kq = t >> 7;
    const int arow = I0 + cr < npad ? I0 + cr : npad - 1, brow = J0 + cr < npad ? J0 + cr : npad - 1;
    constexpr int NCP = S128_KC / 4, CPP = NB / S128_KC, NCH = 2 * NB / S128_KC;
    double raA[NCP], rbA[NCP], raB[NCP], rbB[NCP];
    auto gload = [&](int chunk, double (&ra)[NCP], double (&rb)[NCP]) {
        const int q = chunk / CPP, col0 = (chunk % CPP) * S128_KC;
        const double* Wq = q == 0 ? W0 : W1;
        const double* Ga = Wq + (size_t)arow + (size_t)npad * col0;
        const double* Gb = S + (size_t)brow + (size_t)npad * ((size_t)(k0 + q) * NB + col0);
#pragma unroll
        for (int i = 0; i < NCP; ++i) { ra[i] = Ga[(size_t)npad * (kq + 4 * i)]; rb[i] = Gb[(size_t)npad * (kq + 4 * i)]; }
    };
    auto lstore = [&](int buf, const double (&ra)[NCP], const double (&rb)[NCP]) {
#pragma unroll
        for (int i = 0; i < NCP; ++i) { As[buf][(kq + 4 * i) * S128_LD + cr] = ra[i]; Bs[buf][(kq + 4 * i) * S128_LD + cr] = rb[i]; }
    };
    auto compute = [&](int buf) {
        if (!active) return;
#pragma unroll
        for (int kk = 0; kk < S128_KC; kk += 4) {
            double av[4], bv[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[buf][(kk + lk) * S128_LD + r0w + 16 * a + li];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) bv[b2] = Bs[buf][(kk + lk) * S128_LD + c0w + 16 * b2 + li];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b2], av[a], acc[a][b2], 0, 0, 0);
        }
    };
    gload(0, raA, rbA); lstore(0, raA, rbA); gload(1, raA, rbA); gload(2, raB, rbB);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < NCH; ch += 2) {
        compute(0); lstore(1, raA, rbA); if (ch + 3 < NCH) gload(ch + 3, raA, rbA);
        __syncthreads();
        compute(1); if (ch + 2 < NCH) lstore(0, raB, rbB); if (ch + 4 < NCH) gload(ch + 4, raB, rbB);
        __syncthreads();
    }
    if (!active) return;
    double* Cg = S + (size_t)(I0 + r0w) + (size_t)npad * (J0 + c0w);
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
        double cold[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) cold[a][r] = Cg[(size_t)(16 * a + li) + (size_t)npad * (16 * b2 + lk + 4 * r)];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cg[(size_t)(16 * a + li) + (size_t)npad * (16 * b2 + lk + 4 * r)] = cold[a][r] - acc[a][b2][r];
    }
}
template <class R> void run_variants(R& run, double* S, double* W, int npad, int k0, int jb0, int T128, int ntiles) {
    run("w8 with the operand loads two chunks ahead", [&]() { hipLaunchKernelGGL(syrk128w8pf2, dim3(ntiles), dim3(512), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0); }, true);
    for (int grid : {1, 256, 512}) { char nm[96]; snprintf(nm, 96, "w8 two ahead, first %d tiles only", grid);
        run(nm, [&]() { hipLaunchKernelGGL(syrk128w8pf2, dim3(grid), dim3(512), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0); }, false); }
}
