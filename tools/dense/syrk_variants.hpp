// variants under test (syrk_test.hip)
// p: persistent workgroups (2 per CU) that fetch tiles from a counter; the second workgroup of every CU starts half a tile late, so that one
// workgroup's read-modify-write of C runs beside the other's MFMA loop instead of both waiting for memory at the same time
__device__ int g_counter[64], g_cuslots[8 * 256], g_placement[1024];
__device__ long long g_pst[4][16];
template <int STAG, int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void syrk128p(double* __restrict__ S, const double* __restrict__ W0, const double* __restrict__ W1, int npad, int k0, int jb0, int ntiles, int pass, int nsleep) {
    __shared__ double As[2][S128_KC * S128_LD], Bs[2][S128_KC * S128_LD];
    __shared__ int s_tile[2];
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, lk = lane >> 4;
    const int r0w = (w & 1) * 64, c0w = (w >> 1) * 64;
    if (t == 0) {
        s_tile[0] = atomicAdd(&g_counter[pass], 1);
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        const int key = (int)((xcc & 7) << 8 | ((hw >> 8) & 0xff));
        s_tile[1] = atomicAdd(&g_cuslots[key], 1);
        if (blockIdx.x < 1024) g_placement[blockIdx.x] = key;
    }
    __syncthreads();
    int tix = s_tile[0];
    if (STAG && (s_tile[1] & 1)) for (int i = 0; i < nsleep; ++i) __builtin_amdgcn_s_sleep(127);
    const int cr = t & 127, kq = t >> 7;
    constexpr int NCP = S128_KC / 2, CPP = NB / S128_KC, NCH = 2 * NB / S128_KC;
    int nt = 0; const int sw = blockIdx.x == 0 ? 0 : blockIdx.x == 256 ? 1 : blockIdx.x == 300 ? 2 : blockIdx.x == 511 ? 3 : -1;
    while (tix < ntiles) {
        if (sw >= 0 && t == 0 && nt < 7) { g_pst[sw][2 * nt] = __builtin_amdgcn_s_memtime(); }
        __syncthreads();
        if (t == 0) s_tile[0] = atomicAdd(&g_counter[pass], 1);
        int ti = (int)((sqrt(8.0 * tix + 1.0) - 1.0) * 0.5); while (ti * (ti + 1) / 2 > tix) --ti; while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti; const int tj = tix - ti * (ti + 1) / 2;
        const int I0 = jb0 * NB + 128 * ti, J0 = jb0 * NB + 128 * tj;
        const bool active = I0 + r0w < npad && J0 + c0w < npad && !(ti == tj && c0w > r0w);
        double4_t acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = double4_t{0, 0, 0, 0};
        const int arow = I0 + cr < npad ? I0 + cr : npad - 1, brow = J0 + cr < npad ? J0 + cr : npad - 1;
        double ra[NCP], rb[NCP];
        auto gload = [&](int chunk) {
            const int q = chunk / CPP, col0 = (chunk % CPP) * S128_KC;
            const double* Wq = q == 0 ? W0 : W1;
            const double* Ga = Wq + (size_t)arow + (size_t)npad * col0;
            const double* Gb = S + (size_t)brow + (size_t)npad * ((size_t)(k0 + q) * NB + col0);
#pragma unroll
            for (int i = 0; i < NCP; ++i) { ra[i] = Ga[(size_t)npad * (kq + 2 * i)]; rb[i] = Gb[(size_t)npad * (kq + 2 * i)]; }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < NCP; ++i) { As[buf][(kq + 2 * i) * S128_LD + cr] = ra[i]; Bs[buf][(kq + 2 * i) * S128_LD + cr] = rb[i]; }
        };
        gload(0); lstore(0);
        __syncthreads();
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
            const int buf = (VAR & 2) ? 0 : (ch & 1);
            if (!(VAR & 2) && ch + 1 < NCH) gload(ch + 1);
            if (active) {
#pragma unroll
                for (int kk = 0; kk < S128_KC; kk += 4) {
                    double av[4], bv[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) av[a] = As[buf][(kk + lk) * S128_LD + r0w + 16 * a + li];
#pragma unroll
                    for (int b2 = 0; b2 < 4; ++b2) bv[b2] = Bs[buf][(kk + lk) * S128_LD + c0w + 16 * b2 + li];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[b2], av[a], acc[a][b2], 0, 0, 0);
                }
            }
            if (!(VAR & 2) && ch + 1 < NCH) lstore(buf ^ 1);
            __syncthreads();
        }
        tix = s_tile[0];
        if (sw >= 0 && t == 0 && nt < 7) { g_pst[sw][2 * nt + 1] = __builtin_amdgcn_s_memtime(); } ++nt;
        if (active) {
            double* Cg = S + (size_t)(I0 + r0w) + (size_t)npad * (J0 + c0w);
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) {
                double cold[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cold[a][r] = (VAR & 1) ? 1.0 : Cg[(size_t)(16 * a + li) + (size_t)npad * (16 * b2 + lk + 4 * r)];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { if (VAR & 4) { if (cold[a][r] - acc[a][b2][r] == 1.2345) Cg[0] = 1.0; } else Cg[(size_t)(16 * a + li) + (size_t)npad * (16 * b2 + lk + 4 * r)] = cold[a][r] - acc[a][b2][r]; }
            }
        }
    }
}
template <class R> void run_variants(R& run, double* S, double* W, int npad, int k0, int jb0, int T128, int ntiles) {
    int pass = 0;
    auto zero = [&]() { (void)hipMemset((void*)nullptr, 0, 0); int* c; (void)hipGetSymbolAddress((void**)&c, HIP_SYMBOL(g_counter)); (void)hipMemset(c, 0, 64 * 4); pass = 0; };
    zero(); run("p: persistent, dynamic tiles", [&]() { hipLaunchKernelGGL((syrk128p<0, 0>), dim3(512), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0, ntiles, pass++, 0); }, true);
    zero(); run("p, plain store", [&]() { hipLaunchKernelGGL((syrk128p<0, 1>), dim3(512), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0, ntiles, pass++, 0); }, false);
    zero(); run("p, no operand loads in the loop", [&]() { hipLaunchKernelGGL((syrk128p<0, 2>), dim3(512), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0, ntiles, pass++, 0); }, false);
    zero(); run("p, neither", [&]() { hipLaunchKernelGGL((syrk128p<0, 3>), dim3(512), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0, ntiles, pass++, 0); }, false);
    zero(); run("p, operand loads but no C traffic at all", [&]() { hipLaunchKernelGGL((syrk128p<0, 5>), dim3(512), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0, ntiles, pass++, 0); }, false);
    zero(); run("p, no memory traffic at all", [&]() { hipLaunchKernelGGL((syrk128p<0, 7>), dim3(512), dim3(256), 0, 0, S, W, W + (size_t)npad * 64, npad, k0, jb0, ntiles, pass++, 0); }, false);
    { long long q[4][16]; (void)hipMemcpyFromSymbol(q, HIP_SYMBOL(g_pst), sizeof(q));
      for (int w2 = 0; w2 < 4; ++w2) { printf("  WG %d: tile start / loop end, cycles from the first:", w2 == 0 ? 0 : w2 == 1 ? 256 : w2 == 2 ? 300 : 511); for (int i = 0; i < 8 && q[w2][i]; ++i) printf(" %lld", q[w2][i] - q[0][0]); printf("\n"); } }
    int h[1024]; (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_placement), 4096);
    printf("placement (xcc<<8 | se,sh,cu) of workgroups 0..23: "); for (int i = 0; i < 24; ++i) printf("%x ", h[i]); printf("\n  256..271: "); for (int i = 256; i < 272; ++i) printf("%x ", h[i]); printf("\n");
    int cnt[2048] = {0}; for (int i = 0; i < 512; ++i) cnt[h[i] & 2047]++; int n1 = 0, n2 = 0, n3 = 0; for (int i = 0; i < 2048; ++i) { n1 += cnt[i] == 1; n2 += cnt[i] == 2; n3 += cnt[i] > 2; }
    printf("  CUs with 1 / 2 / more workgroups: %d %d %d\n", n1, n2, n3);
}
