// tools/nd/nd_fuzz.cpp -- the symbolic phase of the tile-sparse solver (csrc/nlls_nd.cpp) on 300 random graphs (paths, grids, random sparse graphs, trees with hubs, nodes of
// 1..128 unknowns, 0..2 border nodes) under the CPU sanitizers (GPU AddressSanitizer is not available on this pool: sanitizers run on the CPU build only).
//   g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++20 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o nd_fuzz tools/nd/nd_fuzz.cpp \
//       nllssolver.jl_amd/csrc/nlls_nd.cpp -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib && ASAN_OPTIONS=detect_leaks=0 ./nd_fuzz
#include "../../nllssolver.jl_amd/csrc/nlls_tsp.hpp"
#include <random>
#include <cstdio>
#include <algorithm>
using namespace nlls;
int main() {
    std::mt19937 rng(1);
    for (int it = 0; it < 300; ++it) {
        int n = 1 + rng() % 1500; int kind = rng() % 5; int d = (kind == 4) ? 1 + rng() % 128 : 6;
        std::vector<std::vector<int32_t>> adj(n);
        auto add = [&](int a, int b) { if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
        if (kind == 0) for (int i = 0; i + 1 < n; ++i) add(i, i + 1);
        else if (kind == 1) { int w = 1 + rng() % 40; for (int i = 0; i < n; ++i) { if (i + 1 < n && (i + 1) % w) add(i, i + 1); if (i + w < n) add(i, i + w); if (i + w + 1 < n && (i + 1) % w) add(i, i + w + 1); } }
        else if (kind == 2) { int m = rng() % (4 * n + 1); for (int e = 0; e < m; ++e) add(rng() % n, rng() % n); }
        else if (kind == 3) { for (int i = 1; i < n; ++i) add(rng() % i, i); int h = rng() % 4; for (int q = 0; q < h; ++q) { int c = rng() % n; for (int i = 0; i < n; i += 2) add(c, i); } }
        else { int m = rng() % (2 * n + 1); for (int e = 0; e < m; ++e) add(rng() % n, rng() % n); }
        for (auto& l : adj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
        int nb = rng() % 3;
        std::vector<int32_t> dof(n + nb, d); for (int q = 0; q < nb; ++q) dof[n + q] = 1 + rng() % 6;
        TspSym sym; bool ok = tsp_symbolic(adj, dof, nb, sym);
        if (!ok) { printf("case %d declined\n", it); continue; }
        for (int v = 0; v < n + nb; ++v) if (sym.tile_of[v] < 0 || sym.tile_of[v] >= sym.nt || sym.row_in_tile[v] + dof[v] > 128) { printf("BAD placement case %d\n", it); return 1; }
        for (int k = 0; k < sym.nt; ++k) for (int32_t i : sym.cstruct[k]) if (i <= k || i >= sym.nt || sym.level[i] <= sym.level[k]) { printf("BAD struct case %d\n", it); return 1; }
    }
    printf("300 random graphs ok\n"); return 0;
}
