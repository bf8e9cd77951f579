// tools/tile_lab.cpp — CPU lab for the clustering behind the two-sweep kernel: tiles of an n x n grid graph, their ring sizes.
//   g++ -O2 -std=c++17 -I cortex.jl_amd/csrc tools/tile_lab.cpp -o /tmp/tile_lab && /tmp/tile_lab 1415 256
#include <cstdio>
#include <cstdlib>
#include <map>

#include "cx_tiling.h"

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 300, cap = argc > 2 ? atoi(argv[2]) : 256;
    const int32_t nv = n * n;
    std::vector<int32_t> off(nv + 1, 0), adj;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            if (i > 0) adj.push_back((i - 1) * n + j);
            if (j > 0) adj.push_back(i * n + j - 1);
            if (j < n - 1) adj.push_back(i * n + j + 1);
            if (i < n - 1) adj.push_back((i + 1) * n + j);
            off[i * n + j + 1] = (int32_t)adj.size();
        }
    std::vector<int32_t> order, end;
    cx::bisect(off, adj, nv, cap, order, end);
    std::vector<int32_t> mark(nv, -1);
    std::map<int, int> hist;
    long own = 0, all = 0; int worst = 0, worst_own = 0, minown = 1 << 30;
    int32_t lo = 0;
    for (size_t t = 0; t < end.size(); t++) {
        std::vector<int32_t> r1, r2;
        for (int32_t i = lo; i < end[t]; i++) mark[order[i]] = (int32_t)t;
        for (int32_t i = lo; i < end[t]; i++) for (int32_t e = off[order[i]]; e < off[order[i] + 1]; e++) if (mark[adj[e]] != (int32_t)t) { mark[adj[e]] = (int32_t)t; r1.push_back(adj[e]); }
        for (int32_t v : r1) for (int32_t e = off[v]; e < off[v + 1]; e++) if (mark[adj[e]] != (int32_t)t) { mark[adj[e]] = (int32_t)t; r2.push_back(adj[e]); }
        const int o = end[t] - lo, a = o + (int)r1.size() + (int)r2.size();
        own += o; all += a; hist[a / 50 * 50]++;
        if (a > worst) { worst = a; worst_own = o; }
        if (o < minown) minown = o;
        lo = end[t];
    }
    printf("grid %d x %d, cap %d: %zu tiles, own/tile %.1f (min %d), loaded/own %.3f, worst tile: %d locals for %d own\n", n, n, cap, end.size(),
           (double)own / end.size(), minown, (double)all / own, worst, worst_own);
    for (auto &kv : hist) printf("  locals %4d..%4d: %d tiles\n", kv.first, kv.first + 49, kv.second);
    return 0;
}
