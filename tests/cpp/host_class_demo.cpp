// tests/cpp/host_class_demo.cpp — the C++ host classes of include/cortex_hip.hpp on the reference's test graphs.
//   g++ -std=c++17 -Iinclude tests/cpp/host_class_demo.cpp -o demo -L cortex.jl_amd -lcortex_hip -Wl,-rpath,$PWD/cortex.jl_amd
// Part 1: the SSM of test/inference_engine_tests.jl:436-481 (T = 6) through cortex::HipProcessor — a batch of mutually
//         independent signals first (process ... flush), then the whole-call override; prints the marginals.
// Part 2: the structured variational SSM of :1032-1071 (n = 8) through cortex::VmpProcessor; prints the posteriors.
// Exit code 77: no GPU (the library has no CPU fallback).
#include <cstdio>
#include <vector>

#include "cortex_hip.hpp"

int main() {
    try {
        {
            const int T = 6;
            const double y[T] = {2.1, 3.9, 6.2, 8.0, 9.7, 12.3};
            cortex::Handle h(cortex::make_config(0, 1, CX_SCHED_FUSED));
            // ids as BipartiteFactorGraphs hands them out: x 1..T, y T+1..2T, likelihood 2T+1..3T, transition 3T+1..4T-1
            std::vector<int64_t> ev, ef, fid, xs, ys, liks;
            std::vector<int32_t> kind;
            std::vector<double> par;
            for (int i = 0; i < T; i++) { ev.push_back(T + 1 + i); ef.push_back(2 * T + 1 + i); ev.push_back(1 + i); ef.push_back(2 * T + 1 + i); }
            for (int i = 0; i < T - 1; i++) { ev.push_back(1 + i); ef.push_back(3 * T + 1 + i); ev.push_back(2 + i); ef.push_back(3 * T + 1 + i); }
            for (int f = 0; f < 2 * T - 1; f++) { fid.push_back(2 * T + 1 + f); kind.push_back(CX_FACTOR_GAUSS_ADDITIVE); par.insert(par.end(), {1.0, 0.0, 0.0, 0.0}); }
            h.graph_create(ev, ef, fid, kind, par);
            for (int i = 0; i < T; i++) { xs.push_back(1 + i); ys.push_back(T + 1 + i); liks.push_back(2 * T + 1 + i); }
            h.set_messages(ys, liks, CX_TO_FACTOR, CX_FORM_POINT, std::vector<double>(y, y + T));
            cortex::HipProcessor proc(h);
            for (int i = 0; i < T; i++) proc.process(CX_ITEM_MESSAGE_TO_VARIABLE, xs[i], liks[i]);   // independent: one launch
            proc.flush();
            const std::vector<double> lik = h.get_messages(xs, liks, CX_TO_VARIABLE);
            for (int i = 0; i < T; i++) std::printf("lik %d %.15g %.15g\n", i + 1, lik[2 * i], lik[2 * i + 1]);
            const std::vector<double> m = proc.update_marginals(xs, T + 2);
            for (int i = 0; i < T; i++) std::printf("x %d %.15g %.15g\n", i + 1, m[2 * i], m[2 * i + 1]);
            std::printf("launches %lld\n", (long long)proc.launches());
            try {
                h.get_marginals({12345});
                std::printf("error MISSING\n");
            } catch (const cortex::Error &e) {
                std::printf("error %d %s\n", e.code, e.what());
            }
        }
        {
            const int n = 8;
            const double y[n] = {0.05, -0.02, 0.11, 0.23, 0.18, 0.31, 0.27, 0.40};
            cortex::Handle h(cortex::make_config(0, 1, CX_SCHED_CHAIN_SCAN, CX_FAMILY_VMP_STRUCTURED));
            // ssnoise 1, obsnoise 2, x 3..n+2, y n+3..2n+2, likelihood 2n+3..3n+2, transition 3n+3..4n+1
            std::vector<int64_t> ev, ef, fid, xs, ys;
            std::vector<int32_t> role, kind;
            for (int i = 0; i < n; i++) {
                const int64_t f = 2 * n + 3 + i;
                ev.push_back(n + 3 + i); ef.push_back(f); role.push_back(CX_ROLE_OUT);
                ev.push_back(3 + i); ef.push_back(f); role.push_back(CX_ROLE_IN);
                ev.push_back(2); ef.push_back(f); role.push_back(CX_ROLE_PRECISION);
                fid.push_back(f); kind.push_back(CX_FACTOR_NORMAL_PRECISION);
                xs.push_back(3 + i); ys.push_back(n + 3 + i);
            }
            for (int i = 0; i < n - 1; i++) {
                const int64_t f = 3 * n + 3 + i;
                ev.push_back(3 + i); ef.push_back(f); role.push_back(CX_ROLE_IN);
                ev.push_back(4 + i); ef.push_back(f); role.push_back(CX_ROLE_OUT);
                ev.push_back(1); ef.push_back(f); role.push_back(CX_ROLE_PRECISION);
                fid.push_back(f); kind.push_back(CX_FACTOR_NORMAL_PRECISION);
            }
            h.graph_create(ev, ef, fid, kind, {}, role);
            cortex::VmpProcessor vmp(h);
            vmp.set_gamma(1, 1.0, 1.0); vmp.set_gamma(2, 1.0, 1.0);
            for (int i = 0; i < n; i++) { vmp.set_normal(xs[i], 0.0, 1.0); vmp.observe(ys[i], y[i]); }
            for (int it = 0; it < 5; it++) {
                vmp.update_marginals(xs);
                vmp.update_marginals({1, 2});
            }
            const std::vector<double> g = vmp.marginals({1, 2}), xm = vmp.marginals(xs);
            std::printf("ssnoise %.15g %.15g\nobsnoise %.15g %.15g\n", g[0], g[1], g[2], g[3]);
            for (int i = 0; i < n; i++) std::printf("q %d %.15g %.15g\n", i + 1, xm[2 * i], xm[2 * i + 1]);
        }
    } catch (const cortex::Error &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return e.code == CX_ERR_NO_DEVICE ? 77 : 1;
    }
    return 0;
}
