// Drives libupright_mi.so through include/upright_mi.hpp the way a C++ MPC node would (mpc_node.cpp:32-60 on the
// reference side).  Inputs are raw little-endian blobs written by the test: the upr_problem struct, body parameters,
// targets and start states.  Prints the solution of every instance; without a GPU the constructor must throw.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "upright_mi.hpp"

static std::vector<char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + p);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static std::vector<double> doubles(const std::string& p) {
    std::vector<char> b = slurp(p);
    std::vector<double> v(b.size() / sizeof(double));
    std::memcpy(v.data(), b.data(), v.size() * sizeof(double));
    return v;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: hpp_demo <dir> <B>\n"); return 2; }
    const std::string dir = argv[1];
    const int B = std::atoi(argv[2]);
    try {
        std::vector<char> pb = slurp(dir + "/problem.bin");
        if (pb.size() != sizeof(upr_problem)) { std::fprintf(stderr, "struct size mismatch: %zu vs %zu\n", pb.size(), sizeof(upr_problem)); return 3; }
        upr_problem P;
        std::memcpy(&P, pb.data(), sizeof(P));
        upright_mi::ControllerInterface ctrl(P, B, doubles(dir + "/body_params.bin"), doubles(dir + "/way_p.bin"));
        std::vector<double> x0 = doubles(dir + "/x0.bin");
        ctrl.set_observation({0.0}, x0);
        ctrl.advance();
        upright_mi::Solution s = ctrl.solution();
        std::vector<double> xo, uo;
        ctrl.evaluate_policy({0.05}, x0, xo, uo);
        std::printf("nx %d nu %d N %d\n", ctrl.state_dim(), ctrl.input_dim(), ctrl.horizon());
        std::printf("xs"); for (double v : s.xs) std::printf(" %.17g", v); std::printf("\n");
        std::printf("us"); for (double v : s.us) std::printf(" %.17g", v); std::printf("\n");
        std::printf("upol"); for (double v : uo) std::printf(" %.17g", v); std::printf("\n");
        if (P.use_feedback_policy) { std::vector<double> K = ctrl.feedback_gains(); double a = 0; for (double v : K) a += v * v; std::printf("Knorm2 %.17g\n", a); }
        // a second control period through tick(): observation (the same state, 10 ms later), warm-started solve, policy at it
        std::vector<double> xt, ut;
        ctrl.tick({0.01}, x0, xt, ut);
        std::printf("utick"); for (double v : ut) std::printf(" %.17g", v); std::printf("\n");
    } catch (const std::runtime_error& e) {
        std::printf("runtime_error: %s\n", e.what());
        return 1;
    }
    return 0;
}
