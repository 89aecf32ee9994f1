// upright_mi.hpp -- C++ face of the C-ABI in upright_mi.h for C++ callers (the ROS nodes of the reference construct
// `upright::ControllerInterface` directly: upright_ros_interface/src/mpc_node.cpp:32-46, mrt_node.cpp:63-71).
//
// The reference class (upright_control/include/upright_control/controller_interface.h:23-95) hands out OCS2 objects
// (`get_mpc()` -> ocs2::MPC_BASE, `get_rollout()`, `get_pinocchio_interface()`); none of those types exists without the
// OCS2 fork, so this twin exposes the same OPERATIONS with plain containers: the calls an MPC node makes on the
// ocs2::MPC_MRT_Interface it builds from `get_mpc()` -- setCurrentObservation, advanceMpc, evaluatePolicy /
// getLinearFeedbackGain, the optimised trajectories -- plus `get_initial_state()`.  Errors are std::runtime_error with
// the library's message, as the reference throws (balancing_constraints.cpp:41-46).  Header only; link libupright_mi.so.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "upright_mi.h"

namespace upright_mi {

struct Solution {
    std::vector<double> ts, xs, us;   // [B][N+1], [B][N+1][nx], [B][N][nu]
};

class ControllerInterface {
   public:
    // body_params[B][nb][10], way_p[B][n_way][3]; B = 1 is the reference's single controller
    ControllerInterface(const upr_problem& problem, int B, const std::vector<double>& body_params, const std::vector<double>& way_p)
        : P_(problem), B_(B) {
        nx_ = 3 * P_.nq;
        nu_ = P_.nq + P_.nf * P_.nc;
        if ((int)body_params.size() != B * P_.nb * 10) throw std::runtime_error("body_params must hold B * nb * 10 values");
        if ((int)way_p.size() != B * P_.n_way * 3) throw std::runtime_error("way_p must hold B * n_way * 3 values");
        h_ = upr_batch_create(&P_, B, body_params.data(), way_p.data());
        if (!h_) throw std::runtime_error(upr_last_error());
    }
    ~ControllerInterface() { if (h_) upr_batch_destroy(h_); }
    ControllerInterface(const ControllerInterface&) = delete;
    ControllerInterface& operator=(const ControllerInterface&) = delete;

    int batch() const { return B_; }
    int state_dim() const { return nx_; }
    int input_dim() const { return nu_; }
    int horizon() const { return P_.N; }

    // ReferenceManager::setTargetTrajectories / MPC reset (mpc_node.cpp:48-52)
    void reset(const std::vector<double>& way_p) { check(upr_batch_reset(h_, way_p.empty() ? nullptr : way_p.data())); }
    // MPC_MRT_Interface::setCurrentObservation: t[B] (or one value for all), x[B][nx]
    void set_observation(const std::vector<double>& t, const std::vector<double>& x) {
        if ((int)x.size() != B_ * nx_) throw std::runtime_error("x must hold B * nx values");
        if ((int)t.size() != B_ && t.size() != 1) throw std::runtime_error("t must hold B values or one");
        check(upr_batch_set_observation(h_, t.data(), t.size() == 1 ? 0 : 1, x.data()));
    }
    // MPC_MRT_Interface::advanceMpc
    void advance() { check(upr_batch_advance(h_)); }
    // MPC_MRT_Interface::evaluatePolicy(t, x, x_opt, u_opt): the feedback law when sqp.use_feedback_policy is set
    void evaluate_policy(const std::vector<double>& t, const std::vector<double>& x, std::vector<double>& x_opt, std::vector<double>& u_opt) {
        x_opt.assign((size_t)B_ * nx_, 0.0); u_opt.assign((size_t)B_ * nu_, 0.0);
        const int stride = t.size() == 1 ? 0 : 1;
        if (P_.use_feedback_policy) check(upr_batch_evaluate_policy(h_, t.data(), stride, x.data(), x_opt.data(), u_opt.data()));
        else check(upr_batch_evaluate(h_, t.data(), stride, x_opt.data(), u_opt.data()));
    }
    // one control period of mpc_node / mrt_node in one call: setCurrentObservation + advanceMpc + evaluatePolicy at the observed
    // state (upr_batch_tick: one upload, one synchronisation; bit-identical to the three calls above)
    void tick(const std::vector<double>& t, const std::vector<double>& x, std::vector<double>& x_opt, std::vector<double>& u_opt) {
        if ((int)x.size() != B_ * nx_) throw std::runtime_error("x must hold B * nx values");
        if ((int)t.size() != B_ && t.size() != 1) throw std::runtime_error("t must hold B values or one");
        x_opt.assign((size_t)B_ * nx_, 0.0); u_opt.assign((size_t)B_ * nu_, 0.0);
        check(upr_batch_tick(h_, t.data(), t.size() == 1 ? 0 : 1, x.data(), x_opt.data(), u_opt.data(), nullptr));
    }
    // control periods step() served by replaying its captured HIP graph (steady state: from the fourth period on)
    long long tick_graph_replays() const { return upr_batch_tick_graph_replays(h_); }
    Solution solution() {
        Solution s;
        s.ts.assign((size_t)B_ * (P_.N + 1), 0.0); s.xs.assign((size_t)B_ * (P_.N + 1) * nx_, 0.0); s.us.assign((size_t)B_ * P_.N * nu_, 0.0);
        check(upr_batch_get_solution(h_, s.ts.data(), s.xs.data(), s.us.data()));
        return s;
    }
    // gains K[B][N][nu][nx] of the linear policy (LinearController::gainArray_)
    std::vector<double> feedback_gains() {
        std::vector<double> K((size_t)B_ * P_.N * nu_ * nx_);
        check(upr_batch_get_feedback(h_, K.data()));
        return K;
    }
    std::vector<double> stats() {
        std::vector<double> s((size_t)B_ * UPR_NSTATS);
        check(upr_batch_get_stats(h_, s.data()));
        return s;
    }
    double last_solve_ms() const { return upr_batch_last_solve_ms(h_); }
    upr_batch* handle() { return h_; }

   private:
    static void check(int rc) { if (rc) throw std::runtime_error(upr_last_error()); }
    upr_problem P_;
    int B_, nx_ = 0, nu_ = 0;
    upr_batch* h_ = nullptr;
};

}  // namespace upright_mi
