// upr_qp3_list.h -- the instantiations of the production QP kernel (upr_qp3.h) that libupright_mi carries, in PARTS: every part
// is compiled as a translation unit of its own (upr_qp3_inst.hip with -DUPR_QP3_PART=k; __graft_entry__.build() runs them side by
// side), upr_api.hip only declares the launchers.  X(nq, nb, nc, nf, N, ROWS, SOFT, DENSE) at 256 lanes; the headline shape
// (9, 1, 4, 3; N = 20; hard rows) with and without state-polytopic rows: Y(NT, ROWS).  (Rounds 1 - 5 also shipped it at 128 and 512
// lanes for A/B runs: both slower, 512 - 655 spilled registers, selected by nothing but UPR_QP_NT -- dropped in round 6;
// UPR_JIT_NT still instantiates them at run time for an experiment.)
// A problem takes the FIRST entry of UPR_QP3_EXTRA that matches it (upr_api.hip, qp3_match).
#pragma once

// (nq, nb, nc, nf, ROWS, SOFT, DENSE) instantiations besides the headline's: the headline shape with slacks on its boxes,
// thing_demo (one body, frictionless, slacks), the upright_robust 8-corner arrangement (star, slacks), and box_arch
// (three stacked bodies that share contacts: dense Schur complement; with the collision rows of obstacles/simple.yaml)
#if defined(UPR_HEADLINE_ONLY) && defined(UPR_EXP_SHAPES)
#define UPR_QP3_EXTRA(X) X(9, 2, 8, 3, 20, false, false, true) X(6, 1, 4, 3, 20, false, false, false) X(9, 7, 28, 3, 20, false, false, false)
#elif defined(UPR_HEADLINE_ONLY) && defined(UPR_EXP_CONFIG3)
#define UPR_QP3_EXTRA(X) X(9, 3, 16, 3, 20, true, false, true) X(9, 8, 32, 1, 20, false, true, false)
#elif defined(UPR_HEADLINE_ONLY)
#define UPR_QP3_EXTRA(X) X(9, 1, 4, 3, 20, false, true, false) X(9, 1, 4, 3, 20, true, true, false)
#else
// part 1: the headline shape with HPIPM slacks (boxes; friction / collision / projectile rows), thing_demo (frictionless)
#define UPR_QP3_PART1(X) X(9, 1, 4, 3, 20, false, true, false) X(9, 1, 4, 3, 20, true, true, false) X(9, 1, 4, 1, 20, false, true, false)
// part 2: the upright_robust 8-corner arrangement (star, frictionless, softened equality)
#define UPR_QP3_PART2(X) X(9, 8, 32, 1, 20, false, true, false)
// part 3: box_arch (three stacked bodies: dense 18 x 18 Schur complement) with the collision rows of obstacles/simple.yaml
#define UPR_QP3_PART3(X) X(9, 3, 16, 3, 20, true, false, true)
// part 4: the six-joint chains (ur10_demo at N = 20 and 10; arm-only runs with friction)
#define UPR_QP3_PART4(X) X(6, 1, 4, 1, 20, false, true, false) X(6, 1, 4, 1, 10, false, true, false) X(6, 1, 4, 3, 20, false, false, false)
// part 5 (round 4): the paper's dice (two stacked bodies, dense 12 x 12) and seven cups (star with friction: upr_qp3_cfg::BIGF)
#define UPR_QP3_PART5(X) X(9, 2, 8, 3, 20, false, false, true) X(9, 7, 28, 3, 20, false, false, false)
// part 6 (round 6): upright_robust at the reference's OWN horizon (upright_robust/config/demos/_base.yaml:62-66: time_horizon 10 s, dt 0.1:
// N = 100) -- the whole-horizon arrays in a far array (upr_qp3_cfg::KFAR)
#define UPR_QP3_PART6(X) X(9, 8, 32, 1, 100, false, true, false)
#define UPR_QP3_NPARTS 7   /* part 0: the headline */
#define UPR_QP3_EXTRA(X) UPR_QP3_PART1(X) UPR_QP3_PART2(X) UPR_QP3_PART3(X) UPR_QP3_PART4(X) UPR_QP3_PART5(X) UPR_QP3_PART6(X)
#endif

// the headline's own instantiations: Y(NT, ROWS)
#define UPR_QP3_PART0(Y) Y(256, false) Y(256, true)
#define UPR_QP3_HEADLINE(Y) UPR_QP3_PART0(Y)
