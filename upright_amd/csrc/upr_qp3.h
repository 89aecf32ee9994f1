// upr_qp3.h -- production QP kernel (third structure).  Same algorithm and arithmetic as upr_qp.h /
// upr_qp2.h (Mehrotra predictor-corrector IPM over a square-root Riccati recursion, Schur-complement
// treatment of the object-dynamics equality, proximal terminal equality); what changes is where the
// data lives and how many dependent phases a knot costs:
//
//   * horizon N and workgroup size NT are template parameters next to (nq, nb, nc, nf);
//   * slacks / multipliers of the state and input boxes live in REGISTERS through the stage-parallel phases: every lane owns a fixed
//     set of (knot, variable) box pairs for the whole solve (parked in a far array around the sweeps, which need every register);
//     the friction-pyramid rows -- the contact-point x cone-facet loop of contact_constraints.h:50-77 -- are a row per lane item;
//   * iterate, step, every per-knot vector (reduced gradients, barrier diagonals, equality residuals) and all problem constants are
//     LDS resident (KFAR instantiations: behind the pointer LK in a far array); global memory holds the read-only linearisation
//     records and the per-knot factors / Riccati feedback that the back-substitutions re-read (Infinity-Cache resident);
//   * the matrix part of the backward sweep runs on TWO waves with the cost-to-go in registers (block-scalar structure of the
//     triple integrator, system_dynamics.h:15-22) and two LDS-only barriers a knot; the vector / forward sweeps on one wave with the
//     state in registers (DPP, v_readlane, ds_bpermute).
// DESIGN.md section 3.2 is the description; profiles/NOTES_r*.md the measurements behind every choice.
#pragma once
#include "upr_kin.h"
#include "upr_qp.h"
#include "upr_qp2.h"

// ROWS_: the instantiation takes state-polytopic rows (collision / projectile).  Problems without such rows run the
// ROWS_ = false instantiation, in which every trace of them folds away (their runtime trip counts and the runtime
// Hessian offset cost 1.8 % of the headline solve when they were compiled in unconditionally); layouts do not depend on it.
// SOFT_: the instantiation carries HPIPM slacks on the lane-owned box rows (state and / or input boxes, selected at run
// time by upr_problem::soft_state_box / soft_input_box): slack, its own barrier pair and the second corrector target
// per row.  Hard problems run the SOFT_ = false instantiation.  (A softened object-dynamics equality, soft_eq, needs no
// state of its own -- it is the runtime regularisation rho_s = 1 / Z of the Schur complement -- and works in both.)
// NB_ > 1 ("star" arrangements: every contact point joins the tray and one balanced body, so no two bodies share a
// contact): the Schur complement of the object-dynamics rows is block diagonal, one 6 x 6 block per (knot, body); all
// Schur-related phases run one lane per block exactly as the single-body kernel runs one lane per knot.
// DENSE_ (with NB_ > 1): arrangements whose bodies share contact points (stacked objects): the Schur complement of a
// knot is one dense 6 NB x 6 NB matrix, assembled, factored and inverted by the knot's lane in registers (the
// instantiation runs one workgroup per CU and may use all 512 registers of a lane).
#ifndef UPR_QP3_INCEK
#define UPR_QP3_INCEK 1
#endif
#ifndef UPR_QP3_GQ
#define UPR_QP3_GQ 5   /* rows of C a lane requests together in the multi-body shapes' passes over C (prep C, forward tail) */
#endif
#ifndef UPR_QP3_SOFT_ROWMEM
#define UPR_QP3_SOFT_ROWMEM 1
#endif
template <int NQ_, int NB_, int NC_, int NF_, int N_, int NT_, bool ROWS_ = true, bool SOFT_ = false, bool DENSE_ = false>
struct upr_qp3_cfg {
    static constexpr int NQ = NQ_, NB = NB_, NC = NC_, NF = NF_, N = N_, NT = NT_;
    static constexpr bool ROWS = ROWS_, SOFT = SOFT_, COUPLED = DENSE_ && NB_ > 1, MULTI = NB_ > 1 && !DENSE_;
    // INCEK (star arrangements without friction: the upright_robust shape): the equality residual ek = e0 + C Zx + Df Zf is formed
    // once and then follows the iterate, ek += alpha (C sx + Df sf), with the C sx the forward sweep's tail has at hand -- one
    // pass over the rows of C (207 KB per instance) per interior-point iteration less: QP launch 6.31 -> 6.00 ms, same iteration
    // counts, plans to 2e-11.  Not for the dense shapes (-2.6 %): their configs run early QPs at the iteration cap, where a change
    // at rounding level moves the unconverged iterates by 1e-3 and with them test_config3_shape...'s comparison of whole SQP
    // runs; those kernels stay bit-identical.  (BIGF keeps ek in the far arrays.)
    static constexpr bool INCEK = UPR_QP3_INCEK && NB_ > 1 && !DENSE_ && NF_ == 1;
    // BIGF: star arrangements WITH friction (the paper's seven cups: nu = 93).  Their force-indexed arrays do not fit the LDS next
    // to the sweeps' working set: Df is kept compact ([force column][the six rows of the body its contact loads] instead of the
    // dense 6 nb x nf nc), Z = Lf^-1 Df' is not staged (the lane of a (knot, body) block forms its twelve columns from the contact
    // factors where it accumulates S_b), and the force part of the back-substitution hf and the equality residual ek -- touched
    // by the flat phases only -- live in the far arrays
    static constexpr bool BIGF = MULTI && NF_ == 3;
    // KFAR (round 6: the reference's own horizon for upright_robust, time_horizon 10 s = 100 knots, _base.yaml:62-66): the whole-horizon
    // arrays -- iterate, step, gradients, barrier diagonals, residuals, P+ b, feed-forward, defects: 475 doubles a knot for the robust
    // shape, 617 KB at N = 100 -- do not fit the LDS.  They live in a far array behind the instance's other far arrays and every
    // access goes through the pointer LK (= the LDS base in all other instantiations: same code, same code objects); LDS keeps the
    // constants, the sweeps' per-knot staging and the reductions.  The lane-owned box rows are streamed slot by slot out of their
    // parked copy instead of living in registers (28 slots of ten values a lane at 256 lanes).  What it costs is latency -- every
    // LDS-ordered hand-over becomes a global one -- and that is accepted: the alternative is the generic kernel, 40x slower per knot.
    static constexpr bool KFAR = N_ > 64;
    static_assert(!KFAR || (MULTI && !BIGF && !ROWS_), "long horizons: star arrangements without friction and without state-polytopic rows (the upright_robust shape)");
    static constexpr int SB = COUPLED ? 6 * NB_ : 6, NKB = COUPLED ? N_ : N_ * NB_;   // Schur block size; number of blocks
    static constexpr int NLS = COUPLED ? 36 * NB_ * NB_ : NB_ * 36;                   // doubles of the inverse Schur factor(s) of a knot
    static constexpr int N1 = N + 1;
    static constexpr int NX = 3 * NQ, NFC = NF * NC, NU = NQ + NFC, NE = 6 * NB;
    static constexpr int NP = (NF == 3) ? 5 * NC : 0;
    static constexpr int NLF = (NF == 3) ? 9 * NC : NC;
    static constexpr int NH = NQ * (NQ + 1) / 2;
    static constexpr int NZ = N1 * NX + N * NU;
    static constexpr int NEN = 3 + 2 * NQ;
    // SW: the matrix part of the backward sweep runs on ONE wave with the cost-to-go in registers (one-body shapes: a lane
    // per 3 x 3 block (joint i, joint j), i <= j, of P; a lane per column of Hux): no workgroup barrier inside the sweep
    static constexpr bool SW = NH <= 64 && 4 * NQ <= 64;
    // VCPRE: Vc = Lsi C of EVERY knot is left in LDS by prep (one-body shapes: 6 x nx per knot).  Otherwise (multi-body shapes:
    // 18 .. 48 rows per knot) the two waves that have no part in the sweep form Vc of the next knot into a double buffer
    static constexpr bool VCPRE = SW && NB_ == 1;
    // lane-owned inequality items
    static constexpr int NXI = N * NX, QX = (NXI + NT - 1) / NT;   // state boxes, knots 1..N
    static constexpr int NUI = N * NU, QU = (NUI + NT - 1) / NT;   // input boxes, knots 0..N-1
    static constexpr int NCI = N * NC, QC = (NCI + NT - 1) / NT;   // contacts (friction rows), knots 0..N-1
    // Riccati store per knot (global)
    static constexpr int SS_V = 0, SS_LJI = SS_V + NQ * NX, SS_YJ = SS_LJI + NQ * NQ, SS_PB = SS_YJ + NQ, SS_STRIDE = ((SS_PB + NX + 1) & ~1);
};

// "far" arrays: per-instance data that only the flat (stage-parallel) phases touch, plus the Riccati feedback
// store that the back-substitutions stream with a one-knot register prefetch.  They live in global memory
// (L2-resident) so that two workgroups fit the 160 KB of LDS of a CU.
#ifndef UPR_QP3_NOMAX
#define UPR_QP3_NOMAX 20   // state-polytopic rows per knot this kernel takes (more: the generic kernel); obstacles/simple.yaml has 20 pairs
#endif
template <class C>
struct upr_qp3_far {
    static constexpr int r2(int n) { return (n + 1) & ~1; }
    static constexpr int g0 = 0, e0 = g0 + r2(C::N * C::NQ), ek = e0 + r2(C::N * C::NE), yf = ek + r2(C::N * C::NE), hf = yf + r2(C::N * C::NFC),
                         ys = hf + r2(C::N * C::NFC), zt = ys + r2(C::N * C::NE), cv = zt, nun = zt + r2(C::N * C::NE), lfi = nun + r2(C::N * C::NE),
                         Ljis = lfi + r2(C::N * C::NLF), ct = Ljis + r2(C::N * (C::SW ? C::NQ * C::NX : C::NH)),   // SW: the dense inverse factor of Hjj, rows strided like K's
                          cl = ct + r2(5 * C::NCI), cc = cl + r2(5 * C::NCI),
                         hee = cc + r2(5 * C::NCI), lsi = hee + r2(C::N * C::NH), Ks = lsi + r2(C::N * C::NLS),
                         // corrector targets of the lane-owned box rows, [slot][lane] (parked here between the corrector's
                         // set-up and its step: 20 registers less to carry through the sweeps)
                         cxr = Ks + r2(C::N * C::NQ * C::NX), rows = cxr + r2((C::SOFT ? 2 : 1) * (2 * C::QX + 2 * C::QU) * C::NT),   // rows: parked (t, lam) of the box rows (SOFT: + sigma, tau, gam; cxr: + the slack pairs' targets)
                         // state-polytopic (collision / projectile) rows of knots 1 .. N-1, at most UPR_QP3_NOMAX per knot: slack,
                         // multiplier, corrector target, affine part d - G xs_q; heew = hee + (1/h) sum_r w_r G_r G_r' (what the
                         // matrix sweep and the costates use in place of hee when such rows exist)
                         ot = rows + r2((C::SOFT ? 10 : 4) * (C::QX + C::QU) * C::NT), ol = ot + (C::N - 1) * UPR_QP3_NOMAX, oc = ol + (C::N - 1) * UPR_QP3_NOMAX,
                         od0 = oc + (C::N - 1) * UPR_QP3_NOMAX, heew = od0 + (C::N - 1) * UPR_QP3_NOMAX,
                         // SOFT: slack, its own barrier pair and the pair's corrector target of the friction-pyramid rows (sfr: sigma,
                         // + 1: tau, + 2: gam, + 3: target; 5 NCI each) and of the state-polytopic rows (sor, the same; (N - 1) NOMAX each)
                         // when slacks.poly_ineq softens them (pybindings.cpp:160-181)
                         sfr = heew + r2(C::N * C::NH), sor = sfr + (C::SOFT ? 4 * r2(5 * C::NCI) : 0),
                         total = sor + ((C::SOFT && C::ROWS) ? 4 * (C::N - 1) * UPR_QP3_NOMAX : 0);
    static constexpr int sfs = r2(5 * C::NCI), sos = (C::N - 1) * UPR_QP3_NOMAX;   // strides between the four arrays
};

// LDS layout (doubles), all compile-time.  KFAR instantiations: the whole-horizon arrays (Z .. dek, Pbs .. gee, prep's staging of
// Z = Lf^-1 Df' and the costates of the full step) are offsets into the far array behind LK instead, the rest is the LDS
template <class C>
struct upr_qp3_lds {
    static constexpr int r2(int n) { return (n + 1) & ~1; }
    static constexpr bool KF = C::KFAR;
    static constexpr int Z = 0, S = Z + r2(C::NZ), gxs = S + r2(C::NZ), wx = gxs + r2(C::N1 * C::NX), gus = wx + r2(C::N1 * C::NX),
                         wu = gus + r2(C::N * C::NU), cs = wu + r2(C::N * C::NU), hf = cs + r2(C::N * C::NX), ys = hf + r2(C::BIGF ? 0 : C::N * C::NFC),
                         zt = ys + r2(C::N * C::NE), cv = zt, ek = zt + r2(C::N * C::NE),
                         dek = ek + r2(C::BIGF ? 0 : C::N * C::NE),   // INCEK: C sx of the step that was taken (the equality residual follows the iterate without a pass over C)
                         kend = dek + r2(C::INCEK ? C::N * C::NE : 0),
                         // constants
                         xlb = KF ? 0 : kend, xub = xlb + r2(C::NX), ulb = xub + r2(C::NX), uub = ulb + r2(C::NU), qd = uub + r2(C::NU),
                         rd = qd + r2(C::NX), xd = rd + r2(C::NU), erow = xd + r2(C::NX), df = erow + r2(3 * (C::NP > 0 ? C::NP : 1)),
                         // working set of the sweeps
                         Pa = df + r2(C::BIGF ? 6 * C::NFC : C::NE * C::NFC), Pb = Pa + r2(C::NX * C::NX), hux = Pb + r2(C::NX * C::NX),
                         // V = Lj^-1 Hux and the packed factor Lj of the knot in work and of the previous one (double buffered:
                         // K = Lj^-T V of the previous knot is formed by an idle wave while wave 0 factors the next)
                         vm = hux + r2(C::NQ * C::NX), lk = vm + 2 * r2(C::NQ * C::NX),
                         hjj = lk + 2 * r2(C::NH), vc = hjj + r2(C::NQ * C::NQ), ck = vc + r2(C::NE * C::NX),
                         // two-wave sweep (C::SW): Vc = Lsi C of knots 1 .. N-1, [col][ne] each -- the first KS of them in the step array
                         // (dead between prep and the vector sweep), the rest from Pa on -- then the sweep's staging: packed Hjj,
                         // Hux / V by columns [nx][HXS], the partial sums of P+ b [nx][HXS]
                         VCS = (C::NE * 2 % 64 == 32 || C::NE * 2 % 64 == 0) ? C::NE + 2 : C::NE /* column stride of Vc: 48 rows = 96 dwords would put every column on the same two bank groups */,
                         VCN = VCS * C::NX, KS = (r2(C::NZ) / VCN < C::N - 1) ? r2(C::NZ) / VCN : C::N - 1, HXS = (C::NQ + 1) & ~1,
                         sw0 = C::VCPRE ? Pa + (C::N - 1 - KS) * VCN : Pa, sw_vcb = sw0 /* (!VCPRE: Vc of the knot in work and of the next one) */,
                         sw_hj = C::VCPRE ? sw0 : sw0 + 2 * VCN, sw_hx = sw_hj + r2(C::NH), sw_pb = sw_hx + (C::NX + C::NQ) * HXS /* (rows nx ..: the identity) */, sw_w = sw_pb + C::NX * HXS /* (the costate of the fused predictor sweep) */,
                         sw_pq = sw_w + r2(C::NX) /* (!VCPRE: the shares of Vc'Vc of waves 2 and 3, [wave][entry][lane]) */,
                         sw_end = sw_pq + (C::VCPRE ? 0 : 2 * 9 * 64),
                         // the scratch region Pa .. yN serves, at different times: the four-wave sweep's working set (Pa .. ck), the
                         // two-wave sweep's staging (sw0 .. sw_end), prep's staging (Z = Lf^-1 Df' at Pa, the Schur complements of
                         // all knots and the state-polytopic rows' (s, w) at hux) and the costates (Pa): sized for the largest of them
                         scr_sweep = ck + r2(C::NE * C::NX), scr_sw = C::SW ? ((KF || sw_end > sw0 + C::NKB * (C::SB * (C::SB + 1) / 2)) ? sw_end : sw0 + r2(C::NKB * (C::SB * (C::SB + 1) / 2))) : 0,
                         // prep's staging of the one-body shapes: Z = Lf^-1 Df' of all knots from Pa, the Schur complements behind it -- at
                         // hux where Z ends before it (nine joints), right behind Z otherwise (six joints with friction: P is 18 x 18)
                         sst = (C::MULTI || C::COUPLED) ? hux : (Pa + r2(C::N * C::NE * C::NFC) > hux ? Pa + r2(C::N * C::NE * C::NFC) : hux),
                         scr_prep = sst + r2((C::MULTI || C::COUPLED) ? 0 : C::N * C::NE * C::NE), scr_rows = hux + (C::ROWS ? 2 * (C::N - 1) * UPR_QP3_NOMAX : 0),
                         scr_a = scr_sweep > scr_sw ? scr_sweep : scr_sw, scr_b = scr_prep > scr_rows ? scr_prep : scr_rows,
                         yN = scr_a > scr_b ? scr_a : scr_b, dyN = yN + r2(C::NEN),
                         eN = dyN + r2(C::NEN), jN = eN + r2(C::NEN), red = jN + r2(3 * C::NQ), RSET = (4 * (C::NT / 64) > 16 ? 4 * (C::NT / 64) : 16) /* two sets of reduction slots */, misc = red + 2 * RSET,
                         // LDS-resident per-knot vectors of the sweeps: P+ b, feed-forward kff = Hjj^-1 huj, dynamics residual
                         prf = misc + 16, lsik = prf + 4 * 16,   /* prf: cycle counters, 16 phases x the first 4 waves */ heek = lsik + r2(C::NLS), lend = heek + 2 * r2(C::NH), Pbs = KF ? kend : lend, kffs = Pbs + r2(C::N * C::NX),
                         bks = kffs + r2(C::N * C::NQ), gee = bks + r2(C::N * C::NX),   // gee: end-effector part of the cost gradient
                         // multi-body shapes: the contacts that load each body (indices as doubles) and their number
                         kend2 = gee + r2(C::N * C::NQ), clist = KF ? lend : kend2, ccnt = clist + r2(C::NB > 1 ? C::NB * C::NC : 0),
                         // stacked bodies (COUPLED): the dense Schur complement of every knot, packed lower triangle, assembled block by block
                         // by a lane per (knot, body pair) and factored by the knot's lane (prep D)
                         sdn = ccnt + r2(C::NB > 1 ? C::NB : 0),
                         // ... and, per body pair (bi >= bj), the contacts that load BOTH bodies (indices as doubles) and their number
                         plist = sdn + r2(C::COUPLED ? C::N * (C::SB * (C::SB + 1) / 2) : 0), pcnt = plist + r2(C::COUPLED ? (C::NB * (C::NB + 1) / 2) * C::NC : 0),
                         total = pcnt + r2(C::COUPLED ? C::NB * (C::NB + 1) / 2 : 0),
                         // prep's staging of Z = Lf^-1 Df' [knot][force][6] (star arrangements) and the costates of the full step: in the
                         // sweeps' working set -- KFAR: whole-horizon arrays of their own
                         zst = KF ? kend2 : Pa, pin = KF ? zst + r2(C::N * C::NFC * 6) : Pa, ktotal = KF ? pin + r2(C::N1 * C::NX) : 0;
};

// global workspace per instance (doubles).  dx / du sit where the line-search kernel expects them.
template <class C>
struct upr_qp3_ws {
    static constexpr int dx = 0, du = C::N1 * C::NX;
    static constexpr int a0 = (C::NZ + 1) & ~1;
    static constexpr int pi = a0, pin = pi + C::N1 * C::NX + (C::N1 * C::NX & 1), nu = pin + C::N1 * C::NX + (C::N1 * C::NX & 1),
                         store = nu + ((C::N * C::NE + 1) & ~1), far = store /* the old per-knot store is gone */, kfar = far + upr_qp3_far<C>::total /* KFAR: the whole-horizon arrays (upr_qp3_lds::ktotal doubles) */, total = ((kfar + upr_qp3_lds<C>::ktotal + 15) & ~15);
};

// knots of register prefetch for the rows / columns of K in the vector and forward sweeps (a step takes 0.7 - 1.1 k
// cycles, an L2 miss 2 - 3.5 k under load)
#ifndef UPR_QP3_KSTAGES
#define UPR_QP3_KSTAGES 6
#endif
// issue priority of the calling wave (0 .. 3): the wave that carries a serial recursion is raised above the co-resident
// workgroup's waves on its SIMD for the duration (UPR_QP3_PRIO=0 at compile time switches it off for A/B runs)
#ifndef UPR_QP3_PRIO
#define UPR_QP3_PRIO 1
#endif
// (Compile-time knobs that measured worse and are gone, with their tables in DESIGN.md section 7: the side work of the factoring wave on
//  waves 2 / 3 (UPR_QP3_OFFLOAD 1 | 2), the feed-forward phase a row per lane (KFF_ROWS), the block wave on another physical wave
//  (PWAVE), raised priorities in the serial feed-forward phase and in the flat phases (PRIO_SERIAL, PRIO_FLAT), two knots per lane in
//  the cooperative Schur factorisation (SCHUR_KQ), the two-pass predictor average (FUSEAFF 0), the separate residual pass (FUSERES 0),
//  the one-wave and the unfused forms of the matrix sweep (SW2 0, FUSEVEC 0), the 128- and 512-lane instantiations.)
// s_setprio of the factoring wave / of the wave that holds the blocks of P in the two-wave matrix sweep, and of the wave that runs the
// corrector's vector sweep / the forward sweeps (the other waves and phases run at 0)
#ifndef UPR_QP3_PRIO_W0
#define UPR_QP3_PRIO_W0 3
#endif
#ifndef UPR_QP3_PRIO_VEC
#define UPR_QP3_PRIO_VEC 3
#endif
#ifndef UPR_QP3_PRIO_FWD
#define UPR_QP3_PRIO_FWD 3
#endif
#ifndef UPR_QP3_PRIO_W1
// (round 5) 1 for the hard-row kernels: the dispatcher puts wave 1 of one workgroup and wave 0 of its co-resident neighbour on the same
// SIMD (tools/probe/simd_probe.hip) -- the wave every other wave waits for is the factoring wave, and with the block wave one step
// below it the headline launch is 0.6 - 1.1 % shorter in six A/B pairs (tools/exp_flags.py; W1 = 2: -1.0 %, 0: -0.6 %, W0 = 2 with
// W1 = 3: + 2.2 %; the vector / forward sweep waves at 2 or 1: no change).  The SOFT kernels measured 1.1 % slower with it and keep 3.
#define UPR_QP3_PRIO_W1 (C::SOFT ? 3 : 1)
#endif
#ifndef UPR_QP3_OCC1
#define UPR_QP3_OCC1 2   /* workgroups per CU the one-body instantiations are compiled for (register budget 512 / (4 waves x this) per SIMD lane) */
#endif
#ifndef UPR_QP3_PREC_MAX
#define UPR_QP3_PREC_MAX 4   /* quad slots of rows of C per lane up to which prep fetches them into registers ahead of phase C */
#endif
#ifndef UPR_QP3_PREV_MAX
#define UPR_QP3_PREV_MAX 8   /* the same for the forward sweep's tail (fetched during the sweep; 8: box_arch since its kernel has no spills, -1 %) */
#endif
#if defined(UPR_HOST_EMU) || !UPR_QP3_PRIO
#define UPR_SETPRIO(p) ((void)0)
#else
#define UPR_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
template <class C>
#define UPR_FORT(i, n) for (int i = tid(); i < (n); i += stride())

struct upr_qp3 {
    typedef upr_qp3_lds<C> O;
    typedef upr_qp3_ws<C> W;
    typedef upr_qp3_far<C> F;
    static constexpr int NQ = C::NQ, NX = C::NX, NU = C::NU, NE = C::NE, NFC = C::NFC, NC = C::NC, NF = C::NF, N = C::N, N1 = C::N1, NT = C::NT;
#ifndef UPR_HOST_EMU
    // which form of the matrix sweep this instantiation runs: two waves (SW2: every device instantiation) or the four-wave form
    // further down (the host emulation)
    static constexpr bool SW2 = C::SW;
    static_assert(!C::SW || NT >= 128, "the two-wave matrix sweep needs two waves");
#else
    static constexpr bool SW2 = false;
#endif
    // LDS-ordered barriers / wave-level ordering points.  KFAR: the whole-horizon arrays are global memory, so every one of them also
    // waits for the global traffic in flight (a full barrier / fence); all other instantiations: exactly the macros
    UPR_HDI static void sync_lds() { if (C::KFAR) { UPR_SYNC(); } else { UPR_SYNC_LDS(); } }
    UPR_HDI static void wsync_lds() { if (C::KFAR) { UPR_WSYNC(); } else { UPR_WSYNC_LDS(); } }
    upr_ctx ctx;
    int wb;   // first lane of this wave (uniform; lives in a scalar register)
    // Lane / work-item index recomputed from the hardware where it is needed: a work-item id kept in a vector
    // register for the whole solve gets spilled, and every reload from scratch waits for ALL global loads in
    // flight (s_waitcnt vmcnt(0)), which serialises the register prefetches of the sweeps.
    UPR_HDI int lane() const {
#ifdef UPR_HOST_EMU
        return ctx.tid;
#else
        int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z));
        return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
#endif
    }
    UPR_HDI int tid() const {
#ifdef UPR_HOST_EMU
        return ctx.tid;
#else
        return wb + lane();
#endif
    }
    UPR_HDI bool wave0() const {
#ifdef UPR_HOST_EMU
        return ctx.tid < 64;
#else
        return wb == 0;
#endif
    }
    UPR_HDI int stride() const {
#ifdef UPR_HOST_EMU
        return ctx.nt;
#else
        return NT;
#endif
    }
    UPR_HDI upr_ctx ctxv() const { upr_ctx c; c.tid = tid(); c.nt = stride(); return c; }
    const upr_problem* P;
    double* L;
    double* LK;  // the whole-horizon arrays: = L (LDS), or a far array of their own (C::KFAR, long horizons)
    double* G;   // far arrays (global)
    const double* xs; const double* us; const double* x0; const double* lin; const double* Dfg;
    double* ws;
    int lin_stride, lin_g, lin_gx, lin_grad, lin_hess, neN;
    int no, lin_obs, hee_w;   // state-polytopic rows per knot (knots 1 .. N-1), their record offset, and F::hee or F::heew
    UPR_HDI const double* orow(int k, int r) const { return lin + (size_t)k * lin_stride + lin_obs + no + r * NQ; }
    double h, h2, h3, sigma_mu;
    int mode;
    bool fbk;   // the feedback gain of the first knot is wanted (use_feedback_policy)
    // lane-owned box rows: [item][0 = lower, 1 = upper]
    // (KFAR: ONE slot of each class in registers at a time -- the loops over the slots are rolled and fetch / park their slot, ldx .. stu)
    static constexpr int QXR = C::KFAR ? 1 : C::QX, QUR = C::KFAR ? 1 : C::QU;
    double tx[QXR][2], lx[QXR][2];
    double tu[QUR][2], lu[QUR][2];
    // SOFT: slack sigma of a softened box row, its own barrier pair (tau, gam); untouched otherwise
    static constexpr int QXS = C::SOFT ? QXR : 1, QUS = C::SOFT ? QUR : 1;
    double sgx[QXS][2], tax[QXS][2], gax[QXS][2];
    double sgu[QUS][2], tau_[QUS][2], gau[QUS][2];
    bool softx, softu;      // which box classes carry slacks (uniform over the workgroup)
    bool softp;             // slacks on the polytopic rows (friction pyramid, collision / projectile rows): slacks.poly_ineq
    double ZL, ZU, zL, zU;  // L2 / L1 penalties of lower / upper rows
    double rho_s;           // regularisation of the Schur complement: UPR_QP_RHO_S, or 1 / Z for a softened equality
    double rho_eq;          // 1 / Z for a softened equality (the row's residual is C dz + e - rho_eq nu), else 0
    double rho_px;          // proximal treatment of a hard equality the forces cannot span (upr_qp.h UPR_QP_RHO_S_PROX): C dz + e = rho_px (nu+ - nu)
    double soft_stat;       // running max of the slack stationarity |Z sigma + z - lam - gam| (residuals)
    bool have_ek;           // INCEK: ek is current (updated with the step) -- prep does not form it again
    static constexpr int NCT0 = 2 * C::QX + 2 * C::QU;                // corrector targets of the rows per lane (F::cxr)
    static constexpr int NCT = (C::SOFT ? 2 : 1) * NCT0;             // SOFT: + the targets of the slack pairs, behind them
    static constexpr int NCTR = C::KFAR ? 1 : NCT;                   // ... of them in registers (KFAR: none, see ineq_sweep's tgt)
    struct zero_targets_t { double v[NCTR]; };
    UPR_HDI static const double (&zero_targets())[NCTR] { static constexpr zero_targets_t z = {}; return z.v; }
    UPR_HDI void load_targets(double (&ctm)[NCTR]) const {
        if (C::KFAR) { ctm[0] = 1.0; return; }
        const int tid_ = tid();
#pragma unroll
        for (int i = 0; i < NCTR; ++i) ctm[i] = G[F::cxr + i * NT + tid_];
    }

    // the corrector's step on the lane-owned rows in one evaluation (rows_dir / rows_trial / rows_apply further down): which
    // instantiations run it, and what a lane keeps between its three parts
    static constexpr bool ONEPASS = !C::SOFT && !C::ROWS && !(NF == 3 && C::QC > 1);
    static constexpr int OP_QR5 = (NF == 3) ? (5 * C::NCI + NT - 1) / NT : 0;
    static constexpr int OP_NB = 2 * C::QX + 2 * C::QU, OP_NR = OP_NB + (OP_QR5 > 0 ? OP_QR5 : 1);
    struct row_dirs { double dt[OP_NR], dl[OP_NR], ft[OP_QR5 > 0 ? OP_QR5 : 1], fl[OP_QR5 > 0 ? OP_QR5 : 1], fc[OP_QR5 > 0 ? OP_QR5 : 1]; };
    // the far operands of the rows' step: the corrector targets of the box rows and (ONEPASS) the friction rows' (t, lam, target)
    UPR_HDI void prefetch_step(double (&ctm)[NCTR], row_dirs& D) const {
        load_targets(ctm);
        if (ONEPASS) {
            const int tid_ = tid();
#pragma unroll
            for (int q = 0; q < OP_QR5; ++q) {
                const int e = tid_ + q * NT, ec = (e < 5 * C::NCI) ? e : 0;
                D.ft[q] = G[F::ct + ec]; D.fl[q] = G[F::cl + ec]; D.fc[q] = G[F::cc + ec];
            }
        }
    }
    // The box rows live in registers through the flat phases only.  Around the sweeps (which need every register) they
    // are parked in global memory [slot][lane]: written once per iteration after the step, read back behind the two
    // forward sweeps (by the idle waves during the sweep, by wave 0 right after it).
    static constexpr int NROWV = 4 * C::QX + 4 * C::QU;
    // SOFT: ten values per box row pair and slot do not fit the registers two workgroups per CU leave a lane (the compiler kept
    // them in scratch, one exposed round trip per use).  Every flat phase reads them from the parked copy instead, all requests
    // together at its top, and they are dead in between.
    static constexpr bool ROWMEM = C::SOFT && C::NB == 1 && C::QX >= 3 && UPR_QP3_SOFT_ROWMEM;   // (the shapes that spilled; the others measured 5 % slower with it)
    static constexpr int QO = ((C::N - 1) * UPR_QP3_NOMAX + C::NT - 1) / C::NT;   // state-polytopic rows per lane (at most)
    // KFAR: slot q of the state / input box rows of this lane out of / into the parked copy (same layout as store_rows)
    UPR_HDI void ldx(int q) {
        if (!C::KFAR) return;
        const double* R = G + F::rows + tid();
        tx[0][0] = R[(4 * q) * NT]; tx[0][1] = R[(4 * q + 1) * NT]; lx[0][0] = R[(4 * q + 2) * NT]; lx[0][1] = R[(4 * q + 3) * NT];
        if (C::SOFT) { const double* Q = R + NROWV * NT;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { sgx[0][s2] = Q[(6 * q + 3 * s2) * NT]; tax[0][s2] = Q[(6 * q + 3 * s2 + 1) * NT]; gax[0][s2] = Q[(6 * q + 3 * s2 + 2) * NT]; } }
    }
    UPR_HDI void stx(int q) const {
        if (!C::KFAR) return;
        double* R = G + F::rows + tid();
        R[(4 * q) * NT] = tx[0][0]; R[(4 * q + 1) * NT] = tx[0][1]; R[(4 * q + 2) * NT] = lx[0][0]; R[(4 * q + 3) * NT] = lx[0][1];
        if (C::SOFT) { double* Q = R + NROWV * NT;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { Q[(6 * q + 3 * s2) * NT] = sgx[0][s2]; Q[(6 * q + 3 * s2 + 1) * NT] = tax[0][s2]; Q[(6 * q + 3 * s2 + 2) * NT] = gax[0][s2]; } }
    }
    UPR_HDI void ldu(int q) {
        if (!C::KFAR) return;
        const double* R = G + F::rows + tid();
        tu[0][0] = R[(4 * C::QX + 4 * q) * NT]; tu[0][1] = R[(4 * C::QX + 4 * q + 1) * NT]; lu[0][0] = R[(4 * C::QX + 4 * q + 2) * NT]; lu[0][1] = R[(4 * C::QX + 4 * q + 3) * NT];
        if (C::SOFT) { const double* Q = R + NROWV * NT;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { sgu[0][s2] = Q[(6 * C::QX + 6 * q + 3 * s2) * NT]; tau_[0][s2] = Q[(6 * C::QX + 6 * q + 3 * s2 + 1) * NT]; gau[0][s2] = Q[(6 * C::QX + 6 * q + 3 * s2 + 2) * NT]; } }
    }
    UPR_HDI void stu(int q) const {
        if (!C::KFAR) return;
        double* R = G + F::rows + tid();
        R[(4 * C::QX + 4 * q) * NT] = tu[0][0]; R[(4 * C::QX + 4 * q + 1) * NT] = tu[0][1]; R[(4 * C::QX + 4 * q + 2) * NT] = lu[0][0]; R[(4 * C::QX + 4 * q + 3) * NT] = lu[0][1];
        if (C::SOFT) { double* Q = R + NROWV * NT;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { Q[(6 * C::QX + 6 * q + 3 * s2) * NT] = sgu[0][s2]; Q[(6 * C::QX + 6 * q + 3 * s2 + 1) * NT] = tau_[0][s2]; Q[(6 * C::QX + 6 * q + 3 * s2 + 2) * NT] = gau[0][s2]; } }
    }
    UPR_HDI void store_rows() const {
        if (C::KFAR) return;   // (every loop over the slots parks its own slot)
        const int tid_ = tid();
        double* R = G + F::rows + tid_;
        if (C::SOFT) {
            double* Q = R + NROWV * NT;
#pragma unroll
            for (int q = 0; q < QXS; ++q)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) { Q[(6 * q + 3 * s2) * NT] = sgx[q][s2]; Q[(6 * q + 3 * s2 + 1) * NT] = tax[q][s2]; Q[(6 * q + 3 * s2 + 2) * NT] = gax[q][s2]; }
#pragma unroll
            for (int q = 0; q < QUS; ++q)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) { Q[(6 * QXS + 6 * q + 3 * s2) * NT] = sgu[q][s2]; Q[(6 * QXS + 6 * q + 3 * s2 + 1) * NT] = tau_[q][s2]; Q[(6 * QXS + 6 * q + 3 * s2 + 2) * NT] = gau[q][s2]; }
        }
#pragma unroll
        for (int q = 0; q < C::QX; ++q) { R[(4 * q) * NT] = tx[q % QXR][0]; R[(4 * q + 1) * NT] = tx[q % QXR][1]; R[(4 * q + 2) * NT] = lx[q % QXR][0]; R[(4 * q + 3) * NT] = lx[q % QXR][1]; }
#pragma unroll
        for (int q = 0; q < C::QU; ++q) { R[(4 * C::QX + 4 * q) * NT] = tu[q % QUR][0]; R[(4 * C::QX + 4 * q + 1) * NT] = tu[q % QUR][1]; R[(4 * C::QX + 4 * q + 2) * NT] = lu[q % QUR][0]; R[(4 * C::QX + 4 * q + 3) * NT] = lu[q % QUR][1]; }
    }
    UPR_HDI void load_rows() {
        if (C::KFAR) return;
        const int tid_ = tid();
        const double* R = G + F::rows + tid_;
        if (C::SOFT) {
            const double* Q = R + NROWV * NT;
#pragma unroll
            for (int q = 0; q < QXS; ++q)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) { sgx[q][s2] = Q[(6 * q + 3 * s2) * NT]; tax[q][s2] = Q[(6 * q + 3 * s2 + 1) * NT]; gax[q][s2] = Q[(6 * q + 3 * s2 + 2) * NT]; }
#pragma unroll
            for (int q = 0; q < QUS; ++q)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) { sgu[q][s2] = Q[(6 * QXS + 6 * q + 3 * s2) * NT]; tau_[q][s2] = Q[(6 * QXS + 6 * q + 3 * s2 + 1) * NT]; gau[q][s2] = Q[(6 * QXS + 6 * q + 3 * s2 + 2) * NT]; }
        }
#pragma unroll
        for (int q = 0; q < C::QX; ++q) { tx[q % QXR][0] = R[(4 * q) * NT]; tx[q % QXR][1] = R[(4 * q + 1) * NT]; lx[q % QXR][0] = R[(4 * q + 2) * NT]; lx[q % QXR][1] = R[(4 * q + 3) * NT]; }
#pragma unroll
        for (int q = 0; q < C::QU; ++q) { tu[q % QUR][0] = R[(4 * C::QX + 4 * q) * NT]; tu[q % QUR][1] = R[(4 * C::QX + 4 * q + 1) * NT]; lu[q % QUR][0] = R[(4 * C::QX + 4 * q + 2) * NT]; lu[q % QUR][1] = R[(4 * C::QX + 4 * q + 3) * NT]; }
    }
    UPR_HDI const double* rec(int k) const { return lin + (size_t)k * lin_stride; }
    UPR_HDI double* Zx(int k) const { return LK + O::Z + k * NX; }
    UPR_HDI double* Zu(int k) const { return LK + O::Z + N1 * NX + k * NU; }
    UPR_HDI double* Sx(int k) const { return LK + O::S + k * NX; }
    UPR_HDI double* Su(int k) const { return LK + O::S + N1 * NX + k * NU; }
    // entries of A = [[1,h,h2],[0,1,h],[0,0,1]] and B = [h3;h2;h] in arithmetic form: a select between the members
    // h, h2, h3 on a lane-dependent index would be turned into an indexed load and pin the whole object in scratch
    UPR_HDI double coefA(int a, int b) const { return ((a == b) ? 1.0 : 0.0) + (((a == 0 && b == 1) || (a == 1 && b == 2)) ? 1.0 : 0.0) * h + ((a == 0 && b == 2) ? 1.0 : 0.0) * h2; }
    UPR_HDI double coefB(int a) const { return ((a == 0) ? 1.0 : 0.0) * h3 + ((a == 1) ? 1.0 : 0.0) * h2 + ((a == 2) ? 1.0 : 0.0) * h; }
    // entry (r, col) of Df (BIGF: compact, valid for the rows of the body the column's contact loads); hf and ek (BIGF: far arrays)
    UPR_HDI double DF(int r, int col) const { return C::BIGF ? L[O::df + col * 6 + (r - 6 * (r / 6))] : L[O::df + r * NFC + col]; }
    UPR_HDI double* hfp() const { return C::BIGF ? G + F::hf : LK + O::hf; }
    UPR_HDI double* ekp() const { return C::BIGF ? G + F::ek : LK + O::ek; }
    // row r of Df times a force-indexed vector.  Multi-body shapes: the row of body r / 6 has entries at the contacts that
    // load that body only (O::clist), four of 32 columns for the robust arrangement
    UPR_HDI double df_dot(int r, const double* vf) const {
        double v = 0.0;
        if (C::NB == 1) { for (int i = 0; i < NFC; ++i) v += L[O::df + r * NFC + i] * vf[i]; }
        else {
            const int bb = r / 6, n = (int)L[O::ccnt + bb];
            for (int j = 0; j < n; ++j) {
                const int ci = (int)L[O::clist + bb * NC + j];
#pragma unroll
                for (int a = 0; a < NF; ++a) v += DF(r, NF * ci + a) * vf[NF * ci + a];
            }
        }
        return v;
    }

    // workgroup reductions.  Inside a wave: four DPP steps (neighbours, pairs, the two quads of a half row, the two halves of a row of
    // sixteen) leave every row's result in all its lanes, the four rows meet through scalar registers (v_readlane) -- no LDS
    // crossbar: the ds_bpermute butterfly of rounds 1 - 5 was twelve dependent permutes, ~1.2 k cycles of the ~1.7 k a reduction
    // cost.  Across the waves: one LDS slot per wave and ONE LDS-only barrier; consecutive reductions alternate between two sets of
    // slots (a wave can only reach the reduction after next through the next one's barrier, behind which nobody reads this one's
    // slots any more).  All lanes must be active.
    int red_par;   // which set of slots the next reduction uses (uniform)
    UPR_HDI static double comb(double a, double b, int op) { return (op == 0) ? a + b : (op == 1 ? (a > b ? a : b) : (a < b ? a : b)); }
#ifndef UPR_HOST_EMU
    UPR_HDI static double wave_reduce(double v, int op) {
        v = comb(v, upr_dpp_quad<0xB1>(v), op);    // quad_perm [1 0 3 2]
        v = comb(v, upr_dpp_quad<0x4E>(v), op);    // quad_perm [2 3 0 1]
        v = comb(v, upr_dpp_quad<0x141>(v), op);   // row_half_mirror
        v = comb(v, upr_dpp_quad<0x140>(v), op);   // row_mirror
        return comb(comb(upr_readlane(v, 0), upr_readlane(v, 16), op), comb(upr_readlane(v, 32), upr_readlane(v, 48), op), op);
    }
#endif
    UPR_HDI double reduce(double v, int op /*0 sum 1 max 2 min*/) {
#ifdef UPR_HOST_EMU
        return upr_reduce(ctx, L + O::red, v, op);
#else
        v = wave_reduce(v, op);
        double* slot = L + O::red + red_par * O::RSET;
        red_par ^= 1;
        if (lane() == 0) slot[wb >> 6] = v;
        sync_lds();
        double r = slot[0];
#pragma unroll
        for (int w = 1; w < (NT >> 6); ++w) r = comb(r, slot[w], op);
        return r;
#endif
    }
    // three maxima and one sum at once
    UPR_HDI void reduce4(double* v) {
#ifdef UPR_HOST_EMU
        v[0] = upr_reduce(ctx, L + O::red, v[0], 1); v[1] = upr_reduce(ctx, L + O::red, v[1], 1);
        v[2] = upr_reduce(ctx, L + O::red, v[2], 1); v[3] = upr_reduce(ctx, L + O::red, v[3], 0);
#else
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = wave_reduce(v[q], q < 3 ? 1 : 0);
        double* slot = L + O::red + red_par * O::RSET;
        red_par ^= 1;
        if (lane() == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) slot[4 * (wb >> 6) + q] = v[q];
        }
        sync_lds();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double r = slot[q];
#pragma unroll
            for (int w = 1; w < (NT >> 6); ++w) r = comb(r, slot[4 * w + q], q < 3 ? 1 : 0);
            v[q] = r;
        }
#endif
    }

    // one inequality row: slack residual, weight and reduced-gradient multiplier for the current mode
    UPR_HDI void row(double c, double ds, double t, double lam, double& cterm, double& s, double& w) const {
        const double rp = c - t;
        const double rt = upr_rcp(t);
        w = lam * rt;
        if (mode == 0) s = w * rp;
        else if (mode == 2) s = -lam;
        else {
            if (mode == 1) { const double dta = ds + rp; const double dla = -lam - w * dta; cterm = dta * dla - sigma_mu; }
            s = (lam * t + cterm + lam * rp) * rt - lam;
        }
    }

    // The same for a softened row c + sigma >= 0, sigma >= 0 with cost 1/2 Z sigma^2 + z sigma (HPIPM slack; the
    // arithmetic of upr_soft_terms, upr_qp.h): the slack step is eliminated from the Newton system, which leaves the
    // effective weight w (Z + w_s) / D and the gradient multiplier (rc + lam rp) / t - w a / D - lam.  cterm / cterm_s
    // carry rc - lam t and rc_s - gam tau (0 in the predictor).
    UPR_HDI void row_soft(double c, double ds, double t, double lam, double sig, double tau, double gam, double Zp, double zp,
                          double& cterm, double& cterm_s, double& s, double& w) const {
        const double rp = c + sig - t, rps = sig - tau;
        const double rt = upr_rcp(t), rtau = upr_rcp(tau);
        const double w0 = lam * rt, ws_ = gam * rtau;
        if (mode == 2) { s = -lam; w = w0; return; }
        const double D = Zp + w0 + ws_, rD = upr_rcp(D);
        const double base = (Zp * sig + zp - lam - gam);
        if (mode == 0) { cterm = 0.0; cterm_s = 0.0; }
        else if (mode == 1) {
            const double a0 = base + (lam * t + lam * rp) * rt + (gam * tau + gam * rps) * rtau;
            const double dsa = -(a0 + w0 * ds) * rD;
            const double dta = ds + rp + dsa, dtaua = dsa + rps;
            const double dla = -lam - w0 * dta, dga = -gam - ws_ * dtaua;
            cterm = dta * dla - sigma_mu; cterm_s = dtaua * dga - sigma_mu;
        }
        const double g0 = (lam * t + cterm + lam * rp) * rt;
        const double a = base + g0 + (gam * tau + cterm_s + gam * rps) * rtau;
        s = g0 - w0 * a * rD - lam;
        w = w0 * (Zp + ws_) * rD;
    }

    // ---- flat phases: per-knot vectors and factors out of the lane-owned rows -------------------------------
    // level 0: reduced gradients + residuals only (KKT check); 1: + back-substitution vectors; 2: + factors
    UPR_HDI void prep(int level) {
        const bool factor = level >= 2;
        // the corrector (level 1) follows the predictor (level 2) at the SAME iterate: what depends on the iterate alone
        // (end-effector gradient, dynamics and equality residuals) is kept, only the barrier terms are rebuilt
        const bool fresh = level != 1;
        const int tid_ = tid();
        // global operands of the later phases, requested here and consumed behind phase A: the end-effector gradient /
        // Hessian rows (g0, hee), the friction rows' (t, lam) and -- on the device, a quad per row -- the rows of C
        constexpr int QG = (N * NQ + NT - 1) / NT;
        double heeP[QG][NQ + 1];
        if (fresh) {
#pragma unroll
            for (int q = 0; q < QG; ++q) {
                const int e = tid_ + q * NT;
                if (e >= NQ && e < N * NQ) {
                    const int k = e / NQ, i = e % NQ;
                    heeP[q][NQ] = G[F::g0 + e];
#pragma unroll
                    for (int j = 0; j < NQ; ++j) heeP[q][j] = G[F::hee + k * C::NH + upr_tri(NQ, i, j)];
                }
            }
        }
        double ctP[C::QC][5], clP[C::QC][5];
        // (round 6) the corrector's phase B multiplies with the inverse contact factor the predictor's call stored: requested here with
        // the rows' operands instead of at its use (with the feed-forward phase's Lj^-1, backward_vec: -0.6 % on the headline launch;
        // the same for the corrector's inverse SCHUR factor of phase D measured neutral to slower and is not done)
        constexpr bool PREF_B = NF == 3 && C::QC == 1;
        double BvP[PREF_B ? C::QC : 1][9];
        constexpr int QCS_ = C::SOFT ? C::QC : 1;
        double csP[QCS_][5], caP[QCS_][5], cgP[QCS_][5];   // softened friction rows: slack, its barrier pair
        if (NF == 3) {
#pragma unroll
            for (int q = 0; q < C::QC; ++q) {
                const int ic = tid_ + q * NT;
                if (ic < C::NCI) {
#pragma unroll
                    for (int r = 0; r < 5; ++r) { ctP[q][r] = G[F::ct + 5 * ic + r]; clP[q][r] = G[F::cl + 5 * ic + r]; }
                    if (PREF_B && level == 1) {   // (the corrector's call multiplies with the inverse contact factor the predictor's call stored)
#pragma unroll
                        for (int a = 0; a < 9; ++a) BvP[q % (PREF_B ? C::QC : 1)][a] = G[F::lfi + (ic / NC) * C::NLF + 9 * (ic % NC) + a];
                    }
                    if (C::SOFT && softp) {
#pragma unroll
                        for (int r = 0; r < 5; ++r) { csP[q % QCS_][r] = G[F::sfr + 5 * ic + r]; caP[q % QCS_][r] = G[F::sfr + F::sfs + 5 * ic + r]; cgP[q % QCS_][r] = G[F::sfr + 2 * F::sfs + 5 * ic + r]; }
                    }
                }
            }
        }
#ifndef UPR_HOST_EMU
        constexpr int CH = (NX + 3) / 4, QR = (N * NE * 4 + NT - 1) / NT;
        // (multi-body shapes have too many rows of C for a register prefetch: they read them where they are used)
        constexpr bool PRE_C = QR <= UPR_QP3_PREC_MAX;
        double ckr[PRE_C ? QR : 1][CH], e0r[PRE_C ? QR : 1];
        // (ROWMEM shapes: requested behind the box rows of phase A, which need the registers)
        auto fetch_c = [&]() {
            if (PRE_C && fresh) {
#pragma unroll
                for (int q = 0; q < QR; ++q) {
                    const int e4 = tid_ + q * NT;
                    if (e4 < N * NE * 4) {
                        const int e = e4 >> 2, part = e4 & 3;
                        const double* Ck = rec(e / NE) + lin_gx + (e % NE) * NX + part * CH;
#pragma unroll
                        for (int c = 0; c < CH; ++c) ckr[q % (PRE_C ? QR : 1)][c] = (part * CH + c < NX) ? Ck[c] : 0.0;
                        e0r[q % (PRE_C ? QR : 1)] = G[F::e0 + e];
                    }
                }
            }
        };
        if (!ROWMEM) fetch_c();
#endif
        // state-polytopic rows d + G (q - q_lin) >= 0 of knots 1 .. N-1 (collision pairs, projectile path): multiplier s and
        // weight w of every row, staged in the LDS the sweeps use for Hux / V until the gradient and Hessian passes below
        if (no > 0) {
            // (the rows of a lane requested together: their Jacobian rows and far-array entries come out of global memory)
            constexpr int QOS = (C::SOFT && C::ROWS) ? QO : 1;
            double gq[QO][NQ], od[QO], otv[QO], olv[QO], osv[QOS][3];
#pragma unroll
            for (int q = 0; q < QO; ++q) {
                const int e = tid_ + q * NT, ec = (e < (N - 1) * no) ? e : 0;
                const int k = 1 + ec / no, r = ec % no, ei = (k - 1) * UPR_QP3_NOMAX + r;
                const double* g = orow(k, r);
#pragma unroll
                for (int i = 0; i < NQ; ++i) gq[q][i] = g[i];
                od[q] = G[F::od0 + ei]; otv[q] = G[F::ot + ei]; olv[q] = G[F::ol + ei];
                if (C::SOFT && C::ROWS && softp) {
#pragma unroll
                    for (int u = 0; u < 3; ++u) osv[q % QOS][u] = G[F::sor + u * F::sos + ei];
                }
            }
#pragma unroll
            for (int q = 0; q < QO; ++q) {
                const int e = tid_ + q * NT;
                if (e >= (N - 1) * no) continue;
                const int k = 1 + e / no, r = e % no, ei = (k - 1) * UPR_QP3_NOMAX + r;
                double c = od[q], ds = 0.0;
#pragma unroll
                for (int i = 0; i < NQ; ++i) { c += gq[q][i] * LK[O::Z + k * NX + i]; ds += gq[q][i] * LK[O::S + k * NX + i]; }
                double sr, wr, ct_ = 0.0;
                if (C::SOFT && C::ROWS && softp) {
                    double cts_ = 0.0;
                    row_soft(c, ds, otv[q], olv[q], osv[q % QOS][0], osv[q % QOS][1], osv[q % QOS][2], ZL, zL, ct_, cts_, sr, wr);
                    if (mode == 1) G[F::sor + 3 * F::sos + ei] = cts_;
                } else row(c, ds, otv[q], olv[q], ct_, sr, wr);
                if (mode == 1) G[F::oc + ei] = ct_;
                L[O::hux + ei] = sr; L[O::hux + (N - 1) * UPR_QP3_NOMAX + ei] = wr;
            }
        }
        ftoc(6, 2);
        if (ROWMEM) load_rows();
        // A: box rows (registers)
#pragma unroll (C::KFAR ? 1 : C::QX)
        for (int q = 0; q < C::QX; ++q) {
            const int ix = tid_ + q * NT;
            if (ix < C::NXI) {
                const int zo = NX + ix, k = 1 + ix / NX, i = ix % NX;
                ldx(q);
                const double X = LK[O::Z + zo], dS = LK[O::S + zo];
                double s0, s1, w0, w1;
                double c0 = 0.0, c1 = 0.0;
                if (C::SOFT && softx) {
                    double d0 = 0.0, d1 = 0.0;
                    row_soft(X - L[O::xlb + i], dS, tx[q % QXR][0], lx[q % QXR][0], sgx[q % QXS][0], tax[q % QXS][0], gax[q % QXS][0], ZL, zL, c0, d0, s0, w0);
                    row_soft(L[O::xub + i] - X, -dS, tx[q % QXR][1], lx[q % QXR][1], sgx[q % QXS][1], tax[q % QXS][1], gax[q % QXS][1], ZU, zU, c1, d1, s1, w1);
                    if (mode == 1) { G[F::cxr + (NCT0 + 2 * q) * NT + tid_] = d0; G[F::cxr + (NCT0 + 2 * q + 1) * NT + tid_] = d1; }
                } else {
                    row(X - L[O::xlb + i], dS, tx[q % QXR][0], lx[q % QXR][0], c0, s0, w0);
                    row(L[O::xub + i] - X, -dS, tx[q % QXR][1], lx[q % QXR][1], c1, s1, w1);
                }
                if (mode == 1) { G[F::cxr + (2 * q) * NT + tid_] = c0; G[F::cxr + (2 * q + 1) * NT + tid_] = c1; }
                const double g = (k < N) ? h * L[O::qd + i] * (X - L[O::xd + i]) : 0.0;   // (the end-effector part is added behind the barrier)
                LK[O::gxs + zo] = g + s0 - s1;
                LK[O::wx + zo] = w0 + w1;
            }
        }
        ftoc(7, 2);
#pragma unroll (C::KFAR ? 1 : C::QU)
        for (int q = 0; q < C::QU; ++q) {
            const int iu = tid() + q * NT;
            if (iu < C::NUI) {
                const int i = iu % NU;
                ldu(q);
                const double U = LK[O::Z + N1 * NX + iu], dS = LK[O::S + N1 * NX + iu];
                double s0, s1, w0, w1;
                double c0 = 0.0, c1 = 0.0;
                if (C::SOFT && softu) {
                    double d0 = 0.0, d1 = 0.0;
                    row_soft(U - L[O::ulb + i], dS, tu[q % QUR][0], lu[q % QUR][0], sgu[q % QUS][0], tau_[q % QUS][0], gau[q % QUS][0], ZL, zL, c0, d0, s0, w0);
                    row_soft(L[O::uub + i] - U, -dS, tu[q % QUR][1], lu[q % QUR][1], sgu[q % QUS][1], tau_[q % QUS][1], gau[q % QUS][1], ZU, zU, c1, d1, s1, w1);
                    if (mode == 1) { G[F::cxr + (NCT0 + 2 * C::QX + 2 * q) * NT + tid_] = d0; G[F::cxr + (NCT0 + 2 * C::QX + 2 * q + 1) * NT + tid_] = d1; }
                } else {
                    row(U - L[O::ulb + i], dS, tu[q % QUR][0], lu[q % QUR][0], c0, s0, w0);
                    row(L[O::uub + i] - U, -dS, tu[q % QUR][1], lu[q % QUR][1], c1, s1, w1);
                }
                if (mode == 1) { G[F::cxr + (2 * C::QX + 2 * q) * NT + tid_] = c0; G[F::cxr + (2 * C::QX + 2 * q + 1) * NT + tid_] = c1; }
                LK[O::gus + iu] = h * L[O::rd + i] * U + s0 - s1;
                LK[O::wu + iu] = w0 + w1;
            }
        }
        ftoc(8, 2);
#ifndef UPR_HOST_EMU
        if (ROWMEM) fetch_c();
#endif
        // dynamics residual of every knot in absolute variables (multiple-shooting defect of the iterate)
        if (fresh) UPR_FORT(e, N * NQ) {
            const int k = e / NQ, j = e % NQ;
            const double* X = Zx(k); const double* Xn = Zx(k + 1); const double* U = Zu(k);
            const double q = X[j], v = X[NQ + j], a = X[2 * NQ + j], u = U[j];
            LK[O::bks + k * NX + j] = q + h * v + h2 * a + h3 * u - Xn[j];
            LK[O::bks + k * NX + NQ + j] = v + h * a + h2 * u - Xn[NQ + j];
            LK[O::bks + k * NX + 2 * NQ + j] = a + h * u - Xn[2 * NQ + j];
        }
        ftoc(9, 2);
        UPR_SYNC(); toc(1);
        // end-effector part of the state gradient (knots 1 .. N-1), kept in gee for the corrector's pass
#pragma unroll
        for (int q = 0; q < QG; ++q) {
            const int e = tid_ + q * NT;
            if (e >= NQ && e < N * NQ) {
                const int k = e / NQ, i = e % NQ;
                double a;
                if (fresh) {
                    a = heeP[q][NQ];
#pragma unroll
                    for (int j = 0; j < NQ; ++j) a += heeP[q][j] * LK[O::Z + k * NX + j];
                    LK[O::gee + e] = a;
                } else a = LK[O::gee + e];
                double go = 0.0;   // G' s of the state-polytopic rows (five rows requested together: one exposed latency per group)
                for (int r0 = 0; r0 < no; r0 += 5) {
                    double gv[5], sv[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) { const int r = r0 + u, rc = (r < no) ? r : no - 1; gv[u] = orow(k, rc)[i]; sv[u] = (r < no) ? L[O::hux + (k - 1) * UPR_QP3_NOMAX + rc] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 5; ++u) go += gv[u] * sv[u];
                }
                LK[O::gxs + k * NX + i] += h * a + go;
            }
        }
        if (no > 0 && factor) UPR_FORT(e, (N - 1) * C::NH) {   // barrier Hessian of those rows next to the end-effector Hessian
            const int k = 1 + e / C::NH, t = e % C::NH;
            int i = 0, rem = t;
            while (rem >= NQ - i) { rem -= NQ - i; ++i; }
            const int j = i + rem;   // upr_tri(NQ, i, j) == t for i <= j
            double v = 0.0;
            const double hee_kt = G[F::hee + k * C::NH + t];
            for (int r0 = 0; r0 < no; r0 += 5) {
                double gi[5], gj[5], wv[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int r = r0 + u, rc = (r < no) ? r : no - 1;
                    const double* g = orow(k, rc);
                    gi[u] = g[i]; gj[u] = g[j];
                    wv[u] = (r < no) ? L[O::hux + (N - 1) * UPR_QP3_NOMAX + (k - 1) * UPR_QP3_NOMAX + rc] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) v += wv[u] * gi[u] * gj[u];
            }
            G[F::heew + k * C::NH + t] = hee_kt + v / h;
        }
        // B: contacts -- friction rows, contact block and its factor, force part of the back-substitution
        ftoc(6, 6);
        if ((C::MULTI || O::sst > O::hux) && no > 0) UPR_SYNC();   // (Z of these shapes extends over the LDS the row multipliers above were staged in)
#pragma unroll
        for (int q = 0; q < C::QC; ++q) {
            const int ic = tid() + q * NT;
            if (ic < C::NCI) {
                const int k = ic / NC, ci = ic % NC;
                if (NF == 3) {
                    const int uo = k * NU + NQ + 3 * ci;
                    const double* f = LK + O::Z + N1 * NX + uo; const double* sf = LK + O::S + N1 * NX + uo;
                    double guf[3] = {LK[O::gus + uo], LK[O::gus + uo + 1], LK[O::gus + uo + 2]};
                    double Hc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int a = 0; a < 3; ++a) Hc[4 * a] = h * L[O::rd + NQ + 3 * ci + a] + LK[O::wu + uo + a];
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        const double* e3 = L + O::erow + 3 * (5 * ci + r);
                        const double c = e3[0] * f[0] + e3[1] * f[1] + e3[2] * f[2];
                        const double ds = e3[0] * sf[0] + e3[1] * sf[1] + e3[2] * sf[2];
                        double s, wgt, ct_ = 0.0;
                        if (C::SOFT && softp) {
                            double cts_ = 0.0;
                            row_soft(c, ds, ctP[q][r], clP[q][r], csP[q % QCS_][r], caP[q % QCS_][r], cgP[q % QCS_][r], ZL, zL, ct_, cts_, s, wgt);
                            if (mode == 1) G[F::sfr + 3 * F::sfs + 5 * ic + r] = cts_;
                        } else row(c, ds, ctP[q][r], clP[q][r], ct_, s, wgt);
                        if (mode == 1) G[F::cc + 5 * ic + r] = ct_;
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            guf[a] += e3[a] * s;
#pragma unroll
                            for (int b2 = 0; b2 < 3; ++b2) Hc[3 * a + b2] += wgt * e3[a] * e3[b2];
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) LK[O::gus + uo + a] = guf[a];
                    ftoc(7, 6);   // (-DUPR_QP3_PROF_FLAT=6: prep B -- 6: up to its start, 7: the five rows, 8: contact factor and Z, 9: the two products)
                    if (level == 0) continue;
                    double* Bk = G + F::lfi + k * C::NLF + 9 * ci;
                    // (round 5: the two products below take the inverse factor out of registers where this call has just formed it
                    //  instead of reading it back through Bk; measured neutral on the headline launch -- the compiler had forwarded
                    //  the stores -- and kept for clarity.  Sub-stamps of this phase: tools/r5_prof.sh "-DUPR_QP3_PROF_FLAT=6")
                    double Bv[9];
                    if (!factor) {
#pragma unroll
                        for (int a = 0; a < 9; ++a) Bv[a] = PREF_B ? BvP[q % (PREF_B ? C::QC : 1)][a] : Bk[a];
                    }
                    if (factor) {
                        if (!upr_chol_inv3(Hc)) L[O::misc] = 1.0;
#pragma unroll
                        for (int a = 0; a < 9; ++a) { Bk[a] = Hc[a]; Bv[a] = Hc[a]; }
                        // Z = Lf^-1 Df' of this contact (S = Z'Z + rho I is assembled in phase C); staged where the sweeps keep P
                        if (C::BIGF) {
                            // (Z is formed by the block's lane in phase D out of the factor stored above)
                        } else if (C::MULTI) {
                            // star arrangement: the contact loads one body only, Z keeps that body's six rows, [knot][force][6]
                            const int b2 = P->contact_body2[ci];
#pragma unroll
                            for (int r6 = 0; r6 < 6; ++r6) {
                                const double* dr = L + O::df + (6 * b2 + r6) * NFC + 3 * ci;
                                const double d0 = dr[0], d1 = dr[1], d2 = dr[2];
                                double* zr = LK + O::zst + (k * NFC + 3 * ci) * 6 + r6;
                                zr[0] = Hc[0] * d0; zr[6] = Hc[3] * d0 + Hc[4] * d1; zr[12] = Hc[6] * d0 + Hc[7] * d1 + Hc[8] * d2;
                            }
                        } else if (!C::COUPLED) {
#pragma unroll
                        for (int r = 0; r < NE; ++r) {
                            const double* dr = L + O::df + r * NFC + 3 * ci;
                            const double d0 = dr[0], d1 = dr[1], d2 = dr[2];
                            double* zr = L + O::Pa + (k * NE + r) * NFC + 3 * ci;
                            zr[0] = Hc[0] * d0; zr[1] = Hc[3] * d0 + Hc[4] * d1; zr[2] = Hc[6] * d0 + Hc[7] * d1 + Hc[8] * d2;
                        }
                        }
                    }
                    ftoc(8, 6);
                    double yv[3], hv[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double v = 0.0;
#pragma unroll
                        for (int b2 = 0; b2 < 3; ++b2) if (b2 <= a) v += Bv[3 * a + b2] * guf[b2];
                        yv[a] = v;
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double v = 0.0;
#pragma unroll
                        for (int b2 = 0; b2 < 3; ++b2) if (b2 >= a) v += Bv[3 * b2 + a] * yv[b2];
                        hv[a] = v;
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) { G[F::yf + k * NFC + 3 * ci + a] = yv[a]; hfp()[k * NFC + 3 * ci + a] = hv[a]; }
                    ftoc(9, 6);
                } else {
                    if (level == 0) continue;
                    const int uo = k * NU + NQ + ci;
                    if (factor) {
                        const double lf0 = 1.0 / sqrt(h * L[O::rd + NQ + ci] + LK[O::wu + uo]);
                        G[F::lfi + k * C::NLF + ci] = lf0;
                        if (C::MULTI) { const int b2 = P->contact_body2[ci]; for (int r6 = 0; r6 < 6; ++r6) LK[O::zst + (k * NFC + ci) * 6 + r6] = lf0 * L[O::df + (6 * b2 + r6) * NFC + ci]; }
                        else if (!C::COUPLED) for (int r = 0; r < NE; ++r) L[O::Pa + (k * NE + r) * NFC + ci] = lf0 * L[O::df + r * NFC + ci];
                    }
                    const double lf = G[F::lfi + k * C::NLF + ci];
                    const double yv = lf * LK[O::gus + uo];
                    G[F::yf + k * NFC + ci] = yv; hfp()[k * NFC + ci] = lf * yv;
                }
            }
        }
        UPR_SYNC(); toc(2);
        // C: equality residual ek = e0 + C Zx + Df Zf, ee = ek - Df hf (-> ys slot); S lower triangle (-> lsi slot)
        const bool fresh_ek = fresh && !(C::INCEK && have_ek);   // (INCEK: ek has followed the iterate since the first iteration)
#ifndef UPR_HOST_EMU
        if (fresh_ek) {
            // C Zx by quads (four column chunks of a row, summed by DPP)
            if (PRE_C) {
#pragma unroll
                for (int q = 0; q < QR; ++q) {
                    const int e4 = tid_ + q * NT;
                    const bool act = e4 < N * NE * 4;
                    const int e = act ? (e4 >> 2) : 0, part = e4 & 3;
                    double v = 0.0;
                    {   // (branch-free, as the quads of the forward tail: see there)
                        const double* zx = LK + O::Z + (e / NE) * NX + part * CH;
                        double zv[CH];
#pragma unroll
                        for (int c = 0; c < CH; ++c) zv[c] = zx[c];
#pragma unroll
                        for (int c = 0; c < CH; ++c) v += (act ? ckr[q % (PRE_C ? QR : 1)][c] : 0.0) * zv[c];
                    }
                    v += upr_dpp_quad<0xB1>(v); v += upr_dpp_quad<0x4E>(v);
                    if (act && part == 0) ekp()[e] = v + e0r[q % (PRE_C ? QR : 1)];
                }
            } else {
                // multi-body shapes: the rows of C come straight from the records, GQ rows of a lane requested together
                // (one exposed latency per group instead of one per row)
                constexpr int GQ = UPR_QP3_GQ;
#pragma unroll 1
                for (int q0 = 0; q0 < QR; q0 += GQ) {
                    double cb[GQ][CH], e0b[GQ];
#pragma unroll
                    for (int g = 0; g < GQ; ++g) {
                        const int e4 = tid_ + (q0 + g) * NT;
                        const bool act = (q0 + g < QR) && e4 < N * NE * 4;
                        const int e = act ? (e4 >> 2) : 0, part = e4 & 3;
                        const double* Ck = rec(e / NE) + lin_gx + (e % NE) * NX + part * CH;
#pragma unroll
                        for (int c = 0; c < CH; ++c) cb[g][c] = (act && part * CH + c < NX) ? Ck[c] : 0.0;
                        e0b[g] = act ? G[F::e0 + e] : 0.0;
                    }
#pragma unroll
                    for (int g = 0; g < GQ; ++g) {
                        const int e4 = tid_ + (q0 + g) * NT;
                        const bool act = (q0 + g < QR) && e4 < N * NE * 4;
                        const int e = act ? (e4 >> 2) : 0, part = e4 & 3;
                        const double* zx = LK + O::Z + (e / NE) * NX + part * CH;
                        double v = 0.0;
#pragma unroll
                        for (int c = 0; c < CH; ++c) v += cb[g][c] * ((part * CH + c < NX) ? zx[c] : 0.0);
                        v += upr_dpp_quad<0xB1>(v); v += upr_dpp_quad<0x4E>(v);
                        if (act && part == 0) ekp()[e] = v + e0b[g];
                    }
                }
            }
            if (C::BIGF) UPR_SYNC(); else sync_lds();   // (BIGF: ek is a far array -- its stores are global)
        }
        UPR_FORT(e, N * NE) {
            const int k = e / NE, r = e % NE;
            double v = ekp()[e], v2 = 0.0;
            if (fresh_ek) {
                v += df_dot(r, LK + O::Z + N1 * NX + k * NU + NQ);
                ekp()[e] = v;
            }
            if (level > 0) v2 = df_dot(r, hfp() + k * NFC);
            LK[O::ys + e] = (rho_px != 0.0 ? v + rho_px * ws[W::nu + e] : v) - v2;
        }
#else
        UPR_FORT(e, N * NE) {
            const int k = e / NE, r = e % NE;
            const double* Ck = rec(k) + lin_gx + r * NX;
            double v, v2 = 0.0;
            if (fresh_ek) {
                v = G[F::e0 + e];
                for (int j = 0; j < NX; ++j) v += Ck[j] * LK[O::Z + k * NX + j];
                v += df_dot(r, LK + O::Z + N1 * NX + k * NU + NQ);
                ekp()[e] = v;
            } else v = ekp()[e];
            if (level > 0) v2 = df_dot(r, hfp() + k * NFC);
            LK[O::ys + e] = (rho_px != 0.0 ? v + rho_px * ws[W::nu + e] : v) - v2;
        }
#endif
        if (factor && !C::MULTI && !C::COUPLED) {
            UPR_FORT(e, N * NE * NE) {
                const int k = e / (NE * NE), r = (e % (NE * NE)) / NE, c = e % NE;
                if (c > r) continue;
                double acc = (r == c) ? rho_s : 0.0;
                const double* zr = L + O::Pa + (k * NE + r) * NFC; const double* zc = L + O::Pa + (k * NE + c) * NFC;
#pragma unroll
                for (int i = 0; i < NFC; ++i) acc += zr[i] * zc[i];
                L[O::sst + k * NE * NE + r * NE + c] = acc;
            }
        }
        UPR_SYNC(); toc(3);
        if (level == 0) return;
        // (phase E's rows of C come out of global memory: fetched here, consumed behind phase D)
        constexpr int QE = (N * NX + NT - 1) / NT;
        constexpr bool PRE_E = QE * NE <= 24;
        double ckp[PRE_E ? QE : 1][PRE_E ? NE : 1];
        if (PRE_E) {
#pragma unroll
            for (int q = 0; q < QE; ++q) {
                const int e = tid_ + q * NT;
                if (e < N * NX) {
                    const double* Ck = rec(e / NX) + lin_gx + e % NX;
#pragma unroll
                    for (int r = 0; r < NE; ++r) ckp[q % (PRE_E ? QE : 1)][r % (PRE_E ? NE : 1)] = Ck[r * NX];
                }
            }
        }
        // D: one lane per Schur block (a knot; a (knot, body) pair of the multi-body shapes): factor (LDS staging -> global),
        // ys = Lsi ee, zt = Lsi' ys
        constexpr int SB = C::SB;
        if (C::COUPLED && factor) {
            // dense S of every knot straight from the inverse contact factors (G[lfi], written in phase B) and Df:
            // S = rho I + sum_c (Lf_c^-1 Df_c')' (Lf_c^-1 Df_c').  A contact loads the rows of at most two bodies, so S is assembled
            // in 6 x 6 blocks, a lane per (knot, body pair): 36 accumulators a lane (rounds 2 - 3e: the knot's lane assembled all
            // 171 entries itself in front of the factorisation, out of a scratch frame).  Every entry sums its contacts in the
            // same order as before.  (Also tried: the factorisation on the knot's lane in place, the inverse a lane per column and
            // the two products a lane per row -- bit-identical and slower: phase D 142 k cycles against 91 k.)
            constexpr int NBL = C::NB * (C::NB + 1) / 2, NPK = SB * (SB + 1) / 2;
            UPR_FORT(e, N * NBL) {
                const int k = e / NBL, bl = e % NBL;
                int bi = 0, b0 = 0;
#pragma unroll
                for (int i = 1; i < C::NB; ++i) if (bl >= i * (i + 1) / 2) { bi = i; b0 = i * (i + 1) / 2; }
                const int bj = bl - b0;
                double acc[6][6];
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) acc[r][c] = (bi == bj && r == c) ? rho_s : 0.0;
                // The contacts of the pair come from its list (O::plist, built once per solve), the inverse contact factors of the
                // first PF of them are requested TOGETHER and the products run without a branch (a factor beyond the list is zero:
                // exact zeros are added).  Rounds 2 - 3 scanned all contacts with a test each: every matching contact then paid its
                // own global round trip -- about 45 k of the 82 k cycles phase D cost per interior-point iteration of configs[2].
                constexpr int PF = (NC < 8) ? NC : 8, NBK = (NF == 3) ? 9 : 1;
                const int npl = (int)L[O::pcnt + bl];
                double Bq[PF][NBK]; int cq[PF];
#pragma unroll
                for (int t = 0; t < PF; ++t) {
                    cq[t] = (int)L[O::plist + bl * NC + t];
#pragma unroll
                    for (int a = 0; a < NBK; ++a) { const double v = G[F::lfi + k * C::NLF + NBK * cq[t] + a]; Bq[t][a] = (t < npl) ? v : 0.0; }
                }
#pragma unroll
                for (int t = 0; t < PF; ++t) {
                    const int ci = cq[t];
                    double Bk[NBK];
#pragma unroll
                    for (int a = 0; a < NBK; ++a) Bk[a] = Bq[t][a];
                    double zi[6][NF], zj[6][NF];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const double* di = L + O::df + (6 * bi + r) * NFC + NF * ci;
                        const double* dj = L + O::df + (6 * bj + r) * NFC + NF * ci;
                        if (NF == 3) {
                            { const double d0 = di[0], d1 = di[1], d2 = di[2]; zi[r][0] = Bk[0] * d0; zi[r][1 % NF] = Bk[3 % NBK] * d0 + Bk[4 % NBK] * d1; zi[r][2 % NF] = Bk[6 % NBK] * d0 + Bk[7 % NBK] * d1 + Bk[8 % NBK] * d2; }
                            { const double d0 = dj[0], d1 = dj[1], d2 = dj[2]; zj[r][0] = Bk[0] * d0; zj[r][1 % NF] = Bk[3 % NBK] * d0 + Bk[4 % NBK] * d1; zj[r][2 % NF] = Bk[6 % NBK] * d0 + Bk[7 % NBK] * d1 + Bk[8 % NBK] * d2; }
                        } else { zi[r][0] = Bk[0] * di[0]; zj[r][0] = Bk[0] * dj[0]; }
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            double v = 0.0;
#pragma unroll
                            for (int a = 0; a < NF; ++a) v += zi[r][a] * zj[c][a];
                            acc[r][c] += v;
                        }
                }
                for (int t = PF; t < npl; ++t) {   // (longer lists: one contact at a time)
                    const int ci = (int)L[O::plist + bl * NC + t];
                    double Bk[NF == 3 ? 9 : 1];
                    if (NF == 3) {
#pragma unroll
                        for (int a = 0; a < 9; ++a) Bk[a] = G[F::lfi + k * C::NLF + 9 * ci + a];
                    } else Bk[0] = G[F::lfi + k * C::NLF + ci];
                    double zi[6][NF], zj[6][NF];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const double* di = L + O::df + (6 * bi + r) * NFC + NF * ci;
                        const double* dj = L + O::df + (6 * bj + r) * NFC + NF * ci;
                        if (NF == 3) {
                            { const double d0 = di[0], d1 = di[1], d2 = di[2]; zi[r][0] = Bk[0] * d0; zi[r][1 % NF] = Bk[3 % (NF == 3 ? 9 : 1)] * d0 + Bk[4 % (NF == 3 ? 9 : 1)] * d1; zi[r][2 % NF] = Bk[6 % (NF == 3 ? 9 : 1)] * d0 + Bk[7 % (NF == 3 ? 9 : 1)] * d1 + Bk[8 % (NF == 3 ? 9 : 1)] * d2; }
                            { const double d0 = dj[0], d1 = dj[1], d2 = dj[2]; zj[r][0] = Bk[0] * d0; zj[r][1 % NF] = Bk[3 % (NF == 3 ? 9 : 1)] * d0 + Bk[4 % (NF == 3 ? 9 : 1)] * d1; zj[r][2 % NF] = Bk[6 % (NF == 3 ? 9 : 1)] * d0 + Bk[7 % (NF == 3 ? 9 : 1)] * d1 + Bk[8 % (NF == 3 ? 9 : 1)] * d2; }
                        } else { zi[r][0] = Bk[0] * di[0]; zj[r][0] = Bk[0] * dj[0]; }
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            double v = 0.0;
#pragma unroll
                            for (int a = 0; a < NF; ++a) v += zi[r][a] * zj[c][a];
                            acc[r][c] += v;
                        }
                }
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        const int R = 6 * bi + r, Cc = 6 * bj + c;
                        if (Cc <= R) L[O::sdn + k * NPK + R * (R + 1) / 2 + Cc] = acc[r][c];
                    }
            }
            UPR_SYNC();
            ftoc(6, 3);
        }
#ifndef UPR_HOST_EMU
        if constexpr (C::COUPLED) {
            // Cooperative form of phase D for the dense Schur complement (round 4): SB lanes per knot, lane i owns ROW i of S through a
            // right-looking Cholesky factorisation -- the pivot column crosses the group through a slot per wave, the reciprocal
            // pivot by a lane shuffle -- then COLUMN i of the inverse factor by forward substitution out of the packed factor in
            // LDS (in place over S), and the two products ys = Lsi ee (a row per lane, out of the packed inverse in LDS) and
            // zt = Lsi' ys (the lane's own column).  Every sum runs in the order of the one-lane form (upr_chol_inv_serial, rounds
            // 2 - 3: one lane per knot, 171 entries in a scratch frame: with the assembly 82 k of the 505 k cycles of an
            // interior-point iteration of configs[2]); the groups of a wave run in lockstep, so wave-local ordering points are
            // all it takes.  The packed inverse stays in LDS for the corrector's call (level 1), which then needs no global operand.
            // (KQ: knots a lane carries side by side.  The chain of a pivot -- reciprocal square root, shuffle, LDS round trip: 640
            // cycles -- is latency, and two independent chains would share it; measured with KQ = 2 the 72 extra registers go to
            // scratch in this 512-register kernel and the phase takes 240 k cycles instead of 68 k: one knot per lane, two passes.)
            constexpr int GP = 64 / SB, KW = GP * (NT / 64), KQ = 1, KPP = KW * KQ, NPS = (N + KPP - 1) / KPP, NPK = SB * (SB + 1) / 2;
            static_assert(GP >= 1 && O::Pa + 64 * KQ * (NT / 64) <= O::yN, "a group per knot inside a wave; pivot-column slots in the sweeps' working set");
            const int ln = lane(), g = ln / SB, i = ln - g * SB, wv = wb >> 6;
            const int gc = (g < GP) ? g : 0, ri = i * (i + 1) / 2;
            double* colb = L + O::Pa + wv * 64 * KQ;
            bool ok = true;
#pragma unroll 1
            for (int pass = 0; pass < NPS; ++pass) {
                int kq[KQ]; bool act[KQ]; double* Sk[KQ];
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    kq[q] = pass * KPP + q * KW + wv * GP + g;
                    act[q] = g < GP && kq[q] < N;
                    if (!act[q]) kq[q] = 0;
                    Sk[q] = L + O::sdn + kq[q] * NPK;
                }
                double col[KQ][SB];
                if (factor) {
                    double row[KQ][SB];
#pragma unroll
                    for (int q = 0; q < KQ; ++q)
#pragma unroll
                        for (int c = 0; c < SB; ++c) row[q][c] = (c <= i) ? Sk[q][ri + c] : 0.0;
#pragma unroll
                    for (int p = 0; p < SB; ++p) {
                        double lp[KQ];
#pragma unroll
                        for (int q = 0; q < KQ; ++q) {
                            double s = row[q][p];
                            if (i == p && !(s > 0.0)) { if (act[q]) ok = false; s = 1.0; }
                            const double idg = upr_rsqrt(i == p ? s : 1.0);
                            const double d = __shfl(idg, gc * SB + p);
                            lp[q] = (i > p) ? row[q][p] * d : ((i == p) ? d : row[q][p]);
                            row[q][p] = lp[q];
                        }
                        if (p + 1 < SB) {
#pragma unroll
                            for (int q = 0; q < KQ; ++q) colb[64 * q + ln] = lp[q];
                            wsync_lds();
                            // (no test on c <= i: the entries of a row beyond its diagonal are never used, and a test would put every
                            //  one of these reads into a branch of its own with a full LDS round trip: measured 2.3 k cycles per pivot)
#pragma unroll
                            for (int q = 0; q < KQ; ++q)
#pragma unroll
                                for (int c = p + 1; c < SB; ++c) row[q][c] -= lp[q] * colb[64 * q + gc * SB + c];
                            wsync_lds();
                        }
                    }
                    ftoc(7, 3);
                    // the factor (reciprocal pivots on the diagonal) in place over S, then column i of its inverse
#pragma unroll
                    for (int q = 0; q < KQ; ++q) if (act[q]) {
#pragma unroll
                        for (int c = 0; c < SB; ++c) if (c <= i) Sk[q][ri + c] = row[q][c];
                    }
                    wsync_lds();
                    // forward substitution in axpy form: once entry r of the column is known it goes into the running sums of all
                    // later entries (seventeen independent operations) -- the dot-product form made every one of its 153 operations
                    // wait for its own LDS read behind the previous one (19 k cycles for two passes); each sum still takes its terms
                    // in the order of the one-lane form
                    // (the running sums live in the column's own registers until their entry is due)
#pragma unroll
                    for (int q = 0; q < KQ; ++q)
#pragma unroll
                        for (int r = 0; r < SB; ++r) col[q][r] = 0.0;
#pragma unroll
                    for (int r = 0; r < SB; ++r)
#pragma unroll
                        for (int q = 0; q < KQ; ++q) {
                            const double rec = Sk[q][r * (r + 1) / 2 + r];
                            col[q][r] = (r < i) ? 0.0 : ((r == i) ? rec : -col[q][r] * rec);
#pragma unroll
                            for (int s2 = r + 1; s2 < SB; ++s2) col[q][s2] += Sk[q][s2 * (s2 + 1) / 2 + r] * col[q][r];   // (col[r] = 0 above the lane's own column: exact zeros)
                        }
                    wsync_lds();   // every lane is through with the factor: the packed inverse takes its place
#pragma unroll
                    for (int q = 0; q < KQ; ++q) if (act[q]) {
                        double* Lsg = G + F::lsi + kq[q] * SB * SB;
#pragma unroll
                        for (int r = 0; r < SB; ++r) { Lsg[r * SB + i] = col[q][r]; if (r >= i) Sk[q][r * (r + 1) / 2 + i] = col[q][r]; }
                    }
                } else {
                    // (level 1: the packed inverse of the predictor's call is still in LDS)
#pragma unroll
                    for (int q = 0; q < KQ; ++q)
#pragma unroll
                        for (int r = 0; r < SB; ++r) { const double v = Sk[q][r * (r + 1) / 2 + ((r >= i) ? i : 0)]; col[q][r] = (r >= i) ? v : 0.0; }
                }
                wsync_lds();
                ftoc(8, 3);
                double yv[KQ], zv[KQ];
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    yv[q] = 0.0;
#pragma unroll
                    for (int m = 0; m < SB; ++m) { const double e = LK[O::ys + kq[q] * SB + m], a = Sk[q][ri + m]; yv[q] += ((m <= i) ? a : 0.0) * e; }
                }
                wsync_lds();
#pragma unroll
                for (int q = 0; q < KQ; ++q) if (act[q]) LK[O::ys + kq[q] * SB + i] = yv[q];
                wsync_lds();
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    zv[q] = 0.0;
#pragma unroll
                    for (int m = 0; m < SB; ++m) zv[q] += col[q][m] * LK[O::ys + kq[q] * SB + m];
                    if (act[q]) LK[O::zt + kq[q] * SB + i] = zv[q];
                }
                wsync_lds();
                ftoc(9, 3);
            }
            if (!ok) L[O::misc] = 1.0;
        } else
#endif
        UPR_FORT(kb, C::NKB) {
            double* Ls = G + F::lsi + kb * SB * SB;
            double Lr[SB * SB];                                   // the inverse factor stays in registers for the two products
            if (factor) {
                bool ok;
                if (C::MULTI) {
                    // S_b = rho I + sum over the forces of this body z z' out of the staged Z (nothing else touches the block)
                    const int k = kb / C::NB, b = kb % C::NB;
                    double Sm[SB * SB];
#pragma unroll
                    for (int r = 0; r < SB; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c) Sm[r * SB + c] = (r == c) ? rho_s : 0.0;
                    for (int ci = 0; ci < NC; ++ci) {
                        if (P->contact_body2[ci] != b) continue;
                        double Bk[C::BIGF ? 9 : 1];
                        if (C::BIGF) {
#pragma unroll
                            for (int a9 = 0; a9 < 9; ++a9) Bk[a9 % (C::BIGF ? 9 : 1)] = G[F::lfi + k * C::NLF + 9 * ci + a9];
                        }
                        for (int a = 0; a < NF; ++a) {
                            const double* z = LK + O::zst + (k * NFC + NF * ci + a) * 6;
                            double zv[SB];
                            if (C::BIGF) {
                                // column a of Z = Lf^-1 Df' of this contact: row a of the inverse factor times the body's rows of Df
#pragma unroll
                                for (int r = 0; r < SB; ++r) {
                                    double v = 0.0;
                                    for (int b3 = 0; b3 <= a; ++b3) v += Bk[(3 * a + b3) % (C::BIGF ? 9 : 1)] * L[O::df + (NF * ci + b3) * 6 + r];
                                    zv[r] = v;
                                }
                            } else {
#pragma unroll
                            for (int r = 0; r < SB; ++r) zv[r] = z[r];
                            }
#pragma unroll
                            for (int r = 0; r < SB; ++r)
#pragma unroll
                                for (int c = 0; c <= r; ++c) Sm[r * SB + c] += zv[r] * zv[c];
                        }
                    }
                    ok = upr_chol_inv_serial<SB>(Sm, Lr);
                } else if (C::COUPLED) {
                    // dense S of the knot, assembled above (LDS, packed lower triangle)
                    const int k = kb;
                    double Sm[SB * SB];
#pragma unroll
                    for (int r = 0; r < SB; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c) Sm[r * SB + c] = L[O::sdn + k * (SB * (SB + 1) / 2) + r * (r + 1) / 2 + c];
                    ok = upr_chol_inv_serial<SB>(Sm, Ls);       // (the inverse goes to memory: factor and inverse together exceed the registers)
                } else ok = upr_chol_inv_serial<SB>(L + O::sst + kb * SB * SB, Lr);
                if (!ok) L[O::misc] = 1.0;
                if (C::COUPLED) {
#pragma unroll
                    for (int r = 0; r < SB; ++r)
#pragma unroll
                        for (int m = 0; m <= r; ++m) Lr[r * SB + m] = Ls[r * SB + m];
                } else {
#pragma unroll
                    for (int e = 0; e < SB * SB; ++e) Ls[e] = Lr[e];
                }
            } else {
#pragma unroll
                for (int r = 0; r < SB; ++r)
#pragma unroll
                    for (int m = 0; m <= r; ++m) Lr[r * SB + m] = Ls[r * SB + m];
            }
#ifndef UPR_HOST_EMU
            if (C::VCPRE && factor) {
                // single-wave sweep: phase E forms Vc = Lsi C of every knot; it reads the factor out of LDS (packed lower
                // triangle, in the sweep's staging area).  That area overlaps the S blocks of the last knots, which other
                // lanes of THIS wave read at the top of their factorisation: all Schur blocks sit on one wave (lockstep),
                // so every read precedes every write.
                static_assert(!C::VCPRE || (C::NKB <= 64 && O::sw0 + C::NKB * (SB * (SB + 1) / 2) <= O::yN), "Lsi staging of phase D");
                UPR_WSYNC();   // (compiler-level: no lane's store below may be hoisted above another lane's reads of S; free at run time)
#pragma unroll
                for (int r = 0; r < SB; ++r)
#pragma unroll
                    for (int m = 0; m <= r; ++m) L[O::sw0 + kb * (SB * (SB + 1) / 2) + r * (r + 1) / 2 + m] = Lr[r * SB + m];
            }
#endif
            double ee[SB], yv[SB];
#pragma unroll
            for (int r = 0; r < SB; ++r) ee[r] = LK[O::ys + kb * SB + r];
#pragma unroll
            for (int r = 0; r < SB; ++r) { double v = 0.0;
#pragma unroll
                for (int m = 0; m <= r; ++m) v += Lr[r * SB + m] * ee[m];
                yv[r] = v; }
#pragma unroll
            for (int r = 0; r < SB; ++r) { double v = 0.0;
#pragma unroll
                for (int m = r; m < SB; ++m) v += Lr[m * SB + r] * yv[m];
                LK[O::ys + kb * SB + r] = yv[r]; LK[O::zt + kb * SB + r] = v; }
        }
        UPR_SYNC(); toc(4);
        // E: cs = C' zt
#pragma unroll
        for (int q = 0; q < QE; ++q) {
            const int e = tid_ + q * NT;
            if (e < N * NX) {
                const int k = e / NX;
                double v = 0.0;
                double cr[NE];   // (every row requested before the first product)
                if (PRE_E) {
#pragma unroll
                    for (int r = 0; r < NE; ++r) cr[r] = ckp[q % (PRE_E ? QE : 1)][r % (PRE_E ? NE : 1)];
                } else {
                    const double* Ck = rec(k) + lin_gx + e % NX;
#pragma unroll
                    for (int r = 0; r < NE; ++r) cr[r] = Ck[r * NX];
                }
#pragma unroll
                for (int r = 0; r < NE; ++r) v += cr[r] * LK[O::zt + k * NE + r];
#ifndef UPR_HOST_EMU
                if (C::VCPRE && factor && k >= 1) {
                    // column e % NX of Vc_k = Lsi_k C_k for the single-wave matrix sweep (its lanes read whole columns)
                    const double* Ls = L + O::sw0 + k * (NE * (NE + 1) / 2);
                    double* vo = L + vca(k) + (e % NX) * O::VCS;
#pragma unroll
                    for (int r = 0; r < NE; ++r) {
                        double w = 0.0;
#pragma unroll
                        for (int m = 0; m <= r; ++m) w += Ls[r * (r + 1) / 2 + m] * cr[m];
                        vo[r] = w;
                    }
                }
#endif
                LK[O::cs + e] = v;
            }
        }
        UPR_SYNC(); toc(5);
    }

    // terminal residual [p_d - p - Jp dq ; v ; a] at the current iterate -> L[eN]
    UPR_HDI void terminal_residual() {
        if (neN > 0) UPR_FORT(q, C::NEN) {
            double v;
            if (q < 3) { v = L[O::misc + 4 + q]; for (int j = 0; j < NQ; ++j) v -= L[O::jN + q * NQ + j] * (Zx(N)[j] - xs[N * NX + j]); }
            else v = Zx(N)[NQ + (q - 3)];
            L[O::eN + q] = v;
        }
    }

    // ---- backward sweep, matrix part: P_k, the factor of Hjj_k and V_k = Lj^-1 Hux_k for k = N-1 .. 0 ------------
    // Per knot: phase 1 (everything that is a function of P+ only) | barrier | wave 0: Cholesky of Hjj in the
    // registers of lane 0, then one lane per column of V by forward substitution -- meanwhile the other waves
    // preload the matrix-core accumulators with sym(A'P+A) + Q~ + Vc'Vc | barrier | P = acc - V'V | barrier.
    // The feedback K = Lj^-T V is not on this critical path: it is formed for all knots at once afterwards.
    // Knot 0 needs only the factor of Hjj_0 (the first state is fixed: no P_0, no K_0).
    UPR_HDI static constexpr int vmb(int k) { return O::vm + (k & 1) * O::r2(NQ * NX); }
    UPR_HDI static constexpr int lkb(int k) { return O::lk + (k & 1) * O::r2(C::NH); }
    // column c of the feedback K_k = Lj_k^-T V_k by back substitution out of the LDS copies of V_k and Lj_k
    UPR_HDI void feedback_column(int k, int c) {
        const double* Lp = L + lkb(k); const double* V = L + vmb(k);
        double kk[NQ];
#pragma unroll
        for (int i = NQ - 1; i >= 0; --i) {
            double t = V[i * NX + c];
#pragma unroll
            for (int m = i + 1; m < NQ; ++m) t -= Lp[m * (m + 1) / 2 + i] * kk[m];
            kk[i] = t * Lp[i * (i + 1) / 2 + i];
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) G[F::Ks + k * NQ * NX + i * NX + c] = kk[i];
    }
#ifndef UPR_HOST_EMU
    // ---- backward sweep, matrix part, on TWO waves with the cost-to-go in registers (C::SW) ---------------------------------------
    // The four-wave form below (now the host emulation's) spends a knot in three workgroup barriers and five LDS round trips, with
    // eight waves of two co-resident workgroups hitting the LDS at the same moments (~9 k cycles per knot).  Here:
    //   * wave 1, lane (bi, bj), bi <= bj (nq (nq + 1) / 2 = 45 lanes) keeps the 3 x 3 block P+[(a, bi)][(c, bj)] in REGISTERS: A'P+A,
    //     its rows of Hux = B'P+A (for both (bi, bj) and the mirrored block), its entry of Hjj = B'P+B + R + W and its partial sums of
    //     P+ b are in-lane arithmetic on nine values (block-scalar structure of the triple integrator, system_dynamics.h:15-22);
    //   * Hjj (packed), Hux (by columns) and the partial sums cross to wave 0 through LDS (barrier A); every lane of wave 0 factors
    //     Hjj for itself (9 dependent pivots) and lane c < nx carries column c of Hux through the eliminations: V = Lj^-1 Hux; lanes
    //     nx .. nx + nq - 1 carry the columns of the identity instead and end with Lj^-1, which the feed-forward phase multiplies
    //     with (kff = Lj^-T Lj^-1 r);
    //   * the same lanes back-substitute their column of the feedback K = Lj^-T V in registers and store it (coalesced);
    //   * V goes back through LDS by columns (barrier B), and lane (bi, bj) of wave 1 updates its block: P = A'P+A + Q~ + Vc'Vc - V'V
    //     with the six columns of V and of Vc = Lsi C it needs (one-body shapes: Vc of every knot was left in LDS by prep's phase E).
    // (A one-wave form of the same sweep -- rounds 3 - 5, UPR_QP3_SW2 = 0 -- was bound by instruction issue: ~960 per knot, a third of
    //  them the factorisation; deleted in round 6.)
    UPR_HDI static int vca(int k) { return (k <= O::KS) ? O::S + (k - 1) * O::VCN : O::Pa + (k - 1 - O::KS) * O::VCN; }
    // (P+ b: wave 1 writes its partial sums BEHIND barrier A and wave 0 reads them behind barrier B of the same knot, i.e. before it
    // arrives at the next barrier A, behind which wave 1 writes the next ones.)
    // Wave 1 keeps the blocks of P (A'P+A, Hux, Hjj, P+ b partial sums; the Vc'Vc part of the update while wave 0 factors; then the V'V
    // part) and wave 0 the columns (factorisation, V, K).  Two LDS-only workgroup barriers per knot hand Hjj / Hux over (A) and V back
    // (B); the remaining waves only take part in the barriers (multi-body shapes: they form Vc of the next knot and take shares of Vc'Vc).
    // start value of the fused predictor recursion, lane c < nx
    UPR_HDI double wt_terminal(int l) const {
        const double irho = 1.0 / UPR_QP_RHO_N;
        double v = LK[O::gxs + N * NX + l];
        if (neN > 0) {
            if (l < NQ) { for (int q = 0; q < 3; ++q) v -= L[O::jN + q * NQ + l] * (L[O::yN + q] + irho * L[O::eN + q]); }
            else v += L[O::yN + 3 + (l - NQ)] + irho * L[O::eN + 3 + (l - NQ)];
        }
        return v;
    }
    // every lane factors Hjj for itself and carries its column (vcl) of Hux / of the identity through the eliminations
    UPR_HDI void sw2_factor(double (&a)[NQ][NQ], double (&hx)[NQ], int vcl, bool& ok) const {
        constexpr int HXS = O::HXS;
#pragma unroll
        for (int i = 0; i < NQ; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) a[i][j] = L[O::sw_hj + i * (i + 1) / 2 + j];
#pragma unroll
        for (int i = 0; i < NQ; ++i) hx[i] = L[O::sw_hx + vcl * HXS + i];
#pragma unroll
        for (int p2 = 0; p2 < NQ; ++p2) {
            const double piv = a[p2][p2];
            ok = ok && (piv > 0.0);   // off the dependent chain: a non-positive pivot poisons the factor with NaN and flags the QP
            const double idg = upr_rsqrt(piv);
#pragma unroll
            for (int i = p2 + 1; i < NQ; ++i) a[i][p2] *= idg;
            hx[p2] *= idg;
#pragma unroll
            for (int j = p2 + 1; j < NQ; ++j) {
#pragma unroll
                for (int i = j; i < NQ; ++i) a[i][j] -= a[i][p2] * a[j][p2];
                hx[j] -= a[j][p2] * hx[p2];
            }
            a[p2][p2] = idg;   // the diagonal keeps its reciprocal
        }
    }
    // the side work of knot k: feedback column (or column of Lj^-1) by back substitution and its store, the sum of P+ b, the fused
    // predictor step (wt: w~_k in, w~_{k-1} out)
    UPR_HDI void sw2_side(int k, const double (&a)[NQ][NQ], const double (&hx)[NQ], const double (&pbv)[NQ], double& wt, int l, bool vl, int vj_,
                          double ca0, double ca1, double ca2, double gk_pre = 0.0, double uk_pre = 0.0) {
        double kk[NQ];
#pragma unroll
        for (int i = NQ - 1; i >= 0; --i) {
            double tt = hx[i];
#pragma unroll
            for (int m = i + 1; m < NQ; ++m) tt -= a[m][i] * kk[m];
            kk[i] = vl ? tt * a[i][i] : hx[i];   // (lanes nx ..: their column of Lj^-1 is what is stored)
        }
        if (l < NX + NQ) {
            double* const dst = G + (vl ? F::Ks + l : F::Ljis + (l - NX)) + k * NQ * NX;
#pragma unroll
            for (int i = 0; i < NQ; ++i) dst[i * NX] = kk[i];
        }
        double pbs;
        {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; q += 3) { s0 += pbv[q]; if (q + 1 < NQ) s1 += pbv[q + 1]; if (q + 2 < NQ) s2 += pbv[q + 2]; }
            pbs = (s0 + s1) + s2;
            if (vl) LK[O::Pbs + k * NX + l] = pbs;
        }
        {
            // the predictor's vector sweep (backward_vec's recursion, mode 0) rides along, K_k out of registers:
            // w_k = w~_k + (P+ b)_k ; rq = gu_k[jerk] + B'w_k (-> the feed-forward phase) ; w~_{k-1} = gx_k + C_k'zt_k + A'w_k - K_k' rq
            // (KFAR: requested by the caller at the top of the knot, in front of the factorisation)
            const double gk = C::KFAR ? gk_pre : ((k >= 1) ? LK[O::gxs + k * NX + (vl ? l : 0)] + LK[O::cs + k * NX + (vl ? l : 0)] : 0.0);
            const double uk = C::KFAR ? uk_pre : LK[O::gus + k * NU + vj_];
            const double wk = wt + pbs;
            if (vl) L[O::sw_w + l] = wk;
            UPR_WSYNC();
            const double w0 = L[O::sw_w + vj_], w1 = L[O::sw_w + NQ + vj_], w2 = L[O::sw_w + 2 * NQ + vj_];
            const double rq = ((h3 * w0 + h2 * w1) + h * w2) + uk;
            if (l < NQ) LK[O::kffs + k * NQ + l] = rq;
            double v0 = gk + ca0 * w0, v1 = ca1 * w1, v2 = ca2 * w2;
#pragma unroll
            for (int m = 0; m < NQ; m += 3) {
                v0 -= kk[m] * upr_readlane(rq, m);
                if (m + 1 < NQ) v1 -= kk[m + 1] * upr_readlane(rq, m + 1);
                if (m + 2 < NQ) v2 -= kk[m + 2] * upr_readlane(rq, m + 2);
            }
            wt = (v0 + v1) + v2;
            UPR_WSYNC();
        }
    }
    UPR_HDI void backward_mat_sw2() {
        constexpr int NBK = C::NH, HXS = O::HXS;
        static_assert(!SW2 || NT >= (C::VCPRE ? 128 : 256), "two waves (four where Vc is formed inside the sweep)");
        const int wave = wb >> 6;
        const int l = lane();
        const double irho = 1.0 / UPR_QP_RHO_N;
        terminal_residual();
        if (!C::VCPRE && wave >= 2) form_vc(N - 1);
        sync_lds();
        // (the barriers A / B inside the sweep hand over LDS staging only -- Hjj, Hux, V, the shares of Vc'Vc -- in KFAR instantiations
        //  too: the whole-horizon arrays the sweep reads were written before the barrier above, the ones it writes are read behind the
        //  full barrier at its end.  As full barriers they made every knot wait for all four waves' far requests: 840 k -> see NOTES_r06)
        if (wave == 1) {
            UPR_SETPRIO(UPR_QP3_PRIO_W1);
            const bool blk = l < NBK;
            const int lc = blk ? l : NBK - 1;      // = upr_tri(NQ, bi, bj)
            int bi = 0, b0 = 0;
#pragma unroll
            for (int i = 1; i < NQ; ++i) { const int st = i * NQ - i * (i - 1) / 2; if (lc >= st) { bi = i; b0 = st; } }
            const int bj = bi + (lc - b0);
            const bool dg = bi == bj;
            double p[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) p[a][c] = 0.0;
            {
                double v = dg ? LK[O::wx + N * NX + bi] : 0.0;
                if (neN > 0) { for (int q = 0; q < 3; ++q) v += irho * L[O::jN + q * NQ + bi] * L[O::jN + q * NQ + bj]; }
                p[0][0] = v;
                const double d1 = LK[O::wx + N * NX + NQ + bi] + ((neN > 0) ? irho : 0.0), d2 = LK[O::wx + N * NX + 2 * NQ + bi] + ((neN > 0) ? irho : 0.0);
                p[1][1] = dg ? d1 : 0.0; p[2][2] = dg ? d2 : 0.0;
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no wait for earlier global loads inside the loop
            double hjd = h * L[O::rd + bi] + LK[O::wu + (N - 1) * NU + bi];
            // KFAR: the knot's defects and barrier diagonals come out of the far array -- requested one knot ahead (out of LDS they are
            // read where they are used)
            double nbj[3], nbi[3], nwx[3];
            if (C::KFAR) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { nbj[c] = LK[O::bks + (N - 1) * NX + c * NQ + bj]; nbi[c] = LK[O::bks + (N - 1) * NX + c * NQ + bi]; nwx[c] = LK[O::wx + (N - 1) * NX + c * NQ + bi]; }
            }
#pragma nounroll
            for (int k = N - 1; k >= 0; --k) {
                // critical path first: Hux (both orientations of the block) and Hjj, straight from the registers of P+
                double hxa[3], hxb[3], up[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double t0 = (c == 0) ? p[0][0] : ((c == 1) ? h * p[0][0] + p[0][1] : h2 * p[0][0] + h * p[0][1] + p[0][2]);
                    const double t1 = (c == 0) ? p[1][0] : ((c == 1) ? h * p[1][0] + p[1][1] : h2 * p[1][0] + h * p[1][1] + p[1][2]);
                    const double t2 = (c == 0) ? p[2][0] : ((c == 1) ? h * p[2][0] + p[2][1] : h2 * p[2][0] + h * p[2][1] + p[2][2]);
                    hxa[c] = h3 * t0 + h2 * t1 + h * t2;                                // Hux[bi][(c, bj)]
                    up[c] = h3 * p[c][0] + h2 * p[c][1] + h * p[c][2];
                }
                hxb[0] = up[0]; hxb[1] = h * up[0] + up[1]; hxb[2] = h2 * up[0] + h * up[1] + up[2];   // Hux[bj][(c, bi)]
                double hj = h3 * up[0] + h2 * up[1] + h * up[2];
                if (dg) hj += hjd;   // h R + W of this knot (requested during the previous knot's update)
                if (blk) {
                    L[O::sw_hj + bj * (bj + 1) / 2 + bi] = hj;
#pragma unroll
                    for (int c = 0; c < 3; ++c) L[O::sw_hx + (c * NQ + bj) * HXS + bi] = hxa[c];
                    if (!dg) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) L[O::sw_hx + (c * NQ + bi) * HXS + bj] = hxb[c];
                    }
                }
                UPR_SYNC_LDS();   // A: Hjj, Hux are in LDS
                toc(6);
                // while wave 0 factors: the partial sums of P+ b (wave 0 adds them up behind barrier B) ...
                const double heek = (k > 0) ? G[hee_w + k * C::NH + lc] : 0.0;   // (used at the very end of the knot)
                {
                    double bjv[3], biv[3], r1[3], r2[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { bjv[c] = C::KFAR ? nbj[c] : LK[O::bks + k * NX + c * NQ + bj]; biv[c] = C::KFAR ? nbi[c] : LK[O::bks + k * NX + c * NQ + bi]; }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        r1[c] = p[c][0] * bjv[0] + p[c][1] * bjv[1] + p[c][2] * bjv[2];     // -> (P+ b)[(c, bi)]
                        r2[c] = p[0][c] * biv[0] + p[1][c] * biv[1] + p[2][c] * biv[2];     // -> (P+ b)[(c, bj)]   (bi < bj)
                    }
                    if (blk) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) L[O::sw_pb + (c * NQ + bi) * HXS + bj] = r1[c];
                        if (!dg) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) L[O::sw_pb + (c * NQ + bj) * HXS + bi] = r2[c];
                        }
                    }
                }
                // ... and p1 = sym(A'P+A) + Q~ + Vc'Vc
                double p1[3][3];
                if (k > 0) {
                    // rows of Vc in chunks of RC: the columns (., bj) and (., bi) of the chunk in registers, the next chunk requested
                    // before this one's products (the multi-body shapes have 18 .. 48 rows: all at once would be 576 registers)
                    constexpr int RC = VC_RC, NCH = VC_N1;   // (multi-body shapes: waves 2 and 3 take the other chunks, vc_share)
                    double wxk[3], qdk[3], o2[3][3], cj[2][3][RC], ci[2][3][RC];
                    const double* Vk = L + (C::VCPRE ? vca(k) : O::sw_vcb + (k & 1) * O::VCN);
                    int bic = bi, bjc = bj;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        wxk[c] = C::KFAR ? nwx[c] : LK[O::wx + k * NX + c * NQ + bi]; qdk[c] = L[O::qd + c * NQ + bi];
#pragma unroll
                        for (int r = 0; r < RC; ++r) { cj[0][c][r] = Vk[(c * NQ + bj) * O::VCS + r]; ci[0][c][r] = Vk[(c * NQ + bi) * O::VCS + r]; }
                    }
                    {
                        double t[3][3];
#pragma unroll
                        for (int a3 = 0; a3 < 3; ++a3) { t[a3][0] = p[a3][0]; t[a3][1] = h * p[a3][0] + p[a3][1]; t[a3][2] = h2 * p[a3][0] + h * p[a3][1] + p[a3][2]; }
#pragma unroll
                        for (int c = 0; c < 3; ++c) { o2[0][c] = t[0][c]; o2[1][c] = h * t[0][c] + t[1][c]; o2[2][c] = h2 * t[0][c] + h * t[1][c] + t[2][c]; }
                        if (dg) { o2[1][0] = o2[0][1]; o2[2][0] = o2[0][2]; o2[2][1] = o2[1][2]; }   // a diagonal block stays exactly symmetric
                    }
#pragma unroll
                    for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            double acc = o2[a3][c];
                            if (a3 == c) acc += dg ? (h * qdk[a3] + wxk[a3]) : 0.0;
                            if (a3 == 0 && c == 0) acc += h * heek;
                            p1[a3][c] = acc;
                        }
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        if (ch + 1 < NCH) {
#pragma unroll
                            for (int c = 0; c < 3; ++c)
#pragma unroll
                                for (int r = 0; r < RC; ++r) {
                                    cj[(ch + 1) & 1][c][r] = Vk[(c * NQ + bjc) * O::VCS + (ch + 1) * RC + r];
                                    ci[(ch + 1) & 1][c][r] = Vk[(c * NQ + bic) * O::VCS + (ch + 1) * RC + r];
                                }
                        }
#pragma unroll
                        for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                double acc = p1[a3][c];
#pragma unroll
                                for (int r = 0; r < RC; ++r) acc += ci[ch & 1][a3][r] * cj[ch & 1][c][r];
                                p1[a3][c] = acc;
                            }
                        if (NCH > 2) asm volatile("" : "+v"(p1[2][2]), "+v"(bic), "+v"(bjc));   // (the chunk after next is not requested before this one is consumed)
                    }
                }
                toc(7);
                UPR_SYNC_LDS();   // B: V is in LDS
                toc(8);
                if (k == 0) break;
                hjd = h * L[O::rd + bi] + LK[O::wu + (k - 1) * NU + bi];
                if (C::KFAR) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { nbj[c] = LK[O::bks + (k - 1) * NX + c * NQ + bj]; nbi[c] = LK[O::bks + (k - 1) * NX + c * NQ + bi]; nwx[c] = LK[O::wx + (k - 1) * NX + c * NQ + bi]; }
                }
                {
                    double vj[3][NQ], vi[3][NQ];
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int m = 0; m < NQ; ++m) {
                            vj[c][m] = L[O::sw_hx + (c * NQ + bj) * HXS + m];
                            vi[c][m] = L[O::sw_hx + (c * NQ + bi) * HXS + m];
                        }
#pragma unroll
                    for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            double acc = p1[a3][c];
                            if (!C::VCPRE) acc += L[O::sw_pq + (3 * a3 + c) * 64 + l] + L[O::sw_pq + (9 + 3 * a3 + c) * 64 + l];
#pragma unroll
                            for (int m = 0; m < NQ; ++m) acc -= vi[a3][m] * vj[c][m];
                            p[a3][c] = acc;
                        }
                }
                UPR_WSYNC();   // (the reads of V precede the next knot's stores of Hux: same wave, in order)
                toc(9);
            }
            UPR_SETPRIO(0);
        } else if (wave == 0) {
            UPR_SETPRIO(UPR_QP3_PRIO_W0);
            const bool vl = l < NX;                // lanes that carry a column of Hux
            const int vcl = (l < NX + NQ) ? l : 0;  // (lanes nx .. nx + nq - 1 read a column of the identity, kept behind Hux)
            if (l >= NX && l < NX + NQ) {
#pragma unroll
                for (int i = 0; i < NQ; ++i) L[O::sw_hx + l * HXS + i] = (l - NX == i) ? 1.0 : 0.0;
            }
            bool ok = true;
            // fused predictor sweep (backward_vec's recursion, mode 0): lane c < nx carries w~_k[c] = w_k[c] - (P+ b)_k[c]
            const int vj_ = vl ? l % NQ : 0, vb_ = vl ? l / NQ : 0;
            const double ca0 = coefA(0, vb_), ca1 = (vb_ >= 1) ? coefA(1, vb_) : 0.0, ca2 = (vb_ >= 2) ? 1.0 : 0.0;
            double wt = 0.0;
            if (vl) wt = wt_terminal(l);
#pragma nounroll
            for (int k = N - 1; k >= 0; --k) {
                UPR_SYNC_LDS();   // A
                toc(6);
                double gk_pre = 0.0, uk_pre = 0.0;
                if (C::KFAR) { gk_pre = (k >= 1) ? LK[O::gxs + k * NX + (vl ? l : 0)] + LK[O::cs + k * NX + (vl ? l : 0)] : 0.0; uk_pre = LK[O::gus + k * NU + vj_]; }
                double a[NQ][NQ], hx[NQ];
                sw2_factor(a, hx, vcl, ok);
                if (k > 0 && vl) {
#pragma unroll
                    for (int m = 0; m < NQ; ++m) L[O::sw_hx + l * HXS + m] = hx[m];
                }
                toc(7);
                UPR_SYNC_LDS();   // B
                toc(8);
                {
                    // off the critical path (wave 1 updates P meanwhile): P+ b, the feedback column by back substitution, its store
                    double pbv[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) pbv[q] = L[O::sw_pb + (vl ? l : 0) * HXS + q];
                    sw2_side(k, a, hx, pbv, wt, l, vl, vj_, ca0, ca1, ca2, gk_pre, uk_pre);
                }
                toc(9);
            }
            if (!ok && l == 0) L[O::misc] = 1.0;
            UPR_SETPRIO(0);
        } else {
            vc_regs vq;
            vcm_regs vm;
            int bi = 0, bj = 0;
            {
                const int lc = (l < C::NH) ? l : C::NH - 1;
                int b0 = 0;
#pragma unroll
                for (int i = 1; i < NQ; ++i) { const int st = i * NQ - i * (i - 1) / 2; if (lc >= st) { bi = i; b0 = st; } }
                bj = bi + (lc - b0);
            }
            if (!C::VCPRE && N - 2 >= 1) { if constexpr (C::COUPLED) { if (wave == 2) form_vcm_load(N - 2, vm); } else form_vc_load(N - 2, vq); }
#pragma nounroll
            for (int k = N - 1; k >= 0; --k) {
                UPR_SYNC_LDS();   // A
                toc(6);
                // Vc of knot k - 1 goes into the buffer whose last readers (knot k + 1) finished before barrier B of that knot; its
                // operands were requested behind that barrier.  Stacked bodies: wave 2 forms it on the matrix cores now, in the
                // long interval; star arrangements: behind barrier B, the shares of Vc_k'Vc_k first
                if (C::COUPLED && wave == 2 && k - 1 >= 1) form_vcm_store(k - 1, vm);
                if (!C::VCPRE && k >= 1) { if (wave == 2) vc_share<0>(k, bi, bj); else vc_share<1>(k, bi, bj); }   // (added by wave 1 behind barrier B)
                toc(7);
                UPR_SYNC_LDS();   // B
                toc(8);
                if (!C::VCPRE && !C::COUPLED && k - 1 >= 1) form_vc_store(k - 1, vq);                                   // (read behind the NEXT barrier A)
                if (!C::VCPRE && k - 2 >= 1) { if constexpr (C::COUPLED) { if (wave == 2) form_vcm_load(k - 2, vm); } else form_vc_load(k - 2, vq); }
                toc(9);
            }
        }
        UPR_SYNC();
    }
    // rows of Vc in chunks of VC_RC; wave 1 takes the first VC_N1 chunks of the products Vc'Vc, waves 2 and 3 the rest
    static constexpr int VC_RC = 6, VC_NCH = NE / VC_RC;
    // (star arrangements, 8 chunks: 4 | 2 | 2 -- waves 2 and 3 also form Vc of the next knot; stacked bodies, 3 chunks: 1 | 0 | 2 --
    // wave 2 forms Vc on the matrix cores and takes no share)
    static constexpr int VC_N1 = C::VCPRE ? VC_NCH : (C::COUPLED ? 1 : VC_NCH / 2), VC_N2 = (C::VCPRE || C::COUPLED) ? 0 : (VC_NCH - VC_N1) / 2;
    static_assert(NE % VC_RC == 0 && VC_N1 >= 1 && VC_N1 + VC_N2 <= VC_NCH, "chunks of six rows, split over three waves");
    // share WHICH (0: wave 2, 1: wave 3) of block (bi, bj)'s part of Vc_k'Vc_k, lane = block as on wave 1 -> sw_pq
    template <int WHICH>
    UPR_HDI void vc_share(int k, int bi, int bj) {
        constexpr int RC = VC_RC, c0 = (WHICH == 0) ? VC_N1 : VC_N1 + VC_N2, c1 = (WHICH == 0) ? VC_N1 + VC_N2 : VC_NCH, NCHS = c1 - c0;
        const int l = lane();
        const double* Vk = L + O::sw_vcb + (k & 1) * O::VCN;
        double acc[3][3], cj[2][3][RC], ci[2][3][RC];
#pragma unroll
        for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[a3][c] = 0.0;
        if (NCHS > 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < RC; ++r) { cj[0][c][r] = Vk[(c * NQ + bj) * O::VCS + c0 * RC + r]; ci[0][c][r] = Vk[(c * NQ + bi) * O::VCS + c0 * RC + r]; }
        }
#pragma unroll
        for (int ch = 0; ch < NCHS; ++ch) {
            if (ch + 1 < NCHS) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < RC; ++r) {
                        cj[(ch + 1) & 1][c][r] = Vk[(c * NQ + bj) * O::VCS + (c0 + ch + 1) * RC + r];
                        ci[(ch + 1) & 1][c][r] = Vk[(c * NQ + bi) * O::VCS + (c0 + ch + 1) * RC + r];
                    }
            }
#pragma unroll
            for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < RC; ++r) acc[a3][c] += ci[ch & 1][a3][r] * cj[ch & 1][c][r];
        }
#pragma unroll
        for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
            for (int c = 0; c < 3; ++c) L[O::sw_pq + (9 * WHICH + 3 * a3 + c) * 64 + l] = acc[a3][c];
    }
    // Vc_k = blockdiag(Lsi_k) C_k into the sweep's double buffer, by the waves from the third on (multi-body shapes): jobs of
    // three rows of one column, operands straight from global memory (the rows of C of the linearisation record, the inverse
    // Schur factor(s) prep left in the far array)
    // jobs: VC_JR rows of one column; the rows of a job lie in one Schur block (six: a whole body of the star arrangements)
    static constexpr int VC_JR = (C::SB % 6 == 0) ? 6 : 3, VC_NJ = (NE / VC_JR) * NX, VC_NL = (NT > 128) ? NT - 128 : 64, VC_R = (VC_NJ + VC_NL - 1) / VC_NL;
    struct vc_regs { double cm[VC_R][C::SB], lr[VC_R][VC_JR][C::SB]; };
    // (the operands are requested a phase ahead of their use: form_vc_load right behind barrier A, form_vc_store behind B)
    UPR_HDI void form_vc_load(int k, vc_regs& q) const {
        constexpr int SBV = C::SB;
        static_assert(C::VCPRE || (SBV % VC_JR == 0 && NT >= 192), "jobs inside one block; at least one wave to run them");
        const int t = tid() - 128;
        const double* Ck = rec(k) + lin_gx;
        const double* Lk = G + F::lsi + k * C::NLS;
#pragma unroll
        for (int r = 0; r < VC_R; ++r) {
            const int f = (t + r * VC_NL < VC_NJ) ? t + r * VC_NL : 0;
            const int g = f / NX, c = f % NX, r0 = VC_JR * g, blk = r0 / SBV, bo = SBV * blk, q0 = r0 - bo;
            const double* Ls = Lk + SBV * SBV * blk;
#pragma unroll
            for (int m = 0; m < SBV; ++m) {
                q.cm[r][m] = Ck[(bo + m) * NX + c];
#pragma unroll
                for (int i = 0; i < VC_JR; ++i) if (SBV > VC_JR || m <= i) q.lr[r][i][m] = Ls[(q0 + i) * SBV + m];   // (one job per block: q0 = 0, the triangle only)
            }
        }
    }
    UPR_HDI void form_vc_store(int k, const vc_regs& q) {
        constexpr int SBV = C::SB;
        const int t = tid() - 128;
        double* out = L + O::sw_vcb + (k & 1) * O::VCN;
#pragma unroll
        for (int r = 0; r < VC_R; ++r) {
            const int f = t + r * VC_NL;
            const int fc = (f < VC_NJ) ? f : 0;
            const int g = fc / NX, c = fc % NX, r0 = VC_JR * g, q0 = r0 - SBV * (r0 / SBV);
            double v[VC_JR];
#pragma unroll
            for (int i = 0; i < VC_JR; ++i) {
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < SBV; ++m) {
                    if (SBV > VC_JR) a += ((m <= q0 + i) ? q.lr[r][i][m] : 0.0) * q.cm[r][m];   // full-length rows, entries above the diagonal masked
                    else if (m <= i) a += q.lr[r][i][m] * q.cm[r][m];
                }
                v[i] = a;
            }
            if (f < VC_NJ) {
#pragma unroll
                for (int i = 0; i < VC_JR; ++i) out[c * O::VCS + r0 + i] = v[i];
            }
        }
    }
    // Dense Schur factor (stacked bodies): Vc = Lsi C (18 x 18 lower triangular times 18 x nx) as v_mfma_f64_16x16x4_f64 on ONE
    // wave, operands straight from global memory (lane l feeds A[l & 15][l >> 4], B[l >> 4][l & 15] and receives
    // D[(l >> 4) + 4 q][l & 15]): 20 loads per lane instead of the 126 of the lane-job form
    static constexpr int VM_TR = (C::SB + 15) / 16, VM_TC = (NX + 15) / 16, VM_NS = (C::SB + 3) / 4;
    struct vcm_regs { double av[VM_TR][VM_NS], bv[VM_TC][VM_NS]; };
    UPR_HDI void form_vcm_load(int k, vcm_regs& q) const {
        constexpr int SBV = C::SB;
        const int ln = lane(), l15 = ln & 15, k4 = ln >> 4;
        const double* Ck = rec(k) + lin_gx;
        const double* Lk = G + F::lsi + k * C::NLS;
#pragma unroll
        for (int s4 = 0; s4 < VM_NS; ++s4) {
            const int m = 4 * s4 + k4, mc = (m < SBV) ? m : SBV - 1;
#pragma unroll
            for (int tr = 0; tr < VM_TR; ++tr) { const int row = 16 * tr + l15, rc = (row < SBV) ? row : SBV - 1; q.av[tr][s4] = Lk[rc * SBV + mc]; }
#pragma unroll
            for (int tc = 0; tc < VM_TC; ++tc) { const int col = 16 * tc + l15, cc = (col < NX) ? col : NX - 1; q.bv[tc][s4] = Ck[mc * NX + cc]; }
        }
    }
    UPR_HDI void form_vcm_store(int k, vcm_regs& q) {
        typedef double v4dv __attribute__((ext_vector_type(4)));
        constexpr int SBV = C::SB;
        const int ln = lane(), l15 = ln & 15, k4 = ln >> 4;
        double* out = L + O::sw_vcb + (k & 1) * O::VCN;
#pragma unroll
        for (int s4 = 0; s4 < VM_NS; ++s4) {
            const int m = 4 * s4 + k4;
#pragma unroll
            for (int tr = 0; tr < VM_TR; ++tr) { const int row = 16 * tr + l15; q.av[tr][s4] = (row < SBV && m < SBV && m <= row) ? q.av[tr][s4] : 0.0; }
#pragma unroll
            for (int tc = 0; tc < VM_TC; ++tc) { const int col = 16 * tc + l15; q.bv[tc][s4] = (m < SBV && col < NX) ? q.bv[tc][s4] : 0.0; }
        }
        v4dv dacc[VM_TR][VM_TC];
#pragma unroll
        for (int tr = 0; tr < VM_TR; ++tr)
#pragma unroll
            for (int tc = 0; tc < VM_TC; ++tc) dacc[tr][tc] = v4dv{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < VM_NS; ++s4)
#pragma unroll
            for (int tr = 0; tr < VM_TR; ++tr) {
                if (4 * s4 > 16 * tr + 15) continue;   // the rows of this tile end before these columns of the factor begin
#pragma unroll
                for (int tc = 0; tc < VM_TC; ++tc) dacc[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(q.av[tr][s4], q.bv[tc][s4], dacc[tr][tc], 0, 0, 0);
            }
#pragma unroll
        for (int tr = 0; tr < VM_TR; ++tr)
#pragma unroll
            for (int tc = 0; tc < VM_TC; ++tc)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) { const int r = 16 * tr + k4 + 4 * qq, col = 16 * tc + l15; if (r < NE && col < NX) out[col * O::VCS + r] = dacc[tr][tc][qq]; }
    }
    UPR_HDI void form_vc(int k) {
        if constexpr (C::COUPLED) { if ((wb >> 6) == 2) { vcm_regs q; form_vcm_load(k, q); form_vcm_store(k, q); } }
        else { vc_regs q; form_vc_load(k, q); form_vc_store(k, q); }
    }
#endif
    UPR_HDI void backward_mat() {
#ifndef UPR_HOST_EMU
        static_assert(C::SW, "device instantiations run the two-wave matrix sweep (chains of up to nine joints)");
        backward_mat_sw2();
#else
#include "upr_qp3_emu_mat.h"   // (the host emulation's plain-loop form of the sweep)
#endif
    }

    // ---- backward sweep, vector part, for the current right-hand side ----------------------------------------
    // With w_k = p_{k+1} + P_{k+1} b_k the recursion is  p_k = gx_k + C_k'zt_k + A'w_k - K_k'(gu_k[jerk] + B'w_k):
    // one wave-local phase per knot (lane i < nx owns p_k[i] and column i of K_k, prefetched one
    // knot ahead).  w_k is kept (in the step array, which the forward sweep overwrites afterwards) for the
    // feed-forward  kff_k = Hjj_k^-1 (gu_k[jerk] + B'w_k)  of all knots at once.
    UPR_HDI double* Wk(int k) const { return LK + O::S + (k + 1) * NX; }
    // fused: the recursion ran on wave 0 of the two-wave matrix sweep (backward_mat_sw2) and left rq_k = gu_k[jerk] + B'w_k in the
    // feed-forward slots: only the feed-forward phase is left
    UPR_HDI void backward_vec(bool fused = false) {
        const double irho = 1.0 / UPR_QP_RHO_N;
        if (!fused) {
        terminal_residual();
        UPR_SYNC();
        }
        toc(11);
        if (!fused && wave0()) {
            UPR_SETPRIO(UPR_QP3_PRIO_VEC);
            UPR_FORT(i, NX) {
                double v = LK[O::gxs + N * NX + i];
                if (neN > 0) {
                    if (i < NQ) { for (int q = 0; q < 3; ++q) v -= L[O::jN + q * NQ + i] * (L[O::yN + q] + irho * L[O::eN + q]); }
                    else v += L[O::yN + 3 + (i - NQ)] + irho * L[O::eN + 3 + (i - NQ)];
                }
                Wk(N - 1)[i] = v + LK[O::Pbs + (N - 1) * NX + i];
            }
            UPR_WSYNC();
#ifndef UPR_HOST_EMU
            // Lane 4 j + b (b < 3) owns component (b, j) of the costate vector and carries it in a REGISTER from knot to
            // knot: the recursion never waits on LDS.  Per knot: B'w + gu of a joint is a sum inside its quad (DPP), the
            // nine of them reach every lane through scalar registers (v_readlane), A'w is quad-local (DPP broadcasts);
            // w is written to LDS for the later phases without being read back here.  Column i of K_k, K_{k-1}, K_{k-2}:
            // three knots of register prefetch cover the L2 / fabric latency.
            static_assert(4 * NQ <= 64, "one quad per joint");
            {
                const int l = lane();
                const bool act = ((l & 3) < 3) && ((l >> 2) < NQ);
                const int j = act ? (l >> 2) : 0, b = act ? (l & 3) : 0;
                const int i = b * NQ + j;
                double kc[UPR_QP3_KSTAGES][NQ];   // column i of K_k, ..., K_{k-S+1}: S knots of register prefetch cover the L2 / fabric latency
#define UPR_LOADKC(dst, kk) do { if (act && (kk) >= 1) { _Pragma("unroll") for (int m = 0; m < NQ; ++m) dst[m] = G[F::Ks + (kk) * NQ * NX + m * NX + i]; } } while (0)
                // operands of a step that do not depend on the recursion (fetched one step ahead)
#define UPR_LOADST(gk, pk, uk, kk) do { if ((kk) >= 1) { gk = LK[O::gxs + (kk) * NX + i] + LK[O::cs + (kk) * NX + i]; pk = LK[O::Pbs + ((kk) - 1) * NX + i]; uk = LK[O::gus + (kk) * NU + j]; } } while (0)
#define UPR_VECSTEP(kcx, gk, pk, uk, kk) do { \
                    double rq = cbq * wv; \
                    rq += upr_dpp_quad<0xB1>(rq); rq += upr_dpp_quad<0x4E>(rq); \
                    rq += uk; \
                    const double w0 = upr_dpp_quad<0x00>(wv), w1 = upr_dpp_quad<0x55>(wv), w2 = upr_dpp_quad<0xAA>(wv); \
                    double v0 = gk + ca0 * w0, v1 = ca1 * w1, v2 = ca2 * w2; \
                    _Pragma("unroll") for (int m = 0; m < NQ; m += 3) { \
                        v0 -= kcx[m] * upr_readlane(rq, 4 * m); \
                        if (m + 1 < NQ) v1 -= kcx[m + 1] * upr_readlane(rq, 4 * (m + 1)); \
                        if (m + 2 < NQ) v2 -= kcx[m + 2] * upr_readlane(rq, 4 * (m + 2)); } \
                    wv = ((v0 + v1) + v2) + pk; \
                    if (act) Wk((kk) - 1)[i] = wv; } while (0)
                // column (b, j) of A' without branches: coefficients of w[(a, j)], zero for a > b
                const double ca0 = coefA(0, b), ca1 = (b >= 1) ? coefA(1, b) : 0.0, ca2 = (b >= 2) ? 1.0 : 0.0;
                const double cbq = act ? coefB(b) : 0.0;
#pragma unroll
                for (int st = 0; st < UPR_QP3_KSTAGES; ++st) {
#pragma unroll
                    for (int m = 0; m < NQ; ++m) kc[st][m] = 0.0;
                    UPR_LOADKC(kc[st], N - 1 - st);
                }
                double wv = Wk(N - 1)[i];
                double gs[2] = {0.0, 0.0}, ps[2] = {0.0, 0.0}, us2[2] = {0.0, 0.0};
                UPR_LOADST(gs[(N - 1) & 1], ps[(N - 1) & 1], us2[(N - 1) & 1], N - 1);
                // fully unrolled (static register indices; inside a loop the compiler's s_waitcnt bookkeeping would also
                // collapse the prefetch stages into one).  KFAR (N = 100): unrolled in groups of KSTAGES knots inside a rolled loop --
                // the stage indices are still static (the group length is the ring's), and the hundred knots' global addresses are
                // formed where they are used: fully unrolled the compiler hoisted and SPILLED them (1 400 scratch instructions inside
                // the two sweeps, every reload a vmcnt(0): 8 k cycles a knot)
                constexpr int UG = C::KFAR ? UPR_QP3_KSTAGES : N - 1;
                static_assert(!C::KFAR || UPR_QP3_KSTAGES % 2 == 0, "the two operand slots keep their parity over a group");
#pragma unroll 1
                for (int k0 = N - 1; k0 >= 1; k0 -= UG) {
#pragma unroll
                    for (int sg = 0; sg < UG; ++sg) {
                        const int k = k0 - sg;
                        if (k >= 1) {
                            UPR_LOADST(gs[(N - sg) & 1], ps[(N - sg) & 1], us2[(N - sg) & 1], k - 1);
                            UPR_VECSTEP(kc[sg % UPR_QP3_KSTAGES], gs[(N - 1 - sg) & 1], ps[(N - 1 - sg) & 1], us2[(N - 1 - sg) & 1], k);
                            UPR_LOADKC(kc[sg % UPR_QP3_KSTAGES], k - UPR_QP3_KSTAGES);
                        }
                    }
                }
#undef UPR_LOADKC
#undef UPR_LOADST
#undef UPR_VECSTEP
            }
#else
            for (int k = N - 1; k >= 1; --k) {
                const double* w = Wk(k);
                UPR_FORT(i, NX) {
                    const int b = i / NQ, j = i % NQ;
                    double v = LK[O::gxs + k * NX + i] + LK[O::cs + k * NX + i];
                    for (int a = 0; a <= b; ++a) v += coefA(a, b) * w[a * NQ + j];
                    for (int m = 0; m < NQ; ++m) v -= G[F::Ks + k * NQ * NX + m * NX + i] * (LK[O::gus + k * NU + m] + h3 * w[m] + h2 * w[NQ + m] + h * w[2 * NQ + m]);
                    Wk(k - 1)[i] = v + LK[O::Pbs + (k - 1) * NX + i];
                }
            }
#endif
            UPR_SETPRIO(0);
        }
#ifndef UPR_HOST_EMU
        // (round 6) the corrector's call: the feed-forward phase below runs on the first lanes of WAVE 1, which request their knot's
        // Lj^-1 (45 values out of the far arrays, stored by the matrix sweep) here, while wave 0 runs the sweep -- on wave 0 those
        // requests went out behind the sweep and the phase waited one far round trip for them
        constexpr bool KFFW1 = SW2 && N <= 64 && NT >= 128;
        const bool kfw = KFFW1 && !fused;
        double liP[KFFW1 ? NQ : 1][KFFW1 ? NQ : 1];
        if (kfw && wb == 64) {
            const int k = (lane() < N) ? lane() : 0;
            const double* Li = G + F::Ljis + k * NQ * NX;
#pragma unroll
            for (int i = 0; i < NQ; ++i)
#pragma unroll
                for (int m = 0; m <= i; ++m) liP[i % (KFFW1 ? NQ : 1)][m % (KFFW1 ? NQ : 1)] = Li[i * NX + m];
        }
#endif
        UPR_SYNC();
        toc(10);
        // feed-forward of every knot: kff = Lj^-T (Lj^-1 huj) by substitution with the packed factor
#ifndef UPR_HOST_EMU
        if constexpr (SW2) {
            // (two-wave matrix sweep: the store holds the dense inverse factor Lj^-1 -- two triangular products, a lane per knot.  A row
            //  per lane with Lj^-1 staged in LDS was measured 1.6 - 4.8 % slower on the headline launch: DESIGN.md section 7)
            for (int k = kfw ? tid() - 64 : tid(); k >= 0 && k < N; k += kfw ? N : stride()) {
                const double* Li = G + F::Ljis + k * NQ * NX;
                const double* w = Wk(k);
                double li[NQ][NQ], tv[NQ], y[NQ];
#pragma unroll
                for (int i = 0; i < NQ; ++i)
#pragma unroll
                    for (int m = 0; m <= i; ++m) li[i][m] = kfw ? liP[i % (KFFW1 ? NQ : 1)][m % (KFFW1 ? NQ : 1)] : Li[i * NX + m];
#pragma unroll
                for (int i = 0; i < NQ; ++i) tv[i] = fused ? LK[O::kffs + k * NQ + i] : LK[O::gus + k * NU + i] + h3 * w[i] + h2 * w[NQ + i] + h * w[2 * NQ + i];
#pragma unroll
                for (int i = 0; i < NQ; ++i) { double t = 0.0;
#pragma unroll
                    for (int m = 0; m <= i; ++m) t += li[i][m] * tv[m];
                    y[i] = t; }
#pragma unroll
                for (int i = 0; i < NQ; ++i) { double t = 0.0;
#pragma unroll
                    for (int m = i; m < NQ; ++m) t += li[m][i] * y[m];
                    LK[O::kffs + k * NQ + i] = t; }
            }
        } else
#endif
        UPR_FORT(k, N) {
            const double* Lp = G + F::Ljis + k * C::NH;
            const double* w = Wk(k);
            double y[NQ], kk[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) { double t = LK[O::gus + k * NU + i] + h3 * w[i] + h2 * w[NQ + i] + h * w[2 * NQ + i];
#pragma unroll
                for (int m = 0; m < i; ++m) t -= Lp[i * (i + 1) / 2 + m] * y[m];
                y[i] = t * Lp[i * (i + 1) / 2 + i]; }
#pragma unroll
            for (int i = NQ - 1; i >= 0; --i) { double t = y[i];
#pragma unroll
                for (int m = i + 1; m < NQ; ++m) t -= Lp[m * (m + 1) / 2 + i] * kk[m];
                kk[i] = t * Lp[i * (i + 1) / 2 + i]; }
#pragma unroll
            for (int i = 0; i < NQ; ++i) LK[O::kffs + k * NQ + i] = kk[i];
        }
        UPR_SYNC();
    }

    // forward sweep, closed-loop form  sx+ = A sx + b - B (K sx + kff): one wave-local phase per knot, lane
    // (block b, joint j) owns sx+[b nq + j] and row j of K_k (prefetched one knot ahead)
    // COST: the corrector's pass -- the costates of the full step follow the tail (their rows of C and of the end-effector
    // Hessian are fetched by the idle waves as well)
    template <bool COST>
    UPR_HDI void forward() {
#ifndef UPR_HOST_EMU
        // The flat tail's global data (rows of C, the Schur factors, the contact factors and yf) are constants of this
        // call: the waves that idle during the serial sweep fetch them into registers meanwhile, and the tail runs on
        // those waves (lane index tl) out of registers and LDS.
        constexpr int NTL = NT - 64, CH = (NX + 3) / 4, QV = (N * NE * 4 + NTL - 1) / NTL, QCT = (C::NCI + NTL - 1) / NTL;
        static_assert((C::KFAR || C::NKB <= NTL) && NTL % 4 == 0, "tail lanes: one per Schur block (KFAR: a loop over the blocks)");
        const int tl = tid() - 64;
        constexpr int QCS = (N * NX + NTL - 1) / NTL;
        constexpr int QH = (N * NQ + NTL - 1) / NTL;
        // (register prefetch of the rows of C only where they fit: the multi-body shapes read them behind the sweep)
        constexpr bool PRE_V = QV <= UPR_QP3_PREV_MAX, PRE_K = QCS * NE <= 24;
        constexpr int SB = C::SB;
        // DIST (dense Schur complements, SB = 6 NB > 6): the two triangular products of a block are dealt out a ROW per lane with the
        // factor's entries read where they are used.  One lane per block out of a register copy of the factor fetched during the
        // sweep, as the star shapes do it, meant 171 registers for 20 lanes of one wave -- the compiler parked them in scratch
        // (150 stores, 46 reloads per lane), and that wave, not the sweep, was what the barrier behind the sweep waited for.
        constexpr bool DIST = SB > 6;
        double ckq[PRE_V ? QV : 1][CH], lsr[DIST ? 1 : SB * SB], bkq[QCT][NF == 3 ? 9 : 1], yfq[QCT][NF == 3 ? 3 : 1];
        double heeq[QH][NQ], ckc[PRE_K ? QCS : 1][PRE_K ? NE : 1];
#endif
        if (wave0()) {
            UPR_SETPRIO(UPR_QP3_PRIO_FWD);
            // knot 0: sx_0 = 0
            UPR_FORT(i, NX) {
                const int b = i / NQ, j = i % NQ;
                const double uj = -LK[O::kffs + j];
                Sx(0)[i] = 0.0;
                Sx(1)[i] = coefB(b) * uj + LK[O::bks + i];
                if (b == 0) Su(0)[j] = uj;
            }
            UPR_WSYNC();
#ifndef UPR_HOST_EMU
            // lane 4 j + b (b < 3) owns x[(b, j)] and carries it in a REGISTER from knot to knot, together with the nq
            // entries K_k[j][b nq ..] (three knots of register prefetch).  Per knot the lane fetches the nq entries of its
            // block from the lanes that own them (ds_bpermute: the LDS crossbar without the memory -- broadcasting all
            // 27 entries through v_readlane was measured slower than the LDS round trip it replaces),
            // the three partial dot products of a joint are summed inside its quad by DPP, and the row of
            // A is quad-local (DPP broadcasts).  x is written to LDS for the later phases, never read back here.
            static_assert(4 * NQ <= 64, "one quad per joint");
            {
                const int l = lane();
                const bool act = ((l & 3) < 3) && ((l >> 2) < NQ);
                const int j = act ? (l >> 2) : 0, b = act ? (l & 3) : 3;
                const int i = (act ? b : 0) * NQ + j;
                double kq[UPR_QP3_KSTAGES][NQ];
#define UPR_LOADKQ(dst, kk) do { if (act && (kk) < N) { _Pragma("unroll") for (int c = 0; c < NQ; ++c) dst[c] = G[F::Ks + (kk) * NQ * NX + j * NX + b * NQ + c]; } } while (0)
#define UPR_LOADST(bk, fk, kk) do { if ((kk) < N) { bk = LK[O::bks + (kk) * NX + i]; fk = LK[O::kffs + (kk) * NQ + j]; } } while (0)
#define UPR_FWDSTEP(kqx, bk, fk, kk) do { \
                    double xs[NQ]; \
                    _Pragma("unroll") for (int c = 0; c < NQ; ++c) xs[c] = upr_bpermute(xv, src + 16 * c); \
                    __builtin_amdgcn_sched_barrier(0);   /* all permutes in flight before the first product */ \
                    double d0 = (b == 0) ? fk : 0.0, d1 = 0.0, d2 = 0.0; \
                    _Pragma("unroll") for (int c = 0; c < NQ; c += 3) { \
                        d0 += kqx[c] * xs[c]; \
                        if (c + 1 < NQ) d1 += kqx[c + 1] * xs[c + 1]; \
                        if (c + 2 < NQ) d2 += kqx[c + 2] * xs[c + 2]; } \
                    double d = (d0 + d1) + d2;   /* (idle lanes: K = 0 and a finite x, exactly 0) */ \
                    d += upr_dpp_quad<0xB1>(d); d += upr_dpp_quad<0x4E>(d); \
                    const double x0 = upr_dpp_quad<0x00>(xv), x1 = upr_dpp_quad<0x55>(xv), x2 = upr_dpp_quad<0xAA>(xv); \
                    xv = ((bk + ra0 * x0) + (ra1 * x1 + ra2 * x2)) - cb * d; \
                    if (act) { Sx((kk) + 1)[i] = xv; if (b == 0) Su(kk)[j] = -d; } } while (0)
                // row (b, j) of A and entry b of B without branches
                const double ra0 = (b == 0) ? 1.0 : 0.0, ra1 = (b == 0) ? h : ((b == 1) ? 1.0 : 0.0), ra2 = (b == 0) ? h2 : ((b == 1) ? h : ((b == 2) ? 1.0 : 0.0));
                const double cb = act ? coefB(b) : 0.0;
                const int src = 4 * (l & 3);   // byte address of lane (b, 0) for ds_bpermute: lane (b, c) is 16 c further
#pragma unroll
                for (int st = 0; st < UPR_QP3_KSTAGES; ++st) {
#pragma unroll
                    for (int c = 0; c < NQ; ++c) kq[st][c] = 0.0;
                    UPR_LOADKQ(kq[st], 1 + st);
                }
                double xv = Sx(1)[i];
                double bs[2] = {0.0, 0.0}, fs[2] = {0.0, 0.0};
                UPR_LOADST(bs[1], fs[1], 1);
                constexpr int UG = C::KFAR ? UPR_QP3_KSTAGES : N - 1;   // (as in the vector sweep)
#pragma unroll 1
                for (int k0 = 1; k0 < N; k0 += UG) {
#pragma unroll
                    for (int sg = 0; sg < UG; ++sg) {
                        const int k = k0 + sg;
                        if (k < N) {
                            UPR_LOADST(bs[(sg + 2) & 1], fs[(sg + 2) & 1], k + 1);
                            UPR_FWDSTEP(kq[sg % UPR_QP3_KSTAGES], bs[(sg + 1) & 1], fs[(sg + 1) & 1], k);
                            UPR_LOADKQ(kq[sg % UPR_QP3_KSTAGES], k + UPR_QP3_KSTAGES);
                        }
                    }
                }
#undef UPR_LOADKQ
#undef UPR_LOADST
#undef UPR_FWDSTEP
            }
#else
            for (int k = 1; k < N; ++k) {
                const double* sx = Sx(k);
                UPR_FORT(i, NX) {
                    const int b = i / NQ, j = i % NQ;
                    const double* Kr = G + F::Ks + k * NQ * NX + j * NX;
                    double d0 = LK[O::kffs + k * NQ + j], d1 = 0.0, d2 = 0.0;
                    for (int c = 0; c < NQ; ++c) { d0 += Kr[c] * sx[c]; d1 += Kr[NQ + c] * sx[NQ + c]; d2 += Kr[2 * NQ + c] * sx[2 * NQ + c]; }
                    const double uj = -(d0 + d1 + d2);
                    double r = coefB(b) * uj + LK[O::bks + k * NX + i];
                    for (int a = b; a < 3; ++a) r += coefA(b, a) * sx[a * NQ + j];
                    Sx(k + 1)[i] = r;
                    if (b == 0) Su(k)[j] = uj;
                }
            }
#endif
            UPR_SETPRIO(0);
#ifndef UPR_HOST_EMU
            if (!ROWMEM) load_rows();   // (wave 0: in flight while the other waves run the tail)
#endif
        }
#ifndef UPR_HOST_EMU
        else {
            if (!ROWMEM) load_rows();
            if (PRE_V) {
#pragma unroll
                for (int q = 0; q < QV; ++q) {
                    const int e4 = tl + q * NTL;
                    if (e4 < N * NE * 4) {
                        const int e = e4 >> 2, part = e4 & 3;
                        const double* Ck = rec(e / NE) + lin_gx + (e % NE) * NX + part * CH;
#pragma unroll
                        for (int c = 0; c < CH; ++c) ckq[q % (PRE_V ? QV : 1)][c] = (part * CH + c < NX) ? Ck[c] : 0.0;
                    }
                }
            }
            if (!DIST && !C::KFAR && tl < C::NKB) {
#pragma unroll
                for (int r = 0; r < SB; ++r)
#pragma unroll
                    for (int m = 0; m <= r; ++m) lsr[r * SB + m] = G[F::lsi + tl * SB * SB + r * SB + m];
            }
            if (COST) {
#pragma unroll
                for (int q = 0; q < QH; ++q) {
                    const int e = tl + q * NTL;
                    if (e < N * NQ) {
                        const int k = e / NQ, i = e % NQ;
#pragma unroll
                        for (int j = 0; j < NQ; ++j) heeq[q][j] = G[hee_w + k * C::NH + upr_tri(NQ, i, j)];
                    }
                }
                if (PRE_K) {
#pragma unroll
                    for (int q = 0; q < QCS; ++q) {
                        const int e = tl + q * NTL;
                        if (e < N * NX) {
                            const double* Ck = rec(e / NX) + lin_gx + e % NX;
#pragma unroll
                            for (int r = 0; r < NE; ++r) ckc[q % (PRE_K ? QCS : 1)][r % (PRE_K ? NE : 1)] = Ck[r * NX];
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < QCT; ++q) {
                const int ic = tl + q * NTL;
                if (ic < C::NCI) {
                    const int k = ic / NC, ci = ic % NC;
                    if (NF == 3) {
#pragma unroll
                        for (int a = 0; a < 9; ++a) bkq[q][a] = G[F::lfi + k * C::NLF + 9 * ci + a];
#pragma unroll
                        for (int a = 0; a < 3; ++a) yfq[q][a] = G[F::yf + k * NFC + 3 * ci + a];
                    } else { bkq[q][0] = G[F::lfi + k * C::NLF + ci]; yfq[q][0] = G[F::yf + k * NFC + ci]; }
                }
            }
        }
#endif
        ftoc(6, 5);   // (-DUPR_QP3_PROF_FLAT=5: slot 6 = each wave's own time up to the barrier behind the forward sweep, 'fwd: sweep' then its wait)
        UPR_SYNC();
        toc(12);
        // flat: cv = C sx ; nu+ = Lsi'(Lsi cv + ys) ; su_f = -Lfi'(yf + Lfi Df' nu+) ; terminal multiplier step
#ifndef UPR_HOST_EMU
        // cv: a quad per row of C (four column chunks), summed by DPP
        if (PRE_V) {
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                const int e4 = tl + q * NTL;
                const bool act = tl >= 0 && e4 < N * NE * 4;
                const int e = act ? (e4 >> 2) : 0, part = e4 & 3;
                double v = 0.0;
                {
                    // (no test per column and none on `act`: a test puts each LDS read into a branch of its own with a full round
                    //  trip -- seven in a row per job.  The coefficients beyond the row's end are zero, what is read there is the
                    //  next knot's first entries: exact zeros are added; lanes without a job compute on knot 0 and store nothing)
                    const double* sx = Sx(e / NE) + part * CH;
                    double sv[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) sv[c] = sx[c];
#pragma unroll
                    for (int c = 0; c < CH; ++c) v += (act ? ckq[q % (PRE_V ? QV : 1)][c] : 0.0) * sv[c];
                }
                v += upr_dpp_quad<0xB1>(v); v += upr_dpp_quad<0x4E>(v);
                if (act && part == 0) { LK[O::cv + e] = v; if (C::INCEK && COST) LK[O::dek + e] = v; }
            }
        } else {
            constexpr int GQ = UPR_QP3_GQ;   // (as in prep: GQ rows of a lane requested together)
#pragma unroll 1
            for (int q0 = 0; q0 < QV; q0 += GQ) {
                double cb[GQ][CH];
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    const int e4 = tl + (q0 + g) * NTL;
                    const bool act = tl >= 0 && (q0 + g < QV) && e4 < N * NE * 4;
                    const int e = act ? (e4 >> 2) : 0, part = e4 & 3;
                    const double* Ck = rec(e / NE) + lin_gx + (e % NE) * NX + part * CH;
#pragma unroll
                    for (int c = 0; c < CH; ++c) cb[g][c] = (act && part * CH + c < NX) ? Ck[c] : 0.0;
                }
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    const int e4 = tl + (q0 + g) * NTL;
                    const bool act = tl >= 0 && (q0 + g < QV) && e4 < N * NE * 4;
                    const int e = act ? (e4 >> 2) : 0, part = e4 & 3;
                    const double* sx = Sx(e / NE) + part * CH;
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < CH; ++c) v += cb[g][c] * ((part * CH + c < NX) ? sx[c] : 0.0);
                    v += upr_dpp_quad<0xB1>(v); v += upr_dpp_quad<0x4E>(v);
                    if (act && part == 0) { LK[O::cv + e] = v; if (C::INCEK && COST) LK[O::dek + e] = v; }
                }
            }
        }
        if (COST) {   // end-effector Hessian part of the costates (the gee slot is free after the corrector's prep)
#pragma unroll
            for (int q = 0; q < QH; ++q) {
                const int e = tl + q * NTL;
                if (tl >= 0 && e < N * NQ) {
                    const double* sx = Sx(e / NQ);
                    double v = 0.0;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) v += heeq[q][j] * sx[j];
                    LK[O::gee + e] = h * v;
                }
            }
        }
        sync_lds();
        // nu+ of a knot by its lane: in place over cv (LDS, what the contact step and the costates read) and to global
        if (DIST) {
            constexpr int NROW = C::NKB * SB, QD = (NROW + NTL - 1) / NTL;
            double tv1[QD];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int q = 0; q < QD; ++q) {
                    const int e = tl + q * NTL;
                    if (tl >= 0 && e < NROW) {
                        const int kb = e / SB, r = e % SB;
                        const double* Ls = G + F::lsi + kb * SB * SB;
                        double lv[SB], xv[SB];
#pragma unroll
                        for (int m = 0; m < SB; ++m) { lv[m] = (pass == 0) ? Ls[r * SB + m] : Ls[m * SB + r]; xv[m] = LK[O::cv + kb * SB + m]; }
                        double v = (pass == 0) ? LK[O::ys + e] : 0.0;
#pragma unroll
                        for (int m = 0; m < SB; ++m) v += (((pass == 0) ? (m <= r) : (m >= r)) ? lv[m] : 0.0) * xv[m];
                        tv1[q] = v;
                    }
                }
                sync_lds();   // (every row of the block has read cv / the first product)
#pragma unroll
                for (int q = 0; q < QD; ++q) {
                    const int e = tl + q * NTL;
                    if (tl >= 0 && e < NROW) { LK[O::cv + e] = tv1[q]; if (pass == 1) G[F::nun + e] = tv1[q]; }
                }
                if (pass == 0) sync_lds();
            }
        } else
        for (int kb = tl; kb >= 0 && kb < C::NKB; kb += NTL) {   // Schur block: a knot, or a (knot, body) pair (one per lane; KFAR: several, its factor fetched here)
            if (C::KFAR) {
#pragma unroll
                for (int r = 0; r < SB; ++r)
#pragma unroll
                    for (int m = 0; m <= r; ++m) lsr[r * SB + m] = G[F::lsi + kb * SB * SB + r * SB + m];
            }
            double cvr[SB], t1[SB];
#pragma unroll
            for (int r = 0; r < SB; ++r) cvr[r] = LK[O::cv + kb * SB + r];
#pragma unroll
            for (int r = 0; r < SB; ++r) { double v = LK[O::ys + kb * SB + r];
#pragma unroll
                for (int m = 0; m <= r; ++m) v += lsr[r * SB + m] * cvr[m];
                t1[r] = v; }
#pragma unroll
            for (int r = 0; r < SB; ++r) { double v = 0.0;
#pragma unroll
                for (int m = r; m < SB; ++m) v += lsr[m * SB + r] * t1[m];
                LK[O::cv + kb * SB + r] = v; G[F::nun + kb * SB + r] = v; }
        }
        sync_lds();
#pragma unroll
        for (int q = 0; q < QCT; ++q) {
            const int ic = tl + q * NTL;
            if (tl >= 0 && ic < C::NCI) {
                const int k = ic / NC, ci = ic % NC;
                if (NF == 3) {
                    double dfn[3], tf[3];
                    const int rb = (C::NB > 1) ? 6 * P->contact_body2[ci] : 0;   // the six rows of the body this contact loads
#pragma unroll
                    for (int a = 0; a < 3; ++a) { double v = 0.0;
#pragma unroll
                        for (int r = 0; r < 6; ++r) v += DF(rb + r, 3 * ci + a) * LK[O::cv + k * NE + rb + r];
                        dfn[a] = v; }
                    if (C::COUPLED && P->contact_body1[ci] >= 0) {   // ... and of the body underneath
                        const int rb1 = 6 * P->contact_body1[ci];
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int r = 0; r < 6; ++r) dfn[a] += L[O::df + (rb1 + r) * NFC + 3 * ci + a] * LK[O::cv + k * NE + rb1 + r];
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) { double v = yfq[q][a];
#pragma unroll
                        for (int b2 = 0; b2 <= a; ++b2) v += bkq[q][3 * a + b2] * dfn[b2];
                        tf[a] = v; }
#pragma unroll
                    for (int a = 0; a < 3; ++a) { double v = 0.0;
#pragma unroll
                        for (int b2 = a; b2 < 3; ++b2) v += bkq[q][3 * b2 + a] * tf[b2];
                        Su(k)[NQ + 3 * ci + a] = -v; }
                } else {
                    double dfn = 0.0;
                    const int rb = (C::NB > 1) ? 6 * P->contact_body2[ci] : 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) dfn += L[O::df + (rb + r) * NFC + ci] * LK[O::cv + k * NE + rb + r];
                    if (C::COUPLED && P->contact_body1[ci] >= 0) {
                        const int rb1 = 6 * P->contact_body1[ci];
#pragma unroll
                        for (int r = 0; r < 6; ++r) dfn += L[O::df + (rb1 + r) * NFC + ci] * LK[O::cv + k * NE + rb1 + r];
                    }
                    const double lf = bkq[q][0];
                    Su(k)[NQ + ci] = -lf * (yfq[q][0] + lf * dfn);
                }
            }
        }
#else
        UPR_FORT(e, N * NE) {
            const int k = e / NE, r = e % NE;
            const double* Ck = rec(k) + lin_gx + r * NX; const double* sx = Sx(k);
            double v = 0.0;
            for (int c = 0; c < NX; ++c) v += Ck[c] * sx[c];
            LK[O::cv + e] = v;
            if (C::INCEK && COST) LK[O::dek + e] = v;
        }
        UPR_SYNC();
        UPR_FORT(kb, C::NKB) {
            constexpr int SB = C::SB;
            const double* Ls = G + F::lsi + kb * SB * SB;
            double t1[SB];
#pragma unroll
            for (int r = 0; r < SB; ++r) { double v = LK[O::ys + kb * SB + r];
#pragma unroll
                for (int m = 0; m <= r; ++m) v += Ls[r * SB + m] * LK[O::cv + kb * SB + m];
                t1[r] = v; }
#pragma unroll
            for (int r = 0; r < SB; ++r) { double v = 0.0;
#pragma unroll
                for (int m = r; m < SB; ++m) v += Ls[m * SB + r] * t1[m];
                LK[O::cv + kb * SB + r] = v; G[F::nun + kb * SB + r] = v; }
        }
        UPR_SYNC();
        for (int q = 0; q < C::QC; ++q) {
            const int ic = tid() + q * NT;
            if (ic < C::NCI) {
                const int k = ic / NC, ci = ic % NC;
                if (NF == 3) {
                    double dfn[3], tf[3];
                    for (int a = 0; a < 3; ++a) { double v = 0.0;
                        if (C::BIGF) { const int rb = 6 * P->contact_body2[ci]; for (int r = 0; r < 6; ++r) v += DF(rb + r, 3 * ci + a) * G[F::nun + k * NE + rb + r]; }
                        else for (int r = 0; r < NE; ++r) v += L[O::df + r * NFC + 3 * ci + a] * G[F::nun + k * NE + r];
                        dfn[a] = v; }
                    const double* Bk = G + F::lfi + k * C::NLF + 9 * ci;
                    for (int a = 0; a < 3; ++a) { double v = G[F::yf + k * NFC + 3 * ci + a]; for (int b2 = 0; b2 <= a; ++b2) v += Bk[3 * a + b2] * dfn[b2]; tf[a] = v; }
                    for (int a = 0; a < 3; ++a) { double v = 0.0; for (int b2 = a; b2 < 3; ++b2) v += Bk[3 * b2 + a] * tf[b2]; Su(k)[NQ + 3 * ci + a] = -v; }
                } else {
                    double dfn = 0.0;
                    for (int r = 0; r < NE; ++r) dfn += L[O::df + r * NFC + ci] * G[F::nun + k * NE + r];
                    const double lf = G[F::lfi + k * C::NLF + ci];
                    Su(k)[NQ + ci] = -lf * (G[F::yf + k * NFC + ci] + lf * dfn);
                }
            }
        }
#endif
        if (neN > 0) UPR_FORT(q, C::NEN) {
            double v;
            if (q < 3) { v = L[O::eN + q]; for (int j = 0; j < NQ; ++j) v -= L[O::jN + q * NQ + j] * Sx(N)[j]; }
            else v = L[O::eN + q] + Sx(N)[NQ + (q - 3)];
            L[O::dyN + q] = v / UPR_QP_RHO_N;
        }
#ifndef UPR_HOST_EMU
        sync_lds();
        if (COST) {
            // costates of the full step, staged where the sweeps keep P (LDS; the update reads them there)
            double* pin = LK + O::pin;
#pragma unroll
            for (int q = 0; q < QCS; ++q) {
                const int e = tl + q * NTL;
                if (tl >= 0 && e < N * NX) {
                    const int k = e / NX, i = e % NX;
                    const double sxi = Sx(k)[i];
                    double v = LK[O::gxs + e] + LK[O::wx + e] * sxi + h * L[O::qd + i] * sxi;
                    if (i < NQ) v += LK[O::gee + k * NQ + i];
                    if (PRE_K) {
#pragma unroll
                        for (int r = 0; r < NE; ++r) v += ckc[q % (PRE_K ? QCS : 1)][r % (PRE_K ? NE : 1)] * LK[O::cv + k * NE + r];
                    } else {
                        const double* Ck = rec(k) + lin_gx + i;
                        double cr[NE];
#pragma unroll
                        for (int r = 0; r < NE; ++r) cr[r] = Ck[r * NX];
#pragma unroll
                        for (int r = 0; r < NE; ++r) v += cr[r] * LK[O::cv + k * NE + r];
                    }
                    pin[e] = v;
                }
            }
            if (tl < 0 && tl + 64 < NX) {
                const int i = tl + 64, e = N * NX + i;
                double v = LK[O::gxs + e] + LK[O::wx + e] * Sx(N)[i];
                if (neN > 0) {
                    if (i < NQ) { for (int q = 0; q < 3; ++q) v -= L[O::jN + q * NQ + i] * (L[O::yN + q] + L[O::dyN + q]); }
                    else v += L[O::yN + 3 + (i - NQ)] + L[O::dyN + 3 + (i - NQ)];
                }
                pin[e] = v;
            }
            sync_lds();
            costate_sums();
        }
#else
        UPR_SYNC();
        if (COST) costates();
#endif
    }

    // pi_k = r_k + A' pi_{k+1}: three running sums per joint, in place (LDS)
    UPR_HDI void costate_sums() {
        double* pin = LK + O::pin;
        UPR_FORT(j, NQ) {
            double pq = pin[N * NX + j], pvv = pin[N * NX + NQ + j], pa = pin[N * NX + 2 * NQ + j];
#pragma unroll
            for (int k = N - 1; k >= 1; --k) {
                const double nq_ = pin[k * NX + j] + pq;
                const double nv_ = pin[k * NX + NQ + j] + h * pq + pvv;
                const double na_ = pin[k * NX + 2 * NQ + j] + h2 * pq + h * pvv + pa;
                pq = nq_; pvv = nv_; pa = na_;
                pin[k * NX + j] = pq; pin[k * NX + NQ + j] = pvv; pin[k * NX + 2 * NQ + j] = pa;
            }
        }
        sync_lds();
    }

    // costates of the full step (host emulation: the same sums out of global memory)
    UPR_HDI void costates() {
        double* pin = LK + O::pin;
        UPR_FORT(e, N1 * NX) {
            const int k = e / NX, i = e % NX;
            const double* sx = Sx(k);
            double v = LK[O::gxs + e] + LK[O::wx + e] * sx[i];
            if (k < N) {
                v += h * L[O::qd + i] * sx[i];
                if (i < NQ) for (int j = 0; j < NQ; ++j) v += h * G[hee_w + k * C::NH + upr_tri(NQ, i, j)] * sx[j];
                const double* Ck = rec(k) + lin_gx;
                for (int q = 0; q < NE; ++q) v += Ck[q * NX + i] * LK[O::cv + k * NE + q];
            } else if (neN > 0) {
                if (i < NQ) { for (int q = 0; q < 3; ++q) v -= L[O::jN + q * NQ + i] * (L[O::yN + q] + L[O::dyN + q]); }
                else v += L[O::yN + 3 + (i - NQ)] + L[O::dyN + 3 + (i - NQ)];
            }
            pin[e] = v;
        }
        UPR_SYNC();
        costate_sums();
    }

    // ---- sweeps over the lane-owned rows with the current step ------------------------------------------------
    //   what 0: alpha_max partial (aux += dt dlam) ; 2: apply ; 3: partial max |rp|, aux += lam t ; 4: 2 then 3 at the new iterate ;
    //   5: partial MIN of the trial products (lam + a dlam)(t + a dt), aux += their sum (the centrality safeguard, UPR_QP_NGAM)
    UPR_HDI void sweep_row(int what, double alpha, double c, double ds, double& t, double& lam, double cterm, double& acc, double* aux) const {
        const double rp = c - t;
        if (what == 3) { const double a = fabs(rp); if (a > acc) acc = a; *aux += lam * t; return; }
        const double dt = ds + rp;
        const double rc = (mode == 0) ? lam * t : lam * t + cterm;
        const double rt = upr_rcp(t);
        const double dl = -(rc + lam * dt) * rt;
        if (what == 0) {
            // acc carries 1 / alpha_max: the row's limits are t / -dt and lam / -dlam (no division, no branch)
            acc = fmax(acc, fmax(-dt * rt, -dl * upr_rcp(lam)));
            if (aux) *aux += dt * dl;   // (predictor: the second-order term of the complementarity average, see solve())
        }
        else if (what == 5) { const double v = (lam + alpha * dl) * (t + alpha * dt); acc = fmin(acc, v); *aux += v; }
        else { t += alpha * dt; lam += alpha * dl; }
    }
    // softened row (see row_soft): what 3 also folds the slack pair into the residuals (|sigma - tau| into acc, gam tau into
    // aux, the slack stationarity into soft_stat)
    UPR_HDI void sweep_row_soft(int what, double alpha, double c, double ds, double& t, double& lam, double& sig, double& tau, double& gam,
                                double Zp, double zp, double cterm, double cterm_s, double& acc, double* aux) {
        const double rp = c + sig - t, rps = sig - tau;
        if (what == 3) {
            const double a = fmax(fabs(rp), fabs(rps)); if (a > acc) acc = a;
            *aux += lam * t + gam * tau;
            soft_stat = fmax(soft_stat, fabs(Zp * sig + zp - lam - gam));
            return;
        }
        const double rc = (mode == 0) ? lam * t : lam * t + cterm, rcs = (mode == 0) ? gam * tau : gam * tau + cterm_s;
        const double rt = upr_rcp(t), rtau = upr_rcp(tau);
        const double w0 = lam * rt, ws_ = gam * rtau, rD = upr_rcp(Zp + w0 + ws_);
        const double a = (Zp * sig + zp - lam - gam) + (rc + lam * rp) * rt + (rcs + gam * rps) * rtau;
        const double dsg = -(a + w0 * ds) * rD;
        const double dt = ds + rp + dsg, dtau = dsg + rps;
        const double dl = -(rc + lam * dt) * rt, dg = -(rcs + gam * dtau) * rtau;
        if (what == 0) { acc = fmax(fmax(acc, fmax(-dt * rt, -dl * upr_rcp(lam))), fmax(-dtau * rtau, -dg * upr_rcp(gam))); if (aux) *aux += dt * dl + dtau * dg; }
        else if (what == 5) { const double v = (lam + alpha * dl) * (t + alpha * dt), vs = (gam + alpha * dg) * (tau + alpha * dtau); acc = fmin(acc, fmin(v, vs)); *aux += v + vs; }
        else { t += alpha * dt; lam += alpha * dl; sig += alpha * dsg; tau += alpha * dtau; gam += alpha * dg; }
    }
    UPR_HDI double ineq_sweep(int what, double alpha, double* aux, const double (&ctm)[NCTR], bool reload = true) {
        ftoc(10, 4);
        if (ROWMEM && reload) load_rows();   // (reload == false: the rows are still those of the sweep just before)
        double acc = (what == 5) ? 1e300 : 0.0;
        const int tid_ = tid();
        // the far-array operands of the friction rows (one row per lane item) and of the state-polytopic rows, each class requested
        // together (SOFT: with the rows' slack pairs and the pairs' corrector targets).  Requested in front of the box rows instead,
        // so that the box rows cover the round trip, measured no gain on the hard kernels and 2 % slower on the SOFT ones.
        constexpr int NR5 = 5 * C::NCI, QR5 = (NR5 + NT - 1) / NT, QS5 = C::SOFT ? QR5 : 1;
        double tv[QR5], lv[QR5], cv5[QR5], ssv[QS5][4];
        auto fetch_fr = [&]() {
#pragma unroll
            for (int q = 0; q < QR5; ++q) {
                const int e = tid_ + q * NT, ec = (e < NR5) ? e : 0;
                tv[q] = G[F::ct + ec]; lv[q] = G[F::cl + ec]; cv5[q] = G[F::cc + ec];
                if (C::SOFT && softp) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) ssv[q % QS5][u] = G[F::sfr + u * F::sfs + ec];
                }
            }
        };
        constexpr int QOS = (C::SOFT && C::ROWS) ? QO : 1;
        double gq[QO][NQ], od[QO], otv[QO], olv[QO], ocv[QO], osv[QOS][4];
        auto fetch_or = [&]() {
#pragma unroll
            for (int q = 0; q < QO; ++q) {
                const int e = tid_ + q * NT, ec = (e < (N - 1) * no) ? e : 0;
                const int k = 1 + ec / no, r = ec % no, ei = (k - 1) * UPR_QP3_NOMAX + r;
                const double* g = orow(k, r);
#pragma unroll
                for (int i = 0; i < NQ; ++i) gq[q][i] = g[i];
                od[q] = G[F::od0 + ei]; otv[q] = G[F::ot + ei]; olv[q] = G[F::ol + ei]; ocv[q] = G[F::oc + ei];
                if (C::SOFT && C::ROWS && softp) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) osv[q % QOS][u] = G[F::sor + u * F::sos + ei];
                }
            }
        };
        // what == 4: the step of what == 2 and then, with the row's value cn at the NEW iterate, the residual terms of what == 3
        // (|cn - t| into acc, lam t into aux) -- what the next iteration's first residual pass would compute from the updated
        // iterate.  cn is formed exactly as that pass forms it: the new primal value z + alpha dz first (upr_step, the same
        // operation as the update of the iterate in solve()), then the row.
        const bool w4 = what == 4;
        const int wstep = w4 ? 2 : what;
        // corrector target i of this lane (KFAR: the targets stay in their far array, ctm[0] only says whether there are any)
        auto tgt = [&](int i) -> double { return C::KFAR ? ((ctm[0] != 0.0) ? G[F::cxr + i * NT + tid_] : 0.0) : ctm[i % NCTR]; };
        auto hard = [&](double c, double ds, double cn, double& t, double& lam, double ct) {
            if (!w4) { sweep_row(what, alpha, c, ds, t, lam, ct, acc, aux); return; }
            double unused = 0.0;
            sweep_row(2, alpha, c, ds, t, lam, ct, unused, nullptr);
            sweep_row(3, 0.0, cn, 0.0, t, lam, 0.0, acc, aux);
        };
        auto soft = [&](double c, double ds, double cn, double& t, double& lam, double& sg, double& ta, double& ga, double Zp, double zp, double ct, double cts) {
            if (!w4) { sweep_row_soft(what, alpha, c, ds, t, lam, sg, ta, ga, Zp, zp, ct, cts, acc, aux); return; }
            double unused = 0.0;
            sweep_row_soft(2, alpha, c, ds, t, lam, sg, ta, ga, Zp, zp, ct, cts, unused, nullptr);
            sweep_row_soft(3, 0.0, cn, 0.0, t, lam, sg, ta, ga, Zp, zp, 0.0, 0.0, acc, aux);
        };
#pragma unroll (C::KFAR ? 1 : C::QX)
        for (int q = 0; q < C::QX; ++q) {
            const int ix = tid_ + q * NT;
            if (ix < C::NXI) {
                const int zo = NX + ix, i = ix % NX;
                ldx(q);
                const double X = LK[O::Z + zo], dS = LK[O::S + zo];
                const double Xn = w4 ? upr_step(X, alpha, dS) : X;
                if (C::SOFT && softx) {
                    soft(X - L[O::xlb + i], dS, Xn - L[O::xlb + i], tx[q % QXR][0], lx[q % QXR][0], sgx[q % QXS][0], tax[q % QXS][0], gax[q % QXS][0], ZL, zL, tgt(2 * q), tgt(NCT0 + 2 * q));
                    soft(L[O::xub + i] - X, -dS, L[O::xub + i] - Xn, tx[q % QXR][1], lx[q % QXR][1], sgx[q % QXS][1], tax[q % QXS][1], gax[q % QXS][1], ZU, zU, tgt(2 * q + 1), tgt(NCT0 + 2 * q + 1));
                } else {
                    hard(X - L[O::xlb + i], dS, Xn - L[O::xlb + i], tx[q % QXR][0], lx[q % QXR][0], tgt(2 * q));
                    hard(L[O::xub + i] - X, -dS, L[O::xub + i] - Xn, tx[q % QXR][1], lx[q % QXR][1], tgt(2 * q + 1));
                }
                if (wstep == 2) stx(q);
            }
        }
ftoc(6, 4);   // (-DUPR_QP3_PROF_FLAT=4: the classes of rows of the sweeps -- slots 6 .. 9: state boxes, input boxes, friction rows, state-polytopic rows)
#pragma unroll (C::KFAR ? 1 : C::QU)
        for (int q = 0; q < C::QU; ++q) {
            const int iu = tid() + q * NT;
            if (iu < C::NUI) {
                const int i = iu % NU;
                ldu(q);
                const double U = LK[O::Z + N1 * NX + iu], dS = LK[O::S + N1 * NX + iu];
                const double Un = w4 ? upr_step(U, alpha, dS) : U;
                if (C::SOFT && softu) {
                    soft(U - L[O::ulb + i], dS, Un - L[O::ulb + i], tu[q % QUR][0], lu[q % QUR][0], sgu[q % QUS][0], tau_[q % QUS][0], gau[q % QUS][0], ZL, zL, tgt(2 * C::QX + 2 * q), tgt(NCT0 + 2 * C::QX + 2 * q));
                    soft(L[O::uub + i] - U, -dS, L[O::uub + i] - Un, tu[q % QUR][1], lu[q % QUR][1], sgu[q % QUS][1], tau_[q % QUS][1], gau[q % QUS][1], ZU, zU, tgt(2 * C::QX + 2 * q + 1), tgt(NCT0 + 2 * C::QX + 2 * q + 1));
                } else {
                    hard(U - L[O::ulb + i], dS, Un - L[O::ulb + i], tu[q % QUR][0], lu[q % QUR][0], tgt(2 * C::QX + 2 * q));
                    hard(L[O::uub + i] - U, -dS, L[O::uub + i] - Un, tu[q % QUR][1], lu[q % QUR][1], tgt(2 * C::QX + 2 * q + 1));
                }
                if (wstep == 2) stu(q);
            }
        }
        ftoc(7, 4);
        if (NF == 3 && C::QC > 1) {
            // several contacts per lane (multi-body shapes): slack, multiplier and corrector term of all their rows requested
            // together (one exposed latency for the lot instead of one per contact)
            double ctv[C::QC][5], clv[C::QC][5], ccv[C::QC][5];
#pragma unroll
            for (int q = 0; q < C::QC; ++q) {
                const int ic = tid_ + q * NT, icc = (ic < C::NCI) ? ic : 0;
#pragma unroll
                for (int r = 0; r < 5; ++r) { ctv[q][r] = G[F::ct + 5 * icc + r]; clv[q][r] = G[F::cl + 5 * icc + r]; ccv[q][r] = G[F::cc + 5 * icc + r]; }
            }
#pragma unroll
            for (int q = 0; q < C::QC; ++q) {
                const int ic = tid_ + q * NT;
                if (ic >= C::NCI) continue;
                const int k = ic / NC, ci = ic % NC, uo = k * NU + NQ + 3 * ci;
                const double* f = LK + O::Z + N1 * NX + uo; const double* sf = LK + O::S + N1 * NX + uo;
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const double* e3 = L + O::erow + 3 * (5 * ci + r);
                    double t = ctv[q][r], lam = clv[q][r];
                    const double fn0 = w4 ? upr_step(f[0], alpha, sf[0]) : f[0], fn1 = w4 ? upr_step(f[1], alpha, sf[1]) : f[1], fn2 = w4 ? upr_step(f[2], alpha, sf[2]) : f[2];
                    if (C::SOFT && softp) {
                        double sg = G[F::sfr + 5 * ic + r], ta = G[F::sfr + F::sfs + 5 * ic + r], ga = G[F::sfr + 2 * F::sfs + 5 * ic + r];
                        soft(e3[0] * f[0] + e3[1] * f[1] + e3[2] * f[2], e3[0] * sf[0] + e3[1] * sf[1] + e3[2] * sf[2], e3[0] * fn0 + e3[1] * fn1 + e3[2] * fn2, t, lam, sg, ta, ga, ZL, zL,
                             ccv[q][r], G[F::sfr + 3 * F::sfs + 5 * ic + r]);
                        if (wstep == 2) { G[F::sfr + 5 * ic + r] = sg; G[F::sfr + F::sfs + 5 * ic + r] = ta; G[F::sfr + 2 * F::sfs + 5 * ic + r] = ga; }
                    } else
                    hard(e3[0] * f[0] + e3[1] * f[1] + e3[2] * f[2], e3[0] * sf[0] + e3[1] * sf[1] + e3[2] * sf[2], e3[0] * fn0 + e3[1] * fn1 + e3[2] * fn2, t, lam, ccv[q][r]);
                    if (wstep == 2) { G[F::ct + 5 * ic + r] = t; G[F::cl + 5 * ic + r] = lam; }
                }
            }
        } else
        if (NF == 3) {
            // one ROW per lane item (rounds 1 - 2: a contact, five rows, per lane -- all of them on the first NCI lanes, whose waves
            // then ran fifteen rows per sweep against ten on the others); the rows of a lane requested together
            fetch_fr();
#pragma unroll
            for (int q = 0; q < QR5; ++q) {
                const int e = tid_ + q * NT;
                if (e >= NR5) continue;
                const int ic = e / 5, r = e % 5, k = ic / NC, ci = ic % NC, uo = k * NU + NQ + 3 * ci;
                const double* f = LK + O::Z + N1 * NX + uo; const double* sf = LK + O::S + N1 * NX + uo;
                const double* e3 = L + O::erow + 3 * (5 * ci + r);
                double t = tv[q], lam = lv[q];
                const double fn0 = w4 ? upr_step(f[0], alpha, sf[0]) : f[0], fn1 = w4 ? upr_step(f[1], alpha, sf[1]) : f[1], fn2 = w4 ? upr_step(f[2], alpha, sf[2]) : f[2];
                if (C::SOFT && softp) {
                    double sg = ssv[q % QS5][0], ta = ssv[q % QS5][1], ga = ssv[q % QS5][2];
                    soft(e3[0] * f[0] + e3[1] * f[1] + e3[2] * f[2], e3[0] * sf[0] + e3[1] * sf[1] + e3[2] * sf[2], e3[0] * fn0 + e3[1] * fn1 + e3[2] * fn2, t, lam, sg, ta, ga, ZL, zL,
                         cv5[q], ssv[q % QS5][3]);
                    if (wstep == 2) { G[F::sfr + e] = sg; G[F::sfr + F::sfs + e] = ta; G[F::sfr + 2 * F::sfs + e] = ga; }
                } else
                hard(e3[0] * f[0] + e3[1] * f[1] + e3[2] * f[2], e3[0] * sf[0] + e3[1] * sf[1] + e3[2] * sf[2], e3[0] * fn0 + e3[1] * fn1 + e3[2] * fn2, t, lam, cv5[q]);
                if (wstep == 2) { G[F::ct + e] = t; G[F::cl + e] = lam; }
            }
        }
        ftoc(8, 4);
        if (no > 0) {
            fetch_or();
#pragma unroll
            for (int q = 0; q < QO; ++q) {
                const int e = tid_ + q * NT;
                if (e >= (N - 1) * no) continue;
                const int k = 1 + e / no, r = e % no, ei = (k - 1) * UPR_QP3_NOMAX + r;
                double c = od[q], ds = 0.0, cn = od[q];
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const double z = LK[O::Z + k * NX + i], sz = LK[O::S + k * NX + i];
                    c += gq[q][i] * z; ds += gq[q][i] * sz;
                    if (w4) cn += gq[q][i] * upr_step(z, alpha, sz);
                }
                double t = otv[q], lam = olv[q];
                if (C::SOFT && C::ROWS && softp) {
                    double sg = osv[q % QOS][0], ta = osv[q % QOS][1], ga = osv[q % QOS][2];
                    soft(c, ds, cn, t, lam, sg, ta, ga, ZL, zL, ocv[q], osv[q % QOS][3]);
                    if (wstep == 2) { G[F::sor + ei] = sg; G[F::sor + F::sos + ei] = ta; G[F::sor + 2 * F::sos + ei] = ga; }
                } else
                hard(c, ds, cn, t, lam, ocv[q]);
                if (wstep == 2) { G[F::ot + ei] = t; G[F::ol + ei] = lam; }
            }
        }
        ftoc(9, 4);
        if (what == 0) acc = acc > 1e-30 ? 1.0 / acc : 1e30;
        return acc;
    }

    // ---- the corrector's step on the lane-owned rows in ONE evaluation (round 6; hard rows without state-polytopic rows) -------------
    // ineq_sweep runs three times behind the corrector's forward sweep -- step length (what 0), the safeguard's trial products
    // (what 5, first iterations), the step itself with the new residuals (what 4) -- and every run forms the row's direction anew:
    // c and ds out of LDS, two reciprocals, the friction rows' (t, lam, target) out of the far arrays.  Here the direction (dt, dlam)
    // of every row of the lane is formed ONCE and kept in registers (twelve rows a lane for the headline shape: 24 values) across the
    // two reductions; the trial products and the step are then two or three operations a row.  Same arithmetic per row as
    // sweep_row (what 0 | 5 | 2 + 3), so iterates are bit-identical to the three-pass form.
    UPR_HDI static void dir_row(double c, double ds, double t, double lam, double cterm, double& dt, double& dl, double& acc) {
        const double rp = c - t;
        dt = ds + rp;
        const double rc = lam * t + cterm;
        const double rt = upr_rcp(t);
        dl = -(rc + lam * dt) * rt;
        acc = fmax(acc, fmax(-dt * rt, -dl * upr_rcp(lam)));
    }
    // directions of all rows of the lane; returns the lane's largest admissible step (as ineq_sweep what 0, mode 3)
    UPR_HDI double rows_dir(const double (&ctm)[NCTR], row_dirs& D) {
        const int tid_ = tid();
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < C::QX; ++q) {
            const int ix = tid_ + q * NT;
            D.dt[2 * q] = 0.0; D.dl[2 * q] = 0.0; D.dt[2 * q + 1] = 0.0; D.dl[2 * q + 1] = 0.0;
            if (ix < C::NXI) {
                const int zo = NX + ix, i = ix % NX;
                const double X = LK[O::Z + zo], dS = LK[O::S + zo];
                dir_row(X - L[O::xlb + i], dS, tx[q % QXR][0], lx[q % QXR][0], ctm[(2 * q) % NCTR], D.dt[2 * q], D.dl[2 * q], acc);
                dir_row(L[O::xub + i] - X, -dS, tx[q % QXR][1], lx[q % QXR][1], ctm[(2 * q + 1) % NCTR], D.dt[2 * q + 1], D.dl[2 * q + 1], acc);
            }
        }
#pragma unroll
        for (int q = 0; q < C::QU; ++q) {
            const int iu = tid_ + q * NT, o = 2 * C::QX + 2 * q;
            D.dt[o] = 0.0; D.dl[o] = 0.0; D.dt[o + 1] = 0.0; D.dl[o + 1] = 0.0;
            if (iu < C::NUI) {
                const int i = iu % NU;
                const double U = LK[O::Z + N1 * NX + iu], dS = LK[O::S + N1 * NX + iu];
                dir_row(U - L[O::ulb + i], dS, tu[q % QUR][0], lu[q % QUR][0], ctm[o % NCTR], D.dt[o], D.dl[o], acc);
                dir_row(L[O::uub + i] - U, -dS, tu[q % QUR][1], lu[q % QUR][1], ctm[(o + 1) % NCTR], D.dt[o + 1], D.dl[o + 1], acc);
            }
        }
#pragma unroll
        for (int q = 0; q < OP_QR5; ++q) {
            const int e = tid_ + q * NT;
            D.dt[OP_NB + q] = 0.0; D.dl[OP_NB + q] = 0.0;
            if (e < 5 * C::NCI) {
                const int ic = e / 5, r = e % 5, k = ic / NC, ci = ic % NC, uo = k * NU + NQ + 3 * ci;
                const double* f = LK + O::Z + N1 * NX + uo; const double* sf = LK + O::S + N1 * NX + uo;
                const double* e3 = L + O::erow + 3 * (5 * ci + r);
                dir_row(e3[0] * f[0] + e3[1] * f[1] + e3[2] * f[2], e3[0] * sf[0] + e3[1] * sf[1] + e3[2] * sf[2], D.ft[q], D.fl[q], D.fc[q], D.dt[OP_NB + q], D.dl[OP_NB + q], acc);
            }
        }
        return acc > 1e-30 ? 1.0 / acc : 1e30;
    }
    // smallest trial product (lam + a dlam)(t + a dt) of the lane's rows and their sum (the centrality safeguard: ineq_sweep what 5)
    UPR_HDI double rows_trial(double alpha, const row_dirs& D, double* aux) const {
        const int tid_ = tid();
        double acc = 1e300;
#pragma unroll
        for (int q = 0; q < C::QX; ++q) if (tid_ + q * NT < C::NXI) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { const double v = (lx[q % QXR][s2] + alpha * D.dl[2 * q + s2]) * (tx[q % QXR][s2] + alpha * D.dt[2 * q + s2]); acc = fmin(acc, v); *aux += v; }
        }
#pragma unroll
        for (int q = 0; q < C::QU; ++q) if (tid_ + q * NT < C::NUI) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { const int o = 2 * C::QX + 2 * q + s2; const double v = (lu[q % QUR][s2] + alpha * D.dl[o]) * (tu[q % QUR][s2] + alpha * D.dt[o]); acc = fmin(acc, v); *aux += v; }
        }
#pragma unroll
        for (int q = 0; q < OP_QR5; ++q) if (tid_ + q * NT < 5 * C::NCI) { const double v = (D.fl[q] + alpha * D.dl[OP_NB + q]) * (D.ft[q] + alpha * D.dt[OP_NB + q]); acc = fmin(acc, v); *aux += v; }
        return acc;
    }
    // the step of the rows, and with the row's value at the NEW iterate the next iteration's first residuals (ineq_sweep what 4):
    // returns the lane's largest |c - t|, adds lam t to *aux
    UPR_HDI double rows_apply(double alpha, row_dirs& D, double* aux) {
        const int tid_ = tid();
        double acc = 0.0;
        auto res = [&](double cn, double t, double lam) { const double a = fabs(cn - t); if (a > acc) acc = a; *aux += lam * t; };
#pragma unroll
        for (int q = 0; q < C::QX; ++q) {
            const int ix = tid_ + q * NT;
            if (ix < C::NXI) {
                const int zo = NX + ix, i = ix % NX;
                const double Xn = upr_step(LK[O::Z + zo], alpha, LK[O::S + zo]);
                tx[q % QXR][0] += alpha * D.dt[2 * q]; lx[q % QXR][0] += alpha * D.dl[2 * q]; tx[q % QXR][1] += alpha * D.dt[2 * q + 1]; lx[q % QXR][1] += alpha * D.dl[2 * q + 1];
                res(Xn - L[O::xlb + i], tx[q % QXR][0], lx[q % QXR][0]); res(L[O::xub + i] - Xn, tx[q % QXR][1], lx[q % QXR][1]);
            }
        }
#pragma unroll
        for (int q = 0; q < C::QU; ++q) {
            const int iu = tid_ + q * NT, o = 2 * C::QX + 2 * q;
            if (iu < C::NUI) {
                const int i = iu % NU;
                const double Un = upr_step(LK[O::Z + N1 * NX + iu], alpha, LK[O::S + N1 * NX + iu]);
                tu[q % QUR][0] += alpha * D.dt[o]; lu[q % QUR][0] += alpha * D.dl[o]; tu[q % QUR][1] += alpha * D.dt[o + 1]; lu[q % QUR][1] += alpha * D.dl[o + 1];
                res(Un - L[O::ulb + i], tu[q % QUR][0], lu[q % QUR][0]); res(L[O::uub + i] - Un, tu[q % QUR][1], lu[q % QUR][1]);
            }
        }
#pragma unroll
        for (int q = 0; q < OP_QR5; ++q) {
            const int e = tid_ + q * NT;
            if (e < 5 * C::NCI) {
                const int ic = e / 5, r = e % 5, k = ic / NC, ci = ic % NC, uo = k * NU + NQ + 3 * ci;
                const double* f = LK + O::Z + N1 * NX + uo; const double* sf = LK + O::S + N1 * NX + uo;
                const double* e3 = L + O::erow + 3 * (5 * ci + r);
                const double fn0 = upr_step(f[0], alpha, sf[0]), fn1 = upr_step(f[1], alpha, sf[1]), fn2 = upr_step(f[2], alpha, sf[2]);
                const double t = D.ft[q] + alpha * D.dt[OP_NB + q], lam = D.fl[q] + alpha * D.dl[OP_NB + q];
                res(e3[0] * fn0 + e3[1] * fn1 + e3[2] * fn2, t, lam);
                G[F::ct + e] = t; G[F::cl + e] = lam;
            }
        }
        return acc;
    }

    // full == false: only the inequality residual and the complementarity average (what the step needs and what
    // decides whether the expensive stationarity / equality residuals can matter at all)
    UPR_HDI void residuals(int ntot, double* res, bool full) {
        if (!full) {
            double lt0 = 0.0;
            const double r_in0 = ineq_sweep(3, 0.0, &lt0, zero_targets());
            res[0] = 0.0; res[1] = 0.0; res[2] = r_in0; res[3] = lt0;
            reduce4(res);
            res[0] = 1e300; res[1] = 1e300;
            res[3] /= (ntot > 0 ? ntot : 1);
            return;
        }
        const double* pi = ws + W::pi; const double* nu = ws + W::nu;
        const int save = mode;
        mode = 2;
        prep(0);
        mode = save;
        double r_stat = 0.0, r_eq = 0.0;
        UPR_FORT(e, N * NX) {
            const int k = 1 + e / NX, i = e % NX;
            double v = LK[O::gxs + k * NX + i] - pi[k * NX + i];
            if (k < N) {
                const double* pn = pi + (k + 1) * NX; const double* Ck = rec(k) + lin_gx;
                const int blk = i / NQ, j = i % NQ;
                for (int a = 0; a <= blk; ++a) v += coefA(a, blk) * pn[a * NQ + j];
                for (int q = 0; q < NE; ++q) v += Ck[q * NX + i] * nu[k * NE + q];
            } else if (neN > 0) {
                if (i < NQ) { for (int q = 0; q < 3; ++q) v -= L[O::jN + q * NQ + i] * L[O::yN + q]; }
                else v += L[O::yN + 3 + (i - NQ)];
            }
            r_stat = fmax(r_stat, fabs(v));
        }
        UPR_FORT(e, N * NU) {
            const int k = e / NU, i = e % NU;
            double v = LK[O::gus + e];
            if (i < NQ) { const double* pn = pi + (k + 1) * NX; v += h3 * pn[i] + h2 * pn[NQ + i] + h * pn[2 * NQ + i]; }
            else if (C::NB == 1) { for (int q = 0; q < NE; ++q) v += L[O::df + q * NFC + (i - NQ)] * nu[k * NE + q]; }
            else {   // (the column of a force has entries in the rows of the bodies its contact loads only)
                const int ci = (i - NQ) / NF, b2 = P->contact_body2[ci], b1 = P->contact_body1[ci];
                for (int q = 0; q < 6; ++q) v += DF(6 * b2 + q, i - NQ) * nu[k * NE + 6 * b2 + q];
                if (b1 >= 0) for (int q = 0; q < 6; ++q) v += L[O::df + (6 * b1 + q) * NFC + (i - NQ)] * nu[k * NE + 6 * b1 + q];
            }
            r_stat = fmax(r_stat, fabs(v));
        }
        UPR_FORT(e, N * NQ) {
            const int k = e / NQ, j = e % NQ;
            const double* X = Zx(k); const double* Xn = Zx(k + 1); const double* U = Zu(k);
            const double q = X[j], v = X[NQ + j], a = X[2 * NQ + j], u = U[j];
            r_eq = fmax(r_eq, fabs(q + h * v + h2 * a + h3 * u - Xn[j]));
            r_eq = fmax(r_eq, fabs(v + h * a + h2 * u - Xn[NQ + j]));
            r_eq = fmax(r_eq, fabs(a + h * u - Xn[2 * NQ + j]));
        }
        UPR_FORT(e, N * NE) r_eq = fmax(r_eq, fabs(ekp()[e] - rho_eq * nu[e]));
        terminal_residual();
        UPR_SYNC();
        if (neN > 0) UPR_FORT(q, C::NEN) r_eq = fmax(r_eq, fabs(L[O::eN + q]));
        double lt = 0.0;
        soft_stat = 0.0;
        const double r_in = ineq_sweep(3, 0.0, &lt, zero_targets());
        if (C::SOFT) r_stat = fmax(r_stat, soft_stat);
        res[0] = r_stat; res[1] = r_eq; res[2] = r_in; res[3] = lt;
        reduce4(res);
        res[3] /= (ntot > 0 ? ntot : 1);
    }

    // Linear feedback gains of this QP (sqp.use_feedback_policy; what feedback_kernel of upr_api.hip gathers for the other QP
    // kernels), out[N][nu][nx], ocs2 sign: jerk rows -K_k of the Riccati recursion; contact-force rows
    // -Hff^-1 Df' S^-1 C_k (the forces follow the state through the object-dynamics rows).  A QP whose factorisation broke
    // down (status 2) has no policy: zeros.
    UPR_HDI void write_feedback(double* out, int status) {
        UPR_SYNC();
        if (status == 2) { UPR_FORT(e, N * NU * NX) out[e] = 0.0; return; }
        UPR_FORT(e, N * NQ * NX) {
            const int k = e / (NQ * NX), rem = e % (NQ * NX);
            out[(size_t)k * NU * NX + rem] = -G[F::Ks + e];
        }
        constexpr int SB = C::SB, NBLK = NE / SB;
        UPR_FORT(e, N * NX) {
            const int k = e / NX, c = e % NX;
            const double* Ck = rec(k) + lin_gx + c;
            double t2[NE];
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const double* Ls = G + F::lsi + k * C::NLS + blk * SB * SB;
                double t1[SB];
#pragma unroll
                for (int r = 0; r < SB; ++r) { double v = 0.0;
#pragma unroll
                    for (int m = 0; m <= r; ++m) v += Ls[r * SB + m] * Ck[(blk * SB + m) * NX];
                    t1[r] = v; }
#pragma unroll
                for (int r = 0; r < SB; ++r) { double v = 0.0;
#pragma unroll
                    for (int m = r; m < SB; ++m) v += Ls[m * SB + r] * t1[m];
                    t2[blk * SB + r] = v; }
            }
            for (int ci = 0; ci < NC; ++ci) {
                double t3[NF];
#pragma unroll
                for (int a = 0; a < NF; ++a) { double v = 0.0;
                    if (C::BIGF) {   // (compact Df: the six rows of the contact's body; t2 is indexed at run time here)
                        const int rb = 6 * P->contact_body2[ci];
                        for (int r = 0; r < 6; ++r) v += DF(rb + r, NF * ci + a) * t2[rb + r];
                    } else {
#pragma unroll
                    for (int r = 0; r < NE; ++r) v += L[O::df + r * NFC + NF * ci + a] * t2[r];
                    }
                    t3[a] = v; }
                double* o = out + ((size_t)k * NU + NQ + NF * ci) * NX + c;
                if (NF == 3) {
                    const double* Bk = G + F::lfi + k * C::NLF + 9 * ci;
                    double y[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) { double v = 0.0;
#pragma unroll
                        for (int b2 = 0; b2 <= a; ++b2) v += Bk[3 * a + b2] * t3[b2 % NF];
                        y[a] = v; }
#pragma unroll
                    for (int a = 0; a < 3; ++a) { double v = 0.0;
#pragma unroll
                        for (int b2 = a; b2 < 3; ++b2) v += Bk[3 * b2 + a] * y[b2];
                        o[(size_t)a * NX] = -v; }
                } else {
                    const double lf = G[F::lfi + k * C::NLF + ci];
                    o[0] = -lf * lf * t3[0];
                }
            }
        }
    }

    // Cycle stamps per phase (upr_batch_qp_profile, tools/dbg_profile.py): compiled in with -DUPR_QP3_PROF only -- a run-time
    // instantiation, UPR_QP3_JIT=2 UPR_JIT_FLAGS=-DUPR_QP3_PROF.  Rounds 1 - 5 shipped them behind a null-pointer test; the ~150
    // uniform branches per interior-point iteration cut the schedule into that many pieces: 1.8 % of the headline launch.
    double* prof; long long tlast;
    UPR_HDI void tic() {
#if !defined(UPR_HOST_EMU) && defined(UPR_QP3_PROF)
        if (prof && lane() == 0) tlast = (long long)__builtin_readcyclecounter();
#endif
    }
    // -DUPR_QP3_PROF_MAT (tools/build_prof.sh, never the production library): the matrix sweep in detail.  Every wave also
    // reads the counter when it ARRIVES at each of the sweep's three barriers, so work and wait separate:
    //   slot 0 / 1: phase 1 work / wait, 2 / 3: factorisation + matrix-core preload + feedback work / wait, 4 / 5: P update
    //   work / wait, 7: wave 0 up to the end of its pivots; everything outside the sweep goes to slot 15.
    UPR_HDI void mtoc(int id) {
#ifdef UPR_QP3_PROF_MAT
        toc(100 + id);
#else
        (void)id;
#endif
    }
    // -DUPR_QP3_PROF_FLAT (experiment builds): slots 6 .. 9 time the parts of a flat phase instead of the matrix sweep
    // (-DUPR_QP3_PROF_FLAT=1: the update, = 2: prep A, = 3: the dense Schur phase, = 4: the row classes of ineq_sweep, = 5: arrival at the forward sweep's barrier)
    UPR_HDI void ftoc(int id, int set = 1) {
#ifdef UPR_QP3_PROF_FLAT
        if (set == UPR_QP3_PROF_FLAT) toc_raw(id);
#else
        (void)id; (void)set;
#endif
    }
    UPR_HDI void toc(int id) {
#ifdef UPR_QP3_PROF_FLAT
        if (id >= 6 && id <= 9) return;
#endif
        toc_raw(id);
    }
    UPR_HDI void toc_raw(int id) {
#if !defined(UPR_HOST_EMU) && defined(UPR_QP3_PROF)
#ifdef UPR_QP3_PROF_MAT
        id = (id >= 100) ? id - 100 : ((id == 6) ? 1 : ((id == 8) ? 3 : ((id == 9) ? 5 : ((id == 7) ? 7 : 15))));
#endif
        // lane 0 of each of the first four waves keeps its own set: a barrier is passed when the LAST wave arrives, and only
        // the per-wave view shows which one that is.  A read costs ~290 cycles; s_memtime is not ordered against VALU work,
        // hence the scheduling fences.
        if (prof && lane() == 0 && wb < 256) {
            __builtin_amdgcn_sched_barrier(0);
            long long t = (long long)__builtin_readcyclecounter();
            L[O::prf + 16 * (wb >> 6) + id] += (double)(t - tlast); tlast = t;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    }

    UPR_HDI void solve(const upr_ctx& c, const upr_qp_args& A, int b, double* lds) {
        // (layout: what prep stages in the sweeps' working set fits it)
        static_assert(C::MULTI || C::COUPLED || (N * NE * NFC <= O::sst - O::Pa && N * NE * NE <= O::yN - O::sst), "prep stages Z and S in the scratch region");
        static_assert(!C::MULTI || C::BIGF || C::KFAR || N * NFC * 6 <= O::yN - O::Pa, "multi-body shapes: Z [knot][force][6] in the sweeps' working set (P .. C_k)");
        static_assert(!C::ROWS || 2 * (N - 1) * UPR_QP3_NOMAX <= O::yN - O::hux, "prep stages the state-polytopic rows' (s, w) there too");
        ctx = c; P = A.P; L = lds; red_par = 0;
#ifdef UPR_HOST_EMU
        wb = 0;
#else
        wb = __builtin_amdgcn_readfirstlane(c.tid & ~63);
#endif
        xs = A.xs + (size_t)b * N1 * NX; us = A.us + (size_t)b * N * NU; x0 = A.x0 + (size_t)b * NX;
        lin = A.lin + (size_t)b * N1 * A.d.lin_stride; Dfg = A.Df + (size_t)b * NE * NFC; ws = A.ws + (size_t)b * A.d.ws_stride; G = ws + W::far; LK = C::KFAR ? ws + W::kfar : lds;
        lin_stride = A.d.lin_stride; lin_g = A.d.lin_g; lin_gx = A.d.lin_gx; lin_grad = A.d.lin_grad; lin_hess = A.d.lin_hess; neN = A.d.neN;
        no = C::ROWS ? A.d.no : 0; lin_obs = A.d.lin_obs; hee_w = (C::ROWS && no > 0) ? F::heew : F::hee;
        h = P->dt; h2 = 0.5 * h * h; h3 = h * h * h / 6.0; sigma_mu = 0.0; mode = 0; fbk = P->use_feedback_policy != 0;
        have_ek = false;
        softx = C::SOFT && P->soft_state_box != 0; softu = C::SOFT && P->soft_input_box != 0; softp = C::SOFT && P->soft_poly != 0;
        ZL = P->soft_L2_lower; ZU = P->soft_L2_upper; zL = P->soft_L1_lower; zU = P->soft_L1_upper; soft_stat = 0.0;
        rho_eq = upr_qp_rho_soft(P); rho_s = upr_qp_rho_s(P, NE, NFC); rho_px = upr_qp_rho_prox(P, NE, NFC);
        prof = A.prof ? A.prof + (size_t)b * 64 : nullptr;
        if (prof) UPR_FORT(i, 64) L[O::prf + i] = 0.0;
        tic();
        // ---- constants and linearisation-point data into LDS
        UPR_FORT(i, NX) { L[O::xlb + i] = P->x_lb[i]; L[O::xub + i] = P->x_ub[i]; L[O::qd + i] = P->Qdiag[i]; L[O::xd + i] = P->xd[i]; }
        UPR_FORT(i, NU) { L[O::ulb + i] = P->u_lb[i]; L[O::uub + i] = P->u_ub[i]; L[O::rd + i] = P->Rdiag[i]; }
        if (NF == 3) UPR_FORT(e, C::NP) { double e3[3]; upr_friction_row_jac(P, e / 5, e % 5, e3); L[O::erow + 3 * e] = e3[0]; L[O::erow + 3 * e + 1] = e3[1]; L[O::erow + 3 * e + 2] = e3[2]; }
        if (C::BIGF) UPR_FORT(e, 6 * NFC) { const int col = e / 6, r6 = e % 6; L[O::df + e] = Dfg[(6 * P->contact_body2[col / NF] + r6) * NFC + col]; }
        else UPR_FORT(e, NE * NFC) L[O::df + e] = Dfg[e];
        if (C::NB > 1) UPR_FORT(bb, C::NB) {
            int n = 0;
            for (int ci = 0; ci < NC; ++ci) if (P->contact_body2[ci] == bb || P->contact_body1[ci] == bb) L[O::clist + bb * NC + n++] = (double)ci;
            L[O::ccnt + bb] = (double)n;
        }
        if (C::COUPLED) UPR_FORT(bl, C::NB * (C::NB + 1) / 2) {
            int bi = 0, b0 = 0;
            for (int i = 1; i < C::NB; ++i) if (bl >= i * (i + 1) / 2) { bi = i; b0 = i * (i + 1) / 2; }
            const int bj = bl - b0;
            int n = 0;
            for (int ci = 0; ci < NC; ++ci) {
                const int b1 = P->contact_body1[ci], b2 = P->contact_body2[ci];
                if ((bi == b1 || bi == b2) && (bj == b1 || bj == b2)) L[O::plist + bl * NC + n++] = (double)ci;
            }
            for (int t = n; t < NC; ++t) L[O::plist + bl * NC + t] = L[O::plist + bl * NC];   // (padding: a valid index)
            if (n == 0) for (int t = 0; t < NC; ++t) L[O::plist + bl * NC + t] = 0.0;
            L[O::pcnt + bl] = (double)n;
        }
        UPR_FORT(e, N1 * NX) { const int k = e / NX; LK[O::Z + e] = (k == 0) ? x0[e] : xs[e]; LK[O::S + e] = 0.0; }
        UPR_FORT(e, N * NU) { LK[O::Z + N1 * NX + e] = us[e]; LK[O::S + N1 * NX + e] = 0.0; }
        UPR_FORT(e, N * C::NH) { const int k = e / C::NH; G[F::hee + e] = rec(k)[lin_hess + e % C::NH]; }
        UPR_FORT(e, 3 * NQ) L[O::jN + e] = rec(N)[lin_hess + e];
        UPR_FORT(q, 3) L[O::misc + 4 + q] = rec(N)[lin_grad + q];
        UPR_FORT(q, C::NEN) { L[O::yN + q] = 0.0; L[O::dyN + q] = 0.0; }
        if (tid() == 0) L[O::misc] = 0.0;
        UPR_FORT(i, W::store) ws[i] = 0.0;
        UPR_SYNC();
        // g0 = gradEE - Hee xs_q ; e0 = g - C xs - Df us_f   (affine parts at the linearisation point)
        UPR_FORT(e, N * NQ) {
            const int k = e / NQ, i = e % NQ;
            double v = rec(k)[lin_grad + i];
            for (int j = 0; j < NQ; ++j) v -= G[F::hee + k * C::NH + upr_tri(NQ, i, j)] * xs[k * NX + j];
            G[F::g0 + e] = v;
        }
        UPR_FORT(e, N * NE) {
            const int k = e / NE, r = e % NE;
            const double* Ck = rec(k) + lin_gx + r * NX;
            double v = rec(k)[lin_g + r];
            for (int j = 0; j < NX; ++j) v -= Ck[j] * xs[k * NX + j];
            if (C::BIGF) v -= df_dot(r, us + k * NU + NQ);
            else for (int i = 0; i < NFC; ++i) v -= L[O::df + r * NFC + i] * us[k * NU + NQ + i];
            G[F::e0 + e] = v;
        }
        if (no > 0) {
            UPR_FORT(e, (N - 1) * no) {
                const int k = 1 + e / no, r = e % no, ei = (k - 1) * UPR_QP3_NOMAX + r;
                const double* g = orow(k, r);
                const double d0 = rec(k)[lin_obs + r];   // value at the linearisation point = at the initial iterate (knots >= 1)
                double v = d0;
                for (int i = 0; i < NQ; ++i) v -= g[i] * xs[k * NX + i];
                const double t = d0 > UPR_QP_THR ? d0 : UPR_QP_THR;
                G[F::od0 + ei] = v; G[F::ot + ei] = t; G[F::ol + ei] = UPR_QP_MU0 / t; G[F::oc + ei] = 0.0;
                if (C::SOFT && C::ROWS && softp) { G[F::sor + ei] = 0.0; G[F::sor + F::sos + ei] = UPR_QP_THR; G[F::sor + 2 * F::sos + ei] = UPR_QP_MU0 / UPR_QP_THR; G[F::sor + 3 * F::sos + ei] = 0.0; }
            }
            UPR_FORT(e, C::NH) G[F::heew + e] = G[F::hee + e];   // knot 0 carries no such rows
        }
        // ---- initial slacks / multipliers
#pragma unroll (C::KFAR ? 1 : C::QX)
        for (int q = 0; q < C::QX; ++q) {
            const int ix = tid() + q * NT;
            tx[q % QXR][0] = tx[q % QXR][1] = 1.0; lx[q % QXR][0] = lx[q % QXR][1] = 0.0;
            if (ix < C::NXI) {
                const int i = ix % NX; const double X = LK[O::Z + NX + ix];
                const double c0 = X - L[O::xlb + i], c1 = L[O::xub + i] - X;
                tx[q % QXR][0] = c0 > UPR_QP_THR ? c0 : UPR_QP_THR; tx[q % QXR][1] = c1 > UPR_QP_THR ? c1 : UPR_QP_THR;
                lx[q % QXR][0] = UPR_QP_MU0 / tx[q % QXR][0]; lx[q % QXR][1] = UPR_QP_MU0 / tx[q % QXR][1];
            }
            if (C::SOFT) for (int s2 = 0; s2 < 2; ++s2) { sgx[q % QXS][s2] = 0.0; tax[q % QXS][s2] = (softx && ix < C::NXI) ? UPR_QP_THR : 1.0; gax[q % QXS][s2] = (softx && ix < C::NXI) ? UPR_QP_MU0 / UPR_QP_THR : 0.0; }
            stx(q);
        }
#pragma unroll (C::KFAR ? 1 : C::QU)
        for (int q = 0; q < C::QU; ++q) {
            const int iu = tid() + q * NT;
            tu[q % QUR][0] = tu[q % QUR][1] = 1.0; lu[q % QUR][0] = lu[q % QUR][1] = 0.0;
            if (iu < C::NUI) {
                const int i = iu % NU; const double U = LK[O::Z + N1 * NX + iu];
                const double c0 = U - L[O::ulb + i], c1 = L[O::uub + i] - U;
                tu[q % QUR][0] = c0 > UPR_QP_THR ? c0 : UPR_QP_THR; tu[q % QUR][1] = c1 > UPR_QP_THR ? c1 : UPR_QP_THR;
                lu[q % QUR][0] = UPR_QP_MU0 / tu[q % QUR][0]; lu[q % QUR][1] = UPR_QP_MU0 / tu[q % QUR][1];
            }
            if (C::SOFT) for (int s2 = 0; s2 < 2; ++s2) { sgu[q % QUS][s2] = 0.0; tau_[q % QUS][s2] = (softu && iu < C::NUI) ? UPR_QP_THR : 1.0; gau[q % QUS][s2] = (softu && iu < C::NUI) ? UPR_QP_MU0 / UPR_QP_THR : 0.0; }
            stu(q);
        }
        store_rows();
        if (NF == 3) for (int q = 0; q < C::QC; ++q) {
            const int ic = tid() + q * NT;
            if (ic < C::NCI) {
                const int k = ic / NC, ci = ic % NC;
                const double* f = LK + O::Z + N1 * NX + k * NU + NQ + 3 * ci;
                for (int r = 0; r < 5; ++r) {
                    const double* e3 = L + O::erow + 3 * (5 * ci + r);
                    const double c0 = e3[0] * f[0] + e3[1] * f[1] + e3[2] * f[2];
                    const double t = c0 > UPR_QP_THR ? c0 : UPR_QP_THR;
                    G[F::ct + 5 * ic + r] = t; G[F::cl + 5 * ic + r] = UPR_QP_MU0 / t; G[F::cc + 5 * ic + r] = 0.0;
                    if (C::SOFT && softp) { G[F::sfr + 5 * ic + r] = 0.0; G[F::sfr + F::sfs + 5 * ic + r] = UPR_QP_THR; G[F::sfr + 2 * F::sfs + 5 * ic + r] = UPR_QP_MU0 / UPR_QP_THR; G[F::sfr + 3 * F::sfs + 5 * ic + r] = 0.0; }
                }
            }
        }
        UPR_SYNC();
        const int ntot = N * (2 * NU + C::NP) + N * 2 * NX + (N - 1) * no + (softx ? N * 2 * NX : 0) + (softu ? N * 2 * NU : 0) + (softp ? N * C::NP + (N - 1) * no : 0);   // each softened row adds the pair (tau, gam)
        double res[4] = {0, 0, 0, 0}, res_next[4] = {0, 0, 0, 0};
        bool have_next = false;
        int it = 0, status = 1;
        const double tol = P->qp_tol;
        const double tol_stat = P->qp_tol_stat > 0.0 ? P->qp_tol_stat : tol;   // HPIPM tol_stat (upright_mi.h)
        toc(15);
        for (;; ++it) {
            // the KKT test can only pass once the complementarity average and the inequality residual are below the
            // tolerance: until then the stationarity / equality residuals are not evaluated
#pragma nounroll
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 0 && have_next) {   // left by the rows' step of the previous iteration (ineq_sweep what == 4)
                    res[0] = 1e300; res[1] = 1e300; res[2] = res_next[2]; res[3] = res_next[3] / (ntot > 0 ? ntot : 1);
                } else residuals(ntot, res, pass == 1);
                if (!((it > 0 && res[2] < tol && res[3] < tol) || it >= P->qp_iter_max)) break;
            }
            toc(0);
#ifdef UPR_HOST_EMU
            if (getenv("UPR_EMU_DEBUG")) printf("v3 it %d res %.3e %.3e %.3e %.3e\n", it, res[0], res[1], res[2], res[3]);
#endif
            if (it > 0 && res[0] < tol_stat && res[1] < tol && res[2] < tol && res[3] < tol) { status = 0; break; }
            if (it >= P->qp_iter_max) break;
            const double mu = res[3];
            mode = 0;
            prep(2);
            backward_mat(); toc(11);
#ifndef UPR_HOST_EMU
            backward_vec(SW2); toc(11);
#else
            backward_vec(); toc(11);
#endif
            if (L[O::misc] != 0.0) {
                status = 2;
#ifndef UPR_HOST_EMU
                load_rows();   // (the exit reads the multipliers of the box rows: with this reload they are dead across the sweeps)
#endif
                break;
            }
            forward<false>(); toc(13);
            double a_aff, mu_aff;
            {
                // step length of the predictor and its complementarity average from ONE pass over the rows: for the affine direction
                // lam dt + t dlam = -lam t row by row, so sum (lam + a dlam)(t + a dt) = (1 - a) sum lam t + a^2 sum dt dlam
                double v4[4] = {0.0, 0.0, 0.0, 0.0};
                v4[0] = -ineq_sweep(0, 0.0, &v4[3], zero_targets());   // (max of -alpha = -min alpha; v4[3]: sum of dt dlam)
                reduce4(v4);
                a_aff = -v4[0];
                if (a_aff > 1.0) a_aff = 1.0;
                mu_aff = (1.0 - a_aff) * mu + a_aff * a_aff * (v4[3] / ntot);
            }
            const double sg = mu_aff / mu;
            sigma_mu = sg * sg * sg * mu;
            if (sigma_mu < UPR_QP_SIGMA_FLOOR * tol) sigma_mu = UPR_QP_SIGMA_FLOOR * tol;
            toc(14);
            mode = 1;
            prep(1);
            backward_vec(); toc(11);
            mode = 3;
            double ctm[NCTR];
            forward<true>(); toc(13);
            // (requested here, not ahead of the forward sweep's costates: that measured 0.6 - 0.8 % SLOWER on the headline launch)
            row_dirs rd;
            prefetch_step(ctm, rd);
            const double a_loc = ONEPASS ? rows_dir(ctm, rd) : ineq_sweep(0, 0.0, nullptr, ctm);
            ftoc(6);
            double a = reduce(a_loc, 2);
            ftoc(7);
            if (a > 1.0) a = 1.0;
            a *= 0.995;   // see upr_qp.h
            // centrality safeguard of the first iterations (UPR_QP_NGAM, upr_qp.h): one more pass over the rows -- the smallest trial
            // product and their sum through one combined reduction
            if (UPR_QP_NGAM > 0.0 && it < UPR_QP_NIT) {
                double v4[4] = {0.0, 0.0, 0.0, 0.0};
                v4[0] = ONEPASS ? -rows_trial(a, rd, &v4[3]) : -ineq_sweep(5, a, &v4[3], ctm, ROWMEM);   // (ROWMEM: the rows come from their parked copy again -- kept live across the reduction they go to scratch)
                reduce4(v4);
                if (!(-v4[0] >= UPR_QP_NGAM * (v4[3] / (ntot > 0 ? ntot : 1)))) a *= UPR_QP_NBT;
            }
            // the rows' step also leaves |c - t| and lam t at the NEW iterate: the next iteration's first residual pass.  (The SOFT
            // instantiations too since their rows no longer live in scratch: 3.20 -> 3.07 ms; before: 3.77 -> 3.92.)
            res_next[0] = 0.0; res_next[1] = 0.0; res_next[3] = 0.0;
            res_next[2] = ONEPASS ? rows_apply(a, rd, &res_next[3]) : ineq_sweep(4, a, &res_next[3], ctm, false);
            store_rows();
            // (the multipliers' old values are requested in front of the barrier: their round trip overlaps it)
            constexpr int QPI = (N1 * NX + NT - 1) / NT, QNU = (N * NE + NT - 1) / NT;
            double pio[QPI], nuo[QNU], nun_[QNU];
            {
                const int tid_ = tid();
#pragma unroll
                for (int q = 0; q < QPI; ++q) { const int e = tid_ + q * NT; pio[q] = ws[W::pi + ((e < N1 * NX) ? e : 0)]; }
#pragma unroll
                for (int q = 0; q < QNU; ++q) { const int e = tid_ + q * NT, ec = (e < N * NE) ? e : 0; nuo[q] = ws[W::nu + ec]; nun_[q] = G[F::nun + ec]; }
            }
            reduce4(res_next);   // (its barriers are the one the update of the iterate needs: every lane is through with Z and S)
            have_next = true;
            ftoc(8);
            {
                const int tid_ = tid();
#pragma unroll
                for (int q = 0; q < QPI; ++q) {
                    const int e = tid_ + q * NT;
                    if (e < N1 * NX) {
                        if (e >= NX) LK[O::Z + e] = upr_step(LK[O::Z + e], a, LK[O::S + e]);
                        ws[W::pi + e] = pio[q] + a * (LK[O::pin + e] - pio[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < QNU; ++q) { const int e = tid_ + q * NT; if (e < N * NE) ws[W::nu + e] = nuo[q] + a * (nun_[q] - nuo[q]); }
            }
            if (C::INCEK) {   // the equality residual at the new iterate: ek += a (C sx + Df sf)
                UPR_FORT(e, N * NE) { const int k = e / NE, r = e % NE; LK[O::ek + e] += a * (LK[O::dek + e] + df_dot(r, LK + O::S + N1 * NX + k * NU + NQ)); }
                have_ek = true;
            }
            UPR_FORT(e, N * NU) LK[O::Z + N1 * NX + e] = upr_step(LK[O::Z + N1 * NX + e], a, LK[O::S + N1 * NX + e]);
            UPR_FORT(q, C::NEN) L[O::yN + q] += a * L[O::dyN + q];
            UPR_SYNC();
            ftoc(9);
            toc(15);
        }
        // ---- result: step from the linearisation point
        UPR_FORT(e, N1 * NX) ws[W::dx + e] = LK[O::Z + e] - xs[e];
        UPR_FORT(e, N * NU) ws[W::du + e] = LK[O::Z + N1 * NX + e] - us[e];
        if (prof) UPR_FORT(i, 64) prof[i] += L[O::prf + i];
        if (A.fb) write_feedback(A.fb + (size_t)b * N * NU * NX, status);
        if (A.kkt) {   // multipliers for upr_batch_qp_kkt, in the generic kernel's slot layout
            double* K = A.kkt + (size_t)b * A.kkt_stride;
            const int ni = A.d.ni_stage, o_nu = N1 * NX, o_y = o_nu + N * NE, o_l = o_y + neN, o_t = o_l + N1 * ni;   // (o_t: the rows' slacks, same slots)
            UPR_FORT(e, N1 * NX) K[e] = ws[W::pi + e];
            UPR_FORT(e, N * NE) K[o_nu + e] = ws[W::nu + e];
            UPR_FORT(q, neN) K[o_y + q] = L[O::yN + q];
            UPR_FORT(e, N1 * ni) { K[o_l + e] = 0.0; K[o_t + e] = 1.0; }
            UPR_SYNC();
            // (the box rows out of their parked copy -- store_rows() behind the last step, or the initial point -- not out of the
            //  registers: an export that keeps (t, lam) alive behind the loop moves the register allocation of the loop itself,
            //  measured + 1 % on the headline launch)
            const double* R = G + F::rows + tid();
#pragma unroll
            for (int q = 0; q < C::QX; ++q) {
                const int ix = tid() + q * NT;
                if (ix < C::NXI) {
                    const int k = 1 + ix / NX, i = ix % NX;
                    K[o_t + k * ni + i] = R[(4 * q) * NT]; K[o_t + k * ni + NX + i] = R[(4 * q + 1) * NT];
                    K[o_l + k * ni + i] = R[(4 * q + 2) * NT]; K[o_l + k * ni + NX + i] = R[(4 * q + 3) * NT];
                }
            }
#pragma unroll
            for (int q = 0; q < C::QU; ++q) {
                const int iu = tid() + q * NT;
                if (iu < C::NUI) {
                    const int k = iu / NU, i = iu % NU;
                    K[o_t + k * ni + 2 * NX + i] = R[(4 * C::QX + 4 * q) * NT]; K[o_t + k * ni + 2 * NX + NU + i] = R[(4 * C::QX + 4 * q + 1) * NT];
                    K[o_l + k * ni + 2 * NX + i] = R[(4 * C::QX + 4 * q + 2) * NT]; K[o_l + k * ni + 2 * NX + NU + i] = R[(4 * C::QX + 4 * q + 3) * NT];
                }
            }
            if (NF == 3) UPR_FORT(e, 5 * C::NCI) { const int ic = e / 5, k = ic / NC, ci = ic % NC, sl = k * ni + 2 * NX + 2 * NU + 5 * ci + e % 5; K[o_l + sl] = G[F::cl + e]; K[o_t + sl] = G[F::ct + e]; }
            if (no > 0) UPR_FORT(e, (N - 1) * no) { const int k = 1 + e / no, r = e % no, sl = k * ni + 2 * NX + 2 * NU + C::NP + r, ei = (k - 1) * UPR_QP3_NOMAX + r; K[o_l + sl] = G[F::ol + ei]; K[o_t + sl] = G[F::ot + ei]; }
        }
        if (tid() == 0) {
            double* st = A.stats + (size_t)b * UPR_NSTATS;
            st[1] = it; st[2] = status; st[6] = res[0]; st[7] = res[1]; st[8] = res[2]; st[9] = res[3];
            upr_qp_store_key(A, b, it);
        }
        UPR_SYNC();
    }
};

template <class C>
static UPR_HDI void upr_qp3_solve(const upr_ctx& ctx, const upr_qp_args& A, int b, double* L) {
    upr_qp3<C> S;
    S.solve(ctx, A, b, L);
}

#ifndef UPR_HOST_EMU
// (Which physical wave plays which part was tried as a function of the SIMD each wave sits on -- tools/probe/simd_probe.hip: the
// dispatcher places the waves 0 .. 3 of the workgroups of a CU on SIMDs (0 2 1 3), (2 1 3 0), (1 3 0 2), ... -- so that the two
// sweep waves of co-resident workgroups never share a SIMD: every such assignment measured 6 - 9 % SLOWER than logical =
// physical (1.90 ms -> 2.01 .. 2.07 ms); the sweep waves want to be the two oldest waves of their workgroup.)
template <class C>
__global__ void __launch_bounds__(C::NT, (C::NT <= 256 && C::NB == 1) ? UPR_QP3_OCC1 : 1) upr_qp3_kernel(upr_qp_args A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    upr_ctx ctx; ctx.tid = threadIdx.x; ctx.nt = C::NT;
    upr_qp3_solve<C>(ctx, A, upr_qp_instance(A, blockIdx.x), smem);
}
#endif
