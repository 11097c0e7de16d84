/*
 * crx.h -- C ABI of libcrx, the MI355X-native batched optimal-control solver behind
 * HybridRobotics/car-racing's planner / MPC-CBF hot path.
 *
 * The reference has NO native boundary for this path: the boundary is three Python call sites
 * that build a CasADi `Opti` problem and hand it to IPOPT.  Each entry point below replaces one
 * of them (file:line into /root/reference/car_racing); INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add at exactly those lines.
 *
 *   crx_planner_solve      <- planning/overtake_traj_planner.py:248-379  generate_traj_per_region
 *                             (Opti build :263-334, opti.solve() :359-364, fallback :365-374),
 *                             batched over what the reference forks one process per region for
 *                             (:182-197)
 *   crx_select             <- planning/overtake_traj_planner.py:205-246  region selection
 *   crx_planner_plan       <- planning/overtake_traj_planner.py:162-246  solve_optimization_problem
 *                             (= crx_planner_solve over all regions + crx_select, one launch chain)
 *   crx_cbf_solve          <- control/control.py:476-607 mpccbf  and  control/control.py:251-473
 *                             mpc_multi_agents  (same NLP family; the latter with a per-stage target)
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; all arrays C-contiguous, float64 / int32.
 *   - `*_dev` variants take DEVICE pointers (e.g. torch tensor .data_ptr()) and a stream handle
 *     (hipStream_t passed as void*, NULL = default stream); they enqueue work and return without
 *     synchronising.  The non-`_dev` variants take HOST pointers, stage through library-owned
 *     device buffers and block until the result is back.
 *   - the caller owns every buffer; the library never keeps a pointer after a call returns.
 *   - return value: 0 on success, negative crx_err on a CALL failure (bad argument, no device,
 *     HIP error; text via crx_last_error()).  A problem that does not converge is NOT a call
 *     failure: it is reported per problem in status[] (crx_status), with X/U holding the last
 *     iterate (mpccbf / mpc_multi_agents consume the last iterate, control.py:600-603,458-469) or
 *     the reference's synthetic fall-back trajectory (planner, overtake_traj_planner.py:365-374).
 *   - not fork-safe (a HIP context does not survive fork()); callers must replace the reference's
 *     Process fan-out (:182-197) with ONE batched call from the parent.
 *
 * State order  x = [vx, vy, wz, epsi, s, ey]   (utils/constants.py:1, control/lmpc_helper.py:133-138)
 * Input order  u = [delta, a]                  (system/vehicle_dynamics.py:10-11)
 */
#ifndef CRX_H
#define CRX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRX_VERSION 400 /* 0.4.0: NOT layout-compatible with 0.3.x -- crx_ipm_opts grew by IPOPT's three UNSCALED termination tolerances
                          (`dual_inf_tol`, `constr_viol_tol`, `compl_inf_tol`: "converged" now means IPOPT's complete test, not the scaled error
                          alone) and by `stall_iters` (the stall rule's budget, a kernel constant until 0.3.x; crx_cbf_desc_default picks it and
                          `restore_iters` by problem class).
                          0.3.0: crx_allgather_winners_dev takes the winners' status and the winner record is SURVEY 8e's {int32 flag; int32
                          status; double X[N+1][6]} (same 632 B at N = 12; 0.2.x: the flag as a double, no status) -- the only signature that
                          changed; every other entry point and every struct layout is 0.2.x's.
                          0.2.1: same ABI as 0.2.0; the crash path of the MPC-CBF NLPs (crx_ipm_opts.slack_start = 2) takes the crash start's barrier
                          parameter from its complementarity and keeps the convexified inertia retry between probes -- other iterates, same end
                          points or other KKT points on crash states; every problem off the crash path is bit-identical to 0.2.0.
                          0.2.0: NOT layout-compatible with 0.1.x -- crx_ipm_opts grew by `reach_screen` and `slack_start` (every descriptor
                          embeds it), the process-global switches crx_set_reach_screen / crx_set_cbf_slack_start / crx_set_timing /
                          crx_last_kernel_ms are gone (options travel in the descriptor, timing is an object: crx_timer_*), new status
                          CRX_STALLED (what CRX_INFEASIBLE used to report without a proof), CRX_MAX_OBS 3 -> 6 (CRX_MAX_VEH = 3 for the planner side).
                          (0.1.3: per-obstacle dimensions, plant noise, crx_game_*, crx_comm_* (RCCL), dispatch order, crx_streams_*;
                           0.1.2: infeasibility certificates, *_masked_dev, CRX_SKIPPED, crx_track_prep_dev;
                           0.1.1: restoration phase, CRX_RESTORED, crx_ipm_opts.restore_iters) */
#define CRX_NX 6
#define CRX_NU 2
#define CRX_MAX_N 24       /* horizon limit (reference runs N=10/12; BASELINE configs go to 20) */
#define CRX_MAX_OBS 6      /* obstacles per MPC-CBF NLP (n_obs_max of crx_cbf_desc).  The reference admits any number (control.py:524-562
                              loops over every vehicle in the window); its scenarios race 3 cars (overtake_planner_test.py).  Up to 3 run on
                              the tuned instantiations, 4..6 on ONE generic instantiation per horizon class (correct, slow: twice the
                              register file).  A caller with more keeps the nearest (crx.hostprep.pack_obstacles warns). */
#define CRX_MAX_VEH 6      /* vehicles of interest per planner scenario (n_veh_max of the scene / prep / select descriptors); regions = + 1.  The
                              reference plans around every vehicle get_overtake_flag returns (overtake_traj_planner.py:62-92); its scenarios race
                              3 cars.  [0.3.0: 3 -> 6, = CRX_MAX_OBS: the tracking NLP behind the planner takes the same vehicles as obstacles.]
                              A scene with more keeps the nearest (crx_scene reports `overflow`). */
#define CRX_MAX_REGIONS (CRX_MAX_VEH + 1)
#define CRX_LMPC_MAX_N 16 /* horizon limit of crx_lmpc_solve (its dense factors share one LDS slice) */
#define CRX_MAX_SS 60      /* safe-set points per learning-MPC QP (reference: 44) */

typedef enum crx_err {
    CRX_OK = 0,
    CRX_ERR_ARG = -1,      /* NULL pointer, N out of range, batch < 0, ... */
    CRX_ERR_NO_DEVICE = -2,
    CRX_ERR_HIP = -3,
    CRX_ERR_NOT_INIT = -4
} crx_err;

typedef enum crx_status {
    CRX_CONVERGED = 0,     /* KKT error <= tol */
    CRX_MAX_ITER = 1,      /* iteration cap, or the line search failed at a point that satisfies the constraints; last iterate returned */
    CRX_INFEASIBLE = 2,    /* PROVED infeasible: a bound already violated by the fixed x0 (quirk Q9 / the planner's ey_0 rows), the
                              reachability screen, or the Farkas certificate over the input box (problems whose rows are all linear:
                              planner QPs, 0-obstacle NLPs, first attempts of the learning-MPC QP).  Never a heuristic. */
    CRX_RESTORED = 3,      /* MPC-CBF NLPs only: the solve went through its crash path (restart from a feasible interior point, or
                              the closed-form slack restoration) and used up opts.restore_iters further iterations without
                              converging.  The returned iterate satisfies every CBF row through its slacks sigma but is not optimal. */
    CRX_SKIPPED = 4,       /* *_masked_dev launches only: active[b] == 0, the problem was left alone (outputs untouched) */
    CRX_STALLED = 5        /* gave up WITHOUT a proof at a point that still violates its constraints: no acceptable step (IPOPT:
                              "restoration failed" / "converged to a point of local infeasibility"), or multipliers past 1e12 (IPOPT's
                              divergence heuristic).  libcrx <= 0.1.3 reported these as CRX_INFEASIBLE.  Like every status != 0 it
                              selects the reference's "solver failed" branch (control.py:600-603: keep the last iterate;
                              overtake_traj_planner.py:365-374: the fall-back trajectory). */
} crx_status;

/* Interior-point options.  Defaults (crx_ipm_opts_default) restate IPOPT 3.x defaults that the
 * reference inherits by passing only print options (control.py:593, overtake_traj_planner.py:335). */
typedef struct crx_ipm_opts {
    double tol;            /* 1e-8  convergence tolerance on the scaled KKT error (IPOPT `tol`); since 0.4.0 necessary, not sufficient: see
                              dual_inf_tol / constr_viol_tol / compl_inf_tol at the end of this struct */
    int32_t max_iter;      /* 200   (IPOPT: 3000; capped, status CRX_MAX_ITER beyond) */
    int32_t restore_iters; /* 50 / 25  (by problem class: see stall_iters below) iterations allowed after the first restoration (CRX_RESTORED beyond); < 0: no restoration
                              phase at all (a failed line search then ends the solve, as in libcrx 0.1.0) */
    double mu_init;        /* 0.1   */
    double kappa_eps;      /* 10    barrier sub-problem tolerance factor */
    double kappa_mu;       /* 0.2   linear decrease factor */
    double theta_mu;       /* 1.5   superlinear decrease power */
    double tau_min;        /* 0.99  fraction-to-the-boundary */
    double slack_push;     /* 1e-2  initial slack floor (bound_push) */
    double grad_scale_max; /* 100   gradient-based row scaling target (nlp_scaling_max_gradient) */
    int32_t reach_screen;  /* 1     reachability screens (0 = off: every problem goes through the interior-point iteration).
                              Planner QPs: the inputs are boxed, so ey_j cannot leave [free response -+ reach_j], reach_j =
                              sum_{m<j} |e_ey' A^m B| (delta_max, a_max)'; a region whose ey bound at some stage lies outside that
                              interval by more than 1e-6 has no feasible trajectory -- a proof, independent of the other rows -- and
                              is answered before the iteration is set up: CRX_INFEASIBLE, iters 0, kkt +inf, X = the reference's
                              fall-back trajectory, cost +inf.  Learning-MPC QP: the same for the first attempt's terminal set
                              (x_N = SS lambd needs component c of x_N inside [min_j SS_cj, max_j SS_cj]); the attempt is skipped,
                              the relaxed second attempt runs as it would have. */
    int32_t slack_start;   /* 2     start of the MPC-CBF NLPs whose slacks cannot stay at zero (crash states: the ego inside, or about to
                              enter, an obstacle's safety set; 3..8 % of the BASELINE draws).  The reference passes no initial guess
                              (control.py:593-599), IPOPT starts at u = 0, sigma = 0 and would enter its restoration phase.
                              0: that start only; a solve that stalls is restored in closed form (slacks raised at the current
                                 inputs) -- libcrx 0.1.x behaviour;
                              1: the slacks start at provable lower bounds of their optimal values (0.1.3's optional switch);
                              2: CRASH PATH (default).  (i) A problem whose slacks are PROVABLY positive at every admissible input
                                 (reach of (s, ey) under the boxed inputs) does not start at zero: the best of a 5 x 5 grid of constant
                                 input pairs by f(u) + w sum sigma(u), sigma(u) = the minimal slack cascade for that u, with that
                                 cascade pushed strictly inside -- a feasible interior point; its barrier parameter starts at a tenth of
                                 the point's mean complementarity (not below mu_init), not at mu_init.  (ii) A solve that started at zero and
                                 stalls on violated CBF rows (no acceptable step, jam, 100 iterations still infeasible) restarts ONCE
                                 from such a point instead of being abandoned.  (iii) On the crash path a reduced Hessian of the wrong
                                 inertia is first retried WITHOUT the reverse-convex part of the CBF curvature (positive definite by
                                 construction) before IPOPT's delta_w schedule; after an iteration that needed that, the next ones start
                                 without it and every 4th of such a run tries the exact Hessian first again.  Problems that never enter the crash path are untouched,
                                 bit for bit.  BASELINE configs[1]: 95.7 -> 100 % of the 256 NLPs converge, longest solve 58 -> 38
                                 iterations; configs[3]: 92.5 -> 98.9 % (oracle). 
                              3: EAGER crash path (0.2.1).  As 2, but EVERY problem whose zero start violates a CBF row starts from the point, not
                                 only the provable crash states (the restart of 2 then never fires).  Faster -- BASELINE configs[1]: longest
                                 solve 36 -> 30 iterations, 0.47 -> 0.39 ms per 256 NLPs; configs[3]: 99.2 -> 99.5 % converged -- and further from the
                                 reference: a near-miss that the zero start solves ends at the local minimum IPOPT reaches from zero, and at
                                 another one from the candidate point; the reference's recorded closed loop is reproduced step by step with 2,
                                 not with 3. */
    /* [0.4.0] IPOPT's complete termination test.  IPOPT accepts a point only if the scaled error is <= tol AND three UNSCALED
     * quantities are below their own tolerances (IpOptErrorConvCheck.cpp; the reference runs IPOPT on defaults, control.py:593,
     * overtake_traj_planner.py:335).  "Unscaled" = without the s_d / s_c normalisation of the error measure and with the
     * gradient-based row scaling of the CBF rows undone (rows in the reference's units).  Until 0.3.x libcrx stopped on the scaled
     * error alone: a crash state with multipliers of 1e7..1e9 (s_d of 1e4..1e7) was reported converged with a complementarity of
     * 2.5e-4 (configs[1]) / 5.5e-4 (configs[3]) -- beyond the 1e-4 IPOPT (and north_star) allow. */
    double dual_inf_tol;    /* 1     max |reduced Lagrangian gradient|, unscaled                (IPOPT dual_inf_tol) */
    double constr_viol_tol; /* 1e-4  max |c_j - t_j| with the row scaling undone               (IPOPT constr_viol_tol) */
    double compl_inf_tol;   /* 1e-4  max t_j nu_j                                              (IPOPT compl_inf_tol) */
    int32_t stall_iters;    /* 100 / 50  MPC-CBF NLPs: a solve that started at the reference's zero point and is still infeasible (by 1e-6)
                              after this many iterations is sent to the crash restart / the closed-form restoration.  A
                              kernel constant until 0.3.x (50 in 0.1..0.2, 100 in 0.3).  crx_ipm_opts_default: 100; crx_cbf_desc_default
                              chooses (stall_iters, restore_iters) by problem class: (50, 25) for N <= 12 with at most one obstacle slot --
                              the class of BASELINE configs[1], where a healthy solve is done after 30 iterations --, (100, 50) for every
                              longer horizon or more obstacles (configs[3]: healthy solves of 53..86 iterations).  With the defaults a
                              solve ends after at most stall_iters + 1 + restore_iters (76 / 151) iterations when it went through a
                              restoration, 1 + 3 restore_iters (76 / 151) when it started on the crash path: below max_iter in both. */
    int32_t qp_method;      /* 0     [0.4.0] the solver-kernel problems whose rows are all linear and whose cost is a convex quadratic -- planner region QPs, MPC-CBF
                              NLPs without an obstacle slot -- have ONE solution, so the path to it is free:
                              0: Mehrotra's predictor-corrector (one factorisation and two solves per iteration, no line search): 6.5 iterations
                                 per region QP instead of 10.3, BASELINE configs[2] 0.285 -> 0.247 ms per 4096 QPs;
                              1: IPOPT's monotone barrier + filter line search, as libcrx <= 0.3 ran them (A/B runs).
                              Same starting point, same termination test, same infeasibility proofs either way.  Problems with CBF rows ignore it; so do
                              the learning-MPC QP and the path planner's QPs (measured on the oracle: 15.5 -> 11.5 iterations against a second solve of
                              a third of an iteration -- no gain). */
} crx_ipm_opts;

/* ---- planner region QP (overtake_traj_planner.py:263-334) ------------------------------------ */
typedef struct crx_planner_desc {
    int32_t N;             /* num_horizon_planner (utils/base.py:391) */
    int32_t reserved0;
    double A[36];          /* row-major 6x6, racing_game_param.matrix_A (:272) */
    double B[12];          /* row-major 6x2, racing_game_param.matrix_B (:273) */
    double w_ref;          /* 20   weight on (ey-ey_bez)^2 and (s-s_ref)^2          (:333-334) */
    double w_dey;          /* 30   weight on (ey_k-ey_{k-1})^2, k=2..N-1            (:325-327) */
    double w_prog;         /* 200  weight on -(s_N - s_0)                           (:328) */
    double vx_max;         /* 5.0  vx_{k+1} <= vx_max                               (:276) */
    double delta_max;      /* 0.5                                                   (:280-281) */
    double a_max;          /* 1.5                                                   (:283-284) */
    double dt_ref;         /* 0.1  literal time step in s_ref and the window test   (:296,:330) */
    double fallback_gain;  /* 1.1  speed factor of the fall-back trajectory         (:367-368) */
    crx_ipm_opts opts;
} crx_planner_desc;

/* ---- MPC-CBF NLP (control.py:492-591 / :270-382) --------------------------------------------- */
typedef struct crx_cbf_desc {
    int32_t N;             /* num_horizon (utils/base.py:281) / num_horizon_ctrl (:390) */
    int32_t n_obs_max;     /* leading dimension of the obstacle arrays, <= CRX_MAX_OBS */
    int32_t per_stage_target; /* 0: xt is [batch][6] (mpccbf); 1: xt is [batch][N+1][6] (mpc_multi_agents :373-382) */
    int32_t degree;        /* 6 (control.py:528 / :312); accepted: 2, 4, 6, 8 */
    double A[36];
    double B[12];
    double Q[6];           /* diag(matrix_Q)  (utils/base.py:277 / :384) */
    double R[2];           /* diag(matrix_R)  (utils/base.py:278 / :385) */
    double delta_max;      /* system_param.delta_max (utils/base.py:709) */
    double a_max;
    double v_min;
    double v_max;
    double ey_max;         /* track.width (control.py:585-586) */
    double alpha;          /* 0.8 mpc_cbf_param.alpha / 0.6 literal (control.py:285) */
    double margin;         /* 0.2 (control.py:527) / 0.15 (:311) */
    double l_sum;          /* l_agent + l_obs (control.py:532-535) = 0.4; per-obstacle values: crx_cbf_solve_dims */
    double w_sum;          /* w_agent + w_obs = 0.2 */
    double w_slack;        /* 1e4 (control.py:560,562) */
    crx_ipm_opts opts;
} crx_cbf_desc;

/* ---- learning-MPC QP (control.py:610-730) ------------------------------------------------------ */
typedef struct crx_lmpc_desc {
    int32_t N;             /* lmpc_param.num_horizon (utils/base.py:359), 12 */
    int32_t n_ss_max;      /* leading dimension of ss / qfun / lambda (num_ss_points, utils/base.py:357), <= CRX_MAX_SS */
    double Q[6];           /* diag(matrix_Q)  (utils/base.py:353; zero by default) */
    double R[2];           /* diag(matrix_R)  (:354)  [1, 0.25] */
    double dR[2];          /* diag(matrix_dR) (:356)  [4, 0] */
    double x_track[6];     /* [5,0,0,0,0,0] (control.py:649) */
    double v_max;          /* vx_i <= v_max, i < N      (control.py:658) */
    double ey_max;         /* |ey_i| <= lap_width, i < N (:659-660) */
    double delta_max;      /* (:662-663) */
    double a_max;          /* (:665-666) */
    double w_x0;           /* 1e4: weight of the squared initial-state relaxation, used by the second attempt only
                              (see crx_lmpc_solve) */
    crx_ipm_opts opts;
} crx_lmpc_desc;

/* ---- region selection (overtake_traj_planner.py:205-246) ------------------------------------- */
typedef struct crx_select_desc {
    int32_t N;
    int32_t n_veh_max;     /* leading dimension V of the obstacle arrays; regions R = V + 1 */
    double veh_length;     /* 0.4 */
    double veh_width;      /* 0.2 */
    double lap_length;     /* obstacle s is wrapped `while s > lap_length` (:216-217) */
    double w_prog;         /* 10  (:209) */
    double w_coll;         /* 100 (:223,:237) */
    double w_switch;       /* 100 (:243) */
} crx_select_desc;

/* ---- planner host prep on the device (planner_helper.py:43-153, overtake_traj_planner.py:277-324) ---- */
typedef struct crx_prep_desc {
    int32_t N;             /* num_horizon_planner */
    int32_t n_veh_max;     /* leading dimension V of the per-vehicle arrays; regions R = V + 1 */
    int32_t n_opt;         /* rows of the optimal-trajectory table (data/optimal_traj/xcurv_*.csv) */
    int32_t reserved0;
    double prediction_factor; /* 0.5  racing_game_param.planning_prediction_factor (planner_helper.py:51-53) */
    double lookahead;      /* 4.0  literal in the end point s3 = s0 + factor * max|dv| + 4 (:51-53) */
    double track_width;    /* track.width */
    double lap_length;
    double veh_length;     /* 0.4 */
    double veh_width;      /* 0.2 */
    double safety_margin;  /* 0.15 (overtake_traj_planner.py:288) */
    double dt_ref;         /* 0.1  literal time step of the window test (:296) */
} crx_prep_desc;

/* ---- plant of the simulator (system/vehicle_dynamics.py:4-49, utils/base.py:897-942) ------------------ */
typedef struct crx_plant_desc {
    int32_t n_sub;         /* 100: explicit Euler sub-steps per control step (`while (i+1)*0.001 <= timestep`, base.py:905) */
    int32_t n_seg;         /* rows of the track table point_and_tangent (racing_env.py:17-120) */
    double dt_sub;         /* 0.001 (base.py:899) */
    double lap_length;
    double m, lf, lr, Iz;  /* BicycleDynamicsParam (base.py:686-697): 1.98, 0.125, 0.125, 0.024 */
    double Df, Cf, Bf;     /* front Pacejka: 0.8*m*g/2, 1.25, 1.0 */
    double Dr, Cr, Br;     /* rear  Pacejka */
} crx_plant_desc;

/* ---- overtake PATH planner QP (planning/overtake_path_planner.py:199-318) ------------------------------- */
typedef struct crx_path_desc {
    int32_t N;             /* num_horizon_planner: N+1 lateral offsets per candidate path */
    int32_t reserved0;
    double alpha;          /* racing_game_param.alpha: weight of the Bezier reference; 1 - alpha on the optimal line (:248-250) */
    double w_rate;         /* 100: weight on (ey_j - ey_{j-1})^2 (:252-253) */
    crx_ipm_opts opts;
} crx_path_desc;

/* library management */
int crx_version(void);
/* device >= 0: HIP device ordinal.  There is no CPU back-end in this library: a missing device is
 * an error (CRX_ERR_NO_DEVICE), never a silent fall-back. */
int crx_init(int device);
void crx_shutdown(void);
const char* crx_last_error(void);
int crx_device_count(void);

void crx_ipm_opts_default(crx_ipm_opts* o);
void crx_planner_desc_default(crx_planner_desc* d, int N, const double* A, const double* B);
void crx_cbf_desc_default(crx_cbf_desc* d, int N, int n_obs_max, const double* A, const double* B);
void crx_select_desc_default(crx_select_desc* d, int N, int n_veh_max, double lap_length);
void crx_lmpc_desc_default(crx_lmpc_desc* d, int N, int n_ss_max);
void crx_path_desc_default(crx_path_desc* d, int N, double alpha);
void crx_plant_desc_default(crx_plant_desc* d, int n_seg, double lap_length);
void crx_prep_desc_default(crx_prep_desc* d, int N, int n_veh_max, int n_opt, double track_width, double lap_length);

/*
 * Planner region QPs.  One problem per (scenario, region).
 *   x0      [batch][6]      ego.xcurv, raw (quirk Q5: the start-line-wrapped copy is only used by
 *                           the host when it builds ey_lb)
 *   bez_s   [batch][N+1]    Bezier reference polyline, s coordinates   (bezier_xcurvs[r,:,0])
 *   bez_ey  [batch][N+1]    ... ey coordinates                         (bezier_xcurvs[r,:,1])
 *   ey_lb   [batch][N]      effective lower bound on ey_k, k=0..N-1: max of -(ey_ub) and the
 *                           neighbour constraints active at k (:286-324, quirk Q2)
 *   ey_ub   [batch]         track.width - 0.5*veh_width (:277-278)
 *   X       [batch][N+1][6] out: solution, or fall-back trajectory when status != 0
 *   U       [batch][N][2]   out: inputs (zeros with the fall-back)
 *   cost    [batch]         out: sol.value(cost) (:364); +inf with the fall-back (:374)
 *   status  [batch]         out: crx_status
 *   kkt     [batch]         out: final scaled KKT error
 *   iters   [batch]         out: interior-point iterations
 */
int crx_planner_solve(const crx_planner_desc* d, int batch, const double* x0, const double* bez_s,
                      const double* bez_ey, const double* ey_lb, const double* ey_ub, double* X,
                      double* U, double* cost, int32_t* status, double* kkt, int32_t* iters);
int crx_planner_solve_dev(const crx_planner_desc* d, int batch, const double* x0,
                          const double* bez_s, const double* bez_ey, const double* ey_lb,
                          const double* ey_ub, double* X, double* U, double* cost, int32_t* status,
                          double* kkt, int32_t* iters, void* stream);

/*
 * Region selection.  Scenario i has n_veh[i] <= V sorted vehicles and n_veh[i]+1 regions.
 *   X        [n_scen][V+1][N+1][6]  per-region trajectories (solution_xvar, transposed)
 *   obs_s    [n_scen][V][N+1]       predicted s of sorted vehicle v (obs_infos[name][4,:])
 *   obs_ey   [n_scen][V][N+1]
 *   old_flag [n_scen]               previous direction_flag, -1 for None (:238-243)
 *   flag     [n_scen]               out: direction_flag (first arg-min, :244)
 *   sel_cost [n_scen][V+1]          out: cost_selection (regions >= n_veh+1: +inf)
 *   best_X   [n_scen][N+1][6]       out: traj_xcurv (:245)
 */
int crx_select(const crx_select_desc* d, int n_scen, const int32_t* n_veh, const double* X,
               const double* obs_s, const double* obs_ey, const int32_t* old_flag, int32_t* flag,
               double* sel_cost, double* best_X);
int crx_select_dev(const crx_select_desc* d, int n_scen, const int32_t* n_veh, const double* X,
                   const double* obs_s, const double* obs_ey, const int32_t* old_flag,
                   int32_t* flag, double* sel_cost, double* best_X, void* stream);

/*
 * One planner step for n_scen scenarios: all V+1 region QPs of every scenario (crx_planner_solve with
 * batch = n_scen*(V+1), scenario-major) followed by crx_select on the same stream, X staying on the
 * device in between.  Replaces solve_optimization_problem (overtake_traj_planner.py:162-246).
 * Arrays as in crx_planner_solve (leading dimension n_scen*(V+1)) and crx_select.
 */
int crx_planner_plan(const crx_planner_desc* d, const crx_select_desc* sd, int n_scen, const double* x0,
                     const double* bez_s, const double* bez_ey, const double* ey_lb, const double* ey_ub,
                     const int32_t* n_veh, const double* obs_s, const double* obs_ey, const int32_t* old_flag,
                     double* X, double* U, double* cost, int32_t* status, double* kkt, int32_t* iters,
                     int32_t* flag, double* sel_cost, double* best_X);
int crx_planner_plan_dev(const crx_planner_desc* d, const crx_select_desc* sd, int n_scen, const double* x0,
                         const double* bez_s, const double* bez_ey, const double* ey_lb, const double* ey_ub,
                         const int32_t* n_veh, const double* obs_s, const double* obs_ey,
                         const int32_t* old_flag, double* X, double* U, double* cost, int32_t* status,
                         double* kkt, int32_t* iters, int32_t* flag, double* sel_cost, double* best_X,
                         void* stream);
/* masked launch (see crx_cbf_solve_masked_dev): active [n_scen] on the device, 0 = none of this scenario's region QPs is
 * solved (their status = CRX_SKIPPED, X / U / cost untouched); the selection still runs for every scenario and, for a
 * skipped one, works on whatever X holds -- the caller ignores flag / best_X of the scenarios it masked out. */
int crx_planner_plan_masked_dev(const crx_planner_desc* d, const crx_select_desc* sd, int n_scen, const int32_t* active,
                                const double* x0, const double* bez_s, const double* bez_ey, const double* ey_lb,
                                const double* ey_ub, const int32_t* n_veh, const double* obs_s, const double* obs_ey,
                                const int32_t* old_flag, double* X, double* U, double* cost, int32_t* status, double* kkt,
                                int32_t* iters, int32_t* flag, double* sel_cost, double* best_X, void* stream);

/*
 * Planner host prep on the device: per scenario, the cubic Bezier reference of
 * every region (planner_helper.get_bezier_control_points :43-135 and get_bezier_curve :138-153, sampled
 * as overtake_traj_planner.py:105-111) and the per-stage ey bounds with the obstacle windows
 * (overtake_traj_planner.py:277-324, quirks Q2/Q5), written directly in the layout crx_planner_solve reads.
 *   x_wrapped [S][6]   start-line-wrapped ego state (xcurv_ego: control points :49,95 and window test :296)
 *   x_raw     [S][6]   ego.xcurv as stored on the vehicle (becomes x0 of every region, :266, quirk Q5)
 *   n_veh     [S]      vehicles of interest, 0..V
 *   veh_info  [S][V][3] rows (s, max ey over the prediction, min ey) in ITERATION order (quirk Q4, :87-92)
 *   max_dv    [S]      agent_info.max_delta_v (planner_helper.py:177-201)
 *   obs_s, obs_ey [S][V][N+1]  predictions of the SORTED vehicles
 *   opt_s, opt_ey [n_opt]      columns 4 and 5 of the optimal-trajectory table (shared by all scenarios;
 *                              precondition: the end point s3 lies inside the table, else the reference raises)
 * outputs, S*(V+1) leading dimension, scenario-major: x0 [.][6], bez_s, bez_ey [.][N+1], ey_lb [.][N], ey_ub [.]
 */
int crx_planner_prep(const crx_prep_desc* d, int n_scen, const double* x_wrapped, const double* x_raw,
                     const int32_t* n_veh, const double* veh_info, const double* max_dv, const double* obs_s,
                     const double* obs_ey, const double* opt_s, const double* opt_ey, double* x0, double* bez_s,
                     double* bez_ey, double* ey_lb, double* ey_ub);
int crx_planner_prep_dev(const crx_prep_desc* d, int n_scen, const double* x_wrapped, const double* x_raw,
                         const int32_t* n_veh, const double* veh_info, const double* max_dv, const double* obs_s,
                         const double* obs_ey, const double* opt_s, const double* opt_ey, double* x0,
                         double* bez_s, double* bez_ey, double* ey_lb, double* ey_ub, void* stream);

/*
 * The front of OvertakeTrajPlanner on the device (remainder): per scenario
 *   which vehicles are of interest   get_overtake_flag -> planner_helper.check_ego_agent_distance (:218-266)
 *   the reference's partial ey "sort" overtake_traj_planner.py:66-76 (quirk Q3: every vehicle is compared with the CURRENT FIRST only)
 *   veh_info rows (s, max ey, min ey)  :87-92, in ITERATION order although consumed as if sorted (quirk Q4)
 *   agent_info.max_delta_v            planner_helper.get_agent_info (:177-201)
 *   predictions of the sorted vehicles  obs_infos[name][4:6, :] (:77-85)
 * Outputs are exactly the vehicle inputs of crx_planner_prep and crx_select.
 *   ego_xcurv [S][6]         the ego vehicle's state (the vehicle object's xcurv: both tests read it)
 *   n_all     [S]            vehicles in the scenario besides the ego, 0..n_all_max, in the reference's dict order
 *   veh_xcurv [S][VA][6]     their current states;  pred_s, pred_ey [S][VA][N+1]  their predictions (get_trajectory_nsteps rows 4, 5)
 *   n_veh [S]                out: vehicles of interest (the planner runs iff > 0);  overflow [S]: of-interest vehicles beyond
 *                            n_veh_max that were dropped (the reference has no limit; libcrx plans around at most CRX_MAX_VEH).
 *                            On overflow the n_veh_max vehicles NEAREST to the ego along the closed lap are kept (ties to the
 *                            earlier one), in dict order -- the closest car is never the one dropped.
 *   order [S][V]             out: vehicle index (0..n_all-1) of sorted vehicle k, -1 beyond n_veh
 *   veh_info [S][V][3], max_dv [S], obs_s, obs_ey [S][V][N+1]   out
 */
typedef struct crx_scene_desc {
    int32_t N;
    int32_t n_all_max;       /* VA: leading dimension of the vehicle arrays */
    int32_t n_veh_max;       /* V <= CRX_MAX_VEH */
    int32_t reserved0;
    double safety_factor;    /* 4.5  RacingGameParam.safety_factor */
    double prediction_factor;/* 0.5  planning_prediction_factor */
    double veh_length;       /* 0.4  ego.param.length */
    double lap_length;
} crx_scene_desc;
void crx_scene_desc_default(crx_scene_desc* d, int N, int n_all_max, int n_veh_max, double lap_length);
int crx_planner_scene(const crx_scene_desc* d, int n_scen, const double* ego_xcurv, const int32_t* n_all, const double* veh_xcurv,
                      const double* pred_s, const double* pred_ey, int32_t* n_veh, int32_t* overflow, int32_t* order,
                      double* veh_info, double* max_dv, double* obs_s, double* obs_ey);
int crx_planner_scene_dev(const crx_scene_desc* d, int n_scen, const double* ego_xcurv, const int32_t* n_all,
                          const double* veh_xcurv, const double* pred_s, const double* pred_ey, int32_t* n_veh,
                          int32_t* overflow, int32_t* order, double* veh_info, double* max_dv, double* obs_s, double* obs_ey,
                          void* stream);

/*
 * One control step of the plant for a batch of vehicles: n_sub explicit Euler
 * sub-steps of the dynamic bicycle with Pacejka tyres (system/vehicle_dynamics.py:4-49) in global and
 * curvilinear coordinates, the curvature looked up from the track table at every sub-step
 * (racing_env.py:225-246), exactly DynamicBicycleModel.forward_dynamics (base.py:897-942).  The reference's bounded
 * process noise (:929-939; ON by default, car_racing/tests/overtake_planner_test.py:41-42 makes zero-noise opt-in) is
 * applied by crx_plant_step_noise_dev from standard-normal draws the CALLER supplies (its RNG, its seed):
 *   noise_z [batch][3] or NULL   z ~ N(0,1) per vehicle for (vx, vy, wz); the kernel forms clip(0.01 z0, +-0.05),
 *                                clip(0.01 z1, +-0.1), clip(0.005 z2, +-0.05) and adds HALF of each to the curvilinear
 *                                velocities only (xglob_next keeps the noise-free ones, as the reference does).  NULL = zero noise.
 *   track [n_seg][6] rows (x, y, psi, s_start, length, curvature);  xglob, xcurv [batch][6];  u [batch][2]
 * crx_plant_step_wrap_dev additionally applies the lap bookkeeping of ModelBase.update_memory (base.py:795-819):
 * s > lap_length -> s -= lap_length, laps[b] += 1 (laps may be NULL).  u_stride = doubles between the inputs of
 * consecutive vehicles (2 for a packed array, 2N to read u_0 straight out of a crx_cbf_solve U array).
 */
int crx_plant_step_wrap_dev(const crx_plant_desc* d, int batch, const double* track, const double* xglob,
                            const double* xcurv, const double* u, int u_stride, double* xglob_next, double* xcurv_next,
                            int32_t* laps, void* stream);
int crx_plant_step_noise_dev(const crx_plant_desc* d, int batch, const double* track, const double* xglob,
                             const double* xcurv, const double* u, int u_stride, const double* noise_z, double* xglob_next,
                             double* xcurv_next, int32_t* laps, void* stream);
int crx_plant_step(const crx_plant_desc* d, int batch, const double* track, const double* xglob, const double* xcurv,
                   const double* u, double* xglob_next, double* xcurv_next);
int crx_plant_step_dev(const crx_plant_desc* d, int batch, const double* track, const double* xglob,
                       const double* xcurv, const double* u, double* xglob_next, double* xcurv_next, void* stream);

/*
 * Overtake PATH planner: one 1-D QP per candidate region
 * (overtake_path_planner.py:199-318).  Variables ey_0..ey_N along fixed s samples;
 *   cost  sum_j (1-alpha)(ey_j - opt_j)^2 + alpha (ey_j - bez_j)^2  +  w_rate sum_{j>=1} (ey_j - ey_{j-1})^2      (:246-253)
 *   ey_0 = e0 (:258), ey_N = eN (:261), lb_j <= ey_j <= ub_j (track box :263-264 and the neighbours' side rows :265-297,
 *   merged by the caller; a row the reference skips is +-inf)
 *   opt, bez, lb, ub [batch][N+1];  e0, eN [batch];  E [batch][N+1];  cost [batch] (inf unless converged, :311)
 * status CRX_INFEASIBLE: an end point violates its own box or lb_j > ub_j somewhere (the reference then keeps IPOPT's
 * debug iterate, :310; here E holds the end points joined by the box-clipped Bezier reference).
 */
int crx_path_solve(const crx_path_desc* d, int batch, const double* opt, const double* bez, const double* lb,
                   const double* ub, const double* e0, const double* eN, double* E, double* cost, int32_t* status,
                   double* kkt, int32_t* iters);
int crx_path_solve_dev(const crx_path_desc* d, int batch, const double* opt, const double* bez, const double* lb,
                       const double* ub, const double* e0, const double* eN, double* E, double* cost, int32_t* status,
                       double* kkt, int32_t* iters, void* stream);

/*
 * Obstacle arrays of control.mpccbf for scripted cars, built on the device (used by device-resident race loops):
 * predictions s_o(t + j dt) = v_o (t + j dt) + s0_o, ey_o constant, j = 0..N, from the cars' clock t
 * (NoDynamicsModel.get_trajectory_nsteps, utils/base.py:879-886, quirk Q6); window filter
 * dist_obs - 2 vx < dist_ego < dist_obs + 2 vx on the lap-folded positions and lap offsets
 * (num_cycle_ego - num_cycle_obs) * lap_length (control/control.py:499-523, 538-540); kept cars packed to the
 * front in vehicle order, the rest zero.  Outputs are exactly the obstacle inputs of crx_cbf_solve.
 *   xcurv [batch][6];  car_s0, car_v, car_ey [batch][V];  obs_s, obs_ey [batch][V][N+1];  lap_off [batch][V];  n_obs [batch]
 */
int crx_cbf_prep_dev(int N, int V, double lap_length, double t, double dt, double safety_time, int batch,
                     const double* xcurv, const double* car_s0, const double* car_v, const double* car_ey,
                     double* obs_s, double* obs_ey, double* lap_off, int32_t* n_obs, void* stream);

/*
 * Inputs of control.mpc_multi_agents' NLP (crx_cbf_solve with per_stage_target = 1) from the planner's outputs, on the device
 * (device-resident race loops): per-stage targets xt_i = [vx_0, 0, 0, 0, 0, f_traj(clip(s_0 + dt_ref i vx_0))] from the
 * selected trajectory (control/control.py:373-382; f_traj = linear interpolation of its (s, ey) samples) and the obstacle
 * arrays: the sorted vehicles' predictions filtered by the +-safety_time*vx window on the lap-folded positions, lap
 * offsets, kept vehicles packed to the front (control.py:293-309, 311-319).  Horizon of the controller = horizon of the
 * planner here (the reference's defaults: 10 and 10).
 *   x [batch][6];  n_veh [batch];  obs_s_in, obs_ey_in [batch][V][N+1] (crx_planner_scene order);  traj [batch][N+1][6] (best_X)
 *   xt [batch][N+1][6];  obs_s, obs_ey [batch][V][N+1];  lap_off [batch][V];  n_obs [batch]
 */
int crx_track_prep_dev(int N, int V, double lap_length, double safety_time, double dt_ref, int batch, const double* x,
                       const int32_t* n_veh, const double* obs_s_in, const double* obs_ey_in, const double* traj, double* xt,
                       double* obs_s, double* obs_ey, double* lap_off, int32_t* n_obs, void* stream);

/*
 * Learning-MPC QPs: control.lmpc (control.py:610-730) after its safe-set
 * selection (:625-639), one problem per batch entry.  LTV affine model x_{i+1} = A_i x_i + B_i u_i + C_i
 * as estimated by the caller (utils/base.py:585-622).
 *   x0 [batch][6], u_old [batch][2]; A [batch][N][36], B [batch][N][12], C [batch][N][6] (row-major);
 *   ss [batch][6][n_ss_max] (columns = safe-set points, :636-638), qfun [batch][n_ss_max], n_ss [batch] in
 *   [1, n_ss_max].
 *   X [batch][N+1][6], U [batch][N][2], lambda [batch][n_ss_max], cost [batch] (the reference's cost :698).
 * status: CRX_CONVERGED = KKT point of the reference's QP (terminal state inside the safe-set hull).
 *   The reference pins its terminal slack to zero (:694-695), so its QP is infeasible whenever the
 *   regression model cannot reach the hull from xcurv; the reference then applies IPOPT's restoration
 *   state (:711-722), which is solver-internal: a least-violation point of ALL equalities of the
 *   full-space problem.  On the recorded infeasible instances the cheapest such violation is a ~1e-2
 *   shift of the initial-state equality (:650).  libcrx therefore repeats a failed problem with that
 *   equality relaxed, x_0 = xcurv + w, cost += w_x0 * w'w, terminal constraint kept; X is the plan from
 *   xcurv + w, U its inputs, status CRX_INFEASIBLE when the first attempt ended by a PROOF (a bound the fixed x_0 violates,
 *   the terminal-set reachability screen, the Farkas certificate over input box x unit simplex) and CRX_STALLED when it ended
 *   without one (multiplier divergence, iteration cap, stagnation on the noise floor, no acceptable step) [0.3.0; <= 0.2.1
 *   reported both as CRX_INFEASIBLE].  A relaxed attempt that does not converge either keeps its own status (1 / 5).
 */
int crx_lmpc_solve(const crx_lmpc_desc* d, int batch, const double* x0, const double* u_old, const double* A,
                   const double* B, const double* C, const double* ss, const double* qfun, const int32_t* n_ss,
                   double* X, double* U, double* lambda, double* cost, int32_t* status, double* kkt,
                   int32_t* iters);
int crx_lmpc_solve_dev(const crx_lmpc_desc* d, int batch, const double* x0, const double* u_old, const double* A,
                       const double* B, const double* C, const double* ss, const double* qfun,
                       const int32_t* n_ss, double* X, double* U, double* lambda, double* cost,
                       int32_t* status, double* kkt, int32_t* iters, void* stream);
/* masked launch: see crx_cbf_solve_masked_dev */
int crx_lmpc_solve_masked_dev(const crx_lmpc_desc* d, int batch, const int32_t* active, const double* x0, const double* u_old,
                              const double* A, const double* B, const double* C, const double* ss, const double* qfun,
                              const int32_t* n_ss, double* X, double* U, double* lambda, double* cost, int32_t* status,
                              double* kkt, int32_t* iters, void* stream);

/*
 * Host work in front of the learning-MPC QP, on the device (second half): per race, the
 * N local LTV stage models of LMPCRacingGame.estimate_ABC (utils/base.py:585-622 -> control/lmpc_helper.py:26-189:
 * kernel-weighted least squares for the vx / vy / wz rows from the two previous laps, analytic Jacobian of the Euler
 * step for the epsi / s / ey rows) and the safe-set points and cost-to-go control.lmpc selects (control.py:625-639 ->
 * lmpc_helper.py:278-293).  Outputs are exactly the A, B, C, ss, qfun inputs of crx_lmpc_solve.
 *   ss_xcurv [batch][n_laps][n_points][6], u_ss [batch][n_laps][n_points][2], qfun [batch][n_laps][n_points]
 *            the sampled safe set (LMPCRacingGame.ss_xcurv / u_ss / Qfun, stored lap-major here)
 *   time_ss  [batch][n_laps]   samples per stored lap;  iter [batch]   index of the running lap (>= 2)
 *   x        [batch][6]        current state (start-line wrapped, as LMPCRacingGame.calc_input passes it)
 *   lin_points [batch][N+1][6], lin_input [batch][N][2]   linearisation points; with from_plan != 0 they are the X and U
 *            of the previous crx_lmpc_solve and are shifted by one stage here (control.py:726-728)
 *   track    [n_seg][6]        rows (x, y, psi, s_start, length, curvature)
 *   A [batch][N][36], B [batch][N][12], C [batch][N][6];  ss_sel [batch][6][M], q_sel [batch][M], M = n_ss_laps * n_ss_per_lap
 *   status   [batch]           0, or 1 if a stage's normal matrix was singular (no stored sample near its linearisation
 *                              point; the reference's cvxopt raises there).  The three regression rows of such a stage are
 *                              left UNTOUCHED in A, B, C: a device-resident loop that reuses its buffers keeps the previous
 *                              step's model of the stage.
 * The reference's normal equations reach condition numbers of 3e11: coefficients agree with the reference's to ~1e-5
 * relative only (any two LAPACK routes differ that much); what the models predict at their own query points agrees tightly.
 */
typedef struct crx_lmpcprep_desc {
    int32_t N;              /* 12   LMPCRacingParam.num_horizon */
    int32_t n_points;       /* rows per lap in the safe-set arrays */
    int32_t n_laps;         /* laps per race in the safe-set arrays */
    int32_t n_ss_per_lap;   /* 22   num_ss_points / num_ss_iter (control.py:627-632) */
    int32_t n_ss_laps;      /* 2    num_ss_iter */
    int32_t max_neighbours; /* 40   utils/base.py:605 */
    int32_t n_seg;          /* rows of the track table */
    int32_t shift;          /* 0    LMPCRacingParam.shift */
    double bandwidth;       /* 5.0  lmpc_helper.py:42 */
    double scale[5];        /* 0.1, 1, 1, 1, 1   feature scaling of the distance (:49-57) */
    double dt;              /* control period */
    double lap_length;
} crx_lmpcprep_desc;
void crx_lmpcprep_desc_default(crx_lmpcprep_desc* d, int N, int n_points, int n_laps, int n_seg, double dt, double lap_length);
int crx_lmpc_prep(const crx_lmpcprep_desc* d, int batch, const double* ss_xcurv, const double* u_ss, const double* qfun,
                  const int32_t* time_ss, const int32_t* iter, const double* x, const double* lin_points,
                  const double* lin_input, int from_plan, const double* track, double* A, double* B, double* C,
                  double* ss_sel, double* q_sel, int32_t* status);
int crx_lmpc_prep_dev(const crx_lmpcprep_desc* d, int batch, const double* ss_xcurv, const double* u_ss, const double* qfun,
                      const int32_t* time_ss, const int32_t* iter, const double* x, const double* lin_points,
                      const double* lin_input, int from_plan, const double* track, double* A, double* B, double* C,
                      double* ss_sel, double* q_sel, int32_t* status, void* stream);
/* masked launch: see crx_cbf_solve_masked_dev (status[b] = CRX_SKIPPED for the races left alone) */
int crx_lmpc_prep_masked_dev(const crx_lmpcprep_desc* d, int batch, const int32_t* active, const double* ss_xcurv, const double* u_ss,
                             const double* qfun, const int32_t* time_ss, const int32_t* iter, const double* x, const double* lin_points,
                             const double* lin_input, int from_plan, const double* track, double* A, double* B, double* C,
                             double* ss_sel, double* q_sel, int32_t* status, void* stream);
/* LMPCRacingGame.add_point (utils/base.py:624-629): the running lap extends the previous lap's safe set past the finish
 * line -- row time_ss[iter-1] + step + 1 of lap iter-1 becomes x + (0,0,0,0,lap_length,0) / u.  Device-resident race loops. */
int crx_lmpc_addpoint_dev(const crx_lmpcprep_desc* d, int batch, double* ss_xcurv, double* u_ss, const int32_t* time_ss,
                          const int32_t* iter, const int32_t* step, const double* x, const double* u, int u_stride,
                          void* stream);
/* LMPCRacingGame.add_trajectory (utils/base.py:631-656), for the races of a device-resident loop that have just crossed the
 * finish line (crossed[b] != 0; the others are left alone): the running lap, logged as the simulator logs it
 * (update_memory, base.py:795-819: log_x [batch][n_points][6] with n_log[b] states, the last one the crossing state with
 * s > lap_length; log_u [batch][n_points][2] with n_log[b] - 1 inputs), becomes lap iter[b] of the race's safe set --
 * states, inputs, time_ss = n_log - 1, Qfun = compute_cost (lmpc_helper.py:11-23) + the reference's count-down pass over
 * the whole column (:647-649) -- then iter[b] += 1, step[b] = 0 (time_in_iter), and the log restarts with x[b] (the wrapped
 * state the new lap starts from), n_log[b] = 1.  status[b] = 1: the race's safe set is full (iter == n_laps), nothing stored. */
int crx_lmpc_addtraj_dev(const crx_lmpcprep_desc* d, int batch, const int32_t* crossed, double* log_x, const double* log_u,
                         int32_t* n_log, double* ss_xcurv, double* u_ss, double* qfun, int32_t* time_ss, int32_t* iter,
                         int32_t* step, const double* x, int32_t* status, void* stream);

/*
 * MPC-CBF NLPs.
 *   x0      [batch][6]
 *   xt      [batch][6] or [batch][N+1][6]   tracking target(s) (d->per_stage_target)
 *   obs_s   [batch][n_obs_max][N+1]         obstacle prediction rows 4 of obs_traj
 *   obs_ey  [batch][n_obs_max][N+1]         rows 5
 *   lap_off [batch][n_obs_max]              (num_cycle_ego - num_cycle_obs) * lap_length; applied
 *                                           to h_i only, not to h_{i+1} (quirk Q1, control.py:539-542)
 *   n_obs   [batch]                         obstacles inside the +-2*vx window (control.py:520-523)
 *   X [batch][N+1][6], U [batch][N][2], sigma [batch][n_obs_max][N+1] (cbf_slack), cost, status,
 *   kkt, iters as above.  No fall-back: the last iterate is returned (control.py:600-603).
 */
int crx_cbf_solve(const crx_cbf_desc* d, int batch, const double* x0, const double* xt,
                  const double* obs_s, const double* obs_ey, const double* lap_off,
                  const int32_t* n_obs, double* X, double* U, double* sigma, double* cost,
                  int32_t* status, double* kkt, int32_t* iters);
int crx_cbf_solve_dev(const crx_cbf_desc* d, int batch, const double* x0, const double* xt,
                      const double* obs_s, const double* obs_ey, const double* lap_off,
                      const int32_t* n_obs, double* X, double* U, double* sigma, double* cost,
                      int32_t* status, double* kkt, int32_t* iters, void* stream);
/* Per-obstacle dimensions.  The reference takes (l_obs, w_obs) from every obstacle vehicle's own CarParam (control.py:529-535);
 * crx_cbf_desc carries ONE pair (l_sum, w_sum) for the common case of identical cars.  obs_dims [batch][n_obs_max][2] =
 * (l_agent + l_obs, w_agent + w_obs) of every obstacle slot of every problem overrides it (slots >= n_obs[b] are not read);
 * NULL = the descriptor's pair for all.  `active` as in the masked launch below (NULL: all).  Precondition: positive, finite
 * entries -- the host-pointer entry point checks and fails the call; the *_dev variants cannot (device memory) and let a
 * non-positive or non-finite entry fall back to the descriptor's pair. */
int crx_cbf_solve_dims(const crx_cbf_desc* d, int batch, const double* x0, const double* xt, const double* obs_s,
                       const double* obs_ey, const double* lap_off, const int32_t* n_obs, const double* obs_dims, double* X,
                       double* U, double* sigma, double* cost, int32_t* status, double* kkt, int32_t* iters);
int crx_cbf_solve_dims_dev(const crx_cbf_desc* d, int batch, const int32_t* active, const double* x0, const double* xt,
                           const double* obs_s, const double* obs_ey, const double* lap_off, const int32_t* n_obs,
                           const double* obs_dims, double* X, double* U, double* sigma, double* cost, int32_t* status, double* kkt,
                           int32_t* iters, void* stream);
/* Dispatch order.  The workgroups of a launch start in launch order and a launch ends with its slowest problem: when the problems
 * that need many iterations are listed first -- a closed loop knows them: the iteration counts of the previous control step -- the
 * launch no longer waits for a straggler that started last (BASELINE configs[3], 16384 NLPs of 8..166 iterations on 1024 resident
 * slots: list scheduling in index order 398 iteration-units, longest first 338).  order [batch] int32 on the device = a permutation of
 * 0..batch-1: workgroup i solves problem order[i]; results land at the problem's own index; NULL = index order.  Entries are
 * clamped to [0, batch); a non-permutation solves some problems twice and others not at all (the caller's business).
 * crx_cbf_solve_ordered_dev is the superset entry point (mask, order, per-obstacle dimensions; each may be NULL). */
int crx_cbf_solve_ordered_dev(const crx_cbf_desc* d, int batch, const int32_t* active, const int32_t* order, const double* x0,
                              const double* xt, const double* obs_s, const double* obs_ey, const double* lap_off,
                              const int32_t* n_obs, const double* obs_dims, double* X, double* U, double* sigma, double* cost,
                              int32_t* status, double* kkt, int32_t* iters, void* stream);
/* The order itself, computed on the device (one launch, a stable counting sort; equal keys keep index order):
 *   crx_order_longest_first_dev  from the iteration counts of the previous solve of these problems (iters [batch], e.g. the iters
 *                                output of the previous control step), longest first;
 *   crx_cbf_order_dev            with no previous solve, from the problem's own inputs (the arrays of crx_cbf_solve_dev): the
 *                                barrier h = (ds/l)^degree + (dey/w)^degree - 1 - margin (control.py:529-537) of the START state
 *                                and of the un-steered PATH (s_0 + j A[4][0] vx_0 along the target ey).  First the cars that start
 *                                inside a safety ellipse (deepest first: the NLPs that need the restoration phase), then those
 *                                whose path enters one, then the rest, nearest first.
 * Problems with active[b] == 0 go last (active may be NULL).  Measured on BASELINE configs[3] (16384 NLPs, one launch): DESIGN.md
 * section 5.6. */
int crx_order_longest_first_dev(int batch, const int32_t* iters, const int32_t* active, int32_t* order, void* stream);
int crx_cbf_order_dev(const crx_cbf_desc* d, int batch, const int32_t* active, const double* x0, const double* xt,
                      const double* obs_s, const double* obs_ey, const double* lap_off, const int32_t* n_obs, const double* obs_dims,
                      int32_t* order, void* stream);
int crx_lmpc_solve_ordered_dev(const crx_lmpc_desc* d, int batch, const int32_t* active, const int32_t* order, const double* x0,
                               const double* u_old, const double* A, const double* B, const double* C, const double* ss,
                               const double* qfun, const int32_t* n_ss, double* X, double* U, double* lambda, double* cost,
                               int32_t* status, double* kkt, int32_t* iters, void* stream);
/* Masked launches (device-resident loops in which every problem of the batch takes ONE of several branches per step, e.g.
 * LMPCRacingGame.calc_input, utils/base.py:456-583: overtake planner + tracking NLP, or learning-MPC): active [batch]
 * int32 on the device, 0 = this problem is not part of the launch -- its wavefront returns at once, status[b] =
 * CRX_SKIPPED, iters[b] = 0, every other output of problem b keeps its previous contents.  active == NULL: all. */
int crx_cbf_solve_masked_dev(const crx_cbf_desc* d, int batch, const int32_t* active, const double* x0, const double* xt,
                             const double* obs_s, const double* obs_ey, const double* lap_off, const int32_t* n_obs, double* X,
                             double* U, double* sigma, double* cost, int32_t* status, double* kkt, int32_t* iters, void* stream);

/*
 * Device-resident racing-game loop: the bookkeeping of LMPCRacingGame.calc_input (utils/base.py:456-583)
 * and of the simulator between the solver launches, so that a control step of B races is a fixed sequence of libcrx launches with
 * nothing returning to the host (crx.montecarlo.GameLaps: traffic -> scene -> masks -> prep -> plan -> track prep -> tracking NLP |
 * regression -> learning-MPC QP -> commit -> add_point -> plant -> log -> add_trajectory).
 *   crx_game_traffic_dev  scripted cars s(t) = v t + s0, ey(t) = ey (NoDynamicsModel, base.py:847-890) at their clock t: states
 *                         veh_xcurv [B][n_cars][6] (s wrapped once past the line like update_memory keeps it) and predictions
 *                         pred_s, pred_ey [B][n_cars][N+1] at t + j dt (unwrapped: quirk Q6) -- the inputs of crx_planner_scene_dev
 *   crx_game_masks_dev    m_overtake = (n_veh > 0), m_lmpc = 1 - m_overtake: the `active` masks of the masked launches; with
 *                         overflow / overflow_seen (both may be NULL) the scene stage's dropped-vehicle counts are accumulated
 *                         per race, so that a sweep can report whether the n_veh_max slots ever were too few
 *   crx_game_commit_dev   after the solves, per race from the branch it is in (overtake == NULL: every race drives by learning MPC):
 *                         u [B][2] the input to apply (u_0 of the tracking NLP U_track [B][Np][2] or of the learning-MPC QP
 *                         U_lmpc [B][N][2]); u_prev <- u_old; learning-MPC branch only: u_old <- u_0, lin_points [B][N+1][6] /
 *                         lin_input [B][N][2] <- the plan X_lmpc / U_lmpc shifted by one stage, last stage repeated
 *                         (control.py:726-728), step_no += 1; addpoint_step = the step index crx_lmpc_addpoint_dev takes (negative in
 *                         the overtake branch: no point is added, utils/base.py:546-551); old_flag <- flag or -1 (may be NULL)
 *   crx_game_log_dev      after the plant: crossed [B] = lap counter advanced; the new state (s unwrapped if it crossed) and the
 *                         applied input appended to the race's lap log log_x [B][n_points][6], log_u [B][n_points][2], n_log += 1
 *                         (ModelBase.update_memory) -- the inputs of crx_lmpc_addtraj_dev
 * A race keeps n_laps laps of safe set (crx_lmpcprep_desc.n_laps); when they are full crx_lmpc_addtraj_dev reports status 1 and the
 * race goes on racing on its last two laps (the reference allocates for the number of laps it is asked to run, utils/base.py:631-656).
 */
int crx_game_traffic_dev(int N, int batch, int n_cars, double lap_length, double t, double dt, const double* car_s0,
                         const double* car_v, const double* car_ey, double* veh_xcurv, double* pred_s, double* pred_ey,
                         void* stream);
int crx_game_masks_dev(int batch, const int32_t* n_veh, const int32_t* overflow, int32_t* m_overtake, int32_t* m_lmpc,
                       int32_t* overflow_seen, void* stream);
int crx_game_commit_dev(int N, int Np, int batch, const int32_t* overtake, const double* U_track, const double* X_lmpc,
                        const double* U_lmpc, const int32_t* flag, double* u, double* u_old, double* u_prev, double* lin_points,
                        double* lin_input, int32_t* step_no, int32_t* addpoint_step, int32_t* old_flag, void* stream);
int crx_game_log_dev(int batch, int n_points, double lap_length, const double* xcurv, const double* u, const int32_t* laps,
                     int32_t* laps_prev, double* log_x, double* log_u, int32_t* n_log, int32_t* crossed, void* stream);

/*
 * Multi-GPU: problems are independent and a planner sweep is sharded by scenario, so the path has
 * exactly ONE exchange step -- an all-gather of the fixed-size winner records {int32 flag; int32 status; double X[N+1][6]} (8 + 48 (N + 1) bytes =
 * 632 B at N = 12) -- issued on RCCL (ncclAllGather over xGMI).  One process per GPU.  The caller owns the rendezvous:
 * rank 0 calls crx_comm_get_unique_id and carries the CRX_COMM_ID_BYTES bytes to the other ranks by whatever it has (MPI,
 * a file, torch.distributed's store); every rank then calls crx_comm_init_rank on the device it gave crx_init (collective:
 * returns when all ranks arrived).  RCCL is resolved at run time (a copy the process already holds, e.g. PyTorch's, else
 * /opt/rocm's); libcrx does not link it.
 *   crx_allgather_winners_dev  packs this rank's n_local winners (flag [n_local] int32, best_X [n_local][N+1][6] -- the
 *       outputs of crx_select / crx_planner_plan -- and status [n_local] int32, the crx_status of the WINNING region's QP, i.e.
 *       status[s * n_regions + flag[s]] of crx_planner_solve: a consumer on another rank can tell a fall-back winner,
 *       overtake_traj_planner.py:365-374, from a solved one; NULL: the field is 0) into send [n_max][rec] (rows past n_local zero:
 *       ragged shards are padded to the largest, n_max, which every rank derives from (n_total, world) alone) and gathers every
 *       rank's block into recv [world][n_max][rec], in rank order, on `stream`.  rec = 1 + 6 (N + 1) 8-byte words; word 0 of a record
 *       holds the two int32 {flag, status} (little endian: flag in the low half), words 1.. the trajectory.  (libcrx <= 0.2.1 carried
 *       the flag as a double in word 0 and no status; 0.3.0 changed the signature.)
 */
#define CRX_COMM_ID_BYTES 128
int crx_comm_get_unique_id(void* id /* [CRX_COMM_ID_BYTES] */);
int crx_comm_init_rank(const void* id, int world, int rank);
int crx_comm_world(void);   /* 0 without a communicator */
int crx_comm_rank(void);    /* -1 without a communicator */
int crx_comm_destroy(void);
int crx_allgather_winners_dev(int n_local, int n_max, int N, const int32_t* flag, const int32_t* status, const double* best_X,
                              double* send, double* recv, void* stream);

/* Streams for device-resident loops that overlap independent launches (the two branches of a racing-game step, concurrent
 * sub-batches of races: crx.montecarlo).  Streams only overlap when they sit on different HARDWARE queues; the HIP runtime hands
 * its GPU_MAX_HW_QUEUES queues (default 4; set the variable before the process touches the GPU) to streams round-robin IN
 * CREATION ORDER.  A framework's stream pool is created interleaved with streams the caller never sees (PyTorch: low- and
 * high-priority pool entries alternate), so consecutive pool streams share queues in a pattern that depends on what the process
 * did before: measured on the 4096-race racing game with two sub-batches, 3.24 ms per step or 3.5 / 3.8 / 4.6 ms depending only
 * on how many pool streams had been handed out earlier (tools/stream_offset_probe.py; which streams share a queue is not a
 * simple function of the creation order either: tools/queue_probe.py).  crx_streams_create therefore MEASURES: it creates n + 8
 * non-blocking streams on the device of crx_init, runs pairs of 150 us single-wavefront spin kernels on them (two streams on
 * different queues finish a pair in the time of one kernel, two on the same queue in the time of two) and returns n streams of
 * which the first *n_concurrent were seen to overlap pairwise (all n when the runtime has that many queues; the rest share);
 * the other candidates are destroyed.  About 30 ms, once.  Pass them as the `stream` argument of the *_dev entry points
 * (PyTorch: torch.cuda.ExternalStream).  crx_streams_destroy synchronises and destroys them. */
int crx_streams_create(int n, void** streams /* [n] out */, int* n_concurrent /* out, may be NULL */);
int crx_streams_destroy(int n, void** streams);

/* Device time of what a caller enqueues between two marks on ONE stream, measured with a pair of HIP events the timer object owns
 * (bench.py's roofline: the solver launch alone).  crx_timer_begin / _end record on `stream` (the stream the bracketed *_dev call
 * is given); crx_timer_ms blocks until the second event has happened and returns the elapsed milliseconds, < 0 on error.  No global
 * state: any number of timers may be live on any number of streams and threads. */
int crx_timer_create(void** timer /* out */);
int crx_timer_destroy(void* timer);
int crx_timer_begin(void* timer, void* stream);
int crx_timer_end(void* timer, void* stream);
double crx_timer_ms(void* timer);

#ifdef __cplusplus
}
#endif
#endif /* CRX_H */
