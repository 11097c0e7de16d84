// crx_kparams.h -- internal launch descriptors shared by crx_kernels.hip and crx_api.hip.
// Not part of the public ABI (include/crx.h is).
#ifndef CRX_KPARAMS_H
#define CRX_KPARAMS_H
#include <stdint.h>

#include "../../include/crx.h"

struct crx_kparams {
    int N, batch, mode /* 0 planner QP, 1 CBF NLP */, per_stage_target, n_obs_max, degree;
    double A[36], B[12];
    double wq[6], wr[2];
    double w_dey, w_prog, w_slack;
    double delta_max, a_max, v_min, v_max, ey_max;
    double alpha, margin, l_sum, w_sum;
    double dt_ref, fallback_gain;
    crx_ipm_opts opts;
    // planner inputs (device pointers)
    const double *x0, *bez_s, *bez_ey, *ey_lb, *ey_ub;
    // cbf inputs
    const double *xt, *obs_s, *obs_ey, *lap_off;
    const int32_t* n_obs;
    const double* obs_dims;  // optional [batch][n_obs_max][2]: (l_agent + l_obs, w_agent + w_obs) per obstacle slot; NULL: l_sum, w_sum
    // outputs
    double *X, *U, *sigma, *cost, *kkt;
    int32_t *status, *iters;
    // optional per-iteration trace of ONE problem (diagnostics): trace[it][8]
    double* trace;
    int trace_problem, trace_rows;
    int poison;   // diagnostics: fill the LDS slice with NaN before set-up (catches reads of stale LDS)
    int kkt_unscaled;   // diagnostics (crx_debug_kkt_unscaled): kkt[b] of a CONVERGED solve = an UNSCALED KKT quantity (no s_d, CBF rows in the reference's units): 1 max, 2 dual, 3 violation, 4 complementarity
    int spec_idle;      // diagnostics (crx_debug_speculation(2)): the two-wave kernel with its second wave never asked
    const int32_t* active;   // optional [batch / active_div]: 0 = leave this problem alone (status CRX_SKIPPED, outputs untouched)
    int active_div;          // problems per mask entry (planner: the regions of a scenario share one entry); 0 or 1: one each
    const int32_t* order;    // optional [batch]: workgroup i solves problem order[i] (longest-first dispatch); NULL: i
    // planner QP: reachability screen (crx_kernels.hip, first thing the kernel does).  reach_row[j] = e_ey' A^j: the free response
    // of ey_j is reach_row[j] . x0; reach_gain[j] = sum_{m<j} |e_ey' A^m B| (delta_max, a_max)': how far the inputs can move it
    int reach_screen;
    double reach_gain[CRX_MAX_N + 1];
    double reach_row[CRX_MAX_N][6];
    // CBF NLP: the slacks start at provable lower bounds of their optimal values (crx_kernels.hip, set-up).  reach_gain (ey) and
    // reach_s (s) [j] = how far the boxed inputs can move the coordinate of stage j from its free response
    int slack_start;
    double reach_s[CRX_MAX_N + 1];
};

struct crx_lmpc_kparams {
    int N, batch, n_ss_max;
    double Q[6], R[2], dR[2], x_track[6];
    double v_max, ey_max, delta_max, a_max, w_x0;
    crx_ipm_opts opts;
    const double *x0, *u_old, *A, *B, *C, *ss, *qfun;
    const int32_t* n_ss;
    double *X, *U, *lambda, *cost, *kkt;
    int32_t *status, *iters;
    double* trace;   // optional per-iteration phase cycles of ONE problem (diagnostics): trace[it][16]
    int trace_problem, trace_rows;
    int poison;      // diagnostics: fill the LDS slice with NaN before set-up
    const int32_t* active;   // optional [batch]: 0 = leave this problem alone (status CRX_SKIPPED, outputs untouched)
    const int32_t* order;    // optional [batch]: workgroup i solves problem order[i] (longest-first dispatch); NULL: i
    int reach_screen;        // terminal-set reachability screen of the first attempt (crx_lmpc.hip)
};

struct crx_select_kparams {
    int N, V, n_scen;
    double veh_length, veh_width, lap_length, w_prog, w_coll, w_switch;
    const int32_t* n_veh;
    const double *X, *obs_s, *obs_ey;
    const int32_t* old_flag;
    int32_t* flag;
    double *sel_cost, *best_X;
};

struct crx_prep_kparams {
    int N, V, n_scen, n_opt;
    double prediction_factor, lookahead, track_width, lap_length, veh_length, veh_width, safety_margin, dt_ref;
    const double *x_wrapped, *x_raw, *veh_info, *max_dv, *obs_s, *obs_ey, *opt_s, *opt_ey;
    const int32_t* n_veh;
    double *x0, *bez_s, *bez_ey, *ey_lb, *ey_ub;
};

struct crx_path_kparams {
    int N, batch;
    double alpha, w_rate;
    crx_ipm_opts opts;
    const double *opt, *bez, *lb, *ub, *e0, *eN;
    double *E, *cost, *kkt;
    int32_t *status, *iters;
};

struct crx_plant_kparams {
    crx_plant_desc d;
    int batch, u_stride, wrap;
    const double *track, *xglob, *xcurv, *u;
    double *xglob_next, *xcurv_next;
    int32_t* laps;
    const double* noise_z;   // optional [batch][3] standard-normal draws: the bounded process noise of base.py:929-939
};

struct crx_cbfprep_kparams {
    int N, V, batch;
    double lap_length, t, dt, safety_time;
    const double *xcurv, *car_s0, *car_v, *car_ey;
    double *obs_s, *obs_ey, *lap_off;
    int32_t* n_obs;
};

struct crx_trackprep_kparams {
    int N, V, batch;
    double lap_length, safety_time, dt_ref;
    const double *x, *obs_s_in, *obs_ey_in, *traj;
    const int32_t* n_veh;
    double *xt, *obs_s, *obs_ey, *lap_off;
    int32_t* n_obs;
};

struct crx_scene_kparams {
    crx_scene_desc d;
    int n_scen;
    const double *ego_xcurv, *veh_xcurv, *pred_s, *pred_ey;
    const int32_t* n_all;
    int32_t *n_veh, *overflow, *order;
    double *veh_info, *max_dv, *obs_s, *obs_ey;
};

struct crx_lmpcprep_kparams {
    crx_lmpcprep_desc d;
    int batch, from_plan;
    const double *ss_xcurv, *u_ss, *qfun;
    const int32_t *time_ss, *iter;
    const double *x, *lin_points, *lin_input, *track;
    double *A, *B, *C, *ss_sel, *q_sel;
    int32_t* status;
    const int32_t* active;   // optional [batch]: 0 = leave this race alone (status CRX_SKIPPED, outputs untouched)
};

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
hipError_t crx_launch_solve(const crx_kparams& kp, int nobs_template, hipStream_t st);
hipError_t crx_launch_solve_spec(const crx_kparams& kp, hipStream_t st);   // crx_kernels_spec.hip: the two-wave instantiations <1, 12, 6, {12, 10}, 1>
hipError_t crx_launch_select(const crx_select_kparams& sp, hipStream_t st);
hipError_t crx_launch_debug_reduce(const double* in, double* out, hipStream_t st);
hipError_t crx_launch_lmpc_addtraj(const crx_lmpcprep_desc& d, int batch, const int32_t* crossed, double* log_x, const double* log_u,
                                   int32_t* n_log, double* ss_xcurv, double* u_ss, double* qfun, int32_t* time_ss, int32_t* iter,
                                   int32_t* step, const double* x, int32_t* status, hipStream_t st);
size_t crx_solve_lds_bytes(int N, int nobs_template);
int crx_solve_resident_per_cu(int N, int nobs_template);
hipError_t crx_launch_path(const crx_path_kparams& pp, hipStream_t st);
hipError_t crx_launch_cbfprep(const crx_cbfprep_kparams& cp, hipStream_t st);

// dispatch order of a solver launch (include/crx.h, "Dispatch order"): a stable counting sort of the batch by a 257-valued key
struct crx_order_kparams {
    int batch, mode;             // mode 0: iteration counts of the previous solve, descending; 1: smallest start barrier, ascending
    const int32_t *iters, *active;
    int32_t* order;
    int V, stride, degree, per_stage_target;   // mode 1: obstacle slots, N + 1, exponent of the ellipse, layout of xt
    double margin, l_sum, w_sum, ds_per_vx;    // ds_per_vx = A[4][0]: progress per stage and unit speed
    const double *x0, *xt, *obs_s, *obs_ey, *lap_off, *obs_dims;
    const int32_t* n_obs;
};
hipError_t crx_launch_order(const crx_order_kparams& op, hipStream_t st);

// device-resident racing-game loop: the bookkeeping between the solver launches (crx.montecarlo.GameLaps / LmpcLaps)
struct crx_game_kparams {
    int N, Np, batch, n_cars, n_points;
    double lap_length, t, dt;
    // traffic
    const double *car_s0, *car_v, *car_ey;
    double *veh_xcurv, *pred_s, *pred_ey;
    // masks
    const int32_t *n_veh, *overflow;
    int32_t *m_overtake, *m_lmpc, *overflow_seen;
    // commit
    const int32_t* overtake;
    const double *U_track, *X_lmpc, *U_lmpc;
    const int32_t* flag;
    double *u, *u_old, *u_prev, *lin_points, *lin_input;
    int32_t *step_no, *addpoint_step, *old_flag;
    // log
    const double* xcurv;
    const int32_t* laps;
    int32_t *laps_prev, *n_log, *crossed;
    double *log_x, *log_u;
};
hipError_t crx_launch_game(int which, const crx_game_kparams& gp, hipStream_t st);
hipError_t crx_launch_plant(const crx_plant_kparams& pk, hipStream_t st);
hipError_t crx_launch_prep(const crx_prep_kparams& pp, hipStream_t st);
hipError_t crx_launch_scene(const crx_scene_kparams& sp, hipStream_t st);
hipError_t crx_launch_trackprep(const crx_trackprep_kparams& tp, hipStream_t st);
hipError_t crx_launch_lmpc(const crx_lmpc_kparams& kp, hipStream_t st);
hipError_t crx_launch_lmpcprep(const crx_lmpcprep_kparams& kp, hipStream_t st);
size_t crx_lmpcprep_lds_bytes(int n_points);
hipError_t crx_launch_lmpc_addpoint(const crx_lmpcprep_desc& d, int batch, double* ss_xcurv, double* u_ss, const int32_t* time_ss,
                                    const int32_t* iter, const int32_t* step, const double* x, const double* u, int u_stride,
                                    hipStream_t st);
size_t crx_lmpc_lds_bytes(int N, int n_ss_max);
int crx_lmpc_resident_per_cu(int N, int n_ss_max);
#endif
#endif
