// crx_prep.hip -- the small kernels around the solvers:
//   crx_prep_kernel     planner host prep: Bezier references and per-stage ey bounds of every region of a scenario
//   crx_plant_kernel    one control step of the plant for a batch of vehicles
//   crx_cbfprep_kernel  obstacle arrays of control.mpccbf for scripted cars
//   crx_path_kernel     the 1-D QPs of the overtake PATH planner
//
// crx_prep_kernel: Bezier references and per-stage ey bounds of every region of a scenario, written in the layout
// crx_solve_kernel<0> reads.
// One wavefront per scenario, lanes over (region, sample).  Closed-form arithmetic, HBM-bound
// (reads ~ (12 + 3V + 2V(N+1)) doubles, writes (V+1)(6 + 3N + 3) doubles per scenario).
//
// Restates (paths into /root/reference/car_racing):
//   planning/planner_helper.py:43-135   get_bezier_control_points
//   planning/planner_helper.py:138-153  get_bezier_curve, sampled at t = j/N (overtake_traj_planner.py:105-111)
//   planning/overtake_traj_planner.py:277-324  ey bounds with the obstacle windows (quirks Q2, Q5)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "crx_kparams.h"
#include "crx_wave.h"

// scipy interp1d(kind="linear") (searchsorted-left, index clipped to [1,n-1], slope form), as used at
// planner_helper.py:121-134
__device__ __forceinline__ double prep_interp(const double* xs, const double* ys, int n, double x) {
    int lo = 0, hi = n;   // first index with xs[i] >= x
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (xs[mid] < x) lo = mid + 1; else hi = mid;
    }
    int i1 = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const int i0 = i1 - 1;
    const double slope = (ys[i1] - ys[i0]) / (xs[i1] - xs[i0]);
    return slope * (x - xs[i0]) + ys[i0];
}

__global__ void __launch_bounds__(WAVE) crx_prep_kernel(const crx_prep_kparams pp) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= pp.n_scen) return;
    const int N = pp.N, V = pp.V, R = V + 1;
    const int nv = min(max(pp.n_veh[s], 0), V);   // device-resident counts cannot be validated on the host: clamp
    const double* xw = pp.x_wrapped + (size_t)6 * s;
    const double* vi = pp.veh_info + (size_t)3 * V * s;
    const double L = pp.lap_length, vw = pp.veh_width, tw = pp.track_width;
    // control points in s (:49-85)
    const double s0 = xw[4];
    double s3 = s0 + pp.prediction_factor * pp.max_dv[s] + pp.lookahead, span;
    if (s0 > s3) { span = s3 + L - s0; s3 = s3 + L; } else { span = s3 - s0; }
    const double s1 = span / 3.0 + s0, s2 = 2.0 * span / 3.0 + s0;
    // end point rides the optimal trajectory (:121-134)
    const double s_end = s3 >= L ? s3 - L : s3;
    const double e3 = s_end <= pp.opt_s[0] ? pp.opt_ey[0] : prep_interp(pp.opt_s, pp.opt_ey, pp.n_opt, s_end);
    const double e0 = xw[5];   // :95 overrides :87-94
    for (int e = lane; e < R * (N + 1); e += WAVE) {
        const int r = e / (N + 1), j = e - r * (N + 1);
        const int rr = r > nv ? nv : r;   // regions beyond this scenario's vehicles repeat the last one (never selected)
        double e12;
        // no vehicle at all (the reference never plans then: get_overtake_flag is false; its code would index an empty
        // list): the single region gets the curve from the ego to the optimal line, and veh_info is not read
        if (nv == 0) e12 = e3;
        else if (rr == 0) e12 = 0.8 * tw - (-vi[3 * rr + 1] - 0.5 * vw) * 0.2;                       // :98-104
        else if (rr == nv) e12 = -0.8 * tw + (vi[3 * (rr - 1) + 1] - 0.5 * vw) * 0.2;           // :106-112
        else e12 = 0.7 * (vi[3 * rr + 1] + 0.5 * vw) + 0.3 * (vi[3 * (rr - 1) + 1] - 0.5 * vw);   // :113-119
        const double t = j * (1.0 / N), u = 1.0 - t;
        const double b0 = pow(u, 3.0), b1 = 3.0 * t * (u * u), b2 = 3.0 * (t * t) * u, b3 = pow(t, 3.0);   // :139-146
        const size_t o = ((size_t)s * R + r) * (N + 1) + j;
        pp.bez_s[o] = s0 * b0 + s1 * b1 + s2 * b2 + s3 * b3;
        pp.bez_ey[o] = e0 * b0 + e12 * b1 + e12 * b2 + e3 * b3;
    }
    // ey bounds (:277-324): both neighbours impose ey >= ey_obs + W + margin inside their s-window (quirk Q2)
    const double ub = tw - 0.5 * vw, win = pp.veh_length + pp.safety_margin;
    for (int e = lane; e < R * N; e += WAVE) {
        const int r = e / N, k = e - r * N;
        const double s_nom = xw[4] + (k * pp.dt_ref) * xw[0];                                   // :296 (wrapped copy, Q5)
        double lb = -ub;
        for (int side = 0; side < 2; side++) {
            const int v = side == 0 ? r - 1 : r;                                                // sorted[r-1], sorted[r]
            if (v < 0 || v >= nv) continue;
            double os = pp.obs_s[((size_t)s * V + v) * (N + 1) + k];
            os = wrap_above(os, L);                                                             // :291-292
            if (s_nom >= os - win && s_nom <= os + win)
                lb = fmax(lb, pp.obs_ey[((size_t)s * V + v) * (N + 1) + k] + vw + pp.safety_margin);
        }
        pp.ey_lb[((size_t)s * R + r) * N + k] = lb;
    }
    for (int e = lane; e < R * 6; e += WAVE) {
        const int r = e / 6, c = e - 6 * r;
        pp.x0[((size_t)s * R + r) * 6 + c] = pp.x_raw[(size_t)6 * s + c];                      // :266 (raw state, Q5)
    }
    if (lane < R) pp.ey_ub[(size_t)s * R + lane] = ub;
}

// crx_scene_kernel: the front of OvertakeTrajPlanner (get_overtake_flag, the partial ey sort, veh_infos, agent_info.max_delta_v,
// predictions in sorted order), one wavefront per scenario: the decisions are a few dozen scalar operations every lane runs
// alike, the lanes share the copying of the predictions.  Restates planner_helper.py:177-201, :218-266 and
// overtake_traj_planner.py:29-42, :66-92 (quirks Q3, Q4).
__global__ void __launch_bounds__(WAVE) crx_scene_kernel(const crx_scene_kparams sp) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= sp.n_scen) return;
    const crx_scene_desc& d = sp.d;
    const int N1 = d.N + 1, VA = d.n_all_max, V = d.n_veh_max;
    const double* ego = sp.ego_xcurv + (size_t)6 * s;
    const double* vx_ = sp.veh_xcurv + (size_t)6 * VA * s;
    const int na = min(max(sp.n_all[s], 0), VA);
    const double L = d.lap_length, s_e = wrap_above(ego[4], L);
    // vehicles of interest, in dict order (get_overtake_flag :29-42 -> check_ego_agent_distance planner_helper.py:218-266)
    int idx[CRX_MAX_VEH], nv = 0, over = 0;
    unsigned long long hits = 0ull;
    int nh = 0;
    for (int v = 0; v < na; v++) {
        const double dv = fabs(ego[0] - vx_[6 * v]);
        const double s_a = wrap_above(vx_[6 * v + 4], L);
        const double ahead = d.safety_factor * d.veh_length + d.prediction_factor * dv, behind = 1.0 * d.veh_length;
        const bool hit = (s_a - s_e <= ahead && s_a >= s_e) || (s_a + L - s_e <= ahead && s_a + L >= s_e) ||
                         (s_e - s_a <= behind && s_a <= s_e) || (s_e + L - s_a <= behind && s_a <= s_e + L);
        if (hit) { hits |= 1ull << v; nh++; }
    }
    // More vehicles of interest than the n_veh_max slots (the reference has no limit): keep the n_veh_max NEAREST along the
    // track (distance on the closed lap, ties to the earlier one), in dict order -- the policy of hostprep.pack_obstacles on the
    // controller side; `overflow` counts the dropped ones.
    if (nh > V) {
        unsigned long long keep = 0ull;
        for (int k = 0; k < V; k++) {
            int best = -1;
            double bd = 0.0;
            for (int v = 0; v < na; v++) {
                if (!((hits >> v) & 1ull) || ((keep >> v) & 1ull)) continue;
                const double g = wrap_above(vx_[6 * v + 4], L) - s_e;
                const double dist = fmin(fabs(g), fmin(fabs(g + L), fabs(g - L)));
                if (best < 0 || dist < bd) { best = v; bd = dist; }
            }
            keep |= 1ull << best;
        }
        over = nh - V;
        hits = keep;
    }
    for (int v = 0; v < na; v++)
        if ((hits >> v) & 1ull) idx[nv++] = v;
    // partial "sort" (:66-76, quirk Q3): a new vehicle goes to the FRONT if its ey >= the current first one's, else to the back
    int ord[CRX_MAX_VEH];
    for (int k = 0; k < nv; k++) {
        const double e = vx_[6 * idx[k] + 5];
        if (k == 0) ord[0] = idx[0];
        else if (e >= vx_[6 * ord[0] + 5]) { for (int q = k; q > 0; q--) ord[q] = ord[q - 1]; ord[0] = idx[k]; }
        else ord[k] = idx[k];          // (e <= first; a NaN would be dropped by the reference, not representable here)
    }
    // agent_info.max_delta_v over the sorted vehicles (planner_helper.py:177-201)
    double mdv = 0.0;
    for (int k = 0; k < nv; k++) mdv = fmax(mdv, fabs(ego[0] - vx_[6 * ord[k]]));
    if (lane == 0) { sp.n_veh[s] = nv; sp.overflow[s] = over; sp.max_dv[s] = mdv; }
    if (lane < V) sp.order[(size_t)V * s + lane] = lane < nv ? ord[lane] : -1;
    // veh_info rows in ITERATION order (:87-92, quirk Q4): (s, max ey over the prediction, min ey)
    if (lane < V) {
        double vi0 = 0.0, mx = 0.0, mn = 0.0;
        if (lane < nv) {
            const double* pe = sp.pred_ey + ((size_t)s * VA + idx[lane]) * N1;
            vi0 = vx_[6 * idx[lane] + 4]; mx = pe[0]; mn = pe[0];
            for (int j = 1; j < N1; j++) { mx = fmax(mx, pe[j]); mn = fmin(mn, pe[j]); }
        }
        double* vi = sp.veh_info + ((size_t)s * V + lane) * 3;
        vi[0] = vi0; vi[1] = mx; vi[2] = mn;
    }
    // predictions of the sorted vehicles
    for (int e = lane; e < V * N1; e += WAVE) {
        const int k = e / N1, j = e - k * N1;
        const size_t src = ((size_t)s * VA + (k < nv ? ord[k] : 0)) * N1 + j;
        sp.obs_s[((size_t)s * V + k) * N1 + j] = k < nv ? sp.pred_s[src] : 0.0;
        sp.obs_ey[((size_t)s * V + k) * N1 + j] = k < nv ? sp.pred_ey[src] : 0.0;
    }
}

// crx_trackprep_kernel: the inputs of control.mpc_multi_agents' NLP from the planner's outputs, one thread per race:
// per-stage targets (control/control.py:373-382: the selected trajectory's ey, linearly interpolated at the clipped nominal s)
// and the window filter / lap offsets / packing of the sorted vehicles' predictions (:293-309).  Arithmetic of
// crx/hostprep.py tracking_targets, cbf_window, pack_obstacles.
__global__ void __launch_bounds__(256) crx_trackprep_kernel(const crx_trackprep_kparams tp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= tp.batch) return;
    const int N = tp.N, V = tp.V, N1 = N + 1;
    const double* x = tp.x + (size_t)6 * b;
    const double* tr = tp.traj + (size_t)b * N1 * 6;
    const double L = tp.lap_length, vx = x[0], se = x[4];
    // targets: s clipped to the trajectory's range, scipy interp1d(kind="linear") (searchsorted-left, index in [1, n-1], slope form)
    const double s_lo = tr[4], s_hi = tr[(size_t)N * 6 + 4];
    for (int i = 0; i <= N; i++) {
        double s = se + vx * tp.dt_ref * i;
        s = s < s_lo ? s_lo : s;
        s = s >= s_hi ? s_hi : s;
        int hi = 0;
        while (hi < N1 && tr[(size_t)hi * 6 + 4] < s) hi++;
        hi = hi < 1 ? 1 : (hi > N1 - 1 ? N1 - 1 : hi);
        const int lo = hi - 1;
        const double slope = (tr[(size_t)hi * 6 + 5] - tr[(size_t)lo * 6 + 5]) / (tr[(size_t)hi * 6 + 4] - tr[(size_t)lo * 6 + 4]);
        double* xt = tp.xt + ((size_t)b * N1 + i) * 6;
        xt[0] = vx; xt[1] = 0.0; xt[2] = 0.0; xt[3] = 0.0; xt[4] = 0.0;
        xt[5] = slope * (s - tr[(size_t)lo * 6 + 4]) + tr[(size_t)lo * 6 + 5];
    }
    // obstacles: +-safety_time*vx window on the lap-folded positions (int() truncation as in the reference), lap offsets, packing
    const double margin = tp.safety_time * vx;
    const double nce = trunc(se / L), dist_ego = se - nce * L;
    const int nv = min(max(tp.n_veh[b], 0), V);
    int n = 0;
    for (int v = 0; v < nv; v++) {
        const double* is = tp.obs_s_in + ((size_t)b * V + v) * N1;
        const double* ie = tp.obs_ey_in + ((size_t)b * V + v) * N1;
        const double nco = trunc(is[0] / L), dist_obs = is[0] - nco * L;
        if (dist_ego > dist_obs - margin && dist_ego < dist_obs + margin) {
            double* os = tp.obs_s + ((size_t)b * V + n) * N1;
            double* oe = tp.obs_ey + ((size_t)b * V + n) * N1;
            for (int j = 0; j < N1; j++) { os[j] = is[j]; oe[j] = ie[j]; }
            tp.lap_off[(size_t)b * V + n] = (nce - nco) * L;
            n++;
        }
    }
    for (int v = n; v < V; v++) {
        double* os = tp.obs_s + ((size_t)b * V + v) * N1;
        double* oe = tp.obs_ey + ((size_t)b * V + v) * N1;
        for (int j = 0; j < N1; j++) { os[j] = 0.0; oe[j] = 0.0; }
        tp.lap_off[(size_t)b * V + v] = 0.0;
    }
    tp.n_obs[b] = n;
}

hipError_t crx_launch_trackprep(const crx_trackprep_kparams& tp, hipStream_t st) {
    if (tp.batch == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_trackprep_kernel, dim3((tp.batch + 255) / 256), dim3(256), 0, st, tp);
    return hipGetLastError();
}

hipError_t crx_launch_scene(const crx_scene_kparams& sp, hipStream_t st) {
    if (sp.n_scen == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_scene_kernel, dim3(sp.n_scen), dim3(WAVE), 0, st, sp);
    return hipGetLastError();
}

hipError_t crx_launch_prep(const crx_prep_kparams& pp, hipStream_t st) {
    if (pp.n_scen == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_prep_kernel, dim3(pp.n_scen), dim3(WAVE), 0, st, pp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Plant: one thread per vehicle, n_sub Euler sub-steps.  Restates
//   system/vehicle_dynamics.py:4-49          one Euler step, global + curvilinear
//   utils/racing_env.py:225-246              curvature lookup (first matching segment, inclusive ends)
//   utils/base.py:897-942                    DynamicBicycleModel.forward_dynamics, zero noise
// Compute-bound on transcendentals (2 atan2, 2 atan, 6 sin/cos per sub-step); HBM traffic 14+12 doubles
// per vehicle per control step.
// ------------------------------------------------------------------------------------------------
#define CRX_PLANT_MAX_SEG 64

__global__ void __launch_bounds__(256) crx_plant_kernel(const crx_plant_kparams pk) {
    __shared__ double seg_lo[CRX_PLANT_MAX_SEG], seg_hi[CRX_PLANT_MAX_SEG], seg_cv[CRX_PLANT_MAX_SEG];
    const crx_plant_desc& d = pk.d;
    for (int i = threadIdx.x; i < d.n_seg; i += blockDim.x) {
        seg_lo[i] = pk.track[6 * i + 3];
        seg_hi[i] = pk.track[6 * i + 3] + pk.track[6 * i + 4];
        seg_cv[i] = pk.track[6 * i + 5];
    }
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= pk.batch) return;
    double vx = pk.xcurv[6 * b], vy = pk.xcurv[6 * b + 1], wz = pk.xcurv[6 * b + 2];
    double epsi = pk.xcurv[6 * b + 3], s = pk.xcurv[6 * b + 4], ey = pk.xcurv[6 * b + 5];
    double psi = pk.xglob[6 * b + 3], X = pk.xglob[6 * b + 4], Y = pk.xglob[6 * b + 5];
    const double delta = pk.u[(size_t)pk.u_stride * b], acc = pk.u[(size_t)pk.u_stride * b + 1], dt = d.dt_sub;
    const double sd = sin(delta), cd = cos(delta);
    for (int it = 0; it < d.n_sub; it++) {
        // curvature at s (wrapped into one lap), first segment with lo <= s <= hi
        double sw = s;
        sw = wrap_below(wrap_above(sw, d.lap_length), d.lap_length);
        double curv = 0.0;
        for (int i = 0; i < d.n_seg; i++)
            if (sw >= seg_lo[i] && sw <= seg_hi[i]) { curv = seg_cv[i]; break; }
        // tyre slip angles and lateral forces (the reference uses lf for the rear axle too, :25)
        const double slip_f = delta - atan2(vy + d.lf * wz, vx);
        const double slip_r = -atan2(vy - d.lf * wz, vx);
        const double Fyf = 2 * d.Df * sin(d.Cf * atan(d.Bf * slip_f));
        const double Fyr = 2 * d.Dr * sin(d.Cr * atan(d.Br * slip_r));
        const double dvx = acc - 1 / d.m * Fyf * sd + wz * vy;
        const double dvy = 1 / d.m * (Fyf * cd + Fyr) - wz * vx;
        const double dwz = 1 / d.Iz * (d.lf * Fyf * cd - d.lr * Fyr);
        const double ce = cos(epsi), se = sin(epsi), cp = cos(psi), sp = sin(psi);
        const double v_long = (vx * ce - vy * se) / (1 - curv * ey);
        const double psi_n = psi + dt * wz;
        const double X_n = X + dt * (vx * cp - vy * sp), Y_n = Y + dt * (vx * sp + vy * cp);
        const double epsi_n = epsi + dt * (wz - v_long * curv);
        const double s_n = s + dt * v_long;
        const double ey_n = ey + dt * (vx * se + vy * ce);
        const double vx_n = vx + dt * dvx, vy_n = vy + dt * dvy, wz_n = wz + dt * dwz;
        vx = vx_n; vy = vy_n; wz = wz_n; epsi = epsi_n; s = s_n; ey = ey_n; psi = psi_n; X = X_n; Y = Y_n;
    }
    if (pk.wrap && s > d.lap_length) {   // ModelBase.update_memory (base.py:795-819)
        s -= d.lap_length;
        if (pk.laps) pk.laps[b] += 1;
    }
    double* g = pk.xglob_next + 6 * (size_t)b;
    double* c = pk.xcurv_next + 6 * (size_t)b;
    g[0] = vx; g[1] = vy; g[2] = wz; g[3] = psi; g[4] = X; g[5] = Y;
    if (pk.noise_z) {
        // bounded process noise (utils/base.py:929-939): clip(z sigma, +-lim), HALF of it added to the curvilinear velocities
        // only -- the global-frame copy stays clean (the next step reads its velocities from xcurv, vehicle_dynamics.py:17-19)
        const double* z = pk.noise_z + 3 * (size_t)b;
        vx += 0.5 * fmax(-0.05, fmin(z[0] * 0.01, 0.05));
        vy += 0.5 * fmax(-0.1, fmin(z[1] * 0.01, 0.1));
        wz += 0.5 * fmax(-0.05, fmin(z[2] * 0.005, 0.05));
    }
    c[0] = vx; c[1] = vy; c[2] = wz; c[3] = epsi; c[4] = s; c[5] = ey;
}

hipError_t crx_launch_plant(const crx_plant_kparams& pk, hipStream_t st) {
    if (pk.batch == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_plant_kernel, dim3((pk.batch + 255) / 256), dim3(256), 0, st, pk);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Obstacle arrays of control.mpccbf for scripted cars (see crx_cbf_prep_dev in include/crx.h): one thread
// per race.  int() of the reference truncates toward zero (control.py:500,519).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) crx_cbfprep_kernel(const crx_cbfprep_kparams cp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= cp.batch) return;
    const int N = cp.N, V = cp.V;
    const double L = cp.lap_length, vx = cp.xcurv[6 * (size_t)b], se = cp.xcurv[6 * (size_t)b + 4];
    const double margin = cp.safety_time * vx;
    const double nce = trunc(se / L), dist_ego = se - nce * L;
    int n = 0;
    for (int v = 0; v < V; v++) {
        const double s0 = cp.car_s0[(size_t)b * V + v], sv = cp.car_v[(size_t)b * V + v], ey = cp.car_ey[(size_t)b * V + v];
        const double s_now = sv * (cp.t + 0 * cp.dt) + s0;
        const double nco = trunc(s_now / L), dist_obs = s_now - nco * L;
        if (dist_ego > dist_obs - margin && dist_ego < dist_obs + margin) {
            double* os = cp.obs_s + ((size_t)b * V + n) * (N + 1);
            double* oe = cp.obs_ey + ((size_t)b * V + n) * (N + 1);
            for (int j = 0; j <= N; j++) {
                const double tj = cp.t + j * cp.dt;
                os[j] = sv * tj + s0;
                oe[j] = ey + 0.0 * tj;
            }
            cp.lap_off[(size_t)b * V + n] = (nce - nco) * L;
            n++;
        }
    }
    for (int v = n; v < V; v++) {
        double* os = cp.obs_s + ((size_t)b * V + v) * (N + 1);
        double* oe = cp.obs_ey + ((size_t)b * V + v) * (N + 1);
        for (int j = 0; j <= N; j++) { os[j] = 0.0; oe[j] = 0.0; }
        cp.lap_off[(size_t)b * V + v] = 0.0;
    }
    cp.n_obs[b] = n;
}

hipError_t crx_launch_cbfprep(const crx_cbfprep_kparams& cp, hipStream_t st) {
    if (cp.batch == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_cbfprep_kernel, dim3((cp.batch + 255) / 256), dim3(256), 0, st, cp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Overtake PATH planner QP (planning/overtake_path_planner.py:199-318): one
// 1-D QP per wavefront, lane i = interior offset ey_{i+1}.  Same interior-point iteration as the other
// kernels; the Newton matrix (tridiagonal cost + barrier diagonal) is factorised dense with the shared
// row-per-lane Cholesky (n <= 23).  HBM: 4(N+1)+2 doubles in, N+1+3 out per problem.
// ------------------------------------------------------------------------------------------------
#define PATH_NV (CRX_MAX_N)
#define PATH_LD (CRX_MAX_N + 1)

__global__ void __launch_bounds__(WAVE) crx_path_kernel(const crx_path_kparams pp) {
    __shared__ double sh[(PATH_NV + 4) * PATH_LD + 3 * PATH_NV + 32];   // +3 rows: the blocked Cholesky reads (not uses) past the last row
    double* Km = sh;
    constexpr int IK = (PATH_NV + 4) * PATH_LD;
    double *v = sh + IK + PATH_NV, *dv = v + PATH_NV, *Fth = dv + PATH_NV, *Fph = Fth + 16;
    const int b = blockIdx.x, lane = threadIdx.x, N = pp.N, n = N - 1;
    if (b >= pp.batch) return;
    const crx_ipm_opts& o = pp.opts;
    const double *op = pp.opt + (size_t)(N + 1) * b, *bz = pp.bez + (size_t)(N + 1) * b;
    const double *lo = pp.lb + (size_t)(N + 1) * b, *hi = pp.ub + (size_t)(N + 1) * b;
    const double a0 = pp.e0[b], aN = pp.eN[b], w1 = 1.0 - pp.alpha, w2 = pp.alpha, wr = pp.w_rate;
    double* Eb = pp.E + (size_t)(N + 1) * b;
    const bool mine = lane < n;
    const int j = mine ? lane + 1 : 1;
    const double lj = lo[j], hj = hi[j], oj = op[j], bj = bz[j];
    // infeasible by inspection: an end point outside its own box, or an empty box
    int bad = (mine && lj > hj) ? 1 : 0;
    if (lane == 0) bad |= !(a0 >= lo[0] - o.tol && a0 <= hi[0] + o.tol && aN >= lo[N] - o.tol && aN <= hi[N] + o.tol);
    bad = wave_max((double)bad) > 0.0;
    if (bad) {
        if (mine) Eb[j] = fmin(fmax(bj, fmin(lj, hj)), fmax(lj, hj));
        if (lane == 0) { Eb[0] = a0; Eb[N] = aN; pp.cost[b] = INFINITY; pp.status[b] = CRX_INFEASIBLE; pp.kkt[b] = INFINITY; pp.iters[b] = 0; }
        return;
    }
    const bool hasL = mine && lj > -INFINITY, hasU = mine && hj < INFINITY;
    const double Hd = 2.0 * (w1 + w2) + 4.0 * wr;
    double g0 = mine ? -2.0 * (w1 * oj + w2 * bj) : 0.0;
    if (mine && lane == 0) g0 -= 2.0 * wr * a0;
    if (mine && lane == n - 1) g0 -= 2.0 * wr * aN;
    double f0 = mine ? w1 * oj * oj + w2 * bj * bj : 0.0;
    f0 = wave_sum(f0) + w1 * (a0 - op[0]) * (a0 - op[0]) + w2 * (a0 - bz[0]) * (a0 - bz[0]) + w1 * (aN - op[N]) * (aN - op[N]) +
         w2 * (aN - bz[N]) * (aN - bz[N]) + wr * a0 * a0 + wr * aN * aN;
    const int m = (int)wave_sum((hasL ? 1.0 : 0.0) + (hasU ? 1.0 : 0.0));
    // per-lane row state: lower row c = v - lo, upper row c = hi - v
    double vi = 0.0, tL = hasL ? fmax(fabs(-lj), o.slack_push) : 1.0, tU = hasU ? fmax(fabs(hj), o.slack_push) : 1.0, nL = hasL ? 1.0 : 0.0,
           nU = hasU ? 1.0 : 0.0;
    if (mine) v[lane] = 0.0;
    SYNC();
    double mu = o.mu_init, theta_min = 0.0, theta_max = INFINITY, E0 = INFINITY, f = f0;
    int nf = 0, st = CRX_MAX_ITER, it = 0;
    for (it = 0;; it++) {
        const double vm = (mine && lane > 0) ? v[lane - 1] : 0.0, vp = (mine && lane + 1 < n) ? v[lane + 1] : 0.0;
        const double g = mine ? Hd * vi - 2.0 * wr * vm - 2.0 * wr * vp + g0 : 0.0;
        const double cL = vi - lj, cU = hj - vi;
        const double rpL = hasL ? cL - tL : 0.0, rpU = hasU ? cU - tU : 0.0;
        const double nus = wave_sum(nL + nU);
        const double sd = m ? fmax(100.0, nus / m) / 100.0 : 1.0;
        const double e_d = wave_max(mine ? fabs(g - nL + nU) : 0.0) / sd;
        const double e_p = wave_max(fmax(fabs(rpL), fabs(rpU)));
        const double e_c = wave_max(fmax(hasL ? tL * nL : 0.0, hasU ? tU * nU : 0.0)) / sd;
        const double theta = wave_sum(fabs(rpL) + fabs(rpU));
        E0 = fmax(e_d, fmax(e_p, e_c));
        if (E0 <= o.tol && e_d * sd <= o.dual_inf_tol && e_p <= o.constr_viol_tol && e_c * sd <= o.compl_inf_tol) { st = CRX_CONVERGED; break; }   // [r6] IPOPT's complete test
        if (it >= o.max_iter) break;
        for (;;) {
            const double e_cm = wave_max(fmax(hasL ? fabs(tL * nL - mu) : 0.0, hasU ? fabs(tU * nU - mu) : 0.0)) / sd;
            if (fmax(e_d, fmax(e_p, e_cm)) <= o.kappa_eps * mu && mu > o.tol / 10.0) {
                mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
                nf = 0;
            } else
                break;
        }
        const double tau = fmax(o.tau_min, 1.0 - mu);
        // Newton matrix (lower triangle, row per lane) with the right-hand side as the extra row
        const double sgL = hasL ? nL / tL : 0.0, sgU = hasU ? nU / tU : 0.0;
        const double rhs = mine ? -g + (hasL ? mu / tL - sgL * rpL : 0.0) - (hasU ? mu / tU - sgU * rpU : 0.0) : 0.0;
        if (mine) {
            for (int q = 0; q < lane; q++) Km[lane * PATH_LD + q] = (q == lane - 1) ? -2.0 * wr : 0.0;
            Km[lane * PATH_LD + lane] = Hd + sgL + sgU;
            Km[n * PATH_LD + lane] = rhs;
        }
        SYNC();
        if (!l_chol(sh, 0, PATH_LD, IK, n, 1, lane)) break;
        double bb[1];
        bb[0] = mine ? Km[n * PATH_LD + lane] : 0.0;
        l_backsub<1>(sh, 0, PATH_LD, IK, n, lane, bb);
        const double d = mine ? bb[0] : 0.0;
        if (mine) dv[lane] = d;
        SYNC();
        const double dtL = hasL ? rpL + d : 0.0, dtU = hasU ? rpU - d : 0.0;
        const double dnL = hasL ? (mu - tL * nL - nL * dtL) / tL : 0.0, dnU = hasU ? (mu - tU * nU - nU * dtU) / tU : 0.0;
        double a_p = 1.0, a_d = 1.0;
        if (dtL < 0.0) a_p = fmin(a_p, -tau * tL / dtL);
        if (dtU < 0.0) a_p = fmin(a_p, -tau * tU / dtU);
        if (dnL < 0.0) a_d = fmin(a_d, -tau * nL / dnL);
        if (dnU < 0.0) a_d = fmin(a_d, -tau * nU / dnU);
        a_p = wave_min(a_p);
        a_d = wave_min(a_d);
        const double dm = (mine && lane > 0) ? dv[lane - 1] : 0.0, dp = (mine && lane + 1 < n) ? dv[lane + 1] : 0.0;
        const double gdv = wave_sum(g * d), qd = wave_sum(mine ? d * (Hd * d - 2.0 * wr * dm - 2.0 * wr * dp) : 0.0);
        const double Dphi = gdv - mu * wave_sum((hasL ? dtL / tL : 0.0) + (hasU ? dtU / tU : 0.0));
        LogAcc l0;
        if (hasL) l0.mul(tL);
        if (hasU) l0.mul(tU);
        const double phi0 = f - mu * l0.wave_total();
        if (it == 0) { theta_min = 1e-4 * fmax(1.0, theta); theta_max = 1e4 * fmax(1.0, theta); }
        double al = a_p, fn = f, tLn = tL, tUn = tU;
        int acc = 0, ftype = 0;
        for (int ls = 0; ls < 40 && al >= 1e-10; ls++) {   // alpha_min: see crx_kernels.hip
            fn = f + al * (gdv + 0.5 * al * qd);
            const double vt = vi + al * d;
            tLn = hasL ? fmax(tL + al * dtL, vt - lj) : 1.0;
            tUn = hasU ? fmax(tU + al * dtU, hj - vt) : 1.0;
            const double thn = wave_sum((hasL ? fabs(vt - lj - tLn) : 0.0) + (hasU ? fabs(hj - vt - tUn) : 0.0));
            LogAcc la;
            if (hasL) la.mul(tLn);
            if (hasU) la.mul(tUn);
            const double phin = fn - mu * la.wave_total();
            int okf = (thn <= theta_max) && (phin == phin);
            for (int i = 0; i < nf && okf; i++)
                if (!(thn < Fth[i] || phin < Fph[i])) okf = 0;
            if (okf) {
                const int sw = (Dphi < 0.0) && (al * pow(-Dphi, 2.3) > pow(theta, 1.1));
                if (theta <= theta_min && sw) {
                    if (phin <= phi0 + 1e-8 * al * Dphi + 10.0 * 2.2e-16 * fabs(phi0)) { acc = 1; ftype = 1; }
                } else if (thn <= (1.0 - 1e-5) * theta || phin <= phi0 - 1e-8 * theta) {
                    acc = 1;
                }
            }
            if (acc) break;
            al *= 0.5;
        }
        if (acc && !ftype && nf < 16) {
            if (lane == 0) { Fth[nf] = (1.0 - 1e-5) * theta; Fph[nf] = phi0 - 1e-8 * theta; }
            nf++;
        }
        if (!acc) break;
        vi += al * d;
        f = fn;
        tL = tLn; tU = tUn;
        if (hasL) nL = fmin(fmax(nL + a_d * dnL, mu / (1e10 * tL)), 1e10 * mu / tL);
        if (hasU) nU = fmin(fmax(nU + a_d * dnU, mu / (1e10 * tU)), 1e10 * mu / tU);
        if (mine) v[lane] = vi;
        SYNC();
    }
    if (mine) Eb[j] = vi;
    if (lane == 0) {
        Eb[0] = a0; Eb[N] = aN;
        pp.cost[b] = st == CRX_CONVERGED ? f : INFINITY;
        pp.status[b] = st; pp.kkt[b] = E0; pp.iters[b] = it;
    }
}

hipError_t crx_launch_path(const crx_path_kparams& pp, hipStream_t st) {
    if (pp.batch == 0) return hipSuccess;
    hipLaunchKernelGGL(crx_path_kernel, dim3(pp.batch), dim3(WAVE), 0, st, pp);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Device-resident racing-game loop (crx.montecarlo.GameLaps): the bookkeeping of
// LMPCRacingGame.calc_input (utils/base.py:456-583) and of the simulator (racing/offboard.py:114-131, base.py:780-819)
// between the solver launches, one thread per race (or per race and car).  Round 2 did this with ~40 element-wise torch
// launches per control step; these four kernels are the same assignments in the same order.
// ------------------------------------------------------------------------------------------------
// scripted cars at their own clock t (NoDynamicsModel, base.py:847-890): s = v t + s0 wrapped once past the line like
// update_memory keeps it (base.py:795-819), predictions v (t + j dt) + s0 unwrapped (get_trajectory_nsteps: quirk Q6)
__global__ void __launch_bounds__(256) crx_game_traffic_kernel(const crx_game_kparams gp) {
#pragma clang fp contract(off)   // multiply, then add, as the element-wise formulation this replaces did (and NumPy does)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= gp.batch * gp.n_cars) return;
    const double v = gp.car_v[i], s0 = gp.car_s0[i], ey = gp.car_ey[i], L = gp.lap_length;
    const double s_now = v * gp.t + s0;
    double* x = gp.veh_xcurv + (size_t)6 * i;
    x[0] = v; x[1] = 0.0; x[2] = 0.0; x[3] = 0.0;
    x[4] = s_now - L * fmax(ceil(s_now / L) - 1.0, 0.0);
    x[5] = ey;
    const int N1 = gp.Np + 1;
    for (int j = 0; j < N1; j++) {
        gp.pred_s[(size_t)i * N1 + j] = v * (gp.t + (double)j * gp.dt) + s0;
        gp.pred_ey[(size_t)i * N1 + j] = ey + 0.0 * ((double)j * gp.dt);
    }
}

// which branch a race takes this step (utils/base.py:463-467: vehicles of interest -> overtake planner + tracking NLP, none ->
// learning MPC), as the `active` masks of the masked launches
__global__ void __launch_bounds__(256) crx_game_masks_kernel(const crx_game_kparams gp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= gp.batch) return;
    const int ot = gp.n_veh[b] > 0 ? 1 : 0;
    gp.m_overtake[b] = ot;
    gp.m_lmpc[b] = 1 - ot;
    if (gp.overflow_seen) gp.overflow_seen[b] += gp.overflow[b];
}

// after the solves: the input the race applies, the learning MPC's hand-over of its plan (u_old, linearisation points shifted
// by one stage with the last one repeated: control.py:726-728 / utils/base.py:504-515), the step counter of add_point and the
// direction flag (utils/base.py:546-551) -- each from the branch the race is in; the other branch's state is left alone
__global__ void __launch_bounds__(256) crx_game_commit_kernel(const crx_game_kparams gp) {
    // one thread per (race, element of the plan hand-over): E = 6 (N + 1) + 2 N elements; thread 0 of a race also does its scalars
    const int N = gp.N, EX = 6 * (N + 1), E = EX + 2 * N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)gp.batch * E) return;
    const int b = (int)(i / E), e = (int)(i - (size_t)b * E);
    const bool ot = gp.overtake && gp.overtake[b] != 0;
    const double* Ul = gp.U_lmpc + (size_t)b * N * 2;
    const double* Xl = gp.X_lmpc + (size_t)b * (N + 1) * 6;
    if (e == 0) {
        const double u0 = ot ? gp.U_track[(size_t)b * gp.Np * 2] : Ul[0], u1 = ot ? gp.U_track[(size_t)b * gp.Np * 2 + 1] : Ul[1];
        gp.u[2 * b] = u0; gp.u[2 * b + 1] = u1;
        gp.u_prev[2 * b] = gp.u_old[2 * b]; gp.u_prev[2 * b + 1] = gp.u_old[2 * b + 1];
        gp.addpoint_step[b] = ot ? -(1 << 20) : gp.step_no[b];
        if (gp.old_flag) gp.old_flag[b] = ot ? gp.flag[b] : -1;
        if (!ot) { gp.u_old[2 * b] = Ul[0]; gp.u_old[2 * b + 1] = Ul[1]; gp.step_no[b] += 1; }
    }
    if (ot) return;
    if (e < EX) {                       // lin_points[k] <- X[k + 1], the last stage repeated
        const int k = e / 6, c = e - 6 * k, ks = k < N ? k + 1 : N;
        gp.lin_points[(size_t)b * EX + e] = Xl[6 * ks + c];
    } else {                            // lin_input[k] <- U[k + 1], the last stage repeated
        const int q = e - EX, k = q >> 1, c = q & 1, ks = k < N - 1 ? k + 1 : N - 1;
        gp.lin_input[(size_t)b * 2 * N + q] = Ul[2 * ks + c];
    }
}

// after the plant: the lap log the simulator keeps (ModelBase.update_memory: the new state -- the one that crossed the line
// with its s unwrapped -- and the applied input) and which races have just completed a lap (-> crx_lmpc_addtraj_dev)
__global__ void __launch_bounds__(256) crx_game_log_kernel(const crx_game_kparams gp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= gp.batch) return;
    const int P = gp.n_points;
    const int cr = gp.laps[b] > gp.laps_prev[b] ? 1 : 0;
    gp.laps_prev[b] = gp.laps[b];
    gp.crossed[b] = cr;
    const int n = gp.n_log[b];
    const int ix = n < P - 1 ? n : P - 1;
    int iu = n - 1; iu = iu < 0 ? 0 : (iu > P - 1 ? P - 1 : iu);
    const double* x = gp.xcurv + (size_t)6 * b;
    double* lx = gp.log_x + ((size_t)b * P + ix) * 6;
    for (int c = 0; c < 6; c++) lx[c] = x[c];
    lx[4] = x[4] + gp.lap_length * (double)cr;
    double* lu = gp.log_u + ((size_t)b * P + iu) * 2;
    lu[0] = gp.u[2 * b]; lu[1] = gp.u[2 * b + 1];
    gp.n_log[b] = n + 1;
}

// ------------------------------------------------------------------------------------------------
// Dispatch order (include/crx.h, "Dispatch order").  A key of 257 values per problem, key 0 dispatched first:
//   mode 0  255 - min(iterations of the previous solve, 255)              (longest first)
//   mode 1  with no previous solve, from the barrier h = (ds/l)^degree + (dey/w)^degree - 1 - margin of two cheap guesses
//           (BASELINE configs[3] draw, iterations by class: 38 / 21 / 14 / 12 on average):
//           0..127    the START state is inside a safety ellipse (h0 < 0: the NLP that needs the restoration phase), deepest first;
//           128..191  the un-steered PATH -- s_j = s_0 + j A[4][0] vx_0 along the target ey_j -- enters one (hP < 0), deepest first;
//           192..255  neither: 2.1 steps per doubling of 1 + hP, nearest first.
//   256     masked-out problems (active[b] == 0): they return at once, last.
// crx_order_key_kernel (mode 1, one thread per problem, the whole chip) leaves the keys in order[]; crx_order_kernel -- ONE
// workgroup of 16 waves -- sorts the batch by key with a STABLE counting sort (wave w owns a contiguous segment; ranks inside a
// 64-problem step by lane comparison), so the order, and with it the timing of the solver launch that uses it, is a pure
// function of the keys.  The keys sit in LDS (2 B each) between the counting and the scatter pass: batch <= CRX_ORDER_LDS_KEYS;
// larger batches evaluate the key twice instead.  HBM: the key's inputs once (twice beyond the LDS limit), 4 B out per problem.
// ------------------------------------------------------------------------------------------------
#define CRX_ORDER_WAVES 16
#define CRX_ORDER_KEYS 257
#define CRX_ORDER_LDS_KEYS 65536

__device__ __forceinline__ bool vx0_finite(const crx_order_kparams& op, int b) {
    return isfinite(op.x0[(size_t)b * 6]) && isfinite(op.x0[(size_t)b * 6 + 4]) && isfinite(op.x0[(size_t)b * 6 + 5]);
}
__device__ __forceinline__ int crx_order_key(const crx_order_kparams& op, int b) {
    if (op.active && op.active[b] == 0) return CRX_ORDER_KEYS - 1;
    if (op.mode == 0) return 255 - min(max(op.iters[b], 0), 255);
    const int V = op.V, n = (V > 0 && op.n_obs) ? min(max(op.n_obs[b], 0), V) : 0;   // n_obs may be NULL for an obstacle-free batch (crx_cbf_order_dev)
    const double s = op.x0[(size_t)b * 6 + 4], ey = op.x0[(size_t)b * 6 + 5];
    double hmin = 1e30;
    for (int o = 0; o < n; o++) {
        const size_t r = (size_t)b * V + o;
        double ls = op.obs_dims ? op.obs_dims[r * 2] : op.l_sum, ws = op.obs_dims ? op.obs_dims[r * 2 + 1] : op.w_sum;
        // same fall-back as the solver kernel (crx_kernels.hip set-up): a non-positive or non-finite device-resident entry -> the descriptor's pair
        if (!(ls > 0.0) || !isfinite(ls)) ls = op.l_sum;
        if (!(ws > 0.0) || !isfinite(ws)) ws = op.w_sum;
        const double ds = (op.obs_s[r * op.stride] + op.lap_off[r] - s) / ls, de = (op.obs_ey[r * op.stride] - ey) / ws;
        double ps = ds * ds, pe = de * de;
        for (int k = 2; k < op.degree; k += 2) { ps *= ds * ds; pe *= de * de; }
        hmin = fmin(hmin, ps + pe - 1.0 - op.margin);
    }
    if (!(hmin == hmin) || !(vx0_finite(op, b))) return 0;   // NaN inputs (device-resident data cannot be validated on the host): first, like the deepest crash state
    if (hmin < 0.0) return min(max((int)((hmin + 1.0 + op.margin) * (128.0 / (1.0 + op.margin))), 0), 127);
    const double vx = op.x0[(size_t)b * 6];
    double hp = 1e30;
    for (int j = 0; j < op.stride; j++) {
        const double sj = s + (double)j * op.ds_per_vx * vx;
        const double eyj = op.per_stage_target ? op.xt[((size_t)b * op.stride + j) * 6 + 5] : op.xt[(size_t)b * 6 + 5];
        for (int o = 0; o < n; o++) {
            const size_t r = (size_t)b * V + o;
            double ls = op.obs_dims ? op.obs_dims[r * 2] : op.l_sum, ws = op.obs_dims ? op.obs_dims[r * 2 + 1] : op.w_sum;
            if (!(ls > 0.0) || !isfinite(ls)) ls = op.l_sum;
            if (!(ws > 0.0) || !isfinite(ws)) ws = op.w_sum;
            const double ds = (op.obs_s[r * op.stride + j] + op.lap_off[r] - sj) / ls, de = (op.obs_ey[r * op.stride + j] - eyj) / ws;
            double ps = ds * ds, pe = de * de;
            for (int k = 2; k < op.degree; k += 2) { ps *= ds * ds; pe *= de * de; }
            hp = fmin(hp, ps + pe - 1.0 - op.margin);
        }
    }
    if (!(hp == hp)) return 128;
    if (hp < 0.0) return 128 + min(max((int)((hp + 1.0 + op.margin) * (64.0 / (1.0 + op.margin))), 0), 63);
    const double t = 192.0 + 2.1 * log2(1.0 + hp);         // 255 at hP = 1e9: an obstacle ~30 ellipse lengths off the path
    return t < 255.0 ? (int)t : 255;
}

__global__ void __launch_bounds__(256) crx_order_key_kernel(const crx_order_kparams op) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < op.batch) op.order[b] = crx_order_key(op, b);
}

// keys_in_order: order[] holds the keys (crx_order_key_kernel ran); cache: the keys fit in LDS
template <bool CACHE>
__global__ void __launch_bounds__(64 * CRX_ORDER_WAVES) crx_order_kernel(const crx_order_kparams op, const int keys_in_order) {
    extern __shared__ int order_lds[];
    int (*hist)[CRX_ORDER_KEYS] = (int (*)[CRX_ORDER_KEYS])order_lds;   // per (wave, key): count, then first output position
    int* start = order_lds + CRX_ORDER_WAVES * CRX_ORDER_KEYS;
    unsigned short* keys = (unsigned short*)(start + CRX_ORDER_KEYS + 1);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int steps = (op.batch + 64 * CRX_ORDER_WAVES - 1) / (64 * CRX_ORDER_WAVES);   // 64-problem steps per wave segment
    const int seg0 = w * steps * 64;
    for (int i = tid; i < CRX_ORDER_WAVES * CRX_ORDER_KEYS; i += blockDim.x) order_lds[i] = 0;
    __syncthreads();
    for (int st = 0; st < steps; st++) {
        const int b = seg0 + st * 64 + lane;
        if (b < op.batch) {
            const int key = keys_in_order ? min(max(op.order[b], 0), CRX_ORDER_KEYS - 1) : crx_order_key(op, b);
            if (CACHE) keys[b] = (unsigned short)key;
            atomicAdd(&hist[w][key], 1);
        }
    }
    __syncthreads();
    if (tid < CRX_ORDER_KEYS) {
        int t = 0;
        for (int v = 0; v < CRX_ORDER_WAVES; v++) t += hist[v][tid];
        start[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k < CRX_ORDER_KEYS; k++) { const int t = start[k]; start[k] = run; run += t; }
    }
    __syncthreads();
    if (tid < CRX_ORDER_KEYS) {
        int run = start[tid];
        for (int v = 0; v < CRX_ORDER_WAVES; v++) { const int t = hist[v][tid]; hist[v][tid] = run; run += t; }
    }
    __syncthreads();
    for (int st = 0; st < steps; st++) {
        const int b = seg0 + st * 64 + lane;
        const int key = b < op.batch ? (CACHE ? (int)keys[b] : crx_order_key(op, b)) : -1;
        int rank = 0, same = 0;
        for (int k = 0; k < 64; k++) {
            const int other = __shfl(key, k);
            rank += (other == key && k < lane);
            same += (other == key);
        }
        const int base = key >= 0 ? hist[w][key] : 0;
        __syncthreads();
        if (key >= 0) {
            op.order[base + rank] = b;
            if (rank == same - 1) hist[w][key] = base + same;   // the last lane of every key of this step moves the key's cursor
        }
        __syncthreads();
    }
}

hipError_t crx_launch_order(const crx_order_kparams& op, hipStream_t st) {
    if (op.batch == 0) return hipSuccess;
    const bool cache = op.batch <= CRX_ORDER_LDS_KEYS;
    const size_t fixed = (size_t)(CRX_ORDER_WAVES * CRX_ORDER_KEYS + CRX_ORDER_KEYS + 1) * sizeof(int);
    const size_t bytes = fixed + (cache ? (size_t)op.batch * sizeof(unsigned short) : 0);
    static int attr_set_on = -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (attr_set_on != dev) {
        hipError_t e = hipFuncSetAttribute((const void*)crx_order_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(fixed + (size_t)CRX_ORDER_LDS_KEYS * sizeof(unsigned short)));
        if (e != hipSuccess) return e;
        attr_set_on = dev;
    }
    int keyed = 0;
    if (op.mode == 1 && cache) {          // the barrier key on the whole chip, left in order[]
        hipLaunchKernelGGL(crx_order_key_kernel, dim3((op.batch + 255) / 256), dim3(256), 0, st, op);
        keyed = 1;
    }
    if (cache) hipLaunchKernelGGL(crx_order_kernel<true>, dim3(1), dim3(64 * CRX_ORDER_WAVES), bytes, st, op, keyed);
    else hipLaunchKernelGGL(crx_order_kernel<false>, dim3(1), dim3(64 * CRX_ORDER_WAVES), bytes, st, op, keyed);
    return hipGetLastError();
}

hipError_t crx_launch_game(int which, const crx_game_kparams& gp, hipStream_t st) {
    if (gp.batch == 0) return hipSuccess;
    const long long n = which == 0 ? (long long)gp.batch * gp.n_cars : (which == 2 ? (long long)gp.batch * (8 * gp.N + 6) : gp.batch);
    if (n == 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    switch (which) {
        case 0: hipLaunchKernelGGL(crx_game_traffic_kernel, grid, block, 0, st, gp); break;
        case 1: hipLaunchKernelGGL(crx_game_masks_kernel, grid, block, 0, st, gp); break;
        case 2: hipLaunchKernelGGL(crx_game_commit_kernel, grid, block, 0, st, gp); break;
        default: hipLaunchKernelGGL(crx_game_log_kernel, grid, block, 0, st, gp); break;
    }
    return hipGetLastError();
}
