"""Host-side mirror (car-racing_amd/{utils,system,planning,racing}) against fixtures produced by the
reference's own solver-free code (tests/golden/harness.npz, planner.npz).  CPU only."""
import os

import numpy as np
import pytest

import conftest


@pytest.fixture(scope="module")
def H():
    return np.load(os.path.join(conftest.GOLDEN, "harness.npz"))


def _track(name, width=1.0):
    from utils import racing_env

    spec = np.genfromtxt(os.path.join(conftest.ROOT, "data/track_layout/%s.csv" % name), delimiter=",")
    return racing_env.ClosedTrack(spec, track_width=width)


@pytest.mark.parametrize("name", ["l_shape", "m_shape", "goggle", "ellipse"])
def test_track_geometry(H, name):
    tr = _track(name)
    assert tr.lap_length == float(H[name + "/lap_length"])
    np.testing.assert_allclose(tr.point_and_tangent, H[name + "/table"], atol=1e-12)
    xy = np.array([tr.get_global_position(a, b) for a, b in zip(H[name + "/s"], H[name + "/ey"])])
    np.testing.assert_allclose(xy, H[name + "/xy"], atol=1e-12)
    psi = np.array([tr.get_orientation(a, b) for a, b in zip(H[name + "/s"], H[name + "/ey"])])
    np.testing.assert_allclose(psi, H[name + "/psi"], atol=1e-12)
    curv = np.array([tr.get_curvature(a) for a in H[name + "/s_curv"]])
    np.testing.assert_array_equal(curv, H[name + "/curv"])


def test_plant_step(H):
    from system import vehicle_dynamics as vd
    from utils import base

    dyn = base.CarParam().dynamics_param
    for i in range(len(H["plant/u"])):
        g, c = vd.vehicle_dynamics(dyn, H["plant/curv"][i], H["plant/xglob"][i], H["plant/xcurv"][i], 0.001, H["plant/u"][i])
        np.testing.assert_allclose(g, H["plant/xglob_next"][i], rtol=0, atol=1e-15)
        np.testing.assert_allclose(c, H["plant/xcurv_next"][i], rtol=0, atol=1e-15)


def test_pid_closed_loop_and_predictions(H):
    import sympy as sp

    from racing import offboard
    from utils import base

    track = _track("l_shape", 0.8)
    ego = offboard.DynamicBicycleModel(name="ego", param=base.CarParam(), system_param=base.SystemParam())
    ego.set_zero_noise()
    ego.set_state_curvilinear(np.zeros(6)); ego.set_state_global(np.zeros(6)); ego.start_logging()
    ego.set_ctrl_policy(offboard.PIDTracking(vt=0.8)); ego.ctrl_policy.set_timestep(0.1)
    ego.set_track(track); ego.ctrl_policy.set_track(track)
    sim = offboard.CarRacingSim(); sim.set_timestep(0.1); sim.set_track(track); sim.add_vehicle(ego)
    ego.ctrl_policy.set_racing_sim(sim)
    sim.sim(sim_time=3.0)
    np.testing.assert_allclose(np.array(ego.xcurv_log), H["pid/xcurv_log"], atol=1e-12)
    np.testing.assert_allclose(np.array(ego.xglob_log), H["pid/xglob_log"], atol=1e-12)
    t = sp.symbols("t")
    car = offboard.NoDynamicsModel(name="car1", param=base.CarParam()); car.set_track(track); car.set_timestep(0.1)
    car.set_state_curvilinear_func(t, 0.2 * t + 4.0, 0.1 + 0.0 * t); car.time = 1.3
    np.testing.assert_allclose(car.get_trajectory_nsteps(0.0, 0.1, 11)[0], H["pred/nodyn"], atol=1e-13)
    d = offboard.DynamicBicycleModel(name="d", param=base.CarParam(), system_param=base.SystemParam())
    d.set_track(track); d.set_timestep(0.1)
    d.set_state_curvilinear(np.array([0.9, 0.02, 0.1, 0.05, 18.9, 0.2])); d.set_state_global(np.array([0.9, 0.02, 0.1, 0.3, 1.0, 0.5]))
    np.testing.assert_allclose(d.get_trajectory_nsteps(11)[0], H["pred/dyn"], atol=1e-13)


def test_interest_window(H):
    from planning import planner_helper as ph
    from utils import base

    par = base.RacingGameParam(timestep=0.1)
    lap = _track("l_shape").lap_length

    class V:
        def __init__(self, xc):
            self.xcurv, self.param = np.array(xc, float), base.CarParam()

    for se, sa, dv, want in H["interest/cases"]:
        got = ph.check_ego_agent_distance(V([1.0, 0, 0, 0, se, 0]), V([1.0 - dv, 0, 0, 0, sa, 0]), par, lap)
        assert got == bool(want), (se, sa, dv)


def test_bezier_sort_and_bounds_vs_reference(golden_planner):
    """planner_helper mirror + hostprep against what the reference computed in each planner scenario."""
    from crx import hostprep
    from planning import planner_helper as ph

    opt = np.genfromtxt(os.path.join(conftest.ROOT, "data/optimal_traj/xcurv_l_shape.csv"), delimiter=",")
    for name in golden_planner.names:
        c = golden_planner.case(name)
        if not bool(c["overtake_flag"]):
            continue
        N = int(c["N"])
        names = [str(x) for x in c["veh_names"]]
        interest = [n for n, f in zip(names, c["veh_is_interest"]) if f]
        xc = {n: c["veh_xcurv"][i] for i, n in enumerate(names)}
        order = ph.sort_by_ey(interest, lambda n: xc[n][5])
        assert order == [str(x) for x in c["sorted_vehicles"]]
        pred = {str(n): c["obs_pred"][i] for i, n in enumerate(c["sorted_vehicles"])}
        vi = np.array([[xc[n][4], pred[n][5].max(), pred[n][5].min()] for n in interest])
        mdv = max(abs(c["x_raw"][0] - xc[n][0]) for n in order)
        cp = ph.bezier_control_points(len(interest), vi, mdv, 0.5, float(c["width"]), float(c["lap_length"]), 0.2, opt,
                                      c["x_wrapped"])
        np.testing.assert_allclose(ph.bezier_polylines(cp, N), c["bezier_xcurvs"], atol=1e-13)
    # wrap helper == the reference's while-loop
    s = np.array([-3.0, 0.0, 5.0, 19.22957795362994, 19.3, 40.0, 60.0])
    lap = 19.22957795362994
    want = s.copy()
    for i in range(len(want)):
        while want[i] > lap:
            want[i] -= lap
    np.testing.assert_allclose(hostprep.wrap_above(s, lap), want, atol=1e-12)
