"""Data-driven scenario builder for the closed-loop tests.

A scenario is DATA (track name and width, controller targets, scripted cars as (name, s0, v, ey) tuples, a lap plan);
`Race` turns it into the mirrored class surface (utils.base / racing.offboard objects) and runs it.  The parameter
values of the two named scenarios are those of the reference's driver scripts (tests/auto_mpccbf_test.py,
tests/auto_racing_game_test.py); the golden fixtures were recorded from exactly these values."""
import os

import numpy as np

import conftest

# (name, s at t = 0 [m], speed [m/s], lateral offset [m]) of the scripted cars: s(t) = v t + s0, ey(t) = ey
MPCCBF = dict(track="l_shape", width=1.0, dt=0.1, vt=0.8, sim_time=40.0,
              cars=[("car1", 4.0, 0.2, 0.1), ("car2", 10.0, 0.2, -0.1)])
RACING_GAME = dict(track="l_shape", width=1.0, dt=0.1, vt=0.7, alpha=0.8, n_planner=10,
                   lap_plan=("pid", "mpc-lti", "lmpc", "lmpc+traffic"),
                   cars=[("car%d" % (i + 1), 5.5 + 2 * i, 0.7 + 0.02 * i, -0.5 + 0.3 * i) for i in range(2)])


def table(kind, name):
    return np.genfromtxt(os.path.join(conftest.ROOT, "data", kind, name + ".csv"), delimiter=",")


def make_track(name="l_shape", width=1.0):
    from utils import racing_env

    return racing_env.ClosedTrack(table("track_layout", name), track_width=width)


class Race:
    """One simulator with an ego (dynamic bicycle, zero noise, starting at rest on the start line) and scripted cars."""

    def __init__(self, track, dt):
        from racing import offboard
        from utils import base

        self.track, self.dt = track, dt
        self.ego = offboard.DynamicBicycleModel(name="ego", param=base.CarParam(edgecolor="black"), system_param=base.SystemParam())
        self.ego.set_timestep(dt)
        self.ego.set_zero_noise()
        for setter in (self.ego.set_state_curvilinear, self.ego.set_state_global):
            setter(np.zeros(6))
        self.ego.set_track(track)
        self.ego.start_logging()
        self.sim = offboard.CarRacingSim()
        self.sim.set_timestep(dt)
        self.sim.set_track(track)
        self.sim.add_vehicle(self.ego)
        self.cars = []

    def policy(self, ctrl, activate=True):
        """Wire a controller object to the track, the clock and the simulator; optionally make it the ego's policy."""
        ctrl.set_timestep(self.dt)
        ctrl.set_track(self.track)
        ctrl.set_racing_sim(self.sim)
        if activate:
            self.ego.set_ctrl_policy(ctrl)
        return ctrl

    def scripted_car(self, name, s0, v, ey, join=True):
        import sympy as sp
        from racing import offboard
        from utils import base

        t = sp.symbols("t")
        car = offboard.NoDynamicsModel(name=name, param=base.CarParam(edgecolor="orange"))
        car.set_track(self.track)
        car.set_state_curvilinear_func(t, float(v) * t + float(s0), float(ey) + 0.0 * t)
        if join:
            self.join(car)
        return car

    def join(self, car):
        car.start_logging()
        self.sim.add_vehicle(car)
        self.cars.append(car)

    def run(self, sim_time, one_lap=False):
        if one_lap:
            self.sim.sim(sim_time=sim_time, one_lap=True, one_lap_name="ego")
        else:
            self.sim.sim(sim_time=sim_time)


def mpccbf_race(spec=MPCCBF, cars=None, sim_time=None):
    """Ego under MPC-CBF (MPCCBFRacing) among scripted cars."""
    from racing import offboard
    from utils import base

    race = Race(make_track(spec["track"], spec["width"]), spec["dt"])
    race.policy(offboard.MPCCBFRacing(base.MPCCBFRacingParam(vt=spec["vt"]), race.ego.system_param))
    for c in (spec["cars"] if cars is None else cars):
        race.scripted_car(*c)
    race.run(spec["sim_time"] if sim_time is None else sim_time)
    return race


def racing_game(spec=RACING_GAME):
    """The lap plan of the racing game: a PID lap and an mpc-lti lap fill the safe set, then learning-MPC laps, the last
    ones with scripted traffic to overtake.  Returns (race, lmpc_controller)."""
    from control.lmpc_helper import LMPCPrediction
    from racing import offboard
    from utils import base

    dt, laps = spec["dt"], len(spec["lap_plan"])
    race = Race(make_track(spec["track"], spec["width"]), dt)
    race.sim.set_opti_traj(table("optimal_traj", "xglob_" + spec["track"]))
    ctrl = {"pid": race.policy(offboard.PIDTracking(vt=spec["vt"], eyt=0.0)),
            "mpc-lti": race.policy(offboard.MPCTracking(base.MPCTrackingParam(vt=spec["vt"], eyt=0.0), race.ego.system_param), activate=False)}
    horizon = 10000 * dt
    lmpc = offboard.LMPCRacingGame(base.LMPCRacingParam(timestep=dt, lap_number=laps, time_lmpc=horizon),
                                   racing_game_param=base.RacingGameParam(timestep=dt, alpha=spec["alpha"], num_horizon_planner=spec["n_planner"]),
                                   system_param=race.ego.system_param)
    lmpc.set_opti_traj(table("optimal_traj", "xcurv_" + spec["track"]), table("optimal_traj", "xglob_" + spec["track"]))
    lmpc.openloop_prediction = LMPCPrediction(lap_number=laps)
    race.policy(lmpc, activate=False)
    lmpc.set_vehicles_track()
    ctrl["lmpc"] = lmpc
    traffic = [race.scripted_car(*c, join=False) for c in spec["cars"]]
    seeded = False
    for lap, what in enumerate(spec["lap_plan"]):
        kind = what.split("+")[0]
        if kind == "lmpc" and not seeded:            # the first learning lap starts from the laps driven so far
            for k in range(lap):
                lmpc.add_trajectory(race.ego, k)
            seeded = True
        race.ego.set_ctrl_policy(ctrl[kind])
        if what.endswith("+traffic") and not race.cars:
            for car in traffic:
                race.join(car)
            for log in ("solver_time", "all_local_trajs", "all_splines", "xcurv_log", "lmpc_prediction", "mpc_cbf_prediction"):
                setattr(race.ego, log, [])
        race.run(horizon if kind == "lmpc" else 90, one_lap=True)
        if kind == "lmpc":
            lmpc.add_trajectory(race.ego, lap)
    return race, lmpc
