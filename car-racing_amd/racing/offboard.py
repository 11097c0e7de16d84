"""Off-board (single-process) simulator classes with the reference's names (racing/offboard.py).

Plotting / GIF writing is reduced to light matplotlib figures: ImageMagick and ffmpeg are not in
this image, and visualisation is out of scope."""
import numpy as np

from utils import base, racing_env
from utils.constants import X_DIM


class PIDTracking(base.PIDTracking):
    pass


class MPCTracking(base.MPCTracking):
    pass


class MPCCBFRacing(base.MPCCBFRacing):
    def __init__(self, mpc_cbf_param, system_param):
        base.MPCCBFRacing.__init__(self, mpc_cbf_param, system_param)
        self.realtime_flag = False


class LMPCRacingGame(base.LMPCRacingGame):
    def __init__(self, lmpc_param, racing_game_param=None, system_param=None):
        base.LMPCRacingGame.__init__(self, lmpc_param, racing_game_param=racing_game_param, system_param=system_param)
        self.realtime_flag = False


class DynamicBicycleModel(base.DynamicBicycleModel):
    """Adds the zero-input kinematic n-step prediction other controllers use for this vehicle
    (reference offboard.py:51-94)."""

    def get_estimation(self, xglob, xcurv):
        curv = racing_env.get_curvature(self.lap_length, self.point_and_tangent, xcurv[4])
        dt = self.timestep
        v_long = (xcurv[0] * np.cos(xcurv[3]) - xcurv[1] * np.sin(xcurv[3])) / (1 - curv * xcurv[5])
        xc = np.zeros((X_DIM,))
        xc[0:3] = xcurv[0:3]
        xc[3] = xcurv[3] + dt * (xcurv[2] - v_long * curv)
        xc[4] = xcurv[4] + dt * v_long
        xc[5] = xcurv[5] + dt * (xcurv[0] * np.sin(xcurv[3]) + xcurv[1] * np.cos(xcurv[3]))
        xg = np.zeros((X_DIM,))
        xg[0:3] = xglob[0:3]
        xg[3] = xglob[3] + dt * xglob[2]
        # the reference assigns component 4 twice and never component 5 (offboard.py:71-76); kept
        xg[4] = xglob[4] + dt * (xglob[0] * np.sin(xglob[3]) + xglob[1] * np.cos(xglob[3]))
        return xc, xg

    def get_trajectory_nsteps(self, n):
        xc_all, xg_all = np.zeros((X_DIM, n)), np.zeros((X_DIM, n))
        xg, xc = self.xglob, self.xcurv
        for j in range(n):
            xc, xg = self.get_estimation(xg, xc)
            while xc[4] > self.lap_length:
                xc[4] = xc[4] - self.lap_length
            xc_all[:, j], xg_all[:, j] = xc, xg
        return xc_all, xg_all


class NoDynamicsModel(base.NoDynamicsModel):
    pass


class CarRacingSim(base.CarRacingSim):
    def __init__(self):
        base.CarRacingSim.__init__(self)
        self.ax = self.fig = None

    def add_vehicle(self, vehicle):
        self.vehicles[vehicle.name] = vehicle
        vehicle.set_track(self.track)
        vehicle.set_timestep(self.timestep)

    def sim(self, sim_time=50.0, one_lap=False, one_lap_name=None, animating_flag=False):
        start_lap = self.vehicles[one_lap_name].laps if one_lap else None
        for _ in range(int(sim_time / self.timestep)):
            for name in self.vehicles:
                self.vehicles[name].forward_one_step(self.vehicles[name].realtime_flag)
            if one_lap and self.vehicles[one_lap_name].laps > start_lap:
                print("lap completed")
                break

    # -- light-weight stand-ins for the reference's figures -------------------------------------------
    def _trace(self, name):
        return np.array(self.vehicles[name].xcurv_log).reshape(-1, X_DIM)

    def plot_state(self, name):
        import matplotlib.pyplot as plt

        tr = self._trace(name)
        fig, axs = plt.subplots(4)
        for ax, col, lab in zip(axs, (0, 1, 3, 5), ("$v_x$", "$v_y$", "$e_{\\psi}$", "$e_y$")):
            ax.plot(np.arange(len(tr)) * self.timestep, tr[:, col])
            ax.set_ylabel(lab)
        plt.close(fig)

    def plot_input(self, name):
        pass

    def plot_simulation(self):
        import matplotlib.pyplot as plt

        fig, ax = plt.subplots()
        self.track.plot_track(ax)
        for name in self.vehicles:
            g = np.array(self.vehicles[name].xglob_log).reshape(-1, X_DIM)
            ax.plot(g[:, 4], g[:, 5], label=name)
        plt.close(fig)

    def animate(self, filename="untitled", ani_time=400, lap_number=None, racing_game=False, imagemagick=False):
        print("animate(%s): skipped (no ImageMagick/ffmpeg in this image)" % filename)
