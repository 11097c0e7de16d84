"""Plant model of the simulator harness: dynamic bicycle with Pacejka tyres, one explicit Euler step
of length delta_t in global and curvilinear coordinates.  Same signature as the reference's
system/vehicle_dynamics.py:4-49 (host-side, out of scope for the GPU)."""
import numpy as np


def vehicle_dynamics(dynamics_param, curv, xglob, xcurv, delta_t, u):
    m, lf, lr, Iz, Df, Cf, Bf, Dr, Cr, Br = dynamics_param.get_params()
    delta, acc = u[0], u[1]
    vx, vy, wz, epsi, s, ey = xcurv[:6]
    psi, X, Y = xglob[3], xglob[4], xglob[5]
    # tyre slip angles and lateral forces (the reference uses lf for the rear axle too, :25)
    slip_f = delta - np.arctan2(vy + lf * wz, vx)
    slip_r = -np.arctan2(vy - lf * wz, vx)
    Fyf = 2 * Df * np.sin(Cf * np.arctan(Bf * slip_f))
    Fyr = 2 * Dr * np.sin(Cr * np.arctan(Br * slip_r))
    dvx = acc - 1 / m * Fyf * np.sin(delta) + wz * vy
    dvy = 1 / m * (Fyf * np.cos(delta) + Fyr) - wz * vx
    dwz = 1 / Iz * (lf * Fyf * np.cos(delta) - lr * Fyr)
    v_long = (vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - curv * ey)
    body = np.array([vx + delta_t * dvx, vy + delta_t * dvy, wz + delta_t * dwz])
    xglob_next = np.empty(len(xglob))
    xcurv_next = np.empty(len(xcurv))
    xglob_next[0:3] = body
    xglob_next[3] = psi + delta_t * wz
    xglob_next[4] = X + delta_t * (vx * np.cos(psi) - vy * np.sin(psi))
    xglob_next[5] = Y + delta_t * (vx * np.sin(psi) + vy * np.cos(psi))
    xcurv_next[0:3] = body
    xcurv_next[3] = epsi + delta_t * (wz - v_long * curv)
    xcurv_next[4] = s + delta_t * v_long
    xcurv_next[5] = ey + delta_t * (vx * np.sin(epsi) + vy * np.cos(epsi))
    return xglob_next, xcurv_next
