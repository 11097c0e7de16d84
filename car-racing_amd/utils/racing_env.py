"""Track geometry: a closed track made of straight and circular-arc segments, in curvilinear
coordinates (s along the centre line, ey to the left).

Host-side support for the simulator harness (out of scope for the GPU).
Same public surface as the reference's utils/racing_env.py (`ClosedTrack`, module-level
`get_curvature`), independent implementation: every segment stores its start pose, and positions
follow in closed form from that pose.  `point_and_tangent` keeps the reference's table layout
[x_end, y_end, psi_end, s_start, length, curvature] because vehicle models read it
(reference utils/base.py:765-769).
"""
import numpy as np

_S_TOL = 0.001  # the reference accepts s up to 1 mm past a segment end (racing_env.py:13,19)


def _wrap_pi(a):
    if a < -np.pi:
        return a + 2.0 * np.pi
    if a > np.pi:
        return a - 2.0 * np.pi
    return a


def _wrap_s(s, lap_length):
    while s > lap_length:
        s = s - lap_length
    while s < 0:
        s = s + lap_length
    return s


def build_table(spec):
    """spec rows: (length, radius) with radius 0 for a straight; returns the (n+1, 6) table whose last
    row is the straight that closes the loop back to the origin with heading 0."""
    n = spec.shape[0]
    tab = np.zeros((n + 1, 6))
    x = y = psi = s0 = 0.0
    for i in range(n):
        length, radius = float(spec[i, 0]), float(spec[i, 1])
        if radius == 0.0:
            x, y = x + length * np.cos(psi), y + length * np.sin(psi)
            curv = 0.0
        else:
            turn = 1.0 if radius >= 0 else -1.0
            r = abs(radius)
            cx, cy = x + r * np.cos(psi + turn * np.pi / 2), y + r * np.sin(psi + turn * np.pi / 2)
            start_ang = _centre_to_start_angle(psi, turn)
            span = length / r
            x, y = cx + r * np.cos(start_ang + turn * span), cy + r * np.sin(start_ang + turn * span)
            psi = _wrap_pi(psi + span * np.sign(radius))
            curv = 1.0 / radius
        tab[i] = (x, y, psi, s0, length, curv)
        s0 += length
    closing = np.hypot(tab[n - 1, 0], tab[n - 1, 1])
    tab[n] = (0.0, 0.0, 0.0, s0, closing, 0.0)
    return tab


def _centre_to_start_angle(psi, turn):
    """Direction from an arc's centre to its start point, folded the way the reference folds it
    (racing_env.py:58-59) so that sums of angles round identically."""
    normal = _wrap_pi(turn * np.pi / 2 + psi)
    return -(np.pi - abs(normal)) * (1.0 if normal >= 0 else -1.0)


def _segment_of(tab, s, inclusive_end=False):
    lo = tab[:, 3]
    hi = lo + tab[:, 4]
    hit = (s >= lo) & ((s <= hi) if inclusive_end else (s < hi + _S_TOL))
    return int(np.nonzero(hit)[0][0])  # first match at a joint


def _pose(tab, s, ey):
    i = _segment_of(tab, s)
    xs, ys, psis = tab[i - 1, 0], tab[i - 1, 1], tab[i - 1, 2]  # i-1 = -1 wraps to the closing row: the origin
    ds = s - tab[i, 3]
    if tab[i, 5] == 0.0:
        psi = tab[i, 2]
        f = ds / tab[i, 4]
        x = (1 - f) * xs + f * tab[i, 0] + ey * np.cos(psi + np.pi / 2)
        y = (1 - f) * ys + f * tab[i, 1] + ey * np.sin(psi + np.pi / 2)
        return x, y, psi
    radius = 1.0 / tab[i, 5]
    turn = 1.0 if radius >= 0 else -1.0
    r = abs(radius)
    cx, cy = xs + r * np.cos(psis + turn * np.pi / 2), ys + r * np.sin(psis + turn * np.pi / 2)
    ang = _centre_to_start_angle(psis, turn) + turn * (ds / (np.pi * r) * np.pi)
    rho = r - turn * ey
    return cx + rho * np.cos(ang), cy + rho * np.sin(ang), ang + np.pi / 2


def get_curvature(lap_length, point_and_tangent, s):
    """Signed curvature at s (reference racing_env.py:225-246; first match at a segment joint)."""
    s = _wrap_s(s, lap_length)
    return point_and_tangent[_segment_of(point_and_tangent, s, inclusive_end=True), 5]


class ClosedTrack:
    def __init__(self, spec, track_width=0.8):
        self.width = track_width
        self.spec = spec
        self.point_and_tangent = build_table(np.asarray(spec, dtype=float))
        self.lap_length = self.point_and_tangent[-1, 3] + self.point_and_tangent[-1, 4]

    def get_global_position(self, s, ey):
        x, y, _ = _pose(self.point_and_tangent, _wrap_s(s, self.lap_length), ey)
        return x, y

    def get_orientation(self, s, ey):
        """Heading of the centre line at s.  On a straight the reference returns the segment's
        heading (racing_env.py:94); on an arc the tangent of the arc (:125)."""
        return _pose(self.point_and_tangent, _wrap_s(s, self.lap_length), ey)[2]

    def get_curvature(self, s):
        return get_curvature(self.lap_length, self.point_and_tangent, s)

    def plot_track(self, ax, center_line=True):
        n = int(np.floor(100 * self.lap_length))
        for off, style in ((self.width, "-b"), (-self.width, "-b")) + (((0.0, "--r"),) if center_line else ()):
            pts = np.array([self.get_global_position(i / 100.0, off) for i in range(n)])
            ax.plot(pts[:, 0], pts[:, 1], style)
