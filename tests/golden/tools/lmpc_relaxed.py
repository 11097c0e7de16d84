"""[r6] Pin libcrx's OWN semantics for a learning-MPC QP the reference makes infeasible.

The reference pins the terminal slack to zero (control/control.py:694-695), so its QP has no feasible point whenever the regression model cannot
reach the safe-set hull -- 81 of the 160 states of the benched `game` loop (tests/golden/game_draw.npz) -- and it then applies whatever IPOPT's
restoration phase left in `opti.debug` (:711-722), which nothing here can restate.  libcrx and the oracle re-solve such a QP with the initial-state
equality relaxed:  x_0 = xcurv + w,  cost += w_x0 w'w  (w_x0 = 1e4), the rows on stage 0 dropped (they are rows on the now free x_0; the first
attempt never imposes them either: on the fixed x_0 they are constants), everything else as the reference states it.  That problem is a strictly
convex QP with ONE solution -- so it can be pinned by a third solver, independently of oracle and kernel:

  * `LmpcQP` writes the QP down explicitly, in the reference's full space (X, U, lambd [, w]), from the problem data the reference itself produced
    for each state (game_draw.npz: x, u_old, its regression's A / B / C, its safe-set selection ss / qfun) -- control.py:610-730 line by line;
  * the UNRELAXED form is checked against the reference: on the 79 feasible instances it must reproduce the certified solutions of the QP the
    reference's own `control.lmpc` built under the recording stand-in (game_draw.npz X / U);
  * the RELAXED form of the 81 infeasible instances goes through tools/nlp_solve.py (HiGHS feasibility, SLSQP, exact active-set polish) and its
    solver-agnostic KKT certificate; the certified solutions are stored in game_draw.npz (relaxed_ok / relaxed_X / relaxed_U / relaxed_w / relaxed_cert).

    python tests/golden/tools/lmpc_relaxed.py          (build container; needs numpy + scipy only -- no reference import, no oracle, no libcrx)
tests/test_draw_fixtures.py (oracle) and tests/test_gpu_parity.py (kernel) compare their relaxed plans with these.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import nlp_solve  # noqa: E402

OUT = os.path.normpath(os.path.join(HERE, ".."))
# LMPCRacingParam (utils/base.py:350-376), SystemParam (:708-713), the literal x_track of control.lmpc (control.py:649); Q = 0 by default
R = np.array([1.0, 0.25]); dR = np.array([4.0, 0.0]); Q = np.zeros(6); X_TRACK = np.array([5.0, 0, 0, 0, 0, 0])
V_MAX, EY_MAX, DELTA_MAX, A_MAX, W_X0 = 10.0, 1.0, 0.5, 1.0, 1e4


class LmpcQP:
    """The QP of control.lmpc for one state, as matrices over z = [X (N+1 x 6), U (N x 2), lambd (M) [, w (6)]].
    relaxed = False: the reference's problem (control.py:650-697; the slack it pins to zero is left out);
    relaxed = True:  x_0 = xcurv + w, cost += W_X0 w'w, no rows on stage 0."""

    def __init__(self, x, u_old, A, B, C, ss, qfun, relaxed):
        N, M = A.shape[0], ss.shape[1]
        self.N, self.M, self.relaxed = N, M, relaxed
        nx, nu = 6 * (N + 1), 2 * N
        self.ix = lambda k, i: 6 * k + i
        self.iu = lambda k, c: nx + 2 * k + c
        self.il = lambda j: nx + nu + j
        self.iw = lambda c: nx + nu + M + c
        n = nx + nu + M + (6 if relaxed else 0)
        self.nvar = n
        H = np.zeros((n, n)); g = np.zeros(n); f0 = 0.0
        for i in range(N + 1):                                   # (x_i - x_track)' Q (x_i - x_track)  (:671-674, :686-689)
            for c in range(6):
                H[self.ix(i, c), self.ix(i, c)] += 2.0 * Q[c]; g[self.ix(i, c)] += -2.0 * Q[c] * X_TRACK[c]; f0 += Q[c] * X_TRACK[c] ** 2
        for i in range(N):
            for c in range(2):
                a = self.iu(i, c)
                H[a, a] += 2.0 * R[c]                            # u' R u  (:675)
                H[a, a] += 2.0 * dR[c]                           # (u_i - u_{i-1})' dR (u_i - u_{i-1})  (:676-685)
                if i == 0:
                    g[a] += -2.0 * dR[c] * u_old[c]; f0 += dR[c] * u_old[c] ** 2
                else:
                    p = self.iu(i - 1, c)
                    H[p, p] += 2.0 * dR[c]; H[a, p] -= 2.0 * dR[c]; H[p, a] -= 2.0 * dR[c]
        for j in range(M):
            g[self.il(j)] += qfun[j]                             # Qfun' lambd  (:696)
        if relaxed:
            for c in range(6):
                H[self.iw(c), self.iw(c)] += 2.0 * W_X0
        self.H, self.g, self.f0 = H, g, f0
        Je, be = [], []
        for c in range(6):                                       # x_0 == xcurv (:651)  /  x_0 - w == xcurv
            r = np.zeros(n); r[self.ix(0, c)] = 1.0
            if relaxed:
                r[self.iw(c)] = -1.0
            Je.append(r); be.append(x[c])
        for i in range(N):                                       # x_{i+1} == A_i x_i + B_i u_i + C_i  (:654-657)
            for c in range(6):
                r = np.zeros(n); r[self.ix(i + 1, c)] = 1.0
                for j in range(6):
                    r[self.ix(i, j)] -= A[i, c, j]
                for j in range(2):
                    r[self.iu(i, j)] -= B[i, c, j]
                Je.append(r); be.append(C[i, c])
        for c in range(6):                                       # x_N == SS lambd  (:691-692)
            r = np.zeros(n); r[self.ix(N, c)] = 1.0
            for j in range(M):
                r[self.il(j)] = -ss[c, j]
            Je.append(r); be.append(0.0)
        r = np.zeros(n)
        for j in range(M):
            r[self.il(j)] = 1.0
        Je.append(r); be.append(1.0)                             # 1' lambd == 1  (:693)
        self.Je, self.be = np.array(Je), np.array(be)
        Ji, bi = [], []                                          # rows as Ji z - bi >= 0
        for i in range(N):
            for c, ub in ((0, DELTA_MAX), (1, A_MAX)):           # (:664-669)
                r = np.zeros(n); r[self.iu(i, c)] = 1.0; Ji.append(r); bi.append(-ub)
                r = np.zeros(n); r[self.iu(i, c)] = -1.0; Ji.append(r); bi.append(-ub)
            if i == 0 and relaxed:
                continue                                         # rows on the free x_0: not part of the relaxed problem
            for comp, sgn, bnd in ((0, -1.0, V_MAX), (5, -1.0, EY_MAX), (5, 1.0, EY_MAX)):   # vx_i <= v_max (:659), |ey_i| <= lap_width (:660-662)
                r = np.zeros(n); r[self.ix(i, comp)] = sgn; Ji.append(r); bi.append(-bnd)
        for j in range(M):
            r = np.zeros(n); r[self.il(j)] = 1.0; Ji.append(r); bi.append(0.0)               # lambd >= 0 (:690)
        self.Ji, self.bi = np.array(Ji), np.array(bi)

    def z0(self):
        return np.zeros(self.nvar)

    def eval_all(self, z):
        z = np.asarray(z, dtype=float)
        return (float(self.f0 + z @ (0.5 * (self.H @ z) + self.g)), self.H @ z + self.g, self.Je @ z - self.be, self.Je, self.Ji @ z - self.bi, self.Ji)

    def unpack(self, z):
        N, M = self.N, self.M
        X = z[: 6 * (N + 1)].reshape(N + 1, 6); U = z[6 * (N + 1): 6 * (N + 1) + 2 * N].reshape(N, 2)
        lam = z[6 * (N + 1) + 2 * N: 6 * (N + 1) + 2 * N + M]
        w = z[6 * (N + 1) + 2 * N + M:] if self.relaxed else np.zeros(6)
        return X, U, lam, w


def main():
    path = os.path.join(OUT, "game_draw.npz")
    z = dict(np.load(path))
    n = len(z["phase"])
    # (1) the explicit QP is the reference's: the feasible instances reproduce the certified solutions of the reference-built graphs
    worst = 0.0
    n_chk = 0
    for r in range(n):
        if int(z["lp_status"][r]) != 0 or not bool(z["success"][r]):
            continue
        qp = LmpcQP(z["x"][r], z["u_old"][r], z["A"][r], z["B"][r], z["C"][r], z["ss"][r], z["qfun"][r], relaxed=False)
        sol, info = nlp_solve.solve_recorded(qp)
        assert info["success"], (r, info["reason"])
        X, U, _, _ = qp.unpack(sol)
        dx, du = np.abs(X - z["X"][r]).max(), np.abs(U - z["U"][r]).max()
        worst = max(worst, dx, du)
        assert dx <= 1e-6 and du <= 1e-6, (r, dx, du)
        n_chk += 1
    print("unrelaxed form vs the reference-built QPs: %d feasible instances reproduced, worst |dX|, |dU| %.1e" % (n_chk, worst), flush=True)
    # (2) the relaxed form of every instance HiGHS found infeasible, through the third solver and its certificate
    N, M = z["A"].shape[1], z["ss"].shape[2]
    out = dict(relaxed_ok=np.zeros(n, dtype=bool), relaxed_X=np.full((n, N + 1, 6), np.nan), relaxed_U=np.full((n, N, 2), np.nan),
               relaxed_w=np.full((n, 6), np.nan), relaxed_lambda=np.full((n, M), np.nan), relaxed_cert=np.full((n, 6), np.nan))
    for r in range(n):
        if int(z["lp_status"][r]) != 2:
            continue
        qp = LmpcQP(z["x"][r], z["u_old"][r], z["A"][r], z["B"][r], z["C"][r], z["ss"][r], z["qfun"][r], relaxed=True)
        sol, info = nlp_solve.solve_recorded(qp)
        if not info["success"]:
            print("instance %d (phase %d race %d): relaxed QP not certified: %s" % (r, z["phase"][r], z["race"][r], info["reason"]), flush=True)
            continue
        X, U, lam, w = qp.unpack(sol)
        out["relaxed_ok"][r] = True
        out["relaxed_X"][r], out["relaxed_U"][r], out["relaxed_w"][r], out["relaxed_lambda"][r] = X, U, w, lam
        out["relaxed_cert"][r] = [info[k] for k in ("f", "stationarity", "eq_violation", "ineq_violation", "min_multiplier", "complementarity")]
    print("relaxed form: %d of %d infeasible instances solved and certified; |w| max %.3e" % (
        out["relaxed_ok"].sum(), int((z["lp_status"] == 2).sum()), np.nanmax(np.abs(out["relaxed_w"]))), flush=True)
    z.update(out)
    np.savez_compressed(path, **z)


if __name__ == "__main__":
    main()
