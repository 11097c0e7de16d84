"""Closed-loop learning-MPC laps on the GPU (the bench's `game` workload); after every step the QP with the most
iterations is inspected and, past a threshold, its inputs are dumped to gpurun_out/lmpc_stragglers.npz for analysis
on the CPU (oracle, tools/parity_trace.py).  Usage: python tools/lmpc_stragglers.py [steps] [threshold]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "car-racing_amd"))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    thr = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    cx = bench.Ctx()
    args = argparse.Namespace()
    w = bench.make_game(cx, args)
    laps = w.step.__self__
    dump, hist = {}, []
    for s in range(steps):
        u_prev = laps.u_old.clone()
        laps.step()
        torch.cuda.synchronize()
        it = laps.ws.iters.cpu().numpy(); st = laps.ws.status.cpu().numpy()
        hist.append((int(it.max()), int(np.median(it)), int((st != 0).sum())))
        b = int(it.argmax())
        if it[b] >= thr and len([k for k in dump if k.startswith('s')]) < 40 * 10:
            k = "s%03d_" % s
            # after step() the state buffers are swapped: xc_next holds the state this step's QPs started from
            dump[k + "A"] = laps.pws.A[b].cpu().numpy(); dump[k + "B"] = laps.pws.B[b].cpu().numpy(); dump[k + "C"] = laps.pws.C[b].cpu().numpy()
            dump[k + "ss"] = laps.pws.ss[b].cpu().numpy(); dump[k + "qfun"] = laps.pws.qfun[b].cpu().numpy()
            dump[k + "x0"] = laps.xc_next[b].cpu().numpy(); dump[k + "u_old"] = u_prev[b].cpu().numpy()
            dump[k + "n_ss"] = np.int32(laps.n_ss[b].item()); dump[k + "iters"] = np.int32(it[b]); dump[k + "status"] = np.int32(st[b])
        if len(sys.argv) > 3 and s % 10 == 5:        # optional third argument: also dump the first 256 QPs of every 10th step
            k = "all%03d_" % s
            for nm, t in (("A", laps.pws.A), ("B", laps.pws.B), ("C", laps.pws.C), ("ss", laps.pws.ss), ("qfun", laps.pws.qfun),
                          ("x0", laps.xc_next), ("u_old", u_prev), ("n_ss", laps.n_ss), ("iters", laps.ws.iters), ("status", laps.ws.status)):
                dump[k + nm] = t[:256].cpu().numpy()
    print("per step (max iters, median, n status != 0):", hist)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "lmpc_stragglers.npz"), **dump)
    print("dumped", len(dump) // 10, "QPs")


if __name__ == "__main__":
    main()
