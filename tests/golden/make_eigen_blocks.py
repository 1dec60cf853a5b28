#!/usr/bin/env python3
"""Generate tests/golden/eigen_blocks.json from REAL reference code.

Runs oracle/_ref/ref_blocks (built by `make -C oracle ref` from the reference's own
SymmetricEigenSolver.h and its vendored Eigen 3.2.92, compiled where they lie under
/root/reference — nothing is copied) on seeded inputs and stores inputs + outputs.
Only runnable in the build container (needs /root/reference); the JSON it writes is
the committed fixture that travels to the GPU box.
"""
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_blocks")


def run(lines):
    p = subprocess.run([REF_BIN], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    out = []
    for ln in p.stdout.strip().split("\n"):
        assert ln != "ERR"
        out.append([float(v) for v in ln.split()])
    return out


def fmt(op, vals):
    return op + " " + " ".join("%.17g" % v for v in vals)


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    rng = np.random.RandomState(20240601)
    fx = {}

    # --- 3x3 symmetric covariances like the voxel grid produces: planar, linear, isotropic, tiny, huge
    covs = []
    for k in range(64):
        kind = k % 4
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if kind == 0:      # planar (wall / ground): one tiny eigenvalue -> eigen-floor branch
            lam = np.array([rng.uniform(1e-5, 5e-4), rng.uniform(0.02, 0.09), rng.uniform(0.02, 0.09)])
        elif kind == 1:    # linear (pole): two tiny
            lam = np.array([rng.uniform(1e-5, 4e-4), rng.uniform(1e-5, 4e-4), rng.uniform(0.03, 0.09)])
        elif kind == 2:    # blob
            lam = rng.uniform(0.01, 0.09, 3)
        else:              # far-from-origin cancellation noise: slightly indefinite allowed
            lam = np.array([rng.uniform(-1e-9, 1e-6), rng.uniform(1e-3, 0.05), rng.uniform(0.01, 0.09)])
        C = (Q * lam) @ Q.T
        C = 0.5 * (C + C.T)
        covs.append([C[0, 0], C[0, 1], C[0, 2], C[1, 1], C[1, 2], C[2, 2]])
    covs.append([0.04, 0.0, 0.0, 0.02, 0.0, 0.01])  # diagonal -> norm == 0 branch
    covs.append([0.0] * 6)                          # all-zero branch
    covs.append([0.03, 0.0, 0.0, 0.03, 0.001, 0.03])
    fx["eig3"] = {"in": covs, "out": run([fmt("eig3", c) for c in covs])}

    mats = [rng.normal(size=9).tolist() for _ in range(16)]
    for c in covs[:16]:
        mats.append([c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]])
    fx["inv3"] = {"in": mats, "out": run([fmt("inv3", m) for m in mats])}

    # --- 6x6 NDT-like Hessians (negative definite-ish, badly scaled between translation and rotation)
    sv_in = []
    for k in range(24):
        J = rng.normal(size=(40, 6)) * np.array([1, 1, 1, 25, 25, 25])
        H = -(J.T @ J) * rng.uniform(0.5, 50.0)
        if k % 6 == 5:     # rank-deficient: kill one direction
            v = rng.normal(size=6); v /= np.linalg.norm(v)
            P = np.eye(6) - np.outer(v, v)
            H = P @ H @ P
        b = rng.normal(size=6) * 100.0
        sv_in.append(H.reshape(-1).tolist() + b.tolist())
    fx["svd6"] = {"in": sv_in, "out": run([fmt("svd6", m) for m in sv_in])}

    # --- pose <-> matrix in f32
    poses = []
    for k in range(32):
        p = np.concatenate([rng.uniform(-60, 60, 3), rng.uniform(-0.2, 0.2, 3)])
        if k % 4 == 1:
            p[3:] = rng.uniform(-3.0, 3.0, 3)
        if k % 8 == 2:
            p[3:] = [0.0, 0.0, 0.0]
        if k % 8 == 3:
            p[3:] = [-2e-3, 3e-3, 0.02]   # slightly negative roll: the flipped Euler chart
        poses.append(p.tolist())
    p2m = run([fmt("p2m", p) for p in poses])
    fx["p2m"] = {"in": poses, "out": p2m}
    fx["m2p"] = {"in": p2m, "out": run([fmt("m2p", m) for m in p2m])}

    # --- Eigen fixed-size-3 association: Matrix3d*Vector3d, dot, the eigen-floor rebuild and the
    #     covariance expression exactly as NS/VoxelGrid.cpp writes them
    mv = [np.concatenate([rng.normal(size=9) * 10 ** rng.uniform(-3, 3), rng.normal(size=3)]).tolist() for _ in range(32)]
    fx["mv3"] = {"in": mv, "out": run([fmt("mv3", m) for m in mv])}
    dt = [(rng.normal(size=6) * 10 ** rng.uniform(-3, 3)).tolist() for _ in range(32)]
    fx["dot3"] = {"in": dt, "out": run([fmt("dot3", m) for m in dt])}
    vl = []
    for c, o in zip(covs[:48], fx["eig3"]["out"][:48]):
        ev = np.array(o[:3]); ec = o[3:]
        ev2 = ev.copy()
        mn = ev2[2] * 0.01
        ev2[0] = max(ev2[0], mn); ev2[1] = max(ev2[1], mn)
        vl.append(list(ec) + ev2.tolist())
    fx["vlv"] = {"in": vl, "out": run([fmt("vlv", m) for m in vl])}
    cx = []
    for k in range(48):
        n = int(rng.randint(6, 400))
        ctr = rng.uniform(-80, 80, 3)
        ctr[2] = rng.uniform(-2, 6)
        pts = (ctr + rng.uniform(-0.5, 0.5, (n, 3)) * np.array([1.0, 1.0, 0.02 if k % 2 else 1.0])).astype(np.float32).astype(np.float64)
        S = np.eye(3)
        s1 = np.zeros(3)
        for p in pts:   # same accumulation as scatterPointsToVoxelGrid (identity seed)
            s1 += p
            S += np.outer(p, p)
        cx.append(S.reshape(-1).tolist() + s1.tolist() + [float(n)])
    fx["covx"] = {"in": cx, "out": run([fmt("covx", m) for m in cx])}

    with open(os.path.join(HERE, "eigen_blocks.json"), "w") as f:
        json.dump(fx, f)
    print("wrote eigen_blocks.json:", {k: len(v["in"]) for k, v in fx.items()})


if __name__ == "__main__":
    main()
