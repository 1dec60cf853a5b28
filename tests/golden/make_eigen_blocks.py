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

    # ---- round 4: the f32 Eigen calls of NS/NormalDistributionsTransform.cpp:336 and icp_registration_manual.cpp:156-176
    # (generated AFTER everything above so that the earlier blocks keep their inputs, draw for draw)
    def rot(ang):
        cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
        return (np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
                np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]))
    rots = []
    for k in range(96):
        R = rot(rng.uniform(-0.3, 0.3, 3) if k % 3 else rng.uniform(-3.1, 3.1, 3)).astype(np.float32)
        if k % 4 == 1:      # a pose that went through f32 products (predict = cur * step): orthonormal to a few ulp only
            R = (R @ rot(rng.uniform(-0.02, 0.02, 3)).astype(np.float32)).astype(np.float32)
        if k % 8 == 6:      # visibly non-orthonormal linear part: the polar step changes it
            R = (R + rng.normal(size=(3, 3)) * 1e-3).astype(np.float32)
        if k == 7:
            R = np.eye(3, dtype=np.float32)
        if k == 15:
            R = rng.normal(size=(3, 3)).astype(np.float32)
        if k == 23:
            R = np.zeros((3, 3), dtype=np.float32); R[0, 1] = R[1, 2] = R[2, 0] = 1.0   # permutation: sorting of equal singular values
        rots.append([float(v) for v in R.reshape(-1)])
    fx["rot3f"] = {"in": rots, "out": run([fmt("rot3f", m) for m in rots])}
    # ... and m2p on 4x4 matrices carrying those linear parts (the full :331-337 expression)
    m2p2 = []
    for m in rots:
        R = np.array(m, dtype=np.float32).reshape(3, 3)
        if not np.all(np.isfinite(R)) or abs(np.linalg.det(R.astype(np.float64))) < 0.5:
            continue
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R
        T[:3, 3] = rng.uniform(-60, 60, 3).astype(np.float32)
        m2p2.append([float(v) for v in T.T.reshape(-1)])
    fx["m2p_f32svd"] = {"in": m2p2, "out": run([fmt("m2p", m) for m in m2p2])}

    h6 = []
    for k in range(48):
        n = int(rng.randint(6, 4000))
        J = (rng.normal(size=(n, 6)) * np.array([1, 1, 1, 20, 20, 20])).astype(np.float32)
        H = (J.T @ J).astype(np.float32)
        if k % 8 == 3:      # general, not symmetric
            H = rng.normal(size=(6, 6)).astype(np.float32)
        if k % 8 == 5:      # exactly singular: a zero row and column (a degenerate correspondence set) -> determinant() == 0
            H[:, 4] = 0.0; H[4, :] = 0.0
        if k % 16 == 7:     # pivoting needed in the first column
            H = H[[2, 0, 1, 3, 5, 4], :]
        B = (rng.normal(size=6) * 50).astype(np.float32)
        h6.append([float(v) for v in H.reshape(-1)] + [float(v) for v in B])
    fx["det6f"] = {"in": [m[:36] for m in h6], "out": run([fmt("det6f", m[:36]) for m in h6])}
    reg = [m for m, d in zip(h6, fx["det6f"]["out"]) if d[0] != 0.0]     # the reference inverts only when det != 0 (:165-169)
    fx["inv6f"] = {"in": reg, "out": run([fmt("inv6f", m) for m in reg])}
    m3 = [[float(v) for v in np.concatenate([rot(rng.uniform(-3, 3, 3)).reshape(-1), rot(rng.uniform(-0.05, 0.05, 3)).reshape(-1)]).astype(np.float32)] for _ in range(32)]
    m3 += [[float(v) for v in rng.normal(size=18).astype(np.float32)] for _ in range(16)]
    fx["mul3f"] = {"in": m3, "out": run([fmt("mul3f", m) for m in m3])}
    acc = []
    for k in range(48):
        R = rot(rng.uniform(-0.5, 0.5, 3)).astype(np.float32)
        pt = (rng.normal(size=3) * np.array([30, 30, 2])).astype(np.float32)
        e = (rng.normal(size=3) * 0.3).astype(np.float32)
        acc.append([float(v) for v in np.concatenate([R.reshape(-1), pt, e])])
    fx["icpacc"] = {"in": acc, "out": run([fmt("icpacc", m) for m in acc])}

    with open(os.path.join(HERE, "eigen_blocks.json"), "w") as f:
        json.dump(fx, f)
    print("wrote eigen_blocks.json:", {k: len(v["in"]) for k, v in fx.items()})


if __name__ == "__main__":
    main()
