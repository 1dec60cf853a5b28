"""Deterministic synthetic HDL-64E-like scans (KITTI data is not in the
reference tree nor in this image, so every BASELINE.json config is restated on these generators).

A scan is ray-cast from a sensor pose into a closed-form street scene (ground plane, building
boxes with seeded setbacks on both sides, cross walls, parked-car boxes, pole cylinders, and
small clutter boxes standing in for vegetation / street furniture) and returned in the SENSOR
frame as (n, 4) float32 [x, y, z, intensity(w)=1].

Rays are ordered azimuth-major (one column of `beams` shots per azimuth step), so every obstacle
only has to be tested against the contiguous slice of columns its azimuth interval covers.
"""
import numpy as np

SENSOR_HEIGHT = 1.73


class Scene:
    def __init__(self, seed: int = 1, length: float = 260.0, clutter: int = 260):
        rng = np.random.RandomState(seed)
        boxes = []
        # buildings: both sides of the street, varying setback / length / height, small gaps
        for side in (-1.0, 1.0):
            x = -60.0
            while x < length:
                ln = rng.uniform(8.0, 22.0)
                # left side: a street canyon 12-16.5 m from the lane; right side: an open lot, 26-44 m setback
                setback = 12.0 + rng.choice([0.0, 0.0, 1.5, 3.0, 4.5]) if side < 0 else rng.uniform(26.0, 44.0)
                h = rng.uniform(5.0, 11.0)
                y0, y1 = (setback, setback + 15.0) if side > 0 else (-setback - 15.0, -setback)
                boxes.append([x, x + ln, y0, y1, -SENSOR_HEIGHT, -SENSOR_HEIGHT + h])
                x += ln + rng.choice([0.0, 0.0, 2.0, 5.0])
        # cross walls closing the street
        boxes.append([-62.0, -60.0, -30.0, 60.0, -SENSOR_HEIGHT, 8.0])
        boxes.append([length, length + 2.0, -30.0, 60.0, -SENSOR_HEIGHT, 8.0])
        # parked cars 4 x 1.8 x 1.5
        n_cars = int(12 * (length + 60.0) / 100.0)
        for _ in range(n_cars):
            cx = rng.uniform(-50.0, length - 10.0)
            cy = -rng.uniform(5.0, 9.5) if rng.rand() < 0.4 else rng.uniform(5.0, 24.0)
            boxes.append([cx - 2.0, cx + 2.0, cy - 0.9, cy + 0.9, -SENSOR_HEIGHT, -SENSOR_HEIGHT + 1.5])
        # clutter: bushes, bins, signs, tree crowns (small boxes, some elevated)
        n_cl = int(clutter * (length + 60.0) / 100.0)
        for _ in range(n_cl):
            cx = rng.uniform(-55.0, length - 5.0)
            cy = -rng.uniform(4.0, 11.8) if rng.rand() < 0.4 else rng.uniform(4.0, 25.0)
            sx, sy, sz = rng.uniform(0.3, 1.6), rng.uniform(0.3, 1.6), rng.uniform(0.4, 2.2)
            z0 = -SENSOR_HEIGHT + (rng.uniform(2.0, 4.5) if rng.rand() < 0.35 else 0.0)
            boxes.append([cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2, z0, z0 + sz])
        self.boxes = np.asarray(boxes, dtype=np.float64)
        n_poles = int(8 * (length + 60.0) / 100.0)
        px = rng.uniform(-50.0, length - 10.0, n_poles)
        py = rng.choice([-1.0, 1.0], n_poles) * rng.uniform(10.0, 11.5, n_poles)
        self.poles = np.stack([px, py], axis=1)
        self.pole_r = 0.3
        self.pole_top = -SENSOR_HEIGHT + 6.0

    # ------------------------------------------------------------------------------------------
    def ground(self, o, d_w, max_range):
        """range to the ground along every ray (n_az, beams); inf where it is not hit: the plane z = -SENSOR_HEIGHT"""
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (-SENSOR_HEIGHT - o[2]) / d_w[:, :, 2]
        tg[~(tg > 0)] = np.inf
        return tg

    def cast(self, T_world_sensor, az, elev, max_range, rng=None):
        """Ray-cast beams x azimuth columns. Returns t (n_az, beams) ranges (inf where nothing is hit)
        and the unit directions in the sensor frame (n_az, beams, 3).  rng: only scenes with foliage volumes draw from it."""
        R = T_world_sensor[:3, :3]
        o = T_world_sensor[:3, 3]
        ce, se = np.cos(elev), np.sin(elev)
        d_s = np.stack([np.outer(np.cos(az), ce), np.outer(np.sin(az), ce), np.outer(np.ones_like(az), se)], axis=2)
        d_w = d_s @ R.T
        n_az = az.shape[0]
        t = np.full((n_az, elev.shape[0]), np.inf)
        t = np.minimum(t, self.ground(o, d_w, max_range))
        daz = 2.0 * np.pi / n_az
        Rt = R.T

        def col_slices(a_lo, a_hi):
            """columns whose azimuth lies in [a_lo, a_hi] (radians, may wrap)"""
            if a_hi - a_lo >= 2.0 * np.pi - 1e-9:
                return [slice(0, n_az)]
            i0 = int(np.floor((a_lo - az[0]) / daz)) - 1
            i1 = int(np.ceil((a_hi - az[0]) / daz)) + 1
            i0m, i1m = i0 % n_az, i1 % n_az
            if i1 - i0 >= n_az:
                return [slice(0, n_az)]
            if i0m <= i1m:
                return [slice(i0m, i1m + 1)]
            return [slice(i0m, n_az), slice(0, i1m + 1)]

        foliage = getattr(self, "foliage", None)
        n_solid = self.boxes.shape[0]
        all_boxes = self.boxes if foliage is None else np.concatenate([self.boxes, foliage[:, :6]], axis=0)
        for bi, b in enumerate(all_boxes):
            # azimuth interval of the box as seen from the sensor (sensor frame), from its 8 corners
            cs = np.array([[b[i], b[2 + j], b[4 + k]] for i in (0, 1) for j in (0, 1) for k in (0, 1)])
            cs_s = (cs - o) @ Rt.T
            ctr = cs_s.mean(axis=0)
            inside = (b[0] <= o[0] <= b[1]) and (b[2] <= o[1] <= b[3])
            if inside:
                sls = [slice(0, n_az)]
            else:
                a_c = np.arctan2(ctr[1], ctr[0])
                da = np.arctan2(cs_s[:, 1], cs_s[:, 0]) - a_c
                da = (da + np.pi) % (2.0 * np.pi) - np.pi
                if da.max() - da.min() > np.pi:
                    sls = [slice(0, n_az)]
                else:
                    lo, hi = a_c + da.min(), a_c + da.max()
                    # bring into the az array's range [az0, az0 + 2pi)
                    shift = np.floor((lo - az[0]) / (2.0 * np.pi)) * 2.0 * np.pi
                    sls = col_slices(lo - shift, hi - shift)
            for sl in sls:
                d = d_w[sl]
                with np.errstate(divide="ignore", invalid="ignore"):
                    inv = 1.0 / d
                    t0 = (b[0::2] - o) * inv
                    t1 = (b[1::2] - o) * inv
                tmin = np.nanmax(np.minimum(t0, t1), axis=2)
                tmax = np.nanmin(np.maximum(t0, t1), axis=2)
                hit = (tmax >= tmin) & (tmax > 0)
                if bi >= n_solid:
                    # foliage volume: a ray travelling through it returns from a depth drawn from an exponential free path
                    # (density rho per metre), or passes through — a volumetric point distribution instead of a facet
                    rho = foliage[bi - n_solid, 6]
                    depth = rng.exponential(1.0 / rho, size=tmin.shape)
                    t_in = np.maximum(tmin, 0.0)
                    tb = t_in + depth
                    hit &= tb < tmax
                else:
                    tb = np.where(tmin > 0, tmin, tmax)
                tb[~hit] = np.inf
                t[sl] = np.minimum(t[sl], tb)
        pole_r_all = getattr(self, "pole_radii", None)
        pole_top_all = getattr(self, "pole_tops", None)
        for pi_, c in enumerate(self.poles):
            pole_r = self.pole_r if pole_r_all is None else pole_r_all[pi_]
            pole_top = self.pole_top if pole_top_all is None else pole_top_all[pi_]
            oc = o[:2] - c
            dist = np.hypot(oc[0], oc[1])
            if dist <= pole_r * 1.5:
                sls = [slice(0, n_az)]
            else:
                c_s = (np.array([c[0], c[1], o[2]]) - o) @ Rt.T
                a_c = np.arctan2(c_s[1], c_s[0])
                half = np.arcsin(min(1.0, pole_r / dist)) + 0.02
                lo, hi = a_c - half, a_c + half
                shift = np.floor((lo - az[0]) / (2.0 * np.pi)) * 2.0 * np.pi
                sls = col_slices(lo - shift, hi - shift)
            for sl in sls:
                d = d_w[sl]
                dxy = d[:, :, :2]
                a = (dxy ** 2).sum(2)
                bq = 2.0 * (dxy @ oc)
                cq = oc @ oc - pole_r ** 2
                disc = bq * bq - 4.0 * a * cq
                ok = disc > 0
                with np.errstate(divide="ignore", invalid="ignore"):
                    tp = (-bq - np.sqrt(np.where(ok, disc, 0.0))) / (2.0 * a)
                z = o[2] + tp * d[:, :, 2]
                pole_base = getattr(self, "pole_base", None)
                ok &= (tp > 0) & (z <= pole_top) & (z >= -SENSOR_HEIGHT + (0.0 if pole_base is None else pole_base[pi_] - 0.5))
                tp[~ok] = np.inf
                t[sl] = np.minimum(t[sl], tp)
        t[t > max_range] = np.inf
        return t, d_s


class DenseScene(Scene):
    """A KITTI-residential-shaped bracket of the headline scene: detached houses set back 14-35 m
    behind front gardens on both sides, a second row of buildings behind the gaps, a row of low-crowned trees on either
    sidewalk (trunk + a crown that returns volumetrically), hedges along and across the property lines, bushes, fences and
    parked cars.  An HDL-64E sees nothing above ~2 m + 3.5 % of the range, so what adds occupied voxels is LOW relief spread
    over open ground, not tall facades: more occupied 1 m voxels, fatter voxel distributions and more neighbours per point
    than the bare street canyon of `Scene`."""

    def __init__(self, seed: int = 1, length: float = 260.0, hedge_prob: float = 0.75, lot_divider_prob: float = 1.0, parked_cars: float = 16.0):
        rng = np.random.RandomState(seed + 77)
        boxes, foliage, poles, pole_r, pole_top = [], [], [], [], []
        g = -SENSOR_HEIGHT

        def bush(cx, cy, r, h, rho=1.2):
            foliage.append([cx - r, cx + r, cy - r, cy + r, g, g + h, rho])

        for side in (-1.0, 1.0):
            x = -60.0
            while x < length:
                ln = rng.uniform(8.0, 18.0)          # house frontage
                lot = ln + rng.uniform(4.0, 10.0)    # lot width incl. the gap to the next house
                setback = rng.uniform(14.0, 22.0) if side < 0 else rng.uniform(18.0, 35.0)
                h = rng.uniform(5.0, 10.0)
                depth = rng.uniform(8.0, 12.0)
                y0, y1 = (setback, setback + depth) if side > 0 else (-setback - depth, -setback)
                boxes.append([x, x + ln, y0, y1, g, g + h])
                # second row behind the gap
                if rng.rand() < 0.6:
                    sb2 = setback + depth + rng.uniform(10.0, 22.0)
                    yy = (sb2, sb2 + 10.0) if side > 0 else (-sb2 - 10.0, -sb2)
                    boxes.append([x + ln - 2.0, x + lot + 4.0, yy[0], yy[1], g, g + rng.uniform(5.0, 9.0)])
                # porch / bay window
                for _ in range(rng.randint(0, 3)):
                    bx = rng.uniform(x, x + ln - 1.5)
                    bw, bd, bz = rng.uniform(1.2, 3.0), rng.uniform(0.5, 1.5), rng.uniform(0.0, 1.0)
                    yy = (setback - bd, setback) if side > 0 else (-setback, -setback + bd)
                    boxes.append([bx, bx + bw, yy[0], yy[1], g + bz, g + bz + rng.uniform(1.5, 3.0)])
                # hedge along the street side of the lot, hedges / fences between lots (across the street direction)
                front = rng.uniform(7.0, 9.0)
                if rng.rand() < hedge_prob:
                    hy = side * front
                    gap0 = rng.uniform(x + 1.0, x + lot - 4.0)    # driveway opening
                    for (a, b) in ((x, gap0), (gap0 + 3.0, x + lot)):
                        if b - a > 1.0:
                            foliage.append([a, b, hy - 0.45, hy + 0.45, g, g + rng.uniform(0.9, 1.7), 1.8])
                ya, yb = sorted((side * front, side * (setback + depth)))
                divider = lot_divider_prob >= 1.0 or rng.rand() < lot_divider_prob
                if rng.rand() < 0.5:
                    if divider:
                        foliage.append([x + lot - 0.4, x + lot + 0.4, ya, yb, g, g + rng.uniform(1.0, 2.0), 1.8])
                    else:
                        rng.uniform(1.0, 2.0)
                else:
                    if divider:
                        boxes.append([x + lot - 0.05, x + lot + 0.05, ya, yb, g, g + rng.uniform(0.9, 1.5)])
                    else:
                        rng.uniform(0.9, 1.5)
                # front garden: bushes, a small tree or two, a car in the driveway
                for _ in range(rng.randint(3, 9)):
                    bush(rng.uniform(x, x + lot), side * rng.uniform(front + 1.0, setback - 0.5), rng.uniform(0.5, 1.5), rng.uniform(0.8, 2.4))
                for _ in range(rng.randint(0, 3)):
                    tx, ty = rng.uniform(x, x + lot), side * rng.uniform(front + 2.0, setback + depth + 8.0)
                    poles.append([tx, ty]); pole_r.append(rng.uniform(0.1, 0.25))
                    trunk = rng.uniform(1.2, 2.2)
                    pole_top.append(g + trunk + 0.3)
                    cr = rng.uniform(1.5, 3.0)
                    foliage.append([tx - cr, tx + cr, ty - cr, ty + cr, g + trunk, g + trunk + rng.uniform(2.5, 5.0), 0.7])
                if rng.rand() < 0.5:
                    cx, cy = rng.uniform(x + 2.0, x + lot - 2.0), side * rng.uniform(front + 2.0, setback - 2.5)
                    boxes.append([cx - 0.9, cx + 0.9, cy - 2.0, cy + 2.0, g, g + 1.5])
                x += lot
            # sidewalk trees with low crowns
            x = -55.0 + rng.uniform(0.0, 5.0)
            while x < length - 5.0:
                ty = side * rng.uniform(5.2, 6.4)
                poles.append([x, ty]); pole_r.append(rng.uniform(0.12, 0.3))
                trunk = rng.uniform(1.7, 2.4)
                pole_top.append(g + trunk + 0.3)
                cr = rng.uniform(1.5, 2.6)
                foliage.append([x - cr, x + cr, ty - cr, ty + cr, g + trunk, g + trunk + rng.uniform(3.0, 5.0), 0.6])
                x += rng.uniform(7.0, 13.0)
        boxes.append([-62.0, -60.0, -70.0, 70.0, g, 8.0])
        boxes.append([length, length + 2.0, -70.0, 70.0, g, 8.0])
        n_cars = int(parked_cars * (length + 60.0) / 100.0)
        for _ in range(n_cars):
            cx = rng.uniform(-50.0, length - 10.0)
            cy = rng.choice([-1.0, 1.0]) * rng.uniform(3.4, 4.4)
            boxes.append([cx - 2.0, cx + 2.0, cy - 0.9, cy + 0.9, g, g + 1.5])
        self.boxes = np.asarray(boxes, dtype=np.float64)
        self.foliage = np.asarray(foliage, dtype=np.float64)   # [x0, x1, y0, y1, z0, z1, density per metre]
        self.poles = np.asarray(poles, dtype=np.float64)
        self.pole_radii = np.asarray(pole_r)
        self.pole_tops = np.asarray(pole_top)
        self.pole_r = 0.3
        self.pole_top = g + 6.0


class TerrainScene(DenseScene):
    """A KITTI-residential-SHAPED scene: the residential street of `DenseScene` on rolling terrain instead
    of a plane — a flat carriageway with kerbs, lawns and verges that rise, dip and undulate by a metre or two, banks and
    ditches — with every house, hedge, tree and car standing on the terrain under it, more and larger volumetric vegetation,
    and fewer, lower occluders close to the road so that the sensor sees a few hundred metres of relief.  What this buys over
    the flat scenes is what SURVEY.md 8 sizes the path with: ~20 k points after the 0.5 m frame filter (relief spreads the
    far ground rings over distinct voxels), >= 20 k valid 1 m voxels in a 20-key-frame local map, and 4+ neighbour voxels
    per source point (a slope crosses voxel layers; foliage fills voxels above the ground ones)."""

    ROAD_HALF = 3.6
    ROUGH = 0.10     # metres, peak

    def __init__(self, seed: int = 1, length: float = 320.0):
        super().__init__(seed, length, hedge_prob=0.25, lot_divider_prob=0.35, parked_cars=8.0)
        rng = np.random.RandomState(seed + 991)
        g = -SENSOR_HEIGHT
        # more vegetation: copses and shrub belts out to +-90 m, single trees on the verges
        fol, poles, pole_r, pole_top = list(self.foliage), list(self.poles), list(self.pole_radii), list(self.pole_tops)
        # apartment blocks / halls behind the gardens: tall enough (12-22 m) to be seen over the first row by the upper laser
        # block, 45-110 m from the road, facades of 25-70 m: the distant vertical structure a real residential scan is full of
        blocks = []
        for side in (-1.0, 1.0):
            x = -50.0 + rng.uniform(0.0, 20.0)
            while x < length - 20.0:
                ln, dep = rng.uniform(25.0, 70.0), rng.uniform(12.0, 20.0)
                y0 = side * rng.uniform(45.0, 95.0)
                blocks.append([x, min(x + ln, length - 2.0), min(y0, y0 + side * dep), max(y0, y0 + side * dep), g, g + rng.uniform(12.0, 22.0)])
                x += ln + rng.uniform(8.0, 30.0)
        self.boxes = np.concatenate([self.boxes, np.asarray(blocks, dtype=np.float64)], axis=0)
        # tree belts (field boundaries, the edge of a wood) roughly parallel to the road, 4-8 m deep and 4-9 m high, broken by gaps
        for side in (-1.0, 1.0):
            for yb in (rng.uniform(100.0, 115.0),):
                x = -58.0 + rng.uniform(0.0, 10.0)
                while x < length - 6.0:
                    ln = rng.uniform(10.0, 30.0)
                    y0 = side * (yb + rng.uniform(-3.0, 3.0))
                    dep = rng.uniform(4.0, 8.0)
                    fol.append([x, min(x + ln, length - 2.0), min(y0, y0 + side * dep), max(y0, y0 + side * dep), g + rng.uniform(0.0, 1.0), g + rng.uniform(4.0, 9.0), rng.uniform(0.35, 0.7)])
                    x += ln + rng.uniform(10.0, 30.0)
        n_copse = int(90 * (length + 60.0) / 100.0)
        for _ in range(n_copse):
            cx, cy = rng.uniform(-58.0, length - 2.0), rng.choice([-1.0, 1.0]) * rng.uniform(8.0, 95.0)
            for _k in range(rng.randint(2, 7)):
                tx, ty = cx + rng.normal(0.0, 3.5), cy + rng.normal(0.0, 3.5)
                if abs(ty) < 5.0:
                    continue
                if rng.rand() < 0.55:   # a tree: trunk + crown
                    trunk = rng.uniform(1.0, 2.5)
                    poles.append([tx, ty]); pole_r.append(rng.uniform(0.1, 0.3)); pole_top.append(g + trunk + 0.3)
                    cr = rng.uniform(1.5, 3.5)
                    fol.append([tx - cr, tx + cr, ty - cr, ty + cr, g + trunk, g + trunk + rng.uniform(2.5, 6.0), rng.uniform(0.5, 0.9)])
                else:                   # a shrub
                    r = rng.uniform(0.6, 2.0)
                    fol.append([tx - r, tx + r, ty - r, ty + r, g, g + rng.uniform(0.8, 2.6), rng.uniform(1.0, 1.8)])
        self.foliage = np.asarray(fol, dtype=np.float64)
        self.poles = np.asarray(poles, dtype=np.float64)
        self.pole_radii, self.pole_tops = np.asarray(pole_r), np.asarray(pole_top)
        # everything stands on the terrain under its footprint's centre (boxes reach 0.8 m into the ground so that no gap
        # opens on the downhill side)
        hb = self.height(0.5 * (self.boxes[:, 0] + self.boxes[:, 1]), 0.5 * (self.boxes[:, 2] + self.boxes[:, 3]))
        self.boxes[:, 4] += hb - 0.8
        self.boxes[:, 5] += hb
        hf = self.height(0.5 * (self.foliage[:, 0] + self.foliage[:, 1]), 0.5 * (self.foliage[:, 2] + self.foliage[:, 3]))
        self.foliage[:, 4] += hf
        self.foliage[:, 5] += hf
        self.pole_base = self.height(self.poles[:, 0], self.poles[:, 1])
        self.pole_tops = self.pole_tops + self.pole_base

    def height(self, x, y):
        """terrain height above the carriageway level (m): 0 on the road, a 12 cm kerb, then rolling ground"""
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        ay = np.abs(y)
        s = np.clip((ay - self.ROAD_HALF - 0.25) / 6.0, 0.0, 1.0)
        blend = s * s * (3.0 - 2.0 * s)                      # smoothstep from the kerb to 6 m behind it
        roll = (0.9 * np.sin(x / 23.0 + 1.0) * np.cos(y / 17.0) + 0.55 * np.sin((x + 2.0 * y) / 9.0) + 0.35 * np.sin(x / 4.1) * np.sin(y / 5.3)
                + 0.028 * (ay - self.ROAD_HALF) * np.sign(np.sin(x / 61.0 + 0.4)))   # banks that rise / fall away from the road
        kerb = 0.12 * np.clip((ay - self.ROAD_HALF) / 0.25, 0.0, 1.0)
        # grass / soil: decimetre-scale roughness of a few centimetres off the carriageway (what keeps real ground voxels from
        # being the millimetre-thin sheets an analytic surface gives)
        rough = self.ROUGH * (0.55 * np.sin(x * 9.1 + 1.3) * np.sin(y * 7.7 + 0.2) + 0.3 * np.sin(x * 23.3 + y * 3.1) * np.cos(y * 19.1 - x * 2.7)
                              + 0.15 * np.sin(x * 41.0 + 0.7) * np.sin(y * 37.0 + 2.1))
        return kerb + blend * (roll + rough)

    def ground(self, o, d_w, max_range):
        """first crossing of every ray with the height field: march (0.4 m steps near, 2.5 m far), then bisect the bracket"""
        g = -SENSOR_HEIGHT
        lim = 150.0 if not np.isfinite(max_range) else float(max_range)
        ts = np.concatenate([np.arange(0.4, 12.0, 0.4), np.arange(12.0, 40.0, 1.0), np.arange(40.0, lim + 2.5, 2.5)])
        n_az, nb = d_w.shape[0], d_w.shape[1]
        out = np.full((n_az, nb), np.inf)
        step = 256
        for c0 in range(0, n_az, step):
            d = d_w[c0:c0 + step].reshape(-1, 3)
            px = o[0] + np.outer(d[:, 0], ts)
            py = o[1] + np.outer(d[:, 1], ts)
            f = (o[2] + np.outer(d[:, 2], ts)) - (g + self.height(px, py))    # > 0 above the ground
            below = f <= 0.0
            hit = below.any(axis=1)
            k = np.argmax(below, axis=1)                                         # first sample at / below the ground
            t1 = ts[k]
            t0 = np.where(k > 0, ts[np.maximum(k - 1, 0)], 0.0)
            dh, t0h, t1h = d[hit], t0[hit], t1[hit]
            for _ in range(14):
                tm = 0.5 * (t0h + t1h)
                fm = (o[2] + dh[:, 2] * tm) - (g + self.height(o[0] + dh[:, 0] * tm, o[1] + dh[:, 1] * tm))
                up = fm > 0.0
                t0h = np.where(up, tm, t0h)
                t1h = np.where(up, t1h, tm)
            res = np.full(d.shape[0], np.inf)
            res[hit] = 0.5 * (t0h + t1h)
            out[c0:c0 + step] = res.reshape(-1, nb)
        return out


def pose_matrix(x=0.0, y=0.0, z=0.0, roll=0.0, pitch=0.0, yaw=0.0):
    """world<-sensor 4x4 (float64), R = Rx(roll) Ry(pitch) Rz(yaw) like the reference's pose chart."""
    cx, sx, cy, sy, cz, sz = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rx @ Ry @ Rz
    T[:3, 3] = [x, y, z]
    return T


def hdl64e_elevations():
    """The HDL-64E's two laser blocks: 32 beams from +2.0 deg down in 1/3 deg steps (to -8.33), 32 from -8.83 deg down in
    1/2 deg steps (to -24.33) — denser towards the horizon than the linear spacing of SURVEY.md 8d's generator, which puts
    three quarters of the returns inside 10 m."""
    return np.deg2rad(np.concatenate([2.0 - np.arange(32) / 3.0, -8.83 - 0.5 * np.arange(32)])[::-1])


def gen_scan(scene: Scene, T_world_sensor, seed: int, beams: int = 64, az_steps: int = 2083,
             max_range: float = 80.0, noise: float = 0.02, clamp: bool = False, layout: str = "linear"):
    """One spinning-lidar scan. beams x az_steps rays, elevation -24.8..+2.0 deg linear (layout "linear") or
    the HDL-64E's two blocks (layout "hdl64e", beams = 64).
    clamp=False: drop returns beyond max_range (HDL-64E config); clamp=True: keep every ray,
    clamped to max_range (cfg5, exactly beams*az_steps points)."""
    rng = np.random.RandomState(seed)
    elev = hdl64e_elevations() if layout == "hdl64e" else np.deg2rad(np.linspace(-24.8, 2.0, beams))
    az = np.arange(az_steps) * (2.0 * np.pi / az_steps)
    t, d_s = scene.cast(np.asarray(T_world_sensor, dtype=np.float64), az, elev, np.inf if clamp else max_range, rng=rng)
    t = t.reshape(-1)
    d_s = d_s.reshape(-1, 3)
    if clamp:
        t = np.minimum(t, max_range)
        keep = np.ones(t.shape, dtype=bool)
    else:
        keep = np.isfinite(t)
    t = t[keep] + rng.normal(0.0, noise, int(keep.sum()))
    pts = d_s[keep] * t[:, None]
    out = np.empty((pts.shape[0], 4), dtype=np.float32)
    out[:, :3] = pts.astype(np.float32)
    out[:, 3] = 1.0
    return out


def trajectory(n: int = 30, step: float = 1.0, yaw_rate: float = 0.004):
    """cfg2: n poses, `step` m apart along +x with a gentle yaw."""
    poses = []
    x = y = yaw = 0.0
    for i in range(n):
        poses.append(pose_matrix(x, y, 0.0, 0.002 * np.sin(0.3 * i), -0.003 * np.cos(0.2 * i), yaw))
        yaw += yaw_rate
        x += step * np.cos(yaw)
        y += step * np.sin(yaw)
    return poses


def to_xyzi32(xyzw):
    """(n,4) float32 -> (n,8) float32 in pcl::PointXYZI layout (32 B per point)."""
    xyzw = np.asarray(xyzw, dtype=np.float32)
    out = np.zeros((xyzw.shape[0], 8), dtype=np.float32)
    out[:, :3] = xyzw[:, :3]
    out[:, 3] = 1.0
    out[:, 4] = xyzw[:, 3] if xyzw.shape[1] > 3 else 1.0
    return out


def transform_points(T, xyzw):
    """pcl::transformPointCloud on the xyz columns (f32), other columns copied."""
    out = np.array(xyzw, dtype=np.float32, copy=True)
    T32 = np.asarray(T, dtype=np.float32)
    out[:, :3] = xyzw[:, :3] @ T32[:3, :3].T + T32[:3, 3]
    return out
