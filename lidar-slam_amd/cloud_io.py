"""Input formats for the harness: KITTI velodyne `.bin` and PCD (ASCII / binary), plus
pcl::removeNaNFromPointCloud, so recorded scans and the reference's saved key-frames / maps
(`pcl::io::savePCDFileBinary`, back_end.cpp:194, viewer.cpp:202,210; read back with `pcl::io::loadPCDFile` in
matching.cpp:155, loop_closing.cpp:134,286,304) can replace the synthetic generator when the data is there.

Everything returns / takes (n, 4) float32 {x, y, z, intensity} — the packed layout the device path consumes; use
`synth.to_xyzi32` for the 32-byte pcl::PointXYZI layout of the host boundary.
"""
import numpy as np


def read_kitti_bin(path, max_points=None):
    """KITTI raw / odometry velodyne scan: flat little-endian float32 records x, y, z, reflectance."""
    a = np.fromfile(path, dtype="<f4")
    if a.size % 4:
        raise ValueError("%s: %d floats is not a multiple of 4" % (path, a.size))
    a = a.reshape(-1, 4)
    if max_points is not None:
        a = a[:max_points]
    return np.ascontiguousarray(a, dtype=np.float32)


def write_kitti_bin(path, xyzi):
    np.ascontiguousarray(xyzi, dtype="<f4").reshape(-1, 4).tofile(path)


def remove_nan(xyzi):
    """pcl::removeNaNFromPointCloud (front_end.cpp:92, matching.cpp:188): drop points whose x, y or z is not finite,
    order kept.  Returns (cloud, kept_indices).  The device kernels skip non-finite points themselves, so this is only
    needed where a point COUNT must match the reference's (e.g. trans_probability = score / #points)."""
    xyzi = np.asarray(xyzi)
    keep = np.isfinite(xyzi[:, :3]).all(axis=1)
    return np.ascontiguousarray(xyzi[keep]), np.nonzero(keep)[0]


_PCD_TYPES = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("U", 8): "<u8",
              ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4", ("I", 8): "<i8"}


def read_pcd(path):
    """PCD v0.7 reader for DATA ascii / binary (what pcl::io::savePCDFileBinary / ASCII of a PointXYZI cloud write:
    FIELDS x y z intensity, SIZE 4 4 4 4, TYPE F F F F, COUNT 1 1 1 1).  Extra fields are ignored, a `_` padding field
    (binary files of the 32-byte PointXYZI carry none, but other writers do) is skipped, a missing intensity reads 0.
    binary_compressed (LZF) is not produced anywhere in the reference and is rejected."""
    with open(path, "rb") as f:
        raw = f.read()
    header = {}
    pos = 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        header[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    fields = header["FIELDS"]
    sizes = [int(v) for v in header["SIZE"]]
    types = header["TYPE"]
    counts = [int(v) for v in header.get("COUNT", ["1"] * len(fields))]
    npts = int(header["POINTS"][0]) if "POINTS" in header else int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
    kind = header["DATA"][0].lower()
    out = np.zeros((npts, 4), dtype=np.float32)
    want = {"x": 0, "y": 1, "z": 2, "intensity": 3}
    if kind == "ascii":
        txt = raw[pos:].decode("ascii", "replace").split()
        ncol = sum(counts)
        vals = np.array(txt[: npts * ncol], dtype=np.float64).reshape(npts, ncol) if npts else np.zeros((0, ncol))
        col = 0
        for name, cnt in zip(fields, counts):
            if name in want:
                out[:, want[name]] = vals[:, col]
            col += cnt
    elif kind == "binary":
        dt = []
        for k, (name, sz, ty, cnt) in enumerate(zip(fields, sizes, types, counts)):
            base = _PCD_TYPES[(ty.upper(), sz)]
            dt.append(("f%d_%s" % (k, name), base, (cnt,)) if cnt != 1 else ("f%d_%s" % (k, name), base))
        rec = np.frombuffer(raw, dtype=np.dtype(dt), count=npts, offset=pos)
        for k, name in enumerate(fields):
            if name in want:
                out[:, want[name]] = rec["f%d_%s" % (k, name)]
    else:
        raise ValueError("%s: PCD DATA %s is not supported (ascii / binary only)" % (path, kind))
    return out


def write_pcd(path, xyzi, binary=True):
    """pcl::io::savePCDFileBinary / savePCDFileASCII of a pcl::PointCloud<pcl::PointXYZI> (unorganised, height 1)."""
    a = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = a.shape[0]
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
           "COUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % (n, n, "binary" if binary else "ascii"))
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        if binary:
            f.write(a.astype("<f4").tobytes())
        else:
            for p in a:
                f.write((" ".join("nan" if np.isnan(v) else repr(float(v)) for v in p) + "\n").encode("ascii"))
