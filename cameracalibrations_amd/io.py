"""JSON calibration files in the reference's on-disk layout (reference src/io.jl:5-32).

`CalibrationIO` (src/io.jl:9-15) as JSON3 + StructTypes write it:
    {"intrinsic": {"linear": [frow, fcol], "translation": [crow, ccol]},        # Diagonal lowered to its diag (:5-7);
                                                                               #   the text kept at src/io.jl:66 is this object
     "extrinsics": [{"linear": [[R11,R21,R31],[R12,R22,R32],[R13,R23,R33]],    # RotationVec is a 3x3 StaticMatrix, and
                     "translation": [tx, ty, tz]}, ...],                       #   StructTypes lowers every AbstractArray as
                                                                               #   ArrayType: nested arrays, first index innermost
     "scale": {"linear": [1/cs, 1/cs, 1/cs]},                                   # LinearMap(SDiagonal(I / checker_size))
     "k": k, "files": [...]}
Only the `Diagonal` case is special-cased by the reference (:5-7); the rotation therefore travels as its
matrix, not as the three rotation-vector fields.  No Julia runs in this image, so that reading of
JSON3 / StructTypes is unverified; `load` accepts every form the rotation could take
(nested 3x3 as above, flat 9 column-major, a 3-vector, or the {"sx","sy","sz"} field form this package
wrote in its first version) and `save` writes the nested matrix.
`load` rebuilds the maps exactly as the constructor does (src/io.jl:19-26): only the numbers
travel, the closures are re-derived.
"""
import json

import numpy as np

from .calibration import Calibration


def rotation_matrix(rvec):
    """Rodrigues: rotation vector -> 3x3 matrix (what Rotations.RotationVec is as an AbstractMatrix)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    K = np.array([[0.0, -r[2], r[1]], [r[2], 0.0, -r[0]], [-r[1], r[0], 0.0]])
    if th < 1e-8:
        a, b = 1.0 - th * th / 6.0, 0.5 - th * th / 24.0
    else:
        a, b = np.sin(th) / th, (1.0 - np.cos(th)) / (th * th)
    return np.eye(3) + a * K + b * (K @ K)


def rotation_vector(R):
    """Inverse Rodrigues: 3x3 rotation matrix -> rotation vector (angle in [0, pi])."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    if not np.allclose(R @ R.T, np.eye(3), atol=1e-6) or np.linalg.det(R) < 0:
        raise ValueError("extrinsics[].linear is not a rotation matrix")
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])     # 2 sin(th) n
    s = 0.5 * float(np.linalg.norm(w))
    c = 0.5 * (float(np.trace(R)) - 1.0)
    th = float(np.arctan2(s, c))
    if c >= 0.0 or s > 1e-6:
        # th <= pi / 2, or the skew part still carries the axis: w = 2 sin(th) n, so r = th n = w th / (2 s);
        # th / (2 s) -> 1 / 2 as th -> 0 (no hole between a "small" and a "near pi" branch: near pi needs c < 0)
        return w * (th / (2.0 * s)) if s > 1e-150 else 0.5 * w
    # th ~ pi: the skew part vanishes; the symmetric part is c I + (1 - c) n n^T exactly, whatever is left of the skew part
    B = (0.5 * (R + R.T) - c * np.eye(3)) / (1.0 - c)
    i = int(np.argmax(np.diag(B)))
    n = B[:, i] / np.sqrt(B[i, i])
    if w @ n < 0:
        n = -n
    return n * th


def _rvec_from_json(lin):
    if isinstance(lin, dict):                                   # field form {"sx","sy","sz"}
        return np.array([lin["sx"], lin["sy"], lin["sz"]], dtype=np.float64)
    a = np.asarray(lin, dtype=np.float64)
    if a.shape == (3,):
        return a
    if a.shape == (3, 3):                                       # JSON3: outer index = column
        return rotation_vector(a.T)
    if a.shape == (9,):                                         # column-major flat
        return rotation_vector(a.reshape(3, 3).T)
    raise ValueError(f"extrinsics[].linear: unsupported layout {a.shape}")


def to_dict(c: Calibration, rotation="matrix"):
    """rotation = "matrix": extrinsics[].linear as the nested 3x3 (outer index = column) that StructTypes' ArrayType
    gives an AbstractMatrix -- PROVISIONAL: read off the JSON3 / StructTypes rules, never checked against a file written by
    the reference (it holds none); "vector": the three rotation-vector components, which `load` here
    also accepts and a Julia-side `RotationVec(v...)` constructor call can take."""
    if rotation not in ("matrix", "vector"):
        raise ValueError("rotation must be 'matrix' or 'vector'")
    s = 1.0 / float(c.checker_size)

    def lin(r):
        if rotation == "vector":
            return [float(v) for v in np.asarray(r, dtype=np.float64).reshape(3)]
        return [[float(v) for v in col] for col in rotation_matrix(r).T]
    return {
        "intrinsic": {"linear": [float(c.f[0]), float(c.f[1])], "translation": [float(c.c[0]), float(c.c[1])]},
        "extrinsics": [{"linear": lin(r), "translation": [float(t[0]), float(t[1]), float(t[2])]}
                       for r, t in zip(c.rvecs, c.tvecs)],
        "scale": {"linear": [s, s, s]},
        "k": float(c.k),
        "files": list(c.files),
    }


def intrinsic_from_dict(d):
    """The `AM` sub-object (src/io.jl:10; the JSON text at :66): -> (f, c)."""
    return np.array(d["linear"], dtype=np.float64).reshape(2), np.array(d["translation"], dtype=np.float64).reshape(2)


def from_dict(d, device=0):
    scale = d["scale"]["linear"]
    if not (scale[0] == scale[1] == scale[2]) or scale[0] == 0:
        raise ValueError("scale must be a uniform non-zero diagonal (reference src/buildcalibrations.jl:4)")
    rv = np.array([_rvec_from_json(e["linear"]) for e in d["extrinsics"]], dtype=np.float64)
    tv = np.array([e["translation"] for e in d["extrinsics"]], dtype=np.float64)
    f, c = intrinsic_from_dict(d["intrinsic"])
    return Calibration(f, c, rv.reshape(-1, 3), tv.reshape(-1, 3), 1.0 / float(scale[0]), float(d["k"]),
                       list(d["files"]), device)


def save(file, c: Calibration, rotation="matrix"):
    """CameraCalibrations.save(file, c) (src/io.jl:28-32); `rotation`: see to_dict (the matrix form is provisional)."""
    with open(file, "w") as io:
        json.dump(to_dict(c, rotation), io)


def load(file, device=0):
    """CameraCalibrations.load(file) (src/io.jl:19-26)."""
    with open(file) as io:
        return from_dict(json.load(io), device)
