"""JSON calibration files in the reference's on-disk layout (reference src/io.jl:5-32).

`CalibrationIO` as JSON3 + StructTypes write it:
    {"intrinsic": {"linear": [frow, fcol], "translation": [crow, ccol]},        # Diagonal lowered to its diag (:5-7)
     "extrinsics": [{"linear": {"sx": .., "sy": .., "sz": ..}, "translation": [tx, ty, tz]}, ...],   # RotationVec fields
     "scale": {"linear": [1/cs, 1/cs, 1/cs]},                                   # LinearMap(SDiagonal(I / checker_size))
     "k": k, "files": [...]}
`load` rebuilds the maps exactly as the constructor does (src/io.jl:19-26): only the numbers
travel, the closures are re-derived.
"""
import json

import numpy as np

from .calibration import Calibration


def to_dict(c: Calibration):
    s = 1.0 / float(c.checker_size)
    return {
        "intrinsic": {"linear": [float(c.f[0]), float(c.f[1])], "translation": [float(c.c[0]), float(c.c[1])]},
        "extrinsics": [{"linear": {"sx": float(r[0]), "sy": float(r[1]), "sz": float(r[2])},
                        "translation": [float(t[0]), float(t[1]), float(t[2])]}
                       for r, t in zip(c.rvecs, c.tvecs)],
        "scale": {"linear": [s, s, s]},
        "k": float(c.k),
        "files": list(c.files),
    }


def from_dict(d, device=0):
    scale = d["scale"]["linear"]
    if not (scale[0] == scale[1] == scale[2]) or scale[0] == 0:
        raise ValueError("scale must be a uniform non-zero diagonal (reference src/buildcalibrations.jl:4)")
    rv = np.array([[e["linear"]["sx"], e["linear"]["sy"], e["linear"]["sz"]] for e in d["extrinsics"]], dtype=np.float64)
    tv = np.array([e["translation"] for e in d["extrinsics"]], dtype=np.float64)
    return Calibration(np.array(d["intrinsic"]["linear"], dtype=np.float64),
                       np.array(d["intrinsic"]["translation"], dtype=np.float64), rv.reshape(-1, 3), tv.reshape(-1, 3),
                       1.0 / float(scale[0]), float(d["k"]), list(d["files"]), device)


def save(file, c: Calibration):
    """CameraCalibrations.save(file, c) (src/io.jl:28-32)."""
    with open(file, "w") as io:
        json.dump(to_dict(c), io)


def load(file, device=0):
    """CameraCalibrations.load(file) (src/io.jl:19-26)."""
    with open(file) as io:
        return from_dict(json.load(io), device)
