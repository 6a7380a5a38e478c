"""JSON calibration files in the reference's layout (reference src/io.jl:5-32, test/runtests.jl:88-98).
Host-side only.  The one piece of JSON the reference itself holds is the intrinsic sub-object quoted at
src/io.jl:66 (kept as data in tests/golden/reference_io_intrinsic.json); the rest of the layout follows from
the `CalibrationIO` field list (src/io.jl:9-15) and the StructTypes rules the reference relies on (:5-7):
`Diagonal` -> its diagonal vector, every other AbstractArray (RotationVec is a 3x3 StaticMatrix) -> nested
arrays, first index innermost.  No Julia in this image: that last rule is stated, not run."""
import json
import os

import numpy as np
import pytest

import cameracalibrations_amd as cc
from cameracalibrations_amd import io as cio

HERE = os.path.dirname(os.path.abspath(__file__))

# field lists of the reference types: CalibrationIO (src/io.jl:9-15), AffineMap{linear, translation},
# LinearMap{linear} (CoordinateTransformations.jl, the types named at src/meta.jl:11-13)
CALIBRATION_IO_FIELDS = ["intrinsic", "extrinsics", "scale", "k", "files"]
AFFINE_FIELDS = ["linear", "translation"]
LINEAR_FIELDS = ["linear"]

R0 = cio.rotation_matrix([0.603, -0.097, -0.018])
R1 = cio.rotation_matrix([0.1, 0.2, 0.3])


def _layout(linear0, linear1):
    return ('{"intrinsic":{"linear":[401.44,401.44],"translation":[189.23,254.72]},'
            '"extrinsics":[{"linear":%s,"translation":[-1.9,-1.06,7.4]},'
            '{"linear":%s,"translation":[0.5,0.25,9.0]}],'
            '"scale":{"linear":[0.4,0.4,0.4]},"k":0.0496,"files":["1.png","my_extrinsic.png"]}'
            % (json.dumps(linear0), json.dumps(linear1)))


FORMS = {
    "nested 3x3, outer index = column (JSON3 ArrayType)": (R0.T.tolist(), R1.T.tolist()),
    "flat 9, column-major": (R0.T.ravel().tolist(), R1.T.ravel().tolist()),
    "rotation vector": ([0.603, -0.097, -0.018], [0.1, 0.2, 0.3]),
    "field form of this package's first version": ({"sx": 0.603, "sy": -0.097, "sz": -0.018}, {"sx": 0.1, "sy": 0.2, "sz": 0.3}),
}


@pytest.mark.parametrize("form", list(FORMS))
def test_load_reference_layout(tmp_path, form):
    f = tmp_path / "calibration.json"
    f.write_text(_layout(*FORMS[form]))
    c = cc.load(str(f))
    assert np.array_equal(c.f, [401.44, 401.44]) and np.array_equal(c.c, [189.23, 254.72])
    assert c.k == 0.0496 and c.checker_size == 2.5
    assert np.abs(c.rvecs[0] - [0.603, -0.097, -0.018]).max() < 1e-15 * 10
    assert np.abs(c.rvecs[1] - [0.1, 0.2, 0.3]).max() < 1e-15 * 10
    assert np.array_equal(c.tvecs[1], [0.5, 0.25, 9.0])
    assert c.files == ["1.png", "my_extrinsic.png"]
    assert c._index(None) == 1          # c(x) picks the first file matching r"extrinsic" (src/meta.jl:90-93)


def test_reference_held_intrinsic_text():
    """src/io.jl:66: the JSON3 text of an `AM` (AffineMap{SDiagonal{2}, SVector{2}}) -- the only on-disk
    bytes the reference holds.  The parser must accept it verbatim and the writer must emit the same
    key order, nesting and number text."""
    txt = open(os.path.join(HERE, "golden", "reference_io_intrinsic.json")).read().strip()
    f, c = cio.intrinsic_from_dict(json.loads(txt))
    assert f.tolist() == [0.8369119437634303, 0.5233244467702733]
    assert c.tolist() == [0.4655095816308332, 0.6672241801983796]
    cal = cc.Calibration(f, c, np.zeros((1, 3)), np.zeros((1, 3)), 1.0, 0.0, ["a"])
    assert json.dumps(cio.to_dict(cal)["intrinsic"], separators=(",", ":")) == txt


def test_emitted_layout_is_calibration_io():
    """key set, key order and nesting == the CalibrationIO field list (src/io.jl:9-15); Diagonal -> vector (:5-7)."""
    rng = np.random.default_rng(3)
    c = cc.Calibration(rng.random(2), rng.random(2), rng.random((3, 3)), rng.random((3, 3)), 2.5, 0.1, ["a", "b", "c"])
    d = json.loads(json.dumps(cio.to_dict(c)))
    assert list(d) == CALIBRATION_IO_FIELDS
    assert list(d["intrinsic"]) == AFFINE_FIELDS and list(d["scale"]) == LINEAR_FIELDS
    assert np.asarray(d["intrinsic"]["linear"]).shape == (2,)        # SDiagonal{2} lowered to its diagonal
    assert np.asarray(d["scale"]["linear"]).shape == (3,) and d["scale"]["linear"] == [0.4, 0.4, 0.4]
    assert len(d["extrinsics"]) == 3
    for e, r in zip(d["extrinsics"], c.rvecs):
        assert list(e) == AFFINE_FIELDS and np.asarray(e["translation"]).shape == (3,)
        M = np.asarray(e["linear"])
        assert M.shape == (3, 3)
        R = M.T                                                       # inner arrays are columns
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-14 and abs(np.linalg.det(R) - 1) < 1e-14
        assert np.abs(R - cio.rotation_matrix(r)).max() == 0.0
    assert isinstance(d["k"], float) and d["files"] == ["a", "b", "c"]


def test_rotation_vector_inverse():
    rng = np.random.default_rng(5)
    for _ in range(200):
        r = rng.normal(size=3)
        r *= rng.uniform(0, np.pi - 1e-3) / np.linalg.norm(r)
        assert np.abs(cio.rotation_vector(cio.rotation_matrix(r)) - r).max() < 1e-12
    for r in ([0, 0, 0], [1e-10, 0, 0], [np.pi, 0, 0], [0, -np.pi + 1e-9, 0], [2.0, 2.0, 1.0]):
        r = np.asarray(r, dtype=np.float64)
        if np.linalg.norm(r) > np.pi:
            r *= (np.pi - 1e-7) / np.linalg.norm(r)
        back = cio.rotation_vector(cio.rotation_matrix(r))
        assert np.abs(cio.rotation_matrix(back) - cio.rotation_matrix(r)).max() < 1e-9
    with pytest.raises(ValueError):
        cio.rotation_vector(np.diag([1.0, 1.0, -1.0]))


def test_rotation_vector_round_trip_over_every_angle():
    """theta from 1e-12 to pi with random axes: no hole between the small-angle and the near-pi branch (rotation vectors of
    5e-7 rad used to come back as their dominant axis component only)."""
    rng = np.random.default_rng(11)
    for th in np.r_[np.logspace(-12, 0, 97), np.linspace(1.0, np.pi, 60), np.pi - np.logspace(-12, -1, 45)]:
        for _ in range(4):
            n = rng.normal(size=3)
            r = n / np.linalg.norm(n) * th
            back = cio.rotation_vector(cio.rotation_matrix(r))
            if np.pi - th < 1e-6:        # at pi the axis sign is free: compare the rotations
                assert np.abs(cio.rotation_matrix(back) - cio.rotation_matrix(r)).max() < 1e-9, th
            else:
                assert np.abs(back - r).max() < 1e-9 * max(th, 1e-3) + 1e-16, (th, back, r)


def test_save_rotation_forms(tmp_path):
    rng = np.random.default_rng(3)
    c = cc.Calibration(rng.random(2), rng.random(2), rng.random((5, 3)), rng.random((5, 3)), 2.5, 0.1, list("abcde"))
    for form, shape in (("matrix", (3, 3)), ("vector", (3,))):
        f = tmp_path / f"c_{form}.json"
        cc.save(str(f), c, rotation=form)
        assert np.asarray(json.loads(f.read_text())["extrinsics"][0]["linear"]).shape == shape
        assert np.abs(cc.load(str(f)).rvecs - c.rvecs).max() < 1e-14
    with pytest.raises(ValueError):
        cio.to_dict(c, rotation="quaternion")


def test_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for _ in range(100):                 # test/runtests.jl:88-98: 100 random CalibrationIO round trips
        n = 5                            # rand(CalibrationIO) holds 5 extrinsics and 5 file names (src/io.jl:17)
        c = cc.Calibration(rng.random(2), rng.random(2), rng.random((n, 3)), rng.random((n, 3)), float(rng.random() + 0.1),
                           float(rng.random()), ["".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), 5)) for _ in range(n)])
        f = tmp_path / "calibration.json"
        cc.save(str(f), c)
        d = cc.load(str(f))
        assert d.files == c.files        # the reference compares only this field
        assert np.array_equal(d.f, c.f) and np.array_equal(d.c, c.c) and d.k == c.k
        assert np.abs(d.rvecs - c.rvecs).max() < 1e-14 and np.array_equal(d.tvecs, c.tvecs)
        assert abs(d.checker_size - c.checker_size) < 1e-15 * c.checker_size
        assert list(json.loads(f.read_text())) == CALIBRATION_IO_FIELDS


@pytest.mark.gpu
@pytest.mark.parametrize("rotation", ["vector", "matrix"])
def test_device_fit_survives_save_and_load(tmp_path, rotation):
    """Row (f3) end to end on the device: fit on the GPU -> `save` -> `load` -> the loaded calibration's device maps against the
    fitted one's (`src/io.jl:19-32`; the reference's own round trip, test/runtests.jl:88-98, compares `files` only).
    rotation="vector" stores the fitted rotation vectors themselves (JSON floats round-trip binary64 exactly): every map is
    BIT-equal.  rotation="matrix" stores R(r) and `load` takes the logarithm again: the vector comes back to a few ulp, the
    maps to 1e-10 px / 1e-12 board units."""
    golden = os.path.join(HERE, "golden", "example_corners.json")
    ex = json.load(open(golden))
    n_corners = tuple(ex["n_corners"])
    imgpointss = [m for m in np.array(ex["corners"], dtype=np.float64)]
    c, eps = cc.fit_corners(imgpointss, n_corners, 2.5, ex["image_size_wh"], files=ex["files"], rng=np.random.default_rng(3))
    f = tmp_path / "calibration.json"
    cc.save(str(f), c, rotation=rotation)
    d = cc.load(str(f))
    assert d.files == c.files and d.k == c.k and np.array_equal(d.f, c.f) and np.array_equal(d.c, c.c)
    assert np.array_equal(d.tvecs, c.tvecs) and d.checker_size == c.checker_size
    rng = np.random.default_rng(5)
    xyz = np.stack([rng.uniform(0, 10, 40), rng.uniform(0, 20, 40), np.zeros(40)], axis=1)
    rc = np.stack([rng.uniform(20, 350, 40), rng.uniform(20, 480, 40)], axis=1)
    for i in range(1, len(c.files) + 1):
        fwd_c, fwd_d = c([cc.XYZ(*p) for p in xyz], i), d([cc.XYZ(*p) for p in xyz], i)
        inv_c, inv_d = c([cc.RowCol(*p) for p in rc], i), d([cc.RowCol(*p) for p in rc], i)
        if rotation == "vector":
            assert np.array_equal(d.rvecs, c.rvecs)
            assert np.array_equal(np.asarray(fwd_c), np.asarray(fwd_d)) and np.array_equal(np.asarray(inv_c), np.asarray(inv_d))
        else:
            assert np.abs(d.rvecs - c.rvecs).max() < 1e-14
            assert np.abs(np.asarray(fwd_c) - np.asarray(fwd_d)).max() < 1e-10
            assert np.abs(np.asarray(inv_c) - np.asarray(inv_d)).max() < 1e-11
    # the metrics of the loaded calibration on the same corners are the fitted one's
    objp = cc.board_template(n_corners) * 2.5
    e2 = cc.calculate_errors(d, imgpointss, objp, 2.5, ex["image_size_wh"], ex["files"], n_corners, rng=np.random.default_rng(3))
    for k in ("reprojection", "projection", "distance", "inverse"):
        assert abs(e2[k] - eps[k]) < (1e-15 if rotation == "vector" else 1e-9), (k, e2[k], eps[k])
