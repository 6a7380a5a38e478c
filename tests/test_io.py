"""JSON calibration files in the reference's layout (reference src/io.jl:5-32, test/runtests.jl:88-98).
Host-side only; the fixture string is what JSON3 writes for a CalibrationIO (field names of
AffineMap / LinearMap / RotationVec, Diagonal lowered to its diagonal)."""
import json

import numpy as np

import cameracalibrations_amd as cc

REFERENCE_LAYOUT = (
    '{"intrinsic":{"linear":[401.44,401.44],"translation":[189.23,254.72]},'
    '"extrinsics":[{"linear":{"sx":0.603,"sy":-0.097,"sz":-0.018},"translation":[-1.9,-1.06,7.4]},'
    '{"linear":{"sx":0.1,"sy":0.2,"sz":0.3},"translation":[0.5,0.25,9.0]}],'
    '"scale":{"linear":[0.4,0.4,0.4]},"k":0.0496,"files":["1.png","my_extrinsic.png"]}'
)


def test_load_reference_layout(tmp_path):
    f = tmp_path / "calibration.json"
    f.write_text(REFERENCE_LAYOUT)
    c = cc.load(str(f))
    assert np.array_equal(c.f, [401.44, 401.44]) and np.array_equal(c.c, [189.23, 254.72])
    assert c.k == 0.0496 and c.checker_size == 2.5
    assert np.array_equal(c.rvecs[0], [0.603, -0.097, -0.018]) and np.array_equal(c.tvecs[1], [0.5, 0.25, 9.0])
    assert c.files == ["1.png", "my_extrinsic.png"]
    assert c._index(None) == 1          # c(x) picks the first file matching r"extrinsic" (src/meta.jl:90-93)


def test_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for _ in range(20):                  # test/runtests.jl:88-98 does 100 random round trips
        n = int(rng.integers(1, 6))
        c = cc.Calibration(rng.random(2), rng.random(2), rng.random((n, 3)), rng.random((n, 3)), float(rng.random() + 0.1),
                           float(rng.random()), ["".join(rng.choice(list("abcdefgh"), 5)) for _ in range(n)])
        f = tmp_path / "c.json"
        cc.save(str(f), c)
        d = cc.load(str(f))
        assert d.files == c.files        # the reference compares only this field
        assert np.array_equal(d.f, c.f) and np.array_equal(d.c, c.c) and d.k == c.k
        assert np.array_equal(d.rvecs, c.rvecs) and np.array_equal(d.tvecs, c.tvecs)
        assert abs(d.checker_size - c.checker_size) < 1e-15 * c.checker_size
        assert set(json.loads(f.read_text())) == {"intrinsic", "extrinsics", "scale", "k", "files"}
