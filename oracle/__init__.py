"""CPU oracle for the camera-model fit (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (cameracalibrations_amd/) never does.  See ccal_oracle.h for what
is restated, from where, and the pinning status ("parity unpinned" beyond the
reference's own `< 1` bar on test/example/corners.json).
"""
from .oracle import *  # noqa: F401,F403
