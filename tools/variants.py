"""Builds variants of libccal_hip.so with extra -D flags into build_variants/<name>.so (git-ignored; they travel to the GPU box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import build as B  # noqa: E402


def build_variant(name, defines):
    out_dir = os.path.join(ROOT, "build_variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".so")
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
    # an entry starting with '@' is passed to the compiler as it is (e.g. @-mllvm,@-amdgpu-kernarg-preload-count=16)
    extra = [d[1:] if d.startswith("@") else "-D" + d for d in defines]
    cmd = ["/opt/rocm/bin/hipcc", *B.FLAGS, "-shared", *extra, "-o", out, *srcs, "-ldl"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    # usage: variants.py name=DEF1,DEF2 name2=DEF3 ...
    for spec in sys.argv[1:]:
        name, _, defs = spec.partition("=")
        print(build_variant(name, [d for d in defs.split(",") if d]))
