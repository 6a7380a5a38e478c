"""Builds variants of libccal_hip.so with extra -D flags into build_variants/<name>.so (git-ignored; they travel to the GPU box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import build as B  # noqa: E402


def build_variant(name, defines):
    out_dir = os.path.join(ROOT, "build_variants")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".so")
    # an entry starting with '@' is passed to the compiler as it is (e.g. @-mllvm,@-amdgpu-kernarg-preload-count=16)
    extra = [d[1:] if d.startswith("@") else "-D" + d for d in defines]
    objs = []
    for src in B.SOURCES:                         # file by file, with the product build's per-file flags (build.EXTRA_FLAGS)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *extra, "-c", os.path.join(B.CSRC, src), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"])
    return out


if __name__ == "__main__":
    # usage: variants.py name=DEF1,DEF2 name2=DEF3 ...
    for spec in sys.argv[1:]:
        name, _, defs = spec.partition("=")
        print(build_variant(name, [d for d in defs.split(",") if d]))
