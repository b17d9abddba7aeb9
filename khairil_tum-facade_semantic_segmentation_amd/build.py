"""Builds libpn2hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python khairil_tum-facade_semantic_segmentation_amd/build.py [--force] [--save-temps]
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libpn2hip.so")
ARCH = "gfx950"

# -ffp-contract=off: FPS / square_distance parity is bit-exact only if no a*b+c is fused behind
# our back; explicit fmaf() calls are where the reference's BLAS fuses.
HIPCC_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH, "-ffp-contract=off",
               "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(REPO, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, save_temps=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", os.path.join(REPO, "include"), "-I", CSRC]
    if save_temps:
        tmp = os.path.join(PKG, "build", "temps")
        os.makedirs(tmp, exist_ok=True)
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    cmd += sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=os.path.join(PKG, "build", "temps") if save_temps else PKG)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True))
