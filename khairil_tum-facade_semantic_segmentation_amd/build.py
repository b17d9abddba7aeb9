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


def _compile_one(args):
    cmd, cwd = args
    subprocess.check_call(cmd, cwd=cwd)


def build(force=False, save_temps=False, verbose=False, jobs=None):
    """One object per source (rebuilt only when the source or a header is newer), compiled in parallel, then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(PKG, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc] + [f for f in HIPCC_FLAGS if f != "-shared"] + ["-c", "-I", os.path.join(REPO, "include"), "-I", CSRC]
    cwd = PKG
    if save_temps:
        cwd = os.path.join(PKG, "build", "temps")
        os.makedirs(cwd, exist_ok=True)
        base += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(REPO, "include", "*.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or save_temps or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            todo.append((base + [src, "-o", obj], cwd))
    if verbose:
        for cmd, _ in todo:
            print(" ".join(cmd))
    with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as pool:
        list(pool.map(_compile_one, todo))
    link = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-fvisibility=hidden"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=PKG)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True))
