"""Builds piccolo_amd/lib/libpiccolo_hip.so from piccolo_amd/csrc/*.hip with hipcc for gfx950.

In-tree on purpose: the .so is git-ignored but travels with the working tree (gpurun snapshot), and the product
refuses to run without it (piccolo_amd/_lib.py) — there is no CPU fallback.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
SO = os.path.join(OUT_DIR, "libpiccolo_hip.so")
ARCH = "gfx950"


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def loss_kernel_source_hash():
    """First 16 hex digits of sha256(csrc/pcl_loss.hip + csrc/pcl_sample_device.h + csrc/pcl_gd_device.h + csrc/pcl_device.h): what pcl_source_hash() of a library built from this
    tree returns.  Counter-derived figures in profiles/roofs.json carry it; bench.py reports them only when it matches."""
    import hashlib
    h = hashlib.sha256()
    for name in ("pcl_loss.hip", "pcl_sample_device.h", "pcl_gd_device.h", "pcl_device.h"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def library_source_hash():
    """First 16 hex digits of sha256 over EVERY source of the library (csrc/*.hip, csrc/*.h, include/piccolo_hip.h, sorted by name): what
    pcl_library_hash() returns.  The per-kernel roofs of the pipeline kernels (profiles/pipeline_roofs.json) carry it."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "piccolo_hip.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "piccolo_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    extra_flags = list(extra_flags) + os.environ.get("PCL_HIPCC_FLAGS", "").split()     # experiments only
    if not force and not stale():
        return SO
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-pthread",
           "-Wall", "-Wno-unused-function", '-DPCL_SOURCE_HASH="%s"' % loss_kernel_source_hash(), '-DPCL_LIBRARY_HASH="%s"' % library_source_hash(),
           "-o", SO] + list(extra_flags) + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
