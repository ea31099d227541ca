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


EXP_SO = os.path.join(OUT_DIR, "libpiccolo_hip_exp.so")     # the EXPERIMENTS build (-DPCL_EXPERIMENTS): tools/ and the XCD-mapping test load it via PCL_SO
HASH_FILE = "pcl_pack.hip"                                    # the one translation unit that carries the two source hashes


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "piccolo_hip.h")]


def stale(so=None):
    so = so or SO
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False, extra_flags=(), experiments=False):
    """One object per .hip file, compiled in parallel (8 jobs: the full library in ~25 s instead of 75), linked into the shared library.
    experiments=True: libpiccolo_hip_exp.so with -DPCL_EXPERIMENTS — the only build that reads the PCL_* knobs of its process
    (csrc/pcl_device.h PCL_KNOB); the shipped libpiccolo_hip.so reads none."""
    from concurrent.futures import ThreadPoolExecutor
    extra_flags = list(extra_flags) + os.environ.get("PCL_HIPCC_FLAGS", "").split()     # (A/B builds of tools/, with PCL_SO)
    so = EXP_SO if experiments else SO
    if not force and not stale(so):
        return so
    obj_dir = os.path.join(OUT_DIR, "obj_exp" if experiments else "obj")
    os.makedirs(obj_dir, exist_ok=True)
    base = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-pthread", "-Wall", "-Wno-unused-function"]
    base += (["-DPCL_EXPERIMENTS"] if experiments else []) + extra_flags
    hashes = ['-DPCL_SOURCE_HASH="%s"' % loss_kernel_source_hash(), '-DPCL_LIBRARY_HASH="%s"' % library_source_hash()]
    newest_header = max(os.path.getmtime(d) for d in _deps() if not d.endswith(".hip"))

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        stamped = os.path.basename(src) == HASH_FILE                 # (its -D values change with any source: always recompiled)
        if not force and not stamped and not extra_flags and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        cmd = base + (hashes if stamped else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, sources()))
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc", "-pthread", "-o", so] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return so


def build_experiments(force=False, verbose=False):
    return build(force=force, verbose=verbose, experiments=True)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
