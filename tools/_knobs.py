"""Imported by every tool BEFORE piccolo_amd: maps the PCL_* experiment variables of the sweep scripts onto the places that still listen.

The shipped library reads no environment variable (csrc/pcl_device.h PCL_KNOB; tests/test_abi.py).  The launch-planning and kernel-form
knobs (PCL_G, PCL_BLOCKS, PCL_XCD_RUNS, PCL_GD_FUSE_BLOCKS, PCL_ZFORM, ...) exist in the EXPERIMENTS build only: if one of them is set
and PCL_SO is not, this module builds lib/libpiccolo_hip_exp.so (seconds when up to date) and points PCL_SO at it.  The Python-level
switches (PCL_PANO_FMT, PCL_TRIM_FMT, PCL_GD_GRAPH, PCL_VERIFY_LEVELS) become attributes of piccolo_amd.ops.EXPERIMENT."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

C_KNOBS = ("PCL_TRIM_BAND_BYTES", "PCL_G", "PCL_BLOCKS", "PCL_XCD_RUNS", "PCL_XCD_GROUPS", "PCL_TRIM_CHUNKS", "PCL_TRIM_RUNS", "PCL_TRIM_XCD_IMAGES", "PCL_GD_FUSE_BLOCKS",
           "PCL_ZPINGPONG", "PCL_FLIP", "PCL_ZFORM", "PCL_ZSECOND", "PCL_BIN_EXACT", "PCL_BIN_DEDUP", "PCL_RESOLVE_THREADS")

if any(k in os.environ for k in C_KNOBS) and "PCL_SO" not in os.environ:
    from piccolo_amd import build as _build
    os.environ["PCL_SO"] = _build.build_experiments()
    print("tools/_knobs.py: %s set -> PCL_SO=%s" % (", ".join(k for k in C_KNOBS if k in os.environ), os.environ["PCL_SO"]), file=sys.stderr)

from piccolo_amd import ops as _ops  # noqa: E402

if os.environ.get("PCL_PANO_FMT") in ("f16", "f32"):
    _ops.EXPERIMENT.pano_fmt = os.environ["PCL_PANO_FMT"]
if os.environ.get("PCL_TRIM_FMT") in ("u8", "u8p", "u8v"):
    _ops.EXPERIMENT.trim_fmt = os.environ["PCL_TRIM_FMT"]
if os.environ.get("PCL_GD_GRAPH") in ("0", "1"):
    _ops.EXPERIMENT.gd_graph = os.environ["PCL_GD_GRAPH"] == "1"
if os.environ.get("PCL_VERIFY_LEVELS") == "1":
    _ops.EXPERIMENT.verify_levels = True
if "PCL_HIST_BATCH_BYTES" in os.environ:
    _ops.HIST_BATCH_BYTES = float(os.environ["PCL_HIST_BATCH_BYTES"])
