#!/usr/bin/env python3
"""Run the REFERENCE's own main.py with its omniloc / utils / parse_utils / color_utils / data_utils replaced by piccolo_amd.

    python dropin/run_reference.py /path/to/piccolo --config configs/stanford_parallel.ini --log logs/run1

A script's own directory always comes first on sys.path, so the reference's modules would shadow the drop-ins if its
main.py were started directly; this launcher puts dropin/ and the repo root in front and then executes main.py from
inside the reference checkout (its relative ./data paths keep working).  Everything else the reference imports
(localize.py, cv2, tensorboard) is its own.
"""
import os
import runpy
import sys

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    ref = os.path.abspath(sys.argv[1])
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here)]
    sys.path.append(ref)
    sys.argv = [os.path.join(ref, "main.py")] + sys.argv[2:]
    os.chdir(ref)
    runpy.run_path(sys.argv[0], run_name="__main__")
