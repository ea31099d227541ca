#!/usr/bin/env python3
"""Entry point with the reference's CLI (main.py:12-16): --config, --log, --override.

dataset = Synthetic runs piccolo_amd.localize.localize_synthetic (no dataset files needed).  For the reference's real
datasets (Stanford2D-3D-S, OmniScenes) run the reference's own main.py on top of this package — dropin/run_reference.py,
see INTEGRATION.md: its localize.py then calls piccolo_amd's omniloc / utils unchanged.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from piccolo_amd.parse_utils import apply_override, parse_ini

    parser = argparse.ArgumentParser()
    parser.add_argument("--config", default=None, type=str, help="Config file to use for running experiments")
    parser.add_argument("--log", default="./log", type=str, help="Log directory for logging accuracy")
    parser.add_argument("--override", default=None, help="Arguments for overriding config")
    args = parser.parse_args()
    cfg = parse_ini(args.config)
    if args.override is not None:
        cfg = apply_override(cfg, args.override)
    os.makedirs(args.log, exist_ok=True)

    import configparser
    out = configparser.ConfigParser()
    out.add_section("Default")
    for key, val in cfg._asdict().items():
        out["Default"][key] = str(val) if key == "name" else str(val).replace("[", "").replace("]", "")
    with open(os.path.join(args.log, "config.ini"), "w") as f:
        out.write(f)

    if cfg.dataset == "Synthetic":
        import numpy as np
        import torch.distributed as dist
        if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import torch
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl")
        from piccolo_amd.localize import localize_synthetic
        table = localize_synthetic(cfg, None, args.log).cpu().numpy()
        if not dist.is_initialized() or dist.get_rank() == 0:
            print("images %d  median t-err %.4f m  median R-err %.3f deg  mean time %.3f s" % (
                len(table), np.median(table[:, 13]), np.median(table[:, 14]), table[:, 15].mean()))
        if dist.is_initialized():
            dist.destroy_process_group()
    elif cfg.dataset in ("Stanford2D-3D-S", "OmniScenes"):
        raise SystemExit("dataset %r needs the reference's dataset harness: run it on top of piccolo_amd with "
                         "`python dropin/run_reference.py /path/to/piccolo --config ... --log ...` (INTEGRATION.md)" % cfg.dataset)
    else:
        raise ValueError(cfg.dataset)


if __name__ == "__main__":
    main()
