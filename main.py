#!/usr/bin/env python3
"""Entry point with the reference's CLI (main.py:12-16): --config, --log, --override.

dataset = Synthetic runs piccolo_amd.localize.localize_synthetic (no dataset files needed); Stanford2D-3D-S and OmniScenes
run piccolo_amd.localize.localize_stanford / localize_omniscenes over ./data/... in the reference's directory layout
(README.md:40-75 of the reference) and write the reference's CSV.  The reference's own main.py can also be run on top of
this package — dropin/run_reference.py, see INTEGRATION.md.  With torch.distributed.run the query images are sharded
over the GPUs:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -- main.py --config ... --log ...
(the `--` matters: torchrun's argument parser otherwise rejects the reference CLI's `--log` as an ambiguous abbreviation of
its own --log-dir / --logs-specs).
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

    if cfg.dataset not in ("Synthetic", "Stanford2D-3D-S", "OmniScenes"):
        raise ValueError(cfg.dataset)
    import numpy as np
    multi_rank = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if multi_rank:
        # dmabuf IPC (bench.py): the HSA runtime reads this when the process first touches the GPU, so it is set
        # before torch is imported and before anything counts or selects a device
        if "PCL_HSA_IPC_MODE_LEGACY" in os.environ:
            os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ["PCL_HSA_IPC_MODE_LEGACY"]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    if multi_rank:
        import torch
        # One process per GPU over RCCL.  PCL_DIST_BACKEND=gloo (the knob bench.py has) lets several ranks share the GPUs that
        # are there — how the test-suite runs this very loop with two ranks on a one-GPU box.
        backend = os.environ.get("PCL_DIST_BACKEND", "nccl")
        local_rank, n_dev = int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count()
        if n_dev < 1:
            raise SystemExit("main.py needs an MI355X: torch.cuda.device_count() == 0")
        if backend == "nccl" and n_dev <= local_rank:
            raise SystemExit("local rank %d but only %d GPU(s) visible: one rank per GPU is required for the RCCL run" % (local_rank, n_dev))
        import datetime
        dev_index = local_rank if backend == "nccl" else local_rank % n_dev
        torch.cuda.set_device(dev_index)
        # an explicit, short timeout: a rank that cannot reach the others ends with a reason instead of hanging for torch's 10-30 minutes
        tmo = datetime.timedelta(seconds=float(os.environ.get("PCL_DIST_TIMEOUT_S", "180")))
        print("main.py rank %s/%s: device cuda:%d of %d visible, backend %s, HSA_ENABLE_IPC_MODE_LEGACY=%s" % (
            os.environ["RANK"], os.environ["WORLD_SIZE"], dev_index, n_dev, backend, os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]), file=sys.stderr, flush=True)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=tmo)
            else:
                dist.init_process_group(backend, timeout=tmo)
        except Exception as exc:                                      # noqa: BLE001
            raise SystemExit("main.py rank %s: init_process_group(%s) FAILED: %s: %s" % (os.environ["RANK"], backend, type(exc).__name__, exc))
    from piccolo_amd import localize
    run = {"Synthetic": localize.localize_synthetic, "Stanford2D-3D-S": localize.localize_stanford,
           "OmniScenes": localize.localize_omniscenes}[cfg.dataset]
    table = run(cfg, None, args.log).cpu().numpy()
    if not dist.is_initialized() or dist.get_rank() == 0:
        done = table[~np.isnan(table[:, 13])]
        print("images %d (%d skipped)  median t-err %.4f m  median R-err %.3f deg  mean time %.3f s" % (
            len(table), len(table) - len(done), np.median(done[:, 13]) if len(done) else float("nan"),
            np.median(done[:, 14]) if len(done) else float("nan"), done[:, 15].mean() if len(done) else float("nan")))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
