"""CPU tests of bench.py's command line: the self-launch of N > 1 ranks.  No GPU, no compute."""
import os
import subprocess
import sys

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


def test_self_launch_decides_from_gpus_and_world_size(monkeypatch):
    """An ordinary rank (N == 1, or WORLD_SIZE already set by torch.distributed.run) is never a launcher."""
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.self_launch([]) is None
    assert bench.self_launch(["--gpus", "1", "--steps", "3"]) is None
    assert bench.self_launch(["--gpus=1"]) is None
    assert bench.self_launch(["--gpus", "x"]) is None          # left to argparse
    assert bench.self_launch(["--gpus", "4", "--help"]) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.self_launch(["--gpus", "4"]) is None           # started by torch.distributed.run: this process IS a rank


def test_launcher_never_imports_torch_or_the_hip_library():
    """The launcher branch runs before `import torch` / the ctypes load of libpiccolo_hip.so: a process that starts other
    programs must not have initialised the GPU.  Checked on the source: everything above the self_launch() call is stdlib."""
    import ast
    src = open(BENCH).read()
    tree = ast.parse(src)
    call_line = next(n.lineno for n in ast.walk(tree) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "self_launch")
    early = set()
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)) and n.lineno < call_line:
            early |= {a.name.split(".")[0] for a in n.names} if isinstance(n, ast.Import) else {(n.module or "").split(".")[0]}
    assert early <= {"argparse", "json", "os", "sys", "time", "socket", "subprocess"}, early
    # nothing in the file replaces the running program
    assert not [n for n in ast.walk(tree) if isinstance(n, ast.Attribute) and n.attr.startswith(("exec", "spawn", "posix_spawn")) and n.attr != "executable"]


def test_launcher_without_a_gpu_fails_loudly_and_prints_no_line():
    """`python bench.py --gpus 2` in the build container: two ranks are started (gloo or not, there is no GPU), each says so,
    the launcher relays their stderr, exits non-zero and writes no JSON line."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(PCL_DIST_BACKEND="gloo"),
                         cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "launching 2 ranks" in out.stderr and "needs an MI355X" in out.stderr
    assert out.stdout.strip() == ""


def test_launcher_relays_rank0_line_and_status(tmp_path):
    """The relay itself, with a stand-in for bench.py's rank body: a script that defines the same self_launch() (imported from
    bench.py's source) and, as a rank, prints a banner and then — rank 0 only — one JSON line.  The launcher's stdout must be
    that line alone; with a failing rank the launcher exits non-zero."""
    src = open(BENCH).read()
    head = src[:src.index("import numpy as np")]               # the stdlib-only part: imports, REPO, self_launch, the launcher branch
    body = '''
import json
rank = int(os.environ["RANK"])
print("banner from rank %d" % rank, flush=True)
if os.environ.get("FAIL_RANK") == str(rank):
    sys.exit(3)
if rank == 0:
    print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "argv": sys.argv[1:]}), flush=True)
'''
    script = tmp_path / "fake_bench.py"
    script.write_text(head + body)
    out = subprocess.run([sys.executable, str(script), "--gpus", "2", "--steps", "5"], env=_clean_env(), cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 2, "argv": ["--gpus", "2", "--steps", "5"]}
    assert "banner from rank 0" in out.stderr and "banner from rank 1" in out.stderr
    bad = subprocess.run([sys.executable, str(script), "--gpus", "2"], env=_clean_env(FAIL_RANK="1"), cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline", "checks", "ranks_seen", "per_rank_ms_per_step", "single_image")


def _check_compact(c, full):
    import json
    text = json.dumps(c)
    assert len(text) < 4096, len(text)
    assert not [k for k in REQUIRED if k not in c], [k for k in REQUIRED if k not in c]
    assert "also" not in c and c["config"]["workload"] and "model" not in c["config"]
    r = c["roofline"]
    assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_hbm_frac", "hbm_measured_frac"}
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert set(c["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    for k in ("value", "ms_per_step"):
        assert abs(c[k] - full[k]) <= 1e-5 * abs(full[k])
    assert c["n_gpus"] == full["n_gpus"] and c["steps"] == full["steps"] and c["warmup"] == full["warmup"]
    assert not [v for v in c.values() if isinstance(v, str) and len(v) > 200]


def test_compact_line_of_recorded_full_lines_is_below_4k_with_every_key():
    """the driver could not parse round 5's 24.5 KB line.  The LAST stdout line is now compact_line(record): on
    the complete records of round 5 — the driver's N = 1 command and the self-launched 8-rank gloo run — it stays below 4 KB and
    carries the headline, config, single_image, roofline, cpu_baseline and checks; the rest lives in the side file."""
    import json
    import bench
    for name in ("bench_driver_command_line.json", "bench_gpus8_self_launched_gloo_one_gpu.json"):
        full = json.load(open(os.path.join(REPO, "profiles", "r05", name)))
        assert len(json.dumps(full)) > 4096                     # (what the driver lost, or nearly)
        c = bench.compact_line(full, os.path.join(REPO, "bench_also.json"))
        _check_compact(c, full)
        assert c["also_file"] == "bench_also.json" and c["also_brief"]
    assert c["n_gpus"] == 8 and c["ranks_seen"] == 8 and c["n1_value_same_build"]["value"] > 0 and c["dist_backend"] == "gloo"


def test_compact_line_sheds_optional_blocks_instead_of_growing():
    """The size is enforced, not hoped for: absurdly long strings and a huge `also` block still give a line below the limit with the
    headline, the roofline and the CPU baseline intact; non-finite floats become null (valid JSON for any parser)."""
    import json
    import bench
    full = json.load(open(os.path.join(REPO, "profiles", "r05", "bench_driver_command_line.json")))
    full["config"]["workload"] = "w" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["also"] = {"side_%d" % i: {"value": float(i)} for i in range(400)}
    full["also"]["error"] = "e" * 5000
    full["roofline"]["valu"]["frac"] = float("nan")
    full["median_t_err_m"] = float("inf")
    c = bench.compact_line(full, None)
    text = json.dumps(c)
    assert len(text) < bench.COMPACT_LIMIT and "NaN" not in text and "Infinity" not in text
    assert c["value"] == bench._r(full["value"]) and c["roofline"]["frac"] and c["cpu_baseline"]["value"] and "also_brief" not in c
    assert c["median_t_err_m"] is None


def test_launcher_retries_once_with_the_other_ipc_mode(tmp_path):
    """if a collective (or the process group's set-up) fails on the first launch, the launcher starts fresh ranks ONCE
    more with the other HSA_ENABLE_IPC_MODE_LEGACY and relays that launch's line; the line says where the mode came from.  A value
    forced with PCL_HSA_IPC_MODE_LEGACY is never second-guessed.  Stand-in rank body as above."""
    import json
    src = open(BENCH).read()
    head = src[:src.index("import numpy as np")]
    body = '''
import json
rank = int(os.environ["RANK"])
mode = os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]
if mode == os.environ["FAIL_IF_IPC"]:
    print("bench.py rank %d/2: init_process_group(nccl) FAILED: RuntimeError: hipIpcGetMemHandle: invalid argument" % rank, file=sys.stderr, flush=True)
    os._exit(3)
if rank == 0:
    print(json.dumps({"ipc_mode_legacy": mode, "ipc_mode_from": os.environ["PCL_IPC_MODE_FROM"]}), flush=True)
'''
    script = tmp_path / "fake_bench.py"
    script.write_text(head + body)
    env = _clean_env(FAIL_IF_IPC="0")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    out = subprocess.run([sys.executable, str(script), "--gpus", "2"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip()) == {"ipc_mode_legacy": "1", "ipc_mode_from": "retry"}
    assert out.stderr.count("launching 2 ranks") == 2 and "launching once more with HSA_ENABLE_IPC_MODE_LEGACY=1" in out.stderr
    # the mode of the environment works: one launch, and the line says "environment"
    ok = subprocess.run([sys.executable, str(script), "--gpus", "2"], env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="1"), cwd=str(tmp_path),
                        capture_output=True, text=True, timeout=600)
    assert ok.returncode == 0 and json.loads(ok.stdout.strip()) == {"ipc_mode_legacy": "1", "ipc_mode_from": "environment"}
    assert ok.stderr.count("launching 2 ranks") == 1
    # both modes fail: two launches, non-zero, no line
    bad = subprocess.run([sys.executable, str(script), "--gpus", "2"], env=dict(env, FAIL_IF_IPC="0", PCL_HSA_IPC_MODE_LEGACY="0"), cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and bad.stdout.strip() == "" and bad.stderr.count("launching 2 ranks") == 1      # forced mode: no retry


def test_roofline_mix_ceiling_is_reproducible_from_the_committed_inputs(tmp_path):
    """`roofline.frac_of_mix_ceiling` / `fp32_flop_frac` must be reproducible by a committed script from committed
    inputs.  tools/roof_mix.py recompiles csrc/pcl_loss.hip to ISA (hipcc -S: no GPU), prices the loop's VALU opcodes with the class costs
    of profiles/r06/valu_rate.txt and must give profiles/r06/roof_mix.json again; the entries of profiles/roofs.json carry those figures
    for the loss-kernel sources they were measured from, and bench.valu_roof turns them into the line's three numbers."""
    import json
    import shutil
    import bench
    from piccolo_amd import build
    committed = json.load(open(os.path.join(REPO, "profiles", "r06", "roof_mix.json")))
    roofs_copy = tmp_path / "roofs.json"
    shutil.copy(os.path.join(REPO, "profiles", "roofs.json"), roofs_copy)
    out = tmp_path / "roof_mix.json"
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "roof_mix.py"), "--roofs", str(roofs_copy), "--out", str(out)], cwd=REPO)
    again = json.load(open(out))
    assert set(again["instances"]) == {"f16", "u8", "f32"}
    for k, inst in again["instances"].items():
        assert inst["valu_instructions_in_loop"] > 500 and 4.0 < inst["mix_ceiling_cycles_per_instr"] < 5.0 and 1.5 < inst["fp32_flops_per_instr"] < 2.0
        assert not [o for o in inst["priced_by_class_default"] if not o.startswith("v_mov")]          # every other opcode has a measured class
    if committed["loss_kernel_source_hash"] == build.loss_kernel_source_hash():                        # (same kernel sources: the same numbers)
        for k in committed["instances"]:
            assert abs(again["instances"][k]["mix_ceiling_cycles_per_instr"] - committed["instances"][k]["mix_ceiling_cycles_per_instr"]) < 1e-9
            assert again["instances"][k]["valu_instructions_in_loop"] == committed["instances"][k]["valu_instructions_in_loop"]
    roofs = json.load(open(os.path.join(REPO, "profiles", "roofs.json")))
    e = roofs["cfg2/poses160/f16"]
    if e["source_hash"] == committed["loss_kernel_source_hash"]:
        assert abs(e["mix_ceiling_cycles_per_instr"] - committed["instances"]["f16"]["mix_ceiling_cycles_per_instr"]) < 1e-9
    v = bench.valu_roof(e, 1_000_000, 160, 0.45)
    assert abs(v["frac_of_mix_ceiling"] - v["frac"] * e["mix_ceiling_cycles_per_instr"] / 4.0) < 1e-12 and 0.9 < v["frac_of_mix_ceiling"] < 1.0
    assert abs(v["fp32_flop_frac"] - e["fp32_flops_per_point_pose"] * 160e6 / 0.45e-3 / 157.3e12) < 1e-12 and 0.3 < v["fp32_flop_frac"] < 0.5
