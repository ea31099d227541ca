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
