"""bench.py as its own multi-GPU launcher (CPU-only checks): the command a plain
`python bench.py --gpus N` re-issues itself as, and the parent's relay of rank 0's line."""
import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_importing_bench_does_not_import_torch_cuda_state(bench):
    # the launcher parent must not touch the GPU: bench.py imports torch only inside main()
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    assert "\nimport torch" not in head and "\nfrom torch" not in head


def test_launcher_command_shape(bench):
    cmd = bench.launcher_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29511,
                                 python="/usr/bin/python3")
    assert cmd[:3] == ["/usr/bin/python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_self_launch_relays_rank0_line_and_status(bench, monkeypatch, capfd, tmp_path):
    child = tmp_path / "child.py"
    child.write_text(
        "import json, sys\n"
        "print('noise from a rank')\n"
        "print(json.dumps({'metric': 'm', 'value': 1.5, 'n_gpus': int(sys.argv[1])}))\n"
        "sys.exit(int(sys.argv[2]))\n")
    for rc in (0, 3):
        monkeypatch.setattr(bench, "launcher_command",
                            lambda argv, gpus, rc=rc: [sys.executable, str(child), str(gpus), str(rc)])
        got = bench.self_launch(["--gpus", "4"], 4)
        out, err = capfd.readouterr()
        assert got == rc
        lines = [l for l in out.splitlines() if l.strip()]
        assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1.5, "n_gpus": 4}
        assert "noise from a rank" in err


def test_self_launch_without_a_result_line_fails(bench, monkeypatch, capfd):
    monkeypatch.setattr(bench, "launcher_command", lambda argv, gpus: [sys.executable, "-c", "print('x')"])
    assert bench.self_launch([], 2) != 0
