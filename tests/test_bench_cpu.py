"""CPU: bench.py's launch contract (the driver runs `python bench.py --gpus N`): without a launcher in the environment the
script starts one rank per GPU as CHILD processes before anything touches the GPU, relays rank 0's JSON line -- and only
that -- to stdout, and passes the children's failure on."""
import importlib.util
import json
import os
import subprocess
import sys
import types

import pytest

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if getattr(mod, "_BLAS_LIMIT", None) is not None:  # (bench.py limits the BLAS pool of ITS process at import: not of this one)
        mod._BLAS_LIMIT.restore_original_limits()
    return mod


def test_self_launch_relays_exactly_one_json_line(monkeypatch, capsys):
    bench = _bench()
    line = json.dumps({"metric": "emulated signals/sec (batched predict)", "value": 1.0, "n_gpus": 2})
    noise = "[Gloo] Rank 0 is connected to 1 peer ranks.\n{\"not\": \"the line\"}\nsome banner\n"
    seen = {}

    def fake_run(cmd, stdout=None, env=None, timeout=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout=(noise + line + "\n").encode())
    monkeypatch.setattr(subprocess, "run", fake_run)
    with pytest.raises(SystemExit) as e:
        bench._self_launch(2, ["--gpus", "2", "--steps", "5"])
    assert e.value.code == 0
    out, err = capsys.readouterr()
    assert out.strip() == line and "Gloo" in err and "the line" in err          # stdout: the line alone; chatter: stderr
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"

    # a child that fails, or prints no line, fails the run
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=3, stdout=(line + "\n").encode()))
    with pytest.raises(SystemExit) as e:
        bench._self_launch(2, [])
    assert e.value.code == 3
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=0, stdout=b"no json here\n"))
    with pytest.raises(SystemExit) as e:
        bench._self_launch(2, [])
    assert e.value.code == 4


def test_main_self_launches_when_no_launcher_is_in_the_environment(monkeypatch):
    bench = _bench()
    called = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])

    def fake_launch(n, argv):
        called["n"], called["argv"] = n, list(argv)
        raise SystemExit(0)
    monkeypatch.setattr(bench, "_self_launch", fake_launch)
    with pytest.raises(SystemExit):
        bench.main()
    assert called == {"n": 4, "argv": ["--gpus", "4", "--steps", "3"]}
