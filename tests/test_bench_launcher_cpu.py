"""CPU: `bench.py --gpus N` starts N ranks itself.  The launcher branch is host logic: checked here with a
mocked device count and a stand-in child process -- no GPU call is made by the parent."""
import io
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_needs_one_gpu_per_rank():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "5"], {}, ndev=8)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5"]
    with pytest.raises(SystemExit) as e:
        bench.launch_command(8, ["--gpus", "8"], {}, ndev=1)
    assert "only 1 GPU" in str(e.value)
    # the labelled rehearsal backend may share cards
    assert bench.launch_command(2, ["--gpus", "2"], {"STABNET_DIST_BACKEND": "gloo"}, ndev=1)


class _FakeProc:
    def __init__(self, lines, rc):
        self.stdout, self._rc = io.StringIO("".join(lines)), rc

    def wait(self):
        return self._rc


@pytest.mark.parametrize("lines,rc,want", [
    (['noise\n', json.dumps({"metric": "m", "n_gpus": 4}) + "\n"], 0, 0),
    ([json.dumps({"metric": "m", "n_gpus": 1}) + "\n"], 0, 3),           # a run that silently used one GPU is an error
    ([], 0, 3),
    ([json.dumps({"metric": "m", "n_gpus": 4}) + "\n"], 7, 7),           # child failure is relayed
])
def test_launch_ranks_checks_the_reported_world(monkeypatch, capsys, lines, rc, want):
    calls = []

    def popen(cmd, **kw):
        calls.append((cmd, kw))
        return _FakeProc(lines, rc)

    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    monkeypatch.setattr(subprocess, "Popen", popen)
    args = types.SimpleNamespace(gpus=4)
    assert bench.launch_ranks(args, ["--gpus", "4"]) == want
    assert calls and calls[0][0][cmd_index(calls[0][0])] == "4"
    assert calls[0][1]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    if lines:
        assert lines[-1] in capsys.readouterr().out                      # rank 0's line is relayed verbatim


def cmd_index(cmd):
    return cmd.index("--nproc-per-node") + 1


def test_main_launches_before_any_gpu_call_and_rejects_a_mismatched_world(monkeypatch):
    seen = {}
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "launch_ranks", lambda a, argv: seen.setdefault("argv", argv) and 0)
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: pytest.fail("GPU touched before the ranks were started"))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen["argv"] == ["--gpus", "2", "--steps", "1"]
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "must agree" in str(e.value)


def test_visible_gpus_reads_sysfs_not_the_runtime(monkeypatch, tmp_path):
    """the launcher parent counts devices from the KFD topology (sysfs) and the *_VISIBLE_DEVICES lists; the GPU
    runtime is only the fallback when sysfs is unreadable."""
    for i, simd in enumerate([0, 0, 256, 256, 256]):                     # two CPU nodes, three GPU agents
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: pytest.fail("runtime touched"))
    assert bench.visible_gpus({}, str(tmp_path)) == 3
    assert bench.visible_gpus({"HIP_VISIBLE_DEVICES": "0,2"}, str(tmp_path)) == 2
    assert bench.visible_gpus({"ROCR_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1,2"}, str(tmp_path)) == 1
    assert bench.visible_gpus({"CUDA_VISIBLE_DEVICES": ""}, str(tmp_path)) == 0
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 5)
    assert bench.visible_gpus({}, str(tmp_path / "absent")) == 5           # no sysfs: the documented fallback
