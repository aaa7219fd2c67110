"""bench.py's host logic that needs no GPU: the N > 1 self-launch."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_a_launcher_refuses_cleanly_on_a_box_with_fewer_devices():
    """`python bench.py --gpus 8` with WORLD_SIZE unset on a box that has fewer than 8 GPUs (this container: none): non-zero exit, the
    distinct_devices message, no JSON line, no CPU baseline run, no hang — within seconds (the time is `import torch`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    dt = time.perf_counter() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert "distinct_devices" in r.stderr and "--gpus 8" in r.stderr
    assert r.stdout.strip() == ""
    assert dt < 60, dt  # (a cold `import torch` on a fresh container is the only cost; 2-3 s warm)


def test_self_launch_starts_one_child_launcher_and_forwards_its_exit_code(monkeypatch, tmp_path):
    """with enough devices the parent starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <args>` as a
    CHILD (subprocess, never exec), hands its own arguments on unchanged and returns the child's code"""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, stdout=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.self_launch(4, ["--gpus", "4", "--steps", "2", "--warmup", "1"])
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
