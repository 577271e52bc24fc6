"""bench.py's self-launcher (`python bench.py --gpus N` without torchrun): child environments, record forwarding,
failure handling.  No GPU: the children here are tiny stand-in scripts."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_dry_launch_prints_n_child_envs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--dry-launch"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_ranks"] == 8 and len(rec["env"]) == 8
    assert [e["RANK"] for e in rec["env"]] == [str(r) for r in range(8)] and [e["LOCAL_RANK"] for e in rec["env"]] == [str(r) for r in range(8)]
    assert all(e["WORLD_SIZE"] == "8" and e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in rec["env"])
    assert len({e["MASTER_PORT"] for e in rec["env"]}) == 1
    assert "--dry-launch" not in rec["argv"] and rec["argv"][-4:] == ["--gpus", "8", "--steps", "3"]


def test_gpus_gt_1_without_devices_fails_loudly():
    """no GPU in the build container: asking for 2 ranks must exit non-zero, not fall back to one GPU and report n_gpus 1"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    import bench
    have = bench.visible_gpu_count()
    if have is None or have < 2:
        assert out.returncode != 0 and ("device" in out.stderr or "GPU" in out.stderr)


def test_visible_gpu_count_reads_sysfs_only(tmp_path):
    """the launcher's parent counts devices from the KFD topology (simd_count > 0), narrowed by the *_VISIBLE_DEVICES variables — never through
    torch / HIP: a missing sysfs tree answers None and the check is skipped"""
    import bench
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):                    # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    root = str(tmp_path)
    assert bench.visible_gpu_count(env={}, kfd_root=root) == 3
    assert bench.visible_gpu_count(env={"HIP_VISIBLE_DEVICES": "0,2"}, kfd_root=root) == 2
    assert bench.visible_gpu_count(env={"ROCR_VISIBLE_DEVICES": "1"}, kfd_root=root) == 1
    assert bench.visible_gpu_count(env={"HIP_VISIBLE_DEVICES": ""}, kfd_root=root) == 0
    assert bench.visible_gpu_count(env={}, kfd_root=str(tmp_path / "absent")) is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("def main") if "def main" in src else len(src)]
    assert "device_count(" not in body


def _launch(tmp_path, body, n=2):
    import bench
    script = tmp_path / "child.py"
    script.write_text("import os, sys, json, time\nrank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n" + body)
    args = argparse.Namespace(gpus=n, dry_launch=False)
    return bench.launch_ranks(args, [], child_cmd=[sys.executable, str(script)], check_devices=False)


def test_launcher_forwards_rank0_record(tmp_path, capfd):
    rc = _launch(tmp_path, "if rank == 0:\n    print('banner noise'); print(json.dumps({'n_gpus': world, 'rccl_ranks': world, 'value': 1.0}))\n", n=3)
    assert rc == 0
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 3, "rccl_ranks": 3, "value": 1.0}


def test_launcher_rejects_wrong_rank_count(tmp_path):
    assert _launch(tmp_path, "if rank == 0:\n    print(json.dumps({'n_gpus': 1, 'rccl_ranks': 1, 'value': 1.0}))\n") == 1


def test_launcher_fails_fast_when_a_rank_dies(tmp_path):
    """rank 1 dies; rank 0 would sit in a collective forever: the launcher terminates it and returns non-zero"""
    t0 = time.time()
    rc = _launch(tmp_path, "if rank == 1:\n    sys.exit(7)\ntime.sleep(600)\n")
    assert rc == 1 and time.time() - t0 < 60


def test_one_gpu_per_rank_guard():
    """`world` ranks must sit on `world` DIFFERENT GPUs — two ranks that landed on one device (mis-set HIP_VISIBLE_DEVICES)
    would surface in RCCL only as a hang or a 'duplicate GPU' abort.  The guard compares the gathered device identities (here: handed in)."""
    import pytest
    import bench
    assert bench.check_one_gpu_per_rank(None, None, 4, 0, identities=["uuid:a", "uuid:b", "uuid:c", "uuid:d"]) == ["uuid:a", "uuid:b", "uuid:c", "uuid:d"]
    with pytest.raises(SystemExit) as e:
        bench.check_one_gpu_per_rank(None, None, 4, 2, identities=["uuid:a", "uuid:b", "uuid:a", "uuid:d"])
    assert "3 distinct GPUs" in str(e.value) and "rank 2" in str(e.value)
    # an identity that cannot be established (no UUID, no PCI address) is not a collision: warn, do not abort
    assert bench.check_one_gpu_per_rank(None, None, 2, 0, identities=[None, "uuid=a/pci=0:1:0"]) == [None, "uuid=a/pci=0:1:0"]


def test_one_gpu_per_rank_guard_over_gloo_world2(tmp_path):
    """the same guard through its collective (all_gather_object) on two host processes: both ranks report the same made-up identity -> both exit non-zero"""
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch.distributed as dist\n"
        "import bench\n"
        "rank, same = int(os.environ['RANK']), sys.argv[1] == 'same'\n"
        "dist.init_process_group('gloo', rank=rank, world_size=2)\n"
        "bench.device_identity = lambda dev: 'uuid:x' if same else f'uuid:{rank}'\n"
        "ids = bench.check_one_gpu_per_rank(dist, None, 2, rank)\n"
        "print('ok', ids)\n"
        "dist.destroy_process_group()\n")
    import socket
    for mode, want in (("same", 1), ("different", 0)):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = [subprocess.Popen([sys.executable, str(child), mode], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=300)[0].decode() for p in procs]
        assert all((p.returncode != 0) == bool(want) for p in procs), outs
        if want:
            assert all("1 distinct GPUs" in o for o in outs), outs


def test_record_line_is_strict_json():
    """a diverged loss must not put NaN / Infinity into the one record line (strict JSON parsers reject them)"""
    import bench
    rec = bench._json_safe({"value": 1.0, "config": {"final_loss": float("nan"), "xs": [float("inf"), 2.0, (3.0, float("-inf"))]}})
    text = json.dumps(rec, allow_nan=False)
    assert json.loads(text) == {"value": 1.0, "config": {"final_loss": None, "xs": [None, 2.0, [3.0, None]]}}
