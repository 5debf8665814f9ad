"""bench.py's own N-rank launcher, on CPU: `python bench.py --gpus 2` must start two ranks itself (the driver's
command has no torchrun in front of it), print ONE JSON line with n_gpus 2 and a verified checksum all-reduce,
and refuse a world size that differs from --gpus.  The ranks run the real call lists of the pass on numpy
buffers over the emulated kernels with the gloo backend -- the line says it is a dry run, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def narrow_emu_buffers():
    """--buffers factory: numpy buffers + emulated kernels, and a narrow pyramid (the emulator is slow)."""
    from maskflownet_amd import hotpath
    from tests.test_hotpath_passes import _EmuBuffers
    for l, c in {6: 6, 5: 4, 4: 4, 3: 4, 2: 4}.items():
        hotpath.CHANNELS[l] = c
    for name in ("tiny", "tiny_train"):   # one 64x64 pair per rank
        hotpath.CONFIGS[name] = (1, 64, 64) + hotpath.CONFIGS[name][3:]
    return _EmuBuffers()


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize("config", ["tiny"])   # (the training pass's bucket all-reduce over two gloo ranks: test_hotpath_passes)
def test_bench_starts_its_own_ranks(config):
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", config, "--backend", "gloo",
              "--buffers", "tests.test_bench_launcher:narrow_emu_buffers", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["checksum_allreduce_ok"] is True
    assert rec["scaling"] == "weak" and rec["config"]["global_batch"] == 2 * rec["config"]["per_gpu_batch"]
    assert rec["config"]["backend"] == "gloo" and "DRY RUN" in rec["data"]
    assert rec["value"] > 0 and rec["ms_per_step"] > 0
    assert "roofline" not in rec and "cpu_baseline" not in rec


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "tiny", "--backend", "gloo",
              "--buffers", "tests.test_bench_launcher:narrow_emu_buffers"],
             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_gloo_backend_needs_dry_run_buffers():
    r = _run(["--gpus", "1", "--backend", "gloo", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "dry run" in r.stderr


@pytest.mark.gpu
def test_rccl_path_runs_on_the_gpu_with_one_rank():
    """The box has one GPU, so N > 1 cannot run here; what can: the very same code path with a real RCCL process group
    of one rank under torch.distributed.run (init, barrier, MAX all-reduce of the step time, checksum all-reduce, and
    for the training pass the all-reduce of the flat gradient bucket)."""
    for config in ("tiny", "tiny_train", "cfg5"):   # cfg5: BASELINE configs[4] at its full per-GPU size -- the 1.1 MB gradient bucket through RCCL
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                            "127.0.0.1", "--master-port", str(29400 + os.getpid() % 500), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                            "--config", config, "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-epe", "--no-e2e"],
                           cwd=ROOT, env=env, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert rec["n_gpus"] == 1 and rec["config"]["backend"] == "nccl" and rec["checksum_allreduce_ok"] is True
        assert rec["value"] > 0 and rec["data"] == "synthetic"
        # the line's self-check for the day N > 1 runs: what the process group says it is, every rank's own rate
        pg = rec["distributed"]["process_group"]
        assert pg["world_size"] == 1 and pg["backend"] == "nccl" and rec["distributed"]["world_size_env"] == 1
        assert "rccl_version" in pg and rec["distributed"]["per_rank_pairs_per_s"]["n"] == 1
        assert abs(rec["distributed"]["per_rank_pairs_per_s"]["max"] - rec["value"]) <= 1e-6 * rec["value"] + 0.01


@pytest.mark.gpu
def test_more_ranks_than_gpus_is_refused():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--config", "tiny"])
    assert r.returncode != 0 and "refusing" in r.stderr
