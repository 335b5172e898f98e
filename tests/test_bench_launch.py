"""bench.py --gpus N from a bare launch: the process (which has not touched the GPU) starts the N ranks under
torch.distributed.run as a child, relays rank 0's JSON line and leaves with the children's status."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_launch_composes_the_torchrun_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7                                # the children's status is the exit status
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bare_launch_two_ranks_on_one_gpu_prints_one_json_line():
    fake = os.path.join(ROOT, "tests", "fake_rccl", "_build", "libfake_rccl.so")
    assert os.path.exists(fake)
    # two ranks on the ONE test GPU: gloo for torch.distributed, the librccl stand-in for the library's own collective
    import tempfile
    import time
    counter = tempfile.NamedTemporaryFile(prefix="bench_gen_", suffix=".txt", delete=False).name
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_ONE_DEVICE="1", VBNMF_RCCL_LIB=fake, BENCH_GEN_COUNTER=counter)
    env.pop("WORLD_SIZE", None)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--small", "--steps", "5", "--warmup", "2",
                        "--no-cpu", "--no-ml"], env=env, capture_output=True, text=True, timeout=900)
    print(f"two-rank rehearsal: {time.perf_counter() - t0:.1f} s wall")
    assert p.returncode == 0, p.stderr[-2000:]
    # ONE generation of each synthetic matrix per node, whatever the number of ranks: the node's local
    # rank 0 generates, the others map its arrays
    gens = sorted(ln.split()[0] for ln in open(counter).read().splitlines())
    os.unlink(counter)
    assert gens == ["c3s", "c5s"], gens
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["value"] > 0
    assert "value_host_stepped" in out and out["warmup_effective"] >= 2 and "setup" in out
    # config C4 and C5 ride in the same line (VERDICT r03 Next #3)
    sweep = out["rank_sweep"]
    assert "error" not in sweep, sweep
    assert sweep["wall_s"] > 0 and sweep["iterations_by_rank"] and set(sweep["iterations_by_rank"]) <= {str(r) for r in range(2, 7)}
    assert len(sweep["per_process"]) == 2 and sorted(r for q in sweep["per_process"] for r in q["ranks"]) == list(range(2, 7))
    split = sweep["one_unit_at_a_time"]                       # the stepping / set-up split comes from a second call, one unit at a time
    assert split["wall_s"] > 0 and all(q["stepping_s"] > 0 and q["setup_s"] > 0 for q in split["per_process"])
    cells = out["cells_partitioned"]
    assert "error" not in cells, cells
    assert cells["value"] > 0 and cells["allreduce_ms"] > 0 and "device-driven" in cells["loop"]
    assert 0 < cells["roofline"]["hbm"]["frac"] < 1 and 0 < cells["roofline"]["fp64"]["frac"] < 1
