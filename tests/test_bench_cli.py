"""bench.py's own code paths on the CPU emulator (toy sizes; the numbers mean nothing): the JSON contract of the N = 1
line and the driver's N = 2 launch line (`python -m torch.distributed.run ... bench.py --gpus 2`, gloo instead of RCCL)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--gaussians", "1000", "--cameras", "4", "--points", "10000", "--steps", "1", "--warmup", "1",
         "--no-parity", "--no-extra", "--no-cpu-baseline"]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _env():
    env = dict(os.environ, G2PC_BENCH_EMULATE="1", G2PC_DIST_BACKEND="gloo")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    return env


def test_bench_line_contract_single_process():
    from emu_util import build_emu
    build_emu()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "points/s" and d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]


def test_bench_two_ranks_under_torch_distributed_run():
    from emu_util import build_emu
    build_emu()
    port = 29800 + (os.getpid() % 150)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _line(r.stdout)                      # exactly ONE line, from rank 0
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["cameras"] == 4 and d["config"]["cameras_per_gpu"] == 2       # the job is split, not multiplied


def test_bare_gpus_flag_relaunches_itself_under_torch_distributed_run():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment used to run ONE rank and print
    n_gpus: 1.  It now re-executes itself under torch.distributed.run; the line carries the per-rank stage breakdown."""
    from emu_util import build_emu
    build_emu()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2
    pr = d["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and sum(x["cameras"] for x in pr) == 4
    for x in pr:
        for k in ("setup_ms", "camera_loop_ms", "exchange_ms", "fixed_ms", "sample_ms"):
            assert x[k] >= 0.0, (k, x)


def test_gpus_flag_that_disagrees_with_the_launch_fails():
    env = dict(_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
