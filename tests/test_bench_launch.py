"""bench.py's launch path without a GPU: `python bench.py --gpus N` started WITHOUT a launcher must spawn its own N ranks
(torch.distributed.run on 127.0.0.1, a free port) and print one JSON line from rank 0; started under torch.distributed.run it
must use the ranks it is given.  --launch-check stops after the rendezvous (gloo where there is no GPU), so no kernel runs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""     # the rendezvous alone: gloo even on a GPU box
    return env


def test_plain_python_spawns_its_own_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["world_size_seen"] == 2 and line["rank_sum"] == 3.0 and line["launch"] == "self"


def test_under_torchrun_uses_the_given_ranks():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                          "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 3 and line["world_size_seen"] == 3 and line["rank_sum"] == 6.0 and line["launch"] == "torchrun"


def test_single_process_launch_check():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check"], env=_env(), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert _json_line(out.stdout)["world_size_seen"] == 1


def test_world_size_mismatch_is_refused():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in out.stderr
