"""`python bench.py --gpus N` typed as a plain command must start N ranks itself (round-1 review: it exited).
The launcher path is exercised here without a GPU: --dry-run-launcher makes every rank rendezvous over gloo,
reduce a value over the ranks and lets rank 0 print the one JSON line, which the parent relays."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=300)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_gpus_2_self_launches_two_ranks():
    rc, out, err = run_bench("--gpus", "2", "--dry-run-launcher")
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                       # stdout carries exactly the one JSON line
    rec = json.loads(lines[0])
    # every rank's own clock suggests another timed step count; they must all run the one they agreed on
    agreed = rec.pop("agreed_steps")
    assert rec.pop("stepped") == agreed >= 20
    assert rec == {"dry_run": True, "n_gpus": 2, "max_over_ranks": 2.0, "gpus_arg": 2}
    assert "torch.distributed.run" in err and "--nproc-per-node=2" in err


def test_under_torchrun_it_is_one_rank():
    """the driver's own form: started under torch.distributed.run it must not launch again"""
    rc, out, err = run_bench("--gpus", "1", "--dry-run-launcher",
                             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert rc == 0, err[-2000:]
    assert json.loads(out.strip())["n_gpus"] == 1
    assert "starting" not in err


def test_gpus_1_plain_does_not_launch():
    rc, out, err = run_bench("--gpus", "1", "--dry-run-launcher")
    assert rc == 0, err[-2000:]
    rec = json.loads(out.strip())
    assert rec.pop("stepped") == rec.pop("agreed_steps")
    assert rec == {"dry_run": True, "n_gpus": 1, "max_over_ranks": 1.0, "gpus_arg": 1}


def test_a_dead_rank_ends_the_launcher_with_an_error():
    """round-2 review: the launcher waited for its ranks without a limit, so one rank that never returns became
    somebody else's 600 s kill.  With BENCH_LAUNCH_TIMEOUT the launcher kills the ranks' process group itself and
    exits non-zero; nothing of the group is left running."""
    import time
    t0 = time.time()
    rc, out, err = run_bench("--gpus", "2", "--dry-run-launcher",
                             env_extra={"BENCH_DRY_RUN_HANG": "1", "BENCH_LAUNCH_TIMEOUT": "25"})
    assert rc == 124, (rc, err[-2000:])
    assert time.time() - t0 < 120
    assert "killing their process group" in err
    assert out.strip() == ""                          # rank 0 printed, but a failed job relays no result line
    time.sleep(1.0)
    left = subprocess.run(["pgrep", "-f", "bench.py --gpus 2 --dry-run-launcher"], stdout=subprocess.PIPE).stdout.decode().split()
    assert left == [], left
