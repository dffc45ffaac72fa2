"""CPU tests of bench.py's multi-GPU launch path (round-2 verdict, item 2a): `python bench.py --gpus N` started
plainly -- the way the driver starts the N = 1 bench -- must bring up its own N ranks as fresh child processes of
`torch.distributed.run` BEFORE the parent has imported torch or touched a GPU, let rank 0's ONE JSON line through and
return the children's exit code.  `--rehearse-cpu` swaps the GPU work for the same rendezvous / sharding / barrier /
max-over-ranks steps on gloo, so the mechanics run here without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + list(argv), capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)


def _json_lines(text):
    return [json.loads(line) for line in text.splitlines() if line.startswith("{")]


def test_plain_launch_starts_its_own_ranks_and_relays_one_json_line():
    r = _run("--gpus", "2", "--rehearse-cpu", "--steps", "7", "--warmup", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["warmup"] == 2 and out["backend"] == "gloo"
    assert out["shards"] == [[0, 174264], [174264, 348528]]           # config 4, contiguous ceil(n/N) row blocks
    assert out["max_over_ranks_s"] == 0.002                            # MAX over ranks, not rank 0's own 0.001


def test_three_ranks_ragged_shards_and_a_column_override():
    r = _run("--gpus", "3", "--rehearse-cpu", "--cols", "10")
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]["shards"] == [[0, 4], [4, 8], [8, 10]]


def test_a_failing_rank_fails_the_parent():
    r = _run("--gpus", "2", "--rehearse-cpu", "--config", "99")        # no such config: every rank raises KeyError
    assert r.returncode != 0
    assert not _json_lines(r.stdout)


def test_the_parent_does_not_import_torch_before_spawning():
    """the child processes must be started before anything in the parent could initialise a GPU: the parent of a
    `--gpus N` launch never even imports torch"""
    code = ("import runpy, sys\n"
            "sys.argv = [%r, '--gpus', '2', '--rehearse-cpu']\n"
            "try:\n"
            "    runpy.run_path(%r, run_name='__main__')\n"
            "except SystemExit as e:\n"
            "    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules, 'parent imported torch'\n"
            "print('PARENT_CLEAN')\n" % (BENCH, BENCH))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0 and "PARENT_CLEAN" in r.stdout, r.stderr[-2000:]


def test_under_torchrun_the_ranks_do_not_spawn_again():
    """the driver's own N > 1 command line: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--rehearse-cpu"],
                       capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(_json_lines(r.stdout)) == 1


def test_world_size_must_match_gpus():
    r = _run("--gpus", "1", "--rehearse-cpu", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
