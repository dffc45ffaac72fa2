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


def test_rehearsal_line_is_self_proving_every_rank_checks_its_own_block():
    """round-3 verdict, item 2: an N > 1 line carries `verified` = AND over the ranks' own checks (a sample of each
    rank's block incl. its first and last row against the plain-C oracle), the world size the process group reports,
    who each rank is, and the per-rank times -- rehearsed here on gloo with the test engine standing in for the device"""
    r = _run("--gpus", "2", "--rehearse-cpu", "--cols", "300", "--check-rows", "48")
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_lines(r.stdout)[0]
    assert out["verified"] is True and out["ranks"] == 2 and len(out["devices"]) == 2
    assert out["per_rank_ms"] == {"min": 1.0, "max": 2.0, "all": [1.0, 2.0]}
    assert [p["rows"] for p in out["per_rank"]] == [[0, 150], [150, 300]]
    for p in out["per_rank"]:
        c = p["check"]
        assert p["verified"] is True and c["failures"] == [] and c["rows_checked"] == 48
        assert c["first_row"] == 0 and c["last_row"] == 149 and c["rows_in_block"] == 150
    assert len({d["uuid"] for d in out["devices"]}) == 2               # two different processes


def test_one_wrong_rank_makes_the_line_unverified_and_the_launch_fail():
    r = _run("--gpus", "2", "--rehearse-cpu", "--cols", "300", "--check-rows", "48", env={"SPC_REHEARSAL_CORRUPT_RANK": "1"})
    assert r.returncode != 0
    out = _json_lines(r.stdout)[0]
    assert out["verified"] is False
    assert [p["verified"] for p in out["per_rank"]] == [True, False]
    assert any("f_T" in f for f in out["per_rank"][1]["check"]["failures"])


def test_sample_rows_always_hold_the_first_and_the_last_row():
    import numpy
    sys.path.insert(0, ROOT)
    import bench
    for n, m in ((10, 4096), (4096, 4096), (4097, 4096), (43566, 4096), (174264, 4096), (150, 48)):
        rows = bench.sample_rows(n, m)
        assert rows[0] == 0 and rows[-1] == n - 1 and len(rows) <= m and len(numpy.unique(rows)) == len(rows)
        assert (numpy.diff(rows) > 0).all() and (n <= m or len(rows) >= m - 2)
