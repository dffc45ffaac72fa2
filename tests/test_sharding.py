"""CPU tests of the N>1 path: column partitioning and the host-side gather, with world_size-2 (and 3)
gloo process groups.  No GPU here, so the per-rank compute is done by a TEST-ONLY engine backed by the
oracle (defined below, injected into ShardedExchange); the product default is the HIP engine."""
import os
import socket
import sys

import numpy
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sp_coupler_amd import sharding, synthetic  # noqa: E402


def test_shard_bounds_cover_all_columns_once():
    for n in (0, 1, 2, 7, 1024, 35718, 348528):
        for w in (1, 2, 3, 4, 8):
            b = sharding.shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
            per = -(-n // w)
            assert all(b[i + 1] - b[i] <= per for i in range(w))
            assert sum(b[i + 1] - b[i] for i in range(w)) == n
    assert sharding.shard_range(348528, 7, 8) == (304962, 348528)       # T511 on 8 GPUs: 43 566 columns each


def test_shard_rows_replicates_shared_grid():
    gcm, zf, zh, prof = synthetic.make_batch(10, 19, 160, seed=1)
    s = sharding.shard_rows(dict(gcm, zf=zf), 10, 1, 3, replicate=("zf",))
    assert s["T"].shape == (4, 19) and s["zf"].shape == (160,) and numpy.array_equal(s["T"], gcm["T"][4:8])
    assert s["Z0M"].shape == (4,)


from tests.fake_engine import OracleEngine        # TEST-ONLY stand-in for sp_coupler_amd.engine.Engine on a GPU-less box


def _worker(rank, world, port, n_cols, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gcm, zf, zh, prof = synthetic.make_batch(n_cols, 19, 160, seed=77)      # every rank builds the same batch
        ex = sharding.ShardedExchange(OracleEngine(), n_cols, rank, world)
        g, zf_d, zh_d, p = ex.upload(gcm, zf, zh, prof)
        assert g["T"].shape[0] == ex.hi - ex.lo
        fwd, bwd = ex.exchange(g, zf_d, zh_d, p, 0.5, 0.5, 450.0)          # builds the lean plans ...
        fwd, bwd = ex.exchange(g, zf_d, zh_d, p, 1.0, 1.0, 900.0)          # ... and relaunches them with new scalars
        assert "Zf" not in fwd and "u" not in fwd                           # the lean K1: no optional outputs
        full = ex.gather({"f_thl": fwd["f_thl"], "idx": fwd["idx"], "f_ps": fwd["f_ps"], "f_T": bwd["f_T"],
                          "f_A": bwd["f_A"]})
        dist.barrier()
        if rank == 0:
            numpy.savez(os.path.join(outdir, "gathered.npz"), **full)
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n_cols", [(2, 10), (3, 7)])
def test_sharded_exchange_gathers_same_result_as_single_rank(tmp_path, world, n_cols):
    mp.spawn(_worker, args=(world, _free_port(), n_cols, str(tmp_path)), nprocs=world, join=True)
    got = numpy.load(os.path.join(str(tmp_path), "gathered.npz"))
    from oracle import spcpl_oracle as orc
    gcm, zf, zh, prof = synthetic.make_batch(n_cols, 19, 160, seed=77)
    f = orc.forward_batched(gcm, prof, zf, zh, 1.0, 900.0)
    b = orc.backward_batched(gcm, f["Zf"], prof, zf, 1.0, 900.0)
    assert numpy.array_equal(got["f_thl"], f["f_thl"]) and numpy.array_equal(got["idx"], f["idx"])
    assert numpy.array_equal(got["f_ps"], f["f_ps"]) and numpy.array_equal(got["f_T"], b["f_T"], equal_nan=True)
    assert numpy.array_equal(got["f_A"], b["f_A"], equal_nan=True)
