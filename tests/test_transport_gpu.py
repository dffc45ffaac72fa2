"""-m gpu: the copy-stream ("piecewise") transport every real batch size takes.

``spcpl.StepBuffers.piecewise`` switches every batch whose arrays reach ``transfer.Arena.PIECE_MIN`` (4 MiB: 3 277
columns at nL = 160 -- configs 3, 4, 5 and every per-GPU shard of them) from one copy per buffer to per-array copies on
the buffers' own copy streams, overlapped with the model objects' getter / setter calls (``transfer.Arena.push / fence /
pull / ready / settle / writable``).  The closed loop of ``driver.Coupler`` at 6 144 columns, 91 <-> 160 levels (there
the [n x nG] arrays, 4.5 MB, travel on their own too, not only the [n x nL] ones), is compared here

* bit for bit with the same run on the single-copy path (``PIECE_MIN`` raised to 2**62), for the batched model protocol
  and the reference's per-LES loop, surface coupling on / off, conservative coarsening on / off, on one ``Engine``, on an
  ``Engine`` with a stream of its own, on two engines sharing the card and on engines with their own streams;
* with the oracle-driven reference sequencing (``tests/ref_driver.RefCoupler``, splib/splib.py:267-352 restated) on a
  sample of columns incl. the first and the last row;
* under model objects whose getters / setters sleep 0-2 ms at random, so copies and host calls interleave differently;
* in the two orderings the fences exist for (commit c1f9f32): a second ``gather_gcm_data`` while the kernel reading the
  first one's upload has not run yet, and a kernel overwriting results whose download nobody has waited for.

Reference semantics of the calls either side of the copies: splib/spcpl.py:341-347, 535-542, 748-766."""
import hashlib
import random
import time

import numpy
import pytest

from sp_coupler_amd import models, spcpl, transfer
from tests.ref_driver import RefCoupler

pytestmark = pytest.mark.gpu

N, NG, NL, STEPS = 6144, 91, 160, 3
SAMPLE = sorted({0, 1, 2, 3, 4, 5, 6, 7, N - 1, N - 2, N - 3, N - 4, 1023, 1024, 2047, 2048, 2049, 3071, 3072, 3276, 3277, 5761, 5762})


def _digest(a):
    return hashlib.blake2b(numpy.ascontiguousarray(a).view(numpy.uint8), digest_size=16).hexdigest()


class _Tap:
    """records a digest of every array the batched model objects receive, in order (what crossed PCIe downwards)"""

    def __init__(self, gcm, ens, sleepy=None):
        self.log = []
        rng = random.Random(sleepy) if sleepy is not None else None

        def nap():
            if rng is not None:
                time.sleep(rng.uniform(0.0, 2e-3))
        o_set, o_tend = ens.set_forcings_batched, getattr(gcm, "set_profile_tendencies", None)
        o_get, o_cf, o_gcm_get = ens.get_profiles_batched, ens.get_cloudfraction_batched, gcm.get_profile_fields

        def set_forcings_batched(**arrays):
            nap()
            for k, v in arrays.items():
                self.log.append(("les", k, _digest(v)))
            return o_set(**arrays)

        def set_profile_tendencies(var, gi, values):
            nap()
            self.log.append(("gcm", var, _digest(values)))
            return o_tend(var, gi, values)

        def get_profiles_batched(keys, out):
            nap()
            return o_get(keys, out)

        def get_cloudfraction_batched(indices, out):
            nap()
            self.log.append(("idx", "idx", _digest(indices)))
            return o_cf(indices, out)

        def get_profile_fields(var, cols, out=None):
            nap()
            return o_gcm_get(var, cols, out=out) if out is not None else o_gcm_get(var, cols)
        ens.set_forcings_batched, ens.get_profiles_batched = set_forcings_batched, get_profiles_batched
        ens.get_cloudfraction_batched = get_cloudfraction_batched
        if o_tend is not None:
            gcm.set_profile_tendencies = set_profile_tendencies
            gcm.get_profile_fields = gcm.get_surface_field = get_profile_fields


def _run(engine, batched, cplsurf, conservative, piece_min=None, sleepy=None, n=N, steps=STEPS, seed=5):
    """closed loop of driver.Coupler; returns everything a torn or misordered copy would change"""
    from sp_coupler_amd.driver import Coupler
    old = transfer.Arena.PIECE_MIN
    if piece_min is not None:
        transfer.Arena.PIECE_MIN = piece_min
    spcpl.set_engine(engine)
    try:
        gcm, ens = models.make_batched_models(n, nG=NG, nL=NL, seed=seed)
        tap = None
        if batched:
            tap = _Tap(gcm, ens, sleepy)
            les = ens
        else:
            gcm.__class__ = models.SyntheticGCM          # the reference's per-column protocol only
            les = [ens[i] for i in range(n)]
        cpl = Coupler(gcm, les, cplsurf=cplsurf, conservative_coarsening=conservative, les_forcing_factor=0.9,
                      gcm_forcing_factor=1.1)
        cpl.run(steps)
        b = spcpl.current_batch()
        assert b.n == n
        res = {"gcm:" + k: _digest(gcm.state[k]) for k in ("U", "V", "T", "SH", "QL", "QI", "A")}
        res.update({"les:" + k: _digest(ens.p[k]) for k in ("U", "V", "THL", "QT", "QL", "PS", "T")})
        res.update({"tend:" + k: _digest(v) for k, v in sorted(ens.tend.items())})
        res["idx"] = _digest(spcpl._index_map(b))
        if tap is not None:
            res["log"] = tuple(tap.log)
        return res, {k: gcm.state[k].copy() for k in ("U", "V", "T", "SH", "QL", "QI", "A")}, b.buf.piecewise, type(b.buf.gcm_in).__name__
    finally:
        transfer.Arena.PIECE_MIN = old
        spcpl.set_engine(None)


def _engine(kind):
    import torch
    from sp_coupler_amd.engine import Engine
    from sp_coupler_amd.multi import MultiDeviceEngine
    dev = torch.device("cuda:0")
    if kind == "one":
        return Engine(dev)
    if kind == "own_stream":                             # an engine that owns its stream, through spcpl.set_engine
        return Engine(dev, stream=torch.cuda.Stream(dev))
    if kind == "two_on_one_card":
        return MultiDeviceEngine([Engine(dev), Engine(dev)], min_cols_per_device=1000)
    if kind == "own_streams":
        return MultiDeviceEngine([Engine(dev, stream=torch.cuda.Stream(dev)), Engine(dev, stream=torch.cuda.Stream(dev)),
                                  Engine(dev, stream=torch.cuda.Stream(dev))], min_cols_per_device=1000)
    if kind == "primary_with_stream_below_threshold":   # the batch stays whole on the primary engine, which owns a stream
        return MultiDeviceEngine([Engine(dev, stream=torch.cuda.Stream(dev)), Engine(dev)], min_cols_per_device=N)
    raise ValueError(kind)


_SINGLE = {}


def _single_copy(batched, cplsurf, conservative):
    """the run every variant is compared with: one plain Engine, single-copy transport (cached per flag combination)"""
    key = (batched, cplsurf, conservative)
    if key not in _SINGLE:
        res, state, piecewise, arena = _run(_engine("one"), batched, cplsurf, conservative, piece_min=2 ** 62)
        assert not piecewise and arena == "Arena"
        _SINGLE[key] = (res, state)
    return _SINGLE[key]


def _same(got, want):
    for k in want:
        assert got[k] == want[k], "%s differs between the two transports" % k


@pytest.mark.parametrize("batched", [True, False])
@pytest.mark.parametrize("cplsurf", [False, True])
@pytest.mark.parametrize("conservative", [False, True])
def test_copy_stream_transport_equals_single_copy_and_the_reference_sequencing(batched, cplsurf, conservative):
    want, state = _single_copy(batched, cplsurf, conservative)
    got, got_state, piecewise, arena = _run(_engine("one"), batched, cplsurf, conservative)
    assert piecewise and arena == "Arena"                # 6144 x 91 x 8 B = 4.3 MiB >= PIECE_MIN: every array on its own
    assert N * NG * 8 >= transfer.Arena.PIECE_MIN
    _same(got, want)
    # ... and both against splib.step's sequencing executed with the oracle on a sample of columns (first / last rows,
    # rows either side of the single-copy threshold)
    gcm_r, ens_r = models.make_batched_models(N, nG=NG, nL=NL, seed=5)
    ref = RefCoupler(gcm_r, [ens_r[i] for i in SAMPLE], cplsurf=cplsurf, les_forcing_factor=0.9, gcm_forcing_factor=1.1,
                     conservative=conservative)
    for _ in range(STEPS):
        ref.step()
    rows = numpy.asarray(SAMPLE) + 1                     # LES i sits in GCM column i + 1 (models.make_batched_models)
    for var in ("U", "V", "T", "SH", "QL", "QI", "A"):
        a, b = got_state[var][rows], gcm_r.state[var][rows]
        assert numpy.abs(a - b).max() <= 1e-11 * max(numpy.abs(b).max(), 1e-30), var      # thl / T pass through the device pow


@pytest.mark.parametrize("kind,batched,cplsurf,conservative", [
    ("own_stream", True, True, False), ("own_stream", False, False, True), ("own_stream", True, False, True),
    ("two_on_one_card", True, True, True), ("two_on_one_card", False, True, False),
    ("own_streams", True, False, False), ("own_streams", True, True, True), ("own_streams", False, False, False),
    ("primary_with_stream_below_threshold", True, True, False)])
def test_engines_with_streams_and_row_blocks_carry_the_same_bits(kind, batched, cplsurf, conservative):
    want, _ = _single_copy(batched, cplsurf, conservative)
    got, _, piecewise, arena = _run(_engine(kind), batched, cplsurf, conservative)
    assert arena == ("ShardedArena" if kind in ("two_on_one_card", "own_streams") else "Arena")
    _same(got, want)


@pytest.mark.parametrize("seed,cplsurf,conservative", [(1, False, False), (2, True, True)])
def test_random_model_latencies_do_not_change_a_bit(seed, cplsurf, conservative):
    """getters / setters that take 0-2 ms at random: the copies overlap different host calls in every run"""
    want, _ = _single_copy(True, cplsurf, conservative)
    got, _, piecewise, _ = _run(_engine("one"), True, cplsurf, conservative, sleepy=seed)
    assert piecewise
    _same(got, want)
    got, _, _, _ = _run(_engine("own_stream"), True, cplsurf, conservative, sleepy=seed + 10)
    _same(got, want)


# ---- the two orderings the fences exist for -------------------------------------------------------------------------
def _delay(stream, ms=30.0):
    """keep ``stream`` busy for roughly ``ms`` (a spin kernel), so that what is enqueued behind it has not started when
    the host goes on"""
    import torch
    with torch.cuda.stream(stream):
        if hasattr(torch.cuda, "_sleep"):
            torch.cuda._sleep(int(ms * 1e-3 * 2.0e9))
        else:
            junk = torch.empty(1 << 28, dtype=torch.float32, device=stream.device)
            for _ in range(int(ms * 4)):
                junk.fill_(1.0)


def _hazard_setup(engine):
    from sp_coupler_amd.driver import Coupler
    spcpl.set_engine(engine)
    gcm, ens = models.make_batched_models(N, nG=NG, nL=NL, seed=11)
    cpl = Coupler(gcm, ens)
    cpl.step()                                           # past the first step: slab means of this geometry are on the device
    return gcm, ens, cpl


@pytest.mark.parametrize("kind", ["one", "own_stream"])
def test_second_gather_does_not_overwrite_inputs_of_a_kernel_that_has_not_run(kind):
    """write-after-read on the upload side: K1 is enqueued behind a busy compute stream; the next gather_gcm_data refills
    and re-sends the GCM state.  Arena.push must make the copy stream wait for that K1 -- its results must be those of
    the FIRST state."""
    import torch
    try:
        eng = _engine(kind)
        gcm, ens, cpl = _hazard_setup(eng)
        dt = gcm.get_timestep()
        cur = eng.stream if eng.stream is not None else torch.cuda.current_stream(eng.device)
        batch = spcpl.gather_gcm_data(gcm, ens, False)
        assert batch.buf.piecewise
        want = {k: v.copy() for k, v in spcpl.forward_batched(batch, batch.buf.les_in.d, dt, 1.0).items() if k != "ql"}
        state1 = {v: gcm.state[v].copy() for v in spcpl.gcm_vars}
        batch = spcpl.gather_gcm_data(gcm, ens, False)                   # the same state again, K1 not launched yet
        _delay(cur)
        host = spcpl.forward_batched(batch, batch.buf.les_in.d, dt, 1.0, wait=False)     # K1 waits behind the delay
        for v in ("U", "V", "T", "SH", "QL"):                            # a different GCM state ...
            gcm.state[v] = gcm.state[v] * 1.5 + 0.25
        spcpl.gather_gcm_data(gcm, ens, False)                           # ... refilled and pushed while K1 is still queued
        batch.buf.fwd_out.ready()
        for k, w in want.items():
            assert numpy.array_equal(host[k], w), "%s: K1 read the inputs of the NEXT gather" % k
        # and the next launch does see the new state
        got2 = spcpl.forward_batched(batch, batch.buf.les_in.d, dt, 1.0)
        assert not numpy.array_equal(got2["f_u"], want["f_u"])
        for v, a in state1.items():
            gcm.state[v] = a
        spcpl.gather_gcm_data(gcm, ens, False)
        got3 = spcpl.forward_batched(batch, batch.buf.les_in.d, dt, 1.0)
        for k, w in want.items():
            assert numpy.array_equal(got3[k], w), k
    finally:
        spcpl.set_engine(None)


@pytest.mark.parametrize("kind", ["one", "own_stream"])
def test_a_kernel_does_not_overwrite_results_nobody_has_collected(kind):
    """the download side: results of launch 1 are still crossing PCIe (nobody called ready()) when launch 2, with another
    forcing factor, is enqueued.  Arena.settle must hold launch 2 back -- the host must see launch 1's results whole."""
    import torch
    try:
        eng = _engine(kind)
        gcm, ens, cpl = _hazard_setup(eng)
        dt = gcm.get_timestep()
        batch = spcpl.gather_gcm_data(gcm, ens, False)
        b = batch.buf
        want = {k: v.copy() for k, v in spcpl.forward_batched(batch, b.les_in.d, dt, 0.5).items() if k != "ql"}
        other = {k: v.copy() for k, v in spcpl.forward_batched(batch, b.les_in.d, dt, 1.0).items() if k != "ql"}
        assert not numpy.array_equal(want["f_u"], other["f_u"])
        for _ in range(5):
            host = spcpl.forward_batched(batch, b.les_in.d, dt, 0.5, wait=False)        # launch 1 + its pulls (37 MB)
            plan = b.plans[("fwd", False)]
            plan.set_scalars(1.0, dt)
            spcpl._launch(batch, plan, "k1")                                            # launch 2 straight behind it
            b.fwd_out.ready()
            for k, w in want.items():
                assert numpy.array_equal(host[k], w), "%s: launch 2 overwrote results still downloading" % k
            eng.synchronize() if hasattr(eng, "synchronize") else torch.cuda.synchronize()
    finally:
        spcpl.set_engine(None)
