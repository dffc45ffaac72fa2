"""CPU test of bench.py's `dropin.verified` machinery (round-3 verdict, item 1): after a Coupler.step on the batched
model protocol, what the model objects RECEIVED must equal a synchronous recompute from what they handed over, and the
oracle on a row sample.  Here the oracle-backed test engine stands for the device (host and device buffers are one), so
the test covers the bookkeeping -- which arrays, which step, which factors -- and that a wrong delivery is reported."""
import numpy

import bench
from sp_coupler_amd import models, spcpl
from sp_coupler_amd.driver import Coupler
from tests.fake_engine import OracleEngine


def _stepped(n=9, steps=3):
    spcpl.set_engine(OracleEngine())
    gcm, ens = models.make_batched_models(n, nG=19, nL=40, seed=3)
    cpl = Coupler(gcm, ens, les_forcing_factor=0.8, gcm_forcing_factor=1.2)
    cpl.run(steps)
    return cpl, gcm, ens


def test_dropin_verify_passes_on_a_correct_step_and_names_the_checks():
    try:
        cpl, gcm, ens = _stepped()
        ok, det = bench.dropin_verify(cpl, gcm, ens, "cpu rehearsal")
        assert ok and det["failures"] == [], det
        assert det["columns"] == 9 and det["transport"] == "one copy per buffer" and len(det["checked"]) == 6
        assert not cpl.firststep and gcm.step == 4                      # the K1 check ran one more step
    finally:
        spcpl.set_engine(None)


def test_dropin_verify_reports_a_tendency_that_arrived_wrong():
    try:
        cpl, gcm, ens = _stepped()
        gi, vals = gcm.tendencies["T"]
        vals[-1, 3] = numpy.nextafter(vals[-1, 3], numpy.inf)         # one ulp, last row
        ok, det = bench.dropin_verify(cpl, gcm, ens, "cpu rehearsal")
        assert not ok and any(f.startswith("k3 f_T") for f in det["failures"]) and not any("f_U" in f for f in det["failures"])
    finally:
        spcpl.set_engine(None)


def test_dropin_verify_reports_a_forcing_that_arrived_wrong(monkeypatch):
    try:
        cpl, gcm, ens = _stepped()
        orig = spcpl.forward_batched

        def torn(*a, **kw):                      # a download that lands wrong: one element of f_v, first row
            host = orig(*a, **kw)
            host["f_v"][0, 0] += 1e-9
            return host
        monkeypatch.setattr(spcpl, "forward_batched", torn)
        ok, det = bench.dropin_verify(cpl, gcm, ens, "cpu rehearsal")
        assert not ok and any(f.startswith("k1 f_v") for f in det["failures"]) and not any("f_u" in f for f in det["failures"])
    finally:
        spcpl.set_engine(None)
