"""-m gpu: walk the dispatch table of the C ABI and bit-check EVERY kernel instantiation a default launch can reach.

Round-2 verdict: three times a dispatch change (a new template instantiation behind a size threshold) shipped without
a parity test that selects it.  This test cannot be forgotten: it asks the library itself, through
``spc_describe_launch`` (include/spc.h -- the text comes from the same ``choose_fwd`` / ``choose_bwd`` the launchers use),
which instantiation it would launch for every column count 1 ... 41 000 of each level geometry, and then launches, for
every distinct instantiation name, the smallest and the largest such batch (plus the size-class boundaries the verdict
lists) with NO environment override and compares all outputs with the plain-C oracle:

    K1 lean (what bench.py times)  f_u f_v f_qt f_ql ql_ref f_ps idx bit-exact; f_thl <= 8 ulp of thl x factor / dt
    K1 full (convert_profiles / cplsurf)  every optional output as tests/test_parity_gpu.check_forward
    K3 / K4   the seven tendencies (+ start_index) bit-exact, -0.0 and NaN positions included

Reference semantics: splib/spcpl.py:171-246, 299-385 (forward), 388-555 with 471-477 / 518-533 (backward), 479-489 +
splib/sputils.py:94-189 (conservative).  Columns are independent, so ONE oracle run over the largest batch of a
geometry serves every prefix of it.
"""
import ctypes

import numpy
import pytest
import torch

from sp_coupler_amd import _abi, synthetic
from tests import oracle_c
from tests.gpu_util import EPS, assert_bits, assert_close_scaled, host, to_dev
from tests.test_parity_gpu import check_backward, check_forward

pytestmark = pytest.mark.gpu

FACTOR, DT = 0.85, 900.0
N_MAX = 41000
FIXED = (1, 200, 256, 257, 300, 512, 513, 700, 1024, 1025, 4096, 16384, 25000, 25001, 40000)     # verdict, item 1
GEOMETRIES = [(91, 160, 0), (137, 512, 0), (19, 160, 0), (91, 160, 3), (60, 100, 0)]              # (nG, nL, pitch padding)
KINDS = {"k1_lean": (0, 1), "k1_full": (0, 3), "k3": (1, 0), "k4": (4, 0)}                        # name -> (pass, flags)


@pytest.fixture(scope="module")
def eng():
    from sp_coupler_amd.engine import Engine
    return Engine("cuda:0")


def _name(lib, n, nG, nL, pad, pass_, flags, elem=8):
    d = _abi.Dims(n, nG, nL, nG + pad, nG + 1 + pad, nL + pad, 1, 0)
    return _abi.describe_launch(lib, d, pass_, flags, elem).split()[0]


def _sweep(lib, nG, nL, pad, elem=8):
    """{kind: {instantiation name: [column counts]}} for n = 1 ... N_MAX"""
    out = {k: {} for k in KINDS}
    for n in range(1, N_MAX + 1):
        for kind, (pass_, flags) in KINDS.items():
            out[kind].setdefault(_name(lib, n, nG, nL, pad, pass_, flags, elem), []).append(n)
    return out


def _cases(sweep, kind):
    ns = {n for n in FIXED if n <= N_MAX}
    for name, where in sweep[kind].items():
        ns.add(where[0])
        ns.add(where[-1])
    return sorted(ns)


def _pad(t, pad):
    if pad == 0 or t.dim() != 2:
        return t
    buf = torch.full((t.shape[0], t.shape[1] + pad), float("nan"), device=t.device, dtype=t.dtype)
    buf[:, :t.shape[1]] = t
    return buf[:, :t.shape[1]]


def _prefix(d, n):
    return {k: v[:n] for k, v in d.items()}


@pytest.mark.parametrize("nG,nL,pad", GEOMETRIES)
def test_every_reachable_instantiation_is_launched_and_bit_checked(eng, nG, nL, pad):
    lib = eng.lib
    sweep = _sweep(lib, nG, nL, pad)
    # big batch of this geometry + ONE oracle run; every case below is a prefix of it
    n_k4 = 6000                                            # K4 has one instantiation per geometry: prefixes up to here
    gcm, zf, zh, prof = synthetic.make_batch_tiled(N_MAX, nG, nL, seed=7700 + nG + pad, base=2048)
    ref_f = oracle_c.forward(gcm, zf, zh, prof, FACTOR, DT, couple_surface=True)
    ref_b = oracle_c.backward(gcm, None, zf, prof, FACTOR, DT)
    thl_scale = numpy.abs(ref_f["thl"]).max()
    g = {k: _pad(v, pad) for k, v in to_dev(gcm, eng.device).items()}
    p = {k: _pad(v, pad) for k, v in to_dev(prof, eng.device).items()}
    zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
    sptr = ctypes.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
    seen = {k: set() for k in KINDS}

    # ---- lean K1 + K3: exactly the plans bench.py / the step path launch (Engine.plan_exchange, launch_raw) --------
    for n in sorted(set(_cases(sweep, "k1_lean")) | set(_cases(sweep, "k3"))):
        gn, pn = _prefix(g, n), _prefix(p, n)
        fp, bp = eng.plan_exchange(gn, zf_d, zh_d, pn, FACTOR, FACTOR, DT)
        for t in list(fp.outputs.values()) + list(bp.outputs.values()):
            t.fill_(float("nan")) if t.is_floating_point() else t.fill_(-7)
        seen["k1_lean"].add(_abi.describe_launch(lib, fp.dims, 0, 1).split()[0])
        seen["k3"].add(_abi.describe_launch(lib, bp.dims, 1, 0).split()[0])
        fp.launch_raw(sptr)
        bp.launch_raw(sptr)
        torch.cuda.synchronize()
        tag = "%d<->%d pad %d n=%d: " % (nG, nL, pad, n)
        F = {k: host(v) for k, v in fp.outputs.items()}
        assert_bits(tag + "idx", F["idx"], ref_f["idx"][:n])
        for k in ("f_u", "f_v", "f_qt", "f_ql", "ql_ref", "f_ps"):
            assert_bits(tag + k, F[k], ref_f[k][:n])
        assert_close_scaled(tag + "f_thl", F["f_thl"], ref_f["f_thl"][:n], 8 * EPS, thl_scale * abs(FACTOR) / DT)
        for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
            assert_bits(tag + k, host(bp.outputs[k]), ref_b[k][:n])

    # ---- full K1 (every optional output, surface coupling) + K3 with start_index and the Zf round trip -------------
    for n in _cases(sweep, "k1_full"):
        gn, pn = _prefix(g, n), _prefix(p, n)
        plan = eng.plan_forward(gn, zf_d, pn, FACTOR, DT, zh=zh_d, want_profiles=True, couple_surface=True)
        seen["k1_full"].add(_abi.describe_launch(lib, plan.dims, 0, 3).split()[0])
        fwd = plan.launch()
        bwd = eng.backward(gn, zf_d, pn, FACTOR, DT, Zf=fwd["Zf"])
        torch.cuda.synchronize()
        check_forward({k: host(v) for k, v in fwd.items()}, {k: v[:n] for k, v in ref_f.items()}, thl_scale)
        check_backward({k: host(v) for k, v in bwd.items()}, {k: v[:n] for k, v in ref_b.items()})

    # ---- K4 (conservative coarsening): one instantiation per geometry ------------------------------------------------
    rng = numpy.random.default_rng(5)
    rho = numpy.ascontiguousarray(1.2 * numpy.exp(-zf / 8000.0)[None, :] * rng.uniform(0.9, 1.1, (n_k4, nL)))
    prof4 = dict(_prefix(prof, n_k4), Rhobf=rho)
    ref_c = oracle_c.backward(_prefix(gcm, n_k4), None, zf, prof4, FACTOR, DT, conservative=True, zh=zh)
    p4 = dict(_prefix(p, n_k4), Rhobf=_pad(torch.from_numpy(rho).to(eng.device), pad))
    for n in sorted({n for n in _cases(sweep, "k4") if n <= n_k4} | {n_k4}):
        pn = _prefix(p4, n)
        plan = eng.plan_backward(_prefix(g, n), zf_d, pn, FACTOR, DT, Zf=None, conservative=True, zh=zh_d)
        seen["k4"].add(_abi.describe_launch(lib, plan.dims, 4, 0).split()[0])
        out = plan.launch()
        torch.cuda.synchronize()
        check_backward({k: host(v) for k, v in out.items()}, {k: v[:n] for k, v in ref_c.items()})

    # ---- the point of the exercise: nothing the library can pick for this geometry went unchecked -----------------------
    for kind in KINDS:
        reachable = set(sweep[kind])
        assert seen[kind] == reachable, "%s: not launched %s; launched but not in the sweep %s" % (
            kind, sorted(reachable - seen[kind]), sorted(seen[kind] - reachable))


def test_fp32_instantiations_agree_with_each_other_and_with_the_fp64_oracle(eng):
    """The float instantiations of the same templates (config 5's tolerance sweep): every name the float dispatch can
    reach for 91 <-> 160 and 137 <-> 512 is launched; the run-time-geometry single-column-slab launch is held to the
    fp64 oracle within fp32 tolerances, and every other instantiation must reproduce ITS bits (same operation order)."""
    from sp_coupler_amd.engine import Engine
    e32 = Engine("cuda:0", dtype=torch.float32)
    lib = e32.lib
    for nG, nL in ((91, 160), (137, 512)):
        sweep = _sweep(lib, nG, nL, 0, elem=4)
        n_max = 30000
        gcm, zf, zh, prof = synthetic.make_batch_tiled(n_max, nG, nL, seed=990 + nG, base=2048)
        f32 = lambda d: {k: torch.from_numpy(v).to(e32.device, torch.float32) for k, v in d.items()}      # noqa: E731
        g, p = f32(gcm), f32(prof)
        zf_d, zh_d = (torch.from_numpy(z).to(e32.device, torch.float32) for z in (zf, zh))
        # baseline: padded pitch -> run-time geometry, one column per workgroup
        gb, pb = {k: _pad(v, 1) for k, v in g.items()}, {k: _pad(v, 1) for k, v in p.items()}
        fb, bb = e32.plan_exchange(gb, zf_d, zh_d, pb, FACTOR, FACTOR, DT, cols_per_block=1)
        fb.launch()
        bb.launch()
        torch.cuda.synchronize()
        base = {k: host(v) for k, v in list(fb.outputs.items()) + list(bb.outputs.items())}
        m = 512
        ref_f = oracle_c.forward(_prefix(gcm, m), zf, zh, _prefix(prof, m), FACTOR, DT, couple_surface=False)
        ref_b = oracle_c.backward(_prefix(gcm, m), None, zf, _prefix(prof, m), FACTOR, DT)
        for k, ref in (("f_u", ref_f), ("f_qt", ref_f), ("f_thl", ref_f), ("f_T", ref_b), ("f_U", ref_b)):
            assert numpy.abs(base[k][:m].astype(numpy.float64) - ref[k]).max() <= 2e-3 * numpy.abs(ref[k]).max(), k
        seen = {"k1_lean": set(), "k3": set()}
        ns = {n for n in (set(_cases(sweep, "k1_lean")) | set(_cases(sweep, "k3"))) if n <= n_max}
        for kind in seen:                       # names whose whole range lies above n_max: none expected, but be explicit
            for name, where in sweep[kind].items():
                assert where[0] <= n_max, (name, where[0])
        for n in sorted(ns):
            fp, bp = e32.plan_exchange(_prefix(g, n), zf_d, zh_d, _prefix(p, n), FACTOR, FACTOR, DT)
            seen["k1_lean"].add(_abi.describe_launch(lib, fp.dims, 0, 1, 4).split()[0])
            seen["k3"].add(_abi.describe_launch(lib, bp.dims, 1, 0, 4).split()[0])
            fp.launch()
            bp.launch()
            torch.cuda.synchronize()
            for k, v in list(fp.outputs.items()) + list(bp.outputs.items()):
                assert_bits("f32 %d<->%d n=%d %s" % (nG, nL, n, k), host(v), base[k][:n])
        for kind in seen:
            want = {nm for nm, where in sweep[kind].items() if where[0] <= n_max}
            assert seen[kind] == want, (kind, sorted(want - seen[kind]))
