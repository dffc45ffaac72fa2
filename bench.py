#!/usr/bin/env python3
"""bench.py -- SP column-exchanges/s of the batched coupling step on MI355X.

One *step* = one pass of the hot path over one batch of synthetic columns:
    K1 forward (convert + interpolate + LES forcings, fused K2 cloud-fraction index map)  ->  K3 backward
    (interpolate back + GCM tendencies), i.e. one *column-exchange* per column.

Workloads (BASELINE.json `configs`; the metric names no config, so N=1 runs the largest single-GPU one):
    N = 1   config 3: T159 full-SP, 35 718 columns, 91 GCM <-> 160 LES levels, fp64, all on one MI355X.
            Extra key `small_batch`: config 2 (1024 columns), the launch-latency-bound case, same measurements.
    N > 1   config 4: T511 full-SP, 348 528 columns, column-sharded over the N ranks (contiguous blocks of
            ceil(n/N) rows, sharding.shard_range) -- STRONG scaling of the stated workload; no data-path
            collective (columns are independent); RCCL only for the barrier and the max-over-ranks of the time.
Inputs are resident in HBM before the timed region; ROTATE distinct batches are cycled (one config-3 batch is
1.6 GB of live arrays, already 6x the 256 MB Infinity Cache; config 2 rotates 8 x 44 MB).

Byte model (LES grid SHARED by all columns as bench.py packs it -> zf/zh are not per-column
traffic): K1+K2 (9 nG + 5 nL + 3) reads + (6 nL + 1) writes fp64 + nG int32; K3 (9 nG + 6 nL) reads + 7 nG writes.
91<->160: 21 028 B + 19 328 B = 40 356 B per column-exchange.

Prints ONE JSON line on rank 0 (contract: see the task statement) with the extra objects `roofline`,
`cpu_baseline`, `small_batch`, and `verified` (outputs of batch 0 after the timed region, bit-compared with the
CPU oracle's outputs that the cpu_baseline leg produced).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy)
KERNEL_LAUNCHES = 400   # back-to-back launches per per-kernel HIP-event measurement (independent of --steps)
HEAT_MS = 40.0          # GPU work in front of every timed loop (clock ramp after idle: Workload.heat)


def algorithmic_bytes(nG, nL, esize=8, shared_grid=True):
    """SURVEY.md section 8(d), per column-exchange; with a shared LES grid the 2 nL (fwd: zf, zh) and nL (bwd: zf)
    grid reads are per-launch constants, not per-column traffic, and are subtracted as 8(d) says."""
    fwd_r = (9 * nG + 6 * nL + 3) - (nL if shared_grid else 0)
    idx_r = 0 if shared_grid else nL
    bwd_r = (9 * nG + 7 * nL) - (nL if shared_grid else 0)
    fwd = fwd_r * esize + (6 * nL + 1) * esize
    idx = idx_r * esize + nG * 4
    bwd = bwd_r * esize + 7 * nG * esize
    return {"forward": fwd, "index": idx, "backward": bwd, "k1_launch": fwd + idx, "k3_launch": bwd,
            "exchange": fwd + idx + bwd}


# ---------------------------------------------------------------------------------------------
# cpu_baseline: the reference's algorithm on the host (oracle/spcpl_oracle.py: serial Python loop over columns,
# one numpy.interp per profile).  Its outputs double as the checker of `verified`.
# ---------------------------------------------------------------------------------------------
def cpu_baseline(gcm, zf, zh, prof, dt, factor, budget_s):
    from oracle import spcpl_oracle as orc
    n = gcm["T"].shape[0]
    lean = {k: v for k, v in prof.items() if k not in ("Rain", "rain_last")}
    done, t0 = 0, time.perf_counter()
    while True:
        f = orc.forward_batched(gcm, lean, zf, zh, factor, dt, couple_surface=False)
        b = orc.backward_batched(gcm, f["Zf"], prof, zf, factor, dt)
        done += n
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    res = {"value": done / el, "unit": "column-exchanges/s", "cores": 1, "kind": "port",
           "sample": "%d column-exchanges (%d passes over the first %d columns of batch 0, NumPy per-column "
                     "loop, %.1f s)" % (done, done // n, n, el)}
    return res, f, b


def cpu_worker_main(path, budget):
    """`python bench.py --cpu-worker PATH BUDGET`: one worker of the all-cores leg -- whole passes over
    its copy of a small batch for BUDGET seconds; prints "<done> <seconds>". Never touches the GPU."""
    import numpy
    from oracle import spcpl_oracle as orc
    z = numpy.load(path)
    gcm = {k[2:]: z[k] for k in z.files if k.startswith("g_")}
    prof = {k[2:]: z[k] for k in z.files if k.startswith("p_")}
    zf, zh = z["zf"], z["zh"]
    n = gcm["T"].shape[0]
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        f = orc.forward_batched(gcm, prof, zf, zh, 1.0, 900.0)
        orc.backward_batched(gcm, f["Zf"], prof, zf, 1.0, 900.0)
        done += n
    print(done, time.perf_counter() - t0)


def cpu_baseline_multicore(gcm, zf, zh, prof, budget_s, workers):
    """The same per-column NumPy loop in `workers` independent CHILD PROCESSES (plain subprocesses with a hard
    timeout -- no pool that could respawn; they never touch the GPU)."""
    import subprocess
    import tempfile
    import numpy
    sub = 128                                   # a small batch per pass keeps every worker inside the budget
    d = tempfile.mkdtemp(prefix="spc_cpu_")
    path = os.path.join(d, "batch.npz")
    numpy.savez(path, zf=zf, zh=zh, **{"g_" + k: v[:sub] for k, v in gcm.items()},
                **{"p_" + k: v[:sub] for k, v in prof.items()})
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(budget_s)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
             for _ in range(workers)]
    res = []
    try:
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=budget_s + 60)
                a, b = out.split()
                res.append((int(a), float(b)))
            except Exception:
                pr.kill()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        try:
            os.remove(path)
            os.rmdir(d)
        except OSError:
            pass
    if not res:
        raise RuntimeError("no cpu worker finished")
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": total / wall, "unit": "column-exchanges/s", "cores": len(res), "kind": "port",
            "sample": "%d column-exchanges in %d processes x %.1f s (NumPy per-column loop)" % (total, len(res), wall)}


def compare_with_oracle(F, B, ref_f, ref_b, factor, dt):
    """HIP outputs (dicts of host arrays) against oracle outputs of the same rows: index map and everything not
    downstream of pow() bit-exact (incl. the sign of zero and NaN positions); f_thl within 8 ulp of thl's scale / dt.
    Returns (list of failures, f_thl statistics)."""
    import numpy
    eps = 2.220446049250313e-16
    bad = []

    def bits(name, got, want):
        if got.shape != want.shape:
            bad.append("%s: shape %s vs %s" % (name, got.shape, want.shape))
            return
        same = (got == want) | (numpy.isnan(got) & numpy.isnan(want))
        if not same.all() or not numpy.array_equal(numpy.signbit(got)[~numpy.isnan(want)], numpy.signbit(want)[~numpy.isnan(want)]):
            bad.append("%s: %d elements differ" % (name, int((~same).sum())))
    if "idx" in F:
        bits("idx", F["idx"], ref_f["idx"].astype(F["idx"].dtype))
    for k in ("f_u", "f_v", "f_qt", "f_ql", "ql_ref", "f_ps"):
        bits(k, F[k], ref_f[k])
    for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A"):
        bits(k, B[k], ref_b[k])
    err = float(numpy.abs(F["f_thl"] - ref_f["f_thl"]).max())
    bound = 8 * eps * float(numpy.abs(ref_f["thl"]).max()) * abs(factor) / dt
    if not err <= bound:
        bad.append("f_thl: max abs err %.3e > %.3e" % (err, bound))
    rel = err / float(numpy.abs(ref_f["f_thl"]).max())
    # element-wise view of the same comparison: f_thl = factor (thl - thl_d) / dt cancels, so an element whose forcing is
    # tiny carries the absolute error of thl (<= 8 ulp of ~300 K, / dt) on a small value; the two CPU oracles (NumPy / C)
    # differ from each other by 4 ulp of thl (tests/test_oracle.py), so no tighter element-wise bar is definable
    d = numpy.abs(F["f_thl"] - ref_f["f_thl"])
    a = numpy.abs(ref_f["f_thl"])
    nz = a > 0
    erel = d[nz] / a[nz]
    ew = {"max_abs_err": err, "abs_err_bound_8ulp_thl_over_dt": bound,
          "max_rel_err_over_all_nonzero_elements": float(erel.max()) if erel.size else 0.0,
          "fraction_of_elements_above_1e-10_relative": float((erel > 1e-10).mean()) if erel.size else 0.0,
          "smallest_abs_f_thl_among_those": float(a[nz][erel > 1e-10].min()) if (erel > 1e-10).any() else None,
          "max_rel_err_where_abs_f_thl_ge_1e-5": float((d[a >= 1e-5] / a[a >= 1e-5]).max()) if (a >= 1e-5).any() else None}
    return bad, {"f_thl_max_rel_err": rel, "f_thl_elementwise": ew}


BIT_EXACT = "idx,f_u,f_v,f_qt,f_ql,ql_ref,f_ps,f_T,f_SH,f_QL,f_QI,f_U,f_V,f_A"


def verify(fplan, bplan, ref_f, ref_b, n_ref, factor, dt):
    """Outputs of batch 0 (left in HBM by the timed launches) against the oracle outputs of the cpu_baseline
    leg (the first ``n_ref`` columns)."""
    F = {k: v.cpu().numpy()[:n_ref] for k, v in fplan.outputs.items()}
    B = {k: v.cpu().numpy()[:n_ref] for k, v in bplan.outputs.items()}
    bad, det = compare_with_oracle(F, B, ref_f, ref_b, factor, dt)
    return (not bad), dict(det, columns_checked=int(n_ref), bit_exact=BIT_EXACT, failures=bad)


def pow_ulp_histogram(eng, pf):
    """iexner(Pf) = (Pf / pref0) ** (-rd / cp) (sputils.py:33-34), the one transcendental of the path, three ways: the HIP
    kernel's own spc_pow, NumPy's `**` (its SIMD pow: what the reference runs) and glibc's pow (what the plain-C oracle
    runs).  Distance in units in the last place, as a histogram.  The two CPU results differ from EACH OTHER on a few
    percent of the points, so bit-parity of thl is not defined by the reference itself; the kernel's pow is within
    0.56 ulp of the exact value (tools/csrc/pow_accuracy.c), i.e. at most 1 ulp from either."""
    import math
    import numpy
    import torch
    x = numpy.ascontiguousarray(pf, dtype=numpy.float64)
    got = eng.exner(torch.from_numpy(x).to(eng.device), inverse=True).cpu().numpy()
    y = (-287.04) / 1004.
    np_pow = (x / 1e5) ** y
    libm = numpy.fromiter((math.pow(v, y) for v in (x / 1e5).ravel()), dtype=numpy.float64, count=x.size).reshape(x.shape)

    def hist(a, b):
        d = numpy.abs(a.view(numpy.int64) - b.view(numpy.int64))
        return {"0": float((d == 0).mean()), "1": float((d == 1).mean()), "2": float((d == 2).mean()), ">2": float((d > 2).mean()),
                "max": int(d.max())}
    return {"points": int(x.size), "hip_vs_numpy_pow": hist(got, np_pow), "hip_vs_glibc_pow": hist(got, libm),
            "numpy_pow_vs_glibc_pow": hist(np_pow, libm), "unit": "fraction of points at that distance (ulp)"}


def sample_rows(n, m=4096):
    """row numbers of a check sample of (at most) m rows of an n-row block: its first and last m/3 rows and m/3 rows
    spread evenly over the rest -- always including row 0 and row n - 1"""
    import numpy
    if n <= m:
        return numpy.arange(n, dtype=numpy.int64)
    a = m // 3
    mid = numpy.linspace(a, n - a - 1, m - 2 * a).astype(numpy.int64)
    return numpy.unique(numpy.concatenate([numpy.arange(a), mid, numpy.arange(n - a, n)]))


# fp32 arithmetic variant (BASELINE config 5's tolerance sweep): outputs against the fp64 oracle ON THE SAME (rounded) INPUTS.
# A forcing / tendency is factor (x_interpolated - x_model) / dt, a difference of nearly equal numbers, so its fp32 error is
# set by the PROFILE's scale: the statistic is max |err| dt / (|factor| max |profile|).  Bars = 5-10x what
# tests/test_fullsize_gpu.py::test_fp32_vs_fp64_tolerance_sweep_config5 measures.
F32_TOL = {"f_u": 1e-4, "f_v": 1e-4, "f_thl": 1e-5, "f_qt": 1e-4, "f_ql": 5e-4, "ql_ref": 5e-4, "f_ps": 1e-5,
           "f_T": 1e-5, "f_SH": 1e-4, "f_QL": 5e-4, "f_QI": 5e-4, "f_U": 1e-4, "f_V": 1e-4, "f_A": 1e-5}
F32_IDX_MISMATCH_MAX = 0.01       # half levels within fp32 rounding of an LES half level land one cell off
F32_MASK_MISMATCH_MAX = 1e-3      # GCM levels within fp32 rounding of the LES top: masked in one arithmetic, not the other


def compare_with_oracle_f32(F, B, ref_f, ref_b, scales, factor, dt):
    """fp32 outputs against fp64 oracle outputs of the same rows; ``scales``: max |profile| per output name.
    Returns (failures, detail)."""
    import numpy
    bad, det = [], {}
    for name, got, want in [(k, F[k], ref_f[k]) for k in ("f_u", "f_v", "f_thl", "f_qt", "f_ql", "ql_ref", "f_ps")] + \
                           [(k, B[k], ref_b[k]) for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A")]:
        got = got.astype(numpy.float64)
        live = ~(numpy.isnan(want) | numpy.isnan(got))
        # masked above the LES top in one arithmetic only (f[0:start_index] *= 0, spcpl.py:527-533): counted, not measured
        off = live & ((want == 0) != (got == 0)) if name in B else numpy.zeros_like(live)
        nanoff = numpy.isnan(want) != numpy.isnan(got)
        mm = float((off | nanoff).mean())
        use = live & ~off
        unit = 1.0 if name == "ql_ref" else abs(factor) / dt
        err = float(numpy.abs(got[use] - want[use]).max() / (scales[name] * unit)) if use.any() else 0.0
        det[name] = {"max_err_over_profile_scale": err, "mask_mismatch_fraction": mm}
        if not err <= F32_TOL[name]:
            bad.append("%s: fp32 error %.3e of the profile scale > %.1e" % (name, err, F32_TOL[name]))
        if mm > F32_MASK_MISMATCH_MAX:
            bad.append("%s: %.2e of the elements masked / NaN in one arithmetic only" % (name, mm))
    if "idx" in F:
        mis = float((F["idx"] != ref_f["idx"].astype(F["idx"].dtype)).mean())
        det["idx_mismatch_fraction"] = mis
        if mis > F32_IDX_MISMATCH_MAX:
            bad.append("idx: %.3e of the level indices differ" % mis)
    return bad, det


def f32_scales(gs, ps, ref_f):
    import numpy
    mx = lambda a: float(numpy.abs(a[numpy.isfinite(a)]).max())                       # noqa: E731
    return {"f_u": mx(ref_f["u"]), "f_v": mx(ref_f["v"]), "f_thl": mx(ref_f["thl"]), "f_qt": mx(ref_f["qt"]),
            "f_ql": max(mx(ref_f["ql_ref"]), 1e-4), "ql_ref": max(mx(ref_f["ql_ref"]), 1e-4), "f_ps": mx(ps["PS"]),
            "f_T": mx(ps["T"]), "f_SH": mx(ps["QT"]), "f_QL": max(mx(ps["QL"]), 1e-4), "f_QI": max(mx(ps["QL"]), 1e-4),
            "f_U": mx(ps["U"]), "f_V": mx(ps["V"]), "f_A": 1.0}


def sample_check(fout, bout, g, p, zf, zh, factor, dt, m=4096):
    """A sample of the rows of one device's block (``sample_rows``): the inputs the kernels read are fetched FROM THE
    DEVICE, run through the plain-C oracle (tests/oracle_c.py: oracle/spc_oracle.c, the independent restatement with the
    ABI's own argument structs) on the host, and compared with the outputs the timed plans left in HBM.  ``fout`` /
    ``bout``: output dicts of the forward / backward plan; ``g`` / ``p``: the device tensors the plans were built on.
    Used where no CPU copy of the batch exists: the ranks of an N > 1 run, the devices of in_process_all_gpus, the
    `config5` / `per_column_grid` legs.  float64 tensors: bit comparison (f_thl within 8 ulp of thl / dt); float32
    tensors: the fp64 oracle on the SAME rounded inputs, tolerances F32_TOL.  zf / zh: [nL] or one row per column."""
    import numpy
    import torch
    from tests import oracle_c
    n = int(g["T"].shape[0])
    f32 = g["T"].dtype == torch.float32
    rows = sample_rows(n, m)
    rt = torch.from_numpy(rows).to(g["T"].device)
    take = lambda t: numpy.ascontiguousarray(t.index_select(0, rt).cpu().numpy())       # noqa: E731
    take64 = lambda t: numpy.ascontiguousarray(take(t), dtype=numpy.float64)             # noqa: E731
    gs = {k: take64(v) for k, v in g.items() if k in ("U", "V", "T", "SH", "QL", "QI", "Pfull", "Phalf", "A", "Zgfull", "Zghalf")}
    ps = {k: take64(v) for k, v in p.items() if k in ("U", "V", "THL", "QT", "QL", "QL_ice", "T", "PS", "A")}
    ps["Rain"], ps["rain_last"] = numpy.zeros(len(rows)), numpy.zeros(len(rows))
    grid = lambda z: take64(z) if z.dim() == 2 else numpy.ascontiguousarray(z.cpu().numpy(), dtype=numpy.float64)   # noqa: E731
    zfn, zhn = grid(zf), grid(zh)
    ref_f = oracle_c.forward(gs, zfn, zhn, ps, factor, dt, couple_surface=False)
    ref_b = oracle_c.backward(gs, None, zfn, ps, factor, dt)
    F = {k: take(v) for k, v in fout.items()}
    B = {k: take(v) for k, v in bout.items()}
    res = {"rows_checked": int(len(rows)), "first_row": int(rows[0]), "last_row": int(rows[-1]), "rows_in_block": n,
           "oracle": "oracle/spc_oracle.c (plain C, glibc pow, fp64) on inputs read back from the device"}
    if f32:
        bad, det = compare_with_oracle_f32(F, B, ref_f, ref_b, f32_scales(gs, ps, ref_f), factor, dt)
        res.update({"fp32_vs_fp64_oracle": det, "tolerances": F32_TOL, "failures": bad})
    else:
        bad, det = compare_with_oracle(F, B, ref_f, ref_b, factor, dt)
        res.update({"f_thl_max_rel_err": det["f_thl_max_rel_err"], "failures": bad})
    return (not bad), res


def device_identity(index):
    """what a reader needs to tell N distinct GPUs from one GPU used N times"""
    import torch
    pr = torch.cuda.get_device_properties(index)
    return {"index": int(index), "name": pr.name, "arch": getattr(pr, "gcnArchName", None),
            "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
            "uuid": str(getattr(pr, "uuid", "")), "cus": int(pr.multi_processor_count), "hbm_GiB": round(pr.total_memory / 2 ** 30, 1)}


def _bits_equal(a, b):
    import numpy
    a, b = numpy.asarray(a), numpy.asarray(b)
    return a.shape == b.shape and bool(((a == b) | ((a != a) & (b != b))).all())


def dropin_verify(cpl, gcm, ens, label):
    """Is what the model objects RECEIVED through the drop-in step (after the uploads, the launches and the downloads of
    whatever transport that batch size takes) what the kernels compute from what the models HANDED OVER?  Called straight
    after the last timed Coupler.step of a batched-protocol leg:
      uploads   the device copy of every array the GCM / LES stand-ins wrote into the pinned buffers == that buffer;
      K3        the seven tendencies the GCM stand-in received == a synchronous recompute (Engine.backward on the same
                device inputs, fresh output tensors, default stream, waited for) == the NumPy oracle on a row sample;
      K1        one more (untimed) step: the forcings the LES ensemble receives == a synchronous Engine.forward on a
                device snapshot of the slab means that step's K1 reads + that step's GCM state, == the oracle on a sample.
    Bit for bit except f_thl against the oracle (8 ulp of thl / dt, as `verified`).  Returns (ok, detail)."""
    import numpy
    import torch
    from oracle import spcpl_oracle as orc
    from sp_coupler_amd import spcpl
    from sp_coupler_amd.transfer import Sharded
    b = spcpl.current_batch()
    buf, eng, n = b.buf, b.engine, b.n
    dt = float(gcm.get_timestep())
    lf, gf = float(cpl.les_forcing_factor), float(cpl.gcm_forcing_factor)
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
    sync()
    bad, checked = [], []
    host = spcpl._to_host
    rows = sample_rows(n, 192)
    zf_h, zh_h = numpy.asarray(b.zf_host), numpy.asarray(b.zh_host)

    def uploads(arena, names, what):
        for k in names:
            if not _bits_equal(host(arena.d[k])[:n], arena.hn[k][:n]):
                bad.append("%s %s: device copy differs from the pinned host array" % (what, k))
        checked.append("%s: %d arrays device == host" % (what, len(names)))
    with eng.on_stream():
        uploads(buf.gcm_in, spcpl.gcm_vars, "h2d_gcm")
        uploads(buf.les_in, spcpl._BWD_KEYS + ("THL", "PS"), "h2d_les")
        # K3 of the last timed step
        r3 = eng.backward(b.gcm, b.zf, {k: buf.les_in.d[k] for k in spcpl._BWD_KEYS}, gf, dt, Zf=None)
        got3 = {}
        for var in spcpl._TEND_VARS:
            rec = gcm.tendencies.get(var)
            got3[var] = None if not isinstance(rec, tuple) else rec[1]
            if got3[var] is None or not _bits_equal(got3[var], host(r3["f_" + var])):
                bad.append("k3 f_%s: what the GCM received differs from the synchronous recompute" % var)
        checked.append("k3: 7 tendencies x %d columns received == recompute" % n)
    g_h = {v: numpy.ascontiguousarray(buf.gcm_in.hn[v][:n][rows]) for v in spcpl.gcm_vars}
    p_h = {k: numpy.ascontiguousarray(buf.les_in.hn[k][rows]) for k in spcpl._BWD_KEYS + ("THL",)}
    Zf = (g_h["Zgfull"] - g_h["Zghalf"][:, -1:]) / orc.grav
    ref_b = orc.backward_batched(g_h, Zf, p_h, zf_h, gf, dt)
    for var in spcpl._TEND_VARS:
        if got3[var] is not None and not _bits_equal(got3[var][rows], ref_b["f_" + var]):
            bad.append("k3 f_%s: differs from the oracle on the row sample" % var)
    checked.append("k3: oracle on %d sample rows (first %d ... last %d)" % (len(rows), rows[0], rows[-1]))
    # K1: the slab means the NEXT step's K1 reads are on the device now
    fkeys = ("U", "V", "THL", "QT", "QL", "PS")
    with eng.on_stream():
        snap = {}
        for k in fkeys:
            t = buf.les_in.d[k]
            snap[k] = Sharded([p_.clone() for p_ in t.parts], t.bounds) if isinstance(t, Sharded) else t.clone()
        snap_h = {k: host(v) for k, v in snap.items()}
    received = {}
    orig = ens.set_forcings_batched

    def tap(**arrays):
        for k, v in arrays.items():
            received[k] = numpy.array(v)
        return orig(**arrays)
    ens.set_forcings_batched = tap
    try:
        cpl.step()
    finally:
        ens.set_forcings_batched = orig
    sync()
    pairs = (("U", "f_u"), ("V", "f_v"), ("THL", "f_thl"), ("QT", "f_qt"), ("SP", "f_ps"), ("QL", "f_ql"), ("QLp", "ql_ref"))
    with eng.on_stream():
        r1 = eng.forward(b.gcm, b.zf, snap, lf, dt, zh=b.zh, want_profiles=False, want_heights=False)
        for key, name in pairs:
            if key not in received or not _bits_equal(received[key], host(r1[name])):
                bad.append("k1 %s: what the LES ensemble received differs from the synchronous recompute" % name)
        checked.append("k1: 7 forcing arrays x %d columns received == recompute" % n)
    g_h = {v: numpy.ascontiguousarray(buf.gcm_in.hn[v][:n][rows]) for v in spcpl.gcm_vars}
    ref_f = orc.forward_batched(g_h, {k: numpy.ascontiguousarray(v[rows]) for k, v in snap_h.items()}, zf_h, zh_h, lf, dt)
    for key, name in pairs:
        if key not in received:
            continue
        if name == "f_thl":
            err = float(numpy.abs(received[key][rows] - ref_f[name]).max())
            bound = 8 * 2.220446049250313e-16 * float(numpy.abs(ref_f["thl"]).max()) * abs(lf) / dt
            if not err <= bound:
                bad.append("k1 f_thl: %.3e from the oracle on the row sample (bound %.3e)" % (err, bound))
        elif not _bits_equal(received[key][rows], ref_f[name]):
            bad.append("k1 %s: differs from the oracle on the row sample" % name)
    checked.append("k1: oracle on %d sample rows" % len(rows))
    return (not bad), {"leg": label, "columns": n, "transport": "per-array copies on copy streams" if buf.piecewise else "one copy per buffer",
                       "arena": type(buf.gcm_in).__name__, "checked": checked, "failures": bad}


def dropin_rate(eng, n_les=1024, steps=30, warmup=5, per_les_steps=5):
    """Column-exchanges/s THROUGH THE DROP-IN API: driver.Coupler.step (gather -> set_les_forcings -> LES -> profiles
    -> set_gcm_tendencies) on the in-process synthetic GCM/LES pair, host buffers in, host buffers out every step
    (PCIe both ways), with the time spent inside the model objects' own methods subtracted.  Two transports:
    the optional batched model protocol (one call per variable for all columns) and the reference's per-LES calls.
    `breakdown`: a second pass of the batched protocol with every copy and launch bracketed by HIP events
    (transfer.StepTrace): bytes and GB/s per PCIe copy, kernel time, and what is left for the host."""
    import torch
    from sp_coupler_amd import models, spcpl, transfer
    from sp_coupler_amd.driver import Coupler
    spcpl.set_engine(None)                   # the engine spcpl.get_engine() picks by itself: what a drop-in user gets

    def engine_keys():
        e = spcpl.get_engine()
        return {"engine": type(e).__name__, "devices": [str(x.device) for x in getattr(e, "engines", [e])]}
    out = dict({"n_cols": n_les, "levels": "91<->160", "unit": "column-exchanges/s",
                "note": "wall time of Coupler.step minus time inside model methods; includes H2D/D2H of every step"}, **engine_keys())
    gcm, ens = models.make_batched_models(n_les, nG=91, nL=160, seed=3)
    cpl = Coupler(gcm, ens)
    import gc
    gc.collect()                             # the big workloads above have just been dropped: not inside the timed passes
    for _ in range(2 * warmup):
        cpl.step()
    passes = []
    for _ in range(3):                       # three passes of `steps` steps; the median is reported, all three are listed
        torch.cuda.synchronize()
        models.model_seconds = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            cpl.step()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        passes.append(((wall - models.model_seconds) / steps * 1e3, models.model_seconds / steps * 1e3))
    ms_c, ms_m = sorted(passes)[1]
    out["batched_protocol"] = {"value": n_les / (ms_c * 1e-3), "steps": steps, "passes": 3,
                               "ms_per_step_coupler": ms_c, "ms_per_step_models": ms_m,
                               "ms_per_step_coupler_all_passes": [p_[0] for p_ in passes]}
    ok_small, det_small = dropin_verify(cpl, gcm, ens, "batched_protocol (%d columns)" % n_les)
    # the same steps again with HIP events around every copy / launch (the events cost a little host time themselves)
    transfer.trace = tr = transfer.StepTrace()
    models.model_seconds = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        cpl.step()
    torch.cuda.synchronize()
    wall_t = time.perf_counter() - t0
    transfer.trace = None
    summ = tr.summary()
    per = {k: {"ms_per_step": v["ms"] / steps, "bytes_per_step": v["bytes"] // steps, "GBs": v["GBs"]} for k, v in summ.items()}
    dev_ms = sum(v["ms"] for v in summ.values()) / steps
    coupler_ms = (wall_t - models.model_seconds) / steps * 1e3
    h2d = [v for k, v in summ.items() if k.startswith("h2d")]
    d2h = [v for k, v in summ.items() if k.startswith("d2h")]
    out["breakdown"] = {"per_step": per, "ms_per_step_coupler_traced": coupler_ms, "ms_per_step_on_device_or_wire": dev_ms,
                        "note": "every array crosses PCIe on its buffer's copy stream while the model fetches / takes the next one: "
                                "the copies overlap the model calls (whose time is subtracted), so the coupler's share can be less "
                                "than the time on the wire",
                        "h2d_GBs": sum(v["bytes"] for v in h2d) / max(sum(v["ms"] for v in h2d), 1e-9) / 1e6,
                        "d2h_GBs": sum(v["bytes"] for v in d2h) / max(sum(v["ms"] for v in d2h), 1e-9) / 1e6,
                        "pcie_bytes_per_step": sum(v["bytes"] for v in h2d + d2h) // steps,
                        "pcie_ms_per_step": sum(v["ms"] for v in h2d + d2h) / steps}
    # the reference's transport: ~35 getter / setter calls per column per step (spcpl.py:341-347, 535-542, 748-766)
    # through the UNCHANGED loop shape of splib.step (splib.py:317-332) -- per-LES spcpl calls on a list of LES objects
    gcm2, ens2 = models.make_batched_models(n_les, nG=91, nL=160, seed=3)
    gcm2.__class__ = models.SyntheticGCM
    cpl2 = Coupler(gcm2, [ens2[i] for i in range(n_les)])
    cpl2.step()
    cpl2.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(per_les_steps):
        cpl2.step()
    torch.cuda.synchronize()
    wall2 = time.perf_counter() - t0
    out["per_les_protocol"] = {"value": n_les * per_les_steps / wall2, "steps": per_les_steps,
                               "ms_per_step": wall2 / per_les_steps * 1e3,
                               "note": "`value`: wall time of the reference's loop shape (35 model calls per column and step) on the "
                                       "synthetic stand-in models, THEIR getters / setters included (NumPy row copies, request objects); "
                                       "`value_null_models`: the same loop on models whose methods cost nothing = the coupler's and "
                                       "driver's own per-column cost, PCIe and kernels included"}
    # the coupler's OWN per-column cost: the same loop on models whose methods cost nothing (no subtraction needed)
    gcm3 = models.NullTendencyGCM(n_les + 4, 91, 3)
    zf3, zh3 = ens2.zf_cache, ens2.zh_cache
    cpl3 = Coupler(gcm3, [models.NullLES(i + 1, zf3, zh3, 91) for i in range(n_les)])
    cpl3.step()
    cpl3.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(per_les_steps):
        cpl3.step()
    torch.cuda.synchronize()
    wall3 = time.perf_counter() - t0
    out["per_les_protocol"]["value_null_models"] = n_les * per_les_steps / wall3
    out["per_les_protocol"]["ms_per_step_null_models"] = wall3 / per_les_steps * 1e3
    # the same batched-protocol step at T159 size (config 3's 35 718 columns: 1.1 GB over PCIe per step)
    spcpl.set_engine(None)
    gcm4, ens4 = models.make_batched_models(35718, nG=91, nL=160, seed=5)
    cpl4 = Coupler(gcm4, ens4)
    for _ in range(2):
        cpl4.step()
    torch.cuda.synchronize()
    models.model_seconds = 0.0
    t0 = time.perf_counter()
    for _ in range(5):
        cpl4.step()
    torch.cuda.synchronize()
    w4 = time.perf_counter() - t0
    m4 = models.model_seconds
    ok_big, det_big = dropin_verify(cpl4, gcm4, ens4, "batched_protocol_35718_columns")
    big = {"value": 35718 * 5 / (w4 - m4), "ms_per_step_coupler": (w4 - m4) / 5 * 1e3,
           "ms_per_step_models": m4 / 5 * 1e3, "steps": 5, **engine_keys(),
           "note": "1.1 GB over PCIe per step, array by array on the transfer buffers' copy streams while the model objects fetch / "
                   "take the next variable: most of it is hidden behind the model calls (whose time is subtracted)"}
    del cpl4, gcm4, ens4
    out["batched_protocol_35718_columns"] = big
    out["verified"] = bool(ok_small and ok_big)
    out["verified_detail"] = [det_small, det_big]
    spcpl.set_engine(None)
    return out


def in_process_all_gpus(ids, factor, dt, args):
    """config 4 (348 528 columns) through multi.MultiDeviceEngine: ONE process, the column batch in row blocks on every
    listed device, one launch plan per device, launches issued device by device (what `spcpl.get_engine()` gives the
    reference's single master process on a multi-GPU node).  Same step and byte model as the headline."""
    import torch
    from sp_coupler_amd import synthetic
    from sp_coupler_amd.engine import Engine
    from sp_coupler_amd.multi import MultiDeviceEngine, describe_partition
    from sp_coupler_amd.transfer import Sharded
    n, nG, nL, seed = synthetic.CONFIGS[4]
    if args.cols:
        n = args.cols
    multi = MultiDeviceEngine([Engine("cuda:%d" % i) for i in ids], min_cols_per_device=1)
    b = multi.bounds_for(n)
    need = ("U", "V", "T", "SH", "QL", "QI", "Pfull", "Phalf", "A", "Zgfull", "Zghalf", "THL", "QT", "QL_ice", "PS")
    parts_g, parts_p, zfs, zhs = [], [], [], []
    for d, e in enumerate(multi.engines):           # each device generates its own rows in place (device-side tiling)
        rows = max(b[d + 1] - b[d], 1)
        g, zf_d, zh_d, p, _ = synthetic.make_batch_tiled_device(e.device, rows, nG, nL, seed=seed + 1000 * d,
                                                                 couple_surface=False, keys=need)
        cut = b[d + 1] - b[d]
        parts_g.append({k: v[:cut] for k, v in g.items()})
        parts_p.append({k: v[:cut] for k, v in p.items()})
        zfs.append(zf_d)
        zhs.append(zh_d)
    gs = {k: Sharded([pg[k] for pg in parts_g], b) for k in parts_g[0]}
    ps = {k: Sharded([pp[k] for pp in parts_p], b) for k in parts_p[0]}
    fp, bp = multi.plan_exchange(gs, Sharded(zfs), Sharded(zhs), ps, factor, factor, dt, cols_per_block=args.cols_per_block)

    def step():
        fp.launch()
        bp.launch()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < HEAT_MS:
        for _ in range(4):
            step()
        multi.synchronize()
    steps = max(20, min(args.steps, 100))
    for _ in range(max(args.warmup, 3)):
        step()
    multi.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    multi.synchronize()
    el = time.perf_counter() - t0
    # every device's block proves itself (sample_check: rows incl. the block's first and last, inputs read back from THAT
    # device, plain-C oracle): the outputs left in HBM by the timed launches
    checks = []
    for d, (pf_, pb_) in enumerate(zip(fp.plans, bp.plans)):
        if pf_ is None:
            continue
        try:
            ok_d, det_d = sample_check(pf_.outputs, pb_.outputs, parts_g[d], parts_p[d], zfs[d], zhs[d], factor, dt)
        except Exception as e:
            ok_d, det_d = False, {"failures": ["sample_check raised %r" % (e,)]}
        checks.append(dict(det_d, device="cuda:%d" % ids[d], rows=[int(b[d]), int(b[d + 1])], verified=bool(ok_d)))
    return {"workload": "config 4: %d synthetic SP columns, ONE process, row blocks on %d device(s) (multi.MultiDeviceEngine)" % (n, len(ids)),
            "devices": [device_identity(i) for i in ids], "distinct_devices": len(set(ids)), "partition": describe_partition(multi, n),
            "verified": all(c["verified"] for c in checks), "verified_detail": checks,
            "value": n * steps / el, "unit": "column-exchanges/s", "steps": steps, "ms_per_step": el / steps * 1e3,
            "note": "no collective, no peer traffic; compare with scaling_anchor (one GPU) and the `--gpus N` lines (one rank per GPU)"}


class Workload:
    """ROTATE batches of one configuration resident in HBM + the exchange plans bench.py times."""

    def __init__(self, eng, n_cols, nG, nL, seed, rotate, factor, dt, cols_per_block=0, keep_host=False, per_column_grid=False):
        from sp_coupler_amd import synthetic
        self.n_cols, self.nG, self.nL, self.rotate = n_cols, nG, nL, rotate
        self.esize = 8 if eng.dtype.itemsize == 8 else 4
        self.bytes = algorithmic_bytes(nG, nL, self.esize, shared_grid=not per_column_grid)
        self.fplans, self.bplans, self.host0 = [], [], None
        need = ("U", "V", "T", "SH", "QL", "QI", "Pfull", "Phalf", "A", "Zgfull", "Zghalf",      # what the two plans read
                "THL", "QT", "QL_ice", "PS")
        for r in range(rotate):
            # columns beyond the first 8192 are tiles of those with a per-tile perturbation, made ON THE DEVICE (same bits
            # as synthetic.make_batch_tiled on the host; tile 0 is the host batch `verified` checks)
            g, zf_d, zh_d, p, host = synthetic.make_batch_tiled_device(eng.device, n_cols, nG, nL, seed=seed + r,
                                                                        couple_surface=False, keys=need, dtype=eng.dtype,
                                                                        per_column_grid=per_column_grid)
            if r == 0:
                self.inputs0 = (g, zf_d, zh_d, p)      # device tensors of batch 0 (sample_check reads its rows back)
            if r == 0 and keep_host:
                self.host0 = host
            fp, bp = eng.plan_exchange(g, zf_d, zh_d, p, factor, factor, dt, cols_per_block=cols_per_block)
            self.fplans.append(fp)
            self.bplans.append(bp)

    def step(self, i, sptr):
        self.fplans[i % self.rotate].launch_raw(sptr)
        self.bplans[i % self.rotate].launch_raw(sptr)

    def heat(self, sptr, ms=HEAT_MS, plans=None):
        """Whole steps (or launches of `plans`) for `ms` of wall time: an MI355X that has idled for >= 50 ms runs its
        first ~20 ms of work 10-13 % slower (clock ramp, profiles/r02_shortrun_clock_ramp.log), so every timed loop of
        this file starts from a GPU that has just been busy for longer than that."""
        import torch
        t0, i = time.perf_counter(), 0
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(16):
                if plans is None:
                    self.step(i, sptr)
                else:
                    plans[i % self.rotate].launch_raw(sptr)
                i += 1
            torch.cuda.synchronize()

    def kernel_times(self, stream, sptr, launches=KERNEL_LAUNCHES):
        """Average duration of K1 alone and K3 alone: HIP events (recorded on the launch stream) around `launches`
        back-to-back launches of the one kernel over the rotating batches (inter-kernel gap ~40 ns in rocprofv3
        traces), after heat(); the figure rocprofv3 --kernel-trace --stats reproduces (profiles/)."""
        import torch
        res = {}
        for name, plans in (("k1", self.fplans), ("k3", self.bplans)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.heat(sptr, plans=plans)
            e0.record(stream)
            for i in range(launches):
                plans[i % self.rotate].launch_raw(sptr)
            e1.record(stream)
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e3 / launches
        return res["k1"], res["k3"]


def config_leg(device, cfg, dtype_name, per_column_grid, factor, dt, steps=100, launches=100, check_rows=4096, cols_per_block=0):
    """One extra workload of the N = 1 line on the SHIPPED library (round-4 verdict, items 1 and 3): BASELINE config ``cfg``
    in the arithmetic type ``dtype_name`` ('f64' / 'f32'), LES grid shared or one row per column (north_star's literal
    layout); whole steps, per-kernel HIP events, algorithmic bytes at the element size, and ``sample_check`` of what the
    timed plans left in HBM (bit comparison for fp64, F32_TOL for fp32)."""
    import torch
    from sp_coupler_amd import synthetic
    from sp_coupler_amd.engine import Engine
    eng = Engine(device, dtype=torch.float64 if dtype_name == "f64" else torch.float32)
    stream = torch.cuda.current_stream(eng.device)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    n, nG, nL, seed = synthetic.CONFIGS[cfg]
    esize = 8 if dtype_name == "f64" else 4
    ab = algorithmic_bytes(nG, nL, esize, shared_grid=not per_column_grid)
    rotate = 1 if n * ab["exchange"] > 2e9 else 2           # one batch of > 2 GB is already > 8x the Infinity Cache
    wl = Workload(eng, n, nG, nL, seed, rotate, factor, dt, cols_per_block, per_column_grid=per_column_grid)
    wl.heat(sptr)
    for i in range(3):
        wl.step(i, sptr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        wl.step(i, sptr)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k1, k3 = wl.kernel_times(stream, sptr, launches=launches)
    try:
        g0, zf0, zh0, p0 = wl.inputs0
        ok, det = sample_check(wl.fplans[0].outputs, wl.bplans[0].outputs, g0, p0, zf0, zh0, factor, dt, m=check_rows)
    except Exception as e:
        ok, det = False, {"failures": ["sample_check raised %r" % (e,)]}
    res = {"workload": "config %d: %d synthetic SP columns on one GPU, %d GCM <-> %d LES levels, %s, %s, %d batch(es) resident in HBM"
                       % (cfg, n, nG, nL, "fp64" if esize == 8 else "fp32", "LES grid PER COLUMN [n_cols x n_lev]" if per_column_grid else "shared LES grid", rotate),
           "dtype": dtype_name, "value": n * steps / el, "unit": "column-exchanges/s", "steps": steps, "ms_per_step": el / steps * 1e3,
           "bytes_per_exchange": ab["exchange"], "hbm_frac_whole_step": n * steps / el * ab["exchange"] / 1e9 / HBM_PEAK_GBS,
           "k1_avg_launch_us": k1, "k3_avg_launch_us": k3, "launches_timed": launches,
           "k1_kernel": wl.fplans[0].describe(), "k3_kernel": wl.bplans[0].describe(),
           "k1_algorithmic_bytes_per_launch": ab["k1_launch"] * n, "k3_algorithmic_bytes_per_launch": ab["k3_launch"] * n,
           "k1_frac": ab["k1_launch"] * n / (k1 * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "k3_frac": ab["k3_launch"] * n / (k3 * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "verified": bool(ok), "check": det}
    del wl
    torch.cuda.empty_cache()
    return res


def live_pmc_traffic(n_cols, timeout_s=100.0):
    """HBM bytes per K1 / K3 launch measured IN THIS RUN (round-4 verdict, weak 13: `roofline.traffic` used to be a constant from a
    builder's earlier run): two child processes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters
    only, as MI355X_MICROARCH.md prescribes) around tools/pmc_run.py -- the timed plans on config-3-sized rotating batches plus
    256-MiB calibration copies of the kernels' own 8 B-per-lane access width -- parsed by tools/pmc_summary.py.  Runs AFTER every
    timed region.  Returns (dict or None, text): any failure (no rocprofv3, a timeout, a counter the box refuses) leaves the
    constant of profiles/traffic.json in place and says so."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="spc_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(d, sub), "--", "python3",
                   os.path.join(ROOT, "tools", "pmc_run.py"), str(n_cols), "2"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d: %s" % (counter, r.returncode, (r.stderr or "")[-200:])
        from tools import pmc_summary
        res = pmc_summary.summarise(os.path.join(d, "fetch"), os.path.join(d, "write"), n_cols, 1 << 28, tag="measured inside this bench run")
        if not res.get("k_forward_bytes_per_launch") or not res.get("k_backward_bytes_per_launch"):
            return None, "the counter files hold no k_forward / k_backward rows"
        return res, "measured in this run: child processes `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` around tools/pmc_run.py, calibrated on 256-MiB copies (tools/pmc_summary.py)"
    except subprocess.TimeoutExpired:
        return None, "a rocprofv3 --pmc pass did not finish in %.0f s" % timeout_s
    except Exception as e:                              # a reported extra, never fatal for the headline
        return None, "live PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def self_launch(n_gpus, argv):
    """Run `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child process (this parent
    has not imported torch, so nothing here has initialised a GPU) and return its exit code; the children's stdout --
    rank 0's ONE JSON line -- and stderr pass straight through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def rehearse_cpu(args, rank, world):
    """The N > 1 path without a GPU (CPU tests of the self-launch and of the self-proving line): gloo rendezvous, this
    rank's shard of the workload, barrier, max-over-ranks of a per-rank time, and the per-rank check / identity / gather
    machinery of the real run -- with the TEST engine (tests/fake_engine.py, NumPy oracle) standing in for the device, so
    `sample_check` compares two independent CPU restatements.  Rank 0 prints the JSON line with `value: null`."""
    import socket
    import torch
    import torch.distributed as dist
    from sp_coupler_amd import sharding, synthetic
    dist.init_process_group("gloo")
    cfg = args.config if args.config is not None else 4
    total_cols = args.cols or synthetic.CONFIGS[cfg][0]
    nG, nL, seed = synthetic.CONFIGS[cfg][1:]
    lo, hi = sharding.shard_range(total_cols, rank, world)
    dist.barrier()
    tt = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    rows = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(rows, torch.tensor([lo, hi]))
    mine = {"rank": rank, "rows": [int(lo), int(hi)], "device": {"index": None, "name": "cpu (rehearsal)", "pci": socket.gethostname(),
                                                                  "uuid": "pid %d" % os.getpid()},
            "ms_per_step": 1.0 * (rank + 1), "verified": None, "check": None}
    if hi - lo <= 1024:                     # small rehearsal shards: run the check for real
        from tests.fake_engine import OracleEngine
        g, zf, zh, p = synthetic.make_batch(hi - lo, nG, nL, seed + 1000 * rank, couple_surface=False)
        t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}                       # noqa: E731
        gt, pt, zft, zht = t(g), t(p), torch.from_numpy(zf), torch.from_numpy(zh)
        fp, bp = OracleEngine().plan_exchange(gt, zft, zht, pt, 1.0, 1.0, 900.0)
        fp.launch(), bp.launch()
        if os.environ.get("SPC_REHEARSAL_CORRUPT_RANK") == str(rank):                     # test hook: one rank's outputs are wrong
            bp.outputs["f_T"][-1, 0] += 1.0
        ok, det = sample_check(fp.outputs, bp.outputs, gt, pt, zft, zht, 1.0, 900.0, m=min(4096, max(3, args.check_rows)))
        mine["verified"], mine["check"] = bool(ok), det
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    if rank == 0:
        ms = [r["ms_per_step"] for r in per_rank]
        ver = [r["verified"] for r in per_rank]
        print(json.dumps({"metric": "SP column-exchanges/sec (GCM<->LES forcing+tendency)", "value": None, "rehearsal": "cpu",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "backend": "gloo",
                          "ranks": dist.get_world_size(), "devices": [r["device"] for r in per_rank],
                          "per_rank_ms": {"min": min(ms), "max": max(ms), "all": ms},
                          "verified": (None if any(v is None for v in ver) else all(ver)),
                          "per_rank": [{k: r[k] for k in ("rank", "rows", "verified", "check")} for r in per_rank],
                          "max_over_ranks_s": float(tt.item()), "shards": [[int(a), int(b)] for a, b in rows],
                          "config": {"workload": "config %d, %d columns over %d ranks" % (cfg, total_cols, world)}}))
    dist.destroy_process_group()
    if rank == 0 and any(v is False for v in [r["verified"] for r in per_rank]):
        sys.exit("bench.py --rehearse-cpu: a rank's outputs differ from the oracle")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", type=int, default=None, help="BASELINE.md config id (default: 3 at N=1, 4 sharded at N>1)")
    ap.add_argument("--cols", type=int, default=None, help="override the TOTAL column count of the workload")
    ap.add_argument("--rotate", type=int, default=None, help="distinct batches cycled through (default 2; 8 for <= 4096 cols)")
    ap.add_argument("--cols-per-block", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip; also skips `verified`)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-cpu-multicore", action="store_true", help="skip the all-cores cpu_baseline extra")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the config-2 extra at N=1")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in API (Coupler.step) rate at N=1")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N>1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal on a one-GPU box: initialise the process group (RCCL), run the "
                    "barrier and the max-reduction of the N > 1 path with ONE rank under torch.distributed.run; the line says so")
    ap.add_argument("--device", type=int, default=None, help="force this HIP device for every rank (rehearsal on a 1-GPU box)")
    ap.add_argument("--no-anchor", action="store_true", help="skip the config-4-on-one-GPU `scaling_anchor` extra at N=1")
    ap.add_argument("--multi-devices", default=None,
                    help="N=1 extra `in_process_all_gpus`: config 4 through multi.MultiDeviceEngine on these device ids (default: every "
                         "visible GPU when there is more than one; '0,0' rehearses the path with two engines on one GPU; 'none' skips)")
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64", help="arithmetic type of the headline workload (f64 as the reference; "
                    "f32: the tolerance-sweep variant, checked against the fp64 oracle within F32_TOL)")
    ap.add_argument("--per-column-grid", action="store_true", help="headline workload with the LES grid packed per column [n_cols x n_lev]")
    ap.add_argument("--no-config5", action="store_true", help="skip the `config5` extra (88 838 columns 137<->512, f64 and f32) at N=1")
    ap.add_argument("--no-per-column-grid-leg", action="store_true", help="skip the `per_column_grid` extra (config 3, grid per column) at N=1")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure `roofline.traffic` in this run (two rocprofv3 --pmc child "
                    "processes after the timed regions, ~1 minute); report the constant of profiles/traffic.json instead")
    ap.add_argument("--check-rows", type=int, default=4096, help="rows per rank / device in the N > 1 output check (sample_check)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="N>1 launch mechanics only (rendezvous, sharding, barrier, max-reduction over gloo; no GPU work): CPU tests")
    args = ap.parse_args()

    # `python bench.py --gpus N` launched plainly (no torchrun around it): start the N ranks as FRESH CHILD PROCESSES
    # before this process has imported torch or touched the GPU, let rank 0's JSON line through, return their exit code.
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (torch.distributed.run must start one rank per GPU)" % (args.gpus, world))
    if args.rehearse_cpu:
        return rehearse_cpu(args, rank, world)

    import numpy
    import torch
    from sp_coupler_amd import sharding, synthetic
    from sp_coupler_amd.engine import Engine

    if world > 1 and args.device is None and torch.cuda.device_count() < world:
        sys.exit("bench.py --gpus %d needs %d visible GPUs, found %d (--device D puts every rank on GPU D: a rehearsal of the "
                 "launch path, not a measurement)" % (world, world, torch.cuda.device_count()))
    if args.device is not None:
        local = args.device
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist
        if args.backend == "nccl":      # RCCL.  The data path needs no collective (barrier + max of the elapsed time only),
            try:                        # so a failing RCCL does not void the measurement -- but the JSON says what ran.
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                probe = torch.zeros(1, device="cuda:%d" % local)
                dist.all_reduce(probe)      # fail here, not in the timed region
                torch.cuda.synchronize()
            except Exception as e:
                print("bench.py: RCCL unavailable (%r): barrier / max-reduction fall back to gloo; reported as "
                      "\"backend\": \"gloo (rccl init failed)\"" % (e,), file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                args.backend = "gloo (rccl init failed)"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(args.backend)

    cfg = args.config if args.config is not None else (3 if world == 1 else 4)
    total_cols, nG, nL, seed = synthetic.CONFIGS[cfg]
    if args.cols:
        total_cols = args.cols
    lo, hi = sharding.shard_range(total_cols, rank, world)       # contiguous row block of this rank
    n_cols = hi - lo
    rotate = args.rotate or (8 if n_cols <= 4096 else 2)
    dt_gcm, factor = 900.0, 1.0
    f32 = args.dtype == "f32"
    eng = Engine("cuda:%d" % local, dtype=torch.float32 if f32 else torch.float64)
    stream = torch.cuda.current_stream(eng.device)
    sptr = ctypes.c_void_p(stream.cuda_stream)

    wl = Workload(eng, n_cols, nG, nL, seed + 1000 * rank, rotate, factor, dt_gcm, args.cols_per_block,
                  keep_host=(rank == 0 and world == 1), per_column_grid=args.per_column_grid)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # The per-kernel HIP-event pass and the copy-rate probe run BEFORE the timed region, each behind Workload.heat():
    # after an idle gap (>= 50 ms, e.g. the host-side batch generation above) an MI355X runs the first ~20 ms of work
    # 10-13 % slower than steady state (profiles/r02_shortrun_clock_ramp.log, tools/shortrun.py), which is longer than
    # the driver's whole `--warmup 5 --steps 20` region.  `cold_clock` below reports the same W + K taken straight
    # after an idle gap.
    ab = wl.bytes                            # at the element size, grid shared or per column
    k1_us = k3_us = None
    kdiag = {}
    copy_gbs = None
    fence()                                  # N > 1: line the ranks up first, so no rank idles (and cools) at the next fence
    if not args.no_kernel_events:
        k1_us, k3_us = wl.kernel_times(stream, sptr)
        # measured device-to-device copy rate of this box (16 B/lane streaming copy, 256 MiB, read + write bytes):
        # the practical HBM ceiling reported next to the 8 TB/s spec peak
        from tools import spc_tools          # measurement instruments (tools/libspc_tools.so), not the product library
        src = torch.empty(1 << 28, dtype=torch.uint8, device=eng.device)
        dst = torch.empty_like(src)
        for _ in range(3):
            spc_tools.stream_copy(dst, src, stream)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(10):
            spc_tools.stream_copy(dst, src, stream)
        c1.record(stream)
        torch.cuda.synchronize()
        copy_gbs = 2.0 * src.numel() * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst

    wl.heat(sptr)
    for i in range(args.warmup):
        wl.step(i, sptr)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        wl.step(i, sptr)
    torch.cuda.synchronize()
    elapsed = own_elapsed = time.perf_counter() - t0
    fence()
    if dist is not None:
        tt = torch.tensor([elapsed], device=eng.device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # spread of the headline: the same W + K region four more times (each behind Workload.heat like the official one);
    # the official `value` stays the FIRST region, timed exactly as the contract says
    spread = None
    if world == 1 and not args.no_kernel_events:
        vals = [total_cols * args.steps / elapsed]
        for _ in range(4):
            wl.heat(sptr)
            for i in range(args.warmup):
                wl.step(i, sptr)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(args.steps):
                wl.step(i, sptr)
            torch.cuda.synchronize()
            vals.append(total_cols * args.steps / (time.perf_counter() - ts))
        sv = sorted(vals)
        spread = {"regions": len(vals), "min": sv[0], "median": sv[len(sv) // 2], "max": sv[-1], "first_is_value": True}

    cold = None
    if world == 1 and not args.no_kernel_events:
        # the same W + K protocol straight after 0.5 s of idle GPU (clocks ramped down): what a short region costs cold
        time.sleep(0.5)
        for i in range(args.warmup):
            wl.step(i, sptr)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(args.steps):
            wl.step(i, sptr)
        torch.cuda.synchronize()
        elc = time.perf_counter() - tc
        cold = {"ms_per_step": elc / args.steps * 1e3, "value": total_cols * args.steps / elc,
                "note": "same warmup+steps after 0.5 s of idle GPU; `value` above is taken with the per-kernel pass run first"}

    small = None
    if world == 1 and not args.no_small_batch and not args.no_kernel_events and cfg != 2:
        n2, nG2, nL2, seed2 = synthetic.CONFIGS[2]
        w2 = Workload(eng, n2, nG2, nL2, seed2, 8, factor, dt_gcm, args.cols_per_block)
        w2.heat(sptr)
        t2 = time.perf_counter()
        for i in range(2000):
            w2.step(i, sptr)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t2
        a2 = algorithmic_bytes(nG2, nL2)
        s1, s3 = w2.kernel_times(stream, sptr)
        small = {"workload": "config 2: 1024 synthetic SP columns, 91 GCM <-> 160 LES levels, fp64, 8 rotating "
                             "batches (cold: 8 x 41 MB > Infinity Cache)", "value": n2 * 2000 / el2,
                 "unit": "column-exchanges/s", "steps": 2000, "ms_per_step": el2 / 2000 * 1e3,
                 "k1_avg_launch_us": s1, "k3_avg_launch_us": s3,
                 "k1_frac": a2["k1_launch"] * n2 / (s1 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                 "k3_frac": a2["k3_launch"] * n2 / (s3 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                 "k1_algorithmic_bytes_per_launch": a2["k1_launch"] * n2,
                 "k3_algorithmic_bytes_per_launch": a2["k3_launch"] * n2}
        del w2

    anchor = None
    if world == 1 and cfg == 3 and not args.cols and not args.no_anchor and not args.no_kernel_events:
        # the N = 1 point of the series the N > 1 runs measure (config 4, 348 528 columns, strong scaling): the whole of
        # config 4 on this ONE GPU (14 GB live; one batch -- 55x the Infinity Cache -- so no rotation is needed)
        n4, nG4, nL4, seed4 = synthetic.CONFIGS[4]
        w4 = Workload(eng, n4, nG4, nL4, seed4, 1, factor, dt_gcm, args.cols_per_block)
        k4steps = max(20, min(args.steps, 100))
        w4.heat(sptr)
        for i in range(max(args.warmup, 3)):
            w4.step(i, sptr)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for i in range(k4steps):
            w4.step(i, sptr)
        torch.cuda.synchronize()
        el4 = time.perf_counter() - t4
        a4 = algorithmic_bytes(nG4, nL4)
        s1, s3 = w4.kernel_times(stream, sptr, launches=60)
        anchor = {"workload": "config 4: 348528 synthetic SP columns on ONE GPU (the series `--gpus N` shards over N GPUs), "
                              "91 GCM <-> 160 LES levels, fp64, shared LES grid, 1 batch of 14 GB resident in HBM",
                  "value": n4 * k4steps / el4, "unit": "column-exchanges/s", "n_gpus": 1, "steps": k4steps,
                  "ms_per_step": el4 / k4steps * 1e3, "k1_avg_launch_us": s1, "k3_avg_launch_us": s3,
                  "k1_frac": a4["k1_launch"] * n4 / (s1 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                  "k3_frac": a4["k3_launch"] * n4 / (s3 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                  "note": "strong-scaling efficiency at N GPUs = value(N) / (N x this value)"}
        del w4
        torch.cuda.empty_cache()

    # BASELINE config 5 (137 <-> 512, the high-vertical-resolution case) in both arithmetic types, and config 3 with the LES
    # grid packed per column: measured on the shipped library in EVERY default N = 1 line (round-4 verdict, items 1 and 3)
    config5 = percol = None
    plain = world == 1 and cfg == 3 and not args.cols and not f32 and not args.per_column_grid and not args.no_kernel_events
    if plain and not args.no_config5:
        config5 = {}
        for dn in ("f64", "f32"):
            try:
                config5[dn] = config_leg(eng.device, 5, dn, False, factor, dt_gcm, steps=60, launches=60, check_rows=args.check_rows)
            except Exception as e:                   # a reported extra, never fatal for the headline
                config5[dn] = {"error": repr(e)}
    if plain and not args.no_per_column_grid_leg:
        try:
            percol = config_leg(eng.device, 3, "f64", True, factor, dt_gcm, steps=200, launches=KERNEL_LAUNCHES, check_rows=args.check_rows)
            if k1_us:
                percol["k1_vs_shared_grid"] = percol["k1_avg_launch_us"] / k1_us
                percol["k3_vs_shared_grid"] = percol["k3_avg_launch_us"] / k3_us
        except Exception as e:
            percol = {"error": repr(e)}

    inproc = None
    ids = None
    if world == 1 and args.multi_devices != "none" and not args.no_kernel_events:
        if args.multi_devices:
            ids = [int(x) for x in args.multi_devices.split(",")]
        elif torch.cuda.device_count() > 1:
            ids = list(range(torch.cuda.device_count()))
    if ids:
        try:
            inproc = in_process_all_gpus(ids, factor, dt_gcm, args)
        except Exception as e:                       # a reported extra, never fatal for the headline
            inproc = {"error": repr(e)}

    dropin = None
    if world == 1 and not args.no_dropin:
        try:
            dropin = dropin_rate(eng)
        except Exception as e:                       # a reported extra, never fatal for the headline
            dropin = {"error": repr(e)}

    # N > 1: every rank proves ITS outputs and says who it is (a line whose ranks all sat on one GPU, or whose shards were
    # never checked, proves nothing): a sample of >= 4096 rows of the rank's block incl. its first and last row, inputs
    # read back from the device, plain-C oracle on the host, compared with what the timed plans left in HBM
    per_rank = None
    if dist is not None:
        g0, zf0, zh0, p0 = wl.inputs0
        try:
            ok_r, det_r = sample_check(wl.fplans[0].outputs, wl.bplans[0].outputs, g0, p0, zf0, zh0, factor, dt_gcm,
                                       m=max(4096, args.check_rows))
        except Exception as e:                          # a rank that cannot check reports that, it does not hang the others
            ok_r, det_r = False, {"failures": ["sample_check raised %r" % (e,)]}
        mine = {"rank": rank, "local_rank": local, "rows": [int(lo), int(hi)], "device": device_identity(local), "verified": bool(ok_r),
                "check": det_r, "ms_per_step": own_elapsed / args.steps * 1e3, "k1_avg_launch_us": k1_us, "k3_avg_launch_us": k3_us,
                "pid": os.getpid()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = total_cols * args.steps / elapsed
    gtxt = "LES grid PER COLUMN [n_cols x n_lev]" if args.per_column_grid else "shared LES grid"
    ttxt = "fp32" if f32 else "fp64"
    if world == 1:
        wtxt = ("config %d: %d synthetic SP columns on one GPU, %d GCM <-> %d LES levels, %s, %s, "
                "%d rotating batches resident in HBM" % (cfg, total_cols, nG, nL, ttxt, gtxt, rotate))
    else:
        wtxt = ("config %d: %d synthetic SP columns column-sharded over %d GPUs (%d per GPU, contiguous row blocks), "
                "%d GCM <-> %d LES levels, %s, %s, %d rotating batches per GPU"
                % (cfg, total_cols, world, n_cols, nG, nL, ttxt, gtxt, rotate))
    out = {
        "metric": "SP column-exchanges/sec (GCM<->LES forcing+tendency)",
        "value": value, "unit": "column-exchanges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": wtxt, "total_cols": total_cols, "n_cols_per_gpu": n_cols, "nG": nG, "nL": nL,
                   "rotate": rotate, "launches_per_step": 2, "parallelism": "columns sharded, no collective"},
        "backend": (args.backend if dist is not None else None),
        "series_note": ("ONE workload per scaling series: `--gpus N` (N > 1) shards config 4 (348 528 columns) over N GPUs -- strong "
                        "scaling; its N = 1 point is `scaling_anchor` of the N = 1 line (config 4 on one GPU).  The N = 1 headline "
                        "`value` is config 3 (35 718 columns, the largest config BASELINE.json places on ONE GPU): do not divide "
                        "value(N) by value(1)"),
        "bytes_per_exchange": ab["exchange"],
        "hbm_frac_whole_step": value / world * ab["exchange"] / 1e9 / HBM_PEAK_GBS,
        "ranks": (dist.get_world_size() if dist is not None else 1),
        "force_dist": (True if (args.force_dist and world == 1 and dist is not None) else None),   # a ONE-rank rehearsal of the N > 1 machinery
        "devices": ([r["device"] for r in per_rank] if per_rank else [device_identity(local)]),
    }
    roof_rank = 0
    if per_rank:
        ms = [r["ms_per_step"] for r in per_rank]
        out["per_rank_ms"] = {"min": min(ms), "max": max(ms), "all": ms}
        out["distinct_devices"] = len({(r["device"]["pci"], r["device"]["uuid"]) for r in per_rank})
        out["per_rank"] = [{k: r[k] for k in ("rank", "local_rank", "rows", "verified", "check", "k1_avg_launch_us", "k3_avg_launch_us", "pid")}
                           for r in per_rank]
        if all(r["k1_avg_launch_us"] is not None for r in per_rank):       # the roofline object is the SLOWEST rank's
            roof_rank = max(range(world), key=lambda i: per_rank[i]["k1_avg_launch_us"])
            k1_us, k3_us = per_rank[roof_rank]["k1_avg_launch_us"], per_rank[roof_rank]["k3_avg_launch_us"]
            n_cols = per_rank[roof_rank]["rows"][1] - per_rank[roof_rank]["rows"][0]
    if k1_us is not None:
        traffic, tsrc, traffic_k3, live_note = None, None, None, None
        if plain and not args.no_live_traffic:
            live, live_note = live_pmc_traffic(n_cols)
            if live is not None:
                traffic, traffic_k3, tsrc = live["k_forward_bytes_per_launch"], live["k_backward_bytes_per_launch"], live_note
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if traffic is None and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))     # PMC passes of tools/gpu_round.sh (separate --pmc runs, calibrated)
                if tj.get("n_cols") == n_cols and (nG, nL) == (91, 160) and not f32 and not args.per_column_grid:
                    traffic = tj.get("k_forward_bytes_per_launch")
                    tsrc = "profiles/traffic.json (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on another box, %s)%s" % (
                        tj.get("tag", "this round"), ("; live measurement not available: " + live_note) if live_note else "")
            except Exception:
                traffic = None
        ach = ab["k1_launch"] * n_cols / (k1_us * 1e-6) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_forward (K1, fused K2 index map), lean hot-path variant",
                           "rank": roof_rank, "rank_choice": ("slowest K1 over the ranks" if per_rank else None),
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": tsrc,
                           "frac_vs_pmc": (traffic / (k1_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                           "algorithmic_bytes_per_launch": ab["k1_launch"] * n_cols, "avg_launch_us": k1_us,
                           "launches_timed": KERNEL_LAUNCHES,
                           "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": (ach / copy_gbs) if copy_gbs else None,
                           "timing": dict(kdiag, method="HIP events on the launch stream around %d back-to-back launches "
                                          "of the kernel alone over the rotating batches" % KERNEL_LAUNCHES),
                           "backward": {"kernel": "k_backward (K3)", "avg_launch_us": k3_us, "traffic": traffic_k3,
                                        "achieved": ab["k3_launch"] * n_cols / (k3_us * 1e-6) / 1e9,
                                        "frac": ab["k3_launch"] * n_cols / (k3_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_launch": ab["k3_launch"] * n_cols}}
    if spread is not None:
        out["value_spread"] = spread
    if cold is not None:
        out["cold_clock"] = cold
    if small is not None:
        out["small_batch"] = small
    if anchor is not None:
        out["scaling_anchor"] = anchor
    if config5 is not None:
        out["config5"] = config5
    if percol is not None:
        out["per_column_grid"] = percol
    if inproc is not None:
        out["in_process_all_gpus"] = inproc
    if dropin is not None:
        out["dropin"] = dropin
        out["dropin_value"] = dropin.get("batched_protocol", {}).get("value")
    out["verified"] = None
    if per_rank:
        out["verified"] = all(r["verified"] for r in per_rank)
        out["verified_detail"] = {"method": "every rank: sample_check (>= 4096 rows of its block incl. first and last row, inputs read "
                                            "back from its device, plain-C oracle) after the timed region; AND over ranks",
                                  "bit_exact": BIT_EXACT,
                                  "failures": [("rank %d: " % r["rank"]) + f for r in per_rank for f in r["check"].get("failures", [])]}
    if args.cpu_seconds > 0 and world == 1:
        gcm, zf, zh, prof = wl.host0
        m = min(n_cols, 8192)             # bounded sample: the first m columns of batch 0, whole passes
        gs = {k: numpy.ascontiguousarray(v[:m]) for k, v in gcm.items()}
        ps = {k: numpy.ascontiguousarray(v[:m]) for k, v in prof.items()}
        if args.per_column_grid:
            zf, zh = numpy.ascontiguousarray(zf[:m]), numpy.ascontiguousarray(zh[:m])
        base, ref_f, ref_b = cpu_baseline(gs, zf, zh, ps, dt_gcm, factor, args.cpu_seconds)
        out["cpu_baseline"] = base
        # batch 0 was last written by the per-kernel loops above with the same inputs: check it
        if f32:      # the fp32 variant: fp64 oracle on the unrounded inputs, tolerances F32_TOL (input rounding included)
            F = {k: v.cpu().numpy()[:m] for k, v in wl.fplans[0].outputs.items()}
            B = {k: v.cpu().numpy()[:m] for k, v in wl.bplans[0].outputs.items()}
            bad, det = compare_with_oracle_f32(F, B, ref_f, ref_b, f32_scales(gs, ps, ref_f), factor, dt_gcm)
            ok, detail = (not bad), {"columns_checked": int(m), "fp32_vs_fp64_oracle": det, "tolerances": F32_TOL, "failures": bad}
        else:
            ok, detail = verify(wl.fplans[0], wl.bplans[0], ref_f, ref_b, m, factor, dt_gcm)
            try:
                detail["iexner_ulp_distance"] = pow_ulp_histogram(eng, gs["Pfull"][:2048])
            except Exception as e:                           # a reported extra
                detail["iexner_ulp_distance"] = {"error": repr(e)}
        out["verified"], out["verified_detail"] = ok, detail
        workers = min(16, os.cpu_count() or 1)
        if workers > 1 and not args.no_cpu_multicore:
            try:
                out["cpu_baseline"]["all_cores"] = cpu_baseline_multicore(gs, zf, zh, ps, min(6.0, args.cpu_seconds), workers)
            except Exception as e:                       # a reported extra, never fatal
                out["cpu_baseline"]["all_cores"] = {"error": repr(e)}
    elif args.cpu_seconds > 0:
        out["cpu_baseline"] = None   # reported at N=1 only
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if out["verified"] is False:
        sys.exit("bench.py: outputs of the timed plans differ from the oracle: %s" % out["verified_detail"]["failures"])
    legs = [("per_column_grid", percol)] + [("config5." + k, v) for k, v in (config5 or {}).items()]
    wrong = [name for name, leg in legs if leg is not None and leg.get("verified") is False]
    if wrong:
        sys.exit("bench.py: outputs of the %s leg(s) differ from the oracle" % ", ".join(wrong))
    if dropin is not None and dropin.get("verified") is False:
        sys.exit("bench.py: the drop-in step delivered something else than the kernels compute: %s"
                 % [f for d in dropin["verified_detail"] for f in d["failures"]])


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-worker":
        cpu_worker_main(sys.argv[2], float(sys.argv[3]))
    else:
        main()
