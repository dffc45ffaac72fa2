"""Deterministic synthetic SP-column batches.

The reference has no synthetic generator on this path: its stand-ins (``splib/spdummy.py:141-172,
253-275``) have 20 levels and lack ``Zgfull/Zghalf/SH/QL/QI/A``.  This generator produces physically
shaped ``[n_cols x n_lev]`` float64 batches with the variable names of ``spcpl.gcm_vars``
(``splib/spcpl.py:32-33``) and of the LES profile dict (``splib/spcpl.py:767``).

Level geometry comes from the bundled case of the reference: the L19 hybrid A/B table decoded from
the GRIB1 headers of ``oifs-input/ICMSHTESTINIT`` (SURVEY.md section 8(d) note) and the DALES grid
``zf = 12.5 + 25 k`` m, ``zh = 25 k`` m (``dales-input/prof.inp.001:3-12``).
Host-side NumPy only; the batches are uploaded to HBM by ``ColumnBatch.from_numpy``.
"""
import numpy

# constants of splib/sputils.py:14-20
pref0, rd, rv, cp, rlv, grav = 1e5, 287.04, 461.5, 1004., 2.53e6, 9.81

L19_A = numpy.array([0, 2000, 4000, 6046.11, 8267.93, 10609.51, 12851.10, 14698.50, 15861.13, 16116.24,
                     15356.93, 13621.46, 11101.56, 8127.14, 5125.14, 2549.97, 783.20, 0, 0, 0], dtype=numpy.float64)
L19_B = numpy.array([0, 0, 0, 3.3899e-4, 3.35719e-3, 1.307004e-2, 3.407715e-2, 7.06498e-2, 0.12591666,
                     0.20119542, 0.29551965, 0.40540922, 0.52493221, 0.64610797, 0.75969839, 0.85643756,
                     0.92874694, 0.97298521, 0.9922815, 1], dtype=numpy.float64)

#: BASELINE.md section 4: config id -> (n_cols, nG, nL, seed)
CONFIGS = {
    1: (2, 19, 160, 20261005),
    2: (1024, 91, 160, 20261006),
    3: (35718, 91, 160, 20261007),
    4: (348528, 91, 160, 20261008),
    5: (88838, 137, 512, 20261009),
}


def hybrid_coefficients(nG):
    """Half-level hybrid coefficients A [Pa], B for nG full levels (nG+1 values, top first)."""
    if nG == 19:
        return L19_A.copy(), L19_B.copy()
    eta19 = numpy.linspace(0.0, 1.0, 20)
    eta = numpy.linspace(0.0, 1.0, nG + 1)
    return numpy.interp(eta, eta19, L19_A), numpy.interp(eta, eta19, L19_B)


def les_grid(nL):
    """(zf, zh) of the LES: DALES RICO-like case for 160 levels, 10 m spacing otherwise."""
    dz = 25.0 if nL == 160 else 10.0
    k = numpy.arange(nL, dtype=numpy.float64)
    return (k + 0.5) * dz, k * dz


def _smooth(x, passes=3):
    if x.shape[1] < 3:
        return x
    for _ in range(passes):
        x = numpy.concatenate([x[:, :1], 0.25 * x[:, :-2] + 0.5 * x[:, 1:-1] + 0.25 * x[:, 2:], x[:, -1:]], axis=1)
    return x


def _batched_interp(x, xp, fp):
    """Row-wise linear interpolation with end clamping (generator use only, not parity-critical).
    x [nL] or [n x nL], xp ascending [n x m], fp [n x m]."""
    n, m = xp.shape
    x2 = numpy.broadcast_to(x, (n, x.shape[-1]))
    big = 4.0 * (max(numpy.abs(xp).max(), numpy.abs(x2).max()) + 1.0)
    off = (numpy.arange(n, dtype=numpy.float64) * big)[:, None]
    flat = (xp + off).ravel()
    j = numpy.searchsorted(flat, (numpy.clip(x2, xp[:, :1], xp[:, -1:]) + off).ravel(), side="right") - 1
    row0 = (numpy.arange(n) * m)[:, None]
    j = numpy.clip(j.reshape(n, -1), row0, row0 + m - 2)
    fpf, xpf = fp.ravel(), xp.ravel()
    w = (numpy.clip(x2, xp[:, :1], xp[:, -1:]) - xpf[j]) / (xpf[j + 1] - xpf[j])
    return fpf[j] + w * (fpf[j + 1] - fpf[j])


def make_gcm_columns(n_cols, nG, seed, couple_surface=True):
    """GCM side of the batch: dict of the 11 ``gcm_vars`` (+7 ``surf_vars``), float64."""
    rng = numpy.random.default_rng(seed)
    A, B = hybrid_coefficients(nG)
    ps = rng.uniform(9.5e4, 1.03e5, size=(n_cols, 1))
    Ph = A[None, :] + B[None, :] * ps                      # [n x nG+1], Ph[:,0] = 0
    Pf = 0.5 * (Ph[:, :-1] + Ph[:, 1:])
    z_apx = 7000.0 * numpy.log(ps / Pf)                    # rough height for shaping T and SH
    T = numpy.maximum(288.0 - 6.5e-3 * z_apx, 210.0) + rng.normal(0.0, 1.0, size=(n_cols, nG))
    SH = 0.015 * numpy.exp(-z_apx / 2500.0) * rng.uniform(0.5, 1.0, size=(n_cols, nG))
    cloudy = (rng.uniform(size=(n_cols, nG)) < 0.10) & (z_apx < 6000.0)
    QL = numpy.where(cloudy, rng.uniform(0.0, 1e-3, size=(n_cols, nG)), 0.0)
    QI = numpy.where(cloudy & (T < 268.0), rng.uniform(0.0, 5e-4, size=(n_cols, nG)), 0.0)
    Acl = numpy.where(cloudy, rng.uniform(0.0, 1.0, size=(n_cols, nG)), 0.0)
    U = _smooth(rng.normal(0.0, 10.0, size=(n_cols, nG)))
    V = _smooth(rng.normal(0.0, 10.0, size=(n_cols, nG)))
    # hydrostatic geopotential: dZ = rd*Tv/(grav*Pf)*dP (commented formula at spcpl.py:180-185)
    Tv = T * (1 + (rv / rd - 1) * SH - (QL + QI))
    dZ = rd * Tv / (grav * Pf) * (Ph[:, 1:] - Ph[:, :-1])
    zsurf_g = rng.uniform(0.0, 2e4, size=(n_cols, 1))      # surface geopotential m^2/s^2
    Zh_m = numpy.concatenate([numpy.cumsum(dZ[:, ::-1], axis=1)[:, ::-1], numpy.zeros((n_cols, 1))], axis=1)
    Zghalf = grav * Zh_m + zsurf_g
    Zgfull = 0.5 * (Zghalf[:, :-1] + Zghalf[:, 1:])
    out = dict(U=U, V=V, T=T, SH=SH, QL=QL, QI=QI, Pfull=Pf, Phalf=Ph, A=Acl, Zgfull=Zgfull, Zghalf=Zghalf)
    if couple_surface:
        out.update(Z0M=rng.uniform(1e-4, 1.0, n_cols), Z0H=rng.uniform(1e-5, 0.1, n_cols),
                   QLflux=-rng.uniform(0, 1e-6, n_cols), QIflux=-rng.uniform(0, 1e-7, n_cols),
                   SHflux=-rng.uniform(0, 1e-4, n_cols), TLflux=-rng.uniform(0, 200.0, n_cols),
                   TSflux=rng.uniform(-100.0, 50.0, n_cols))
    return {k: numpy.ascontiguousarray(v, dtype=numpy.float64) for k, v in out.items()}


def make_les_profiles(gcm, nL, seed, per_column_grid=False):
    """LES side: grids and slab means = forward-interpolated GCM profile + Gaussian noise
    (sigma 0.5 m/s, 0.2 K, 2e-4), keys of ``spcpl.get_les_profiles`` (splib/spcpl.py:767)."""
    rng = numpy.random.default_rng(seed + 7919)
    n, nG = gcm["T"].shape
    zf, zh = les_grid(nL)
    if per_column_grid:  # stretch each column's grid a little (exercises les_grid_shared = 0)
        s = rng.uniform(0.9, 1.1, size=(n, 1))
        zf, zh = zf[None, :] * s, zh[None, :] * s
    Zf = (gcm["Zgfull"] - gcm["Zghalf"][:, -1:]) / grav
    thl_ = (gcm["T"] - rlv * (gcm["QL"] + gcm["QI"]) / cp) * (gcm["Pfull"] / pref0) ** (-rd / cp)
    qt_ = gcm["SH"] + gcm["QL"] + gcm["QI"]
    xp = Zf[:, ::-1]
    u = _batched_interp(zf, xp, gcm["U"][:, ::-1])
    v = _batched_interp(zf, xp, gcm["V"][:, ::-1])
    thl = _batched_interp(zf, xp, thl_[:, ::-1])
    qt = _batched_interp(zf, xp, qt_[:, ::-1])
    ql = _batched_interp(zf, xp, gcm["QL"][:, ::-1])
    pf = _batched_interp(zf, xp, gcm["Pfull"][:, ::-1])
    shape = (n, nL)
    prof = dict(
        U=u + rng.normal(0, 0.5, shape), V=v + rng.normal(0, 0.5, shape),
        THL=thl + rng.normal(0, 0.2, shape), QT=numpy.abs(qt + rng.normal(0, 2e-4, shape)),
        QL=numpy.abs(ql + rng.normal(0, 2e-5, shape)) * (rng.uniform(size=shape) < 0.3),
        PS=gcm["Phalf"][:, -1] + rng.normal(0, 50.0, n),
        Rain=rng.uniform(0, 1e-2, n), rain_last=rng.uniform(0, 5e-3, n),
        presf=pf, Rhof=pf / (rd * 290.0), Rhobf=pf / (rd * 290.0) * rng.uniform(0.98, 1.02, shape),
        QR=rng.uniform(0, 1e-5, shape), A=rng.uniform(0.0, 1.0, (n, nG)),
    )
    prof["QL_ice"] = rng.uniform(0, 1, shape) * prof["QL"]
    prof["T"] = prof["THL"] * (pf / pref0) ** (rd / cp) + rlv * prof["QL"] / cp + rng.normal(0, 0.1, shape)
    prof = {k: numpy.ascontiguousarray(v_, dtype=numpy.float64) for k, v_ in prof.items()}
    return numpy.ascontiguousarray(zf), numpy.ascontiguousarray(zh), prof


def make_batch(n_cols, nG=91, nL=160, seed=20261006, couple_surface=True, per_column_grid=False):
    """Returns (gcm dict, zf, zh, les profile dict). dt = 900 s and factor = 1 are the bundled
    case's values (oifs-input/fort.4:52, splib/splib.py:46,57) and are chosen by the caller."""
    gcm = make_gcm_columns(n_cols, nG, seed, couple_surface)
    zf, zh, prof = make_les_profiles(gcm, nL, seed, per_column_grid)
    return gcm, zf, zh, prof


def make_batch_tiled(n_cols, nG=91, nL=160, seed=20261006, base=8192, couple_surface=True):
    """``n_cols`` columns built from a ``base``-column synthetic batch, tiled with a per-tile perturbation of the
    state (heights untouched, so they stay monotone): the generator itself needs minutes for the 3.5e5 columns of
    config 4.  Used for the full-size benchmark / test batches; every column still differs from every other."""
    if n_cols <= base:
        return make_batch(n_cols, nG, nL, seed, couple_surface)
    gcm, zf, zh, prof = make_batch(base, nG, nL, seed, couple_surface)
    reps = -(-n_cols // base)
    tile = numpy.repeat(numpy.arange(reps, dtype=numpy.float64), base)[:n_cols]

    def rep(a):
        return numpy.ascontiguousarray(numpy.concatenate([a] * reps, axis=0)[:n_cols])
    g = {k: rep(v) for k, v in gcm.items()}
    p = {k: rep(v) for k, v in prof.items()}
    g["T"] += 0.01 * tile[:, None]
    g["U"] *= (1.0 + 1e-3 * tile[:, None])
    p["THL"] += 0.02 * tile[:, None]
    p["V"] -= 0.05 * tile[:, None]
    p["PS"] += tile
    return g, zf, zh, p


def make_batch_tiled_device(device, n_cols, nG=91, nL=160, seed=20261006, base=8192, couple_surface=True, keys=None,
                            dtype=None, per_column_grid=False):
    """make_batch_tiled with the tiling done ON THE DEVICE: only the ``base`` columns are generated and cross PCIe, the
    tiles and their perturbations are torch ops in HBM (IEEE add / multiply: the same bits as the host version, and
    tile 0 IS the base batch).  Returns (gcm, zf, zh, prof) of device tensors plus the host base batch.
    ``dtype``: the tensors are generated in float64 and rounded ONCE at the end (torch.float32: the fp32 arithmetic
    variant's inputs).  ``per_column_grid``: zf / zh are [n_cols x nL], one (slightly stretched) grid per column --
    north_star's literal layout, ``h = les.zf_cache`` per LES at splib/spcpl.py:222 -- instead of one shared [nL] vector."""
    import torch
    host = make_batch(min(n_cols, base), nG, nL, seed, couple_surface, per_column_grid=per_column_grid)
    gcm, zf, zh, prof = host
    up = lambda d: {k: torch.from_numpy(v).to(device) for k, v in d.items() if keys is None or k in keys}   # noqa: E731
    g, p = up(gcm), up(prof)
    zf_d, zh_d = torch.from_numpy(zf).to(device), torch.from_numpy(zh).to(device)
    if n_cols > base:
        reps = -(-n_cols // base)
        tile = torch.arange(reps, dtype=torch.float64, device=device).repeat_interleave(base)[:n_cols]
        rep = lambda t: t.repeat(*((reps,) + (1,) * (t.dim() - 1)))[:n_cols].contiguous()                   # noqa: E731
        g, p = {k: rep(v) for k, v in g.items()}, {k: rep(v) for k, v in p.items()}
        if per_column_grid:
            zf_d, zh_d = rep(zf_d), rep(zh_d)
        g["T"] += 0.01 * tile[:, None]
        g["U"] *= (1.0 + 1e-3 * tile[:, None])
        p["THL"] += 0.02 * tile[:, None]
        p["V"] -= 0.05 * tile[:, None]
        p["PS"] += tile
    if dtype is not None and dtype != torch.float64:
        g, p = {k: v.to(dtype) for k, v in g.items()}, {k: v.to(dtype) for k, v in p.items()}
        zf_d, zh_d = zf_d.to(dtype), zh_d.to(dtype)
    return g, zf_d, zh_d, p, host


def make_config(cfg_id, n_cols=None):
    n, nG, nL, seed = CONFIGS[cfg_id]
    return make_batch(n if n_cols is None else n_cols, nG, nL, seed)
