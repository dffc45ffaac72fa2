"""TEST INFRASTRUCTURE ONLY -- CPU (NumPy) restatement of the reference coupling math.

This module is the *oracle* for the hot path named in BASELINE.json:
the per-column arithmetic of ``splib/spcpl.py`` and the helpers of ``splib/sputils.py`` in
the reference (CloudResolvingClimateModeling/sp-coupler), restated on plain float64 NumPy
arrays (SI values, no AMUSE units).  It may be imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- never by the
product package ``sp_coupler_amd``.

Pinning status
--------------
* The reference cannot be imported in the build container (``omuse``/``amuse`` are absent and stay
  absent), so no output of the reference itself is available.
* PINNED by the reference's own tests: ``exner``/``iexner``/``rms`` (``splib/test/sputils_test.py:10-39``)
  and the cloud-fraction level-index map (``splib/test/spcpl_test.py:10-16`` with the dummy LES grid of
  ``splib/spdummy.py:219-222,261-262,319-321``) -- see ``tests/golden/reference_known_answers.json``.
* The interpolation/searchsorted arithmetic calls the very same third-party routines the reference
  calls (``numpy.interp`` at ``sputils.py:86`` and ``numpy.searchsorted`` at ``sputils.py:91``).
* Everything else (interpolated profiles, forcings, tendencies) follows the reference line by line
  but is **parity unpinned** by any reference fixture: the reference's tests hold none for it.

Per-column functions follow the reference's operation order exactly; citations are
``file:line`` relative to the reference root.
"""
import numpy

# Physical constants -- splib/sputils.py:14-20 (values only; units are SI-coherent, factor 1)
pref0 = 1e5      # Pa
rd = 287.04      # J/kg/K
rv = 461.5       # J/kg/K
cp = 1004.       # J/kg/K
rlv = 2.53e6     # J/kg
grav = 9.81      # m/s^2

# Variable lists -- splib/spcpl.py:32-33
gcm_vars = ["U", "V", "T", "SH", "QL", "QI", "Pfull", "Phalf", "A", "Zgfull", "Zghalf"]
surf_vars = ["Z0M", "Z0H", "QLflux", "QIflux", "SHflux", "TLflux", "TSflux"]


def rms(a):
    """splib/sputils.py:23-24"""
    return numpy.sqrt(numpy.mean(a ** 2))


def exner(p):
    """splib/sputils.py:28-29"""
    return (p / pref0) ** (rd / cp)


def iexner(p):
    """splib/sputils.py:33-34"""
    return (p / pref0) ** (-rd / cp)


def interp(x, xp, fp):
    """splib/sputils.py:82-86 -- numpy.interp on the bare numbers."""
    return numpy.interp(x, xp, fp)


def searchsorted(a, v, **kwargs):
    """splib/sputils.py:88-91"""
    return numpy.searchsorted(a, v, **kwargs)


def interp_restated(x, xp, fp):
    """Scalar-loop restatement of ``numpy.interp`` (numpy/_core/src/multiarray/compiled_base.c,
    ``arr_interp``, numpy 2.2) -- the formula the C oracle and the HIP kernels implement.
    Checked bit-for-bit against ``numpy.interp`` in tests/test_oracle.py."""
    x = numpy.asarray(x, dtype=numpy.float64)
    xp = numpy.asarray(xp, dtype=numpy.float64)
    fp = numpy.asarray(fp, dtype=numpy.float64)
    n = len(xp)
    out = numpy.empty(x.shape, dtype=numpy.float64)
    with numpy.errstate(all="ignore"):
        for i, xv in enumerate(x):
            if n == 1:                      # numpy's lenxp == 1 branch: fp[0] for every x, NaN included
                out[i] = fp[0]
                continue
            if numpy.isnan(xv):
                out[i] = xv
                continue
            if xv > xp[n - 1]:
                out[i] = fp[n - 1]
                continue
            if xv < xp[0]:
                out[i] = fp[0]
                continue
            j = int(numpy.searchsorted(xp, xv, side="right")) - 1
            if j == n - 1:
                out[i] = fp[j]
            elif xp[j] == xv:
                out[i] = fp[j]
            else:
                slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j])
                r = slope * (xv - xp[j]) + fp[j]
                if numpy.isnan(r):
                    r = slope * (xv - xp[j + 1]) + fp[j + 1]
                    if numpy.isnan(r) and fp[j] == fp[j + 1]:
                        r = fp[j]
                out[i] = r
    return out


def integral(a, b, z, q, w=None):
    """splib/sputils.py:94-161 -- integral of piecewise-constant q(z) from a to b, optionally
    weighted; returns None when an end point is outside z (sputils.py:113-115)."""
    if a < z[0] or a > z[-1] or b < z[0] or b > z[-1]:
        return None
    sign = 1
    if a > b:
        sign = -1
        a, b = b, a
    ia = 0
    while z[ia + 1] < a:
        ia += 1
    ib = ia
    while z[ib + 1] < b:
        ib += 1
    if w is None:
        S = (q[ia:ib + 1] * (z[ia + 1:ib + 2] - z[ia:ib + 1])).sum()
        Sa = q[ia] * (a - z[ia])
        Sb = q[ib] * (z[ib + 1] - b)
        return (S - Sa - Sb) * sign
    S = (w[ia:ib + 1] * q[ia:ib + 1] * (z[ia + 1:ib + 2] - z[ia:ib + 1])).sum()
    Sa = w[ia] * q[ia] * (a - z[ia])
    Sb = w[ib] * q[ib] * (z[ib + 1] - b)
    Sw = (w[ia:ib + 1] * (z[ia + 1:ib + 2] - z[ia:ib + 1])).sum()
    Swa = w[ia] * (a - z[ia])
    Swb = w[ib] * (z[ib + 1] - b)
    return (S - Sa - Sb) / (Sw - Swa - Swb) * sign


def interp_c(Zh, zh, q, rho):
    """splib/sputils.py:173-189 -- conservative fine->coarse interpolation. Zh descending,
    zh ascending; levels whose upper edge is not below the LES top stay 0 (sputils.py:187)."""
    Q = numpy.zeros(len(Zh) - 1)
    for i in range(len(Q)):
        if Zh[i] < zh[-1]:
            Q[i] = integral(Zh[i + 1], Zh[i], zh, q, rho)
    return Q


def interp_rho(Zh, zh, rho):
    """splib/sputils.py:191-197 -- a density on the coarser grid (unweighted integral / layer thickness)."""
    RHO = numpy.zeros(len(Zh) - 1)
    for i in range(len(RHO)):
        if Zh[i] < zh[-1]:
            RHO[i] = integral(Zh[i + 1], Zh[i], zh, rho) / (Zh[i] - Zh[i + 1])
    return RHO


def cloud_fraction_indices(zh, Zh):
    """splib/spcpl.py:26 and 764: searchsorted(zh, Zh, side='right')[:-1][::-1]"""
    return searchsorted(zh, Zh, side="right")[:-1:][::-1]


def convert_surface_fluxes(col):
    """splib/spcpl.py:136-167. ``col`` maps variable name -> value (profiles 1-D, fluxes scalar)."""
    Ph = col["Phalf"]
    T = col["T"]
    rho = Ph[-1] / (rd * T[-1])                                              # spcpl.py:153
    wqt = - (col["QLflux"] + col["QIflux"] + col["SHflux"]) / rho            # spcpl.py:159
    wthl = - col["TSflux"] * iexner(Ph[-1]) / (cp * rho)                     # spcpl.py:161
    return col["Z0M"], col["Z0H"], wthl, wqt


def convert_profiles(col, zf):
    """splib/spcpl.py:171-246. Returns a dict with the tuple the reference returns
    (u, v, thl, qt, ps, ql) plus the cached heights and the spifs diagnostics."""
    U, V, T, SH, QL, QI, Pf, Ph, A, Zgfull, Zghalf = (col[v] for v in gcm_vars)
    c = rv / rd - 1                                                          # spcpl.py:175
    Tv = T * (1 + c * SH - (QL + QI))                                        # spcpl.py:176
    Zh = (Zghalf - Zghalf[-1]) / grav                                        # spcpl.py:197
    Zf = (Zgfull - Zghalf[-1]) / grav                                        # spcpl.py:198
    thl_ = (T - (rlv * (QL + QI)) / cp) * iexner(Pf)                         # spcpl.py:214
    qt_ = SH + QL + QI                                                       # spcpl.py:215
    h = zf                                                                   # spcpl.py:222
    thl = interp(h, Zf[::-1], thl_[::-1])                                    # spcpl.py:224
    qt = interp(h, Zf[::-1], qt_[::-1])                                      # spcpl.py:225
    ql = interp(h, Zf[::-1], QL[::-1])                                       # spcpl.py:226
    u = interp(h, Zf[::-1], U[::-1])                                         # spcpl.py:227
    v = interp(h, Zf[::-1], V[::-1])                                         # spcpl.py:228
    return dict(u=u, v=v, thl=thl, qt=qt, ps=Ph[-1], ql=ql, Zf=Zf, Zh=Zh, Tv=Tv, THL=thl_, QT=qt_)


def set_les_forcings(col, zf, prof, dt_gcm, factor, rain_last=0.0, couple_surface=False):
    """splib/spcpl.py:299-385 (arithmetic only). ``prof`` holds the LES slab means
    U,V,THL,QT,QL [nL], PS and Rain (scalars) -- spcpl.py:302-323."""
    cp_ = convert_profiles(col, zf)                                          # spcpl.py:300
    u, v, thl, qt, ps, ql = (cp_[k] for k in ("u", "v", "thl", "qt", "ps", "ql"))
    rain = prof.get("Rain", 0.0)
    rainrate = (rain - rain_last) / dt_gcm                                   # spcpl.py:325
    out = dict(cp_)
    out["f_u"] = factor * (u - prof["U"]) / dt_gcm                           # spcpl.py:328
    out["f_v"] = factor * (v - prof["V"]) / dt_gcm                           # spcpl.py:329
    out["f_thl"] = factor * (thl - prof["THL"]) / dt_gcm                     # spcpl.py:330
    out["f_qt"] = factor * (qt - prof["QT"]) / dt_gcm                        # spcpl.py:331
    out["f_ps"] = factor * (ps - prof["PS"]) / dt_gcm                        # spcpl.py:332
    out["f_ql"] = factor * (ql - prof["QL"]) / dt_gcm                        # spcpl.py:333
    out["ql_ref"] = ql                                                       # spcpl.py:347-348
    out["rainrate"] = rainrate
    if couple_surface:
        z0m, z0h, wthl, wqt = convert_surface_fluxes(col)                    # spcpl.py:360
        out.update(z0m=z0m, z0h=z0h, wthl=wthl, wqt=wqt)
    return out


def set_gcm_tendencies(col, Zf, Zh, zf, zh, prof, dt_gcm, factor=1, conservative=False):
    """splib/spcpl.py:388-555 (arithmetic only). ``prof`` holds U,V,THL,QT,QL,QL_ice,T [nL],
    A [nG] (in the order get_cloudfraction(indices) returns it), Rhobf [nL] (conservative only)."""
    U, V, T, SH, QL, QI, Pf, Ph, A, Zgfull, Zghalf = (col[v] for v in gcm_vars)
    h = zf
    u_d = prof["U"]
    v_d = prof["V"]
    thl_d = prof["THL"]
    qt_d = prof["QT"]
    ql_d = prof["QL"]
    ql_ice_d = prof["QL_ice"]
    ql_water_d = ql_d - ql_ice_d                                             # spcpl.py:402
    A_d = prof["A"][::-1]                                                    # spcpl.py:404
    pf = interp(h, Zf[::-1], Pf[::-1])                                       # spcpl.py:408
    t = thl_d * exner(pf) + rlv * ql_d / cp                                  # spcpl.py:409
    t_d = prof["T"]                                                          # spcpl.py:411
    ft = dt_gcm                                                              # spcpl.py:427
    if not conservative:
        t_d = interp(Zf, h, t_d)                                             # spcpl.py:471
        qt_d = interp(Zf, h, qt_d)                                           # spcpl.py:472
        ql_d = interp(Zf, h, ql_d)                                           # spcpl.py:473
        ql_water_d = interp(Zf, h, ql_water_d)                               # spcpl.py:474
        ql_ice_d = interp(Zf, h, ql_ice_d)                                   # spcpl.py:475
        u_d = interp(Zf, h, u_d)                                             # spcpl.py:476
        v_d = interp(Zf, h, v_d)                                             # spcpl.py:477
    else:
        rhobf_d = prof["Rhobf"]
        t_d = interp_c(Zh, zh, t_d, rhobf_d)                                 # spcpl.py:482
        qt_d = interp_c(Zh, zh, qt_d, rhobf_d)                               # spcpl.py:483
        ql_d = interp_c(Zh, zh, ql_d, rhobf_d)                               # spcpl.py:484
        ql_water_d = interp_c(Zh, zh, ql_water_d, rhobf_d)                   # spcpl.py:485
        ql_ice_d = interp_c(Zh, zh, ql_ice_d, rhobf_d)                       # spcpl.py:486
        u_d = interp_c(Zh, zh, u_d, rhobf_d)                                 # spcpl.py:487
        v_d = interp_c(Zh, zh, v_d, rhobf_d)                                 # spcpl.py:488
    start_index = int(searchsorted(-Zf, -h[-1]))                             # spcpl.py:498
    with numpy.errstate(all="ignore"):
        f_T = factor * (t_d - T) / ft                                        # spcpl.py:518
        f_SH = factor * ((qt_d - ql_d) - SH) / ft                            # spcpl.py:519
        f_QL = factor * (ql_water_d - QL) / ft                               # spcpl.py:520
        f_QI = factor * (ql_ice_d - QI) / ft                                 # spcpl.py:521
        f_U = factor * (u_d - U) / ft                                        # spcpl.py:524
        f_V = factor * (v_d - V) / ft                                        # spcpl.py:525
        f_A = factor * (A_d - A) / ft                                        # spcpl.py:526
        for f in (f_T, f_SH, f_QL, f_QI, f_U, f_V, f_A):                     # spcpl.py:527-533
            f[0:start_index] *= 0
    return dict(f_T=f_T, f_SH=f_SH, f_QL=f_QL, f_QI=f_QI, f_U=f_U, f_V=f_V, f_A=f_A,
                start_index=start_index, pf=pf, t=t, ql_water=prof["QL"] - prof["QL_ice"], A_d=A_d)


def output_column_conversion(profile):
    """splib/spcpl.py:251-267 (in place, like the reference)."""
    c = rv / rd - 1
    profile['Tv'] = profile['T'] * (1 + c * profile['SH'] - (profile['QL'] + profile['QI']))
    Zghalf = profile['Zghalf']
    Zgfull = profile['Zgfull']
    Zh = (Zghalf - Zghalf[-1]) / grav
    Zf = (Zgfull - Zghalf[-1]) / grav
    profile['Zh'] = Zh[1:]
    profile['Zf'] = Zf[:]
    profile['Psurf'] = profile['Ph'][-1]
    profile['Ph'] = profile['Ph'][1:]
    profile['THL'] = (profile['T'] - (rlv * (profile['QL'] + profile['QI'])) / cp) * iexner(profile['Pf'])
    profile['QT'] = profile['SH'] + profile['QL'] + profile['QI']


# ---------------------------------------------------------------------------------------------
# Batched drivers: the reference's serial ``for les in les_models`` loops (splib/splib.py:317-323
# and 330-332) over rows of [n_cols x n_lev] arrays.  Same names/layout as the product's batch.
# ---------------------------------------------------------------------------------------------
def _row(d, i, names):
    return {k: d[k][i] for k in names if k in d}


def _grid(z, i):
    z = numpy.asarray(z)
    return z if z.ndim == 1 else z[i]


def forward_batched(gcm, les, zf, zh=None, factor=1.0, dt=900.0, couple_surface=False):
    """gcm: dict of [n x nG] / [n x (nG+1)] arrays (+ surface [n]); les: dict U,V,THL,QT,QL [n x nL],
    PS [n], optional Rain, rain_last [n]. Returns dict of stacked outputs (+ idx when zh given)."""
    n = gcm["T"].shape[0]
    rows = []
    for i in range(n):
        col = _row(gcm, i, gcm_vars + surf_vars)
        prof = {k: les[k][i] for k in ("U", "V", "THL", "QT", "QL", "PS")}
        prof["Rain"] = les["Rain"][i] if "Rain" in les else 0.0
        rl = les["rain_last"][i] if "rain_last" in les else 0.0
        r = set_les_forcings(col, _grid(zf, i), prof, dt, factor, rl, couple_surface)
        if zh is not None:
            r["idx"] = cloud_fraction_indices(_grid(zh, i), r["Zh"]).astype(numpy.int32)
        rows.append(r)
    return {k: numpy.stack([numpy.asarray(r[k]) for r in rows]) for k in rows[0]}


def backward_batched(gcm, Zf, les, zf, factor=1.0, dt=900.0, conservative=False, Zh=None, zh=None):
    n = gcm["T"].shape[0]
    rows = []
    for i in range(n):
        col = _row(gcm, i, gcm_vars)
        prof = {k: les[k][i] for k in ("U", "V", "THL", "QT", "QL", "QL_ice", "T", "A")}
        if conservative:
            prof["Rhobf"] = les["Rhobf"][i]
        r = set_gcm_tendencies(col, Zf[i], None if Zh is None else Zh[i], _grid(zf, i),
                               None if zh is None else _grid(zh, i), prof, dt, factor, conservative)
        rows.append(r)
    return {k: numpy.stack([numpy.asarray(r[k]) for r in rows]) for k in rows[0]}
