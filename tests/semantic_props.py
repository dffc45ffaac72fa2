"""Semantic (metamorphic) properties of the coupling step that follow from the REFERENCE'S TEXT, not from the restatement in
oracle/spcpl_oracle.py (round-4 verdict, weak 1 / next 2).

The reference's own fixtures pin four functions (exner, iexner, rms, the cloud-fraction index map); everything else in the
oracle is a line-by-line restatement, and the kernels are compared with that restatement -- a transcription slip copied into
both (a swapped argument, a wrong sign of rd/cp, the ql_water / ql_ice split) would pass every parity test.  Each property
below states something the reference's lines imply whatever the implementation, evaluates its expected value with plain NumPy
(``numpy.interp`` where the reference calls it, counts and overlaps by brute force), and is run against TWO implementations:
the NumPy oracle on the CPU (tests/test_semantic_oracle.py) and the HIP kernels through the C ABI (tests/test_semantic_gpu.py).
``tools/mutation_control.py`` shows that each property fails when the kernel line it guards is perturbed
(profiles/r05_mutation_control.log).

An implementation is an object with
    forward(gcm, zf, zh, prof, factor, dt)           -> dict f_u f_v f_thl f_qt f_ql ql_ref f_ps idx u v thl qt
    backward(gcm, zf, zh, prof, factor, dt, conservative) -> dict f_T f_SH f_QL f_QI f_U f_V f_A start_index
    interp_c(Zh, zh, q, rho) / interp_rho(Zh, zh, rho) -> [n x nG]
    les_temperature(gcm, zf, prof)                   -> (pf, t) on LES levels (spcpl.py:408-409)
    surface(gcm, zf, zh, prof)                       -> dict z0m z0h wthl wqt: the surface branch of set_les_forcings (spcpl.py:359-364)
    surface_alone(Ph_s, T_s, QLflux, QIflux, SHflux, TSflux) -> (wthl, wqt): convert_surface_fluxes on its own (spcpl.py:136-167)
    gcm_diagnostics(gcm)                             -> dict Tv THL QT Zf Zh on GCM levels (spcpl.py:176, 197-198, 214-215; output_column_conversion :251-267)
    rainrate(gcm, zf, zh, prof)                      -> [n]: (rain - rain_last) / dt of set_les_forcings (spcpl.py:316-325), prof with Rain, rain_last
    nudge(fields, R, constantT)                      -> dict qt thl beta a qt_std of ONE LES after spcpl.variability_nudge (spcpl.py:613-744);
                                                        fields: qt qsat thl ql [itot x jtot x k], ql_av qt_av presf ql_ref [k]; R [itot x jtot]
all on NumPy arrays [n_cols x n_lev].
"""
import numpy

from sp_coupler_amd import synthetic

EPS = 2.220446049250313e-16
# splib/sputils.py:14-20
pref0, rd, cp, rlv, grav = 1e5, 287.04, 1004., 2.53e6, 9.81
DT = 900.0
FORCINGS = ("f_u", "f_v", "f_thl", "f_qt", "f_ql")
TENDENCIES = ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A")


#: level geometry the properties run on; the test files also run them at BASELINE config 5's 137 <-> 512 (set_geometry)
GEOMETRY = {"nG": 91, "nL": 160}


def set_geometry(nG, nL):
    GEOMETRY.update(nG=nG, nL=nL)


def batch(n=40, seed=777):
    return synthetic.make_batch(n, GEOMETRY["nG"], GEOMETRY["nL"], seed, couple_surface=False)


def heights(gcm):
    """Zf, Zh as splib/spcpl.py:197-198 writes them: (Zg - Zghalf[-1]) / grav"""
    zs = gcm["Zghalf"][:, -1:]
    return (gcm["Zgfull"] - zs) / grav, (gcm["Zghalf"] - zs) / grav


# (a) spcpl.py:328-333: f_x = factor (x - x_d) / dt -- LES slab means equal to the forward-interpolated GCM profile
#     => every forcing is exactly zero, whatever factor and dt are
def prop_zero_forcings_when_the_les_equals_the_interpolated_gcm_profile(impl):
    gcm, zf, zh, prof = batch()
    r = impl.forward(gcm, zf, zh, prof, 1.0, DT)
    same = dict(prof, U=r["u"], V=r["v"], THL=r["thl"], QT=r["qt"], QL=r["ql_ref"], PS=gcm["Phalf"][:, -1].copy())
    r2 = impl.forward(gcm, zf, zh, same, 1.7, 450.0)
    for k in FORCINGS + ("f_ps",):
        assert (r2[k] == 0).all(), "%s: %d elements are not zero" % (k, int((r2[k] != 0).sum()))
    # ... and each forcing IS that line of the reference applied to the profile the implementation returns (sign, factor,
    # dt, and which LES mean belongs to which field): spcpl.py:328-333
    for k, x, x_d in (("f_u", "u", "U"), ("f_v", "v", "V"), ("f_thl", "thl", "THL"), ("f_qt", "qt", "QT"), ("f_ql", "ql_ref", "QL")):
        assert numpy.array_equal(r[k], 1.0 * (r[x] - prof[x_d]) / DT), k
    assert numpy.array_equal(r["f_ps"], 1.0 * (gcm["Phalf"][:, -1] - prof["PS"]) / DT)


# (b) spcpl.py:214 + sputils.py:28-34: thl_ = (T - rlv (QL + QI) / cp) iexner(Pf) -- an isentropic, condensate-free column
#     T = theta (p / pref0) ** (rd / cp) has thl == theta at EVERY level, hence at every interpolated LES level
def prop_isentropic_column_has_constant_thl(impl):
    gcm, zf, zh, prof = batch()
    theta = 300.0
    g = dict(gcm, T=theta * (gcm["Pfull"] / pref0) ** (rd / cp), QL=numpy.zeros_like(gcm["QL"]), QI=numpy.zeros_like(gcm["QI"]))
    r = impl.forward(g, zf, zh, prof, 1.0, DT)
    err = numpy.abs(r["thl"] - theta).max()
    assert err <= 4 * numpy.spacing(theta), "thl differs from theta by %.3e (%.1f ulp)" % (err, err / numpy.spacing(theta))
    # the backward direction, spcpl.py:408-409: t = thl_d exner(pf) + rlv ql_d / cp with thl_d == theta, ql_d == 0 gives back the
    # isentropic temperature at the interpolated pressure
    p2 = dict(prof, THL=numpy.full_like(prof["THL"], theta), QL=numpy.zeros_like(prof["QL"]), QL_ice=numpy.zeros_like(prof["QL"]))
    pf, t = impl.les_temperature(g, zf, p2)
    want = theta * (pf / pref0) ** (rd / cp)
    assert (numpy.abs(t - want) <= 4 * numpy.spacing(want)).all()
    # and a condensate load lowers thl by exactly rlv q / cp times the same Exner factor (the sign of the latent term)
    g2 = dict(g, QL=numpy.full_like(gcm["QL"], 1e-3))
    r2 = impl.forward(g2, zf, zh, prof, 1.0, DT)
    lo = r["thl"] - r2["thl"]
    assert (lo > 0).all() and (lo < rlv * 1e-3 / cp * 2.0).all() and (lo > rlv * 1e-3 / cp * 0.9).all()


# (c) spcpl.py:402, 519-521: f_SH + f_QL + f_QI = factor (interp(qt_d) - (SH + QL + QI)) / dt -- the condensate split
#     (ql_water = ql - ql_ice) cancels in the sum because interpolation is linear
def prop_total_water_tendency_closes(impl):
    gcm, zf, zh, prof = batch()
    Zf, _ = heights(gcm)
    factor = 0.6
    r = impl.backward(gcm, zf, zh, prof, factor, DT, False)
    got = r["f_SH"] + r["f_QL"] + r["f_QI"]
    for c in range(gcm["T"].shape[0]):
        si = int(r["start_index"][c])
        want = factor * (numpy.interp(Zf[c], zf, prof["QT"][c]) - (gcm["SH"][c] + gcm["QL"][c] + gcm["QI"][c])) / DT
        tol = 16 * EPS * numpy.abs(prof["QT"][c]).max() * factor / DT
        assert numpy.abs(got[c, si:] - want[si:]).max() <= tol, (c, numpy.abs(got[c, si:] - want[si:]).max(), tol)
        assert (got[c, :si] == 0).all()
    # each part alone has the sign the reference's comment gives it: with QL_ice == QL all condensate is ice -> f_QL = -QL / dt
    p2 = dict(prof, QL_ice=prof["QL"].copy())
    r2 = impl.backward(gcm, zf, zh, p2, 1.0, DT, False)
    for c in range(gcm["T"].shape[0]):
        si = int(r2["start_index"][c])
        assert numpy.array_equal(r2["f_QL"][c, si:], 1.0 * (0.0 - gcm["QL"][c, si:]) / DT)


def _overlap_integral(z, w, lo, hi):
    """integral of the piecewise-constant w (w[j] on [z[j], z[j+1]]) over [lo, hi], by brute force over the cells"""
    tot = 0.0
    for j in range(len(z) - 1):
        tot += w[j] * max(0.0, min(z[j + 1], hi) - max(z[j], lo))
    return tot


# (d) sputils.py:94-189: interp_c returns the rho-weighted MEAN of the piecewise-constant profile over each coarse layer, so
#     sum over the covered layers of mean x (integral of rho over the layer) = integral of rho q over their union; a constant
#     profile returns that constant; layers whose top lies at or above the fine grid's top stay zero
def prop_conservative_coarsening_conserves(impl):
    gcm, zf, zh, prof = batch(24)
    _, Zh = heights(gcm)
    q, rho = prof["QT"], prof["Rhobf"]
    Q = impl.interp_c(Zh, zh, q, rho)
    R = impl.interp_rho(Zh, zh, rho)                      # layer-mean density: integral of rho over the layer / its depth
    const = impl.interp_c(Zh, zh, numpy.full_like(q, 0.0123), rho)
    for c in range(q.shape[0]):
        covered = Zh[c, :-1] < zh[-1]                     # sputils.py:186
        assert covered.any() and not covered.all()
        assert (Q[c, ~covered] == 0).all() and (const[c, ~covered] == 0).all()
        assert numpy.abs(const[c, covered] - 0.0123).max() <= 8 * EPS * 0.0123
        top = Zh[c, :-1][covered].max()
        mass = (Q[c, covered] * R[c, covered] * (Zh[c, :-1] - Zh[c, 1:])[covered]).sum()
        want = _overlap_integral(zh, rho[c] * q[c], 0.0, top)
        assert abs(mass - want) <= 1e-12 * abs(want), (c, mass, want)
        # every layer mean lies between the smallest and the largest cell value it covers
        assert (Q[c, covered] >= q[c].min() * (1 - 1e-12)).all() and (Q[c, covered] <= q[c].max() * (1 + 1e-12)).all()
    # the same through the backward pass's conservative branch (spcpl.py:479-489): with a GCM state of zero, factor = dt = 1,
    # the tendencies ARE the layer means
    zero = {k: numpy.zeros_like(v) for k, v in gcm.items() if k in ("T", "SH", "QL", "QI", "U", "V", "A")}
    g0 = dict(gcm, **zero)
    r = impl.backward(g0, zf, zh, prof, 1.0, 1.0, True)
    QT = impl.interp_c(Zh, zh, prof["T"], rho)
    for c in range(q.shape[0]):
        si = int(r["start_index"][c])
        assert numpy.array_equal(r["f_T"][c, si:], QT[c, si:])
        assert numpy.array_equal(r["f_U"][c, si:], impl.interp_c(Zh[c:c + 1], zh, prof["U"][c:c + 1], rho[c:c + 1])[0, si:])


# (e) spcpl.py:498, 527-533: start_index = searchsorted(-Zf, -h[-1]) = the number of GCM full levels above the LES top;
#     f[0:start_index] *= 0 leaves +-0 (or NaN) there and touches nothing below
def prop_masking_above_the_les_top(impl):
    gcm, zf, zh, prof = batch()
    Zf, _ = heights(gcm)
    r = impl.backward(gcm, zf, zh, prof, 1.0, DT, False)
    n, nG = gcm["T"].shape
    si_brute = (Zf > zf[-1]).sum(axis=1)                  # Zf descends with the level index: the levels above the LES top
    assert numpy.array_equal(r["start_index"], si_brute)
    assert (si_brute > 0).all() and (si_brute < nG).all()
    above = numpy.arange(nG)[None, :] < si_brute[:, None]
    for k in TENDENCIES:
        f = r[k]
        assert ((f[above] == 0) | numpy.isnan(f[above])).all(), k
        if k in ("f_T", "f_SH", "f_U", "f_V", "f_A"):     # random inputs: a zero below the LES top would be a masked level
            assert (f[~above] != 0).all(), k
    # the sign survives the multiplication by zero (x *= 0 keeps the sign of x): spcpl.py:527
    assert numpy.signbit(r["f_T"][above]).any() and (~numpy.signbit(r["f_T"][above])).any()


# (f) spcpl.py:26 / 764: idx = searchsorted(zh, Zh, side='right')[:-1][::-1] -- idx[m] is the NUMBER of LES half levels at or
#     below GCM half level nG-1-m, equal heights included
def prop_index_map_is_a_count(impl):
    gcm, zf, zh, prof = batch()
    n, nG = gcm["T"].shape
    # make some GCM half levels coincide exactly with LES half levels (side='right' counts them)
    g = {k: v.copy() for k, v in gcm.items()}
    zs = g["Zghalf"][:, -1:].copy()
    _, Zh0 = heights(gcm)
    cands = zh[(zh > 0) & (zh < Zh0[:, nG - 21].min())]               # LES half levels below every column's 21st-lowest GCM half level
    target = cands[numpy.linspace(0, len(cands) - 1, 20).astype(int)]
    assert len(numpy.unique(target)) == 20
    Zgh = g["Zghalf"]
    Zgh[:, nG - 20:nG] = (grav * target[::-1])[None, :] + zs
    g["Zghalf"] = Zgh
    g["Zgfull"] = 0.5 * (Zgh[:, :-1] + Zgh[:, 1:])
    _, Zh = heights(g)
    hits = numpy.isin(Zh, zh).sum()
    assert hits >= 4 * n, "the construction does not produce exactly equal heights (%d)" % hits     # (grav x height / grav is exact for most, not all)
    r = impl.forward(g, zf, zh, prof, 1.0, DT)
    want = numpy.empty((n, nG), dtype=numpy.int64)
    for m in range(nG):
        want[:, m] = (zh[None, :] <= Zh[:, nG - 1 - m][:, None]).sum(axis=1)
    assert numpy.array_equal(r["idx"].astype(numpy.int64), want)
    assert (numpy.diff(want, axis=1) >= 0).all() and want.max() == len(zh)


# (g) spcpl.py:224-228: the GCM arrays run top-down, numpy.interp needs ascending abscissae: u = interp(h, Zf[::-1], U[::-1]) --
#     the kernels never materialise the reversal (index arithmetic while staging)
def prop_reversal_is_index_arithmetic_only(impl):
    gcm, zf, zh, prof = batch()
    Zf, _ = heights(gcm)
    r = impl.forward(gcm, zf, zh, prof, 1.0, DT)
    for c in range(gcm["T"].shape[0]):
        assert numpy.array_equal(r["u"][c], numpy.interp(zf, Zf[c, ::-1], gcm["U"][c, ::-1])), c
        assert numpy.array_equal(r["v"][c], numpy.interp(zf, Zf[c, ::-1], gcm["V"][c, ::-1])), c
        assert numpy.array_equal(r["ql_ref"][c], numpy.interp(zf, Zf[c, ::-1], gcm["QL"][c, ::-1])), c
        qt_ = gcm["SH"][c] + gcm["QL"][c] + gcm["QI"][c]                                   # spcpl.py:215: ALL water, ice included
        assert numpy.array_equal(r["qt"][c], numpy.interp(zf, Zf[c, ::-1], qt_[::-1])), c
    # cloud fraction comes back reversed: A_d = profile["A"][::-1] (spcpl.py:404), f_A = factor (A_d - A) / dt (spcpl.py:526)
    b = impl.backward(gcm, zf, zh, prof, 1.0, DT, False)
    for c in range(gcm["T"].shape[0]):
        si = int(b["start_index"][c])
        assert numpy.array_equal(b["f_A"][c, si:], 1.0 * (prof["A"][c, ::-1] - gcm["A"][c])[si:] / DT)


# (h) splib.py:317, 330: every column is coupled on its own -- permuting the columns permutes the results, and a column's
#     result does not depend on what else is in the batch
def prop_columns_are_independent(impl):
    n = 2601          # enough columns for the kernels to put SEVERAL into one workgroup's slab (and an odd tail)
    gcm, zf, zh, prof = batch(n)
    perm = numpy.random.default_rng(3).permutation(n)
    r, b = impl.forward(gcm, zf, zh, prof, 1.0, DT), impl.backward(gcm, zf, zh, prof, 1.0, DT, False)
    gp, pp = {k: numpy.ascontiguousarray(v[perm]) for k, v in gcm.items()}, {k: numpy.ascontiguousarray(v[perm]) for k, v in prof.items()}
    rp, bp = impl.forward(gp, zf, zh, pp, 1.0, DT), impl.backward(gp, zf, zh, pp, 1.0, DT, False)
    for k in FORCINGS + ("ql_ref", "f_ps", "idx"):
        assert numpy.array_equal(rp[k], r[k][perm]), k
    for k in TENDENCIES + ("start_index",):
        assert numpy.array_equal(bp[k], b[k][perm], equal_nan=True), k
    one = impl.forward({k: v[5:6] for k, v in gcm.items()}, zf, zh, {k: v[5:6] for k, v in prof.items()}, 1.0, DT)
    for k in FORCINGS:
        assert numpy.array_equal(one[k][0], r[k][5]), k


# (i) spcpl.py:136-167: rho = Ph[-1] / (rd T[-1]) (ideal gas at the surface: SURFACE pressure, LOWEST full level);
#     wqt = -(QLflux + QIflux + SHflux) / rho (all three water fluxes, OpenIFS positive downward -> DALES positive upward);
#     wthl = -TSflux iexner(Ph[-1]) / (cp rho) (only the SENSIBLE heat); z0m / z0h handed through
def prop_surface_fluxes_are_the_ifs_fluxes_over_the_surface_density(impl):
    gcm, zf, zh, prof = synthetic.make_batch(40, GEOMETRY["nG"], GEOMETRY["nL"], 778, couple_surface=True)
    n = gcm["T"].shape[0]
    # a surface state whose density is EXACTLY one: Ph_s = rd T_s, so the kinematic fluxes are the mass fluxes with the sign turned
    g = {k: v.copy() for k, v in gcm.items()}
    g["T"][:, -1] = numpy.linspace(250.0, 310.0, n)
    g["Phalf"][:, -1] = rd * g["T"][:, -1]
    ways = {"in the forward pass": impl.surface(g, zf, zh, prof),
            "on its own": dict(zip(("wthl", "wqt"), impl.surface_alone(g["Phalf"][:, -1], g["T"][:, -1], g["QLflux"], g["QIflux"], g["SHflux"], g["TSflux"])))}
    for way, r in ways.items():
        assert numpy.array_equal(r["wqt"], -(g["QLflux"] + g["QIflux"] + g["SHflux"])), "wqt " + way
        want = -g["TSflux"] * (g["Phalf"][:, -1] / pref0) ** (-rd / cp) / cp
        assert (numpy.abs(r["wthl"] - want) <= 4 * numpy.spacing(numpy.abs(want))).all(), "wthl " + way
    r = ways["in the forward pass"]
    assert numpy.array_equal(r["z0m"], g["Z0M"]) and numpy.array_equal(r["z0h"], g["Z0H"])
    # signs: a downward (positive, OpenIFS) moisture / sensible heat flux is a negative (upward-positive, DALES) kinematic flux
    assert (numpy.sign(r["wqt"]) == -numpy.sign(g["QLflux"] + g["QIflux"] + g["SHflux"])).all()
    assert (numpy.sign(r["wthl"]) == -numpy.sign(g["TSflux"])).all()
    # which inputs matter: the density is that of the LOWEST full level and the SURFACE half level -- every other level can
    # change without a trace, a lowest level twice as warm is air half as dense (an exact power of two through every step)
    g2 = {k: v.copy() for k, v in g.items()}
    g2["T"][:, :-1] *= 1.25
    g2["Phalf"][:, :-1] *= 0.75
    r2 = impl.surface(g2, zf, zh, prof)
    assert numpy.array_equal(r2["wqt"], r["wqt"]) and numpy.array_equal(r2["wthl"], r["wthl"])
    g3 = {k: v.copy() for k, v in g.items()}
    g3["T"][:, -1] *= 2.0
    r3 = impl.surface(g3, zf, zh, prof)
    assert numpy.array_equal(r3["wqt"], 2.0 * r["wqt"]) and numpy.array_equal(r3["wthl"], 2.0 * r["wthl"])
    # which fluxes matter: the latent heat flux TLflux nowhere, TSflux only in wthl, each water flux in wqt with weight one
    for name in ("QLflux", "QIflux", "SHflux"):
        g4 = {k: v.copy() for k, v in g.items()}
        g4[name] = g[name] + 0.25
        r4 = impl.surface(g4, zf, zh, prof)
        assert numpy.array_equal(r4["wthl"], r["wthl"]), name
        assert numpy.array_equal(r4["wqt"], -(g4["QLflux"] + g4["QIflux"] + g4["SHflux"])), name
    g5 = {k: v.copy() for k, v in g.items()}
    g5["TSflux"] = 4.0 * g["TSflux"]
    g5["TLflux"] = g["TLflux"] - 7.0
    r5 = impl.surface(g5, zf, zh, prof)
    assert numpy.array_equal(r5["wqt"], r["wqt"]) and numpy.array_equal(r5["wthl"], 4.0 * r["wthl"])


# (j) spcpl.py:613-744, the variability nudge of one LES, level by level:
#     * GCM cloud (ql_ref > 1e-9): beta with  mean(max(beta (qt - qt_av) + qt_av - qsat, 0)) = ql_ref  (:646-648, 676), applied as
#       qt += (beta - 1)(qt - qt_av) (:724-725) -- so the level's mean condensate BECOMES ql_ref and its mean qt stays;
#     * no root in [0, 5] (:669-673) -> additive noise a R with mean(max(qt + a R - qsat, 0)) = ql_ref, only if that means MORE
#       cloud (:712-719); beta = 1 (:722);
#     * GCM clear but LES cloudy (:679-683): beta that makes the wettest point exactly saturated;
#     * otherwise untouched (:697);  constantT (:726-733): thl moves so that T = thl exner(p) + rlv ql / cp stays, point by point.
def prop_variability_nudge_reaches_the_gcm_cloud_amount(impl):
    from tests.test_vnudge import make_les_fields
    itot, jtot, ktot = 32, 24, 40
    f = make_les_fields(itot, jtot, ktot, seed=5)
    cloudy = numpy.nonzero(f["ql_av"] > 1e-6)[0]
    f["ql_ref"][cloudy[0]] = 5e-8                            # a GCM cloud amount between 1e-9 and 1e-6: still "significant" (:665)
    rng = numpy.random.default_rng(99)
    R = rng.normal(size=(itot, jtot))
    R -= R.sum() / (itot * jtot)
    r = impl.nudge(f, R, True)
    qt, qsat, N = f["qt"], f["qsat"], itot * jtot
    mean_ql = lambda x, k: numpy.maximum(x - qsat[:, :, k], 0).sum() / N         # noqa: E731
    seen = set()
    for k in range(ktot):
        x, av, ref = qt[:, :, k], f["qt_av"][k], f["ql_ref"][k]
        new = r["qt"][:, :, k]
        touched = True
        if ref > 1e-9:
            lo, hi = mean_ql(0 * (x - av) + av, k) - ref, mean_ql(5 * (x - av) + av, k) - ref
            if lo > 0 or hi < 0:                                                  # no root: additive noise, or nothing
                if ref > f["ql_av"][k]:
                    seen.add("additive")
                    assert r["a"][k] > 0 and numpy.array_equal(new, x + r["a"][k] * R), k
                    # brentq stops within xtol = 2e-12 (+ 4 eps a) of the root; the mean condensate moves by at most mean|R| per unit of a
                    assert abs(mean_ql(new, k) - ref) <= 4e-12 * numpy.abs(R).mean() + 4 * EPS * ref, (k, mean_ql(new, k), ref)
                else:
                    seen.add("no root, less cloud wanted")
                    assert numpy.array_equal(new, x), k
                assert r["beta"][k] == 1
            else:
                seen.add("multiplicative")
                b = r["beta"][k]
                assert 0 <= b < 5 and numpy.array_equal(new, x + (b - 1) * (x - av)), k
                assert abs(mean_ql(new, k) - ref) <= 4e-12 * numpy.abs(x - av).mean() + 4 * EPS * ref, (k, mean_ql(new, k), ref)
                assert abs(new.mean() - x.mean()) <= 8 * EPS * abs(av) * (1 + abs(b)), k        # the level's water is redistributed, not changed
        elif f["ql_av"][k] > ref:
            i, j = numpy.unravel_index(numpy.argmax(x - qsat[:, :, k]), x.shape)
            b = r["beta"][k]
            if (qsat[i, j, k] - av) / (x[i, j] - av) < 0:
                seen.add("mean already saturated")
                assert b == 1 and numpy.array_equal(new, x), k
            elif b == 1 and (qsat[i, j, k] - av) / (x[i, j] - av) >= 5:
                assert numpy.array_equal(new, x), k
            else:
                seen.add("barely unsaturated")
                assert numpy.array_equal(new, x + (b - 1) * (x - av)), k
                assert abs(new[i, j] - qsat[i, j, k]) <= 8 * EPS * qsat[i, j, k], k              # the wettest point: exactly saturated
        else:
            seen.add("untouched")
            touched = False
            assert r["beta"][k] == 1 and numpy.array_equal(new, x), k
            assert numpy.array_equal(r["thl"][:, :, k], f["thl"][:, :, k]), k
        if touched:                                                               # constantT: the temperature of every point stays
            ex = (f["presf"][k] / pref0) ** (rd / cp)
            T0 = f["thl"][:, :, k] * ex + rlv / cp * f["ql"][:, :, k]
            T1 = r["thl"][:, :, k] * ex + rlv / cp * numpy.maximum(new - qsat[:, :, k], 0)
            assert numpy.abs(T1 - T0).max() <= 16 * numpy.spacing(300.0), (k, numpy.abs(T1 - T0).max())
        assert abs(r["qt_std"][k] - numpy.sqrt(((new - new.mean()) ** 2).mean())) <= 1e-12 * new.mean(), k
    assert {"multiplicative", "additive", "barely unsaturated", "untouched"} <= seen, seen
    # without constantT thl is nobody's business
    r2 = impl.nudge(f, R, False)
    assert numpy.array_equal(r2["qt"], r["qt"]) and (r2["thl"] is None or numpy.array_equal(r2["thl"], f["thl"]))


# (k) spcpl.py:471-477, 518-525: f_x = factor (x_d at the GCM level - x_gcm) / dt for T, U, V -- a RELAXATION: applied for
#     dt / factor it lands the GCM level on the LES profile.  With LES profiles LINEAR in height the profile at a GCM height is
#     known in closed form (the line itself; the end values outside the LES levels, where numpy.interp clamps), whatever the
#     interpolation code does; three different lines, so that T, U and V cannot stand in for each other.
#     spcpl.py:316-325: rainrate = (rain - rain_last) / dt.
def prop_tendencies_relax_the_gcm_towards_the_les_profile(impl):
    gcm, zf, zh, prof = batch()
    Zf, _ = heights(gcm)
    n = gcm["T"].shape[0]
    lines = {"T": (285.0, -6.5e-3), "U": (3.0, 2.0e-3), "V": (-8.0, 5.0e-4)}
    p2 = dict(prof)
    for k, (a, b) in lines.items():
        p2[k] = numpy.tile(a + b * zf, (n, 1))
    factor = 0.8
    r = impl.backward(gcm, zf, zh, p2, factor, DT, False)
    zc = numpy.clip(Zf, zf[0], zf[-1])
    for k, (a, b) in lines.items():
        landed = gcm[k] + r["f_" + k] * DT / factor
        want = a + b * zc
        for c in range(n):
            si = int(r["start_index"][c])
            tol = 64 * EPS * max(numpy.abs(gcm[k][c]).max(), numpy.abs(want[c]).max())
            assert numpy.abs(landed[c, si:] - want[c, si:]).max() <= tol, (k, c, numpy.abs(landed[c, si:] - want[c, si:]).max(), tol)
    # the sign: an LES warmer than the GCM warms it
    p3 = dict(p2, T=numpy.tile(400.0 + 0 * zf, (n, 1)))
    r3 = impl.backward(gcm, zf, zh, p3, 1.0, DT, False)
    for c in range(n):
        si = int(r3["start_index"][c])
        assert (r3["f_T"][c, si:] > 0).all()
    # rain: the amount fallen since the last exchange per unit time (exactly representable numbers: the result is exact)
    rate = 2.0 ** -10 * (1 + numpy.arange(n) % 4)
    p4 = dict(prof, rain_last=numpy.full(n, 0.5), Rain=0.5 + DT * rate)
    assert numpy.array_equal(impl.rainrate(gcm, zf, zh, p4), rate)


# (l) spcpl.py:175-176, 197-198, 214-215 (and output_column_conversion, :251-267), the GCM-level diagnostics:
#     * Tv is the temperature at which DRY air has the density of the mixture: with vapour q_v and condensate q_c per unit mass,
#       p = rho T (rd (1 - q_v - q_c) + rv q_v) (Dalton: dry air + vapour carry the pressure, the condensate only weighs), so
#       p / (rd Tv) must be that density;
#     * QT is all water, THL of an isentropic condensate-free column is its potential temperature;
#     * Zh / Zf are heights above the SURFACE (the last half level): Zh ends with exactly 0, g Zh + Zg_surface gives the geopotential back.
def prop_gcm_level_diagnostics_mean_what_their_names_say(impl):
    gcm, zf, zh, prof = batch()
    rv = 461.5                                                                    # sputils.py:16
    d = impl.gcm_diagnostics(gcm)
    qv, qc, p = gcm["SH"], gcm["QL"] + gcm["QI"], gcm["Pfull"]
    rho = p / (gcm["T"] * (rd * (1 - qv - qc) + rv * qv))
    assert (numpy.abs(p / (rd * d["Tv"]) - rho) <= 8 * EPS * rho).all()
    assert (d["Tv"][qc > 1e-5] < (gcm["T"] * (1 + (rv / rd - 1) * qv))[qc > 1e-5]).all()     # condensate only weighs: it LOWERS Tv
    assert numpy.array_equal(d["QT"], gcm["SH"] + gcm["QL"] + gcm["QI"])
    theta = 300.0
    g = dict(gcm, T=theta * (gcm["Pfull"] / pref0) ** (rd / cp), QL=numpy.zeros_like(gcm["QL"]), QI=numpy.zeros_like(gcm["QI"]))
    assert (numpy.abs(impl.gcm_diagnostics(g)["THL"] - theta) <= 4 * numpy.spacing(theta)).all()
    assert (d["Zh"][:, -1] == 0).all() and (numpy.diff(d["Zh"], axis=1) < 0).all()
    zs = gcm["Zghalf"][:, -1:]
    assert (numpy.abs(d["Zh"] * grav + zs - gcm["Zghalf"]) <= 4 * EPS * numpy.abs(gcm["Zghalf"]).max()).all()
    assert (numpy.abs(d["Zf"] * grav + zs - gcm["Zgfull"]) <= 4 * EPS * numpy.abs(gcm["Zgfull"]).max()).all()


PROPERTIES = [v for k, v in sorted(globals().items()) if k.startswith("prop_")]
