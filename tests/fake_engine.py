"""TEST-ONLY engine with the interface of sp_coupler_amd.engine.Engine, computing with the CPU oracle on
torch CPU tensors.  It exists so that the HOST LOGIC of the product (spcpl fan-out, profile registry,
driver sequencing, sharding, spifs writer) can be exercised by the `-m "not gpu"` suite; it is never
importable from the product package and never used on a GPU box's product path."""
import numpy
import torch

from oracle import spcpl_oracle as orc


def _np(d):
    return {k: v.numpy() for k, v in d.items()}


def _t(d, out=None):
    """results as torch CPU tensors; written INTO the caller's ``out`` tensors where given (the product hands
    views of its transfer buffers, like it does to the HIP engine's plans)"""
    res = {}
    for k, v in d.items():
        t = torch.from_numpy(numpy.ascontiguousarray(v))
        if out is not None and k in out:
            out[k].copy_(t.to(out[k].dtype))
            t = out[k]
        res[k] = t
    return res


def _into(res, out):
    if out:
        for k, t in out.items():
            if k in res:
                t.copy_(res[k].to(t.dtype))
                res[k] = t
    return res


class _OraclePlan:
    """plan object with the interface of sp_coupler_amd.engine._Plan (set_scalars / launch / outputs): the product's
    host logic drives the test engine exactly like the HIP one"""

    class _Args:
        factor = dt = 0.0

    def __init__(self, run, factor=0.0, dt=1.0):
        self._run, self.args, self.outputs = run, _OraclePlan._Args(), {}
        self.args.factor, self.args.dt = factor, dt

    def set_scalars(self, factor, dt):
        self.args.factor, self.args.dt = float(factor), float(dt)

    def launch(self, stream=None):
        self.outputs = self._run(self.args.factor, self.args.dt)
        return self.outputs


class OracleEngine:
    device, dtype = torch.device("cpu"), torch.float64

    stream = None

    def arena(self, specs, rows=None):
        from sp_coupler_amd.transfer import Arena
        return Arena(self.device, specs)

    def on_stream(self):
        import contextlib
        return contextlib.nullcontext()

    # -- the K7 operators (Engine.exner / interp / searchsorted / interp_c / rms) row by row through the oracle ----------
    @staticmethod
    def _rows(*ts):
        n = max(int(t.shape[0]) if t.dim() == 2 else 1 for t in ts)
        return n, [(lambda r, a=t.numpy(): a[r] if a.ndim == 2 else a) for t in ts]

    def exner(self, p, inverse=False, stream=None):
        return torch.from_numpy(numpy.asarray((orc.iexner if inverse else orc.exner)(p.numpy())))

    def interp(self, x, xp, fp, stream=None):
        if xp.shape[-1] != fp.shape[-1]:
            raise ValueError("fp and xp are not of the same length")
        n, (gx, gxp, gfp) = self._rows(x, xp, fp)
        with numpy.errstate(all="ignore"):
            out = numpy.stack([numpy.interp(gx(r), gxp(r), gfp(r)) for r in range(n)])
        return torch.from_numpy(out[0] if x.dim() == xp.dim() == fp.dim() == 1 else out)

    def searchsorted(self, a, v, side="left", stream=None):
        n, (ga, gv) = self._rows(a, v)
        out = numpy.stack([numpy.searchsorted(ga(r), gv(r), side=side) for r in range(n)]).astype(numpy.int64)
        return torch.from_numpy(out[0] if a.dim() == v.dim() == 1 else out)

    def interp_c(self, Zh, zh, q, rho=None, mode="interp_c", stream=None):
        n, (gZ, gz, gq) = self._rows(Zh, zh, q)
        gr = self._rows(rho)[1][0] if rho is not None else None
        rows = []
        with numpy.errstate(all="ignore"):
            for r in range(n):
                if mode == "interp_c":
                    rows.append(orc.interp_c(gZ(r), gz(r), gq(r), gr(r)))
                elif mode == "interp_rho":
                    rows.append(orc.interp_rho(gZ(r), gz(r), gq(r)))
                else:
                    Z = gZ(r)
                    vals = [orc.integral(Z[k + 1], Z[k], gz(r), gq(r), gr(r) if gr else None) for k in range(len(Z) - 1)]
                    rows.append(numpy.array([numpy.nan if v is None else v for v in vals], dtype=numpy.float64))
        out = numpy.stack(rows)
        return torch.from_numpy(out[0] if Zh.dim() == q.dim() == 1 else out)

    def rms(self, a, stream=None):
        x = a.numpy()
        return torch.from_numpy(numpy.asarray(orc.rms(x) if x.ndim == 1 else numpy.array([orc.rms(r) for r in x])))

    def to_devices(self, host_array, rows=None, n_cols=None):
        return torch.from_numpy(numpy.ascontiguousarray(host_array)).to(self.device, self.dtype)

    def plan_forward(self, g, zf, p, factor, dt, zh=None, **kw):
        return _OraclePlan(lambda f, d: self.forward(g, zf, p, f, d, zh=zh, **kw), factor, dt)

    def plan_backward(self, g, zf, p, factor, dt, **kw):
        return _OraclePlan(lambda f, d: self.backward(g, zf, p, f, d, **kw), factor, dt)

    def plan_exchange(self, g, zf, zh, p, factor_les, factor_gcm, dt, cols_per_block=0):
        lean = {k: v for k, v in p.items() if k not in ("Rain", "rain_last")}
        return (self.plan_forward(g, zf, lean, factor_les, dt, zh=zh, want_profiles=False, want_heights=False),
                self.plan_backward(g, zf, p, factor_gcm, dt, Zf=None))

    def plan_diagnostics(self, g, zf=None, prof=None, out=None, **kw):
        return _OraclePlan(lambda f, d: _into(self.diagnostics(g, zf, prof), out))

    def plan_cloud_indices(self, zh, Zh, out=None, **kw):
        return _OraclePlan(lambda f, d: _into({"idx": self.cloud_indices(zh, Zh)}, {"idx": out} if out is not None else None))

    def forward(self, g, zf, p, factor, dt, zh=None, want_profiles=False, want_heights=True, couple_surface=False, out=None, **kw):
        pn = _np(p)
        r = orc.forward_batched(_np(g), pn, zf.numpy(), None if zh is None else zh.numpy(), factor, dt,
                                couple_surface=couple_surface)
        keep = ["f_u", "f_v", "f_thl", "f_qt", "f_ql", "ql_ref", "f_ps"]
        if want_profiles:
            keep += ["u", "v", "thl", "qt", "ps"]
        if want_heights:
            keep += ["Zf", "Zh"]
        if zh is not None:
            keep += ["idx"]
        if "Rain" in pn and "rain_last" in pn:
            keep += ["rainrate"]
        if couple_surface:
            keep += ["z0m", "z0h", "wthl", "wqt"]
        return _t({k: r[k] for k in keep}, out)

    def backward(self, g, zf, p, factor, dt, Zf=None, conservative=False, zh=None, Zh=None, out=None, **kw):
        gn = _np(g)
        if Zf is None:
            Zf_n = (gn["Zgfull"] - gn["Zghalf"][:, -1:]) / orc.grav
        else:
            Zf_n = Zf.numpy()
        Zh_n = (gn["Zghalf"] - gn["Zghalf"][:, -1:]) / orc.grav if (conservative and Zh is None) else (
            None if Zh is None else Zh.numpy())
        pn = _np(p)
        pn.setdefault("THL", numpy.zeros_like(pn["T"]))   # only feeds the oracle's `t` diagnostic
        r = orc.backward_batched(gn, Zf_n, pn, zf.numpy(), factor, dt, conservative=conservative, Zh=Zh_n,
                                 zh=None if zh is None else zh.numpy())
        res = {k: r[k] for k in ("f_T", "f_SH", "f_QL", "f_QI", "f_U", "f_V", "f_A")}
        res["start_index"] = r["start_index"].astype(numpy.int32)
        return _t(res, out)

    def cloud_indices(self, zh, Zh, **kw):
        zhn, Zhn = zh.numpy(), Zh.numpy()
        rows = [orc.cloud_fraction_indices(zhn if zhn.ndim == 1 else zhn[i], Zhn[i]) for i in range(Zhn.shape[0])]
        return torch.from_numpy(numpy.stack(rows).astype(numpy.int32))

    def diagnostics(self, g, zf=None, prof=None, **kw):
        gn = _np(g)
        n = gn["T"].shape[0]
        out = {k: [] for k in ("Tv", "THL", "QT", "Zf", "Zh")}
        les = {k: [] for k in ("pf", "t", "ql_water")}
        for i in range(n):
            col = {k: gn[k][i] for k in ("T", "SH", "QL", "QI", "Zgfull", "Zghalf")}
            col.update(U=gn["T"][i], V=gn["T"][i], A=gn["T"][i], Pfull=gn["Pfull"][i], Phalf=gn["Zghalf"][i])
            z = numpy.zeros(2) if zf is None else (zf.numpy() if zf.dim() == 1 else zf.numpy()[i])
            c = orc.convert_profiles(col, z)
            for k in out:
                out[k].append(c[k])
            if zf is not None and prof is not None:
                pn = {k: v.numpy()[i] for k, v in prof.items()}
                pf = orc.interp(z, c["Zf"][::-1], col["Pfull"][::-1])
                les["pf"].append(pf)
                les["t"].append(pn["THL"] * orc.exner(pf) + orc.rlv * pn["QL"] / orc.cp)
                les["ql_water"].append(pn["QL"] - pn["QL_ice"])
        res = {k: numpy.stack(v) for k, v in out.items()}
        if les["pf"]:
            res.update({k: numpy.stack(v) for k, v in les.items()})
        return _t(res)

    def variability_nudge(self, qt, qsat, R, ql_av, qt_av, ql_ref, presf=None, thl=None, ql=None, constantT=False, **kw):
        """oracle/vnudge_oracle.py per column; qt (and thl) updated in place like the HIP engine does"""
        from oracle import vnudge_oracle as vo
        n, _, _, k = qt.shape
        res = {name: numpy.empty((n, k)) for name in ("beta", "a", "qt_std")}
        res["status"] = numpy.empty((n, k), dtype=numpy.int32)
        for i in range(n):
            r = vo.variability_nudge(qt[i].numpy(), qsat[i].numpy(), ql_av[i].numpy(), qt_av[i].numpy(), presf[i].numpy(),
                                     ql_ref[i].numpy(), R[i].numpy(), 1.0, constantT,
                                     thl=None if thl is None else thl[i].numpy(), ql=None if ql is None else ql[i].numpy())
            assert r["error"] is None, r["error"]
            qt[i].copy_(torch.from_numpy(r["qt"]))
            if constantT:
                thl[i].copy_(torch.from_numpy(r["thl"]))
            for name in ("beta", "a", "qt_std", "status"):
                res[name][i] = r[name]
        return {name: torch.from_numpy(v) for name, v in res.items()}

    def surface_fluxes(self, Ph_s, T_s, QLflux, QIflux, SHflux, TSflux, **kw):
        rho = Ph_s.numpy() / (orc.rd * T_s.numpy())
        wqt = -(QLflux.numpy() + QIflux.numpy() + SHflux.numpy()) / rho
        wthl = -TSflux.numpy() * orc.iexner(Ph_s.numpy()) / (orc.cp * rho)
        return torch.from_numpy(wthl), torch.from_numpy(wqt)
