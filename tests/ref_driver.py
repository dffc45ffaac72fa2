"""Test-side reference driver: splib.step's sequencing (splib/splib.py:267-352) executed with the
ORACLE's per-column functions (serial Python loop over columns, like the reference), on the same
synthetic model objects as the product driver.  Used to check the closed loop end to end."""
import numpy

from oracle import spcpl_oracle as orc


class RefCoupler:
    def __init__(self, gcm, les_models, cplsurf=False, les_forcing_factor=1.0, gcm_forcing_factor=1.0,
                 conservative=False):
        self.gcm, self.les_models = gcm, list(les_models)
        self.conservative = conservative
        self.cplsurf, self.lf, self.gf = cplsurf, les_forcing_factor, gcm_forcing_factor
        self.firststep = True
        self.profiles = {}
        self.log = []           # every setter call: (kind, column, name, array)

    def step(self):
        gcm = self.gcm
        t, dt = gcm.get_model_time(), gcm.get_timestep()
        gcm.evolve_model_until_cloud_scheme()
        gcm.evolve_model_cloud_scheme()
        gcm.step += 1
        cols = [les.grid_index for les in self.les_models]
        data = {v: gcm.get_profile_fields(v, cols) for v in orc.gcm_vars}                     # spcpl.py:62-67
        if self.cplsurf:
            data.update({v: gcm.get_surface_field(v, cols) for v in orc.surf_vars})
        heights = {}
        for i, les in enumerate(self.les_models):                                             # splib.py:317-323
            col = {k: v[i] for k, v in data.items()}
            if self.firststep:
                prof = {"U": les.get_profile_U(), "V": les.get_profile_V(), "THL": les.get_profile_THL(),
                        "QT": les.get_profile_QT(), "QL": les.get_profile_QL(), "PS": les.get_surface_pressure(),
                        "Rain": les.get_rain()}
            else:
                prof = self.profiles[les]
            f = orc.set_les_forcings(col, les.zf_cache, prof, dt, self.lf, getattr(les, "rain", 0.0), self.cplsurf)
            les.rain = prof["Rain"]
            heights[les] = (f["Zf"], f["Zh"])
            for name, key in (("U", "f_u"), ("V", "f_v"), ("THL", "f_thl"), ("QT", "f_qt"), ("surface_pressure", "f_ps"),
                              ("QL", "f_ql")):
                getattr(les, "set_tendency_" + name)(f[key])
                self.log.append(("les", les.grid_index, key, numpy.array(f[key])))
            les.set_ref_profile_QL(f["ql_ref"])
            self.log.append(("les", les.grid_index, "ql_ref", numpy.array(f["ql_ref"])))
            if self.cplsurf:
                les.set_z0m_surf(f["z0m"]); les.set_z0h_surf(f["z0h"]); les.set_wt_surf(f["wthl"]); les.set_wq_surf(f["wqt"])
                for key in ("z0m", "z0h", "wthl", "wqt"):
                    self.log.append(("les", les.grid_index, key, numpy.array(f[key])))
        new_profiles = {}
        for les in self.les_models:                                                           # splib.py:554-594
            les.evolve_model(t + dt, exactEnd=True)
            Zf, Zh = heights[les]
            idx = orc.cloud_fraction_indices(les.zh_cache, Zh)                                # spcpl.py:764
            self.log.append(("idx", les.grid_index, "idx", numpy.array(idx)))
            new_profiles[les] = {"U": les.get_profile_U(), "V": les.get_profile_V(), "presf": les.get_presf(),
                                 "Rhof": les.get_rhof(), "Rhobf": les.get_rhobf(), "THL": les.get_profile_THL(),
                                 "QT": les.get_profile_QT(), "QL": les.get_profile_QL(),
                                 "QL_ice": les.get_profile_QL_ice(), "QR": les.get_profile_QR(),
                                 "PS": les.get_surface_pressure(), "T": les.get_profile_T(),
                                 "A": les.get_cloudfraction(idx), "Rain": les.get_rain()}
        self.profiles = new_profiles
        for i, les in enumerate(self.les_models):                                             # splib.py:330-332
            col = {k: v[i] for k, v in data.items()}
            Zf, Zh = heights[les]
            b = orc.set_gcm_tendencies(col, Zf, Zh, les.zf_cache, les.zh_cache, self.profiles[les], dt, self.gf,
                                       conservative=self.conservative)
            for var, key in (("U", "f_U"), ("V", "f_V"), ("T", "f_T"), ("SH", "f_SH"), ("QL", "f_QL"), ("QI", "f_QI"),
                             ("A", "f_A")):
                gcm.set_profile_tendency(var, les.grid_index, b[key])
                self.log.append(("gcm", les.grid_index, key, numpy.array(b[key])))
        gcm.evolve_model_from_cloud_scheme()
        self.firststep = False
