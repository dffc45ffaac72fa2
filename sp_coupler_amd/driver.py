"""Per-step driver: the sequencing of ``splib.step`` / ``step_spinup`` (``splib/splib.py:267-352,
355-402``) around the batched coupling kernels -- gather -> forcings -> (LES) -> profiles -> tendencies,
``firststep`` handling, ``profiles`` carry-over and the ``timing.txt`` row.  It drives any pair of
model objects that satisfy the reference's duck-typed contract (``sp_coupler_amd.models`` ships an
in-process synthetic pair).  The orchestration around it (process launch, config, restart files) is
out of scope."""
import time

from . import spcpl
from .models import RequestsPool


class Coupler:
    def __init__(self, gcm, les_models, cplsurf=False, les_forcing_factor=1.0, gcm_forcing_factor=1.0,
                 conservative_coarsening=False, qt_forcing="sp", les_spinup=0, output_column_indices=None, write=False):
        self.ensemble = spcpl._is_ensemble(les_models)          # LES side offers the batched protocol (models.py)
        self.gcm, self.les_models = gcm, (les_models if self.ensemble else list(les_models))
        self.cplsurf = cplsurf                                   # splib/splib.py:67
        self.les_forcing_factor = les_forcing_factor             # splib/splib.py:57
        self.gcm_forcing_factor = gcm_forcing_factor             # splib/splib.py:46
        self.conservative_coarsening = conservative_coarsening   # splib/splib.py:69
        self.qt_forcing = qt_forcing                             # splib/splib.py:68
        self.les_spinup = les_spinup
        self.output_column_indices = output_column_indices
        self.write = write
        self.firststep = True
        self.profiles = {}
        self.timing_rows = []                                    # rows of timing.txt (splib/splib.py:340-343)
        if not self.ensemble:
            for les in self.les_models:
                if not hasattr(les, "zf_cache"):
                    les.zh_cache, les.zf_cache = les.get_zh(), les.get_zf()    # splib/splib.py:152-153

    def initialize_output(self, les_spinup_steps=1):
        """splib.initialize (splib/splib.py:192-193): the spifs record the FIRST step writes into is opened at
        initialisation with the spin-up time stamp; ``step`` appends a record only from the second step on."""
        if self.write and spcpl.writer is not None:
            spcpl.writer.update_time(self.les_spinup / max(1, les_spinup_steps))

    # splib/splib.py:554-594 (async branch): evolve every LES, then fetch its slab means
    def step_les_models(self, model_time, offset=0):
        if self.ensemble:                       # one evolve call and one getter round for ALL columns
            t0 = time.time()
            self.les_models.evolve_model_batched(model_time + offset)
            wall = time.time() - t0
            diag = self.conservative_coarsening or (self.write and spcpl.writer is not None)
            return [wall], spcpl.get_les_profiles_batched(self.les_models, True, diagnostics=diag)
        pool = RequestsPool()
        reqs, profile_reqs = [], {}
        pending = pool.requests
        for les in self.les_models:
            req = les.evolve_model(model_time + offset, exactEnd=True)
            reqs.append(req)
            pending.append(req)
            prof = spcpl.get_les_profiles(les, True)
            profile_reqs[les] = prof
            pending.extend(prof.values())
        pool.waitall()
        # the request dicts go on as they are: spcpl resolves them column-block-wise when it packs the next launch
        # (the reference resolves them here, one .result() per variable and column, splib.py:586-590)
        walls = [r.result() if hasattr(r, "result") else 0.0 for r in reqs]
        return walls, profile_reqs

    # splib/splib.py:267-352
    def step(self):
        gcm = self.gcm
        t = gcm.get_model_time()
        delta_t = gcm.get_timestep()
        starttime = time.time()
        w1 = -time.time()
        if self.write and spcpl.writer is not None:
            if not self.firststep:                                            # splib.py:287-288
                spcpl.writer.update_time(t + self.les_spinup + delta_t)
            elif spcpl.writer.step < 0:                                       # record 0 comes from initialize()
                self.initialize_output()
        if gcm.first_half_step_done:
            gcm.first_half_step_done = False
        else:
            gcm.evolve_model_until_cloud_scheme()
            gcm.evolve_model_cloud_scheme()
        w1 += time.time()
        gcm.step += 1
        delta_t = gcm.get_timestep()

        wg = -time.time()
        spcpl.gather_gcm_data(gcm, self.les_models, self.cplsurf, self.output_column_indices, write=self.write)
        wg += time.time()

        wf = -time.time()
        pool = RequestsPool()
        for req in spcpl.set_les_forcings_batched(self.les_models, gcm, True, self.firststep, self.profiles,
                                                  dt_gcm=delta_t, factor=self.les_forcing_factor,
                                                  couple_surface=self.cplsurf, qt_forcing=self.qt_forcing,
                                                  write=self.write):
            pool.requests.extend(req.values())
        pool.waitall()
        wf += time.time()

        les_wall, self.profiles = self.step_les_models(t + delta_t, offset=self.les_spinup)

        wt = -time.time()
        spcpl.set_gcm_tendencies_batched(gcm, self.les_models, self.profiles, dt_gcm=delta_t,
                                         factor=self.gcm_forcing_factor, write=self.write,
                                         conservative=self.conservative_coarsening)
        wt += time.time()
        w2 = -time.time()
        gcm.evolve_model_from_cloud_scheme()
        w2 += time.time()
        self.timing_rows.append((starttime, w1, wg, wf, wt, w2) + tuple(les_wall))
        self.firststep = False

    # splib/splib.py:355-402 (forcings with dt = spinup length, no GCM tendencies)
    def step_spinup(self, spinup_length, les_spinup_forcing_factor=1.0):
        if not len(self.les_models):
            return
        probe = [self.les_models] if self.ensemble else self.les_models
        if any(getattr(les, "_spc_batch", None) is None for les in probe):   # splib.py:196 gathered at init
            spcpl.gather_gcm_data(self.gcm, self.les_models, self.cplsurf, write=self.write)
        t_les = self.les_models.model_time if self.ensemble else self.les_models[0].get_model_time()
        pool = RequestsPool()
        for req in spcpl.set_les_forcings_batched(self.les_models, self.gcm, True, self.firststep, self.profiles,
                                                  dt_gcm=spinup_length, factor=les_spinup_forcing_factor,
                                                  couple_surface=self.cplsurf, qt_forcing=self.qt_forcing,
                                                  write=self.write):
            pool.requests.extend(req.values())
        pool.waitall()
        _, self.profiles = self.step_les_models(t_les + spinup_length, offset=0)
        if self.write and spcpl.writer is not None:                           # splib/splib.py:388-391
            spcpl.write_les_profiles_batched(self.les_models)
        self.firststep = False

    def run(self, nsteps):
        for _ in range(nsteps):
            self.step()
