"""sp_coupler_amd -- MI355X-native batched superparameterization coupling step.

Drop-in for the per-step coupling math of CloudResolvingClimateModeling/sp-coupler
(``splib/spcpl.py`` + ``splib/sputils.py`` helpers): all SP columns packed as [n_cols x n_lev]
tensors in HBM and processed by hand-written gfx950 HIP kernels behind a C ABI (include/spc.h).

    from sp_coupler_amd import spcpl          # reference-named API (gather_gcm_data, set_les_forcings, ...)
    from sp_coupler_amd.engine import Engine  # batched tensor API

The HIP extension must be built first (``__graft_entry__.build()``); nothing here falls back to CPU.
"""
__version__ = "0.1.0"
