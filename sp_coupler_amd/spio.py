"""spifs output: the variable list of the reference's ``spifs.nc``
(``splib/spio.py:133-164,176-210``, README table) written ONCE PER VARIABLE PER STEP for all columns,
instead of ~40 slice writes per column per step (``spcpl.py:230-244,352-376,412-425,545-555``).

The reference writes netCDF-4 with one group per column through the ``netCDF4`` package, which is not
available here; this writer produces netCDF-3 classic through ``scipy.io.netcdf_file`` with the same
variable names, units and f4 storage (``spio.py:153``) but a batched layout: dimensions
``Time`` (record), ``column``, ``zf`` (LES levels), ``oifs_height`` (GCM levels); a column's group in the
reference corresponds to index ``column`` here (``grid_index[column]`` holds the GCM grid index).
``read_column`` returns what ``examples/access-spifs-nc.py`` reads from a group.
"""
import threading

import numpy
from scipy.io import netcdf_file

# (name, unit) on LES levels -- spio.py:133-152
LES_LEVEL_VARS = (('u', 'm/s'), ('v', 'm/s'), ('thl', 'K'), ('qt', '1'), ('ql', '1'), ('ql_ice', '1'), ('ql_water', '1'),
                  ('qr', '1'), ('t', 'K'), ('t_', 'K'), ('f_u', 'm/s'), ('f_v', 'm/s'), ('f_thl', 'K/s'), ('f_qt', '1/s'),
                  ('presf', 'Pa/s'), ('rhof', 'kg/m^3'), ('rhobf', 'kg/m^3'), ('qt_std', '1'), ('qt_alpha', '1/s'),
                  ('qt_beta', '1'))
# on GCM levels -- spio.py:158-164 (tendencies) and 176-190 (state)
GCM_LEVEL_VARS = (('f_U', 'm/s'), ('f_V', 'm/s'), ('f_T', 'K/s'), ('f_SH', '1/s'), ('f_QL', '1/s'), ('f_QI', '1/s'),
                  ('f_A', '1/s'), ('U', 'm/s'), ('V', 'm/s'), ('T', 'K'), ('SH', '1'), ('QL', '1'), ('QI', '1'), ('Pf', 'Pa'),
                  ('Ph', 'Pa'), ('Tv', 'K'), ('Zf', 'm'), ('Zh', 'm'), ('THL', 'K'), ('QT', '1'), ('A', '1'), ('A_d', '1'))
# per-column scalars -- spio.py:195-210
SURFACE_VARS = (('Psurf', 'Pa'), ('rain', 'kg / m^2'), ('rainrate', 'kg / m^2h'), ('z0m', 'm'), ('z0h', 'm'),
                ('wthl', 'K m/s'), ('wqt', 'kg/kg m/s'), ('TLflux', 'W/m^2'), ('TSflux', 'W/m^2'), ('SHflux', 'kg / m^2s'),
                ('QLflux', 'kg / m^2s'), ('QIflux', 'kg / m^2s'))


class SpifsWriter:
    def __init__(self, path, grid_indices, lats, lons, zf, nG, start_time="", with_surf_vars=True):
        self.lock = threading.Lock()                                          # spio.py:24
        self.step = -1                                                        # spio.py:27
        self.path = path
        self.n = len(grid_indices)
        f = self.f = netcdf_file(path, "w")
        f.createDimension("Time", None)                                       # spio.py:100
        f.createDimension("column", self.n)
        f.createDimension("zf", len(zf))                                      # spio.py:96
        f.createDimension("oifs_height", nG)                                  # spio.py:98
        t = f.createVariable("Time", "f4", ("Time",))
        t.units = "s since " + str(start_time)                                # spio.py:122
        v = f.createVariable("zf", "f4", ("zf",))
        v[:] = numpy.asarray(zf, dtype=numpy.float32)
        v.units = "m"
        for name, data, typ in (("grid_index", grid_indices, "i4"), ("lat", lats, "f4"), ("lon", lons, "f4")):
            v = f.createVariable(name, typ, ("column",))
            v[:] = numpy.asarray(data)
        for name, unit in LES_LEVEL_VARS:
            f.createVariable(name, "f4", ("Time", "column", "zf")).units = unit
        for name, unit in GCM_LEVEL_VARS:
            f.createVariable(name, "f4", ("Time", "column", "oifs_height")).units = unit
        for name, unit in SURFACE_VARS[: (12 if with_surf_vars else 3)]:
            f.createVariable(name, "f4", ("Time", "column")).units = unit

    def update_time(self, t):
        """spio.update_time (spio.py:68-72): append a record"""
        self.step = self.f.variables["Time"].shape[0]
        self.f.variables["Time"][self.step] = float(t)
        for var in self.f.variables.values():   # netCDF-3 records are written whole: grow every record variable
            if var.isrec and var.shape[0] <= self.step:
                var[self.step] = numpy.full(var.shape[1:], numpy.nan, dtype=numpy.float32)

    def write(self, rows=None, **arrays):
        """one slice write per variable for the whole batch: arrays[name] is [n x levels] or [n];
        ``rows`` selects a subset of columns (extra output columns); an array with fewer rows than the file
        has columns addresses rows 0..m-1 (the SP columns of a file that also holds extra output columns)."""
        with self.lock:
            for name, arr in arrays.items():
                var = self.f.variables.get(name)
                if var is None:
                    raise KeyError("Attempt to write profile to uninitialized variable %s" % name)   # spio.py:240
                a = numpy.asarray(arr, dtype=numpy.float32)
                sel = rows                               # per VARIABLE: arrays of one call may differ in their row count
                if sel is None and a.ndim >= 1 and a.shape[0] < self.n:
                    sel = numpy.arange(a.shape[0])       # the SP columns occupy the FIRST rows; extra output
                                                         # columns (spcpl.py:89-129) follow and keep their values
                if sel is None:
                    var[self.step] = a
                else:
                    cur = numpy.array(var[self.step]) if var.shape[0] > self.step else numpy.zeros(var.shape[1:], "f4")
                    cur[numpy.asarray(sel)] = a
                    var[self.step] = cur

    def sync(self):
        with self.lock:                                                       # spio.sync_root, spio.py:76-84
            self.f.flush()

    def close(self):
        with self.lock:
            self.f.close()


def _native(a):
    a = numpy.array(a)
    return a.astype(a.dtype.newbyteorder("="))       # netCDF-3 stores big-endian


def read_column(path, column):
    """dict name -> array[Time, ...] of one column (what the reference example reads from a group)"""
    with netcdf_file(path, "r", mmap=False) as f:
        out = {"Time": _native(f.variables["Time"][:]), "zf": _native(f.variables["zf"][:])}
        for name, var in f.variables.items():
            if len(var.dimensions) >= 2 and var.dimensions[:2] == ("Time", "column"):
                out[name] = _native(var[:, column])
            elif var.dimensions == ("column",):
                out[name] = _native(var[column])
    return out


def read_record_nearest(path, t, names):
    """the record whose Time is nearest to ``t`` (what spcpl.set_gcm_tendencies_from_file looks up, splib/spcpl.py:560):
    returns (record index, its Time, grid_index[column], {name: [column x levels] float64})"""
    with netcdf_file(path, "r", mmap=False) as f:
        times = _native(f.variables["Time"][:]).astype(numpy.float64)
        if times.size == 0:
            raise ValueError("%s holds no record" % path)
        ti = int(numpy.abs(times - float(t)).argmin())
        data = {n: _native(f.variables[n][ti]).astype(numpy.float64) for n in names}
        return ti, float(times[ti]), _native(f.variables["grid_index"][:]), data
