#!/usr/bin/env python3
"""Read a spifs file written by sp_coupler_amd.spio.SpifsWriter -- the counterpart of the reference's
examples/access-spifs-nc.py for the batched layout (one `column` dimension instead of one netCDF-4
group per column, same variable names / units / f4 storage).

usage: python examples/access_spifs.py [spifs.nc] [column] [time_index]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sp_coupler_amd import spio  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "spifs.nc"
    column = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    time_index = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    col = spio.read_column(path, column)
    # superparameterized columns carry the LES-level variables too (the reference tells them apart by the number of
    # variables in the group); here every column has every variable and the LES ones stay NaN for the others
    print("column %d: grid_index %d lat %.3f lon %.3f, %d records, time %g" % (
        column, int(col["grid_index"]), float(col["lat"]), float(col["lon"]), len(col["Time"]), float(col["Time"][time_index])))
    print()
    Zf, T, SH, U, V = (col[k][time_index] for k in ("Zf", "T", "SH", "U", "V"))
    for i in range(len(Zf)):                                      # height, temperature, specific humidity, winds
        print("%8.1f %5.1f %6.4f %4.1f %4.1f" % (Zf[i], T[i], SH[i], U[i], V[i]))


if __name__ == "__main__":
    main()
