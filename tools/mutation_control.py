#!/usr/bin/env python3
"""Mutation control of the semantic tests (round-4 verdict, next 2: "each property fails when the corresponding line of the
kernel is perturbed").  For the shipped library and for every mutant build/mutants/libspc_mutantN.so (tools/build_mutants.sh:
ONE kernel line perturbed each, -DSPC_MUTANT=N) the properties of tests/semantic_props.py run through the HIP kernels, in ONE
process (Engine(lib_path=...)).  Expected: the shipped library passes all, every mutant fails at least the property that
guards its line.  Prints a table; exit status 1 if a mutant survives or the shipped library fails.
usage: python tools/mutation_control.py > profiles/r05_mutation_control.log"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MUTANTS = {
    1: ("K1 thl: exponent +rd/cp instead of -rd/cp (exner for iexner, sputils.py:28-34)", "isentropic_column_has_constant_thl"),
    2: ("K1 forcings: u_d and v_d swapped (spcpl.py:328-329)", "zero_forcings_when_the_les_equals_the_interpolated_gcm_profile"),
    3: ("K3 f_QL from ql instead of ql_water = ql - ql_ice (spcpl.py:402, 520)", "total_water_tendency_closes"),
    4: ("K3 masking one level too far: k <= start_index (spcpl.py:527-533)", "masking_above_the_les_top"),
    5: ("K2 index map with side='left' instead of 'right' (spcpl.py:764)", "index_map_is_a_count"),
    6: ("K1 staging: U not reversed (spcpl.py:227)", "reversal_is_index_arithmetic_only"),
    7: ("K7 interp_c: numerator without the weight rho (sputils.py:152-154)", "conservative_coarsening_conserves"),
    8: ("K1 thl: latent term added instead of subtracted (spcpl.py:214)", "isentropic_column_has_constant_thl"),
    9: ("K3 cloud fraction A_d read from the neighbouring column of the slab (spcpl.py:404)", "columns_are_independent"),
    10: ("K5 t: exponent -rd/cp instead of +rd/cp (spcpl.py:409)", "isentropic_column_has_constant_thl"),
    11: ("K4 f_T: numerator without the weight rho (spcpl.py:482, sputils.py:152)", "conservative_coarsening_conserves"),
    12: ("K1 qt_ = SH + QL, the ice forgotten (spcpl.py:215)", "reversal_is_index_arithmetic_only"),
    13: ("K3 f_SH from qt instead of qt - ql (spcpl.py:519: SH is vapour only)", "total_water_tendency_closes"),
    14: ("K1 f_ps with the opposite sign (spcpl.py:332)", "zero_forcings_when_the_les_equals_the_interpolated_gcm_profile"),
    15: ("K1 surface branch: density from T at the model top instead of the lowest level (spcpl.py:153)", "surface_fluxes_are_the_ifs_fluxes_over_the_surface_density"),
    16: ("k_surface: wqt without the ice flux QIflux (spcpl.py:159)", "surface_fluxes_are_the_ifs_fluxes_over_the_surface_density"),
    17: ("K1 surface branch: wthl with exner instead of iexner (spcpl.py:161)", "surface_fluxes_are_the_ifs_fluxes_over_the_surface_density"),
    18: ("K6 update: qt += (beta - 1) qt, the level mean forgotten (spcpl.py:724)", "variability_nudge_reaches_the_gcm_cloud_amount"),
    19: ("K6 constantT: dTHL with the opposite sign (spcpl.py:731)", "variability_nudge_reaches_the_gcm_cloud_amount"),
    20: ("K6: 'significant cloud' threshold 1e-6 instead of 1e-9 (spcpl.py:665)", "variability_nudge_reaches_the_gcm_cloud_amount"),
    21: ("K6 additive noise subtracted instead of added (spcpl.py:716-719)", "variability_nudge_reaches_the_gcm_cloud_amount"),
    22: ("K3 f_U from the LES v instead of u (spcpl.py:524)", "tendencies_relax_the_gcm_towards_the_les_profile"),
    23: ("K1 rainrate with the opposite sign (spcpl.py:325)", "tendencies_relax_the_gcm_towards_the_les_profile"),
    24: ("K3 f_T with the opposite sign (spcpl.py:518)", "tendencies_relax_the_gcm_towards_the_les_profile"),
    25: ("K5 Tv: the condensate load added instead of subtracted (spcpl.py:176)", "gcm_level_diagnostics_mean_what_their_names_say"),
    26: ("K5 QT without the ice (spcpl.py:215)", "gcm_level_diagnostics_mean_what_their_names_say"),
    27: ("K5 Zh above the lowest FULL-level interface instead of the surface (spcpl.py:197)", "gcm_level_diagnostics_mean_what_their_names_say"),
}


def run(lib_path):
    from tests import semantic_props as sp
    from tests.test_semantic_gpu import HipImpl
    impl = HipImpl(lib_path)
    failed = []
    for prop in sp.PROPERTIES:
        try:
            prop(impl)
        except AssertionError:
            failed.append(prop.__name__[5:])
        except Exception:
            failed.append(prop.__name__[5:] + " (raised: %s)" % traceback.format_exc().strip().splitlines()[-1])
    return failed


def main():
    import torch
    print("mutation control of tests/test_semantic_gpu.py on %s" % torch.cuda.get_device_name(0))
    bad = 0
    clean = run(None)
    from tests import semantic_props as sp
    print("shipped library: %d properties, failed: %s" % (len(sp.PROPERTIES), clean or "none"))
    bad += bool(clean)
    for n, (what, guard) in sorted(MUTANTS.items()):
        path = os.path.join(ROOT, "build", "mutants", "libspc_mutant%d.so" % n)
        if not os.path.exists(path):
            print("mutant %2d: NOT BUILT (%s)" % (n, path))
            bad += 1
            continue
        failed = run(path)
        ok = guard in [f.split(" ")[0] for f in failed]
        bad += not ok
        print("mutant %2d: %s\n           guarded by %s: %s; all failing: %s" % (n, what, guard, "DETECTED" if ok else "SURVIVED", failed or "none"))
    print("result: %s" % ("every mutant detected, shipped library clean" if not bad else "%d problem(s)" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
