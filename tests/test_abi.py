"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/spc.h declares (no compute call is made without a GPU), and the ctypes mirror matches it."""
import ctypes
import os
import re
import subprocess

import pytest

import __graft_entry__ as ge
from sp_coupler_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ge.build_hip()
    return _abi.load_library()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "spc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound(lib):
    declared = _declared_symbols()
    assert len(declared) >= 13
    assert sorted(_abi.PROTOTYPES) == declared          # python mirror covers exactly the header
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_error_text(lib):
    assert lib.spc_abi_version() == _abi.ABI_VERSION == 4
    assert isinstance(lib.spc_last_error(), bytes)


def test_struct_layout_matches_c_compiler(tmp_path):
    """sizeof/offsetof as gcc sees include/spc.h == the ctypes mirror."""
    probes = [("spc_dims", _abi.Dims, ["n_cols", "nG", "nL", "pitchG", "pitchGh", "pitchL", "les_grid_shared", "cols_per_block"]),
              ("spc_forward_args", _abi.ForwardArgs, ["U", "zf", "rain_last", "factor", "dt", "f_u", "idx", "Z0M", "wqt"]),
              ("spc_backward_args", _abi.BackwardArgs, ["T", "A_prof", "rhobf_d", "conservative", "factor", "dt", "f_T", "start_index"]),
              ("spc_diagnostics_args", _abi.DiagnosticsArgs, ["T", "zf", "Tv", "ql_water"]),
              ("spc_vnudge_args", _abi.VnudgeArgs, ["n_cols", "itot", "ktot", "constantT", "qt", "R", "presf", "beta", "status", "work", "work_bytes"]),
              ("spc_interp_args", _abi.InterpArgs, ["n_rows", "n_x", "n_xp", "pitch_x", "pitch_out", "x", "fp", "out"]),
              ("spc_searchsorted_args", _abi.SearchsortedArgs, ["n_rows", "n_a", "n_v", "pitch_a", "pitch_out", "a", "out", "side_right"]),
              ("spc_interp_c_args", _abi.InterpCArgs, ["n_rows", "nG", "nL", "pitch_Zh", "pitch_zh", "pitch_q", "pitch_out", "Zh", "rho", "out", "mode"])]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "spc.h"', 'int main(void){']
    for cname, _, fields in probes:
        lines.append('printf("%%zu\\n", sizeof(%s));' % cname)
        for f in fields:
            lines.append('printf("%%zu\\n", offsetof(%s, %s));' % (cname, f))
    lines.append('return 0;}')
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = []
    for _, cls, fields in probes:
        want.append(ctypes.sizeof(cls))
        want += [getattr(cls, f).offset for f in fields]
    assert got == want


def test_invalid_arguments_are_rejected_without_a_gpu(lib):
    """Argument validation happens on the host before any launch: error codes and texts, no crash."""
    d = _abi.Dims(4, 91, 160, 91, 92, 160, 1, 0)
    a = _abi.ForwardArgs()                                   # all pointers NULL
    assert lib.spc_forward_f64(ctypes.byref(d), ctypes.byref(a), None) == _abi.SPC_ERR_INVALID_ARGUMENT
    assert b"NULL" in lib.spc_last_error()
    bad = _abi.Dims(4, 91, 160, 90, 92, 160, 1, 0)           # pitch < levels
    assert lib.spc_forward_f64(ctypes.byref(bad), ctypes.byref(a), None) == _abi.SPC_ERR_INVALID_ARGUMENT
    assert b"pitch" in lib.spc_last_error()
    neg = _abi.Dims(-1, 91, 160, 91, 92, 160, 1, 0)
    assert lib.spc_backward_f64(ctypes.byref(neg), ctypes.byref(_abi.BackwardArgs()), None) == _abi.SPC_ERR_INVALID_ARGUMENT
    with pytest.raises(_abi.SpcInvalidArgument):
        _abi.check(lib, lib.spc_cloud_indices_f64(ctypes.byref(d), None, None, None, None))
    empty = _abi.Dims(0, 91, 160, 91, 92, 160, 1, 0)          # empty batch is a no-op, not an error
    assert lib.spc_forward_f64(ctypes.byref(empty), ctypes.byref(a), None) == 0
    assert lib.spc_backward_f64(ctypes.byref(empty), ctypes.byref(_abi.BackwardArgs()), None) == 0


def test_cols_per_block_heuristic(lib):
    small = _abi.Dims(1024, 91, 160, 91, 92, 160, 1, 0)
    big = _abi.Dims(348528, 91, 160, 91, 92, 160, 1, 0)
    assert lib.spc_pick_cols_per_block(ctypes.byref(small), 0) == 4      # 4 columns per 1024-thread workgroup
    assert lib.spc_pick_cols_per_block(ctypes.byref(big), 0) == 8
    tall = _abi.Dims(348528, 137, 512, 137, 138, 512, 1, 0)   # backward LDS: 6*512+137 doubles per column
    assert 1 <= lib.spc_pick_cols_per_block(ctypes.byref(tall), 1) <= 2


def test_describe_launch_names_the_instantiation_the_launcher_would_pick(lib):
    """spc_describe_launch is pure host logic (the occupancy query falls back without a device): the documented
    size classes of the K1 / K3 dispatch read back as instantiation names."""
    def d(n, nG=91, nL=160, pad=0, cb=0):
        return _abi.Dims(n, nG, nL, nG + pad, nG + 1 + pad, nL + pad, 1, cb)
    k1 = lambda n, **kw: _abi.describe_launch(lib, d(n, **kw), 0, 1).split()[0]       # noqa: E731
    k3 = lambda n, **kw: _abi.describe_launch(lib, d(n, **kw), 1, 0).split()[0]       # noqa: E731
    assert k1(200) == "k_forward<f64,lean,91,160,wt=1,blk=256,pre=1>"
    assert k1(300) == "k_forward<f64,lean,91,160,wt=1,blk=512,pre=1>"
    assert k1(1024) == "k_forward<f64,lean,91,160,wt=1,blk=1024,pre=1>"
    assert k1(1025) == "k_forward<f64,lean,91,160,wt=1,blk=256,pre=0>"
    assert k1(35718) == "k_forward<f64,lean,91,160,wt=0,blk=256,pre=0>"
    assert k1(35718, pad=3) == "k_forward<f64,lean,0,0,wt=0,blk=256,pre=0>"          # padded pitch: run-time geometry
    assert k1(1024, nG=137, nL=512) == "k_forward<f64,lean,137,512,wt=1,blk=256,pre=1>"
    assert k3(1024) == "k_backward<f64,91,160,wt=1,blk=1024,pre=1>"
    assert k3(1025) == "k_backward<f64,91,160,wt=1,blk=256,pre=0>"
    assert k3(2880) == "k_backward<f64,91,160,wt=1,blk=256,pre=0>"            # K3 stores write-through up to 14 MiB written
    assert k3(2881) == "k_backward<f64,91,160,wt=0,blk=256,pre=0>"
    assert k3(25000) == "k_backward<f64,91,160,wt=0,blk=256,pre=0>"
    assert k3(25001) == "k_backward<f64,91,160,wt=0,blk=256,pre=1>"
    assert k3(4096, nG=19) == "k_backward<f64,19,160,wt=1,blk=256,pre=0>"
    full = _abi.describe_launch(lib, d(4096), 0, 3)
    assert full.startswith("k_forward<f64,full,91,160,wt=0,blk=256,pre=0> cb=")        # FULL: plain stores only
    assert _abi.describe_launch(lib, d(4096), 4, 0).startswith("k_backward_cons3<f64,91,160,cb=")
    # K4's footprint is what lets FIVE two-column workgroups share a CU: 25 of gfx950's 1 280-byte LDS allocation granules
    k4 = _abi.describe_launch(lib, d(35718), 4, 0)
    assert " cb=2 " in k4 and int(k4.split("lds=")[1].split()[0]) <= 25 * 1280, k4
    assert _abi.describe_launch(lib, d(4096, nL=400), 4, 0).startswith("k_backward_cons2<f64,0,0,pd=2> cb=")
    assert _abi.describe_launch(lib, d(4096, nL=2000), 4, 0).startswith("k_backward_cons2<f64,0,0,pd=-1> cb=")
    # float: multi-round launches of the compile-time geometries with an even slab take the 8-byte-access kernels (spc_f32v.hpp)
    assert _abi.describe_launch(lib, d(4096), 0, 1, 4).startswith("k_forward_f32v<91,160,wt=1>")
    assert _abi.describe_launch(lib, d(35718), 1, 0, 4).startswith("k_backward<f32,91,160,wt=0,blk=256,pre=1> cb=2 ")     # K3<float>: two columns
    assert _abi.describe_launch(lib, d(1000), 0, 1, 4).startswith("k_forward<f32,lean,91,160,wt=1,blk=1024,pre=1>")      # one round: scalar
    assert _abi.describe_launch(lib, d(4096, pad=1), 0, 1, 4).startswith("k_forward<f32,lean,0,0,")                    # run-time geometry: scalar
    txt = _abi.describe_launch(lib, d(35718), 0, 1)
    fields = dict(kv.split("=") for kv in txt.split()[1:])
    assert int(fields["cb"]) * int(fields["grid"]) >= 35718 > int(fields["cb"]) * (int(fields["grid"]) - 1)
    with pytest.raises(_abi.SpcInvalidArgument):
        _abi.describe_launch(lib, d(10), 7, 0)
    assert lib.spc_vnudge_workspace_bytes(2, 64, 64, 160) == 2 * 2 * 64 * 64 * 160 * 8
    assert lib.spc_vnudge_workspace_bytes(2, 128, 128, 160) == 2 * 2 * 128 * 128 * 160 * 8   # large planes stream from it
    assert lib.spc_vnudge_workspace_bytes(2, 0, 64, 160) < 0


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_abi.SpcLibraryError):
        _abi.load_library(str(tmp_path / "libspc_hip.so"))


def test_engine_refuses_to_run_without_gpu(lib):
    import torch
    from sp_coupler_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine()


def test_sputils_operators_validate_on_the_host(lib):
    """K7 entry points (spc_exner / interp / searchsorted / interp_c / rms): argument checks before any launch"""
    E = _abi.SPC_ERR_INVALID_ARGUMENT
    assert lib.spc_exner_f64(0, None, None, 0, None) == 0                       # empty: no-op
    assert lib.spc_exner_f64(8, None, None, 0, None) == E and b"NULL" in lib.spc_last_error()
    assert lib.spc_exner_f32(-1, None, None, 1, None) == E
    a = _abi.InterpArgs(4, 160, 0, 160, 91, 91, 160, 1, 1, 1, 1)                # empty sample array: numpy raises ValueError
    assert lib.spc_interp_f64(ctypes.byref(a), None) == E and b"empty" in lib.spc_last_error()
    a = _abi.InterpArgs(4, 160, 91, 100, 91, 91, 160, 1, 1, 1, 1)               # pitch_x < n_x
    assert lib.spc_interp_f64(ctypes.byref(a), None) == E and b"pitch" in lib.spc_last_error()
    a = _abi.InterpArgs(4, 160, 91, 0, 0, 91, 160, None, None, None, None)
    assert lib.spc_interp_f32(ctypes.byref(a), None) == E and b"NULL" in lib.spc_last_error()
    assert lib.spc_interp_f64(ctypes.byref(_abi.InterpArgs(0, 160, 91, 0, 0, 91, 160)), None) == 0
    s = _abi.SearchsortedArgs(4, 160, 92, 0, 92, 10, 1, 1, 1, 1, 0)             # pitch_out < n_v
    assert lib.spc_searchsorted_f64(ctypes.byref(s), None) == E
    c = _abi.InterpCArgs(4, 91, 160, 92, 0, 160, 91, 1, 1, 1, None, 1, 0, 0)    # interp_c without weights
    assert lib.spc_interp_c_f64(ctypes.byref(c), None) == E and b"rho" in lib.spc_last_error()
    c = _abi.InterpCArgs(4, 91, 1, 92, 0, 160, 91, 1, 1, 1, 1, 1, 0, 0)         # one grid point bounds no cell
    assert lib.spc_interp_c_f64(ctypes.byref(c), None) == E
    c = _abi.InterpCArgs(4, 91, 160, 92, 0, 160, 91, 1, 1, 1, 1, 1, 7, 0)
    assert lib.spc_interp_c_f32(ctypes.byref(c), None) == E and b"mode" in lib.spc_last_error()
    assert lib.spc_rms_f64(3, 160, 100, None, None, None) == E
    assert lib.spc_rms_f64(0, 160, 160, None, None, None) == 0


def test_sputils_module_fails_loudly_without_a_gpu():
    """the GPU twin of splib/sputils.py has no CPU path: without a HIP device the first call raises"""
    import numpy
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sp_coupler_amd import spcpl, sputils
    spcpl.set_engine(None)
    assert sputils.rd == 287.04 and sputils.cp == 1004. and sputils.pref0 == 1e5          # sputils.py:14-20
    for call in (lambda: sputils.exner(1e5), lambda: sputils.interp(numpy.zeros(2), numpy.arange(3.), numpy.arange(3.)),
                 lambda: sputils.rms(numpy.ones(3))):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


def test_launch_heuristics_follow_the_cu_count_of_the_device(lib, monkeypatch):
    """round-4 verdict, weak 10: the residency rules (pick_cb rule 1 / rule 2, pick_cb_cons3, small_block, the PRE thresholds,
    the K7 slab height) count rounds of workgroups against the CUs of the CURRENT device (hipDeviceAttributeMultiprocessorCount,
    256 on an MI355X in SPX mode, fewer on a CPX / DPX partition) instead of a literal 256.  SPC_CUS overrides the query, so the
    heuristic can be walked without a GPU: every size class scales with the CU count, and spc_describe_launch says which count
    it used."""
    def d(n, nG=91, nL=160):
        return _abi.Dims(n, nG, nL, nG, nG + 1, nL, 1, 0)
    desc = lambda n, pass_, flags=0: _abi.describe_launch(lib, d(n), pass_, flags)        # noqa: E731
    field = lambda txt, k: dict(kv.split("=") for kv in txt.split()[1:])[k]               # noqa: E731
    seen = {}
    for cus in (32, 64, 128, 256):
        monkeypatch.setenv("SPC_CUS", str(cus))
        assert field(desc(1000, 0, 1), "cus") == str(cus)
        # small_block: cus < n <= 2 cus -> 512 threads, <= 4 cus -> 1024 threads, beyond -> 256-thread workgroups without the
        # prologue prefetch (the measured 257 / 513 / 1025 boundaries of 256 CUs)
        assert "blk=256,pre=1" in desc(cus, 0, 1)
        assert "blk=512,pre=1" in desc(cus + 1, 0, 1) and "blk=512,pre=1" in desc(2 * cus, 0, 1)
        assert "blk=1024,pre=1" in desc(2 * cus + 1, 0, 1) and "blk=1024,pre=1" in desc(4 * cus, 0, 1)
        assert "blk=256,pre=0" in desc(4 * cus + 1, 0, 1)
        # K3's prologue prefetch comes back where the kernel has saturated: 25 000 columns on 256 CUs, pro rata elsewhere
        assert ",pre=0>" in desc(25000 * cus // 256, 1) and ",pre=1>" in desc(25000 * cus // 256 + 1, 1)
        # rule 1 of pick_cb: the smallest slab whose grid is resident at once -- the column count up to which K3 keeps two
        # columns per workgroup is proportional to the CU count (without a device the occupancy query falls back to 4
        # workgroups per CU: 8 columns per CU)
        n1 = min(n for n in range(4 * cus + 1, 40 * cus) if field(desc(n, 1), "cb") != "2") - 1
        assert n1 == 8 * cus
        seen[cus] = n1
    assert seen[64] == 2 * seen[32] and seen[128] == 2 * seen[64] and seen[256] == 2 * seen[128], seen
    monkeypatch.delenv("SPC_CUS")
    assert field(desc(1000, 0, 1), "cus") == "256"        # no device here: the MI355X's count
