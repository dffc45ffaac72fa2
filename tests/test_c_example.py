"""examples/c_abi_demo.c: the drop-in boundary used from plain C (gcc, no Python / torch / HIP headers).  CPU: it builds
against include/spc.h + libspc_hip.so and fails loudly without a device.  GPU: the same inputs through the Python engine
give the same bits."""
import os
import subprocess

import numpy
import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, NG, NL = 8, 20, 32


def build_demo(tmp_path):
    ge.build_hip()
    exe = str(tmp_path / "c_abi_demo")
    lib_dir = os.path.join(ROOT, "sp_coupler_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", exe, "-L" + lib_dir, "-lspc_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_c_demo_builds_and_fails_loudly_without_a_device(tmp_path):
    import torch
    exe = build_demo(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: see the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_demo_matches_the_python_engine_bit_for_bit(tmp_path):
    import torch
    from sp_coupler_amd.engine import Engine
    exe = build_demo(tmp_path)
    dump = str(tmp_path / "dump.bin")
    r = subprocess.run([exe, dump], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of %d elements differ" % (N * NG) in r.stdout
    raw = numpy.fromfile(dump, dtype=numpy.float64)
    pos = 0

    def take(*shape):
        nonlocal pos
        n = int(numpy.prod(shape))
        a = raw[pos:pos + n].reshape(shape)
        pos += n
        return a
    G, Gh, L = (N, NG), (N, NG + 1), (N, NL)
    names = [("U", G), ("V", G), ("T", G), ("SH", G), ("QL", G), ("QI", G), ("A", G), ("Pfull", G), ("Phalf", Gh), ("Zgfull", G),
             ("Zghalf", Gh), ("zf", (NL,)), ("zh", (NL,)), ("u_d", L), ("v_d", L), ("thl_d", L), ("qt_d", L), ("ql_d", L), ("qi_d", L),
             ("t_d", L), ("A_prof", G), ("ps_d", (N,)),
             ("f_u", L), ("f_thl", L), ("ql_ref", L), ("f_ps", (N,)), ("Zf", G), ("f_T", G), ("f_U", G), ("f_A", G), ("interp", G)]
    d = {k: take(*shape) for k, shape in names}
    assert pos == raw.size
    eng = Engine("cuda:0")
    dev = lambda a: torch.from_numpy(numpy.ascontiguousarray(a)).to(eng.device)      # noqa: E731
    gcm = {k: dev(d[k]) for k in ("U", "V", "T", "SH", "QL", "QI", "A", "Pfull", "Phalf", "Zgfull", "Zghalf")}
    prof = {"U": dev(d["u_d"]), "V": dev(d["v_d"]), "THL": dev(d["thl_d"]), "QT": dev(d["qt_d"]), "QL": dev(d["ql_d"]), "PS": dev(d["ps_d"]),
            "QL_ice": dev(d["qi_d"]), "T": dev(d["t_d"]), "A": dev(d["A_prof"])}
    fwd = eng.forward(gcm, dev(d["zf"]), prof, 1.0, 900.0, zh=dev(d["zh"]))
    bwd = eng.backward(gcm, dev(d["zf"]), prof, 1.0, 900.0, Zf=fwd["Zf"])
    torch.cuda.synchronize()
    for k in ("f_u", "f_thl", "ql_ref", "f_ps", "Zf"):
        assert numpy.array_equal(fwd[k].cpu().numpy(), d[k]), k
    for k in ("f_T", "f_U", "f_A"):
        assert numpy.array_equal(bwd[k].cpu().numpy(), d[k], equal_nan=True), k
    assert numpy.array_equal(eng.interp(fwd["Zf"], dev(d["zf"]), dev(d["u_d"])).cpu().numpy(), d["interp"])
