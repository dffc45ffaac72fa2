#!/usr/bin/env python3
"""What does one vector instruction occupy its SIMD for on this GPU?  (tools/csrc/spc_tools.hip: k_issue -- one wave per SIMD,
8 independent copies of the instruction per round.)  Used to price the fp32 arithmetic variant's quotients, which go through
fp64 (csrc/spc_hip.hip: the fp32 path), against v_fma_f32 = 4 cycles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import spc_tools
OPS = ["v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rndne_f64", "v_frexp_mant_f64", "v_ldexp_f64", "v_div_fixup_f64",
       "v_fma_f32", "v_rcp_f32", "v_div_fixup_f32", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_div_scale_f64", "v_div_scale_f32",
       "v_exp_f32", "v_log_f32", "v_cvt_f64_i32"]
tl = spc_tools.load()
s = torch.cuda.current_stream(); sp = ctypes.c_void_p(s.cuda_stream)
inp = (torch.rand(64, dtype=torch.float64, device="cuda") + 1.0)
out = torch.empty(256, dtype=torch.float64, device="cuda")
n = 200000
base = None
for op, name in enumerate(OPS):
    for _ in range(2):
        assert tl.spc_probe_issue(op, out.data_ptr(), inp.data_ptr(), n, sp) == 0, tl.spc_tools_last_error()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(5):
        tl.spc_probe_issue(op, out.data_ptr(), inp.data_ptr(), n, sp)
    b.record(s); torch.cuda.synchronize()
    ns = a.elapsed_time(b) * 1e6 / 5 / (8 * n)
    if name == "v_fma_f32":
        base = ns
    print("%-18s %7.3f ns per instruction and wave" % (name, ns), flush=True)
print("(v_fma_f32 = 4 cycles: 1 cycle = %.3f ns)" % (base / 4))
