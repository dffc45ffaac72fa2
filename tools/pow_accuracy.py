#!/usr/bin/env python3
"""Build and run the host accuracy sweep of spc_pow (tools/csrc/pow_accuracy.c includes the DEVICE's own source,
sp_coupler_amd/csrc/spc_pow.h, and compares with powl in 80-bit arithmetic).  usage: tools/pow_accuracy.py [points]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = "/tmp/spc_pow_accuracy"
subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(ROOT, "tools", "csrc", "pow_accuracy.c"), "-lm"], check=True)
subprocess.run([exe] + sys.argv[1:2], check=True)
