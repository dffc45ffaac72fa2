#!/usr/bin/env python3
"""Negative control for tests/test_transport_gpu.py's two ordering tests: with the fence under test switched OFF the test
must FAIL (the copy tears), otherwise it would prove nothing.  Switched off here, in the tool, never in the product:
  * Arena.settle -> no-op                      (a kernel may overwrite results whose download nobody has collected)
  * wait_stream inside Arena.push -> no-op     (an upload may overwrite inputs of a kernel that has not run yet)
Prints one line per (control, engine kind): "tears as expected" or "NOT DETECTED"."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sp_coupler_amd import transfer  # noqa: E402
from tests import test_transport_gpu as t  # noqa: E402


def expect_failure(name, fn, *a):
    """Whether an unfenced copy overtakes the kernel depends on which hardware queues the two streams land on (HIP maps
    its streams onto a few in-order hardware queues; two streams on ONE queue are serialised whatever the program says), so
    the control is repeated with the stream pool advanced by one each time until the race shows."""
    keep = []
    for attempt in range(8):
        try:
            fn(*a)
        except AssertionError as e:
            print("%-60s tears as expected (attempt %d): %s" % (name, attempt + 1, str(e).splitlines()[0][:80]), flush=True)
            return True
        keep.append(torch.cuda.Stream("cuda:0"))     # shifts every later stream to the next slot of torch's pool
    print("%-60s NOT DETECTED (the test passed 8 times with the fence off)" % name, flush=True)
    return False


ok = True
for kind in ("one", "own_stream"):
    t.test_a_kernel_does_not_overwrite_results_nobody_has_collected(kind)          # sanity: passes with the fences on
    t.test_second_gather_does_not_overwrite_inputs_of_a_kernel_that_has_not_run(kind)
    orig_settle = transfer.Arena.settle
    transfer.Arena.settle = lambda self: None
    try:
        ok &= expect_failure("settle() off, engine %s" % kind, t.test_a_kernel_does_not_overwrite_results_nobody_has_collected, kind)
    finally:
        transfer.Arena.settle = orig_settle
    orig_push, orig_wait, off = transfer.Arena.push, torch.cuda.Stream.wait_stream, {"v": False}

    def wait(self, other, _o=orig_wait):
        if not off["v"]:
            _o(self, other)

    def push(self, *a, _p=orig_push, **k):
        off["v"] = True
        try:
            return _p(self, *a, **k)
        finally:
            off["v"] = False
    torch.cuda.Stream.wait_stream, transfer.Arena.push = wait, push
    try:
        ok &= expect_failure("push()'s wait for the compute stream off, engine %s" % kind,
                             t.test_second_gather_does_not_overwrite_inputs_of_a_kernel_that_has_not_run, kind)
    finally:
        torch.cuda.Stream.wait_stream, transfer.Arena.push = orig_wait, orig_push
print("hazard control: %s" % ("every switched-off fence was detected" if ok else "A FENCE COULD BE REMOVED UNNOTICED"))
