#!/usr/bin/env python3
"""bench.py -- SP column-exchanges/s of the batched coupling step on MI355X.

One *step* = one pass of the hot path over one batch of synthetic columns:
    K1 forward (convert + interpolate + LES forcings, fused K2 cloud-fraction index map)  ->  K3 backward
    (interpolate back + GCM tendencies), i.e. one *column-exchange* per column.
Workload at N=1: BASELINE.json configs[1] -- 1024 synthetic SP columns, 91 GCM <-> 160 LES levels, fp64.
Inputs are resident in HBM before the timed region; ROTATE distinct batches (default 8 x 44 MB of
live arrays > 256 MB Infinity Cache) are cycled so the kernels stream from HBM, not from cache.
N>1 (launched by torch.distributed.run): weak scaling, every rank owns its own 1024-column batches,
no data-path collective (columns are independent); only the barrier and the max-over-ranks of the
elapsed time use RCCL.

Prints ONE JSON line on rank 0 (contract: see the task statement); extra keys `roofline` and
`cpu_baseline` as specified there.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy)


def algorithmic_bytes(nG, nL, esize=8):
    """SURVEY.md section 8(d) / BASELINE.md section 3, per column-exchange."""
    fwd = (9 * nG + 6 * nL + 3) * esize + (6 * nL + 1) * esize
    idx = nL * esize + nG * 4
    bwd = (9 * nG + 7 * nL) * esize + 7 * nG * esize
    return {"forward": fwd, "index": idx, "backward": bwd, "k1_launch": fwd + idx, "k3_launch": bwd,
            "exchange": fwd + idx + bwd}


def cpu_baseline(gcm, zf, zh, prof, dt, factor, budget_s):
    """The reference's algorithm on the host: serial Python loop over columns, one numpy.interp per
    profile (oracle/spcpl_oracle.py).  Bounded sample: whole passes over the batch until budget_s."""
    from oracle import spcpl_oracle as orc
    n = gcm["T"].shape[0]
    done, t0 = 0, time.perf_counter()
    while True:
        f = orc.forward_batched(gcm, prof, zf, zh, factor, dt, couple_surface=False)
        orc.backward_batched(gcm, f["Zf"], prof, zf, factor, dt)
        done += n
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {"value": done / el, "unit": "column-exchanges/s", "cores": 1, "kind": "port",
            "sample": "%d column-exchanges (%d passes over the %d-column batch, NumPy per-column loop, "
                      "%.1f s)" % (done, done // n, n, el)}


def cpu_worker_main(path, budget):
    """`python bench.py --cpu-worker PATH BUDGET`: one worker of the all-cores leg -- whole passes over
    its copy of a small batch for BUDGET seconds; prints "<done> <seconds>". Never touches the GPU."""
    import numpy
    from oracle import spcpl_oracle as orc
    z = numpy.load(path)
    gcm = {k[2:]: z[k] for k in z.files if k.startswith("g_")}
    prof = {k[2:]: z[k] for k in z.files if k.startswith("p_")}
    zf, zh = z["zf"], z["zh"]
    n = gcm["T"].shape[0]
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        f = orc.forward_batched(gcm, prof, zf, zh, 1.0, 900.0)
        orc.backward_batched(gcm, f["Zf"], prof, zf, 1.0, 900.0)
        done += n
    print(done, time.perf_counter() - t0)


def cpu_baseline_multicore(gcm, zf, zh, prof, budget_s, workers):
    """the same per-column NumPy loop in `workers` independent CHILD PROCESSES
    (plain subprocesses with a hard timeout -- no pool that could respawn; they never touch the GPU)."""
    import subprocess
    import tempfile
    import numpy
    sub = 128                                   # a small batch per pass keeps every worker inside the budget
    d = tempfile.mkdtemp(prefix="spc_cpu_")
    path = os.path.join(d, "batch.npz")
    numpy.savez(path, zf=zf, zh=zh, **{"g_" + k: v[:sub] for k, v in gcm.items()},
                **{"p_" + k: v[:sub] for k, v in prof.items()})
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(budget_s)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
             for _ in range(workers)]
    res = []
    try:
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=budget_s + 60)
                a, b = out.split()
                res.append((int(a), float(b)))
            except Exception:
                pr.kill()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        try:
            os.remove(path)
            os.rmdir(d)
        except OSError:
            pass
    if not res:
        raise RuntimeError("no cpu worker finished")
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": total / wall, "unit": "column-exchanges/s", "cores": len(res), "kind": "port",
            "sample": "%d column-exchanges in %d processes x %.1f s (NumPy per-column loop)" % (total, len(res), wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.md config id (2 = 1024 cols, 91<->160)")
    ap.add_argument("--cols", type=int, default=None, help="override columns per GPU")
    ap.add_argument("--rotate", type=int, default=8, help="distinct batches cycled through (cache defeat)")
    ap.add_argument("--cols-per-block", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-cpu-multicore", action="store_true", help="skip the all-cores cpu_baseline extra")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N>1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="force this HIP device for every rank (rehearsal on a 1-GPU box)")
    args = ap.parse_args()

    import numpy
    import torch
    from sp_coupler_amd import synthetic
    from sp_coupler_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    if args.device is not None:
        local = args.device
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                probe = torch.zeros(1, device="cuda:%d" % local)
                dist.all_reduce(probe)                      # fail here, not in the timed region
                torch.cuda.synchronize()
            except Exception as e:                          # the data path needs no collective: gloo is enough
                print("bench.py: RCCL unavailable (%r), using gloo for barrier/max" % (e,), file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                args.backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(args.backend)

    n_cols, nG, nL, seed = synthetic.CONFIGS[args.config]
    if args.cols:
        n_cols = args.cols
    dt_gcm, factor = 900.0, 1.0
    eng = Engine("cuda:%d" % local)
    stream = torch.cuda.current_stream()
    sptr = ctypes.c_void_p(stream.cuda_stream)

    fplans, bplans, host0 = [], [], None
    for r in range(args.rotate):
        gcm, zf, zh, prof = synthetic.make_batch(n_cols, nG, nL, seed=seed + 1000 * rank + r, couple_surface=False)
        # hot path only: the rain-rate diagnostic (spcpl.py:325, written to spifs only) is not part of the byte model
        prof = {k: v for k, v in prof.items() if k not in ("Rain", "rain_last")}
        if r == 0:
            host0 = (gcm, zf, zh, prof)
        g = {k: torch.from_numpy(v).to(eng.device) for k, v in gcm.items()}
        p = {k: torch.from_numpy(v).to(eng.device) for k, v in prof.items()}
        zf_d, zh_d = torch.from_numpy(zf).to(eng.device), torch.from_numpy(zh).to(eng.device)
        # K1 writes only the arrays the algorithmic-byte model counts (+ idx); K3 recomputes Zf from geopotential
        fplans.append(eng.plan_forward(g, zf_d, p, factor, dt_gcm, zh=zh_d, want_heights=False,
                                       cols_per_block=args.cols_per_block))
        bplans.append(eng.plan_backward(g, zf_d, p, factor, dt_gcm, Zf=None, want_start_index=False,
                                        cols_per_block=args.cols_per_block))
    R = args.rotate

    def step(i):
        fplans[i % R].launch_raw(sptr)
        bplans[i % R].launch_raw(sptr)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fence()
    if dist is not None:
        tt = torch.tensor([elapsed], device=eng.device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-kernel durations with HIP events on the launch stream (same plans, same rotating batches).
    # (a) event pairs bracketing every kernel of `ke` further steps; an event pair costs a few us of its
    #     own on this stack, measured live with empty pairs and subtracted;
    # (b) K1 alone / K3 alone, `ke` back-to-back launches between two events (inter-kernel gap ~40 ns
    #     in the rocprofv3 trace), the figure the roofline uses.
    ab = algorithmic_bytes(nG, nL)
    k1_us = k3_us = None
    kdiag = {}
    if not args.no_kernel_events:
        ke = min(args.steps, 400)
        E = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
        evs = [[E() for _ in range(3)] for _ in range(ke)]
        emp = [[E() for _ in range(2)] for _ in range(64)]
        torch.cuda.synchronize()
        for i in range(ke):
            evs[i][0].record(stream)
            fplans[i % R].launch_raw(sptr)
            evs[i][1].record(stream)
            bplans[i % R].launch_raw(sptr)
            evs[i][2].record(stream)
        for a, b in emp:
            a.record(stream)
            b.record(stream)
        torch.cuda.synchronize()
        empty = float(numpy.median([a.elapsed_time(b) for a, b in emp])) * 1e3
        kdiag["event_pair_k1_us"] = float(numpy.mean([e[0].elapsed_time(e[1]) for e in evs])) * 1e3
        kdiag["event_pair_k3_us"] = float(numpy.mean([e[1].elapsed_time(e[2]) for e in evs])) * 1e3
        kdiag["empty_event_pair_us"] = empty
        res = {}
        for name, plans in (("k1", fplans), ("k3", bplans)):
            e0, e1 = E(), E()
            for i in range(32):
                plans[i % R].launch_raw(sptr)
            torch.cuda.synchronize()
            e0.record(stream)
            for i in range(ke):
                plans[i % R].launch_raw(sptr)
            e1.record(stream)
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e3 / ke
        k1_us, k3_us = res["k1"], res["k3"]
        # the same kernels re-launched on ONE batch (inputs resident in the 256 MB Infinity Cache): reported as the
        # warm half of the cold/warm pair BASELINE.md section 3 asks for; never used for `value` or `roofline.frac`
        for name, plans in (("k1_warm", fplans), ("k3_warm", bplans)):
            e0, e1 = E(), E()
            for i in range(32):
                plans[0].launch_raw(sptr)
            torch.cuda.synchronize()
            e0.record(stream)
            for i in range(ke):
                plans[0].launch_raw(sptr)
            e1.record(stream)
            torch.cuda.synchronize()
            kdiag[name + "_us"] = e0.elapsed_time(e1) * 1e3 / ke

    # measured device-to-device copy rate of this box (16 B/lane streaming copy, 256 MiB, read + write bytes):
    # the practical HBM ceiling reported next to the 8 TB/s spec peak
    copy_gbs = None
    if not args.no_kernel_events:
        src = torch.empty(1 << 28, dtype=torch.uint8, device=eng.device)
        dst = torch.empty_like(src)
        for _ in range(3):
            eng.stream_copy(dst, src, stream)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(10):
            eng.stream_copy(dst, src, stream)
        c1.record(stream)
        torch.cuda.synchronize()
        copy_gbs = 2.0 * src.numel() * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_cols = n_cols * world * args.steps
    value = total_cols / elapsed
    out = {
        "metric": "SP column-exchanges/sec (GCM<->LES forcing+tendency)",
        "value": value, "unit": "column-exchanges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config %d: %d synthetic SP columns per GPU, %d GCM <-> %d LES levels, fp64, "
                               "%d rotating batches resident in HBM" % (args.config, n_cols, nG, nL, R),
                   "n_cols_per_gpu": n_cols, "nG": nG, "nL": nL, "rotate": R,
                   "launches_per_step": 2, "parallelism": "columns sharded, no collective"},
        "bytes_per_exchange": ab["exchange"],
        "hbm_frac_whole_step": value / world * ab["exchange"] / 1e9 / HBM_PEAK_GBS,
    }
    if k1_us is not None:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))     # PMC passes of tools/gpu_round.sh (separate --pmc runs, calibrated)
                if tj.get("n_cols") == n_cols and (nG, nL) == (91, 160):
                    traffic = tj.get("k_forward_bytes_per_launch")
            except Exception:
                traffic = None
        ach = ab["k1_launch"] * n_cols / (k1_us * 1e-6) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_forward<double,false> (K1+K2 fused)", "achieved": ach,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "algorithmic_bytes_per_launch": ab["k1_launch"] * n_cols, "avg_launch_us": k1_us,
                           "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": (ach / copy_gbs) if copy_gbs else None,
                           "timing": dict(kdiag, method="HIP events around %d back-to-back launches of the kernel "
                                          "alone on the launch stream" % min(args.steps, 400)),
                           "backward": {"kernel": "k_backward<double> (K3)", "avg_launch_us": k3_us,
                                        "achieved": ab["k3_launch"] * n_cols / (k3_us * 1e-6) / 1e9,
                                        "algorithmic_bytes_per_launch": ab["k3_launch"] * n_cols}}
    if args.cpu_seconds > 0 and world == 1:
        gcm, zf, zh, prof = host0
        out["cpu_baseline"] = cpu_baseline(gcm, zf, zh, prof, dt_gcm, factor, args.cpu_seconds)
        workers = min(16, os.cpu_count() or 1)
        if workers > 1 and not args.no_cpu_multicore:
            try:
                out["cpu_baseline"]["all_cores"] = cpu_baseline_multicore(gcm, zf, zh, prof, min(6.0, args.cpu_seconds), workers)
            except Exception as e:                       # a reported extra, never fatal
                out["cpu_baseline"]["all_cores"] = {"error": repr(e)}
    elif args.cpu_seconds > 0:
        out["cpu_baseline"] = None   # reported at N=1 only
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-worker":
        cpu_worker_main(sys.argv[2], float(sys.argv[3]))
    else:
        main()
