#!/usr/bin/env python3
"""bench.py -- chi-square association scan throughput on MI355X.

A "step" is one pass of the hot path (assoc scan kernel + chi-square statistics
kernel, and for N > 1 the gather of the result blocks to rank 0) over one
device-resident synthetic cohort shard.  Inputs are generated ON DEVICE before
the timed region (SURVEY.md 8d generator), so nothing crosses PCIe while timing.

  python bench.py --gpus N --steps K --warmup W [--workload c2|m8|c4tdt] [...]

N = 1 default workload is BASELINE.json configs[1] ("c2": 1M biallelic SNP x 10k
case/control): the metric's own 10M x 50k cohort is 500 GB and does not fit one
GPU.  "m8" is the per-GPU shard of that metric cohort at 8 GPUs (1.25M x 50k,
62.5 GB); "c5" one 100 GB tile of a GPU's shard of the 40M x 100k cohort.
Scaling is weak: every rank scans one such shard.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (variants per GPU, samples, description)
    "c2": (1_000_000, 10_000, "assoc --chisq, synthetic 1M biallelic SNP x 10k case/control (BASELINE configs[1])"),
    "m8": (1_250_000, 50_000, "assoc --chisq, per-GPU shard (1/8) of the 10M SNP x 50k metric cohort"),
    "c5": (1_000_000, 100_000, "assoc --chisq, one 100 GB tile (1 of 5) of a GPU's shard of the 40M SNP x 100k cohort on 8 GPUs (BASELINE configs[4])"),
    "smoke": (20_000, 2_000, "assoc --chisq, tiny smoke cohort"),
    # secondary configs (not the headline metric; run with --workload):
    "c3": (1_000_000, 10_000, "assoc --fisher on the 1M x 10k cohort (BASELINE configs[2]): scan + Fisher p-pass"),
    "c4": (2_000_000, 15_000, "tdt, 2M SNP x 5k trios (BASELINE configs[3]): trio scan + TDT statistics"),
}
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--variants", type=int, default=0, help="override variants per GPU")
    ap.add_argument("--samples", type=int, default=0, help="override samples")
    ap.add_argument("--gather-chunks", type=int, default=1,
                    help="N>1: the shard is scanned in this many variant blocks, each followed by its own "
                         "asynchronous result gather; with the default 1 the whole step's gather overlaps "
                         "the next step's scan (result blocks are double-buffered)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="bracket the scan kernel with HIP events on every n-th timed step (an event pair costs "
                         "tens of microseconds of queue bubbles, so not every step carries one)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1; gloo is a rehearsal mode (blocks are staged through host "
                         "memory and several ranks may share one GPU), never a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target wall time of each CPU baseline leg")
    ap.add_argument("--option", action="append", default=[], help="engine option key=value")
    return ap.parse_args()


def pmc_traffic(workload, variants, samples, pitch, kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/*_pmc_traffic_<workload>.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate runs, gfx950 correction applied).  None when no matching profile."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_%s.json" % workload))):
        try:
            d = json.load(open(f))
            if d["variants"] == variants and d["samples"] == samples and d["row_pitch_bytes"] == pitch:
                for k, v in d["kernels"].items():
                    if kernel.startswith(k):
                        best = v["hbm_bytes_per_launch"]
        except Exception:
            pass
    return best


def cpu_baseline(n_samples, cond, target_s):
    """Oracle (oracle/hpgv_oracle.c) timed on the host cores of this box: the
    reference's worker structure (OpenMP workers pulling 200-variant batches),
    text-faithful scan = per genotype strdup + parse + free as assoc.c:50-57."""
    from oracle import pyoracle as orc
    cores = orc.effective_cpus()                        # affinity mask capped by the cgroup CPU quota
    # pilot to size the bounded sample
    pilot = 200 * cores
    sec, used = orc.baseline_assoc_text(0, pilot, n_samples, cond, cores)
    rate = pilot / max(sec, 1e-9)
    n_text = int(min(max(rate * target_s, pilot), 5_000_000))
    n_text = max(200, (n_text // 200) * 200)
    sec, used = orc.baseline_assoc_text(0, n_text, n_samples, cond, cores)
    text_rate = n_text / sec
    # packed leg: same int8 matrix in host RAM
    n_packed = int(min(4_000_000_000 // max(n_samples, 1), 400_000))
    gt = orc.synth_matrix(0, n_packed, n_samples, n_samples)
    reps, psec = 0, 0.0
    while psec < target_s / 2 and reps < 50:
        s, _ = orc.baseline_assoc_packed(gt, n_samples, cond, cores)
        psec += s
        reps += 1
    packed_rate = n_packed * reps / psec
    return {
        "value": text_rate, "unit": "variants/s", "cores": used, "kind": "port",
        "sample": "text-faithful scan (strdup+parse+free per genotype, assoc.c:50-57) of %d synthetic variants x %d samples, "
                  "OpenMP workers on 200-variant batches, %.1f s" % (n_text, n_samples, sec),
        "packed_value": packed_rate,
        "packed_sample": "same loop on the packed int8 matrix in host RAM: %d variants x %d passes, %.1f s" % (n_packed, reps, psec),
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; this engine has no CPU path", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    hpgv = importlib.import_module("hpg-variant_amd")
    from importlib import import_module
    sharding = import_module("hpg-variant_amd.sharding")

    V, N, desc = WORKLOADS[args.workload]
    if args.variants:
        V = args.variants
    if args.samples:
        N = args.samples

    eng = hpgv.Engine(dev_index)
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    kind = {"c3": "fisher", "c4": "tdt"}.get(args.workload, "chisq")
    cond = (np.arange(N) % 2).astype(np.uint8)          # odd samples are cases
    fam = None
    if kind == "tdt":
        n_tr = N // 3                                   # trio k = columns (3k, 3k+1, 3k+2), child sex alternating
        kk = np.arange(n_tr)
        fam = (3 * kk, 3 * kk + 1, np.arange(n_tr + 1), 3 * kk + 2, (kk % 2).astype(np.uint8))
        nA, nU, pitch = eng.set_families(3 * n_tr, *fam)      # (fast trios, slow families, pitch)
        which, res_bytes, payload, scan_name = hpgv.LAYOUT_TDT, 32, 32, "k_tdt_scan"
    else:
        nA, nU, pitch = eng.set_cohort(cond)
        which = hpgv.LAYOUT_ASSOC
        scan_name = "k_assoc_scan" if "pipeline=0" in args.option else "k_assoc_scan_pipe"
        res_bytes, payload = (40, 40) if kind == "chisq" else (32, 32)
        if kind == "fisher":
            # ln(i!) table with num_samples * 10 entries, as assoc_runner.c:164-166 builds it (an INPUT of the pass)
            lf_table = np.concatenate([[0.0], np.cumsum(np.log(np.arange(1, N * 10, dtype=np.float64)))])
            eng.set_logfact(lf_table)
    head = 16 if kind != "tdt" else 8                   # integer tallies at the front of a result block

    # device memory owned by torch (plumbing); raw pointers cross the C ABI
    gt = torch.empty(V * pitch, dtype=torch.uint8, device=dev)
    res = torch.empty(res_bytes * V, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    v0 = rank * V                                       # global variant ids of this shard
    eng.synth(which, v0, V, gt.data_ptr(), sp)
    torch.cuda.synchronize()

    chunks = max(1, args.gather_chunks) if world > 1 else 1
    bounds = [sharding.variant_range(c, chunks, V) for c in range(chunks)]
    # a block's result = tallies[n] | f64 arrays[n] ...: each block lives in its own tensor, so the
    # gather of block i (overlapping the scan of block i+1) needs no repacking
    # two generations of result blocks: the gathers of step k are only waited for at the end of step k+1
    # (they overlap that step's scans), so the blocks of step k+1 must not reuse step k's buffers
    gens = 1 if world == 1 else 2
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    chunk_res = [[res] if world == 1 else [torch.empty(res_bytes * (hi - lo), dtype=torch.uint8, device=dev) for lo, hi in bounds]
                 for _ in range(gens)]
    recv = None
    if world > 1 and rank == 0:
        recv = [[[torch.empty(res_bytes * (hi - lo), dtype=torch.uint8, device=comm_dev) for _ in range(world)]
                 for lo, hi in bounds] for _ in range(gens)]
    pending = []                                       # works of the previous step

    def scan_block(lo, n, b):
        g = gt.data_ptr() + lo * pitch
        if kind == "tdt":
            eng.tdt_scan(g, n, b, None, sp)
        else:
            eng.assoc_scan(g, n, b, None, sp)

    def stats_block(n, b):
        if kind == "chisq":
            eng.assoc_chisq(b, n, b + 16 * n, b + 24 * n, b + 32 * n, sp)
        elif kind == "fisher":
            eng.assoc_fisher(b, n, b + 16 * n, b + 24 * n, sp)
        else:
            eng.tdt_stats(b, n, b + 8 * n, b + 16 * n, b + 24 * n, sp)

    step_no = [0]

    def step(ev=None):
        g = step_no[0] % gens
        step_no[0] += 1
        works = []
        for c, (lo, hi) in enumerate(bounds):
            n = hi - lo
            blk = chunk_res[g][c]
            b = blk.data_ptr()
            if ev and c == 0:
                ev[0].record(stream)
            scan_block(lo, n, b)
            if ev and c == 0:
                ev[1].record(stream)
            stats_block(n, b)
            if world > 1:
                send = blk if args.backend == "nccl" else blk.cpu()      # gloo rehearsal: through host memory
                _, w = sharding.gather_blocks(send, [res_bytes * n] * world, dst=0, async_op=True,
                                              out_bufs=recv[g][c] if recv else None)
                works.append(w)
        # the previous step's gathers have had this whole step to finish
        for w in pending:
            w.wait()
        pending[:] = works

    def drain():
        for w in pending:
            w.wait()
        pending[:] = []

    def barrier():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    every = max(1, args.event_every)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range((args.steps + every - 1) // every)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k // every] if k % every == 0 else None)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    scan_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    scan_variants = bounds[0][1] - bounds[0][0]
    bytes_per_variant = N + payload                     # N x 1 B + result payload
    achieved = scan_variants * bytes_per_variant / (scan_ms * 1e-3) / 1e9

    # ---- measured streaming-read ceiling of the same buffer (report both fractions) -----------
    probe_gbps = None
    if rank == 0 and world == 1:
        try:
            probe_ms = eng.read_probe(gt.data_ptr(), V * pitch, 5)
            probe_gbps = V * pitch / (probe_ms * 1e-3) / 1e9
        except Exception:
            probe_gbps = None

    # ---- parity spot check: the oracle as CHECKER of the timed run's outputs (not timed, not on the product path)
    parity = None
    parity_failed = False
    if rank == 0:
        from oracle import pyoracle as orc
        n0 = bounds[0][1] - bounds[0][0]
        g_last = (step_no[0] - 1) % gens
        # blocks to check: this rank's first block and, for N > 1, the first block GATHERED from the last rank
        sources = [(chunk_res[g_last][0], v0)]
        if world > 1:
            sources.append((recv[g_last][0][world - 1], (world - 1) * V))
        idx = np.unique(np.concatenate([np.arange(min(256, n0)), np.arange(0, n0, 1000), np.arange(max(0, n0 - 256), n0)]))
        if kind == "fisher":
            idx = idx[:: max(1, len(idx) // 400)]
        ok = True

        def same(got, exp):
            with np.errstate(invalid="ignore"):
                return bool(np.all((np.abs(got - exp) <= 1e-10 * np.maximum(1, np.abs(exp))) | (np.isnan(got) & np.isnan(exp))))
        ncol = N if kind != "tdt" else 3 * (N // 3)
        for blk, vbase in sources:
            ints = blk[: head * n0].view(torch.int32).view(n0, head // 4).cpu().numpy()
            stats = blk[head * n0:].view(torch.float64).view(-1, n0).cpu().numpy()
            for lo in range(0, len(idx), 512):
                sel = idx[lo: lo + 512]
                rows = np.stack([orc.synth_matrix(vbase + int(v), 1, ncol, ncol)[0] for v in sel])
                if kind == "tdt":
                    t1, t2 = orc.tdt_counts(rows, *fam)
                    ok &= bool(np.array_equal(ints[sel], np.stack([t1, t2], 1)))
                    exp = orc.tdt_stats(t1, t2)
                    ok &= all(same(stats[j][sel], exp[j]) for j in range(3))
                else:
                    A1, A2, U1, U2 = orc.assoc_counts(rows, cond)
                    ok &= bool(np.array_equal(ints[sel], np.stack([A1, A2, U1, U2], 1)))
                    if kind == "chisq":
                        exp = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
                        ok &= all(same(stats[j][sel], exp[j]) for j in range(3))
                    else:
                        odds, _, p = orc.assoc_stats(orc.TASK_FISHER, A1, A2, U1, U2, lf_table)
                        ok &= same(stats[0][sel], odds) and same(stats[1][sel], p)
        parity = {"checked_variants": int(len(idx)) * len(sources), "blocks": len(sources), "ok": bool(ok)}

    if rank == 0:
        total_variants = V * world * args.steps
        out = {
            "metric": {"chisq": "variants/s chi2 assoc", "fisher": "variants/s fisher assoc", "tdt": "variants/s tdt"}[kind],
            "value": total_variants / elapsed,
            "unit": "variants/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (on-device splitmix64 cohort, HWE genotypes, 1% missing, odd samples are cases)",
            "config": {"workload": "%s: %s" % (args.workload, desc), "variants_per_gpu": V, "samples": N,
                       ("affected" if kind != "tdt" else "trios"): nA, ("unaffected" if kind != "tdt" else "multi_child_families"): nU,
                       "row_pitch_bytes": pitch,
                       "parallelism": "variant-sharded x%d%s" % (world, ", result gather to rank 0 (%s) in %d blocks overlapped with the scans" % (args.backend, chunks) if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic(args.workload, scan_variants, N, pitch, scan_name) if world == 1 else None,
                         "kernel": scan_name, "kernel_ms": scan_ms, "kernel_samples": len(evs),
                         "stream_read_probe_GBps": probe_gbps,
                         "frac_of_probe": (achieved / probe_gbps) if probe_gbps else None,
                         "algorithmic_bytes_per_variant": bytes_per_variant,
                         "variants_per_launch": scan_variants},
            "parity": parity,
        }
        if world == 1 and not args.no_cpu_baseline and kind == "chisq":
            out["cpu_baseline"] = cpu_baseline(N, cond, args.cpu_seconds)
        print(json.dumps(out))
        sys.stdout.flush()
        if parity is not None and not parity["ok"]:
            print("bench.py: the timed run's outputs DIFFER from the oracle on the sampled variants: the number above is invalid",
                  file=sys.stderr)
            parity_failed = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
