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
62.5 GB).  Scaling is weak: every rank scans one such shard.

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
    "smoke": (20_000, 2_000, "assoc --chisq, tiny smoke cohort"),
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
    ap.add_argument("--gather-chunks", type=int, default=8,
                    help="N>1: the shard is scanned in this many variant blocks so that the result "
                         "gather of block i overlaps the scan of block i+1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target wall time of each CPU baseline leg")
    ap.add_argument("--option", action="append", default=[], help="engine option key=value")
    return ap.parse_args()


def cpu_baseline(n_samples, cond, target_s):
    """Oracle (oracle/hpgv_oracle.c) timed on the host cores of this box: the
    reference's worker structure (OpenMP workers pulling 200-variant batches),
    text-faithful scan = per genotype strdup + parse + free as assoc.c:50-57."""
    from oracle import pyoracle as orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    hpgv = importlib.import_module("hpg-variant_amd")
    from importlib import import_module
    sharding = import_module("hpg-variant_amd.sharding")

    V, N, desc = WORKLOADS[args.workload]
    if args.variants:
        V = args.variants
    if args.samples:
        N = args.samples

    eng = hpgv.Engine(local_rank)
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    cond = (np.arange(N) % 2).astype(np.uint8)          # odd samples are cases
    nA, nU, pitch = eng.set_cohort(cond)

    # device memory owned by torch (plumbing); raw pointers cross the C ABI
    gt = torch.empty(V * pitch, dtype=torch.uint8, device=dev)
    lay = sharding.result_block_layout(V)
    res = torch.empty(lay["bytes"], dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    v0 = rank * V                                       # global variant ids of this shard
    eng.synth(hpgv.LAYOUT_ASSOC, v0, V, gt.data_ptr(), sp)
    torch.cuda.synchronize()

    base = res.data_ptr()
    chunks = max(1, args.gather_chunks) if world > 1 else 1
    bounds = [sharding.variant_range(c, chunks, V) for c in range(chunks)]
    # per-chunk result blocks are contiguous sub-blocks, so a chunk's gather needs no repacking:
    # block c = counts[lo:hi] | odds[lo:hi] | chisq[lo:hi] | p[lo:hi] lives in its own tensor
    chunk_res = [torch.empty(40 * (hi - lo), dtype=torch.uint8, device=dev) for lo, hi in bounds] if world > 1 else None
    recv = None
    if world > 1 and rank == 0:
        recv = [[torch.empty(40 * (hi - lo), dtype=torch.uint8, device=dev) for _ in range(world)] for lo, hi in bounds]

    def step(ev=None):
        if world == 1:
            if ev:
                ev[0].record(stream)
            eng.assoc_scan(gt.data_ptr(), V, base + lay["counts"], None, sp)
            if ev:
                ev[1].record(stream)
            eng.assoc_chisq(base + lay["counts"], V, base + lay["odds"], base + lay["chisq"], base + lay["p"], sp)
            return None
        works = []
        for c, (lo, hi) in enumerate(bounds):
            n = hi - lo
            b = chunk_res[c].data_ptr()
            if ev and c == 0:
                ev[0].record(stream)
            eng.assoc_scan(gt.data_ptr() + lo * pitch, n, b, None, sp)
            if ev and c == 0:
                ev[1].record(stream)
            eng.assoc_chisq(b, n, b + 16 * n, b + 24 * n, b + 32 * n, sp)
            _, w = sharding.gather_blocks(chunk_res[c], [40 * n] * world, dst=0, async_op=True,
                                           out_bufs=recv[c] if recv else None)
            works.append(w)
        for w in works:
            w.wait()
        return None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    scan_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    scan_variants = V if world == 1 else (bounds[0][1] - bounds[0][0])
    bytes_per_variant = N + 40                          # N x 1 B + 40 B result payload
    achieved = scan_variants * bytes_per_variant / (scan_ms * 1e-3) / 1e9

    # ---- parity spot check against the oracle (not timed) -----------------------
    parity = None
    if rank == 0:
        from oracle import pyoracle as orc
        if world == 1:
            counts = res[: 16 * V].view(torch.int32).view(V, 4).cpu().numpy()
            stats = res[16 * V:].view(torch.float64).view(3, V).cpu().numpy()
        else:
            n0 = bounds[0][1] - bounds[0][0]
            counts = chunk_res[0][: 16 * n0].view(torch.int32).view(n0, 4).cpu().numpy()
            stats = chunk_res[0][16 * n0:].view(torch.float64).view(3, n0).cpu().numpy()
        nchk = counts.shape[0]
        idx = np.unique(np.concatenate([np.arange(min(256, nchk)), np.arange(0, nchk, 1000),
                                        np.arange(max(0, nchk - 256), nchk)]))
        ok = True
        for lo in range(0, len(idx), 512):
            sel = idx[lo: lo + 512]
            rows = np.stack([orc.synth_matrix(v0 + int(v), 1, N, N)[0] for v in sel])
            A1, A2, U1, U2 = orc.assoc_counts(rows, cond)
            odds, chisq, p = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
            ok &= bool(np.array_equal(counts[sel], np.stack([A1, A2, U1, U2], 1)))
            for got, exp in ((stats[0][sel], odds), (stats[1][sel], chisq), (stats[2][sel], p)):
                with np.errstate(invalid="ignore"):
                    ok &= bool(np.all((np.abs(got - exp) <= 1e-10 * np.maximum(1, np.abs(exp))) |
                                      (np.isnan(got) & np.isnan(exp))))
        parity = {"checked_variants": int(len(idx)), "ok": ok}

    if rank == 0:
        total_variants = V * world * args.steps
        out = {
            "metric": "variants/s chi2 assoc",
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
                       "affected": nA, "unaffected": nU, "row_pitch_bytes": pitch,
                       "parallelism": "variant-sharded x%d%s" % (world, ", result gather to rank 0 in %d overlapped blocks" % chunks if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": "k_assoc_scan", "kernel_ms": scan_ms,
                         "algorithmic_bytes_per_variant": bytes_per_variant,
                         "variants_per_launch": scan_variants},
            "parity": parity,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, cond, args.cpu_seconds)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
