#!/usr/bin/env python3
"""bench.py -- per-variant statistics scan throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload m|c2|c3|c4|c5|m8|stats] [...]

A "step" is one pass of the hot path over one synthetic cohort: scan kernel +
statistics kernel over every variant of the rank's shard, and for N > 1 the
gather of the result blocks to rank 0 (RCCL, overlapped with the next scan).
Inputs are generated ON DEVICE (SURVEY.md 8d generator, bit-identical to the
oracle's), so nothing crosses PCIe while timing.

Default workload "m" is the cohort BASELINE.json's metric is quoted on: 10M SNP
x 50k samples (500 GB), STRONG-scaled: rank g scans variants [g*10M/N, (g+1)*10M/N).
A rank's shard is cut into tiles of at most --tile-gb (126 GB: SURVEY 8d row M).
When the whole shard fits the GPU's free memory (N >= 2 on a 288 GiB part) it is
resident and the K steps are timed in ONE region, barrier + device sync on both
sides (the contract).  When it does not fit (N = 1: 500 GB) ONE tile buffer is
regenerated on the device before each tile's scan, OUTSIDE the timed region: the
timed region is then the sum of the per-tile segments, each bracketed by a device
sync (+ barrier), and `config.timed_region` says so.  The other workloads are
one resident shard per rank (weak scaling): c2 / c3 / c4 / c5 are BASELINE
configs[1..4]'s single-GPU shapes, m8 the 1/8 shard of the metric cohort, stats
the vcf-stats counters + Hardy-Weinberg on 1M x 10k.

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts its own N
ranks: the parent makes no GPU call, spawns one fresh child per rank (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) and relays rank
0's JSON line.  Under torch.distributed.run the ranks are used as given.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (kind, variants, samples, scaling, description); strong: `variants` is the whole cohort,
    # weak: `variants` per GPU
    "m": ("chisq", 10_000_000, 50_000, "strong",
          "assoc --chisq, the metric cohort: synthetic 10M biallelic SNP x 50k case/control (BASELINE metric)"),
    "c2": ("chisq", 1_000_000, 10_000, "weak", "assoc --chisq, synthetic 1M biallelic SNP x 10k case/control (BASELINE configs[1])"),
    "c3": ("fisher", 1_000_000, 10_000, "weak", "assoc --fisher on the 1M x 10k cohort (BASELINE configs[2]): scan + Fisher p-pass"),
    "c4": ("tdt", 2_000_000, 15_000, "weak", "tdt, 2M SNP x 5k trios (BASELINE configs[3]): trio scan + TDT statistics"),
    "c5": ("chisq", 1_000_000, 100_000, "weak",
           "assoc --chisq, one 100 GB tile (1 of 5) of a GPU's shard of the 40M SNP x 100k cohort on 8 GPUs (BASELINE configs[4])"),
    "c5full": ("chisq", 40_000_000, 100_000, "strong",
               "assoc --chisq, the whole 40M SNP x 100k cohort (4 TB) of BASELINE configs[4], variant-sharded over the ranks: on 8 GPUs "
               "5M variants = 4 tiles of 125 GB per rank, regenerated on the device before each tile's scan; meant for --gpus 8"),
    "m8": ("chisq", 1_250_000, 50_000, "weak", "assoc --chisq, per-GPU shard (1/8) of the 10M SNP x 50k metric cohort"),
    "stats": ("stats", 1_000_000, 10_000, "weak",
              "vcf stats: genotype / allele / missing counters + Hardy-Weinberg on 1M SNP x 10k samples (get_variants_stats)"),
    "smoke": ("chisq", 20_000, 2_000, "weak", "assoc --chisq, tiny smoke cohort"),
}
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# FP64 / 64-bit vector issue peak used for the Fisher p-pass: 256 CUs x 4 SIMDs x 2.4 GHz, one wave64
# instruction per 4 cycles per SIMD (FP64 FMA rate = 78.6 TFLOP/s spec = 16 lanes/clk/SIMD)
VALU64_PEAK_GINST = 256 * 4 * 2.4 / 4.0
# the cheapest instruction sequence for ONE hypergeometric term exp(konst - lf[x] - lf[r1-x] - lf[c1-x] - lf[d0+x]) added to a sum,
# on the reference's own table (see roofline.algorithmic_note): 2 address ops + 4 subtractions + 18 exp + 1 add
FISHER_INSTS_PER_TERM = 25
METRIC = {"chisq": "variants/s chi2 assoc", "fisher": "variants/s fisher assoc", "tdt": "variants/s tdt",
          "stats": "variants/s vcf stats"}
# (bytes of integer tallies at the front of a result block, bytes of a result record = SURVEY 8d payload)
RESULT = {"chisq": (16, 40), "fisher": (16, 32), "tdt": (8, 32), "stats": (32, 48)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="m", choices=sorted(WORKLOADS))
    ap.add_argument("--variants", type=int, default=0, help="override the workload's variant count")
    ap.add_argument("--samples", type=int, default=0, help="override samples")
    ap.add_argument("--tile-gb", type=float, default=126.0, help="largest genotype tile kept in one buffer (GB)")
    ap.add_argument("--resident", choices=["auto", "no"], default="auto",
                    help="no: regenerate every tile before its scan even when the shard would fit (rehearses the N = 1 path)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="bracket the scan kernel with HIP events on every n-th timed launch (an event pair costs "
                         "tens of microseconds of queue bubbles, so not every launch carries one)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1; gloo is a rehearsal mode (blocks are staged through host "
                         "memory and several ranks may share one GPU), never a measurement")
    ap.add_argument("--fisher-pieces", type=int, default=1,
                    help="Fisher workload, experiment: pieces of a tile whose Fisher pass runs on a second stream beside the next piece's scan "
                         "(measured: no gain, profiles/experiments_that_did_not_pay.md; 1 = one after the other)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives all N GPUs through the C ABI's group context (hpgv_create_multi + hpgv_group_*: "
                         "shards scanned on per-device streams, results gathered onto device 0 over the communicator the "
                         "group owns, ncclCommInitAll); no torch.distributed.  What a C host would run")
    ap.add_argument("--devices", default="", help="--single-process: device ids of the group, e.g. 0,0 to rehearse on one GPU")
    ap.add_argument("--force-process-group", action="store_true",
                    help="N = 1: initialise the process group all the same and send the result blocks through its gather "
                         "(one real RCCL rank through the N > 1 code path; a check, not a measurement)")
    ap.add_argument("--configs", choices=["auto", "none", "all"], default="auto",
                    help="attach BASELINE configs[1..3] + stats (c2, c3, c4, stats) as \"configs\" to the line: auto = when "
                         "the default workload runs at N = 1 without overrides")
    ap.add_argument("--config-steps", type=int, default=10)
    ap.add_argument("--config-warmup", type=int, default=3)
    ap.add_argument("--config-cpu-seconds", type=float, default=3.0)
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target wall time of each CPU baseline leg")
    ap.add_argument("--option", action="append", default=[], help="engine option key=value")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# parent: start one fresh process per rank (no GPU call is made here)
# ---------------------------------------------------------------------------------------------------------
def count_gpus_no_hip():
    """GPUs this process may use, WITHOUT any HIP / torch call (the parent of the ranks must not initialise a GPU): the KFD
    topology's nodes with SIMDs (/sys/class/kfd), capped by the render nodes the container is given (/dev/dri/renderD*) and by
    a *_VISIBLE_DEVICES list.  A rank that still finds no device of its own ends with exit code 2 and says so."""
    import glob
    n_kfd = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(prop):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n_kfd += 1
        except (OSError, ValueError):
            pass
    n_render = len(glob.glob("/dev/dri/renderD*"))
    if not os.path.exists("/dev/kfd"):
        return 0
    n = min(n_kfd, n_render) if n_kfd and n_render else max(n_kfd, n_render)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([x for x in os.environ[var].split(",") if x.strip()]))
    return n


def launch_ranks(args):
    n = args.gpus
    if args.backend == "nccl":
        have = count_gpus_no_hip()
        if have < n:
            print("bench.py: --gpus %d but this machine shows %d GPU(s); one rank per GPU over RCCL needs %d "
                  "(rehearse with --backend gloo, which lets ranks share a GPU)" % (n, have, n), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.PIPE if r == 0 else sys.stderr       # only rank 0 prints the JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.append(procs[0].stdout.read().decode()), daemon=True)
    reader.start()
    rc = 0
    live = list(procs)
    while live:                                         # a rank that dies takes the job down: the others would wait for it
        time.sleep(0.2)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                if p.returncode and not rc:
                    rc = p.returncode
                    for q in live:
                        q.terminate()                   # exactly the children started above
    reader.join(timeout=10.0)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------
# helpers of a rank
# ---------------------------------------------------------------------------------------------------------
def pmc_profile(workload, kernel, variants, samples, pitch):
    """Per-launch figures of `kernel` from the committed PMC passes (profiles/*_pmc_*_<workload>.json:
    rocprofv3 --pmc in separate runs, gfx950 corrections applied).  Returns the kernel's dict or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_*_%s.json" % workload))):
        try:
            d = json.load(open(f))
            if d["samples"] == samples and d["row_pitch_bytes"] == pitch:
                for k, v in d["kernels"].items():
                    if kernel.startswith(k) and (v.get("variants_per_launch", d.get("variants")) == variants or "valu_insts_per_variant" in v):
                        best = dict(best or {}, **v)
                        # where each replayed figure comes from: the LAST file that supplied it
                        best.setdefault("_source", {}).update({key: "profiles/" + os.path.basename(f) for key in v})
        except Exception:
            pass
    return best


REPLAYED = "%s (builder's rocprofv3 --pmc pass of this workload, committed; REPLAYED here, not measured in this run)"


def bounded_rate(run, pilot, target_s, cap):
    """Times run(n) on a pilot, then on a sample sized for about target_s seconds; returns (rate, n, seconds, threads)."""
    sec, used = run(pilot)
    n = int(min(max(pilot / max(sec, 1e-9) * target_s, pilot), cap))
    n = max(200, (n // 200) * 200)
    sec, used = run(n)
    return n / sec, n, sec, used


def cpu_baseline(kind, n_samples, cond, fam, lf_table, target_s):
    """Oracle (oracle/hpgv_oracle.c) timed on the host cores of this box: the reference's worker structure
    (OpenMP workers pulling 200-variant batches, hpg-variant.conf:33), text-faithful = per genotype strdup +
    parse + free as assoc.c:50-57 / tdt.c:97-108,150-157 do."""
    from oracle import pyoracle as orc
    import numpy as np
    cores = orc.effective_cpus()                        # affinity mask capped by the cgroup CPU quota
    pilot = 200 * cores
    if kind == "chisq":
        run, what = (lambda n: orc.baseline_assoc_text(0, n, n_samples, cond, cores)), "assoc.c:50-57 scan + chi-square"
    elif kind == "fisher":
        run, what = (lambda n: orc.baseline_fisher_text(0, n, n_samples, cond, lf_table, cores)), \
            "assoc.c:50-57 scan + Fisher's exact test on the ln(i!) table (assoc.c:69-75)"
    elif kind == "tdt":
        run, what = (lambda n: orc.baseline_tdt_text(0, n, n_samples, *fam, cores)), \
            "tdt.c:41-271 family loop (sample_ids look-ups resolved once, not per variant)"
    else:
        run, what = (lambda n: orc.baseline_stats_text(0, n, n_samples, cores)), \
            "get_variants_stats counters + Hardy-Weinberg per variant (definition of this repo: hpg-libs is absent)"
    rate, n, sec, used = bounded_rate(run, pilot, target_s, 5_000_000)
    out = {"value": rate, "unit": "variants/s", "cores": used, "kind": "port",
           "sample": "text-faithful %s, strdup+parse+free per genotype, of %d synthetic variants x %d samples, "
                     "OpenMP workers on 200-variant batches, %.1f s" % (what, n, n_samples, sec)}
    if kind == "chisq":                                 # packed leg: same int8 matrix in host RAM
        n_packed = int(min(4_000_000_000 // max(n_samples, 1), 400_000))
        gt = orc.synth_matrix(0, n_packed, n_samples, n_samples)
        reps, psec = 0, 0.0
        while psec < target_s / 2 and reps < 50:
            s, _ = orc.baseline_assoc_packed(gt, n_samples, cond, cores)
            psec += s
            reps += 1
        out["packed_value"] = n_packed * reps / psec
        out["packed_sample"] = "same loop on the packed int8 matrix in host RAM: %d variants x %d passes, %.1f s" % (n_packed, reps, psec)
    return out


def measure(args, wl, steps, warmup, cpu_seconds, main, world, rank, dev, dev_index, pg):
    """One workload on this rank's GPU: returns (the JSON object on rank 0 / None elsewhere, parity_failed).  `main`: the
    workload named on the command line (its --variants / --samples / --option overrides apply to it alone).  `pg`: a process
    group exists and the result blocks are gathered through it (N > 1, or --force-process-group at N = 1)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    hpgv = importlib.import_module("hpg-variant_amd")
    sharding = importlib.import_module("hpg-variant_amd.sharding")

    kind, V_all, N, scaling, desc = WORKLOADS[wl]
    if args.variants and main:
        V_all = args.variants
    if args.samples and main:
        N = args.samples
    total_per_step = V_all if scaling == "strong" else V_all * world

    eng = hpgv.Engine(dev_index)
    for kv in (args.option if main else []):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    cond = (np.arange(N) % 2).astype(np.uint8)          # odd samples are cases
    fam, lf_table = None, None
    if kind == "tdt":
        n_tr = N // 3                                   # trio k = columns (3k, 3k+1, 3k+2), child sex alternating
        kk = np.arange(n_tr)
        fam = (3 * kk, 3 * kk + 1, np.arange(n_tr + 1), 3 * kk + 2, (kk % 2).astype(np.uint8))
        nA, nU, pitch = eng.set_families(3 * n_tr, *fam)      # (fast trios, slow families, pitch)
        which, scan_name, stats_name = hpgv.LAYOUT_TDT, "k_tdt_scan", "k_tdt_stats"
    elif kind == "stats":
        pitch = eng.set_stats_cohort(N)
        nA, nU = N, 0
        which, scan_name, stats_name = hpgv.LAYOUT_STATS, "k_stats_scan_hs", "k_stats_hwe"
    else:
        nA, nU, pitch = eng.set_cohort(cond)
        which = hpgv.LAYOUT_ASSOC
        scan_name = "k_assoc_scan" if "pipeline=0" in args.option else "k_assoc_scan_pipe"
        stats_name = "k_assoc_chisq" if kind == "chisq" else "k_assoc_fisher"
        if kind == "fisher":
            # ln(i!) table with num_samples * 10 entries, as assoc_runner.c:164-166 builds it (an INPUT of the pass)
            lf_table = np.concatenate([[0.0], np.cumsum(np.log(np.arange(1, N * 10, dtype=np.float64)))])
            eng.set_logfact(lf_table)
    head, res_bytes = RESULT[kind]

    # ---- tiles of the shard (strong: rank g scans [g*V/G, (g+1)*V/G); weak: one shard of V_all per rank) -----------------
    free_b, total_b = torch.cuda.mem_get_info(dev)
    avail = torch.tensor([free_b], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
    if pg:                                                           # every rank plans with the tightest rank's memory
        dist.all_reduce(avail, op=dist.ReduceOp.MIN)
    tile_bytes = min(int(args.tile_gb * 1e9), int(0.9 * (int(avail.item()) - (8 << 30))))     # a tile buffer must fit whatever the part
    plan = sharding.plan_tiles(rank, world, V_all, pitch, max(tile_bytes, pitch), strong=(scaling == "strong"))
    v_lo, V, n_tiles, per_tile, tiles = plan["v_lo"], plan["n"], plan["n_tiles"], plan["per_tile"], plan["tiles"]
    res_need = 2 * res_bytes * per_tile * n_tiles * (1 + (world if rank == 0 and pg else 0))
    resident = args.resident == "auto" and (V * pitch + res_need + (6 << 30) <= free_b)
    if pg:                                                           # every rank takes the same path
        flag = torch.tensor([1 if resident else 0], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        resident = bool(flag.item())
    n_bufs = n_tiles if resident else 1

    # device memory owned by torch (plumbing); raw pointers cross the C ABI
    gt = [torch.empty(max(per_tile, 1) * pitch, dtype=torch.uint8, device=dev) for _ in range(n_bufs)]
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    def generate(ti, buf):
        lo, hi = tiles[ti]
        if hi > lo:
            eng.synth(which, v_lo + lo, hi - lo, buf.data_ptr(), sp)

    if resident:
        for ti in range(n_tiles):
            generate(ti, gt[ti])
    torch.cuda.synchronize()

    # result blocks: tallies[n] | f64 arrays[n] ..., one block per tile and generation.  Two generations for
    # N > 1: the gathers of step k are only waited for at the end of step k+1 (they overlap its scans).
    gens = 2 if pg else 1
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    blocks = [[torch.empty(res_bytes * max(hi - lo, 1), dtype=torch.uint8, device=dev) for lo, hi in tiles] for _ in range(gens)]
    # sizes of every rank's block of tile ti (known to all ranks: they follow from variant_range)
    def tile_sizes(ti):
        return [res_bytes * c for c in plan["counts"][ti]]

    recv = None
    if pg and rank == 0:
        recv = [[[torch.empty(max(max(tile_sizes(ti)), 1), dtype=torch.uint8, device=comm_dev) for _ in range(world)]
                 for ti in range(n_tiles)] for _ in range(gens)]
    pending = []                                        # gather works of the previous step
    side = torch.cuda.Stream(device=dev) if kind == "fisher" else None
    side_ev = [torch.cuda.Event() for _ in range(2)] if kind == "fisher" else None
    side_done = torch.cuda.Event() if kind == "fisher" else None

    def run_tile(ti, g, buf, ev=None):
        lo, hi = tiles[ti]
        n = hi - lo
        blk = blocks[g][ti]
        b = blk.data_ptr()
        if n > 0 and kind == "fisher" and ev is None and args.fisher_pieces > 1:
            # the scan is bound by HBM, the Fisher pass by arithmetic: the tile goes through in pieces, piece p's Fisher pass on a
            # second stream beside piece p + 1's scan.  Launches that carry events (every n-th) run one after the other, so that
            # the kernels' own durations are measured alone.
            P = min(args.fisher_pieces, n)
            for pi in range(P):
                a, z = n * pi // P, n * (pi + 1) // P
                eng.assoc_scan(buf.data_ptr() + a * pitch, z - a, b + 16 * a, None, sp)
                side_ev[pi % 2].record(stream)
                side.wait_event(side_ev[pi % 2])
                eng.assoc_fisher(b + 16 * a, z - a, b + 16 * n + 8 * a, b + 24 * n + 8 * a, side.cuda_stream)
            side_done.record(side)
            stream.wait_event(side_done)
        elif n > 0:
            if ev:
                ev[0].record(stream)
            if kind == "tdt":
                eng.tdt_scan(buf.data_ptr(), n, b, None, sp)
            elif kind == "stats":
                eng.stats_scan(buf.data_ptr(), n, b, sp)
            else:
                eng.assoc_scan(buf.data_ptr(), n, b, None, sp)
            if ev:
                ev[1].record(stream)
                ev[2].record(stream)
            if kind == "chisq":
                eng.assoc_chisq(b, n, b + 16 * n, b + 24 * n, b + 32 * n, sp)
            elif kind == "fisher":
                eng.assoc_fisher(b, n, b + 16 * n, b + 24 * n, sp)
            elif kind == "tdt":
                eng.tdt_stats(b, n, b + 8 * n, b + 16 * n, b + 24 * n, sp)
            else:
                eng.stats_hwe(b, n, b + 32 * n, b + 40 * n, sp)
            if ev:
                ev[3].record(stream)
        if pg:
            sizes = tile_sizes(ti)
            send = blk[: res_bytes * n] if args.backend == "nccl" else blk[: res_bytes * n].cpu()   # gloo rehearsal: via host
            _, w = sharding.gather_blocks(send, sizes, dst=0, async_op=True, out_bufs=recv[g][ti] if recv else None, force=True)
            return [w]
        return []

    def drain():
        for w in pending:
            w.wait()
        pending[:] = []

    def barrier():
        drain()
        torch.cuda.synchronize()
        if pg:
            dist.barrier()
        torch.cuda.synchronize()

    every = max(1, args.event_every)
    evs = []
    launch_no = [0]

    def maybe_events(timed):
        if not timed:
            return None
        launch_no[0] += 1
        if (launch_no[0] - 1) % every:
            return None
        e = tuple(torch.cuda.Event(enable_timing=True) for _ in range(4))
        evs.append(e)
        return e

    step_no = [0]
    elapsed = 0.0
    if resident:
        def step(timed):
            g = step_no[0] % gens
            step_no[0] += 1
            works = []
            for ti in range(n_tiles):
                works += run_tile(ti, g, gt[ti], maybe_events(timed))
            for w in pending:                           # the previous step's gathers have had this whole step to finish
                w.wait()
            pending[:] = works
        for _ in range(warmup):
            step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        for k in range(warmup + steps):
            timed = k >= warmup
            g = step_no[0] % gens
            step_no[0] += 1
            for ti in range(n_tiles):
                generate(ti, gt[0])                     # outside the timed region: the tile is resident when its segment starts
                barrier()
                t0 = time.perf_counter()
                pending[:] = run_tile(ti, g, gt[0], maybe_events(timed))
                barrier()
                if timed:
                    elapsed += time.perf_counter() - t0
    if pg:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    scan_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs])) if evs else float("nan")
    stats_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in evs])) if evs else float("nan")
    scan_variants = tiles[0][1] - tiles[0][0]
    bytes_per_variant = N + res_bytes                   # N x 1 B + result payload
    achieved = scan_variants * bytes_per_variant / (scan_ms * 1e-3) / 1e9

    # ---- measured streaming-read ceiling of the same buffer (report both fractions) -----------
    probe_gbps = None
    if rank == 0 and world == 1 and main:
        try:
            nb = scan_variants * pitch
            probe_ms = eng.read_probe(gt[0].data_ptr(), nb, 5)
            probe_gbps = nb / (probe_ms * 1e-3) / 1e9
        except Exception:
            probe_gbps = None

    # ---- parity spot check: the oracle as CHECKER of the timed run's outputs (not timed, not on the product path)
    parity = None
    parity_failed = False
    out = None
    if rank == 0:
        from oracle import pyoracle as orc
        g_last = (step_no[0] - 1) % gens
        # blocks to check: this rank's first and last tile and, for N > 1, the last tile GATHERED from the last rank
        sources = [(blocks[g_last][0], v_lo + tiles[0][0], tiles[0][1] - tiles[0][0])]
        if n_tiles > 1 and tiles[-1][1] > tiles[-1][0]:
            sources.append((blocks[g_last][-1], v_lo + tiles[-1][0], tiles[-1][1] - tiles[-1][0]))
        if pg:
            r = world - 1
            r_lo = sharding.plan_tiles(r, world, V_all, pitch, max(tile_bytes, pitch), strong=(scaling == "strong"))["v_lo"]
            nb = tile_sizes(n_tiles - 1)[r] // res_bytes
            if nb > 0:
                sources.append((recv[g_last][n_tiles - 1][r], r_lo + (n_tiles - 1) * per_tile, nb))
        ok, checked = True, 0
        fisher_terms = []

        def same(got, exp):
            with np.errstate(invalid="ignore"):
                return bool(np.all((np.abs(got - exp) <= 1e-10 * np.maximum(1, np.abs(exp))) | (np.isnan(got) & np.isnan(exp))))
        ncol = N if kind != "tdt" else 3 * (N // 3)
        for blk, vbase, n0 in sources:
            idx = np.unique(np.concatenate([np.arange(min(256, n0)), np.arange(0, n0, 1000), np.arange(max(0, n0 - 256), n0)]))
            if len(sources) > 1 or N > 20_000:
                idx = idx[:: max(1, len(idx) // 600)]
            if kind in ("fisher", "stats"):
                idx = idx[:: max(1, len(idx) // 400)]
            checked += len(idx)
            ints = blk[: head * n0].view(torch.int32).view(n0, head // 4).cpu().numpy()
            stats = blk[head * n0: res_bytes * n0].view(torch.float64).view(-1, n0).cpu().numpy()
            for lo in range(0, len(idx), 256):
                sel = idx[lo: lo + 256]
                rows = np.stack([orc.synth_matrix(vbase + int(v), 1, ncol, ncol)[0] for v in sel])
                if kind == "tdt":
                    t1, t2 = orc.tdt_counts(rows, *fam)
                    ok &= bool(np.array_equal(ints[sel], np.stack([t1, t2], 1)))
                    exp = orc.tdt_stats(t1, t2)
                    ok &= all(same(stats[j][sel], exp[j]) for j in range(3))
                elif kind == "stats":
                    for j, v in enumerate(sel):
                        vs = orc.variant_stats(rows[j], 2)
                        exp8 = list(vs.genotypes_count)[:4] + [vs.missing_genotypes, vs.missing_alleles,
                                                               vs.alleles_count[0], vs.alleles_count[1]]
                        ok &= list(ints[v]) == exp8
                        ok &= same(stats[0][v: v + 1], np.array([vs.hw_chi2])) and same(stats[1][v: v + 1], np.array([vs.hw_p]))
                else:
                    A1, A2, U1, U2 = orc.assoc_counts(rows, cond)
                    ok &= bool(np.array_equal(ints[sel], np.stack([A1, A2, U1, U2], 1)))
                    if kind == "chisq":
                        exp = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
                        ok &= all(same(stats[j][sel], exp[j]) for j in range(3))
                    else:
                        odds, _, p = orc.assoc_stats(orc.TASK_FISHER, A1, A2, U1, U2, lf_table)
                        ok &= same(stats[0][sel], odds) and same(stats[1][sel], p)
                        fisher_terms.append(orc.fisher_terms_needed(A1, A2, U1, U2, lf_table, 1e-22))
        parity = {"checked_variants": int(checked), "blocks": len(sources), "ok": bool(ok)}
        fisher_terms_mean = float(np.mean(np.concatenate(fisher_terms))) if fisher_terms else None
        fisher_terms_n = int(sum(len(t) for t in fisher_terms))

    if rank == 0:
        total_variants = total_per_step * steps
        par = "variant-sharded x%d (%s scaling)" % (world, scaling)
        if pg:
            par += ", result blocks gathered to rank 0 (%s), each step's gather overlapped with the next step's scan" % args.backend
        config = {"workload": "%s: %s" % (wl, desc), "variants": total_per_step, "variants_per_gpu": V, "samples": N,
                  ("affected" if kind in ("chisq", "fisher") else "trios" if kind == "tdt" else "columns"): nA,
                  ("unaffected" if kind in ("chisq", "fisher") else "multi_child_families" if kind == "tdt" else "groups"): nU,
                  "row_pitch_bytes": pitch, "tiles_per_gpu": n_tiles, "variants_per_tile": per_tile,
                  "resident": bool(resident), "hbm_total_GB": round(total_b / 1e9, 1),
                  "timed_region": ("one region over all steps, barrier + device sync on both sides" if resident else
                                   "sum over steps and tiles of [scan + statistics%s of one tile], each segment between device syncs%s; "
                                   "the tile buffer (%.1f GB) is regenerated on the device before its segment, outside the timed region"
                                   % (" + gather" if pg else "", " and barriers" if pg else "", per_tile * pitch / 1e9)),
                  "parallelism": par}
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": scan_name, "kernel_ms": scan_ms, "kernel_samples": len(evs),
                "read_probe_GBps": probe_gbps,
                "read_probe_note": "a plain (unpipelined) streaming-read kernel over the same buffer: informational, NOT a ceiling",
                "algorithmic_bytes_per_variant": bytes_per_variant,
                "variants_per_launch": scan_variants}
        roof["measured_in_this_run"] = ["achieved", "frac", "kernel_ms", "kernel_samples", "read_probe_GBps"]
        roof["traffic_source"] = None
        prof = pmc_profile(wl, scan_name, scan_variants, N, pitch) if world == 1 else None
        if prof and "hbm_bytes_per_launch" in prof:
            roof["traffic"] = prof["hbm_bytes_per_launch"]
            roof["traffic_source"] = REPLAYED % prof["_source"]["hbm_bytes_per_launch"]
        out = {
            "metric": METRIC[kind], "value": total_variants / elapsed, "unit": "variants/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u8",
            "data": "synthetic (on-device splitmix64 cohort, HWE genotypes, 1% missing, odd samples are cases)",
            "config": config,
        }
        if pg:
            out["rccl_ranks"] = dist.get_world_size() if args.backend == "nccl" else 0
        if kind == "fisher":
            # the p-pass dominates this workload and is bound by FP64 / 64-bit vector issue, not by HBM: instructions per
            # variant come from the committed PMC pass (tools/fisher_prof.sh), the duration from this run's HIP events
            fprof = pmc_profile(wl, "k_assoc_fisher", scan_variants, N, pitch) or {}
            ipv = fprof.get("valu_insts_per_variant")
            ginst = (ipv * scan_variants / (stats_ms * 1e-3) / 1e9) if ipv else None
            # The ALGORITHMIC work of the p-pass, counted here on the parity sample by the oracle: the hypergeometric terms each
            # variant's p-value needs under the kernel's own tail cut (a tail stops below 10^-fisher_cut_exp of its largest
            # term) x the cheapest per-term sequence the definition allows on this ISA (FISHER_INSTS_PER_TERM), 64 terms per
            # wave instruction.  achieved / frac are priced on THAT: instructions the kernel spends beyond it (boundary
            # searches, partly filled turns, per-variant setup) lower the fraction.  The old figure -- the kernel's own
            # instruction stream over the issue peak -- stays as issue_utilisation.
            alg_ipv = (fisher_terms_mean * FISHER_INSTS_PER_TERM / 64.0) if fisher_terms_mean else None
            alg_ginst = (alg_ipv * scan_variants / (stats_ms * 1e-3) / 1e9) if alg_ipv else None
            src = fprof.get("_source", {})
            out["roofline"] = {"bound": "valu", "achieved": alg_ginst, "peak": VALU64_PEAK_GINST, "unit": "Ginst/s",
                               "frac": (alg_ginst / VALU64_PEAK_GINST) if alg_ginst else None,
                               "frac_algorithmic": (alg_ginst / VALU64_PEAK_GINST) if alg_ginst else None,
                               "algorithmic_insts_per_variant": alg_ipv,
                               "algorithmic_terms_per_variant": fisher_terms_mean,
                               "algorithmic_insts_per_term": FISHER_INSTS_PER_TERM,
                               "algorithmic_note": ("terms: oracle count on this run's parity sample (%d variants) of the tables with P <= P_obs(1+1e-7) above "
                                                    "10^-%d of their tail's largest term; per term 2 address ops + 4 f64 subtractions + 18 for exp "
                                                    "(range reduction 4, cvt/and/shift 3, polynomial 6, scale 3, underflow select 2) + 1 add, "
                                                    "all priced at the 4-cycle 64-bit rate (profiles/r03_valu_instruction_costs.txt)"
                                                    % (fisher_terms_n or 0, 22)),
                               "issue_utilisation": (ginst / VALU64_PEAK_GINST) if ginst else None,
                               "issue_utilisation_note": "the kernel's OWN vector instructions per second over the issue peak: says the SIMDs are busy, not that the work is minimal",
                               "traffic": fprof.get("hbm_bytes_per_launch"),
                               "traffic_source": (REPLAYED % src["hbm_bytes_per_launch"]) if "hbm_bytes_per_launch" in src else None,
                               "kernel": "k_assoc_fisher", "kernel_ms": stats_ms, "kernel_samples": len(evs),
                               "valu_insts_per_variant": ipv,
                               "valu_insts_source": (REPLAYED % src["valu_insts_per_variant"]) if "valu_insts_per_variant" in src else None,
                               "measured_in_this_run": ["kernel_ms", "kernel_samples", "algorithmic_terms_per_variant", "achieved", "frac", "frac_algorithmic"],
                               "variants_per_launch": scan_variants,
                               "peak_note": "wave64 FP64 / 64-bit vector instructions: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles",
                               "overlap_note": ("value: the tile goes through in %d pieces, a piece's Fisher pass on a second stream beside the next "
                                                "piece's scan; kernel_ms: the launches that carry events (every %d-th of the timed region) run the two "
                                                "kernels one after the other, alone" % (args.fisher_pieces, every)) if args.fisher_pieces > 1 else None}
            out["roofline_scan"] = roof
        else:
            out["roofline"] = roof
            out["roofline"]["stats_kernel"] = stats_name
            out["roofline"]["stats_kernel_ms"] = stats_ms
        out["parity"] = parity
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, N if kind != "tdt" else 3 * (N // 3), cond, fam, lf_table, cpu_seconds)
        if parity is not None and not parity["ok"]:
            print("bench.py: workload %s: the timed run's outputs DIFFER from the oracle on the sampled variants: its number is invalid" % wl,
                  file=sys.stderr)
            parity_failed = True
    eng.close()
    del gt, blocks, recv
    torch.cuda.empty_cache()
    return out, parity_failed


def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; running with %d rank(s)" % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; this engine has no CPU path", file=sys.stderr)
        return 2
    if local_rank >= torch.cuda.device_count() and args.backend == "nccl":      # before set_device, which would raise
        print("bench.py: rank %d wants GPU %d but this machine shows %d GPU(s): one rank per GPU over RCCL"
              % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
        return 2
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = world > 1 or args.force_process_group
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:             # a forced one-rank process group: any free port (two runs side by side)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    out, failed = measure(args, args.workload, args.steps, args.warmup, args.cpu_seconds, True, world, rank, dev, dev_index, pg)
    # ---- the other BASELINE configs in the same driver-timed line (N = 1, default workload, no overrides): a few seconds each
    want_configs = args.configs == "all" or (args.configs == "auto" and args.workload == "m" and world == 1
                                              and not args.variants and not args.samples and not args.option)
    if rank == 0 and world == 1 and want_configs and out is not None:
        out["configs"] = {}
        for name in ("c2", "c3", "c4", "stats"):
            if name == args.workload:
                continue
            sub, f = measure(args, name, args.config_steps, args.config_warmup, args.config_cpu_seconds, False, world, rank, dev, dev_index, False)
            failed = failed or f
            keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling", "dtype", "roofline", "roofline_scan",
                    "parity", "cpu_baseline")
            out["configs"][name] = dict({k: sub[k] for k in keep if k in sub},
                                        config={k: sub["config"][k] for k in ("workload", "variants", "samples", "row_pitch_bytes")})
    if rank == 0 and out is not None:
        print(json.dumps(out))
        sys.stdout.flush()
    if pg:
        dist.barrier()
        dist.destroy_process_group()
    return 3 if failed else 0


def worker_group(args):
    """--single-process: the north star's shards + gather behind the C ABI, one process, no torch.distributed."""
    import numpy as np
    hpgv = importlib.import_module("hpg-variant_amd")
    kind, V_all, N, scaling, desc = WORKLOADS[args.workload]
    V_all = args.variants or V_all
    N = args.samples or N
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    G = len(devices)
    V = V_all if scaling == "strong" else V_all * G
    try:
        grp = hpgv.Engine(devices)
    except hpgv.HpgvError as e:
        print("bench.py: %s" % e, file=sys.stderr)
        return 2
    for kv in args.option:
        k, v = kv.split("=")
        grp.set_option(k, int(v))
    cond = (np.arange(N) % 2).astype(np.uint8)
    fam = lf_table = None
    if kind == "tdt":
        n_tr = N // 3
        kk = np.arange(n_tr)
        fam = (3 * kk, 3 * kk + 1, np.arange(n_tr + 1), 3 * kk + 2, (kk % 2).astype(np.uint8))
        nA, nU, pitch = grp.set_families(3 * n_tr, *fam)
        which = hpgv.LAYOUT_TDT
    elif kind == "stats":
        pitch = grp.set_stats_cohort(N)
        nA, nU, which = N, 0, hpgv.LAYOUT_STATS
    else:
        nA, nU, pitch = grp.set_cohort(cond)
        which = hpgv.LAYOUT_ASSOC
        if kind == "fisher":
            lf_table = np.concatenate([[0.0], np.cumsum(np.log(np.arange(1, N * 10, dtype=np.float64)))])
            grp.set_logfact(lf_table)
    head, res_bytes = RESULT[kind]
    ranks = grp.group_comm_init()
    members = [grp.member(k) for k in range(G)]
    shards = []
    try:
        for k, m in enumerate(members):
            lo, hi = grp.group_shard(V, k)
            p = m.alloc(max(hi - lo, 1) * pitch)
            for v0 in range(lo, hi, 1 << 24):                  # the generator takes an int count
                m.synth(which, v0, min(1 << 24, hi - v0), p.value + (v0 - lo) * pitch)
            shards.append(p)
        for m in members:
            m.sync()
        m0 = members[0]
        # two sets of result arrays: the gather of step k runs under the scans of step k + 1
        sets = [[m0.alloc(head * max(V, 1))] + [m0.alloc(8 * max(V, 1)) for _ in range(3)] for _ in range(2)]
    except hpgv.HpgvError as e:
        print("bench.py: the shards of this workload do not fit %d device(s) resident: %s" % (G, e), file=sys.stderr)
        return 2

    def step(k):
        ints, f0, f1, f2 = sets[k & 1]
        if kind == "chisq":
            grp.group_assoc(hpgv.TASK_CHISQ, shards, V, ints, f0, f1, f2)
        elif kind == "fisher":
            grp.group_assoc(hpgv.TASK_FISHER, shards, V, ints, f0, None, f2)
        elif kind == "tdt":
            grp.group_tdt(shards, V, ints, f0, f1, f2)
        else:
            grp.group_stats(shards, V, ints, f1, f2, None)

    for k in range(args.warmup):
        step(k)
    grp.group_sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    grp.group_sync()
    elapsed = time.perf_counter() - t0

    # the scan kernel alone on member 0's shard (HIP events on its stream, option "profile"), for the roofline line
    lo0, hi0 = grp.group_shard(V, 0)
    m0.set_option("profile", 1)
    tmp = m0.alloc(head * max(hi0 - lo0, 1))
    scan_ms = []
    for _ in range(3):
        if kind == "tdt":
            m0.tdt_scan(shards[0], hi0 - lo0, tmp, None, None)
        elif kind == "stats":
            m0.stats_scan(shards[0], hi0 - lo0, tmp, None)
        else:
            m0.assoc_scan(shards[0], hi0 - lo0, tmp, None, None)
        scan_ms.append(m0.last_kernel_ms()[0])
    m0.set_option("profile", 0)
    scan_ms = float(np.median(scan_ms))
    achieved = (hi0 - lo0) * (N + res_bytes) / (scan_ms * 1e-3) / 1e9

    # parity: the GATHERED arrays on member 0 against the oracle, sampled over every member's shard
    from oracle import pyoracle as orc
    last = sets[(args.warmup + args.steps - 1) & 1]
    idx = []
    for k in range(G):
        lo, hi = grp.group_shard(V, k)
        n = hi - lo
        if n:
            idx.append(lo + np.unique(np.concatenate([np.arange(min(64, n)), np.arange(0, n, max(1000, n // 100)), np.arange(max(0, n - 64), n)])))
    idx = np.concatenate(idx) if idx else np.zeros(0, np.int64)
    if kind in ("fisher", "stats"):
        idx = idx[:: max(1, len(idx) // 400)]
    ints = m0.d2h(last[0], (V, head // 4), np.int32)
    fl = [m0.d2h(last[i], (V,), np.float64) for i in (1, 2, 3)]
    ncol = N if kind != "tdt" else 3 * (N // 3)

    def same(got, exp):
        with np.errstate(invalid="ignore"):
            return bool(np.all((np.abs(got - exp) <= 1e-10 * np.maximum(1, np.abs(exp))) | (np.isnan(got) & np.isnan(exp))))
    ok = True
    for a in range(0, len(idx), 256):
        sel = idx[a: a + 256]
        rows = np.stack([orc.synth_matrix(int(v), 1, ncol, ncol)[0] for v in sel])
        if kind == "tdt":
            t1, t2 = orc.tdt_counts(rows, *fam)
            exp = orc.tdt_stats(t1, t2)
            ok &= bool(np.array_equal(ints[sel], np.stack([t1, t2], 1))) and all(same(fl[j][sel], exp[j]) for j in range(3))
        elif kind == "stats":
            for j, v in enumerate(sel):
                vs = orc.variant_stats(rows[j], 2)
                exp8 = list(vs.genotypes_count)[:4] + [vs.missing_genotypes, vs.missing_alleles, vs.alleles_count[0], vs.alleles_count[1]]
                ok &= list(ints[v]) == exp8 and same(fl[1][v: v + 1], np.array([vs.hw_chi2])) and same(fl[2][v: v + 1], np.array([vs.hw_p]))
        else:
            A1, A2, U1, U2 = orc.assoc_counts(rows, cond)
            ok &= bool(np.array_equal(ints[sel], np.stack([A1, A2, U1, U2], 1)))
            if kind == "chisq":
                exp = orc.assoc_stats(orc.TASK_CHISQ, A1, A2, U1, U2)
                ok &= all(same(fl[j][sel], exp[j]) for j in range(3))
            else:
                odds, _, p = orc.assoc_stats(orc.TASK_FISHER, A1, A2, U1, U2, lf_table)
                ok &= same(fl[0][sel], odds) and same(fl[2][sel], p)
    out = {
        "metric": METRIC[kind], "value": V * args.steps / elapsed, "unit": "variants/s", "n_gpus": G, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "u8",
        "data": "synthetic (on-device splitmix64 cohort, HWE genotypes, 1% missing, odd samples are cases)",
        "config": {"workload": "%s: %s" % (args.workload, desc), "variants": V, "variants_per_gpu": -(-V // G), "samples": N,
                   "row_pitch_bytes": pitch, "resident": True, "devices": devices,
                   "timed_region": "one region over all steps between two hpgv_group_sync",
                   "parallelism": "ONE process, group context over %d device(s) behind the C ABI (hpgv_group_*): variant shards on "
                                  "per-device streams, results gathered onto device 0 over the group's own communicator "
                                  "(ncclCommInitAll), each step's gather under the next step's scans" % G},
        "rccl_ranks": ranks,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "traffic_source": None, "measured_in_this_run": ["achieved", "frac", "kernel_ms"],
                     "kernel": "scan kernel of member 0's shard, alone", "kernel_ms": scan_ms,
                     "algorithmic_bytes_per_variant": N + res_bytes, "variants_per_launch": hi0 - lo0},
        "parity": {"checked_variants": int(len(idx)), "ok": bool(ok), "what": "arrays gathered on member 0 vs the oracle, sampled over every member's shard"},
    }
    print(json.dumps(out))
    sys.stdout.flush()
    for m in members:
        m.close()
    grp.close()
    if not ok:
        print("bench.py: the gathered results DIFFER from the oracle on the sampled variants: the number above is invalid", file=sys.stderr)
        return 3
    return 0


def main():
    args = parse_args()
    if args.single_process:
        sys.exit(worker_group(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
