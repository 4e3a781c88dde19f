#!/bin/bash
# The scaling curve of BASELINE.json's metric in ONE command, for the day a node with several MI355X is at hand:
#
#   tools/scale_node.sh [outdir]              # N in 1 2 4 8 (as many GPUs as the node shows), real RCCL
#   tools/scale_node.sh --rehearse [outdir]   # the same commands on ONE GPU: gloo ranks / group members sharing device 0
#                                             # (checks every command line and code path; its numbers mean nothing)
#
# Per N: (a) bench.py --gpus N            one process per GPU, torch.distributed over RCCL (what the driver launches);
#        (b) bench.py --gpus N --single-process   ONE process, the C ABI's group context (hpgv_group_*, ncclCommInitAll);
# then   (c) the whole configs[4] cohort (40M x 100k) on all GPUs, (d) the file runner over a bgzip VCF with the devices of
#        the node behind libhpgv_host.so (HPGV_DEVICES=all).  Every JSON line is kept as it is printed; scale_summary.json
#        lists metric value, n_gpus, rccl_ranks, roofline.frac and the scaling efficiency against N = 1 of each form.
# The summaries worth keeping go to profiles/scale_<date>_*.json by hand (gpurun_out/ is scratch).
set -u
REHEARSE=0
if [ "${1:-}" = "--rehearse" ]; then REHEARSE=1; shift; fi
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/scale_$(date +%Y%m%d_%H%M%S)}"
mkdir -p "$OUT"
cd "$ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python3 -c "import bench; print(bench.count_gpus_no_hip())" 2>/dev/null || echo 1)
[ "$HAVE" -ge 1 ] 2>/dev/null || HAVE=1
echo "scale_node: $HAVE GPU(s) visible, rehearse=$REHEARSE, output in $OUT"
NS="1 2 4 8"
run() {  # name, command...
    local name="$1"; shift
    echo "== $name: $*"
    timeout -k 10 1500 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
    local rc=$?
    echo "   rc=$rc $(head -c 220 "$OUT/$name.json")"
    return 0
}
for N in $NS; do
    if [ "$REHEARSE" = 1 ]; then
        [ "$N" -gt 4 ] && continue                                   # a GPU box allows few processes on its card
        SMALL="--variants 400000 --samples 10000 --steps 3 --warmup 1 --no-cpu-baseline --configs none"
        if [ "$N" = 1 ]; then run "ranks_n$N" python3 bench.py --gpus 1 $SMALL
        else run "ranks_n$N" python3 bench.py --gpus "$N" --backend gloo $SMALL; fi
        DEV=$(python3 -c "print(','.join(['0'] * $N))")
        run "group_n$N" python3 bench.py --gpus "$N" --single-process --devices "$DEV" $SMALL
    else
        [ "$N" -gt "$HAVE" ] && { echo "== N=$N skipped: the node shows $HAVE GPU(s)"; continue; }
        run "ranks_n$N" python3 bench.py --gpus "$N"
        run "group_n$N" python3 bench.py --gpus "$N" --single-process --no-cpu-baseline
    fi
done
if [ "$REHEARSE" = 1 ]; then
    run "c5full_rehearsal" python3 bench.py --workload c5full --gpus 2 --backend gloo --variants 200000 --samples 100000 --steps 1 --warmup 0 --no-cpu-baseline
    HPGV_DEVICES=0,0 run "file_runner_devices" python3 tools/bench_file_runner.py 2000 20000 bgzf 64
else
    [ "$HAVE" -ge 2 ] && run "c5full_n$HAVE" python3 bench.py --workload c5full --gpus "$HAVE" --steps 1 --warmup 0
    HPGV_DEVICES=all run "file_runner_devices" python3 tools/bench_file_runner.py 10000 200000 bgzf 256
fi
python3 - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    if os.path.basename(f) == "scale_summary.json":
        continue
    for line in open(f):
        line = line.strip()
        if not line.startswith("{"):
            continue
        try:
            d = json.loads(line)
        except ValueError:
            continue
        if "metric" in d:
            rows.append({"run": os.path.basename(f)[:-5], "metric": d["metric"], "value": d["value"], "unit": d.get("unit"), "n_gpus": d.get("n_gpus"),
                         "ms_per_step": d.get("ms_per_step"), "scaling": d.get("scaling"), "rccl_ranks": d.get("rccl_ranks"),
                         "roofline_frac": (d.get("roofline") or {}).get("frac"), "parity_ok": (d.get("parity") or {}).get("ok"),
                         "workload": (d.get("config") or {}).get("workload")})
base = {}
for r in rows:
    form = r["run"].split("_n")[0]
    if r["n_gpus"] == 1:
        base[form] = r["value"]
for r in rows:
    form = r["run"].split("_n")[0]
    if form in base and r["n_gpus"]:
        r["efficiency_vs_n1"] = r["value"] / (base[form] * r["n_gpus"])
json.dump(rows, open(os.path.join(out, "scale_summary.json"), "w"), indent=1)
for r in rows:
    print("%-22s n=%s value=%.4g %s frac=%s ranks=%s eff=%s parity=%s" % (r["run"], r["n_gpus"], r["value"], r["unit"], r["roofline_frac"], r["rccl_ranks"],
                                                                   r.get("efficiency_vs_n1"), r["parity_ok"]))
PY
echo "scale_node: done, $OUT/scale_summary.json"
