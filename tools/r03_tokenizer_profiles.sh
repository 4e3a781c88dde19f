#!/bin/bash
# round 3, the tokenizer after the shape recogniser: times, per-kernel durations, instruction counters and HBM traffic of
# tools/bench_tokenize.py on 16 000 x 10 k and 800 000 x 200 (text resident on the device).  Files under gpurun_out/r03/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
: > $O/tok_bench.jsonl
for shape in "10000 16000" "200 800000"; do
  for tiles in 1 2; do
    TOK_TILES=$tiles python3 $R/tools/bench_tokenize.py $shape | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); d['tokenizer_tiles']=$tiles; print(json.dumps(d))" >> $O/tok_bench.jsonl || exit 1
  done
done
cat $O/tok_bench.jsonl
HEAD=7 $R/tools/prof_kernels.sh tok_10k tools/bench_tokenize.py 10000 16000 || exit 1
HEAD=7 $R/tools/prof_kernels.sh tok_200 tools/bench_tokenize.py 200 800000 || exit 1
for tag in 10k 200; do
  [ $tag = 10k ] && shape="10000 16000" || shape="200 800000"
  i=0
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    HEAD=1 $R/tools/prof_pmc.sh tokpmc_${tag}_$i "$c" tools/bench_tokenize.py $shape > /dev/null || exit 1
  done
done
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
out = {}
for tag in ("10k", "200"):
    k = {}
    for path in sorted(glob.glob("%s/tokpmc_%s_*_pmc.csv" % (O, tag))):
        for row in csv.DictReader(open(path)):
            name = row["kernel"].replace("hpgv::", "")
            if "tok" not in name: continue
            for c, v in row.items():
                if c not in ("kernel", "dispatches"): k.setdefault(name, {})[c] = float(v)
    for name, d in k.items():
        if "FETCH_SIZE" in d: d["hbm_read_GB"] = round(d["FETCH_SIZE"] * 2048 / 1e9, 4)      # KiB, doubled on gfx950
        if "WRITE_SIZE" in d: d["hbm_write_GB"] = round(d["WRITE_SIZE"] * 1024 / 1e9, 4)
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d: d["valu_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
    out[tag] = k
json.dump(out, open(O + "/tokenizer_pmc.json", "w"), indent=1)
for tag in out:
    for name, d in out[tag].items():
        print(tag, name, {c: d[c] for c in ("valu_per_wave", "hbm_read_GB", "hbm_write_GB") if c in d})
PY
