cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/epm_clk -o clk --output-format csv -- python3 $R/tools/bench_epistasis.py 16384 10000 10 > $O/epm_clk.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/epm_clk0 -o clk --output-format csv -- python3 $R/tools/bench_epistasis.py 16384 10000 10 --option=epi_pairs_mfma=0 > $O/epm_clk0.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
for d in ("epm_clk","epm_clk0"):
    dur={}
    for f in glob.glob("$O/%s/**/*kernel_trace.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_epi_pairs" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    tot_c=0;tot_t=0
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_epi_pairs" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                tot_c+=float(r["Counter_Value"]); tot_t+=dur[r["Dispatch_Id"]]
    print(d,"cycles",tot_c,"ns",tot_t,"GHz",tot_c/max(tot_t,1))
PY
