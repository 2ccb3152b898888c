# Halo2 k = 23 profile set (run through gpurun): rocprofv3 kernel stats of tools/halo2_bench.py 23, then --pmc passes (separate runs) with the
# SQ counters that tell a VALU-bound kernel from a stalled one.  Summaries land in gpurun_out/prof_h2/; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_h2
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/halo2_bench.py 23 > $O/halo2_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT64 --output-format csv -d $O/pmc_sq -- python tools/halo2_bench.py 23 > /dev/null 2> $O/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_lds -- python tools/halo2_bench.py 23 > /dev/null 2> $O/pmc_lds.err
python - <<'PY'
import csv, glob, os, collections, shutil
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/prof_h2")
for d in ("pmc_sq","pmc_lds"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(os.path.join(O,d,"**","*counter_collection.csv"),recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]; agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    with open(os.path.join(O,d+"_summary.txt"),"w") as out:
        for k in sorted(agg, key=lambda k:-sum(agg[k].values())):
            out.write(k[:90]+": "+", ".join("%s=%.4g (n=%d)"%(c,v/cnt[(k,c)],cnt[(k,c)]) for c,v in agg[k].items())+"\n")
    shutil.rmtree(os.path.join(O,d),ignore_errors=True)
for f in glob.glob(os.path.join(O,"stats","**","*kernel_trace.csv"),recursive=True): os.remove(f)
for f in glob.glob(os.path.join(O,"stats","**","*_agent_info.csv"),recursive=True): os.remove(f)
PY
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -12 $O/kernel_stats.csv | cut -c1-150
head -12 $O/pmc_sq_summary.txt | cut -c1-400
head -8 $O/pmc_lds_summary.txt | cut -c1-400
tail -3 $O/pmc_lds.err
