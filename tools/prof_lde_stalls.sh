# Where the waves of the two LDE passes spend their cycles (round 6): two rocprofv3 --pmc passes over tools/prof_lde.py with the SQ wait / LDS counters.
# SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md); SQ_BUSY_CYCLES per shader engine.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/ldest_a -- python tools/prof_lde.py 12 > /dev/null 2> $O/ldest_a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $O/ldest_b -- python tools/prof_lde.py 12 > /dev/null 2> $O/ldest_b.err
python - <<'PY'
import csv, glob, os, collections
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/prof_r06")
with open(os.path.join(O,"lde_stalls.txt"),"w") as out:
    for d in ("ldest_a","ldest_b"):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for f in glob.glob(os.path.join(O,d,"**","*counter_collection.csv"),recursive=True):
            for row in csv.DictReader(open(f)):
                k=row["Kernel_Name"]
                if "ntt_" not in k: continue
                agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
        for k in agg:
            out.write(k[:80]+": "+", ".join("%s=%.5g"%(c,v/cnt[(k,c)]) for c,v in sorted(agg[k].items()))+"\n")
print(open(os.path.join(O,"lde_stalls.txt")).read())
PY
