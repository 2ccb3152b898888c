# stall picture of the LDE passes: a few --pmc passes over one LDE shape (2^17 -> 2^20 x 135)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/pl; rocprofv3 --pmc $set --output-format csv -d /tmp/pl -- python $R/tools/prof_ntt.py > /dev/null 2> /tmp/pl.err || { echo "set [$set] failed: $(tail -1 /tmp/pl.err)"; continue; }
  python - <<'P'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("/tmp/pl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("gl355::","")
        if not k.startswith("ntt_"): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in agg:
    print(k[:40], {c: "%.4g" % (v/cnt[(k,c)]) for c,v in agg[k].items()})
P
done
