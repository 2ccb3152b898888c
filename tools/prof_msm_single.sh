timeout 400 python -m pytest tests/test_gpu_bn254_curve.py tests/test_gpu_halo2.py -x -q 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/msm_stats -- python $GRAFT_REPO_ROOT/tools/prof_fr_fft.py 23 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/msm_stats/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in list(csv.DictReader(open(f))):
    if "msm_" in r["Name"]:
        tot+=float(r["TotalDurationNs"])/2
        print(r["Name"][:50], r["Calls"], round(float(r["AverageNs"])/1e6,3), round(float(r["TotalDurationNs"])/2e6,2))
print("per MSM kernel ms", tot/1e6)
PY
find gpurun_out/msm_stats -name "*kernel_trace.csv" -delete
