# Round-4 closing set (GPU box): full GPU suite, smoke, the default bench line, the Halo2 prover at k = 17 / 20 / 23 with its kernel profile, the
# aggregation flow.  Outputs under gpurun_out/final_r04/; the summaries are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r04
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py > $O/bench_recursive.json 2> $O/bench_recursive.err
python tools/halo2_bench.py 17 20 23 > $O/halo2_bench.json 2> $O/halo2_bench.err
python tools/aggregate_native.py 8 > $O/aggregate_native.json 2> $O/aggregate_native.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/h2stats -- python tools/halo2_bench.py 23 > /dev/null 2> $O/h2stats.err
find $O/h2stats -name "*kernel_trace.csv" -delete
find $O/h2stats -name "*_agent_info.csv" -delete
tail -3 $O/gpu_tests.txt; cat $O/smoke.txt | tail -1
