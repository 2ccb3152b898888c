# contexts:lock-step-units:replay-threads sweep of the default bench workload (run on the GPU box); WAIT=poll|spin|sleep
mkdir -p gpurun_out
for cfg in ${SWEEP:-22:1:2 16:4:2 8:4:2 12:4:2 6:8:2 8:8:2 12:8:2 4:16:2 6:16:2}; do
  c=${cfg%%:*}; r=${cfg#*:}; b=${r%%:*}; t=${r#*:}
  GL355_BENCH_WAIT=${WAIT:-poll} GL355_BENCH_REPLAY_THREADS=$t GL355_BENCH_BATCH_UNITS=$b python bench.py --threads $c --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - "$c" "$b" "$t" "${WAIT:-poll}" <<'P'
import json,sys
try:
    d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1])
    print("wait %s contexts %s B %s replay-threads %s: %.1f units/s, host_cpu_ms_per_unit %s" % (sys.argv[4], sys.argv[1], sys.argv[2], sys.argv[3], d['value'], d['config'].get('host_cpu_ms_per_unit')), flush=True)
except Exception as e:
    print("contexts %s B %s: FAILED %r" % (sys.argv[1], sys.argv[2], e)); print(open('gpurun_out/sw.err').read()[-800:])
P
done
