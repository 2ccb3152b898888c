#!/bin/bash
# A/B of the 16-lane Poseidon permutation's partial rounds (profiles/r06_lanes_form3_ab.txt): the tree as built against a library with the dense form everywhere.
# Build the variant HERE first (the GPU box only runs it):
#   cd stark-verifier_amd/csrc && mkdir -p ../lib/variants && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPSD_LANES_FORM=0 -c merkle.hip -o /tmp/merkle_v0.o \
#     && hipcc -shared -o ../lib/variants/libgl355_lanesform0.so /tmp/merkle_v0.o $(ls build/*.o | grep -v /merkle.o) -ldl -pthread
#   (only merkle.hip reads PSD_LANES_FORM; link flags as the Makefile's $(LIB) rule)
# then: gpurun -- 'bash tools/ab_lanes_form.sh'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q -k "merkle or commit or tree or hash or poseidon" > gpurun_out/r06/lanes_tests.txt 2>&1; tail -2 gpurun_out/r06/lanes_tests.txt
rm -f gpurun_out/r06/ab_lanes3.txt
KBENCH_WARM=5 KBENCH_REPS_X=5 bash tools/ab_lib.sh gpurun_out/r06/ab_lanes3.txt python tools/kbench.py merkle
grep -v amdgpu gpurun_out/r06/ab_lanes3.txt | grep "==\|2^16\|2^20 L=4" | cut -c1-250
L=stark-verifier_amd/lib
cp $L/libgl355.so /tmp/main.so
for v in main lanesform0 main lanesform0; do
  if [ $v = main ]; then cp /tmp/main.so $L/libgl355.so; else cp $L/variants/libgl355_$v.so $L/libgl355.so; fi
  python bench.py --steps 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'units/s', d['value'], 'latency', d['latency_single_unit_ms']['median_ms'], d['latency_single_unit_ms']['min_ms'])"
done
cp /tmp/main.so $L/libgl355.so
