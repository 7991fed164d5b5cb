#!/bin/bash
# r03j: v210 encoder stores back to plain (write amplification check) -- PMC passes of the 8K v210 and 1080p RGB workloads, their bench lines
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
python -m pytest tests/test_gpu_dxt.py -q -x 2>&1 | tail -1
bash tools/pmc_workloads.sh > $OUT/pmc_workloads.log 2>&1; cp gpurun_out/pmc_workloads/*.txt $OUT/
python tools/pmc_to_json.py v210_dxt5_8k_x4 "dxt_encode_kernel<6, 6" "rocprof passes of round 3 (profiles/r03_pmc_8k_v210.txt), dxt_encode_kernel<v210,DXT5,ties even>" $OUT/8k-v210.txt
python tools/pmc_to_json.py rgb_dxt1_1080p_x64 "dxt_encode_kernel<4, 1" "rocprof passes of round 3 (profiles/r03_pmc_1080p_rgb_dxt1.txt), dxt_encode_kernel<RGB,DXT1,ties even>" $OUT/1080p-rgb-dxt1.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
for wl in 8k-v210 1080p-rgb-dxt1; do python bench.py --workload $wl --no-e2e > $OUT/bench_$wl.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_$wl.json; done
