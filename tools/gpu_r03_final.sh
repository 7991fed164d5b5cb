#!/bin/bash
# Round-3 evidence in one gpurun call: GPU tests, PMC passes of the four bench workloads (-> profiles/pmc_traffic.json), the driver-contract
# bench lines, the rocprofv3 kernel trace of the default bench command, the all-kernels and decoder tables, a long random search.
# Everything lands in gpurun_out/r03/; tools/copy_evidence_r03.sh copies what is kept to profiles/.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
bash tools/pmc_collect.sh r03 > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_r03/summary.txt $OUT/pmc_summary.txt; rm -f gpurun_out/pmc_r03/*.db
python tools/pmc_to_json.py uyvy_dxt5_4k_x16 "dxt_encode_kernel<2, 6" "rocprof passes of round 3 (profiles/r03_pmc_uyvy_dxt5_4k_x16.txt), dxt_encode_kernel<UYVY,DXT5,ties even> with the fast index stages" $OUT/pmc_summary.txt
bash tools/pmc_workloads.sh > $OUT/pmc_workloads.log 2>&1; cp gpurun_out/pmc_workloads/*.txt $OUT/
python tools/pmc_to_json.py v210_dxt5_8k_x4 "dxt_encode_kernel<6, 6" "rocprof passes of round 3 (profiles/r03_pmc_8k_v210.txt), dxt_encode_kernel<v210,DXT5,ties even>" $OUT/8k-v210.txt
python tools/pmc_to_json.py rgb_dxt1_1080p_x64 "dxt_encode_kernel<4, 1" "rocprof passes of round 3 (profiles/r03_pmc_1080p_rgb_dxt1.txt), dxt_encode_kernel<RGB,DXT1,ties even>" $OUT/1080p-rgb-dxt1.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench_line.json
for wl in 8k-v210 1080p-rgb-dxt1 4k-uyvy-jpeg420; do python bench.py --workload $wl --no-e2e > $OUT/bench_$wl.json 2>> $OUT/bench.err; done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-e2e > $OUT/trace.log 2>&1 )
python tools/pmc_summary.py $OUT/trace/bench_results.db > $OUT/kernel_trace.txt 2>&1; head -4 $OUT/kernel_trace.txt | cut -c1-160; tail -1 $OUT/trace.log | cut -c1-300
rm -rf $OUT/trace
python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; grep -c . $OUT/kernels_table.txt
python tools/bench_decode.py --json $OUT/decode.json > $OUT/decode.txt 2>&1
timeout 900 python tools/find_dxt_mismatch.py 8000 2>&1 | tail -3 > $OUT/find_dxt.txt; cat $OUT/find_dxt.txt
timeout 300 python tools/find_module_mismatch.py 2>&1 | tail -2 > $OUT/find_module.txt; cat $OUT/find_module.txt
ls $OUT
