#!/bin/bash
# Round 4, session M: the whole GPU suite, NUMA soak A/B, traffic counters of the JPEG encoder's kernels, the tables re-taken over 2.4 GB
# of rotating buffers, the traffic entries of pmc_traffic.json, the bench line.  Everything lands in gpurun_out/r04m/.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04m; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
CFGS="dxt:DXT5 dxt:DXT5:numa=0 jpeg:q=75:restart=4 jpeg:q=75:restart=4:numa=0 dxt:DXT5 dxt:DXT5:numa=0 jpeg:q=75:restart=4:batch=8 jpeg:q=75:restart=4:batch=8:workers=1" REPEAT=400 timeout 600 bash tools/soak.sh > $OUT/soak.txt 2>&1; grep -E "^==|THROUGHPUT" $OUT/soak.txt | paste - - | cut -c1-150
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/tools/bench_jpeg_batch.py --only batch --calls 40"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt -o t -- $CMD > $OUT/jt.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/jp1 -o p -- $CMD > $OUT/jp1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/jp3 -o p -- $CMD > $OUT/jp3.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/jp2 -o p -- $CMD > $OUT/jp2.log 2>&1 )
python tools/pmc_summary.py $(find $OUT/jt $OUT/jp1 $OUT/jp2 $OUT/jp3 -name "*.db") 2>&1 | grep -v "copyBuffer\|roll_cuda\|elementwise\|fillBuffer\|CatArray\|at::native" | sed "s#$ROOT/##" > $OUT/jpeg_batch_traffic.txt
grep "^pmc" $OUT/jpeg_batch_traffic.txt | cut -c1-12,60-150; rm -rf $OUT/jt $OUT/jp1 $OUT/jp2 $OUT/jp3
bash tools/pmc_workloads.sh > $OUT/pmc_workloads.log 2>&1; cp gpurun_out/pmc_workloads/*.txt $OUT/
python tools/pmc_to_json.py uyvy_jpeg420_4k_x8 "uyvy_jpeg_fast_kernel" "rocprof passes of round 4 (profiles/r04_pmc_4k_uyvy_jpeg420.txt), uyvy_jpeg_fast_kernel<420> batched (the configs[3] front end; the -c jpeg module's fused encoder kernel never writes the coefficients)" $OUT/4k-uyvy-jpeg420.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_line.json
python bench.py --workload 4k-uyvy-jpeg420 --no-e2e > $OUT/bench_4k-uyvy-jpeg420.json 2>> $OUT/bench.err
timeout 900 python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; grep -c . $OUT/kernels_table.txt; tail -3 $OUT/kernels_table.txt
timeout 600 python tools/bench_pixfmt_all.py --json $OUT/pixfmt_all_8k.json > $OUT/pixfmt_all_8k.txt 2>&1; tail -3 $OUT/pixfmt_all_8k.txt
timeout 300 python tools/bench_decode.py --json $OUT/decode.json > $OUT/decode.txt 2>&1; tail -3 $OUT/decode.txt
ls $OUT
