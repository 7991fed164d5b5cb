#!/bin/bash
# Round 4: the JPEG part of tools/gpu_r04_final.sh again (after the encoder kernel's divisions went), plus the whole GPU suite; same output
# directory, so tools/copy_evidence_r04.sh picks the newer files up.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04z; mkdir -p $OUT
FILTER="copyBuffer\|roll_cuda\|elementwise\|fillBuffer\|CatArray\|at::native"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -2 $OUT/pytest.log
( cd /tmp && export TMPDIR=/tmp
  CMD="python $ROOT/tools/bench_jpeg_batch.py --only batch --calls 40"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt -o t -- $CMD > $OUT/jt.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/jp0 -o p -- $CMD > $OUT/jp0.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/jp1 -o p -- $CMD > $OUT/jp1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/jp3 -o p -- $CMD > $OUT/jp3.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/jp2 -o p -- $CMD > $OUT/jp2.log 2>&1
  for sub in 422 444; do timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/jt$sub -o t -- $CMD --sub $sub > $OUT/jt$sub.log 2>&1; done )
python tools/pmc_summary.py $(find $OUT/jt $OUT/jp0 $OUT/jp1 $OUT/jp2 $OUT/jp3 -name "*.db") 2>&1 | grep -v "$FILTER" | sed "s#$ROOT/##" > $OUT/jpeg_batch_pmc.txt
python tools/pmc_summary.py $(find $OUT/jt422 $OUT/jt444 -name "*.db") 2>&1 | grep -v "$FILTER" | sed "s#$ROOT/##" > $OUT/jpeg_batch_trace_422_444.txt
grep -A1 "^kernel.*jpeg" $OUT/jpeg_batch_pmc.txt | head -4 | cut -c1-140; grep "^pmc.*jpeg_code.*SQ_INSTS_VALU\|^pmc.*jpeg_code.*SQ_WAVES\|^pmc.*jpeg_code.*ACTIVE" $OUT/jpeg_batch_pmc.txt | cut -c60-150
rm -rf $OUT/jt $OUT/jp0 $OUT/jp1 $OUT/jp2 $OUT/jp3 $OUT/jt422 $OUT/jt444
python tools/pmc_to_json.py uyvy_jpeg_encode_4k_x8 "jpeg_code_kernel<3, 420>" "rocprof passes of round 4 (profiles/r04_jpeg_batch_pmc.txt), jpeg_code_kernel<3,420>: the fused encoder kernel of ug_hip_jpeg_encoder_encode_batch, 8 frames per launch (jpeg_gather_kernel beside it moves the stream bytes once more)" $OUT/jpeg_batch_pmc.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
for sub in 420 422; do UG_JPEG_PROF=1 timeout 120 python tools/bench_jpeg_batch.py --sub $sub --only batch --calls 40 2>&1 | grep "UG_JPEG_PROF" | sed "s/^/$sub /"; done > $OUT/jpeg_phase_clock.txt; cat $OUT/jpeg_phase_clock.txt
python bench.py --workload 4k-uyvy-jpeg-encode --no-e2e > $OUT/bench_4k-uyvy-jpeg-encode.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_4k-uyvy-jpeg-encode.json
{ timeout 100 python tools/bench_jpeg_batch.py; timeout 100 python tools/bench_jpeg_batch.py --n 16 --only batch; timeout 100 python tools/bench_jpeg_batch.py --sub 422; timeout 100 python tools/bench_jpeg_batch.py --sub 444;
  timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 7680x4320 --n 4; timeout 100 python tools/bench_jpeg_batch.py --sub 422 --size 1920x1080 --n 16;
  UG_JPEG_LOOKBACK=1 timeout 100 python tools/bench_jpeg_batch.py --only batch | sed 's/^/UG_JPEG_LOOKBACK=1 /'; UG_JPEG_FUSED=0 timeout 100 python tools/bench_jpeg_batch.py --only batch | sed 's/^/UG_JPEG_FUSED=0 /'; } 2>&1 | grep "per call" > $OUT/jpeg_batch_all.txt; cat $OUT/jpeg_batch_all.txt
for sub in 420 422; do for q in 50 75 90 95 98 100; do timeout 60 python tools/bench_jpeg_batch.py --sub $sub --q $q --only batch --seconds 0.3 2>&1 | grep "per call" | tail -1; done; done > $OUT/jpeg_quality_sweep.txt; cat $OUT/jpeg_quality_sweep.txt | cut -c1-120
timeout 600 python tools/find_encode_mismatch.py 3000 2>&1 | tail -1 > $OUT/find_encode.txt; cat $OUT/find_encode.txt
timeout 600 python tools/find_libjpeg_mismatch.py 3000 2>&1 | grep -v "amdgpu.ids\|JPEG\]\|APP14" | tail -1 > $OUT/find_libjpeg.txt; cat $OUT/find_libjpeg.txt
