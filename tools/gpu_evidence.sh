#!/bin/bash
# Round evidence in one gpurun call: GPU tests, the driver-contract bench lines, rocprofv3 kernel trace + PMC passes of the headline
# bench, per-kernel tables (+ their rocprofv3 trace), the all-pairs decoders[] table, decoder table, end-to-end / module-level / soak
# raw outputs, the reference's CPU converters on all host cores.  Everything lands in gpurun_out/<tag>/; copy what is to be kept to profiles/.
TAG=${1:-r02}
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q --maxfail=20 2>&1 | grep -v "lavc_vid_conv" | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench_line.json
for wl in 8k-v210 1080p-rgb-dxt1; do python bench.py --workload $wl --no-e2e > $OUT/bench_$wl.json 2>> $OUT/bench.err; done
python bench.py --workload 4k-uyvy-jpeg420 > $OUT/bench_4k-uyvy-jpeg420.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-e2e > $OUT/trace.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/trace/bench_results.db > $OUT/kernel_trace.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace_jpeg -o bench -- python $ROOT/bench.py --workload 4k-uyvy-jpeg420 --no-cpu-baseline > $OUT/trace_jpeg.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/trace_jpeg/bench_results.db > $OUT/kernel_trace_jpeg420.txt 2>&1
cd $ROOT
bash tools/pmc_collect.sh $TAG > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary.txt
python tools/bench_kernels.py --json $OUT/kernels.json > $OUT/kernels_table.txt 2>&1; tail -3 $OUT/kernels_table.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace_kernels -o k -- python $ROOT/tools/bench_kernels.py > $OUT/trace_kernels.log 2>&1)
python tools/pmc_summary.py $OUT/trace_kernels/k_results.db > $OUT/all_kernels_trace.txt 2>&1
python tools/bench_pixfmt_all.py --json $OUT/pixfmt_all_8k.json > $OUT/pixfmt_all_8k.txt 2>&1; tail -1 $OUT/pixfmt_all_8k.txt | cut -c1-300
python tools/bench_decode.py --json $OUT/decode.json > $OUT/decode.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace_jpegenc -o j -- python $ROOT/tools/jpeg_profile.py 3840 2160 4 > $OUT/trace_jpegenc.log 2>&1)
python tools/pmc_summary.py $OUT/trace_jpegenc/j_results.db > $OUT/jpeg_encoder_trace.txt 2>&1
python tools/e2e_bench.py --workload all > $OUT/e2e_bench.txt 2>&1; cat $OUT/e2e_bench.txt | cut -c1-200
bash tools/module_fps.sh > $OUT/module_fps.txt 2>&1; tail -4 $OUT/module_fps.txt
bash tools/soak.sh > $OUT/soak.txt 2>&1; tail -4 $OUT/soak.txt
tools/dxt5_16lane_experiment > $OUT/16lane.txt 2>&1
python tools/cpu_reference_bench.py --json $OUT/cpu_reference_pixfmt.json > $OUT/cpu_reference_pixfmt.txt 2>&1; tail -3 $OUT/cpu_reference_pixfmt.txt
rm -rf $OUT/trace $OUT/trace_jpeg $OUT/trace_kernels $OUT/trace_jpegenc gpurun_out/pmc_$TAG/*.db
ls $OUT
python tools/bench_jpeg_decode.py --json $OUT/jpeg_decode.json > $OUT/jpeg_decode.txt 2>&1; head -3 $OUT/jpeg_decode.txt
python tools/bench_jpeg_decode.py --width 7680 --height 4320 --configs 1 --json $OUT/jpeg_decode_8k.json >> $OUT/jpeg_decode.txt 2>&1
bash tools/pmc_jpeg_dec.sh > $OUT/pmc_jpeg_dec.log 2>&1; cp gpurun_out/pmc_jpeg_dec/summary.txt $OUT/jpeg_decoder_pmc.txt
