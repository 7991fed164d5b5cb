#!/bin/bash
# HBM traffic + VALU counters of the dominant kernel of the other bench workloads (separate --pmc passes, --kernel-trace only); GPU box.
#   bash tools/pmc_workloads.sh   -> gpurun_out/pmc_workloads/<workload>.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_workloads
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in ${WORKLOADS:-8k-v210 1080p-rgb-dxt1 4k-uyvy-jpeg420 4k-uyvy-jpeg-encode}; do
  CMD="python $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --launches-per-step 16 --no-cpu-baseline --no-e2e --no-configs --no-parity-check"
  rm -rf /tmp/pw; mkdir -p /tmp/pw
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES -d /tmp/pw -o sq -- $CMD > /tmp/pw/sq.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o write -- $CMD > /tmp/pw/write.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d /tmp/pw -o tcc -- $CMD > /tmp/pw/tcc.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/pw/*.db 2>&1 | grep -E "^==|dxt_encode_kernel|uyvy_jpeg|jpeg_code_kernel" | grep -v "^kernel" > $OUT/$wl.txt
  grep -E "pmc" $OUT/$wl.txt | cut -c1-12,70-160
done
