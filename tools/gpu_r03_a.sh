# round 3, GPU session A: full GPU test suite, bench line, fp64/int microbench, DXT decoder counters
O=gpurun_out/r03a; mkdir -p $O
( time python -m pytest tests -m gpu -q --maxfail=20 --durations=15 ) 2>&1 | grep -v "lavc_vid_conv\|Using CUDA FFmpeg" | tail -60 > $O/pytest.log
tail -8 $O/pytest.log
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 2500 $O/bench_line.json; tail -3 $O/bench.err
UG_MB_F64=1 tools/valu_microbench > $O/valu_microbench_f64.txt 2>&1; tail -16 $O/valu_microbench_f64.txt
bash tools/pmc_dxt_decode.sh r03a > $O/pmc_dxtdec.log 2>&1; tail -60 $O/pmc_dxtdec.log
