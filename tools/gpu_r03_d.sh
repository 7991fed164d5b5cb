O=gpurun_out/r03d; mkdir -p $O
( time python -m pytest tests -m gpu -q --maxfail=20 ) 2>&1 | grep -v "lavc_vid_conv\|Using CUDA FFmpeg" | tail -30 > $O/pytest.log; tail -6 $O/pytest.log
bash tools/ab_nt.sh 2>&1 | tee $O/ab_nt_pixfmt.txt
