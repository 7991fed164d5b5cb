O=gpurun_out/r03h; mkdir -p $O
( time python tools/find_dxt_mismatch.py 6000 ) 2>&1 | tail -12 | tee $O/find_dxt.txt
( time python tools/find_pixfmt_mismatch.py 8000 ) 2>&1 | tail -12 | tee $O/find_pixfmt.txt
( time python tools/find_encode_mismatch.py 600 ) 2>&1 | tail -12 | tee $O/find_jpeg_encode.txt
( time python tools/find_module_mismatch.py 120 ) 2>&1 | tail -12 | tee $O/find_module.txt
bash tools/ab_ntload.sh 2>&1 | tee $O/ab_ntload.txt
