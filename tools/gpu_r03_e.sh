O=gpurun_out/r03e; mkdir -p $O
python tools/e2e_bench.py --sweep --seconds 2 > $O/e2e_sweep.jsonl 2> $O/e2e_sweep.err; tail -3 $O/e2e_sweep.err; python - <<PY
import json
for l in open("$O/e2e_sweep.jsonl"):
    d=json.loads(l)
    if d.get("probe")=="link": print("link", d["streams_per_direction"], "streams: h2d", d["h2d_gbs"], "d2h", d["d2h_gbs"], "bidir each", d["bidir_each_gbs"])
    else: print(d["workload"], d["mode"], "depth", d["in_flight"], "fps", d["fps"], "copy-only", d["copy_only_fps"], "frac", d["frac_of_copy_only"], "h2d", d["h2d_gbs"], "d2h", d["d2h_gbs"])
PY
python tools/bench_kernels.py --json $O/kernels.json > $O/kernels_table.txt 2>&1; tail -3 $O/kernels_table.txt
python tools/bench_decode.py --json $O/decode.json > $O/decode.txt 2>&1; grep -- "->" $O/decode.txt
