#!/bin/bash
# after the 5-waves-per-SIMD class became the default for unpaired end-to-end batches: the headline line again (reference timed beside it),
# the three PMC passes at 2 M reads per launch, a kernel trace, and -- time permitting -- the 862-run regression table on the device
T=${1:-r04zy}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O/se150; cd $R; export TMPDIR=/tmp
P=$O/se150
(timeout 400 python bench.py --steps 25 --warmup 5 2>$P/bench.err | tail -1) > $P/bench_prelim.json
cd /tmp
CMD="python $R/bench.py --steps 1 --warmup 1 --reads 2000000 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/a -- $CMD > $P/a.json 2> $P/a.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/b -- $CMD > $P/b.json 2> $P/b.err
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/c -- $CMD > $P/c.json 2> $P/c.err
cd $R
python tools/pmc_summary.py $P 2000000 2 > $P/pmc_summary.txt 2>&1; grep "k_align" $P/pmc_summary.txt | cut -c1-160 | head -12
find $P -name "*.csv" -size +1M -delete
[ -s $P/pmc_traffic.json ] && cp $P/pmc_traffic.json $R/profiles/${T}_pmc_traffic.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $P/bench_under_rocprof.json 2> $P/rocprof.err
f=$(find $P/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $P/kernel_stats.csv && grep "k_align_reads" "$f" | cut -c1-40,300-420
find $P/trace -name "*.csv" -size +1M -delete
cd $R
# the bench line's roofline.traffic / instruction_issue come from the PMC file: take the line again now that it exists (no reference run: the first line has it)
(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>>$P/bench.err | tail -1) > $P/bench_nocpu.json
python3 - <<PY
import json
a = json.load(open("$P/bench_prelim.json")); b = json.load(open("$P/bench_nocpu.json"))
a["roofline"]["traffic"] = b["roofline"]["traffic"]; a["roofline"]["traffic_source"] = b["roofline"]["traffic_source"]
a["roofline"]["traffic_over_algorithmic"] = b["roofline"]["traffic"] / a["roofline"]["algorithmic_bytes_per_launch"] if b["roofline"]["traffic"] else None
a["roofline"]["instruction_issue"] = b["roofline"]["instruction_issue"]
json.dump(a, open("$P/bench.json", "w"))
c = a["config"]; print("se150", round(a["value"]), "reads/s", c["kernel_ms_per_step"], "parity", c.get("parity_identical"), "flagged", c.get("reads_overflowed"), "cpu", a["cpu_baseline"] and round(a["cpu_baseline"]["value"]))
print("  roofline", a["roofline"]["frac"], a["roofline"]["traffic"], a["roofline"]["traffic_over_algorithmic"], a["roofline"]["instruction_issue"])
PY
(timeout 420 python -m pytest -q -x -m gpu tests/test_simple_tests.py 2>&1 | tail -3) | tee $O/pytest_table.log
