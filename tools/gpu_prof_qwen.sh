#!/bin/bash
# rocprofv3 kernel trace + stats of the Qwen2.5-VL-7B bench leg (configs[4]); summary in gpurun_out/prof_qwen/
P=gpurun_out/prof_qwen; mkdir -p $P; export PYTHONUNBUFFERED=1; REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/$P -o bench -- python $REPO/tools/bench_qwen.py --batch 64 --steps 2 --warmup 1 > $REPO/$P/bench_under_rocprof.log 2>&1
echo "rocprof exit $?"; cd $REPO
python tools/rocpd_summary.py $P/bench_results.db $P/summary "rocprofv3 --kernel-trace --stats -- python tools/bench_qwen.py --batch 64 --steps 2 --warmup 1" > /dev/null 2>&1
rm -f $P/*.db; find $P -name "*kernel_trace*" -size +20M -delete
head -30 $P/summary.md | cut -c1-170
