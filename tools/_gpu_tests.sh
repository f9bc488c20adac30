cd /root/repo
timeout 900 python -m pytest tests/test_examples.py tests/test_rowdot.py tests/test_golden.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for i in 1 2; do timeout 300 python bench.py --only lreg > gpurun_out/lreg_$i.json 2>gpurun_out/lreg_$i.err; tail -c 400 gpurun_out/lreg_$i.err; done
