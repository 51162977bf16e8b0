mkdir -p gpurun_out/r03f
python -m pytest tests -m gpu -q -x > gpurun_out/r03f/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03f/pytest_gpu.log
python bench.py > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err
python bench.py --mode train --steps 20 --warmup 5 --dtype bf16 > gpurun_out/r03f/train_bf16.json 2>> gpurun_out/r03f/bench.err
python bench.py --mode train --steps 20 --warmup 5 > gpurun_out/r03f/train_fp32.json 2>> gpurun_out/r03f/bench.err
python bench.py --mode train --steps 20 --warmup 5 --model campnet --dtype bf16 > gpurun_out/r03f/campnet_bf16.json 2>> gpurun_out/r03f/bench.err
python bench.py --mode train --steps 20 --warmup 5 --model campnet > gpurun_out/r03f/campnet_fp32.json 2>> gpurun_out/r03f/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03f/smoke.log 2>&1
tail -3 gpurun_out/r03f/pytest_gpu.log; cat gpurun_out/r03f/*.json | cut -c1-400; tail -2 gpurun_out/r03f/smoke.log
