set -x
python tools/config_sweep.py --only wide --sites 32768 --steps 6 > gpurun_out/r06_variants_at_bench_size.txt 2>&1
python tools/config_sweep.py --only merged_hybrid_250 --sites 32768 --steps 6 >> gpurun_out/r06_variants_at_bench_size.txt 2>&1
python tools/config_sweep.py --only single_tech_softplus --sites 8192 --steps 6 >> gpurun_out/r06_variants_at_bench_size.txt 2>&1
python tools/config_sweep.py --only single_tech_addendum --sites 8192 --steps 6 >> gpurun_out/r06_variants_at_bench_size.txt 2>&1
python tools/config_sweep.py --sites 4096 > gpurun_out/r06_config_sweep_1gpu.txt 2>&1
(time python bench.py) > gpurun_out/r06_bench2.json 2> gpurun_out/r06_bench2.err
tail -5 gpurun_out/r06_bench2.err
cat gpurun_out/r06_variants_at_bench_size.txt
