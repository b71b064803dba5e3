mkdir -p gpurun_out/r06e; O=gpurun_out/r06e
(timeout 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log)
(EPOS_H2_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_h2.py tests/test_gpu_layers.py tests/test_gpu_configs.py tests/test_gpu_net.py -x -q > $O/gpu_tests_persist.log 2>&1; tail -3 $O/gpu_tests_persist.log)
COMMON="--steps 60 --warmup 8 --no-cpu-baseline --traffic static" REPS=3 bash tools/ab_env.sh "EPOS_PAD_ROWS=1" "EPOS_PAD_ROWS=2" > $O/ab_pad_level.txt 2>&1; cat $O/ab_pad_level.txt
COMMON="--steps 60 --warmup 8 --no-cpu-baseline --traffic static" REPS=3 bash tools/ab_env.sh "EPOS_H2_PERSIST=0" "EPOS_H2_PERSIST=1" > $O/ab_persist_c2.txt 2>&1; cat $O/ab_persist_c2.txt
COMMON="--steps 40 --warmup 6 --no-cpu-baseline --traffic static --batch-per-gpu 4" REPS=2 bash tools/ab_env.sh "EPOS_H2_PERSIST=0" "EPOS_H2_PERSIST=1" > $O/ab_persist_c3.txt 2>&1; cat $O/ab_persist_c3.txt
COMMON="--steps 20 --warmup 3 --no-cpu-baseline --traffic static --model-variant resnet_v1_101_beta --num-objs 15 --batch-per-gpu 8 --pipeline-depth 2" REPS=2 bash tools/ab_env.sh "EPOS_H2_PERSIST=0" "EPOS_H2_PERSIST=1" > $O/ab_persist_c5.txt 2>&1; cat $O/ab_persist_c5.txt
for P in 0 1; do echo "== EPOS_H2_PERSIST=$P"; EPOS_H2_PERSIST=$P python tools/bench_one_gemm_h2.py 19200 728 728 3 --time 2>&1 | tail -2; EPOS_H2_PERSIST=$P python tools/bench_one_gemm_h2.py 19200 4032 256 3 --time 2>&1 | tail -2;  EPOS_H2_PERSIST=$P python tools/bench_one_gemm_h2.py 76800 128 128 3 --time 2>&1 | tail -2; done > $O/persist_micro.txt 2>&1; cat $O/persist_micro.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_driver_cmd.err | tail -1 > $O/bench_driver_cmd.json; python -c "
import json; d=json.load(open('$O/bench_driver_cmd.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('in_step'), d['cpu_baseline'])"
