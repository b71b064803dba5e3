mkdir -p gpurun_out/r06f; O=gpurun_out/r06f
(timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_infer_cli.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log)
COMMON="--steps 60 --warmup 8 --no-cpu-baseline --traffic static" REPS=3 bash tools/ab_args.sh "--launch-queue 1" "--launch-queue 2" > $O/ab_launch_queue_c2.txt 2>&1; cat $O/ab_launch_queue_c2.txt
COMMON="--steps 60 --warmup 8 --no-cpu-baseline --traffic static --sparse-heads" REPS=2 bash tools/ab_args.sh "--launch-queue 1" "--launch-queue 2" > $O/ab_launch_queue_sparse.txt 2>&1; cat $O/ab_launch_queue_sparse.txt
COMMON="--steps 40 --warmup 6 --no-cpu-baseline --traffic static --batch-per-gpu 4" REPS=2 bash tools/ab_args.sh "--launch-queue 1" "--launch-queue 2" > $O/ab_launch_queue_c3.txt 2>&1; cat $O/ab_launch_queue_c3.txt
timeout 900 python tools/infer_end_to_end.py --frames 600 --out $O/infer_end_to_end.txt > $O/e2e.log 2>&1; cut -c1-330 $O/infer_end_to_end.txt
