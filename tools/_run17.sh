cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3u
for st in 10 30 60; do
  echo "== steps $st"
  timeout 600 python bench.py --steps $st --warmup 2 --num-objs 1 --objs-per-image 1 --no-cpu-baseline --traffic static --no-roofline > gpurun_out/r3u/c1.out 2> gpurun_out/r3u/c1.err; echo rc=$?; tail -1 gpurun_out/r3u/c1.err | cut -c1-200
done
echo "== depth1"
timeout 600 python bench.py --steps 60 --warmup 2 --num-objs 1 --objs-per-image 1 --no-cpu-baseline --traffic static --no-roofline --pipeline-depth 1 > gpurun_out/r3u/c1.out 2> gpurun_out/r3u/c1.err; echo rc=$?; tail -1 gpurun_out/r3u/c1.err | cut -c1-200
echo "== roofline on, steps 10"
timeout 600 python bench.py --steps 10 --warmup 2 --num-objs 1 --objs-per-image 1 --no-cpu-baseline --traffic static > gpurun_out/r3u/c1.out 2> gpurun_out/r3u/c1.err; echo rc=$?; tail -1 gpurun_out/r3u/c1.err | cut -c1-200
