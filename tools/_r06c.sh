mkdir -p gpurun_out/r06c; O=gpurun_out/r06c
timeout 700 python tools/infer_end_to_end.py --frames 600 --out $O/infer_end_to_end.txt > $O/e2e.log 2>&1; cat $O/infer_end_to_end.txt | cut -c1-260
for L in epos_amd/lib/libepos_hip_r06base.so epos_amd/lib/libepos_hip.so; do echo "== $L"; EPOS_HIP_LIB=/root/repo/$L python tools/bench_dw.py --h2 2>&1 | tail -9; done > $O/bench_dw_h2.txt 2>&1; cat $O/bench_dw_h2.txt
bash tools/ab_lib.sh /root/repo/epos_amd/lib/libepos_hip_r06base.so /root/repo/epos_amd/lib/libepos_hip.so > $O/ab_dw_kernarg.txt 2>&1; cat $O/ab_dw_kernarg.txt
timeout 900 bash tools/ab_prune.sh > $O/ransac_pruning_ab.txt 2>&1; cat $O/ransac_pruning_ab.txt | cut -c1-400
for f in 0.3 0.5 0.7; do python bench.py --planted-poses --planted-outliers $f --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --pipeline-depth 3 --steps 40 --warmup 5 --no-cpu-baseline --traffic static 2>$O/c4_planted_$f.err | tail -1 > $O/bench_planted_c4_$f.json; done
python bench.py --height 540 --width 720 --num-objs 30 --objs-per-image 8 --instances 2 --pipeline-depth 3 --steps 40 --warmup 5 --no-cpu-baseline --traffic static 2>/dev/null | tail -1 > $O/bench_c4.json
for f in $O/bench_planted_c4_*.json $O/bench_c4.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d.get('planted') and {k:d['planted'][k] for k in ('planted_poses','recovered_within_1deg_5mm','rot_err_deg_max','trans_err_mm_max','inlier_correspondences_per_object_mean','ok')}, (d.get('serial_depth1') or {}).get('stage_ms'))"; done
timeout 600 bash tools/pmc_dw.sh > $O/pmc_dw.json 2>$O/pmc_dw.err; cat $O/pmc_dw.json | head -60
