set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
python tools/diag_concurrent.py > gpurun_out/r3e/diag_concurrent.txt 2>&1; tail -20 gpurun_out/r3e/diag_concurrent.txt | cut -c1-330
