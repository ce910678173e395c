#!/bin/bash
# The reference's OWN training script, unmodified, on the MI355X through gshell_amd.compat (tests/ref_script_harness.py in `cuda` mode).
#   build box:  tools/run_ref_train_script_gpu.sh stage       # copies the three reference files the script needs into the git-ignored .ref_scratch/
#               gpurun -- tools/run_ref_train_script_gpu.sh run [iterations]      # -> gpurun_out/r06/ref_train_script_gpu.log
#               tools/run_ref_train_script_gpu.sh clean       # removes the scratch copy again
# (/root/reference does not exist on the GPU box; nothing of it is committed.)
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$root"
case "$1" in
  stage)
    mkdir -p .ref_scratch/render
    cp /root/reference/train_gshelltet_deepfashion.py .ref_scratch/
    cp /root/reference/render/material.py /root/reference/render/texture.py .ref_scratch/render/ ;;
  run)
    mkdir -p gpurun_out/r06
    work=$(mktemp -d)
    python tests/ref_script_harness.py "$root/.ref_scratch" "$work" cuda "${2:-60}" > gpurun_out/r06/ref_train_script_gpu.log 2>&1
    echo "rc=$?" >> gpurun_out/r06/ref_train_script_gpu.log
    grep -n "iter=\|PSNR\|AVERAGES\|rc=\|Error" gpurun_out/r06/ref_train_script_gpu.log | tail -20 ;;
  clean)
    rm -rf .ref_scratch ;;
esac
