cd $GRAFT_REPO_ROOT
bash tools/final_check.sh
bash tools/profile_step.sh r05r > /dev/null 2>&1
head -3 gpurun_out/r05r/train_step_kernels.md
