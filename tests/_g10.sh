cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tests/ab_step.py cur pre > gpurun_out/r03h_ab_reverse.txt 2>&1; cat gpurun_out/r03h_ab_reverse.txt
bash tests/gpu_session.sh r03h bench prof pmc 2>&1 | tail -60
