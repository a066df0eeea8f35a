cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python tools/batch_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_batch_probe.txt | cut -c1-400
