cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_hip_c8.py -q -m gpu -k "space_to_depth" 2>&1 | tail -3
timeout -k 10 300 python tools/s2d_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_s2d_probe_interleaved.txt
