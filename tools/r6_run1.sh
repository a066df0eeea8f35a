cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout -k 10 600 python tools/dump_separated_decoder.py > gpurun_out/r6_dump_sep.log 2>&1; echo rc $?
ls -la gpurun_out/sep_sd_d.pt
timeout -k 10 900 python bench.py > gpurun_out/r6_bench0.json 2> gpurun_out/r6_bench0.err; echo rc $?
tail -c 1500 gpurun_out/r6_bench0.json
