set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout 2400 python tools/chain_error_probe.py 8 > gpurun_out/r4h/chain_error_probe_8_blocks.json 2> gpurun_out/r4h/probe.err
tail -c 2500 gpurun_out/r4h/chain_error_probe_8_blocks.json; tail -3 gpurun_out/r4h/probe.err
