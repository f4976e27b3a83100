mkdir -p gpurun_out/r6i && cd /root/repo
python tools/exp/cfg3_knobs.py 3 occupancy > gpurun_out/r6i/cfg3_occupancy.txt 2> gpurun_out/r6i/cfg3_occupancy.err; echo "rc=$?"; cat gpurun_out/r6i/cfg3_occupancy.txt; tail -3 gpurun_out/r6i/cfg3_occupancy.err
