mkdir -p gpurun_out/r04serial
s=$(date +%s)
python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r04serial/suite.txt 2>&1
echo "rc=$? total_s=$(( $(date +%s) - s ))" | tee gpurun_out/r04serial/time.txt
tail -3 gpurun_out/r04serial/suite.txt
python -m pytest tests/ -q -m gpu --durations=15 -p no:cacheprovider -k "cli or compat" > gpurun_out/r04serial/durations.txt 2>&1
tail -25 gpurun_out/r04serial/durations.txt
