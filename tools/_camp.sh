mkdir -p gpurun_out/r04camp
python tools/fuzz_campaign_programs.py 42100 700 8 > gpurun_out/r04camp/programs2.txt 2>&1
tail -3 gpurun_out/r04camp/programs2.txt
