mkdir -p gpurun_out/r04camp
python tools/fuzz_campaign_programs.py 41100 160 8 > gpurun_out/r04camp/programs.txt 2>&1
tail -3 gpurun_out/r04camp/programs.txt
