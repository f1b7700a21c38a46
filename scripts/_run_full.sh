cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4c_pytest.log 2>&1; tail -3 gpurun_out/r4c_pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s3 -- python3 scripts/time_stem.py 256 2 > gpurun_out/r04_time_stem.txt 2>&1; f=$(ls -t gpurun_out/prof_s3/*/*kernel_stats.csv | head -1); grep stem_fused $f | cut -d, -f1-4 >> gpurun_out/r04_time_stem.txt; grep -v amdgpu gpurun_out/r04_time_stem.txt
R50_LIB=$PWD/implementation_phd_lab_vision_amd/libr50hip_stamp.so timeout -k 10 100 python scripts/stamp_stem.py > gpurun_out/r04_stamps_stem.txt 2>&1; grep -v amdgpu gpurun_out/r04_stamps_stem.txt
timeout -k 10 300 python3 bench.py > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err; cut -c1-400 gpurun_out/r4c_bench.json
