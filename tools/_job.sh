export TMPDIR=/tmp
bash tools/profile_r06.sh r06 > gpurun_out/r06_profile.log 2>&1
tail -5 gpurun_out/r06_profile.log
