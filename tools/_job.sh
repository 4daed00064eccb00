set -o pipefail
bash tools/profile_round.sh r04 > gpurun_out/prof_r04_stdout.txt 2>&1; echo "rc $?"
tail -30 gpurun_out/prof_r04/log.txt
ls gpurun_out/prof_r04 | head -80
