set -o pipefail
mkdir -p gpurun_out/r04aa
bash tools/sweep_env.sh r04aa/jg2 "--workload network --no-cpu-baseline" "PGBP_TUNING=chunk_max_tasks=3072" "PGBP_TUNING=chunk_max_tasks=4096" "PGBP_TUNING=chunk_max_tasks=6144" "PGBP_TUNING=chunk_max_tasks=8192" "PGBP_TUNING=chunk_max_tasks=12288" "PGBP_TUNING=chunk_max_tasks=4096,chunk_bins=512" "PGBP_TUNING=chunk_max_tasks=8192,chunk_bins=512" "PGBP_TUNING=chunk_max_tasks=3072" || exit 1
bash tools/sweep_env.sh r04aa/be2 "--workload network --graph bethe --no-cpu-baseline" "PGBP_TUNING=chunk_max_tasks=3072" "PGBP_TUNING=chunk_max_tasks=4096" "PGBP_TUNING=chunk_max_tasks=6144" "PGBP_TUNING=chunk_max_tasks=8192" "PGBP_TUNING=chunk_max_tasks=4096,chunk_bins=512" "PGBP_TUNING=chunk_max_tasks=8192,chunk_bins=512" || exit 1
