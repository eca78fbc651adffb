BENCH_ARGS="--preset nyanko_ljspeech --gemm bf16" bash scripts/r2_prof.sh r2t_nyanko_bf16
BENCH_ARGS="--preset deepvoice3_vctk --gemm bf16" bash scripts/r2_prof.sh r2t_vctk_bf16
