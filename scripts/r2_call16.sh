mkdir -p gpurun_out
T='tests/test_gpu_preset_scale.py::test_preset_train_step_matches_oracle[f16x3-nyanko_ljspeech]'
for f in test_audio test_gpu_ddp test_gpu_kernels test_gpu_model; do
  timeout 600 python -m pytest tests/$f.py "$T" -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r2o_bisect_$f.log
  echo "== $f: $(tail -1 gpurun_out/r2o_bisect_$f.log)"
  grep -h "FAILED" gpurun_out/r2o_bisect_$f.log
done
