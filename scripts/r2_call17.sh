mkdir -p gpurun_out
T='tests/test_gpu_preset_scale.py::test_preset_train_step_matches_oracle[f16x3-nyanko_ljspeech]'
i=0
for k in "test_training_forward_backward_matches_oracle or test_train_step_matches_reference_golden" "test_graphed_train_step_matches_reference_golden" "test_bf16_mode_forward_and_gradients or test_graphed_decode_equals_eager_decode or test_eval_after_training_uses_current_weights" "test_resume_from_reference_checkpoint or test_save_load_step" "test_fast_decode_equals or test_priority_freq_weight"; do
  i=$((i+1))
  timeout 600 python -m pytest tests/test_gpu_model.py "$T" -m gpu -q -p no:cacheprovider -k "($k) or test_preset_train_step_matches_oracle" 2>&1 | tail -4 > gpurun_out/r2p_bisect_$i.log
  echo "== [$k]: $(tail -1 gpurun_out/r2p_bisect_$i.log)"
done
