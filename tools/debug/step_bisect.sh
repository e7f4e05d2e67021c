T='python -m pytest tests/test_train_step_gpu.py -x -q -k "test_step_gradients_match_oracle and densenet and disc"'
for v in "OTGAN_DENSE_SPLIT=0 OTGAN_WN_BATCHED=0" "OTGAN_DENSE_SPLIT=0 OTGAN_WN_BATCHED=0 OTGAN_DISABLE_WINO_PLAIN3=1" "OTGAN_WN_BATCHED=0" "OTGAN_DISABLE_WINO_PLAIN3=1"; do
  echo "== $v"; env $v timeout 600 bash -c "$T" 2>&1 | grep -E "AssertionError:|passed|failed" | head -3
done
