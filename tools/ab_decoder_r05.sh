cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -3
python -m pytest tests/test_decoder.py -x -q -m gpu 2>&1 | tail -3
for cfg in "base:MSS_WGRAD_KSPLIT=0" "ksplit:MSS_WGRAD_KSPLIT=1" "ksplit_bn64_32:MSS_WGRAD_KSPLIT=1 MSS_GEMM_BN64_MINK=32" "ksplit_bn64_17:MSS_WGRAD_KSPLIT=1 MSS_GEMM_BN64_MINK=17"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for split in 0 1; do
    echo "== $name split=$split"
    env $envs MSS_GEMM_SPLIT=$split python tools/bench_decoder.py 2>/dev/null | grep c4_n16
  done
done
