cd $GRAFT_REPO_ROOT
for lib in libfnn_hip.so libfnn_exp.so; do
  for i in 1 2; do
  FNN_LIB=$PWD/fast-nnunet_amd/csrc/$lib python tools/layer_time.py 32 32 32 160 48 48 2>&1 | grep "op time"
  done
  FNN_LIB=$PWD/fast-nnunet_amd/csrc/$lib python tools/layer_time.py 32 32 32 160 48 48 32 2>&1 | grep "op time"
  FNN_LIB=$PWD/fast-nnunet_amd/csrc/$lib python tools/layer_time.py 32 64 64 80 24 24 2>&1 | grep "op time"
done
