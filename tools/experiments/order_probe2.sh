cd $GRAFT_REPO_ROOT
A=tests/test_sparse_gpu.py::test_dynamic_voxelize_mean
T=tests/test_train_step_gpu.py::test_static_step_and_graph_reproduce_the_exact_shape_step
for e in GLX_X=1 GLX_BEV_SPARSE_FIRST=0 GLX_PREPACK=0 GLX_OWN_CONV3X3=0 GLX_OWN_DECONV=0 GLX_OWN_S2_FWD=0 "GLX_OWN_CONV3X3=0 GLX_OWN_DECONV=0 GLX_BEV_SPARSE_FIRST=0"; do
  echo "== $e: $(env $e timeout 300 python -m pytest $A $T -x -q -m gpu 2>&1 | tail -1 | cut -c1-80)"
done
