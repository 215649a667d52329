import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# HarvestFeatures with two agents runs four envs per wave only from 2 048 envs per launch up (launch_feat_step); the parity tests
# use small batches, so the test process takes the packed kernels at every size (read once per process by the library).  The
# one-env kernels of that configuration and the default threshold are covered by child processes in test_feat_quad_gpu.py.
os.environ.setdefault("CE_FEAT_QUAD_MIN_ENVS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
