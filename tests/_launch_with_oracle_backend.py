"""`python -m kge_amd.libkge_plugin.launch` for the CPU test suite: the same launcher, with the oracle-backed stand-in
scoring backend (tests/test_sharded_gloo_cpu.OracleBackend) assigned to sharded_job.SHARD_BACKEND first -- job.device cpu
has no kernels to run on.  Test infrastructure: the package itself imports nothing named by an environment variable."""
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
for pth in (here, os.path.join(os.path.dirname(here), "oracle"), os.path.dirname(here)):
    if pth not in sys.path:
        sys.path.insert(0, pth)

import kge_amd.libkge_plugin.sharded_job as sj  # noqa: E402
from test_sharded_gloo_cpu import OracleBackend  # noqa: E402
from kge_amd.libkge_plugin.launch import main  # noqa: E402

sj.SHARD_BACKEND = OracleBackend
main()
