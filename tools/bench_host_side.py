"""developer helper / CPU test driver: bench.py's control flow (stream plan, prepare, warm-up and timed passes, barriers and the MAX
all-reduce of the N > 1 path, statistics pass, the JSON line) against the stand-in HIP runtime of tests/hoststub.  No kernel runs, the
numbers mean nothing; what is checked is that every rank gets through and rank 0 prints one well-formed line.
  python tools/bench_host_side.py
  VVR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/bench_host_side.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.device_count = lambda: 1
import vvdec_amd
import test_host_glue as T
vvdec_amd._LIBPATH = T.LIB
import bench
sys.argv = ["bench.py", "--width", "256", "--height", "128", "--steps", "6", "--warmup", "4", "--gop", "4", "--intra-period", "8", "--irap-lookahead", "2",
            "--streams", "3", "--slots", "10", "--no-cpu-baseline", "--verify", "0"] + os.environ.get("VVR_BENCH_EXTRA", "").split()
bench.main()
