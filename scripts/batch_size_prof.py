"""One batch size of the parser path for rocprofv3 --kernel-trace: which kernels make a batch size slow.
    rocprofv3 --kernel-trace --stats -d out -o name -- python3 scripts/batch_size_prof.py B [steps] [serial|pipelined]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import torch  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
serial = (sys.argv[3] if len(sys.argv) > 3 else "serial") == "serial"
dev = torch.device("cuda", 0)
bench.CONFIGS[1] = dict(B=B, L=32, name="sweep")
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 4, 0, dev, "")
bench.run_steps(p, batches, steps, serial=serial)
torch.cuda.synchronize()
print("done", B, steps, "serial" if serial else "pipelined")
