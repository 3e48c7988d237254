"""Device-resident throughput of the parser path against the batch size (L = 32): what `process` can reach at best when it
merges reference batches into larger device batches.   python scripts/batch_size_sweep.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 712, 1024]:
    bench.CONFIGS[1] = dict(B=B, L=32, name="sweep")
    p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 4, 0, dev, "")
    bench.run_steps(p, batches, 8)
    torch.cuda.synchronize()
    n = max(8, 25600 // B)
    t0 = time.perf_counter()
    bench.run_steps(p, batches, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bench.run_steps(p, batches, 4, serial=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_steps(p, batches, 8, serial=True)
    torch.cuda.synchronize()
    ds = (time.perf_counter() - t0) / 8
    print("B = %4d: %.3f ms per batch, %.0f sentences/s   (one stream: %.3f ms per batch, %.0f sentences/s; in flight %d)"
          % (B, dt / n * 1e3, n * B / dt, ds * 1e3, B / ds, p.in_flight_for(33)), flush=True)
    del p, batches
    torch.cuda.empty_cache()
