"""bench.py with another number of batches in flight for short sentences: python scripts/inflight_sweep.py N [bench args]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
n = int(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402
from cube_amd.networks.parser import Parser  # noqa: E402

orig = Parser.in_flight_for
Parser.in_flight_for = lambda self, T: n if T < self.PAIR_MIN_T else orig(self, T)
bench.main()
