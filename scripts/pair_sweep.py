"""bench.py --config 2 with other group sizes of the later stages (Parser.PAIR): python scripts/pair_sweep.py PAIR [bench args]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
pair = int(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402
from cube_amd.networks.parser import Parser  # noqa: E402

Parser.PAIR = pair
Parser.N_SLOTS = Parser.N_GROUPS = 2 * pair + 1
bench.main()
