"""Wall-time ratio  oracle (the CPU restatement bench.py times on the GPU host) : GENUINE reference.

Build container only (needs /root/reference; uses the import stand-ins of tests/golden/make_golden.py for the
two uninstalled packages).  SURVEY 8d / BASELINE.md 3: the GPU host cannot run the reference, so the bench's
`cpu_baseline` is the oracle; this script records, on the same 8 cores and the same inputs, how the oracle's
sentences/s relate to those of `cube.networks.parser.Parser.process` itself, so that "x times the host-CPU
reference" is traceable to the real reference.  Workloads = BASELINE configs 0-2, bounded samples:

    config 0: 20-token sentences, reference default batch_size=4 (what Cube.__call__ runs, api.py:125)
    config 1: 32-token sentences, batch_size 4 / 32 / 256
    config 2: 128-token sentences, batch_size 8

    python scripts/ref_ratio.py [--quick]   ->  profiles/r2_ref_ratio.json
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.argv_saved, sys.argv = sys.argv, sys.argv[:1]
import make_golden as mg  # noqa: E402  (imports the reference with the stand-ins; does not run its main)
sys.argv = sys.argv_saved

import torch  # noqa: E402

from cube_amd import synth  # noqa: E402
from oracle import parser_oracle  # noqa: E402
from tests import cases  # noqa: E402


def main():
    quick = "--quick" in sys.argv
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    ref, cfg, enc, sd = mg.build_ref_parser(cases.FULL)
    ref_col = mg.REF["MorphoCollate"](mg.ref_encodings(enc))
    ora = parser_oracle.OracleParser(cfg, enc, sd, [300], quant="fp32")
    work = [("config0 20 tok bs4", 20, 4, 16 if quick else 48), ("config1 32 tok bs4", 32, 4, 16 if quick else 48),
            ("config1 32 tok bs32", 32, 32, 32 if quick else 96), ("config1 32 tok bs256", 32, 256, 256),
            ("config2 128 tok bs8", 128, 8, 8 if quick else 16)]
    rows = []
    for name, L, bs, n in work:
        doc = synth.make_document(enc, [L] * n, seed=4000 + L + bs, ext_word_emb=(300,))
        rdoc = mg.ref_document(doc)
        odoc = synth.make_document(enc, [L] * n, seed=4000 + L + bs, ext_word_emb=(300,))
        t0 = time.perf_counter()
        ref.process(rdoc, ref_col, batch_size=bs, num_workers=0)
        t_ref = time.perf_counter() - t0
        t0 = time.perf_counter()
        with torch.no_grad():
            ora.process(odoc, batch_size=bs)
        t_ora = time.perf_counter() - t0
        same = sum(str(a) == str(b) for a, b in zip(rdoc.sentences, odoc.sentences))
        row = dict(workload=name, sentences=n, tokens=L, batch_size=bs, reference_s=t_ref, oracle_s=t_ora,
                   reference_sent_per_s=n / t_ref, oracle_sent_per_s=n / t_ora, oracle_over_reference=t_ref / t_ora,
                   identical_sentences=same)
        print(json.dumps(row), flush=True)
        rows.append(row)
    out = dict(host="build container, %d threads, torch %s CPU kernels" % (threads, torch.__version__),
               note="genuine reference = cube.networks.parser.Parser.process (adobe/NLP-Cube 3.0) imported with stand-ins for "
                    "pytorch_lightning / fasttext; oracle = oracle/parser_oracle.py, the `cpu_baseline` of bench.py; same "
                    "weights (seed 1234), same documents; oracle_over_reference > 1 means the oracle is FASTER than the "
                    "reference, i.e. speed-ups quoted against the oracle understate speed-ups against the reference",
               rows=rows)
    with open(os.path.join(ROOT, "profiles", "r2_ref_ratio.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
