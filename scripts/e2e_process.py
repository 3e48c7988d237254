"""End-to-end Parser.process(Document) throughput incl. featurisation, upload and write-back (the secondary
metric of SURVEY 8d), with the host stages of one batch timed one by one.

    python scripts/e2e_process.py [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from cube_amd import synth  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 1, 0, dev, "")
    col = MorphoCollate(enc)
    out = {"rows": []}
    n = 16384
    doc = synth.make_document(enc, [32] * 1024, seed=77, ext_word_emb=(300,))
    # 16k DISTINCT sentences (copies of one block would let Parser.dedup_words skip most of the character CNN)
    big = synth.make_document(enc, [32] * 1024, seed=78, ext_word_emb=(300,))
    k = 0
    while len(big.sentences) < n:
        big.sentences.extend(synth.make_document(enc, [32] * 1024, seed=1078 + k, ext_word_emb=(300,)).sentences)
        k += 1
    for bs, merge in ((256, True), (32, True), (4, True), (4, False)):  # merged device batches: <= 16384 padded tokens
        m = n if merge else 1024
        sub = big if merge else doc
        p.process(sub, col, batch_size=bs, merge=merge)  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.process(sub, col, batch_size=bs, merge=merge)
        dt = time.perf_counter() - t0
        row = dict(batch_size=bs, merge=merge, sentences=m, seconds=dt, sentences_per_s=m / dt)
        print(json.dumps(row), flush=True)
        out["rows"].append(row)

    def tensor_bytes(o):
        if torch.is_tensor(o):
            return o.numel() * o.element_size()
        if isinstance(o, dict):
            return sum(tensor_bytes(v) for v in o.values())
        if isinstance(o, (list, tuple)):
            return sum(tensor_bytes(v) for v in o)
        return 0
    # what process(max_tokens=32768) holds once it has seen its largest batches (grow-only workspace, 7 result slots, staging ring)
    out["footprint"] = dict(device_bytes_peak=int(torch.cuda.max_memory_allocated()), device_bytes_reserved=int(torch.cuda.memory_reserved()),
                            pinned_staging_bytes=int(tensor_bytes(p.__dict__.get("_staging", []))), staging_sets=len(p.__dict__.get("_staging", [])))
    print(json.dumps(out["footprint"]), flush=True)
    # the same document with word frequencies as text has them (Zipf over a 20000-word pool): what Parser.dedup_words
    # (one character-CNN row per distinct (spelling, language) pair of a batch) is for
    zdoc = synth.make_document(enc, [32] * 1024, seed=9078, ext_word_emb=(300,))
    k = 0
    while len(zdoc.sentences) < n:
        zdoc.sentences.extend(synth.make_document(enc, [32] * 1024, seed=10078 + k, ext_word_emb=(300,)).sentences)
        k += 1
    synth.zipf_words(zdoc)
    for flag in (False, True):
        p.dedup_words = flag
        p.process(zdoc, col, batch_size=256)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.process(zdoc, col, batch_size=256)
        dt = time.perf_counter() - t0
        h = p._featurize(zdoc.sentences[:256], col)
        row = dict(zipf_text=True, dedup_words=flag, sentences=len(zdoc.sentences), sentences_per_s=len(zdoc.sentences) / dt,
                   char_cnn_words_of_8192_tokens=int(h["N"]))
        p._staging_release(h["staging"])
        print(json.dumps(row), flush=True)
        out["rows"].append(row)
    p.dedup_words = True
    # host stages of one 256-sentence batch, one by one (single thread, no overlap)
    batch = big.sentences[:256]
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        h = p._featurize(batch, col)
        p._staging_release(h["staging"])
    t_feat = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        X = col.collate_fn(batch, targets=False)
        hs = p._prepare_host(X)
    t_py = (time.perf_counter() - t) / reps
    h = p._featurize(batch, col)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        bt = p._upload(dict(h, staging=None))
    torch.cuda.synchronize()
    t_up = (time.perf_counter() - t) / reps
    p._staging_release(h["staging"])
    t = time.perf_counter()
    tickets = []
    for _ in range(reps):
        tk = p.submit(bt)
        t_sub_end = time.perf_counter()
        p.collect(tk)
    t_sub_col = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        tk = p.submit(bt)
        tickets.append(tk)
        if len(tickets) == 3:
            p.collect(tickets.pop(0))
    t_sub = (time.perf_counter() - t) / reps
    while tickets:
        res, status = p.collect(tickets.pop(0))
    t = time.perf_counter()
    for _ in range(reps):
        p._write_back(batch, 0, res, status)
    t_wb = (time.perf_counter() - t) / reps
    out["host_stages_ms_per_256_sentences"] = dict(
        featurize_c=t_feat * 1e3, collate_plus_prepare_python=t_py * 1e3, upload_incl_sync=t_up * 1e3,
        submit_pipelined=t_sub * 1e3, submit_collect_serial=t_sub_col * 1e3, write_back_c=t_wb * 1e3)
    print(json.dumps(out["host_stages_ms_per_256_sentences"]), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
