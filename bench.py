"""Benchmark of the parser+tagger inference hot path (BASELINE.json metric: sentences/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|2|3]

One "step" = one pass of the hot path over one batch of synthetic sentences whose collated
inputs are already resident in HBM: device index arrays + external embeddings ->
heads / labels / UPOS / XPOS / ATTRS ids on the host (one D->H copy).  Config 1 (default,
BASELINE.json configs[1]): B=256 sentences x L=32 tokens, default ParserConfig, SURVEY.md §8d
vocabulary, random-init weights (seed 1234).  Config 2: B=64 x L=128.

For N>1 launch with torch.distributed.run (one rank per GPU); ranks shard the sentences (every
rank processes its own B-sentence batches: weak scaling, no collective on the data path); the
timed region is bracketed by barrier + synchronize and the max over ranks is reported.
Config 3 (--config 3) is the STRONG-scaling job of SURVEY 8d/8e: one fixed corpus of 262 144 ragged
sentences (64 repetitions of a 4096-sentence sample; lengths ~ clip(N(22,10), 3, 80), three flavours),
each rank takes its contiguous shard, runs it in length-bucketed batches of <= 8192 padded tokens and
the [n_words, 5] int32 result arrays are gathered on rank 0 (RCCL `gather`; nothing at N=1).  A step is
one pass over the whole corpus; value = corpus sentences x steps / wall time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the gated Conv1d GEMM of the
character CNN, timed live with HIP events on the launch stream) and `cpu_baseline` (the oracle,
i.e. the CPU restatement of the reference path, timed on a bounded sample on this host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "nlp-cube_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {1: dict(B=256, L=32, name="en parser+tagger B=256 x L=32 (BASELINE configs[1])"),
           2: dict(B=64, L=128, name="long sentences B=64 x L=128 (BASELINE configs[2])"),
           # SURVEY §8d config 3, scaled to a sample: ragged lengths, length-bucketed batches of <= 8192 tokens
           3: dict(B=None, L=None, n_sent=4096, name="ragged multi-flavour corpus L~clip(N(22,10),3,80), ewt/lines/gum, "
                                                        "262144 sentences = 64 x a 4096-sentence sample, sharded over the "
                                                        "ranks, bucketed <=8192 tokens (BASELINE configs[3], SURVEY 8d config 3)")}
BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
I8_DENSE_PEAK_TOPS = 5000.0      # same table: I8 32x32x32 = 2x BF16 per clock
SUSTAINED_I8_TOPS, SUSTAINED_BF16_TFLOPS = 3530.0, 1890.0  # measured, profiles/r2_mfma_peak.txt


def build_model_and_batches(cfg_id, n_batches, rank, device, precision, ext=(300,)):
    from cube_amd import synth
    from cube_amd.io_utils.config import ParserConfig
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.utils import MorphoCollate
    c = CONFIGS[cfg_id]
    cfg = ParserConfig()
    enc = synth.make_encodings()
    sd = synth.make_parser_state_dict(cfg, enc, list(ext), seed=1234)
    p = Parser(cfg, enc, language_codes=synth.EN_LANGUAGE_CODES, ext_word_emb=list(ext))
    if precision:
        p.precision = precision
    p.load_state_dict(sd)
    p.to(device)
    col = MorphoCollate(enc)
    batches, docs = [], []
    if cfg_id == 3:
        # the same base sample on every rank (seed without the rank): the corpus is ONE fixed document
        from cube_amd.sharding import bucketed_batches
        rng = np.random.default_rng(3000)
        lens = np.clip(np.round(rng.normal(22, 10, size=c["n_sent"])), 3, 80).astype(np.int64)
        flav = np.array([synth.EN_LANGUAGE_CODES.index(f) for f in ("en_ewt", "en_lines", "en_gum")])
        doc = synth.make_document(enc, lens, seed=3000, ext_word_emb=tuple(ext), lang_ids=flav[rng.integers(0, 3, size=len(lens))])
        woff = np.concatenate([[0], np.cumsum(lens)])
        for idx in bucketed_batches(lens, 8192):
            X = col.collate_fn([doc.sentences[i] for i in idx], targets=False)
            bt = p._prepare(X)
            n = lens[idx]
            rows = np.repeat(np.arange(len(idx)), n)
            cols = np.arange(int(n.sum())) - np.repeat(np.cumsum(n) - n, n)
            bt["_scatter"] = (rows, cols, np.repeat(woff[idx], n) + cols)  # result [5, B, L] -> rows of the [n_words, 5] array
            batches.append(bt)
        return p, cfg, enc, sd, batches, [doc]
    for i in range(n_batches):
        doc = synth.make_document(enc, [c["L"]] * c["B"], seed=1000 * (rank + 1) + i, ext_word_emb=tuple(ext))
        X = col.collate_fn(doc.sentences)
        X.pop("y_upos")
        batches.append(p._prepare(X))
        docs.append(doc)
    return p, cfg, enc, sd, batches, docs


TIMED_EVERY = int(os.environ.get("CUBE_BENCH_TIMED_EVERY", "5"))   # HIP events around every n-th char-conv launch of the timed region
SETUP_STEPS = int(os.environ.get("CUBE_BENCH_SETUP_STEPS", "48"))  # untimed passes before the warm-up steps (see main)


def run_steps(p, batches, n, serial=False):
    """n steps, each: device-resident batch -> five int32 arrays on the host (`Parser.submit` /
    `collect`: the latency-bound second phase of a step overlaps the first phase of the next).
    serial (or CUBE_BENCH_SERIAL=1, the profiling aid): the same steps on ONE stream, kernels never overlap."""
    out = None
    if serial or os.environ.get("CUBE_BENCH_SERIAL") == "1":
        for i in range(n):
            r = p.decode_batch_prepared(batches[i % len(batches)])
            out = r["packed"].cpu()
        return out
    from collections import deque
    inflight = deque()
    for i in range(n):
        inflight.append(p.submit(batches[i % len(batches)]))
        if len(inflight) >= p.in_flight_for(batches[0]["T"]):  # the host runs that many batches ahead of the one it waits for
            out = p.collect(inflight.popleft())
    while inflight:
        out = p.collect(inflight.popleft())
    return out


def run_corpus_pass(p, batches, reps, ids, n_words_base):
    """Config 3, one pass over this rank's shard = `reps` repetitions of the base sample: every bucketed batch
    through submit / collect, results scattered into the shard's [n_words, 5] int32 array."""
    from collections import deque
    inflight = deque()

    def retire():
        rep, bt, ticket = inflight.popleft()
        res, status = p.collect(ticket)
        rows, cols, dest = bt["_scatter"]
        ids[rep * n_words_base + dest] = res[:, rows, cols].T
        return int((status != 0).sum())

    bad = 0
    for rep in range(reps):
        for bt in batches:
            inflight.append((rep, bt, p.submit(bt)))
            if len(inflight) >= p.in_flight_for(bt["T"]):
                bad += retire()
    while inflight:
        bad += retire()
    return bad


def cpu_baseline(cfg, enc, sd, docs, batch_size, budget_s=15.0, ext=(300,)):
    """The oracle (CPU restatement of the reference path, fp32) on a bounded sample: batches of
    `batch_size` sentences of the bench workload until ~budget_s seconds of CPU work are done."""
    from oracle import parser_oracle
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    net = parser_oracle.OracleParser(cfg, enc, sd, list(ext), quant="fp32")
    sents = [s for d in docs for s in d.sentences]
    done = 0
    t0 = time.perf_counter()
    with torch.no_grad():
        while True:
            s = done % len(sents)
            net.decode_batch(parser_oracle.collate(sents[s:s + batch_size], enc))
            done += batch_size
            dt = time.perf_counter() - t0
            if dt >= budget_s:
                break
    return dict(value=done / dt, unit="sentences/s", cores=threads, kind="port",
                sample="%d sentences x %d tokens, batch_size %d, oracle fp32 (PyTorch-CPU + NumPy CLE), %.1f s"
                       % (done, len(sents[0].words), batch_size, dt))


def end_to_end(p, enc, docs, n_target=32768):
    """Secondary number: `Parser.process(Document)` from Sentence objects to filled words — C
    featurisation, upload, the device pipeline, write-back — at batch_size 256 and at the reference's default 4
    (reference batches of equal padding geometry merged into device batches, results unchanged)."""
    from cube_amd import synth
    from cube_amd.io_utils.objects import Document
    from cube_amd.networks.utils import MorphoCollate
    col = MorphoCollate(enc)
    doc = Document()
    ext = tuple(e.shape[0] for e in docs[0].sentences[0].words[0].emb)
    L = len(docs[0].sentences[0].words)
    i = 0
    while len(doc.sentences) < n_target:
        # fresh sentences, not copies of the bench batches: repeated sentences would let `Parser.dedup_words` skip most
        # of the character CNN, which a real document does not offer to that extent
        doc.sentences.extend(synth.make_document(enc, [L] * 256, seed=500000 + i, ext_word_emb=ext).sentences)
        i += 1
    out = {}
    for bs in (256, 4):
        p.process(doc, col, batch_size=bs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.process(doc, col, batch_size=bs)
        dt = time.perf_counter() - t0
        out["batch_size_%d" % bs] = len(doc.sentences) / dt
    # the same with word frequencies as text has them: repeated words share one character-CNN row (Parser.dedup_words)
    synth.zipf_words(doc)
    p.process(doc, col, batch_size=256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p.process(doc, col, batch_size=256)
    out["batch_size_256_zipf_words"] = len(doc.sentences) / (time.perf_counter() - t0)
    out.update(unit="sentences/s", sentences=len(doc.sentences),
               what="Parser.process(Document) end to end: featurisation + upload + device + write-back, one host process; "
                    "zipf_words: the same sentences with Zipf-distributed words (repeated words share a character-CNN row)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--precision", default="")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end process(Document) secondary number")
    ap.add_argument("--no-roofline-pass", action="store_true", help="profiling aid: skip the one-stream kernel-timing pass "
                    "(the roofline then reports the events of the pipelined region)")
    ap.add_argument("--corpus-reps", type=int, default=64, help="config 3: repetitions of the 4096-sentence sample")
    ap.add_argument("--ext", default="1x300", help="external embeddings per word as KxE: 1x300 (fastText-like, the "
                    "BASELINE config) or 13x768 (xlm-roberta-base-like, SURVEY 8d secondary input)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if dist:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=device)

    c = CONFIGS[args.config]
    n_batches = 4
    k, e = args.ext.lower().split("x")
    ext = (int(e),) * int(k)
    p, cfg, enc, sd, batches, docs = build_model_and_batches(args.config, n_batches, rank, device, args.precision, ext)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    # set-up, not a step and not part of the W warm-up steps: size the grow-only workspace of every pipeline slot (7 groups:
    # with fewer passes than that the first timed steps would still allocate) and load every kernel's code object for
    # each of the distinct synthetic batches, and let the GPU leave its idle clocks (measured: the first ~25 steps after a
    # pause run 4 % slower) -- so that the K timed steps measure the path, not the box's power state.  Reported as
    # config.setup_steps.
    setup_steps = max(len(batches), SETUP_STEPS) if args.config != 3 else len(batches)
    # a generation-2 collection of the interpreter inside a 28 ms region would cost it tens of milliseconds (seen once in 15
    # regions: 48k instead of 179k sentences/s): collect now and keep the collector off until the regions are over.  BEFORE
    # the set-up steps: the collection takes ~0.1 s of host time with the GPU idle, and the first steps after such a pause
    # run on the GPU's way back up from its idle clocks (collecting right before the timed region: 180k -> 167k)
    import gc
    gc.collect()
    gc.disable()
    run_steps(p, batches, setup_steps)
    barrier()
    if args.config == 3:
        from cube_amd.sharding import gather_result_ids
        if args.corpus_reps % world:
            raise SystemExit("--corpus-reps must be a multiple of the number of ranks")
        reps = args.corpus_reps // world                      # this rank's contiguous shard of the corpus
        n_words_base = int(sum(len(s.words) for s in docs[0].sentences))
        ids = np.zeros((reps * n_words_base, 5), dtype=np.int32)
        sizes = [reps * n_words_base] * world

        def one_pass():
            bad = run_corpus_pass(p, batches, reps, ids, n_words_base)
            parts = gather_result_ids(ids, sizes, rank, world, device) if world > 1 else [ids]
            return bad, parts
        for _ in range(args.warmup):
            one_pass()
        barrier()
        p.start_kernel_timing("char_conv")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            bad, parts = one_pass()
        barrier()
        dt = time.perf_counter() - t0
        kt = p.stop_kernel_timing()
        assert bad == 0, "%d sentences without a tree" % bad
        if rank == 0:  # every shard is `reps` copies of the same sample: the gathered arrays must all agree
            for part in parts:
                assert np.array_equal(part[:n_words_base], parts[0][:n_words_base])
    else:
        run_steps(p, batches, args.warmup)
        barrier()
        # live timing of the dominant kernel inside the timed region: every 5th of its launches (4 per step: the sample walks
        # through the four layers), so that the measurement itself costs the region 0.3 % instead of 1.5 %
        p.start_kernel_timing("char_conv", every=TIMED_EVERY)
        t0 = time.perf_counter()
        run_steps(p, batches, args.steps)
        barrier()
        dt = time.perf_counter() - t0
        kt = p.stop_kernel_timing()
        # the same K steps twice more, each bracketed the same way: how far one box's timed regions scatter (`repeats`;
        # `value` stays the FIRST region's figure)
        repeat_dt = []
        for _ in range(2 if args.steps <= 200 else 0):
            barrier()
            t1 = time.perf_counter()
            run_steps(p, batches, args.steps)
            barrier()
            repeat_dt.append(time.perf_counter() - t1)
    gc.enable()
    # kernel-timing pass for the roofline: the same steps on ONE stream, so that the HIP events around a launch
    # bracket the kernel alone (in the pipelined region above they also bracket the time the launch waits for CUs
    # held by the other stages' kernels, and the kernel shares the chip with them while it runs)
    kt_alone = None
    if rank == 0 and args.config != 3 and not args.no_roofline_pass:
        n_alone = max(1, min(args.steps, 20))
        run_steps(p, batches, 2, serial=True)
        torch.cuda.synchronize()
        p.start_kernel_timing("char_conv")
        run_steps(p, batches, n_alone, serial=True)
        torch.cuda.synchronize()
        kt_alone = p.stop_kernel_timing()
    if args.config == 3:
        repeat_dt = []
    if dist:
        t = torch.tensor([dt] + repeat_dt, dtype=torch.float64, device=device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt, repeat_dt = float(t[0].item()), [float(x) for x in t[1:].tolist()]
    if rank != 0:
        if dist:
            td.destroy_process_group()
        return

    if args.config == 3:  # strong scaling: a step is one pass over the whole fixed corpus
        corpus = args.corpus_reps * len(docs[0].sentences)
        sent_per_s = args.steps * corpus / dt
    else:
        sent_per_s = args.steps * c["B"] * world / dt
    # ---- roofline of the dominant kernel: char-CNN gated conv GEMM, layers >= 1
    bt = batches[0]
    NW = bt["N"] * bt["W"]
    lay = p._lay
    kc, nf = lay.char_half + lay.lang, cfg.char_filter_size
    # algorithmic (what the reference computes, SURVEY §8d F_char): 2 * N*W rows * 5 taps * (256+64) * 512
    flops_per_launch = 2.0 * NW * 5 * kc * nf
    # executed by this implementation: ragged rows, language channels folded into the bias table, 3 bf16 passes
    rows_exec = float(np.mean(bt["char_rows"][2:]))  # output rows of layers >= 1 (tiles run over output rows)
    passes = 3 if p.precision == "bf16x3" else 1
    # int8x2 fixed point (Parser.char_int8, the default): three int8 slice products per channel pair
    i8 = "xc_q1" in p._buf._b
    peak = I8_DENSE_PEAK_TOPS if i8 else BF16_DENSE_PEAK_TFLOPS
    flops_executed = 2.0 * rows_exec * 5 * lay.char_half * nf * passes
    ms_pipelined = kt["ms_mean"]
    if kt_alone is None or not kt_alone["ms_mean"]:
        kt_alone = kt
    ms = kt_alone["ms_mean"]
    achieved = flops_per_launch / (ms * 1e-3) / 1e12 if ms else 0.0
    traffic = None
    tf = os.path.join(ROOT, "profiles", "r3_traffic_i8.json" if i8 else "r1_traffic.json")
    if i8 and not os.path.exists(tf):
        tf = os.path.join(ROOT, "profiles", "r2_traffic_i8.json")
    if args.config == 1 and world == 1 and os.path.exists(tf):  # PMC pass of this workload, see the file
        with open(tf) as f:
            traffic = json.load(f)["traffic_bytes_per_launch"]
    # `frac` describes the TIMED region (three stage streams: the launch shares the chip with the other stages' kernels and
    # its events also bracket the wait for CUs they hold); the kernel alone on the chip -- what rocprofv3 --kernel-trace of
    # the one-stream run shows -- is `alone`
    ms_region = ms_pipelined if ms_pipelined else ms
    achieved_region = flops_per_launch / (ms_region * 1e-3) / 1e12 if ms_region else 0.0
    roofline = dict(bound="mfma", achieved=achieved_region, peak=peak, unit="TFLOP/s",
                    frac=achieved_region / peak, traffic=traffic,
                    timed_every=TIMED_EVERY if args.config != 3 else 1,
                    alone=dict(ms_per_launch=ms, achieved=achieved, frac=achieved / peak, launches=kt_alone["n"],
                               note="the same launches in a pass of the same steps on ONE stream inside this run: the kernel "
                                    "alone on the chip (rocprofv3 --kernel-trace of CUBE_BENCH_SERIAL=1 bench.py: "
                                    "profiles/*_serial_*)"),
                    frac_is="ALGORITHMIC FLOPs of the reference (all N*W padded rows, 320 input channels) / launch time in the "
                            "timed region / dense peak; the matrix-core utilisation proper is executed_frac (kernel alone)",
                    executed_frac=(flops_executed / (ms * 1e-3) / 1e12 / peak) if ms else 0.0,
                    traffic_source=("PMC pass of this workload recorded in " + os.path.relpath(tf, ROOT)) if traffic else None, kernel_dtype="int8x2 (i32 accumulate)" if i8 else p.precision,
                    peak_is="dense int8 MFMA (2x bf16)" if i8 else "dense bf16 MFMA",
                    # what the chip HOLDS with nothing but MFMAs on random operands (power limit; the nominal peak is reached
                    # only with near-constant operands): scripts/micro/mfma_peak.hip, profiles/r2_mfma_peak.txt
                    sustained_peak_measured=SUSTAINED_I8_TOPS if i8 else SUSTAINED_BF16_TFLOPS,
                    executed_frac_of_sustained=(flops_executed / (ms * 1e-3) / 1e12 /
                                                (SUSTAINED_I8_TOPS if i8 else SUSTAINED_BF16_TFLOPS)) if ms else 0.0,
                    frac_of_bf16_peak=achieved / BF16_DENSE_PEAK_TFLOPS,
                    kernel="gemm_kernel<GATE%s> (char conv, layers>=1)" % (", CfgI8, int8x2" if i8 else ""),
                    launches=kt["n"], ms_per_launch=ms_region,
                    measured="HIP events around every `timed_every`-th launch of the kernel (the sample walks through its four layers), on "
                             "its stream, in the timed region; `alone`: around every launch of the one-stream pass",
                    pipelined=dict(ms_per_launch=ms_pipelined, launches=kt["n"],
                                   achieved=flops_per_launch / (ms_pipelined * 1e-3) / 1e12 if ms_pipelined else 0.0,
                                   frac=flops_per_launch / (ms_pipelined * 1e-3) / 1e12 / peak if ms_pipelined else 0.0,
                                   note="the same events inside the timed (three-stream) region: they also bracket the wait "
                                        "for CUs held by the other stages' kernels, which then share the chip with the launch"),
                    flops_per_launch=flops_per_launch, padded_W=bt["W"], rows_algorithmic=NW, rows_executed=rows_exec,
                    mfma_flops_executed_per_launch=flops_executed,
                    mfma_executed_tflops=flops_executed / (ms * 1e-3) / 1e12 if ms else 0.0,
                    )
    out = {
        "metric": "sentences/sec (node) parser+tagger inference", "value": sent_per_s, "unit": "sentences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.config == 3 else "weak", "vs_baseline": None,
        # the default precision = the mode tests/test_gpu_parity.py holds to north_star's 99.9 % on the two full-size batches
        # further regions of the same K steps in this run, same units as `value` (max over ranks each): the box's scatter
        "repeats": [args.steps * c["B"] * world / x for x in repeat_dt] if args.config != 3 else [],
        "dtype": p.precision + ("+int8x2" if i8 else ""),
        "precision": dict(mode=p.precision, char_int8=bool(p.char_int8), word_int8=bool(p.word_int8), word0_int8=bool(p.word_int8 and p.word0_int8), lstm_int8=bool(p.lstm_int8),
                          what="bf16x3 = hi/lo bf16 planes, hi*hi + hi*lo + lo*hi on the matrix cores, fp32 accumulate; int8x2 = two "
                               "int8 slices (16-bit fixed point) for the gated convolutions of layers >= 1 and for h in the BiLSTM "
                               "recurrence (W_hh 14 bits), exact int32 accumulate"),
        "data": "synthetic (SURVEY.md 8d: random-init weights seed 1234, synthetic vocabulary and sentences)",
        "config": {"workload": c["name"], "B": c["B"], "L": c["L"], "padded_W": bt["W"], "ext_emb": list(ext),
                   "sharding": "sentences sharded across ranks, no collective on the data path",
                   "setup_steps": setup_steps},
        "roofline": roofline,
    }
    if args.config == 3:
        out["config"].update(B=None, L=None, corpus_sentences=args.corpus_reps * len(docs[0].sentences),
                             sharding="one fixed corpus, contiguous shard per rank, length-bucketed batches of <= 8192 padded "
                                      "tokens, [n_words, 5] int32 results gathered on rank 0 (RCCL gather)")
    if not args.no_e2e and world == 1 and args.config == 1:
        out["secondary"] = end_to_end(p, enc, docs)
    if not args.no_cpu_baseline and world == 1:
        # the oracle at the GPU batch size and at the reference's shipped batch_size = 4 (parser.py:404, what
        # Cube.__call__ runs); its speed relative to the genuine reference: profiles/r2_ref_ratio.json
        gpu_bs = {1: 256, 2: 64, 3: 32}[args.config]
        base = cpu_baseline(cfg, enc, sd, docs, gpu_bs, budget_s=12.0, ext=ext)
        base["at_reference_batch_size_4"] = cpu_baseline(cfg, enc, sd, docs, 4, budget_s=8.0, ext=ext)
        rr = os.path.join(ROOT, "profiles", "r2_ref_ratio.json")
        if os.path.exists(rr):
            with open(rr) as f:
                rows = json.load(f)["rows"]
            base["oracle_over_genuine_reference"] = {
                "what": "wall-time ratio reference : oracle measured in the build container (8 cores, identical output), "
                        "scripts/ref_ratio.py: the oracle is FASTER than adobe/NLP-Cube's own Parser.process, so speed-ups "
                        "against it understate speed-ups against the reference",
                "min": min(r["oracle_over_reference"] for r in rows), "max": max(r["oracle_over_reference"] for r in rows)}
        out["cpu_baseline"] = base
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
