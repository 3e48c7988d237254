"""Packed weights as a file (include/cube_hip.h: cube_packfile_*, nlp-cube_amd/csrc/packfile.hip): what a host that cannot
run the Python packer reads.  CPU only: the reader is host code, no GPU is touched."""
import ctypes
import os
import struct

import numpy as np
import pytest
import torch

from tests import cases
from cube_amd import hip
from cube_amd.networks import packing

NP_DTYPES = {0: np.float32, 1: np.int16, 2: np.int8, 3: np.int32, 4: np.int64, 5: np.uint8}


def _open(path):
    lib = hip.load()
    h = ctypes.c_void_p()
    rc = lib.cube_packfile_open(path.encode(), ctypes.byref(h))
    return lib, h, rc


def _entries(lib, h):
    out = {}
    for i in range(lib.cube_packfile_count(h)):
        e = hip.PackEntry()
        assert lib.cube_packfile_entry(h, i, ctypes.byref(e)) == 0
        shape = tuple(e.dims[d] for d in range(e.ndim))
        buf = ctypes.string_at(e.data, e.nbytes)
        assert e.data % 256 == 0 or e.nbytes == 0
        out[e.name.decode()] = (e.dtype, np.frombuffer(buf, dtype=NP_DTYPES[e.dtype]).reshape(shape))
    return out


@pytest.mark.parametrize("which", ["parser", "tagger", "lemmatizer", "tokenizer"])
def test_packed_file_holds_the_packers_bytes(tmp_path, which):
    """Every tensor and integer of the packer's output comes back from the C reader under its dotted name with the same
    type, shape and bytes; the layout's column offsets ride along."""
    if which == "parser":
        from cube_amd.networks.parser import Parser
        cfg, enc, sd = cases.parser_model(cases.TINY)
        m = Parser(cfg, enc, ext_word_emb=list(cases.TINY["ext"]))
    elif which == "tagger":
        from cube_amd.networks.tagger import Tagger
        cfg, enc, sd = cases.tagger_model(cases.TINY)
        m = Tagger(cfg, enc, ext_word_emb=list(cases.TINY["ext"]))
    elif which == "lemmatizer":
        from cube_amd.networks.lemmatizer import Lemmatizer
        cfg, enc, sd = cases.lemmatizer_model(cases.LEMMA_SMALL)
        m = Lemmatizer(cfg, enc)
    else:
        from cube_amd.networks.tokenizer import Tokenizer
        cfg, enc, sd = cases.tokenizer_model(cases.TOKENIZER_SMALL)
        m = Tokenizer(cfg, enc, ext_word_emb=list(cases.TOKENIZER_SMALL["ext"]))
    m.load_state_dict(sd)
    path = str(tmp_path / "model.cubepack")
    n = m.save_packed(path)
    lib, h, rc = _open(path)
    assert rc == 0, lib.cube_last_error()
    try:
        got = _entries(lib, h)
        want = packing.flatten_packed(m._pk_cpu, m._lay)
        assert n == len(want) == len(got) and n > 50
        for name, v in want:
            code, arr = got[name]
            if isinstance(v, int):
                assert code == 4 and arr.shape == () and int(arr) == v, name
                val = ctypes.c_longlong()
                assert lib.cube_packfile_int(h, name.encode(), ctypes.byref(val)) == 0 and val.value == v
            else:
                assert code == packing.PACK_DTYPES[v.dtype] and arr.shape == tuple(v.shape), name
                ref = (v.view(torch.int16) if v.dtype == torch.bfloat16 else v).numpy()
                assert np.array_equal(arr, ref), name
        # what a C host looks up to fill cube_gemm_args for, e.g., the second word-level gated convolution
        for key in () if which in ("lemmatizer", "tokenizer") else ("word_convs.1.W.hi", "word_convs.1.W.lo", "word_convs.1.bias", "word_convs.1.Kc", "word_convs.1.N",
                    "word_convs.1.half", "word_convs.1.taps", "layout.xw0_w", "layout.xw_w", "word_emb.hi") + \
                (("lstm.0.whh.hi", "lstm.0.whh8.hi", "lstm.0.whh8.lo", "lstm.0.whh8.scale") if which == "parser" else ()):
            e = hip.PackEntry()
            assert lib.cube_packfile_find(h, key.encode(), ctypes.byref(e)) == 0, key
        e = hip.PackEntry()
        assert lib.cube_packfile_find(h, b"no.such.entry", ctypes.byref(e)) != 0
        assert b"no.such.entry" in lib.cube_last_error()
        val = ctypes.c_longlong()
        assert lib.cube_packfile_int(h, b"word_convs.1.bias", ctypes.byref(val)) != 0    # a tensor, not a scalar
    finally:
        lib.cube_packfile_close(h)


def test_packed_file_reader_refuses_damaged_files(tmp_path):
    lib = hip.load()
    pk = {"a": torch.arange(6, dtype=torch.float32).view(2, 3), "n": 7, "p": hip.Pair.split(torch.randn(4, 8))}
    good = str(tmp_path / "good.cubepack")
    assert packing.save_packed(pk, good) == 4
    raw = open(good, "rb").read()
    lib_, h, rc = _open(good)
    assert rc == 0
    got = _entries(lib, h)
    lib.cube_packfile_close(h)
    assert sorted(got) == ["a", "n", "p.hi", "p.lo"] and np.array_equal(got["a"][1], pk["a"].numpy())

    def bad(name, data):
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        _, hh, rc_ = _open(path)
        assert rc_ != 0 and not hh.value, name
        return lib.cube_last_error()

    assert b"bad magic" in bad("magic", b"CUBEPACX" + raw[8:])
    assert b"too short" in bad("short", raw[:20])
    assert b"version" in bad("version", raw[:8] + struct.pack("<I", 2) + raw[12:])
    data_off = struct.unpack("<Q", raw[24:32])[0]
    assert bad("cut", raw[:data_off + 8])                       # the tensors run past the end of the file
    assert bad("table", raw[:16] + struct.pack("<Q", 1 << 40) + raw[24:])   # entry table larger than the file
    with pytest.raises(TypeError):                              # an operand the format cannot hold is an error, not a gap
        packing.save_packed({"x": torch.zeros(2, dtype=torch.float64)}, str(tmp_path / "f64"))
    _, hh, rc_ = _open(str(tmp_path / "does_not_exist"))
    assert rc_ != 0 and b"cannot open" in lib.cube_last_error()
    lib.cube_packfile_close(None)                               # closing nothing is allowed
