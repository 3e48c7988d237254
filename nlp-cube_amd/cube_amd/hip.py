"""ctypes binding of libcubehip.so (C ABI declared in include/cube_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).  There is no CPU
fallback: if the shared object is missing or a call fails this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcubehip.so")

EPI_F32, EPI_BF16, EPI_BF16_TANH, EPI_BF16_RELU, EPI_GATE, EPI_LSTM = 0, 1, 2, 3, 4, 5
ABI_VERSION = 19  # include/cube_hip.h; a stale libcubehip.so is refused

c_int, c_void_p, c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t


MAX_CONV_GROUPS = 32
I8_SHIFT = 8    # CUBE_I8_SHIFT
I8_ONE = 127 << I8_SHIFT  # CUBE_I8_ONE: int8x2 fixed point, x = (256*hi + lo) / I8_ONE (16 bits)
LSTM8_HONE = I8_ONE  # hidden states of cube_bilstm_recurrent_i8: the same 16-bit planes


class GatherDesc(ctypes.Structure):
    """cube_gather_desc (include/cube_hip.h): one gather of cube_gather_rows_batch."""
    _fields_ = [("dst", c_void_p), ("dst_lo", c_void_p), ("ld_dst", c_int), ("col0", c_int), ("width", c_int),
                ("table", c_void_p), ("table_lo", c_void_p), ("ld_table", c_int), ("idx", c_void_p), ("dst_rows", c_void_p),
                ("n", c_int)]


GATHER_BATCH_MAX = 8


class ConvGroups(ctypes.Structure):
    """cube_conv_groups (include/cube_hip.h): ragged sequence groups of one conv layer, host memory."""
    _fields_ = [("n", c_int)] + [(k, c_int * MAX_CONV_GROUPS) for k in
                                 ("row0_in", "n_rows", "seq_in", "row0_out", "seq_out", "word0")]


class GemmArgs(ctypes.Structure):
    _fields_ = [
        ("A", c_void_p), ("A_lo", c_void_p), ("lda", c_int), ("M", c_int), ("seq_len", c_int), ("taps", c_int),
        ("Kc", c_int),
        ("W", c_void_p), ("W_lo", c_void_p), ("N", c_int), ("bias", c_void_p),
        ("epilogue", c_int), ("n_valid", c_int), ("accumulate", c_int),
        ("out0", c_void_p), ("out0_lo", c_void_p), ("ldo0", c_int), ("col0", c_int),
        ("rowmap", c_void_p),
        ("res", c_void_p), ("ld_res", c_int), ("res_mode", c_int),
        ("out1", c_void_p), ("out1_lo", c_void_p), ("ldo1", c_int), ("col1", c_int),
        ("out2", c_void_p), ("out2_lo", c_void_p), ("ldo2", c_int), ("col2", c_int),
        ("bias_tab", c_void_p), ("row_group", c_void_p), ("n_groups", c_int),
        ("groups", ctypes.POINTER(ConvGroups)),
        ("out0_int8", c_int),
        ("batch", c_int), ("stride_A", ctypes.c_longlong), ("stride_W", ctypes.c_longlong), ("stride_bias", c_int),
        ("stride_col0", c_int),
    ]


class PackEntry(ctypes.Structure):
    """cube_pack_entry (include/cube_hip.h): one tensor or integer scalar of a packed-weights file."""
    _fields_ = [("name", ctypes.c_char_p), ("dtype", c_int), ("ndim", c_int), ("dims", ctypes.c_int64 * 8),
                ("data", c_void_p), ("nbytes", ctypes.c_uint64)]


class LemmaAttendArgs(ctypes.Structure):
    """cube_lemma_attend_args (include/cube_hip.h)."""
    _fields_ = [
        ("q", c_void_p), ("ld_q", c_int), ("taps", c_int), ("P", c_int), ("ld_tap", c_int),
        ("col_char", c_int), ("n_char", c_int), ("col_case", c_int), ("n_case", c_int),
        ("v", c_void_p), ("e_enc", c_void_p), ("ld_e", c_int),
        ("enc", c_void_p), ("enc_lo", c_void_p), ("ld_enc", c_int), ("C", c_int),
        ("B", c_int), ("S", c_int),
        ("char_emb", c_void_p), ("char_emb_lo", c_void_p), ("ld_emb", c_int), ("E", c_int),
        ("x", c_void_p), ("x_lo", c_void_p), ("ld_x", c_int), ("col_ctx", c_int), ("col_emb", c_int),
        ("step", c_int), ("attend", c_int),
        ("out_char", c_void_p), ("out_case", c_void_p), ("ld_out", c_int),
        ("eol", c_int), ("done", c_void_p), ("n_done", c_void_p),
    ]


class Pair:
    """A bf16 tensor as hi (+ optional lo) planes: x ~= hi + lo."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo=None):
        self.hi, self.lo = hi, lo

    @staticmethod
    def split(x, with_lo=True):
        """fp32 tensor -> Pair of bf16 planes (host or device)."""
        import torch
        hi = x.to(torch.bfloat16)
        lo = (x.to(torch.float32) - hi.to(torch.float32)).to(torch.bfloat16) if with_lo else None
        return Pair(hi, lo)

    def float(self):
        f = self.hi.float()
        return f if self.lo is None else f + self.lo.float()

    def to(self, device):
        return Pair(self.hi.to(device), None if self.lo is None else self.lo.to(device))


# name -> (restype, argtypes); every symbol declared in include/cube_hip.h
SIGNATURES = {
    "cube_last_error": (ctypes.c_char_p, []),
    "cube_abi_version": (c_int, []),
    "cube_gemm_bf16": (c_int, [ctypes.POINTER(GemmArgs), c_void_p]),
    "cube_gemm_i8x2": (c_int, [ctypes.POINTER(GemmArgs), c_void_p, c_void_p]),
    "cube_gather_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_int, c_void_p]),
    "cube_gather_rows_batch": (c_int, [ctypes.POINTER(GatherDesc), c_int, c_void_p]),
    "cube_tag_argmax": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cube_pack_tags": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cube_copy_to_pinned": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "cube_masked_mean": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                 c_void_p, c_int, c_void_p, c_void_p]),
    "cube_char_conv0_lookup": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGroups), c_int, c_int,
                                       c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "cube_masked_mean_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_int, c_void_p, c_void_p]),
    "cube_char_conv0_lookup_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGroups), c_int, c_int,
                                          c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "cube_ext_finish": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cube_split_f32": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "cube_packfile_open": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_void_p)]),
    "cube_packfile_count": (c_int, [c_void_p]),
    "cube_packfile_entry": (c_int, [c_void_p, c_int, ctypes.POINTER(PackEntry)]),
    "cube_packfile_find": (c_int, [c_void_p, ctypes.c_char_p, ctypes.POINTER(PackEntry)]),
    "cube_packfile_int": (c_int, [c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)]),
    "cube_packfile_close": (None, [c_void_p]),
    "cube_quant_i8x2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "cube_subword_pool": (c_int, [c_void_p, c_int, c_size_t, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                  c_void_p]),
    "cube_argmax_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cube_bilstm_recurrent": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                      c_int, c_void_p]),
    "cube_bilstm_lo_bits": (c_int, [c_int]),
    "cube_bilstm_recurrent_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_int, c_int, c_void_p]),
    "cube_biaffine_arc": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p]),
    "cube_mst_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cube_mst_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cube_init": (c_int, []),
    "cube_mst_decode_noalloc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cube_label_score": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                 c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cube_lemma_attend": (c_int, [ctypes.POINTER(LemmaAttendArgs), c_void_p]),
    "cube_lstm_cell": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                               c_void_p, c_int, c_void_p]),
}

_lib = None


class CubeHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CubeHipError("libcubehip.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the parser path)" % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so); it must be in the process
    # before this library resolves its libamdhip64 dependency, or two runtimes end up loaded and
    # device pointers from one are unknown to the other ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.cube_abi_version() != ABI_VERSION:
        raise CubeHipError("%s has ABI version %d, this package needs %d — rebuild it (__graft_entry__.build())"
                           % (LIB_PATH, lib.cube_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        raise CubeHipError("%s failed (%d): %s" % (what, code, load().cube_last_error().decode("utf-8", "replace")))


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_handle():
    """HIP stream of the CURRENT device.  Every launch path of a model runs under `on_device`, which makes
    the model's device current first: a stream of another device with this model's pointers would fault."""
    import torch
    # the raw handle straight from torch's C layer: torch.cuda.current_stream() builds a Stream object and resolves device
    # indices in Python (~10 us; a small batch asks ~20 times)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def on_device(fn):
    """Method decorator: run with the object's `_device` as the current HIP device (the reference lets a
    model live on any device, `Cube.load(lang, device='cuda:1')`, api.py:207): the stream the kernels are
    launched on, the LDS reservations of the library and torch's allocations all follow the current device."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = getattr(self, "_device", None)
        if dev is None or dev.type != "cuda":
            return fn(self, *args, **kwargs)
        import torch
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    return wrapper


def _planes(x, col=0):
    """(hi pointer, lo pointer) of a Pair / tensor / None, advanced by `col` bf16 elements."""
    if x is None:
        return None, None
    if isinstance(x, Pair):
        return x.hi.data_ptr() + 2 * col, (None if x.lo is None else x.lo.data_ptr() + 2 * col)
    return x.data_ptr() + 2 * col, None


def gemm(A, lda, M, Kc, W, N, bias, epilogue, out0=None, ldo0=0, col0=0, taps=1, seq_len=0, n_valid=0, accumulate=0,
         rowmap=None, res=None, ld_res=0, res_mode=0, out1=None, ldo1=0, col1=0, out2=None, ldo2=0, col2=0,
         a_col=0, bias_tab=None, row_group=None, groups=None, stream=None, w_scale=None, out0_int8=False,
         batch=1, stride_a=0, stride_w=0, stride_bias=0, stride_col0=0):
    """A, W, out0 (bf16 epilogues), out1, out2: `Pair`s; out0 of CUBE_EPI_F32: an fp32 tensor.
    bf16x3 is used when both A and W carry a lo plane.  a_col: first column of A (bf16 elements).
    w_scale (f32 [N]): A, W and out0 are int8x2 planes (cube_gemm_i8x2; lda, Kc, ldo0 in int8 elements).
    out0_int8: bf16 operands, but out0 receives the gate values as int8x2 planes (ldo0 in int8 elements).
    batch > 1: that many same-shape Linear problems in one launch (strides in elements, see include/cube_hip.h)."""
    # one constructor call in field order (40 attribute stores on a ctypes structure cost more than the launch itself)
    A_hi, A_lo = _planes(A, a_col)
    W_hi, W_lo = _planes(W)
    if A_lo is None or W_lo is None:
        A_lo = W_lo = None
    if epilogue == EPI_F32:
        o0, o0_lo = ptr(out0), None
    else:
        o0, o0_lo = _planes(out0)
    o1, o1_lo = _planes(out1)
    o2, o2_lo = _planes(out2)
    a = GemmArgs(A_hi, A_lo, lda, M, seq_len, taps, Kc,
                 W_hi, W_lo, N, ptr(bias),
                 epilogue, n_valid, accumulate,
                 o0, o0_lo, ldo0, col0,
                 ptr(rowmap),
                 ptr(res), ld_res, res_mode,
                 o1, o1_lo, ldo1, col1,
                 o2, o2_lo, ldo2, col2,
                 ptr(bias_tab), ptr(row_group), 0 if bias_tab is None else bias_tab.shape[0])
    if groups is not None:
        a.groups = ctypes.pointer(groups)
    if out0_int8:
        a.out0_int8 = 1
    if batch != 1:
        a.batch, a.stride_A, a.stride_W, a.stride_bias, a.stride_col0 = batch, stride_a, stride_w, stride_bias, stride_col0
    else:
        a.batch = 1
    if w_scale is not None:
        if a_col or col0:
            raise ValueError("int8x2 operands start at column 0")
        check(load().cube_gemm_i8x2(ctypes.byref(a), w_scale.data_ptr(), stream if stream is not None else stream_handle()),
              "cube_gemm_i8x2")
        return
    check(load().cube_gemm_bf16(ctypes.byref(a), stream if stream is not None else stream_handle()), "cube_gemm_bf16")


def masked_mean(layers, weights, ld_x, word_len, N, W, C, out, ld_out, word_rows=None, stream=None, i8=False):
    """layers: list of Pair [N*W, ld_x] (or ragged, with word_rows i32 [n_layers, N]); out: Pair [N, ld_out].
    i8: the layers are int8x2 planes (ld_x in int8 elements)."""
    n = len(layers)
    his = (c_void_p * n)(*[l.hi.data_ptr() for l in layers])
    los = (c_void_p * n)(*[l.lo.data_ptr() for l in layers]) if layers[0].lo is not None else None
    ws = (ctypes.c_float * n)(*weights)
    fn = load().cube_masked_mean_i8 if i8 else load().cube_masked_mean
    check(fn(his, los, ws, n, ld_x, word_len.data_ptr(), N, W, C, out.hi.data_ptr(), ptr(out.lo),
             ld_out, ptr(word_rows), stream if stream is not None else stream_handle()),
          "cube_masked_mean")
