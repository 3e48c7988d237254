"""Dry-run checker of the host side of the C ABI (CPU only, no kernels run).

`TraceLib` stands in for libcubehip.so: every entry point records its arguments and checks that
each pointer the host hands over lies inside a live tensor and covers every byte the kernel
behind that entry point reads or writes (extents restated from include/cube_hip.h).  Index arrays
are host memory in a dry run, so gathers are checked with their real row numbers.  This is the
"operand shapes match what the kernel and its grid assume" check, run before anything goes to a
GPU.  It computes nothing: outputs stay zero, so it says nothing about values.
"""
import ctypes

import numpy as np
import torch

from cube_amd import hip


class ExtentError(AssertionError):
    pass


def _walk(obj, out):
    if isinstance(obj, hip.Pair):
        _walk(obj.hi, out)
        _walk(obj.lo, out)
    elif isinstance(obj, torch.Tensor):
        st = obj.untyped_storage()
        out.append((st.data_ptr(), st.nbytes()))
    elif isinstance(obj, dict):
        for v in obj.values():
            _walk(v, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _walk(v, out)


class TraceLib:
    """Fake library object: same attribute names as the ctypes library in cube_amd.hip."""

    def __init__(self, roots):
        self.roots = roots          # callables returning objects whose tensors are "live"
        self.calls = []
        self.extra = []             # tensors registered by the test (inputs it owns)

    # ---- registry --------------------------------------------------------------------------
    def _regions(self):
        out = []
        for r in self.roots:
            _walk(r(), out)
        _walk(self.extra, out)
        return out

    def _need(self, what, ptr, nbytes, align=1):
        if ptr is None or nbytes <= 0:
            return
        ptr = int(ptr)
        if ptr % align:
            raise ExtentError("%s: pointer not %d-byte aligned" % (what, align))
        for base, size in self._regions():
            if base <= ptr < base + size:
                if ptr + nbytes > base + size:
                    raise ExtentError("%s: needs %d bytes, only %d left in its tensor" % (what, nbytes, base + size - ptr))
                return
        raise ExtentError("%s: pointer is not inside any live tensor (freed temporary?)" % what)

    def _ints(self, what, ptr, n):
        self._need(what, ptr, 4 * n, 4)
        return np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(int(ptr))).copy()

    # ---- entry points ----------------------------------------------------------------------
    def cube_last_error(self):
        return b""

    def cube_abi_version(self):
        return hip.ABI_VERSION

    def cube_gemm_bf16(self, ref, stream):
        a = ref._obj
        self.calls.append(("gemm", a.epilogue, a.M, a.N, a.Kc, a.taps))
        M, N, Kc, taps = a.M, a.N, a.Kc, a.taps
        assert Kc % 32 == 0 and N % 128 == 0 and a.lda >= Kc and a.lda % 8 == 0
        assert taps == 1 or a.groups or (a.seq_len > 0 and M % a.seq_len == 0)
        nbat = max(a.batch, 1)  # same-shape Linear problems in one launch
        if nbat > 1:
            assert taps == 1 and a.epilogue != hip.EPI_GATE and not a.rowmap and not a.groups and not a.accumulate
            assert a.stride_A % 8 == 0 and a.stride_W % 8 == 0
        for nm, p in (("A", a.A), ("A_lo", a.A_lo)):
            self._need("gemm " + nm, p, 2 * ((nbat - 1) * a.stride_A + (M - 1) * a.lda + Kc), 16)
        for nm, p in (("W", a.W), ("W_lo", a.W_lo)):
            self._need("gemm " + nm, p, 2 * ((nbat - 1) * a.stride_W + N * taps * Kc), 16)
        gate = a.epilogue == hip.EPI_GATE
        cell = a.epilogue == hip.EPI_LSTM  # n_valid = hidden units; the bias covers every packed gate column
        nv = a.n_valid if a.n_valid > 0 else (N // 2 if gate else N)
        if cell:
            assert taps == 1 and nbat == 1 and not a.rowmap and not a.groups and not a.accumulate and not a.bias_tab
            assert 0 < nv and (nv + 15) // 16 * 64 <= N and a.res and a.out0
            # h must not land in memory the launch reads as A (other tiles may still be reading it): the two are
            # different buffers (the decoder alternates between two state buffers)
            a_lo, a_hi = a.A, a.A + 2 * ((M - 1) * a.lda + Kc)
            o_lo = a.out0 + 2 * a.col0
            o_hi = o_lo + 2 * ((M - 1) * a.ldo0 + nv)
            assert o_hi <= a_lo or a_hi <= o_lo, "LSTM epilogue writes h into the rows of A"
        self._need("gemm bias", a.bias, 4 * ((nbat - 1) * a.stride_bias + (N if gate or cell else nv)), 4)
        out_rows, n_words = M, M
        if a.groups:
            cg = a.groups.contents
            assert gate and not a.res and not a.rowmap and 1 <= cg.n <= 32
            out_rows = n_words = 0
            for g in range(cg.n):
                assert cg.seq_in[g] > 0 and cg.n_rows[g] % cg.seq_in[g] == 0 and 0 < cg.seq_out[g] <= cg.seq_in[g]
                assert cg.row0_in[g] >= 0 and cg.row0_in[g] + cg.n_rows[g] <= M
                nw = cg.n_rows[g] // cg.seq_in[g]
                out_rows = max(out_rows, cg.row0_out[g] + nw * cg.seq_out[g])
                n_words = max(n_words, cg.word0[g] + nw)
        if a.bias_tab:
            assert gate and a.row_group and taps <= 5 and a.n_groups > 0
            grp = self._ints("gemm row_group", a.row_group, n_words)
            assert grp.min() >= 0 and grp.max() < a.n_groups
            self._need("gemm bias_tab", a.bias_tab, 4 * a.n_groups * 32 * N, 4)
        rows = out_rows
        if a.rowmap:
            rm = self._ints("gemm rowmap", a.rowmap, M)
            rows = int(rm.max()) + 1
        esz = 4 if a.epilogue == hip.EPI_F32 else 2
        if a.out0_int8:  # gate values as int8x2 planes for a next layer in cube_gemm_i8x2
            assert gate and a.res and a.res_mode in (0, 1, 2) and a.out0 and a.out0_lo
            esz = 1
        last_col = a.col0 + (nbat - 1) * a.stride_col0
        assert a.ldo0 >= last_col + nv or not a.out0
        for nm, p in (("out0", a.out0), ("out0_lo", a.out0_lo)):
            self._need("gemm " + nm, p, esz * ((rows - 1) * a.ldo0 + last_col + nv), esz)
        if gate or cell:
            self._need("gemm res", a.res, 4 * ((M - 1) * a.ld_res + nv), 4)
            for nm, p, ld, c in (("out1", a.out1, a.ldo1, a.col1), ("out1_lo", a.out1_lo, a.ldo1, a.col1),
                                 ("out2", a.out2, a.ldo2, a.col2), ("out2_lo", a.out2_lo, a.ldo2, a.col2)):
                self._need("gemm " + nm, p, 2 * ((M - 1) * ld + c + nv), 2)
        return 0

    def cube_gemm_i8x2(self, ref, w_scale, stream):
        """int8x2 planes: 1 byte per element, 64-channel chunks, gate epilogue into int8 planes."""
        a = ref._obj
        self.calls.append(("gemm_i8x2", a.epilogue, a.M, a.N, a.Kc, a.taps))
        M, N, Kc, taps = a.M, a.N, a.Kc, a.taps
        assert a.epilogue == hip.EPI_GATE and a.A_lo and a.W_lo and bool(a.out0) == bool(a.out0_lo) and not a.out0_int8
        assert not a.rowmap and (a.res or (a.out0 and not a.out1 and not a.out2))
        assert Kc % 64 == 0 and N % 128 == 0 and a.lda >= Kc and a.lda % 16 == 0
        nv = a.n_valid if a.n_valid > 0 else N // 2
        assert nv % 16 == 0 and (not a.out0 or (a.ldo0 % 16 == 0 and a.col0 % 16 == 0 and a.ldo0 >= a.col0 + nv))
        if a.res:  # word-level stack: fp32 residual stream, x + res side outputs, four channels per access
            assert a.res_mode in (0, 1) and not a.groups and a.ld_res % 4 == 0 and a.ld_res >= nv
            self._need("gemm_i8x2 res", a.res, 4 * ((M - 1) * a.ld_res + nv), 16)
            for nm, p, ld, c in (("out1", a.out1, a.ldo1, a.col1), ("out1_lo", a.out1_lo, a.ldo1, a.col1),
                                 ("out2", a.out2, a.ldo2, a.col2), ("out2_lo", a.out2_lo, a.ldo2, a.col2)):
                if p:
                    assert ld % 4 == 0 and c % 4 == 0 and ld >= c + nv
                self._need("gemm_i8x2 " + nm, p, 2 * ((M - 1) * ld + c + nv), 8)
        for nm, p in (("A", a.A), ("A_lo", a.A_lo)):
            self._need("gemm_i8x2 " + nm, p, (M - 1) * a.lda + Kc, 16)
        for nm, p in (("W", a.W), ("W_lo", a.W_lo)):
            self._need("gemm_i8x2 " + nm, p, N * taps * Kc, 16)
        self._need("gemm_i8x2 w_scale", w_scale, 4 * N, 4)
        self._need("gemm_i8x2 bias", a.bias, 4 * N, 4)
        out_rows, n_words = M, M
        if a.groups:
            cg = a.groups.contents
            assert 1 <= cg.n <= 32
            out_rows = n_words = 0
            for g in range(cg.n):
                assert cg.seq_in[g] > 0 and cg.n_rows[g] % cg.seq_in[g] == 0 and 0 < cg.seq_out[g] <= cg.seq_in[g]
                assert cg.row0_in[g] >= 0 and cg.row0_in[g] + cg.n_rows[g] <= M
                nw = cg.n_rows[g] // cg.seq_in[g]
                out_rows = max(out_rows, cg.row0_out[g] + nw * cg.seq_out[g])
                n_words = max(n_words, cg.word0[g] + nw)
        else:
            assert taps == 1 or (a.seq_len > 0 and M % a.seq_len == 0)
        if a.bias_tab:
            assert a.row_group and taps <= 5 and a.n_groups > 0
            grp = self._ints("gemm_i8x2 row_group", a.row_group, n_words)
            assert grp.min() >= 0 and grp.max() < a.n_groups
            self._need("gemm_i8x2 bias_tab", a.bias_tab, 4 * a.n_groups * 32 * N, 4)
        for nm, p in (("out0", a.out0), ("out0_lo", a.out0_lo)):
            self._need("gemm_i8x2 " + nm, p, (out_rows - 1) * a.ldo0 + a.col0 + nv, 16)
        return 0

    def cube_gather_rows(self, dst, dst_lo, ld_dst, col0, width, table, table_lo, ld_table, idx, dst_rows, n, stream):
        self.calls.append(("gather", n, width))
        if n <= 0 or width <= 0:
            return 0
        assert col0 >= 0 and col0 + width <= ld_dst and width <= ld_table
        src_max = int(self._ints("gather idx", idx, n).max()) if idx else 0
        assert src_max >= 0
        if dst_rows:
            dr = self._ints("gather dst_rows", dst_rows, n)
            dmax = int(dr.max())
        else:
            dmax = n - 1
        assert dst_lo is None or table_lo is not None
        for nm, t, d in (("", table, dst), ("_lo", table_lo if dst_lo else None, dst_lo)):
            self._need("gather table" + nm, t, 2 * (src_max * ld_table + width), 2)
            self._need("gather dst" + nm, d, 2 * (dmax * ld_dst + col0 + width), 2)
        return 0

    def cube_gather_rows_batch(self, descs, n_desc, stream):
        assert 1 <= n_desc <= hip.GATHER_BATCH_MAX
        written = []
        for k in range(n_desc):
            d = descs[k]
            self.cube_gather_rows(d.dst, d.dst_lo, d.ld_dst, d.col0, d.width, d.table, d.table_lo, d.ld_table, d.idx, d.dst_rows,
                                  d.n, stream)
            written.append((d.dst, d.ld_dst, d.col0, d.col0 + d.width))
        for i in range(n_desc):   # no ordering between the gathers of a launch: same buffer -> disjoint columns
            for j in range(i):
                a_, b_ = written[i], written[j]
                assert a_[0] != b_[0] or a_[1] != b_[1] or a_[3] <= b_[2] or b_[3] <= a_[2], ("gathers overlap", i, j)
        return 0

    def cube_tag_argmax(self, x, ld, M, col_a, n_a, out_a, col_b, n_b, out_b, table, table_lo, ld_table, width, dst, dst_lo,
                        ld_dst, col_dst, stream):
        self.calls.append(("tag_argmax", M, n_a, n_b))
        assert n_a > 0 and col_a >= 0 and col_a + n_a <= ld
        self._need("tag_argmax x", x, 4 * ((M - 1) * ld + max(col_a + n_a, (col_b + n_b) if out_b else 0)), 4)
        self._need("tag_argmax out_a", out_a, 4 * M, 4)
        if out_b:
            assert n_b > 0 and col_b + n_b <= ld
            self._need("tag_argmax out_b", out_b, 4 * M, 4)
        if dst:
            assert table and width <= ld_table and col_dst + width <= ld_dst and (dst_lo is None or table_lo is not None)
            for nm, t, d in (("", table, dst), ("_lo", table_lo if dst_lo else None, dst_lo)):
                self._need("tag_argmax table" + nm, t, 2 * ((n_a - 1) * ld_table + width), 2)
                self._need("tag_argmax dst" + nm, d, 2 * ((M - 1) * ld_dst + col_dst + width), 2)
        return 0

    def cube_pack_tags(self, tags, cap, B, T, root, out, stream):
        self.calls.append(("pack_tags", B, T))
        assert cap >= B * T and root in (0, 1)
        self._need("pack_tags tags", tags, 4 * 3 * cap, 4)
        self._need("pack_tags out", out, 4 * 3 * B * (T - root), 4)
        return 0

    def cube_copy_to_pinned(self, src, dst, nbytes, stream):
        self.calls.append(("copy_to_pinned", nbytes))
        assert nbytes % 4 == 0
        self._need("copy_to_pinned src", src, nbytes, 4)
        return 0

    def cube_masked_mean_i8(self, x_hi, x_lo, weight, n_layers, ld_x, length, N, W, C, out, out_lo, ld_out, word_rows,
                            stream):
        assert x_lo is not None and C % 16 == 0 and ld_x % 16 == 0
        return self.cube_masked_mean(x_hi, x_lo, weight, n_layers, ld_x, length, N, W, C, out, out_lo, ld_out, word_rows,
                                     stream, esz=1)

    def cube_masked_mean(self, x_hi, x_lo, weight, n_layers, ld_x, length, N, W, C, out, out_lo, ld_out, word_rows,
                         stream, esz=2):
        self.calls.append(("masked_mean", n_layers, N, W, C))
        assert 0 < n_layers <= 8 and C <= ld_x and C <= ld_out
        lens = np.clip(self._ints("masked_mean len", length, N), 0, W)
        wr = self._ints("masked_mean word_rows", word_rows, n_layers * N).reshape(n_layers, N) if word_rows else None
        for k in range(n_layers):
            last = int((wr[k] + np.maximum(lens, 1) - 1).max()) if wr is not None else N * W - 1
            self._need("masked_mean x_hi[%d]" % k, x_hi[k], esz * (last * ld_x + C), 16 if esz == 1 else 2)
            if x_lo is not None:
                self._need("masked_mean x_lo[%d]" % k, x_lo[k], esz * (last * ld_x + C), 16 if esz == 1 else 2)
        for nm, p in (("out", out), ("out_lo", out_lo)):
            self._need("masked_mean " + nm, p, 2 * ((N - 1) * ld_out + C), 2)
        return 0

    def cube_char_conv0_lookup_i8(self, T, btab, cc, lang_word, groups, taps, half, n_combo, zero_code, out, out_lo, ld_out,
                                  stream):
        assert out_lo is not None
        return self.cube_char_conv0_lookup(T, btab, cc, lang_word, groups, taps, half, n_combo, zero_code, out, out_lo, ld_out,
                                           stream, esz=1)

    def cube_char_conv0_lookup(self, T, btab, cc, lang_word, groups, taps, half, n_combo, zero_code, out, out_lo, ld_out, stream,
                               esz=2):
        cg = groups.contents
        assert -1 <= zero_code < n_combo
        self.calls.append(("conv0_lookup", cg.n, half))
        assert taps in (1, 3, 5) and half % 4 == 0 and ld_out >= half and 1 <= cg.n <= 32
        total = n_in = n_words = 0
        for g in range(cg.n):
            assert cg.n_rows[g] % cg.seq_in[g] == 0 and 0 < cg.seq_out[g] <= cg.seq_in[g] and cg.row0_out[g] == total
            nw = cg.n_rows[g] // cg.seq_in[g]
            total += nw * cg.seq_out[g]
            n_in = max(n_in, cg.row0_in[g] + cg.n_rows[g])
            n_words = max(n_words, cg.word0[g] + nw)
        ids = self._ints("conv0 cc", cc, n_in)
        assert ids.min() >= 0 and ids.max() < n_combo
        lw = self._ints("conv0 lang_word", lang_word, n_words)
        self._need("conv0 T", T, 4 * taps * n_combo * 2 * half, 16)
        self._need("conv0 btab", btab, 4 * (int(lw.max()) + 1) * 32 * 2 * half, 16)
        for nm, p in (("out", out), ("out_lo", out_lo)):
            self._need("conv0 " + nm, p, esz * ((total - 1) * ld_out + half), 8)
        return 0

    def cube_ext_finish(self, acc, ld_acc, B, L, root, P, K, parts, part_stride, da, da_lo, ld_a, col_a, db, db_lo, ld_b, col_b,
                        stream):
        self.calls.append(("ext_finish", B, L, P))
        assert P <= ld_acc and col_a + P <= ld_a and (db is None or col_b + P <= ld_b)
        assert parts >= 1 and (parts == 1 or (part_stride >= P and (parts - 1) * part_stride + P <= ld_acc))
        self._need("ext_finish acc", acc, 4 * ((B * L - 1) * ld_acc + (parts - 1) * part_stride + P), 4)
        assert root in (0, 1)
        last = B * (L + root) - 1
        for nm, p, ld, c in (("a", da, ld_a, col_a), ("a_lo", da_lo, ld_a, col_a), ("b", db, ld_b, col_b),
                             ("b_lo", db_lo, ld_b, col_b)):
            self._need("ext_finish dst_" + nm, p, 2 * (last * ld + c + P), 2)
        return 0

    def cube_init(self):
        return 0

    def cube_quant_i8x2(self, x, x_lo, ld_x, rows, cols, inv_scale, q, q_lo, ld_q, stream):
        self.calls.append(("quant_i8x2", rows, cols))
        assert cols % 8 == 0 and ld_x % 8 == 0 and ld_q % 8 == 0 and cols <= ld_x and cols <= ld_q and q and q_lo
        self._need("quant_i8x2 x", x, 2 * ((rows - 1) * ld_x + cols), 16)
        if x_lo:
            self._need("quant_i8x2 x_lo", x_lo, 2 * ((rows - 1) * ld_x + cols), 16)
        self._need("quant_i8x2 inv_scale", inv_scale, 4 * cols, 16)
        self._need("quant_i8x2 q", q, (rows - 1) * ld_q + cols, 8)
        self._need("quant_i8x2 q_lo", q_lo, (rows - 1) * ld_q + cols, 8)
        return 0

    def cube_split_f32(self, x, n, hi, lo, stream):
        # x / hi / lo are allocated with exactly n elements by the caller (`_upload`) in the same statement, and
        # freed in stream order: only the contract on n and the alignments can be wrong
        self.calls.append(("split_f32", n))
        assert n % 4 == 0 and int(x) % 16 == 0 and int(hi) % 8 == 0 and (lo is None or int(lo) % 8 == 0)
        return 0

    def cube_argmax_rows(self, x, ld, col0, n, M, out, stream):
        self.calls.append(("argmax", M, n))
        assert n > 0 and col0 >= 0 and col0 + n <= ld
        self._need("argmax x", x, 4 * ((M - 1) * ld + col0 + n), 4)
        self._need("argmax out", out, 4 * M, 4)
        return 0

    def cube_bilstm_recurrent_i8(self, gx, whi, wlo, wscale, B, T, H, lo_bits, out, out_lo, ld_out, col0, stream):
        self.calls.append(("bilstm_i8", B, T, H))
        U = (H + 15) // 16
        KP = 32 * ((U + 1) // 2)
        n64, tail = KP // 64, int(KP % 64 != 0)
        assert col0 + 2 * H <= ld_out
        self._need("bilstm_i8 gx", gx, 4 * B * T * 2 * U * 64, 4)
        self._need("bilstm_i8 whi", whi, 2 * U * 4 * (n64 * 1024 + tail * 512), 16)
        assert lo_bits in (4, 6)
        tailq = (16 * U - 64 * n64 + 7) // 8 if tail else 0
        self._need("bilstm_i8 wlo", wlo, 2 * U * 4 * ((n64 * 512 + tail * 256) if lo_bits == 4 else (n64 * 768 + tailq * 128)), 16)
        self._need("bilstm_i8 wscale", wscale, 4 * 2 * U * 64, 4)
        for nm, p in (("out", out), ("out_lo", out_lo)):
            self._need("bilstm_i8 " + nm, p, 2 * ((B * T - 1) * ld_out + col0 + 2 * H), 2)
        return 0

    def cube_bilstm_recurrent(self, gx, whh, whh_lo, B, T, H, out, out_lo, ld_out, col0, stream):
        self.calls.append(("bilstm", B, T, H))
        U, KS = (H + 15) // 16, (H + 31) // 32
        assert col0 + 2 * H <= ld_out and (whh_lo is None) == (out_lo is None)
        self._need("bilstm gx", gx, 4 * B * T * 2 * U * 64, 4)
        for nm, p in (("whh", whh), ("whh_lo", whh_lo)):
            self._need("bilstm " + nm, p, 2 * 2 * U * 4 * KS * 512, 16)
        for nm, p in (("out", out), ("out_lo", out_lo)):
            self._need("bilstm " + nm, p, 2 * ((B * T - 1) * ld_out + col0 + 2 * H), 2)
        return 0

    def cube_biaffine_arc(self, u, u_lo, ld_u, h2, h2_lo, ld_h2, col_h2, D, B, T, logits, probs, stream):
        self.calls.append(("biaffine", B, T, D))
        assert D % 16 == 0 and D <= ld_u and col_h2 + D <= ld_h2 and ld_u % 8 == 0 and ld_h2 % 8 == 0 and col_h2 % 8 == 0
        for nm, p in (("u", u), ("u_lo", u_lo)):
            self._need("biaffine " + nm, p, 2 * ((B * T - 1) * ld_u + D), 16)
        for nm, p in (("h2", h2), ("h2_lo", h2_lo)):
            self._need("biaffine " + nm, p, 2 * ((B * T - 1) * ld_h2 + col_h2 + D), 16)
        for nm, p in (("logits", logits), ("probs", probs)):
            self._need("biaffine " + nm, p, 4 * B * (T - 1) * T, 4)
        return 0

    @staticmethod
    def cube_mst_workspace_bytes(B, L):
        a16 = lambda x: (x + 15) // 16 * 16  # noqa: E731
        nt = B * L
        n = L + 1
        idb = 2 if n > 255 else 1  # bytes per node id
        s_bytes, rec_bytes = 8 * n * (n | 1), a16(idb * (n * n + 3 * n + 16))
        root_blocks = min(nt, 3072 if idb == 1 else max(8, min(3072, (2 << 30) // (s_bytes + 2 * (n * n + 3 * n + 16)))))
        b = 16 + 3 * a16(4 * B) + 3 * a16(4 * nt) + 2 * a16(8 * nt) + a16(nt * n * idb)  # + bounds, first-candidate lists
        if n > 130:  # long-sentence mode: score matrices of the first pass in the workspace
            b += s_bytes * B
        if n >= 96:  # the rooted re-runs keep theirs there from 96 nodes on (one per workgroup), and their records
            b += (s_bytes + rec_bytes) * root_blocks
        if idb == 2:  # 16-bit node ids: the contraction records of the first pass too
            b += rec_bytes * B
        return b

    def cube_mst_decode(self, probs, lens, B, L, heads, status, ws, ws_bytes, stream):
        self.calls.append(("mst", B, L))
        assert L + 1 <= 2048 and ws_bytes >= self.cube_mst_workspace_bytes(B, L)
        self._need("mst probs", probs, 4 * B * L * (L + 1), 4)
        self._need("mst lens", lens, 4 * B, 4)
        self._need("mst heads", heads, 4 * B * L, 4)
        self._need("mst status", status, 4 * B, 4)
        self._need("mst workspace", ws, ws_bytes, 16)
        return 0

    def cube_label_score(self, hl, hl_lo, ld_hl, col_x2, D, wl, wl_lo, bias, n_labels, heads, lens, B, L, logits,
                         labels, stream):
        self.calls.append(("label", B, L, D, n_labels))
        assert D % 32 == 0 and D <= col_x2 and col_x2 + D <= ld_hl and ld_hl % 8 == 0 and col_x2 % 8 == 0
        T = L + 1
        for nm, p in (("hl", hl), ("hl_lo", hl_lo)):
            self._need("label " + nm, p, 2 * ((B * T - 1) * ld_hl + col_x2 + D), 16)
        for nm, p in (("wl", wl), ("wl_lo", wl_lo)):
            self._need("label " + nm, p, 2 * n_labels * D * D, 16)
        self._need("label bias", bias, 4 * n_labels, 4)
        h = self._ints("label heads", heads, B * L)
        assert h.min() >= 0 and h.max() <= L
        self._need("label lens", lens, 4 * B, 4)
        self._need("label logits", logits, 4 * B * L * n_labels, 4)
        self._need("label labels", labels, 4 * B * L, 4)
        return 0


    def cube_lemma_attend(self, ref, stream):
        a = ref._obj
        self.calls.append(("lemma_attend", a.B, a.S, a.step, a.attend))
        B, S = a.B, a.S
        assert a.step >= 0 and a.col_emb >= 0 and a.col_emb + a.E <= a.ld_x
        self._need("attend q", a.q, 4 * B * a.ld_q, 4)
        for nm, p in (("x", a.x), ("x_lo", a.x_lo)):
            self._need("attend " + nm, p, 2 * B * a.ld_x, 16)
        if a.step > 0:
            assert a.step <= a.ld_out and a.col_char + a.n_char <= a.ld_q and a.col_case + a.n_case <= a.ld_q
            assert a.E <= a.ld_emb
            for nm, p in (("out_char", a.out_char), ("out_case", a.out_case)):
                self._need("attend " + nm, p, 4 * ((B - 1) * a.ld_out + a.step), 4)
            self._need("attend done", a.done, 4 * B, 4)
            self._need("attend n_done", a.n_done, 4, 4)
            for nm, p in (("char_emb", a.char_emb), ("char_emb_lo", a.char_emb_lo)):
                self._need("attend " + nm, p, 2 * ((a.n_char - 1) * a.ld_emb + a.E), 2)   # any arg-max < n_char
        if a.attend:
            assert 0 < S <= 256 and 0 < a.P <= 128 and a.taps in (1, 3, 5) and a.P <= a.ld_tap and a.taps * a.ld_tap <= a.ld_q
            assert a.P <= a.ld_e and 0 < a.C <= 512 and a.C % 8 == 0 and a.ld_enc % 8 == 0 and a.C <= a.ld_enc
            assert a.col_ctx % 8 == 0 and a.ld_x % 8 == 0 and a.col_ctx + a.C <= a.ld_x
            self._need("attend v", a.v, 4 * a.P, 4)
            self._need("attend e_enc", a.e_enc, 4 * ((B * S - 1) * a.ld_e + a.P), 4)
            for nm, p in (("enc", a.enc), ("enc_lo", a.enc_lo)):
                self._need("attend " + nm, p, 2 * ((B * S - 1) * a.ld_enc + a.C), 16)
        return 0

    def cube_lstm_cell(self, gates, ld_g, H, c, B, h, h_lo, ld_h, col_h, c16, c16_lo, col_c, stream):
        self.calls.append(("lstm_cell", B, H))
        assert ld_g >= 4 * H and col_h >= 0 and col_h + H <= ld_h
        self._need("cell gates", gates, 4 * ((B - 1) * ld_g + 4 * H), 4)
        self._need("cell c", c, 4 * B * H, 4)
        for nm, p in (("h", h), ("h_lo", h_lo)):
            self._need("cell " + nm, p, 2 * ((B - 1) * ld_h + col_h + H), 2)
        if c16:
            assert col_c >= 0 and col_c + H <= ld_h
            for nm, p in (("c16", c16), ("c16_lo", c16_lo)):
                self._need("cell " + nm, p, 2 * ((B - 1) * ld_h + col_c + H), 2)
        return 0


def dry_lemmatizer(cfg, enc, sd, precision, monkeypatch):
    """A Lemmatizer whose device is the host and whose library is a TraceLib."""
    from cube_amd.networks import packing
    from cube_amd.networks import parser as parser_mod
    from cube_amd.networks.lemmatizer import Lemmatizer
    m = Lemmatizer(cfg, enc)
    m.precision = precision
    m.load_state_dict(sd)
    m._device = torch.device("cpu")
    m._split = precision == "bf16x3"
    m._pk = packing.to_device(m._pk_cpu, "cpu", with_lo=m._split)
    m._buf = parser_mod._Buffers("cpu", m._split)
    lib = TraceLib([lambda: m._pk, lambda: m._buf._b, lambda: getattr(m, "_live", None)])
    monkeypatch.setattr(hip, "load", lambda: lib)
    monkeypatch.setattr(hip, "stream_handle", lambda: 0)
    monkeypatch.setattr(m, "_require_gpu", lambda: None)
    return m, lib


def dry_net(cls, case_cfg, enc, sd, ext, precision, monkeypatch):
    """A Parser / Tagger whose device is the host and whose library is a TraceLib."""
    from cube_amd.networks import packing
    from cube_amd.networks import parser as parser_mod
    p = cls(case_cfg, enc, ext_word_emb=list(ext))
    p.precision = precision
    p.load_state_dict(sd)
    p._device = torch.device("cpu")
    p._split = precision == "bf16x3"
    p._pk = packing.to_device(p._pk_cpu, "cpu", with_lo=p._split)
    p._buf = parser_mod._Buffers("cpu", p._split)
    lib = TraceLib([lambda: p._pk, lambda: p._buf._b, lambda: getattr(p, "_last_bt", None)])
    monkeypatch.setattr(hip, "load", lambda: lib)
    monkeypatch.setattr(hip, "stream_handle", lambda: 0)
    monkeypatch.setattr(p, "_require_gpu", lambda: None)
    return p, lib


def dry_parser(case_cfg, enc, sd, ext, precision, monkeypatch):
    from cube_amd.networks.parser import Parser
    return dry_net(Parser, case_cfg, enc, sd, ext, precision, monkeypatch)
