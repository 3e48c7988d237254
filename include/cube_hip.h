/* libcubehip — C ABI of the MI355X (gfx950) kernels behind the NLP-Cube parser/tagger
 * inference path.  Plain C: pointers are DEVICE pointers unless a name ends in _host, sizes
 * are ints, every call is asynchronous on the given hipStream_t (passed as void*), returns
 * 0 on success or a negative code (text via cube_last_error()).  The library allocates
 * nothing: the caller owns every buffer, including the workspaces.
 *
 * The reference has no FFI; each entry point replaces PyTorch/NumPy calls the reference
 * makes on this path (file:line under the reference checkout):
 *
 *   cube_gemm_bf16          torch.conv1d + BatchNorm1d(eval) + tanh*sigmoid gating
 *                           (cube/networks/modules.py:334-344,462-479; parser.py:161-176),
 *                           nn.Linear (+tanh/relu) (modules.py:28-52; parser.py:181-203;
 *                           modules.py:568-579), LSTM input projections (parser.py:71-72,201)
 *   cube_gather_rows        nn.Embedding lookups + torch.cat (modules.py:451-457;
 *                           parser.py:140-154), unpack() (utils.py:17-32)
 *   cube_masked_mean        WordGram masked mean (modules.py:480-483)
 *   cube_char_conv0_lookup  WordGram embeddings + first ConvNorm/BatchNorm/gate (modules.py:451-472)
 *   cube_split_f32          the fp32 -> operand-plane cast of the external embeddings (parser.py:130-136)
 *   cube_ext_finish         word_emb_ext / K, tanh (parser.py:137-138,195-197; tagger.py:118-119)
 *   cube_argmax_rows        torch.argmax over tag/label logits (parser.py:186,428-431)
 *   cube_subword_pool       sub-word -> word averaging of LMHelperHF._compute_we / apply (lm.py:183-208)
 *   cube_bilstm_recurrent[_i8]  torch.lstm recurrence, 1 layer, both directions (parser.py:201)
 *   cube_biaffine_arc       PairwiseBilinear bmm + torch.softmax (modules.py:518-533;
 *                           parser.py:214-216,425)
 *   cube_mst_decode         ChuLiuEdmondsDecoder.decode (cube/networks/utils.py:565-577,
 *                           520-563, 445-518, 404-443)
 *   cube_label_score        Parser._get_labels gather + nn.Bilinear (parser.py:218-233;
 *                           modules.py:536-547)
 *   cube_lemma_attend       Lemmatizer greedy step: torch.argmax + _char_emb feedback
 *                           (cube/networks/lemmatizer.py:160-167), Attention.forward
 *                           (modules.py:145-169), context bmm (lemmatizer.py:144-145)
 *   cube_lstm_cell          one time step of the decoder nn.LSTM cell (lemmatizer.py:152-154)
 */
#ifndef CUBE_HIP_H
#define CUBE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CUBE_OK 0
#define CUBE_ERR_ARG (-1)      /* bad argument / shape the kernels do not support   */
#define CUBE_ERR_ALIGN (-2)    /* pointer or leading dimension not 16-byte aligned  */
#define CUBE_ERR_LAUNCH (-3)   /* HIP reported a launch error                       */

const char* cube_last_error(void);
int cube_abi_version(void);

/* ---- fused bf16 MFMA GEMM / implicit Conv1d(k=taps) ------------------------------------
 * C[r, n] = bias[n] + sum_{tap, c} A[r + tap - taps/2, c] * Wt[n, tap, c]
 * where a row r belongs to the sequence r / seq_len and taps outside it read zeros
 * (Conv1d zero padding).  A: bf16 [M, lda]; W: bf16 [N, taps*Kc], n-major, K laid out in
 * 32-channel chunks with the taps of a chunk adjacent: W[n, (kc*taps + tap)*32 + c32] =
 * Wt[n, tap, kc*32 + c32]; Kc % 32 == 0, N % 128 == 0, lda % 8 == 0.  fp32 accumulation. */
enum {
  CUBE_EPI_F32 = 0,        /* out0 f32 [M, ldo0]: cols col0..col0+n_valid = C (+= if accumulate)  */
  CUBE_EPI_BF16 = 1,       /* out0 bf16 = C                                                      */
  CUBE_EPI_BF16_TANH = 2,  /* out0 bf16 = tanh(C)                                                */
  CUBE_EPI_BF16_RELU = 3,  /* out0 bf16 = relu(C)                                                */
  CUBE_EPI_GATE = 4,       /* gated conv layer, see below                                        */
  CUBE_EPI_LSTM = 5        /* one LSTM cell step on the gate pre-activations, see below          */
};
/* Precision.  Every bf16 tensor may come as a pair of planes, x ~= hi + lo ("*_lo" pointers,
 * same shape and leading dimension as the hi plane).  With A_lo and W_lo given ("bf16x3")
 * each product is hi*hi + hi*lo + lo*hi on the matrix cores: fp32-grade results, 3 MFMA
 * passes.  With both NULL ("bf16") one pass.  bf16 outputs are written as hi (and, when the
 * matching *_lo pointer is given, lo = bf16(v - hi)).
 *
 * CUBE_EPI_GATE: N = 2*half output channels packed so that column
 *   n = nb*128 + wc*64 + part*32 + j  holds channel (part ? half : 0) + nb*64 + wc*32 + j;
 * g = tanh(C_a) * sigmoid(C_b) per channel ch.  Then
 *   res (optional) res_mode 0: res[r,ch]  = g     1, 3: res[r,ch] += g     2: res[r,ch] += 2g
 *   out0 (bf16, optional): [row, col0+ch] = g     (res_mode 3: = res_after, the running sum — the
 *                          tokenizer feeds every layer the sum of all previous gates, tokenizer.py:116-122)
 *   out1/out2 (bf16, optional): [row, col+ch] = res_after + g   (x + res of the reference)
 *
 * Broadcast input channels folded into the bias (CUBE_EPI_GATE only).  The reference
 * concatenates the language embedding to every position of a sequence before each Conv1d
 * (modules.py:470, parser.py:165-175): those channels are constant along the sequence, so their
 * contribution to C[r, n] depends only on the row's language and on WHICH taps of r fall inside
 * its sequence.  With bias_tab / row_group given, bias[n] is replaced by
 *   bias_tab[(row_group[r] * 32 + tapmask(r)) * N + n],   tapmask bit t = tap t is inside,
 * computed at load time (bias + sum over inside taps of W_lang[n, tap, :] . emb[group]), and the
 * GEMM runs over the remaining channels only.  bias_tab f32 [n_groups, 32, N]; row_group i32 [M]. */

/* CUBE_EPI_LSTM (cube_gemm_bf16, one plain Linear problem: taps 1, no rowmap / groups / accumulate / batch): one step of
 * a torch.nn.LSTM layer for M independent rows -- the decoder of the lemmatizer, lemmatizer.py:131,149 (self._decoder,
 * two layers, one time step per call).  A[r, :] = [x_t | h_{t-1}] of row r, W = [W_ih | W_hh] with the 4*units gate
 * rows (torch order i, f, g, o) packed so that the four pre-activations of a hidden unit meet in two neighbouring lanes:
 *   column n = ub*64 + part*32 + 2*j + odd  holds gate (2*part + odd) of unit ub*16 + j      (j < 16, part, odd < 2),
 * N >= ceil(units / 16) * 64 rounded up to a multiple of 128, bias = b_ih + b_hh in the same order, n_valid = units.
 *   res  f32 [M, ld_res]: the cell state, read as c_{t-1} and overwritten with c_t = f*c + i*g
 *   out0 bf16 [M, ldo0] (hi, optional lo): [r, col0 + u] = h_t = o * tanh(c_t)
 *   out1 bf16 (optional): [r, col1 + u] = c_t (the attention query of lemmatizer.py:144 reads the cell state)
 * out0 must not overlap the columns of A the launch reads (other workgroups may still be reading them): a caller that
 * feeds h back keeps two activation buffers and alternates.  Same bits as CUBE_EPI_F32 followed by cube_lstm_cell.  */

/* Ragged sequence groups (CUBE_EPI_GATE without res / rowmap).  The rows of A are n groups; group
 * g owns rows [row0_in, row0_in + n_rows) as sequences of seq_in rows (n_rows % seq_in == 0; its
 * sequences are words word0, word0 + 1, ... for row_group, which is then indexed per WORD).
 * All seq_in positions are computed with Conv1d zero padding at both ends of the sequence; the
 * first seq_out <= seq_in positions of sequence j are written to row row0_out + j*seq_out + pos of
 * the outputs, the others are dropped.  This is how the character CNN skips the padded tail of
 * short words: layer l only needs positions < len + 2*(layers - 1 - l) (modules.py:462-481: the
 * masked mean reads positions < len of every layer, each Conv1d(k=5) looks 2 positions ahead).
 * Row tiles never straddle groups.  A HOST struct, read at launch.                              */
#define CUBE_MAX_CONV_GROUPS 32
typedef struct cube_conv_groups {
  int n;
  int row0_in[CUBE_MAX_CONV_GROUPS], n_rows[CUBE_MAX_CONV_GROUPS], seq_in[CUBE_MAX_CONV_GROUPS];
  int row0_out[CUBE_MAX_CONV_GROUPS], seq_out[CUBE_MAX_CONV_GROUPS], word0[CUBE_MAX_CONV_GROUPS];
} cube_conv_groups;

typedef struct cube_gemm_args {
  const void* A; const void* A_lo; int lda; int M; int seq_len; int taps; int Kc;
  const void* W; const void* W_lo; int N; const float* bias;
  int epilogue; int n_valid; int accumulate;
  void* out0; void* out0_lo; int ldo0; int col0;
  const int32_t* rowmap;     /* optional: output row = rowmap[r]; negative = drop the row */
  float* res; int ld_res; int res_mode;
  void* out1; void* out1_lo; int ldo1; int col1;
  void* out2; void* out2_lo; int ldo2; int col2;
  const float* bias_tab; const int32_t* row_group; int n_groups;
  const cube_conv_groups* groups;   /* optional, host memory */
  int out0_int8;             /* cube_gemm_bf16, gate epilogue with res: out0 / out0_lo are int8x2 planes of the gate value
                                (ldo0, col0 in bytes) for a next layer that runs in cube_gemm_i8x2 */
  /* batch > 1 (cube_gemm_bf16, Linear layers: taps 1, no gate / rowmap / groups): `batch` problems of the same shape in
   * ONE launch -- problem z reads A + z*stride_A, W + z*stride_W (elements, both planes), bias + z*stride_bias and writes
   * out0 at column col0 + z*stride_col0.  Several GEMMs that each fill a fraction of the chip (the first layers of the K
   * external-embedding MLPs, parser.py:130-138) then run on the large tiles as one grid.  0 and 1 mean one problem. */
  int batch; long long stride_A; long long stride_W; int stride_bias; int stride_col0;
} cube_gemm_args;
int cube_gemm_bf16(const cube_gemm_args* args, void* stream);

/* dst[dst_rows ? dst_rows[i] : i, col0 .. col0+width) = table[idx ? idx[i] : 0, 0..width)   (bf16;
 * the lo planes, when dst_lo is given, in the same launch)                                   */
int cube_gather_rows(void* dst, void* dst_lo, int ld_dst, int col0, int width, const void* table, const void* table_lo,
                     int ld_table, const int32_t* idx, const int32_t* dst_rows, int n, void* stream);

/* Up to CUBE_GATHER_BATCH_MAX cube_gather_rows in ONE launch (the embedding look-ups and concatenations of one stage that
 * depend on the batch's index arrays only: parser.py:140-156,172-175,180-183).  descs: HOST array.  The gathers must not
 * write the same elements (no ordering between them).                                                               */
#define CUBE_GATHER_BATCH_MAX 8
typedef struct {
  void* dst; void* dst_lo; int ld_dst; int col0; int width;
  const void* table; const void* table_lo; int ld_table;
  const int32_t* idx; const int32_t* dst_rows; int n;
} cube_gather_desc;
int cube_gather_rows_batch(const cube_gather_desc* descs, int n_desc, void* stream);

/* Tag decisions of a row in one launch (parser.py:186-192, 428-431): out_a[r] = first maximum of x[r, col_a .. col_a+n_a),
 * out_b[r] likewise for a second range (out_b may be NULL), and -- when dst is given -- the embedding row table[out_a[r]]
 * (width values, bf16 planes) into dst[r, col_dst .. col_dst+width): UPOS arg-max + _upos_emb look-up (parser.py:186-190),
 * or the ATTRS and XPOS arg-maxes over the two halves of one logits matrix.                                          */
int cube_tag_argmax(const float* x, int ld, int M, int col_a, int n_a, int32_t* out_a, int col_b, int n_b, int32_t* out_b,
                    const void* table, const void* table_lo, int ld_table, int width, void* dst, void* dst_lo, int ld_dst,
                    int col_dst, void* stream);

/* tags int32 [3, cap] = (upos, attrs, xpos) decisions of B*T positions (root rows included when root = 1) ->
 * out int32 [3, B, T-root] = (upos, xpos, attrs) of the words: rows 2..4 of the result array (parser.py:428-431).    */
int cube_pack_tags(const int32_t* tags, int cap, int B, int T, int root, int32_t* out, void* stream);

/* The result arrays of a batch -> page-locked host memory (parser.py:432-447 reads them with .detach().cpu().numpy()):
 * `bytes` (a multiple of 4) from device memory `src` to `dst_pinned`, written by a kernel's own stores on `stream` instead
 * of a hipMemcpyAsync.  Why: a D->H copy that is enqueued early and waits for its batch's kernels sits in one of the
 * runtime's in-order copy queues all that time, and the H->D uploads of the NEXT batches queue up behind it (measured:
 * every second or third upload of Parser.process started only when the previous batch's decode had finished, DESIGN
 * section 9.8).  dst_pinned must be device-visible pinned memory (hipHostMalloc / torch pin_memory); it is valid on the host
 * once an event recorded on `stream` after the call has completed.                                                   */
int cube_copy_to_pinned(const void* src, void* dst_pinned, size_t bytes, void* stream);

/* WordGram tail: out[n, c] = bf16( sum_k weight[k] * sum_{p < len[n]} x_k[n*W + p, c] / max(len[n], 1) )
 * over the n_layers gated-layer outputs x_k (bf16 [N*W, ld_x], hi [+ lo] planes); x_hi / x_lo /
 * weight are HOST arrays of n_layers entries (x_lo may be NULL).  With weight = 1,..,1,2 this is
 * the reference's (x + res) masked mean, res = sum of all layers, x = the last one.
 * word_rows (optional, device i32 [n_layers, N]): first row of word n in layer k's buffer when the
 * layers are laid out in ragged sequence groups (cube_conv_groups); NULL = row n*W in every layer. */
#define CUBE_MAX_MEAN_LAYERS 8
int cube_masked_mean(const void* const* x_hi, const void* const* x_lo, const float* weight, int n_layers,
                     int ld_x, const int32_t* len, int N, int W, int C,
                     void* out, void* out_lo, int ld_out, const int32_t* word_rows, void* stream);

/* First gated layer of the character CNN as table look-ups (modules.py:451-455,462-472): its input
 * at a position is [tok_emb[char] | case_emb[case] | lang_emb[lang]], so the Conv1d(k=taps) + BN
 * pre-activation of a position is  sum over the taps inside the word of
 *   T[tap][cc[input row]][:]          T f32 [taps, n_combo, 2*half] = W_tap . [tok_emb | case_emb] per
 *                                     (character, case) combination, columns [tanh half | sigmoid half]
 * + btab[lang_word[word]][tap mask][:]  (bias + language channels, as bias_tab above, same column order)
 * and the output is tanh(.)*sigmoid(.) as bf16 hi (+ lo) rows [row, 0..half).  Rows follow `groups`
 * (cube_conv_groups of layer 0: cc is indexed by INPUT row, the output rows of the groups must be
 * contiguous from 0).  HBM/L2-bound (taps + 1 table rows per position) instead of an MFMA GEMM
 * over K = taps * (char_emb + case_emb).  zero_code: a (character, case) combination whose rows of T are
 * all zero in every tap -- the padding character, whose embedding rows are zero (modules.py:430-431,
 * padding_idx = 0) -- or -1: its rows are not read (the positions past a word's end, which layer 0 computes
 * for the layers above it, are most of the look-ups).                                              */
int cube_char_conv0_lookup(const float* T, const float* btab, const int32_t* cc, const int32_t* lang_word,
                           const cube_conv_groups* groups, int taps, int half, int n_combo, int zero_code,
                           void* out, void* out_lo, int ld_out, void* stream);

/* ---- int8x2 fixed point for the gated convolutions whose inputs are gate outputs (|x| < 1) --------
 * Same reference calls as cube_gemm_bf16 / cube_char_conv0_lookup / cube_masked_mean (modules.py:462-483:
 * character-CNN layers >= 1), other operand format.  An activation is two int8 planes,
 *   x = (256*hi + lo) / CUBE_I8_ONE,   hi in [-127, 127], lo in [-128, 127]   (|x| <= 1, step 3.1e-5: 16 bits),
 * a weight row n likewise with its own scale: w[n, k] = (256*hi + lo) * wmax[n] / CUBE_I8_ONE.  The
 * products hi*hi and hi*lo + lo*hi are accumulated EXACTLY in int32 on the matrix cores
 * (v_mfma_i32_32x32x32_i8, twice the bf16 rate; lo*lo, 2^-16 relative, is dropped) and
 *   C[r, n] = (256*HH + CROSS) * w_scale[n] + bias,    w_scale[n] = 256 * wmax[n] / CUBE_I8_ONE^2.
 * cube_gemm_i8x2: cube_gemm_args as above with A, A_lo int8 [M, lda], W, W_lo int8 [N, taps*Kc] laid out
 * in 64-channel chunks (W[n, (kc*taps + tap)*64 + c64]), Kc % 64 == 0, lda % 16 == 0, lda / Kc / ldo0 /
 * col0 counted in int8 elements; CUBE_EPI_GATE only, no rowmap; out0, out0_lo int8 planes of the gate
 * values (n_valid, ldo0, col0 multiples of 16; may both be NULL when res is given).  res (fp32 residual
 * stream, res_mode 0: res = g, 1: res += g) and the bf16 x + res side outputs out1 / out2 as for
 * cube_gemm_bf16 (the word-level stack; ld_res, ldo1/2, col1/2 multiples of 4, without sequence groups).  w_scale f32 [N] in packed column
 * order.  bias / bias_tab / row_group / groups as for cube_gemm_bf16; w_scale, bias and bias_tab 16-byte
 * aligned (the epilogue reads four floats at a time).
 * cube_char_conv0_lookup_i8 / cube_masked_mean_i8: the bf16 entry points with int8x2 planes as the
 * gate-value output / as the layer inputs (ld_out resp. ld_x in int8 elements, multiples of 16 resp. C % 16 == 0). */
#define CUBE_I8_SHIFT 8
#define CUBE_I8_ONE 32512 /* 127 << CUBE_I8_SHIFT */
int cube_gemm_i8x2(const cube_gemm_args* args, const float* w_scale, void* stream);
int cube_char_conv0_lookup_i8(const float* T, const float* btab, const int32_t* cc, const int32_t* lang_word,
                              const cube_conv_groups* groups, int taps, int half, int n_combo, int zero_code,
                              void* out, void* out_lo, int ld_out, void* stream);
int cube_masked_mean_i8(const void* const* x_hi, const void* const* x_lo, const float* weight, int n_layers,
                        int ld_x, const int32_t* len, int N, int W, int C,
                        void* out, void* out_lo, int ld_out, const int32_t* word_rows, void* stream);

/* bf16 hi (+ lo) rows -> int8x2 planes with one scale per COLUMN: q[r, c] = round(clamp(x[r, c] * inv_scale[c], -1, 1) *
 * CUBE_I8_ONE) = 256*hi + lo.  Input of a cube_gemm_i8x2 layer whose activations are not gate outputs -- the first
 * word-level gated convolution reads the concatenation [word emb | char vector | lang emb | ext] (parser.py:149-158,
 * tagger.py:121-129): the packer takes 1 / inv_scale[c] = the largest magnitude the column's block can hold (the embedding
 * table's maximum; 1 for tanh outputs; the parser's root row included) and folds it into the layer's weights.  cols, ld_x
 * (bf16 elements) and ld_q (bytes) multiples of 8; inv_scale f32 [cols], 0 for padding columns; x_lo may be NULL.      */
int cube_quant_i8x2(const void* x, const void* x_lo, int ld_x, int rows, int cols, const float* inv_scale, void* q, void* q_lo,
                    int ld_q, void* stream);

/* e = tanh(sum_j acc[r, j*part_stride + 0..P) / K)  (acc f32 [B*L, ld_acc], row r = b*L + i; j < parts, summed in
 * that order: parts = 1 when the GEMMs accumulated in place, parts = K after one batched launch of the K second layers);
 * bf16(e) -> dst_a[(b*(L+root) + i + root), col_a..+P) and dst_b[same row, col_b..+P);
 * root = 1: the destination keeps a root row in front of every sentence (parser.py:195-197),
 * root = 0: it does not (tagger.py:111-119).  dst_b may be NULL.                            */
int cube_ext_finish(const float* acc, int ld_acc, int B, int L, int root, int P, int K, int parts, int part_stride,
                    void* dst_a, void* dst_a_lo, int ld_a, int col_a,
                    void* dst_b, void* dst_b_lo, int ld_b, int col_b, void* stream);

/* hi[i] = bf16(x[i]) (round to nearest even), lo[i] = bf16(x[i] - hi[i]) (lo may be NULL): the hi/lo planes of
 * an fp32 tensor — the external embeddings of a batch (parser.py:130-131: x_word_embeddings arrive as fp32
 * from the LM helper, lm.py:199-208).  n % 4 == 0.                                                        */
int cube_split_f32(const float* x, size_t n, void* hi, void* lo, void* stream);

/* Sub-word piece -> word pooling of the external-embedding stage: LMHelperHF._compute_we (lm.py:183-197) and the cut
 * into per-layer vectors of LMHelperHF.apply (lm.py:199-208), on hidden states that are on the device.
 * hidden: HOST array of n_layers device pointers (<= CUBE_POOL_MAX_LAYERS), layer k f32 [sentences, pieces, E] with
 * strides stride_s / stride_p (floats); word w covers the word_n[w] consecutive pieces from (word_sent[w], word_pos[w]).
 * out f32 [n_layers, n_words, E]: n = 0 -> zeros; n = 1 -> the piece; n >= 2 -> ((h0 + h0) + h1 + ... + h(n-2)) / n --
 * the reference's loop adds the first piece to itself in place and never reaches the last one.  fp32, additions in
 * that order, IEEE division: bit-equal to the reference's NumPy arithmetic.  E, strides % 4 == 0.             */
#define CUBE_POOL_MAX_LAYERS 32
int cube_subword_pool(const void* const* hidden, int n_layers, size_t stride_s, size_t stride_p, int E,
                      const int32_t* word_sent, const int32_t* word_pos, const int32_t* word_n, int n_words,
                      float* out, void* stream);

/* out[r] = first index of the maximum of x[r, col0 .. col0+n)   (x f32 [M, ld])             */
int cube_argmax_rows(const float* x, int ld, int col0, int n, int M, int32_t* out, void* stream);

/* One bidirectional LSTM layer's recurrence over a padded batch.
 * gx f32 [B*T, 2*G] (G = 64*U, U = ceil(H/16)): input projections + both biases, gate columns
 *   packed as dir*G + u*64 + gate*16 + j for hidden unit u*16+j (gate order i,f,g,o).
 * whh bf16 [2][U][4][KS][64][8] (KS = ceil(H/32)): recurrent weights in MFMA-fragment order:
 *   element [d][u][g][ks][lane][j] = W_hh_d[g*H + u*16 + (lane&15)][ks*32 + (lane>>4)*8 + j] (0 beyond H).
 * out bf16 [B*T, ld_out]: h_fwd -> cols [col0, col0+H), h_bwd -> [col0+H, col0+2H).        */
int cube_bilstm_recurrent(const float* gx, const void* whh, const void* whh_lo, int B, int T, int H,
                          void* out, void* out_lo, int ld_out, int col0, void* stream);

/* The same recurrence (torch.lstm, parser.py:71-72,201) with the hidden-state weights RESIDENT in VGPRs and LDS and
 * the product h_{t-1} . W_hh^T in two-slice fixed point on the int8 matrix cores: h = (256*hi + lo) / CUBE_I8_ONE
 * (|h| < 1, 16 bits), a weight row n with its own scale, an int8 high slice and a low slice of lo_bits = 6 or 4 bits:
 * w[n, k] = (2^lo_bits * hi + lo) * wmax[n] / (127 * 2^lo_bits); hi*hi, hi*lo and lo*hi accumulated exactly in int32
 * (lo*lo dropped); gates, cell state and the input projections stay fp32.  cube_bilstm_lo_bits(H) = the widest low
 * slice that fits the LDS of a CU for this rnn_size (6 up to H = 208, else 4): what the packer should produce.
 * gx, out, out_lo, ld_out, col0 as for cube_bilstm_recurrent (out_lo may be NULL).  KP = H rounded up to 32,
 * n64 = KP / 64, tail = KP % 64 != 0:
 *   whi   int8 [2][U][4][n64*1024 + tail*512]: fragments [lane][16] of hi[row u*64 + g*16 + (lane & 15)]
 *         [ks*64 + (lane >> 4)*16 + j], then the tail [lane][8] at k = n64*64 + (lane >> 4)*8 + j
 *   wlo   the unsigned values v = lo + 2^(lo_bits-1) of the same fragments, packed:
 *         lo_bits 4: uint8 [2][U][4][n64*512 + tail*256], two to a byte: 32-bit word m of a lane, byte b =
 *                    v[k0 + 8m + b] | v[k0 + 8m + 4 + b] << 4
 *         lo_bits 6: uint8 [2][U][4][n64*768 + tailq*128], a lane's 16 values in three 32-bit words: byte b of word m
 *                    (m < 3) = v[4m + b] | ((v[12 + b] >> 2m) & 3) << 6; per fragment words 0, 1 of all lanes
 *                    ([lane][8]), then word 2 of all lanes ([lane][4]); the tail as whole bytes [lane][8] for the
 *                    first tailq = ceil((16U - 64*n64) / 8) groups of 16 lanes only (the others meet h = 0)
 *   wscale f32 [2][U*64]: wmax[n] / (CUBE_I8_ONE * 127) in packed gate-row order.
 * Both tensors gx and out must be smaller than 2 GiB (32-bit lane offsets).                                */
int cube_bilstm_lo_bits(int H);
int cube_bilstm_recurrent_i8(const float* gx, const void* whi, const void* wlo, const float* wscale, int B, int T, int H,
                             int lo_bits, void* out, void* out_lo, int ld_out, int col0, void* stream);

/* Arc scores + row softmax.  u bf16 [B*T, ld_u] (= [h1;1] W), h2 bf16 [B*T, ld_h2] (cols
 * col_h2..col_h2+D), D % 32 == 0.  logits/probs f32 [B, T-1, T]: [b,i,j] = u[b,i+1,:].h2[b,j,:],
 * probs = softmax over all T columns (pads included, as the reference does).                */
int cube_biaffine_arc(const void* u, const void* u_lo, int ld_u, const void* h2, const void* h2_lo, int ld_h2,
                      int col_h2, int D, int B, int T, float* logits, float* probs, void* stream);

/* Chu-Liu-Edmonds single-root decoding.  probs f32 [B, L, L+1], lens int32 [B] ->
 * heads int32 [B, L] (0 at pads), status int32 [B] (0 ok, 1 = no tree: the reference asserts,
 * 2 = the scores are not probabilities (NaN / negative entries made a contraction that does not
 * shrink the graph): the kernel stops instead of looping).  L <= 2047 (the reference has no limit;
 * up to 129 words the score matrix of a task lives in LDS, beyond in the workspace, from 255 words
 * with 16-bit node ids).  workspace: cube_mst_workspace_bytes(B, L) bytes, 16-byte aligned.     */
size_t cube_mst_workspace_bytes(int B, int L);
int cube_mst_decode(const float* probs, const int32_t* lens, int B, int L,
                    int32_t* heads, int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* cube_mst_decode for callers that CANNOT ALLOCATE (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results as
 * cube_mst_decode without the workspace: the task list, candidate trees and scores live in a built-in
 * device array of CUBE_MST_NOALLOC_BYTES that every GPU the library runs on holds until the process
 * ends.  Calls with cube_mst_workspace_bytes(B, L) > CUBE_MST_NOALLOC_BYTES are refused (CUBE_ERR_ARG):
 * that covers B*L <= 8192 padded tokens up to L = 94 (longer sentences keep their score matrices in the
 * workspace and need cube_mst_decode).  The call allocates and frees nothing.  All calls on one GPU share
 * that GPU's array: calls that overlap on one GPU (two streams, two threads) race — serialise them.
 * cube_init() resolves the array's address for the CURRENT device (once per device; the first
 * cube_mst_decode_noalloc on a device does the same if it was not called — call it before a graph
 * capture starts).                                                                               */
#define CUBE_MST_NOALLOC_BYTES (4u << 20)
int cube_init(void);
int cube_mst_decode_noalloc(const float* probs, const int32_t* lens, int B, int L,
                            int32_t* heads, int32_t* status, void* stream);

/* Label scores for the decoded heads.  hl bf16 [B*T, ld_hl]: x1 = cols [0, D), x2 = cols
 * [col_x2, col_x2+D) (each with its constant-1 column inside D), D % 32 == 0.
 * wl bf16 [n_labels][D][D] ([o][q][p] = W[o,p,q]); bias f32 [n_labels];
 * logits f32 [B, L, n_labels]: [b,i,o] = x1[b,i+1]^T W[o] x2[b, head] + bias[o], head =
 * heads[b,i] if i < lens[b] else 0.  labels int32 [B, L] = argmax_o.                        */
int cube_label_score(const void* hl, const void* hl_lo, int ld_hl, int col_x2, int D,
                     const void* wl, const void* wl_lo, const float* bias,
                     int n_labels, const int32_t* heads, const int32_t* lens, int B, int L,
                     float* logits, int32_t* labels, void* stream);

/* ---- lemmatizer decoder ---------------------------------------------------
 * One greedy decoding step for B words between the GEMMs of the step.  q f32 [B, ld_q] is the output
 * of the per-step "query" GEMM: columns [t*ld_tap, t*ld_tap + P) hold W_att[:, query channels, tap t]
 * . query (the query is the top layer's CELL state, lemmatizer.py:144), columns [col_char, +n_char)
 * and [col_case, +n_case) the character / case logits of the PREVIOUS decoder step.
 *   step > 0: out_char/out_case[b, step-1] = first arg-max of those logits; done[b] / *n_done are
 *             updated when the character is `eol`; x[b, col_emb..+E) = char_emb[arg-max].
 *   step = 0: x[b, col_emb..+E) = 0 (no previous character, lemmatizer.py:147-149).
 *   attend != 0: energy[t] = v . tanh(e_enc[b*S+t, :] + sum over taps t+k-taps/2 inside [0, S) of
 *             q[b, k*ld_tap + :]) (Conv1d(k=taps) over [query | encoder_output] with the query
 *             constant along the word, e_enc = the encoder part + bias, computed once per batch);
 *             att = softmax over all S positions; x[b, col_ctx..+C) = sum_t att[t] * enc[b*S+t, 0..C).
 * enc, char_emb, x: bf16 hi (+ lo) planes.  S <= 256, P <= 128, C <= 512, C % 8 == 0.              */
typedef struct cube_lemma_attend_args {
  const float* q; int ld_q; int taps; int P; int ld_tap;
  int col_char; int n_char; int col_case; int n_case;
  const float* v; const float* e_enc; int ld_e;
  const void* enc; const void* enc_lo; int ld_enc; int C;
  int B; int S;
  const void* char_emb; const void* char_emb_lo; int ld_emb; int E;
  void* x; void* x_lo; int ld_x; int col_ctx; int col_emb;
  int step; int attend;
  int32_t* out_char; int32_t* out_case; int ld_out;
  int eol; int32_t* done; int32_t* n_done;
} cube_lemma_attend_args;
int cube_lemma_attend(const cube_lemma_attend_args* args, void* stream);

/* LSTM cell: gates f32 [B, ld_g] = W_ih x + W_hh h + b, columns [i | f | g | o] of H each (PyTorch
 * order); c f32 [B, H] updated in place; h = o * tanh(c) -> bf16 hi (+ lo) at h[b, col_h..+H) (row
 * stride ld_h); optionally c as bf16 hi (+ lo) at c16[b, col_c..+H) (same stride).               */
int cube_lstm_cell(const float* gates, int ld_g, int H, float* c, int B, void* h, void* h_lo, int ld_h, int col_h,
                   void* c16, void* c16_lo, int col_c, void* stream);

/* ---- packed weights from a file: for hosts that cannot run the Python packer --------------------------------------
 * The weight operands of every entry point above are LAYOUTS (BatchNorm folded in, gate-column order, chunked K, int8x2
 * slices + scales, bias tables, LSTM gate-row orders, W_hh in fragment order) that cube_amd/networks/packing.py builds
 * from a reference state_dict (the reference keeps nn.Module parameters, parser.py:23-95; there is no counterpart of a
 * packed file in it).  `Parser.save_packed(path)` / `packing.save_packed` write them once, offline; these host-only calls
 * map such a file and hand out every tensor by name -- e.g. "word_convs.1.W8.hi", "word_convs.1.w_scale",
 * "char_convs.2.bias_tab" -- with its type and shape, and the integers that go into the argument structs -- "word_convs.1.Kc",
 * "word_convs.1.N", "layout.xw0_w" ... (names: the keys of the packer's dict joined by dots, list indices as numbers,
 * `hi` / `lo` for the planes of a bf16 or int8x2 pair, `layout.*` for the column offsets of the activation buffers).
 * The host copies `data` (valid until cube_packfile_close, 256-byte aligned in the file) to device memory it owns.
 * cube_packfile_open allocates the handle (the only allocation of this library that outlives a call);
 * cube_packfile_close frees it.  No GPU is touched.                                                                    */
enum { CUBE_PACK_F32 = 0, CUBE_PACK_BF16 = 1, CUBE_PACK_I8 = 2, CUBE_PACK_I32 = 3, CUBE_PACK_I64 = 4, CUBE_PACK_U8 = 5 };
#define CUBE_PACK_MAX_DIMS 8
typedef struct cube_pack_entry {
  const char* name;      /* owned by the handle */
  int dtype; int ndim;   /* CUBE_PACK_*; ndim 0 = scalar */
  int64_t dims[CUBE_PACK_MAX_DIMS];
  const void* data;      /* host pointer into the mapped file */
  uint64_t nbytes;
} cube_pack_entry;
typedef struct cube_packfile cube_packfile;
int cube_packfile_open(const char* path, cube_packfile** out);
int cube_packfile_count(const cube_packfile* f);
int cube_packfile_entry(const cube_packfile* f, int i, cube_pack_entry* out);
int cube_packfile_find(const cube_packfile* f, const char* name, cube_pack_entry* out);
int cube_packfile_int(const cube_packfile* f, const char* name, long long* value);
void cube_packfile_close(cube_packfile* f);

#ifdef __cplusplus
}
#endif
#endif
