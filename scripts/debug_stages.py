"""Stage-by-stage comparison of the HIP path with the bf16-contract oracle (debug aid, GPU box)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
from tests import cases
from oracle import parser_oracle
from cube_amd.networks.parser import Parser
from cube_amd.networks.utils import MorphoCollate

def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)), float(np.abs(a-b).mean()/max(np.abs(b).mean(),1e-30))

case = cases.FULL if len(sys.argv) < 2 or sys.argv[1] == "full" else cases.TINY
key = sys.argv[2] if len(sys.argv) > 2 else "full_8x32"
cfg, enc, sd = cases.parser_model(case)
doc = cases.document(case, enc, key)
p = Parser(cfg, enc, ext_word_emb=list(case["ext"])); p.load_state_dict(sd); p.to("cuda:0")
X = MorphoCollate(enc).collate_fn(doc.sentences); X.pop("y_upos")
r = p.decode_batch(X); torch.cuda.synchronize()
ora = parser_oracle.OracleParser(cfg, enc, sd, list(case["ext"]), quant="bf16")
torch.set_num_threads(16)
with torch.no_grad():
    (att, pre, upos, xpos, attrs, *_), inter = ora.forward(parser_oracle.collate(doc.sentences, enc), return_intermediates=True)
bt = r["bt"]; B, L, T, N, W = bt["B"], bt["L"], bt["T"], bt["N"], bt["W"]
lay = p._lay; buf = p._buf._b
def view(name, rows, ld): return buf[name][:rows*ld].view(rows, ld).float().cpu().numpy()
ch, wh, lg, P, H = lay.char_half, lay.word_half, lay.lang, lay.P, cfg.rnn_size
xw0 = view("xw0", B*T, lay.xw0_w).reshape(B, T, -1)
sl = X["x_sent_len"].numpy()
ce = parser_oracle.unpack(inter["char_emb_packed"], sl, L).numpy()
print("char_emb", rel(xw0[:, 1:, lay.xw0_char:lay.xw0_char+ch], parser_oracle.q_bf16(torch.from_numpy(ce)).numpy()))
print("ext", rel(xw0[:, 1:, lay.xw0_ext:lay.xw0_ext+P], parser_oracle.q_bf16(inter["word_emb_ext"]).numpy()))
xl0 = view("xl0", B*T, packing_w := ((wh+lg+P+31)//32*32)).reshape(B, T, -1)
print("x_parse", rel(xl0[:, :, :wh], parser_oracle.q_bf16(inter["x_parse"]).numpy()))
xm = view("xm", B*T, (wh+lg+31)//32*32).reshape(B, T, -1)
print("hidden", rel(xm[:, 1:, :wh], parser_oracle.q_bf16(inter["hidden"][:, :, :wh]).numpy()))
nl = cfg.rnn_layers
xl = view("xla" if (nl-1) % 2 == 0 else "xlb", B*T, (2*H+lg+31)//32*32).reshape(B, T, -1)
print("rnn_out", rel(xl[:, :, :2*H], parser_oracle.q_bf16(inter["rnn_out"]).numpy()))
pp = view("pp", B*T, r["pp_w"]).reshape(B, T, -1)
print("pre_parsing", rel(pp[:, :, :cfg.pre_parser_size], pre.numpy()))
print("arc", rel(r["arc_logits"].cpu().numpy(), att.numpy()))
print("upos", rel(r["upos_logits"].cpu().numpy(), upos.numpy()))
