"""Debug: forward and backward intermediates of the fused decoder layer vs the torch restatement."""
import sys, os, math
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import torch, torch.nn.functional as F
import test_gpu_dense as T
from demf_amd import fused, ops
dev = torch.device("cuda")
B, Q, H, L, P, E, Fd = [int(v) for v in sys.argv[3:10]] if len(sys.argv) > 9 else (3, 256, 8, 4, 2, 256, 1024)
shapes = ((50, 70), (25, 35), (13, 18), (7, 9))
p_attn, p_ffn = float(sys.argv[1]), float(sys.argv[2])
fused.rng_state(dev, seed=99)
c = T._make_case(B, Q, H, L, P, E, Fd, shapes, seed=B * 100 + Q)
dims = (B, Q, H, L, P, p_attn, p_ffn, 1e-5)
R = B * Q
ones = lambda n: torch.ones(n, device="cuda")
nxt = fused.peek_next_rng(dev)   # the (seed, step) the training forward below draws
masks = dict(
    attn=fused.dropout_mask(B * H * Q * Q, p_attn, fused.OP_ATTN, dev, state=nxt) if p_attn else ones(B * H * Q * Q),
    ln1=fused.dropout_mask(R * E, p_attn, fused.OP_LN1, dev, state=nxt) if p_attn else ones(R * E),
    ln2=fused.dropout_mask(R * E, p_attn, fused.OP_LN2, dev, state=nxt) if p_attn else ones(R * E),
    ffn=fused.dropout_mask(R * Fd, p_ffn, fused.OP_FFN, dev, state=nxt) if p_ffn else ones(R * Fd),
    ln3=fused.dropout_mask(R * E, p_ffn, fused.OP_LN3, dev, state=nxt) if p_ffn else ones(R * E))
prm = {k: v.clone().requires_grad_() for k, v in c["prm"].items()}
gout = T._r(R, E, seed=5)
fused._DEBUG = {}
x = c["x"].clone().requires_grad_(); pos = c["pos"].clone().requires_grad_(); pts = c["pts"].clone().requires_grad_()
out = fused.FusedDecoderLayer.apply(x, pos, pts, c["tokens"], c["keep4"], c["shapes"], c["lsi"], c["M"],
                                    c["ab"], c["vr"], dims, True, *[prm[k] for k in T.PARAM_ORDER])
out.backward(gout)
D = fused._DEBUG
# torch side with retained intermediate gradients
prm2 = {k: v.detach().clone().requires_grad_() for k, v in c["prm"].items()}
xx = c["x"].clone().requires_grad_(); pp = c["pos"].clone().requires_grad_(); pt = c["pts"].clone().requires_grad_()
Dh = E // H
keep = {}
def K(name, t):
    t.retain_grad(); keep[name] = t; return t
qkv = K("qkv", torch.cat([(xx + pp) @ prm2["in_w"][:2 * E].t() + prm2["in_b"][:2 * E], xx @ prm2["in_w"][2 * E:].t() + prm2["in_b"][2 * E:]], 1))
hd = lambda t_: t_.reshape(B, Q, H, Dh).permute(0, 2, 1, 3)
sc = K("sc", hd(qkv[:, :E]) @ hd(qkv[:, E:2 * E]).transpose(-1, -2) / math.sqrt(Dh))
pd = K("pd", torch.softmax(sc, -1) * masks["attn"].view(B, H, Q, Q))
att = K("att", (pd @ hd(qkv[:, 2 * E:])).permute(0, 2, 1, 3).reshape(R, E))
ao = K("ao", att @ prm2["out_w"].t() + prm2["out_b"])
x1 = K("x1", F.layer_norm(xx + ao * masks["ln1"].view(R, E), (E,), prm2["g1"], prm2["b1"], 1e-5))
qp = x1 + pp
raw = K("raw", torch.cat([qp @ prm2["off_w"].t() + prm2["off_b"], qp @ prm2["aw_w"].t() + prm2["aw_b"]], 1))
HLP = H * L * P
off = raw[:, :2 * HLP].reshape(B, Q, H, L, P, 2)
aw = torch.softmax(raw[:, 2 * HLP:].reshape(B, Q, H, L * P), -1).view(B, Q, H, L, P)
p4 = torch.cat([pt, torch.ones_like(pt[:, :1])], -1).view(B, Q, 4) @ c["M"].transpose(1, 2)
uv = p4[..., :2] / p4[..., 2:3]
uv = torch.clamp(uv * c["ab"][:, None, 0::2] + c["ab"][:, None, 1::2], 0, 1)
ref = uv[:, :, None] * c["vr"][:, None]
norm = torch.stack([c["shapes"][:, 1], c["shapes"][:, 0]], -1).float()
loc = K("loc", ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :])
aw = K("aw", aw)
value = (c["tokens"] @ prm2["vp_w"].t() + prm2["vp_b"]) * c["keep4"][..., :1]
mo = K("mo", ops.MultiScaleDeformableAttnFunction.apply(value.view(B, -1, H, Dh).contiguous(), c["shapes"], c["lsi"], loc.contiguous(), aw.contiguous(), 64).view(R, E))
co = K("co", mo @ prm2["op_w"].t() + prm2["op_b"])
x2 = K("x2", F.layer_norm(x1 + co * masks["ln2"].view(R, E), (E,), prm2["g2"], prm2["b2"], 1e-5))
z0 = K("z0", x2 @ prm2["f0_w"].t() + prm2["f0_b"])
hid = torch.relu(z0) * masks["ffn"].view(R, -1)
fo = K("fo", hid @ prm2["f1_w"].t() + prm2["f1_b"])
x3 = F.layer_norm(x2 + fo * masks["ln3"].view(R, E), (E,), prm2["g3"], prm2["b3"], 1e-5)
x3.backward(gout)
def cmp(n, a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    print(f"{n:8s} max|d| {float((a - b).abs().max()):.3e}  scale {float(b.abs().max()):.3e}  rel-l2 {float((a - b).norm() / b.norm()):.2e}")
cmp("out", out, x3)
cmp("df", D["df"], keep["fo"].grad)
cmp("dh", D["dh"], keep["z0"].grad)
cmp("dx2", D["dx2"], keep["x2"].grad)
cmp("dco", D["dco"], keep["co"].grad)
cmp("dmo", D["dmo"], keep["mo"].grad)
cmp("dloc", D["dloc"] + D["dloc2"], keep["loc"].grad)
cmp("dw", D["dw"] + D["dw2"], keep["aw"].grad)
cmp("draw", D["draw"], keep["raw"].grad)
cmp("dx1", D["dx1"], keep["x1"].grad)
cmp("dao", D["dao"], keep["ao"].grad)
cmp("datt", D["datt"], keep["att"].grad)
cmp("ds", D["ds"], keep["sc"].grad)
cmp("dqkv", D["dqkv"], keep["qkv"].grad)
cmp("dx", D["dx"], xx.grad)
cmp("dpos", D["dpos"], pp.grad)
cmp("dpts", pts.grad, pt.grad)
for k in T.PARAM_ORDER:
    cmp("d_" + k, prm[k].grad, prm2[k].grad)
