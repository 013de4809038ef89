"""One encoder layer's five launches (csrc/rows_gemm.hip + the raw-input MSDA), each timed with HIP events on the
stream it is launched on, against its own roofline: algorithmic flop / bytes (inputs read once, outputs written once,
weights ignored), TF/s of fp32-grade products (of the 416.7 TF/s the three-term mode can issue = bf16 MFMA peak / 6)
and GB/s (of 8 TB/s).  8 scenes x 18 609 tokens, embed 256, FFN 1024, P = 4.  usage: python tools/enc_layer_roofline.py [f32|bf16]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from demf_amd import ops, _ffi
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
ops.set_compute_dtype(mode)
planes = 1 if mode == "bf16" else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, shapes, C, F, H, P = 8, ((100, 140), (50, 70), (25, 35), (13, 18)), 256, 1024, 8, 4
S = sum(h * w for h, w in shapes); R = B * S
r = lambda *s: torch.randn(*s, device=dev)
x, pos, mask = r(R, C), r(R, C), torch.rand(R, device=dev) < 0.05
w_in, b_in = ops.split_planes(r(640, C) / 16, planes), r(640)
w_o, b_o, w0, b0, w1, b1 = ops.split_planes(r(C, C) / 16, planes), r(C), ops.split_planes(r(F, C) / 16, planes), r(F), \
    ops.split_planes(r(C, F) / 32, planes), r(C)
g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
raw, samp, x1, hid, xn = r(R, 640), r(R, C), r(R, C), r(R, F), r(R, C)
raw[:, :256] *= 0.5
ref = torch.rand(B, S, 4, 2, device=dev)
shp = torch.tensor(shapes, dtype=torch.long, device=dev)
lsi = torch.tensor([0, 14000, 17500, 18375], dtype=torch.long, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
rows = [
    ("rows_gemm in-proj N=640 (+pos on 384 cols, mask)", lambda: ops.rows_gemm(x, w_in, b_in, raw, a2=pos, a2_cols=384, row_mask=mask, mask_col0=384),
     2.0 * R * 256 * 640, 4.0 * R * (256 + 256 + 640)),
    ("msda_fwd_raw (one wave per query)", lambda: ops.msda_fwd_raw(raw, 384, 0, 256, ref, shp, lsi, B, S, H, C // H, P, samp),
     2.0 * R * H * 16 * 4 * 32, 4.0 * R * (640 + 256)),
    ("rows_gemm out-proj + residual + LayerNorm", lambda: ops.rows_gemm(samp, w_o, b_o, x1, ln=(x, g, be, 1e-5)),
     2.0 * R * 256 * 256, 4.0 * R * 3 * 256),
    ("rows_gemm FFN up + ReLU N=1024", lambda: ops.rows_gemm(x1, w0, b0, hid, relu=True), 2.0 * R * 256 * 1024, 4.0 * R * (256 + 1024)),
    ("rows_gemm FFN down K=1024", lambda: ops.rows_gemm(hid, w1, b1, samp), 2.0 * R * 256 * 1024, 4.0 * R * (1024 + 256)),
    ("rows_ln_pos (residual + LayerNorm)", lambda: _ffi.call("demf_rows_ln_pos_f32", R, C, samp.data_ptr(), x1.data_ptr(), g.data_ptr(),
                                                           be.data_ptr(), 1e-5, None, xn.data_ptr(), None, st), 0.0, 4.0 * R * 3 * 256),
]
out, tot = [], 0.0
for name, fn, flop, byts in rows:
    us = t(fn); tot += us
    out.append(dict(kernel=name, us=round(us, 1), gflop=round(flop * 1e-9, 1), mb=round(byts * 1e-6, 1),
                    tflops=round(flop / us * 1e-6, 1), mfma_frac=round(flop / us * 1e-6 / (416.7 if planes == 3 else 2500.0), 3),
                    gbps=round(byts / us * 1e-3, 0), hbm_frac=round(byts / us * 1e-3 / 8000.0, 3)))
print(json.dumps(dict(mode=mode, rows=R, layer_us=round(tot, 1), six_layers_ms=round(6 * tot * 1e-3, 2), launches=out), indent=1))
