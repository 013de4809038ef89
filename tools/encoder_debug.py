import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import fixtures
from oracle.model import OracleImageStream
from demf_amd.modules import ImageStream
gold = np.load("/root/repo/tests/golden/ref_encoder.npz")
cfg = fixtures.TINY_IMAGE_STREAM
ref = OracleImageStream(**cfg); fixtures.seed_weights(ref, 4)
m = ImageStream(**cfg); fixtures.seed_weights(m, 4); m.cuda()
img, metas = fixtures.make_images(4)
neck = [torch.from_numpy(gold[f"neck{i}"]) for i in range(4)]
enc = m.img_encoder
spatial = [tuple(f.shape[-2:]) for f in neck]
st = enc._static(metas, spatial, torch.device("cuda"))
# oracle intermediates
oe = ref.img_encoder
B = 2; ih, iw = metas[0]["batch_input_shape"]
img_masks = torch.ones((B, ih, iw))
for i in range(B):
    h, w, _ = metas[i]["img_shape"]; img_masks[i, :h, :w] = 0
import torch.nn.functional as F
masks = [F.interpolate(img_masks[None], size=f.shape[-2:]).to(torch.bool).squeeze(0) for f in neck]
print("mask equal:", torch.equal(torch.cat([x.flatten(1) for x in masks], 1), st["mask_flatten"].cpu()))
pos = [oe.positional_encoding(x) for x in masks]
pos_f = torch.cat([p.flatten(2).transpose(1, 2) for p in pos], 1)
print("pos maxdiff:", float((torch.cat(st["pos"], 1).cpu() - pos_f).abs().max()))
vr = torch.stack([oe.valid_ratio(x) for x in masks], 1)
print("vr diff:", float((st["valid_ratios"].cpu() - vr).abs().max()))
rp = oe.reference_points(spatial, vr)
print("ref pts diff:", float((st["reference_points"].cpu() - rp).abs().max()))
# layer by layer
tokens_o = torch.cat([f.flatten(2).transpose(1, 2) for f in neck], 1)
pos_o = torch.cat([p.flatten(2).transpose(1, 2) + oe.level_embeds[l].view(1, 1, -1) for l, p in enumerate(pos)], 1)
ss = torch.as_tensor(spatial, dtype=torch.long); lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
tokens_p = tokens_o.cuda(); pos_p = torch.cat([p + enc.level_embeds[l].view(1, 1, -1) for l, p in enumerate(st["pos"])], 1)
q = tokens_o.permute(1, 0, 2)
with torch.no_grad():
    for li, (lo, lp) in enumerate(zip(oe.encoder.layers, enc.encoder.layers)):
        # attention only
        a_o = lo.attentions[0](q, q, q, None, query_pos=pos_o.permute(1, 0, 2), key_padding_mask=torch.cat([x.flatten(1) for x in masks], 1),
                               spatial_shapes=ss, reference_points=rp, level_start_index=lsi)
        a_p = lp.attentions[0](tokens_p, None, None, None, query_pos=pos_p, key_padding_mask=st["mask_flatten"],
                               reference_points=st["reference_points"], spatial_shapes=st["spatial_shapes"], level_start_index=st["level_start_index"])
        d = (a_p.cpu() - a_o.permute(1, 0, 2)).abs()
        print(f"layer {li} attention maxdiff {float(d.max()):.3e} at", np.unravel_index(int(d.argmax()), d.shape))
        q = lo(q, query_pos=pos_o.permute(1, 0, 2), query_key_padding_mask=torch.cat([x.flatten(1) for x in masks], 1), spatial_shapes=ss, reference_points=rp, level_start_index=lsi)
        tokens_p = lp(tokens_p, pos_p, st["mask_flatten"], reference_points=st["reference_points"], spatial_shapes=st["spatial_shapes"], level_start_index=st["level_start_index"])
        d = (tokens_p.cpu() - q.permute(1, 0, 2)).abs()
        print(f"layer {li} output maxdiff {float(d.max()):.3e}")
