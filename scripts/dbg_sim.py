import sys, torch
sys.path.insert(0, "/root/repo")
from pope_amd.matcher import dense_match
g = torch.Generator().manual_seed(1)
h, w = 9, 11
f = (torch.randn(1, h * w, 64, generator=g) * 4).cuda()
a = dense_match(f, f, (h, w), (h, w), (h * 14, w * 14), precision="f32")
b = dense_match(f, f, (h, w), (h, w), (h * 14, w * 14), precision="f16x3")
print(len(a["i_ids"]), len(b["i_ids"]))
ca, cb = a["conf_matrix"][0], b["conf_matrix"][0]
d = (ca - cb).abs()
print("max diff", float(d.max()), "at", divmod(int(d.argmax()), 99))
print(ca[:3, :6]); print(cb[:3, :6]); print(cb[96:99, 93:99]); print(ca[96:99, 93:99])
