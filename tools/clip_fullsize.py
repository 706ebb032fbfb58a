"""Full-size (ViT-H-14) run of the two OpenCLIP condition encoders on libvcx: wall time per call and rel-L2 against the fp32
CPU oracle (oracle/clip_oracle.py) on the same synthetic weights.  python tools/clip_fullsize.py [--no-oracle]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.weights import synth_input, synth_state_dict
from viewcrafter_amd.lvdm.modules.encoders import condition as cond

def load_synth(m):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth_state_dict(shapes)
    m.load_state_dict(sd, strict=True)
    return sd

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): y = fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n, y

def rel_l2(a, b): return float((a.float().cpu() - b.float()).norm() / b.float().norm())

res = {}
cfg = cond.CLIP_CONFIGS["ViT-H-14"]
txt = cond.FrozenOpenCLIPEmbedder(layer="penultimate").eval(); sd_t = load_synth(txt); txt = txt.cuda()
tokens = cond.tokenize([""])
tokens[0, 1:12] = torch.arange(1000, 1011); tokens[0, 12] = 49407
res["text_s"], yt = timed(lambda: txt.encode_with_transformer(tokens))
img = cond.FrozenOpenCLIPImageEmbedderV2().eval(); sd_i = load_synth(img); img = img.cuda()
x = torch.tanh(synth_input("clip_full", (1, 3, 576, 1024)))
res["image_s"], yi = timed(lambda: img(x.cuda()))
res["text_shape"], res["image_shape"] = list(yt.shape), list(yi.shape)
if "--no-oracle" not in sys.argv:
    from oracle import clip_oracle as C
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    t, v = cfg["text"], cfg["vision"]
    with torch.no_grad():
        res["text_rel_l2"] = rel_l2(yt, C.clip_text_forward(sd_t, tokens, t["heads"], t["layers"], 1))
        res["image_rel_l2"] = rel_l2(yi, C.clip_image_forward(sd_i, x, v["width"] // v["head_width"], v["layers"], v["patch_size"]))
print(json.dumps(res))
