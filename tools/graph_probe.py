import sys, time, torch
sys.path[:0] = ["/root/repo", "/root/repo/clip-based-cross-modal-hash_amd"]
from xmh import ops, retrieval as R
from xmh.models.dcmht import DCMHT
from xmh.utils.config import Config
from xmh.models import weights as W
model = DCMHT.from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64).cuda().eval()
B = 100
image = W.synth_images(5, B).cuda()
ids, _ = W.synth_text(5, B); ids = ids.cuda()
for name, fn in (("image", lambda: model.encode_image(image)), ("text", lambda: model.encode_text(ids))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    t_enq = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 10
    # graph capture
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            out = fn()
        torch.cuda.synchronize()
        ref = fn()
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(out, ref)
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 10
        print(name, "enqueue %.3f ms  total %.3f ms  graph %.3f ms  same=%s" % (t_enq * 1e3, t_all * 1e3, t_graph * 1e3, same))
    except Exception as e:
        print(name, "enqueue %.3f ms  total %.3f ms  graph capture failed: %r" % (t_enq * 1e3, t_all * 1e3, e))
