"""The image tower and the text tower of one evaluation forward are independent (reference runners/base.py:250-257 runs them one
after the other): here the text tower is enqueued on a second HIP stream and runs under the image tower.  Their kernels fill
each other's tails -- a ViT-B/32 GEMM of a 100-image batch leaves CUs idle in its last round of tiles and while it stores C, the
text tower's grids are smaller than the chip -- so the pair costs less than the sum.  Same kernels, same bits.

Allocator safety: every forward takes its workspace from torch's stream-aware caching allocator on the stream it runs on; the
side stream starts after everything already enqueued on the caller's stream (inputs are ready) and the caller's stream waits
for it before the results are used; blocks freed later are reused on their own stream behind those waits."""
from __future__ import annotations

import os

import torch

_side = {}


def run_both(fn_image, fn_text):
    """-> (fn_image(), fn_text()); XMH_TOWER_STREAMS=0 runs them back to back on the current stream."""
    if os.environ.get("XMH_TOWER_STREAMS", "1") == "0" or not torch.cuda.is_available():
        return fn_image(), fn_text()
    cur = torch.cuda.current_stream()
    side = _side.get(cur.device)
    if side is None:
        side = _side[cur.device] = torch.cuda.Stream(device=cur.device)
    side.wait_stream(cur)                # the side stream starts behind what is on the caller's stream NOW: the inputs, not this image forward
    img = fn_image()                     # enqueued first: the packed text tower reads its row count back (one small device-to-host
    with torch.cuda.stream(side):        # copy behind the wait above), and while the host waits for it the GPU has this to run
        txt = fn_text()
    cur.wait_stream(side)
    _record(txt, cur)
    return img, txt


def _record(out, stream):
    """the text tower's outputs were allocated on the side stream and are consumed on the caller's: tell the caching allocator, so
    that a block freed while the caller's stream still reads it is not handed to the side stream again (the next run_both may
    run under ANOTHER current stream, whose wait_stream does not fence this one)"""
    if isinstance(out, torch.Tensor):
        if out.is_cuda:
            out.record_stream(stream)
    elif isinstance(out, (tuple, list)):
        for o in out:
            _record(o, stream)
    elif isinstance(out, dict):
        for o in out.values():
            _record(o, stream)
