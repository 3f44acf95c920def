"""GPU parity of the evaluation image transform (xmh_image_preprocess_u8) against the oracle and the Pillow-made goldens:
uint8 stage bit-exact, float stage bit-exact (same IEEE float32 operations in the same order)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from xmh.dataset.preprocess import GpuEvalTransform
    return GpuEvalTransform(224)


def test_goldens_bit_exact(tf):
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    n = 0
    while "img%d" % n in g:
        img = torch.from_numpy(g["img%d" % n])
        assert np.array_equal(tf.resize_u8(img).cpu().numpy()[0], g["resized%d" % n]), n
        n += 1
    assert np.array_equal(tf(torch.from_numpy(g["img0"])).cpu().numpy()[0], g["tensor0"])


@pytest.mark.parametrize("B,H,W", [(3, 375, 500), (2, 224, 224), (5, 100, 80), (2, 640, 427), (1, 224, 300), (2, 33, 517), (1, 1200, 1600),
                                   (4, 1, 1), (2, 2, 3)])
def test_batches_match_oracle(tf, B, H, W):
    from oracle import preprocess as O
    gen = torch.Generator().manual_seed(H * 7 + W)
    imgs = torch.randint(0, 256, (B, H, W, 3), generator=gen, dtype=torch.uint8)
    imgs[0, : H // 2] = 255                                   # a saturated half: overshoot must clip, not wrap
    imgs[-1, :, : W // 2] = 0
    got_u8 = tf.resize_u8(imgs).cpu().numpy()
    got_f = tf(imgs.cuda()).cpu().numpy()
    for b in range(B):
        want = O.resize_bicubic_u8(imgs[b].numpy(), 224, 224)
        assert np.array_equal(got_u8[b], want), b
        assert np.array_equal(got_f[b], O.to_tensor_normalize(want)), b


def test_rejects_wrong_inputs(tf):
    with pytest.raises(TypeError):
        tf(torch.zeros(1, 8, 8, 3))
    with pytest.raises(ValueError):
        tf(torch.zeros(1, 3, 8, 8, dtype=torch.uint8))


def test_feeds_the_encoder(tf):
    """raw uint8 photos -> GPU transform -> CLIP image tower: same features as the host-transformed float batch."""
    from oracle import preprocess as O
    from xmh.models.dcmht import DCMHT
    from xmh.utils.config import Config
    model = DCMHT.from_config(Config({"clip_path": "synthetic:3:layers=1"}), output_dim=16).cuda().eval()
    gen = torch.Generator().manual_seed(9)
    imgs = torch.randint(0, 256, (3, 120, 160, 3), generator=gen, dtype=torch.uint8)
    host = torch.from_numpy(np.stack([O.eval_transform(i.numpy()) for i in imgs]))
    assert torch.equal(model.encode_image(tf(imgs)), model.encode_image(host.cuda()))
