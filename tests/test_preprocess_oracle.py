"""Pin the N1 oracle (oracle/preprocess_oracle.py) against the real third-party code it restates: Pillow's bicubic resize
(bit-exact, uint8) and transformers' CLIPImageProcessor (the call the reference makes, modeling_utils.py:150-152)."""
import numpy as np
import pytest

from oracle import preprocess_oracle as P

SIZES = [(300, 400), (1000, 640), (224, 224), (150, 97), (512, 768), (225, 224), (37, 1201)]


@pytest.mark.parametrize("hw", SIZES)
def test_resize_is_bit_exact_vs_pillow(hw):
    from PIL import Image
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    img = (rng.random((*hw, 3)) * 255).astype(np.uint8)
    oh, ow = P.resized_shape(*hw, 224)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
    assert np.array_equal(P.resize_bicubic_u8(img, oh, ow), ref)


@pytest.mark.parametrize("hw", SIZES[:5])
@pytest.mark.parametrize("size", [224, 336])
def test_pipeline_matches_clip_image_processor(hw, size):
    from PIL import Image
    from transformers import CLIPImageProcessor
    rng = np.random.default_rng(hw[0] + hw[1] + size)
    img = (rng.random((*hw, 3)) * 255).astype(np.uint8)
    proc = CLIPImageProcessor(size={"shortest_edge": size}, crop_size={"height": size, "width": size})
    ref = proc(Image.fromarray(img), return_tensors="np").pixel_values[0]
    got = P.clip_preprocess(img, size)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("hw", [(90, 120), (480, 640), (640, 480), (224, 224), (224, 500), (37, 1000), (1080, 1920)])
def test_product_tables_match_oracle(hw):
    """the host-side table builder of the HIP path (visualcla/preprocess.py) against the oracle's"""
    from visualcla.preprocess import plan_tables
    H, W = hw
    S = 224
    oh, ow = P.resized_shape(H, W, S)
    top, left = (oh - S) // 2, (ow - S) // 2
    (hl, hn, hk), (vl, vn, vk) = plan_tables(H, W, S)
    for got, want in zip((hl, hn, hk), P.coeff_tables(W, ow, left, S)):
        assert np.array_equal(got, want)
    for got, want in zip((vl, vn, vk), P.coeff_tables(H, oh, top, S)):
        assert np.array_equal(got, want)
