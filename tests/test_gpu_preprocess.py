"""Next row N1: the HIP image-preprocessing path (vcla_image_preprocess through visualcla.preprocess.GpuClipImageProcessor)
against the integer-exact oracle and against transformers' CLIPImageProcessor itself.  fp32 output must be bit-identical
(same integer resize, same float op order); bf16 output is the round-to-nearest-even of it."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as P

pytestmark = pytest.mark.gpu

SIZES = [(300, 400), (1000, 640), (224, 224), (150, 97), (512, 768), (225, 224), (37, 1201), (1080, 1920), (224, 640)]


def _img(hw, seed=0):
    rng = np.random.default_rng(seed + hw[0] * 7 + hw[1])
    return (rng.random((*hw, 3)) * 255).astype(np.uint8)


@pytest.mark.parametrize("hw", SIZES)
@pytest.mark.parametrize("size", [224, 336, 32])
def test_fp32_is_bit_exact_vs_oracle(hw, size):
    from visualcla.preprocess import GpuClipImageProcessor
    proc = GpuClipImageProcessor(size=size)
    img = _img(hw)
    got = proc(img).pixel_values
    assert got.shape == (1, 3, size, size) and got.dtype == torch.float32 and got.is_cuda
    want = P.clip_preprocess(img, size)
    assert np.array_equal(got[0].cpu().numpy(), want)


@pytest.mark.parametrize("hw", SIZES[:5])
def test_matches_hf_clip_image_processor_on_pil_input(hw):
    from PIL import Image
    from transformers import CLIPImageProcessor
    from visualcla.preprocess import GpuClipImageProcessor
    hf = CLIPImageProcessor()
    proc = GpuClipImageProcessor.from_hf(hf)
    pil = Image.fromarray(_img(hw, 3))
    ref = hf(pil, return_tensors="pt").pixel_values
    got = proc(pil).pixel_values.cpu()
    assert torch.equal(got, ref)


def test_bf16_output_batch_and_extremes():
    from visualcla.preprocess import GpuClipImageProcessor
    proc = GpuClipImageProcessor(size=224, dtype=torch.bfloat16)
    imgs = [_img((300, 400), 1), np.zeros((256, 256, 3), np.uint8), np.full((500, 333, 3), 255, np.uint8)]
    # a checkerboard drives the bicubic overshoot into the clip-to-[0,255] branch
    cb = ((np.indices((448, 448)).sum(0) // 3) % 2 * 255).astype(np.uint8)
    imgs.append(np.repeat(cb[:, :, None], 3, 2))
    got = proc(imgs).pixel_values
    assert got.shape == (4, 3, 224, 224) and got.dtype == torch.bfloat16
    for i, im in enumerate(imgs):
        want = torch.from_numpy(P.clip_preprocess(im, 224)).to(torch.bfloat16)
        assert torch.equal(got[i].cpu(), want)


def test_rejects_non_rgb_u8():
    from visualcla.preprocess import GpuClipImageProcessor
    proc = GpuClipImageProcessor(size=224)
    with pytest.raises(ValueError):
        proc(np.zeros((10, 10), np.uint8))
    with pytest.raises(ValueError):
        proc(np.zeros((10, 10, 3), np.float32))


def test_batched_entry_is_bit_exact_and_keeps_input_order():
    """vcla_image_preprocess_batch: N same-sized images in one launch pair; a mixed list is grouped by size and scattered back"""
    from visualcla.preprocess import GpuClipImageProcessor
    proc = GpuClipImageProcessor(size=224)
    same = np.stack([_img((480, 640), s) for s in range(9)])
    got = proc.preprocess_batch(torch.from_numpy(same).pin_memory())
    assert got.shape == (9, 3, 224, 224)
    for i in range(9):
        assert np.array_equal(got[i].cpu().numpy(), P.clip_preprocess(same[i], 224))
    mixed = [_img((480, 640), 1), _img((300, 400), 2), _img((480, 640), 3), _img((224, 224), 4), _img((300, 400), 5)]
    got = proc(mixed).pixel_values
    for i, im in enumerate(mixed):
        assert np.array_equal(got[i].cpu().numpy(), P.clip_preprocess(im, 224)), i
