"""Device image pipeline against the reference's formulas (data/src/utils.py:7-25 behind torchvision's ToTensor,
RandomHorizontalFlip of data/src/cifar10.py:17-20), restated with tensor ops on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_preprocess(u8_bhwc, flip):
    x = u8_bhwc.permute(0, 3, 1, 2).to(torch.float32).div(255)        # ToTensor
    if flip is not None:
        x = torch.where(flip.view(-1, 1, 1, 1).bool(), x.flip(-1), x)  # RandomHorizontalFlip (applied before ToTensor)
    x = x * 255
    return x / 256 - 0.5


@pytest.mark.parametrize("shape", [(64, 32, 32, 3), (5, 64, 64, 3), (3, 7, 9, 1)])
def test_preprocess_bit_identical(shape):
    from nf_distillation_amd.data import preprocess_images
    g = torch.Generator().manual_seed(3)
    u = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
    flip = torch.rand(shape[0], generator=g) < 0.5
    for f in (None, flip):
        got = preprocess_images(u.cuda(), None if f is None else f.cuda()).cpu()
        assert torch.equal(got, _ref_preprocess(u, f))
    # every byte value
    allv = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1)
    assert torch.equal(preprocess_images(allv.cuda()).cpu(), _ref_preprocess(allv, None))


def test_postprocess_matches_reference_and_round_trips():
    from nf_distillation_amd.data import postprocess_images, preprocess_images
    g = torch.Generator().manual_seed(4)
    x = torch.randn(6, 3, 16, 16, generator=g) * 0.4
    ref = torch.clamp(x, -0.5, 0.5) + 0.5
    ref = torch.clamp(ref * 255, 0, 255).byte().permute(0, 2, 3, 1)
    assert torch.equal(postprocess_images(x.cuda()).cpu(), ref)
    u = torch.randint(0, 256, (4, 8, 8, 3), dtype=torch.uint8, generator=g)
    # pre- then post-processing: x = u/256 - 0.5 -> (x + 0.5) * 255 = u * 255/256 -> truncation loses at most one level
    back = postprocess_images(preprocess_images(u.cuda())).cpu().int()
    assert int((u.int() - back).abs().max()) <= 1
