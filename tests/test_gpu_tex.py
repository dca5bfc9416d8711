"""Parity tests proper for the texture path: HIP on a real MI355X, through the C-ABI, vs the CPU oracle."""
import os
import numpy as np
import pytest
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,n,seed", [(64, 2, 1), (52, 3, 5), (256, 5, 2), (512, 2, 7)])
def test_gpu_texture_bit_exact(oracle, gpu_codec, size, n, seed):
    import synth
    tex = synth.texture_sequence(n, size=size, seed=seed)
    assert gpu_codec.encode_texture_segment(tex) == oracle.ktx2_encode(tex)


def test_gpu_flat_single_layer(oracle, gpu_codec):
    flat = [np.full((16, 16, 4), 255, np.uint8)]
    flat[0][..., :3] = (12, 200, 77)
    assert gpu_codec.encode_texture_segment(flat) == oracle.ktx2_encode(flat)


def test_gpu_reference_texture_reencode(oracle, gpu_codec):
    """Real captured content (decoded reference segment, 1024^2 x 5): byte-exact vs oracle, fixture-like bpp."""
    ref = open(os.path.join(GOLDEN, "00000.ktx2"), "rb").read()
    d = oracle.ktx2_decode(ref)
    src = [im[::-1].copy() for im in d.images]
    k = gpu_codec.encode_texture_segment(src)
    assert k == oracle.ktx2_encode(src)
    assert 0.8 * len(ref) < len(k) < 1.1 * len(ref)


def test_gpu_full_size_segment(oracle, gpu_codec):
    """BASELINE headline shape: 2048^2 x KTX2_BATCH_SIZE=5 ETC1S video segment -> decodes with the pinned decoder,
    I/P slice pattern, skip blocks on static content, PSNR floor; and byte-exact vs the oracle."""
    import synth
    tex = synth.texture_sequence(5, size=2048, seed=0)
    k = gpu_codec.encode_texture_segment(tex)
    d = oracle.ktx2_decode(k)
    assert (d.width, d.height, d.layers) == (2048, 2048, 5) and d.slice_flags == [0, 2, 2, 2, 2]
    assert all(8 * l - b <= 7 for l, b in zip(d.slice_len, d.slice_bits_used))
    nb = 512 * 512
    assert all(0.5 * nb < s < 0.9 * nb for s in d.slice_skip[1:])
    assert min(oracle.psnr(d.images[l], tex[l][::-1]) for l in range(5)) > 30.0
    assert k == oracle.ktx2_encode(tex)


def test_gpu_batched_segments_equal_single_calls(oracle, gpu_codec):
    """uvol_encode_texture_segments (one launch per stage for the whole batch) == per-segment calls == oracle."""
    import synth
    segs = [synth.texture_sequence(3, size=128, seed=s) for s in (1, 2, 3, 4)]
    segs.append([np.full((128, 128, 4), v, np.uint8) for v in (10, 10, 200)])
    res = gpu_codec.encode_texture_segments(segs)
    for seg, r in zip(segs, res):
        assert r == oracle.ktx2_encode(seg)
