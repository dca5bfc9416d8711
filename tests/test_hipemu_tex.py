"""Host-logic check of the product's texture kernels through the tests/hipemu shim (no GPU)."""
import numpy as np


def test_hipemu_texture_matches_oracle_bytes(oracle, hipemu_lib):
    import synth, uvol
    cd = uvol.Codec(lib_path=hipemu_lib)
    for size, n, seed in [(64, 2, 1), (52, 3, 5)]:
        tex = synth.texture_sequence(n, size=size, seed=seed)
        assert cd.encode_texture_segment(tex) == oracle.ktx2_encode(tex)
    cd.close()
