import os
import numpy as np


def test_yuv_reader_roundtrip(tmp_path):
    # reference: TVideoIOYuv::read / readPlane (TVideoIOYuv.cpp:680, :247): planar, luma first
    from hmme import yuv
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, size=(36, 64)).astype(np.uint8) for _ in range(3)]
    p = str(tmp_path / "t.yuv")
    yuv.write_luma_420(p, frames)
    assert os.path.getsize(p) == 3 * yuv.frame_bytes(64, 36)
    for i, f in enumerate(frames):
        assert np.array_equal(yuv.read_luma(p, 64, 36, i), f)
    hi = (rng.integers(0, 1024, size=(36, 64))).astype("<u2")
    with open(p, "wb") as fh:
        fh.write(hi.tobytes()); fh.write(np.zeros(36 * 64 // 2, "<u2").tobytes())
    assert np.array_equal(yuv.read_luma(p, 64, 36, 0, file_bit_depth=10), hi)
