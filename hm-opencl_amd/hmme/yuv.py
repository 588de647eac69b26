"""Frame feeder: planar YUV reader for the luma plane (the step before the ME path; reference
TLibVideoIO/TVideoIOYuv.cpp:680 `read`, :247 `readPlane`: 8-bit files hold one byte per sample,
higher bit depths two bytes little-endian; 4:2:0 chroma follows luma and is skipped here because
integer ME never reads it).  Padding to HM's 80-sample margin is done on the device by
hmme_plane_upload_* (TComPicYuv::extendPicBorder, TComPicYuv.cpp:214-262)."""
import numpy as np


def frame_bytes(width, height, file_bit_depth=8, chroma="420"):
    bps = 1 if file_bit_depth <= 8 else 2
    luma = width * height * bps
    c = {"400": 0, "420": luma // 2, "422": luma, "444": 2 * luma}[chroma]
    return luma + c


def read_luma(path, width, height, frame, file_bit_depth=8, chroma="420"):
    """-> (height, width) uint8 (8-bit files) or uint16 (16-bit little-endian files) luma of picture `frame`"""
    bps = 1 if file_bit_depth <= 8 else 2
    off = frame * frame_bytes(width, height, file_bit_depth, chroma)
    dt = np.uint8 if bps == 1 else np.dtype("<u2")
    a = np.fromfile(path, dtype=dt, count=width * height, offset=off)
    if a.size != width * height:
        raise ValueError(f"{path}: picture {frame} is beyond the end of the file")
    return a.reshape(height, width)


def write_luma_420(path, frames):
    """write 8-bit 4:2:0 pictures (luma given, chroma = 128): test helper"""
    with open(path, "wb") as f:
        for y in frames:
            y = np.ascontiguousarray(y, dtype=np.uint8)
            f.write(y.tobytes())
            f.write(np.full(y.size // 2, 128, np.uint8).tobytes())
