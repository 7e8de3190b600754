"""PNG frames decoded on the GPU (abub_png_decode_dev: gather + wave-per-frame inflate + unfilter) against zlib / PIL:
every deflate block type, every filter type, palette images, IDAT chunkings, corrupt and truncated streams (status must
say so and nothing may be written out of bounds), full-size frames, one frame of the reference's own sample data."""
import io
import os
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))


def chunk(typ, data):
    return struct.pack(">I", len(data)) + typ + data + struct.pack(">I", zlib.crc32(typ + data) & 0xFFFFFFFF)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def filter_rows(img, filters):
    """PNG filtering of an 8-bit single-channel image; filters[y] in 0..4 -> bytes of all scanlines"""
    H, W = img.shape
    out = bytearray()
    prev = np.zeros(W, dtype=np.int32)
    for y in range(H):
        cur = img[y].astype(np.int32)
        ft = int(filters[y])
        left = np.concatenate(([0], cur[:-1]))
        upleft = np.concatenate(([0], prev[:-1]))
        if ft == 0:
            pred = np.zeros(W, dtype=np.int32)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            pa, pb, pc = np.abs(prev - upleft), np.abs(left - upleft), np.abs(left + prev - 2 * upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        out.append(ft)
        out += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def make_png(img, filters, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, idat=1 << 16, palette=None, memlevel=8):
    H, W = img.shape
    raw = filter_rows(img, filters)
    co = zlib.compressobj(level, zlib.DEFLATED, 15, memlevel, strategy)
    z = co.compress(raw) + co.flush()
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 3 if palette is not None else 0, 0, 0, 0))
    if palette is not None:
        png += chunk(b"PLTE", bytes(palette))
    if idat == 0:  # an empty chunk in front, then one-byte chunks, then the rest
        png += chunk(b"IDAT", b"")
        for k in range(min(5, len(z))):
            png += chunk(b"IDAT", z[k:k + 1])
        png += chunk(b"IDAT", z[5:])
    else:
        for k in range(0, len(z), idat):
            png += chunk(b"IDAT", z[k:k + idat])
    return png + chunk(b"IEND", b""), z


def expect_grey(img, palette):
    if palette is None:
        return img
    pal = np.frombuffer(bytes(palette) + bytes(768 - len(palette)), dtype=np.uint8).reshape(256, 3).astype(np.int64)
    lut = ((pal[:, 0] * 9797 + pal[:, 1] * 19234 + pal[:, 2] * 3737 + 16384) >> 15).astype(np.uint8)
    lut[len(palette) // 3:] = 0
    return lut[img]


def images(rs, W, H):
    yy, xx = np.mgrid[0:H, 0:W]
    noise = rs.randint(-3, 4, (H, W))
    return {
        "smooth+noise": np.clip(60 + (xx * 60) // max(W, 1) + (yy * 30) // max(H, 1) + noise, 0, 255).astype(np.uint8),
        "random": rs.randint(0, 256, (H, W)).astype(np.uint8),
        "constant": np.full((H, W), 7, dtype=np.uint8),
        "ramp": ((xx + yy) & 255).astype(np.uint8),
        "blocks": ((xx // 7 + yy // 3) % 5 * 50).astype(np.uint8),
    }


def run(files, W, H):
    from autobub3hs_amd import hip
    out, st = hip.png_decode(files, W, H)
    torch.cuda.synchronize()
    return out.cpu().numpy(), st


@pytest.mark.parametrize("W,H", [(4, 1), (8, 3), (64, 2), (100, 37), (256, 64), (320, 160), (516, 33), (2048, 9)])
def test_every_filter_and_block_type(W, H):
    rs = np.random.RandomState(W * 1000 + H)
    files, want = [], []
    strategies = [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY),
                  (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (6, zlib.Z_FILTERED)]
    for name, img in images(rs, W, H).items():
        for ft in (0, 1, 2, 3, 4, "mixed"):
            filters = rs.randint(0, 5, H) if ft == "mixed" else np.full(H, ft)
            level, strat = strategies[len(files) % len(strategies)]
            png, _ = make_png(img, filters, level, strat, idat=[1 << 16, 8192, 100, 0][len(files) % 4], memlevel=[8, 1, 9][len(files) % 3])
            files.append(png)
            want.append(img)
    got, st = run(files, W, H)
    assert (st == 0).all(), st
    for k in range(len(files)):
        assert np.array_equal(got[k], want[k]), k


def test_palette_images():
    rs = np.random.RandomState(5)
    W, H = 320, 100
    files, want = [], []
    for npal in (256, 200, 2):
        pal = rs.randint(0, 256, 3 * npal).astype(np.uint8).tobytes()
        img = rs.randint(0, npal, (H, W)).astype(np.uint8)
        img[:, :50] = np.clip(60 + rs.randint(-2, 3, (H, 50)), 0, npal - 1)
        png, _ = make_png(img, rs.randint(0, 5, H), 6, palette=pal)
        files.append(png)
        want.append(expect_grey(img, pal))
    # grey palette (what the cameras of the reference's sample frames write): identity
    pal = bytes(v for i in range(256) for v in (i, i, i))
    img = rs.randint(0, 256, (H, W)).astype(np.uint8)
    files.append(make_png(img, np.zeros(H), 1, palette=pal)[0])
    want.append(img)
    got, st = run(files, W, H)
    assert (st == 0).all(), st
    for k in range(len(files)):
        assert np.array_equal(got[k], want[k]), k


def test_pil_encoder_all_levels():
    from PIL import Image
    from autobub3hs_amd import synth
    W, H = 1280, 1024
    spec = synth.random_spec(W, H, 12, 3, 0)
    fr = np.asarray(synth.render_event(W, H, spec, 3, 0))
    files, want = [], []
    for k, lvl in enumerate((0, 1, 3, 6, 9)):
        b = io.BytesIO()
        Image.fromarray(fr[k]).save(b, format="PNG", compress_level=lvl)
        files.append(b.getvalue())
        want.append(fr[k])
    b = io.BytesIO()
    Image.fromarray(fr[5]).convert("P").save(b, format="PNG")  # PIL's own palette image
    files.append(b.getvalue())
    pimg = Image.open(io.BytesIO(files[-1]))
    want.append(expect_grey(np.asarray(pimg), bytes(pimg.getpalette())))
    got, st = run(files, W, H)
    assert (st == 0).all(), st
    for k in range(len(files)):
        assert np.array_equal(got[k], want[k]), k


def test_unsupported_files_are_refused_not_decoded():
    from PIL import Image
    W, H = 64, 16
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (H, W)).astype(np.uint8)
    ok, _ = make_png(img, np.zeros(H))
    b16 = io.BytesIO()
    Image.fromarray((img.astype(np.uint16) << 8)).save(b16, format="PNG")  # 16-bit grey
    rgb = io.BytesIO()
    Image.fromarray(np.stack([img] * 3, -1)).save(rgb, format="PNG")
    other = make_png(rs.randint(0, 256, (H, W + 4)).astype(np.uint8), np.zeros(H))[0]  # another geometry
    got, st = run([ok, b16.getvalue(), rgb.getvalue(), other, b"not a png at all", ok], W, H)
    assert list(st) == [0, -1, -1, -1, -1, 0]
    assert np.array_equal(got[0], img) and np.array_equal(got[5], img)
    assert not got[1:5].any()


def test_corrupt_and_truncated_streams_agree_with_zlib():
    """Random damage to the compressed data: where zlib (and the scanline size) accepts the stream, the GPU must produce the
    same image; where zlib refuses, the status must be non-zero.  Neighbouring frames of the batch stay intact."""
    rs = np.random.RandomState(11)
    W, H = 128, 40
    base = images(rs, W, H)["smooth+noise"]
    good, _ = make_png(base, rs.randint(0, 5, H), 6)
    files, verdict = [], []
    for trial in range(120):
        img = base if trial % 3 else rs.randint(0, 256, (H, W)).astype(np.uint8)
        filters = rs.randint(0, 5, H)
        level, strat = [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)][trial % 4]
        raw = filter_rows(img, filters)
        co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strat)
        z = bytearray(co.compress(raw) + co.flush())
        kind = trial % 5
        if kind == 0:
            z = z[:rs.randint(0, len(z))]                      # truncated
        elif kind == 1:
            z[rs.randint(0, len(z))] ^= 1 << rs.randint(0, 8)  # one bit
        elif kind == 2:
            for _ in range(8):
                z[rs.randint(0, len(z))] = rs.randint(0, 256)
        elif kind == 3:
            z[-4:] = bytes(rs.randint(0, 256, 4).tolist())     # checksum
        else:
            z[2 + rs.randint(0, min(40, len(z) - 2))] ^= 0xFF  # the first block's header / tables
        z = bytes(z)
        try:
            d = zlib.decompressobj()
            res = d.decompress(z)
            okz = d.eof and len(res) == H * (W + 1) and all(res[y * (W + 1)] <= 4 for y in range(H))
        except zlib.error:
            okz, res = False, b""
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b"")
        files += [good, png]
        verdict.append((okz, res))
    got, st = run(files, W, H)
    from PIL import Image
    for t, (okz, res) in enumerate(verdict):
        assert st[2 * t] == 0 and np.array_equal(got[2 * t], base), t  # the intact neighbour
        if okz:
            assert st[2 * t + 1] == 0, (t, st[2 * t + 1])
            png = files[2 * t + 1]
            want = np.asarray(Image.open(io.BytesIO(png)))
            assert np.array_equal(got[2 * t + 1], want), t
        else:
            assert st[2 * t + 1] != 0, t


def test_reference_sample_frame():
    """One frame of the reference's sample data as its camera software wrote it (cam_masks/40l-19, 1680 x 1050, 8-bit grey
    palette, filter None, 15 IDAT chunks): committed as data under tests/golden."""
    from PIL import Image
    path = os.path.join(HERE, "golden", "sample_40l19_cam1_image30.png")
    data = open(path, "rb").read()
    im = Image.open(io.BytesIO(data))
    want = expect_grey(np.asarray(im), bytes(im.getpalette()))
    got, st = run([data, data], 1680, 1050)
    assert (st == 0).all(), st
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


def test_a_batch_of_full_size_frames_and_host_decoder_agreement():
    """64 full-size frames, as the ingestion path sends them; the host decoder of the same build gives the same pixels."""
    from PIL import Image
    from autobub3hs_amd import synth
    W, H = 1280, 1024
    spec = synth.random_spec(W, H, 16, 9, 1)
    fr = np.asarray(synth.render_event(W, H, spec, 9, 1))
    files = []
    for k in range(64):
        b = io.BytesIO()
        Image.fromarray(fr[k % 16]).save(b, format="PNG", compress_level=1 if k % 2 else 6)
        files.append(b.getvalue())
    got, st = run(files, W, H)
    assert (st == 0).all(), st
    for k in range(64):
        assert np.array_equal(got[k], fr[k % 16]), k
